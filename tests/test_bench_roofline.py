"""bench.py's roofline arithmetic on hand-made statistics (no GPU): every kernel family gives a fraction that follows from the
formula in its note, stays at or below 1 for the measured figures of the tracked runs, and the counter file is ignored when it
was taken from other kernel sources."""
import json
import os
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _scene(program=1, triangles=36, normal_mode=1):
    return types.SimpleNamespace(program=program, num_triangles=triangles, normal_mode=normal_mode)


def _stats(**kw):
    st = {k: 0 for k in ("rays", "rays_traced", "samples", "lane_slots", "bvh_node_fetches", "bvh_triangle_fetches", "grid_triangle_tests",
                         "shaded_hits", "loop_triangle_visits", "loop_full_tests", "lane_walk_records")}
    st["dispatches"] = 1
    st.update(kw)
    return st


def test_grid_kernel_fraction_follows_the_flop_table():
    # the benchmark step as profiles/r02_bench_default.json records it
    line = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_default.json")))
    ev = line["roofline"]["events_per_launch"]
    st = _stats(**{k: int(v) for k, v in ev.items()})
    ms = line["roofline"]["kernel_ms"]
    r = bench.roofline(st, ms, _scene(), "reference", None, "none")
    assert r["kernel"] == "pt_render_kernel<4,false>" and r["bound"] == "valu" and r["unit"] == "TFLOP/s"
    F = bench.FLOP
    flop = (ev["shaded_hits"] * F["shade_reference"] + ev["samples"] * F["sample_end"] + ev["rays_traced"] * F["grid_query"]
            + ev["grid_triangle_tests"] * F["test_origin_records"] + ev["shaded_hits"] * F["test_accept"])
    assert r["executed_flop_per_launch"] == pytest.approx(flop)
    assert r["frac"] == pytest.approx(flop / (ms * 1e-3) * 1e-12 / bench.PEAK_FP32_UNFUSED_TFLOPS)
    assert r["frac"] == pytest.approx(line["roofline"]["frac"], rel=1e-9) and 0.2 < r["frac"] <= 1.0
    assert r["traffic"] is None and r["pmc"] == "none"                       # no counter entry: nothing made up
    assert r["brute_force_equivalent"]["frac_algorithmic"] > 1.0 and "frac" not in r["brute_force_equivalent"]


def test_loop_kernels_use_the_measured_skip_share_or_say_that_they_cannot():
    st = _stats(rays=1000, rays_traced=640, samples=100, lane_slots=1280, shaded_hits=500, loop_triangle_visits=20 * 36)
    low = bench.roofline(st, 1.0, _scene(), "reference", None, "none")
    assert low["kernel"] == "pt_render_kernel<1,false>" and isinstance(low["loop_tests_past_backface_skip"], str)
    with_share = bench.roofline(st, 1.0, _scene(), "reference", {"loop_tests_past_backface_skip": 0.5}, None)
    assert with_share["loop_tests_past_backface_skip"] == 0.5
    lanes = 20 * 36 * 64 * (640 / 1280)
    assert with_share["executed_flop_per_launch"] - low["executed_flop_per_launch"] == pytest.approx(lanes * bench.FLOP["loop_rest"] * 0.5)
    counted = bench.roofline(dict(st, loop_full_tests=20 * 36 // 4), 1.0, _scene(), "reference", {"loop_tests_past_backface_skip": 0.5}, None)
    assert counted["loop_tests_past_backface_skip"] == 0.25                   # the build's own count wins over the profile's


def test_walk_kernels_are_priced_in_bytes_against_the_level_their_records_fit():
    st = _stats(rays=5000, rays_traced=1000, samples=900, lane_slots=6400, grid_triangle_tests=26000, lane_walk_records=4000, shaded_hits=700)
    small = bench.roofline(st, 0.001, _scene(triangles=3936), "materials", None, "none")
    assert small["kernel"] == "pt_render_decoupled_kernel" and small["bound"] == "l2" and small["unit"] == "GB/s"
    nbytes = 22000 * 64 + 4000 * 48
    assert small["algorithmic_bytes_per_launch"] == nbytes
    assert small["frac"] == pytest.approx(nbytes / 1e-6 * 1e-9 / 16800.0)
    big = bench.roofline(st, 0.001, _scene(triangles=999744), "materials", {"l2_hit_rate": 0.6}, None)
    assert big["bound"] == "infinity_cache" and big["peak"] == 8600.0
    dep = big["dependent_gather"]
    rate = 26000 / 1e-6 * 1e-9
    ceiling = 1.0 / (0.6 / dep["l2_resident_Grecords_per_s"] + 0.4 / dep["beyond_l2_Grecords_per_s"])
    assert dep["achieved_Grecords_per_s"] == pytest.approx(rate) and dep["frac_of_ceiling"] == pytest.approx(rate / ceiling)
    assert dep["beyond_l2_Grecords_per_s"] < dep["l2_resident_Grecords_per_s"]
    wave = bench.roofline(_stats(rays=100, rays_traced=100, samples=10, lane_slots=6400, bvh_node_fetches=300, bvh_triangle_fetches=100),
                          0.001, _scene(triangles=3936), "reference", None, "none")
    assert wave["kernel"] == "pt_render_kernel<2,false>" and wave["algorithmic_bytes_per_launch"] == 400 * 64


def test_tracked_counter_entries_belong_to_the_current_kernels_or_are_ignored(tmp_path, monkeypatch):
    doc = json.load(open(bench.PMC_SIDECAR))
    key = bench.config_key("cornell", 1920, 1080, 256, "reference", "auto")
    assert key in doc["entries"]
    entry, note = bench.pmc_entry(key)
    if entry is None:
        assert "stale" in note            # kernels changed after the last profile: the line then carries no counter figures
    else:
        assert entry["kernel_hash"] == bench.kernel_source_hash() and entry["hbm_bytes_per_launch"] > 0
    stale = dict(doc, entries={key: dict(doc["entries"][key], kernel_hash="0" * 16)})
    p = tmp_path / "pmc.json"
    p.write_text(json.dumps(stale))
    monkeypatch.setattr(bench, "PMC_SIDECAR", str(p))
    entry, note = bench.pmc_entry(key)
    assert entry is None and "stale" in note
    assert bench.pmc_entry("no|such|configuration")[0] is None


def test_profile_key_ignores_comments_and_layout_but_not_code():
    """The hash that ties a counter profile to the kernel sources looks at code only: a comment or a line break changes nothing,
    a constant does; a comment marker inside a string literal is text."""
    a = 'int f() { return 3; }  // three\nconst char* s = "a // b";\n'
    b = '/* note */ int f()\n{\n    return 3;\n}\nconst char* s = "a // b";'
    c = 'int f() { return 4; }\nconst char* s = "a // b";'
    d = 'int f() { return 3; }\nconst char* s = "a";'
    assert bench.code_only(a) == bench.code_only(b)
    assert bench.code_only(a) != bench.code_only(c)
    assert bench.code_only(a) != bench.code_only(d)
