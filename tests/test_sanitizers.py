"""The CPU side under AddressSanitizer + UndefinedBehaviorSanitizer (host compiler; GPU code has no sanitizer here): the OBJ/MTL
reader on every asset and on 10,000 mutated OBJ/MTL texts, the scene construction, the workloads, the Framebuffer stores, and
the parity oracle, all in one statically instrumented executable (oracle/Makefile target `asan`, driver
tests/native/sanitize_main.cpp). A finding aborts the run; the oracle's results under the sanitizers must equal the regular
build's, checksum for checksum."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import pt_oracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_build", "sanitize_check")


def fnv1a(*arrays):
    h = 1469598103934665603
    for a in arrays:
        for b in np.ascontiguousarray(a).view(np.uint8).reshape(-1).tolist():
            h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


@pytest.fixture(scope="module")
def report():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("no sanitizer build on this machine: " + r.stderr[-300:])   # (a host compiler without the static runtimes)
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
        r = subprocess.run([BIN, os.path.join(ROOT, "assets"), tmp, "10000"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr and "LeakSanitizer" not in r.stderr, r.stderr[-4000:]
    lines = [l.split() for l in r.stdout.strip().splitlines()]
    assert lines[-1] == ["done"]
    return lines


def test_reader_and_scene_code_run_clean_on_assets_and_mutated_files(report):
    assets = {l[1]: l for l in report if l[0] == "asset"}
    # positions, triangles, materials per file: what the reference's own reader returns (tests/golden/obj_*.npz)
    for name in ("cornell_box", "monkeyhead2", "suzanne", "catmark_torus_creases0", "cube", "no_material", "monkeyhead"):
        d = np.load(os.path.join(ROOT, "tests", "golden", f"obj_{name}.npz"))
        assert [int(v) for v in assets[name][2:5]] == [len(d["v"]), len(d["idx_v"]), len(d["mat_names"])]
    fuzz = [l for l in report if l[0] == "fuzz"][0]
    assert int(fuzz[1]) == 10000 and int(fuzz[2]) == 10000      # every mutated file was read to its end and built into a scene
    assert {l[1] for l in report if l[0] == "workload"} == {"cornell", "monkeyhead2", "world", "torus", "torus_distorted"}
    assert len([l for l in report if l[0] == "framebuffer"]) == 4


def test_oracle_under_the_sanitizers_equals_the_regular_build(report):
    from pathtracer_amd.scenes import SceneData
    sc = SceneData("monkeyhead2", 40, 24)
    table = np.zeros((10, 4), np.float32)
    table[:4] = [[0.5, 0.5, 0.5, 0.0], [0.85, 0.3, 0.2, 0.0], [0.25, 0.7, 0.35, 0.0], [1.0, 0.9, 0.7, 3.0]]
    t = np.arange(sc.num_triangles)
    mids = np.where(t % 11 == 5, 3, t % 3).astype(np.int32)
    got = {l[1]: (int(l[2]), int(l[3], 16)) for l in report if l[0] == "oracle" and len(l) == 4}
    for name, variant, normal_mode, shading in (("mesh", 1, 0, 0), ("mesh_geo", 1, 1, 0), ("world", 0, 0, 0), ("materials", 1, 0, 1),
                                                ("materials_unit", 1, 1, 2)):
        scene = po.Scene(sc.vertices, sc.indices, sc.spheres, table if shading else sc.materials, mids)
        u = po.uniforms(sc.camera, 40, 24, sc.num_triangles, 10, variant, normal_mode, shading)
        rgba, rays = po.render(scene, u, 1, 2)
        rgba, r2 = po.render(scene, u, 3, 1, rgba, window=(5, 3, 20, 11))
        sums, r3 = po.render_sum(scene, u, 2, 3, 2)
        ids, ts, _ = po.primary(scene, u)
        want = fnv1a(rgba, sums, ids, ts, po.present(rgba))
        assert got[name] == (rays + r2 + r3, want), name
    scene = po.Scene(sc.vertices, sc.indices, sc.spheres, sc.materials, mids)
    u = po.uniforms(sc.camera, 40, 24, sc.num_triangles, 10, 1, 0, 0)
    sums, rays = po.cpu_tracer(scene, u, 1, 1, 2, max_t=99.9, seed=18)
    assert got["cpu_tracer"] == (rays, fnv1a(sums))
    draws = po.rng_stream(3, 5, 7.0, 16)
    a = np.array([0.0, 1.0, -40.0, 6.2831, 1e-30, 0.5, 130.0, -130.0], np.float32)
    b = np.array([0.5, 1.0 / 1025.0, 2.0, 0.0, 4.0, 0.25, 1.0, 3.0], np.float32)
    out = np.stack([po.lp_math(k, a, b) for k in ("sin", "cos", "exp2", "pow")])
    math_line = [l for l in report if l[:2] == ["oracle", "math"]][0]
    assert int(math_line[2], 16) == fnv1a(draws, out)
