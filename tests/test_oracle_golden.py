"""The oracle (oracle/pt_oracle.c) against what the REFERENCE produced in the build container:
its fragment shader run on Mesa llvmpipe (tests/golden/case_*.npz, lp_math.npz, lp_rng.npz; made by
oracle/gen_golden.py). Everything here is bit-for-bit: primary-hit ids, t, ray direction and the RGBA32F
accumulation after 1..16 passes."""
import numpy as np
import pytest

import pt_oracle as po
from helpers import CASES, GOLDEN, bits, case_shading_mode, load_case, n_bit_mismatch


@pytest.mark.parametrize("kind,col", [("sin", 0), ("cos", 0), ("exp2", 1), ("pow", 2)])
def test_llvmpipe_math_known_answers(kind, col):
    d = np.load(f"{GOLDEN}/lp_math.npz")
    a, ref = d["inputs"], d["lp_out"]
    out = po.lp_math(kind, a[:, col], a[:, 3])
    idx = {"sin": 0, "cos": 1, "exp2": 2, "pow": 3}[kind]
    assert n_bit_mismatch(out, ref[:, idx]) == 0


def test_rng_known_answers():
    d = np.load(f"{GOLDEN}/lp_rng.npz")
    for s in (1, 7, 1000):
        img = d[f"draws_s{s}"]
        H, W, _ = img.shape
        for y in range(0, H, 5):
            for x in range(0, W, 7):
                assert n_bit_mismatch(po.rng_stream(x, y, s, 4), img[y, x]) == 0, (s, x, y)


def test_world_tables_match_glibc_stream():
    d = np.load(f"{GOLDEN}/world_tables_glibc.npz")
    sph, mat = po.reference_spheres_and_materials()
    assert n_bit_mismatch(sph, d["spheres"]) == 0 and n_bit_mismatch(mat, d["materials"]) == 0
    # PathTracer.cpp:1220-1244: first sphere with glibc's rand() after srand(18)
    np.testing.assert_allclose(sph[0], [2.574437, 2.477976, 9.276695, 0.312389], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name", CASES)
def test_case_bit_exact(name):
    fx = load_case(name)
    W, H = int(fx["width"]), int(fx["height"])
    x0, y0, w, h = (int(v) for v in fx["crop"])
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"], fx.get("material_ids"))
    u = po.uniforms(fx["camera"], W, H, len(fx["indices"]), int(fx["depth"]), int(fx["variant"]), int(fx["normal_mode"]),
                    case_shading_mode(fx))
    ids, ts, rds = po.primary(scene, u, window=(x0, y0, w, h), threads=8)
    win = (slice(y0, y0 + h), slice(x0, x0 + w))
    assert np.array_equal(ids[win], fx["lp_id"])
    hit = fx["lp_id"] >= 0
    assert hit.any()
    assert n_bit_mismatch(ts[win][hit], fx["lp_t"][hit]) == 0
    assert n_bit_mismatch(rds[win][..., 0], fx["lp_rdx"]) == 0
    assert n_bit_mismatch(rds[win][..., 1], fx["lp_rdy"]) == 0
    rgba, done = None, 0
    for k in sorted(int(s) for s in fx["spp"]):
        rgba, rays = po.render(scene, u, done + 1, k - done, rgba, window=(x0, y0, w, h), threads=8)
        done = k
        assert rays >= w * h * 1
        assert n_bit_mismatch(rgba[win][..., :3], fx[f"lp_rgb_{k}"]) == 0, f"{name}: radiance after {k} passes"
        assert np.all(rgba[win][..., 3] == 1.0)
    if "lp_present_rgba8" in fx:
        assert np.array_equal(po.present(rgba), fx["lp_present_rgba8"])


def test_camera_reference_setup():
    # PathTracer.cpp:1160-1162 at the reference's 1280x720: eye 0, looking +z, vfov 60
    cam = po.camera(60.0, 1280, 720, (0, 0, 0), (0, 0, 1), (0, 1, 0))
    t = np.float32(np.tan(np.float64(np.float32(60.0)) * np.pi / 360.0))
    np.testing.assert_array_equal(cam[9:12], np.array([0, 2 * t, 0], np.float32))
    np.testing.assert_array_equal(cam[3:6], np.array([-t * np.float32(1280 / 720), -t, 1], np.float32))


def test_sample_sum_equals_mean_within_rounding():
    fx = load_case("monkey_mesh_128")
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(fx["camera"], 128, 128, len(fx["indices"]), 10, 1, 0)
    win = (40, 40, 32, 32)
    mean, _ = po.render(scene, u, 1, 8, window=win, threads=4)
    s0, r0 = po.render_sum(scene, u, 1, 2, 4, window=win, threads=4)   # samples 1,3,5,7
    s1, r1 = po.render_sum(scene, u, 2, 2, 4, window=win, threads=4)   # samples 2,4,6,8
    merged = (s0 + s1)[40:72, 40:72, :3] * np.float32(1 / 8)
    assert np.abs(merged - mean[40:72, 40:72, :3]).max() < 2e-6
