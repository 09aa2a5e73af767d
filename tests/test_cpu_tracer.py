"""PT_PROGRAM_CPU_TRACER: the reference's CPU tracer (Scene::color, PathTracer.cpp:932-1052, dead code in its own build).

Three links of the chain:
  reference -> oracle   statistical. tests/golden/cpu_tracer_stats.npz holds per-pixel mean and variance over 256 frames from
                        the reference's OWN code (oracle/_ref/cpu_tracer: its lines compiled where they lie; oracle/
                        gen_cpu_tracer_stats.py). The reference draws from one serial rand() stream, so no sample-by-sample
                        comparison exists; the oracle's restatement (counter-based generator) must agree in distribution.
  oracle -> MI355X      bit for bit (same generator, same fp32 operation order).
  MI355X -> reference   statistical again, directly, at the fixture's size.
"""
import os

import numpy as np
import pytest

import pt_oracle as po
from helpers import GOLDEN, n_bit_mismatch


def _fixture():
    return dict(np.load(os.path.join(GOLDEN, "cpu_tracer_stats.npz")))


def _z_scores(mean_a, var_a, mean_b, var_b, n):
    random = (var_a > 1e-6) | (var_b > 1e-6)
    z = (mean_a - mean_b)[random] / np.sqrt((var_a + var_b)[random] / n)
    return z, random


def _stats_from_sums(sum1, sum2, n):
    mean = sum1 / n
    return mean, np.maximum(sum2 / n - mean * mean, 0.0)


def _check_against_reference(fx, mean, var, n):
    z, random = _z_scores(fx["mean"].astype(np.float64), fx["var"].astype(np.float64), mean, var, n)
    assert random.sum() > 5000
    assert abs(z.mean()) < 0.06, z.mean()                 # no bias: 6,300 channels, standard error about 0.013
    assert 0.93 < z.std() < 1.07, z.std()                 # differences as large as independent sampling predicts, no larger
    assert np.abs(z).max() < 5.5
    det = ~random
    # pixels that get the same value every frame (sky, metal and glass spheres): the same arithmetic, the same value
    # (the reference keeps an fp32 running sum of 256 values: rounding of about 4e-6)
    assert np.abs(fx["mean"].astype(np.float64) - mean)[det].max() < 2e-5
    both = (fx["var"] > 1e-4) & (var > 1e-4)
    assert 0.97 < np.median(var[both] / fx["var"][both]) < 1.03
    assert abs(mean.mean() - fx["mean"].mean()) < 2e-4


def test_oracle_restatement_agrees_with_the_reference_in_distribution():
    fx = _fixture()
    W, H, N = int(fx["width"]), int(fx["height"]), int(fx["passes"])
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(fx["camera"], W, H, len(fx["indices"]), 10, 1, 0)
    s1 = np.zeros((H, W, 3), np.float64)
    s2 = np.zeros((H, W, 3), np.float64)
    rays = 0
    for k in range(N):
        img, r = po.cpu_tracer(scene, u, 1 + k, 1, 1, max_t=float(fx["max_t"]), seed=99, threads=8)
        rays += r
        s1 += img[..., :3]
        s2 += img[..., :3].astype(np.float64) ** 2
    mean, var = _stats_from_sums(s1, s2, N)
    _check_against_reference(fx, mean, var, N)
    assert abs(rays / float(fx["rays"]) - 1.0) < 0.005      # rays per frame: 8,023,599 in the reference's run


def test_oracle_sums_are_additive_and_seeded():
    fx = _fixture()
    W, H = int(fx["width"]), int(fx["height"])
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(fx["camera"], W, H, len(fx["indices"]), 10, 1, 0)
    win = (32, 16, 64, 48)
    a, _ = po.cpu_tracer(scene, u, 1, 1, 6, seed=5, window=win, threads=4)
    b, _ = po.cpu_tracer(scene, u, 1, 2, 3, seed=5, window=win, threads=4)     # samples 1, 3, 5
    b, _ = po.cpu_tracer(scene, u, 2, 2, 3, b, seed=5, window=win, threads=4)  # then 2, 4, 6 on top
    assert np.abs(a - b).max() < 1e-5 and np.all(a[16:64, 32:96, 3] == 6) and np.all(a[:16] == 0)
    c, _ = po.cpu_tracer(scene, u, 1, 1, 6, seed=6, window=win, threads=4)
    assert n_bit_mismatch(a, c) > 1000


def test_framebuffer_keeps_sums_in_tile_traversal_order(tmp_path):
    """Framebuffer::storeSumsTileOrder against the layout the reference's own loop produced (the fixture's display bytes are
    in pixel order, its sums were read in tile order and un-tiled by the generator: tiling them again must give the
    reference's accumulationBuffer order)."""
    import ctypes as C
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    src.write_text('#include <cstdio>\n#include "Framebuffer.h"\nint main(){ Framebuffer fb; fb.create(48, 35); float img[48*35*4];\n'
                   'for (int i = 0; i < 48*35; ++i) { img[4*i] = i; img[4*i+1] = 0.25f; img[4*i+2] = 1.0f; img[4*i+3] = 4; }\n'
                   'fb.storeSumsTileOrder(img, 4);\n'
                   'for (int k = 0; k < 3*16*16*2; ++k) printf("%g ", fb.accumulationBuffer[3*k]);\n'
                   'const Color* c = (const Color*)fb.colorBuffer; printf("\\n%d %d %d %d\\n", c[5].R, c[5].G, c[5].B, c[5].A); return 0; }\n')
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-O1", "-I", os.path.join(root, "pathtracer_amd", "host"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True).splitlines()
    order = np.array([float(v) for v in out[0].split()])
    # 48x35 -> 3x2 whole tiles; k-th entry = k-th pixel of the loops th, tw, j, i
    want = []
    for th in range(2):
        for tw in range(3):
            for j in range(16):
                for i in range(16):
                    want.append(48 * (16 * th + j) + 16 * tw + i)
    assert np.array_equal(order, np.array(want, float))
    # colour of pixel 5: fromLinear(z, y, x) of sum / samples = (sqrt(0.25), sqrt(0.0625), sqrt(1.25 -> 1)) x 255.49, truncated
    assert out[1].split() == ["127", "63", "255", "255"]


@pytest.mark.gpu
def test_device_equals_oracle_bit_for_bit():
    from pathtracer_amd import Tracer, capi
    fx = _fixture()
    W, H = int(fx["width"]), int(fx["height"])
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    ntri = len(fx["indices"])
    with Tracer(0) as t:
        prog = t.program_create(capi.PROGRAM_CPU_TRACER)
        t.use_program(prog)
        t.upload_spheres(fx["spheres"])
        t.upload_materials(fx["materials"])
        t.upload_mesh(fx["vertices"], fx["indices"])
        t.set_accum_mode(capi.ACCUM_SUM)
        t.target_create(W, H)
        t.set_uniforms(fx["camera"], ntri, 10)
        t.set_cpu_tracer(1234, float(fx["max_t"]))
        u = po.uniforms(fx["camera"], W, H, ntri, 10, 1, 0)
        t.dispatch(1, 6, 1)
        t.finish()
        ref, rays = po.cpu_tracer(scene, u, 1, 1, 6, max_t=float(fx["max_t"]), seed=1234, threads=8)
        assert n_bit_mismatch(t.readback(), ref) == 0
        assert t.stats()["rays"] == rays
        # two more dispatches on top (strided sample indices), a window, another seed, no mesh, bounce limits
        t.dispatch(7, 3, 2)
        t.dispatch(8, 3, 2)
        t.finish()
        ref, _ = po.cpu_tracer(scene, u, 7, 2, 3, ref, max_t=float(fx["max_t"]), seed=1234, threads=8)
        ref, _ = po.cpu_tracer(scene, u, 8, 2, 3, ref, max_t=float(fx["max_t"]), seed=1234, threads=8)
        assert n_bit_mismatch(t.readback(), ref) == 0
        t.target_clear()
        t.set_window(13, 9, 70, 50)
        t.set_cpu_tracer(77, 5.0)          # a short camera ray: the far spheres are out of its reach
        t.dispatch_static(1, 40, 1)        # (the allocation-free dispatch serves the program too)
        t.finish()
        ref, _ = po.cpu_tracer(scene, u, 1, 1, 40, max_t=5.0, seed=77, window=(13, 9, 70, 50), threads=8)
        assert n_bit_mismatch(t.readback(), ref) == 0
        t.set_window()
        for depth, tris in ((1, ntri), (3, 0), (0, ntri)):
            t.target_clear()
            t.set_uniforms(fx["camera"], tris, depth)
            t.set_cpu_tracer(3, float(fx["max_t"]))
            t.dispatch(1, 4, 1)
            t.finish()
            u2 = po.uniforms(fx["camera"], W, H, tris, depth, 1, 0)
            ref, _ = po.cpu_tracer(scene, u2, 1, 1, 4, max_t=float(fx["max_t"]), seed=3, threads=8)
            assert n_bit_mismatch(t.readback(), ref) == 0, (depth, tris)
        t.program_destroy(prog)


@pytest.mark.gpu
def test_device_agrees_with_the_reference_in_distribution():
    from pathtracer_amd import Tracer, capi
    fx = _fixture()
    W, H, N = int(fx["width"]), int(fx["height"]), int(fx["passes"])
    with Tracer(0) as t:
        prog = t.program_create(capi.PROGRAM_CPU_TRACER)
        t.use_program(prog)
        t.upload_spheres(fx["spheres"])
        t.upload_materials(fx["materials"])
        t.upload_mesh(fx["vertices"], fx["indices"])
        t.set_accum_mode(capi.ACCUM_SUM)
        t.target_create(W, H)
        t.set_uniforms(fx["camera"], len(fx["indices"]), 10)
        t.set_cpu_tracer(2026, float(fx["max_t"]))
        s1 = np.zeros((H, W, 3), np.float64)
        s2 = np.zeros((H, W, 3), np.float64)
        for k in range(N):                   # one frame per dispatch, like the reference's loop: per-frame radiance for the variance
            t.target_clear()
            t.dispatch(1 + k, 1, 1)
            t.finish()
            img = t.readback()[..., :3].astype(np.float64)
            s1 += img
            s2 += img * img
        rays = t.stats()["rays"]
        t.program_destroy(prog)
    mean, var = _stats_from_sums(s1, s2, N)
    _check_against_reference(fx, mean, var, N)
    assert abs(rays / float(fx["rays"]) - 1.0) < 0.005


@pytest.mark.gpu
def test_cpp_host_runs_the_cpu_tracer_branch(tmp_path):
    """The C++ host's render loop with the CPU tracer selected (the branch the reference compiles without USE_GLSL): sums
    arrive in Framebuffer::accumulationBuffer's tile-traversal order, display colours in pixel order; against the oracle."""
    from pathtracer_amd import host_api
    from pathtracer_amd.scenes import ASSETS
    fx = _fixture()
    W, H = 160, 96
    rgb_sum, rgba8, stats = host_api.render(os.path.join(ASSETS, "monkeyhead2.obj"), W, H, passes=5, samples_per_dispatch=2, program_kind=2)
    res = {"rgb_sum": rgb_sum, "rgba8": rgba8, "rays": stats["rays"]}
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(fx["camera"], W, H, len(fx["indices"]), 10, 1, 0)
    ref, rays = po.cpu_tracer(scene, u, 1, 1, 5, max_t=float(fx["max_t"]), seed=18, threads=8)
    tiled = []
    for th in range(H // 16):
        for tw in range(W // 16):
            tiled.append(ref[16 * th:16 * th + 16, 16 * tw:16 * tw + 16, :3].reshape(-1, 3))
    assert n_bit_mismatch(res["rgb_sum"].reshape(-1, 3), np.concatenate(tiled)) == 0
    assert res["rays"] == rays
    mean = ref[..., :3] * np.float32(1.0 / 5.0)
    want = (np.float32(255.49) * np.clip(np.sqrt(mean[..., ::-1]), 0, 1)).astype(np.uint8)
    assert np.array_equal(res["rgba8"][..., :3], want) and np.all(res["rgba8"][..., 3] == 255)
