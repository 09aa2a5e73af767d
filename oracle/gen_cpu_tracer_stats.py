#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY: per-pixel statistics of the REFERENCE's own CPU tracer (oracle/_ref/cpu_tracer: Scene::color and
the tile loop compiled from the lines where they lie in /root/reference/PathTracer.cpp) on the reference's own scene
(monkeyhead2.obj at z + 3 with the reference's vertex/normal pairing, its 10 spheres and materials from glibc's rand() after
srand(18), its camera), written to tests/golden/cpu_tracer_stats.npz: mean and variance of the per-frame radiance of every
pixel over N frames, the display bytes, the ray count. The reference draws from one serial rand() stream, so this is what a
parallel implementation can be compared with: statistics, not samples.

Also checks the oracle's restatement (pt_oracle.c: pto_cpu_tracer, counter-based generator) against those statistics and
prints the report that tests/test_cpu_tracer.py asserts.

Usage:  make -C oracle ref && python3 oracle/gen_cpu_tracer_stats.py          (needs /root/reference)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pt_oracle as po  # noqa: E402

BIN = os.path.join(HERE, "_ref", "cpu_tracer")
W, H, PASSES, SEED = 160, 96, 256, 18


def tile_to_pixel_order(a, width, height, tile=16):
    """Framebuffer::accumulationBuffer is filled in tile-traversal order (PathTracer.cpp:1326-1357): k-th entry = k-th pixel
    visited by the loops th, tw, j, i. Returns the [H, W, C] image."""
    a = a.reshape(-1, a.shape[-1])
    out = np.zeros((height, width, a.shape[-1]), a.dtype)
    k = 0
    for th in range(height // tile):
        for tw in range(width // tile):
            block = a[k:k + tile * tile].reshape(tile, tile, -1)
            out[th * tile:(th + 1) * tile, tw * tile:(tw + 1) * tile] = block
            k += tile * tile
    return out


def compare_statistics(mean_a, var_a, mean_b, var_b, n):
    """z-scores of the difference of two per-pixel means over n samples each, and how the deterministic pixels compare."""
    # (the reference's per-frame radiance is recovered as a difference of its fp32 running sum, which leaves rounding noise of
    # about 1e-10 in the variance of pixels that get the same value every frame: "random" starts well above that)
    random = (var_a > 1e-6) | (var_b > 1e-6)
    z = (mean_a - mean_b)[random] / np.sqrt((var_a + var_b)[random] / n)
    det = ~random
    ratio = var_b[random & (var_a > 1e-6)] / var_a[random & (var_a > 1e-6)]
    return {"random_channels": int(random.sum()), "z_mean": float(z.mean()), "z_std": float(z.std()), "z_abs_max": float(np.abs(z).max()),
            "deterministic_channels": int(det.sum()),
            "deterministic_max_abs_diff": float(np.abs(mean_a - mean_b)[det].max()) if det.any() else 0.0,
            "variance_ratio_median": float(np.median(ratio)), "image_mean_a": float(mean_a.mean()), "image_mean_b": float(mean_b.mean())}


def main():
    from helpers import load_case
    fx = load_case("monkey_mesh_128")          # the reference's mesh as its host builds it + the glibc sphere / material tables
    tmp = tempfile.mkdtemp(prefix="cput_")
    fx["vertices"].astype(np.float32).tofile(f"{tmp}/v.f32")
    fx["indices"].astype(np.int32).tofile(f"{tmp}/i.i32")
    fx["spheres"].astype(np.float32).tofile(f"{tmp}/s.f32")
    fx["materials"].astype(np.float32).tofile(f"{tmp}/m.f32")
    job = f"""width {W}
height {H}
passes {PASSES}
seed {SEED}
fov 60
near 0.1
far 100
eye 0 0 0
target 0 0 1
up 0 1 0
spheres {tmp}/s.f32
materials {tmp}/m.f32
vertices {tmp}/v.f32
indices {tmp}/i.i32
out_sum {tmp}/sum.f32
out_sumsq {tmp}/sumsq.f32
out_color {tmp}/color.u8
"""
    open(f"{tmp}/job.txt", "w").write(job)
    r = subprocess.run([BIN, f"{tmp}/job.txt"], capture_output=True, text=True, check=True)
    info = json.loads(r.stdout.strip().splitlines()[-1])
    s = tile_to_pixel_order(np.fromfile(f"{tmp}/sum.f32", np.float32).reshape(-1, 3), W, H).astype(np.float64)
    sq = tile_to_pixel_order(np.fromfile(f"{tmp}/sumsq.f32", np.float32).reshape(-1, 3), W, H).astype(np.float64)
    color = np.fromfile(f"{tmp}/color.u8", np.uint8).reshape(H, W, 4)
    mean = s / PASSES
    var = np.maximum(sq / PASSES - mean * mean, 0.0)
    cam = po.camera(60.0, W, H, (0, 0, 0), (0, 0, 1), (0, 1, 0))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cpu_tracer_stats.npz"), width=W, height=H, passes=PASSES, seed=SEED,
                        rays=info["rays"], camera=cam, max_t=np.float32(100.0) - np.float32(0.1), mean=mean.astype(np.float32), var=var.astype(np.float32),
                        color=color, vertices=fx["vertices"], indices=fx["indices"], spheres=fx["spheres"], materials=fx["materials"])
    # the restatement against it
    scene = po.Scene(fx["vertices"], fx["indices"], fx["spheres"], fx["materials"])
    u = po.uniforms(cam, W, H, len(fx["indices"]), 10, 1, 0)
    mine = np.zeros((H, W, 3), np.float64)
    mine_sq = np.zeros((H, W, 3), np.float64)
    rays = 0
    for k in range(PASSES):
        img, r_ = po.cpu_tracer(scene, u, 1 + k, 1, 1, max_t=float(np.float32(100.0) - np.float32(0.1)), seed=7, threads=8)
        rays += r_
        mine += img[..., :3]
        mine_sq += img[..., :3].astype(np.float64) ** 2
    m2 = mine / PASSES
    v2 = np.maximum(mine_sq / PASSES - m2 * m2, 0.0)
    rep = compare_statistics(mean, var, m2, v2, PASSES)
    rep["rays_reference"] = info["rays"]
    rep["rays_restatement"] = rays
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
