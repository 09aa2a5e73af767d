#!/usr/bin/env python3
"""A/B timing of experimental builds of the HIP library in ONE process, rounds interleaved (kernel variants are
compared on the same device, same clocks, same data). Build the variants first on the CPU box:

    python tools/ab_bench.py --build            # compiles pathtracer_amd/lib/libpt_hip__<tag>.so for every variant
    python tools/ab_bench.py --workload cornell --spp 64 --rounds 5     # on the GPU box
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {
    "head": None,   # the kernel sources as committed (git show HEAD:...), to compare the working tree ("base") against
    "base": [],
    "vfirst": ["-DPT_V_VFIRST"],
    "nosky": ["-DPT_V_NO_SKY"],
    "maxilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "memclause": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"],
    "nounroll": ["-fno-unroll-loops"],
    "o2": ["-O2"],
    "countfull": ["-DPT_V_COUNT_FULL"],  # all-triangles loops: counts the tests past the wave-uniform back-face skip (tools/pmc_collect.py uses it)
    "split07": ["-DPT_V_SPREAD_SPLIT=0.7f"],   # split factor of spread-out views 0.7 / 1 instead of 0.85
    "split1": ["-DPT_V_SPREAD_SPLIT=1.0f"],
    # tuning of the decoupled walk (materials mode through the hierarchy)
    "leaf1": ["-DPT_BVH_WIDE_LEAF_MAX=1"], "leaf4": ["-DPT_BVH_WIDE_LEAF_MAX=4"], "steps1": ["-DPT_WALK_NODE_STEPS=1"], "steps2": ["-DPT_WALK_NODE_STEPS=2"], "steps2b24": ["-DPT_WALK_NODE_STEPS=2", "-DPT_SHADE_BATCH=24"],
    "steps2b32": ["-DPT_WALK_NODE_STEPS=2", "-DPT_SHADE_BATCH=32"], "steps4": ["-DPT_WALK_NODE_STEPS=4"],
    "waves4": ["-DPT_DECOUPLED_MIN_WAVES=4"], "waves6": ["-DPT_DECOUPLED_MIN_WAVES=6"],
    "batch12": ["-DPT_SHADE_BATCH=12"], "batch24": ["-DPT_SHADE_BATCH=24"], "b16": ["-DPT_SHADE_BATCH=16"], "b32": ["-DPT_SHADE_BATCH=32"], "b40": ["-DPT_SHADE_BATCH=40"],
    "s1": ["-DPT_WALK_NODE_STEPS=1"], "s3": ["-DPT_WALK_NODE_STEPS=3"],
    "reshade": ["-DPT_V_RESHADE"],   # materials loop: a lane whose sample ended in Shade() shades its next sample's camera hit in the same trip
    "sub4": ["-DPT_SCHED_SUB_LOG=2"], "sub16": ["-DPT_SCHED_SUB_LOG=4"],   # work list: 4 / 16 cost buckets to an octave (16 / 4 octaves) instead of 8
    "live": ["-DPT_V_LIVE_COUNT"],   # direction grid: lanes still in the loop summed over the wave trips -> bvh_node_fetches (lane_slots minus that = slots of lanes whose pixel is finished)
    "iters": ["-DPT_V_COUNT_ITERS"],   # direction grid: list-loop trips per wave in bvh_node_fetches, lanes with >1 test in bvh_triangle_fetches
}


def build_committed(ptbuild, tag):
    """Compiles the HIP library from the sources of the last commit into libpt_hip__<tag>.so."""
    import subprocess
    import tempfile
    out = os.path.join(ptbuild.LIB, f"libpt_hip__{tag}.so")
    with tempfile.TemporaryDirectory() as tmp:
        csrc = os.path.join(tmp, "pathtracer_amd", "csrc")   # same relative layout: the sources include ../../include/pt_hip.h
        os.makedirs(csrc)
        os.makedirs(os.path.join(tmp, "include"))
        names = ["pt_kernels.hip", "pt_bvh.hip", "pt_grid.hip", "pt_capi.hip", "pt_internal.h"]
        for f in names:
            open(os.path.join(csrc, f), "wb").write(subprocess.check_output(["git", "-C", ROOT, "show", f"HEAD:pathtracer_amd/csrc/{f}"]))
        open(os.path.join(tmp, "include", "pt_hip.h"), "wb").write(subprocess.check_output(["git", "-C", ROOT, "show", "HEAD:include/pt_hip.h"]))
        subprocess.check_call([ptbuild.HIPCC] + ptbuild.DEVICE_FLAGS + ["-shared", "-o", out] + [os.path.join(csrc, f) for f in names[:4]])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--workload", default="cornell")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", default=None)
    ap.add_argument("--accel", type=int, default=0, help="0 auto, 1 brute, 2 bvh, 3 cull")
    ap.add_argument("--shading", default="reference")
    args = ap.parse_args()
    from pathtracer_amd import build as ptbuild
    names = [v for v in VARIANTS if not args.only or v in args.only.split(",")]
    if args.build:
        for v in names:
            if VARIANTS[v] is None:
                print(build_committed(ptbuild, v))
            else:
                print(ptbuild.build_hip(force=True, extra_flags=VARIANTS[v], tag=v))
        return
    import numpy as np
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(args.workload, args.width, args.height)
    tracers, ref = {}, None
    for v in names:
        t = Tracer(0, os.path.join(ROOT, "pathtracer_amd", "lib", f"libpt_hip__{v}.so"))
        scene.upload(t, accel=(args.accel in (0, 2)), shading=args.shading)
        t.set_accel_mode(args.accel)
        t.target_create(scene.width, scene.height)
        t.enable_timing(True)
        t.dispatch(1, 2, 1)
        t.finish()
        img = t.readback()
        if ref is None:
            ref = img
        same = bool(np.array_equal(img.view(np.uint32), ref.view(np.uint32)))
        tracers[v] = (t, same)
    times = {v: [] for v in names}
    util = {}
    for _ in range(args.rounds):
        for v in names:
            t, _ = tracers[v]
            t.target_clear()
            t.reset_stats()
            t.dispatch(1, args.spp, 1)
            t.finish()
            st = t.stats()
            times[v].append(st["last_dispatch_ms"])
            util[v] = (st["rays"], st["rays_traced"] / max(1, st["lane_slots"]))
    for v in names:
        rays, u = util[v]
        ms = statistics.median(times[v])
        print(json.dumps({"variant": v, "median_ms": round(ms, 3), "min_ms": round(min(times[v]), 3),
                          "Grays_per_s": round(rays / ms * 1e-6, 2), "lane_utilisation": round(u, 4), "tri_records_per_wave_ray": round(st["bvh_triangle_fetches"] * 64 / max(1, st["lane_slots"]), 2),
                          "node_records_per_wave_ray": round(st["bvh_node_fetches"] * 64 / max(1, st["lane_slots"]), 3),
                          "grid_tests_per_traced_ray": round(st["grid_triangle_tests"] / max(1, st["rays_traced"]), 3),
                          "same_image_as_first": tracers[v][1]}))


if __name__ == "__main__":
    main()
