#!/usr/bin/env python3
"""One warm-up and N timed dispatches of a workload: the program to put behind rocprofv3."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cornell")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--lib", default=None)
    ap.add_argument("--accel", default="auto", choices=["auto", "brute", "bvh", "cull", "grid"])
    ap.add_argument("--shading", default="reference", choices=["reference", "materials"])
    ap.add_argument("--cold", action="store_true", help="no warm-up dispatch: the first timed dispatch has no tile costs to go by")
    ap.add_argument("--no-reuse", action="store_true", help="evaluate the camera ray for every sample, like the shader")
    args = ap.parse_args()
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(args.workload, args.width, args.height)
    t = Tracer(0, args.lib)
    scene.upload(t, shading=args.shading)
    t.target_create(scene.width, scene.height)
    if args.accel == "bvh":
        import time
        t0 = time.perf_counter()
        t.build_accel()
        t.finish()
        build_s = time.perf_counter() - t0
        t.set_accel_mode(2)
    else:
        build_s = 0.0
        t.set_accel_mode({"brute": 1, "cull": 3, "auto": 0, "grid": 4}[args.accel])
    t.set_primary_reuse(not args.no_reuse)
    t.enable_timing(True)
    if not args.cold:
        t.dispatch(1, 1, 1)
        t.finish()
    t.reset_stats()
    per_rep = []
    for _ in range(args.reps):
        t.target_clear()
        t.dispatch(1, args.spp, 1)
        t.finish()
        per_rep.append(round(t.stats()["last_dispatch_ms"], 3))
    st = t.stats()
    ms = st["total_dispatch_ms"] / st["dispatches"]
    print(json.dumps({"workload": args.workload, "size": [scene.width, scene.height], "spp": args.spp, "triangles": scene.num_triangles,
                      "accel": args.accel, "shading": args.shading, "bvh_build_s": round(build_s, 4), "kernel_ms": ms, "per_rep_ms": per_rep, "Grays_per_s": st["rays"] / st["dispatches"] / ms * 1e-6, "traced_Grays_per_s": st["rays_traced"] / st["dispatches"] / ms * 1e-6,
                      "Gsamples_per_s": st["samples"] / st["dispatches"] / ms * 1e-6, "primary_reuse": not args.no_reuse,
                      "rays_per_sample": st["rays"] / st["samples"], "lane_utilisation": st["rays_traced"] / max(1, st["lane_slots"]),
                      "triangle_records_per_wave_ray": st["bvh_triangle_fetches"] * 64 / st["lane_slots"],
                      "grid_tests_per_traced_ray": st["grid_triangle_tests"] / max(1, st["rays_traced"]), "grid_entries": st["grid_entries"],
                      "node_records_per_wave_ray": st["bvh_node_fetches"] * 64 / st["lane_slots"],
                      "loop_triangle_visits": st["loop_triangle_visits"], "loop_full_tests": st["loop_full_tests"]}))
    t.close()


if __name__ == "__main__":
    main()
