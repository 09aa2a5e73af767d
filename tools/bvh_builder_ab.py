#!/usr/bin/env python3
"""Materials mode through the hierarchy made by the device's Morton-code builder, by the host's surface-area builder and by the
device's cluster-merging builder (PT_BVH_BUILDER=morton / sah / ploc): build time, dispatch time, records fetched per query, the self-check, and whether the images agree."""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="instanced:16,suzanne:64,monkeyhead2:64,torus_distorted:16")
    ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--shading", default="materials")
    ap.add_argument("--accel", default="auto")
    ap.add_argument("--lib", default=None, help="a variant library (pathtracer_amd.build.build_hip(tag=...))")
    ap.add_argument("--builders", default="morton,sah,ploc,binned")
    args = ap.parse_args()
    from pathtracer_amd import Tracer, capi
    from pathtracer_amd.scenes import SceneData
    for item in args.workloads.split(","):
        wl, spp = item.split(":")
        spp = int(spp)
        scene = SceneData(wl, None, None)
        digests = {}
        for builder in args.builders.split(","):
            os.environ["PT_BVH_BUILDER"] = builder
            t = Tracer(0, args.lib)
            with t:
                scene.upload(t, shading=args.shading, accel=False)
                t.target_create(scene.width, scene.height)
                t.finish()
                t0 = time.perf_counter()
                t.build_accel()
                t.finish()
                build_s = time.perf_counter() - t0
                os.environ.pop("PT_BVH_BUILDER")
                if args.accel == "bvh":
                    t.set_accel_mode(capi.ACCEL_BVH)
                t.enable_timing(True)
                t.dispatch(1, 1, 1)
                t.dispatch(2, 1, 1)
                t.finish()
                t.reset_stats()
                ms = []
                for _ in range(args.reps):
                    t.target_clear()
                    t.dispatch(1, spp, 1)
                    t.finish()
                    ms.append(round(t.stats()["last_dispatch_ms"], 3))
                st = t.stats()
                digests[builder] = hashlib.sha256(t.readback().tobytes()).hexdigest()
                compared, bad_id, bad_t = t.verify_accel(2)
                print(json.dumps({"workload": wl, "spp": spp, "triangles": scene.num_triangles, "builder": builder, "build_s": round(build_s, 4),
                                  "dispatch_ms": ms, "steps_per_traced_ray": st["grid_triangle_tests"] / max(1, st["rays_traced"]),
                                  "triangle_records_per_traced_ray": st["lane_walk_records"] / max(1, st["rays_traced"]),
                                  "wave_walk_nodes": st["bvh_node_fetches"], "self_check": [compared, bad_id, bad_t]}), flush=True)
        print(json.dumps({"workload": wl, "images_equal": len(set(digests.values())) == 1, "sha256": sorted(set(digests.values()))[0][:16]}), flush=True)


if __name__ == "__main__":
    main()
