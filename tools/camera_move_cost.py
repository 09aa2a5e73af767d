#!/usr/bin/env python3
"""What a camera move costs: per-triangle records, compaction, LBVH emit and direction grid are rebuilt at the first
dispatch after the camera position changed. Prints host wall time of that dispatch minus a plain one."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="instanced")
    ap.add_argument("--variant", default=None, help="an experimental library built by tools/ab_bench.py --build")
    ap.add_argument("--shading", default="reference", choices=["reference", "materials"])
    args = ap.parse_args()
    import numpy as np
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(args.workload)
    t = Tracer(0, os.path.join(ROOT, "pathtracer_amd", "lib", f"libpt_hip__{args.variant}.so")) if args.variant else Tracer(0)
    scene.upload(t, shading=args.shading)
    t.target_create(scene.width, scene.height)

    def frame():
        t0 = time.perf_counter()
        t.dispatch(1, 1, 1)
        t.finish()
        return (time.perf_counter() - t0) * 1e3

    frame()
    plain = min(frame() for _ in range(5))
    moved = []
    for k in range(4):
        cam = np.array(scene.camera, np.float32).copy()
        cam[0:3] += np.float32(0.01 * (k + 1))      # position
        cam[3:6] += np.float32(0.01 * (k + 1))      # lower-left corner moves with it
        t.set_uniforms(cam, scene.num_triangles, 10)
        moved.append(frame())
    print(json.dumps({"workload": args.workload, "shading": args.shading, "library": args.variant or "current", "triangles": scene.num_triangles, "plain_frame_ms": round(plain, 3),
                      "frame_after_camera_move_ms": [round(m, 3) for m in moved], "grid_entries": t.stats()["grid_entries"]}))
    t.close()


if __name__ == "__main__":
    main()
