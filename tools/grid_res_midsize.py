#!/usr/bin/env python3
"""Direction-grid resolution on meshes between the small assets and the million triangles: n jittered copies of suzanne.obj
(the large scene's generator with fewer instances and columns), 1920x1080, 256 spp per dispatch, static camera; for each mesh
the resolutions given, forced with PT_GRID_RES (read when a context is created), and the library's own choice ("auto").

    python3 tools/grid_res_midsize.py --instances 8,25,64 --res auto,1024,1536,2048
"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--instances", default="8,25,64")
    ap.add_argument("--res", default="auto,1024,1536,2048")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--reps", type=int, default=8)
    args = ap.parse_args()
    import numpy as np
    from pathtracer_amd import Tracer, host_api
    from pathtracer_amd.scenes import ASSETS, SceneData, instanced_mesh
    base = host_api.ObjFile(os.path.join(ASSETS, "suzanne.obj"), ASSETS + "/").mesh_untranslated
    for n in [int(x) for x in args.instances.split(",")]:
        columns = max(1, int(round(n ** 0.5 * 1.33)))
        scene = SceneData("instanced", 1920, 1080)   # camera, spheres and normal mode of the large scene
        scene.vertices, scene.indices = instanced_mesh(base["vertices"], base["indices"], instances=n, columns=columns, unit=30.0)
        scene.num_triangles = len(scene.indices)
        rows = (n + columns - 1) // columns
        dist = max(columns, rows * 16.0 / 9.0) * 2.8 * 30.0 * 1.0   # far enough to see the whole grid at 60 degrees
        scene.camera = host_api.camera(60.0, 1920, 1080, (0.0, 0.0, -dist), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0))
        scene.material_ids = np.zeros(scene.num_triangles, np.int32)
        for r in args.res.split(","):
            if r == "auto":
                os.environ.pop("PT_GRID_RES", None)
            else:
                os.environ["PT_GRID_RES"] = r
            t = Tracer(0)
            scene.upload(t)
            t.target_create(scene.width, scene.height)
            t.enable_timing(True)
            t.dispatch(1, 1, 1)
            t.finish()
            ms = []
            for _ in range(args.reps):
                t.target_clear()
                t.reset_stats()
                t.dispatch(1, args.spp, 1)
                t.finish()
                st = t.stats()
                ms.append(st["last_dispatch_ms"])
            print(json.dumps({"instances": n, "triangles": scene.num_triangles, "res": r, "median_ms": round(statistics.median(ms[len(ms) // 2:]), 3),
                              "min_ms": round(min(ms), 3), "tests_per_evaluated_query": round(st["grid_triangle_tests"] / max(1, st["rays_traced"]), 3),
                              "grid_entries": st["grid_entries"]}), flush=True)
            t.close()


if __name__ == "__main__":
    main()
