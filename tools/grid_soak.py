#!/usr/bin/env python3
"""Soak test of the direction grid: many random camera positions (inside, outside, close to surfaces) on several scenes,
direction grid against the LBVH and the all-triangles loop; prints the number of differing words (must be 0)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cameras", type=int, default=40)
    ap.add_argument("--spp", type=int, default=6)
    ap.add_argument("--seed", type=int, default=2024)
    args = ap.parse_args()
    from pathtracer_amd import Tracer, capi, host_api
    from pathtracer_amd.scenes import SceneData
    rng = np.random.default_rng(args.seed)
    total_bad, total_rays = 0, 0
    for workload, W, H, modes in (("cornell", 320, 240, (capi.ACCEL_BRUTE, capi.ACCEL_BVH)), ("monkeyhead2", 320, 240, (capi.ACCEL_BRUTE, capi.ACCEL_BVH)),
                                  ("suzanne", 320, 240, (capi.ACCEL_BVH,)), ("instanced", 640, 360, (capi.ACCEL_BVH,))):
        sc = SceneData(workload, W, H)
        t = Tracer(0)
        sc.upload(t)
        t.target_create(W, H)
        lo, hi = sc.vertices[:, :3].min(0), sc.vertices[:, :3].max(0)
        ext = hi - lo
        for k in range(args.cameras):
            kind = k % 4
            if kind == 0:      # inside the bounding box
                eye = lo + rng.uniform(0.05, 0.95, 3) * ext
            elif kind == 1:    # outside, looking in
                d = rng.normal(size=3); d /= np.linalg.norm(d)
                eye = 0.5 * (lo + hi) + d * np.linalg.norm(ext) * rng.uniform(0.6, 2.5)
            elif kind == 2:    # very close to a vertex
                v = sc.vertices[rng.integers(len(sc.vertices)), :3]
                eye = v + rng.normal(size=3) * 1e-3 * np.linalg.norm(ext)
            else:              # in the plane of a triangle, beside it
                tri = sc.vertices[sc.indices[rng.integers(len(sc.indices))].astype(np.int64), :3]
                eye = tri[0] + 1.7 * (tri[1] - tri[0]) + 0.9 * (tri[2] - tri[0])
            target = lo + rng.uniform(0.2, 0.8, 3) * ext
            cam = host_api.camera(rng.uniform(40, 110), W, H, eye, target, (0.0, 1.0, 0.05))
            t.set_uniforms(cam, sc.num_triangles, 10)
            out = []
            for mode in (capi.ACCEL_GRID,) + modes:
                t.set_accel_mode(mode)
                t.target_clear()
                t.reset_stats()
                ids, ts = t.readback_primary()
                t.dispatch(1, args.spp, 1)
                t.finish()
                out.append((ids, ts, t.readback(), t.stats()["rays"]))
            for other in out[1:]:
                bad = int((other[0] != out[0][0]).sum()) + int((other[1].view(np.uint32) != out[0][1].view(np.uint32)).sum()) + \
                    int((other[2].view(np.uint32) != out[0][2].view(np.uint32)).sum()) + int(other[3] != out[0][3])
                total_bad += bad
                if bad:
                    print(json.dumps({"workload": workload, "camera": k, "eye": [float(x) for x in eye], "differing_words": bad}), flush=True)
            total_rays += out[0][3]
        t.close()
        print(json.dumps({"workload": workload, "cameras": args.cameras, "differing_words_so_far": total_bad, "rays_so_far": int(total_rays)}), flush=True)
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
