#!/usr/bin/env python3
"""Sparse views: the gathered hit pixels (PT_PIXEL_LIST=1) against the window's 8x8 blocks (=0), one process, dispatches interleaved.
    python tools/pixel_list_ab.py [--shading materials]"""
import argparse
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shading", default="reference")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--configs", default="instanced:256,suzanne:256,monkeyhead2:256,torus:64,cornell:256")
    args = ap.parse_args()
    import numpy as np
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    for cfg in args.configs.split(","):
        workload, spp = cfg.split(":")
        spp = int(spp)
        scene = SceneData(workload, None, None)
        tracers = {}
        for mode in ("0", "1"):
            os.environ["PT_PIXEL_LIST"] = mode
            t = Tracer(0)
            scene.upload(t, accel=True, shading=args.shading)
            t.target_create(scene.width, scene.height)
            t.enable_timing(True)
            for _ in range(4):
                t.target_clear()
                t.dispatch(1, spp, 1)
            t.finish()
            tracers[mode] = t
        same = bool(np.array_equal(tracers["0"].readback().view(np.uint32), tracers["1"].readback().view(np.uint32)))
        times = {"0": [], "1": []}
        util = {}
        for _ in range(args.rounds):
            for mode, t in tracers.items():
                t.target_clear()
                t.reset_stats()
                t.dispatch(1, spp, 1)
                t.finish()
                st = t.stats()
                times[mode].append(st["last_dispatch_ms"])
                util[mode] = st["rays_traced"] / max(1, st["lane_slots"])
                if mode == "1":
                    hits = st["gathered_pixels"]
        print(json.dumps({"workload": workload, "spp": spp, "shading": args.shading, "blocks_ms": round(statistics.median(times["0"]), 3),
                          "gathered_ms": round(statistics.median(times["1"]), 3), "lane_utilisation": [round(util["0"], 4), round(util["1"], 4)],
                          "pixels_that_see_the_mesh": round(hits / (scene.width * scene.height), 4), "same_image": same}))
        for t in tracers.values():
            t.close()


if __name__ == "__main__":
    main()
