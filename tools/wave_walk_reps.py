#!/usr/bin/env python3
"""The per-wave hierarchy walk of the reference shading (PT_ACCEL_BVH): time of four dispatches in a row and the nodes fetched, per
hierarchy builder. usage: wave_walk_reps.py <workload> <samples> <builders, comma separated>   (PT_SPLIT_FACTOR=f to sweep the split)"""
import os, sys, json
sys.path.insert(0, os.getcwd())
from pathtracer_amd import Tracer, capi
from pathtracer_amd.scenes import SceneData
wl, spp = sys.argv[1], int(sys.argv[2])
scene = SceneData(wl, None, None)
for builder in sys.argv[3].split(","):
    os.environ["PT_BVH_BUILDER"] = builder
    t = Tracer(0)
    with t:
        scene.upload(t, shading="reference", accel=False)
        t.target_create(scene.width, scene.height)
        t.build_accel()
        t.set_accel_mode(capi.ACCEL_BVH)
        t.enable_timing(True)
        t.dispatch(1, 1, 1); t.dispatch(2, 1, 1); t.finish()
        ms = []
        for rep in range(4):
            t.reset_stats()
            t.target_clear()
            t.dispatch(1, spp, 1)
            t.finish()
            st = t.stats()
            ms.append(round(st["last_dispatch_ms"], 2))
        print(wl, builder, os.environ.get("PT_SPLIT_FACTOR"), ms, st["bvh_node_fetches"], flush=True)
