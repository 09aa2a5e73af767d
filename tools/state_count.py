#!/usr/bin/env python3
"""Lane-state shares of pt_render_decoupled_kernel from the measurement build (-DPT_V_STATE_COUNT, which reports its counts in
the statistics fields the kernel does not otherwise use). usage: state_count.py --lib <variant .so> --workload W --spp N"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="instanced")
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--lib", required=True)
    a = ap.parse_args()
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(a.workload)
    t = Tracer(0, a.lib)
    scene.upload(t, shading="materials")
    t.target_create(scene.width, scene.height)
    t.enable_timing(True)
    t.dispatch(1, 1, 1)
    t.dispatch(2, a.spp, 1)
    t.finish()
    t.reset_stats()
    t.target_clear()
    t.dispatch(1, a.spp, 1)
    t.finish()
    s = t.stats()
    trips = s["bvh_node_fetches"]
    slots = trips * 64
    out = {"workload": a.workload, "spp": a.spp, "ms": s["last_dispatch_ms"], "wave_loop_trips": trips,
           "walking_share": s["bvh_triangle_fetches"] / slots, "waiting_share": s["loop_triangle_visits"] / slots,
           "done_share": 1 - (s["bvh_triangle_fetches"] + s["loop_triangle_visits"]) / slots,
           "lanes_per_shade_batch": s["loop_full_tests"] / max(1, s["lane_slots"] / 64),
           "shade_batches_per_trip": (s["lane_slots"] / 64) / trips,
           "node_and_record_steps_per_walking_lane_trip": s["grid_triangle_tests"] / max(1, s["bvh_triangle_fetches"]),
           "rays_traced": s["rays_traced"], "shaded": s["shaded_hits"]}
    print(json.dumps(out))
    t.close()


if __name__ == "__main__":
    main()
