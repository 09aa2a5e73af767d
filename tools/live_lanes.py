#!/usr/bin/env python3
"""Direction-grid kernel: where the lane slots of the wave trips go. Needs the counting build
(python tools/ab_bench.py --build --only live). Prints, per workload and form of the loop, the shares of
  query     lane slots that evaluated a closest-hit query,
  no_query  lanes still in the loop without a query in that trip (their sample ended in Shade(), or, second form, they carry a list),
  finished  lanes whose pixel has no samples left while other lanes of the wave go on (the tail inside a tile)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    lib = os.path.join(ROOT, "pathtracer_amd", "lib", "libpt_hip__live.so")
    for workload, spp in (("cornell", 256), ("instanced", 256), ("suzanne", 256), ("monkeyhead2", 256)):
        for carry in ("0", "1"):
            os.environ["PT_GRID_CARRY"] = carry
            scene = SceneData(workload, None, None)
            t = Tracer(0, lib)
            scene.upload(t, accel=True, shading="reference")
            t.target_create(scene.width, scene.height)
            t.enable_timing(True)
            for _ in range(3):
                t.target_clear()
                t.reset_stats()
                t.dispatch(1, spp, 1)
                t.finish()
            st = t.stats()
            slots, live, traced = st["lane_slots"], st["bvh_node_fetches"], st["rays_traced"]
            print(json.dumps({"workload": workload, "spp": spp, "lists_carried": int(carry), "ms": round(st["last_dispatch_ms"], 3),
                              "query": round(traced / slots, 4), "no_query": round((live - traced) / slots, 4), "finished": round((slots - live) / slots, 4)}))
            del t


if __name__ == "__main__":
    main()
