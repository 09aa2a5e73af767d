#!/usr/bin/env python3
"""The direction-grid kernel's two loop forms side by side (PT_GRID_CARRY=0: the wave runs its list loop; =1: list remainders are carried
into the next trip; unset: the library decides from the view's statistics): ms per dispatch, interleaved rounds, images compared.
usage: grid_carry_ab.py <workload> <samples> [rounds]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    wl, spp = sys.argv[1], int(sys.argv[2])
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    import numpy as np
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(wl)
    tracers = {}
    for name, env in (("loop", "0"), ("carry", "1"), ("auto", None)):
        if env is None:
            os.environ.pop("PT_GRID_CARRY", None)
        else:
            os.environ["PT_GRID_CARRY"] = env
        t = Tracer(0)
        scene.upload(t)
        t.target_create(scene.width, scene.height)
        t.enable_timing(True)
        t.dispatch(1, 2, 1)
        t.dispatch(3, spp, 1)
        t.finish()
        tracers[name] = t
    os.environ.pop("PT_GRID_CARRY", None)
    times = {k: [] for k in tracers}
    images = {}
    for _ in range(rounds):
        for name, t in tracers.items():
            t.target_clear()
            t.reset_stats()
            t.dispatch(1, spp, 1)
            t.finish()
            st = t.stats()
            times[name].append(st["last_dispatch_ms"])
            images[name] = (t.readback().copy(), st["rays"], st["rays_traced"])
    ref = images["loop"]
    for name in tracers:
        same = bool(np.array_equal(images[name][0].view(np.uint32), ref[0].view(np.uint32))) and images[name][1:] == ref[1:]
        print(json.dumps({"workload": wl, "spp": spp, "form": name, "median_ms": round(statistics.median(times[name]), 3), "min_ms": round(min(times[name]), 3),
                          "same_image_and_ray_counts": same}))


if __name__ == "__main__":
    main()
