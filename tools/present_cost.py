#!/usr/bin/env python3
"""What the present pass costs: host wall time of pt_present_rgba8 (copy.fs + 8-bit conversion on the device, then the
copy of the 8-bit image to host memory) and of a whole reference-style frame: one sample, finish, present."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cornell")
    args = ap.parse_args()
    import torch   # (before the library touches the device: torch finds no GPU when it initialises second)
    torch.cuda.init()
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    scene = SceneData(args.workload)
    t = Tracer(0)
    scene.upload(t)
    t.target_create(scene.width, scene.height)
    for s in range(1, 4):
        t.dispatch(s, 1, 1)
    t.finish()
    t.present()
    present = []
    for _ in range(20):
        t0 = time.perf_counter()
        t.present()
        present.append((time.perf_counter() - t0) * 1e3)
    frame = []
    for s in range(4, 24):
        t0 = time.perf_counter()
        t.dispatch(s, 1, 1)
        t.finish()
        t.present()
        frame.append((time.perf_counter() - t0) * 1e3)
    # the same into page-locked host memory (a torch pinned tensor): the copy then runs at the link's speed
    from pathtracer_amd.capi import check
    pinned = torch.empty((scene.height, scene.width, 4), dtype=torch.uint8, pin_memory=True)
    check(t._lib.pt_present_rgba8(t._ctx, pinned.data_ptr()), t._lib)
    present_pinned = []
    for _ in range(20):
        t0 = time.perf_counter()
        check(t._lib.pt_present_rgba8(t._ctx, pinned.data_ptr()), t._lib)
        present_pinned.append((time.perf_counter() - t0) * 1e3)
    same = bool((pinned.numpy() == t.present()).all())
    present.sort()
    frame.sort()
    present_pinned.sort()
    print(json.dumps({"workload": args.workload, "size": [scene.width, scene.height], "present_ms_median": round(present[10], 3),
                      "present_into_pinned_memory_ms_median": round(present_pinned[10], 3), "pinned_equals_pageable": same,
                      "frame_with_present_ms_median": round(frame[10], 3), "frames_per_s": round(1e3 / frame[10])}))
    t.close()


if __name__ == "__main__":
    main()
