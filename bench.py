#!/usr/bin/env python3
"""Throughput of the hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one pass of the hot path over one batch: `--spp` progressive samples of every pixel of the workload's frame
(default: cornell_box.obj at 1920x1080, 256 spp -- the configuration the metric is quoted on), rendered by ONE kernel
launch with the scene and the accumulation image resident in HBM. With N > 1 every rank renders its own residue
class of the sample indices (weak scaling: per-GPU work is fixed) into a running sum, and the sums are merged by one
RCCL all-reduce per step inside the timed region; the exchange of a step runs beside the rendering of the next one, and
the last step's exchange is waited for before the clock stops.

Prints one JSON line on rank 0: whole-job Mrays/s (ray = one closest-hit query = one bounce-loop iteration, the
reference's definition), the roofline of the render kernel and a CPU baseline from the oracle on the host cores.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# fp32 vector peak of the part (MI355X_MICROARCH.md): 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz x 2 flop (FMA).
PEAK_FP32_VALU_TFLOPS = 157.3
# The numerics contract forbids fusing a*b+c (the reference's software GL does not fuse), so a multiply and an add
# are two instructions: the ceiling for THIS arithmetic is half the FMA peak.
PEAK_FP32_UNFUSED_TFLOPS = PEAK_FP32_VALU_TFLOPS / 2
# Algorithmic work of one ray-triangle test as the shader writes it (pathtracing.fs:136-151): 43 flop, mul and add
# counted separately; flop per ray = 43 x triangles.
FLOP_PER_TRI_TEST = 43


def cpu_baseline(scene, spp_hint, budget_s=25.0):
    """The oracle (CPU restatement of the reference shader) on the same scene/camera/resolution, on a bounded sample:
    the full frame for as many passes as fit the time budget."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pt_oracle as po
    threads = min(16, os.cpu_count() or 1)
    sc = po.Scene(scene.vertices, scene.indices, scene.spheres, scene.materials)
    u = po.uniforms(scene.camera, scene.width, scene.height, scene.num_triangles, 10,
                    po.VARIANT_MESH if scene.program == 1 else po.VARIANT_WORLD, scene.normal_mode)
    # calibrate on a strip, then run whole-frame passes within the budget
    strip = max(8, scene.height // 16)
    t0 = time.perf_counter()
    _, rays = po.render(sc, u, 1, 1, window=(0, (scene.height - strip) // 2, scene.width, strip), threads=threads)
    dt = time.perf_counter() - t0
    per_pass = dt * scene.height / strip
    passes = int(max(1, min(spp_hint, budget_s / max(per_pass, 1e-6))))
    t0 = time.perf_counter()
    _, rays = po.render(sc, u, 1, passes, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": rays / dt * 1e-6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{scene.name} {scene.width}x{scene.height}, passes 1..{passes} of the full frame, "
                      f"{rays} rays in {dt:.2f} s (oracle/pt_oracle.c, {threads} threads)"}


def trace(msg):
    """Progress lines on stderr when PT_BENCH_TRACE=1 (finding where a multi-process run stops)."""
    if os.environ.get("PT_BENCH_TRACE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cornell")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per step and per GPU")
    ap.add_argument("--partition", default="sample", choices=["sample", "tile"],
                    help="with --gpus N > 1: 'sample' = every GPU renders its own --spp samples of the whole frame (weak scaling, "
                         "all-reduce of sums); 'tile' = the GPUs share the rows of ONE frame of --spp samples in interleaved blocks "
                         "(strong scaling, bit-identical to one GPU, merged by an exact sum)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-scene", action="store_true", help="skip the short 1M-triangle run reported under 'large_scene'")
    ap.add_argument("--lean", action="store_true", help="warm-up and timed steps only (no CPU baseline, no 1M-triangle run, no "
                    "no-reuse comparison): what to put behind rocprofv3 so that the kernel statistics cover exactly these launches")
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline = args.no_second_scene = True

    import torch
    import torch.distributed as dist
    from pathtracer_amd import Tracer, capi, dist as ptdist
    from pathtracer_amd.scenes import SceneData

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # the CPU baseline, the 1M-triangle run and the no-reuse comparison are reported by the single-GPU run only
        # (they happen after the timed region on rank 0 and would keep the other ranks waiting at the final barrier)
        args.no_cpu_baseline = args.no_second_scene = args.lean = True
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    # Rehearsal knobs for a one-GPU box (never set by the driver): PT_BENCH_SHARE_GPU=1 puts every rank on device 0 and
    # PT_BENCH_BACKEND=gloo replaces RCCL, which refuses two ranks on one device.
    if os.environ.get("PT_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    backend = os.environ.get("PT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # Every step renders a frame from scratch: the camera-ray hits the library would keep from the previous dispatch of the
    # same view are not carried from step to step (one query in about 400 per pixel and step, but a step stays self-contained).
    os.environ.setdefault("PT_PRIMARY_CACHE", "0")
    trace("process group up")
    scene = SceneData(args.workload, args.width, args.height)
    W, H, spp = scene.width, scene.height, args.spp
    tracer = Tracer(local_rank)
    stream = torch.cuda.Stream(device=device)
    tracer.set_stream(stream.cuda_stream)
    scene.upload(tracer)
    # the accumulation image lives in a torch tensor so that RCCL can reduce it in place
    image = torch.zeros((H, W, 4), dtype=torch.float32, device=device)
    tracer.target_bind_external(image.data_ptr(), image.data_ptr(), W, H)
    sample_split = world > 1 and args.partition == "sample"
    tile_split = world > 1 and args.partition == "tile"
    tracer.set_accum_mode(capi.ACCUM_SUM if sample_split else capi.ACCUM_MEAN)
    if tile_split:
        first, stride, count = 1, 1, spp
        tracer.set_row_interleave(16, rank, world)
    else:
        first, stride, count = ptdist.sample_indices(rank, world, spp * world)
    tracer.enable_timing(True)
    # the exchange of a frame's sums runs beside the rendering of the next frame (PT_BENCH_OVERLAP=0: in line, as one step)
    merger = None
    if sample_split and os.environ.get("PT_BENCH_OVERLAP", "1") != "0":
        merger = ptdist.OverlappedSampleMerge(image, spp * world)

    def step():
        with torch.cuda.stream(stream):
            image.zero_()
            if not sample_split and not tile_split:
                image[..., 3] = 1.0
            tracer.dispatch(first, count, stride)
            if merger is not None:
                merger.submit(image)
            elif sample_split:
                ptdist.merge_sample_split(image, spp * world)
            elif tile_split:
                ptdist.merge_tile_split(image, H)   # rows of other ranks are zero here: the sum moves bytes, nothing else

    def fence():
        if merger is not None:
            with torch.cuda.stream(stream):
                merger.finish()   # the last frame's exchange belongs to the timed region
        stream.synchronize()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    trace("scene uploaded")
    for _ in range(args.warmup):
        step()
    trace("warm-up issued")
    fence()
    trace("warm-up done")
    tracer.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    trace("timed steps done")
    st = tracer.stats()

    # the same steps with the camera ray evaluated for every sample, as the shader does (outside the timed region, for the record)
    plain, plain_elapsed = None, 0.0
    if not args.lean:
        tracer.set_primary_reuse(False)
        for _ in range(2):
            step()
        fence()
        tracer.reset_stats()
        t1 = time.perf_counter()
        for _ in range(3):
            step()
        fence()
        plain_elapsed = (time.perf_counter() - t1) / 3
        plain = tracer.stats()
        tracer.set_primary_reuse(True)

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    counts = torch.tensor([float(st["rays"]), float(st["samples"])], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays, samples = float(counts[0].item()), float(counts[1].item())

    # second half of the headline metric: the ~1M-triangle scene through the LBVH (short run, rank 0, outside the timed region)
    large = None
    if rank == 0 and not args.no_second_scene and args.workload != "instanced":
        big = SceneData("instanced")
        t2 = Tracer(local_rank)
        big.upload(t2)
        t2.target_create(big.width, big.height)
        t2.enable_timing(True)
        big_spp = 256               # per dispatch; the configuration's 4096 spp would be sixteen of these
        for _ in range(2):          # warm-up: from the second dispatch on the work list rests on measured tile costs
            t2.target_clear()
            t2.dispatch(1, big_spp, 1)
        t2.finish()
        t2.reset_stats()
        for _ in range(8):          # includes the scheduler runs that fall into this stretch (the list is rebuilt now and then)
            t2.target_clear()
            t2.dispatch(1, big_spp, 1)
        t2.finish()
        s2 = t2.stats()
        s2 = dict(s2, last_dispatch_ms=s2["total_dispatch_ms"] / s2["dispatches"], rays=s2["rays"] / s2["dispatches"],
                  rays_traced=s2["rays_traced"] / s2["dispatches"],
                  samples=s2["samples"] / s2["dispatches"], lane_slots=s2["lane_slots"] / s2["dispatches"],
                  bvh_node_fetches=s2["bvh_node_fetches"] / s2["dispatches"], bvh_triangle_fetches=s2["bvh_triangle_fetches"] / s2["dispatches"])
        large = {"workload": f"instanced suzanne x254, {big.num_triangles} triangles, {big.width}x{big.height}, {big_spp} spp per dispatch, direction grid (PT_ACCEL_AUTO; mean of 8 dispatches)",
                 "Mrays_per_s": s2["rays"] / s2["last_dispatch_ms"] * 1e-3, "traced_Mrays_per_s": s2["rays_traced"] / s2["last_dispatch_ms"] * 1e-3,
                 "kernel_ms": s2["last_dispatch_ms"],
                 "rays_per_sample": s2["rays"] / s2["samples"],
                 "triangle_tests_per_evaluated_query": s2["grid_triangle_tests"] / max(1, s2["dispatches"]) / max(1, s2["rays_traced"]),
                 "grid_list_entries": s2["grid_entries"],
                 "gathered_GB_per_s": (s2["grid_triangle_tests"] / max(1, s2["dispatches"]) * 68 + s2["rays_traced"] * 8) / (s2["last_dispatch_ms"] * 1e-3) * 1e-9,
                 "brute_force_equivalent_TFLOP_per_s": s2["rays_traced"] * FLOP_PER_TRI_TEST * big.num_triangles / (s2["last_dispatch_ms"] * 1e-3) * 1e-12}
        t2.close()

    if rank == 0:
        kernel_ms = st["total_dispatch_ms"] / max(1, st["dispatches"])
        # the roofline prices the closest-hit queries the kernel evaluated (reused camera-ray answers cost nothing)
        rays_per_launch = st["rays_traced"] / max(1, st["dispatches"])
        flop_per_launch = rays_per_launch * FLOP_PER_TRI_TEST * scene.num_triangles
        achieved = flop_per_launch / (kernel_ms * 1e-3) * 1e-12 if kernel_ms > 0 else 0.0
        out = {
            "metric": "Mrays/s (whole node) at 1080p Cornell box + 1M-tri scene, 1/2/4/8 MI355X",
            "value": rays / elapsed * 1e-6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if tile_split else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": {"cornell": "synthetic camera on the reference cornell_box.obj", "instanced": "synthetic: 254 jittered instances of the reference suzanne.obj (fixed LCG seed)"}.get(args.workload, "reference OBJ asset, reference camera"),
            "config": {"workload": f"{scene.name} {W}x{H}, {spp} spp per {'frame (one frame per step, rows shared by the GPUs)' if tile_split else 'GPU per step'}, depth 10, " +
                                   (f"{scene.num_triangles} triangles, closest hit through the direction grid (PT_ACCEL_AUTO)"
                                    if scene.program == 1 else "the reference's 10 spheres (IntersectWorld)"),
                       "parallelism": "single GPU" if world == 1 else (f"image rows in interleaved blocks of 16 over {world} GPUs, merged by an exact sum (RCCL all-reduce)"
                                                                           if tile_split else f"sample split over {world} GPUs + RCCL all-reduce of the sums" +
                                                                           (", overlapped with the next step's rendering" if merger is not None else ""))},
            "rays_definition": "closest-hit queries as the reference counts them: one per bounce-loop iteration of every sample "
                               "(PathTracer.cpp:947). The shader has no pixel jitter, so the first query of every sample of a pixel "
                               "is the same; it is evaluated once per dispatch and pixel and reused (bit-identical image). "
                               "'rays_traced_per_s' counts only evaluated queries; 'no_primary_reuse' is the same step without the reuse.",
            "primary_reuse": True,
            "rays_traced_per_s": st["rays_traced"] * world / elapsed if world == 1 else None,
            "no_primary_reuse": None if plain is None else
                                {"value": plain["rays"] / 3 / plain_elapsed * 1e-6 * world, "unit": "Mrays/s", "ms_per_step": plain_elapsed * 1e3,
                                 "kernel_ms": plain["total_dispatch_ms"] / max(1, plain["dispatches"]), "note": "rank 0 clock, 3 steps after 2 warm-up steps"},
            "samples_per_s": samples / elapsed,
            "rays_per_sample": rays / samples if samples else 0.0,
            "roofline": {
                "bound": "valu",
                "achieved": achieved,
                "peak": PEAK_FP32_VALU_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_VALU_TFLOPS,
                "peak_unfused": PEAK_FP32_UNFUSED_TFLOPS,
                "frac_unfused": achieved / PEAK_FP32_UNFUSED_TFLOPS,
                "traffic": None,
                "kernel": "pt_render_kernel<4,false> (direction grid)" if scene.program == 1 else "pt_render_kernel<0,false> (spheres)",
                "kernel_ms": kernel_ms,
                "algorithmic_flop_per_launch": flop_per_launch,
                "note": "algorithmic flop = evaluated closest-hit queries x 43 x triangles, the shader's formulation (mul and add "
                        "counted separately; FMA is not permitted by the numerics contract); shading and RNG work is not counted. "
                        "The kernel executes far fewer: a query tests only the triangles listed in the cube-map cell of its direction "
                        "(1-2 on average), so this algorithmic rate can exceed the machine's peak; what the kernel is bound by is "
                        "VALU issue of the shading arithmetic (valu_issue_busy, from the PMC profile).",
            },
        }
        if st["bvh_node_fetches"] > 0:
            # LBVH traversal: the dominant cost is fetching node and triangle records (64 B each, once per wave);
            # they come from L2 / Infinity Cache, priced here against the HBM peak as the outermost bound
            fetched = (st["bvh_node_fetches"] * 64 + st["bvh_triangle_fetches"] * 64) / max(1, st["dispatches"])
            gbs = fetched / (kernel_ms * 1e-3) * 1e-9
            out["config"]["workload"] = out["config"]["workload"].replace("closest hit through the direction grid (PT_ACCEL_AUTO)", "device LBVH, wave-coherent traversal")
            out["roofline"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None,
                               "kernel": "pt_render_kernel<bvh>", "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": fetched,
                               "brute_force_equivalent_TFLOP_per_s": achieved,
                               "note": "bytes = (node records + triangle records) x 64, counted in-kernel once per wave fetch; "
                                       "latency-bound pointer chasing served by L2/Infinity Cache, far below the HBM roof"}
        if large:
            out["large_scene"] = large
        # HBM bytes per launch from rocprofv3 PMC passes of this same workload (profiles/r01_pmc_cornell_grid_v6.json:
        # FETCH_SIZE 20762 KB doubled as the gfx950 correction prescribes, plus WRITE_SIZE 158514 KB, of which 33 MB are
        # the image and the rest the parked samples of split tiles); counters cannot be read from inside this run
        if args.workload == "cornell" and (W, H, spp) == (1920, 1080, 256):
            out["roofline"]["traffic"] = 2 * 20762.0 * 1024 + 158514.03 * 1024
            out["roofline"]["valu_issue_busy"] = 0.74
            out["roofline"]["hbm_GB_per_s"] = out["roofline"]["traffic"] / (kernel_ms * 1e-3) * 1e-9   # against the 8000 GB/s peak: the path is not HBM-bound
            out["roofline"]["hbm_frac"] = out["roofline"]["hbm_GB_per_s"] / 8000.0
        # the reference shader itself on Mesa llvmpipe, same scene, camera and passes: recorded in the build container (the
        # shader text does not travel to the GPU box), see oracle/time_c1.py --c2
        gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "c2_cornell_1080p.json")
        if args.workload == "cornell" and (W, H, spp) == (1920, 1080, 256) and os.path.exists(gold):
            lp = json.load(open(gold))["llvmpipe"]
            out["llvmpipe_recorded"] = {"Mrays_per_s": lp["Mrays_per_s"], "ms_per_pass_median": lp["pass_ms_median"], "cores": lp["threads"],
                                        "renderer": lp["renderer"], "note": "reference pathtracing.fs, passes 1..256 of this workload, timed around glDrawArrays..glFinish "
                                        "in the build container, not on this machine; its final image is bit-identical to this run's (tests/golden/c2_cornell_1080p.json)"}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, spp)
        print(json.dumps(out), flush=True)

    tracer.close()
    if world > 1:
        dist.barrier()   # rank 0 reports (CPU baseline, large scene) after the timed region: the ranks leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
