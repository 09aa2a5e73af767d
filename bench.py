#!/usr/bin/env python3
"""Throughput of the hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only starts N children (one rank per GPU, RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set), relays rank 0's JSON line and exits non-zero if any child does; it never
touches a GPU itself. Under torch.distributed.run (WORLD_SIZE already set) it is one of the ranks.

A step = one pass of the hot path over one batch: `--spp` progressive samples of every pixel of the workload's frame
(default: cornell_box.obj at 1920x1080, 256 spp -- the configuration the metric is quoted on), rendered by ONE dispatch
with the scene and the accumulation image resident in HBM. With N > 1, `--partition sample` (default): every rank renders
its own residue class of the sample indices (weak scaling: per-GPU work is fixed) into a running sum, and the sums are
merged by one RCCL all-reduce per step inside the timed region, beside the rendering of the next step; `--partition tile`:
the ranks share the rows of ONE frame in interleaved blocks (strong scaling, bit-identical to one GPU), merged by an
all-gather of the owned rows, also beside the next step.

Prints one JSON line on rank 0: whole-job Mrays/s (ray = one closest-hit query actually evaluated; the count by the
reference's definition, one per bounce-loop iteration with reused camera-ray answers included, is reported beside it), the roofline of the render kernel from the work it executed, and a CPU baseline from the oracle.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# fp32 vector peak of the part (MI355X_MICROARCH.md): 256 CUs x 4 SIMD x 32 lanes x 2.4 GHz x 2 flop (FMA).
PEAK_FP32_FMA_TFLOPS = 157.3
# The numerics contract forbids fusing a*b+c (the reference's software GL does not fuse), so a multiply and an add are
# two instructions: the ceiling for THIS arithmetic is one fp32 operation per lane and issue slot, half the FMA peak.
PEAK_FP32_UNFUSED_TFLOPS = PEAK_FP32_FMA_TFLOPS / 2
# Algorithmic work of one ray-triangle test as the shader writes it (pathtracing.fs:136-151): 43 flop, mul and add
# counted separately; the shader's loop costs 43 x triangles per ray. Reported as `brute_force_equivalent`, never as `frac`.
FLOP_PER_TRI_TEST_SHADER = 43

# fp32 operations per event AS EXECUTED by the kernels, counted at source level in pt_kernels.hip (add, sub, mul, floor,
# min, max, division, reciprocal and square root one each, a fused multiply-add two; compares, selects, conversions, moves,
# integer and address arithmetic are not counted, and a division or square root counts as ONE although it issues 4-9
# instructions). Events come from the kernel's own counters (PtStats).
FLOP = {
    "rand01": 13,           # Rng::rand01: 4 add, 3 (2 mul + add), 2 fract (floor + sub), 1 mul, add (the shader's final mod(r, 1.0) is the identity on its argument and is not executed)
    "normalize": 10,        # dot3 (3 mul + 2 add), sqrt, reciprocal, 3 mul
    "cross": 9,
    "reflect": 12,          # dot3, 2*, 3 mul, 3 sub
    "pow": 41,              # lp_log2 25 (quotient 12, polynomial 13) + mul + lp_exp2 15 (lower clamp only: the argument is <= 0)
    "sincos_pair": 32,      # lp_sincos<true> and <false> of one angle: shared reduction and both polynomials 28, two selections 4
    "exp2": 14,             # without clamps (the Fresnel exponent lies in [-8.66, 0])
    # shade(): reflect 12, two rand01 26, tangent frame 2 x (cross 9 + normalize 10) 38, pow 41, sin_t 4, phi 2, sincos 32,
    # hemi 2, new direction 15, dot(N, rd) 5, clamps 4, Fresnel exp2 14 + 4, specular 3, throughput 2  -> 204
    "shade_reference": 204,
    # shade_materials(), Lambert branch: hit point 6, two rand01 26, tangent frame 38, pow 41, sin_t 4, phi 2, sincos 32, hemi 2,
    # new direction 15, new origin 6, throughput 3 -> 175 (the emitter branch is 6)
    "shade_materials": 175,
    "vertex_normal": 29,    # mesh_normal(): barycentrics 4, blend 15, normalize 10 (not for geometric normals)
    "sample_end": 21,       # sky 12 (fs:408-410), running mean 9 (fs:414)
    "grid_query": 9,        # reciprocal of the major axis, 2 x (mul, add, mul, floor)
    "test_origin_records": 25,  # test against an origin-relative record: cross 9, det 5, u 5, v 5, u + v 1
    "test_accept": 2,       # 1/det, t
    "test_from_first": 14,  # rays with their own origin: cross 9, det 5 (always) ...
    "test_from_rest": 23,   # ... ap 3, cross 9, u 5, v 5, u + v 1 (past the wave-uniform back-face skip)
    "loop_first": 14,       # all-triangles loop over origin-relative records: cross 9 + det 5 (always) ...
    "loop_rest": 11,        # ... u 5, v 5, u + v 1 (past the skip)
    "spheres_query": 70,    # 10 x (dot3 5, B*B - C 2)
    "sphere_accept": 15,    # sqrt, two roots, hit point 6, normal 6
}

# Where a gather-bound walk's records come from, by their footprint (MI355X_MICROARCH.md, "Indexed rows" and "L2"):
# random rows out of the XCD's L2, out of the Infinity Cache, out of HBM (GB/s chip-wide).
GATHER_PEAK = (("l2", 4 << 20, 16800.0), ("infinity_cache", 256 << 20, 8600.0), ("hbm", 1 << 62, 8000.0))


def code_only(text):
    """C/C++ source without its comments and layout: string literals stay as they are, // and /* */ comments go, runs of white
    space count as one blank."""
    import re
    text = re.sub(r'"(?:\\.|[^"\\\n])*"|//[^\n]*|/\*.*?\*/', lambda m: m.group(0) if m.group(0).startswith('"') else " ", text, flags=re.S)
    return " ".join(text.split())


def kernel_source_hash():
    """Identifies the kernels a profile was taken from: the code of the sources (comments and layout left out) + build flags."""
    h = hashlib.sha256()
    for rel in ("pathtracer_amd/csrc/pt_kernels.hip", "pathtracer_amd/csrc/pt_internal.h", "pathtracer_amd/csrc/pt_grid.hip",
                "pathtracer_amd/csrc/pt_bvh.hip", "pathtracer_amd/csrc/pt_capi.hip"):   # (the last: grid resolution, work list, launch sizes)
        h.update(code_only(open(os.path.join(ROOT, rel), "r", encoding="utf-8", errors="replace").read()).encode())
    from pathtracer_amd import build as ptbuild
    h.update(" ".join(ptbuild.DEVICE_FLAGS).encode())
    return h.hexdigest()[:16]


PMC_SIDECAR = os.path.join(ROOT, "profiles", "r03_pmc.json")


def config_key(workload, width, height, spp, shading, accel):
    return f"{workload}|{width}x{height}|{spp}spp|{shading}|{accel}"


def pmc_entry(key):
    """Counter-derived figures for this exact configuration from the tracked profile written by tools/pmc_collect.py, or
    None when there is none or it was taken from other kernel sources."""
    try:
        doc = json.load(open(PMC_SIDECAR))
    except (OSError, ValueError):
        return None, "no profiles/r03_pmc.json"
    e = doc.get("entries", {}).get(key)
    if e is None:
        return None, "no PMC profile of this configuration"
    if e.get("kernel_hash") != kernel_source_hash():
        return None, f"PMC profile is stale (taken from kernel sources {e.get('kernel_hash')}, now {kernel_source_hash()})"
    return e, None


def gather_probe_rates():
    """Record rates (G records/s, chip-wide) of tools/gather_probe.hip on MI355X -- every lane chasing its own chain of 64-byte
    records with 16-byte loads at 20 waves per CU, 64 lanes active: from a footprint that stays in the L2s, and from one that
    does not -- out of the tracked result file, or None."""
    try:
        rows = json.load(open(os.path.join(ROOT, "profiles", "r02_gather_probe.json")))["rows"]
    except (OSError, ValueError, KeyError):
        return None
    pick = {r["footprint_MiB"]: r["Grecords_per_s"] for r in rows
            if r["bytes_per_record"] == 64 and r["waves_per_cu"] == 20 and r["active_lanes"] == 64}
    if 2 not in pick or 128 not in pick:
        return None
    return {"l2_resident": pick[2], "beyond_l2": pick[128]}


def which_kernel(st, program, shading):
    if program == 0:
        return "pt_render_kernel<0,false>", "spheres"
    if shading == "materials":
        if st["grid_triangle_tests"] > 0:
            return "pt_render_decoupled_kernel", "lane_walk"
        if st["bvh_node_fetches"] > 0:
            return "pt_render_kernel<2,true>", "wave_walk"
        return "pt_render_kernel<1,true>", "loop_from"
    if st["grid_triangle_tests"] > 0:   # (the two forms of the loop do the same arithmetic: one flop table)
        return ("pt_render_kernel<7,false>" if st.get("grid_lists_carried") else "pt_render_kernel<4,false>"), "grid"
    if st["bvh_node_fetches"] > 0:
        return "pt_render_kernel<2,false>", "wave_walk"
    return "pt_render_kernel<1,false>", "loop"


def roofline(st, kernel_ms, scene, shading, pmc, pmc_note):
    """The render kernel against the ceiling that bounds it, from the work it executed in the timed launches."""
    n = max(1, st["dispatches"])
    per = {k: st[k] / n for k in ("rays", "rays_traced", "samples", "lane_slots", "bvh_node_fetches", "bvh_triangle_fetches",
                                  "grid_triangle_tests", "shaded_hits", "loop_triangle_visits", "loop_full_tests", "lane_walk_records")}
    kernel, kind = which_kernel(st, scene.program, shading)
    sec = kernel_ms * 1e-3
    lane_util = per["rays_traced"] / per["lane_slots"] if per["lane_slots"] else 0.0
    out = {"kernel": kernel, "kernel_ms": kernel_ms, "lane_utilisation": lane_util,
           "events_per_launch": {k: v for k, v in per.items() if v}}
    if st.get("gathered_pixels"):   # a sparse view: the waves' tiles are 64 gathered hit pixels each, the sky pixels have a pass of their own
        out["gathered_pixels"] = st["gathered_pixels"]
    brute = per["rays_traced"] * FLOP_PER_TRI_TEST_SHADER * scene.num_triangles
    if kind in ("wave_walk", "lane_walk"):
        # dependent record fetches. Per-wave walk: 64-byte node and triangle records, one scalar fetch serves the wave; per-lane
        # walk: every lane gathers its own 64-byte four-child nodes and 48-byte triangle records.
        if kind == "lane_walk":
            recs = per["grid_triangle_tests"]
            nbytes = (recs - per["lane_walk_records"]) * 64.0 + per["lane_walk_records"] * 48.0
            footprint = scene.num_triangles * (48 + 32)   # a record per triangle, about a node per two triangles
        else:
            recs = per["bvh_node_fetches"] + per["bvh_triangle_fetches"]
            nbytes = recs * 64.0
            footprint = scene.num_triangles * 128     # node + triangle record per triangle
        level, _, peak = next(g for g in GATHER_PEAK if footprint <= g[1])
        gbs = nbytes / sec * 1e-9 if sec > 0 else 0.0
        out.update({"bound": level if level != "hbm" else "hbm", "achieved": gbs, "peak": peak, "unit": "GB/s", "frac": gbs / peak,
                    "hbm_peak": 8000.0, "frac_of_hbm_peak": gbs / 8000.0,
                    "algorithmic_bytes_per_launch": nbytes, "records_footprint_bytes": footprint,
                    "records_per_evaluated_query": recs / per["rays_traced"] if per["rays_traced"] else 0.0,
                    "note": "bytes = " + ("64-byte four-child nodes and 48-byte triangle records" if kind == "lane_walk" else "64-byte node and triangle records")
                            + " fetched by the hierarchy walk (counted in-kernel: "
                            + ("per lane" if kind == "lane_walk" else "once per wave, scalar fetches") +
                            "), priced against the random-row gather rate of the level their footprint fits "
                            "(MI355X_MICROARCH.md: L2 16.8, Infinity Cache 8.6, HBM 8.0 TB/s)"})
        probe = gather_probe_rates() if kind == "lane_walk" else None
        if probe and sec > 0:
            # the same access shape measured bare (tools/gather_probe.hip): what the memory side allows a walk that does nothing
            # but fetch. With the L2 hit rate of the profiled run the two rates blend into the ceiling for THIS mix of hits.
            rate = recs / sec * 1e-9
            dep = {"achieved_Grecords_per_s": rate, "l2_resident_Grecords_per_s": probe["l2_resident"],
                   "beyond_l2_Grecords_per_s": probe["beyond_l2"], "frac_of_l2_resident": rate / probe["l2_resident"],
                   "source": "profiles/r02_gather_probe.json (tools/gather_probe.hip: 64-byte records, 20 waves per CU, 64 lanes)"}
            h = pmc.get("l2_hit_rate") if pmc else None
            if h is not None:
                ceiling = 1.0 / (h / probe["l2_resident"] + (1.0 - h) / probe["beyond_l2"])
                dep.update({"l2_hit_rate": h, "ceiling_at_that_hit_rate_Grecords_per_s": ceiling, "frac_of_ceiling": rate / ceiling})
            out["dependent_gather"] = dep
    else:
        shade = FLOP["shade_materials"] if shading == "materials" else FLOP["shade_reference"]
        if scene.program != 0 and scene.normal_mode == 0:
            shade += FLOP["vertex_normal"]
        flop = per["shaded_hits"] * shade + per["samples"] * FLOP["sample_end"]
        if kind == "grid":
            flop += per["rays_traced"] * FLOP["grid_query"] + per["grid_triangle_tests"] * FLOP["test_origin_records"]
            flop += per["shaded_hits"] * FLOP["test_accept"]
        elif kind == "spheres":
            flop += per["rays_traced"] * FLOP["spheres_query"] + per["shaded_hits"] * FLOP["sphere_accept"]
        else:
            # records visited per wave trip x the lanes that wanted a hit (the others are masked off)
            per_lane_tests = per["loop_triangle_visits"] * 64.0 * lane_util
            first, rest = (FLOP["test_from_first"], FLOP["test_from_rest"]) if kind == "loop_from" else (FLOP["loop_first"], FLOP["loop_rest"])
            past_skip = per["loop_full_tests"] / per["loop_triangle_visits"] if per["loop_triangle_visits"] else 0.0
            out["loop_tests_past_backface_skip"] = past_skip
            if not per["loop_full_tests"]:
                # the product build does not count them (one VALU instruction per record): the tracked profile of this
                # configuration carries the share measured with the counting build, else only the part before the skip is priced
                if pmc and pmc.get("loop_tests_past_backface_skip") is not None:
                    past_skip = pmc["loop_tests_past_backface_skip"]
                    out["loop_tests_past_backface_skip"] = past_skip
                    out["loop_tests_past_backface_skip_source"] = "profiles/r03_pmc.json (counting build, same configuration)"
                else:
                    out["loop_tests_past_backface_skip"] = "not counted in the product build: only the part of a test before the skip is priced"
            flop += per_lane_tests * (first + rest * past_skip) + per["shaded_hits"] * FLOP["test_accept"]
        tf = flop / sec * 1e-12 if sec > 0 else 0.0
        out.update({"bound": "valu", "achieved": tf, "peak": PEAK_FP32_UNFUSED_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_UNFUSED_TFLOPS,
                    "peak_fma_nominal": PEAK_FP32_FMA_TFLOPS, "frac_of_fma_nominal": tf / PEAK_FP32_FMA_TFLOPS,
                    "executed_flop_per_launch": flop,
                    "note": "achieved = fp32 operations the kernel executed (events counted in-kernel x operations per event counted in "
                            "pt_kernels.hip, bench.py FLOP table; a division or square root counts as one, compares/selects/moves/integer "
                            "work as none) / kernel time; peak = the fp32 rate without fused multiply-add, which the numerics contract "
                            "forbids (half of the nominal 157.3)"})
    out["brute_force_equivalent"] = {"TFLOP_per_s": brute / sec * 1e-12 if sec > 0 else 0.0, "frac_algorithmic": brute / sec * 1e-12 / PEAK_FP32_FMA_TFLOPS if sec > 0 else 0.0,
                                     "note": "evaluated queries x 43 x triangles: what the shader's own all-triangles loop would execute for the "
                                             "same queries; not work this kernel does"}
    out["traffic"] = None
    if pmc is not None:
        out["traffic"] = pmc.get("hbm_bytes_per_launch")
        out["pmc"] = {k: pmc[k] for k in ("source", "kernel_hash", "valu_insts_per_launch", "valu_issue_share", "valu_lane_ops_frac_of_unfused_peak",
                                          "active_lane_share_of_valu", "fp32_share_of_valu_insts", "hw_fp32_flop_per_launch_active_lanes",
                                          "wait_inst_share", "wait_any_share", "l2_hit_rate", "fetch_bytes_per_launch", "write_bytes_per_launch",
                                          "kernel_ms", "kernel_ms_rocprof_stats_avg") if k in pmc}
        if pmc.get("hw_fp32_flop_per_launch_active_lanes") and pmc.get("kernel_ms") and out.get("unit") == "TFLOP/s":
            # the same quantity from the hardware's instruction counters of the profiled run (add + mul + 2 fma + transcendental
            # instructions x 64 x the share of lanes switched on)
            out["frac_from_counters"] = pmc["hw_fp32_flop_per_launch_active_lanes"] / (pmc["kernel_ms"] * 1e-3) * 1e-12 / PEAK_FP32_UNFUSED_TFLOPS
        if out["traffic"] and sec > 0:
            out["hbm_GB_per_s"] = out["traffic"] / (pmc.get("kernel_ms", kernel_ms) * 1e-3) * 1e-9
            out["hbm_frac"] = out["hbm_GB_per_s"] / 8000.0
    else:
        out["pmc"] = pmc_note
    return out


def cpu_baseline(scene, spp_hint, shading, budget_s=25.0):
    """The oracle (CPU restatement of the reference shader) on the same scene/camera/resolution, on a bounded sample:
    the full frame for as many passes as fit the time budget."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pt_oracle as po
    threads = min(16, os.cpu_count() or 1)
    if shading == "materials":
        sc = po.Scene(scene.vertices, scene.indices, scene.spheres, scene.material_table, scene.material_ids)
    else:
        sc = po.Scene(scene.vertices, scene.indices, scene.spheres, scene.materials)
    u = po.uniforms(scene.camera, scene.width, scene.height, scene.num_triangles, 10,
                    po.VARIANT_MESH if scene.program == 1 else po.VARIANT_WORLD, scene.normal_mode,
                    po.SHADE_MATERIALS if shading == "materials" else po.SHADE_REFERENCE)
    # calibrate on a strip, then run whole-frame passes within the budget (a large mesh: a strip only). The oracle tests
    # every triangle for every ray, about 2e9 tests per second on these threads: the strip is sized for about ten seconds
    strip = max(8, scene.height // 16)
    if scene.program == 1 and scene.num_triangles > 0:
        strip = int(max(1, min(strip, 2.0e10 / scene.num_triangles / 1.5 // scene.width)))
    win = (0, (scene.height - strip) // 2, scene.width, strip)
    t0 = time.perf_counter()
    _, rays = po.render(sc, u, 1, 1, window=win, threads=threads)
    dt = time.perf_counter() - t0
    per_pass = dt * scene.height / strip
    if per_pass > budget_s:
        return {"value": rays / dt * 1e-6, "unit": "Mrays/s", "cores": threads, "kind": "port",
                "sample": f"{scene.name} {scene.width}x{scene.height}, pass 1 of the middle {strip} rows, {rays} rays in {dt:.2f} s "
                          f"(oracle/pt_oracle.c, {threads} threads; a whole pass would take {per_pass:.0f} s)"}
    # whole-frame passes in chunks until about twelve seconds of CPU work are done (the strip's timing carries the start-up of
    # the threads and overestimates a pass, so the chunks are sized as they go)
    target_s, done, rays, spent = min(12.0, budget_s), 0, 0, 0.0
    chunk = int(max(1, min(spp_hint, 2.0 / max(per_pass, 1e-6))))
    while done < spp_hint and spent < target_s:
        chunk = min(chunk, spp_hint - done)
        t0 = time.perf_counter()
        _, r = po.render(sc, u, 1 + done, chunk, threads=threads)
        dt = time.perf_counter() - t0
        rays, spent, done = rays + r, spent + dt, done + chunk
        chunk = int(max(1, min(4 * chunk, (target_s - spent) / max(dt / chunk, 1e-6) + 1)))
    return {"value": rays / spent * 1e-6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{scene.name} {scene.width}x{scene.height}, passes 1..{done} of the full frame, "
                      f"{rays} rays in {spent:.2f} s (oracle/pt_oracle.c, {threads} threads)"}


def trace(msg):
    """Progress lines on stderr when PT_BENCH_TRACE=1 (finding where a multi-process run stops)."""
    if os.environ.get("PT_BENCH_TRACE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cornell")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--spp", type=int, default=256, help="samples per pixel per step and per GPU")
    ap.add_argument("--shading", default="reference", choices=["reference", "materials"],
                    help="'reference' = the shader as committed; 'materials' = the diffuse/emissive mode with the scene's MTL-derived table")
    ap.add_argument("--accel", default="auto", choices=["auto", "brute", "bvh", "grid"])
    ap.add_argument("--partition", default="sample", choices=["sample", "tile"],
                    help="with --gpus N > 1: 'sample' = every GPU renders its own --spp samples of the whole frame (weak scaling, "
                         "all-reduce of sums); 'tile' = the GPUs share the rows of ONE frame of --spp samples in interleaved blocks "
                         "(strong scaling, bit-identical to one GPU, merged by an all-gather of the owned rows)")
    ap.add_argument("--dump-image", default=None, help="rank 0 writes the last step's merged image ([H, W, 4] float32 .npy) here")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-second-scene", action="store_true", help="skip the short 1M-triangle and materials runs reported beside the headline")
    ap.add_argument("--lean", action="store_true", help="warm-up and timed steps only (no CPU baseline, no second scenes, no "
                    "no-reuse comparison): what to put behind rocprofv3 so that the kernel statistics cover exactly these launches")
    args = ap.parse_args(argv)
    if args.lean:
        args.no_cpu_baseline = args.no_second_scene = True
    return args


def launch_ranks(args):
    """--gpus N > 1 started directly: N child processes, one per GPU, before this process has touched a GPU (it never does).
    Returns the exit code: 0 only if every rank exited 0; rank 0's stdout (the JSON line) is relayed."""
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    env["WORLD_SIZE"] = str(args.gpus)
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout (one JSON line at its end) is read by a thread, so its pipe can never fill up
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    code = 0
    pending = set(range(args.gpus))
    while pending:
        for r in sorted(pending):
            rc = procs[r].poll()
            if rc is not None:
                pending.discard(r)
                if rc != 0:
                    code = code or rc
                    print(f"[bench] rank {r} exited with {rc}", file=sys.stderr, flush=True)
        if code and pending:
            # a failed rank leaves the others waiting in a collective: give them a moment, then end them
            deadline = time.time() + 5.0
            while time.time() < deadline and any(procs[r].poll() is None for r in pending):
                time.sleep(0.1)
            for r in pending:
                if procs[r].poll() is None:
                    procs[r].terminate()
            for r in pending:
                try:
                    procs[r].wait(timeout=20)
                except subprocess.TimeoutExpired:
                    procs[r].kill()
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = out0[0] if out0 else b""
    sys.stdout.write(out0.decode("utf-8", "replace"))
    sys.stdout.flush()
    return code


def side_run(local_rank, workload, shading, spp, steps, width=None, height=None):
    """A short run of another configuration on rank 0, outside the timed region: mean of `steps` dispatches after two warm-ups."""
    from pathtracer_amd import Tracer
    from pathtracer_amd.scenes import SceneData
    sc = SceneData(workload, width, height)
    t = Tracer(local_rank)
    sc.upload(t, shading=shading)
    t.target_create(sc.width, sc.height)
    t.enable_timing(True)
    for _ in range(2):          # from the second dispatch on the work list rests on measured tile costs
        t.target_clear()
        t.dispatch(1, spp, 1)
    t.finish()
    t.reset_stats()
    for _ in range(steps):      # includes the scheduler runs that fall into this stretch (the list is rebuilt now and then)
        t.target_clear()
        t.dispatch(1, spp, 1)
    t.finish()
    st = t.stats()
    t.close()
    ms = st["total_dispatch_ms"] / max(1, st["dispatches"])
    key = config_key(workload, sc.width, sc.height, spp, shading, "auto")
    pmc, note = pmc_entry(key)
    n = max(1, st["dispatches"])
    return {"workload": f"{sc.name} {sc.width}x{sc.height}, {sc.num_triangles} triangles, {spp} spp per dispatch, shading {shading} (mean of {steps} dispatches)",
            "Mrays_per_s": st["rays_traced"] / n / ms * 1e-3, "reference_definition_Mrays_per_s": st["rays"] / n / ms * 1e-3,
            "kernel_ms": ms, "rays_per_sample": st["rays"] / max(1, st["samples"]),
            "roofline": roofline(st, ms, sc, shading, pmc, note)}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    from pathtracer_amd import Tracer, capi, dist as ptdist
    from pathtracer_amd.scenes import SceneData

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        # the CPU baseline, the second scenes and the no-reuse comparison are reported by the single-GPU run only
        # (they happen after the timed region on rank 0 and would keep the other ranks waiting at the final barrier)
        args.no_cpu_baseline = args.no_second_scene = args.lean = True
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    scene.upload(tracer, shading=args.shading)
    tracer.set_accel_mode({"auto": capi.ACCEL_AUTO, "brute": capi.ACCEL_BRUTE, "bvh": capi.ACCEL_BVH, "grid": capi.ACCEL_GRID}[args.accel])
    # the accumulation image lives in a torch tensor so that RCCL can work on it in place
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
    # the exchange of a frame runs beside the rendering of the next frame (PT_BENCH_OVERLAP=0: in line, as one step)
    overlap = os.environ.get("PT_BENCH_OVERLAP", "1") != "0"
    merger = None
    if sample_split and overlap:
        merger = ptdist.OverlappedSampleMerge(image, spp * world)
    gather = ptdist.TileGather(image, rank, world, 16, overlapped=overlap) if tile_split else None
    merged = [image]

    def step():
        with torch.cuda.stream(stream):
            image.zero_()
            if not sample_split:
                image[..., 3] = 1.0
            tracer.dispatch(first, count, stride)
            if merger is not None:
                merger.submit(image)
            elif sample_split:
                ptdist.merge_sample_split(image, spp * world)
            elif tile_split:
                gather.submit(image)   # owned rows travel; nothing is added

    def fence():
        with torch.cuda.stream(stream):
            if merger is not None:
                merged[0] = merger.finish()   # the last frame's exchange belongs to the timed region
            if gather is not None:
                merged[0] = gather.finish()
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
    issue_ms = []   # how long each step's calls held the host (a dispatch that rebuilds its work list reads two counters back)
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        issue_ms.append((time.perf_counter() - ts) * 1e3)
    fence()
    elapsed = time.perf_counter() - t0
    trace("timed steps done")
    st = tracer.stats()
    if args.dump_image and rank == 0:
        np.save(args.dump_image, merged[0].detach().cpu().numpy())

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
    counts = torch.tensor([float(st["rays"]), float(st["samples"]), float(st["rays_traced"])], dtype=torch.float64, device=device)
    if world > 1:
        if backend != "nccl":
            t, counts = t.cpu(), counts.cpu()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    rays, samples, traced = float(counts[0].item()), float(counts[1].item()), float(counts[2].item())

    # second half of the headline metric (the ~1M-triangle scene) and the shading the north star names (diffuse/emissive
    # materials): short runs of the single-GPU line, outside the timed region (a multi-GPU run reports its own workload only)
    large = materials = materials_large = None
    if rank == 0 and world == 1 and not args.no_second_scene:
        if args.workload != "instanced":
            large = side_run(local_rank, "instanced", "reference", 256, 8)
        if args.shading != "materials":
            materials = side_run(local_rank, "cornell", "materials", 256, 4)
            materials_large = side_run(local_rank, "instanced", "materials", 16, 4)

    if rank == 0:
        kernel_ms = st["total_dispatch_ms"] / max(1, st["dispatches"])
        key = config_key(args.workload, W, H, spp, args.shading, args.accel)
        pmc, pmc_note = pmc_entry(key) if world == 1 else (None, "single-GPU profile only")
        roof = roofline(st, kernel_ms, scene, args.shading, pmc, pmc_note)
        how = {"grid": "closest hit through the direction grid", "wave_walk": "device LBVH, one walk per wave", "lane_walk": "device LBVH, one walk per lane",
               "loop": "the shader's loop over all triangles", "loop_from": "the shader's loop over all triangles", "spheres": "the reference's 10 spheres (IntersectWorld)"}[which_kernel(st, scene.program, args.shading)[1]]
        out = {
            "metric": "Mrays/s (whole node) at 1080p Cornell box + 1M-tri scene, 1/2/4/8 MI355X",
            "value": traced / elapsed * 1e-6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if tile_split else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": {"cornell": "synthetic camera on the reference cornell_box.obj",
                     "instanced": "synthetic: 254 jittered instances of the reference suzanne.obj (fixed LCG seed)",
                     "torus_distorted": "reference catmark_torus_creases0.obj + a deterministically distorted copy of it (fixed LCG seed) standing in for "
                                        "data/distordedmesh/*.bin, which nothing in the reference loads"}.get(args.workload, "reference OBJ asset, reference camera"),
            "config": {"workload": f"{scene.name} {W}x{H}, {spp} spp per {'frame (one frame per step, rows shared by the GPUs)' if tile_split else 'GPU per step'}, depth 10, "
                                   f"{scene.num_triangles if scene.program == 1 else 0} triangles, {how}, shading {args.shading}",
                       "parallelism": "single GPU" if world == 1 else (f"image rows in interleaved blocks of 16 over {world} GPUs, merged by an all-gather of the owned rows (RCCL)" +
                                                                           (", overlapped with the next step's rendering" if overlap else "")
                                                                           if tile_split else f"sample split over {world} GPUs + RCCL all-reduce of the sums" +
                                                                           (", overlapped with the next step's rendering" if merger is not None else ""))},
            "value_reference_definition": rays / elapsed * 1e-6,
            "host_issue_ms_per_step": {"median": float(np.median(issue_ms)), "max": float(np.max(issue_ms)),
                                       "note": "time this rank's host spent inside each step's calls (dispatch and merge submission). With the "
                                               "library's timing switched on, as here, a dispatch first collects the previous dispatch's events, so "
                                               "the host runs one step ahead of the device and the median is the device's step time; what matters is "
                                               "how far the maximum lies above it (a dispatch that rebuilds its work list also waits for two counters)"},
            "rays_definition": "value: closest-hit queries actually EVALUATED per second (a ray is counted where it is traced, as the "
                               "reference's CPU tracer does at PathTracer.cpp:947). The shader has no pixel jitter, so the first query "
                               "of every sample of a pixel is the same; it is evaluated once per dispatch and pixel and reused "
                               "(bit-identical image) and counts once. value_reference_definition counts one query per bounce-loop "
                               "iteration of every sample, reused ones included (what the reference's own loop would have traced for "
                               "this image); 'no_primary_reuse' is the same step with every query evaluated.",
            "primary_reuse": True,
            "no_primary_reuse": None if plain is None else
                                {"value": plain["rays"] / 3 / plain_elapsed * 1e-6 * world, "unit": "Mrays/s", "ms_per_step": plain_elapsed * 1e3,
                                 "kernel_ms": plain["total_dispatch_ms"] / max(1, plain["dispatches"]), "note": "rank 0 clock, 3 steps after 2 warm-up steps"},
            "samples_per_s": samples / elapsed,
            "rays_per_sample": rays / samples if samples else 0.0,
            "roofline": roof,
        }
        if large:
            out["large_scene"] = large
        if materials:
            out["materials_mode"] = {"cornell": materials, "large_scene": materials_large}
        # the reference shader itself on Mesa llvmpipe, same scene, camera and passes: recorded in the build container (the
        # shader text does not travel to the GPU box), see oracle/time_c1.py --c2
        gold = os.path.join(ROOT, "tests", "golden", "c2_cornell_1080p.json")
        if args.workload == "cornell" and args.shading == "reference" and (W, H, spp) == (1920, 1080, 256) and os.path.exists(gold):
            lp = json.load(open(gold))["llvmpipe"]
            out["llvmpipe_recorded"] = {"Mrays_per_s": lp["Mrays_per_s"], "ms_per_pass_median": lp["pass_ms_median"], "cores": lp["threads"],
                                        "renderer": lp["renderer"], "note": "reference pathtracing.fs, passes 1..256 of this workload, timed around glDrawArrays..glFinish "
                                        "in the build container, not on this machine; its final image is bit-identical to this run's (tests/golden/c2_cornell_1080p.json)"}
        if not args.no_cpu_baseline and world == 1:   # (the CPU figure belongs to the single-GPU line; the other ranks would only wait for it)
            out["cpu_baseline"] = cpu_baseline(scene, spp, args.shading)
        print(json.dumps(out), flush=True)

    tracer.close()
    if world > 1:
        dist.barrier()   # rank 0 reports after the timed region: the ranks leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
