#!/usr/bin/env python3
"""Hardware counters and kernel statistics of one bench.py configuration, for the tracked profile bench.py reads.

On the GPU box:    python3 tools/pmc_collect.py <tag> [bench.py arguments, e.g. --workload instanced --spp 64]
  runs `bench.py --lean <arguments>` five times under rocprofv3 (one counter group per run, each within the hardware's
  slots; the last run is --kernel-trace --stats) and writes
      gpurun_out/pmc/<tag>.json               the entry: per-launch averages over the timed launches of the render kernel
      gpurun_out/pmc/<tag>_kernel_stats.csv   rocprofv3's kernel statistics of the same command
      gpurun_out/pmc/<tag>_bench.json         the bench line of the kernel-trace run
Here, afterwards:  python3 tools/pmc_collect.py --merge
  folds every gpurun_out/pmc/*.json into profiles/r03_pmc.json (keyed by configuration) and copies the statistics files
  to profiles/r03_kernel_stats_<tag>.csv.

Every entry carries the hash of the kernel sources it was taken from; bench.py ignores entries whose hash is not the
current one. Units as rocprofv3 reports them: FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE is doubled (on gfx950 it reports
half the bytes of wide coalesced reads, MI355X_MICROARCH.md "HBM"); SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in
quad-cycles; GRBM_GUI_ACTIVE summed over the 8 XCDs.
"""
import csv
import glob
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "pmc")

PASSES = {
    "sq1": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVES",
            "GRBM_GUI_ACTIVE"],
    "sq2": ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_THREAD_CYCLES_VALU", "SQ_WAIT_ANY",
            "SQ_ACTIVE_INST_ANY", "SQ_INSTS_SMEM"],
    "fetch": ["FETCH_SIZE", "TCC_HIT_sum"],
    "write": ["WRITE_SIZE", "TCC_MISS_sum"],
    # the vector memory path of a compute unit: address unit (TA), L1 (TCP), data return (TD)
    "mem": ["TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum",
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TD_TD_BUSY_sum", "TD_TC_STALL_sum", "SQ_INSTS_VMEM_RD", "SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES"],
}


def short_name(kernel_name):
    m = re.search(r"pt_render_kernel<[^>]*>|pt_render_decoupled_kernel|pt_cpu_tracer_kernel", kernel_name)
    return m.group(0).replace(" ", "") if m else kernel_name


def collect(tag, bench_args):
    import bench
    a = bench.parse_args(["--lean"] + bench_args)
    os.makedirs(OUT, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd_tail = ["python3", os.path.join(ROOT, "bench.py"), "--lean"] + bench_args
    counters = {}
    kernel_name = None
    for name, group in PASSES.items():
        d = os.path.join(OUT, f"{tag}_{name}")
        shutil.rmtree(d, ignore_errors=True)
        r = subprocess.run(["timeout", "-s", "KILL", "120", "rocprofv3", "--pmc"] + group + ["-d", d, "-o", "p", "--output-format", "csv", "--"] + cmd_tail,
                           cwd="/tmp", env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print(f"pass {name} failed ({r.returncode}):\n{r.stderr[-1500:]}", file=sys.stderr)
            sys.exit(1)
        rows = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        # the render kernel with the most time in it = the most dispatches x grid size is not known here: take the render
        # kernel variant that was dispatched most often, and of its dispatches the last `steps` (the timed ones)
        by_kernel = {}
        for r_ in rows:
            if "pt_render_" in r_["Kernel_Name"] or "pt_cpu_tracer_kernel" in r_["Kernel_Name"]:
                by_kernel.setdefault(r_["Kernel_Name"], set()).add(int(r_["Dispatch_Id"]))
        if not by_kernel:
            print(f"pass {name}: no render kernel dispatch in the counter file", file=sys.stderr)
            sys.exit(1)
        kernel_name = max(by_kernel, key=lambda k: len(by_kernel[k]))
        timed = sorted(by_kernel[kernel_name])[-a.steps:]
        for r_ in rows:
            if r_["Kernel_Name"] == kernel_name and int(r_["Dispatch_Id"]) in timed:
                counters[r_["Counter_Name"]] = counters.get(r_["Counter_Name"], 0.0) + float(r_["Counter_Value"]) / len(timed)
        shutil.rmtree(d, ignore_errors=True)
        print(f"pass {name}: {len(timed)} timed launches of {short_name(kernel_name)}", flush=True)
    # kernel statistics of the same command
    d = os.path.join(OUT, f"{tag}_trace")
    shutil.rmtree(d, ignore_errors=True)
    r = subprocess.run(["timeout", "-s", "KILL", "180", "rocprofv3", "--kernel-trace", "--stats", "-d", d, "-o", "p", "--output-format", "csv", "--"] + cmd_tail,
                       cwd="/tmp", env=env, capture_output=True, text=True)
    if r.returncode != 0:
        print(f"kernel-trace run failed ({r.returncode}):\n{r.stderr[-1500:]}", file=sys.stderr)
        sys.exit(1)
    bench_line = json.loads(r.stdout.strip().splitlines()[-1])
    json.dump(bench_line, open(os.path.join(OUT, f"{tag}_bench.json"), "w"), indent=1)
    stats_file = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats_file, os.path.join(OUT, f"{tag}_kernel_stats.csv"))
    stats = {r_["Name"]: r_ for r_ in csv.DictReader(open(stats_file))}
    shutil.rmtree(d, ignore_errors=True)
    ks = stats[kernel_name]
    kernel_ms_trace = float(ks["AverageNs"]) * 1e-6
    kernel_ms_events = bench_line["roofline"]["kernel_ms"]

    c = counters
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0                       # per XCD = kernel duration in shader cycles
    simds = 256 * 4
    # SQ_THREAD_CYCLES_VALU adds the active lanes of every VALU instruction: its ratio to 64 x instructions is the share of
    # lanes that EXEC left switched on
    lane_util_hw = c["SQ_THREAD_CYCLES_VALU"] / (c["SQ_INSTS_VALU"] * 64.0)
    fp32_insts = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_TRANS_F32"]
    fp32_flop_insts = c["SQ_INSTS_VALU_ADD_F32"] + c["SQ_INSTS_VALU_MUL_F32"] + 2 * c["SQ_INSTS_VALU_FMA_F32"] + c["SQ_INSTS_VALU_TRANS_F32"]
    sec = kernel_ms_events * 1e-3
    entry = {
        "kernel": short_name(kernel_name),
        "kernel_hash": bench.kernel_source_hash(),
        "source": "rocprofv3 --pmc passes of `python3 bench.py --lean " + " ".join(bench_args) + f"` (tools/pmc_collect.py {tag}); averages per launch over the "
                  f"{a.steps} timed launches",
        "counters_per_launch": {k: v for k, v in sorted(c.items())},
        "kernel_ms": kernel_ms_events,
        "kernel_ms_rocprof_stats_avg": kernel_ms_trace,
        "kernel_stats_calls": int(ks["Calls"]),
        "shader_cycles": cycles,
        "shader_clock_GHz": cycles / (kernel_ms_trace * 1e-3) * 1e-9,
        "valu_insts_per_launch": c["SQ_INSTS_VALU"],
        # a wave64 VALU instruction holds a SIMD-32 for 2 cycles
        "valu_issue_share": c["SQ_INSTS_VALU"] * 2.0 / (cycles * simds),
        "wait_inst_share": c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"],
        "wait_any_share": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "active_lane_share_of_valu": lane_util_hw,
        "fp32_arith_insts_per_launch": fp32_insts,
        "fp32_share_of_valu_insts": fp32_insts / c["SQ_INSTS_VALU"],
        "hw_fp32_flop_per_launch_all_lanes": fp32_flop_insts * 64.0,
        # add + mul + 2 fma + transcendental instructions x 64 lanes x the active share: what the hardware counted of the
        # quantity bench.py models from its event counters (executed_flop_per_launch)
        "hw_fp32_flop_per_launch_active_lanes": fp32_flop_insts * 64.0 * lane_util_hw,
        "bench_model_flop_per_launch": bench_line["roofline"].get("executed_flop_per_launch"),
        "valu_lane_ops_frac_of_unfused_peak": c["SQ_INSTS_VALU"] * 64.0 / sec / (bench.PEAK_FP32_UNFUSED_TFLOPS * 1e12),
        "fetch_bytes_per_launch": 2.0 * c["FETCH_SIZE"] * 1024.0,
        "write_bytes_per_launch": c["WRITE_SIZE"] * 1024.0,
        "hbm_bytes_per_launch": 2.0 * c["FETCH_SIZE"] * 1024.0 + c["WRITE_SIZE"] * 1024.0,
        "l2_hit_rate": c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) if c.get("TCC_HIT_sum") is not None and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) else None,
        # shares of the kernel's cycles, averaged over the 256 compute units; L1 accesses = 64-byte line look-ups
        "ta_busy_share": c["TA_TA_BUSY_sum"] / (256.0 * cycles) if "TA_TA_BUSY_sum" in c else None,
        "td_busy_share": c["TD_TD_BUSY_sum"] / (256.0 * cycles) if "TD_TD_BUSY_sum" in c else None,
        "l1_accesses_per_cycle_per_cu": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256.0 * cycles) if "TCP_TOTAL_CACHE_ACCESSES_sum" in c else None,
        "l1_miss_share": c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"] if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") else None,
        "l1_pending_stall_share": c["TCP_PENDING_STALL_CYCLES_sum"] / (256.0 * cycles) if "TCP_PENDING_STALL_CYCLES_sum" in c else None,
        "l1_accesses_per_vmem_read_inst": c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"] if c.get("SQ_INSTS_VMEM_RD") and "TCP_TOTAL_CACHE_ACCESSES_sum" in c else None,
        "bench_roofline_of_the_trace_run": {k: bench_line["roofline"][k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel_ms")},
    }
    # all-triangles loops: the share of record visits that goes past the wave-uniform back-face skip is counted only by the
    # measurement build of the library (-DPT_V_COUNT_FULL, one more VALU instruction per record), when it has been built
    variant = os.path.join(ROOT, "pathtracer_amd", "lib", "libpt_hip__countfull.so")
    if "<1," in short_name(kernel_name) and os.path.exists(variant):
        cmd = ["python3", os.path.join(ROOT, "tools", "one_dispatch.py"), "--lib", variant, "--workload", a.workload, "--spp", str(a.spp),
               "--shading", a.shading, "--accel", a.accel, "--reps", "1"]
        r = subprocess.run(["timeout", "-k", "10", "120"] + cmd, capture_output=True, text=True)
        if r.returncode == 0:
            d1 = json.loads(r.stdout.strip().splitlines()[-1])
            if d1["loop_triangle_visits"]:
                entry["loop_tests_past_backface_skip"] = d1["loop_full_tests"] / d1["loop_triangle_visits"]
                entry["loop_tests_past_backface_skip_source"] = "tools/one_dispatch.py with the -DPT_V_COUNT_FULL build, same workload"
    cfgw = bench_line["config"]["workload"].split()
    w, h = cfgw[1].rstrip(",").split("x")
    entry["key"] = bench.config_key(a.workload, int(w), int(h), a.spp, a.shading, a.accel)
    json.dump(entry, open(os.path.join(OUT, f"{tag}.json"), "w"), indent=1)
    print(json.dumps({k: entry[k] for k in ("key", "kernel", "kernel_ms", "kernel_ms_rocprof_stats_avg", "valu_issue_share", "wait_inst_share",
                                            "fp32_share_of_valu_insts", "active_lane_share_of_valu", "hbm_bytes_per_launch", "l2_hit_rate", "ta_busy_share", "td_busy_share",
                                            "l1_accesses_per_cycle_per_cu", "l1_miss_share", "l1_pending_stall_share", "l1_accesses_per_vmem_read_inst")}, indent=1))


def merge():
    path = os.path.join(ROOT, "profiles", "r03_pmc.json")
    doc = json.load(open(path)) if os.path.exists(path) else {"entries": {}}
    doc["about"] = "written by tools/pmc_collect.py --merge from rocprofv3 runs on MI355X; bench.py reads the entry of its configuration"
    for f in sorted(glob.glob(os.path.join(OUT, "*.json"))):
        if f.endswith("_bench.json"):
            continue
        e = json.load(open(f))
        tag = os.path.basename(f)[:-5]
        e["tag"] = tag
        doc["entries"][e["key"]] = e
        s = os.path.join(OUT, f"{tag}_kernel_stats.csv")
        if os.path.exists(s):
            shutil.copy(s, os.path.join(ROOT, "profiles", f"r03_kernel_stats_{tag}.csv"))
        b = os.path.join(OUT, f"{tag}_bench.json")
        if os.path.exists(b):
            shutil.copy(b, os.path.join(ROOT, "profiles", f"r03_bench_under_rocprof_{tag}.json"))
        print("merged", e["key"])
    json.dump(doc, open(path, "w"), indent=1)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "--merge":
        merge()
    elif len(sys.argv) >= 2:
        collect(sys.argv[1], sys.argv[2:])
    else:
        print(__doc__)
        sys.exit(2)
