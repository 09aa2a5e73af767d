#!/usr/bin/env python3
"""Opcode-class histogram of one kernel's ISA (static counts, from the -save-temps assembly).

  python3 tools/isa_histogram.py [kernel-name-substring, default pt_render_kernelILi4ELb0E] [--blocks]

Builds pt_kernels.hip with -save-temps (pathtracer_amd/lib/temps/), cuts out the kernel and counts instructions per class:
fp32 arithmetic (add/sub/mul/fma/mac/min/max/floor/fract/rcp/rsq/sqrt/...), fp32 compares, selects and moves (v_cndmask,
v_mov, v_readlane ...), conversions, integer/bit/address VALU, vector memory, LDS, scalar ALU, scalar memory, branches and
waits. With --blocks: the same per basic block (label), largest first, to see where the non-arithmetic work sits.
Static counts say what the code consists of, not how often each block runs; the run-time mix is what the SQ_INSTS_VALU_*
counters of tools/pmc_collect.py report.
"""
import collections
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FP_ARITH = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad|max|min|max3|min3|med3|floor|fract|ceil|trunc|rndne|rcp|rsq|sqrt|exp|log|sin|cos|ldexp|div_fixup|div_fmas|div_scale|frexp_mant|pk_mul|pk_add|pk_fma)(_legacy)?_f32")
CLASSES = [
    ("fp32_arith", FP_ARITH),
    ("fp32_compare", re.compile(r"^v_cmpx?_\w+_f32|^v_cmp_class_f32")),
    ("select_move", re.compile(r"^v_(cndmask|mov|readlane|readfirstlane|writelane|swap|accvgpr|permlane|bfrev)|^v_mov_b")),
    ("convert", re.compile(r"^v_cvt_")),
    ("int_valu", re.compile(r"^v_")),
    ("vmem", re.compile(r"^(global|buffer|flat|scratch)_")),
    ("lds", re.compile(r"^ds_")),
    ("smem", re.compile(r"^s_(load|buffer_load|store)")),
    ("wait_branch", re.compile(r"^s_(waitcnt|cbranch|branch|barrier|nop|endpgm|sleep|setprio)")),
    ("salu", re.compile(r"^s_")),
]


def classify(op):
    for name, rx in CLASSES:
        if rx.match(op):
            return name
    return "other"


def kernel_body(asm, needle):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and needle in l and l.rstrip().split(":")[0].endswith("PtKernelParams"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    needle = args[0] if args else "pt_render_kernelILi4ELb0E"
    from pathtracer_amd import build
    build.build_hip(force=True, save_temps=True)
    asm = open(os.path.join(build.LIB, "temps", "pt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    body = kernel_body(asm, needle)
    total = collections.Counter()
    blocks = collections.OrderedDict()
    cur = "entry"
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", t)
            if m:
                cur = m.group(1)
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = m.group(1)
            continue
        op = t.split()[0]
        if not re.match(r"^[a-z]", op):
            continue
        c = classify(op)
        total[c] += 1
        blocks.setdefault(cur, collections.Counter())[c] += 1
    valu = sum(total[k] for k in ("fp32_arith", "fp32_compare", "select_move", "convert", "int_valu"))
    out = {"kernel": needle, "instructions": sum(total.values()), "valu": valu, "classes": dict(total),
           "share_of_valu": {k: round(total[k] / valu, 4) for k in ("fp32_arith", "fp32_compare", "select_move", "convert", "int_valu")}}
    print(json.dumps(out, indent=1))
    if "--blocks" in sys.argv:
        for name, c in sorted(blocks.items(), key=lambda kv: -sum(kv[1].values()))[:25]:
            print(f"{name:12s} {sum(c.values()):5d}  " + "  ".join(f"{k}={v}" for k, v in c.most_common()))


if __name__ == "__main__":
    main()
