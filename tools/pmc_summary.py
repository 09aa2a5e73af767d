#!/usr/bin/env python3
"""Sums the rocprofv3 counter CSVs written by tools/pmc_passes.sh for the LAST dispatch of each kernel."""
import csv, glob, json, os, sys
tag = sys.argv[1]
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
out = {}
for d in sorted(glob.glob(os.path.join(root, f"pmc_{tag}_*"))):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        last = {}
        for r in rows:
            k = r["Kernel_Name"]
            last[k] = max(last.get(k, 0), int(r["Dispatch_Id"]))
        for r in rows:
            k = r["Kernel_Name"]
            if int(r["Dispatch_Id"]) != last[k] or "pt_render" not in k: continue
            out.setdefault(k, {})
            out[k][r["Counter_Name"]] = out[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(json.dumps(out, indent=1))
