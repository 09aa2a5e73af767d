#!/usr/bin/env python3
"""Kernel timeline of the last camera move in a rocprofv3 --kernel-trace database of tools/camera_move_cost.py
(rocprofv3 --kernel-trace -d DIR -o move -- python3 tools/camera_move_cost.py; then this script on DIR/move_results.db)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name, start, end from kernels order by start"))
i0 = [i for i, r in enumerate(rows) if "prepare_triangles" in r[0]][-1]
t0 = rows[i0][1]
for r in rows[i0:i0 + 60]:
    name = re.sub(r"\(.*", "", r[0].replace("(anonymous namespace)::", ""))[:60]
    print(f"{(r[1] - t0) / 1e3:9.1f} us  +{(r[2] - r[1]) / 1e3:8.1f} us  {name}")
    if "pt_render_kernel" in r[0]:
        break
