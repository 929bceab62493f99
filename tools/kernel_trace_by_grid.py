#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid size): the library launches one kernel name at several
pyramid levels, which `--stats` averages together; the grid size tells the levels apart (largest grid = level 0).
Usage: tools/kernel_trace_by_grid.py <..._kernel_trace.csv> > by_grid.csv"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(list)
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        name = r["Kernel_Name"]
        if "lk::" not in name:
            continue
        grid = int(r.get("Grid_Size", 0) or 0) or int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1))
        acc[(name.replace("void ", "").split("(")[0], grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print('"Name","Grid_Size","Calls","AverageNs","MinNs","MaxNs"')
for (name, grid), v in sorted(acc.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
    print(f'"{name}",{grid},{len(v)},{sum(v) / len(v):.1f},{min(v)},{max(v)}')
