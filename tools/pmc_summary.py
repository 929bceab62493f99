#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs (tools/prof_pmc.sh): per kernel and counter, the MAXIMUM value over
its dispatches = the level-0 launch (every counter grows with the level size; grid sizes alone do not
identify the level because the strip-height heuristic can give levels 0 and 1 equal grids)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> [values of largest-grid dispatches]
grid = defaultdict(int)
rows = []
for f in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            name = r["Kernel_Name"]
            if not name.startswith(("lk::", "void lk::")):
                continue
            short = name.replace("void ", "").split("(")[0]
            g = int(r["Grid_Size"]) if "Grid_Size" in r else int(r.get("Grid_Size_X", 0))
            rows.append((short, g, r["Counter_Name"], float(r["Counter_Value"]), r))
            grid[short] = max(grid[short], g)
for short, g, cname, val, r in rows:
    acc[short][cname].append(val)
import json
traffic = {}
for k in sorted(acc):
    print(f"== {k}  (grid {grid[k]} work-items)")
    c = {n: max(v) for n, v in acc[k].items()}
    for n in sorted(c):
        print(f"   {n:32s} {c[n]:18.1f}")
    if "FETCH_SIZE" in c:
        # FETCH_SIZE is in KiB and (gfx950) counts 128-B requests as 64 B: x2 (MI355X_MICROARCH.md, HBM)
        print(f"   -> HBM read  ~ {c['FETCH_SIZE'] * 1024 * 2 / 1e6:10.1f} MB/launch (FETCH_SIZE KiB x2 correction)")
    if "WRITE_SIZE" in c:
        print(f"   -> HBM write ~ {c['WRITE_SIZE'] * 1024 / 1e6:10.1f} MB/launch")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        traffic[k] = {"grid_work_items": grid[k], "hbm_read_bytes": c["FETCH_SIZE"] * 1024 * 2,
                      "hbm_write_bytes": c["WRITE_SIZE"] * 1024,
                      "note": "rocprofv3 --pmc FETCH_SIZE (KiB, x2 gfx950 read correction) / WRITE_SIZE (KiB), "
                              "of the level-0 launch (max over dispatches)"}
    if "SQ_WAVES" in c and c["SQ_WAVES"]:
        w = c["SQ_WAVES"]
        for n in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM"):
            if n in c:
                print(f"   -> {n}/wave = {c[n] / w:10.1f}")

if len(sys.argv) > 2:
    # one JSON per workload ("<cols>x<rows>x<pairs per launch>"), stamped with the digest of the sources the LOADED library was
    # built from; tools/merge_pmc_traffic.py folds them into profiles/pmc_traffic.json, which bench.py reads per workload
    workload = sys.argv[3] if len(sys.argv) > 3 else "1920x1080x32"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import __graft_entry__ as ge
    # optional 4th argument: the kernel-trace-by-grid CSV of the same command (tools/kernel_trace_by_grid.py): the mean
    # duration of each kernel's largest-grid (level-0) dispatches is stored next to its traffic
    if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
        best = {}
        with open(sys.argv[4]) as fh:
            for r in csv.DictReader(fh):
                k, g = r["Name"], int(r["Grid_Size"])
                if k not in best or g > best[k][0]:
                    best[k] = (g, float(r["AverageNs"]) / 1e6, int(r["Calls"]))
        for k, (g, ms, n) in best.items():
            traffic.setdefault(k, {"grid_work_items": g})
            traffic[k]["rocprof_avg_launch_ms"] = round(ms, 4)
            traffic[k]["rocprof_dispatches"] = n
    with open(sys.argv[2], "w") as fh:
        json.dump({"_workload": workload, "_csrc_sha": ge.lib_digest() or "unstamped", **traffic}, fh, indent=1)
