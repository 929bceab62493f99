#!/usr/bin/env python3
"""Fold per-workload PMC / kernel-trace records (tools/pmc_summary.py -> *_pmc_traffic.json) into profiles/pmc_traffic.json:
{"workloads": {"<cols>x<rows>x<pairs>": {"_csrc_sha": ..., "<kernel>": {...}}}}.  bench.py attaches a record to its line only when
the record's source digest is the tree's and the loaded library's.   usage: merge_pmc_traffic.py out.json in1.json [in2.json ...]"""
import json
import sys

out = {"workloads": {}}
for f in sys.argv[2:]:
    d = json.load(open(f))
    out["workloads"][d["_workload"]] = d
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("merged", sorted(out["workloads"]))
