"""Summarise the per-wave segment stamps of a -DLK_SPLIT_STAMPS build (LKFLOW_STAMPS_OUT=file):
tools/split_stamps_summary.py file [n_pe_waves]"""
import sys
import numpy as np

d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 4          # waves are ordered B (nb), S (5), F
d = d[d[:, 6] > 0]
for name, sel in (("B", d[:, 7] < nb), ("S", (d[:, 7] >= nb) & (d[:, 7] < nb + 5)), ("F", d[:, 7] >= nb + 5)):
    w = d[sel]
    if not len(w):
        continue
    per = w[:, :4] / w[:, 6:7]
    tot = per.sum(axis=1)
    labels = {"F": ("loads+products", "-", "rotate(vm wait)", "barrier"), "B": ("-", "solve+store", "-", "barrier"),
              "S": ("lds+hsum", "ring+write", "-", "barrier")}[name]
    print(f"{name}: {len(w)} waves, {w[:,6].mean():.0f} steps, cycles/step total {tot.mean():.0f} (p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f})")
    for k, lab in enumerate(labels):
        print(f"   {lab:16s} {per[:,k].mean():8.1f}  ({100*per[:,k].mean()/tot.mean():.0f} %)")
