import numpy as np, sys
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
a = a[a[:, 6] > 0]
rows = a[:, 6]
names = ["loads+products", "dpp+lds-write", "barrier", "lds-read+hsum+vert", "rotate(vmcnt)", "epilogue"]
tot = a[:, :6].sum(axis=1)
print("waves", len(a), "rows/wave", rows.mean(), "cycles/row", (tot / rows).mean())
for k, n in enumerate(names):
    print(f"  {n:22s} {(a[:, k] / rows).mean():9.1f} cycles/row  {100 * a[:, k].sum() / tot.sum():5.1f}%")
