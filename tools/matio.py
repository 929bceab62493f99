#!/usr/bin/env python3
"""Golden text I/O in the reference's own format.

saveMat (ParallelVideoPyr.cpp:193-204 / SerialPyr.cpp:108-119) writes one matrix row per line,
values printed with "%lf" (6 decimals) separated by single spaces; the checkers
(ParallelBasicPyr.cpp:774-832) re-read it with operator>> and compare at abs 1e-6, stopping at the
first mismatch.  These helpers let a user diff lkflow output against files produced by their own
OpenCV build of the reference.

    python tools/matio.py compare UPyrTest.txt mine.txt [--thresh 1e-6]
    python tools/matio.py npy2txt flow.npy out.txt
"""
import sys

import numpy as np


def save_mat(m, path):
    """saveMat(M, s): "%lf" per element, space separated, one row per line (path gets no suffix here)."""
    m = np.asarray(m, dtype=np.float64)
    assert m.ndim == 2
    with open(path, "w") as fh:
        for row in m:
            fh.write(" ".join("%f" % v for v in row))
            fh.write("\n")


def load_mat(path):
    return np.loadtxt(path, dtype=np.float64, ndmin=2)


def compare(ref, mine, thresh=1e-6):
    """The reference's check: first element with |a - b| > thresh, or None.  Returns (ok, report)."""
    if ref.shape != mine.shape:
        return False, f"shape mismatch: {ref.shape} vs {mine.shape}"
    d = np.abs(ref - mine)
    bad = np.argwhere(d > thresh)
    if bad.size == 0:
        return True, f"Success!  max |diff| = {d.max():.3e} over {d.size} elements"
    r, c = bad[0]
    return False, (f"Found mismatch\ncurrElem index: {r},{c}\nvalue: {mine[r, c]}\nCorrect value: {ref[r, c]}\n"
                   f"({len(bad)} elements beyond {thresh})")


def main(argv):
    if len(argv) >= 3 and argv[0] == "compare":
        thresh = float(argv[argv.index("--thresh") + 1]) if "--thresh" in argv else 1e-6
        ok, rep = compare(load_mat(argv[1]), load_mat(argv[2]), thresh)
        print(rep)
        return 0 if ok else 1
    if len(argv) == 3 and argv[0] == "npy2txt":
        save_mat(np.load(argv[1]), argv[2])
        return 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
