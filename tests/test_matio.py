"""CPU suite: the reference's text golden format (saveMat / compare), tools/matio.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_save_load_compare_roundtrip(tmp_path, goldens):
    import matio
    g = goldens["UPyrTest"][:40, :50]
    p = tmp_path / "u.txt"
    matio.save_mat(g, p)
    first = open(p).readline().split()
    assert len(first) == 50 and all(len(t.split(".")[1]) == 6 for t in first)     # "%lf" = 6 decimals
    back = matio.load_mat(p)
    assert np.array_equal(back, g)                       # golden values are 6-decimal numbers already
    ok, rep = matio.compare(back, g)
    assert ok and rep.startswith("Success!")
    g2 = g.copy(); g2[3, 7] += 2e-6
    ok, rep = matio.compare(g, g2)
    assert not ok and "3,7" in rep
