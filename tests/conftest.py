import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def table_u8():
    """The reference's table1/table2 crop Rect(0,0,640,448) as uint8 (tests/golden/make_golden.py)."""
    return tuple(np.fromfile(os.path.join(GOLDEN, f"{n}_640x448.u8"), dtype=np.uint8).reshape(448, 640)
                 for n in ("table1", "table2"))


@pytest.fixture(scope="session")
def table_f64(oracle, table_u8):
    return tuple(oracle.u8_to_f64(a) for a in table_u8)


@pytest.fixture(scope="session")
def goldens():
    """U2Test, V2Test, UPyrTest, VPyrTest, V2 of the reference as float64 (6-decimal text values)."""
    z = np.load(os.path.join(GOLDEN, "goldens.npz"))
    return {k: z[k].astype(np.float64) / 1e6 for k in z.files}


@pytest.fixture(scope="session")
def lkflow():
    """The ctypes binding; builds liblkflow.so first if it is missing or stale (the .so is git-ignored, so a fresh
    checkout has none -- hipcc cross-compiles gfx950 without a GPU, about a minute)."""
    import lkflow_amd
    if not os.environ.get("LKFLOW_LIB"):
        import __graft_entry__
        __graft_entry__.build_hip()          # no-op unless the library is missing or was built from other sources (digest)
    return lkflow_amd


@pytest.fixture(scope="session")
def gpu_ctx(lkflow):
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    ctx = lkflow.Context(0)
    yield ctx
    ctx.close()
