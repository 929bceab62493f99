"""CPU suite: the C-ABI library loads and exports every symbol include/lkflow.h declares; argument
validation that needs no GPU.  No compute call is made here."""
import ctypes
import os

import pytest


def test_library_exports_every_header_symbol(lkflow):
    L = lkflow.lib()
    names = lkflow.header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), f"{n} declared in lkflow.h but not exported by liblkflow.so"
    # the Python binding must cover the whole header too
    assert sorted(lkflow.SIGNATURES) == names


def test_library_does_not_link_the_oracle(lkflow):
    # product path must be HIP only: no dependency on oracle/ artefacts
    import subprocess
    out = subprocess.run(["ldd", lkflow.LIB_PATH], capture_output=True, text=True).stdout
    assert "lk_oracle" not in out and "lk_cpu_omp" not in out
    assert "libamdhip64" in out


def test_max_levels_matches_reference_rule(lkflow, oracle):
    for w in list(range(1, 70)) + [240, 640, 1080, 1920, 3264, 3840, 7680]:
        for h in (1, 2, 3, 4, 6, 8, 40, 160, 448, 1080, 2160, 2448, 4320):
            assert lkflow.getMaxLayer(w, h) == oracle.max_layer(w, h)
    assert lkflow.getMaxLayer(0, 5) == 0


def test_strerror_and_version(lkflow):
    L = lkflow.lib()
    assert L.lkflow_strerror(0) == b"ok"
    for code in range(-6, 0):
        assert len(L.lkflow_strerror(code)) > 3
    assert b"gfx950" in L.lkflow_version()


def test_null_context_is_rejected(lkflow):
    L = lkflow.lib()
    assert L.lkflow_ctx_create(None, 0) == lkflow.LKFLOW_E_ARG
    assert L.lkflow_ctx_reserve(None, 8, 8, 1, 1) == lkflow.LKFLOW_E_ARG
    assert L.lkflow_ctx_set_option(None, 1, 0) == lkflow.LKFLOW_E_ARG
    assert L.lkflow_ctx_workspace_bytes(None) == 0
    assert L.lkflow_compute_f64(None, None, None, 8, 8, 64, None, None, 64, 1, 3, 0) == lkflow.LKFLOW_E_ARG
    L.lkflow_ctx_destroy(None)  # no-op


def test_no_gpu_means_loud_failure(lkflow):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(lkflow.LKFlowError) as e:
        lkflow.Context(0)
    assert e.value.status in (lkflow.LKFLOW_E_NODEVICE, lkflow.LKFLOW_E_HIP)


def test_missing_library_raises(lkflow, monkeypatch):
    monkeypatch.setattr(lkflow, "_LIB", None)
    monkeypatch.setattr(lkflow, "LIB_PATH", os.path.join(os.path.dirname(lkflow.LIB_PATH), "nope.so"))
    with pytest.raises(ImportError):
        lkflow.lib()
