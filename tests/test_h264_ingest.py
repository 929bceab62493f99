"""Video ingest: optical-flow_amd/h264_ingest.py, the in-repo H.264 Constrained-Baseline decoder that stands in for
cv::VideoCapture (no decoder library in the image), and the committed luma fixture of the reference's pacman.mp4.

What can be checked without a reference decoder (the pixel values themselves are UNPINNED): the CAVLC tables are complete prefix codes
(a wrong code word collides with another or leaves a hole), the decoder's arithmetic kernels satisfy the standard's
identities, the fixture is what the decoder produces from the reference's file (when that file is present: never on the GPU box),
every slice parses to exactly its stop bit, and IDR pictures restart without a jump."""
import os
import sys
from fractions import Fraction

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "optical-flow_amd"))
import h264_ingest as h  # noqa: E402

FIXTURE = os.path.join(ROOT, "tests", "golden", "pacman_240x160_y.npz")
MP4 = "/root/reference/pacman.mp4"


def test_cavlc_tables_are_complete_prefix_codes():
    """Tables 9-5 / 9-7 / 9-8 / 9-9 / 9-10: every table fills the code space exactly, except that five tables leave the numerically
    smallest word(s) of their longest length unused (the all-zero end of the code space: start-code emulation).  VLC.__init__ has
    already asserted that no two code words overlap."""
    holes = {"coeff_token[0]": 2, "coeff_token[1]": 2, "coeff_token[2]": 1, "total_zeros[1]": 1, "run_before[7]": 1}   # unused longest words
    for v in h.ALL_VLCS:
        n = holes.get(v.name, 0)
        assert v.kraft() == 1 - Fraction(n, 1 << v.maxlen), v.name
        assert all(e is None for e in v.lut[:n]) and all(e is not None for e in v.lut[n:]), v.name
    assert sorted(h.CBP_INTRA) == list(range(48)) and sorted(h.CBP_INTER) == list(range(48))
    # every (TotalCoeff, TrailingOnes) with TrailingOnes <= min(TotalCoeff, 3) has a code word, nothing else has
    for v in h.COEFF_TOKEN:
        assert sorted(s for _, _, s in v.entries) == [4 * t + o for t in range(17) for o in range(min(t, 3) + 1)]


def test_exp_golomb_and_bit_reader():
    b = h.Bits(bytes([0b10100110, 0b01000010, 0b11000000]))       # ue: 0, 1, 2, 3, 4 ... = 1 010 011 00100 00101 1(stop)
    assert [b.ue() for _ in range(5)] == [0, 1, 2, 3, 4]
    assert not b.more_data()
    b = h.Bits(bytes([0b01001100, 0b10000101]))                   # se: 010 -> +1, 011 -> -1, 00100 -> +2, 00101 -> -2
    assert [b.se() for _ in range(4)] == [1, -1, 2, -2]
    assert h.unescape(bytes([0, 0, 3, 1, 0, 0, 3, 0, 0, 3])) == bytes([0, 0, 1, 0, 0, 0, 0])


def test_transforms_and_interpolation_identities():
    # a DC-only block: the inverse transform spreads (dc + 32) >> 6 evenly
    d = np.zeros((4, 4), np.int64); d[0, 0] = 640
    assert (h._itrans4x4(d) == 10).all()
    # quarter-sample interpolation of a constant plane is the constant at every fractional position; integer motion is a copy
    ref = np.full((32, 32), 77, np.uint8)
    for mvx in range(-6, 7):
        for mvy in (-5, 0, 2, 3, 7):
            assert (h.mc_luma(ref, 8, 8, 8, 8, mvx, mvy) == 77).all()
            assert (h.mc_chroma(ref, 4, 4, 4, 4, mvx, mvy) == 77).all()
    rng = np.random.default_rng(3)
    ref = rng.integers(0, 256, (40, 48)).astype(np.uint8)
    assert (h.mc_luma(ref, 16, 8, 16, 8, -8, 12) == ref[11:19, 14:30]).all()
    # the half-sample filter at a vertical step edge: (1, -5, 20, 20, -5, 1) / 32
    ref = np.zeros((16, 16), np.uint8); ref[:, 8:] = 160
    assert int(h.mc_luma(ref, 7, 4, 1, 1, 2, 0)[0, 0]) == (16 * 160 + 16) >> 5
    # plane prediction reproduces a linear ramp it is given the border of
    top = [10 + 2 * x for x in range(16)]; left = [10 + 3 * y for y in range(16)]
    p = h.pred_plane(top, left, 10 - 2 - 3 + 5, 16)              # (exactness is not required by the standard: within rounding)
    want = np.array([[10 + 2 * x + 3 * y + 5 for x in range(16)] for y in range(16)])
    assert np.abs(p - want).max() <= 6


def test_fixture_is_the_pacman_stream():
    y = np.load(FIXTURE)["Y"]
    assert y.shape == (245, 160, 240) and y.dtype == np.uint8
    # a dark game screen with bright sparse content; frames change a little at a time, the five IDR pictures (samples 0, 60, 120, 143,
    # 203 of the file's sync table) included: an IDR picture that followed drifted P pictures would jump
    assert 20 < y.mean() < 40 and y.max() > 200
    d = np.abs(y[1:].astype(np.int16) - y[:-1].astype(np.int16)).reshape(244, -1).mean(axis=1)
    assert d.max() < 8.0 and np.median(d) < 2.0
    for k in (60, 120, 143, 203):
        assert d[k - 1] < 4.0, (k, d[k - 1])


@pytest.mark.skipif(not os.path.exists(MP4), reason="the reference's pacman.mp4 is not on this machine (GPU box)")
def test_decoder_reproduces_the_fixture_and_consumes_every_slice():
    """The whole file again (about 15 s): 245 samples, one slice each, every slice parsed to exactly its rbsp_stop_one_bit
    (Decoder raises otherwise), the pictures equal to the committed fixture."""
    buf = open(MP4, "rb").read()
    sps, pps, nls, samples, sync = h.demux_avc(buf)
    assert len(samples) == 245 and nls == 4 and [i for i, s in enumerate(sync) if s] == [0, 60, 120, 143, 203]
    s = h.parse_sps(sps[0])
    assert (s["profile"], s["mb_w"], s["mb_h"], s["num_ref_frames"], s["poc_type"]) == (66, 15, 10, 1, 2)
    y, cb, cr, stats = h.decode_mp4(MP4, with_chroma=True)
    assert np.array_equal(y, np.load(FIXTURE)["Y"])
    assert stats["slice_bits_left"] == 0
    assert sum(stats[k] for k in stats if k != "slice_bits_left") == 245 * 150           # every macroblock of every picture
    assert cb.shape == (245, 80, 120) and 100 < cb.mean() < 156 and 100 < cr.mean() < 156   # chroma planes near neutral


# ---------------------------------------------------------------------------------------------- the C-ABI / C++ reader
PKG = os.path.join(ROOT, "optical-flow_amd")
INGEST_LIB = os.path.join(PKG, "liblkingest.so")


def _ingest_lib():
    import ctypes
    if not os.path.exists(INGEST_LIB):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build_ingest()
    return ctypes.CDLL(INGEST_LIB)


def test_ingest_library_exports_every_declared_symbol():
    """include/lkingest.h <-> optical-flow_amd/liblkingest.so (host-only, g++): every declared function is exported; the library does
    not link the HIP runtime, the flow library or the oracle; a missing file is LKINGEST_E_IO, a NULL argument LKINGEST_E_ARG."""
    import ctypes
    import re
    import subprocess
    L = _ingest_lib()
    hdr = open(os.path.join(ROOT, "include", "lkingest.h")).read()
    names = sorted(set(re.findall(r"\b(lkingest_[a-z_]+)\s*\(", hdr)))
    assert len(names) == 6, names
    for n in names:
        assert hasattr(L, n), n
    needed = subprocess.run(["readelf", "-d", INGEST_LIB], capture_output=True, text=True).stdout
    assert "amdhip" not in needed and "lkflow" not in needed and "oracle" not in needed
    h = ctypes.c_void_p()
    assert L.lkingest_open(ctypes.byref(h), b"/nonexistent/file.mp4") == -2 and not h.value
    assert L.lkingest_open(None, b"x") == -1
    assert L.lkingest_info(None, None, None, None) == -1


@pytest.mark.skipif(not os.path.exists(MP4), reason="the reference's pacman.mp4 is not on this machine (GPU box)")
def test_cpp_reader_equals_the_python_decoder(tmp_path):
    """tests/cpp/video_file_check.cpp: lkflow::VideoFile (include/lkingest.hpp over the C ABI; a second, scalar implementation of
    the decoder in C++) reads the reference's file like the reference's loop reads its capture; all 245 luma planes equal the
    Python decoder's fixture byte for byte, the first BGR frame equals yuv420_to_bgr of the fixture's planes."""
    import subprocess
    _ingest_lib()
    exe = str(tmp_path / "video_file_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "video_file_check.cpp"), "-o", exe, "-L", PKG, "-llkingest", f"-Wl,-rpath,{PKG}"])
    luma, bgr = str(tmp_path / "luma.u8"), str(tmp_path / "bgr.u8")
    r = subprocess.run([exe, MP4, luma, bgr], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "240 x 160, 245 frames" in r.stdout and "frames read: 245" in r.stdout
    f = np.load(FIXTURE)
    assert np.array_equal(np.fromfile(luma, np.uint8).reshape(245, 160, 240), f["Y"])
    assert np.array_equal(np.fromfile(bgr, np.uint8).reshape(160, 240, 3), h.yuv420_to_bgr(f["Y"][:1], f["Cb"][:1], f["Cr"][:1])[0])


@pytest.mark.skipif(not os.path.exists(MP4), reason="the reference's pacman.mp4 is not on this machine (GPU box)")
def test_cpp_reader_under_sanitizers_on_damaged_files(tmp_path):
    """The reader is fed files: a damaged one must end in a status, not in an out-of-range access.  The decoder and the harness built
    with -fsanitize=address,undefined: the real file decodes clean, and 60 mutated copies (tools/fuzz_ingest.py: flipped bytes in the
    sample tables, parameter sets and slice data, truncations, bit flips in the first IDR slice; the builder-side runs did 1500) each end
    with status 0 or 1 and no sanitizer report."""
    import subprocess
    lib = str(tmp_path / "liblkingest_asan.so")
    exe = str(tmp_path / "vfc_asan")
    san = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]
    subprocess.check_call(["g++", "-std=c++20", "-fPIC", "-shared", *san, os.path.join(PKG, "ingest", "h264_ingest.cpp"), "-o", lib])
    subprocess.check_call(["g++", "-std=c++17", *san, "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "video_file_check.cpp"),
                           "-o", exe, "-L", str(tmp_path), "-llkingest_asan", f"-Wl,-rpath,{tmp_path}"])
    r = subprocess.run([exe, MP4, str(tmp_path / "l.u8"), str(tmp_path / "b.u8")], capture_output=True, text=True)
    assert r.returncode == 0 and "frames read: 245" in r.stdout and "Sanitizer" not in r.stderr, r.stderr[-2000:]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ingest.py"), "11", "60", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CRASH" not in r.stdout, r.stdout[-3000:]
