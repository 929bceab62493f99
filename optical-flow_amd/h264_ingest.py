"""Video ingest for BASELINE config 3: `pacman.mp4` -> luma frames, with no decoder library in the image.

The reference ships `pacman.mp4` (ISO-BMFF, one `avc1` track: H.264 Constrained Baseline, level 1.3, 240 x 160, CAVLC, frame
pictures only, plus an AAC track) and reads video with cv::VideoCapture (ParallelVideoPyr.cpp:769-776); this image has no
ffmpeg / OpenCV / PyAV / rocDecode, so this module IS the decoder: an ISO-BMFF sample-table demuxer and an H.264 decoder for what the
file uses -- ITU-T H.264 (03/2005) clauses 7.3 (syntax), 8.2.4 (reference lists, sliding window / MMCO 1), 8.3 (intra 4x4 / 16x16 /
chroma prediction, I_PCM), 8.4 (P macroblocks: motion-vector prediction, quarter-sample luma and eighth-sample chroma
interpolation, several reference frames), 8.5 (CAVLC residuals, 4x4 integer transform, DC transforms), 8.7 (deblocking filter),
9.1 / 9.2 (Exp-Golomb, CAVLC tables).  Not handled (raises NotImplementedError): CABAC, B slices, interlace, FMO / ASO, 8x8
transform, weighted prediction, long-term references, several slices per picture with deblocking across them.

PARITY STATUS: no reference decoder exists here to compare with, so the pixel values are UNPINNED.  What is checked (tests/
test_h264_ingest.py): every VLC table is a complete prefix code, every slice of the file parses to exactly its last bit (a wrong
table or syntax element desynchronises within a macroblock or two), every IDR picture restarts cleanly, and the frames look right
(profiles/ holds a contact sheet).  Host-side ingest only: nothing here is on the timed path, and the product path never imports it.

    frames = decode_mp4("/root/reference/pacman.mp4")        # uint8 [245, 160, 240] luma (Y of the decoded YCbCr 4:2:0)
"""
import struct

import numpy as np


# ------------------------------------------------------------------------------------------------ ISO-BMFF (mp4) demux
def _boxes(buf, lo, hi):
    while lo + 8 <= hi:
        size, typ = struct.unpack(">I4s", buf[lo:lo + 8])
        hdr = 8
        if size == 1:
            size = struct.unpack(">Q", buf[lo + 8:lo + 16])[0]
            hdr = 16
        elif size == 0:
            size = hi - lo
        if size < hdr:
            break
        yield typ, lo + hdr, lo + size
        lo += size


def _find(buf, lo, hi, path):
    for typ, a, b in _boxes(buf, lo, hi):
        if typ == path[0]:
            if len(path) == 1:
                return a, b
            r = _find(buf, a, b, path[1:])
            if r:
                return r
    return None


def demux_avc(buf):
    """-> (sps list, pps list, nal_length_size, [sample bytes in decoding order], [is_sync]) of the first avc1 track."""
    moov = _find(buf, 0, len(buf), [b"moov"])
    for typ, a, b in _boxes(buf, *moov):
        if typ != b"trak":
            continue
        stbl = _find(buf, a, b, [b"mdia", b"minf", b"stbl"])
        if not stbl:
            continue
        stsd = _find(buf, *stbl, [b"stsd"])
        if buf[stsd[0] + 12:stsd[0] + 16] != b"avc1":
            continue
        entry = stsd[0] + 8                                   # version/flags + entry count
        avcc = _find(buf, entry + 8 + 78, entry + struct.unpack(">I", buf[entry:entry + 4])[0], [b"avcC"])
        c = buf[avcc[0]:avcc[1]]
        nls = (c[4] & 3) + 1
        p = 6
        sps, pps = [], []
        for _ in range(c[5] & 31):
            n = struct.unpack(">H", c[p:p + 2])[0]
            sps.append(bytes(c[p + 2:p + 2 + n])); p += 2 + n
        npps = c[p]; p += 1
        for _ in range(npps):
            n = struct.unpack(">H", c[p:p + 2])[0]
            pps.append(bytes(c[p + 2:p + 2 + n])); p += 2 + n
        a_, _b = _find(buf, *stbl, [b"stsz"])
        fixed, count = struct.unpack(">II", buf[a_ + 4:a_ + 12])
        sizes = [fixed] * count if fixed else list(struct.unpack(">%dI" % count, buf[a_ + 12:a_ + 12 + 4 * count]))
        co = _find(buf, *stbl, [b"stco"])
        if co:
            n = struct.unpack(">I", buf[co[0] + 4:co[0] + 8])[0]
            chunks = list(struct.unpack(">%dI" % n, buf[co[0] + 8:co[0] + 8 + 4 * n]))
        else:
            co = _find(buf, *stbl, [b"co64"])
            n = struct.unpack(">I", buf[co[0] + 4:co[0] + 8])[0]
            chunks = list(struct.unpack(">%dQ" % n, buf[co[0] + 8:co[0] + 8 + 8 * n]))
        sc = _find(buf, *stbl, [b"stsc"])
        n = struct.unpack(">I", buf[sc[0] + 4:sc[0] + 8])[0]
        stsc = [struct.unpack(">III", buf[sc[0] + 8 + 12 * i:sc[0] + 20 + 12 * i]) for i in range(n)]
        samples, k = [], 0
        for ci, off in enumerate(chunks):
            per = 0
            for first, spc, _desc in stsc:
                if ci + 1 >= first:
                    per = spc
            for _ in range(per):
                if k >= count:
                    break
                samples.append(bytes(buf[off:off + sizes[k]])); off += sizes[k]; k += 1
        sync = [True] * count
        ss = _find(buf, *stbl, [b"stss"])
        if ss:
            n = struct.unpack(">I", buf[ss[0] + 4:ss[0] + 8])[0]
            keys = set(struct.unpack(">%dI" % n, buf[ss[0] + 8:ss[0] + 8 + 4 * n]))
            sync = [(i + 1) in keys for i in range(count)]
        return sps, pps, nls, samples, sync
    raise ValueError("no avc1 track")


# ------------------------------------------------------------------------------------------------ bit reader (RBSP)
class Bits:
    """MSB-first reader over an RBSP (emulation-prevention bytes already removed), clause 7.2."""

    def __init__(self, rbsp):
        self.n = len(rbsp) * 8
        self.v = int.from_bytes(rbsp, "big")
        self.pos = 0
        # position of the rbsp_stop_one_bit: the last 1 bit
        self.end = self.n - ((self.v & -self.v).bit_length()) if self.v else 0

    def u(self, k):
        if k == 0:
            return 0
        if self.pos + k > self.n:
            raise EOFError("read past the end of the NAL unit")
        r = (self.v >> (self.n - self.pos - k)) & ((1 << k) - 1)
        self.pos += k
        return r

    def peek(self, k):
        rem = self.n - self.pos
        if rem >= k:
            return (self.v >> (rem - k)) & ((1 << k) - 1)
        return (self.v & ((1 << rem) - 1)) << (k - rem)

    def ue(self):
        z = 0
        while self.u(1) == 0:
            z += 1
            if z > 32:
                raise ValueError("bad Exp-Golomb code")
        return (1 << z) - 1 + self.u(z)

    def se(self):
        k = self.ue()
        return (k + 1) >> 1 if k & 1 else -(k >> 1)

    def te(self, cmax):
        if cmax > 1:
            return self.ue()
        return 1 - self.u(1)

    def more_data(self):
        return self.pos < self.end

    def align(self):
        self.pos = (self.pos + 7) & ~7


def unescape(nal):
    out, z, i = bytearray(), 0, 0
    for b in nal:
        if z >= 2 and b == 3:
            z = 0
            continue
        out.append(b)
        z = z + 1 if b == 0 else 0
    return bytes(out)


# ------------------------------------------------------------------------------------------------ parameter sets (7.3.2.1 / 7.3.2.2)
def parse_sps(nal):
    b = Bits(unescape(nal[1:]))
    s = {"profile": b.u(8), "constraints": b.u(8), "level": b.u(8), "id": b.ue()}
    if s["profile"] in (100, 110, 122, 244, 44, 83, 86, 118, 128):
        raise NotImplementedError("High profiles")
    s["log2_max_frame_num"] = b.ue() + 4
    s["poc_type"] = b.ue()
    if s["poc_type"] == 0:
        s["log2_max_poc_lsb"] = b.ue() + 4
    elif s["poc_type"] == 1:
        s["delta_pic_order_always_zero"] = b.u(1)
        b.se(); b.se()
        for _ in range(b.ue()):
            b.se()
    s["num_ref_frames"] = b.ue()
    s["gaps_allowed"] = b.u(1)
    s["mb_w"] = b.ue() + 1
    s["mb_h"] = b.ue() + 1
    if not b.u(1):
        raise NotImplementedError("interlace (frame_mbs_only_flag = 0)")
    b.u(1)                                                   # direct_8x8_inference_flag
    s["crop"] = (0, 0, 0, 0)
    if b.u(1):
        s["crop"] = (b.ue(), b.ue(), b.ue(), b.ue())         # left, right, top, bottom in chroma sample units (x2 for 4:2:0 luma)
    return s


def parse_pps(nal):
    b = Bits(unescape(nal[1:]))
    p = {"id": b.ue(), "sps": b.ue(), "cabac": b.u(1), "pic_order_present": b.u(1)}
    if p["cabac"]:
        raise NotImplementedError("CABAC")
    if b.ue() != 0:
        raise NotImplementedError("slice groups (FMO)")
    p["num_ref_idx_l0"] = b.ue() + 1
    b.ue()
    if b.u(1) or b.u(2):
        raise NotImplementedError("weighted prediction")
    p["init_qp"] = 26 + b.se()
    b.se()
    p["chroma_qp_offset"] = b.se()
    p["deblock_control"] = b.u(1)
    p["constrained_intra"] = b.u(1)
    p["redundant_pic_cnt"] = b.u(1)
    return p


# ------------------------------------------------------------------------------------------------ CAVLC tables (9.2, Tables 9-5 .. 9-10)
# [nC class][4 * TotalCoeff + TrailingOnes] = (length, code); length 0: no such combination
_CT_LEN = [
    [1, 0, 0, 0, 6, 2, 0, 0, 8, 6, 3, 0, 9, 8, 7, 5, 10, 9, 8, 6, 11, 10, 9, 7, 13, 11, 10, 8, 13, 13, 11, 9, 13, 13, 13, 10,
     14, 14, 13, 11, 14, 14, 14, 13, 15, 15, 14, 14, 15, 15, 15, 14, 16, 15, 15, 15, 16, 16, 16, 15, 16, 16, 16, 16, 16, 16, 16, 16],
    [2, 0, 0, 0, 6, 2, 0, 0, 6, 5, 3, 0, 7, 6, 6, 4, 8, 6, 6, 4, 8, 7, 7, 5, 9, 8, 8, 6, 11, 9, 9, 6, 11, 11, 11, 7,
     12, 11, 11, 9, 12, 12, 12, 11, 12, 12, 12, 11, 13, 13, 13, 12, 13, 13, 13, 13, 13, 14, 13, 13, 14, 14, 14, 13, 14, 14, 14, 14],
    [4, 0, 0, 0, 6, 4, 0, 0, 6, 5, 4, 0, 6, 5, 5, 4, 7, 5, 5, 4, 7, 5, 5, 4, 7, 6, 6, 4, 7, 6, 6, 4, 8, 7, 7, 5,
     8, 8, 7, 6, 9, 8, 8, 7, 9, 9, 8, 8, 9, 9, 9, 8, 10, 9, 9, 9, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10],
]
_CT_BITS = [
    [1, 0, 0, 0, 5, 1, 0, 0, 7, 4, 1, 0, 7, 6, 5, 3, 7, 6, 5, 3, 7, 6, 5, 4, 15, 6, 5, 4, 11, 14, 5, 4, 8, 10, 13, 4,
     15, 14, 9, 4, 11, 10, 13, 12, 15, 14, 9, 12, 11, 10, 13, 8, 15, 1, 9, 12, 11, 14, 13, 8, 7, 10, 9, 12, 4, 6, 5, 8],
    [3, 0, 0, 0, 11, 2, 0, 0, 7, 7, 3, 0, 7, 10, 9, 5, 7, 6, 5, 4, 4, 6, 5, 6, 7, 6, 5, 8, 15, 6, 5, 4, 11, 14, 13, 4,
     15, 10, 9, 4, 11, 14, 13, 12, 8, 10, 9, 8, 15, 14, 13, 12, 11, 10, 9, 12, 7, 11, 6, 8, 9, 8, 10, 1, 7, 6, 5, 4],
    [15, 0, 0, 0, 15, 14, 0, 0, 11, 15, 13, 0, 8, 12, 14, 12, 15, 10, 11, 11, 11, 8, 9, 10, 9, 14, 13, 9, 8, 10, 9, 8, 15, 14, 13, 13,
     11, 14, 10, 12, 15, 10, 13, 12, 11, 14, 9, 12, 8, 10, 13, 8, 13, 7, 9, 12, 9, 12, 11, 10, 5, 8, 7, 6, 1, 4, 3, 2],
]
_CDC_LEN = [2, 0, 0, 0, 6, 1, 0, 0, 6, 6, 3, 0, 6, 7, 7, 6, 6, 8, 8, 7]
_CDC_BITS = [1, 0, 0, 0, 7, 1, 0, 0, 4, 6, 1, 0, 3, 3, 2, 5, 2, 3, 2, 0]
_TZ_LEN = [
    [1, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 9], [3, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 6, 6, 6, 6], [4, 3, 3, 3, 4, 4, 3, 3, 4, 5, 5, 6, 5, 6],
    [5, 3, 4, 4, 3, 3, 3, 4, 3, 4, 5, 5, 5], [4, 4, 4, 3, 3, 3, 3, 3, 4, 5, 4, 5], [6, 5, 3, 3, 3, 3, 3, 3, 4, 3, 6], [6, 5, 3, 3, 3, 2, 3, 4, 3, 6],
    [6, 4, 5, 3, 2, 2, 3, 3, 6], [6, 6, 4, 2, 2, 3, 2, 5], [5, 5, 3, 2, 2, 2, 4], [4, 4, 3, 3, 1, 3], [4, 4, 2, 1, 3], [3, 3, 1, 2], [2, 2, 1], [1, 1]]
_TZ_BITS = [
    [1, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 3, 2, 1], [7, 6, 5, 4, 3, 5, 4, 3, 2, 3, 2, 3, 2, 1, 0], [5, 7, 6, 5, 4, 3, 4, 3, 2, 3, 2, 1, 1, 0],
    [3, 7, 5, 4, 6, 5, 4, 3, 3, 2, 2, 1, 0], [5, 4, 3, 7, 6, 5, 4, 3, 2, 1, 1, 0], [1, 1, 7, 6, 5, 4, 3, 2, 1, 1, 0], [1, 1, 5, 4, 3, 3, 2, 1, 1, 0],
    [1, 1, 1, 3, 3, 2, 2, 1, 0], [1, 0, 1, 3, 2, 1, 1, 1], [1, 0, 1, 3, 2, 1, 1], [0, 1, 1, 2, 1, 3], [0, 1, 1, 1, 1], [0, 1, 1, 1], [0, 1, 1], [0, 1]]
_CTZ_LEN = [[1, 2, 3, 3], [1, 2, 2], [1, 1]]
_CTZ_BITS = [[1, 1, 1, 0], [1, 1, 0], [1, 0]]
_RUN_LEN = [[1, 1], [1, 2, 2], [2, 2, 2, 2], [2, 2, 2, 3, 3], [2, 2, 3, 3, 3, 3], [2, 3, 3, 3, 3, 3, 3], [3, 3, 3, 3, 3, 3, 3, 4, 5, 6, 7, 8, 9, 10, 11]]
_RUN_BITS = [[1, 0], [1, 1, 0], [3, 2, 1, 0], [3, 2, 1, 1, 0], [3, 2, 3, 2, 1, 0], [3, 0, 1, 3, 2, 5, 4], [7, 6, 5, 4, 3, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1]]


class VLC:
    """A prefix code: symbols decoded by table look-up on the next `maxlen` bits."""

    def __init__(self, lens, codes, name):
        self.name = name
        self.entries = [(l, c, s) for s, (l, c) in enumerate(zip(lens, codes)) if l]
        self.maxlen = max(l for l, _, _ in self.entries)
        self.lut = [None] * (1 << self.maxlen)
        for l, c, s in self.entries:
            assert c < (1 << l), (name, s)
            lo = c << (self.maxlen - l)
            for k in range(lo, lo + (1 << (self.maxlen - l))):
                assert self.lut[k] is None, (name, "codes overlap", s, self.lut[k])
                self.lut[k] = (l, s)

    def kraft(self):
        """Fraction of the code space in use (1 = complete prefix code)."""
        from fractions import Fraction
        return sum(Fraction(1, 1 << l) for l, _, _ in self.entries)

    def read(self, b):
        e = self.lut[b.peek(self.maxlen)]
        if e is None:
            raise ValueError("invalid code in " + self.name)
        b.pos += e[0]
        return e[1]


COEFF_TOKEN = [VLC(_CT_LEN[k], _CT_BITS[k], "coeff_token[%d]" % k) for k in range(3)]
COEFF_TOKEN_CDC = VLC(_CDC_LEN, _CDC_BITS, "coeff_token[chroma DC]")
TOTAL_ZEROS = [VLC(l, c, "total_zeros[%d]" % (k + 1)) for k, (l, c) in enumerate(zip(_TZ_LEN, _TZ_BITS))]
TOTAL_ZEROS_CDC = [VLC(l, c, "total_zeros[chroma DC %d]" % (k + 1)) for k, (l, c) in enumerate(zip(_CTZ_LEN, _CTZ_BITS))]
RUN_BEFORE = [VLC(l, c, "run_before[%d]" % (k + 1)) for k, (l, c) in enumerate(zip(_RUN_LEN, _RUN_BITS))]
ALL_VLCS = COEFF_TOKEN + [COEFF_TOKEN_CDC] + TOTAL_ZEROS + TOTAL_ZEROS_CDC + RUN_BEFORE

# coded_block_pattern me(v), Table 9-4 (chroma_format_idc 1): codeNum -> cbp
CBP_INTRA = [47, 31, 15, 0, 23, 27, 29, 30, 7, 11, 13, 14, 39, 43, 45, 46, 16, 3, 5, 10, 12, 19, 21, 26, 28, 35, 37, 42, 44, 1, 2, 4, 8, 17, 18, 20, 24,
             6, 9, 22, 25, 32, 33, 34, 36, 40, 38, 41]
CBP_INTER = [0, 16, 1, 2, 4, 8, 32, 3, 5, 10, 12, 15, 47, 7, 11, 13, 14, 6, 9, 31, 35, 37, 42, 44, 33, 34, 36, 40, 39, 43, 45, 46, 17, 18, 20, 24,
             19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41]
ZIGZAG = [(0, 0), (1, 0), (0, 1), (0, 2), (1, 1), (2, 0), (3, 0), (2, 1), (1, 2), (0, 3), (1, 3), (2, 2), (3, 1), (3, 2), (2, 3), (3, 3)]   # (x, y)
BLK_XY = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (3, 0), (2, 1), (3, 1), (0, 2), (1, 2), (0, 3), (1, 3), (2, 2), (3, 2), (2, 3), (3, 3)]  # blkIdx -> (x, y) / 4
BLK_IDX = {xy: i for i, xy in enumerate(BLK_XY)}
_V = [(10, 16, 13), (11, 18, 14), (13, 20, 16), (14, 23, 18), (16, 25, 20), (18, 29, 23)]       # normAdjust4x4 (8.5.9)
LEVEL_SCALE = np.zeros((6, 4, 4), np.int64)                  # [qP % 6][y][x], flat scaling matrix (16 *)
for _m in range(6):
    for _y in range(4):
        for _x in range(4):
            LEVEL_SCALE[_m, _y, _x] = 16 * (_V[_m][0] if (_x % 2 == 0 and _y % 2 == 0) else (_V[_m][1] if (_x % 2 == 1 and _y % 2 == 1) else _V[_m][2]))
QPC = list(range(30)) + [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18]
TC0 = [(0, 0, 0)] * 17 + [(0, 0, 1)] * 4 + [(0, 1, 1)] * 2 + [(1, 1, 1)] * 4 + [(1, 1, 2)] * 4 + [(1, 2, 3)] * 2 + [(2, 2, 3), (2, 2, 4), (2, 3, 4), (2, 3, 4),
      (3, 3, 5), (3, 4, 6), (3, 4, 6), (4, 5, 7), (4, 5, 8), (4, 6, 9), (5, 7, 10), (6, 8, 11), (6, 8, 13), (7, 10, 14), (8, 11, 16), (9, 12, 18), (10, 13, 20),
      (11, 15, 23), (13, 17, 25)]
assert len(QPC) == 52 and len(ALPHA) == 52 and len(BETA) == 52 and len(TC0) == 52


def residual_block(b, nC, max_coeff):
    """7.3.5.3.1 / 9.2: -> (coefficients in scan order [max_coeff], TotalCoeff)."""
    if nC < 0:
        tok = COEFF_TOKEN_CDC.read(b)
    elif nC < 2:
        tok = COEFF_TOKEN[0].read(b)
    elif nC < 4:
        tok = COEFF_TOKEN[1].read(b)
    elif nC < 8:
        tok = COEFF_TOKEN[2].read(b)
    else:
        v = b.u(6)                                          # 6-bit FLC: xxxxyy = (TotalCoeff - 1, TrailingOnes); 000011 = (0, 0)
        tok = 0 if v == 3 else ((v >> 2) + 1) * 4 + (v & 3)
    total, t1 = tok >> 2, tok & 3
    coef = [0] * max_coeff
    if total == 0:
        return coef, 0
    if total > max_coeff or t1 > total:
        raise ValueError("coeff_token out of range")
    level = []
    suffix_len = 1 if (total > 10 and t1 < 3) else 0
    for i in range(total):
        if i < t1:
            level.append(1 - 2 * b.u(1))
            continue
        prefix = 0
        while b.u(1) == 0:
            prefix += 1
            if prefix > 32:
                raise ValueError("level_prefix too long")
        code = min(15, prefix) << suffix_len
        if prefix == 14 and suffix_len == 0:
            size = 4
        elif prefix >= 15:
            size = prefix - 3
        else:
            size = suffix_len
        if size:
            code += b.u(size)
        if prefix >= 15 and suffix_len == 0:
            code += 15
        if prefix >= 16:
            code += (1 << (prefix - 3)) - 4096
        if i == t1 and t1 < 3:
            code += 2
        lv = (code + 2) >> 1 if code % 2 == 0 else (-code - 1) >> 1
        level.append(lv)
        if suffix_len == 0:
            suffix_len = 1
        if abs(lv) > (3 << (suffix_len - 1)) and suffix_len < 6:
            suffix_len += 1
    zeros_left = 0
    if total < max_coeff:
        zeros_left = (TOTAL_ZEROS_CDC if max_coeff == 4 else TOTAL_ZEROS)[total - 1].read(b)
        if zeros_left + total > max_coeff:
            raise ValueError("total_zeros out of range")
    run = [0] * total
    for i in range(total - 1):
        if zeros_left > 0:
            run[i] = RUN_BEFORE[min(zeros_left, 7) - 1].read(b)
            zeros_left -= run[i]
            if zeros_left < 0:
                raise ValueError("run_before out of range")
    run[total - 1] = zeros_left
    k = -1
    for i in range(total - 1, -1, -1):
        k += run[i] + 1
        coef[k] = level[i]
    return coef, total


# ------------------------------------------------------------------------------------------------ transforms (8.5.10 - 8.5.12)
def _itrans4x4(d):
    """Inverse 4x4 core transform of scaled coefficients d [4][4] (rows = y) -> residual [(x + 32) >> 6]."""
    d = np.asarray(d, np.int64)
    e0 = d[:, 0] + d[:, 2]; e1 = d[:, 0] - d[:, 2]; e2 = (d[:, 1] >> 1) - d[:, 3]; e3 = d[:, 1] + (d[:, 3] >> 1)
    f = np.stack([e0 + e3, e1 + e2, e1 - e2, e0 - e3], axis=1)
    g0 = f[0] + f[2]; g1 = f[0] - f[2]; g2 = (f[1] >> 1) - f[3]; g3 = f[1] + (f[3] >> 1)
    h = np.stack([g0 + g3, g1 + g2, g1 - g2, g0 - g3], axis=0)
    return (h + 32) >> 6


def _scale4x4(coef, qp, skip_dc):
    """Scan-order coefficients -> scaled 4x4 array d[y][x] (8.5.12.1, flat matrices); the DC entry is left 0 when it comes
    from a DC transform (Intra16x16 luma, chroma)."""
    c = np.zeros((4, 4), np.int64)
    for k, (x, y) in enumerate(ZIGZAG):
        c[y, x] = coef[k]
    ls = LEVEL_SCALE[qp % 6]
    if qp >= 24:
        d = (c * ls) << (qp // 6 - 4)
    else:
        d = (c * ls + (1 << (3 - qp // 6))) >> (4 - qp // 6)
    if skip_dc:
        d[0, 0] = 0
    return d


_H4 = np.array([[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]], np.int64)


def _luma_dc(coef, qp):
    """Intra16x16 DC: scan-order 16 levels -> dcY[4][4] (index = 4x4 block (x, y) of the macroblock), 8.5.10."""
    c = np.zeros((4, 4), np.int64)
    for k, (x, y) in enumerate(ZIGZAG):
        c[y, x] = coef[k]
    f = _H4 @ c @ _H4
    ls = int(LEVEL_SCALE[qp % 6, 0, 0])
    if qp >= 36:
        return (f * ls) << (qp // 6 - 6)
    return (f * ls + (1 << (5 - qp // 6))) >> (6 - qp // 6)


def _chroma_dc(coef, qpc):
    c = np.array([[coef[0], coef[1]], [coef[2], coef[3]]], np.int64)
    a = np.array([[1, 1], [1, -1]], np.int64)
    f = a @ c @ a
    return ((f * int(LEVEL_SCALE[qpc % 6, 0, 0])) << (qpc // 6)) >> 5


def _clip8(a):
    return np.clip(a, 0, 255)


# ------------------------------------------------------------------------------------------------ inter prediction samples (8.4.2.2)
def _fetch(plane, x0, y0, w, h):
    H, W = plane.shape
    ys = np.clip(np.arange(y0, y0 + h), 0, H - 1)
    xs = np.clip(np.arange(x0, x0 + w), 0, W - 1)
    return plane[np.ix_(ys, xs)].astype(np.int64)


def _tap6(a, axis):
    n = a.shape[axis] - 5
    s = [slice(None)] * 2

    def sl(k):
        s2 = list(s); s2[axis] = slice(k, k + n); return a[tuple(s2)]
    return sl(0) - 5 * sl(1) + 20 * sl(2) + 20 * sl(3) - 5 * sl(4) + sl(5)


def mc_luma(ref, x, y, w, h, mvx, mvy):
    dx, dy = mvx & 3, mvy & 3
    blk = _fetch(ref, x + (mvx >> 2) - 2, y + (mvy >> 2) - 2, w + 5, h + 5)      # rows / cols -2 .. +2 (+3)
    G = blk[2:2 + h, 2:2 + w]
    if dx == 0 and dy == 0:
        return G
    b1 = _tap6(blk, 1)                                       # [h + 5][w]: half-sample columns, every row -2 .. h + 2
    h1 = _tap6(blk, 0)                                       # [h][w + 5]: half-sample rows, every column -2 .. w + 2
    bfull = _clip8((b1 + 16) >> 5)
    hfull = _clip8((h1 + 16) >> 5)
    b, s = bfull[2:2 + h], bfull[3:3 + h]
    hh, m = hfull[:, 2:2 + w], hfull[:, 3:3 + w]
    H_ = blk[2:2 + h, 3:3 + w]                               # integer sample to the right
    M_ = blk[3:3 + h, 2:2 + w]                               # integer sample below
    if dy == 0:
        return b if dx == 2 else ((G + b + 1) >> 1 if dx == 1 else (H_ + b + 1) >> 1)
    if dx == 0:
        return hh if dy == 2 else ((G + hh + 1) >> 1 if dy == 1 else (M_ + hh + 1) >> 1)
    if dx == 2 or dy == 2:
        j = _clip8((_tap6(b1, 0) + 512) >> 10)
        if dx == 2 and dy == 2:
            return j
        if dx == 2:
            return (j + (b if dy == 1 else s) + 1) >> 1
        return (j + (hh if dx == 1 else m) + 1) >> 1
    return ((b if dy == 1 else s) + (hh if dx == 1 else m) + 1) >> 1


def mc_chroma(ref, x, y, w, h, mvx, mvy):
    dx, dy = mvx & 7, mvy & 7
    blk = _fetch(ref, x + (mvx >> 3), y + (mvy >> 3), w + 1, h + 1)
    A, B, C, D = blk[:h, :w], blk[:h, 1:], blk[1:, :w], blk[1:, 1:]
    return ((8 - dx) * (8 - dy) * A + dx * (8 - dy) * B + (8 - dx) * dy * C + dx * dy * D + 32) >> 6


# ------------------------------------------------------------------------------------------------ intra prediction (8.3)
def _f3(a, b, c):
    return (a + 2 * b + c + 2) >> 2


def pred_intra4x4(mode, top, left, tl):
    """top: 8 samples or None, left: 4 or None, tl: sample or None -> pred[4][4] (rows y)."""
    p = np.zeros((4, 4), np.int64)
    if mode == 2:
        if top is not None and left is not None:
            v = (sum(top[:4]) + sum(left) + 4) >> 3
        elif left is not None:
            v = (sum(left) + 2) >> 2
        elif top is not None:
            v = (sum(top[:4]) + 2) >> 2
        else:
            v = 128
        p[:] = v
        return p
    if mode == 0:
        p[:] = np.array(top[:4])[None, :]
        return p
    if mode == 1:
        p[:] = np.array(left)[:, None]
        return p
    if mode in (3, 7):
        t = list(top)
        for y in range(4):
            for x in range(4):
                if mode == 3:
                    p[y, x] = (t[6] + 3 * t[7] + 2) >> 2 if (x == 3 and y == 3) else _f3(t[x + y], t[x + y + 1], t[x + y + 2])
                else:
                    k = x + (y >> 1)
                    p[y, x] = (t[k] + t[k + 1] + 1) >> 1 if y % 2 == 0 else _f3(t[k], t[k + 1], t[k + 2])
        return p
    if mode == 8:
        l = list(left)
        for y in range(4):
            for x in range(4):
                z = x + 2 * y
                k = y + (x >> 1)
                if z > 5:
                    p[y, x] = l[3]
                elif z == 5:
                    p[y, x] = (l[2] + 3 * l[3] + 2) >> 2
                elif z % 2 == 0:
                    p[y, x] = (l[k] + l[k + 1] + 1) >> 1
                else:
                    p[y, x] = _f3(l[k], l[k + 1], l[k + 2])
        return p
    # modes 4, 5, 6 use e = [l3, l2, l1, l0, tl, t0, t1, t2, t3]: p[x, -1] = e[5 + x], p[-1, y] = e[3 - y]
    e = [left[3], left[2], left[1], left[0], tl, top[0], top[1], top[2], top[3]]
    for y in range(4):
        for x in range(4):
            if mode == 4:
                i = 4 + x - y
                p[y, x] = _f3(e[i - 1], e[i], e[i + 1])
            elif mode == 5:
                z = 2 * x - y
                k = x - (y >> 1)
                if z >= 0 and z % 2 == 0:
                    p[y, x] = (e[5 + k - 1] + e[5 + k] + 1) >> 1
                elif z >= 0:
                    p[y, x] = _f3(e[5 + k - 2], e[5 + k - 1], e[5 + k])
                elif z == -1:
                    p[y, x] = _f3(e[3], e[4], e[5])
                else:
                    p[y, x] = _f3(e[3 - (y - 1)], e[3 - (y - 2)], e[3 - (y - 3)])
            else:
                z = 2 * y - x
                k = y - (x >> 1)
                if z >= 0 and z % 2 == 0:
                    p[y, x] = (e[3 - (k - 1)] + e[3 - k] + 1) >> 1
                elif z >= 0:
                    p[y, x] = _f3(e[3 - (k - 2)], e[3 - (k - 1)], e[3 - k])
                elif z == -1:
                    p[y, x] = _f3(e[3], e[4], e[5])
                else:
                    p[y, x] = _f3(e[5 + x - 1], e[5 + x - 2], e[5 + x - 3])
    return p


def pred_plane(top, left, tl, n):
    """Intra16x16 / chroma plane prediction: top, left of n samples (n = 16 or 8), tl = p[-1, -1]."""
    half = n // 2
    t = [tl] + list(top)                                     # t[1 + x] = p[x, -1]
    l = [tl] + list(left)
    hs = sum((k + 1) * (t[1 + half + k] - t[1 + half - 2 - k]) for k in range(half))
    vs = sum((k + 1) * (l[1 + half + k] - l[1 + half - 2 - k]) for k in range(half))
    a = 16 * (left[n - 1] + top[n - 1])
    if n == 16:
        b, c = (5 * hs + 32) >> 6, (5 * vs + 32) >> 6
    else:
        b, c = (34 * hs + 32) >> 6, (34 * vs + 32) >> 6
    xs = np.arange(n) - (half - 1)
    return _clip8((a + b * xs[None, :] + c * xs[:, None] + 16) >> 5)


# ------------------------------------------------------------------------------------------------ picture decoding
def _next_qp(qp, delta):
    if not -26 <= delta <= 25:                               # 7.4.5
        raise ValueError("mb_qp_delta out of range")
    return (qp + delta + 52) % 52


class Picture:
    def __init__(self, mb_w, mb_h):
        self.mb_w, self.mb_h = mb_w, mb_h
        W, H = 16 * mb_w, 16 * mb_h
        self.Y = np.zeros((H, W), np.uint8)
        self.C = [np.zeros((H // 2, W // 2), np.uint8), np.zeros((H // 2, W // 2), np.uint8)]
        self.frame_num = 0
        self.uid = -1                                         # identity for "same reference picture" tests


class Decoder:
    """H.264 Constrained Baseline, frame pictures, one slice per picture (what pacman.mp4 holds)."""

    def __init__(self, sps, pps):
        self.sps, self.pps = sps, pps
        self.mb_w, self.mb_h = sps["mb_w"], sps["mb_h"]
        self.refs = []                                        # short-term reference pictures, most recent first
        self.npic = 0
        self.stats = {"I4x4": 0, "I16x16": 0, "IPCM": 0, "P16x16": 0, "P16x8": 0, "P8x16": 0, "P8x8": 0, "PSkip": 0, "slice_bits_left": 0}

    # ---- slice header (7.3.3)
    def slice_header(self, b, nal_type, nal_ref_idc):
        sps, pps = self.sps, self.pps
        h = {"first_mb": b.ue(), "type": b.ue() % 5}
        if h["type"] not in (0, 2):
            raise NotImplementedError("slice type %d" % h["type"])
        if b.ue() != pps["id"]:
            raise NotImplementedError("several picture parameter sets")
        h["frame_num"] = b.u(sps["log2_max_frame_num"])
        if nal_type == 5:
            b.ue()                                            # idr_pic_id
        if sps["poc_type"] == 0:
            b.u(sps["log2_max_poc_lsb"])
            if pps["pic_order_present"]:
                b.se()
        elif sps["poc_type"] == 1 and not sps["delta_pic_order_always_zero"]:
            b.se()
            if pps["pic_order_present"]:
                b.se()
        if pps["redundant_pic_cnt"]:
            b.ue()
        h["num_ref"] = pps["num_ref_idx_l0"]
        h["reorder"] = []
        if h["type"] == 0:
            if b.u(1):
                h["num_ref"] = b.ue() + 1
            if b.u(1):                                        # ref_pic_list_reordering_flag_l0
                while True:
                    idc = b.ue()
                    if idc == 3:
                        break
                    h["reorder"].append((idc, b.ue()))
        h["mmco"] = []
        h["idr"] = nal_type == 5
        if nal_ref_idc:
            if nal_type == 5:
                b.u(1)
                if b.u(1):
                    raise NotImplementedError("long-term reference")
            elif b.u(1):
                while True:
                    op = b.ue()
                    if op == 0:
                        break
                    if op != 1:
                        raise NotImplementedError("memory_management_control_operation %d" % op)
                    h["mmco"].append((op, b.ue()))
        h["qp"] = pps["init_qp"] + b.se()
        if not 0 <= h["qp"] <= 51:
            raise ValueError("slice QP out of range")
        h["deblock"], h["alpha_off"], h["beta_off"] = 0, 0, 0
        if pps["deblock_control"]:
            h["deblock"] = b.ue()
            if h["deblock"] != 1:
                h["alpha_off"] = 2 * b.se()
                h["beta_off"] = 2 * b.se()
        if h["first_mb"] != 0:
            raise NotImplementedError("several slices per picture")
        return h

    def ref_list(self, h):
        """8.2.4.2.1 + 8.2.4.3: short-term pictures by descending PicNum, then the slice's reordering commands."""
        max_fn = 1 << self.sps["log2_max_frame_num"]

        def picnum(p):
            return p.frame_num if p.frame_num <= h["frame_num"] else p.frame_num - max_fn
        lst = sorted(self.refs, key=picnum, reverse=True)
        pred = h["frame_num"]
        for k, (idc, v) in enumerate(h["reorder"]):
            if idc > 1:
                raise NotImplementedError("long-term reordering")
            pred = (pred - (v + 1)) % max_fn if idc == 0 else (pred + (v + 1)) % max_fn
            want = pred if pred <= h["frame_num"] else pred - max_fn
            pic = next(p for p in lst if picnum(p) == want)
            lst = lst[:k] + [pic] + [p for p in lst[k:] if p is not pic]
        lst = lst[:h["num_ref"]]
        if h["type"] == 0 and len(lst) < h["num_ref"]:
            raise ValueError("missing reference picture")
        return lst

    # ---- one access unit = one slice NAL
    def decode_nal(self, nal):
        nal_type, nal_ref_idc = nal[0] & 31, (nal[0] >> 5) & 3
        if nal_type not in (1, 5):
            return None
        b = Bits(unescape(nal[1:]))
        h = self.slice_header(b, nal_type, nal_ref_idc)
        if h["idr"]:
            self.refs = []
        pic = Picture(self.mb_w, self.mb_h)
        pic.frame_num = h["frame_num"]
        pic.uid = self.npic
        self.npic += 1
        self._decode_slice(b, h, pic, self.ref_list(h) if h["type"] == 0 else [])
        if h["deblock"] != 1:
            self._deblock(pic, h)
        if nal_ref_idc:
            for op, v in h["mmco"]:                           # MMCO 1: unmark a short-term picture
                max_fn = 1 << self.sps["log2_max_frame_num"]
                want = (h["frame_num"] - (v + 1))
                self.refs = [p for p in self.refs if (p.frame_num if p.frame_num <= h["frame_num"] else p.frame_num - max_fn) != want]
            self.refs.insert(0, pic)
            del self.refs[max(self.sps["num_ref_frames"], 1):]   # sliding window (8.2.5.3)
        return pic

    # ---- slice data (7.3.4) and macroblocks (7.3.5)
    def _decode_slice(self, b, h, pic, reflist):
        mb_w, mb_h = self.mb_w, self.mb_h
        W4, H4 = 4 * mb_w, 4 * mb_h
        Y = pic.Y.astype(np.int64)
        C = [c.astype(np.int64) for c in pic.C]
        nmb = mb_w * mb_h
        # per-4x4 state (lists of lists: scalar indexing of numpy arrays is slow)
        nnzY = [[0] * W4 for _ in range(H4)]
        nnzC = [[[0] * (W4 // 2) for _ in range(H4 // 2)] for _ in range(2)]
        ipm = [[-1] * W4 for _ in range(H4)]                  # Intra4x4PredMode; -1: not an I4x4 block (counts as DC when available)
        mvs = [[(0, 0)] * W4 for _ in range(H4)]
        refi = [[-2] * W4 for _ in range(H4)]                 # -2: not decoded yet, -1: intra, >= 0: refIdxL0
        refu = [[-1] * W4 for _ in range(H4)]                 # uid of the reference picture (deblocking)
        mb_intra = [False] * nmb
        mb_qp = [0] * nmb
        mb_done = [False] * nmb
        qp = h["qp"]
        is_p = h["type"] == 0
        cqo = self.pps["chroma_qp_offset"]

        def qpc_of(q):
            return QPC[min(max(q + cqo, 0), 51)]

        def nc_luma(bx, by):
            a = nnzY[by][bx - 1] if bx > 0 else None
            t = nnzY[by - 1][bx] if by > 0 else None
            if a is not None and t is not None:
                return (a + t + 1) >> 1
            return a if a is not None else (t if t is not None else 0)

        def nc_chroma(ci, bx, by):
            a = nnzC[ci][by][bx - 1] if bx > 0 else None
            t = nnzC[ci][by - 1][bx] if by > 0 else None
            if a is not None and t is not None:
                return (a + t + 1) >> 1
            return a if a is not None else (t if t is not None else 0)

        def avail4(bx, by, cur_mb):
            """Is the 4x4 block (global units) available to the current macroblock: inside the picture and already decoded."""
            if bx < 0 or by < 0 or bx >= W4 or by >= H4:
                return False
            m = (by >> 2) * mb_w + (bx >> 2)
            if m == cur_mb:
                return refi[by][bx] != -2
            return m < cur_mb

        def mv_neighbour(bx, by, cur_mb):
            if not avail4(bx, by, cur_mb):
                return None
            r = refi[by][bx]
            return (r if r >= 0 else -1, mvs[by][bx] if r >= 0 else (0, 0))

        def pred_mv(bx, by, w4, h4, ref, cur_mb, shape, part):
            A = mv_neighbour(bx - 1, by, cur_mb)
            B = mv_neighbour(bx, by - 1, cur_mb)
            Cn = mv_neighbour(bx + w4, by - 1, cur_mb)
            if Cn is None:
                Cn = mv_neighbour(bx - 1, by - 1, cur_mb)
            if shape == "16x8":
                if part == 0 and B is not None and B[0] == ref:
                    return B[1]
                if part == 1 and A is not None and A[0] == ref:
                    return A[1]
            elif shape == "8x16":
                if part == 0 and A is not None and A[0] == ref:
                    return A[1]
                if part == 1 and Cn is not None and Cn[0] == ref:
                    return Cn[1]
            if B is None and Cn is None and A is not None:
                return A[1]
            na = A if A is not None else (-1, (0, 0))
            nb = B if B is not None else (-1, (0, 0))
            ncn = Cn if Cn is not None else (-1, (0, 0))
            same = [n for n in (na, nb, ncn) if n[0] == ref]
            if len(same) == 1:
                return same[0][1]
            return (sorted((na[1][0], nb[1][0], ncn[1][0]))[1], sorted((na[1][1], nb[1][1], ncn[1][1]))[1])

        def set_motion(bx, by, w4, h4, ref, mv):
            uid = reflist[ref].uid
            for yy in range(by, by + h4):
                for xx in range(bx, bx + w4):
                    mvs[yy][xx] = mv; refi[yy][xx] = ref; refu[yy][xx] = uid

        def inter_pred(mbx, mby, bx, by, w4, h4, ref, mv):
            rp = reflist[ref]
            x, y, w, hh = 4 * bx, 4 * by, 4 * w4, 4 * h4
            Y[y:y + hh, x:x + w] = mc_luma(rp.Y, x, y, w, hh, mv[0], mv[1])
            for ci in range(2):
                C[ci][y // 2:(y + hh) // 2, x // 2:(x + w) // 2] = mc_chroma(rp.C[ci], x // 2, y // 2, w // 2, hh // 2, mv[0], mv[1])

        def add_luma_residual(bx, by, d):
            x, y = 4 * bx, 4 * by
            Y[y:y + 4, x:x + 4] = _clip8(Y[y:y + 4, x:x + 4] + _itrans4x4(d))

        def chroma_residual(mbx, mby, cbp_c, qpc_):
            """Chroma DC / AC of both components (7.3.5.3) added to the prediction already in C."""
            dcs = []
            for ci in range(2):
                if cbp_c & 3:
                    coef, _ = residual_block(b, -1, 4)
                    dcs.append(_chroma_dc(coef, qpc_))
                else:
                    dcs.append(np.zeros((2, 2), np.int64))
            for ci in range(2):
                for k in range(4):
                    cx, cy = 2 * mbx + (k & 1), 2 * mby + (k >> 1)
                    if cbp_c & 2:
                        coef, tot = residual_block(b, nc_chroma(ci, cx, cy), 15)
                        nnzC[ci][cy][cx] = tot
                        d = _scale4x4([0] + coef, qpc_, True)
                    else:
                        d = np.zeros((4, 4), np.int64)
                    d[0, 0] = dcs[ci][k >> 1, k & 1]
                    if d.any():
                        x, y = 4 * cx, 4 * cy
                        C[ci][y:y + 4, x:x + 4] = _clip8(C[ci][y:y + 4, x:x + 4] + _itrans4x4(d))

        def intra_chroma_pred(mbx, mby, mode, left_ok, top_ok):
            for ci in range(2):
                P = C[ci]
                x0, y0 = 8 * mbx, 8 * mby
                top = [int(v) for v in P[y0 - 1, x0:x0 + 8]] if top_ok else None
                left = [int(v) for v in P[y0:y0 + 8, x0 - 1]] if left_ok else None
                if mode == 0:
                    for k in range(4):
                        xo, yo = 4 * (k & 1), 4 * (k >> 1)
                        st = sum(top[xo:xo + 4]) if top_ok else None
                        sl = sum(left[yo:yo + 4]) if left_ok else None
                        if (xo, yo) in ((0, 0), (4, 4)):
                            if st is not None and sl is not None:
                                v = (st + sl + 4) >> 3
                            elif st is not None or sl is not None:
                                v = ((st if st is not None else sl) + 2) >> 2
                            else:
                                v = 128
                        elif (xo, yo) == (4, 0):
                            v = (st + 2) >> 2 if st is not None else ((sl + 2) >> 2 if sl is not None else 128)
                        else:
                            v = (sl + 2) >> 2 if sl is not None else ((st + 2) >> 2 if st is not None else 128)
                        P[y0 + yo:y0 + yo + 4, x0 + xo:x0 + xo + 4] = v
                elif mode == 1:
                    P[y0:y0 + 8, x0:x0 + 8] = np.array(left)[:, None]
                elif mode == 2:
                    P[y0:y0 + 8, x0:x0 + 8] = np.array(top)[None, :]
                else:
                    P[y0:y0 + 8, x0:x0 + 8] = pred_plane(top, left, int(P[y0 - 1, x0 - 1]), 8)

        mb = 0
        prev_skipped = False
        while True:
            if is_p:
                run = b.ue()
                for _ in range(run):                          # P_Skip (8.4.1.1)
                    if mb >= nmb:
                        raise ValueError("mb_skip_run past the picture")
                    mbx, mby = mb % mb_w, mb // mb_w
                    bx, by = 4 * mbx, 4 * mby
                    A = mv_neighbour(bx - 1, by, mb)
                    B = mv_neighbour(bx, by - 1, mb)
                    if A is None or B is None or (A[0] == 0 and A[1] == (0, 0)) or (B[0] == 0 and B[1] == (0, 0)):
                        mv = (0, 0)
                    else:
                        mv = pred_mv(bx, by, 4, 4, 0, mb, "16x16", 0)
                    set_motion(bx, by, 4, 4, 0, mv)
                    inter_pred(mbx, mby, bx, by, 4, 4, 0, mv)
                    mb_qp[mb] = qp
                    mb_done[mb] = True
                    self.stats["PSkip"] += 1
                    mb += 1
                if not b.more_data():
                    break
            if mb >= nmb:
                raise ValueError("macroblock data past the picture")
            mbx, mby = mb % mb_w, mb // mb_w
            bx0, by0 = 4 * mbx, 4 * mby
            left_ok, top_ok = mbx > 0, mby > 0
            mb_type = b.ue()
            intra_type = None
            if is_p and mb_type >= 5:
                intra_type = mb_type - 5
            elif not is_p:
                intra_type = mb_type
            if intra_type is not None:
                mb_intra[mb] = True
                for yy in range(by0, by0 + 4):
                    for xx in range(bx0, bx0 + 4):
                        refi[yy][xx] = -1; refu[yy][xx] = -1; mvs[yy][xx] = (0, 0)
                if intra_type == 25:                          # I_PCM
                    b.align()
                    for yy in range(16):
                        for xx in range(16):
                            Y[16 * mby + yy, 16 * mbx + xx] = b.u(8)
                    for ci in range(2):
                        for yy in range(8):
                            for xx in range(8):
                                C[ci][8 * mby + yy, 8 * mbx + xx] = b.u(8)
                    for yy in range(by0, by0 + 4):
                        for xx in range(bx0, bx0 + 4):
                            nnzY[yy][xx] = 16
                    for ci in range(2):
                        for yy in range(2 * mby, 2 * mby + 2):
                            for xx in range(2 * mbx, 2 * mbx + 2):
                                nnzC[ci][yy][xx] = 16
                    mb_qp[mb] = 0
                    mb_done[mb] = True
                    self.stats["IPCM"] += 1
                    mb += 1
                    if not b.more_data():
                        break
                    continue
                if intra_type == 0:                           # I_4x4: the sixteen prediction modes (8.3.1.1)
                    self.stats["I4x4"] += 1
                    modes = []
                    for k in range(16):
                        x4, y4 = BLK_XY[k]
                        gx, gy = bx0 + x4, by0 + y4

                        def nmode(nx, ny):
                            if nx < 0 or ny < 0:
                                return None                   # not available -> DC for the whole derivation
                            m = ipm[ny][nx]
                            return m if m >= 0 else 2
                        ma, mb_ = nmode(gx - 1, gy), nmode(gx, gy - 1)
                        predm = 2 if (ma is None or mb_ is None) else min(ma, mb_)
                        if b.u(1):
                            mode = predm
                        else:
                            rem = b.u(3)
                            mode = rem if rem < predm else rem + 1
                        ipm[gy][gx] = mode
                        modes.append(mode)
                    chroma_mode = b.ue()
                    cbp = CBP_INTRA[b.ue()]
                else:
                    self.stats["I16x16"] += 1
                    i16mode = (intra_type - 1) % 4
                    cbp = (((intra_type - 1) // 4) % 3) << 4 | (15 if intra_type >= 13 else 0)
                    chroma_mode = b.ue()
                if cbp > 0 or intra_type != 0:
                    qp = _next_qp(qp, b.se())
                mb_qp[mb] = qp
                qpc_ = qpc_of(qp)
                x0, y0 = 16 * mbx, 16 * mby
                if intra_type == 0:
                    for k in range(16):
                        x4, y4 = BLK_XY[k]
                        gx, gy = bx0 + x4, by0 + y4
                        x, y = 4 * gx, 4 * gy
                        t_ok, l_ok = gy > 0, gx > 0
                        top = left = tl = None
                        if t_ok:
                            top = [int(v) for v in Y[y - 1, x:x + 4]]
                            # top-right: above (or above-right) macroblock row, or an earlier block of this macroblock
                            if y4 == 0:
                                tr_ok = gx + 1 < W4
                            else:
                                tr_ok = x4 < 3 and BLK_IDX[(x4 + 1, y4 - 1)] < k
                            top += [int(v) for v in Y[y - 1, x + 4:x + 8]] if tr_ok else [top[3]] * 4
                        if l_ok:
                            left = [int(v) for v in Y[y:y + 4, x - 1]]
                        if t_ok and l_ok:
                            tl = int(Y[y - 1, x - 1])
                        Y[y:y + 4, x:x + 4] = pred_intra4x4(modes[k], top, left, tl)
                        if cbp & (1 << (2 * (y4 >> 1) + (x4 >> 1))):
                            coef, tot = residual_block(b, nc_luma(gx, gy), 16)
                            nnzY[gy][gx] = tot
                            if tot:
                                add_luma_residual(gx, gy, _scale4x4(coef, qp, False))
                else:
                    top = [int(v) for v in Y[y0 - 1, x0:x0 + 16]] if top_ok else None
                    left = [int(v) for v in Y[y0:y0 + 16, x0 - 1]] if left_ok else None
                    if i16mode == 0:
                        Y[y0:y0 + 16, x0:x0 + 16] = np.array(top)[None, :]
                    elif i16mode == 1:
                        Y[y0:y0 + 16, x0:x0 + 16] = np.array(left)[:, None]
                    elif i16mode == 2:
                        if top_ok and left_ok:
                            v = (sum(top) + sum(left) + 16) >> 5
                        elif top_ok or left_ok:
                            v = (sum(top if top_ok else left) + 8) >> 4
                        else:
                            v = 128
                        Y[y0:y0 + 16, x0:x0 + 16] = v
                    else:
                        Y[y0:y0 + 16, x0:x0 + 16] = pred_plane(top, left, int(Y[y0 - 1, x0 - 1]), 16)
                    coef, _ = residual_block(b, nc_luma(bx0, by0), 16)
                    dc = _luma_dc(coef, qp)
                    for k in range(16):
                        x4, y4 = BLK_XY[k]
                        gx, gy = bx0 + x4, by0 + y4
                        if cbp & 15:
                            ac, tot = residual_block(b, nc_luma(gx, gy), 15)
                            nnzY[gy][gx] = tot
                            d = _scale4x4([0] + ac, qp, True)
                        else:
                            d = np.zeros((4, 4), np.int64)
                        d[0, 0] = dc[y4, x4]
                        if d.any():
                            add_luma_residual(gx, gy, d)
                intra_chroma_pred(mbx, mby, chroma_mode, left_ok, top_ok)
                chroma_residual(mbx, mby, cbp >> 4, qpc_)
            else:
                # ---- P macroblock (7.3.5.1 / 7.3.5.2, 8.4.1)
                for yy in range(by0, by0 + 4):
                    for xx in range(bx0, bx0 + 4):
                        refi[yy][xx] = -2                    # (already so: fresh picture state)
                nref = len(reflist)
                parts = []                                    # (bx, by, w4, h4, shape, part index)
                if mb_type == 0:
                    parts = [(0, 0, 4, 4, "16x16", 0)]; self.stats["P16x16"] += 1
                elif mb_type == 1:
                    parts = [(0, 0, 4, 2, "16x8", 0), (0, 2, 4, 2, "16x8", 1)]; self.stats["P16x8"] += 1
                elif mb_type == 2:
                    parts = [(0, 0, 2, 4, "8x16", 0), (2, 0, 2, 4, "8x16", 1)]; self.stats["P8x16"] += 1
                elif mb_type in (3, 4):
                    self.stats["P8x8"] += 1
                    sub = [b.ue() for _ in range(4)]
                    if max(sub) > 3:
                        raise ValueError("sub_mb_type")
                else:
                    raise ValueError("mb_type %d" % mb_type)
                if mb_type in (3, 4):
                    refs = [0] * 4
                    if mb_type == 3 and h["num_ref"] > 1:
                        refs = [b.te(h["num_ref"] - 1) for _ in range(4)]
                    groups = []
                    for k in range(4):
                        ox, oy = 2 * (k & 1), 2 * (k >> 1)
                        shp = [(2, 2)], [(2, 1), (2, 1)], [(1, 2), (1, 2)], [(1, 1)] * 4
                        offs = {0: [(0, 0)], 1: [(0, 0), (0, 1)], 2: [(0, 0), (1, 0)], 3: [(0, 0), (1, 0), (0, 1), (1, 1)]}[sub[k]]
                        for (w4, h4), (dx, dy) in zip(shp[sub[k]], offs):
                            groups.append((ox + dx, oy + dy, w4, h4, refs[k]))
                    mvd = [(b.se(), b.se()) for _ in groups]
                    for (px, py, w4, h4, ref), d in zip(groups, mvd):
                        if ref >= nref:
                            raise ValueError("ref_idx out of range")
                        pm = pred_mv(bx0 + px, by0 + py, w4, h4, ref, mb, "sub", 0)
                        mv = (pm[0] + d[0], pm[1] + d[1])
                        set_motion(bx0 + px, by0 + py, w4, h4, ref, mv)
                        inter_pred(mbx, mby, bx0 + px, by0 + py, w4, h4, ref, mv)
                else:
                    refs = [0] * len(parts)
                    if h["num_ref"] > 1:
                        refs = [b.te(h["num_ref"] - 1) for _ in parts]
                    mvd = [(b.se(), b.se()) for _ in parts]
                    for (px, py, w4, h4, shape, pi), ref, d in zip(parts, refs, mvd):
                        if ref >= nref:
                            raise ValueError("ref_idx out of range")
                        pm = pred_mv(bx0 + px, by0 + py, w4, h4, ref, mb, shape, pi)
                        mv = (pm[0] + d[0], pm[1] + d[1])
                        set_motion(bx0 + px, by0 + py, w4, h4, ref, mv)
                        inter_pred(mbx, mby, bx0 + px, by0 + py, w4, h4, ref, mv)
                cbp = CBP_INTER[b.ue()]
                if cbp > 0:
                    qp = _next_qp(qp, b.se())
                mb_qp[mb] = qp
                qpc_ = qpc_of(qp)
                for k in range(16):
                    x4, y4 = BLK_XY[k]
                    if cbp & (1 << (2 * (y4 >> 1) + (x4 >> 1))):
                        gx, gy = bx0 + x4, by0 + y4
                        coef, tot = residual_block(b, nc_luma(gx, gy), 16)
                        nnzY[gy][gx] = tot
                        if tot:
                            add_luma_residual(gx, gy, _scale4x4(coef, qp, False))
                chroma_residual(mbx, mby, cbp >> 4, qpc_)
            mb_done[mb] = True
            mb += 1
            if not b.more_data():
                break
        if mb != nmb:
            raise ValueError("slice ends after %d of %d macroblocks" % (mb, nmb))
        self.stats["slice_bits_left"] += b.end - b.pos
        if b.pos != b.end:
            raise ValueError("slice data ends %d bits before the stop bit" % (b.end - b.pos))
        pic.Y[:] = Y
        pic.C[0][:] = C[0]; pic.C[1][:] = C[1]
        pic.state = (nnzY, mvs, refu, mb_intra, mb_qp)

    # ---- deblocking filter (8.7), whole picture, macroblocks in raster order
    def _deblock(self, pic, h):
        nnzY, mvs, refu, mb_intra, mb_qp = pic.state
        mb_w, mb_h = self.mb_w, self.mb_h
        Y = pic.Y.astype(np.int64)
        C = [c.astype(np.int64) for c in pic.C]
        cqo = self.pps["chroma_qp_offset"]
        a_off, b_off = h["alpha_off"], h["beta_off"]

        def qpc_of(q):
            return QPC[min(max(q + cqo, 0), 51)]

        def bs_of(px, py, qx, qy, mb_edge, intra_p, intra_q):
            if intra_p or intra_q:
                return 4 if mb_edge else 3
            if nnzY[py][px] or nnzY[qy][qx]:
                return 2
            if refu[py][px] != refu[qy][qx]:
                return 1
            mp, mq = mvs[py][px], mvs[qy][qx]
            return 1 if (abs(mp[0] - mq[0]) >= 4 or abs(mp[1] - mq[1]) >= 4) else 0

        def filt(P, Q, bS, qpav, luma):
            """P, Q: [4 or 2][n] samples p0.. / q0.. across the edge (views are written back by the caller); bS: [n]."""
            ia = min(max(qpav + a_off, 0), 51)
            ib = min(max(qpav + b_off, 0), 51)
            alpha, beta = ALPHA[ia], BETA[ib]
            p0, p1, q0, q1 = P[0], P[1], Q[0], Q[1]
            on = (bS > 0) & (np.abs(p0 - q0) < alpha) & (np.abs(p1 - p0) < beta) & (np.abs(q1 - q0) < beta)
            if not on.any():
                return None
            tc0 = np.array([TC0[ia][k - 1] if 0 < k < 4 else 0 for k in bS], np.int64)
            if luma:
                p2, q2 = P[2], Q[2]
                ap, aq = np.abs(p2 - p0) < beta, np.abs(q2 - q0) < beta
                tc = tc0 + ap + aq
            else:
                tc = tc0 + 1
            delta = np.clip(((q0 - p0) * 4 + (p1 - q1) + 4) >> 3, -tc, tc)
            np0, nq0 = _clip8(p0 + delta), _clip8(q0 - delta)
            out = {"p0": np0, "q0": nq0}
            if luma:
                out["p1"] = np.where(ap, p1 + np.clip((p2 + ((p0 + q0 + 1) >> 1) - 2 * p1) >> 1, -tc0, tc0), p1)
                out["q1"] = np.where(aq, q1 + np.clip((q2 + ((p0 + q0 + 1) >> 1) - 2 * q1) >> 1, -tc0, tc0), q1)
                out["p2"], out["q2"] = p2, q2
            strong = bS == 4
            if strong.any():
                if luma:
                    p3, q3 = P[3], Q[3]
                    small = np.abs(p0 - q0) < ((alpha >> 2) + 2)
                    sp, sq = ap & small, aq & small
                    out["p0"] = np.where(strong, np.where(sp, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, (2 * p1 + p0 + q1 + 2) >> 2), out["p0"])
                    out["p1"] = np.where(strong, np.where(sp, (p2 + p1 + p0 + q0 + 2) >> 2, p1), out["p1"])
                    out["p2"] = np.where(strong & sp, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3, p2)
                    out["q0"] = np.where(strong, np.where(sq, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, (2 * q1 + q0 + p1 + 2) >> 2), out["q0"])
                    out["q1"] = np.where(strong, np.where(sq, (p0 + q0 + q1 + q2 + 2) >> 2, q1), out["q1"])
                    out["q2"] = np.where(strong & sq, (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3, q2)
                else:
                    out["p0"] = np.where(strong, (2 * p1 + p0 + q1 + 2) >> 2, out["p0"])
                    out["q0"] = np.where(strong, (2 * q1 + q0 + p1 + 2) >> 2, out["q0"])
            for k in list(out):
                src = {"p0": p0, "p1": p1, "q0": q0, "q1": q1, "p2": P[2] if luma else None, "q2": Q[2] if luma else None}[k]
                out[k] = np.where(on, out[k], src)
            return out

        for mb in range(mb_w * mb_h):
            mbx, mby = mb % mb_w, mb // mb_w
            bx0, by0 = 4 * mbx, 4 * mby
            x0, y0 = 16 * mbx, 16 * mby
            iq = mb_intra[mb]
            for vertical in (True, False):
                for e in range(4):
                    if e == 0 and ((vertical and mbx == 0) or (not vertical and mby == 0)):
                        continue
                    if vertical:
                        pmb = mb - 1 if e == 0 else mb
                        bS4 = [bs_of(bx0 + e - 1, by0 + k, bx0 + e, by0 + k, e == 0, mb_intra[pmb], iq) for k in range(4)]
                    else:
                        pmb = mb - mb_w if e == 0 else mb
                        bS4 = [bs_of(bx0 + k, by0 + e - 1, bx0 + k, by0 + e, e == 0, mb_intra[pmb], iq) for k in range(4)]
                    if not any(bS4):
                        continue
                    bS = np.repeat(np.array(bS4, np.int64), 4)
                    qpav = (mb_qp[pmb] + mb_qp[mb] + 1) >> 1
                    if vertical:
                        x = x0 + 4 * e
                        P = [Y[y0:y0 + 16, x - 1 - k] for k in range(4)]
                        Q = [Y[y0:y0 + 16, x + k] for k in range(4)]
                    else:
                        y = y0 + 4 * e
                        P = [Y[y - 1 - k, x0:x0 + 16] for k in range(4)]
                        Q = [Y[y + k, x0:x0 + 16] for k in range(4)]
                    o = filt([p.copy() for p in P], [q.copy() for q in Q], bS, qpav, True)
                    if o is not None:
                        for k in range(3):
                            P[k][:] = o["p%d" % k]; Q[k][:] = o["q%d" % k]
                    if e % 2 == 0:                            # chroma edges 0 and 1 lie on luma edges 0 and 2
                        bSc = np.repeat(np.array(bS4, np.int64), 2)
                        qpc_av = (qpc_of(mb_qp[pmb]) + qpc_of(mb_qp[mb]) + 1) >> 1
                        for ci in range(2):
                            Pl = C[ci]
                            if vertical:
                                x = 8 * mbx + 2 * e
                                Pc = [Pl[8 * mby:8 * mby + 8, x - 1 - k] for k in range(2)]
                                Qc = [Pl[8 * mby:8 * mby + 8, x + k] for k in range(2)]
                            else:
                                y = 8 * mby + 2 * e
                                Pc = [Pl[y - 1 - k, 8 * mbx:8 * mbx + 8] for k in range(2)]
                                Qc = [Pl[y + k, 8 * mbx:8 * mbx + 8] for k in range(2)]
                            o = filt([p.copy() for p in Pc], [q.copy() for q in Qc], bSc, qpc_av, False)
                            if o is not None:
                                Pc[0][:] = o["p0"]; Qc[0][:] = o["q0"]
        pic.Y[:] = Y
        pic.C[0][:] = C[0]; pic.C[1][:] = C[1]


def decode_mp4(path, with_chroma=False, progress=None):
    """All pictures of the first avc1 track in decoding (= output: Baseline, no B pictures) order: uint8 [n, H, W] luma
    (and, with_chroma, the two chroma planes [n, H / 2, W / 2] each)."""
    buf = open(path, "rb").read()
    sps_l, pps_l, nls, samples, _sync = demux_avc(buf)
    sps, pps = parse_sps(sps_l[0]), parse_pps(pps_l[0])
    dec = Decoder(sps, pps)
    ys, cbs, crs = [], [], []
    for i, s in enumerate(samples):
        p = 0
        while p + nls <= len(s):
            n = int.from_bytes(s[p:p + nls], "big")
            nal = s[p + nls:p + nls + n]
            p += nls + n
            t = nal[0] & 31
            if t == 7:
                dec.sps = parse_sps(nal)
            elif t == 8:
                dec.pps = parse_pps(nal)
            pic = dec.decode_nal(nal)
            if pic is not None:
                ys.append(pic.Y.copy()); cbs.append(pic.C[0].copy()); crs.append(pic.C[1].copy())
        if progress:
            progress(i, len(samples))
    cl, cr, ct, cb = sps["crop"]
    H, W = 16 * sps["mb_h"], 16 * sps["mb_w"]
    Yv = np.stack(ys)[:, 2 * ct:H - 2 * cb, 2 * cl:W - 2 * cr]
    if not with_chroma:
        return Yv
    return Yv, np.stack(cbs)[:, ct:H // 2 - cb, cl:W // 2 - cr], np.stack(crs)[:, ct:H // 2 - cb, cl:W // 2 - cr], dec.stats


def yuv420_to_bgr(y, cb, cr):
    """Decoded YCbCr 4:2:0 planes ([n, H, W], [n, H/2, W/2] x 2, uint8) -> BGR frames uint8 [n, H, W, 3], the layout
    cv::VideoCapture::read hands the reference (ParallelVideoPyr.cpp:776).  ITU-R BT.601 limited range, the usual 8-bit integer
    form (298 / 409 / 100 / 208 / 516, >> 8), chroma replicated 2 x 2.  UNPINNED like the decoder: the reference's route is
    libswscale, whose chroma up-sampling filter and rounding are not reproduced here."""
    c = y.astype(np.int32) - 16
    d = np.repeat(np.repeat(cb.astype(np.int32) - 128, 2, axis=-2), 2, axis=-1)
    e = np.repeat(np.repeat(cr.astype(np.int32) - 128, 2, axis=-2), 2, axis=-1)
    r = (298 * c + 409 * e + 128) >> 8
    g = (298 * c - 100 * d - 208 * e + 128) >> 8
    b = (298 * c + 516 * d + 128) >> 8
    return np.clip(np.stack([b, g, r], axis=-1), 0, 255).astype(np.uint8)
