#!/usr/bin/env python3
"""Re-encode the reference's own test fixtures into small binary files under tests/golden/.

Run in the build container only (needs /root/reference and Pillow); the GPU box never runs it.
Nothing here is source text of the reference: inputs are its test images / golden value tables.

  table1_640x448.u8 / table2_640x448.u8
      gray decode of testdata/table{1,2}.jpg cropped Rect(0,0,640,448)  (SerialPyr.cpp:223-226).
      OpenCV 2.4 imread(path, 0) asks libjpeg for JCS_GRAYSCALE (the Y plane); Pillow's
      Image.draft('L', size) requests the same libjpeg output colour space, whereas
      .convert('L') from RGB differs by up to 2 grey levels.
  goldens.npz
      U2Test/V2Test (single-level LK), UPyrTest/VPyrTest (7-level pyramid), V2 (LK after the
      in-place warp) as int32 = round(value * 1e6): the text files print "%lf" = 6 decimals, so
      this is lossless (SerialPyr.cpp:108-119 saveMat).
"""
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def decode_gray(path):
    im = Image.open(path)
    im.draft("L", im.size)
    assert im.mode == "L", im.mode
    return np.asarray(im, dtype=np.uint8)


def main():
    for name in ("table1", "table2"):
        g = decode_gray(os.path.join(REF, "testdata", name + ".jpg"))
        crop = np.ascontiguousarray(g[0:448, 0:640])
        crop.tofile(os.path.join(OUT, f"{name}_640x448.u8"))
        print(name, g.shape, "->", crop.shape, "mean", crop.mean())

    # BASELINE config 0b input: hand1/hand2 (3264x2448, unused by every reference source, no golden).
    # A centred 816x612 crop (= the 3264x2448 frame / 4 in each dimension, so 3 levels are legal:
    # 816 = 2^4*51, 612 = 2^2*153) keeps the fixture small; EXIF orientation ignored like OpenCV 2.4.
    for name in ("hand1", "hand2"):
        g = decode_gray(os.path.join(REF, "testdata", name + ".jpg"))
        assert g.shape == (2448, 3264), g.shape
        crop = np.ascontiguousarray(g[918:918 + 612, 1224:1224 + 816])
        crop.tofile(os.path.join(OUT, f"{name}_816x612.u8"))
        print(name, g.shape, "->", crop.shape, "mean", crop.mean())

    arrays = {}
    for name in ("U2Test", "V2Test", "UPyrTest", "VPyrTest", "V2"):
        a = np.loadtxt(os.path.join(REF, name + ".txt"), dtype=np.float64)
        assert a.shape == (448, 640), a.shape
        q = np.rint(a * 1e6)
        assert np.all(np.abs(q) < 2**31 - 1)
        assert np.max(np.abs(q / 1e6 - a)) < 1e-9      # lossless for 6-decimal text
        arrays[name] = q.astype(np.int32)
        print(name, "min", a.min(), "max", a.max())
    np.savez_compressed(os.path.join(OUT, "goldens.npz"), **arrays)
    print("wrote", os.path.getsize(os.path.join(OUT, "goldens.npz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
