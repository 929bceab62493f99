"""Decode /root/reference/pacman.mp4 (BASELINE config 3's input) with the in-repo H.264 ingest
(optical-flow_amd/h264_ingest.py -- this image holds no video decoder) and commit the luma frames as a fixture, so that the GPU box
(where /root/reference does not exist) runs config 3 on the real stream:

    python tests/golden/make_pacman_fixture.py        # -> tests/golden/pacman_240x160_y.npz
                                                      #    Y: uint8 [245, 160, 240]; Cb, Cr: uint8 [245, 80, 120] (4:2:0)

Data only: decoded pixels of a data file the reference ships, no reference source.  The pixel values are UNPINNED (no reference decoder
here to compare with): see the header of h264_ingest.py for what is checked instead.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "optical-flow_amd"))
import h264_ingest  # noqa: E402

if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/pacman.mp4"
    Y, Cb, Cr, _stats = h264_ingest.decode_mp4(src, with_chroma=True)
    assert Y.shape == (245, 160, 240) and Y.dtype == np.uint8, Y.shape
    np.savez_compressed(os.path.join(HERE, "pacman_240x160_y.npz"), Y=Y, Cb=Cb, Cr=Cr)
    print("frames", Y.shape, "mean luma %.2f" % Y.mean())
