"""The one reference translation unit that compiles without OpenCV/Eigen -- src/extra/utils.cc
(SD_SLAM::Random) -- is built into oracle/_ref/libref_utils.so (oracle/Makefile `ref`, build
container only) and pins the oracle's restatement of the RANSAC draw."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_utils.so")


@pytest.mark.skipif(not os.path.exists(REF), reason="oracle/_ref not built (reference absent on this machine)")
def test_random_matches_reference_binary():
    from sdslam_amd.synth import glibc_rand_stream
    ref = C.CDLL(REF)
    fn = ref._ZN7SD_SLAM6RandomEii          # int SD_SLAM::Random(int, int)
    fn.restype = C.c_int
    fn.argtypes = [C.c_int, C.c_int]
    libc = C.CDLL("libc.so.6")
    libc.srand(1)                            # the reference never seeds: default state == seed 1
    rs = glibc_rand_stream(4000)
    sizes = np.random.default_rng(0).integers(1, 2000, size=4000)
    for r, d in zip(rs.tolist(), sizes.tolist()):
        mn = 3
        exp = int((float(r) / (2147483647.0 + 1.0)) * d + mn)   # oracle / device formula
        assert fn(mn, mn + d - 1) == exp
