"""ctypes binding of the CPU ORACLE (oracle/build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (sdslam_amd)."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build(fast: bool = False) -> str:
    target = "build/liboracle_fast.so" if fast else "build/liboracle.so"
    subprocess.check_call(["make", "-s", "-C", _HERE, target])
    return os.path.join(_HERE, target)


_libs = {}


def lib(fast: bool = False):
    if fast not in _libs:
        path = os.path.join(_HERE, "build", "liboracle_fast.so" if fast else "liboracle.so")
        if not os.path.exists(path):
            build(fast)
        L = C.CDLL(path)
        L.orc_orb_create.restype = C.c_void_p
        L.orc_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int]
        L.orc_orb_destroy.argtypes = [C.c_void_p]
        L.orc_fast_atan2.restype = C.c_float
        L.orc_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orc_ic_angle.restype = C.c_float
        L.orc_ic_angle.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.orc_brief.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p]
        _libs[fast] = L
    return _libs[fast]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OrbOracle:
    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, th_fast=20, fast_build=False):
        self.L = lib(fast_build)
        self.nlevels = nlevels
        self.h = C.c_void_p(self.L.orc_orb_create(nfeatures, scale_factor, nlevels, th_fast))
        self.cap = max(4 * nfeatures, 64)

    def __del__(self):
        try:
            self.L.orc_orb_destroy(self.h)
        except Exception:
            pass

    def tables(self):
        n = self.nlevels
        sf, isf, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orc_orb_tables(self.h, _p(sf), _p(isf), _p(s2), _p(is2), _p(quota), _p(umax))
        return dict(sf=sf, inv_sf=isf, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        kps = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = self.L.orc_orb_extract(self.h, _p(img), img.shape[1], img.shape[0], img.strides[0],
                                   _p(kps), _p(desc), self.cap)
        assert n <= self.cap
        return kps[:n].copy(), desc[:n].copy()

    def level(self, l, padded=False):
        w, h = C.c_int(), C.c_int()
        assert self.L.orc_orb_level_info(self.h, l, C.byref(w), C.byref(h)) == 0
        W, H = (w.value + 38, h.value + 38) if padded else (w.value, h.value)
        out = np.zeros((H, W), np.uint8)
        self.L.orc_orb_level_copy(self.h, l, int(padded), _p(out), W)
        return out

    def blurred(self, l):
        w, h = C.c_int(), C.c_int()
        self.L.orc_orb_level_info(self.h, l, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        if self.L.orc_orb_blurred_copy(self.h, l, _p(out), w.value) != 0:
            return None
        return out

    def level_keypoints(self, l):
        kps = np.zeros(self.cap, KP_DTYPE)
        n = self.L.orc_orb_level_keypoints(self.h, l, _p(kps), self.cap)
        return kps[:n].copy()

    def cell_totals(self, l):
        out = np.zeros(4096, np.int32)
        n = self.L.orc_orb_cell_totals(self.h, l, _p(out), 4096)
        return out[:n].copy()

    def ic_angle(self, img, x, y):
        img = np.ascontiguousarray(img, np.uint8)
        return float(self.L.orc_ic_angle(self.h, _p(img), img.strides[0], float(x), float(y)))

    def brief(self, blurred, x, y, angle):
        blurred = np.ascontiguousarray(blurred, np.uint8)
        d = np.zeros(32, np.uint8)
        self.L.orc_brief(self.h, _p(blurred), blurred.strides[0], float(x), float(y), float(angle), _p(d))
        return d


# ---- stage-level helpers -------------------------------------------------------------
def fast(img, threshold=20, nonmax=True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    kps = np.zeros(max(cap, 1), KP_DTYPE)
    n = lib().orc_fast(_p(img), img.shape[1], img.shape[0], img.strides[0], threshold, int(nonmax), _p(kps), cap)
    return kps[:n].copy()


def fast_score(img, x, y, threshold=20):
    img = np.ascontiguousarray(img, np.uint8)
    ptr = C.c_void_p(img.ctypes.data + y * img.strides[0] + x)
    return lib().orc_fast_score(ptr, img.strides[0], threshold)


def resize_linear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_linear(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def border101(src, b=19):
    src = np.ascontiguousarray(src, np.uint8)
    h, w = src.shape
    dst = np.zeros((h + 2 * b, w + 2 * b), np.uint8)
    lib().orc_border101(_p(src), w, h, src.strides[0], _p(dst), w + 2 * b, b)
    return dst


def blur7(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().orc_blur7(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def gauss_taps():
    k = np.zeros(7, np.int32)
    lib().orc_gauss_taps(_p(k))
    return k


def fast_atan2(y, x):
    return float(lib().orc_fast_atan2(float(y), float(x)))


def retain_best(kps, n_points):
    buf = np.ascontiguousarray(kps.copy())
    n = lib().orc_retain_best(_p(buf), len(buf), n_points)
    return buf[:n].copy()
