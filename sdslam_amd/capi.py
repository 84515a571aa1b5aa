"""ctypes binding of libsdslam_hip.so (C ABI: include/sdslam_hip.h).

Host-side mirror of the reference interface for the hot path: the class/method names follow
the reference (`ORBextractor.__call__` == `ORBextractor::operator()`, reference
src/ORBextractor.h:38-70).  Nothing here computes: every method forwards to the HIP library
and raises SdError on a non-zero status.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28

SD_OK = 0


class SdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sdslam_hip status {code}: {msg}")
        self.code = code


def lib_path() -> str:
    return os.path.join(_HERE, "libsdslam_hip.so")


_lib = None


def lib():
    """Load libsdslam_hip.so.  Fails loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        p = lib_path()
        if not os.path.exists(p):
            raise SdError(-1, f"{p} is missing: build it with `python -m sdslam_amd.build` "
                              "(hipcc, gfx950). There is no CPU fallback.")
        L = C.CDLL(p)
        L.sd_last_error.restype = C.c_char_p
        L.sd_version.restype = C.c_char_p
        L.sd_orb_stage_name.restype = C.c_char_p
        L.sd_orb_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_void_p)]
        L.sd_orb_destroy.argtypes = [C.c_void_p]
        L.sd_orb_plan_info.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                       C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sd_orb_extract_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.sd_orb_extract_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                  C.c_size_t]
        L.sd_orb_download.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.sd_orb_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_void_p]
        L.sd_orb_level_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.sd_orb_level_copy.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.sd_orb_debug_blurred.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.sd_orb_debug_cell_counts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.sd_orb_debug_level_keys.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.sd_orb_scale_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.sd_orb_features_per_level.argtypes = [C.c_void_p, C.c_void_p]
        L.sd_orb_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.sd_orb_sync.argtypes = [C.c_void_p]
        L.sd_orb_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.sd_orb_stage_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.sd_orb_stage_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.sd_orb_stage_name.argtypes = [C.c_int]
        L.sd_orb_levels.argtypes = [C.c_void_p]
        L.sd_dev_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        L.sd_dev_free.argtypes = [C.c_void_p]
        L.sd_dev_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.sd_dev_download.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.sd_hamming.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def _check(rc):
    if rc != SD_OK:
        raise SdError(rc, lib().sd_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def device_count() -> int:
    return int(lib().sd_device_count())


def hamming(a: np.ndarray, b: np.ndarray) -> int:
    """ORBmatcher::DescriptorDistance (reference src/ORBmatcher.cc:1459-1473)."""
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    assert a.size == 32 and b.size == 32
    return int(lib().sd_hamming(_p(a), _p(b)))


def plan_info(nfeatures, scale_factor, nlevels, th_fast, w, h):
    """Geometry the extractor uses for a w x h frame (host-only; no GPU needed)."""
    lv = np.zeros((nlevels, 8), np.int32)
    zones = np.zeros((65536, 6), np.int32)
    n = C.c_int32()
    nbytes = C.c_uint64()
    _check(lib().sd_orb_plan_info(nfeatures, scale_factor, nlevels, th_fast, w, h, _p(lv), _p(zones), len(zones),
                                  C.byref(n), C.byref(nbytes)))
    return dict(levels=lv, cells=zones[:n.value].copy(), bytes_per_frame=nbytes.value)


class DeviceBuffer:
    """Raw HBM allocation (for harnesses that stage inputs on the device themselves)."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        self.nbytes = nbytes
        _check(lib().sd_dev_alloc(nbytes, C.byref(self.ptr)))

    def upload(self, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(lib().sd_dev_upload(self.ptr, _p(arr), arr.nbytes))

    def free(self):
        if self.ptr:
            lib().sd_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class ORBextractor:
    """Mirror of SD_SLAM::ORBextractor (reference src/ORBextractor.h:38-70) over sd_orb_*."""

    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, thFAST=20, max_w=640, max_h=480, max_batch=1,
                 device=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.nfeatures, self.nlevels, self.max_batch = nfeatures, nlevels, max_batch
        _check(self.L.sd_orb_create(nfeatures, scaleFactor, nlevels, thFAST, max_w, max_h, max_batch, device,
                                    C.byref(self.h)))
        self.cap = nfeatures

    def close(self):
        if self.h:
            self.L.sd_orb_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- getters (GetLevels / GetScaleFactors / ...) ---
    def GetLevels(self):
        return self.nlevels

    def _tables(self):
        n = self.nlevels
        arrs = [np.zeros(n, np.float32) for _ in range(4)]
        _check(self.L.sd_orb_scale_tables(self.h, *[_p(a) for a in arrs]))
        return arrs

    def GetScaleFactors(self):
        return self._tables()[0]

    def GetInverseScaleFactors(self):
        return self._tables()[1]

    def GetScaleSigmaSquares(self):
        return self._tables()[2]

    def GetInverseScaleSigmaSquares(self):
        return self._tables()[3]

    def features_per_level(self):
        q = np.zeros(self.nlevels, np.int32)
        _check(self.L.sd_orb_features_per_level(self.h, _p(q)))
        return q

    # --- operator() ---
    def __call__(self, image: np.ndarray):
        """(keypoints[KP_DTYPE], descriptors[N,32]) of one grey frame; pyramid stays on device."""
        image = np.ascontiguousarray(image, np.uint8)
        if image.size == 0:
            return np.zeros(0, KP_DTYPE), np.zeros((0, 32), np.uint8)
        k, d, n = self.extract_batch(image[None])
        return k[0, :n[0]].copy(), d[0, :n[0]].copy()

    def extract_batch(self, images: np.ndarray):
        images = np.ascontiguousarray(images, np.uint8)
        B, H, W = images.shape
        kps = np.zeros((B, self.cap), KP_DTYPE)
        desc = np.zeros((B, self.cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        _check(self.L.sd_orb_extract_batch(self.h, _p(images), B, W, H, images.strides[1], images.strides[0],
                                           _p(kps), _p(desc), self.cap, _p(n)))
        return kps, desc, n

    def extract_batch_device(self, d_ptr, B, W, H, stride=None, frame_stride=None):
        stride = stride or W
        frame_stride = frame_stride or stride * H
        _check(self.L.sd_orb_extract_batch_device(self.h, d_ptr, B, W, H, stride, frame_stride))

    def download(self, frame0=0, n_frames=1):
        kps = np.zeros((n_frames, self.cap), KP_DTYPE)
        desc = np.zeros((n_frames, self.cap, 32), np.uint8)
        n = np.zeros(n_frames, np.int32)
        _check(self.L.sd_orb_download(self.h, frame0, n_frames, _p(kps), _p(desc), self.cap, _p(n)))
        return kps, desc, n

    # --- pyramid / diagnostics ---
    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        _check(self.L.sd_orb_level_info(self.h, level, C.byref(w), C.byref(h)))
        return w.value, h.value

    def level(self, level, frame=0, padded=False):
        w, h = self.level_size(level)
        if padded:
            w, h = w + 38, h + 38
        out = np.zeros((h, w), np.uint8)
        _check(self.L.sd_orb_level_copy(self.h, frame, level, int(padded), _p(out), w))
        return out

    def blurred(self, level, frame=0):
        w, h = self.level_size(level)
        out = np.zeros((h, w), np.uint8)
        _check(self.L.sd_orb_debug_blurred(self.h, frame, level, _p(out), w))
        return out

    def cell_counts(self, level, frame=0):
        out = np.zeros(4096, np.int32)
        n = C.c_int()
        _check(self.L.sd_orb_debug_cell_counts(self.h, frame, level, _p(out), 4096, C.byref(n)))
        return out[:n.value].copy()

    def level_keys(self, level, frame=0):
        out = np.zeros(max(self.cap, 1), np.uint32)
        n = C.c_int()
        _check(self.L.sd_orb_debug_level_keys(self.h, frame, level, _p(out), len(out), C.byref(n)))
        return out[:n.value].copy()

    # --- stream / timing ---
    def set_stream(self, stream_ptr):
        _check(self.L.sd_orb_set_stream(self.h, C.c_void_p(stream_ptr) if stream_ptr else None))

    def sync(self):
        _check(self.L.sd_orb_sync(self.h))

    def set_profiling(self, on=True):
        _check(self.L.sd_orb_set_profiling(self.h, int(on)))

    def stage_names(self):
        return [self.L.sd_orb_stage_name(i).decode() for i in range(self.L.sd_orb_num_stages())]

    def stage_ms(self):
        n = self.L.sd_orb_num_stages()
        ms = np.zeros(n, np.float32)
        _check(self.L.sd_orb_stage_ms(self.h, _p(ms), n))
        return ms

    def stage_bytes(self):
        n = self.L.sd_orb_num_stages()
        b = np.zeros(n, np.float64)
        _check(self.L.sd_orb_stage_bytes(self.h, _p(b), n))
        return b
