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
    # SD_LIB: an instrumented build of the same library (profiling tools only; e.g. -DSD_PNP_PROF stage timers)
    return os.environ.get("SD_LIB") or os.path.join(_HERE, "libsdslam_hip.so")


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
        L.sd_set_option.argtypes = [C.c_char_p, C.c_int]
        L.sd_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.sd_option_name.restype = C.c_char_p
        L.sd_option_name.argtypes = [C.c_int]
        _lib = L
    return _lib


def _check(rc):
    if rc != SD_OK:
        raise SdError(rc, lib().sd_last_error().decode())


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def set_option(name: str, value: int):
    """sd_set_option: process-wide tuning / test switch (include/sdslam_hip.h lists them; the library reads no environment)."""
    _check(lib().sd_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_int(0)
    _check(lib().sd_get_option(name.encode(), C.byref(v)))
    return v.value


def option_names():
    L = lib()
    return [L.sd_option_name(i).decode() for i in range(L.sd_option_count())]


class options:
    """with options({"extract.use_graph": 1}): ... -- set for the block, restore afterwards."""

    def __init__(self, values):
        self.values, self.saved = dict(values), {}

    def __enter__(self):
        for k, v in self.values.items():
            self.saved[k] = get_option(k)
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            set_option(k, v)
        return False


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


def pinned_array(shape, dtype=np.uint8):
    """numpy array over page-locked host memory (sd_host_alloc); keep the returned owner alive."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    L = lib()
    p = C.c_void_p()
    L.sd_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    _check(L.sd_host_alloc(n, C.byref(p)))
    buf = (C.c_uint8 * n).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dtype).reshape(shape)

    class _Owner:
        def __init__(self, ptr):
            self.ptr = ptr

        def __del__(self):
            try:
                L.sd_host_free.argtypes = [C.c_void_p]
                L.sd_host_free(self.ptr)
            except Exception:
                pass
    return arr, _Owner(p)


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

    def set_distortion(self, fx, fy, cx, cy, k1, k2=0.0, p1=0.0, p2=0.0, k3=0.0):
        """Frame::UndistortKeyPoints parameters (mK as CV_32F, mDistCoef); k1 == 0 disables."""
        self.L.sd_orb_set_distortion.argtypes = [C.c_void_p] + [C.c_float] * 9
        _check(self.L.sd_orb_set_distortion(self.h, fx, fy, cx, cy, k1, k2, p1, p2, k3))

    def download_undistorted(self, frame0=0, n_frames=1):
        kps = np.zeros((n_frames, self.cap), KP_DTYPE)
        self.L.sd_orb_download_undistorted.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _check(self.L.sd_orb_download_undistorted(self.h, frame0, n_frames, _p(kps), self.cap))
        return kps

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

    def stream_fence(self, stream_ptr, direction):
        """sd_orb_stream_fence: 0 = the caller's stream waits for the extractions queued so far, 1 = later extractions wait for it."""
        self.L.sd_orb_stream_fence.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _check(self.L.sd_orb_stream_fence(self.h, C.c_void_p(stream_ptr), direction))

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


def _cm(T):
    """4x4 (row-major math) -> 16 doubles column-major (Eigen::Matrix4d::data())."""
    return np.ascontiguousarray(np.asarray(T, np.float64).T).ravel()


def _from_cm(v):
    return np.asarray(v, np.float64).reshape(4, 4).T.copy()


class Tracker:
    """Batched TrackWithMotionModel context over two resident extractors (sd_track_*):
    ImageAlign.ComputePose -> ORBmatcher.SearchByProjection -> PnPsolver.iterate
    (reference src/Tracking.cc:654-718)."""

    MODE_FRAME, MODE_KF, MODE_KF_FAST, MODE_KFKF = 0, 1, 2, 3

    def __init__(self, cur: ORBextractor, ref: ORBextractor, max_points=1000, max_batch=1, pnp_max_iterations=300):
        self.L = lib()
        L = self.L
        L.sd_track_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.sd_track_destroy.argtypes = [C.c_void_p]
        L.sd_track_set_camera.argtypes = [C.c_void_p] + [C.c_float] * 9
        L.sd_track_set_last.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
        L.sd_track_set_poses.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.sd_track_set_rand.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.sd_track_align.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.sd_track_match.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int]
        L.sd_track_pnp.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
        L.sd_track_get_align.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.sd_track_get_matches.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.sd_track_get_pnp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.sd_track_set_profiling.argtypes = [C.c_void_p, C.c_int]
        L.sd_track_stage_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        self.cur, self.ref = cur, ref
        self.M, self.B, self.cap = max_points, max_batch, cur.cap
        self.h = C.c_void_p()
        _check(L.sd_track_create(cur.h, ref.h, max_points, max_batch, pnp_max_iterations, C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.sd_track_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, fx, fy, cx, cy, bf=0.0, bounds=(0.0, 640.0, 0.0, 480.0)):
        _check(self.L.sd_track_set_camera(self.h, fx, fy, cx, cy, bf, *[float(b) for b in bounds]))

    def set_last(self, frame0, cases):
        """cases: list of dict(valid, Xw, desc, octave, angle, obs) (one per frame)."""
        n, M = len(cases), self.M
        n_last = np.zeros(n, np.int32)
        valid = np.zeros((n, M), np.uint8)
        Xw = np.zeros((n, M, 3), np.float64)
        desc = np.zeros((n, M, 32), np.uint8)
        octave = np.zeros((n, M), np.int32)
        angle = np.zeros((n, M), np.float32)
        obs = np.zeros((n, M), np.int32)
        for i, c in enumerate(cases):
            k = len(c["valid"])
            assert k <= M
            n_last[i] = k
            valid[i, :k], Xw[i, :k], desc[i, :k] = c["valid"], c["Xw"], c["desc"]
            octave[i, :k], angle[i, :k], obs[i, :k] = c["octave"], c["angle"], c["obs"]
        _check(self.L.sd_track_set_last(self.h, frame0, n, _p(n_last), _p(valid), _p(Xw), _p(desc), _p(octave), _p(angle),
                                        _p(obs)))

    def set_poses(self, frame0, T_ref_list, T_cur_list):
        n = len(T_ref_list)
        a = np.stack([_cm(T) for T in T_ref_list])
        b = np.stack([_cm(T) for T in T_cur_list])
        _check(self.L.sd_track_set_poses(self.h, frame0, n, _p(a), _p(b)))

    def set_rand(self, frame0, rand_values):
        r = np.ascontiguousarray(rand_values, np.int32)
        assert r.ndim == 2
        _check(self.L.sd_track_set_rand(self.h, frame0, r.shape[0], _p(r), r.shape[1]))

    def set_uright(self, frame0, uright):
        u = np.ascontiguousarray(uright, np.float32)
        self.L.sd_track_set_uright.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _check(self.L.sd_track_set_uright(self.h, frame0, u.shape[0], _p(u), u.shape[1]))

    def stereo_from_depth(self, depth):
        """Frame::ComputeStereoFromRGBD for the current frames; depth: [n, H, W] float32."""
        d = np.ascontiguousarray(depth, np.float32)
        self.L.sd_track_stereo_from_depth.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t]
        _check(self.L.sd_track_stereo_from_depth(self.h, d.shape[0], _p(d), d.shape[2], d.shape[1], d.shape[2], d.shape[1] * d.shape[2]))

    def get_stereo(self, frame0, n):
        u = np.zeros((n, self.cap), np.float32)
        d = np.zeros((n, self.cap), np.float32)
        self.L.sd_track_get_stereo.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _check(self.L.sd_track_get_stereo(self.h, frame0, n, _p(u), _p(d), self.cap))
        return u, d

    def set_local(self, frame0, pts_list, kp_claimed=None):
        """pts_list: per frame dict(cand, Xw, normal, min_dist, max_dist, mf_max_dist, desc, obs) (TrackLocalMap's points)."""
        n, M = len(pts_list), self.M
        nl = np.array([len(p["cand"]) for p in pts_list], np.int32)

        def pad(key, shape, dt):
            a = np.zeros((n,) + shape, dt)
            for i, p in enumerate(pts_list):
                v = np.asarray(p[key], dt)
                a[i, :len(v)] = v
            return a
        cand, Xw, nr = pad("cand", (M,), np.uint8), pad("Xw", (M, 3), np.float64), pad("normal", (M, 3), np.float64)
        mn, mx, mf = pad("min_dist", (M,), np.float32), pad("max_dist", (M,), np.float32), pad("mf_max_dist", (M,), np.float32)
        md, ob = pad("desc", (M, 32), np.uint8), pad("obs", (M,), np.int32)
        kc = None
        if kp_claimed is not None:
            kc = np.zeros((n, self.cap), np.uint8)
            for i, k in enumerate(kp_claimed):
                kc[i, :len(k)] = k
        self.L.sd_track_set_local.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 10
        _check(self.L.sd_track_set_local(self.h, frame0, n, _p(nl), _p(cand), _p(Xw), _p(nr), _p(mn), _p(mx), _p(mf), _p(md), _p(ob),
                                         _p(kc) if kc is not None else None))

    def match_local(self, n_frames, th=1.0, nnratio=0.8, cos_limit=0.5):
        self.L.sd_track_match_local.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float]
        _check(self.L.sd_track_match_local(self.h, n_frames, float(th), float(nnratio), float(cos_limit)))

    def get_local(self, frame0, n):
        M = self.M
        lm = np.zeros((n, self.cap), np.int32)
        nm = np.zeros(n, np.int32)
        iv = np.zeros((n, M), np.uint8)
        pr = np.zeros((n, M, 3), np.float32)
        lv = np.zeros((n, M), np.int32)
        cs = np.zeros((n, M), np.float32)
        self.L.sd_track_get_local.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int] + [C.c_void_p] * 5
        _check(self.L.sd_track_get_local(self.h, frame0, n, _p(lm), self.cap, _p(nm), _p(iv), _p(pr), _p(lv), _p(cs)))
        return dict(match=lm, n=nm, in_view=iv.astype(bool), proj=pr, level=lv, cos=cs)

    def pose_opt(self, n_frames, source=0):
        self.L.sd_track_pose_opt.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _check(self.L.sd_track_pose_opt(self.h, n_frames, int(source)))

    def get_pose_opt(self, frame0, n):
        T = np.zeros((n, 16))
        out = np.zeros((n, self.cap), np.uint8)
        info = np.zeros((n, 8), np.int32)
        self.L.sd_track_get_pose_opt.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _check(self.L.sd_track_get_pose_opt(self.h, frame0, n, _p(T), _p(out), self.cap, _p(info)))
        return dict(T=[_from_cm(t) for t in T], outlier=out.astype(bool), n_initial=info[:, 0], n_bad=info[:, 1], rounds=info[:, 2],
                    iterations=info[:, 3], lm_trials=info[:, 4], n_inliers=info[:, 5])

    def track_with_motion_model(self, n_frames, th=15.0, mono=True, align_mode=0, min_matches=20, min_inliers=10):
        """Tracking::TrackWithMotionModel (src/Tracking.cc:654-718) for the batch; align_mode 1 = against the reference
        keyframe (TrackReferenceKeyFrame), -1 = align_image_ off."""
        self.L.sd_track_with_motion_model.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        _check(self.L.sd_track_with_motion_model(self.h, n_frames, int(align_mode), th, int(mono), min_matches, min_inliers))

    def get_tracked(self, frame0, n):
        info = np.zeros((n, 4), np.int32)
        self.L.sd_track_get_tracked.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _check(self.L.sd_track_get_tracked(self.h, frame0, n, _p(info)))
        return dict(status=info[:, 0], nmatches=info[:, 1], nmatches_map=info[:, 2], retried=info[:, 3])

    def track_local_map(self, n_frames, th=1.0, nnratio=0.8, cos_limit=0.5, min_inliers=30):
        """Tracking::TrackLocalMap (src/Tracking.cc:720-751) on top of the frame-to-frame matches and the current pose."""
        self.L.sd_track_local_map.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]
        _check(self.L.sd_track_local_map(self.h, n_frames, th, nnratio, cos_limit, min_inliers))

    def get_local_map(self, frame0, n):
        mm = np.zeros((n, self.cap), np.int32)
        info = np.zeros((n, 4), np.int32)
        self.L.sd_track_get_local_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _check(self.L.sd_track_get_local_map(self.h, frame0, n, _p(mm), self.cap, _p(info)))
        return dict(match=mm, status=info[:, 0], n_points=info[:, 1], n_inliers=info[:, 2], n_local=info[:, 3])

    def set_current_broadcast(self, cur_frame):
        """One current frame against many keyframes (-1: slot f <-> current frame f)."""
        self.L.sd_track_set_current_broadcast.argtypes = [C.c_void_p, C.c_int]
        _check(self.L.sd_track_set_current_broadcast(self.h, int(cur_frame)))

    def relocalize(self, n_keyframes, cur_frame=0, th=15.0, mono=True, min_matches=20, min_good=10):
        """Tracking::Relocalization (src/Tracking.cc:1064-1097), one batch slot per keyframe attempt."""
        win = C.c_int32(-1)
        st = np.zeros((n_keyframes, 3), np.int32)
        self.L.sd_track_relocalize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _check(self.L.sd_track_relocalize(self.h, n_keyframes, cur_frame, th, int(mono), min_matches, min_good, C.byref(win), _p(st)))
        return int(win.value), st

    def detect_loop(self, n_keyframes, cur_frame=0, excluded=None):
        """Candidate search of LoopClosing::DetectLoop (src/LoopClosing.cc:115-149)."""
        ex = None if excluded is None else np.ascontiguousarray(excluded, np.uint8)
        cand = np.zeros(n_keyframes, np.int32)
        n = C.c_int32(0)
        best = C.c_double(0)
        err = np.zeros(n_keyframes)
        self.L.sd_track_detect_loop.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                C.c_void_p]
        _check(self.L.sd_track_detect_loop(self.h, n_keyframes, cur_frame, None if ex is None else _p(ex), _p(cand), n_keyframes,
                                           C.byref(n), C.byref(best), _p(err)))
        return dict(candidates=cand[:n.value].copy(), best_error=best.value, errors=err)

    def align(self, n_frames, mode=0):
        _check(self.L.sd_track_align(self.h, n_frames, mode))

    def match(self, n_frames, th=8.0, mono=True, check_ori=True):
        _check(self.L.sd_track_match(self.h, n_frames, th, int(mono), int(check_ori)))

    def pnp(self, n_frames, probability=0.99, min_inliers=8, max_iterations=300, min_set=4, epsilon=0.4, th2=5.991,
            n_iterations=None):
        n_iterations = max_iterations if n_iterations is None else n_iterations
        _check(self.L.sd_track_pnp(self.h, n_frames, probability, min_inliers, max_iterations, min_set, epsilon, th2,
                                   n_iterations))

    def pnp_iterate(self, n_frames, n_iterations):
        """A further PnPsolver::iterate(n_iterations) on the solvers the last pnp() call constructed."""
        self.L.sd_track_pnp_iterate.argtypes = [C.c_void_p, C.c_int, C.c_int]
        _check(self.L.sd_track_pnp_iterate(self.h, n_frames, int(n_iterations)))

    def set_matches(self, frame0, cur_match):
        """CurrentFrame.mvpMapPoints of the slots from the caller: [n, cap'] indices into the last-frame arrays, -1 = NULL."""
        m = np.ascontiguousarray(cur_match, np.int32)
        assert m.ndim == 2
        self.L.sd_track_set_matches.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        _check(self.L.sd_track_set_matches(self.h, frame0, m.shape[0], _p(m), m.shape[1]))

    def get_align(self, frame0, n):
        T = np.zeros((n, 16))
        err = np.zeros(n)
        ok = np.zeros(n, np.int32)
        iters = np.zeros((n, 16), np.int32)
        chi2 = np.zeros(n)
        _check(self.L.sd_track_get_align(self.h, frame0, n, _p(T), _p(err), _p(ok), _p(iters), _p(chi2)))
        return dict(T=[_from_cm(t) for t in T], error=err, ok=ok.astype(bool), iters=iters, chi2=chi2)

    def get_matches(self, frame0, n):
        cm = np.zeros((n, self.cap), np.int32)
        nm = np.zeros(n, np.int32)
        _check(self.L.sd_track_get_matches(self.h, frame0, n, _p(cm), self.cap, _p(nm)))
        return cm, nm

    def get_pnp(self, frame0, n):
        T = np.zeros((n, 16), np.float32)
        inl = np.zeros((n, self.cap), np.uint8)
        info = np.zeros((n, 8), np.int32)
        _check(self.L.sd_track_get_pnp(self.h, frame0, n, _p(T), _p(inl), self.cap, _p(info)))
        return dict(T=T.reshape(n, 4, 4), inliers=inl.astype(bool), ok=info[:, 0].astype(bool), n_inliers=info[:, 1],
                    no_more=info[:, 2].astype(bool), iterations=info[:, 3], N=info[:, 4], min_inliers=info[:, 5],
                    max_its=info[:, 6], refined=info[:, 7].astype(bool))

    def set_point_flags(self, frame0, has_mp_cur, has_mp_ref):
        a, b = np.ascontiguousarray(has_mp_cur, np.uint8), np.ascontiguousarray(has_mp_ref, np.uint8)
        assert a.ndim == 2 and a.shape == b.shape
        self.L.sd_track_set_point_flags.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _check(self.L.sd_track_set_point_flags(self.h, frame0, a.shape[0], _p(a), _p(b), a.shape[1]))

    def search_by_points(self, n_frames, nnratio=0.75, check_ori=True):
        """ORBmatcher::SearchByPoints(currentKF, pKF, matches): brute-force Hamming between two keyframes' map points."""
        self.L.sd_track_search_by_points.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int]
        _check(self.L.sd_track_search_by_points(self.h, n_frames, float(nnratio), int(check_ori)))

    def get_point_matches(self, frame0, n):
        m = np.zeros((n, self.cap), np.int32)
        nm = np.zeros(n, np.int32)
        self.L.sd_track_get_point_matches.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        _check(self.L.sd_track_get_point_matches(self.h, frame0, n, _p(m), self.cap, _p(nm)))
        return m, nm

    def features_in_area(self, frame, x, y, r, min_level=-1, max_level=-1, want_grid=False):
        """Frame::GetFeaturesInArea on the device grid of current frame `frame` (debug read-out)."""
        idx = np.zeros(self.cap, np.int32)
        n = C.c_int32(0)
        grid = np.zeros((64, 48), np.int32) if want_grid else None
        self.L.sd_track_debug_features_in_area.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int,
                                                           C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        _check(self.L.sd_track_debug_features_in_area(self.h, frame, x, y, r, min_level, max_level, _p(idx), self.cap, C.byref(n),
                                                      _p(grid) if want_grid else None))
        return (idx[:n.value].copy(), grid) if want_grid else idx[:n.value].copy()

    def pack_records(self, n_frames, source, d_ptr):
        """Queue the packing of the batch's result records (n_frames x 20 f64) into device memory at `d_ptr`."""
        self.L.sd_track_pack_records.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _check(self.L.sd_track_pack_records(self.h, n_frames, int(source), C.c_void_p(d_ptr)))

    def stream_fence(self, stream_ptr, direction):
        """direction 0: the caller's stream waits for the tracking stream; 1: the tracking stream waits for the caller's."""
        self.L.sd_track_stream_fence.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _check(self.L.sd_track_stream_fence(self.h, C.c_void_p(stream_ptr) if stream_ptr else None, int(direction)))

    def set_profiling(self, on=True):
        _check(self.L.sd_track_set_profiling(self.h, int(on)))

    def stage_ms(self):
        ms = np.zeros(3, np.float32)
        _check(self.L.sd_track_stage_ms(self.h, _p(ms), 3))
        return ms


def debug_epnp(Xw, uv, K):
    """Device EPnP on explicit correspondences (diagnostics)."""
    L = lib()
    L.sd_debug_epnp.argtypes = [C.c_int, C.c_void_p, C.c_void_p] + [C.c_double] * 4 + [C.c_void_p] * 3
    Xw = np.ascontiguousarray(Xw, np.float64)
    uv = np.ascontiguousarray(uv, np.float64)
    R, t, e = np.zeros(9), np.zeros(3), C.c_double()
    _check(L.sd_debug_epnp(len(Xw), _p(Xw), _p(uv), float(K[0]), float(K[1]), float(K[2]), float(K[3]), _p(R), _p(t),
                           C.byref(e)))
    return R.reshape(3, 3), t, e.value
