"""Seeded synthetic inputs.  No datasets exist offline, so
every test/bench input is regenerated from an integer seed with numpy PCG64."""
import numpy as np


def make_image(seed: int, w: int = 640, h: int = 480) -> np.ndarray:
    """C1 generator: mid-gray + 400 random rectangles + 200 filled discs + +-4 noise
    (counts scale with image area so 1280x720 frames keep the same corner density)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    area = (w * h) / (640.0 * 480.0)
    img = np.full((h, w), 128, dtype=np.int32)
    n_rect, n_disc = int(round(400 * area)), int(round(200 * area))
    for _ in range(n_rect):
        rw, rh = rng.integers(8, 65, size=2)
        x0 = int(rng.integers(-16, w))
        y0 = int(rng.integers(-16, h))
        g = int(rng.integers(0, 256))
        img[max(y0, 0):max(y0 + int(rh), 0), max(x0, 0):max(x0 + int(rw), 0)] = g
    for _ in range(n_disc):
        r = int(rng.integers(3, 13))
        cx = int(rng.integers(0, w))
        cy = int(rng.integers(0, h))
        g = int(rng.integers(0, 256))
        y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, h), max(cx - r, 0), min(cx + r + 1, w)
        yy, xx = np.mgrid[y0:y1, x0:x1]
        m = (yy - cy) ** 2 + (xx - cx) ** 2 <= r * r
        img[y0:y1, x0:x1][m] = g
    img += rng.integers(-4, 5, size=(h, w))
    return np.clip(img, 0, 255).astype(np.uint8)


def make_batch(seed0: int, n: int, w: int = 640, h: int = 480) -> np.ndarray:
    return np.stack([make_image(seed0 + i, w, h) for i in range(n)])


# ----------------------------------------------------------------------------------------
# C3/C4 scene: textured plane Z = 2 m seen from two nearby poses
# ----------------------------------------------------------------------------------------
FX, FY, CX, CY = 500.0, 500.0, 320.0, 240.0      # Examples/Example.yaml-style pinhole, no distortion


def se3_exp(upsilon, omega_deg):
    """Rigid transform from translation-first twist (metres, degrees); plain Rodrigues."""
    w = np.deg2rad(np.asarray(omega_deg, np.float64))
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        R, V = np.eye(3) + K, np.eye(3)
    else:
        R = np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ np.asarray(upsilon, np.float64)
    return T


def _bilinear(tex, x, y):
    h, w = tex.shape
    x = np.clip(x, 0, w - 1.001)
    y = np.clip(y, 0, h - 1.001)
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    fx, fy = x - x0, y - y0
    t = tex.astype(np.float64)
    v = (t[y0, x0] * (1 - fx) * (1 - fy) + t[y0, x0 + 1] * fx * (1 - fy) + t[y0 + 1, x0] * (1 - fx) * fy +
         t[y0 + 1, x0 + 1] * fx * fy)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def surface_z(X, Y, depth=2.0):
    """Gently curved scene surface Z = f(X, Y) (non-planar on purpose: EPnP as restated in the
    reference has no planar special case)."""
    return depth + 0.3 * np.sin(1.3 * X + 0.4) * np.cos(1.1 * Y - 0.2) + 0.1 * X


def intersect_surface(Ow, dw, depth=2.0, iters=24):
    """Ray/surface intersection by fixed-point iteration on the ray parameter (|slope| < 1)."""
    s = (depth - Ow[2]) / dw[..., 2]
    for _ in range(iters):
        X = Ow[0] + dw[..., 0] * s
        Y = Ow[1] + dw[..., 1] * s
        s = (surface_z(X, Y, depth) - Ow[2]) / dw[..., 2]
    return Ow + dw * s[..., None]


def render_plane_view(tex, Tcw, w=640, h=480, depth=2.0):
    """Image of the textured scene surface; the texture (2x the image resolution) is painted on
    the surface along Z, i.e. texel = 2 * projection of (X, Y) onto the canonical Z=depth view."""
    R, t = Tcw[:3, :3], Tcw[:3, 3]
    v, u = np.mgrid[0:h, 0:w].astype(np.float64)
    rays = np.stack([(u - CX) / FX, (v - CY) / FY, np.ones_like(u)], -1)        # camera frame
    Rwc = R.T
    Ow = -Rwc @ t
    dw = rays @ Rwc.T                                                          # world directions
    Xw = intersect_surface(Ow, dw, depth)
    ur = FX * Xw[..., 0] / depth + CX
    vr = FY * Xw[..., 1] / depth + CY
    return _bilinear(tex, 2.0 * ur + 0.5, 2.0 * vr + 0.5)


def make_scene(seed: int, upsilon=(0.02, -0.01, 0.015), omega_deg=(0.4, -0.3, 0.5), w=640, h=480):
    """Two views of the textured surface: ref pose = identity, cur pose = Exp(upsilon, omega)."""
    tex = make_image(seed, 2 * w, 2 * h)
    T_ref = np.eye(4)
    T_cur = se3_exp(upsilon, omega_deg)
    return dict(ref=render_plane_view(tex, T_ref, w, h), cur=render_plane_view(tex, T_cur, w, h),
                T_ref=T_ref, T_cur=T_cur, K=(FX, FY, CX, CY), depth=2.0)


def backproject_on_surface(xy, depth=2.0):
    """World points (ref pose = identity) hit by the rays through ref-image pixels."""
    xy = np.asarray(xy, np.float64)
    rays = np.stack([(xy[:, 0] - CX) / FX, (xy[:, 1] - CY) / FY, np.ones(len(xy))], -1)
    return intersect_surface(np.zeros(3), rays, depth)


def tracking_case(seed: int, ref_kps, ref_desc, max_points=300, **kw):
    """Last-frame data of a TrackWithMotionModel step: the first `max_points` ref keypoints carry
    map points (the surface points they see), described by the ref descriptors."""
    n = len(ref_kps)
    valid = np.zeros(n, np.uint8)
    valid[:min(n, max_points)] = 1
    Xw = backproject_on_surface(np.stack([ref_kps["x"], ref_kps["y"]], 1))
    return dict(valid=valid, Xw=Xw, desc=ref_desc.copy(), octave=ref_kps["octave"].astype(np.int32).copy(),
                angle=ref_kps["angle"].astype(np.float32).copy(), obs=np.ones(n, np.int32))


def keyframe_case(kps, desc, T_kf, max_points=300):
    """Map points of a keyframe seen at pose T_kf (world = the scene's frame): the first `max_points` keypoints carry
    the surface points they see.  Same dict as tracking_case (what Tracker.set_last takes)."""
    n = len(kps)
    R, t = T_kf[:3, :3], T_kf[:3, 3]
    Ow = -R.T @ t
    rays = np.stack([(kps["x"].astype(np.float64) - CX) / FX, (kps["y"].astype(np.float64) - CY) / FY, np.ones(n)], -1)
    Xw = intersect_surface(Ow, rays @ R, 2.0) if n else np.zeros((0, 3))
    valid = np.zeros(n, np.uint8)
    valid[:min(n, max_points)] = 1
    return dict(valid=valid, Xw=np.ascontiguousarray(Xw), desc=desc.copy(), octave=kps["octave"].astype(np.int32).copy(),
                angle=kps["angle"].astype(np.float32).copy(), obs=np.ones(n, np.int32))


def planted_matches(seed, kps, T_cw, n_match, outlier_frac, noise_px=0.0, n_exact_inliers=None):
    """A match vector the tracker did NOT produce (PnPsolver / PoseOptimization take any vpMapPointMatches): keypoint i of the
    current frame is paired with map point i, whose world position is the back-projection of the keypoint at a random depth
    under the true pose T_cw (+ pixel noise) -- or a gross outlier.
    Returns (last, cm, truth): `last` = dict for Tracker.set_last / the oracle (one map point per keypoint slot), cm[i] = i for
    the n_match chosen keypoints else -1, truth[i] = True where the pair is a planted inlier."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n = len(kps)
    chosen = np.sort(rng.choice(n, size=n_match, replace=False))
    n_in = int(round(n_match * (1.0 - outlier_frac))) if n_exact_inliers is None else n_exact_inliers
    inl = np.zeros(n, bool)
    inl[rng.choice(chosen, size=n_in, replace=False)] = True
    R, t = T_cw[:3, :3], T_cw[:3, 3]
    xy = np.stack([kps["x"], kps["y"]], 1).astype(np.float64)
    xy_n = xy + rng.normal(size=xy.shape) * noise_px
    z = rng.uniform(1.0, 5.0, n)
    Xc = np.stack([(xy_n[:, 0] - CX) / FX * z, (xy_n[:, 1] - CY) / FY * z, z], 1)
    Xw = (Xc - t) @ R                       # R^T (Xc - t)
    bad = ~inl
    Xw[bad] = rng.uniform(-2.0, 2.0, size=(int(bad.sum()), 3)) + np.array([0, 0, 3.0])
    cm = np.full(n, -1, np.int32)
    cm[chosen] = chosen
    last = dict(valid=np.ones(n, np.uint8), Xw=np.ascontiguousarray(Xw), desc=np.zeros((n, 32), np.uint8),
                octave=kps["octave"].astype(np.int32).copy(), angle=kps["angle"].astype(np.float32).copy(), obs=np.ones(n, np.int32))
    return last, cm, inl & (cm >= 0)


def glibc_rand_stream(n, seed=1):
    """n raw rand() values of glibc's TYPE_3 additive-feedback generator seeded with `seed`.
    The reference never seeds: SD_SLAM::Random draws from the default seed-1 state (reference
    src/extra/utils.cc:23-26).  Restated so harnesses do not perturb the process RNG."""
    r = [0] * 34
    r[0] = seed
    for i in range(1, 31):
        hi, lo = divmod(r[i - 1], 127773)
        w = 16807 * lo - 2836 * hi
        if w < 0:
            w += 2147483647
        r[i] = w
    for i in range(31, 34):
        r[i] = r[i - 31]
    out = []
    for i in range(34, 344 + n):
        v = (r[i - 31] + r[i - 3]) & 0xFFFFFFFF
        r.append(v)
        if i >= 344:
            out.append(v >> 1)
    return np.array(out, np.int32)


def local_map_case(seed, kps, desc, T_cur, n_extra=300, scale_factor=1.2, nlevels=8):
    """Synthetic TrackLocalMap input for a current frame with keypoints `kps` / descriptors `desc` seen at pose
    T_cur: one local map point per keypoint (the scene-surface point it sees, descriptor = the keypoint's with a few
    flipped bits, normal roughly along the viewing ray, distance-invariance interval around the true distance and the
    keypoint's octave), plus `n_extra` points that fail one of isInFrustum's tests or have no good match (behind the
    camera, outside the image, too near / far, grazing view, random descriptor).  Returns the dict the oracle and
    Tracker.set_local take, in a shuffled but deterministic order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    R, t = T_cur[:3, :3], T_cur[:3, 3]
    Ow = -R.T @ t
    n = len(kps)
    rays = np.stack([(kps["x"].astype(np.float64) - CX) / FX, (kps["y"].astype(np.float64) - CY) / FY, np.ones(n)], -1)
    Xw = intersect_surface(Ow, rays @ R, 2.0)                                     # world directions = R^T ray
    PO = Xw - Ow
    d = np.linalg.norm(PO, axis=1)
    sf = scale_factor ** kps["octave"].astype(np.float64)
    mf_max = (d * sf * rng.uniform(0.85, 1.15, n)).astype(np.float32)          # mfMaxDistance = dist * scale at creation
    mf_min = (mf_max / np.float32(scale_factor ** (nlevels - 1))).astype(np.float32)
    nrm = PO / d[:, None] + rng.normal(size=(n, 3)) * 0.25                        # viewCos = PO.Pn / dist, mostly > 0.9
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    dsc = desc.copy()
    flips = rng.integers(0, 256, size=(n, 6))
    for k in range(6):
        dsc[np.arange(n), flips[:, k] // 8] ^= (1 << (flips[:, k] % 8)).astype(np.uint8)
    pts = dict(cand=np.ones(n, np.uint8), Xw=Xw, normal=nrm, min_dist=np.float32(0.8) * mf_min, max_dist=np.float32(1.2) * mf_max,
               mf_max_dist=mf_max, desc=dsc, obs=rng.integers(0, 4, n).astype(np.int32))
    ex = n_extra
    eX = Ow + (rng.normal(size=(ex, 3)) * np.array([1.5, 1.2, 2.5]) + np.array([0, 0, 1.0])) @ R
    ed = np.linalg.norm(eX - Ow, axis=1)
    emf = (ed * rng.uniform(0.3, 3.0, ex)).astype(np.float32)
    en = rng.normal(size=(ex, 3))
    en /= np.linalg.norm(en, axis=1)[:, None]
    extra = dict(cand=(rng.random(ex) > 0.1).astype(np.uint8), Xw=eX, normal=en, min_dist=np.float32(0.8) * emf / np.float32(3.58),
                 max_dist=np.float32(1.2) * emf, mf_max_dist=emf, desc=rng.integers(0, 256, size=(ex, 32)).astype(np.uint8),
                 obs=rng.integers(0, 3, ex).astype(np.int32))
    out = {k: np.concatenate([pts[k], extra[k]]) for k in pts}
    perm = rng.permutation(n + ex)
    return {k: np.ascontiguousarray(v[perm]) for k, v in out.items()}
