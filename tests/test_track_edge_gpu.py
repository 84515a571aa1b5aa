"""GPU parity, edge inputs of the tracking stages (round-2 advisor findings and VERDICT r2 task 1):
  * a tracker behind an extractor with so few features that whole pyramid levels have no grid cell (no FAST launch on
    those levels: ImageAlign's "pyramid ready" event must still cover them);
  * a 4-level pyramid: ImageAlign needs max_level_ = 4 (reference src/ImageAlign.cc:36-39,57 "Not enough pyramid levels");
  * PnP solvers whose inputs were replaced since sd_track_pnp: sd_track_pnp_iterate must refuse, not continue on stale state;
  * alignment on nearly textureless images (H close to singular: the LDLT-driven accept / stop branches)."""
import numpy as np
import pytest

from sdslam_amd import synth

pytestmark = pytest.mark.gpu
K = (synth.FX, synth.FY, synth.CX, synth.CY)
BOUNDS = (0.0, 640.0, 0.0, 480.0)
POSE_TOL = 1e-5


@pytest.fixture(scope="module")
def sd():
    import sdslam_amd
    if sdslam_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests need a real MI355X")
    return sdslam_amd


def _levels(o, n):
    return [o.level(l) for l in range(n)]


@pytest.mark.parametrize("nfeatures", [40, 12])
def test_tracker_behind_an_extractor_with_few_features(sd, oracle, nfeatures):
    """nfeatures 40 / 12 at 8 x 1.2: level quotas of 9...2 features, i.e. levels whose grid has no cell at all (quota < 7 gives
    levelCols = 0, src/ORBextractor.cc:474) and -- for 12 -- no cell on ANY level of the merged FAST launch (3...7), which is
    then skipped.  ImageAlign reads levels 4, 3, 2 of exactly those pyramids, back to back with the extraction (no host
    synchronisation in between), on 16 frames per launch and five times over: poses, iteration counts and errors equal the
    oracle's every time (a pyramid level read before its resize had finished would show up as a different chi2)."""
    cfg = (nfeatures, 1.2, 8, 20)
    B = 16
    scenes = [synth.make_scene(40 + (i % 4), *[((0.02, -0.01, 0.015), (0.4, -0.3, 0.5)), ((-0.03, 0.02, -0.01), (-0.6, 0.2, 0.3))][i % 2])
              for i in range(4)]
    info = sd.plan_info(*cfg, 640, 480)
    cells_per_level = [int(((info["cells"][:, 0] == l) & (info["cells"][:, 3] > 0)).sum()) for l in range(8)]
    assert 0 in cells_per_level, cells_per_level           # the situation under test really occurs
    if nfeatures == 12:
        assert sum(cells_per_level[3:]) == 0               # no merged FAST launch at all
    cur, ref = sd.ORBextractor(*cfg, 640, 480, B), sd.ORBextractor(*cfg, 640, 480, B)
    from sdslam_amd.capi import DeviceBuffer
    fr = np.stack([scenes[i % 4]["cur"] for i in range(B)])
    d = DeviceBuffer(fr.nbytes)
    d.upload(fr)
    ref.extract_batch(np.stack([scenes[i % 4]["ref"] for i in range(B)]))
    trk = sd.Tracker(cur, ref, max_points=400, max_batch=B)
    trk.set_camera(*K, 0.0, BOUNDS)
    # map points: a fixed grid of ref-image pixels back-projected on the scene surface (the few keypoints are not enough)
    gx, gy = np.meshgrid(np.linspace(60, 580, 20), np.linspace(50, 430, 15))
    px = np.stack([gx.ravel(), gy.ravel()], 1)
    Xw = synth.backproject_on_surface(px)
    n = len(Xw)
    last = dict(valid=np.ones(n, np.uint8), Xw=Xw, desc=np.zeros((n, 32), np.uint8), octave=np.zeros(n, np.int32),
                angle=np.zeros(n, np.float32), obs=np.ones(n, np.int32))
    trk.set_last(0, [last] * B)
    T0 = [synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ scenes[i % 4]["T_cur"] for i in range(B)]
    want = []
    for i in range(4):
        oc, orf = oracle.OrbOracle(*cfg), oracle.OrbOracle(*cfg)
        oc.extract(scenes[i]["cur"])
        orf.extract(scenes[i]["ref"])
        tab = oc.tables()
        want.append(oracle.align(_levels(oc, 8), _levels(orf, 8), tab["inv_sf"], tab["sf"], Xw[:300], scenes[i]["T_ref"], T0[i], K, mode=0))
    assert all(w["ok"] for w in want)
    for rep in range(5):
        trk.set_poses(0, [scenes[i % 4]["T_ref"] for i in range(B)], T0)
        cur.extract_batch_device(d.ptr, B, 640, 480)
        trk.align(B, 0)
        g = trk.get_align(0, B)
        for i in range(B):
            w = want[i % 4]
            assert g["ok"][i] and np.array_equal(g["iters"][i][:8], w["iters"]), (rep, i, g["iters"][i][:8], w["iters"])
            assert np.abs(g["T"][i] - w["T"]).max() <= POSE_TOL and abs(g["chi2"][i] - w["chi2"]) <= 1e-9 * max(1.0, abs(w["chi2"])), (rep, i)
    trk.close()
    cur.close()
    ref.close()


def test_image_align_needs_five_pyramid_levels(sd, oracle):
    """A 4-level extractor through sd_track_align: the reference logs "Not enough pyramid levels" and returns false
    (src/ImageAlign.cc:57-60) in every mode; the pose is untouched and TrackWithMotionModel falls back to the prediction
    (src/Tracking.cc:669-672) -- equal to the oracle's composition."""
    cfg = (1000, 2.0, 4, 20)
    s = synth.make_scene(20)
    cur, ref = sd.ORBextractor(*cfg, 640, 480, 1), sd.ORBextractor(*cfg, 640, 480, 1)
    ck, cd, cn = cur.extract_batch(s["cur"][None])
    rk, rd, rn = ref.extract_batch(s["ref"][None])
    oc, orf = oracle.OrbOracle(*cfg), oracle.OrbOracle(*cfg)
    ock, ocd = oc.extract(s["cur"])
    ork, ord_ = orf.extract(s["ref"])
    assert np.array_equal(ock, ck[0, :cn[0]]) and np.array_equal(ork, rk[0, :rn[0]])
    last = synth.tracking_case(20, ork, ord_)
    trk = sd.Tracker(cur, ref, 1000, 1)
    trk.set_camera(*K, 0.0, BOUNDS)
    trk.set_last(0, [last])
    T0 = synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ s["T_cur"]
    tab = oc.tables()
    for mode in (0, 1, 2, 3):
        trk.set_poses(0, [s["T_ref"]], [T0])
        trk.align(1, mode)
        g = trk.get_align(0, 1)
        r = oracle.align(_levels(oc, 4), _levels(orf, 4), tab["inv_sf"], tab["sf"], last["Xw"][last["valid"] != 0], s["T_ref"], T0, K, mode=mode)
        assert not r["ok"] and not g["ok"][0], mode
        assert np.abs(g["T"][0] - T0).max() == 0 and not g["iters"][0].any()
    trk.set_poses(0, [s["T_ref"]], [T0])
    trk.track_with_motion_model(1, th=15.0, mono=True, align_mode=0)
    tw, po = trk.get_tracked(0, 1), trk.get_pose_opt(0, 1)
    cm, _ = trk.get_matches(0, 1)
    r = oracle.track_with_motion_model(_levels(oc, 4), _levels(orf, 4), tab, ock, ocd, BOUNDS, K, s["T_ref"], T0, last, 15.0, mono=True)
    assert (tw["status"][0], tw["nmatches"][0], tw["nmatches_map"][0], tw["retried"][0]) == (r["status"], r["nmatches"], r["nmatches_map"], r["retried"])
    assert np.array_equal(cm[0, :len(ock)], r["match"]) and np.abs(po["T"][0] - r["T"]).max() <= POSE_TOL
    assert r["status"] == 2            # the prediction alone is good enough here: tracked without alignment
    trk.close()
    cur.close()
    ref.close()


def test_pnp_iterate_refuses_solvers_whose_inputs_were_replaced(sd, oracle):
    """The reference's PnPsolver owns copies of its inputs (src/PnPsolver.cc:71-110); the tracker's solvers read the live
    match vector / keypoints.  Whatever replaces those between sd_track_pnp and sd_track_pnp_iterate must end the solvers:
    a loud SD_ERR_INVALID_ARG instead of inlier masks that index another correspondence list."""
    cfg = (1000, 1.2, 8, 20)
    s = synth.make_scene(20)
    cur, ref = sd.ORBextractor(*cfg, 640, 480, 1), sd.ORBextractor(*cfg, 640, 480, 1)
    cur.extract_batch(s["cur"][None])
    rk, rd, rn = ref.extract_batch(s["ref"][None])
    last = synth.tracking_case(20, rk[0, :rn[0]], rd[0, :rn[0]])
    trk = sd.Tracker(cur, ref, 1000, 1, 300)
    trk.set_camera(*K, 0.0, BOUNDS)
    trk.set_last(0, [last])
    trk.set_poses(0, [s["T_ref"]], [s["T_cur"]])
    rs = synth.glibc_rand_stream(1200)

    def build():
        trk.set_rand(0, rs[None])
        trk.match(1, 8.0, True, True)
        trk.pnp(1, 0.99, 10, 200, 4, 0.28, 5.991, 5)

    build()
    trk.pnp_iterate(1, 5)                                    # the legitimate continuation works
    first = trk.get_pnp(0, 1)
    assert first["ok"][0]
    cm, _ = trk.get_matches(0, 1)
    for what, action in [("match", lambda: trk.match(1, 8.0, True, True)),
                         ("set_matches", lambda: trk.set_matches(0, cm[:1])),
                         ("set_last", lambda: trk.set_last(0, [last])),
                         ("set_rand", lambda: trk.set_rand(0, rs[None])),
                         ("track_with_motion_model", lambda: trk.track_with_motion_model(1, th=8.0)),
                         ("re-extraction", lambda: cur.extract_batch(s["cur"][None]))]:
        build()
        action()
        with pytest.raises(sd.SdError):
            trk.pnp_iterate(1, 5)
        build()                                              # and a fresh sd_track_pnp makes it usable again
        trk.pnp_iterate(1, 5)
        again = trk.get_pnp(0, 1)
        assert again["iterations"][0] == first["iterations"][0] and np.array_equal(again["inliers"], first["inliers"]), what
    for bad in (1, 2):
        with pytest.raises(sd.SdError):
            trk.pnp(1, 0.99, 10, 50, bad, 0.4, 5.991, 50)    # minSet below 3: pinned by nothing, refused
    trk.close()
    cur.close()
    ref.close()


def _smooth_image(seed, kind):
    """Nearly textureless frames: `ramp` varies along x only (every patch gradient is (g, 0): H has rank <= 3 -- the 2 x 6
    projection Jacobian's first row -- whatever the points), `blobs` is a sum of a few very wide Gaussians (rank 6 but tiny,
    badly scaled curvature), both with +-1 noise so that residuals are not identically zero."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:480, 0:640].astype(np.float64)
    if kind == "ramp":
        img = 40.0 + xx * 0.25
    else:
        img = np.full((480, 640), 90.0)
        for _ in range(5):
            cx, cy, sg, a = rng.uniform(0, 640), rng.uniform(0, 480), rng.uniform(150, 300), rng.uniform(20, 60)
            img += a * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
    img += rng.integers(-1, 2, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("kind", ["ramp", "blobs"])
def test_image_align_on_nearly_textureless_frames(sd, oracle, kind):
    """k_align reduces H per point and per level in a tree, the reference adds 4800 J J^T terms one by one; on
    well-textured scenes that is 1e-12 in the pose.  Here H is (nearly) singular and Eigen's pivoted LDLT works on rounding
    noise.  What is pinned: the DECISIONS -- return value, per-level iteration counts -- for 24 / 60 / 300 points in modes 0
    and 2, because they hang on the float residual chain (reproduced operation by operation), and the pose wherever the
    step itself is well defined.  `ramp` (rank-deficient by construction): the LDLT solution of a singular system is
    rounding noise in BOTH implementations; such a slot is reported as not comparable (parity unpinned for it) unless the
    decisions happen to agree, and the test demands agreement only for `blobs`, where H is regular."""
    cfg = (1000, 1.2, 8, 20)
    img_ref, img_cur = _smooth_image(5, kind), _smooth_image(6, kind)
    counts = [24, 60, 300]
    B = len(counts)
    cur, ref = sd.ORBextractor(*cfg, 640, 480, B), sd.ORBextractor(*cfg, 640, 480, B)
    cur.extract_batch(np.stack([img_cur] * B))
    ref.extract_batch(np.stack([img_ref] * B))
    oc, orf = oracle.OrbOracle(*cfg), oracle.OrbOracle(*cfg)
    oc.extract(img_cur)
    orf.extract(img_ref)
    tab = oc.tables()
    rng = np.random.default_rng(8)
    px = np.stack([rng.uniform(40, 600, 300), rng.uniform(40, 440, 300)], 1)
    Xw = synth.backproject_on_surface(px)
    trk = sd.Tracker(cur, ref, 400, B)
    trk.set_camera(*K, 0.0, BOUNDS)
    lasts = []
    for c in counts:
        lasts.append(dict(valid=(np.arange(300) < c).astype(np.uint8), Xw=Xw, desc=np.zeros((300, 32), np.uint8), octave=np.zeros(300, np.int32),
                          angle=np.zeros(300, np.float32), obs=np.ones(300, np.int32)))
    trk.set_last(0, lasts)
    T0 = synth.se3_exp((0.002, -0.001, 0.001), (0.03, 0.02, -0.02))
    agree, total = 0, 0
    for mode in (0, 2):
        trk.set_poses(0, [np.eye(4)] * B, [T0] * B)
        trk.align(B, mode)
        g = trk.get_align(0, B)
        for i, c in enumerate(counts):
            r = oracle.align(_levels(oc, 8), _levels(orf, 8), tab["inv_sf"], tab["sf"], Xw[:c], np.eye(4), T0, K, mode=mode)
            same = g["ok"][i] == r["ok"] and np.array_equal(g["iters"][i][:8], r["iters"])
            total += 1
            agree += bool(same)
            if kind == "blobs":
                assert same, (mode, c, g["ok"][i], r["ok"], g["iters"][i][:8], r["iters"])
            if same and r["ok"]:
                # same decisions => same sequence of accepted steps; the steps differ by H's conditioning times 1e-16
                tol = POSE_TOL if kind == "blobs" else 1e-3
                assert np.abs(g["T"][i] - r["T"]).max() <= tol, (kind, mode, c, np.abs(g["T"][i] - r["T"]).max())
            assert np.isfinite(g["T"][i]).all()
    print(f"{kind}: decisions agree in {agree} of {total} slots")
    trk.close()
    cur.close()
    ref.close()


def test_two_host_threads_on_two_handle_sets(sd, oracle):
    """SURVEY section 8(b) "Threading": Tracking runs on the caller's thread while LoopClosing::DetectLoop runs
    ImageAlign(KF, KF) on its own (reference src/LoopClosing.cc:133); include/sdslam_hip.h promises that different handles
    are independent.  One process, two host threads (ctypes releases the GIL for the duration of a call), each with its own
    extractor pair + tracker: thread A loops extraction + TrackWithMotionModel on four frame pairs, thread B loops the
    DetectLoop candidate search of one keyframe against 64 keyframes.  Every result of every iteration equals what the same
    handles gave single-threaded (and that equals the oracle: a spot check on thread A's first frame)."""
    import threading
    from sdslam_amd.capi import DeviceBuffer
    cfg = (1000, 1.2, 8, 20)
    # ---- thread A's world
    scenes = [synth.make_scene(20 + i) for i in range(4)]
    curA, refA = sd.ORBextractor(*cfg, 640, 480, 4), sd.ORBextractor(*cfg, 640, 480, 4)
    rk, rd, rn = refA.extract_batch(np.stack([s["ref"] for s in scenes]))
    trkA = sd.Tracker(curA, refA, 1000, 4)
    trkA.set_camera(*K, 0.0, BOUNDS)
    lasts = [synth.tracking_case(i, rk[i, :rn[i]], rd[i, :rn[i]]) for i in range(4)]
    trkA.set_last(0, lasts)
    T0 = [synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ s["T_cur"] for s in scenes]
    trkA.set_poses(0, [s["T_ref"] for s in scenes], T0)
    fr = np.stack([s["cur"] for s in scenes])
    dA = DeviceBuffer(fr.nbytes)
    dA.upload(fr)

    def step_a():
        curA.extract_batch_device(dA.ptr, 4, 640, 480)
        trkA.track_with_motion_model(4, th=8.0, mono=True, align_mode=0)
        tw, po = trkA.get_tracked(0, 4), trkA.get_pose_opt(0, 4)
        cm, _ = trkA.get_matches(0, 4)
        return np.stack([tw["status"], tw["nmatches"], tw["nmatches_map"]]), po["T"].copy(), cm.copy()

    # ---- thread B's world: one current keyframe against 64 keyframes (8 distinct views, tiled)
    tex = synth.make_image(71, 1280, 960)
    T_kf = [synth.se3_exp((0.01 * (i % 3 - 1), 0.008 * (i % 4 - 2), 0.005 * i), (0.2 * (i % 5 - 2), -0.1 * (i % 3), 0.15 * i)) for i in range(8)]
    im_kf = np.stack([synth.render_plane_view(tex, T) for T in T_kf])
    T_c = synth.se3_exp((0.015, -0.01, 0.01), (0.3, -0.2, 0.4))
    curB, refB = sd.ORBextractor(*cfg, 640, 480, 1), sd.ORBextractor(*cfg, 640, 480, 64)
    curB.extract_batch(synth.render_plane_view(tex, T_c)[None])
    kk, kd, kn = refB.extract_batch(im_kf[np.arange(64) % 8])
    trkB = sd.Tracker(curB, refB, 1000, 64)
    trkB.set_camera(*K, 0.0, BOUNDS)
    trkB.set_last(0, [synth.keyframe_case(kk[i, :kn[i]], kd[i, :kn[i]], T_kf[i % 8], max_points=400) for i in range(64)])
    trkB.set_poses(0, [T_kf[i % 8] for i in range(64)], [np.eye(4)] * 64)
    excluded = [1 if i % 11 == 0 else 0 for i in range(64)]

    def step_b():
        g = trkB.detect_loop(64, cur_frame=0, excluded=excluded)
        return np.array(g["candidates"]), np.array(g["errors"]), np.float64(g["best_error"])

    a0, b0 = step_a(), step_b()
    assert (a0[0][0] == 2).all() and len(b0[0]) >= 1
    # spot check against the oracle (thread A, frame 0)
    oc, orf = oracle.OrbOracle(*cfg), oracle.OrbOracle(*cfg)
    ock, ocd = oc.extract(scenes[0]["cur"])
    orf.extract(scenes[0]["ref"])
    r = oracle.track_with_motion_model(_levels(oc, 8), _levels(orf, 8), oc.tables(), ock, ocd, BOUNDS, K, scenes[0]["T_ref"], T0[0], lasts[0], 8.0,
                                       mono=True)
    assert a0[0][1][0] == r["nmatches"] and np.array_equal(a0[2][0, :len(ock)], r["match"]) and np.abs(a0[1][0] - r["T"]).max() <= POSE_TOL
    errors = []

    def run(step, want, n):
        try:
            for k in range(n):
                got = step()
                for x, y in zip(got, want):
                    if not np.array_equal(x, y):
                        raise AssertionError(f"iteration {k}: result differs from the single-threaded one")
        except Exception as e:   # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    ta = threading.Thread(target=run, args=(step_a, a0, 25))
    tb = threading.Thread(target=run, args=(step_b, b0, 25))
    ta.start()
    tb.start()
    ta.join(300)
    tb.join(300)
    assert not ta.is_alive() and not tb.is_alive()
    assert not errors, errors
    for h in (trkA, trkB, curA, refA, curB, refB):
        h.close()
