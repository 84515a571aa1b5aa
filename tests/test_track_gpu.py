"""GPU parity: sd_track_* (ImageAlign, SearchByProjection, PnP RANSAC) vs the CPU oracle on
the same seeded two-view scenes.  Bars: matches bit-exact (integer work); poses within 1e-5
absolute on every entry (BASELINE north_star); iteration counts / inlier masks reported equal."""
import numpy as np
import pytest

from sdslam_amd import synth

pytestmark = pytest.mark.gpu
K = (synth.FX, synth.FY, synth.CX, synth.CY)
CFG = (1000, 1.2, 8, 20)
# both pyramids of SURVEY D3: "p8" = BASELINE's 8 levels x 1.2, "p5" = the reference's own default, 5 levels x 2.0
# (src/Config.cc:48-51): ImageAlign then works on 40x30 / 80x60 / 160x120 images (src/ImageAlign.cc:36-39,57,382) and the
# search radius / octave window on scale factors 2^n (src/ORBmatcher.cc:999-1004)
CFGS = {"p8": CFG, "p5": (1000, 2.0, 5, 20)}
BOUNDS = (0.0, 640.0, 0.0, 480.0)
POSE_TOL = 1e-5


@pytest.fixture(scope="module")
def sd():
    import sdslam_amd
    if sdslam_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests need a real MI355X")
    return sdslam_amd


@pytest.fixture(scope="module", params=["p8", "p5"])
def rig(sd, oracle, request):
    """4 scenes with different motions; extractors + oracle state for each; once per pyramid."""
    CFG = CFGS[request.param]
    NL = CFG[2]
    motions = [((0.02, -0.01, 0.015), (0.4, -0.3, 0.5)), ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),
               ((-0.03, 0.02, -0.01), (-0.6, 0.2, 0.3)), ((0.01, 0.03, 0.02), (0.2, 0.5, -0.8))]
    B = len(motions)
    scenes = [synth.make_scene(20 + i, *m) for i, m in enumerate(motions)]
    cur = sd.ORBextractor(*CFG, 640, 480, B)
    ref = sd.ORBextractor(*CFG, 640, 480, B)
    ck, cd, cn = cur.extract_batch(np.stack([s["cur"] for s in scenes]))
    rk, rd, rn = ref.extract_batch(np.stack([s["ref"] for s in scenes]))
    oras = []
    for i, s in enumerate(scenes):
        oc, orf = oracle.OrbOracle(*CFG), oracle.OrbOracle(*CFG)
        ock, ocd = oc.extract(s["cur"])
        ork, ord_ = orf.extract(s["ref"])
        assert np.array_equal(ock, ck[i, :cn[i]]) and np.array_equal(ork, rk[i, :rn[i]])
        oras.append(dict(oc=oc, orf=orf, ck=ock, cd=ocd, rk=ork, rd=ord_, last=synth.tracking_case(i, ork, ord_),
                         tab=oc.tables()))
    trk = sd.Tracker(cur, ref, max_points=1000, max_batch=B, pnp_max_iterations=300)
    trk.set_camera(*K, 0.0, BOUNDS)
    trk.set_last(0, [o["last"] for o in oras])
    yield dict(B=B, scenes=scenes, cur=cur, ref=ref, trk=trk, oras=oras, cfg=CFG, NL=NL, name=request.param)
    trk.close()
    cur.close()
    ref.close()


@pytest.fixture(params=[1, 0], ids=["split", "single"])
def match_mode(sd, request):
    """Both forms of SearchByProjection(Frame, Frame): candidate kernel + one-wave assignment kernel (the default; a frame whose
    candidates do not fit its HBM list is handed to the single kernel) and the single 39-KB kernel alone."""
    with sd.options({"track.match_split": request.param}):
        yield request.param


def _oracle_align(oracle, o, s, T0, mode=0):
    pc = [o["oc"].level(l) for l in range(len(o["tab"]["sf"]))]
    pr = [o["orf"].level(l) for l in range(len(o["tab"]["sf"]))]
    Xw = o["last"]["Xw"][o["last"]["valid"] != 0]
    return oracle.align(pc, pr, o["tab"]["inv_sf"], o["tab"]["sf"], Xw, s["T_ref"], T0, K, mode=mode)


@pytest.mark.parametrize("init", ["identity", "perturbed_truth"])
def test_image_align_matches_oracle(sd, oracle, rig, init):
    trk, B = rig["trk"], rig["B"]
    if init == "identity":
        T0 = [np.eye(4) for _ in range(B)]
    else:
        T0 = [synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ s["T_cur"] for s in rig["scenes"]]
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T0)
    trk.align(B, mode=0)
    g = trk.get_align(0, B)
    for i in range(B):
        r = _oracle_align(oracle, rig["oras"][i], rig["scenes"][i], T0[i])
        assert g["ok"][i] == r["ok"]
        assert np.abs(g["T"][i] - r["T"]).max() <= POSE_TOL, (i, np.abs(g["T"][i] - r["T"]).max())
        assert np.array_equal(g["iters"][i][:len(r["iters"])], r["iters"]), (i, g["iters"][i][:len(r["iters"])], r["iters"])
        assert abs(g["error"][i] - r["error"]) <= 1e-7 * max(1.0, abs(r["error"]))
        assert abs(g["chi2"][i] - r["chi2"]) <= 1e-9 * max(1.0, abs(r["chi2"]))


def test_image_align_modes(sd, oracle, rig):
    trk, B = rig["trk"], rig["B"]
    T0 = [np.eye(4) for _ in range(B)]
    for mode in (2, 3):
        trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T0)
        trk.align(B, mode=mode)
        g = trk.get_align(0, B)
        for i in range(B):
            o = rig["oras"][i]
            pc = [o["oc"].level(l) for l in range(rig["NL"])]
            pr = [o["orf"].level(l) for l in range(rig["NL"])]
            Xw = o["last"]["Xw"][o["last"]["valid"] != 0]
            r = oracle.align(pc, pr, o["tab"]["inv_sf"], o["tab"]["sf"], Xw, rig["scenes"][i]["T_ref"], T0[i], K, mode=mode)
            assert g["ok"][i] == r["ok"], (mode, i)
            assert np.array_equal(g["iters"][i][:len(r["iters"])], r["iters"])
            if r["ok"] and mode != 3:
                assert np.abs(g["T"][i] - r["T"]).max() <= POSE_TOL
            else:
                assert np.abs(g["T"][i] - T0[i]).max() == 0      # pose untouched
            assert abs(g["error"][i] - r["error"]) <= 1e-7 * max(1.0, abs(r["error"]))


def test_image_align_ragged_point_counts(sd, oracle, rig):
    """Point counts that fill neither the 100-point (modes 2, 3) nor the 300-point (modes 0, 1) capacity, scattered validity
    flags, and one frame with just eight points: the input class behind the device fault of round 1 (a development build of
    k_align read the per-point LDS projections of slots beyond the gathered points -- uninitialised coordinates -- and
    used them as image addresses).  Every thread of the current kernel owns one point and idles when
    it has none; this test pins that for all four modes."""
    trk, B = rig["trk"], rig["B"]
    rng = np.random.default_rng(4)
    # (fewer than three points make H rank deficient: there the reference's own result is decided by the rounding of its
    # per-pixel accumulation order, which no parallel sum reproduces -- DESIGN.md section 3; not a parity case)
    counts = [37, 8, 101, 299]
    cases = []
    for i in range(B):
        c = {k: v.copy() for k, v in rig["oras"][i]["last"].items()}
        idx = np.flatnonzero(c["valid"])
        keep = rng.choice(idx, size=counts[i], replace=False)
        c["valid"][:] = 0
        c["valid"][keep] = 1
        cases.append(c)
    trk.set_last(0, cases)
    try:
        T0 = [synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ s["T_cur"] for s in rig["scenes"]]
        for mode in (0, 1, 2, 3):
            Tinit = T0 if mode < 2 else [np.eye(4) for _ in range(B)]
            trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], Tinit)
            trk.align(B, mode=mode)
            g = trk.get_align(0, B)
            for i in range(B):
                o = rig["oras"][i]
                pc = [o["oc"].level(l) for l in range(rig["NL"])]
                pr = [o["orf"].level(l) for l in range(rig["NL"])]
                Xw = cases[i]["Xw"][cases[i]["valid"] != 0]
                r = oracle.align(pc, pr, o["tab"]["inv_sf"], o["tab"]["sf"], Xw, rig["scenes"][i]["T_ref"], Tinit[i], K, mode=mode)
                assert g["ok"][i] == r["ok"] and np.array_equal(g["iters"][i][:len(r["iters"])], r["iters"]), (mode, i, g["iters"][i][:len(r["iters"])], r["iters"])
                if r["ok"] and mode != 3:
                    assert np.abs(g["T"][i] - r["T"]).max() <= POSE_TOL, (mode, i)
                else:
                    assert np.abs(g["T"][i] - Tinit[i]).max() == 0
    finally:
        trk.set_last(0, [o["last"] for o in rig["oras"]])


def test_search_by_projection_bit_exact(sd, oracle, rig, match_mode):
    trk, B = rig["trk"], rig["B"]
    # th = 64: windows of hundreds of pixels overflow the LDS candidate list (per-point slow path of the kernel)
    for th, check_ori, use_truth in [(8.0, True, True), (16.0, True, False), (8.0, False, True), (64.0, True, True)]:
        T = [s["T_cur"] if use_truth else synth.se3_exp((0.004, 0, 0), (0, 0.1, 0)) @ s["T_cur"] for s in rig["scenes"]]
        trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T)
        trk.match(B, th=th, mono=True, check_ori=check_ori)
        cm, nm = trk.get_matches(0, B)
        for i in range(B):
            o = rig["oras"][i]
            n, ocm = oracle.search_by_projection(o["ck"], o["cd"], o["tab"]["sf"], BOUNDS, K, T[i], rig["scenes"][i]["T_ref"],
                                                 o["last"], th=th, mono=True, check_ori=check_ori)
            assert nm[i] == n, (th, check_ori, i, nm[i], n)
            assert np.array_equal(cm[i, :len(ocm)], ocm)
            assert (cm[i, len(ocm):] == -1).all()
            assert n > 50


def test_match_overwrite_and_claim_semantics(sd, oracle, rig, match_mode):
    """obs == 0 map points may be overwritten by later ones; obs > 0 ones block their keypoint."""
    trk, B = rig["trk"], rig["B"]
    cases = []
    for i in range(B):
        c = {k: v.copy() for k, v in rig["oras"][i]["last"].items()}
        c["obs"][::3] = 0
        # duplicate some points so several map points compete for one keypoint
        c["Xw"][1:200:2] = c["Xw"][0:199:2]
        c["desc"][1:200:2] = c["desc"][0:199:2]
        c["octave"][1:200:2] = c["octave"][0:199:2]
        cases.append(c)
    trk.set_last(0, cases)
    T = [s["T_cur"] for s in rig["scenes"]]
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T)
    trk.match(B, th=8.0, mono=True, check_ori=True)
    cm, nm = trk.get_matches(0, B)
    for i in range(B):
        o = rig["oras"][i]
        n, ocm = oracle.search_by_projection(o["ck"], o["cd"], o["tab"]["sf"], BOUNDS, K, T[i], rig["scenes"][i]["T_ref"],
                                             cases[i], th=8.0)
        assert nm[i] == n and np.array_equal(cm[i, :len(ocm)], ocm)
    trk.set_last(0, [o["last"] for o in rig["oras"]])     # restore


def test_rgbd_stereo_gates_bit_exact(sd, oracle, rig, match_mode):
    """RGB-D frames: ComputeStereoFromRGBD (mvuRight/mvDepth) and the bForward / bBackward / uRight gates of
    SearchByProjection (src/ORBmatcher.cc:965-966,999-1004,1020-1025).  bf = 4 (baseline 7.7 mm) so that the
    rig's centimetre motions land on both sides of the mb threshold."""
    trk, B = rig["trk"], rig["B"]
    bf = 4.0
    mb = np.float32(bf) / np.float32(K[0])
    yy, xx = np.mgrid[0:480, 0:640].astype(np.float32)
    depth = np.stack([(1.5 + 0.8 * np.sin(xx * 0.013 + i) * np.cos(yy * 0.017)).astype(np.float32) for i in range(B)])
    depth[:, 100:140, :] = 0.0            # holes: no depth -> mvuRight = -1
    depth[:, :, 300:310] = -1.0
    try:
        trk.set_camera(*K, bf, BOUNDS)
        trk.stereo_from_depth(depth)
        ur, dd = trk.get_stereo(0, B)
        ours = []
        for i in range(B):
            o = rig["oras"][i]
            our, odd = oracle.stereo_from_rgbd(o["ck"], o["ck"], depth[i], bf)
            n = len(our)
            assert np.array_equal(ur[i, :n], our) and np.array_equal(dd[i, :n], odd)
            assert (ur[i, n:] == -1).all() and (our == -1).any() and (our > 0).any()
            ours.append(our)
        seen = set()
        for th, scale in [(15.0, 1.0), (7.0, 1.0), (15.0, 0.0)]:
            # scale 0: current pose == last pose -> neither forward nor backward
            T = [s["T_cur"] if scale else s["T_ref"] for s in rig["scenes"]]
            trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T)
            trk.match(B, th=th, mono=False, check_ori=True)
            cm, nm = trk.get_matches(0, B)
            for i in range(B):
                o = rig["oras"][i]
                Tl = rig["scenes"][i]["T_ref"]
                twc = -T[i][:3, :3].T @ T[i][:3, 3]
                tlc = Tl[:3, :3] @ twc + Tl[:3, 3]
                seen.add("f" if tlc[2] > mb else ("b" if -tlc[2] > mb else "n"))
                n, ocm = oracle.search_by_projection(o["ck"], o["cd"], o["tab"]["sf"], BOUNDS, K, T[i], Tl, o["last"], th=th,
                                                     mono=False, check_ori=True, u_right=ours[i], mbf=bf, mb=mb)
                assert nm[i] == n, (th, i, nm[i], n)
                assert np.array_equal(cm[i, :len(ocm)], ocm)
                # and the gate matters: the monocular result differs somewhere in the batch
        assert seen == {"f", "b", "n"}, seen
    finally:
        trk.set_camera(*K, 0.0, BOUNDS)
        trk.set_uright(0, np.full((B, 8), -1, np.float32))
        trk.stereo_from_depth(np.zeros((B, 480, 640), np.float32))   # resets every mvuRight to -1


def test_pnp_ransac_matches_oracle(sd, oracle, rig):
    trk, B = rig["trk"], rig["B"]
    T = [s["T_cur"] for s in rig["scenes"]]
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T)
    trk.match(B, th=8.0, mono=True, check_ori=True)
    cm, nm = trk.get_matches(0, B)
    rs = oracle.glibc_rand_stream(4 * 200)
    trk.set_rand(0, np.tile(rs, (B, 1)))
    trk.pnp(B, 0.99, 10, 200, 4, 0.28, 5.991, 200)
    g = trk.get_pnp(0, B)
    for i in range(B):
        o = rig["oras"][i]
        n = len(o["ck"])
        valid = (cm[i, :n] >= 0).astype(np.uint8)
        Xw = np.zeros((n, 3))
        Xw[valid != 0] = o["last"]["Xw"][cm[i, :n][valid != 0]]
        p = oracle.PnPOracle(valid, np.stack([o["ck"]["x"], o["ck"]["y"]], 1), o["ck"]["octave"], o["tab"]["sigma2"], Xw, K)
        p.set_ransac(0.99, 10, 200, 4, 0.28, 5.991)
        r = p.iterate(200, rs)
        pr = p.params()
        assert (g["N"][i], g["min_inliers"][i], g["max_its"][i]) == (pr["N"], pr["min_inliers"], pr["max_its"])
        assert g["ok"][i] == r["ok"] and g["no_more"][i] == r["no_more"]
        assert g["iterations"][i] == r["iterations"], (i, g["iterations"][i], r["iterations"])
        assert g["n_inliers"][i] == r["n_inliers"]
        assert np.array_equal(g["inliers"][i, :n], r["inliers"])
        assert np.abs(g["T"][i] - r["T"]).max() <= POSE_TOL
        # and the solve is geometrically sane (EPnP on a noisy, gently curved scene: cm-level)
        assert np.abs(r["T"][:3, 3] - T[i][:3, 3]).max() < 0.06


def test_pnp_degenerate_inputs(sd, oracle, rig):
    trk, B = rig["trk"], rig["B"]
    # no valid map points -> no matches -> N < minInliers -> empty Mat + bNoMore
    cases = []
    for i in range(B):
        c = {k: v.copy() for k, v in rig["oras"][i]["last"].items()}
        c["valid"][:] = 0
        cases.append(c)
    trk.set_last(0, cases)
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], [s["T_cur"] for s in rig["scenes"]])
    trk.match(B)
    cm, nm = trk.get_matches(0, B)
    assert (nm == 0).all() and (cm == -1).all()
    trk.set_rand(0, np.tile(oracle.glibc_rand_stream(800), (B, 1)))
    trk.pnp(B, 0.99, 10, 200, 4, 0.28, 5.991, 200)
    g = trk.get_pnp(0, B)
    assert (~g["ok"]).all() and g["no_more"].all() and (g["n_inliers"] == 0).all() and (g["T"] == 0).all()
    # ImageAlign with no points -> false, pose untouched
    trk.align(B, 0)
    a = trk.get_align(0, B)
    assert (~a["ok"]).all()
    trk.set_last(0, [o["last"] for o in rig["oras"]])


def test_epnp_device_vs_oracle(sd, oracle):
    """compute_pose alone: well-conditioned n-point sets and rank-deficient 4-point minimal sets
    (the 12x12 Gram matrix then has a 4-D null space whose basis is decided by last-bit
    rounding: equality here shows the device follows the oracle's Jacobi rotations exactly)."""
    from sdslam_amd.capi import debug_epnp
    rng = np.random.default_rng(11)
    for trial in range(12):
        n = 4 if trial < 8 else int(rng.integers(6, 80))
        T = synth.se3_exp(rng.normal(size=3) * 0.1, rng.normal(size=3) * 5.0)
        Xc = np.stack([rng.uniform(-1.2, 1.2, n), rng.uniform(-0.9, 0.9, n), rng.uniform(1.0, 5.0, n)], 1)
        Xw = ((Xc - T[:3, 3]) @ T[:3, :3]).astype(np.float32).astype(np.float64)
        Xc = Xw @ T[:3, :3].T + T[:3, 3]
        uv = np.stack([K[0] * Xc[:, 0] / Xc[:, 2] + K[2], K[1] * Xc[:, 1] / Xc[:, 2] + K[3]], 1)
        uv = (uv + rng.normal(size=uv.shape) * 0.3).astype(np.float32).astype(np.float64)
        R, t, e = oracle.epnp(Xw, uv, K)
        Rg, tg, eg = debug_epnp(Xw, uv, K)
        assert np.abs(R - Rg).max() <= 1e-9 and np.abs(t - tg).max() <= 1e-9, (trial, n, np.abs(t - tg).max())


@pytest.mark.parametrize("opts", [{}, {"extract.fast0_early": 0}, {"extract.pyr_early": 1}, {"track.align_start": 0}, {"track.align_start": 1}],
                         ids=["default", "fast0_late", "pyr_early", "align_after_extraction", "align_after_pyramid"])
def test_pipelined_steps_match_isolated_steps(sd, oracle, opts):
    """Back-to-back steps without host synchronisation (extraction of batch n+1 overlaps tracking of batch n on
    the double-buffered extractor; level-0 FAST -- and optionally the resize chain -- of batch n+1 start behind the
    SELECTION of batch n, beside its descriptors) give exactly the results of the same steps run one at a time, under
    every setting of the scheduling options."""
    with sd.options(opts):
        _pipelined_vs_isolated(sd)


def _pipelined_vs_isolated(sd):
    B = 4
    scenes_a = [synth.make_scene(60 + i, (0.02, -0.01, 0.015), (0.4, -0.3, 0.5)) for i in range(B)]
    scenes_b = [synth.make_scene(70 + i, (-0.02, 0.02, -0.01), (-0.5, 0.2, 0.3)) for i in range(B)]
    cur = sd.ORBextractor(*CFG, 640, 480, B)
    ref = sd.ORBextractor(*CFG, 640, 480, B)
    trk = sd.Tracker(cur, ref, max_points=1000, max_batch=B, pnp_max_iterations=200)
    trk.set_camera(*K, 0.0, BOUNDS)
    rs = synth.glibc_rand_stream(800)
    trk.set_rand(0, np.tile(rs, (B, 1)))
    from sdslam_amd.capi import DeviceBuffer

    def prepare(scenes):
        rk, rd, rn = ref.extract_batch(np.stack([s["ref"] for s in scenes]))
        trk.set_last(0, [synth.tracking_case(i, rk[i, :rn[i]], rd[i, :rn[i]]) for i in range(B)])
        trk.set_poses(0, [s["T_ref"] for s in scenes], [s["T_cur"] for s in scenes])

    def run(d_frames):
        cur.extract_batch_device(d_frames.ptr, B, 640, 480)
        trk.align(B, 0)
        trk.match(B, 8.0, True, True)
        trk.pnp(B, 0.99, 10, 200, 4, 0.28, 5.991, 200)

    def results():
        al, (cm, nm), pn = trk.get_align(0, B), trk.get_matches(0, B), trk.get_pnp(0, B)
        k, d, n = cur.download(0, B)
        return np.stack(al["T"]), cm.copy(), nm.copy(), pn["T"].copy(), pn["inliers"].copy(), k.copy(), d.copy(), n.copy()

    bufs = []
    for scenes in (scenes_a, scenes_b):
        fr = np.stack([s["cur"] for s in scenes])
        db = DeviceBuffer(fr.nbytes)
        db.upload(fr)
        bufs.append(db)
    # isolated: B alone (same last-frame state as in the pipelined run below)
    prepare(scenes_b)
    run(bufs[1])
    iso = results()
    # pipelined: A, then B, then A, then B without any host sync in between; last results must equal `iso`
    for j in (0, 1, 0, 1):
        run(bufs[j])
    pip = results()
    for x, y in zip(iso, pip):
        assert np.array_equal(x, y)
    assert pip[2].min() > 50


def test_local_map_search_bit_exact(sd, oracle, rig):
    """TrackLocalMap's search (isInFrustum + PredictScale + SearchByProjection(F, vpMapPoints, th)): in-view flags,
    projections, predicted levels and the assignment vector equal the oracle's, for th = 1 / 3 / 5 (mono, RGB-D,
    after relocalisation: src/Tracking.cc:931-937), with claimed keypoints, stereo gates and a list overflow."""
    trk, B = rig["trk"], rig["B"]
    log_sf = np.log(np.float32(rig["cfg"][1]))
    T = [s["T_cur"] for s in rig["scenes"]]
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T)
    cases = [synth.local_map_case(100 + i, rig["oras"][i]["ck"], rig["oras"][i]["cd"], T[i], n_extra=300 - 100 * (i % 2),
                                  scale_factor=rig["cfg"][1], nlevels=rig["NL"]) for i in range(B)]
    # cap the local maps at the tracker's max_points
    cases = [{k: v[:1000] for k, v in c.items()} for c in cases]
    claimed = []
    for i in range(B):
        c = np.zeros(len(rig["oras"][i]["ck"]), np.uint8)
        c[i::7] = 1
        claimed.append(c)
    ur = np.full((B, 1000), -1, np.float32)
    for i in range(B):
        n = len(rig["oras"][i]["ck"])
        ur[i, :n:3] = rig["oras"][i]["ck"]["x"][::3] - 3.0     # some keypoints carry a right coordinate
    try:
        trk.set_camera(*K, 4.0, BOUNDS)
        trk.set_uright(0, ur)
        trk.set_local(0, cases, kp_claimed=claimed)
        for th, nn in [(1.0, 0.8), (3.0, 0.8), (5.0, 0.6), (40.0, 0.9)]:     # th = 40 overflows the LDS candidate list
            trk.match_local(B, th=th, nnratio=nn)
            g = trk.get_local(0, B)
            for i in range(B):
                o = rig["oras"][i]
                n = len(o["ck"])
                r = oracle.search_local_points(o["ck"], o["cd"], o["tab"]["sf"], log_sf, BOUNDS, K, 4.0, T[i], cases[i], th=th, nnratio=nn,
                                               u_right=ur[i, :n], kp_claimed=claimed[i])
                M = len(cases[i]["cand"])
                assert np.array_equal(g["in_view"][i, :M], r["in_view"]), (th, i)
                assert np.array_equal(g["proj"][i, :M], r["proj"]) and np.array_equal(g["level"][i, :M], r["level"])
                assert np.array_equal(g["cos"][i, :M], r["cos"])
                assert g["n"][i] == r["n"], (th, i, g["n"][i], r["n"])
                assert np.array_equal(g["match"][i, :n], r["match"]), (th, i)
                assert (g["match"][i, n:] == -1).all()
                if th == 1.0:
                    assert r["n"] > 200
    finally:
        trk.set_camera(*K, 0.0, BOUNDS)
        trk.stereo_from_depth(np.zeros((B, 480, 640), np.float32))   # resets every mvuRight to -1


def test_pose_optimization_matches_oracle(sd, oracle, rig):
    """Optimizer::PoseOptimization on the frame-to-frame matches (source 0) and on the local-map matches (source 1),
    mono and with stereo observations: pose within 1e-5 of the oracle's g2o restatement, identical outlier flags
    and return values."""
    trk, B = rig["trk"], rig["B"]
    T = [synth.se3_exp((0.004, -0.003, 0.002), (0.01, 0.008, -0.012)) @ s["T_cur"] for s in rig["scenes"]]
    trk.set_last(0, [o["last"] for o in rig["oras"]])
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], [s["T_cur"] for s in rig["scenes"]])
    trk.match(B, th=8.0, mono=True, check_ori=True)
    cm, nm = trk.get_matches(0, B)
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T)          # start PoseOptimization from a perturbed pose
    inv_s2 = rig["oras"][0]["tab"]["inv_sigma2"]
    ur = np.full((B, 1000), -1, np.float32)
    for stereo in (False, True):
        try:
            if stereo:
                for i in range(B):
                    o = rig["oras"][i]
                    n = len(o["ck"])
                    Xw = o["last"]["Xw"][np.maximum(cm[i, :n], 0)]
                    zc = (Xw @ rig["scenes"][i]["T_cur"][:3, :3].T + rig["scenes"][i]["T_cur"][:3, 3])[:, 2]
                    ur[i, :n] = np.where(np.arange(n) % 3 == 0, o["ck"]["x"] - 40.0 / zc, -1).astype(np.float32)
                trk.set_camera(*K, 40.0, BOUNDS)
                trk.set_uright(0, ur)
            trk.pose_opt(B, source=0)
            g = trk.get_pose_opt(0, B)
            for i in range(B):
                o = rig["oras"][i]
                n = len(o["ck"])
                has = cm[i, :n] >= 0
                r = oracle.pose_optimization(o["ck"], has, o["last"]["Xw"][np.maximum(cm[i, :n], 0)], inv_s2, K, T[i],
                                             u_right=ur[i, :n] if stereo else None, bf=40.0 if stereo else 0.0)
                assert np.abs(g["T"][i] - r["T"]).max() <= POSE_TOL, (stereo, i, np.abs(g["T"][i] - r["T"]).max())
                assert np.array_equal(g["outlier"][i, :n], r["outlier"]) and not g["outlier"][i, n:].any()
                assert g["n_inliers"][i] == r["n_inliers"] and g["n_initial"][i] == r["info"][0] and g["n_bad"][i] == r["info"][1]
                # (g2o iteration / LM trial counts are not compared: once converged, the sign of rho is summation-order noise)
                assert g["rounds"][i] == 4 and 4 <= g["iterations"][i] <= 40 and g["lm_trials"][i] >= g["iterations"][i]
                assert np.abs(g["T"][i][:3, 3] - rig["scenes"][i]["T_cur"][:3, 3]).max() < 5e-3     # and it converges to the truth
        finally:
            trk.set_camera(*K, 0.0, BOUNDS)
            trk.stereo_from_depth(np.zeros((B, 480, 640), np.float32))
    # source 1: local-map matches
    Tc = [s["T_cur"] for s in rig["scenes"]]
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], Tc)
    cases = [{k: v[:1000] for k, v in synth.local_map_case(200 + i, rig["oras"][i]["ck"], rig["oras"][i]["cd"], Tc[i], scale_factor=rig["cfg"][1],
                                                            nlevels=rig["NL"]).items()} for i in range(B)]
    trk.set_local(0, cases)
    trk.match_local(B, th=1.0, nnratio=0.8)
    lm = trk.get_local(0, B)["match"]
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T)
    trk.pose_opt(B, source=1)
    g = trk.get_pose_opt(0, B)
    for i in range(B):
        o = rig["oras"][i]
        n = len(o["ck"])
        has = lm[i, :n] >= 0
        r = oracle.pose_optimization(o["ck"], has, cases[i]["Xw"][np.maximum(lm[i, :n], 0)], inv_s2, K, T[i])
        assert np.abs(g["T"][i] - r["T"]).max() <= POSE_TOL
        assert np.array_equal(g["outlier"][i, :n], r["outlier"]) and g["n_inliers"][i] == r["n_inliers"]
    # degenerate: fewer than 3 correspondences -> pose untouched, return value 0
    empty = [dict(o["last"]) for o in rig["oras"]]
    for c in empty:
        c["valid"] = np.zeros_like(c["valid"])
    trk.set_last(0, empty)
    trk.set_poses(0, [s["T_ref"] for s in rig["scenes"]], T)
    trk.match(B, th=8.0, mono=True, check_ori=True)
    trk.pose_opt(B, source=0)
    g = trk.get_pose_opt(0, B)
    for i in range(B):
        assert g["n_inliers"][i] == 0 and g["rounds"][i] == 0 and np.abs(g["T"][i] - T[i]).max() == 0
    trk.set_last(0, [o["last"] for o in rig["oras"]])


def test_track_with_motion_model_decisions(sd, oracle, rig, match_mode):
    """Tracking::TrackWithMotionModel as one call (src/Tracking.cc:654-718): the per-frame decisions taken on the device
    (failed alignment keeps the prediction, wider-window retry from the prediction, the two failure exits, outlier discard,
    nmatchesMap) equal the oracle's stage-by-stage composition; final pose within 1e-5, final mvpMapPoints identical."""
    trk, B = rig["trk"], rig["B"]
    scenes, oras = rig["scenes"], rig["oras"]
    seen = set()
    lasts = [dict(o["last"]) for o in oras]
    lasts[3]["obs"] = np.where(np.arange(len(lasts[3]["obs"])) % 40 == 0, 1, 0).astype(np.int32)   # few points "in the map"
    trk.set_last(0, lasts)
    try:
        cases = [  # (align_mode, th, prior perturbation per frame: translation scale, rotation scale)
            (0, 15.0, [(1.0, 1.0)] * 4),
            (1, 15.0, [(1.0, 1.0)] * 4),
            (-1, 1.5, [(0.3, 0.3), (1.0, 1.0), (2.0, 2.0), (6.0, 6.0)]),
            (-1, 0.6, [(0.5, 0.5), (1.5, 1.5), (3.0, 3.0), (0.2, 0.2)]),
            (0, 2.0, [(40.0, 40.0), (1.0, 1.0), (25.0, 25.0), (1.0, 1.0)]),
        ]
        for align_mode, th, pert in cases:
            T0 = [synth.se3_exp((0.004 * a, -0.003 * a, 0.002 * a), (0.05 * b, 0.03 * b, -0.04 * b)) @ s["T_cur"]
                  for (a, b), s in zip(pert, scenes)]
            trk.set_poses(0, [s["T_ref"] for s in scenes], T0)
            T_al = None
            if align_mode >= 0:      # the device's aligned poses (<= 1e-12 from the oracle's): the matcher is compared on these
                trk.align(B, align_mode)
                T_al = trk.get_align(0, B)["T"]
                trk.set_poses(0, [s["T_ref"] for s in scenes], T0)
            trk.track_with_motion_model(B, th=th, mono=True, align_mode=align_mode)
            tw, gp, (cm, nm), ga = trk.get_tracked(0, B), trk.get_pose_opt(0, B), trk.get_matches(0, B), trk.get_align(0, B)
            for i in range(B):
                o = oras[i]
                n = len(o["ck"])
                pc = [o["oc"].level(l) for l in range(rig["NL"])]
                pr = [o["orf"].level(l) for l in range(rig["NL"])]
                T_dev_al = None if T_al is None else T_al[i]
                r = oracle.track_with_motion_model(pc, pr, o["tab"], o["ck"], o["cd"], BOUNDS, K, scenes[i]["T_ref"], T0[i], lasts[i], th,
                                                   mono=True, align_mode=align_mode, T_aligned=T_dev_al)
                key = (align_mode, th, i)
                assert tw["status"][i] == r["status"], (key, tw["status"][i], r["status"], tw["nmatches"][i], r["nmatches"])
                assert tw["retried"][i] == r["retried"], key
                assert tw["nmatches"][i] == r["nmatches"] and tw["nmatches_map"][i] == r["nmatches_map"], key
                assert np.array_equal(cm[i, :n], r["match"]) and (cm[i, n:] == -1).all(), key
                assert np.abs(gp["T"][i] - r["T"]).max() <= POSE_TOL, (key, np.abs(gp["T"][i] - r["T"]).max())
                assert np.abs(ga["T"][i] - r["T"]).max() <= POSE_TOL          # the frame's pose IS the result
                assert not gp["outlier"][i].any()                             # flags cleared by the discard
                if r["status"] == 2 and th == 15.0:     # good prior, aligned: it also converges to the truth
                    assert np.abs(gp["T"][i][:3, 3] - scenes[i]["T_cur"][:3, 3]).max() < 5e-3
                seen.add((int(r["status"]), int(r["retried"])))
    finally:
        trk.set_last(0, [o["last"] for o in oras])
    # every exit of the function was taken by some frame: tracked with / without retry, few matches after the retry,
    # few inliers
    assert {(2, 0), (2, 1), (0, 1)} <= seen and any(s == 1 for s, _ in seen), seen


def test_motion_model_retry_list_longer_than_its_grid(sd, oracle, rig):
    """TrackWithMotionModel's retry pass walks a device-built list of frames with a grid of 128 one-wave workgroups: 320 slots
    (the rig's four frame pairs, 80 times) with priors that send at least half of them through the retry make every workgroup
    take several list entries; every slot equals the oracle's result for its scene (match vector, decisions, pose)."""
    CFG, NL = rig["cfg"], rig["NL"]
    scenes, oras = rig["scenes"], rig["oras"]
    REP, B = 80, 320
    cur = sd.ORBextractor(*CFG, 640, 480, B)
    ref = sd.ORBextractor(*CFG, 640, 480, B)
    trk = None
    try:
        cur.extract_batch(np.stack([scenes[i % 4]["cur"] for i in range(B)]))
        ref.extract_batch(np.stack([scenes[i % 4]["ref"] for i in range(B)]))
        trk = sd.Tracker(cur, ref, max_points=1000, max_batch=B, pnp_max_iterations=300)
        trk.set_camera(*K, 0.0, BOUNDS)
        trk.set_last(0, [oras[i % 4]["last"] for i in range(B)])
        th = 0.6
        pert = [(0.5, 0.5), (1.5, 1.5), (3.0, 3.0), (2.5, 2.5)]
        T0 = [synth.se3_exp((0.004 * a, -0.003 * a, 0.002 * a), (0.05 * b, 0.03 * b, -0.04 * b)) @ s["T_cur"] for (a, b), s in zip(pert, scenes)]
        trk.set_poses(0, [scenes[i % 4]["T_ref"] for i in range(B)], [T0[i % 4] for i in range(B)])
        trk.track_with_motion_model(B, th=th, mono=True, align_mode=-1)
        tw, gp, (cm, nm) = trk.get_tracked(0, B), trk.get_pose_opt(0, B), trk.get_matches(0, B)
        exp = []
        for i in range(4):
            o = oras[i]
            pc = [o["oc"].level(l) for l in range(NL)]
            pr = [o["orf"].level(l) for l in range(NL)]
            exp.append(oracle.track_with_motion_model(pc, pr, o["tab"], o["ck"], o["cd"], BOUNDS, K, scenes[i]["T_ref"], T0[i], o["last"], th,
                                                      mono=True, align_mode=-1))
        assert sum(int(r["retried"]) for r in exp) * REP > 128, [r["retried"] for r in exp]   # more list entries than workgroups
        for b in range(B):
            r, n = exp[b % 4], len(oras[b % 4]["ck"])
            assert tw["status"][b] == r["status"] and tw["retried"][b] == r["retried"] and tw["nmatches"][b] == r["nmatches"], b
            assert np.array_equal(cm[b, :n], r["match"]) and (cm[b, n:] == -1).all(), b
            assert np.abs(gp["T"][b] - r["T"]).max() <= POSE_TOL, b
    finally:
        if trk is not None:
            trk.close()
        cur.close()
        ref.close()


def test_track_local_map_after_motion_model(sd, oracle, rig):
    """Tracking::TrackLocalMap (src/Tracking.cc:720-751) chained behind TrackWithMotionModel without a host round trip:
    the local search is closed where the frame match has observations, PoseOptimization runs over the union of both
    match vectors, mnMatchesInliers decides.  Everything equals the oracle's composition of the stages."""
    trk, B = rig["trk"], rig["B"]
    scenes, oras = rig["scenes"], rig["oras"]
    log_sf = np.log(np.float32(rig["cfg"][1]))
    lasts = [dict(o["last"]) for o in oras]
    for i, l in enumerate(lasts):       # a third of the last frame's points are not in the map yet (Observations() == 0)
        l["obs"] = (np.arange(len(l["obs"])) % 3 != i % 3).astype(np.int32)
    lasts[2]["valid"] = (np.arange(len(lasts[2]["valid"])) < 25).astype(np.uint8)       # frame 2 enters with few frame matches
    cases = [{k: v[:1000] for k, v in synth.local_map_case(300 + i, oras[i]["ck"], oras[i]["cd"], scenes[i]["T_cur"], scale_factor=rig["cfg"][1],
                                                            nlevels=rig["NL"]).items()} for i in range(B)]
    cases[3] = {k: v[:12] for k, v in cases[3].items()}                                  # frame 3 has an almost empty local map
    T0 = [synth.se3_exp((0.004, -0.003, 0.002), (0.05, 0.03, -0.04)) @ s["T_cur"] for s in scenes]
    trk.set_last(0, lasts)
    trk.set_local(0, cases)
    try:
        for th, min_inl in ((1.0, 30), (3.0, 30), (5.0, 2000)):
            trk.set_poses(0, [s["T_ref"] for s in scenes], T0)
            trk.track_with_motion_model(B, th=8.0, mono=True, align_mode=0)
            fm, _ = trk.get_matches(0, B)
            T_mm = trk.get_pose_opt(0, B)["T"]
            trk.track_local_map(B, th=th, min_inliers=min_inl)
            g, gp, gl = trk.get_local_map(0, B), trk.get_pose_opt(0, B), trk.get_local(0, B)
            for i in range(B):
                o = oras[i]
                n = len(o["ck"])
                r = oracle.track_local_map(o["ck"], o["cd"], o["tab"], log_sf, BOUNDS, K, T_mm[i], fm[i, :n], lasts[i], cases[i], th=th,
                                           min_inliers=min_inl)
                key = (th, i)
                assert np.array_equal(gl["match"][i, :n], r["local_match"]), key
                want = np.where(r["local_match"] >= 0, r["local_match"] + 1000, r["frame_match"])
                assert np.array_equal(g["match"][i, :n], want) and (g["match"][i, n:] == -1).all(), key
                assert g["n_points"][i] == r["n_points"] and g["n_local"][i] == r["n_local"], key
                assert np.array_equal(gp["outlier"][i, :n], r["outlier"]), key
                assert g["n_inliers"][i] == r["n_inliers"] and g["status"][i] == r["status"], (key, g["n_inliers"][i], r["n_inliers"])
                assert np.abs(gp["T"][i] - r["T"]).max() <= POSE_TOL, (key, np.abs(gp["T"][i] - r["T"]).max())
                if th == 1.0:
                    assert (r["status"] == 2) == (i != 3 or r["n_inliers"] >= 30)
                    if i < 2:
                        assert r["n_local"] > 100 and r["status"] == 2
                else:
                    assert th != 5.0 or r["status"] == 1          # gate above what any frame reaches
    finally:
        trk.set_last(0, [o["last"] for o in oras])


def test_track_with_motion_model_rgbd(sd, oracle, rig):
    """The same call on RGB-D frames (bMono = false): mvuRight from the depth image gates the search
    (src/ORBmatcher.cc:1020-1025) and adds stereo edges to PoseOptimization; results equal the oracle's composition."""
    trk, B = rig["trk"], rig["B"]
    scenes, oras = rig["scenes"], rig["oras"]
    bf = 4.0
    mb = np.float32(bf) / np.float32(K[0])
    depth = np.zeros((B, 480, 640), np.float32)
    for i, s in enumerate(scenes):      # depth of the scene surface along each pixel's ray, holes every 5th column
        v, u = np.mgrid[0:480, 0:640].astype(np.float64)
        R, t = s["T_cur"][:3, :3], s["T_cur"][:3, 3]
        rays = np.stack([(u - K[2]) / K[0], (v - K[3]) / K[1], np.ones_like(u)], -1)
        Ow = -R.T @ t
        Xw = synth.intersect_surface(Ow, rays @ R, 2.0)
        zc = (Xw @ R.T + t)[..., 2]
        depth[i] = zc.astype(np.float32)
        depth[i, :, ::5] = 0
    T0 = [synth.se3_exp((0.004, -0.003, 0.002), (0.05, 0.03, -0.04)) @ s["T_cur"] for s in scenes]
    try:
        trk.set_camera(*K, bf, BOUNDS)
        trk.stereo_from_depth(depth)
        ur, _ = trk.get_stereo(0, B)
        for align_mode, th in ((0, 8.0), (-1, 2.0)):
            trk.set_poses(0, [s["T_ref"] for s in scenes], T0)
            T_al = None
            if align_mode >= 0:
                trk.align(B, align_mode)
                T_al = trk.get_align(0, B)["T"]
                trk.set_poses(0, [s["T_ref"] for s in scenes], T0)
            trk.track_with_motion_model(B, th=th, mono=False, align_mode=align_mode)
            tw, gp, (cm, nm) = trk.get_tracked(0, B), trk.get_pose_opt(0, B), trk.get_matches(0, B)
            for i in range(B):
                o = oras[i]
                n = len(o["ck"])
                assert (ur[i, :n] >= 0).sum() > 300 and (ur[i, :n] < 0).sum() > 50       # stereo and mono edges
                pc = [o["oc"].level(l) for l in range(rig["NL"])]
                pr = [o["orf"].level(l) for l in range(rig["NL"])]
                r = oracle.track_with_motion_model(pc, pr, o["tab"], o["ck"], o["cd"], BOUNDS, K, scenes[i]["T_ref"], T0[i], o["last"], th,
                                                   mono=False, align_mode=align_mode, u_right=ur[i, :n], mbf=bf, mb=mb,
                                                   T_aligned=None if T_al is None else T_al[i])
                key = (align_mode, i)
                assert (tw["status"][i], tw["retried"][i], tw["nmatches"][i], tw["nmatches_map"][i]) == \
                       (r["status"], r["retried"], r["nmatches"], r["nmatches_map"]), key
                assert np.array_equal(cm[i, :n], r["match"]), key
                assert np.abs(gp["T"][i] - r["T"]).max() <= POSE_TOL, (key, np.abs(gp["T"][i] - r["T"]).max())
                assert r["status"] == 2
    finally:
        trk.set_camera(*K, 0.0, BOUNDS)
        trk.stereo_from_depth(np.zeros((B, 480, 640), np.float32))


@pytest.fixture(scope="module", params=["p8", "p5"])
def kfmap(sd, oracle, request):
    """One current frame and 8 keyframes of a small map: six see the current frame's scene from nearby or distant
    poses, two show another place.  Slot order = the order the reference would try / list them."""
    tex_a, tex_b = synth.make_image(71, 1280, 960), synth.make_image(72, 1280, 960)
    T_cur = synth.se3_exp((0.015, -0.01, 0.01), (0.3, -0.2, 0.4))
    kf_motion = [(tex_b, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),                 # another place
                 (tex_a, (0.30, -0.20, 0.10), (6.0, -5.0, 8.0)),            # same place, far away: alignment diverges
                 (tex_a, (0.02, -0.005, 0.012), (0.35, -0.1, 0.3)),         # close
                 (tex_b, (0.01, 0.0, 0.0), (0.1, 0.0, 0.0)),
                 (tex_a, (0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),                 # close
                 (tex_a, (0.05, 0.03, -0.02), (1.0, 0.8, -0.9)),            # medium
                 (tex_a, (0.012, -0.012, 0.008), (0.3, -0.25, 0.45)),       # closest
                 (tex_a, (-0.04, 0.02, 0.03), (-0.7, 0.5, 0.2))]            # medium
    NK = len(kf_motion)
    CFG = CFGS[request.param]
    T_kf = [synth.se3_exp(u, w) for _, u, w in kf_motion]
    cur = sd.ORBextractor(*CFG, 640, 480, 1)
    ref = sd.ORBextractor(*CFG, 640, 480, NK)
    im_cur = synth.render_plane_view(tex_a, T_cur)
    im_kf = np.stack([synth.render_plane_view(t, T) for (t, _, _), T in zip(kf_motion, T_kf)])
    ck, cd, cn = cur.extract_batch(im_cur[None])
    rk, rd, rn = ref.extract_batch(im_kf)
    oc = oracle.OrbOracle(*CFG)
    ock, ocd = oc.extract(im_cur)
    assert np.array_equal(ock, ck[0, :cn[0]])
    kfs = []
    for i in range(NK):
        orf = oracle.OrbOracle(*CFG)
        ork, ord_ = orf.extract(im_kf[i])
        assert np.array_equal(ork, rk[i, :rn[i]])
        kfs.append(dict(orf=orf, last=synth.keyframe_case(ork, ord_, T_kf[i], max_points=400)))
    trk = sd.Tracker(cur, ref, max_points=1000, max_batch=NK, pnp_max_iterations=300)   # cur holds ONE frame
    trk.set_camera(*K, 0.0, BOUNDS)
    trk.set_last(0, [k["last"] for k in kfs])
    yield dict(NK=NK, T_cur=T_cur, T_kf=T_kf, trk=trk, cur=cur, ref=ref, oc=oc, ck=ock, cd=ocd, kfs=kfs, tab=oc.tables(), NL=CFG[2],
               name=request.param)
    trk.close()
    cur.close()
    ref.close()


def test_relocalization_over_all_keyframes(sd, oracle, kfmap):
    """Tracking::Relocalization (src/Tracking.cc:1064-1097) with every keyframe attempt as one batch slot against the
    broadcast current frame: per slot the three stages equal the oracle's sequential ImageAlign(frame, kf, fast) ->
    SearchByProjection -> PoseOptimization, and the winner is where the reference's loop stops."""
    m, trk, NK = kfmap, kfmap["trk"], kfmap["NK"]
    trk.set_poses(0, m["T_kf"], m["T_kf"])                  # mCurrentFrame.SetPose(kf->GetPose())
    th = 15.0
    winner, st = trk.relocalize(NK, cur_frame=0, th=th, mono=True)
    ga, (cm, nm), gp = trk.get_align(0, NK), trk.get_matches(0, NK), trk.get_pose_opt(0, NK)
    pc = [m["oc"].level(l) for l in range(m["NL"])]
    n = len(m["ck"])
    expect = -1
    for i in range(NK):
        k = m["kfs"][i]
        pr = [k["orf"].level(l) for l in range(m["NL"])]
        Xw = k["last"]["Xw"][k["last"]["valid"] != 0]
        ra = oracle.align(pc, pr, m["tab"]["inv_sf"], m["tab"]["sf"], Xw, m["T_kf"][i], m["T_kf"][i], K, mode=2)
        assert ga["ok"][i] == ra["ok"] == st[i, 0], i
        T_al = ra["T"] if ra["ok"] else m["T_kf"][i]
        assert np.abs(ga["T"][i] - T_al).max() <= POSE_TOL
        # the later stages of every slot are checked from the DEVICE's aligned pose (a 1e-12 pose difference may flip a
        # float comparison in the matcher; that sensitivity is the aligner's tolerance, not the matcher's)
        nmo, ocm = oracle.search_by_projection(m["ck"], m["cd"], m["tab"]["sf"], BOUNDS, K, ga["T"][i], m["T_kf"][i], k["last"],
                                               th=th, mono=True, check_ori=True)
        assert nm[i] == nmo == st[i, 1] and np.array_equal(cm[i, :n], ocm), i
        has = ocm >= 0
        rp = oracle.pose_optimization(m["ck"], has, k["last"]["Xw"][np.maximum(ocm, 0)], m["tab"]["inv_sigma2"], K, ga["T"][i])
        assert gp["n_inliers"][i] == rp["n_inliers"] == st[i, 2], i
        assert np.array_equal(gp["outlier"][i, :n], rp["outlier"])
        assert np.abs(gp["T"][i] - rp["T"]).max() <= POSE_TOL
        if expect < 0 and ra["ok"] and nmo >= 20 and rp["n_inliers"] >= 10:
            expect = i
    assert winner == expect
    # the wrong-place and too-far keyframes fail one of the gates.  8 x 1.2: the first close keyframe (slot 2) wins; 5 x 2.0: the
    # fast alignment starts on a 40 x 30 image, where slots 2 ... 6 exceed its 0.01 error gate (src/ImageAlign.cc:159-163) in the
    # oracle and on the device alike, and the loop stops at slot 7
    assert winner == {"p8": 2, "p5": 7}[m["name"]]
    assert np.abs(gp["T"][winner][:3, 3] - m["T_cur"][:3, 3]).max() < 5e-3
    # no keyframe of the right place -> Relocalization returns false
    trk.set_last(0, [m["kfs"][0]["last"], m["kfs"][3]["last"]])
    try:
        # (slot 1 now pairs kf 3's points with kf 1's pyramid: garbage in, "no" out)
        w2, st2 = trk.relocalize(2, cur_frame=0, th=th, mono=True)
        assert w2 == -1
    finally:
        trk.set_last(0, [k["last"] for k in m["kfs"]])


def test_relocalization_rgbd_broadcast(sd, oracle, kfmap):
    """RGB-D relocalisation: the ONE current frame's mvuRight row (from its depth image) serves every keyframe slot --
    search gates and stereo edges equal the oracle's per-keyframe sequence."""
    m, trk, NK = kfmap, kfmap["trk"], kfmap["NK"]
    bf = 4.0
    mb = np.float32(bf) / np.float32(K[0])
    v, u = np.mgrid[0:480, 0:640].astype(np.float64)
    R, t = m["T_cur"][:3, :3], m["T_cur"][:3, 3]
    rays = np.stack([(u - K[2]) / K[0], (v - K[3]) / K[1], np.ones_like(u)], -1)
    Xs = synth.intersect_surface(-R.T @ t, rays @ R, 2.0)
    depth = (Xs @ R.T + t)[..., 2].astype(np.float32)
    depth[:, ::4] = 0
    n = len(m["ck"])
    try:
        trk.set_camera(*K, bf, BOUNDS)
        trk.stereo_from_depth(depth[None])                # one current frame in the cur extractor
        ur = trk.get_stereo(0, 1)[0][0, :n]
        assert (ur >= 0).sum() > 300 and (ur < 0).sum() > 100
        trk.set_poses(0, m["T_kf"], m["T_kf"])
        winner, st = trk.relocalize(NK, cur_frame=0, th=15.0, mono=False)
        ga, (cm, nm), gp = trk.get_align(0, NK), trk.get_matches(0, NK), trk.get_pose_opt(0, NK)
        expect = -1
        for i in range(NK):
            k = m["kfs"][i]
            nmo, ocm = oracle.search_by_projection(m["ck"], m["cd"], m["tab"]["sf"], BOUNDS, K, ga["T"][i], m["T_kf"][i], k["last"],
                                                   th=15.0, mono=False, check_ori=True, u_right=ur, mbf=bf, mb=mb)
            assert nm[i] == nmo == st[i, 1] and np.array_equal(cm[i, :n], ocm), i
            rp = oracle.pose_optimization(m["ck"], ocm >= 0, k["last"]["Xw"][np.maximum(ocm, 0)], m["tab"]["inv_sigma2"], K, ga["T"][i],
                                          u_right=ur, bf=bf)
            assert gp["n_inliers"][i] == rp["n_inliers"] == st[i, 2] and np.array_equal(gp["outlier"][i, :n], rp["outlier"]), i
            assert np.abs(gp["T"][i] - rp["T"]).max() <= POSE_TOL
            if expect < 0 and st[i, 0] and nmo >= 20 and rp["n_inliers"] >= 10:
                expect = i
        assert winner == expect and winner >= 0
    finally:
        trk.set_camera(*K, 0.0, BOUNDS)
        trk.stereo_from_depth(np.zeros((1, 480, 640), np.float32))


def test_detect_loop_candidates(sd, oracle, kfmap):
    """LoopClosing::DetectLoop's candidate search (src/LoopClosing.cc:115-149): KF-KF ImageAlign of the current keyframe
    against all keyframes in one launch, then the reference's loop (exclusions, skip-after-failure, 1.5 x best)."""
    m, trk, NK = kfmap, kfmap["trk"], kfmap["NK"]
    trk.set_poses(0, m["T_kf"], [np.eye(4)] * NK)
    # level-4 KF-KF alignment rarely exceeds the 0.03 gate on this scene; the failure the loop meets in practice is a
    # keyframe without map points ("No points to track!").  Slots 3 and 5 are such keyframes, so the reference's
    # skip-after-failure hides slots 4 and 6 -- the closest keyframe -- unless an exclusion shifts the walk.
    lasts = [dict(k["last"]) for k in m["kfs"]]
    for i in (3, 5):
        lasts[i]["valid"] = np.zeros_like(lasts[i]["valid"])
    trk.set_last(0, lasts)
    pc = [m["oc"].level(l) for l in range(m["NL"])]
    ok, err = [], []
    for i in range(NK):
        pr = [m["kfs"][i]["orf"].level(l) for l in range(m["NL"])]
        Xw = lasts[i]["Xw"][lasts[i]["valid"] != 0]
        r = oracle.align(pc, pr, m["tab"]["inv_sf"], m["tab"]["sf"], Xw, m["T_kf"][i], np.eye(4), K, mode=3)
        ok.append(r["ok"]); err.append(r["error"])

    def reference_loop(excluded):
        cand, best, i = {}, 1e10, 0
        while i < NK:                           # for (i = 0; i < kfs.size(); i++)
            if not excluded[i]:
                if not ok[i]:
                    i += 1                      # "Skip some keyframes"
                else:
                    cand[i] = err[i]
                    best = min(best, err[i])
            i += 1
        return sorted(j for j, e in cand.items() if e < best * 1.5), best

    for excluded in ([0] * NK, [0, 0, 1, 0, 0, 0, 1, 0], [1] * NK, [0, 1, 0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 1, 0, 0]):
        g = trk.detect_loop(NK, cur_frame=0, excluded=excluded)
        want, best = reference_loop(excluded)
        assert list(g["candidates"]) == want, (excluded, g, want)
        assert abs(g["best_error"] - best) <= 1e-7 * max(1.0, abs(best))
        for i in range(NK):
            assert abs(g["errors"][i] - err[i]) <= 1e-7 * max(1.0, abs(err[i]))
    assert sum(ok) >= 3 and not all(ok)
    g = trk.detect_loop(NK, cur_frame=0, excluded=None)
    assert list(g["candidates"]) == reference_loop([0] * NK)[0] and 6 not in g["candidates"]
    shifted = [0, 0, 0, 1, 0, 1, 0, 0]                      # excluding the point-less keyframes un-hides slots 4 and 6
    g = trk.detect_loop(NK, cur_frame=0, excluded=shifted)
    assert list(g["candidates"]) == reference_loop(shifted)[0]
    if m["name"] == "p8":                                   # (on the 40 x 30 level-4 images of 5 x 2.0 slot 6 is aligned, but not within 1.5 x best)
        assert 6 in g["candidates"]
    trk.set_last(0, [k["last"] for k in m["kfs"]])


def test_degenerate_frames_through_the_whole_chain(sd, oracle):
    """A textureless current frame (0 keypoints), an empty last frame and an empty local map go through every stage
    without faults and give the reference's 'nothing to do' outcomes; a partial batch (n < max_batch) works."""
    B = 3
    sc = synth.make_scene(90, (0.02, -0.01, 0.015), (0.4, -0.3, 0.5))
    flat = np.full((480, 640), 127, np.uint8)
    cur = sd.ORBextractor(*CFG, 640, 480, B)
    ref = sd.ORBextractor(*CFG, 640, 480, B)
    kps, desc, n = cur.extract_batch(np.stack([flat, sc["cur"]]))          # 2 of 3 frames
    assert n[0] == 0 and n[1] > 500
    rk, rd, rn = ref.extract_batch(np.stack([sc["ref"], sc["ref"]]))
    trk = sd.Tracker(cur, ref, max_points=1000, max_batch=B, pnp_max_iterations=200)
    trk.set_camera(*K, 0.0, BOUNDS)
    last = synth.tracking_case(1, rk[0, :rn[0]], rd[0, :rn[0]])
    empty = {k: v[:0] for k, v in last.items()}
    trk.set_last(0, [last, empty])
    trk.set_poses(0, [sc["T_ref"]] * 2, [sc["T_cur"]] * 2)
    trk.set_rand(0, np.tile(synth.glibc_rand_stream(800), (2, 1)))
    trk.align(2, 0)
    al = trk.get_align(0, 2)
    assert al["ok"][0] and not al["ok"][1]                       # frame 1: "No points to track!" -> false, pose untouched
    assert np.abs(al["T"][1] - sc["T_cur"]).max() == 0
    trk.set_poses(0, [sc["T_ref"]] * 2, [sc["T_cur"]] * 2)
    trk.match(2, 8.0, True, True)
    cm, nm = trk.get_matches(0, 2)
    assert nm[0] == 0 and nm[1] == 0 and (cm == -1).all()        # no keypoints / no map points
    trk.pnp(2, 0.99, 10, 200, 4, 0.28, 5.991, 200)
    pn = trk.get_pnp(0, 2)
    assert not pn["ok"].any() and pn["no_more"].all() and (pn["N"] == 0).all()
    trk.pose_opt(2, 0)
    po = trk.get_pose_opt(0, 2)
    assert (po["n_inliers"] == 0).all() and np.abs(po["T"][0] - sc["T_cur"]).max() == 0
    pts = synth.local_map_case(3, kps[1, :n[1]], desc[1, :n[1]], sc["T_cur"])
    pts = {k: v[:1000] for k, v in pts.items()}
    trk.set_local(0, [pts, {k: v[:0] for k, v in pts.items()}])
    trk.match_local(2, 1.0, 0.8)
    lm = trk.get_local(0, 2)
    assert lm["n"][0] == 0 and lm["n"][1] == 0 and (lm["match"] == -1).all()
    assert lm["in_view"][0].sum() > 500 and not lm["in_view"][1].any()      # points project into the flat frame, nothing to match


def test_two_thousand_features_through_both_tracking_calls(sd, oracle):
    """The tracker's capacity limits (2000 keypoints -> 2048-entry key arrays, 32 edges per PoseOptimization lane, windows
    with more candidates than a 32-lane half at th = 30): TrackWithMotionModel and TrackLocalMap still equal the oracle."""
    cfg = (2000, 1.2, 8, 12)
    s = synth.make_scene(33)
    cur, ref = sd.ORBextractor(*cfg, 640, 480, 1), sd.ORBextractor(*cfg, 640, 480, 1)
    ck, cd, cn = cur.extract_batch(s["cur"][None])
    rk, rd, rn = ref.extract_batch(s["ref"][None])
    assert cn[0] == 2000 and rn[0] == 2000
    oc, orf = oracle.OrbOracle(*cfg), oracle.OrbOracle(*cfg)
    ock, ocd = oc.extract(s["cur"])
    ork, ord_ = orf.extract(s["ref"])
    assert np.array_equal(ock, ck[0, :cn[0]]) and np.array_equal(ork, rk[0, :rn[0]])
    last = synth.tracking_case(33, ork, ord_, max_points=2000)
    last["obs"] = (np.arange(len(last["obs"])) % 4 != 0).astype(np.int32)
    trk = sd.Tracker(cur, ref, 2048, 1, 300)
    trk.set_camera(*K, 0.0, BOUNDS)
    trk.set_last(0, [last])
    T0 = synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ s["T_cur"]
    tab = oc.tables()
    pc, pr = [oc.level(l) for l in range(8)], [orf.level(l) for l in range(8)]
    n = len(ock)
    for th in (8.0, 30.0):
        trk.set_poses(0, [s["T_ref"]], [T0])
        trk.align(1, 0)
        T_al = trk.get_align(0, 1)["T"][0]
        trk.set_poses(0, [s["T_ref"]], [T0])
        trk.track_with_motion_model(1, th=th)
        tw, po = trk.get_tracked(0, 1), trk.get_pose_opt(0, 1)
        cm, _ = trk.get_matches(0, 1)
        r = oracle.track_with_motion_model(pc, pr, tab, ock, ocd, BOUNDS, K, s["T_ref"], T0, last, th, T_aligned=T_al)
        assert (tw["status"][0], tw["nmatches"][0], tw["nmatches_map"][0]) == (r["status"], r["nmatches"], r["nmatches_map"])
        assert np.array_equal(cm[0, :n], r["match"]) and np.abs(po["T"][0] - r["T"]).max() <= POSE_TOL
        assert r["status"] == 2 and r["nmatches"] > 1000
        pts = {k: v[:2048] for k, v in synth.local_map_case(5, ock, ocd, s["T_cur"], n_extra=40).items()}
        trk.set_local(0, [pts])
        trk.track_local_map(1, th=1.0)
        tl, po2, lm = trk.get_local_map(0, 1), trk.get_pose_opt(0, 1), trk.get_local(0, 1)
        r2 = oracle.track_local_map(ock, ocd, tab, np.log(np.float32(1.2)), BOUNDS, K, po["T"][0], cm[0, :n], last, pts, th=1.0)
        assert np.array_equal(lm["match"][0, :n], r2["local_match"]) and np.array_equal(po2["outlier"][0, :n], r2["outlier"])
        assert (tl["status"][0], tl["n_points"][0], tl["n_inliers"][0]) == (r2["status"], r2["n_points"], r2["n_inliers"])
        assert np.abs(po2["T"][0] - r2["T"]).max() <= POSE_TOL
        assert r2["n_points"] > 1700


def test_fresh_tracker_has_no_matches(sd, oracle):
    """Before any search has run the match vectors mean "no map point": PoseOptimization / TrackLocalMap called first
    find nothing to optimise instead of reading point 0 for every keypoint."""
    cur = sd.ORBextractor(*CFG, 640, 480, 1)
    ref = sd.ORBextractor(*CFG, 640, 480, 1)
    img = synth.make_image(9)
    cur.extract_batch(img[None])
    ref.extract_batch(img[None])
    trk = sd.Tracker(cur, ref, max_points=200, max_batch=1)
    trk.set_camera(*K, 0.0, BOUNDS)
    trk.set_poses(0, [np.eye(4)], [np.eye(4)])
    cm, nm = trk.get_matches(0, 1)
    assert (cm == -1).all()
    for source in (0, 1, 2):
        trk.pose_opt(1, source=source)
        g = trk.get_pose_opt(0, 1)
        assert g["n_initial"][0] == 0 and g["n_inliers"][0] == 0 and np.abs(g["T"][0] - np.eye(4)).max() == 0
    trk.track_local_map(1)
    tl = trk.get_local_map(0, 1)
    assert tl["status"][0] == 1 and tl["n_points"][0] == 0 and (tl["match"] == -1).all()


def test_tracker_errors_are_loud(sd):
    """Capacity / argument violations come back as SdError, never as silent truncation."""
    cur = sd.ORBextractor(*CFG, 640, 480, 2)
    ref = sd.ORBextractor(*CFG, 640, 480, 2)
    trk = sd.Tracker(cur, ref, max_points=100, max_batch=2, pnp_max_iterations=50)
    with pytest.raises(sd.SdError):
        trk.align(1, 0)                                   # camera not set
    trk.set_camera(*K, 0.0, BOUNDS)
    with pytest.raises(sd.SdError):
        trk.align(1, 0)                                   # nothing extracted yet
    img = synth.make_image(1)
    cur.extract_batch(img[None])
    ref.extract_batch(img[None])
    with pytest.raises(sd.SdError):
        trk.align(2, 0)                                   # only one frame was extracted
    with pytest.raises(sd.SdError):
        trk.align(1, 7)                                   # bad mode
    with pytest.raises(sd.SdError):
        trk.pnp(1, 0.99, 10, 300, 4, 0.4, 5.991, 300)     # iterations exceed the handle's pnp_max_iterations
    with pytest.raises(sd.SdError):
        trk.pnp(1, 0.99, 10, 50, 5, 0.4, 5.991, 50)       # minSet != 4
    with pytest.raises(sd.SdError):
        trk.pose_opt(1, source=3)
    with pytest.raises(sd.SdError):
        trk.track_with_motion_model(1, align_mode=2)      # only -1, 0, 1
    with pytest.raises(sd.SdError):
        trk.track_with_motion_model(2)                    # only one frame was extracted
    with pytest.raises(sd.SdError):
        trk.set_current_broadcast(2)                      # no such frame in the cur extractor
    with pytest.raises(sd.SdError):
        trk.relocalize(2, cur_frame=1)                    # frame 1 was not extracted
    big = dict(cand=np.ones(101, np.uint8), Xw=np.zeros((101, 3)), normal=np.zeros((101, 3)), min_dist=np.zeros(101, np.float32),
               max_dist=np.ones(101, np.float32), mf_max_dist=np.ones(101, np.float32), desc=np.zeros((101, 32), np.uint8),
               obs=np.zeros(101, np.int32))
    with pytest.raises((sd.SdError, ValueError)):
        trk.set_local(0, [big])                           # more local points than max_points
    with pytest.raises(sd.SdError):
        sd.Tracker(cur, ref, max_points=5000, max_batch=2)   # beyond the matcher's index width
