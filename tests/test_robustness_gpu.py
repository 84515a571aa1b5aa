"""Regression tests for the round-1 advisor findings: the rotation-event list of k_match when the map holds
more points than the frame has keypoints, the extractor handle after a failed geometry change, the tracker behind an
extraction replayed as a hipGraph, and the validation of host-supplied octaves."""
import os
import subprocess
import sys

import numpy as np
import pytest

from sdslam_amd import synth

pytestmark = pytest.mark.gpu
K = (synth.FX, synth.FY, synth.CX, synth.CY)
BOUNDS = (0.0, 640.0, 0.0, 480.0)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sd():
    import sdslam_amd
    if sdslam_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests need a real MI355X")
    return sdslam_amd


def test_more_map_points_than_keypoints_with_overwrites(sd, oracle):
    """max_points (1000) > keypoint capacity (400-feature extractor, KP2 = 512): ~900 valid last-frame points with
    Observations() == 0 overwrite each other's assignments, so the rotation histogram records more events than there are
    keypoints (one per ASSIGNMENT, src/ORBmatcher.cc:1045-1052)."""
    s = synth.make_scene(31)
    cfg_small, cfg_big = (400, 1.2, 8, 20), (1000, 1.2, 8, 20)
    cur, ref = sd.ORBextractor(*cfg_small, 640, 480, 1), sd.ORBextractor(*cfg_small, 640, 480, 1)
    ck, cd, cn = cur.extract_batch(s["cur"][None])
    ref.extract_batch(s["ref"][None])
    big = oracle.OrbOracle(*cfg_big)
    rk, rd = big.extract(s["ref"])                                   # the map: 1000 points seen in the last frame
    last = synth.tracking_case(31, rk, rd, max_points=900)
    last["obs"][:] = 0                                               # nothing is ever "claimed": later points overwrite
    # three copies of every point a few centimetres apart: they land in the same windows and fight for the same keypoints
    for k in ("valid", "Xw", "desc", "octave", "angle", "obs"):
        last[k] = np.concatenate([last[k][:330]] * 3 + [last[k][990:]])[:1000]
    last["Xw"] = last["Xw"] + np.random.default_rng(1).normal(size=last["Xw"].shape) * 0.004
    assert int(last["valid"].sum()) > cur.cap
    trk = sd.Tracker(cur, ref, max_points=1000, max_batch=1, pnp_max_iterations=8)
    trk.set_camera(*K, 0.0, BOUNDS)
    trk.set_last(0, [last])
    tab = oracle.OrbOracle(*cfg_small).tables()
    n = cn[0]
    for th, ori in ((8.0, True), (30.0, True), (30.0, False)):
        trk.set_poses(0, [s["T_ref"]], [s["T_cur"]])
        trk.match(1, th, True, ori)
        cm, nm = trk.get_matches(0, 1)
        on, ocm = oracle.search_by_projection(ck[0, :n], cd[0, :n], tab["sf"], BOUNDS, K, s["T_cur"], s["T_ref"], last, th=th, check_ori=ori)
        assert nm[0] == on and np.array_equal(cm[0, :n], ocm), (th, ori, nm[0], on)
    assert on > 100


def test_extractor_survives_a_failed_geometry_change(sd, oracle):
    img = synth.make_image(5)
    ext = sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 1)
    k0, d0 = ext(img)
    k_tiny, _ = ext(np.full((2, 2), 7, np.uint8))   # a legal geometry (every level keeps >= 1 pixel) without a single corner
    assert len(k_tiny) == 0
    with pytest.raises(sd.SdError):
        ext(np.full((1, 1), 7, np.uint8))          # level sizes collapse to zero: refused, nothing may be left half-built
    k1, d1 = ext(img)
    assert np.array_equal(k0, k1) and np.array_equal(d0, d1)
    ok, od = oracle.OrbOracle(1000, 1.2, 8, 20).extract(img)
    assert np.array_equal(k1, ok) and np.array_equal(d1, od)
    k2, d2 = ext(synth.make_image(6, 320, 240))     # and a real geometry change still works afterwards
    ok2, od2 = oracle.OrbOracle(1000, 1.2, 8, 20).extract(synth.make_image(6, 320, 240))
    assert np.array_equal(k2, ok2) and np.array_equal(d2, od2)


def test_invalid_octave_is_refused(sd):
    s = synth.make_scene(20)
    cur, ref = sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 1), sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 1)
    cur.extract_batch(s["cur"][None])
    rk, rd, rn = ref.extract_batch(s["ref"][None])
    trk = sd.Tracker(cur, ref, 1000, 1, 8)
    last = synth.tracking_case(20, rk[0, :rn[0]], rd[0, :rn[0]])
    last["octave"][5] = 8                            # nlevels = 8: valid octaves are 0..7
    with pytest.raises(sd.SdError):
        trk.set_last(0, [last])
    last["octave"][5] = -1
    with pytest.raises(sd.SdError):
        trk.set_last(0, [last])
    last["valid"][5] = 0                             # an invalid entry's octave is never used
    trk.set_last(0, [last])


def test_tracker_behind_graph_replayed_extraction(sd):
    """Option "extract.use_graph": inside a captured graph the pyramid-done event is a graph node, not an event record; the
    tracker's early start (ImageAlign right behind the pyramid / the FAST launches) must then wait for the whole
    extraction.  Identical results with and without the graph."""
    from sdslam_amd.capi import DeviceBuffer
    s = [synth.make_scene(20 + i) for i in range(2)]
    fr = np.stack([x["cur"] for x in s])
    d = DeviceBuffer(fr.nbytes)
    d.upload(fr)
    outs = []
    for flag in (0, 1):
        with sd.options({"extract.use_graph": flag}):
            cur, ref = sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 2), sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 2)
            rk, rd, rn = ref.extract_batch(np.stack([x["ref"] for x in s]))
            trk = sd.Tracker(cur, ref, 1000, 2, 200)
            trk.set_camera(*K, 0.0, BOUNDS)
            trk.set_last(0, [synth.tracking_case(i, rk[i, :rn[i]], rd[i, :rn[i]]) for i in range(2)])
            trk.set_poses(0, [x["T_ref"] for x in s], [synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ x["T_cur"] for x in s])
            trk.set_rand(0, np.tile(synth.glibc_rand_stream(800), (2, 1)))
            for _ in range(3):                      # replayed graph from the second call on
                cur.extract_batch_device(d.ptr, 2, 640, 480)
                trk.align(2, 0)
                trk.match(2, 8.0, True, True)
                trk.pnp(2, 0.99, 10, 200, 4, 0.28, 5.991, 200)
            al, (cm, nm), pn = trk.get_align(0, 2), trk.get_matches(0, 2), trk.get_pnp(0, 2)
            outs.append(dict(T=np.stack(al["T"]), iters=al["iters"].copy(), cm=cm.copy(), nm=nm.copy(), pT=pn["T"].copy(), inl=pn["inliers"].copy()))
            trk.close()
            cur.close()
            ref.close()
    for k in outs[0]:
        assert np.array_equal(outs[0][k], outs[1][k]), k
    assert outs[0]["nm"].min() > 50


def test_cpp_frame_overloads_run_the_reference_call_sequences(sd, oracle, tmp_path):
    """The C++ drop-in path end to end on the GPU: tests/native/facade_frame.cc drives FrameTracker's overloads with
    reference-shaped Frame / KeyFrame / MapPoint types through the bodies of TrackWithMotionModel (src/Tracking.cc:668-693),
    SearchLocalPoints' search (:937), PnPsolver(F, matches) + find(), TrackReferenceKeyFrame (:583-644), one turn of
    Relocalization (:1069-1092) and one of DetectLoop (src/LoopClosing.cc:132-134); every printed result is compared with the
    oracle's composition of the same calls on the same two frames."""
    import struct
    s = synth.make_scene(20)
    cfg = (1000, 1.2, 8, 20)
    oc, orf = oracle.OrbOracle(*cfg), oracle.OrbOracle(*cfg)
    ck, cd = oc.extract(s["cur"])
    rk, rd = orf.extract(s["ref"])
    last = synth.tracking_case(20, rk, rd)
    T0 = synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ s["T_cur"]
    tab = oc.tables()
    pc, pr = [oc.level(l) for l in range(8)], [orf.level(l) for l in range(8)]
    # ---- the oracle's TrackWithMotionModel body (needed first: the local map's isInFrustum results hang on its pose)
    al = oracle.align(pc, pr, tab["inv_sf"], tab["sf"], last["Xw"][last["valid"] != 0], s["T_ref"], T0, K, 0)
    Ta = al["T"] if al["ok"] else T0
    nm, cm = oracle.search_by_projection(ck, cd, tab["sf"], BOUNDS, K, Ta, s["T_ref"], last, th=8.0)
    Xw = np.zeros((len(ck), 3))
    Xw[cm >= 0] = last["Xw"][cm[cm >= 0]]
    po = oracle.pose_optimization(ck, cm >= 0, Xw, tab["inv_sigma2"], K, Ta)
    kept = (cm >= 0) & (po["outlier"] == 0)                      # after "Discard outliers"
    pts = {k: v[:1000] for k, v in synth.local_map_case(7, ck, cd, s["T_cur"]).items()}
    lm = oracle.search_local_points(ck, cd, tab["sf"], np.log(np.float32(1.2)), BOUNDS, K, 0.0, po["T"], pts, th=1.0, nnratio=0.8,
                                    kp_claimed=kept.astype(np.uint8))
    raw = tmp_path / "frames.bin"
    with open(raw, "wb") as f:
        f.write(s["cur"].tobytes())
        f.write(s["ref"].tobytes())
        f.write(np.ascontiguousarray(s["T_ref"].T).tobytes())
        f.write(np.ascontiguousarray(T0.T).tobytes())
        idx = np.flatnonzero(last["valid"])
        f.write(struct.pack("<i", len(idx)))
        for i in idx:
            f.write(struct.pack("<i3d", int(i), *last["Xw"][i]))
        M = len(pts["cand"])
        f.write(struct.pack("<i", M))
        for i in range(M):
            f.write(struct.pack("<3d", *pts["Xw"][i]) + pts["desc"][i].tobytes() +
                    struct.pack("<ii3fif", int(pts["obs"][i]), int(lm["in_view"][i]), *[float(v) for v in lm["proj"][i]], int(lm["level"][i]),
                                float(lm["cos"][i])))
    exe = str(tmp_path / "facade_frame")
    libdir = os.path.dirname(sd.lib_path())
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-DRUN_ON_GPU", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "facade_frame.cc"), "-o", exe, "-L", libdir, "-lsdslam_hip",
                           f"-Wl,-rpath,{libdir}"])
    out = subprocess.run([exe, str(raw)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout[-500:], out.stderr[-2000:])
    lines = {l.split()[0]: l.split()[1:] for l in out.stdout.splitlines() if l and l.split()[0].isupper()}

    def pose(vals):
        return np.array([float(v) for v in vals]).reshape(4, 4).T

    # ---- TrackWithMotionModel
    res = lines["RESULT"]
    N, nmatches, ngood, nout = (int(v) for v in res[:4])
    T = pose(res[4:20])
    got_match = np.array([int(v) for v in lines["MATCH"]], np.int32)
    assert N == len(ck) and nmatches == nm and np.array_equal(got_match, cm)
    assert ngood == po["n_inliers"] and nout == int(po["outlier"][cm >= 0].sum())
    assert np.abs(T - po["T"]).max() <= 1e-5
    # ---- SearchByProjection(F, vpMapPoints, th) on the caller's isInFrustum results
    assert int(lines["RESULTLOCAL"][0]) == lm["n"] and lm["n"] > 100
    assert np.array_equal(np.array([int(v) for v in lines["MATCHLOCAL"]], np.int32), lm["match"])
    # ---- PnPsolver(F, frame matches) + find(): the unseeded rand() stream is glibc's seed-1 stream
    assert np.array_equal(np.array([int(v) for v in lines["PNPMATCH"]]) != 0, kept)
    Xk = np.zeros((len(ck), 3))
    Xk[kept] = last["Xw"][cm[kept]]
    p = oracle.PnPOracle(kept.astype(np.uint8), np.stack([ck["x"], ck["y"]], 1), ck["octave"], tab["sigma2"], Xk, K)
    p.set_ransac(0.99, 10, 200, 4, 0.28, 5.991)
    r = p.iterate(200, synth.glibc_rand_stream(1200))
    rp = lines["RESULTPNP"]
    assert int(rp[0]) == int(r["ok"]) == 1 and int(rp[1]) == r["n_inliers"]
    assert np.array_equal(np.array([int(v) for v in lines["INLPNP"]]) != 0, r["inliers"] != 0)
    assert np.abs(np.array([float(v) for v in rp[2:18]]).reshape(4, 4) - r["T"]).max() <= 1e-5
    # ---- TrackReferenceKeyFrame: the keyframe's points in ITS std::set order (pointer order of this run, printed)
    order = np.array([int(v) for v in lines["SETORDER"]])
    assert sorted(order) == sorted(np.flatnonzero(last["valid"]))
    rk_ = lines["RESULTKF"]
    nm_kf, ngood_kf, nmap_kf, retried = (int(v) for v in rk_[:4])
    T_al, T_fin = pose(rk_[4:20]), pose(rk_[20:36])
    alk = oracle.align(pc, pr, tab["inv_sf"], tab["sf"], last["Xw"][order[:300]], s["T_ref"], s["T_ref"], K, 1)
    assert alk["ok"] and np.abs(T_al - alk["T"]).max() <= 1e-5
    nmk, cmk = oracle.search_by_projection(ck, cd, tab["sf"], BOUNDS, K, T_al, s["T_ref"], last, th=8.0)   # from the device's aligned pose
    assert retried == 0 and nmk >= 20 and np.array_equal(np.array([int(v) for v in lines["MATCHKF"]], np.int32), cmk)
    Xk2 = np.zeros((len(ck), 3))
    Xk2[cmk >= 0] = last["Xw"][cmk[cmk >= 0]]
    pok = oracle.pose_optimization(ck, cmk >= 0, Xk2, tab["inv_sigma2"], K, T_al)
    assert ngood_kf == pok["n_inliers"] and nm_kf == nmk - int(pok["outlier"][cmk >= 0].sum())
    assert nmap_kf == int(((cmk >= 0) & (pok["outlier"] == 0)).sum())            # every point has Observations() > 0 here
    assert np.abs(T_fin - pok["T"]).max() <= 1e-5 and np.abs(T_fin[:3, 3] - s["T_cur"][:3, 3]).max() < 5e-3
    # ---- one turn of Relocalization: ComputePose(F, KF, fast) from the keyframe's pose
    rr = lines["RESULTRELOC"]
    ok_r, nm_r, ngood_r = (int(v) for v in rr[:3])
    T_al_r, T_fin_r = pose(rr[4:20]), pose(rr[20:36])
    alr = oracle.align(pc, pr, tab["inv_sf"], tab["sf"], last["Xw"][order[:300]], s["T_ref"], s["T_ref"], K, 2)
    assert ok_r == int(alr["ok"]) and abs(float(rr[3]) - alr["error"]) <= 1e-7 * max(1.0, abs(alr["error"]))
    if alr["ok"]:
        assert np.abs(T_al_r - alr["T"]).max() <= 1e-5
        nmr, cmr = oracle.search_by_projection(ck, cd, tab["sf"], BOUNDS, K, T_al_r, s["T_ref"], last, th=8.0)
        assert nm_r == nmr
        if nmr >= 20:
            Xr = np.zeros((len(ck), 3))
            Xr[cmr >= 0] = last["Xw"][cmr[cmr >= 0]]
            por = oracle.pose_optimization(ck, cmr >= 0, Xr, tab["inv_sigma2"], K, T_al_r)
            assert ngood_r == por["n_inliers"] and np.abs(T_fin_r - por["T"]).max() <= 1e-5
    else:
        assert nm_r == -1 and np.abs(T_al_r - s["T_ref"]).max() == 0
    # ---- one turn of DetectLoop: ComputePose(KF, KF)
    rl = lines["RESULTLOOP"]
    all_ = oracle.align(pc, pr, tab["inv_sf"], tab["sf"], last["Xw"][order[:300]], s["T_ref"], np.eye(4), K, 3)
    assert int(rl[0]) == int(all_["ok"]) and abs(float(rl[1]) - all_["error"]) <= 1e-7 * max(1.0, abs(all_["error"]))
