"""End-to-end tracking cases behind tests/golden/e2e_{p5,p8}_seed*.npz, shared by the golden
generator (tools/gen_golden.py), the CPU test that the oracle still reproduces them and the GPU test that the HIP path
matches them without running the oracle.

Both pyramids: "p8" = BASELINE's 8 levels x 1.2, "p5" = the reference's own default, 5 levels x 2.0 (reference
src/Config.cc:48-51), where ImageAlign's levels 4, 3, 2 are 40 x 30 / 80 x 60 / 160 x 120 images (src/ImageAlign.cc:36-39,57,382)
and the matcher's radius / octave window run on scale factors 2^n (src/ORBmatcher.cc:999-1004)."""
import numpy as np

from sdslam_amd import synth

K = (synth.FX, synth.FY, synth.CX, synth.CY)
BOUNDS = (0.0, 640.0, 0.0, 480.0)
CFGS = {"p8": (1000, 1.2, 8, 20), "p5": (1000, 2.0, 5, 20)}
SEEDS = (20, 21, 22, 23)
MOTIONS = [((0.02, -0.01, 0.015), (0.4, -0.3, 0.5)), ((0.0, 0.0, 0.0), (0.0, 0.0, 0.0)),
           ((-0.03, 0.02, -0.01), (-0.6, 0.2, 0.3)), ((0.01, 0.03, 0.02), (0.2, 0.5, -0.8))]
ALIGN_MODES = (0, 2, 3)       # (Frame, Frame) / (Frame, KeyFrame, fast) / (KeyFrame, KeyFrame)


def scene(seed):
    return synth.make_scene(seed, *MOTIONS[seed % 4])


def prior(s):
    return synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ s["T_cur"]


def last_frame(seed, rk, rd):
    last = synth.tracking_case(seed, rk, rd)
    last["obs"] = (np.arange(len(last["obs"])) % 3 != 0).astype(np.int32)
    return last


def oracle_case(O, cfg, seed):
    """Everything the golden of (cfg, seed) holds, computed by the oracle."""
    nl = cfg[2]
    s = scene(seed)
    oc, orf = O.OrbOracle(*cfg), O.OrbOracle(*cfg)
    ck, cd = oc.extract(s["cur"])
    rk, rd = orf.extract(s["ref"])
    last = last_frame(seed, rk, rd)
    tab = oc.tables()
    pc, pr = [oc.level(l) for l in range(nl)], [orf.level(l) for l in range(nl)]
    T0 = prior(s)
    out = dict(T0=T0, n_cur=np.int32(len(ck)), n_ref=np.int32(len(rk)))
    Xw_last = last["Xw"][last["valid"] != 0]
    for mode in ALIGN_MODES:
        Ti = T0 if mode == 0 else np.eye(4)
        al = O.align(pc, pr, tab["inv_sf"], tab["sf"], Xw_last, s["T_ref"], Ti, K, mode)
        out.update({f"al{mode}_T": al["T"], f"al{mode}_iters": al["iters"], f"al{mode}_ok": np.int32(al["ok"]),
                    f"al{mode}_error": np.float64(al["error"])})
    nm, cm = O.search_by_projection(ck, cd, tab["sf"], BOUNDS, K, out["al0_T"], s["T_ref"], last, th=8.0)
    valid = (cm >= 0).astype(np.uint8)
    Xw = np.zeros((len(ck), 3))
    Xw[valid != 0] = last["Xw"][cm[valid != 0]]
    p = O.PnPOracle(valid, np.stack([ck["x"], ck["y"]], 1), ck["octave"], tab["sigma2"], Xw, K)
    p.set_ransac(0.99, 10, 200, 4, 0.28, 5.991)
    r = p.iterate(200, synth.glibc_rand_stream(800))
    po = O.pose_optimization(ck, valid, Xw, tab["inv_sigma2"], K, T0)
    tw = O.track_with_motion_model(pc, pr, tab, ck, cd, BOUNDS, K, s["T_ref"], T0, last, 8.0, mono=True)
    out.update(n_matches=np.int32(nm), cur_match=cm, pnp_T=r["T"], pnp_inliers=r["inliers"],
               pnp_info=np.array([r["ok"], r["iterations"], r["n_inliers"]], np.int32),
               po_T=po["T"], po_outlier=po["outlier"], po_n_inliers=np.int32(po["n_inliers"]),
               tw_info=np.array([tw["status"], tw["nmatches"], tw["nmatches_map"], tw["retried"]], np.int32), tw_T=tw["T"], tw_match=tw["match"])
    return out
