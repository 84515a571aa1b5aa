#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the CPU oracle (the reference ships no golden vectors and
cannot be built here, so these pin the ORACLE's behaviour against regressions and let the GPU
tests run against committed data).  Inputs are regenerated from seeds, never stored.
    python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from sdslam_amd import synth            # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
K = (synth.FX, synth.FY, synth.CX, synth.CY)
BOUNDS = (0.0, 640.0, 0.0, 480.0)

for name, cfg in (("p8", (1000, 1.2, 8, 20)), ("p5", (1000, 2.0, 5, 20))):
    for seed in (0, 1, 2, 3):
        e = O.OrbOracle(*cfg)
        k, d = e.extract(synth.make_image(seed))
        np.savez_compressed(os.path.join(OUT, f"orb_{name}_seed{seed}.npz"), kps=k, desc=d,
                            cell_totals=np.concatenate([e.cell_totals(l) for l in range(cfg[2])]),
                            level_checksum=np.array([int(e.level(l).astype(np.uint64).sum()) for l in range(cfg[2])]))

cfg = (1000, 1.2, 8, 20)
for seed in (20,):
    s = synth.make_scene(seed)
    oc, orf = O.OrbOracle(*cfg), O.OrbOracle(*cfg)
    ck, cd = oc.extract(s["cur"])
    rk, rd = orf.extract(s["ref"])
    last = synth.tracking_case(seed, rk, rd)
    tab = oc.tables()
    T0 = synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ s["T_cur"]
    al = O.align([oc.level(l) for l in range(8)], [orf.level(l) for l in range(8)], tab["inv_sf"], tab["sf"],
                 last["Xw"][last["valid"] != 0], s["T_ref"], T0, K, 0)
    nm, cm = O.search_by_projection(ck, cd, tab["sf"], BOUNDS, K, al["T"], s["T_ref"], last, th=8.0)
    valid = (cm >= 0).astype(np.uint8)
    Xw = np.zeros((len(ck), 3))
    Xw[valid != 0] = last["Xw"][cm[valid != 0]]
    rs = synth.glibc_rand_stream(800)
    p = O.PnPOracle(valid, np.stack([ck["x"], ck["y"]], 1), ck["octave"], tab["sigma2"], Xw, K)
    p.set_ransac(0.99, 10, 200, 4, 0.28, 5.991)
    r = p.iterate(200, rs)
    np.savez_compressed(os.path.join(OUT, f"track_seed{seed}.npz"), T0=T0, align_T=al["T"], align_iters=al["iters"],
                        align_error=al["error"], align_chi2=al["chi2"], n_matches=nm, cur_match=cm,
                        pnp_T=r["T"], pnp_inliers=r["inliers"], pnp_iterations=r["iterations"], pnp_n_inliers=r["n_inliers"])
    # stages added after the first fixtures: undistortion, local-map search, PoseOptimization (same scene)
    Kd, dist = (517.3, 516.5, 318.6, 255.3), (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)
    und = O.undistort_points(np.stack([ck["x"], ck["y"]], 1), Kd, dist)
    pts = {k: v[:1000] for k, v in synth.local_map_case(7, ck, cd, s["T_cur"]).items()}
    lm = O.search_local_points(ck, cd, tab["sf"], np.log(np.float32(1.2)), BOUNDS, K, 0.0, s["T_cur"], pts, th=1.0, nnratio=0.8)
    po = O.pose_optimization(ck, valid, Xw, tab["inv_sigma2"], K, T0)
    np.savez_compressed(os.path.join(OUT, f"track2_seed{seed}.npz"), undist=und, lm_match=lm["match"], lm_n=lm["n"],
                        lm_in_view=lm["in_view"], lm_level=lm["level"], lm_proj=lm["proj"], po_T=po["T"], po_outlier=po["outlier"],
                        po_n_inliers=po["n_inliers"])
    # the reference's composite functions on the same scene: TrackWithMotionModel (normal prior; a prior bad enough for
    # the wider-window retry with align_image_ off) and TrackLocalMap behind the first
    last3 = dict(last)
    last3["obs"] = (np.arange(len(last["obs"])) % 3 != 0).astype(np.int32)
    pc, pr = [oc.level(l) for l in range(8)], [orf.level(l) for l in range(8)]
    tw = O.track_with_motion_model(pc, pr, tab, ck, cd, BOUNDS, K, s["T_ref"], T0, last3, 8.0, mono=True)
    T0b = synth.se3_exp((0.008, -0.006, 0.004), (0.1, 0.06, -0.08)) @ s["T_cur"]
    twb = O.track_with_motion_model(pc, pr, tab, ck, cd, BOUNDS, K, s["T_ref"], T0b, last3, 1.0, mono=True, align_mode=-1)
    T0c = synth.se3_exp((0.012, -0.009, 0.006), (0.15, 0.09, -0.12)) @ s["T_cur"]
    twc = O.track_with_motion_model(pc, pr, tab, ck, cd, BOUNDS, K, s["T_ref"], T0c, last3, 1.0, mono=True, align_mode=-1)
    assert (twb["status"], twb["retried"], twc["status"], twc["retried"]) == (2, 1, 0, 1)
    tl = O.track_local_map(ck, cd, tab, np.log(np.float32(1.2)), BOUNDS, K, tw["T"], tw["match"], last3, pts, th=1.0)
    np.savez_compressed(os.path.join(OUT, f"track3_seed{seed}.npz"), T0b=T0b,
                        tw_info=np.array([tw["status"], tw["nmatches"], tw["nmatches_map"], tw["retried"]]), tw_T=tw["T"], tw_match=tw["match"],
                        twb_info=np.array([twb["status"], twb["nmatches"], twb["nmatches_map"], twb["retried"]]), twb_T=twb["T"],
                        twb_match=twb["match"], T0c=T0c,
                        twc_info=np.array([twc["status"], twc["nmatches"], twc["nmatches_map"], twc["retried"]]), twc_match=twc["match"],
                        tl_info=np.array([tl["status"], tl["n_points"], tl["n_inliers"], tl["n_local"]]), tl_T=tl["T"],
                        tl_local_match=tl["local_match"], tl_outlier=tl["outlier"])

# End-to-end goldens of the tracking step for BOTH pyramids -- p5 = the reference's own default
# (src/Config.cc:48-51) -- and four scenes; what they hold: tests/e2e_cases.py
sys.path.insert(0, os.path.join(ROOT, "tests"))
import e2e_cases as EC                  # noqa: E402
for name, cfg_ in EC.CFGS.items():
    for seed in EC.SEEDS:
        g = EC.oracle_case(O, cfg_, seed)
        np.savez_compressed(os.path.join(OUT, f"e2e_{name}_seed{seed}.npz"), **g)
        print(f"e2e {name} seed {seed}: align iters {g['al0_iters'].tolist()} ok {int(g['al0_ok'])}/{int(g['al2_ok'])}/{int(g['al3_ok'])}, "
              f"{int(g['n_matches'])} matches, pnp {g['pnp_info'].tolist()}, poseopt inliers {int(g['po_n_inliers'])}, tw {g['tw_info'].tolist()}")
print("golden written to", OUT, sorted(os.listdir(OUT)))

# PnP RANSAC on planted-outlier match vectors (tests/pnp_cases.py): every iterate() call of every scenario
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pnp_cases as PC                  # noqa: E402
s = synth.make_scene(20)
oc = O.OrbOracle(*cfg)
ck, cd = oc.extract(s["cur"])
tab = oc.tables()
gold = {}
for name, (kw, params, calls) in PC.SCENARIOS.items():
    last, cm, truth = PC.planted(7, ck, s["T_cur"], **kw)
    rs = synth.glibc_rand_stream(PC.rand_needed(params, calls))
    res, pr = PC.run_oracle(O, ck, tab["sigma2"], last, cm, params, calls, rs)
    gold[name + "_params"] = np.array([pr["N"], pr["min_inliers"], pr["max_its"]])
    gold[name + "_info"] = np.array([[r["ok"], r["iterations"], r["n_inliers"], r["no_more"]] for r in res], np.int32)
    gold[name + "_inliers"] = np.stack([r["inliers"] for r in res])
    gold[name + "_T"] = np.stack([r["T"] for r in res])
np.savez_compressed(os.path.join(OUT, "pnp_ransac_seed20.npz"), **gold)
print("pnp golden:", {k: v.tolist() for k, v in gold.items() if k.endswith("_info")})
