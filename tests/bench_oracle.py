"""The oracle's side of bench.py's step for ONE scene, all three pose-solver variants -- a picklable job so that
tests/test_bench_step_gpu.py can run the 64 benchmarked scenes x 2 pyramids through the CPU oracle in a process pool
(CPU-only workers: they never touch the GPU)."""
import numpy as np


def oracle_step(job):
    scene, cfg, K, bounds, rs, pnp = job
    from oracle import oracle as O
    from sdslam_amd import synth
    nl = cfg[2]
    oc, orf = O.OrbOracle(*cfg), O.OrbOracle(*cfg)
    ck, cd = oc.extract(scene["cur"])
    rk, rd = orf.extract(scene["ref"])
    tab = oc.tables()
    last = synth.tracking_case(0, rk, rd)                       # what bench.Workload builds from the GPU's ref keypoints
    T0 = synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04)) @ scene["T_cur"]
    pc, pr = [oc.level(l) for l in range(nl)], [orf.level(l) for l in range(nl)]
    al = O.align(pc, pr, tab["inv_sf"], tab["sf"], last["Xw"][last["valid"] != 0], scene["T_ref"], T0, K, 0)
    nm, cm = O.search_by_projection(ck, cd, tab["sf"], bounds, K, al["T"], scene["T_ref"], last, th=8.0)
    valid = (cm >= 0).astype(np.uint8)
    Xw = np.zeros((len(ck), 3))
    Xw[valid != 0] = last["Xw"][cm[valid != 0]]
    p = O.PnPOracle(valid, np.stack([ck["x"], ck["y"]], 1), ck["octave"], tab["sigma2"], Xw, K)
    p.set_ransac(pnp["probability"], pnp["min_inliers"], pnp["max_iterations"], 4, pnp["epsilon"], pnp["th2"])
    r_pnp = p.iterate(pnp["max_iterations"], rs)
    po = O.pose_optimization(ck, valid, Xw, tab["inv_sigma2"], K, al["T"])
    tw = O.track_with_motion_model(pc, pr, tab, ck, cd, bounds, K, scene["T_ref"], T0, last, 8.0, mono=True)
    return dict(ck=ck, cd=cd, rk=rk, rd=rd, al=al, nm=nm, cm=cm, pnp=r_pnp, pnp_params=p.params(), po=po, tw=tw)
