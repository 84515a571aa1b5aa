#!/usr/bin/env python3
"""bench.py -- tracked frames/s of the per-frame tracking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B]

Workload (BASELINE.json metric: "tracked frames/sec (ORB+ImageAlign+PnP) at 640x480, 1000 kp"):
one *step* tracks a batch of B independent 640x480 frames, all resident in HBM:
    ORB extract (8 levels x1.2, 1000 kp)  ->  ImageAlign (levels 4,3,2, <=30 GN its each)
    ->  SearchByProjection (th 8, mono, orientation check)  ->  PnP RANSAC (<=200 its, EPnP)
i.e. the TrackWithMotionModel sequence of SURVEY.md §3.2 / config C4.  The last frames'
pyramids, map points and the predicted poses are set up (untimed) beforehand, exactly as they
would be left behind by the previous tracking step.  With N > 1 the driver launches one rank
per GPU (torch.distributed / RCCL); frames are independent, so ranks share nothing on the data
path (weak scaling); the fixed-size per-frame pose records are all-gathered once.
Rank 0 prints ONE JSON line.

Variants of the step (same metric name, `config.pose_solver` says which): --pose-solver poseopt replaces PnP RANSAC by
Optimizer::PoseOptimization (what the reference's TrackWithMotionModel really calls); motion_model runs the
whole Tracking::TrackWithMotionModel as one device-side call; track adds Tracking::TrackLocalMap over a ~1000-point
local map.  --orb-only times the extraction alone, --res WxH other frame sizes, --batch 1 the single-frame latency.

Extra objects in the line:
  roofline     -- the dominant kernel stage: algorithmic bytes per launch / its mean
                  duration, measured with HIP events on the launch stream inside the timed region
  cpu_baseline -- the CPU oracle (a port of the reference path; the reference itself needs
                  OpenCV/Eigen and cannot be built here) timed on this host, 1 core, bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable
CFG = (1000, 1.2, 8, 20)
W, H = 640, 480
BOUNDS = (0.0, 640.0, 0.0, 480.0)
PNP = dict(probability=0.99, min_inliers=10, max_iterations=200, min_set=4, epsilon=0.28, th2=5.991)


def make_cases(n_unique, seed0, w=640, h=480):
    """n_unique two-view scenes with slightly different motions."""
    from sdslam_amd import synth
    rng = np.random.default_rng(seed0)
    scenes = []
    for i in range(n_unique):
        ups = np.array([0.02, -0.01, 0.015]) + rng.normal(size=3) * 0.004
        om = np.array([0.4, -0.3, 0.5]) + rng.normal(size=3) * 0.1
        scenes.append(synth.make_scene(seed0 + i, tuple(ups), tuple(om), w, h))
    return scenes


def cpu_baseline(scenes, lasts, T0s, rs, budget_s=15.0, pose_solver="pnp", locals_=None):
    """The oracle's full tracking step (-O3 -march=native build) on this host: 1 thread."""
    from oracle import oracle as O
    from sdslam_amd import synth
    K = (synth.FX, synth.FY, synth.CX, synth.CY)
    O.lib(True)
    ora_ref = [O.OrbOracle(*CFG, fast_build=True) for _ in scenes]
    for o, s in zip(ora_ref, scenes):
        o.extract(s["ref"])
    ref_pyr = [[o.level(l) for l in range(CFG[2])] for o in ora_ref]
    cur = O.OrbOracle(*CFG, fast_build=True)
    tab = cur.tables()
    t0 = time.perf_counter()
    n = 0
    t_stage = np.zeros(4)
    while True:
        i = n % len(scenes)
        s, last = scenes[i], lasts[i]
        ta = time.perf_counter()
        ck, cd = cur.extract(s["cur"])
        tb = time.perf_counter()
        pc = [cur.level(l) for l in range(CFG[2])]
        tb2 = time.perf_counter()
        if pose_solver in ("motion_model", "track"):      # the reference functions as a whole (oracle composition of the same stages)
            r = O.track_with_motion_model(pc, ref_pyr[i], tab, ck, cd, BOUNDS, K, s["T_ref"], T0s[i], last, 8.0, mono=True)
            tl = time.perf_counter()
            if pose_solver == "track":
                O.track_local_map(ck, cd, tab, np.log(np.float32(CFG[1])), BOUNDS, K, r["T"], r["match"], last, locals_[i], th=1.0)
            te = time.perf_counter()
            t_stage += [tb - ta, tl - tb2, te - tl, 0.0]
            n += 1
            if time.perf_counter() - t0 > budget_s or n >= 1000:
                break
            continue
        r = O.align(pc, ref_pyr[i], tab["inv_sf"], tab["sf"], last["Xw"][last["valid"] != 0], s["T_ref"], T0s[i], K, 0)
        tc = time.perf_counter()
        nm, cm = O.search_by_projection(ck, cd, tab["sf"], BOUNDS, K, r["T"], s["T_ref"], last, th=8.0)
        td = time.perf_counter()
        valid = (cm >= 0).astype(np.uint8)
        Xw = np.zeros((len(ck), 3))
        Xw[valid != 0] = last["Xw"][cm[valid != 0]]
        if pose_solver == "pnp":
            p = O.PnPOracle(valid, np.stack([ck["x"], ck["y"]], 1), ck["octave"], tab["sigma2"], Xw, K)
            p.set_ransac(PNP["probability"], PNP["min_inliers"], PNP["max_iterations"], 4, PNP["epsilon"], PNP["th2"])
            p.iterate(PNP["max_iterations"], rs)
        else:
            O.pose_optimization(ck, valid, Xw, tab["inv_sigma2"], K, r["T"])
        te = time.perf_counter()
        t_stage += [tb - ta, tc - tb2, td - tc, te - td]
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 1000:
            break
    dt = t_stage.sum()      # excludes the Python-side pyramid copies between the C calls
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} VGA frame pairs, full step (ORB extract + ImageAlign + SearchByProjection + "
                      f"{'PnP RANSAC' if pose_solver == 'pnp' else 'PoseOptimization'}"
                      f"{', composed as Tracking::TrackWithMotionModel (align+match+pose under image_align)' if pose_solver in ('motion_model', 'track') else ''}"
                      f"{' + Tracking::TrackLocalMap (under search_by_projection)' if pose_solver == 'track' else ''}), "
                      f"1 thread of {os.cpu_count()} host cpus",
            "ms_per_frame": {"orb_extract": t_stage[0] / n * 1e3, "image_align": t_stage[1] / n * 1e3,
                             "search_by_projection": t_stage[2] / n * 1e3, "pose_solve": t_stage[3] / n * 1e3}}


STAGE_KERNELS = {"pyramid": ["k_pyr_resize", "k_pyr_edges", "k_pyr_rows", "k_pyr_level"], "fast_nms": ["k_fast_cells"],
                 "select": ["k_select_level"], "blur": ["k_blur"], "orient_desc": ["k_orient_desc"], "image_align": ["k_align"],
                 "search_by_projection": ["k_match"], "pnp_ransac": ["k_pnp"]}
PMC_FRAMES = 1024       # frames per launch in the committed PMC passes (tools/run_profiles.sh: default batch)
N_SIMD, CLK_GHZ = 1024, 2.4


def pmc_for_stage(stage, batch):
    """HBM bytes per launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, MI355X_MICROARCH.md) and VALU issue
    figures of the stage's kernels from the newest committed profiles/*pmc_summary*.json; (None, None) if absent.
    PMC counters cannot be collected from inside this process; the passes are rocprofv3 runs of this same command."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary_b1024.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    tot, valu, dur = 0.0, 0.0, 0.0
    found = False
    def entry(name):      # templated kernels appear as "void k_align<4>"
        for key, val in d.items():
            if key == name or key.replace("void ", "").split("<")[0] == name:
                return val
        return None
    for k in STAGE_KERNELS.get(stage, []):
        e = entry(k)
        if not e or "FETCH_bytes_corrected_per_launch" not in e:
            continue
        found = True
        # launches per step: relative to a kernel that runs once per step (the pyramid and FAST kernels run per level)
        calls = max(1, round(e.get("duration_ns_samples", 1) / max(d.get("k_select_level", {}).get("duration_ns_samples", 1), 1)))
        tot += calls * (e["FETCH_bytes_corrected_per_launch"] + e.get("WRITE_bytes_per_launch", 0.0))
        valu += calls * e.get("SQ_INSTS_VALU_per_launch", 0.0)
        dur += calls * e.get("duration_ns_per_launch", 0.0)
    if not found:
        return None, None
    sc = batch / PMC_FRAMES
    min_ms = valu * sc / N_SIMD * 4 / (CLK_GHZ * 1e9) * 1e3      # one wave64 VALU instruction per 4 cycles per SIMD
    return tot * sc, {"wave_insts_per_launch": valu * sc, "min_ms_at_full_issue": min_ms,
                      "profiled_kernel_ms": dur * sc / 1e6, "source": os.path.basename(files[-1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="frames per GPU per step")
    ap.add_argument("--unique", type=int, default=8, help="distinct synthetic scenes (tiled to --batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--orb-only", action="store_true", help="time ORB extraction alone (configs[1] without tracking)")
    ap.add_argument("--res", default="640x480", help="frame size WxH (BASELINE configs[4] uses 1280x720 frames; the metric is quoted at 640x480)")
    ap.add_argument("--pose-solver", choices=["pnp", "poseopt", "motion_model", "track"], default="pnp",
                    help="pnp: PnPsolver RANSAC (the BASELINE metric); poseopt: Optimizer::PoseOptimization, the pose solve the reference's "
                         "TrackWithMotionModel really calls (SURVEY D1) -- reported under the same metric name with config.pose_solver set; "
                         "motion_model: the whole Tracking::TrackWithMotionModel as one call (retry search, failure exits, outlier discard); "
                         "track: motion_model followed by Tracking::TrackLocalMap over a ~1000-point local map per frame")
    args = ap.parse_args()
    global W, H, BOUNDS
    W, H = (int(v) for v in args.res.lower().split("x"))
    BOUNDS = (0.0, float(W), 0.0, float(H))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"

    import sdslam_amd
    from sdslam_amd import synth
    from sdslam_amd.capi import DeviceBuffer

    B = args.batch
    K = (synth.FX, synth.FY, synth.CX, synth.CY)
    nu = max(1, min(args.unique, B))
    scenes = make_cases(nu, 1000 + 100 * rank, W, H)
    idx = [i % nu for i in range(B)]
    cur_frames = np.stack([scenes[i]["cur"] for i in idx])
    ref_frames = np.stack([scenes[i]["ref"] for i in idx])
    d_cur = DeviceBuffer(cur_frames.nbytes)
    d_cur.upload(cur_frames)

    cur = sdslam_amd.ORBextractor(*CFG, W, H, B, device=local_rank)
    ref = sdslam_amd.ORBextractor(*CFG, W, H, B, device=local_rank)
    trk = sdslam_amd.Tracker(cur, ref, max_points=1000, max_batch=B, pnp_max_iterations=PNP["max_iterations"])
    trk.set_camera(*K, 0.0, BOUNDS)

    # ---- untimed setup: what the previous tracking step leaves behind
    rk, rd, rn = ref.extract_batch(ref_frames)                       # last frames' pyramids stay resident
    lasts_u = [synth.tracking_case(i, rk[i, :rn[i]], rd[i, :rn[i]]) for i in range(nu)]
    trk.set_last(0, [lasts_u[i] for i in idx])
    pert = synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04))   # motion-model prediction error
    T0_u = [pert @ s["T_cur"] for s in scenes]
    T_ref = [scenes[i]["T_ref"] for i in idx]
    T0 = [T0_u[i] for i in idx]
    rs = synth.glibc_rand_stream(4 * PNP["max_iterations"])
    trk.set_rand(0, np.tile(rs, (B, 1)))

    def step():
        cur.extract_batch_device(d_cur.ptr, B, W, H)
        if not args.orb_only and args.pose_solver in ("motion_model", "track"):
            trk.track_with_motion_model(B, th=8.0, mono=True, align_mode=0)
            if args.pose_solver == "track":
                trk.track_local_map(B, th=1.0)
        elif not args.orb_only:
            trk.align(B, 0)
            trk.match(B, 8.0, True, True)
            if args.pose_solver == "pnp":
                trk.pnp(B, PNP["probability"], PNP["min_inliers"], PNP["max_iterations"], PNP["min_set"], PNP["epsilon"],
                        PNP["th2"], PNP["max_iterations"])
            else:
                trk.pose_opt(B, 0)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    trk.set_poses(0, T_ref, T0)         # last-frame poses + motion-model predictions (resident, like the frames)
    locals_u = None
    if args.pose_solver == "track" and not args.orb_only:      # the local map of every frame (UpdateLocalMap is the caller's)
        ck, cd, cn = cur.extract_batch(np.stack([s_["cur"] for s_ in scenes]))
        locals_u = [{k: v[:1000] for k, v in synth.local_map_case(500 + i, ck[i, :cn[i]], cd[i, :cn[i]], scenes[i]["T_cur"]).items()}
                    for i in range(nu)]
        trk.set_local(0, [locals_u[i] for i in idx])
    for _ in range(args.warmup):
        step()
    cur.sync()
    cur.set_profiling(True)
    trk.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()                          # every step re-aligns from the predicted pose (Tprior is not overwritten)
    barrier()
    dt = time.perf_counter() - t0
    orb_ms = cur.stage_ms()
    trk_ms = trk.stage_ms() if not args.orb_only else np.zeros(3, np.float32)
    if args.pose_solver == "track" and not args.orb_only:
        trk_ms[1:] *= 2      # the timers average per call; search and PoseOptimization run twice per step here
    cur.set_profiling(False)
    trk.set_profiling(False)

    # ---- results: fixed-size per-frame records (pose + counts), all-gathered across ranks
    al = trk.get_align(0, B)
    if args.pose_solver == "pnp" or args.orb_only:
        pn = trk.get_pnp(0, B)
    elif args.pose_solver == "motion_model":
        po, tw = trk.get_pose_opt(0, B), trk.get_tracked(0, B)
        pn = dict(ok=tw["status"] == 2, n_inliers=tw["nmatches_map"], iterations=po["iterations"], N=po["n_initial"])
    elif args.pose_solver == "track":
        po, tl = trk.get_pose_opt(0, B), trk.get_local_map(0, B)
        pn = dict(ok=tl["status"] == 2, n_inliers=tl["n_inliers"], iterations=po["iterations"], N=po["n_initial"])
    else:
        po = trk.get_pose_opt(0, B)
        pn = dict(ok=po["n_inliers"] >= 10, n_inliers=po["n_inliers"], iterations=po["iterations"], N=po["n_initial"])
    cm, nm = trk.get_matches(0, B)
    from sdslam_amd import dist_util
    rec = dist_util.pack_records([t.T.ravel() for t in al["T"]], al["ok"], nm, pn["n_inliers"], pn["ok"])
    dev = torch.device("cuda", local_rank)
    dt = dist_util.max_over_ranks(dt, dist, dev)                      # MAX over ranks
    all_rec = dist_util.gather_records(rec, B * world, dist, dev)     # the only inter-GPU traffic
    assert all_rec.shape == (B * world, dist_util.RECORD_F64)

    if rank == 0:
        total_frames = B * args.steps * world
        names = cur.stage_names() + ["image_align", "search_by_projection", "pnp_ransac" if args.pose_solver == "pnp" else "pose_optimization"]
        stage_ms = np.concatenate([orb_ms, trk_ms])
        sbytes = list(cur.stage_bytes())
        # algorithmic bytes of the tracking stages, from what this run actually did
        P = float(np.mean([min(300, int(l["valid"].sum())) for l in lasts_u]))
        its = float(al["iters"][:, :8].sum(axis=1).mean())
        lv = sdslam_amd.plan_info(*CFG, W, H)["levels"]
        px_l = float(sum(lv[l, 0] * lv[l, 1] for l in (2, 3, 4)))
        sbytes += [its * P * (25 + 64) + 2 * px_l,                       # ImageAlign
                   P * (32 + 10 * 32),                                   # windowed Hamming (c ~ 10)
                   float(pn["iterations"].mean()) * (4 * 20 + float(pn["N"].mean()) * 24)]   # PnP
        dom = int(np.argmax(stage_ms))
        achieved = sbytes[dom] * B / (stage_ms[dom] * 1e-3) / 1e9
        traffic, valu = pmc_for_stage(names[dom], B)
        terr = float(np.mean([np.abs(al["T"][b][:3, 3] - scenes[idx[b]]["T_cur"][:3, 3]).max() for b in range(min(B, nu))]))
        line = {
            "metric": "tracked frames/sec (ORB+ImageAlign+PnP) at 640x480, 1000 kp",
            "value": total_frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"ORB extract only (BASELINE configs[1] without matching), {W}x{H}" if args.orb_only else
                                    f"BASELINE configs[3] at the configs[1] pyramid: {W}x{H}, 8-level x1.2, 1000 kp; "
                                    "ORB extract + ImageAlign (levels 4,3,2) + SearchByProjection + " +
                                    ("PnP RANSAC 200 its" if args.pose_solver == "pnp" else "Optimizer::PoseOptimization (g2o LM, 4x10 its)") +
                                    (", as one Tracking::TrackWithMotionModel call" if args.pose_solver in ("motion_model", "track") else "") +
                                    (" + Tracking::TrackLocalMap (th 1, <=1000 local points)" if args.pose_solver == "track" else "")),
                       "pose_solver": args.pose_solver,
                       "frames_per_gpu_per_step": B, "unique_scenes": nu, "inputs": "resident in HBM",
                       "pose_records": "all-gathered over RCCL" if world > 1 else "single GPU"},
            "stages_ms_per_step": {nm_: float(ms) for nm_, ms in zip(names, stage_ms)},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_frame": float(sbytes[dom]), "ms_per_step": float(stage_ms[dom]),
                         "note": "integer/byte kernel limited by VALU issue, not HBM (DESIGN.md section 6); traffic = HBM bytes "
                                 "per launch from the committed rocprofv3 FETCH_SIZE/WRITE_SIZE passes (profiles/), scaled to this batch",
                         "valu_issue": valu},
            "tracking": {"align_ok": int(al["ok"].sum()), "mean_gn_iterations": its, "mean_matches": float(nm.mean()),
                         "pnp_ok": int(pn["ok"].sum()), "mean_pnp_inliers": float(pn["n_inliers"].mean()),
                         "mean_pnp_iterations": float(pn["iterations"].mean()), "align_translation_err_m": terr},
        }
        if not args.no_cpu_baseline and world == 1 and not args.orb_only:
            line["cpu_baseline"] = cpu_baseline(scenes, lasts_u, T0_u, rs, pose_solver=args.pose_solver, locals_=locals_u)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line, default=float))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
