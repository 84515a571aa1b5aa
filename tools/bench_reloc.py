#!/usr/bin/env python3
"""Measurement of SURVEY 8(f)-4: one current frame against a map of N keyframes.

  relocalize    Tracking::Relocalization (src/Tracking.cc:1064-1097): ImageAlign(frame, kf, fast) ->
                SearchByProjection(frame, kf) -> PoseOptimization for EVERY keyframe as one batch
  detect_loop   LoopClosing::DetectLoop's candidate search (src/LoopClosing.cc:115-149): KF-KF ImageAlign, level 4

python tools/bench_reloc.py [n_keyframes=1024] [steps=10]     -> one JSON line
The keyframe pyramids / map points are resident (they are the map); the timed region is the batched call including
its result download.  The CPU figure is the oracle doing the same attempts one keyframe after another, 1 thread, on a
sample of the keyframes (worst case of the reference's loop: no early winner).
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import sdslam_amd  # noqa: E402
from sdslam_amd import synth  # noqa: E402

NK = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K = (synth.FX, synth.FY, synth.CX, synth.CY)
NU = 8          # distinct keyframe views, tiled over the N slots

tex = synth.make_image(71, 1280, 960)
T_cur = synth.se3_exp((0.015, -0.01, 0.01), (0.3, -0.2, 0.4))
rng = np.random.Generator(np.random.PCG64(5))
T_kf = [synth.se3_exp(rng.normal(size=3) * 0.02, rng.normal(size=3) * 0.5) for _ in range(NU)]
im_cur = synth.render_plane_view(tex, T_cur)
im_kf = np.stack([synth.render_plane_view(tex, T) for T in T_kf])
idx = [i % NU for i in range(NK)]

cur = sdslam_amd.ORBextractor(*bench.CFG, bench.W, bench.H, 1)
ref = sdslam_amd.ORBextractor(*bench.CFG, bench.W, bench.H, NK)
cur.extract_batch(im_cur[None])
rk, rd, rn = ref.extract_batch(im_kf[idx])
lasts = [synth.keyframe_case(rk[i, :rn[i]], rd[i, :rn[i]], T_kf[i], max_points=400) for i in range(NU)]
trk = sdslam_amd.Tracker(cur, ref, max_points=1000, max_batch=NK)
trk.set_camera(*K, 0.0, bench.BOUNDS)
trk.set_last(0, [lasts[i] for i in idx])
poses = [T_kf[i] for i in idx]


def timed(fn):
    for _ in range(2):
        fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    return (time.perf_counter() - t0) / steps


trk.set_poses(0, poses, poses)
t_reloc = timed(lambda: trk.relocalize(NK, 0, th=15.0, mono=True, min_matches=10 ** 6))    # gate never passes: all attempts count
win, st = trk.relocalize(NK, 0, th=15.0, mono=True)
trk.set_poses(0, poses, [np.eye(4)] * NK)
t_loop = timed(lambda: trk.detect_loop(NK, 0))
g = trk.detect_loop(NK, 0)

# CPU: the oracle's sequential loop over a sample of the keyframes
from oracle import oracle as O  # noqa: E402

O.lib(True)
oc = O.OrbOracle(*bench.CFG, fast_build=True)
ock, ocd = oc.extract(im_cur)
pc = [oc.level(l) for l in range(8)]
tab = oc.tables()
orf = []
for i in range(NU):
    o = O.OrbOracle(*bench.CFG, fast_build=True)
    o.extract(im_kf[i])
    orf.append([o.level(l) for l in range(8)])
n_cpu, t_cpu_reloc, t_cpu_loop = 0, 0.0, 0.0
t_start = time.perf_counter()
while time.perf_counter() - t_start < 10.0 and n_cpu < NK:
    i = n_cpu % NU
    Xw = lasts[i]["Xw"][lasts[i]["valid"] != 0]
    ta = time.perf_counter()
    r = O.align(pc, orf[i], tab["inv_sf"], tab["sf"], Xw, T_kf[i], T_kf[i], K, mode=2)
    T = r["T"] if r["ok"] else T_kf[i]
    nm, cm = O.search_by_projection(ock, ocd, tab["sf"], bench.BOUNDS, K, T, T_kf[i], lasts[i], th=15.0, mono=True, check_ori=True)
    O.pose_optimization(ock, cm >= 0, lasts[i]["Xw"][np.maximum(cm, 0)], tab["inv_sigma2"], K, T)
    tb = time.perf_counter()
    O.align(pc, orf[i], tab["inv_sf"], tab["sf"], Xw, T_kf[i], np.eye(4), K, mode=3)
    tc = time.perf_counter()
    t_cpu_reloc += tb - ta
    t_cpu_loop += tc - tb
    n_cpu += 1

print(json.dumps({
    "workload": f"1 VGA frame vs {NK} keyframes (8x1.2 pyramid, <=400 map points each)",
    "relocalize": {"ms": t_reloc * 1e3, "keyframes_per_s": NK / t_reloc, "winner": int(win),
                   "slots_passing_all_gates": int(((st[:, 0] != 0) & (st[:, 1] >= 20) & (st[:, 2] >= 10)).sum())},
    "detect_loop": {"ms": t_loop * 1e3, "keyframes_per_s": NK / t_loop, "n_candidates": int(len(g["candidates"]))},
    "cpu_oracle_1thread": {"relocalize_keyframes_per_s": n_cpu / t_cpu_reloc, "detect_loop_keyframes_per_s": n_cpu / t_cpu_loop,
                           "sample": f"{n_cpu} keyframe attempts"},
}))
