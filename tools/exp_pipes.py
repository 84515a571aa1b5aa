#!/usr/bin/env python3
"""Experiment: the bench workload split over P independent handle sets (sub-batches in flight on
separate streams).  python tools/exp_pipes.py <total_batch> <pipes> <steps>"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

import bench  # noqa: E402
import sdslam_amd  # noqa: E402
from sdslam_amd import synth  # noqa: E402
from sdslam_amd.capi import DeviceBuffer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
b = B // P
nu = 8
K = (synth.FX, synth.FY, synth.CX, synth.CY)
scenes = bench.make_cases(nu, 1000)
idx = [i % nu for i in range(b)]
cur_frames = np.stack([scenes[i]["cur"] for i in idx])
ref_frames = np.stack([scenes[i]["ref"] for i in idx])
d_cur = DeviceBuffer(cur_frames.nbytes)
d_cur.upload(cur_frames)
PN = bench.PNP
pipes = []
for p in range(P):
    cur = sdslam_amd.ORBextractor(*bench.CFG, bench.W, bench.H, b)
    ref = sdslam_amd.ORBextractor(*bench.CFG, bench.W, bench.H, b)
    trk = sdslam_amd.Tracker(cur, ref, max_points=1000, max_batch=b, pnp_max_iterations=PN["max_iterations"])
    trk.set_camera(*K, 0.0, bench.BOUNDS)
    rk, rd, rn = ref.extract_batch(ref_frames)
    lasts = [synth.tracking_case(i, rk[i, :rn[i]], rd[i, :rn[i]]) for i in range(nu)]
    trk.set_last(0, [lasts[i] for i in idx])
    pert = synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04))
    trk.set_poses(0, [scenes[i]["T_ref"] for i in idx], [pert @ scenes[i]["T_cur"] for i in idx])
    trk.set_rand(0, np.tile(synth.glibc_rand_stream(4 * PN["max_iterations"]), (b, 1)))
    pipes.append((cur, ref, trk))


def step():
    for cur, ref, trk in pipes:
        cur.extract_batch_device(d_cur.ptr, b, bench.W, bench.H)
        trk.align(b, 0)
        trk.match(b, 8.0, True, True)
        trk.pnp(b, PN["probability"], PN["min_inliers"], PN["max_iterations"], PN["min_set"], PN["epsilon"], PN["th2"], PN["max_iterations"])


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
pn = pipes[-1][2].get_pnp(0, b)
print(f"B={B} pipes={P}: {B * steps / dt:.0f} frames/s, {dt / steps * 1e3:.2f} ms/step, pnp_ok {int(pn['ok'].sum())}/{b}")
