#!/usr/bin/env python3
"""Stage cycle counters of k_pnp on the bench workload (needs a library built with
SD_EXTRA_FLAGS=-DSD_PNP_PROF python -m sdslam_amd.build --force).  Prints shader-clock cycles
of lane 0 per stage, averaged per frame."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import sdslam_amd  # noqa: E402
from sdslam_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nu = 8
K = (synth.FX, synth.FY, synth.CX, synth.CY)
scenes = bench.make_cases(nu, 1000)
idx = [i % nu for i in range(B)]
cur = sdslam_amd.ORBextractor(*bench.CFG, bench.W, bench.H, B)
ref = sdslam_amd.ORBextractor(*bench.CFG, bench.W, bench.H, B)
trk = sdslam_amd.Tracker(cur, ref, max_points=1000, max_batch=B, pnp_max_iterations=200)
trk.set_camera(*K, 0.0, bench.BOUNDS)
rk, rd, rn = ref.extract_batch(np.stack([scenes[i]["ref"] for i in idx]))
lasts = [synth.tracking_case(i, rk[i, :rn[i]], rd[i, :rn[i]]) for i in range(nu)]
trk.set_last(0, [lasts[i] for i in idx])
pert = synth.se3_exp((0.003, -0.002, 0.001), (0.05, 0.02, -0.04))
trk.set_poses(0, [scenes[i]["T_ref"] for i in idx], [pert @ scenes[i]["T_cur"] for i in idx])
trk.set_rand(0, np.tile(synth.glibc_rand_stream(800), (B, 1)))
cur.extract_batch(np.stack([scenes[i]["cur"] for i in idx]))
trk.align(B, 0)
trk.match(B, 8.0, True, True)
P = bench.PNP
L = sdslam_amd.capi.lib()
out = (C.c_ulonglong * 32)()


def run():
    trk.pnp(B, P["probability"], P["min_inliers"], P["max_iterations"], P["min_set"], P["epsilon"], P["th2"], P["max_iterations"])


aout = (C.c_ulonglong * 16)()
L.sd_debug_align_prof(aout, 1)
for _ in range(3):
    trk.align(B, 0)
L.sd_debug_align_prof(aout, 1)
av = np.array(list(aout), np.float64) / (B * 3)
an = ["gather+init", "precompute patches (per level)", "-", "residuals (project, loads, 16 px)", "H/Jres wave reductions + barrier",
      "float chi2 chain (wave 0) || LDLT+exp (wave 1) + barrier", "decisions + next pose (tid 0)", "barrier"]
print("k_align phases (cycles of thread 0 per frame)")
for i, nme in enumerate(an):
    print(f"  {nme:30s} {av[i]:12.0f}")
print(f"  {'total':30s} {av[:8].sum():12.0f}   iterations {trk.get_align(0, B)['iters'][:, :8].sum(axis=1).mean():.2f}")
trk.match(B, 8.0, True, True)

run()
assert L.sd_debug_pnp_prof(out, 1) == 0, "library built without -DSD_PNP_PROF"
reps = 3
for _ in range(reps):
    run()
assert L.sd_debug_pnp_prof(out, 1) == 0
v = np.array(list(out), np.float64) / (B * reps)
names = {0: "choose_ctrl", 1: "barycentric", 2: "MtM", 3: "svd12", 4: "L6x10/rho", 5: "find_betas(svd6)", 6: "gauss_newton",
         7: "R_t(pcs,abt,svd3)", 8: "reproj_err", 10: "gather+params", 11: "loop-top", 12: "minimal-set EPnP (16 lanes)",
         13: "inlier masks x16", 14: "accept+refine"}
for base, title in ((0, "EPnP minimal sets (lane 0 of 16)"), (16, "EPnP refit (lead lane)")):
    print(title)
    for i in range(9):
        print(f"  {names[i]:24s} {v[base + i]:12.0f}")
    print(f"  {'total':24s} {v[base:base + 9].sum():12.0f}")
print("k_pnp phases")
for i in range(10, 15):
    print(f"  {names[i]:24s} {v[i]:12.0f}")
print(f"  {'total':24s} {v[10:15].sum():12.0f}")
pn = trk.get_pnp(0, B)
print("mean iterations", pn["iterations"].mean(), "ok", pn["ok"].sum())
