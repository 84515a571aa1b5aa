import os, sys, numpy as np
sys.path.insert(0, '.')
import sdslam_amd
from sdslam_amd import synth
g = np.load('tests/golden/track_seed20.npz')
K = (synth.FX, synth.FY, synth.CX, synth.CY)
s = synth.make_scene(20)
cfg = (1000, 1.2, 8, 20)
cur, ref = sdslam_amd.ORBextractor(*cfg, 640, 480, 1), sdslam_amd.ORBextractor(*cfg, 640, 480, 1)
cur.extract_batch(s["cur"][None])
rk, rd, rn = ref.extract_batch(s["ref"][None])
trk = sdslam_amd.Tracker(cur, ref, 1000, 1, 300)
trk.set_camera(*K, 0.0, (0., 640., 0., 480.))
trk.set_last(0, [synth.tracking_case(20, rk[0, :rn[0]], rd[0, :rn[0]])])
trk.set_poses(0, [s["T_ref"]], [g["T0"]])
trk.align(1, 0); trk.match(1, 8.0, True, True)
trk.set_rand(0, synth.glibc_rand_stream(800)[None])
for rep in range(2):
    trk.pnp(1, 0.99, 10, 200, 4, 0.28, 5.991, 200)
    pn = trk.get_pnp(0, 1)
    a = pn["inliers"][0, :len(g["pnp_inliers"])]; b = g["pnp_inliers"]
    print(rep, pn["iterations"], pn["n_inliers"], g["pnp_iterations"], g["pnp_n_inliers"], a.sum(), b.sum())
    print(" gpu-only idx", np.nonzero(a & ~b)[0][:20], " gold-only", np.nonzero(b & ~a)[0][:20])
    print(" dT", np.abs(pn["T"][0] - g["pnp_T"]).max())

import ctypes as C
L = sdslam_amd.lib()
L.sd_track_debug_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
idx = np.zeros(1000, np.uint16)
L.sd_track_debug_read(trk.h, 1, 0, idx.ctypes.data_as(C.c_void_p), idx.nbytes)
cm, nm = trk.get_matches(0, 1)
exp = np.nonzero(cm[0] >= 0)[0]
print("N", len(exp), "g_idx[:80]", idx[:80].tolist())
print("exp   [:80]", exp[:80].tolist())
print("equal", np.array_equal(idx[:len(exp)], exp))
