#!/usr/bin/env python3
"""Phase cycle counters of k_fast_cells on the bench frames (library built with SD_EXTRA_FLAGS=-DSD_PNP_PROF, at most 256
frames): cycles of thread 0 per phase, summed over a frame's cells.  (The selection kernels carried timers until they were
split into k_select_quota / _cells / _bigcells / _final; a rocprofv3 kernel trace times those.)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import sdslam_amd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scenes = bench.make_cases(8, 1000)
frames = np.stack([scenes[i % 8]["cur"] for i in range(B)])
cur = sdslam_amd.ORBextractor(*bench.CFG, bench.W, bench.H, B)
L = sdslam_amd.capi.lib()
out = (C.c_ulonglong * 64)()
cur.extract_batch(frames)
assert L.sd_debug_sel_prof(out, 1) == 0, "library built without -DSD_PNP_PROF"
reps = 3
for _ in range(reps):
    cur.extract_batch(frames)
assert L.sd_debug_sel_prof(out, 1) == 0
v = (np.array(list(out), np.float64) / (B * reps)).reshape(8, 8)
fn = ["stage+clear", "A compass", "B1 ring test", "B2 scores", "(barrier)", "C nms+emit", "(barrier)"]
print("k_fast_cells, cycles of thread 0 per frame (sum over the 148 cells):")
v[:, 7] *= reps   # the FAST records are per workgroup and overwritten by every call: they hold ONE call, not `reps`
for i, n in enumerate(fn):
    print(f"  {n:16s} {v[i, 7]:12.0f}")
print(f"  {'(own tile words loaded + stored, first part of stage)':16s} {v[7, 7]:12.0f}")
print(f"  {'total':16s} {v[:8, 7].sum():12.0f}")
