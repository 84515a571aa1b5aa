#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-buffer entry points: sd_orb_extract_batch with frames in
pageable host memory and results copied back, and the single-frame drop-in sd_orb_extract."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
import sdslam_amd  # noqa: E402

B = 1024
scenes = bench.make_cases(8, 1000)
frames = np.stack([scenes[i % 8]["cur"] for i in range(B)])
ext = sdslam_amd.ORBextractor(*bench.CFG, bench.W, bench.H, B)
ext.extract_batch(frames)
t0 = time.perf_counter()
for _ in range(5):
    k, d, n = ext.extract_batch(frames)
dt = (time.perf_counter() - t0) / 5
print(f"sd_orb_extract_batch, host frames in / keypoints+descriptors out: {B / dt:.0f} frames/s ({dt * 1e3:.1f} ms per {B} frames, "
      f"{frames.nbytes / dt / 1e9:.1f} GB/s of input)")
from sdslam_amd.capi import pinned_array  # noqa: E402
pf, owner = pinned_array(frames.shape)
pf[...] = frames
ext.extract_batch(pf)
t0 = time.perf_counter()
for _ in range(5):
    k, d, n = ext.extract_batch(pf)
dt = (time.perf_counter() - t0) / 5
print(f"same, frames in page-locked host memory (sd_host_alloc): {B / dt:.0f} frames/s ({dt * 1e3:.1f} ms per {B} frames, "
      f"{frames.nbytes / dt / 1e9:.1f} GB/s of input)")
one = sdslam_amd.ORBextractor(*bench.CFG, bench.W, bench.H, 1)
one(frames[0])
t0 = time.perf_counter()
for i in range(200):
    one(frames[i % 8])
dt = (time.perf_counter() - t0) / 200
print(f"sd_orb_extract (single-frame drop-in, host in / host out): {1 / dt:.0f} frames/s ({dt * 1e3:.3f} ms per frame)")
