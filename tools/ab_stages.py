import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bench
scenes = bench.make_cases(8, 1000)
wl = bench.Workload(scenes, 1024, 0, "pnp")
import torch
trk = wl.trk
wl.step(); trk.get_tracked(0, 1)
def t(fn, n=20):
    fn(); trk.get_tracked(0,1)
    t0=time.perf_counter()
    for _ in range(n): fn()
    trk.get_tracked(0,1)
    return (time.perf_counter()-t0)/n*1e3
print("align alone ms", t(lambda: trk.align(1024, 0)))
print("match alone ms", t(lambda: trk.match(1024, 8.0, True, True)))
print("pnp alone ms", t(lambda: trk.pnp(1024, 0.99, 10, 200, 4, 0.28, 5.991, 200)))
print("step ms", t(wl.step, 40))
al = trk.get_align(0, 1024)
print("gn iters", al["iters"][:, :8].sum(axis=1).mean(), "ok", al["ok"].sum())
ones = np.ones((1024, wl.cur.cap), np.uint8)
trk.set_point_flags(0, ones, ones)
print("search_by_points (1000 x 1000 brute force) alone ms", t(lambda: trk.search_by_points(1024, 0.75, True)))
m12, n12 = trk.get_point_matches(0, 1024)
print("mean point matches", n12.mean())
