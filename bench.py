#!/usr/bin/env python3
"""bench.py -- tracked frames/s of the per-frame tracking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--batch B]

A "step" is one pass of the hot path over one batch of B synthetic 640x480 frames that are
already resident in HBM (the batched-frames mode of SURVEY.md §8e).  With N > 1 the driver
launches one rank per GPU (torch.distributed / RCCL); frames are independent, so ranks share
nothing on the data path (weak scaling) and only the fixed-size per-frame result records are
all-gathered.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the dominant kernel stage: algorithmic bytes per launch / its mean
                  duration measured with HIP events on the launch stream inside the timed region
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


def cpu_baseline(images, cfg, budget_s=12.0):
    """Oracle (-O3 -march=native build) on the host cores of this box: 1 thread, bounded sample."""
    from oracle import oracle as O
    ora = O.OrbOracle(*cfg, fast_build=True)
    ora.extract(images[0])   # warm
    t0 = time.perf_counter()
    n = 0
    while True:
        ora.extract(images[n % len(images)])
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 2000:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": f"{n} VGA frames, ORB extract (8 levels x1.2, 1000 kp), 1 thread, "
                      f"{os.cpu_count()} host cpus visible"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--unique", type=int, default=32, help="distinct synthetic frames (tiled to --batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"

    import sdslam_amd
    from sdslam_amd.capi import DeviceBuffer
    from sdslam_amd.synth import make_image

    W, H, B = 640, 480, args.batch
    cfg = (1000, 1.2, 8, 20)          # BASELINE configs[1]: 8-level pyramid, 1000 kp
    uniq = [make_image(10_000 * rank + i, W, H) for i in range(min(args.unique, B))]
    frames = np.stack([uniq[i % len(uniq)] for i in range(B)])
    dbuf = DeviceBuffer(frames.nbytes)
    dbuf.upload(frames)

    ext = sdslam_amd.ORBextractor(*cfg, W, H, B, device=local_rank)

    def step():
        ext.extract_batch_device(dbuf.ptr, B, W, H)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ext.sync()
    ext.set_profiling(True)           # HIP events around each stage, on the launch stream
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    stage_ms = ext.stage_ms()         # mean per stage over the timed steps
    ext.set_profiling(False)

    # pose/result records of all ranks (fixed-size; the only inter-GPU traffic)
    kps, desc, n = ext.download(0, min(B, 4))
    if dist is not None:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rec = torch.from_numpy(n.astype(np.int32)).cuda()
        out = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(out, rec)

    if rank == 0:
        total_frames = B * args.steps * world
        names = ext.stage_names()
        sbytes = ext.stage_bytes()
        dom = int(np.argmax(stage_ms))
        launches = {"pyramid": cfg[2]}.get(names[dom], 1)
        achieved = sbytes[dom] * B / (stage_ms[dom] * 1e-3) / 1e9
        line = {
            "metric": "tracked frames/sec (ORB+ImageAlign+PnP) at 640x480, 1000 kp",
            "value": total_frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: 640x480, 8-level x1.2 pyramid, 1000 kp, ORB extract "
                                   "(stages so far: pyramid, FAST+NMS, select, blur, orientation+rBRIEF)",
                       "frames_per_gpu_per_step": B, "unique_frames": len(uniq), "inputs": "resident in HBM"},
            "stages_ms_per_step": {nm: float(ms) for nm, ms in zip(names, stage_ms)},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "algorithmic_bytes_per_frame": float(sbytes[dom]), "launches_per_step": launches,
                         "ms_per_step": float(stage_ms[dom])},
            "keypoints_first_frames": [int(x) for x in n],
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(uniq, cfg)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
