"""Multi-GPU plumbing for the batched-frames mode: frames are independent, so
ranks own contiguous shards and exchange nothing on the data path; one all-gather collects the
fixed-size per-frame result records.  Backend-agnostic (nccl == RCCL on the GPUs, gloo in CPU tests)."""
import numpy as np

RECORD_F64 = 20   # 16 pose entries (column-major) + align_ok, n_matches, pnp_inliers, pnp_ok


def shard_range(n_frames: int, rank: int, world: int):
    """Contiguous [lo, hi) of the frames owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_records(poses_cm, align_ok, n_matches, pnp_inliers, pnp_ok) -> np.ndarray:
    n = len(poses_cm)
    rec = np.zeros((n, RECORD_F64), np.float64)
    rec[:, :16] = np.asarray(poses_cm, np.float64).reshape(n, 16)
    rec[:, 16], rec[:, 17], rec[:, 18], rec[:, 19] = align_ok, n_matches, pnp_inliers, pnp_ok
    return rec


def gather_records(rec: np.ndarray, n_total: int, dist=None, device=None) -> np.ndarray:
    """All-gather per-rank record blocks (ragged shards are padded to the largest shard)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rec
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    cap = max(shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world))
    buf = np.zeros((cap, RECORD_F64), np.float64)
    buf[:len(rec)] = rec
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    parts = []
    for r in range(world):
        lo, hi = shard_range(n_total, r, world)
        parts.append(out[r].cpu().numpy()[:hi - lo])
    return np.concatenate(parts, 0)


def all_gather_records(rec_t, out_t, dist):
    """The per-step collective of the batched-frames mode: every rank's [B, RECORD_F64] f64 block -> [world * B, RECORD_F64],
    rank-major, tensor to tensor (device memory with nccl/RCCL; the same call runs on CPU tensors with gloo).  Equal shard
    sizes (weak scaling: B frames per GPU); ragged shards go through gather_records."""
    assert out_t.shape[0] == rec_t.shape[0] * dist.get_world_size() and out_t.shape[1:] == rec_t.shape[1:]
    dist.all_gather_into_tensor(out_t, rec_t)
    return out_t


def max_over_ranks(value: float, dist=None, device=None) -> float:
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    import torch
    t = torch.tensor([value], dtype=torch.float64)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
