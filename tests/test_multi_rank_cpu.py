"""N > 1 path on CPU: 2 gloo ranks shard a frame batch, each processes its shard (with the CPU
oracle standing in for the GPU stage -- this test is about the shard / gather / max-time logic,
which is backend-agnostic), and the gathered records must equal the single-process result."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from sdslam_amd import dist_util
from sdslam_amd.synth import make_image
from oracle import oracle as O
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
N = 5                                   # ragged on purpose: shards of 3 and 2 frames
lo, hi = dist_util.shard_range(N, rank, world)
ora = O.OrbOracle(300, 1.2, 8, 20)
recs = []
for f in range(lo, hi):
    k, d = ora.extract(make_image(50 + f, 320, 240))
    pose = np.eye(4).T.ravel() * (f + 1)
    recs.append((pose, 1, len(k), int(d.sum()) % 1000, f % 2))
rec = dist_util.pack_records([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs], [r[3] for r in recs], [r[4] for r in recs])
allrec = dist_util.gather_records(rec, N, dist)
tmax = dist_util.max_over_ranks(1.0 + rank, dist)
# the per-step collective bench.py issues inside its timed region (tensor to tensor; device tensors over RCCL there)
B = 4
mine = torch.arange(B * dist_util.RECORD_F64, dtype=torch.float64).reshape(B, -1) + 1000.0 * rank
both = torch.zeros((B * world, dist_util.RECORD_F64), dtype=torch.float64)
for step in range(3):        # reused buffers, like the double-buffered records of the bench
    dist_util.all_gather_records(mine + step, both, dist)
    for r in range(world):
        exp = torch.arange(B * dist_util.RECORD_F64, dtype=torch.float64).reshape(B, -1) + 1000.0 * r + step
        assert torch.equal(both[r * B:(r + 1) * B], exp), (rank, r, step)
if rank == 0:
    np.save(sys.argv[2], allrec)
    assert tmax == float(world), tmax
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_gloo_shard_and_gather(tmp_path, oracle):
    from sdslam_amd import dist_util
    from sdslam_amd.synth import make_image
    assert [dist_util.shard_range(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    assert [dist_util.shard_range(8192, r, 8)[1] - dist_util.shard_range(8192, r, 8)[0] for r in range(8)] == [1024] * 8
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "rec.npy"
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, str(out)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    ora = oracle.OrbOracle(300, 1.2, 8, 20)
    exp = []
    for f in range(5):
        k, d = ora.extract(make_image(50 + f, 320, 240))
        exp.append(np.concatenate([np.eye(4).T.ravel() * (f + 1), [1, len(k), int(d.sum()) % 1000, f % 2]]))
    assert np.array_equal(got, np.stack(exp))
