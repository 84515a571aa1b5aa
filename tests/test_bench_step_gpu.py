"""Parity AT THE BENCHMARKED SHAPE: the exact Workload.step() bench.py times -- 1024 frames per launch, the low-priority
tracking stream, the double-buffered extractor outputs, the large-batch kernel variants (k_align<4>, 1-wave k_pose_opt) --
run twice back to back without host synchronisation, every one of the 1024 slots compared with the oracle's result for
its scene: keypoints / descriptors / matches bit-exact, ImageAlign / PnP / PoseOptimization poses <= 1e-5 with identical
iteration counts, inlier masks and outlier flags.  Plus BASELINE configs[4]'s frame size (1280x720) through the whole
tracking step."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
POSE_TOL = 1e-5
NU = 64                      # bench.py's --unique default: every benchmarked scene is compared, not a sample of them
PYRAMIDS = ["8x1.2", "5x2.0"]   # BASELINE's pyramid and the reference's own default (src/Config.cc:48-51)


@pytest.fixture(scope="module")
def pool():
    import multiprocessing as mp
    import os
    n = max(1, min(16, len(os.sched_getaffinity(0))))
    with mp.get_context("spawn").Pool(n) as p:      # spawn: fresh CPU-only interpreters, whatever this process did with the GPU
        yield p


@pytest.fixture(scope="module")
def scenes(pool):
    import bench
    return bench.make_cases(NU, 1000, pool=pool)


@pytest.fixture(scope="module")
def oracle_steps(scenes, pool):
    """(pyramid) -> the oracle's results for the 64 scenes, computed once per pyramid in the pool."""
    import bench
    import bench_oracle
    from sdslam_amd import synth
    cache = {}

    def get(pyr, sc=None, bounds=None):
        key = pyr if sc is None else None
        if key is not None and key in cache:
            return cache[key]
        cfg = bench.parse_pyramid(pyr)
        K = (synth.FX, synth.FY, synth.CX, synth.CY)
        rs = synth.glibc_rand_stream(4 * bench.PNP["max_iterations"])
        jobs = [(s, cfg, K, bounds or (0.0, 640.0, 0.0, 480.0), rs, bench.PNP) for s in (sc or scenes)]
        out = pool.map(bench_oracle.oracle_step, jobs)
        if key is not None:
            cache[key] = out
        return out
    return get


def _compare(wl, ora, solver, rec_host):
    B = wl.B
    al, cm, nm, pn = wl.results()
    kps, desc, n = wl.cur.download(0, B)
    for b in range(B):
        o = ora[wl.idx[b]]
        nk = len(o["ck"])
        assert n[b] == nk and np.array_equal(kps[b, :nk], o["ck"]) and np.array_equal(desc[b, :nk], o["cd"]), b
        if solver in ("pnp", "poseopt"):
            r = o["al"]
            assert al["ok"][b] == r["ok"] and np.array_equal(al["iters"][b][:len(r["iters"])], r["iters"]), (b, al["iters"][b][:8], r["iters"])
            assert np.abs(al["T"][b] - r["T"]).max() <= POSE_TOL, (b, np.abs(al["T"][b] - r["T"]).max())
            assert nm[b] == o["nm"] and np.array_equal(cm[b, :nk], o["cm"]), b
            assert np.abs(rec_host[b, :16].reshape(4, 4).T - (pn["T"][b] if solver == "pnp" else wl.po["T"][b])).max() == 0
        if solver == "pnp":
            r, pr = o["pnp"], o["pnp_params"]
            assert (pn["N"][b], pn["min_inliers"][b], pn["max_its"][b]) == (pr["N"], pr["min_inliers"], pr["max_its"])
            assert (bool(pn["ok"][b]), int(pn["iterations"][b]), int(pn["n_inliers"][b]), bool(pn["no_more"][b])) == \
                (r["ok"], r["iterations"], r["n_inliers"], r["no_more"]), b
            assert np.array_equal(pn["inliers"][b, :nk], r["inliers"]) and np.abs(pn["T"][b] - r["T"]).max() <= POSE_TOL, b
            assert rec_host[b, 16:].tolist() == [float(al["ok"][b]), float(nm[b]), float(r["n_inliers"]), float(r["ok"])]
        elif solver == "poseopt":
            r = o["po"]
            g = wl.po
            assert g["n_inliers"][b] == r["n_inliers"] and np.array_equal(g["outlier"][b, :nk], r["outlier"]), b
            assert np.abs(g["T"][b] - r["T"]).max() <= POSE_TOL, (b, np.abs(g["T"][b] - r["T"]).max())
        else:
            r = o["tw"]
            tw, g = wl.tw, wl.po
            assert (tw["status"][b], tw["nmatches"][b], tw["nmatches_map"][b], tw["retried"][b]) == \
                (r["status"], r["nmatches"], r["nmatches_map"], r["retried"]), b
            assert np.array_equal(cm[b, :nk], r["match"]) and np.abs(g["T"][b] - r["T"]).max() <= POSE_TOL, b
            assert rec_host[b, 18] == r["nmatches_map"] and rec_host[b, 19] == float(r["status"] == 2)


@pytest.mark.parametrize("pyramid", PYRAMIDS)
@pytest.mark.parametrize("solver", ["pnp", "poseopt", "motion_model"])
def test_bench_step_1024_frames_every_slot(scenes, oracle_steps, solver, pyramid, monkeypatch):
    _bench_step_1024(scenes, oracle_steps, solver, pyramid, monkeypatch)


@pytest.mark.parametrize("opts", [{"extract.fast0_early": 0}, {"extract.pyr_early": 1}, {"track.match_split": 0}],
                         ids=["fast0_late", "pyr_early", "match_single"])
def test_bench_step_1024_frames_scheduling_options(scenes, oracle_steps, opts, monkeypatch):
    """The same 1024 slots under the non-default scheduling / kernel-form options: results never depend on an option."""
    import sdslam_amd
    with sdslam_amd.options(opts):
        _bench_step_1024(scenes, oracle_steps, "pnp", "8x1.2", monkeypatch)


def _bench_step_1024(scenes, oracle_steps, solver, pyramid, monkeypatch):
    import bench
    from sdslam_amd.capi import DeviceBuffer, lib, _p
    B = 1024
    ora = oracle_steps(pyramid)         # (first: CPU pool work before this test's GPU work)
    monkeypatch.setattr(bench, "CFG", bench.parse_pyramid(pyramid))
    wl = bench.Workload(scenes, B, 0, solver)
    assert wl.nu == NU
    for i in range(NU):                 # the map points bench.py seeds from the GPU's ref keypoints are the oracle's
        assert all(np.array_equal(wl.lasts_u[i][k], v) for k, v in wl.synth.tracking_case(0, ora[i]["rk"], ora[i]["rd"]).items()), i
    rec = [DeviceBuffer(B * 160), DeviceBuffer(B * 160)]
    wl.attach_records(rec[0].ptr.value, rec[1].ptr.value)
    wl.step()
    wl.step()                           # back to back: extraction of step 2 overlaps tracking of step 1
    wl.po = wl.trk.get_pose_opt(0, B)   # (synchronises)
    wl.tw = wl.trk.get_tracked(0, B)
    rec_host = np.zeros((B, 20))
    lib().sd_dev_download(_p(rec_host), rec[1].ptr, rec_host.nbytes)     # step 2 wrote buffer 1
    _compare(wl, ora, solver, rec_host)
    if solver == "pnp":   # the default scenes are the easy case; say so where the numbers are checked
        assert max(o["pnp"]["iterations"] for o in ora) <= 40
    assert all(o["al"]["ok"] for o in ora) and min(o["nm"] for o in ora) >= 100 and all(o["tw"]["status"] == 2 for o in ora)
    wl.trk.close()
    wl.cur.close()
    wl.ref.close()


def test_tracking_step_1280x720(oracle_steps):
    """BASELINE configs[4]'s frame size through extract -> align -> match -> PnP and -> PoseOptimization, two scene pairs."""
    import bench
    from sdslam_amd.capi import DeviceBuffer, lib, _p
    w, h = 1280, 720
    sc = bench.make_cases(2, 4000, w, h)
    bounds = (0.0, float(w), 0.0, float(h))
    for solver in ("pnp", "poseopt"):
        wl = bench.Workload(sc, 2, 0, solver, w=w, h=h)
        rec = [DeviceBuffer(2 * 160), DeviceBuffer(2 * 160)]
        wl.attach_records(rec[0].ptr.value, rec[1].ptr.value)
        wl.step()
        wl.po = wl.trk.get_pose_opt(0, 2)
        wl.tw = wl.trk.get_tracked(0, 2)
        rec_host = np.zeros((2, 20))
        lib().sd_dev_download(_p(rec_host), rec[0].ptr, rec_host.nbytes)
        ora = oracle_steps("8x1.2", sc, bounds)
        _compare(wl, ora, solver, rec_host)
        assert all(o["al"]["ok"] for o in ora) and min(o["nm"] for o in ora) >= 100


@pytest.mark.parametrize("shape", ["vga_b64", "configs4_shard"])
def test_bench_two_ranks_rehearsal(shape):
    """bench.py's N > 1 control flow on this one-GPU box: `--gpus 2` spawns two ranks (both on GPU 0, gloo over host memory:
    SD_BENCH_REHEARSAL), every step packs the records on the device and gathers them inside the timed region, rank 0 checks
    its own block of the gathered tensor against its records and prints one line.  (The RCCL leg itself -- device tensors
    straight into all_gather_into_tensor -- needs two GPUs; the same call is exercised on CPU tensors in
    tests/test_multi_rank_cpu.py.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SD_BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    # configs4_shard: the exact per-rank workload of BASELINE configs[4] on 8 GPUs (8192 x 1280x720 frames / 8 = 1024 per rank)
    B, extra = (64, []) if shape == "vga_b64" else (1024, ["--res", "1280x720"])
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", str(B),
                        "--unique", "4", "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["frames_per_gpu_per_step"] == B and "REHEARSAL" in line["config"]["pose_records"]
    assert line["tracking"]["pnp_ok"] == B and line["value"] > 0
