"""ORBmatcher::SearchByPoints (reference src/ORBmatcher.cc:1209-1301): brute-force Hamming between two keyframes' map
points.  CPU: the oracle against an independent numpy statement of the loop.  GPU: k_search_points vs the oracle on scene
pairs, on near-duplicate (tiled) images where the greedy vbMatched2 exclusion matters, with the order-dependent fallback
forced (option "track.bf_list_k"), and the direct read-out of the device grid (Frame::GetFeaturesInArea) vs oracle.features_in_area."""
import os

import numpy as np
import pytest

from sdslam_amd import synth

CFG = (1000, 1.2, 8, 20)
BOUNDS = (0.0, 640.0, 0.0, 480.0)
K = (synth.FX, synth.FY, synth.CX, synth.CY)


def numpy_search_by_points(ang1, d1, h1, ang2, d2, h2, nnratio, check_ori):
    """Independent restatement: full distance matrix, then the greedy loop."""
    D = np.unpackbits(d1[:, None, :] ^ d2[None, :, :], axis=2).sum(axis=2).astype(np.int64)
    N1, N2 = len(d1), len(d2)
    m = np.full(N1, -1, np.int64)
    taken = np.zeros(N2, bool)
    bins = {}
    for i in range(N1):
        if not h1[i]:
            continue
        ok = np.flatnonzero((h2 != 0) & ~taken)
        if len(ok) == 0:
            continue
        row = D[i, ok]
        o = np.argsort(row, kind="stable")
        b1, j = int(row[o[0]]), int(ok[o[0]])
        b2 = int(row[o[1]]) if len(o) > 1 else 256
        if b1 < 50 and np.float32(b1) < np.float32(nnratio) * np.float32(b2):
            m[i] = j
            taken[j] = True
            rot = np.float32(ang1[i]) - np.float32(ang2[j])
            if rot < 0:
                rot = np.float32(rot + np.float32(360.0))
            b = int(np.round(np.float32(rot * np.float32(1.0 / 30))))      # no exact .5 products occur for these angles
            bins.setdefault(0 if b == 30 else b, []).append(i)
    if check_ori and bins:
        cnt = np.zeros(30, np.int64)
        for b, v in bins.items():
            cnt[b] = len(v)
        order = sorted(range(30), key=lambda b: (-cnt[b], b))
        keep = [order[0]]
        if cnt[order[1]] >= 0.1 * cnt[order[0]] and cnt[order[1]] > 0:
            keep.append(order[1])
            if cnt[order[2]] >= 0.1 * cnt[order[0]] and cnt[order[2]] > 0:
                keep.append(order[2])
        for b, v in bins.items():
            if b not in keep:
                m[v] = -1
    return int((m >= 0).sum()), m


def test_oracle_search_by_points_vs_numpy(oracle):
    rng = np.random.default_rng(5)
    for trial in range(6):
        N1, N2 = int(rng.integers(40, 90)), int(rng.integers(40, 90))
        base = rng.integers(0, 256, size=(12, 32)).astype(np.uint8)

        def noisy(n):
            d = base[rng.integers(0, 12, n)].copy()
            for _ in range(3):
                d[np.arange(n), rng.integers(0, 32, n)] ^= (1 << rng.integers(0, 8, n)).astype(np.uint8)
            return d
        d1, d2 = noisy(N1), noisy(N2)
        k1, k2 = np.zeros(N1, oracle.KP_DTYPE), np.zeros(N2, oracle.KP_DTYPE)
        k1["angle"], k2["angle"] = rng.uniform(0, 360, N1).astype(np.float32), rng.uniform(0, 360, N2).astype(np.float32)
        if trial % 2:
            k2["angle"][:] = 0
            k1["angle"][:] = rng.choice([10.0, 100.0, 200.0], N1).astype(np.float32)
        h1, h2 = (rng.random(N1) > 0.2).astype(np.uint8), (rng.random(N2) > 0.2).astype(np.uint8)
        for nnratio in (0.75, 1.6):
            for ori in (True, False):
                n, m = oracle.search_by_points(k1, d1, h1, k2, d2, h2, nnratio, ori)
                n_ref, m_ref = numpy_search_by_points(k1["angle"], d1, h1, k2["angle"], d2, h2, nnratio, ori)
                assert n == n_ref and np.array_equal(m, m_ref), (trial, nnratio, ori)
                assert n == (m >= 0).sum() and len(set(m[m >= 0])) == n          # a pKF point is given away once
    # no map points on one side -> nothing
    n, m = oracle.search_by_points(k1, d1, np.zeros(N1, np.uint8), k2, d2, h2)
    assert n == 0 and (m == -1).all()


# ------------------------------------------------------------------------------------------------ GPU
def _tiled(seed, noise_seed):
    tile = synth.make_image(seed, 80, 80)
    img = np.tile(tile, (6, 8)).astype(np.int32)
    rng = np.random.default_rng(noise_seed)
    return np.clip(img + rng.integers(-6, 7, size=img.shape), 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def bf_rig(oracle):
    import sdslam_amd
    if sdslam_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests need a real MI355X")
    s0, s1 = synth.make_scene(20), synth.make_scene(23, (0.01, 0.03, 0.02), (0.2, 0.5, -0.8))
    pairs = [(s0["cur"], s0["ref"]), (s1["cur"], s1["ref"]), (_tiled(3, 1), _tiled(3, 2)), (s0["cur"], s0["cur"])]
    B = len(pairs)
    cur, ref = sdslam_amd.ORBextractor(*CFG, 640, 480, B), sdslam_amd.ORBextractor(*CFG, 640, 480, B)
    k1, d1, n1 = cur.extract_batch(np.stack([p[0] for p in pairs]))
    k2, d2, n2 = ref.extract_batch(np.stack([p[1] for p in pairs]))
    trk = sdslam_amd.Tracker(cur, ref, max_points=1000, max_batch=B, pnp_max_iterations=8)
    trk.set_camera(*K, 0.0, BOUNDS)
    return dict(B=B, trk=trk, cur=cur, k1=k1, d1=d1, n1=n1, k2=k2, d2=d2, n2=n2)


@pytest.mark.gpu
@pytest.mark.parametrize("klist", [None, 1, 2])
def test_hip_search_by_points_matches_oracle(oracle, bf_rig, klist):
    r, B, trk = bf_rig, bf_rig["B"], bf_rig["trk"]
    cap = r["k1"].shape[1]
    rng = np.random.default_rng(9)
    import sdslam_amd
    sdslam_amd.set_option("track.bf_list_k", klist or 4)
    try:
        for flags in ("all", "random"):
            h1 = np.ones((B, cap), np.uint8) if flags == "all" else (rng.random((B, cap)) > 0.3).astype(np.uint8)
            h2 = np.ones((B, cap), np.uint8) if flags == "all" else (rng.random((B, cap)) > 0.3).astype(np.uint8)
            trk.set_point_flags(0, h1, h2)
            for nnratio in (0.75, 1.5):
                for ori in (True, False):
                    trk.search_by_points(B, nnratio, ori)
                    m, nm = trk.get_point_matches(0, B)
                    tot = 0
                    for b in range(B):
                        n1, n2 = r["n1"][b], r["n2"][b]
                        n, om = oracle.search_by_points(r["k1"][b, :n1], r["d1"][b, :n1], h1[b, :n1], r["k2"][b, :n2], r["d2"][b, :n2],
                                                        h2[b, :n2], nnratio, ori)
                        assert nm[b] == n, (flags, nnratio, ori, b, nm[b], n)
                        assert np.array_equal(m[b, :n1], om) and (m[b, n1:] == -1).all(), (flags, nnratio, ori, b)
                        tot += n
                    assert tot > 100      # the comparison is not vacuous
    finally:
        sdslam_amd.set_option("track.bf_list_k", 4)


@pytest.mark.gpu
def test_device_grid_features_in_area(oracle, bf_rig):
    """a13 / a14 directly: Frame::AssignFeaturesToGrid occupancy and Frame::GetFeaturesInArea's index list, in order."""
    r, trk = bf_rig, bf_rig["trk"]
    rng = np.random.default_rng(3)
    for b in (0, 2):
        n = r["n1"][b]
        kps = r["k1"][b, :n]
        # mGrid occupancy: PosInGrid with round() (src/Frame.cc:323-332)
        # C round() = half away from zero (numpy rounds half to even; level-0 keypoints sit on exact .5 products)
        posx = np.floor((kps["x"] - np.float32(0)) * np.float32(64.0 / 640.0) + np.float32(0.5)).astype(int)
        posy = np.floor((kps["y"] - np.float32(0)) * np.float32(48.0 / 480.0) + np.float32(0.5)).astype(int)
        ok = (posx >= 0) & (posx < 64) & (posy >= 0) & (posy < 48)
        want = np.zeros((64, 48), np.int32)
        np.add.at(want, (posx[ok], posy[ok]), 1)
        idx, grid = trk.features_in_area(b, 320.0, 240.0, 30.0, want_grid=True)
        assert np.array_equal(grid, want)
        queries = [(320.0, 240.0, 30.0, -1, -1), (5.0, 5.0, 40.0, -1, -1), (639.0, 479.0, 25.0, 0, 2), (100.5, 300.25, 8.0, 1, -1),
                   (-50.0, 200.0, 20.0, -1, -1), (320.0, 240.0, 400.0, -1, -1), (700.0, 100.0, 70.0, -1, 3)]
        queries += [(float(rng.uniform(0, 640)), float(rng.uniform(0, 480)), float(rng.uniform(2, 90)), int(rng.integers(-1, 4)),
                     int(rng.integers(-1, 8))) for _ in range(40)]
        # queries that MUST hit (the non-vacuity bound is not bent to the data any more): centred near a
        # keypoint, radius beyond the offset, level window around the keypoint's own octave (or open)
        sure = []
        for i in rng.choice(n, 40, replace=False):
            o = int(kps["octave"][i])
            lo, hi = [(-1, -1), (o, o), (max(o - 1, 0), o + 1), (0, -1)][int(rng.integers(0, 4))]
            sure.append((float(kps["x"][i]) + float(rng.uniform(-1.5, 1.5)), float(kps["y"][i]) + float(rng.uniform(-1.5, 1.5)),
                         float(rng.uniform(3, 40)), lo, hi))
        nonempty = 0
        for q, (x, y, rad, lo, hi) in enumerate(sure + queries):
            got = trk.features_in_area(b, x, y, rad, lo, hi)
            exp = oracle.features_in_area(kps, BOUNDS, x, y, rad, lo, hi)
            assert np.array_equal(got, exp), (b, x, y, rad, lo, hi, len(got), len(exp))
            assert q >= len(sure) or len(exp) > 0, (b, x, y, rad, lo, hi)
            nonempty += len(exp) > 0
        assert nonempty >= len(sure)
