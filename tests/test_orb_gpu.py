"""GPU parity: sd_orb_* (HIP, through the C ABI) vs the CPU oracle on the same seeded inputs.
Bar: bit-exact pyramid, FAST candidate counts, blurred levels, keypoints (identity AND order),
angles, responses, octaves and descriptors."""
import numpy as np
import pytest

from sdslam_amd.synth import make_image

pytestmark = pytest.mark.gpu

CONFIGS = {
    "P8": (1000, 1.2, 8, 20),   # BASELINE config: 8 levels x 1.2
    "P5": (1000, 2.0, 5, 20),   # reference default (src/Config.cc:48-51)
}


@pytest.fixture(scope="module")
def sd():
    import sdslam_amd
    if sdslam_amd.device_count() < 1:
        pytest.fail("no HIP device: the gpu-marked tests need a real MI355X")
    return sdslam_amd


def _compare_frame(oracle, ext, ora, img, frame, nlevels, stages=True):
    ok, od = ora.extract(img)
    if stages:
        for l in range(nlevels):
            assert np.array_equal(ext.level(l, frame, padded=True), ora.level(l, padded=True)), f"pyramid level {l}"
            assert np.array_equal(ext.cell_counts(l, frame), ora.cell_totals(l)), f"FAST counts level {l}"
            keys = ext.level_keys(l, frame)
            lk = ora.level_keypoints(l)
            exp = (lk["response"].astype(np.uint32) << 24) | (lk["y"].astype(np.uint32) << 12) | lk["x"].astype(np.uint32)
            assert np.array_equal(keys, exp), f"selected keys level {l}"
            ob = ora.blurred(l)
            if ob is not None:
                assert np.array_equal(ext.blurred(l, frame), ob), f"blur level {l}"
    return ok, od


@pytest.mark.parametrize("cfg", ["P8", "P5"])
def test_extract_single_frame_bit_exact(sd, oracle, cfg):
    nf, sf, nl, th = CONFIGS[cfg]
    ext = sd.ORBextractor(nf, sf, nl, th, 640, 480, 1)
    ora = oracle.OrbOracle(nf, sf, nl, th)
    for seed in range(3):
        img = make_image(seed)
        k, d = ext(img)
        ok, od = _compare_frame(oracle, ext, ora, img, 0, nl)
        assert len(k) == len(ok)
        for f in ("x", "y", "size", "angle", "response", "octave", "class_id"):
            assert np.array_equal(k[f], ok[f]), f"keypoint field {f} (seed {seed})"
        assert np.array_equal(d, od), f"descriptors (seed {seed})"
    ext.close()


def test_extract_batch_matches_oracle(sd, oracle):
    nf, sf, nl, th = CONFIGS["P8"]
    B = 6
    ext = sd.ORBextractor(nf, sf, nl, th, 640, 480, B)
    ora = oracle.OrbOracle(nf, sf, nl, th)
    imgs = np.stack([make_image(100 + i) for i in range(B)])
    kps, desc, n = ext.extract_batch(imgs)
    for i in range(B):
        ok, od = _compare_frame(oracle, ext, ora, imgs[i], i, nl, stages=(i in (0, B - 1)))
        assert n[i] == len(ok)
        assert np.array_equal(kps[i, :n[i]], ok)
        assert np.array_equal(desc[i, :n[i]], od)
    ext.close()


def test_edge_inputs(sd, oracle):
    nf, sf, nl, th = CONFIGS["P8"]
    ext = sd.ORBextractor(nf, sf, nl, th, 752, 480, 2)
    ora = oracle.OrbOracle(nf, sf, nl, th)
    # empty image -> no keypoints, no error (src/ORBextractor.cc:622-623)
    k, d = ext(np.zeros((0, 0), np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    # flat image: FAST finds nothing anywhere
    flat = np.full((480, 640), 77, np.uint8)
    k, d = ext(flat)
    ok, od = ora.extract(flat)
    assert len(k) == 0 and len(ok) == 0
    # few corners: quota redistribution paths with mostly-empty cells
    sparse = np.full((480, 640), 100, np.uint8)
    sparse[100:140, 100:160] = 220
    sparse[300:330, 400:470] = 10
    k, d = ext(sparse)
    ok, od = ora.extract(sparse)
    assert np.array_equal(k, ok) and np.array_equal(d, od)
    # a different geometry on the same handle (EuRoC-sized, odd width, non-contiguous stride)
    big = make_image(7, 800, 480)[:, :752]
    k, d = ext(big)
    ok, od = ora.extract(np.ascontiguousarray(big))
    assert np.array_equal(k, ok) and np.array_equal(d, od)
    # pure noise: the densest candidate lists (ties everywhere, stresses retainBest replay)
    rng = np.random.default_rng(3)
    noise = rng.integers(0, 256, size=(480, 640)).astype(np.uint8)
    k, d = ext(noise)
    ok, od = ora.extract(noise)
    assert np.array_equal(k, ok) and np.array_equal(d, od)
    ext.close()


@pytest.mark.parametrize("opts", [{"extract.select_small_cap": 48}, {"extract.select_small_cap": 48, "extract.select_big_cap": 160}])
def test_selection_paths(sd, oracle, opts):
    """The selection's rarely used paths give the same keys: cells larger than k_select_cells' buffer (trimmed by
    k_select_bigcells in its LDS buffer) and cells larger than that one too (serial replay in HBM).  Textured frame and pure
    noise (densest lists, ties everywhere)."""
    with sd.options(opts):
        ext = sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 1)
        ora = oracle.OrbOracle(1000, 1.2, 8, 20)
        rng = np.random.default_rng(5)
        for img in (make_image(61), rng.integers(0, 256, size=(480, 640)).astype(np.uint8)):
            k, d = ext(img)
            ok, od = _compare_frame(oracle, ext, ora, img, 0, 8)
            assert len(ok) > 500
            assert np.array_equal(k, ok) and np.array_equal(d, od)
        ext.close()


@pytest.mark.parametrize("opts", [{"extract.fast_merge_from": 3}, {"extract.fast_merge_from": 5}, {"extract.fast_merge_from": 8}, {"extract.fast_merge_from": 1},
                                  {"extract.fast_lds_kb": 12, "extract.fast_lds_whole_kb": 12}, {"extract.fast0_from_frames": 0}],
                         ids=["merge3", "merge5", "merge_none", "merge_all", "small_strips", "fast0_from_pyramid"])
def test_fast_plan_options(sd, oracle, opts):
    """The FAST launch plan never changes results: which levels share a launch (default: 6...), the LDS budgets that decide
    whole-cell vs strip processing (12 KB forces several strips per cell on every level), level 0 from the padded pyramid."""
    with sd.options(opts):
        ext = sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 2)
        ora = oracle.OrbOracle(1000, 1.2, 8, 20)
        rng = np.random.default_rng(7)
        imgs = np.stack([make_image(71), rng.integers(0, 256, size=(480, 640)).astype(np.uint8)])
        kps, desc, n = ext.extract_batch(imgs)
        for b in range(2):
            ok, od = ora.extract(imgs[b])
            assert len(ok) > 500 and n[b] == len(ok), (opts, b)
            assert np.array_equal(kps[b, :n[b]], ok) and np.array_equal(desc[b, :n[b]], od), (opts, b)
        ext.close()


@pytest.mark.parametrize("shape,cfg", [((479, 637), (1000, 1.2, 8, 20)), ((242, 321), (500, 1.2, 6, 20)),
                                       ((480, 640), (1000, 2.0, 4, 20)), ((360, 486), (800, 1.5, 5, 12))])
def test_odd_geometries_all_stages(sd, oracle, shape, cfg):
    """Odd widths / byte-unaligned rows (generic level-0 path), exact-2x levels (INTER_AREA fast path), a 1.5x
    pyramid: every stage compared with the oracle (padded pyramid levels, FAST counts, selected keys, blur)."""
    H, W = shape
    ext = sd.ORBextractor(*cfg, W, H, 2)
    ora = oracle.OrbOracle(*cfg)
    imgs = np.stack([np.ascontiguousarray(make_image(40 + i, max(W, 640), max(H, 480))[:H, :W]) for i in range(2)])
    kps, desc, n = ext.extract_batch(imgs)
    for i in range(2):
        ok, od = _compare_frame(oracle, ext, ora, imgs[i], i, cfg[2])
        assert n[i] == len(ok)
        assert np.array_equal(kps[i, :n[i]], ok) and np.array_equal(desc[i, :n[i]], od)
    ext.close()


@pytest.mark.parametrize("shape,cfg", [((480, 752), (1000, 1.2, 8, 20)), ((376, 1241), (2000, 1.2, 8, 20)), ((240, 320), (500, 1.2, 8, 20)),
                                       ((1080, 1920), (2000, 1.2, 8, 20)), ((600, 800), (1500, 1.3, 6, 15)),
                                       ((480, 640), (1000, 1.1, 12, 20)), ((97, 131), (200, 1.2, 4, 20))])
def test_more_geometries(sd, oracle, shape, cfg):
    """EuRoC / KITTI / QVGA / 1080p / other pyramids (12 levels, 1.3x) / a tiny frame: keypoints and descriptors of two frames per
    geometry equal the oracle's.  Exercises what depends on the plan: FAST launches per level vs the merged small levels, whole-cell
    vs strip staging, level-0 FAST straight from the frames, the selection's per-geometry buffers."""
    H, W = shape
    base = make_image(77)
    img = np.ascontiguousarray(np.tile(base, (3, 3))[:H, :W]) if (H > 480 or W > 640) else np.ascontiguousarray(base[:H, :W])
    frames = np.stack([img, img[::-1].copy()])
    ext = sd.ORBextractor(*cfg, W, H, 2)
    ora = oracle.OrbOracle(*cfg)
    k, d, n = ext.extract_batch(frames)
    for i in range(2):
        ok, od = ora.extract(frames[i])
        assert n[i] == len(ok) and len(ok) > 100
        assert np.array_equal(k[i, :n[i]], ok) and np.array_equal(d[i, :n[i]], od)
    ext.close()


@pytest.mark.parametrize("shape", [(48, 64), (47, 67), (49, 65), (48, 200), (100, 63), (57, 77), (120, 66)])
def test_pyramid_paths_at_their_size_thresholds(sd, oracle, shape):
    """k_pyr_split switches paths by level size: border rows as second stores from 48 rows up (single reflections), branch-free
    column reflection in the level-0 copy from 64 columns up, the per-group / per-row tables for every resized level.  Frames and
    levels right at and around those limits (and widths with every residue of w + 38 mod 4): padded pyramid, keypoints and
    descriptors equal the oracle's."""
    H, W = shape
    cfg = (150, 1.2, 3, 20)
    frames = np.stack([np.ascontiguousarray(make_image(90 + i)[i * 7:i * 7 + H, i * 5:i * 5 + W]) for i in range(3)])
    ext = sd.ORBextractor(*cfg, W, H, 3)
    ora = oracle.OrbOracle(*cfg)
    k, d, n = ext.extract_batch(frames)
    for i in range(3):
        ok, od = ora.extract(frames[i])
        for l in range(cfg[2]):
            assert np.array_equal(ext.level(l, i, padded=True), ora.level(l, padded=True)), f"pyramid level {l} of frame {i}"
        assert n[i] == len(ok)
        assert np.array_equal(k[i, :n[i]], ok) and np.array_equal(d[i, :n[i]], od)
    ext.close()


def test_errors_are_loud(sd):
    ext = sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 1)
    with pytest.raises(sd.SdError):
        ext(np.zeros((481, 640), np.uint8))       # larger than max_h
    with pytest.raises(sd.SdError):
        ext.extract_batch(np.zeros((2, 480, 640), np.uint8))   # exceeds max_batch
    ext.close()


def test_undistort_keypoints_bit_exact(sd, oracle):
    """Frame::UndistortKeyPoints (src/Frame.cc:333-363): mvKeysUn for a TUM1-like camera vs the oracle's
    cvUndistortPoints restatement; k1 == 0 -> mvKeysUn == mvKeys."""
    Kd = (517.3, 516.5, 318.6, 255.3)
    dist = (0.2624, -0.9531, -0.0054, 0.0026, 1.1633)
    imgs = np.stack([make_image(70 + i) for i in range(2)])
    ext = sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 2)
    ext.set_distortion(*Kd, *dist)
    kps, desc, n = ext.extract_batch(imgs)
    un = ext.download_undistorted(0, 2)
    for i in range(2):
        k = kps[i, :n[i]]
        xy = np.stack([k["x"], k["y"]], 1)
        ref = oracle.undistort_points(xy, Kd, dist)
        u = un[i, :n[i]]
        assert np.array_equal(u["x"], ref[:, 0]) and np.array_equal(u["y"], ref[:, 1])
        assert np.abs(ref - xy).max() > 0.5                       # the distortion is not a no-op
        for fld in ("size", "angle", "response", "octave"):
            assert np.array_equal(u[fld], k[fld])
    ext.set_distortion(*Kd, 0.0, -0.9531, -0.0054, 0.0026, 1.1633)   # k1 == 0: copy
    kps2, _, n2 = ext.extract_batch(imgs)
    un2 = ext.download_undistorted(0, 2)
    for i in range(2):
        assert np.array_equal(un2[i, :n2[i]], kps2[i, :n2[i]])


def test_hipgraph_replay_matches_direct_launches(sd, oracle):
    """Option "extract.use_graph" (captured multi-stream pipeline replayed as one hipGraph): same bits as direct launches, on
    first capture and on replays, after a geometry change, with a second argument set, and when the option is switched
    off again on the same handle."""
    ora = oracle.OrbOracle(1000, 1.2, 8, 20)
    imgs = np.stack([make_image(3), make_image(4)])
    exp = [ora.extract(im) for im in imgs]
    ext = sd.ORBextractor(1000, 1.2, 8, 20, 640, 480, 2)
    with sd.options({"extract.use_graph": 1}):
        for rep in range(3):                       # capture, replay, replay
            k, d, n = ext.extract_batch(imgs)
            for i in range(2):
                assert np.array_equal(k[i, :n[i]], exp[i][0]) and np.array_equal(d[i, :n[i]], exp[i][1]), (rep, i)
        small = make_image(5)[:240, :320].copy()    # geometry change drops the graphs
        ks, ds = ext(small)
        es = oracle.OrbOracle(1000, 1.2, 8, 20).extract(small)
        assert np.array_equal(ks, es[0]) and np.array_equal(ds, es[1])
        k, d, n = ext.extract_batch(imgs[:1])       # different batch size: new capture
        assert np.array_equal(k[0, :n[0]], exp[0][0])
    assert sd.get_option("extract.use_graph") == 0
    k, d, n = ext.extract_batch(imgs)               # direct launches again
    for i in range(2):
        assert np.array_equal(k[i, :n[i]], exp[i][0]) and np.array_equal(d[i, :n[i]], exp[i][1])
    ext.close()
