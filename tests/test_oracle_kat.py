"""Known-answer tests that pin the CPU oracle's OpenCV-primitive restatements from first
principles.  The reference ships no tests or golden vectors, so
these are the only pins the oracle has ("parity unpinned" against a real OpenCV build)."""
import math

import numpy as np
import pytest

from sdslam_amd.synth import make_image

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3),
        (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]


def brute_fast_scores(img, t):
    """Independent definition: corner iff 9 contiguous ring pixels all > v+t or all < v-t;
    score = (largest t' for which it is still a corner) = max over arcs of min|diff| - 1."""
    h, w = img.shape
    im = img.astype(np.int32)
    sc = np.zeros((h, w), np.int32)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            v = im[y, x]
            d = [v - im[y + dy, x + dx] for dx, dy in RING]
            best = -1
            for s in range(16):
                arc = [d[(s + k) % 16] for k in range(9)]
                best = max(best, min(arc), min(-a for a in arc))
            if best > t:
                sc[y, x] = best - 1
    return sc


def brute_fast(img, t):
    sc = brute_fast_scores(img, t)
    h, w = img.shape
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            s = sc[y, x]
            if s == 0:
                continue
            nb = sc[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, s))
    return out


def test_fast_matches_bruteforce_definition(oracle):
    rng = np.random.default_rng(1)
    for seed in range(3):
        img = make_image(seed, 96, 64)[:48, :72].copy()
        got = oracle.fast(img, 20, True)
        exp = brute_fast(img, 20)
        assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == exp
        assert len(exp) > 5
    # pure noise: many adjacent corners -> exercises plateau / NMS ties
    img = rng.integers(0, 256, size=(40, 40)).astype(np.uint8)
    got = oracle.fast(img, 10, True)
    exp = brute_fast(img, 10)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in got] == exp


def test_fast_threshold_edge_and_arc_length(oracle):
    def patch(n_bright, delta):
        img = np.full((16, 16), 100, np.uint8)
        for k in range(n_bright):
            dx, dy = RING[k]
            img[8 + dy, 8 + dx] = 100 + delta
        return img
    # exactly 9 contiguous brighter by t+1 -> corner with score t; by t -> not a corner
    k = oracle.fast(patch(9, 21), 20, True)
    assert [(int(a["x"]), int(a["y"]), int(a["response"])) for a in k if (a["x"], a["y"]) == (8, 8)] == [(8, 8, 20)]
    assert not [a for a in oracle.fast(patch(9, 20), 20, True) if (a["x"], a["y"]) == (8, 8)]
    # arc of 8 is not a corner
    assert not [a for a in oracle.fast(patch(8, 60), 20, True) if (a["x"], a["y"]) == (8, 8)]
    assert oracle.fast_score(patch(9, 60), 8, 8, 20) == 59
    # 3-px exclusion band: corner centre at x=2 is never reported
    img = np.full((16, 16), 100, np.uint8)
    img[:, :3] = 200
    assert all(a["x"] >= 3 and a["x"] <= 12 and a["y"] >= 3 and a["y"] <= 12 for a in oracle.fast(img, 20, True))


def test_resize_exact_2x_is_box_filter(oracle):
    img = make_image(3, 64, 48)
    got = oracle.resize_linear(img, 32, 24)
    a = img.astype(np.int32)
    exp = (a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2
    assert np.array_equal(got, exp.astype(np.uint8))


def test_resize_bilinear_fixed_point_formula(oracle):
    img = make_image(4, 640, 480)
    dw, dh = 533, 400
    got = oracle.resize_linear(img, dw, dh)

    def coeffs(dn, sn):
        scale = 1.0 / (dn / sn)
        ofs, c = [], []
        for d in range(dn):
            f = np.float32((d + 0.5) * scale - 0.5)
            s = int(math.floor(f))
            f = np.float32(f - np.float32(s))
            a1 = int(np.rint(np.float32(f * np.float32(2048))))
            a0 = int(np.rint(np.float32((np.float32(1) - f) * np.float32(2048))))
            ofs.append(s)
            c.append((a0, a1))
        return ofs, c

    xo, xc = coeffs(dw, 640)
    yo, yc = coeffs(dh, 480)
    a = img.astype(np.int64)
    xo = np.array(xo)
    x0 = a[:, xo] * np.array([c[0] for c in xc]) + a[:, np.minimum(xo + 1, 639)] * np.array([c[1] for c in xc])
    yo = np.array(yo)
    b0 = np.array([c[0] for c in yc])[:, None]
    b1 = np.array([c[1] for c in yc])[:, None]
    exp = (((b0 * (x0[yo] >> 4)) >> 16) + ((b1 * (x0[np.minimum(yo + 1, 479)] >> 4)) >> 16) + 2) >> 2
    assert np.array_equal(got, exp.astype(np.uint8))
    # within 1 grey level of float bilinear
    ramp = np.tile(np.arange(640, dtype=np.uint8)[None, :] // 3, (480, 1))
    r = oracle.resize_linear(ramp, dw, dh).astype(np.float64)
    fx = (np.arange(dw) + 0.5) * 640 / dw - 0.5
    assert np.abs(r[0] - np.interp(fx, np.arange(640), ramp[0])).max() <= 1.0


def test_border_reflect101(oracle):
    img = np.arange(6 * 7, dtype=np.uint8).reshape(6, 7)
    got = oracle.border101(img, 3)
    exp = np.pad(img, 3, mode="reflect")
    assert np.array_equal(got, exp)
    img = make_image(5, 50, 40)
    assert np.array_equal(oracle.border101(img, 19), np.pad(img, 19, mode="reflect"))


def test_gaussian_blur_taps_and_impulse(oracle):
    k = oracle.gauss_taps()
    g = np.exp(-((np.arange(7) - 3.0) ** 2) / 8.0)
    g /= g.sum()
    assert np.array_equal(k, np.rint(g * 256).astype(np.int32))
    assert k.tolist() == [18, 34, 49, 55, 49, 34, 18]  # sums to 257 (OpenCV <=3.4.1 fixed-point bias)
    img = np.zeros((21, 21), np.uint8)
    img[10, 10] = 255
    got = oracle.blur7(img)
    exp = np.zeros((21, 21), np.int64)
    exp[7:14, 7:14] = (np.outer(k, k) * 255 + (1 << 15)) >> 16
    assert np.array_equal(got, exp.astype(np.uint8))
    # reflect-101 at the borders == blurring the padded image
    src = make_image(6, 40, 30)
    pad = np.pad(src, 3, mode="reflect").astype(np.int64)
    rows = sum(k[i] * pad[3:-3, i:i + 40] for i in range(7))
    rows_full = sum(k[i] * pad[:, i:i + 40] for i in range(7))
    exp = (sum(k[i] * rows_full[i:i + 30] for i in range(7)) + (1 << 15)) >> 16
    assert rows.shape == (30, 40)
    assert np.array_equal(oracle.blur7(src), np.clip(exp, 0, 255).astype(np.uint8))


def test_fast_atan2_accuracy_and_quadrants(oracle):
    rng = np.random.default_rng(0)
    for _ in range(2000):
        y, x = rng.integers(-100000, 100000, size=2)
        if x == 0 and y == 0:
            continue
        ref = math.degrees(math.atan2(y, x)) % 360.0
        got = oracle.fast_atan2(y, x)
        err = abs(got - ref)
        assert min(err, 360 - err) < 0.3
    assert oracle.fast_atan2(0, 1) == 0.0
    assert abs(oracle.fast_atan2(1, 0) - 90.0) < 1e-4
    assert abs(oracle.fast_atan2(0, -1) - 180.0) < 1e-4
    assert abs(oracle.fast_atan2(-1, 0) - 270.0) < 1e-4


def test_orientation_single_bright_pixel(oracle):
    e = oracle.OrbOracle(1000, 1.2, 8, 20)
    for (u, v) in [(5, 0), (0, 7), (-6, 3), (4, -9), (-3, -3)]:
        img = np.zeros((41, 41), np.uint8)
        img[20 + v, 20 + u] = 200
        assert e.ic_angle(img, 20, 20) == pytest.approx(oracle.fast_atan2(v * 200, u * 200), abs=0)
    # outside the r=15 disc -> no moment
    img = np.zeros((41, 41), np.uint8)
    img[20 + 11, 20 + 11] = 200   # umax[11] = 10
    assert e.ic_angle(img, 20, 20) == oracle.fast_atan2(0, 0)
    assert e.tables()["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def test_quota_tables(oracle):
    assert oracle.OrbOracle(1000, 2.0, 5, 20).tables()["quota"].tolist() == [516, 258, 129, 65, 32]
    assert oracle.OrbOracle(1000, 1.2, 8, 20).tables()["quota"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    t = oracle.OrbOracle(1000, 1.2, 8, 20).tables()
    sf = np.float32(1)
    for i in range(8):
        assert t["sf"][i] == sf
        assert t["inv_sf"][i] == np.float32(1) / sf
        sf = np.float32(np.float64(sf) * np.float64(np.float32(1.2)))


def test_retain_best_keeps_top_n_multiset(oracle):
    rng = np.random.default_rng(2)
    for n, keep in [(50, 7), (300, 100), (10, 10), (10, 0), (9, 20)]:
        kps = np.zeros(n, oracle.KP_DTYPE)
        kps["x"] = np.arange(n)
        kps["response"] = rng.integers(20, 40, size=n)
        out = oracle.retain_best(kps, keep)
        if keep >= n:
            assert np.array_equal(out, kps)
            continue
        if keep == 0:
            assert len(out) == 0
            continue
        srt = np.sort(kps["response"])[::-1]
        assert len(out) >= keep
        assert sorted(out["response"][:keep].tolist(), reverse=True) == srt[:keep].tolist()
        # all ties of the boundary response are kept (hence the reference's explicit resize)
        assert len(out) == int((kps["response"] >= srt[keep - 1]).sum())


def test_brief_rotation_consistency(oracle):
    """A blob rotated by 90 degrees with the angle rotated by 90 gives the same bits."""
    e = oracle.OrbOracle(1000, 1.2, 8, 20)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(61, 61)).astype(np.uint8)
    d0 = e.brief(img, 30, 30, 0.0)
    d90 = e.brief(np.rot90(img, -1).copy(), 30, 30, 90.0)   # clockwise in image coords (y down)
    assert np.unpackbits(d0 ^ d90).sum() <= 2


def test_extract_end_to_end_shapes(oracle):
    img = make_image(0)
    for cfg, nexp in [((1000, 1.2, 8, 20), 1000), ((1000, 2.0, 5, 20), None)]:
        e = oracle.OrbOracle(*cfg)
        k, d = e.extract(img)
        assert d.shape == (len(k), 32)
        if nexp:
            assert len(k) == nexp
        assert (np.diff(k["octave"]) >= 0).all()           # level-major order
        assert (k["class_id"] == -1).all()
        sf = e.tables()["sf"]
        assert np.array_equal(k["size"], np.floor(31 * sf[k["octave"]].astype(np.float32)).astype(np.float32))
        # level coordinates are integers scaled by the level factor after description
        lv = k["octave"]
        xs = k["x"] / sf[lv]
        assert np.abs(xs - np.rint(xs)).max() < 1e-3
        # P5 level 4 (40x30) has a degenerate grid -> no keypoints
        if cfg[2] == 5:
            assert (k["octave"] < 4).all()
