"""Planted-outlier PnP RANSAC scenarios shared by the CPU oracle tests, the GPU parity tests and the golden generator.

A scenario is a match vector the tracker did NOT produce (PnPsolver takes any vpMapPointMatches, reference
src/PnPsolver.cc:71-110): keypoint i of the current frame is paired with map point i, whose world position is either the
back-projection of the keypoint at a random depth under the true pose (+ pixel noise) or a gross outlier."""
import numpy as np

from sdslam_amd import synth

K = (synth.FX, synth.FY, synth.CX, synth.CY)


planted = synth.planted_matches


def oracle_solver(O, kps, sigma2, last, cm):
    n = len(kps)
    valid = (cm >= 0).astype(np.uint8)
    Xw = np.zeros((n, 3))
    Xw[valid != 0] = last["Xw"][cm[valid != 0]]
    return O.PnPOracle(valid, np.stack([kps["x"], kps["y"]], 1), kps["octave"], sigma2, Xw, K)


# name -> (planted kwargs, SetRansacParameters args (prob, minInliers, maxIts, minSet, eps, th2), iterate() call sequence)
SCENARIOS = {
    # C4's parameters on a 30 % outlier set: accepted after a few hypotheses; the later calls re-enter a solver that has
    # already returned a refined pose (Refine() succeeds again on the unchanged best set)
    "out30_c4": (dict(n_match=300, outlier_frac=0.30, noise_px=0.5), (0.99, 10, 200, 4, 0.28, 5.991), [200, 5, 5]),
    # the reference's default parameters (src/PnPsolver.h:74) on a 50 % outlier set, noisy inliers
    "out50_default": (dict(n_match=300, outlier_frac=0.50, noise_px=0.5), (0.99, 8, 300, 4, 0.4, 5.991), [300]),
    # exactly minInliers exact inliers: every all-inlier hypothesis reaches `>= minInliers`, its refit finds the same set
    # and fails the strict `> minInliers` (src/PnPsolver.cc:274): runs to maxIts, returns the best hypothesis + bNoMore
    "refit_rejected": (dict(n_match=60, outlier_frac=0.0, n_exact_inliers=30), (0.99, 10, 200, 4, 0.5, 5.991), [200]),
    # 65 % outliers, chunked iterate(5) calls (the `||` of src/PnPsolver.cc:177 makes the first call run to maxIts)
    "out65_chunked": (dict(n_match=400, outlier_frac=0.65, noise_px=0.3), (0.99, 10, 200, 4, 0.2, 5.991), [5, 5, 20, 40]),
    # nothing but outliers: no hypothesis ever reaches minInliers -> empty Mat, bNoMore
    "all_outliers": (dict(n_match=200, outlier_frac=1.0), (0.99, 10, 120, 4, 0.28, 5.991), [120, 10]),
    # minimal sets other than 4 (SetRansacParameters takes any minSet; EPnP on 5 / 6 / 8 points)
    "minset5": (dict(n_match=250, outlier_frac=0.30, noise_px=0.5), (0.99, 10, 100, 5, 0.28, 5.991), [100, 3]),
    "minset6_out50": (dict(n_match=250, outlier_frac=0.50, noise_px=0.5), (0.99, 10, 100, 6, 0.3, 5.991), [100]),
    "minset8": (dict(n_match=200, outlier_frac=0.20, noise_px=0.3), (0.99, 10, 60, 8, 0.4, 5.991), [60]),
    # fewer points than EPnP's four control points: the covariance of the set is rank deficient, no hypothesis reaches
    # minInliers in either implementation (empty Mat + bNoMore); pins that degenerate sets are handled, not that they work
    "minset3_degenerate": (dict(n_match=250, outlier_frac=0.30, noise_px=0.5), (0.99, 10, 60, 3, 0.28, 5.991), [60]),
}


def rand_needed(params, calls):
    """Upper bound of the rand() values the call sequence can consume."""
    upper = 0
    for k, n in enumerate(calls):
        upper = max(params[2], upper + n) if k else max(params[2], n)
    return params[3] * upper


def run_oracle(O, kps, sigma2, last, cm, params, calls, rs):
    """The oracle's solver through the call sequence; returns the list of iterate() results (+ solver parameters)."""
    p = oracle_solver(O, kps, sigma2, last, cm)
    p.set_ransac(*params)
    out, consumed = [], 0
    for n in calls:
        r = p.iterate(n, rs[consumed:])
        consumed = params[3] * r["iterations"]
        out.append(r)
    return out, p.params()
