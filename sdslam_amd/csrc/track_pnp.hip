// PnPsolver (RANSAC over EPnP) on MI355X, one wavefront per frame.
//
// Replaces SD_SLAM::PnPsolver (reference src/PnPsolver.cc): gather (:71-110), SetRansacParameters
// (:120-155), iterate (:162-244), Refine (:246-286), CheckInliers (:289-315), EPnP (:348-901) and
// SD_SLAM::Random (src/extra/utils.cc:23-26).  PnPsolver has no caller in the reference
// (SURVEY D1); it is built because BASELINE's north_star names it.
//
// Parallel shape: RANSAC draws depend only on the rand() stream, not on earlier hypotheses, so
// the 64 lanes of the wave evaluate 64 consecutive iterations' minimal-set EPnP at once (fp64,
// one-sided Jacobi SVD per lane); inlier masks are ballots over the correspondences; then the
// reference's SEQUENTIAL accept/refine logic (best-so-far, `>=` to consider, `>` to accept the
// refit, early return) is replayed in iteration order.  The refit EPnP (all best inliers) runs
// on one lane.  OpenCV's cvSVD/cvSolve/cvInvert are restated as the same one-sided Jacobi
// iteration order as OpenCV 3.2's JacobiSVDImpl_, so the 4-point (rank-deficient) null-space
// bases agree with the oracle (hypot is the host libm's, restated in sd_hypot.h).
#include <hip/hip_runtime.h>

#include <cfloat>

#include "orb_internal.h"
#include "sd_hypot.h"
#include "track_internal.h"

namespace sd {

#define PNP_MAXN 1024
#define PNP_CHUNK 16   // RANSAC hypotheses evaluated side by side (typical runs accept within the first few)
#define PNP_WORDS (PNP_MAXN / 64)

// Lane-interleaved LDS array: element i of this lane lives at p[i * PNP_CHUNK], so the
// PNP_CHUNK lanes that solve hypotheses side by side hit distinct banks.  The big per-lane EPnP
// work arrays (12x12 Gram / singular-vector matrix, L_6x10) live here instead of in private
// (scratch) memory: with 9 KB of scratch per lane the kernel was bound by scratch traffic.
struct LArr {
  double* p;
  __device__ __forceinline__ double& operator[](int i) const { return p[i * PNP_CHUNK]; }
  __device__ __forceinline__ LArr operator+(int off) const { return LArr{p + off * PNP_CHUNK}; }
};

// ---- one-sided Jacobi SVD (OpenCV 3.2 JacobiSVDImpl_<double>), n <= 12 --------------------
// Split in two: the rotation sweeps (jacobi_sweeps / jacobi_sweeps12_reg) and the common tail
// (final norms, descending sort, normalisation / zero-singular-value fill-in).
__device__ __noinline__ void jacobi_sweeps(double* At, int astep, double* W, double* Vt, int vstep, int m, int n) {
  const double eps = DBL_EPSILON * 10;
  int i, j, k, iter, max_iter = m > 30 ? m : 30;
  double c, s, sd;
  for (i = 0; i < n; i++) {
    for (k = 0, sd = 0; k < m; k++) {
      double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = sd;
    for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
    Vt[i * vstep + i] = 1;
  }
  for (iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (i = 0; i < n - 1; i++)
      for (j = i + 1; j < n; j++) {
        double *Ai = At + i * astep, *Aj = At + j * astep;
        double a = W[i], p = 0, b = W[j];
        for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = sdsc::hypot_glibc(p, beta);
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        a = b = 0;
        for (k = 0; k < m; k++) {
          double t0 = c * Ai[k] + s * Aj[k];
          double t1 = -s * Ai[k] + c * Aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          a += t0 * t0;
          b += t1 * t1;
        }
        W[i] = a;
        W[j] = b;
        changed = true;
        double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
        for (k = 0; k < n; k++) {
          double t0 = c * Vi[k] + s * Vj[k];
          double t1 = -s * Vi[k] + c * Vj[k];
          Vi[k] = t0;
          Vj[k] = t1;
        }
      }
    if (!changed) break;
  }
}

// The 12x12 case (EPnP's M^T M, evaluated per RANSAC hypothesis) with the matrix held in
// REGISTERS: all (i, j, k) loops are unrolled so every index is a compile-time constant; the
// operation sequence is exactly jacobi_sweeps' (same sums in the same order), only the storage
// differs.  The right singular vectors are not needed by any 12x12 caller, so Vt is left as the
// identity (it only rides along in the tail's row swaps).
template <typename Ptr>
__device__ void jacobi_sweeps12_reg(Ptr At, double* W) {
  const double eps = DBL_EPSILON * 10;
  double a[12][12], w[12];
#pragma unroll
  for (int i = 0; i < 12; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
      a[i][k] = At[i * 12 + k];
      sd += a[i][k] * a[i][k];
    }
    w[i] = sd;
  }
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
#pragma unroll
    for (int i = 0; i < 11; i++) {
#pragma unroll
      for (int j = i + 1; j < 12; j++) {
        double p = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) p += a[i][k] * a[j][k];
        if (!(fabs(p) <= eps * sqrt(w[i] * w[j]))) {
          p *= 2;
          double c, s;
          const double beta = w[i] - w[j], gamma = sdsc::hypot_glibc(p, beta);
          if (beta < 0) {
            const double delta = (gamma - beta) * 0.5;
            s = sqrt(delta / gamma);
            c = p / (gamma * s * 2);
          } else {
            c = sqrt((gamma + beta) / (gamma * 2));
            s = p / (gamma * c * 2);
          }
          double na = 0, nb = 0;
#pragma unroll
          for (int k = 0; k < 12; k++) {
            const double t0 = c * a[i][k] + s * a[j][k];
            const double t1 = -s * a[i][k] + c * a[j][k];
            a[i][k] = t0;
            a[j][k] = t1;
            na += t0 * t0;
            nb += t1 * t1;
          }
          w[i] = na;
          w[j] = nb;
          changed = true;
        }
      }
    }
    if (!changed) break;
  }
#pragma unroll
  for (int i = 0; i < 12; i++) {
    W[i] = w[i];
#pragma unroll
    for (int k = 0; k < 12; k++) At[i * 12 + k] = a[i][k];
  }
}

template <typename Ptr>
__device__ __noinline__ void jacobi_finish(Ptr At, int astep, double* W, double* Wout, double* Vt, int vstep, int m, int n, int n1) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  int i, j, k, iter;
  double s, sd;
  for (i = 0; i < n; i++) {
    for (k = 0, sd = 0; k < m; k++) {
      double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = sqrt(sd);
  }
  for (i = 0; i < n - 1; i++) {
    j = i;
    for (k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      double tw = W[i]; W[i] = W[j]; W[j] = tw;
      for (k = 0; k < m; k++) { double t = At[i * astep + k]; At[i * astep + k] = At[j * astep + k]; At[j * astep + k] = t; }
      if (Vt)
        for (k = 0; k < n; k++) { double t = Vt[i * vstep + k]; Vt[i * vstep + k] = Vt[j * vstep + k]; Vt[j * vstep + k] = t; }
    }
  }
  for (i = 0; i < n; i++) Wout[i] = W[i];
  unsigned long long rng = 0x12345678ull;
  for (i = 0; i < n1; i++) {
    sd = i < n ? W[i] : 0;
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
      const double val0 = 1. / m;
      for (k = 0; k < m; k++) {
        rng = (unsigned long long)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
        double val = ((unsigned)rng & 256) != 0 ? val0 : -val0;
        At[i * astep + k] = val;
      }
      for (iter = 0; iter < 2; iter++) {
        for (j = 0; j < i; j++) {
          sd = 0;
          for (k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
          double asum = 0;
          for (k = 0; k < m; k++) {
            double t = At[i * astep + k] - sd * At[j * astep + k];
            At[i * astep + k] = t;
            asum += fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
          for (k = 0; k < m; k++) At[i * astep + k] *= asum;
        }
      }
      sd = 0;
      for (k = 0; k < m; k++) {
        double t = At[i * astep + k];
        sd += t * t;
      }
      sd = sqrt(sd);
    }
    s = sd > minval ? 1 / sd : 0.;
    for (k = 0; k < m; k++) At[i * astep + k] *= s;
  }
}

__device__ void jacobi_svd(double* At, int astep, double* Wout, double* Vt, int vstep, int m, int n, int n1) {
  double W[12];
  jacobi_sweeps(At, astep, W, Vt, vstep, m, n);
  jacobi_finish(At, astep, W, Wout, Vt, vstep, m, n, n1);
}

// cvSVD of the symmetric 12x12 M^T M held (in place) in `A`: on return rows of A are the left
// singular vectors (descending singular values).  A^T == A, so no transpose copy is needed; the
// right singular vectors are not needed by EPnP and are not formed (they only ride along in
// JacobiSVDImpl_'s final row swaps).
// Same sweeps with the matrix left in (lane-interleaved) LDS: ~1/4 of the registers of the
// register-resident version, so the PnP wave no longer owns a whole SIMD's register file and
// other kernels (the next batch's pyramid / FAST) can run beside it.  Identical operation order.
template <typename Ptr>
__device__ __noinline__ void jacobi_sweeps12_mem(Ptr A, double* W) {
  const double eps = DBL_EPSILON * 10;
  for (int i = 0; i < 12; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const double t = A[i * 12 + k];
      sd += t * t;
    }
    W[i] = sd;
  }
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
    for (int i = 0; i < 11; i++)
      for (int j = i + 1; j < 12; j++) {
        const Ptr Ai = A + i * 12, Aj = A + j * 12;
        double ai[12], aj[12];
#pragma unroll
        for (int k = 0; k < 12; k++) {
          ai[k] = Ai[k];
          aj[k] = Aj[k];
        }
        double p = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) p += ai[k] * aj[k];
        const double a = W[i], b = W[j];
        if (fabs(p) <= eps * sqrt(a * b)) continue;
        p *= 2;
        double c, s;
        const double beta = a - b, gamma = sdsc::hypot_glibc(p, beta);
        if (beta < 0) {
          const double delta = (gamma - beta) * 0.5;
          s = sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = sqrt((gamma + beta) / (gamma * 2));
          s = p / (gamma * c * 2);
        }
        double na = 0, nb = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) {
          const double t0 = c * ai[k] + s * aj[k];
          const double t1 = -s * ai[k] + c * aj[k];
          Ai[k] = t0;
          Aj[k] = t1;
          na += t0 * t0;
          nb += t1 * t1;
        }
        W[i] = na;
        W[j] = nb;
        changed = true;
      }
    if (!changed) break;
  }
}

template <typename Ptr>
__device__ void svd_sym12_inplace(Ptr A, double* Wout) {
  double W[12];
#ifdef SD_PNP_REG_JACOBI
  jacobi_sweeps12_reg(A, W);
#else
  jacobi_sweeps12_mem(A, W);
#endif
  jacobi_finish(A, 12, W, Wout, (double*)nullptr, 12, 12, 12, 12);
}

// SVD of a square row-major n x n matrix (n = 3 or 12): Ut rows = left vectors, Vt rows = right
__device__ void svd_square(const double* A, int n, double* W, double* Ut, double* Vt) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Ut[i * n + j] = A[j * n + i];
  jacobi_svd(Ut, n, W, Vt, n, n, n, n);
}

// cvSolve(A (6 x n), b, x, CV_SVD)
__device__ void solve_svd6(const double* A, int n, const double* b, double* x) {
  double a[30], v[25], w[5];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 6; j++) a[i * 6 + j] = A[j * n + i];
  jacobi_svd(a, 6, w, v, n, 6, n, n);
  for (int i = 0; i < n; i++) x[i] = 0;
  double threshold = 0;
  for (int i = 0; i < n; i++) threshold += w[i];
  threshold *= DBL_EPSILON * 2;
  for (int i = 0; i < n; i++) {
    double wi = w[i];
    if (fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    double s = 0;
    for (int j = 0; j < 6; j++) s += a[i * 6 + j] * b[j];
    s *= wi;
    for (int j = 0; j < n; j++) x[j] = x[j] + s * v[i * n + j];
  }
}

__device__ void invert_svd3(const double* A, double* Ainv) {
  double w[3], ut[9], vt[9];
  svd_square(A, 3, w, ut, vt);
  for (int i = 0; i < 9; i++) Ainv[i] = 0;
  double threshold = (w[0] + w[1] + w[2]) * DBL_EPSILON * 2;
  for (int i = 0; i < 3; i++) {
    double wi = w[i];
    if (fabs(wi) <= threshold) continue;
    wi = 1 / wi;
    double buffer[3];
    for (int j = 0; j < 3; j++) buffer[j] = ut[i * 3 + j] * wi;
    for (int r = 0; r < 3; r++)
      for (int j = 0; j < 3; j++) Ainv[r * 3 + j] += vt[i * 3 + r] * buffer[j];
  }
}

__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ double dist2_3(const double* p1, const double* p2) {
  return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
}

__device__ void qr_solve64(double* pA, double* pb, double* pX) {   // 6 x 4 Householder QR (src/PnPsolver.cc:812-901)
  const int nr = 6, nc = 4;
  double A1[4], A2[4];
  double* ppAkk = pA;
  for (int k = 0; k < nc; k++) {
    double *ppAik = ppAkk, eta = fabs(*ppAik);
    for (int i = k + 1; i < nr; i++) {
      double elt = fabs(*ppAik);
      if (eta < elt) eta = elt;
      ppAik += nc;
    }
    if (eta == 0) {
      A1[k] = A2[k] = 0.0;
      return;
    } else {
      double *q = ppAkk, sum = 0.0, inv_eta = 1. / eta;
      for (int i = k; i < nr; i++) {
        *q *= inv_eta;
        sum += *q * *q;
        q += nc;
      }
      double sigma = sqrt(sum);
      if (*ppAkk < 0) sigma = -sigma;
      *ppAkk += sigma;
      A1[k] = sigma * *ppAkk;
      A2[k] = -eta * sigma;
      for (int j = k + 1; j < nc; j++) {
        double *q2 = ppAkk, sum2 = 0;
        for (int i = k; i < nr; i++) {
          sum2 += *q2 * q2[j - k];
          q2 += nc;
        }
        double tau = sum2 / A1[k];
        q2 = ppAkk;
        for (int i = k; i < nr; i++) {
          q2[j - k] -= tau * *q2;
          q2 += nc;
        }
      }
    }
    ppAkk += nc + 1;
  }
  double* ppAjj = pA;
  for (int j = 0; j < nc; j++) {
    double *ppAij = ppAjj, tau = 0;
    for (int i = j; i < nr; i++) {
      tau += *ppAij * pb[i];
      ppAij += nc;
    }
    tau /= A1[j];
    ppAij = ppAjj;
    for (int i = j; i < nr; i++) {
      pb[i] -= tau * *ppAij;
      ppAij += nc;
    }
    ppAjj += nc + 1;
  }
  pX[nc - 1] = pb[nc - 1] / A2[nc - 1];
  for (int i = nc - 2; i >= 0; i--) {
    double *ppAij = pA + i * nc + (i + 1), sum = 0;
    for (int j = i + 1; j < nc; j++) {
      sum += *ppAij * pX[j];
      ppAij++;
    }
    pX[i] = (pb[i] - sum) / A2[i];
  }
}

struct EpnpCam { double fu, fv, uc, vc; };

// EPnP on n correspondences held in pws (3n) / us (2n); alphas (4n), pcs (3n) are work arrays.
// WAVE = false: every lane solves its own problem (RANSAC minimal sets).
// WAVE = true : all 64 lanes of the wave call with IDENTICAL arguments (the refit over the best
//               inlier set): the sequential parts run redundantly (same cost as one lane), the
//               M^T M accumulation -- the only O(n * 144) part -- is spread over the lanes, one or
//               two of the 78 upper-triangle entries per lane, each summed over the correspondences
//               in the reference's order (bit-identical sums), and exchanged through `lds_mtm`.
template <bool WAVE, typename Ptr>
__device__ __noinline__ double epnp_compute_pose(int n, const double* pws, const double* us, double* alphas, double* pcs,
                                    const EpnpCam cam, double R[3][3], double t[3], Ptr ut /* 144 */, Ptr L /* 60 */,
                                    double* lds_mtm = nullptr) {
  double cws[4][3], ccs[4][3];
  // In WAVE mode only lane 0 runs the sequential parts (63 idle lanes issue no private-memory
  // traffic); every lane joins the barriers and the M^T M accumulation.
  const bool lead = !WAVE || (threadIdx.x & 63) == 0;
  if (lead) {
  // choose_control_points
  cws[0][0] = cws[0][1] = cws[0][2] = 0;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
  for (int j = 0; j < 3; j++) cws[0][j] /= n;
  {
    double pw0tpw0[9], dc[3], uct[9], vtmp[9];
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += (pws[3 * k + a] - cws[0][a]) * (pws[3 * k + b] - cws[0][b]);
        pw0tpw0[a * 3 + b] = s;
      }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < a; b++) pw0tpw0[a * 3 + b] = pw0tpw0[b * 3 + a];
    svd_square(pw0tpw0, 3, dc, uct, vtmp);
    for (int i = 1; i < 4; i++) {
      double k = sqrt(dc[i - 1] / n);
      for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j];
    }
  }
  // compute_barycentric_coordinates
  {
    double cc[9], ci[9];
    for (int i = 0; i < 3; i++)
      for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
    invert_svd3(cc, ci);
    for (int i = 0; i < n; i++) {
      const double* pi = pws + 3 * i;
      double* a = alphas + 4 * i;
      for (int j = 0; j < 3; j++)
        a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
      a[0] = 1.0f - a[1] - a[2] - a[3];
    }
  }
  }  // lead
  // M^T M accumulated row by row (rows 2i, 2i+1 of M; same k-order as cvMulTransposed), built
  // in place in `ut` (it is symmetric, so it equals the transposed copy cvSVD would make)
  double d[12];
  if (WAVE) {
    const int lane = threadIdx.x & 63;
    __syncthreads();   // alphas[] (written by the lead lane) visible; lds_mtm free
    for (int e = lane; e < 78; e += 64) {
      int a = 0, rem = e;
      while (rem >= 12 - a) { rem -= 12 - a; a++; }
      const int b = a + rem;
      const int ka = a / 3, ca = a - 3 * ka, kb = b / 3, cb = b - 3 * kb;
      double acc = 0;
      for (int i = 0; i < n; i++) {
        const double aa = alphas[4 * i + ka], ab = alphas[4 * i + kb];
        const double u = us[2 * i], v = us[2 * i + 1];
        const double m1a = ca == 0 ? aa * cam.fu : (ca == 1 ? 0.0 : aa * (cam.uc - u));
        const double m1b = cb == 0 ? ab * cam.fu : (cb == 1 ? 0.0 : ab * (cam.uc - u));
        const double m2a = ca == 0 ? 0.0 : (ca == 1 ? aa * cam.fv : aa * (cam.vc - v));
        const double m2b = cb == 0 ? 0.0 : (cb == 1 ? ab * cam.fv : ab * (cam.vc - v));
        acc += m1a * m1b;
        acc += m2a * m2b;
      }
      lds_mtm[a * 12 + b] = acc;
      lds_mtm[b * 12 + a] = acc;
    }
    __syncthreads();
    if (lead)
      for (int i = 0; i < 144; i++) ut[i] = lds_mtm[i];
  } else {
    for (int i = 0; i < 144; i++) ut[i] = 0;
    for (int i = 0; i < n; i++) {
      const double* as = alphas + 4 * i;
      const double u = us[2 * i], v = us[2 * i + 1];
      double M1[12], M2[12];
      for (int k = 0; k < 4; k++) {
        M1[3 * k] = as[k] * cam.fu;
        M1[3 * k + 1] = 0.0;
        M1[3 * k + 2] = as[k] * (cam.uc - u);
        M2[3 * k] = 0.0;
        M2[3 * k + 1] = as[k] * cam.fv;
        M2[3 * k + 2] = as[k] * (cam.vc - v);
      }
      for (int a = 0; a < 12; a++)
        for (int b = a; b < 12; b++) {
          double acc = ut[a * 12 + b];
          acc += M1[a] * M1[b];
          acc += M2[a] * M2[b];
          ut[a * 12 + b] = acc;
        }
    }
    for (int a = 0; a < 12; a++)
      for (int b = 0; b < a; b++) ut[a * 12 + b] = ut[b * 12 + a];
  }
  double bestR[3][3], bestT[3], best_err = 0;
  if (lead) {
  svd_sym12_inplace(ut, d);
  // compute_L_6x10 / compute_rho
  double rho[6];
  {
    const Ptr v4[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
      int a = 0, b = 1;
      for (int j = 0; j < 6; j++) {
        dv[i][j][0] = v4[i][3 * a] - v4[i][3 * b];
        dv[i][j][1] = v4[i][3 * a + 1] - v4[i][3 * b + 1];
        dv[i][j][2] = v4[i][3 * a + 2] - v4[i][3 * b + 2];
        b++;
        if (b > 3) {
          a++;
          b = a + 1;
        }
      }
    }
    for (int i = 0; i < 6; i++) {
      Ptr row = L + 10 * i;
      row[0] = dot3(dv[0][i], dv[0][i]);
      row[1] = 2.0f * dot3(dv[0][i], dv[1][i]);
      row[2] = dot3(dv[1][i], dv[1][i]);
      row[3] = 2.0f * dot3(dv[0][i], dv[2][i]);
      row[4] = 2.0f * dot3(dv[1][i], dv[2][i]);
      row[5] = dot3(dv[2][i], dv[2][i]);
      row[6] = 2.0f * dot3(dv[0][i], dv[3][i]);
      row[7] = 2.0f * dot3(dv[1][i], dv[3][i]);
      row[8] = 2.0f * dot3(dv[2][i], dv[3][i]);
      row[9] = dot3(dv[3][i], dv[3][i]);
    }
    rho[0] = dist2_3(cws[0], cws[1]);
    rho[1] = dist2_3(cws[0], cws[2]);
    rho[2] = dist2_3(cws[0], cws[3]);
    rho[3] = dist2_3(cws[1], cws[2]);
    rho[4] = dist2_3(cws[1], cws[3]);
    rho[5] = dist2_3(cws[2], cws[3]);
  }
  for (int variant = 1; variant <= 3; variant++) {
    double betas[4];
    if (variant == 1) {
      double l[24], b4[4];
      for (int i = 0; i < 6; i++) {
        l[4 * i] = L[10 * i];
        l[4 * i + 1] = L[10 * i + 1];
        l[4 * i + 2] = L[10 * i + 3];
        l[4 * i + 3] = L[10 * i + 6];
      }
      solve_svd6(l, 4, rho, b4);
      if (b4[0] < 0) {
        betas[0] = sqrt(-b4[0]);
        betas[1] = -b4[1] / betas[0];
        betas[2] = -b4[2] / betas[0];
        betas[3] = -b4[3] / betas[0];
      } else {
        betas[0] = sqrt(b4[0]);
        betas[1] = b4[1] / betas[0];
        betas[2] = b4[2] / betas[0];
        betas[3] = b4[3] / betas[0];
      }
    } else if (variant == 2) {
      double l[18], b3[3];
      for (int i = 0; i < 6; i++) {
        l[3 * i] = L[10 * i];
        l[3 * i + 1] = L[10 * i + 1];
        l[3 * i + 2] = L[10 * i + 2];
      }
      solve_svd6(l, 3, rho, b3);
      if (b3[0] < 0) {
        betas[0] = sqrt(-b3[0]);
        betas[1] = (b3[2] < 0) ? sqrt(-b3[2]) : 0.0;
      } else {
        betas[0] = sqrt(b3[0]);
        betas[1] = (b3[2] > 0) ? sqrt(b3[2]) : 0.0;
      }
      if (b3[1] < 0) betas[0] = -betas[0];
      betas[2] = 0.0;
      betas[3] = 0.0;
    } else {
      double l[30], b5[5];
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 5; j++) l[5 * i + j] = L[10 * i + j];
      solve_svd6(l, 5, rho, b5);
      if (b5[0] < 0) {
        betas[0] = sqrt(-b5[0]);
        betas[1] = (b5[2] < 0) ? sqrt(-b5[2]) : 0.0;
      } else {
        betas[0] = sqrt(b5[0]);
        betas[1] = (b5[2] > 0) ? sqrt(b5[2]) : 0.0;
      }
      if (b5[1] < 0) betas[0] = -betas[0];
      betas[2] = b5[3] / betas[0];
      betas[3] = 0.0;
    }
    // gauss_newton (5 iterations)
    for (int k = 0; k < 5; k++) {
      double a[24], b[6], x[4] = {0, 0, 0, 0};
      for (int i = 0; i < 6; i++) {
        const Ptr rowL = L + i * 10;
        double* rowA = a + i * 4;
        rowA[0] = 2 * rowL[0] * betas[0] + rowL[1] * betas[1] + rowL[3] * betas[2] + rowL[6] * betas[3];
        rowA[1] = rowL[1] * betas[0] + 2 * rowL[2] * betas[1] + rowL[4] * betas[2] + rowL[7] * betas[3];
        rowA[2] = rowL[3] * betas[0] + rowL[4] * betas[1] + 2 * rowL[5] * betas[2] + rowL[8] * betas[3];
        rowA[3] = rowL[6] * betas[0] + rowL[7] * betas[1] + rowL[8] * betas[2] + 2 * rowL[9] * betas[3];
        b[i] = rho[i] - (rowL[0] * betas[0] * betas[0] + rowL[1] * betas[0] * betas[1] + rowL[2] * betas[1] * betas[1] +
                         rowL[3] * betas[0] * betas[2] + rowL[4] * betas[1] * betas[2] + rowL[5] * betas[2] * betas[2] +
                         rowL[6] * betas[0] * betas[3] + rowL[7] * betas[1] * betas[3] + rowL[8] * betas[2] * betas[3] +
                         rowL[9] * betas[3] * betas[3]);
      }
      qr_solve64(a, b, x);
      for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
    // compute_R_and_t: ccs, pcs, sign, Horn/Arun alignment, reprojection error
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0f;
    for (int i = 0; i < 4; i++) {
      const Ptr v = ut + 12 * (11 - i);
      for (int j = 0; j < 4; j++)
        for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
    for (int i = 0; i < n; i++) {
      const double* a = alphas + 4 * i;
      double* pc = pcs + 3 * i;
      for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
    if (pcs[2] < 0.0) {
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
      for (int i = 0; i < n; i++) {
        pcs[3 * i] = -pcs[3 * i];
        pcs[3 * i + 1] = -pcs[3 * i + 1];
        pcs[3 * i + 2] = -pcs[3 * i + 2];
      }
    }
    double Rv[3][3], tv[3];
    {
      double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
      for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) {
          pc0[j] += pcs[3 * i + j];
          pw0[j] += pws[3 * i + j];
        }
      for (int j = 0; j < 3; j++) {
        pc0[j] /= n;
        pw0[j] /= n;
      }
      double abt[9], abt_d[3], u3[9], v3[9];
      for (int i = 0; i < 9; i++) abt[i] = 0;
      for (int i = 0; i < n; i++) {
        const double* pc = pcs + 3 * i;
        const double* pw = pws + 3 * i;
        for (int j = 0; j < 3; j++) {
          abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
          abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
          abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
        }
      }
      svd_square(abt, 3, abt_d, u3, v3);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rv[i][j] = u3[i] * v3[j] + u3[3 + i] * v3[3 + j] + u3[6 + i] * v3[6 + j];
      const double det = Rv[0][0] * Rv[1][1] * Rv[2][2] + Rv[0][1] * Rv[1][2] * Rv[2][0] + Rv[0][2] * Rv[1][0] * Rv[2][1] -
                         Rv[0][2] * Rv[1][1] * Rv[2][0] - Rv[0][1] * Rv[1][0] * Rv[2][2] - Rv[0][0] * Rv[1][2] * Rv[2][1];
      if (det < 0) {
        Rv[2][0] = -Rv[2][0];
        Rv[2][1] = -Rv[2][1];
        Rv[2][2] = -Rv[2][2];
      }
      tv[0] = pc0[0] - dot3(Rv[0], pw0);
      tv[1] = pc0[1] - dot3(Rv[1], pw0);
      tv[2] = pc0[2] - dot3(Rv[2], pw0);
    }
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) {
      const double* pw = pws + 3 * i;
      double Xc = dot3(Rv[0], pw) + tv[0];
      double Yc = dot3(Rv[1], pw) + tv[1];
      double inv_Zc = 1.0 / (dot3(Rv[2], pw) + tv[2]);
      double ue = cam.uc + cam.fu * Xc * inv_Zc;
      double ve = cam.vc + cam.fv * Yc * inv_Zc;
      double u = us[2 * i], v = us[2 * i + 1];
      sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    const double err = sum2 / n;
    // N = 1; if (e2 < e1) N = 2; if (e3 < e[N]) N = 3
    if (variant == 1 || err < best_err) {
      best_err = err;
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) bestR[i][j] = Rv[i][j];
        bestT[i] = tv[i];
      }
    }
  }
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R[i][j] = bestR[i][j];
    t[i] = bestT[i];
  }
  }  // lead
  return best_err;
}

// CheckInliers for correspondence i under (R, t): src/PnPsolver.cc:289-315
__device__ __forceinline__ bool pnp_is_inlier(const double* Rt, const float* q /* {u,v,X,Y,Z,maxErr} */, const EpnpCam cam) {
  const float* p2 = q;
  const float* p3 = q + 2;
  const float maxErr = q[5];
  float Xc = (float)(Rt[0] * p3[0] + Rt[1] * p3[1] + Rt[2] * p3[2] + Rt[9]);
  float Yc = (float)(Rt[3] * p3[0] + Rt[4] * p3[1] + Rt[5] * p3[2] + Rt[10]);
  float invZc = (float)(1 / (Rt[6] * p3[0] + Rt[7] * p3[1] + Rt[8] * p3[2] + Rt[11]));
  double ue = cam.uc + cam.fu * Xc * invZc;
  double ve = cam.vc + cam.fv * Yc * invZc;
  float distX = (float)(p2[0] - ue);
  float distY = (float)(p2[1] - ve);
  float error2 = distX * distX + distY * distY;
  return error2 < maxErr;
}

__global__ __launch_bounds__(64, 4) void k_pnp(const sd_keypoint* __restrict__ kps_all, const int32_t* __restrict__ nkp_all,
                                            TrackBuffers tb, TrackCam tcam, const float* __restrict__ sigma2, PnpParams pp) {
  // gathered correspondences live in HBM (read-mostly, L2-resident): {u, v, X, Y, Z, maxErr} f32
  __shared__ double s_work[PNP_CHUNK * (144 + 60)];   // per-lane EPnP matrices, lane-interleaved
  __shared__ double s_Rt[PNP_CHUNK][12];
  __shared__ unsigned long long s_mask[PNP_CHUNK][PNP_WORDS];
  __shared__ unsigned long long s_best[PNP_WORDS], s_ref[PNP_WORDS];
  __shared__ int s_cnt[PNP_CHUNK];
  __shared__ double s_RtRef[12];
  __shared__ double s_mtm[144];
  const int f = blockIdx.x, lane = threadIdx.x;
  const int cap = tb.kp_cap;
  const sd_keypoint* kps = kps_all + (size_t)f * cap;
  const int nkp = min(nkp_all[f], cap);   // mvpMapPointMatches.size()
  const int32_t* cm = tb.cur_match + (size_t)f * cap;
  const double* Xw = tb.Xw + (size_t)f * tb.max_points * 3;
  int32_t* info = tb.pnp_info + (size_t)f * 8;
  uint8_t* inl_out = tb.pnp_inliers + (size_t)f * cap;
  float* T_out = tb.pnp_T + (size_t)f * 16;
  const EpnpCam cam = {(double)tcam.ffx, (double)tcam.ffy, (double)tcam.fcx, (double)tcam.fcy};
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  float* g_p = tb.pnp_pts + (size_t)f * cap * 6;
  uint16_t* g_idx = tb.pnp_kpidx + (size_t)f * cap;
  const int wl = lane < PNP_CHUNK ? lane : 0;
  const LArr w_ut{s_work + wl}, w_L{s_work + 144 * PNP_CHUNK + wl};

  for (int i = lane; i < cap; i += 64) inl_out[i] = 0;
  // ---- ctor gather, in keypoint order
  int N = 0;
  for (int base = 0; base < nkp; base += 64) {
    const int i = base + lane;
    const int m = i < nkp ? cm[i] : -1;
    const bool fl = m >= 0;
    const unsigned long long bal = __ballot(fl);
    const int pos = N + __popcll(bal & lt);
    if (fl && pos < PNP_MAXN) {
      const sd_keypoint kp = kps[i];
      float* q = g_p + (size_t)pos * 6;
      q[0] = kp.x;
      q[1] = kp.y;
      q[2] = (float)Xw[(size_t)m * 3];
      q[3] = (float)Xw[(size_t)m * 3 + 1];
      q[4] = (float)Xw[(size_t)m * 3 + 2];
      q[5] = sigma2[kp.octave] * pp.th2;
      g_idx[pos] = (uint16_t)i;
    }
    N += __popcll(bal);
  }
  N = min(N, PNP_MAXN);
  __syncthreads();
  // ---- SetRansacParameters
  float eps = pp.epsilon;
  int minInl = pp.min_inliers;
  {
    int nMin = (int)(N * eps);
    if (nMin < minInl) nMin = minInl;
    if (nMin < pp.min_set) nMin = pp.min_set;
    minInl = nMin;
    if (eps < (float)minInl / N) eps = (float)minInl / N;
  }
  int maxIts;
  {
    int nIt;
    if (minInl == N) nIt = 1;
    else nIt = (int)ceil(log(1 - pp.probability) / log(1 - pow((double)eps, 3.0)));
    maxIts = max(1, min(nIt, pp.max_iterations));
  }
  if (lane == 0) {
    info[0] = 0; info[1] = 0; info[2] = 0; info[3] = 0; info[4] = N; info[5] = minInl; info[6] = maxIts; info[7] = 0;
    for (int i = 0; i < 16; i++) T_out[i] = 0.f;
  }
  if (N < minInl) {
    if (lane == 0) info[2] = 1;   // bNoMore
    return;
  }
  const int total = max(maxIts, pp.n_iterations);   // while (mnIterations < maxIts || nCurrent < nIterations)
  const int nwords = (N + 63) >> 6;
  const int32_t* rs = tb.rand_stream + (size_t)f * pp.rand_per_frame;
  int best = 0, accepted = 0, acc_iters = 0, acc_cnt = 0;
  float bestT[12];
  for (int i = 0; i < 12; i++) bestT[i] = 0.f;
  double* scratch = tb.pnp_scratch + (size_t)f * cap * 12;

  for (int c0 = 0; c0 < total; c0 += PNP_CHUNK) {
    const int it = c0 + lane;
    const int nact = min(PNP_CHUNK, total - c0);
    if (lane < nact) {
      // minimal set: 4 draws without replacement from mvAllIndices via swap-with-back removal
      int modp[4], modv[4], nmod = 0, size = N;
      double pws[12], us[8], alphas[16], pcs[12];
      for (int k = 0; k < pp.min_set && k < 4; k++) {
        const int ridx = 4 * it + k;
        const int r = ridx < pp.rand_per_frame ? rs[ridx] : 0;
        const int randi = (int)(((double)r / (2147483647.0 + 1.0)) * size + 0);
        int val = randi, backv = size - 1;
        for (int q = 0; q < nmod; q++) {
          if (modp[q] == randi) val = modv[q];
          if (modp[q] == size - 1) backv = modv[q];
        }
        bool found = false;
        for (int q = 0; q < nmod; q++)
          if (modp[q] == randi) { modv[q] = backv; found = true; }
        if (!found) { modp[nmod] = randi; modv[nmod] = backv; nmod++; }
        size--;
        const float* q = g_p + (size_t)val * 6;
        pws[3 * k] = q[2];
        pws[3 * k + 1] = q[3];
        pws[3 * k + 2] = q[4];
        us[2 * k] = q[0];
        us[2 * k + 1] = q[1];
      }
      double R[3][3], t[3];
      epnp_compute_pose<false>(4, pws, us, alphas, pcs, cam, R, t, w_ut, w_L);
      for (int i = 0; i < 9; i++) s_Rt[lane][i] = R[i / 3][i % 3];
      for (int i = 0; i < 3; i++) s_Rt[lane][9 + i] = t[i];
    }
    __syncthreads();
    // inlier masks of the chunk's hypotheses
    for (int h = 0; h < nact; h++) {
      int cnt = 0;
      for (int w = 0; w < nwords; w++) {
        const int i = w * 64 + lane;
        bool in = false;
        if (i < N) in = pnp_is_inlier(s_Rt[h], g_p + (size_t)i * 6, cam);
        const unsigned long long bal = __ballot(in);
        if (lane == 0) s_mask[h][w] = bal;
        cnt += __popcll(bal);
      }
      if (lane == 0) s_cnt[h] = cnt;
    }
    __syncthreads();
    // sequential accept / refine replay
    for (int h = 0; h < nact; h++) {
      const int cnt = s_cnt[h];
      if (cnt < minInl) continue;
      if (cnt > best) {
        best = cnt;
        for (int w = lane; w < nwords; w += 64) s_best[w] = s_mask[h][w];
        for (int i = 0; i < 12; i++) bestT[i] = (float)s_Rt[h][i];
        __syncthreads();
        // Refine(): EPnP on the best inlier set (wave-cooperative), then CheckInliers
        double* pws = scratch;
        double* us = scratch + 3 * (size_t)cap;
        double* alphas = scratch + 5 * (size_t)cap;
        double* pcs = scratch + 9 * (size_t)cap;
        if (lane == 0) {
          int n = 0;
          for (int i = 0; i < N; i++)
            if ((s_best[i >> 6] >> (i & 63)) & 1ull) {
              const float* q = g_p + (size_t)i * 6;
              pws[3 * n] = q[2];
              pws[3 * n + 1] = q[3];
              pws[3 * n + 2] = q[4];
              us[2 * n] = q[0];
              us[2 * n + 1] = q[1];
              n++;
            }
        }
        __syncthreads();
        {
          double R[3][3], t[3];
          epnp_compute_pose<true>(best, pws, us, alphas, pcs, cam, R, t, w_ut, w_L, s_mtm);   // best == popcount(s_best)
          if (lane == 0) {
            for (int i = 0; i < 9; i++) s_RtRef[i] = R[i / 3][i % 3];
            for (int i = 0; i < 3; i++) s_RtRef[9 + i] = t[i];
          }
        }
        __syncthreads();
        int rcnt = 0;
        for (int w = 0; w < nwords; w++) {
          const int i = w * 64 + lane;
          bool in = false;
          if (i < N) in = pnp_is_inlier(s_RtRef, g_p + (size_t)i * 6, cam);
          const unsigned long long bal = __ballot(in);
          if (lane == 0) s_ref[w] = bal;
          rcnt += __popcll(bal);
        }
        __syncthreads();
        if (rcnt > minInl) {   // accepted: mRefinedTcw / mvbRefinedInliers are returned (written in the common tail)
          accepted = 1;
          acc_iters = c0 + h + 1;
          acc_cnt = rcnt;
        }
      }
      // cnt >= minInl but no new best: Refine() would refit the same best set and fail again
      if (accepted) break;
    }
    __syncthreads();
    if (accepted) break;
  }
  // ---- common tail (uniform control flow): write the returned pose and inlier flags
  __syncthreads();
  if (accepted) {
    for (int i = lane; i < N; i += 64)
      if ((s_ref[i >> 6] >> (i & 63)) & 1ull) inl_out[g_idx[i]] = 1;
    if (lane == 0) {
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T_out[r * 4 + c] = (float)s_RtRef[r * 3 + c];
        T_out[r * 4 + 3] = (float)s_RtRef[9 + r];
      }
      T_out[12] = T_out[13] = T_out[14] = 0.f;
      T_out[15] = 1.f;
      info[0] = 1; info[1] = acc_cnt; info[2] = 0; info[3] = acc_iters; info[7] = 1;
    }
    return;
  }
  // ---- iterations exhausted
  if (lane == 0) {
    info[2] = 1;
    info[3] = total;
  }
  if (best >= minInl) {
    for (int i = lane; i < N; i += 64)
      if ((s_best[i >> 6] >> (i & 63)) & 1ull) inl_out[g_idx[i]] = 1;
    if (lane == 0) {
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T_out[r * 4 + c] = bestT[r * 3 + c];
        T_out[r * 4 + 3] = bestT[9 + r];
      }
      T_out[12] = T_out[13] = T_out[14] = 0.f;
      T_out[15] = 1.f;
      info[0] = 1;
      info[1] = best;
    }
  }
}

// diagnostics: EPnP alone on explicit correspondences (one lane)
__global__ void k_epnp_debug(int n, const double* pws, const double* us, double* work, EpnpCam cam, double* out13) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double R[3][3], t[3];
  double ut[144], L[60];
  double e = epnp_compute_pose<false>(n, pws, us, work, work + 4 * (size_t)n, cam, R, t, (double*)ut, (double*)L);
  for (int i = 0; i < 9; i++) out13[i] = R[i / 3][i % 3];
  for (int i = 0; i < 3; i++) out13[9 + i] = t[i];
  out13[12] = e;
}

int run_epnp_debug(int n, const double* Xw, const double* uv, double fx, double fy, double cx, double cy, double* R9, double* t3,
                   double* err) {
  double *d_p = nullptr, *d_u = nullptr, *d_w = nullptr, *d_o = nullptr;
  SD_HIP_CHECK(hipMalloc(&d_p, sizeof(double) * 3 * n));
  SD_HIP_CHECK(hipMalloc(&d_u, sizeof(double) * 2 * n));
  SD_HIP_CHECK(hipMalloc(&d_w, sizeof(double) * 7 * n));
  SD_HIP_CHECK(hipMalloc(&d_o, sizeof(double) * 13));
  SD_HIP_CHECK(hipMemcpy(d_p, Xw, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
  SD_HIP_CHECK(hipMemcpy(d_u, uv, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
  EpnpCam cam = {fx, fy, cx, cy};
  hipLaunchKernelGGL(k_epnp_debug, dim3(1), dim3(64), 0, 0, n, d_p, d_u, d_w, cam, d_o);
  double out[13];
  SD_HIP_CHECK(hipMemcpy(out, d_o, sizeof(out), hipMemcpyDeviceToHost));
  for (int i = 0; i < 9; i++) R9[i] = out[i];
  for (int i = 0; i < 3; i++) t3[i] = out[9 + i];
  if (err) *err = out[12];
  (void)hipFree(d_p); (void)hipFree(d_u); (void)hipFree(d_w); (void)hipFree(d_o);
  return SD_OK;
}

int launch_pnp(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_sigma2, const PnpParams& pp,
               int n_frames, hipStream_t s) {
  hipLaunchKernelGGL(k_pnp, dim3(n_frames), dim3(64), 0, s, (cur->have_dist ? cur->d_kps_un : cur->d_kps), cur->d_nout, tb, cam, d_sigma2, pp);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

}  // namespace sd
