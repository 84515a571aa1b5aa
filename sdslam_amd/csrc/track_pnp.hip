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

#include <algorithm>
#include <cfloat>
#include <cstdlib>

#include "orb_internal.h"
#include "sd_hypot.h"
#include "track_internal.h"

namespace sd {

#define PNP_MAXN 2048   // = the tracker's keypoint capacity limit (sd_track_create)
#ifndef PNP_CHUNK
#define PNP_CHUNK 8    // RANSAC hypotheses evaluated side by side (typical runs accept within the first few);
                       // 3 lanes per hypothesis (one per EPnP beta variant)
#endif
#define PNP_WORDS (PNP_MAXN / 64)
#define PNP_MAXSET 64   // largest RANSAC minimal set (mRansacMinSet) the general path draws

// Lane-interleaved LDS array: element i of this lane lives at p[i * PNP_CHUNK], so the
// PNP_CHUNK lanes that solve hypotheses side by side hit distinct banks.  The big per-lane EPnP
// work arrays (12x12 Gram / singular-vector matrix, L_6x10) live here instead of in private
// (scratch) memory: with 9 KB of scratch per lane the kernel was bound by scratch traffic.
// Optional stage timers (build with -DSD_PNP_PROF; read through sd_debug_pnp_prof): shader-clock
// cycles of lane 0 per stage, summed over frames.  Slots 0-8: EPnP stages of the RANSAC minimal
// sets, 16-24: of the refit, 10-15: k_pnp phases.
#ifdef SD_PNP_PROF
__device__ unsigned long long g_pnp_prof[32];
#define PROF_DECL long long _pt = clock64()
#define PROF(i)                                                                                       \
  do {                                                                                                \
    long long _n = clock64();                                                                         \
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_pnp_prof[i], (unsigned long long)(_n - _pt));           \
    _pt = _n;                                                                                         \
  } while (0)
#else
#define PROF_DECL
#define PROF(i)
#endif

// The solvers below are real calls (one instruction stream per routine; inlining them all leaves the register allocator with
// > 500 live VGPRs).  A device function does not inherit its kernel's launch bounds: left alone it is compiled for the default
// 1024-thread workgroup, i.e. a 128-VGPR budget, and the unrolled fp64 Jacobi / QR bodies then spill inside their loops (r2: 2.1 KB
// of scratch per lane, 120 MB of scratch writes per launch).  k_pnp runs one 64-lane wave per frame at one wave per SIMD, so its
// callees get that kernel's budget stated explicitly.
#ifndef PNP_FN
#define PNP_FN static __noinline__ __attribute__((disable_tail_calls))
#endif

// LDS pointers keep their address space (ds_read / ds_write); through a generic double* every access
// becomes a flat_load that waits on both the LDS and the global counters.
typedef __attribute__((address_space(3))) double ldsd;
#define LDS_PTR(shared_array) ((ldsd*)(shared_array))
struct LArr {
  ldsd* p;
  __device__ __forceinline__ ldsd& operator[](int i) const { return p[i * PNP_CHUNK]; }
  __device__ __forceinline__ LArr operator+(int off) const { return LArr{p + off * PNP_CHUNK}; }
};

// ---- one-sided Jacobi SVD (OpenCV 3.2 JacobiSVDImpl_<double>), n <= 12 --------------------
// The 12 x 12 case (EPnP's M^T M) keeps the matrix and the squared norms W in lane-interleaved LDS
// (rotation sweeps: jacobi_sweeps12_coop, six lanes per matrix on the independent pairs of an anti-diagonal;
// tail: jacobi_finish12); the small cases are unrolled into
// registers (jacobi_svd_small).  Nothing lives in private memory: a dependent scratch access costs
// a global-memory round trip, which dominated the first version of this kernel.
// Tail of JacobiSVDImpl_ for the 12 x 12 case (final norms, descending selection sort of the rows,
// normalisation / zero-singular-value fill-in), matrix and W in (lane-interleaved) LDS.
template <typename Ptr>
__device__ PNP_FN void jacobi_finish12(Ptr At, Ptr W) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  constexpr int m = 12, n = 12;
  int i, j, k, iter;
  double s, sd;
  for (i = 0; i < n; i++) {
    sd = 0;
#pragma unroll
    for (k = 0; k < m; k++) {
      double t = At[i * m + k];
      sd += t * t;
    }
    W[i] = sqrt(sd);
  }
  for (i = 0; i < n - 1; i++) {
    j = i;
    double wj = W[i];
    for (k = i + 1; k < n; k++) {
      const double wk = W[k];
      if (wj < wk) {
        j = k;
        wj = wk;
      }
    }
    if (i != j) {
      double tw = W[i]; W[i] = W[j]; W[j] = tw;
#pragma unroll
      for (k = 0; k < m; k++) { double t = At[i * m + k]; At[i * m + k] = At[j * m + k]; At[j * m + k] = t; }
    }
  }
  unsigned long long rng = 0x12345678ull;
  for (i = 0; i < n; i++) {
    sd = W[i];
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
      const double val0 = 1. / m;
      for (k = 0; k < m; k++) {
        rng = (unsigned long long)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
        double val = ((unsigned)rng & 256) != 0 ? val0 : -val0;
        At[i * m + k] = val;
      }
      for (iter = 0; iter < 2; iter++) {
        for (j = 0; j < i; j++) {
          sd = 0;
#pragma unroll
          for (k = 0; k < m; k++) sd += At[i * m + k] * At[j * m + k];
          double asum = 0;
#pragma unroll
          for (k = 0; k < m; k++) {
            double t = At[i * m + k] - sd * At[j * m + k];
            At[i * m + k] = t;
            asum += fabs(t);
          }
          asum = asum > eps * 100 ? 1 / asum : 0;
#pragma unroll
          for (k = 0; k < m; k++) At[i * m + k] *= asum;
        }
      }
      sd = 0;
#pragma unroll
      for (k = 0; k < m; k++) {
        double t = At[i * m + k];
        sd += t * t;
      }
      sd = sqrt(sd);
    }
    s = sd > minval ? 1 / sd : 0.;
#pragma unroll
    for (k = 0; k < m; k++) At[i * m + k] *= s;
  }
}

// Zero-singular-value fill-in of row i (JacobiSVDImpl_'s tail): rare, so it stays generic and
// works on a private-memory copy.  Returns the row's new norm.
__device__ PNP_FN double jacobi_fill_row(double* At, int astep, int m, int i, unsigned long long* rng_io, double sd) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  unsigned long long rng = *rng_io;
  for (int ii = 0; ii < 100 && sd <= minval; ii++) {
    const double val0 = 1. / m;
    for (int k = 0; k < m; k++) {
      rng = (unsigned long long)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
      double val = ((unsigned)rng & 256) != 0 ? val0 : -val0;
      At[i * astep + k] = val;
    }
    for (int iter = 0; iter < 2; iter++) {
      for (int j = 0; j < i; j++) {
        sd = 0;
        for (int k = 0; k < m; k++) sd += At[i * astep + k] * At[j * astep + k];
        double asum = 0;
        for (int k = 0; k < m; k++) {
          double t = At[i * astep + k] - sd * At[j * astep + k];
          At[i * astep + k] = t;
          asum += fabs(t);
        }
        asum = asum > eps * 100 ? 1 / asum : 0;
        for (int k = 0; k < m; k++) At[i * astep + k] *= asum;
      }
    }
    sd = 0;
    for (int k = 0; k < m; k++) {
      double t = At[i * astep + k];
      sd += t * t;
    }
    sd = sqrt(sd);
  }
  *rng_io = rng;
  return sd;
}

// The small SVDs of EPnP (3 x 3, and 6 x {3,4,5} inside cvSolve) with compile-time shapes: every
// loop is unrolled so the matrices live in REGISTERS (the generic version keeps them in private
// memory, where each dependent access costs a scratch round trip).  Same operation order as
// jacobi_sweeps + jacobi_finish; At = N rows of length M (the transposed input), Vt = N x N.
template <int M, int N>
__device__ __forceinline__ void jacobi_svd_small(double (&At)[N][M], double (&Wout)[N], double (&Vt)[N][N]) {
  const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
  double W[N];
#pragma unroll
  for (int i = 0; i < N; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
    W[i] = sd;
#pragma unroll
    for (int k = 0; k < N; k++) Vt[i][k] = k == i ? 1 : 0;
  }
  for (int iter = 0; iter < 30; iter++) {   // max_iter = max(m, 30)
    bool changed = false;
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
#pragma unroll
      for (int j = i + 1; j < N; j++) {
        double p = 0;
#pragma unroll
        for (int k = 0; k < M; k++) p += At[i][k] * At[j][k];
        if (!(fabs(p) <= eps * sqrt(W[i] * W[j]))) {
          p *= 2;
          double c, s;
          const double beta = W[i] - W[j], gamma = sdsc::hypot_glibc(p, beta);
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
          for (int k = 0; k < M; k++) {
            const double t0 = c * At[i][k] + s * At[j][k];
            const double t1 = -s * At[i][k] + c * At[j][k];
            At[i][k] = t0;
            At[j][k] = t1;
            na += t0 * t0;
            nb += t1 * t1;
          }
          W[i] = na;
          W[j] = nb;
          changed = true;
#pragma unroll
          for (int k = 0; k < N; k++) {
            const double t0 = c * Vt[i][k] + s * Vt[j][k];
            const double t1 = -s * Vt[i][k] + c * Vt[j][k];
            Vt[i][k] = t0;
            Vt[j][k] = t1;
          }
        }
      }
    }
    if (!changed) break;
  }
#pragma unroll
  for (int i = 0; i < N; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
    W[i] = sqrt(sd);
  }
#pragma unroll
  for (int i = 0; i < N - 1; i++) {   // selection sort, descending
    int j = i;
    double wj = W[i];
#pragma unroll
    for (int k = i + 1; k < N; k++)
      if (wj < W[k]) {
        j = k;
        wj = W[k];
      }
#pragma unroll
    for (int jj = i + 1; jj < N; jj++)
      if (jj == j) {
        const double tw = W[i]; W[i] = W[jj]; W[jj] = tw;
#pragma unroll
        for (int k = 0; k < M; k++) { const double t = At[i][k]; At[i][k] = At[jj][k]; At[jj][k] = t; }
#pragma unroll
        for (int k = 0; k < N; k++) { const double t = Vt[i][k]; Vt[i][k] = Vt[jj][k]; Vt[jj][k] = t; }
      }
  }
#pragma unroll
  for (int i = 0; i < N; i++) Wout[i] = W[i];
  unsigned long long rng = 0x12345678ull;
#pragma unroll
  for (int i = 0; i < N; i++) {
    double sd = W[i];
    if (sd <= minval) {   // vanished singular value: build an orthogonal row (generic path on a copy)
      double tmp[N * M];
      for (int r = 0; r < N; r++)
        for (int k = 0; k < M; k++) tmp[r * M + k] = At[r][k];
      sd = jacobi_fill_row(tmp, M, M, i, &rng, sd);
#pragma unroll
      for (int k = 0; k < M; k++) At[i][k] = tmp[i * M + k];
    }
    const double s = sd > minval ? 1 / sd : 0.;
#pragma unroll
    for (int k = 0; k < M; k++) At[i][k] *= s;
  }
}

// cvSVD of the symmetric 12x12 M^T M held (in place) in `A`: on return rows of A are the left
// singular vectors (descending singular values).  A^T == A, so no transpose copy is needed; the
// right singular vectors are not needed by EPnP and are not formed (they only ride along in
// JacobiSVDImpl_'s final row swaps).
// One (i, j) step of JacobiSVDImpl_'s rotation sweeps (for (i) for (j > i) in cyclic order, up to 30 sweeps, stop after a
// sweep without rotation); rows are copied to registers, everything else stays in LDS.  Returns true when the rows were rotated.
template <typename Ptr>
__device__ __forceinline__ bool jacobi_pair12(Ptr A, Ptr W, int i, int j) {
  const double eps = DBL_EPSILON * 10;
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
  if (fabs(p) <= eps * sqrt(a * b)) return false;
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
  return true;
}

// The sweeps with up to six lanes per matrix.  In the cyclic order (0,1), (0,2), ..., (10,11) a pair only
// depends on the previous pairs that touched row i or row j; unrolling that recurrence gives pair (i, j) the
// earliest step i + j, so the pairs of one anti-diagonal (at most six) are independent of each other and a sweep takes
// 21 steps instead of 66.  Every pair executes exactly the scalar code above on the same inputs as in the sequential
// order, so the result has the same bits; only independent pairs change places.  Called by ALL 64 lanes of the (single-wave) workgroup: lane
// g < 6 of a matrix takes the g-th pair of the step; `act` = this lane's matrix exists.  A matrix whose sweep rotated
// nothing is finished (the sequential loop's `break`).
template <typename Ptr>
__device__ PNP_FN void jacobi_sweeps12_coop(Ptr A, Ptr W, int g, bool act, unsigned long long group_mask) {
  if (act && g < 6)
    for (int i = g; i < 12; i += 6) {
      double sd = 0;
#pragma unroll
      for (int k = 0; k < 12; k++) {
        const double t = A[i * 12 + k];
        sd += t * t;
      }
      W[i] = sd;
    }
  __syncthreads();
  bool live = act;
  for (int iter = 0; iter < 30; iter++) {
    bool changed = false;
    for (int st = 1; st <= 21; st++) {
      const int i = max(0, st - 11) + g, j = st - i;
      if (live && g < 6 && i < j) changed |= jacobi_pair12(A, W, i, j);
      __syncthreads();   // one wave: orders this step's LDS writes before the next step's reads
    }
    const unsigned long long any = __ballot(changed);
    live = live && (any & group_mask) != 0;
    if (!__any(live)) break;
  }
}

// cvSVD of a row-major 3 x 3 matrix: Ut rows = left vectors, Vt rows = right vectors
__device__ __forceinline__ void svd3(const double* A, double* W, double* Ut, double* Vt) {
  double at[3][3], w[3], vt[3][3];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) at[i][j] = A[j * 3 + i];
  jacobi_svd_small<3, 3>(at, w, vt);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    W[i] = w[i];
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Ut[i * 3 + j] = at[i][j];
      Vt[i * 3 + j] = vt[i][j];
    }
  }
}

// The three find_betas variants solve 6 x 4, 6 x 3 and 6 x 5 systems on three lanes of ONE wave: as three template
// instantiations they are three instruction streams that the wave executes one after the other (each with one lane active).
// This is the same one-sided Jacobi SVD + back-substitution with the column count N as a per-lane RUN-TIME value on a
// 5-row register layout: every loop is unrolled over the 5 rows and row i takes part iff i < N, so the lanes share one
// instruction stream and run side by side.  The operations a lane performs on its real rows, and their order, are exactly
// those of jacobi_svd_small<6, N> / solve_svd6<N> (the pairs (i, j), i < j < N, of the 5-row cyclic order are the N-row
// cyclic order; a sweep over a converged matrix rotates nothing), so the results are bit-identical.
// r3: A is read where it lies -- column i of the system is column col[i] of the 6 x 10 matrix L in LDS (the variant's choice of
// find_betas_approx_{1,2,3}) -- and b / x are registers of the (inlining) caller: nothing goes through private memory.
template <typename Ptr>
__device__ __forceinline__ void solve_svd6_n(int N, Ptr L, const int (&col)[5], const double (&b)[6], double (&x)[5]) {
  constexpr int M = 6, NM = 5;
  const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
  double At[NM][M], Vt[NM][NM], W[NM];
#pragma unroll
  for (int i = 0; i < NM; i++) {
#pragma unroll
    for (int j = 0; j < M; j++) At[i][j] = i < N ? L[10 * j + col[i]] : 0.0;
    double sd = 0;
#pragma unroll
    for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
    W[i] = sd;
#pragma unroll
    for (int k = 0; k < NM; k++) Vt[i][k] = k == i ? 1 : 0;
  }
  bool live = true;   // this lane's matrix still rotates
  for (int iter = 0; iter < 30; iter++) {   // max_iter = max(m, 30)
    bool changed = false;
#pragma unroll
    for (int i = 0; i < NM - 1; i++) {
#pragma unroll
      for (int j = i + 1; j < NM; j++) {
        if (live && j < N) {
          double p = 0;
#pragma unroll
          for (int k = 0; k < M; k++) p += At[i][k] * At[j][k];
          if (!(fabs(p) <= eps * sqrt(W[i] * W[j]))) {
            p *= 2;
            double c, sn;
            const double beta = W[i] - W[j], gamma = sdsc::hypot_glibc(p, beta);
            if (beta < 0) {
              const double delta = (gamma - beta) * 0.5;
              sn = sqrt(delta / gamma);
              c = p / (gamma * sn * 2);
            } else {
              c = sqrt((gamma + beta) / (gamma * 2));
              sn = p / (gamma * c * 2);
            }
            double na = 0, nb = 0;
#pragma unroll
            for (int k = 0; k < M; k++) {
              const double t0 = c * At[i][k] + sn * At[j][k];
              const double t1 = -sn * At[i][k] + c * At[j][k];
              At[i][k] = t0;
              At[j][k] = t1;
              na += t0 * t0;
              nb += t1 * t1;
            }
            W[i] = na;
            W[j] = nb;
            changed = true;
#pragma unroll
            for (int k = 0; k < NM; k++) {
              const double t0 = c * Vt[i][k] + sn * Vt[j][k];
              const double t1 = -sn * Vt[i][k] + c * Vt[j][k];
              Vt[i][k] = t0;
              Vt[j][k] = t1;
            }
          }
        }
      }
    }
    live = live && changed;   // the sequential loop's `if (!changed) break`
    if (!__any(live)) break;
  }
#pragma unroll
  for (int i = 0; i < NM; i++) {
    double sd = 0;
#pragma unroll
    for (int k = 0; k < M; k++) sd += At[i][k] * At[i][k];
    W[i] = sqrt(sd);
  }
#pragma unroll
  for (int i = 0; i < NM - 1; i++) {   // selection sort, descending, over the lane's N rows
    int j = i;
    double wj = W[i];
#pragma unroll
    for (int k = i + 1; k < NM; k++)
      if (k < N && wj < W[k]) {
        j = k;
        wj = W[k];
      }
#pragma unroll
    for (int jj = i + 1; jj < NM; jj++)
      if (jj == j) {
        const double tw = W[i]; W[i] = W[jj]; W[jj] = tw;
#pragma unroll
        for (int k = 0; k < M; k++) { const double t = At[i][k]; At[i][k] = At[jj][k]; At[jj][k] = t; }
#pragma unroll
        for (int k = 0; k < NM; k++) { const double t = Vt[i][k]; Vt[i][k] = Vt[jj][k]; Vt[jj][k] = t; }
      }
  }
  double w[NM];
#pragma unroll
  for (int i = 0; i < NM; i++) w[i] = W[i];
  unsigned long long rng = 0x12345678ull;
#pragma unroll
  for (int i = 0; i < NM; i++) {
    if (i < N) {
      double sd = W[i];
      if (sd <= minval) {   // vanished singular value: build an orthogonal row (generic path on a copy of the N rows)
        double tmp[NM * M];
        for (int r = 0; r < NM; r++)
          for (int k = 0; k < M; k++) tmp[r * M + k] = At[r][k];
        sd = jacobi_fill_row(tmp, M, M, i, &rng, sd);
#pragma unroll
        for (int k = 0; k < M; k++) At[i][k] = tmp[i * M + k];
      }
      const double sc = sd > minval ? 1 / sd : 0.;
#pragma unroll
      for (int k = 0; k < M; k++) At[i][k] *= sc;
    }
  }
  // cvSolve's back substitution
  double xr[NM];
#pragma unroll
  for (int i = 0; i < NM; i++) xr[i] = 0;
  double threshold = 0;
#pragma unroll
  for (int i = 0; i < NM; i++)
    if (i < N) threshold += w[i];
  threshold *= DBL_EPSILON * 2;
#pragma unroll
  for (int i = 0; i < NM; i++) {
    if (i < N) {
      double wi = w[i];
      if (!(fabs(wi) <= threshold)) {
        wi = 1 / wi;
        double sacc = 0;
#pragma unroll
        for (int j = 0; j < M; j++) sacc += At[i][j] * b[j];
        sacc *= wi;
#pragma unroll
        for (int j = 0; j < NM; j++) xr[j] = xr[j] + sacc * Vt[i][j];
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NM; i++) x[i] = i < N ? xr[i] : 0.0;
}

__device__ void invert_svd3(const double* A, double* Ainv) {
  double w[3], ut[9], vt[9];
  svd3(A, w, ut, vt);
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

// 6 x 4 Householder QR solve (src/PnPsolver.cc:812-901) with every index a compile-time constant
// (registers).  NB the reference's pivot search reads rows k .. nr-2 (it re-reads A[k][k] and never
// looks at the last row); that is kept.  A is row-major 6 x 4.
__device__ __forceinline__ void qr_solve64(const double (&pA)[24], const double (&pb)[6], double (&pX)[4]) {
  constexpr int nr = 6, nc = 4;
  double A[nr][nc], b[nr], A1[nc], A2[nc];
#pragma unroll
  for (int i = 0; i < nr; i++) {
    b[i] = pb[i];
#pragma unroll
    for (int j = 0; j < nc; j++) A[i][j] = pA[i * nc + j];
  }
  bool singular = false;
#pragma unroll
  for (int k = 0; k < nc; k++) {
    if (singular) break;
    double eta = fabs(A[k][k]);
#pragma unroll
    for (int i = k + 1; i < nr; i++) {
      const double elt = fabs(A[i - 1][k]);
      if (eta < elt) eta = elt;
    }
    if (eta == 0) {
      singular = true;   // "God damnit, A is singular, this shouldn't happen." -- x is left untouched
    } else {
      double sum = 0.0;
      const double inv_eta = 1. / eta;
#pragma unroll
      for (int i = k; i < nr; i++) {
        A[i][k] *= inv_eta;
        sum += A[i][k] * A[i][k];
      }
      double sigma = sqrt(sum);
      if (A[k][k] < 0) sigma = -sigma;
      A[k][k] += sigma;
      A1[k] = sigma * A[k][k];
      A2[k] = -eta * sigma;
#pragma unroll
      for (int j = k + 1; j < nc; j++) {
        double sum2 = 0;
#pragma unroll
        for (int i = k; i < nr; i++) sum2 += A[i][k] * A[i][j];
        const double tau = sum2 / A1[k];
#pragma unroll
        for (int i = k; i < nr; i++) A[i][j] -= tau * A[i][k];
      }
    }
  }
  if (singular) return;
#pragma unroll
  for (int j = 0; j < nc; j++) {
    double tau = 0;
#pragma unroll
    for (int i = j; i < nr; i++) tau += A[i][j] * b[i];
    tau /= A1[j];
#pragma unroll
    for (int i = j; i < nr; i++) b[i] -= tau * A[i][j];
  }
  double X[nc];
  X[nc - 1] = b[nc - 1] / A2[nc - 1];
#pragma unroll
  for (int i = nc - 2; i >= 0; i--) {
    double sum = 0;
#pragma unroll
    for (int j = i + 1; j < nc; j++) sum += A[i][j] * X[j];
    X[i] = (b[i] - sum) / A2[i];
  }
#pragma unroll
  for (int i = 0; i < nc; i++) pX[i] = X[i];
}

struct EpnpCam { double fu, fv, uc, vc; };

// ---- EPnP building blocks (src/PnPsolver.cc:348-901), shared by the minimal-set and the refit solvers ----

// choose_control_points tail: control points 1..3 from the centred covariance (pw0tpw0, full 3x3)
__device__ void epnp_control_points(const double* pw0tpw0, int n, double cws[4][3]) {
  double dc[3], uct[9], vtmp[9];
  svd3(pw0tpw0, dc, uct, vtmp);
  for (int i = 1; i < 4; i++) {
    double k = sqrt(dc[i - 1] / n);
    for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j];
  }
}

// compute_barycentric_coordinates: CC^-1 (cvInvert, CV_SVD)
__device__ void epnp_cc_inverse(const double cws[4][3], double ci[9]) {
  double cc[9];
  for (int i = 0; i < 3; i++)
    for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
  invert_svd3(cc, ci);
}

__device__ __forceinline__ void epnp_alphas(const double* pi, const double cws[4][3], const double* ci, double* a) {
  for (int j = 0; j < 3; j++)
    a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
  a[0] = 1.0f - a[1] - a[2] - a[3];
}

// L (6 x 10) and rho (6) of EPnP's beta system (src/PnPsolver.cc:716-764), table-driven.
//   pair p = (kPairA[p], kPairB[p]) enumerates the six control-point pairs (0,1) (0,2) (0,3) (1,2) (1,3) (2,3);
//   d_v[p] = (control point kPairA[p] - control point kPairB[p]) of null-space vector v (v = 0..3 <-> rows 11, 10, 9, 8 of ut);
//   column c of L pairs null-space vectors (kColU[c], kColV[c]): L[p][c] = (u == v ? 1 : 2) * <d_u[p], d_v[p]>,
//   the doubled entries being the cross terms of (sum_v beta_v d_v[p])^2;  rho[p] = |cw_a - cw_b|^2.
// Operation order inside a dot product and the float literal of the doubling are what bit parity needs; everything else is indexing.
__device__ const int8_t kPairA[6] = {0, 0, 0, 1, 1, 2}, kPairB[6] = {1, 2, 3, 2, 3, 3};
__device__ const int8_t kColU[10] = {0, 0, 1, 0, 1, 2, 0, 1, 2, 3}, kColV[10] = {0, 1, 1, 2, 2, 2, 3, 3, 3, 3};
template <typename Ptr>
__device__ void epnp_L_rho(Ptr ut, const double cws[4][3], Ptr L, double rho[6], bool write_L) {
  if (write_L) {
    for (int p = 0; p < 6; p++) {
      double d[4][3];   // the pair's difference vector in each of the four null-space vectors
      for (int v = 0; v < 4; v++) {
        const Ptr nv = ut + 12 * (11 - v);
        for (int k = 0; k < 3; k++) d[v][k] = nv[3 * kPairA[p] + k] - nv[3 * kPairB[p] + k];
      }
      Ptr row = L + 10 * p;
      for (int c = 0; c < 10; c++) {
        const double dp = dot3(d[kColU[c]], d[kColV[c]]);
        row[c] = kColU[c] == kColV[c] ? dp : 2.0f * dp;
      }
    }
  }
  for (int p = 0; p < 6; p++) rho[p] = dist2_3(cws[kPairA[p]], cws[kPairB[p]]);
}

// find_betas_approx_{1,2,3} followed by gauss_newton (5 iterations).  The three variants depend
// only on L and rho, never on each other, so they run on three lanes side by side.
struct Rho6 { double r[6]; };
struct Betas4 { double b[4]; };
// r3: L stays in LDS, rho arrives and the betas leave BY VALUE (registers), the 6 x N least-squares solve and the five Gauss-Newton
// QR solves are inlined here: the routine is a leaf with no operand in private memory (r2: l[30], a[24], b[6], x[4], bs[5] written to
// scratch for the callees to read back, 205 doubles per call).
template <typename Ptr>
__device__ PNP_FN void epnp_betas(int variant, Ptr L, const Rho6 rho_v, Betas4& out) {
  PROF_DECL;
  // find_betas_approx_{1,2,3} (src/PnPsolver.cc:700-780): the variant picks its columns of L, one shared solver
  const int N = variant == 1 ? 4 : (variant == 2 ? 3 : 5);
  const int col[5] = {0, 1, variant == 1 ? 3 : 2, variant == 1 ? 6 : 3, 4};
  double rho[6], bs[5], betas[4];
#pragma unroll
  for (int i = 0; i < 6; i++) rho[i] = rho_v.r[i];
  solve_svd6_n(N, L, col, rho, bs);
  if (variant == 1) {
    if (bs[0] < 0) {
      betas[0] = sqrt(-bs[0]);
      betas[1] = -bs[1] / betas[0];
      betas[2] = -bs[2] / betas[0];
      betas[3] = -bs[3] / betas[0];
    } else {
      betas[0] = sqrt(bs[0]);
      betas[1] = bs[1] / betas[0];
      betas[2] = bs[2] / betas[0];
      betas[3] = bs[3] / betas[0];
    }
  } else if (variant == 2) {
    if (bs[0] < 0) {
      betas[0] = sqrt(-bs[0]);
      betas[1] = (bs[2] < 0) ? sqrt(-bs[2]) : 0.0;
    } else {
      betas[0] = sqrt(bs[0]);
      betas[1] = (bs[2] > 0) ? sqrt(bs[2]) : 0.0;
    }
    if (bs[1] < 0) betas[0] = -betas[0];
    betas[2] = 0.0;
    betas[3] = 0.0;
  } else {
    if (bs[0] < 0) {
      betas[0] = sqrt(-bs[0]);
      betas[1] = (bs[2] < 0) ? sqrt(-bs[2]) : 0.0;
    } else {
      betas[0] = sqrt(bs[0]);
      betas[1] = (bs[2] > 0) ? sqrt(bs[2]) : 0.0;
    }
    if (bs[1] < 0) betas[0] = -betas[0];
    betas[2] = bs[3] / betas[0];
    betas[3] = 0.0;
  }
  PROF(5);
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
  PROF(6);
#pragma unroll
  for (int i = 0; i < 4; i++) out.b[i] = betas[i];
}

// compute_ccs
template <typename Ptr>
__device__ __forceinline__ void epnp_ccs(Ptr ut, const double* betas, double ccs[4][3]) {
  for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0f;
  for (int i = 0; i < 4; i++) {
    const Ptr v = ut + 12 * (11 - i);
    for (int j = 0; j < 4; j++)
      for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
  }
}

// estimate_R_and_t tail: ABt -> R, t (Arun), given the centroids
__device__ void epnp_Rt_from_abt(const double* abt, const double* pc0, const double* pw0, double Rv[3][3], double tv[3]) {
  double abt_d[3], u3[9], v3[9];
  svd3(abt, abt_d, u3, v3);
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

__device__ __forceinline__ double epnp_reproj_term(const double Rv[3][3], const double tv[3], const double* pw, double u, double v,
                                                   const EpnpCam cam) {
  double Xc = dot3(Rv[0], pw) + tv[0];
  double Yc = dot3(Rv[1], pw) + tv[1];
  double inv_Zc = 1.0 / (dot3(Rv[2], pw) + tv[2]);
  double ue = cam.uc + cam.fu * Xc * inv_Zc;
  double ve = cam.vc + cam.fv * Yc * inv_Zc;
  return sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
}

// ---- EPnP on a RANSAC minimal set (n = 4) -------------------------------------------------------
// Called by ALL 64 lanes.  Lane layout: hypothesis h = lane % PNP_CHUNK, beta variant = lane /
// PNP_CHUNK + 1 (lanes >= 3 * PNP_CHUNK idle).  The three lanes of a hypothesis compute the cheap
// front part redundantly (lock-step, so it costs nothing), the variant-1 lane builds M^T M, six lanes
// per hypothesis run the 12 x 12 Jacobi sweeps (jacobi_sweeps12_coop), the variant-1 lane finishes the SVD
// and writes L into the hypothesis' LDS view, then each lane follows its own
// find_betas variant / Gauss-Newton / R,t / reprojection error.  Returns that variant's error.
// r3: inlined into its one caller; the four correspondences are fetched by `gather(k, pw, u)` where they lie (twice: they are not
// kept across the find_betas call), and the barycentric coordinates wait in the lane's slot of the (by then idle) M-row scratch
// area while epnp_betas runs -- that call uses the whole register file, and what is alive across it would otherwise go to scratch.
template <typename Ptr, typename G>
__device__ __forceinline__ double epnp_minimal(bool active, int variant, G&& gather, const EpnpCam cam, double (&Rv)[3][3],
                                               double (&tv)[3], Ptr ut, Ptr L, ldsd* Mrows /* PNP_CHUNK x 4 x 24 */,
                                               int nhyp /* hypotheses in flight: lanes h < nhyp are active */) {
  constexpr int n = 4;
  double cws[4][3], alphas[16], rho[6];
  double pws[12], us[8];
  PROF_DECL;
  if (active) {
    for (int k = 0; k < n; k++) gather(k, pws + 3 * k, us + 2 * k);
    cws[0][0] = cws[0][1] = cws[0][2] = 0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[0][j] /= n;
    double pw0tpw0[9], ci[9];
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++) {
        double s = 0;
        for (int k = 0; k < n; k++) s += (pws[3 * k + a] - cws[0][a]) * (pws[3 * k + b] - cws[0][b]);
        pw0tpw0[a * 3 + b] = s;
      }
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < a; b++) pw0tpw0[a * 3 + b] = pw0tpw0[b * 3 + a];
    epnp_control_points(pw0tpw0, n, cws);
    PROF(0);
    epnp_cc_inverse(cws, ci);
    for (int i = 0; i < n; i++) epnp_alphas(pws + 3 * i, cws, ci, alphas + 4 * i);
    PROF(1);
  }
  // M^T M (rows 2i, 2i+1 of M; per entry the same k-order as cvMulTransposed: + M1a M1b, + M2a M2b for i = 0..3 from 0.0).
  // The variant-1 lane of every hypothesis parks its 8 rows of M in LDS, then the 78 upper-triangle entries of all
  // hypotheses are spread over the 64 lanes (one lane alone, with a read-modify-write of the LDS matrix per term, spent
  // 160 k cycles here); the result is written into `ut` of its hypothesis, both triangles (symmetric, so it equals the
  // transposed copy cvSVD would make).
  if (active && variant == 1) {
    const int h = threadIdx.x % PNP_CHUNK;
    for (int i = 0; i < n; i++) {
      const double* as = alphas + 4 * i;
      const double u = us[2 * i], v = us[2 * i + 1];
      ldsd* row = Mrows + (h * 4 + i) * 24;
      for (int k = 0; k < 4; k++) {
        row[3 * k] = as[k] * cam.fu;
        row[3 * k + 1] = 0.0;
        row[3 * k + 2] = as[k] * (cam.uc - u);
        row[12 + 3 * k] = 0.0;
        row[12 + 3 * k + 1] = as[k] * cam.fv;
        row[12 + 3 * k + 2] = as[k] * (cam.vc - v);
      }
    }
  }
  __syncthreads();
  for (int t = threadIdx.x; t < nhyp * 78; t += 64) {
    const int h = t / 78;
    int q = t - h * 78, a = 0;
    while (q >= 12 - a) { q -= 12 - a; a++; }
    const int b = a + q;
    const ldsd* rows = Mrows + h * 4 * 24;
    double acc = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      acc += rows[i * 24 + a] * rows[i * 24 + b];
      acc += rows[i * 24 + 12 + a] * rows[i * 24 + 12 + b];
    }
    ldsd* uh = ut.p - (threadIdx.x % PNP_CHUNK) + h;   // hypothesis h's lane-interleaved view of the work area
    uh[(a * 12 + b) * PNP_CHUNK] = acc;
    uh[(b * 12 + a) * PNP_CHUNK] = acc;
  }
  __syncthreads();
  PROF(2);
  {
    // 12 x 12 SVD: sweeps with six lanes per hypothesis (lanes h, h + 8, ..., h + 40 share hypothesis h's LDS view),
    // tail on the variant-1 lane
    const int lane = threadIdx.x, h = lane % PNP_CHUNK, g = lane / PNP_CHUNK;
    __syncthreads();
    const bool act_h = __shfl((int)active, h) != 0;   // lane h is the variant-1 lane of hypothesis h
    jacobi_sweeps12_coop(ut, ut + 144, g, act_h, 0x0000010101010101ull << h);
    if (active && variant == 1) jacobi_finish12(ut, ut + 144);
    PROF(3);
  }
  epnp_L_rho(ut, cws, L, rho, active && variant == 1);
  PROF(4);
  __syncthreads();   // ut / L of every hypothesis visible to its variant lanes
  double err = 0;
  ldsd* park = Mrows + (threadIdx.x & 63) * 16;   // 3 * PNP_CHUNK lanes x 16 doubles <= PNP_CHUNK x 4 x 24 (the M rows were consumed above)
  static_assert(3 * 16 <= 4 * 24, "alpha parking area");
  if (active) {
    for (int i = 0; i < 16; i++) park[i] = alphas[i];
    double betas[4], ccs[4][3], pcs[12];
    {
      Rho6 rv;
      for (int i = 0; i < 6; i++) rv.r[i] = rho[i];
      Betas4 bt;
      epnp_betas(variant, L, rv, bt);
      for (int i = 0; i < 4; i++) betas[i] = bt.b[i];
    }
    // the compiler knows that epnp_betas touches nothing but its arguments and would carry the values below across the call in
    // registers (i.e. in scratch): make it read them again
    asm volatile("" ::: "memory");
    for (int i = 0; i < 16; i++) alphas[i] = park[i];
    for (int k = 0; k < n; k++) gather(k, pws + 3 * k, us + 2 * k);
    PROF(9);   // (timed inside: slots 5 / 6)
    epnp_ccs(ut, betas, ccs);
    for (int i = 0; i < n; i++) {
      const double* a = alphas + 4 * i;
      for (int j = 0; j < 3; j++) pcs[3 * i + j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
    if (pcs[2] < 0.0)   // solve_for_sign (the sign of ccs is not used afterwards)
      for (int i = 0; i < 3 * n; i++) pcs[i] = -pcs[i];
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
    double abt[9];
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
    epnp_Rt_from_abt(abt, pc0, pw0, Rv, tv);
    PROF(7);
    double sum2 = 0.0;
    for (int i = 0; i < n; i++) sum2 += epnp_reproj_term(Rv, tv, pws + 3 * i, us[2 * i], us[2 * i + 1], cam);
    err = sum2 / n;
    PROF(8);
  }
  return err;
}

// ---- ordered sums over the correspondences, wave-cooperative ------------------------------------
// out[k] = sum_i term(i)[k], accumulated in i order from 0.0 exactly like the reference's
// sequential loops: the 64 lanes evaluate the terms of 64 consecutive i, park them in LDS, and
// lane k (< K) adds them up in order.  `terms` holds 64 * (K | 1) doubles; all lanes must call.
template <int K, typename F>
__device__ __forceinline__ void ordered_sums(int n, ldsd* terms, ldsd* out, F&& term) {
  constexpr int KP = K | 1;
  const int lane = threadIdx.x & 63;
  double acc = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    if (i < n) {
      double tv[K];
      term(i, tv);
#pragma unroll
      for (int k = 0; k < K; k++) terms[lane * KP + k] = tv[k];
    }
    __syncthreads();
    const int cnt = min(64, n - base);
    if (lane < K) {
#pragma unroll 8
      for (int j = 0; j < cnt; j++) acc += terms[j * KP + lane];
    }
    __syncthreads();
  }
  if (lane < K) out[lane] = acc;
  __syncthreads();
}

// ---- EPnP refit over the best inlier set (n >= 4), one wave, all lanes call with identical args ----
// Everything O(n) is spread over the lanes: per-correspondence work (alphas, pcs, reprojection
// terms) is lane-parallel, every reduction is an ordered_sums (bit-identical to the sequential
// loops of the reference).  M^T M: one lane per non-zero upper-triangle entry (62 of 78; the other
// 16 pair an x- with a y-column of M and stay +0), inputs staged through LDS.  The 12 x 12 SVD
// runs on lane 0, the three beta variants on lanes 0..2.  ut / L: lane 0's LDS views.
// lds: terms[64 * 9], red[64], mtm[144].  Writes {R (row-major), t} to Rt_out[12] (LDS).
// r3: pws / us are read where they lie -- TIn = float: the compacted copy of the gathered f32 correspondences (PnPsolver narrows
// its 3-D points to Point3f and takes kp.pt, src/PnPsolver.cc:92-93; (double) of the same float is the same double) -- and the
// barycentric coordinates (alphas) and camera-frame points (pcs) are RECOMPUTED from them wherever the reference reads its
// alphas[] / pcs[] arrays (same expressions on the same inputs: same bits).  Round 2 kept all four as double arrays in HBM:
// 96 bytes written and re-read several times per inlier, a third of the kernel's 19 x algorithmic traffic.
template <typename Ptr, typename TIn>
__device__ PNP_FN void epnp_refit_wave(int n, const TIn* pws, const TIn* us, const EpnpCam cam, ldsd* Rt_out, Ptr ut, Ptr L,
                                               ldsd* terms, ldsd* red, ldsd* mtm) {
  const int lane = threadIdx.x & 63;
  double cws[4][3], ci[9];
  auto PW = [&](int i, double* p) {
    p[0] = (double)pws[3 * i];
    p[1] = (double)pws[3 * i + 1];
    p[2] = (double)pws[3 * i + 2];
  };
  PROF_DECL;
  // choose_control_points
  ordered_sums<3>(n, terms, red, [&](int i, double* tv) { PW(i, tv); });
  for (int j = 0; j < 3; j++) cws[0][j] = red[j] / n;
  ordered_sums<6>(n, terms, red, [&](int i, double* tv) {
    double pw[3];
    PW(i, pw);
    const double d0 = pw[0] - cws[0][0], d1 = pw[1] - cws[0][1], d2 = pw[2] - cws[0][2];
    tv[0] = d0 * d0; tv[1] = d0 * d1; tv[2] = d0 * d2;
    tv[3] = d1 * d1; tv[4] = d1 * d2; tv[5] = d2 * d2;
  });
  if (lane == 0) {
    const double pw0tpw0[9] = {red[0], red[1], red[2], red[1], red[3], red[4], red[2], red[4], red[5]};
    epnp_control_points(pw0tpw0, n, cws);
    epnp_cc_inverse(cws, ci);
    for (int i = 0; i < 9; i++) {
      red[16 + i] = cws[1 + i / 3][i % 3];
      red[32 + i] = ci[i];
    }
    for (int j = 0; j < 3; j++) red[48 + j] = cws[0][j];
  }
  __syncthreads();
  for (int i = 0; i < 9; i++) {
    cws[1 + i / 3][i % 3] = red[16 + i];
    ci[i] = red[32 + i];
  }
  PROF(16);
  // compute_barycentric_coordinates: on demand
  auto AL = [&](int i, double* a) {
    double pw[3];
    PW(i, pw);
    epnp_alphas(pw, cws, ci, a);
  };
  for (int e = lane; e < 144; e += 64) mtm[e] = 0;
  PROF(17);
  // M^T M
  {
    int ea = 0, eb = 0, cnt = 0;
    for (int a = 0; a < 12; a++)
      for (int b = a; b < 12; b++) {
        const int ca = a % 3, cb = b % 3;
        if ((ca == 0 && cb == 1) || (ca == 1 && cb == 0)) continue;
        if (cnt == lane) { ea = a; eb = b; }
        cnt++;
      }
    const bool mine = lane < 62;
    const int ka = ea / 3, ca = ea % 3, kb = eb / 3, cb = eb % 3;
    const bool has1 = ca != 1 && cb != 1, has2 = ca != 0 && cb != 0;
    double acc = 0;
    for (int base = 0; base < n; base += 64) {
      const int i = base + lane;
      __syncthreads();
      if (i < n) {
        ldsd* row = terms + lane * 7;
        double a[4];
        AL(i, a);
        for (int k = 0; k < 4; k++) row[k] = a[k];
        row[4] = cam.uc - (double)us[2 * i];
        row[5] = cam.vc - (double)us[2 * i + 1];
      }
      __syncthreads();
      const int c = min(64, n - base);
      if (mine) {
#pragma unroll 4
        for (int j = 0; j < c; j++) {
          const ldsd* row = terms + j * 7;
          const double aa = row[ka], ab = row[kb];
          const double m1a = aa * (ca == 0 ? cam.fu : row[4]), m1b = ab * (cb == 0 ? cam.fu : row[4]);
          const double m2a = aa * (ca == 1 ? cam.fv : row[5]), m2b = ab * (cb == 1 ? cam.fv : row[5]);
          acc += has1 ? m1a * m1b : 0.0;
          acc += has2 ? m2a * m2b : 0.0;
        }
      }
    }
    if (mine) {
      mtm[ea * 12 + eb] = acc;
      mtm[eb * 12 + ea] = acc;
    }
  }
  __syncthreads();
  PROF(18);
  double rho[6];
  for (int i = lane; i < 144; i += 64) ut[i] = mtm[i];
  __syncthreads();
  jacobi_sweeps12_coop(ut, ut + 144, lane, true, 0x3full);   // six lanes on the one matrix
  if (lane == 0) jacobi_finish12(ut, ut + 144);
  PROF(19);
  epnp_L_rho(ut, cws, L, rho, lane == 0);
  __syncthreads();
  PROF(20);
  if (lane < 3) {
    Rho6 rv;
    for (int i = 0; i < 6; i++) rv.r[i] = rho[i];
    Betas4 bt;
    epnp_betas(lane + 1, L, rv, bt);
    for (int i = 0; i < 4; i++) red[4 * lane + i] = bt.b[i];
  }
  __syncthreads();
  // control points and their inverse again from LDS (slots untouched since they were broadcast): nothing of them stays in
  // registers across the epnp_betas call, which uses the whole register file
  asm volatile("" ::: "memory");
  for (int i = 0; i < 12; i++) cws[i / 3][i % 3] = red[i < 3 ? 48 + i : 13 + i];
  for (int i = 0; i < 9; i++) ci[i] = red[32 + i];
  double betas3[3][4];
  for (int i = 0; i < 12; i++) betas3[i / 4][i % 4] = red[i];
  __syncthreads();
  PROF(22);
  double bestR[3][3], bestT[3], best_err = 0;
  for (int variant = 1; variant <= 3; variant++) {
    double ccs[4][3];
    epnp_ccs(ut, betas3[variant - 1], ccs);
    // compute_pcs + solve_for_sign (decided by correspondence 0): pc = sum_k alpha_k ccs_k, negated as a whole when pcs[0].z < 0
    bool flip = false;
    auto PC = [&](int i, double* pc) {
      double a[4];
      AL(i, a);
      for (int j = 0; j < 3; j++) {
        const double x = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
        pc[j] = flip ? -x : x;
      }
    };
    {
      double pc_first[3];
      PC(0, pc_first);
      flip = pc_first[2] < 0.0;
    }
    // estimate_R_and_t
    double pc0[3], pw0[3], Rv[3][3], tv[3];
    ordered_sums<6>(n, terms, red, [&](int i, double* t6) {
      PC(i, t6);
      PW(i, t6 + 3);
    });
    for (int j = 0; j < 3; j++) {
      pc0[j] = red[j] / n;
      pw0[j] = red[3 + j] / n;
    }
    ordered_sums<9>(n, terms, red, [&](int i, double* t9) {
      double pc[3], pw[3];
      PC(i, pc);
      PW(i, pw);
      for (int j = 0; j < 3; j++) {
        t9[3 * j] = (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
        t9[3 * j + 1] = (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
        t9[3 * j + 2] = (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
      }
    });
    if (lane == 0) {
      double abt[9];
      for (int i = 0; i < 9; i++) abt[i] = red[i];
      epnp_Rt_from_abt(abt, pc0, pw0, Rv, tv);
      for (int i = 0; i < 9; i++) red[16 + i] = Rv[i / 3][i % 3];
      for (int i = 0; i < 3; i++) red[25 + i] = tv[i];
    }
    __syncthreads();
    for (int i = 0; i < 9; i++) Rv[i / 3][i % 3] = red[16 + i];
    for (int i = 0; i < 3; i++) tv[i] = red[25 + i];
    PROF(23);
    ordered_sums<1>(n, terms, red, [&](int i, double* t1) {
      double pw[3];
      PW(i, pw);
      t1[0] = epnp_reproj_term(Rv, tv, pw, (double)us[2 * i], (double)us[2 * i + 1], cam);
    });
    const double err = red[0] / n;
    __syncthreads();
    PROF(24);
    // N = 1; if (e2 < e1) N = 2; if (e3 < e[N]) N = 3
    if (variant == 1 || err < best_err) {
      best_err = err;
      for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) bestR[i][j] = Rv[i][j];
        bestT[i] = tv[i];
      }
    }
  }
  if (lane == 0) {
    for (int i = 0; i < 9; i++) Rt_out[i] = bestR[i / 3][i % 3];
    for (int i = 0; i < 3; i++) Rt_out[9 + i] = bestT[i];
    red[63] = best_err;   // the reprojection error of the winner (a function that RETURNS a value keeps the callee-saved
                          // register convention: 108 VGPRs saved and restored per call; a void one with local linkage does not)
  }
  __syncthreads();
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

// kGeneral: minimal sets of a size other than 4 (one hypothesis at a time through the wave-cooperative solver); the reference only ever
// constructs 4 (src/PnPsolver.h:74).  Two instantiations: the common kernel then carries neither that path's call site nor its state.
template <bool kGeneral>
__global__ __launch_bounds__(64, 2) __attribute__((disable_tail_calls)) void k_pnp(const sd_keypoint* __restrict__ kps_all, const int32_t* __restrict__ nkp_all,
                                            TrackBuffers tb, TrackCam tcam, const float* __restrict__ sigma2, PnpParams pp,
                                            int n_frames) {
  // gathered correspondences live in HBM (read-mostly, L2-resident): {u, v, X, Y, Z, maxErr} f32
  __shared__ double s_work[PNP_CHUNK * (156 + 60)];   // per-hypothesis EPnP matrices, lane-interleaved
  __shared__ double s_Rt[PNP_CHUNK][12];
  __shared__ unsigned long long s_mask[PNP_CHUNK][PNP_WORDS];
  __shared__ unsigned long long s_best[PNP_WORDS], s_ref[PNP_WORDS];
  __shared__ int s_cnt[PNP_CHUNK];
  __shared__ double s_RtRef[12];
  __shared__ float s_bestT[12];
  // One scratch area for two call-local uses (one wave per workgroup, the calls are sequential): the rows of M of the chunk's
  // minimal sets inside epnp_minimal, and the ordered-sum terms / MtM / reduction slots inside epnp_refit_wave.  Keeping them
  // apart cost 6 KB of the 30 KB this kernel holds per frame -- four frames per CU, beside k_fast_cells workgroups of 24-39 KB.
  constexpr int kScrMrows = PNP_CHUNK * 4 * 24, kScrRefit = 64 * 9 + 144 + 64;
  __shared__ double s_scr[kScrMrows > kScrRefit ? kScrMrows : kScrRefit];
  double* const s_Mrows = s_scr;
  double* const s_terms = s_scr;
  double* const s_mtm = s_scr + 64 * 9;
  double* const s_red = s_scr + 64 * 9 + 144;
  __shared__ int s_modp[PNP_MAXSET], s_modv[PNP_MAXSET], s_draw[PNP_MAXSET];   // general minimal sets (mRansacMinSet != 4)
  const int lane = threadIdx.x;
  // one wave per frame.  (r2 walked the frames with a persistent loop for a capped-grid experiment, "track.pnp_grid_cap": the
  // compiler hoisted everything invariant in that loop -- log / pow of the RANSAC parameters, the constants of the fp64 division
  // and square-root expansions -- to the kernel entry and kept it alive, i.e. in scratch, across every call below.)
  const int f = blockIdx.x;
  if (f >= n_frames) return;
  const int cap = tb.kp_cap;
  const int fc = tb.cur_bcast >= 0 ? tb.cur_bcast : f;   // current-frame slot (TrackBuffers::cur_bcast)
  const sd_keypoint* kps = kps_all + (size_t)fc * cap;
  const int nkp = min(nkp_all[fc], cap);   // mvpMapPointMatches.size()
  const int32_t* cm = tb.cur_match + (size_t)f * cap;
  const double* Xw = tb.Xw + (size_t)f * tb.max_points * 3;
  int32_t* info = tb.pnp_info + (size_t)f * 8;
  uint8_t* inl_out = tb.pnp_inliers + (size_t)f * cap;
  float* T_out = tb.pnp_T + (size_t)f * 16;
  const EpnpCam cam = {(double)tcam.ffx, (double)tcam.ffy, (double)tcam.fcx, (double)tcam.fcy};
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  float* g_p = tb.pnp_pts + (size_t)f * cap * 6;
  uint16_t* g_idx = tb.pnp_kpidx + (size_t)f * cap;
  const int hyp = lane % PNP_CHUNK, variant = lane / PNP_CHUNK + 1;   // lanes >= 3 * PNP_CHUNK: no hypothesis
  const LArr w_ut{LDS_PTR(s_work) + hyp}, w_L{LDS_PTR(s_work) + 156 * PNP_CHUNK + hyp};
  const LArr r_ut{LDS_PTR(s_work)}, r_L{LDS_PTR(s_work) + 156 * PNP_CHUNK};             // the refit uses hypothesis 0's storage

  PROF_DECL;
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
  PROF(10);
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
  if (!pp.resume && lane == 0) {   // a freshly constructed solver
    tb.pnp_state[(size_t)f * 4] = 0;
    tb.pnp_state[(size_t)f * 4 + 1] = 0;
    tb.pnp_state[(size_t)f * 4 + 2] = 0;
  }
  if (N < minInl) {
    if (lane == 0) {
      info[2] = 1;   // bNoMore
      info[3] = pp.resume ? tb.pnp_state[(size_t)f * 4] : 0;
    }
    return;
  }
  // ---- solver state that outlives an iterate() call (src/PnPsolver.h: mnIterations, mnBestInliers, mvbBestInliers,
  // mBestTcw); refine_ok = what Refine() returns for the CURRENT best set (it is a pure function of that set)
  int32_t* st = tb.pnp_state + (size_t)f * 4;
  unsigned long long* st_mask = tb.pnp_best_mask + (size_t)f * PNP_WORDS;
  float* st_T = tb.pnp_best_T + (size_t)f * 12;
  const int nwords = (N + 63) >> 6;
  int start = 0, best = 0, refine_ok = 0;
  if (lane < 12) s_bestT[lane] = 0.f;   // mBestTcw (LDS: twelve floats per lane would live across every solver call)
  if (pp.resume) {
    start = st[0];
    best = st[1];
    refine_ok = st[2];
    if (lane < 12) s_bestT[lane] = st_T[lane];
    for (int w = lane; w < nwords; w += 64) s_best[w] = st_mask[w];
  }
  // while (mnIterations < mRansacMaxIts || nCurrentIterations < nIterations)
  const int total = max(maxIts, start + pp.n_iterations);
  const int32_t* rs = tb.rand_stream + (size_t)f * pp.rand_per_frame;
  int accepted = 0, acc_iters = 0, acc_cnt = 0;
  float* scratch = tb.pnp_scratch + (size_t)f * cap * 5;
  const int mset = kGeneral ? pp.min_set : 4;
  const int chunk = kGeneral ? 1 : PNP_CHUNK;
  __syncthreads();

  for (int c0 = start; c0 < total; c0 += chunk) {
    PROF(11);
    const int nact = min(chunk, total - c0);
    if constexpr (!kGeneral) {
      const int it = c0 + hyp;
      const bool active = lane < 3 * PNP_CHUNK && hyp < nact;
      // minimal set: 4 draws without replacement from mvAllIndices via swap-with-back removal
      int modp[4], modv[4], nmod = 0, size = N;
      int draw[4];
      for (int k = 0; k < 4; k++) {
        const int ridx = 4 * it + k;
        const int r = (active && ridx < pp.rand_per_frame) ? rs[ridx] : 0;
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
        draw[k] = val;
      }
      double R[3][3], t[3];
      const double err = epnp_minimal(
          active, variant,
          [&](int k, double* pw, double* u) {
            const float* q = g_p + (size_t)draw[k] * 6;
            pw[0] = q[2];
            pw[1] = q[3];
            pw[2] = q[4];
            u[0] = q[0];
            u[1] = q[1];
          },
          cam, R, t, w_ut, w_L, LDS_PTR(s_Mrows), nact);
      // N = 1; if (e2 < e1) N = 2; if (e3 < e[N]) N = 3   (src/PnPsolver.cc:385-389)
      const double e1 = __shfl(err, hyp), e2 = __shfl(err, hyp + PNP_CHUNK), e3 = __shfl(err, hyp + 2 * PNP_CHUNK);
      int win = 1;
      double be = e1;
      if (e2 < be) { win = 2; be = e2; }
      if (e3 < be) win = 3;
      if (active && variant == win) {
        for (int i = 0; i < 9; i++) s_Rt[hyp][i] = R[i / 3][i % 3];
        for (int i = 0; i < 3; i++) s_Rt[hyp][9 + i] = t[i];
      }
    } else {
      // general minimal set (mRansacMinSet != 4; the reference only ever constructs 4, src/PnPsolver.h:74, but the setter takes
      // any): one hypothesis at a time through the wave-cooperative solver the refit uses.  The draws are the serial
      // swap-with-back removal of src/PnPsolver.cc:185-194 on lane 0 (list of modified positions instead of the N-entry copy).
      if (lane == 0) {
        int size = N, nmod = 0;
        for (int k = 0; k < mset; k++) {
          const int ridx = mset * c0 + k;
          const int r = ridx < pp.rand_per_frame ? rs[ridx] : 0;
          const int randi = (int)(((double)r / (2147483647.0 + 1.0)) * size + 0);
          int val = randi, backv = size - 1;
          for (int q = 0; q < nmod; q++) {
            if (s_modp[q] == randi) val = s_modv[q];
            if (s_modp[q] == size - 1) backv = s_modv[q];
          }
          bool found = false;
          for (int q = 0; q < nmod; q++)
            if (s_modp[q] == randi) { s_modv[q] = backv; found = true; }
          if (!found) { s_modp[nmod] = randi; s_modv[nmod] = backv; nmod++; }
          size--;
          s_draw[k] = val;
        }
      }
      __syncthreads();
      float* pws = scratch;
      float* us = scratch + 3 * (size_t)cap;
      if (lane < mset) {
        const float* q = g_p + (size_t)s_draw[lane] * 6;
        pws[3 * lane] = q[2];
        pws[3 * lane + 1] = q[3];
        pws[3 * lane + 2] = q[4];
        us[2 * lane] = q[0];
        us[2 * lane + 1] = q[1];
      }
      __syncthreads();
      epnp_refit_wave(mset, (const float*)pws, (const float*)us, cam, LDS_PTR(s_Rt[0]), r_ut, r_L, LDS_PTR(s_terms), LDS_PTR(s_red),
                      LDS_PTR(s_mtm));
    }
    __syncthreads();
    PROF(12);
    // inlier masks of the chunk's hypotheses.  r3: a correspondence is fetched ONCE and tested against every hypothesis of the chunk
    // (was: hypothesis-major, 8 x 4 rounds of six global loads each, every one waited for)
    {
      int cnts[PNP_CHUNK];
#pragma unroll
      for (int h = 0; h < PNP_CHUNK; h++) cnts[h] = 0;
      for (int w = 0; w < nwords; w++) {
        const int i = w * 64 + lane;
        float q[6];
#pragma unroll
        for (int k = 0; k < 6; k++) q[k] = i < N ? g_p[(size_t)i * 6 + k] : 0.f;
#pragma unroll
        for (int h = 0; h < PNP_CHUNK; h++) {
          if (h < nact) {
            const bool in = i < N && pnp_is_inlier(s_Rt[h], q, cam);
            const unsigned long long bal = __builtin_amdgcn_ballot_w64(in);
            if (lane == 0) s_mask[h][w] = bal;
            cnts[h] += __popcll(bal);
          }
        }
      }
#pragma unroll
      for (int h = 0; h < PNP_CHUNK; h++)
        if (lane == 0 && h < nact) s_cnt[h] = cnts[h];
    }
    __syncthreads();
    PROF(13);
    // sequential accept / refine replay
    for (int h = 0; h < nact; h++) {
      const int cnt = s_cnt[h];
      if (cnt < minInl) continue;
      const bool new_best = cnt > best;
      if (new_best) {
        best = cnt;
        for (int w = lane; w < nwords; w += 64) s_best[w] = s_mask[h][w];
        if (lane < 12) s_bestT[lane] = (float)s_Rt[h][lane];
      }
      // Refine() runs on every hypothesis that reaches minInliers (src/PnPsolver.cc:216) but is a pure function of the best
      // set: without a new best it repeats its last outcome -- a failure again (skipped), or, on a solver that has already
      // returned a refined pose (a later iterate() call), the same success again
      if (new_best || refine_ok) {
        __syncthreads();
        // Refine(): EPnP on the best inlier set (wave-cooperative), then CheckInliers
        float* pws = scratch;                    // the best set's correspondences, compacted (20 bytes each)
        float* us = scratch + 3 * (size_t)cap;
        {
          int n0 = 0;
          for (int w = 0; w < nwords; w++) {
            const unsigned long long bits = s_best[w];
            if ((bits >> lane) & 1ull) {
              const int n = n0 + __popcll(bits & lt);
              const float* q = g_p + (size_t)(w * 64 + lane) * 6;
              pws[3 * n] = q[2];
              pws[3 * n + 1] = q[3];
              pws[3 * n + 2] = q[4];
              us[2 * n] = q[0];
              us[2 * n + 1] = q[1];
            }
            n0 += __popcll(bits);
          }
        }
        __syncthreads();
        epnp_refit_wave(best, (const float*)pws, (const float*)us, cam, LDS_PTR(s_RtRef), r_ut, r_L, LDS_PTR(s_terms), LDS_PTR(s_red),
                        LDS_PTR(s_mtm));   // best == popcount(s_best)
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
        refine_ok = rcnt > minInl;
        if (refine_ok) {   // accepted: mRefinedTcw / mvbRefinedInliers are returned (written in the common tail)
          accepted = 1;
          acc_iters = c0 + h + 1;
          acc_cnt = rcnt;
        }
      }
      if (accepted) break;
    }
    __syncthreads();
    PROF(14);
    if (accepted) break;
  }
  // ---- common tail (uniform control flow): solver state, then the returned pose and inlier flags
  __syncthreads();
  for (int w = lane; w < nwords; w += 64) st_mask[w] = s_best[w];
  if (lane == 0) {
    st[0] = accepted ? acc_iters : total;
    st[1] = best;
    st[2] = refine_ok;
    for (int i = 0; i < 12; i++) st_T[i] = s_bestT[i];
  }
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
  // ---- iterations exhausted (mnIterations >= mRansacMaxIts holds whenever the loop ends without a return)
  if (lane == 0) {
    info[2] = 1;
    info[3] = total;
  }
  if (best >= minInl) {
    for (int i = lane; i < N; i += 64)
      if ((s_best[i >> 6] >> (i & 63)) & 1ull) inl_out[g_idx[i]] = 1;
    if (lane == 0) {
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T_out[r * 4 + c] = s_bestT[r * 3 + c];
        T_out[r * 4 + 3] = s_bestT[9 + r];
      }
      T_out[12] = T_out[13] = T_out[14] = 0.f;
      T_out[15] = 1.f;
      info[0] = 1;
      info[1] = best;
    }
  }
}

// diagnostics: EPnP alone on explicit correspondences (one lane)
__global__ __launch_bounds__(64, 2) __attribute__((disable_tail_calls)) void k_epnp_debug(int n, const double* pws, const double* us, EpnpCam cam, double* out13) {
  __shared__ double s_work[PNP_CHUNK * (156 + 60)];
  __shared__ double s_mtm[144], s_terms[64 * 9], s_red[64], s_Rt[12];
  __shared__ double s_Mrows[PNP_CHUNK * 4 * 24];
  const int lane = threadIdx.x;
  const LArr ut{LDS_PTR(s_work)}, L{LDS_PTR(s_work) + 156 * PNP_CHUNK};
  double e;
  if (n == 4) {   // the RANSAC minimal-set solver: lanes 0, PNP_CHUNK, 2 * PNP_CHUNK run the three variants
    double R[3][3], t[3];
    const int hyp = lane % PNP_CHUNK, variant = lane / PNP_CHUNK + 1;
    const bool active = hyp == 0 && variant <= 3;
    const LArr h_ut{LDS_PTR(s_work) + hyp}, h_L{LDS_PTR(s_work) + 156 * PNP_CHUNK + hyp};   // per-hypothesis views, as in k_pnp
    const double err = epnp_minimal(
        active, variant,
        [&](int k, double* pw, double* u) {
          for (int j = 0; j < 3; j++) pw[j] = pws[3 * k + j];
          for (int j = 0; j < 2; j++) u[j] = us[2 * k + j];
        },
        cam, R, t, h_ut, h_L, LDS_PTR(s_Mrows), 1);
    const double e1 = __shfl(err, 0), e2 = __shfl(err, PNP_CHUNK), e3 = __shfl(err, 2 * PNP_CHUNK);
    int win = 1;
    e = e1;
    if (e2 < e) { win = 2; e = e2; }
    if (e3 < e) { win = 3; e = e3; }
    if (active && variant == win) {
      for (int i = 0; i < 9; i++) s_Rt[i] = R[i / 3][i % 3];
      for (int i = 0; i < 3; i++) s_Rt[9 + i] = t[i];
    }
    __syncthreads();
  } else {
    epnp_refit_wave(n, pws, us, cam, LDS_PTR(s_Rt), ut, L, LDS_PTR(s_terms), LDS_PTR(s_red), LDS_PTR(s_mtm));
    e = s_red[63];
  }
  if (lane == 0) {
    for (int i = 0; i < 12; i++) out13[i] = s_Rt[i];
    out13[12] = e;
  }
}

int run_epnp_debug(int n, const double* Xw, const double* uv, double fx, double fy, double cx, double cy, double* R9, double* t3,
                   double* err) {
  double *d_p = nullptr, *d_u = nullptr, *d_o = nullptr;
  SD_HIP_CHECK(hipMalloc(&d_p, sizeof(double) * 3 * n));
  SD_HIP_CHECK(hipMalloc(&d_u, sizeof(double) * 2 * n));
  SD_HIP_CHECK(hipMalloc(&d_o, sizeof(double) * 13));
  SD_HIP_CHECK(hipMemcpy(d_p, Xw, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
  SD_HIP_CHECK(hipMemcpy(d_u, uv, sizeof(double) * 2 * n, hipMemcpyHostToDevice));
  EpnpCam cam = {fx, fy, cx, cy};
  hipLaunchKernelGGL(k_epnp_debug, dim3(1), dim3(64), 0, 0, n, d_p, d_u, cam, d_o);
  double out[13];
  SD_HIP_CHECK(hipMemcpy(out, d_o, sizeof(out), hipMemcpyDeviceToHost));
  for (int i = 0; i < 9; i++) R9[i] = out[i];
  for (int i = 0; i < 3; i++) t3[i] = out[9 + i];
  if (err) *err = out[12];
  (void)hipFree(d_p); (void)hipFree(d_u); (void)hipFree(d_o);
  return SD_OK;
}

int read_pnp_prof(unsigned long long* out32, int reset) {
#ifdef SD_PNP_PROF
  SD_HIP_CHECK(hipDeviceSynchronize());
  SD_HIP_CHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_pnp_prof), 32 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[32] = {};
    SD_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_pnp_prof), z, sizeof(z)));
  }
  return SD_OK;
#else
  set_error("library built without -DSD_PNP_PROF");
  return SD_ERR_INVALID_ARG;
#endif
}

int launch_pnp(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_sigma2, const PnpParams& pp,
               int n_frames, hipStream_t s) {
  hipLaunchKernelGGL(pp.min_set == 4 ? k_pnp<false> : k_pnp<true>, dim3(n_frames), dim3(64), 0, s, (cur->have_dist ? cur->d_kps_un : cur->d_kps), cur->d_nout, tb, cam, d_sigma2, pp,
                     n_frames);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

}  // namespace sd
