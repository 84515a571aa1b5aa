// Optimizer::PoseOptimization on MI355X: the motion-only bundle adjustment the reference runs after every
// SearchByProjection (src/Tracking.cc:693, 729), one wavefront per frame.
//
// Replaces (reference file:line)
//   Optimizer::PoseOptimization                                   src/Optimizer.cc:221-415
//   g2o OptimizationAlgorithmLevenberg::solve & friends           src/extra/g2o/core/optimization_algorithm_levenberg.cpp:60-187
//   g2o BlockSolver::buildSystem / BaseUnaryEdge::constructQuadraticForm / RobustKernelHuber
//                                                                 src/extra/g2o/core/block_solver.hpp:502-604, base_unary_edge.hpp:43-72,
//                                                                 robust_kernel_impl.cpp:78-91
//   g2o LinearSolverDense (Eigen::LDLT, isPositive)               src/extra/g2o/solvers/linear_solver_dense.h:65-116
//   EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose   src/extra/g2o/types/types_six_dof_expmap.{h,cpp}
//   SE3Quat::exp / operator* / map, VertexSE3Expmap::oplusImpl    src/extra/g2o/types/se3quat.h, types_six_dof_expmap.h:73-76
//
// Shape: the graph has ONE 6-DoF vertex and up to ~1000 unary edges, so an LM step is two reductions over the
// edges (robust chi2; 21 + 6 entries of H and b) around a 6x6 solve.  One frame = one workgroup of 4 waves; thread t
// owns keypoints t, t + 256, ...: it re-derives each edge from the resident arrays (undistorted keypoint, level sigma,
// matched map point) instead of storing per-edge state, accumulates in fp64, the waves combine with xor-butterflies
// and across waves through LDS in a fixed order (every thread ends with the same bits).  Thread 0 factorises
// H + lambda I in LDS (pivoted LDLT, Eigen semantics incl. isPositive); all threads replay the scalar LM logic in
// lock step.  g2o adds edges in keypoint order and sums them sequentially; the
// butterfly order differs, so this stage is compared at a tolerance (pose 1e-5, identical outlier flags), like
// ImageAlign's H.  Quirks kept: every round restarts from the frame's initial pose; edges are classified with the
// errors of the last computeActiveErrors (after a rejected trial: the rejected estimate's) -- kept as "the estimate
// at which errors were last computed" rather than per-edge error storage; Huber removed after the third round.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <cstdlib>

#include "orb_internal.h"
#include "track_internal.h"

namespace sd {

struct Se3q {
  double q[4];   // x, y, z, w
  double t[3];
};

__device__ __forceinline__ void q_normalize(double* q) {
  if (q[3] < 0)
    for (int i = 0; i < 4; i++) q[i] *= -1;
  const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}

__device__ __forceinline__ void q_from_R(const double m[3][3], double* q) {   // Eigen QuaternionBase::operator=(MatrixBase)
  double t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[2][1] - m[1][2]) * t;
    q[1] = (m[0][2] - m[2][0]) * t;
    q[2] = (m[1][0] - m[0][1]) * t;
  } else {
    // i = index of the largest diagonal entry, written without run-time indexing
    const bool i1 = m[1][1] > m[0][0];
    const bool i2 = m[2][2] > (i1 ? m[1][1] : m[0][0]);
    if (i2) {          // i = 2, j = 0, k = 1
      t = sqrt(m[2][2] - m[0][0] - m[1][1] + 1.0);
      q[2] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[1][0] - m[0][1]) * t;
      q[0] = (m[0][2] + m[2][0]) * t;
      q[1] = (m[1][2] + m[2][1]) * t;
    } else if (i1) {   // i = 1, j = 2, k = 0
      t = sqrt(m[1][1] - m[2][2] - m[0][0] + 1.0);
      q[1] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[0][2] - m[2][0]) * t;
      q[2] = (m[2][1] + m[1][2]) * t;
      q[0] = (m[0][1] + m[1][0]) * t;
    } else {           // i = 0, j = 1, k = 2
      t = sqrt(m[0][0] - m[1][1] - m[2][2] + 1.0);
      q[0] = 0.5 * t;
      t = 0.5 / t;
      q[3] = (m[2][1] - m[1][2]) * t;
      q[1] = (m[1][0] + m[0][1]) * t;
      q[2] = (m[2][0] + m[0][2]) * t;
    }
  }
}

__device__ __forceinline__ void q_mul(const double* a, const double* b, double* r) {
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z; r[3] = w;
}

__device__ __forceinline__ void q_rotate(const double* q, const double* v, double* r) {
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  for (int i = 0; i < 3; i++) uv[i] += uv[i];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; i++) r[i] = v[i] + q[3] * uv[i] + c[i];
}

__device__ __forceinline__ Se3q se3_from_Rt(const double R[3][3], const double* t) {
  Se3q s;
  q_from_R(R, s.q);
  for (int i = 0; i < 3; i++) s.t[i] = t[i];
  q_normalize(s.q);
  return s;
}

__device__ __forceinline__ Se3q se3q_mul(const Se3q& a, const Se3q& b) {
  Se3q r = a;
  double rt[3];
  q_rotate(a.q, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] += rt[i];
  q_mul(a.q, b.q, r.q);
  q_normalize(r.q);
  return r;
}

__device__ __noinline__ Se3q se3q_exp(const double* update) {   // SE3Quat::exp
  const double omega[3] = {update[0], update[1], update[2]}, upsilon[3] = {update[3], update[4], update[5]};
  const double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  const double Om[3][3] = {{0, -omega[2], omega[1]}, {omega[2], 0, -omega[0]}, {-omega[1], omega[0], 0}};
  double Om2[3][3], R[3][3], V[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Om2[i][j] = Om[i][0] * Om[0][j] + Om[i][1] * Om[1][j] + Om[i][2] * Om[2][j];
  if (theta < 0.00001) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) V[i][j] = R[i][j] = ((i == j ? 1.0 : 0.0) + Om[i][j]) + Om2[i][j];
  } else {
    const double a = sin(theta) / theta, b = (1 - cos(theta)) / (theta * theta), c = (theta - sin(theta)) / pow(theta, 3.0);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        R[i][j] = ((i == j ? 1.0 : 0.0) + a * Om[i][j]) + b * Om2[i][j];
        V[i][j] = ((i == j ? 1.0 : 0.0) + b * Om[i][j]) + c * Om2[i][j];
      }
  }
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = V[i][0] * upsilon[0] + V[i][1] * upsilon[1] + V[i][2] * upsilon[2];
  return se3_from_Rt(R, t);
}

// Eigen 3.3 LDLT (pivoting on the largest |diagonal|, sign tracking) + solve, 6x6 in LDS, one lane.
// Returns 1 if isPositive().  Same elimination as ldlt_solve6 in track_align.hip.
#define PA_(r, c) A[(r) * 6 + (c)]
__device__ __noinline__ int ldlt6_solve_sign(double* A, const double* b, double* x) {
  const int n = 6;
  int tr[6];
  double temp[6];
  int sign = 0;   // 0 zero, 1 positive semidefinite, -1 negative semidefinite, 2 indefinite
  bool found_zero_pivot = false;
  for (int k = 0; k < n; ++k) {
    int big = k;
    double best = fabs(PA_(k, k));
    for (int i = k + 1; i < n; i++)
      if (fabs(PA_(i, i)) > best) { best = fabs(PA_(i, i)); big = i; }
    tr[k] = big;
    if (k != big) {
      const int s = n - big - 1;
      for (int j = 0; j < k; j++) { double t = PA_(k, j); PA_(k, j) = PA_(big, j); PA_(big, j) = t; }
      for (int i = 0; i < s; i++) { double t = PA_(big + 1 + i, k); PA_(big + 1 + i, k) = PA_(big + 1 + i, big); PA_(big + 1 + i, big) = t; }
      { double t = PA_(k, k); PA_(k, k) = PA_(big, big); PA_(big, big) = t; }
      for (int i = k + 1; i < big; ++i) { double t = PA_(i, k); PA_(i, k) = PA_(big, i); PA_(big, i) = t; }
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = PA_(j, j) * PA_(k, j);
      double s = 0;
      for (int j = 0; j < k; j++) s += PA_(k, j) * temp[j];
      PA_(k, k) -= s;
      for (int i = 0; i < rs; i++) {
        double t = 0;
        for (int j = 0; j < k; j++) t += PA_(k + 1 + i, j) * temp[j];
        PA_(k + 1 + i, k) -= t;
      }
    }
    const double akk = PA_(k, k);
    const bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      sign = 0;
      for (int j = 0; j < n; j++) tr[j] = j;
      break;
    }
    if (valid) {
      for (int i = 0; i < rs; i++) PA_(k + 1 + i, k) /= akk;
    } else {
      found_zero_pivot = true;
    }
    if (sign == 1) { if (akk < 0) sign = 2; }
    else if (sign == -1) { if (akk > 0) sign = 2; }
    else if (sign == 0) { if (akk > 0) sign = 1; else if (akk < 0) sign = -1; }
    if (found_zero_pivot && valid) sign = 2;
  }
  if (!(sign == 1 || sign == 0)) return 0;
  double y[6];
  for (int i = 0; i < n; i++) y[i] = b[i];
  for (int k = 0; k < n; k++) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) y[i] -= PA_(i, j) * y[j];
  for (int i = 0; i < n; i++) {
    if (fabs(PA_(i, i)) > DBL_MIN) y[i] /= PA_(i, i);
    else y[i] = 0;
  }
  for (int i = n - 1; i >= 0; i--)
    for (int j = i + 1; j < n; j++) y[i] -= PA_(j, i) * y[j];
  for (int k = n - 1; k >= 0; k--) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < n; i++) x[i] = y[i];
  return 1;
}

struct PoCam { double fx, fy, cx, cy, bf; };

// One unary edge as the kernel re-derives it from the resident arrays.  The loads behind it form a dependent chain
// (match index -> map point; keypoint -> octave -> sigma), so the edge loops fetch PO_U edges with straight-line,
// index-clamped loads before touching any of them: PO_U chains in flight per thread instead of one.
constexpr int PO_U = 4;
struct PoEdge {
  int i;        // keypoint index
  int m;        // map point index, < 0: no edge in this slot
  int outl;     // current mvbOutlier flag
  float x, y, ur;
  double info;  // invSigma2 of the keypoint's octave
  double Xw[3];
};

// error of one edge at estimate T; returns chi2 (= invSigma2 * |e|^2); e[] filled
__device__ __forceinline__ double po_error(const Se3q& T, const double* Xw, double ox, double oy, double our, bool stereo, double info,
                                           const PoCam& cam, double* e, double* p) {
  double r[3];
  q_rotate(T.q, Xw, r);
  for (int i = 0; i < 3; i++) p[i] = r[i] + T.t[i];
  if (!stereo) {
    const double px = p[0] / p[2], py = p[1] / p[2];
    e[0] = ox - (px * cam.fx + cam.cx);
    e[1] = oy - (py * cam.fy + cam.cy);
    e[2] = 0;
    return e[0] * (info * e[0]) + e[1] * (info * e[1]);
  }
  const float invz = (float)(1.0 / p[2]);   // const float invz = 1.0f / trans_xyz[2]
  const double r0 = p[0] * invz * cam.fx + cam.cx, r1 = p[1] * invz * cam.fy + cam.cy, r2 = r0 - cam.bf * invz;
  e[0] = ox - r0;
  e[1] = oy - r1;
  e[2] = our - r2;
  return (e[0] * (info * e[0]) + e[1] * (info * e[1])) + e[2] * (info * e[2]);
}

// sum over the wave, every lane gets the same bits (device library's DPP reduction; the order is a fixed tree, like the
// xor-butterfly it replaces -- this stage is compared at a tolerance, see the header)
extern "C" __device__ __attribute__((const)) double __ockl_wfred_add_f64(double);
__device__ __forceinline__ double wave_sum(double v) { return __ockl_wfred_add_f64(v); }

// One frame = one workgroup of W waves (W = 1 for large batches, where the waves of other frames fill the SIMDs and the
// replicated scalar LM logic of extra waves would only cost; W = 4 for small batches, where the edge loops are the
// critical path): the edges are spread over all threads; sums are combined per wave by butterflies and across waves
// through LDS in a fixed order, so that every thread continues with the same bits.
constexpr int PO_NRED = 28;
template <int W>
struct PoRed {
  double part[W][PO_NRED];
  int cnt[W][2];
};
template <int N, int W>
__device__ __forceinline__ void block_sum(double (&v)[N], PoRed<W>& R, int wave, int lane) {
  static_assert(N <= PO_NRED, "reduction scratch too small");
#pragma unroll
  for (int k = 0; k < N; k++) v[k] = wave_sum(v[k]);
  if (W == 1) return;
  __syncthreads();   // earlier readers of R are done
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < N; k++) R.part[wave][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; k++) {
    double t = R.part[0][k];
#pragma unroll
    for (int w = 1; w < W; w++) t += R.part[w][k];
    v[k] = t;
  }
}
// a, b: wave-uniform partial counts
template <int W>
__device__ __forceinline__ void block_count2(int& a, int& b, PoRed<W>& R, int wave, int lane) {
  if (W == 1) return;
  __syncthreads();
  if (lane == 0) {
    R.cnt[wave][0] = a;
    R.cnt[wave][1] = b;
  }
  __syncthreads();
  int ta = 0, tb2 = 0;
#pragma unroll
  for (int w = 0; w < W; w++) {
    ta += R.cnt[w][0];
    tb2 += R.cnt[w][1];
  }
  a = ta;
  b = tb2;
}

// Tracking::TrackLocalMap after PoseOptimization (src/Tracking.cc:730-751): mnMatchesInliers = points of mvpMapPoints that
// are not outliers and have observations; tracked iff >= min_inliers (30).  Nothing is discarded here.
template <int W>
__device__ void tlm_tail(const TrackBuffers& tb, int f, int nkp, const int32_t* match, const uint8_t* outl, int nInitial, int min_inliers,
                         PoRed<W>& R) {
  constexpr int PO_THREADS = 64 * W;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int M = tb.max_points;
  const int32_t* obs_lo = tb.obs + (size_t)f * M;
  const int32_t* obs_hi = tb.lm_obs + (size_t)f * M - M;
  int ninl = 0, nloc = 0;
  for (int i0 = 0; i0 < nkp; i0 += PO_THREADS) {
    const int i = i0 + tid;
    bool inl = false, loc = false;
    if (i < nkp && match[i] >= 0) {
      const int m = match[i];
      loc = m >= M;
      inl = !outl[i] && (m >= M ? obs_hi : obs_lo)[m] > 0;
    }
    ninl += __popcll(__ballot(inl));
    nloc += __popcll(__ballot(loc));
  }
  block_count2(ninl, nloc, R, wave, lane);
  if (tid == 0) {
    tb.tl_info[(size_t)f * 4 + 0] = ninl >= min_inliers ? 2 : 1;
    tb.tl_info[(size_t)f * 4 + 1] = nInitial;
    tb.tl_info[(size_t)f * 4 + 2] = ninl;
    tb.tl_info[(size_t)f * 4 + 3] = nloc;
  }
}

// source 0: map points of the frame-to-frame match (tb.cur_match -> tb.Xw); 1: of the local-map search (tb.lm_match -> tb.lm_Xw)
template <int W>
__global__ __launch_bounds__(64 * W) void k_pose_opt(const sd_keypoint* __restrict__ kps_all, const int32_t* __restrict__ nkp_all, TrackBuffers tb,
                                                TrackCam tcam, const float* __restrict__ inv_sigma2, int source, int n_frames,
                                                int min_matches, int min_inliers) {
  __shared__ double s_A[36], s_b[6], s_x[8];
  __shared__ int s_ok;
  constexpr int PO_THREADS = 64 * W;
  __shared__ PoRed<W> s_red;
  // a thread's keypoints that carry a map point, in ascending order (its edges): [position][thread]; the edge loops walk
  // these lists instead of all keypoints (a frame-to-frame match gives ~ a quarter of the keypoints a point)
  constexpr int PO_SLOTS = 2048 / PO_THREADS;   // keypoint capacity of the tracker (<= 2048) / threads
  __shared__ uint16_t s_idx[PO_SLOTS][PO_THREADS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
    __syncthreads();
    const int cap = tb.kp_cap, M = tb.max_points;
    const int fc = tb.cur_bcast >= 0 ? tb.cur_bcast : f;   // current-frame slot (TrackBuffers::cur_bcast)
    const sd_keypoint* kps = kps_all + (size_t)fc * cap;
    const int nkp = min(nkp_all[fc], cap);
    int32_t* match = (source == 0 ? tb.cur_match : source == 1 ? tb.lm_match : tb.un_match) + (size_t)f * cap;
    // source 2: indices >= M address the local-map arrays (XW below)
    const double* Xw_lo = (source == 1 ? tb.lm_Xw : tb.Xw) + (size_t)f * M * 3;
    const double* Xw_hi = tb.lm_Xw + (size_t)f * M * 3 - (size_t)M * 3;
#define XW(m, k) (((m) >= M ? Xw_hi : Xw_lo)[(size_t)(m) * 3 + (k)])
    const float* uright = tb.uright + (size_t)fc * cap;
    uint8_t* outl = tb.po_outlier + (size_t)f * cap;
    double* T_out = tb.po_T + (size_t)f * 16;
    int32_t* info = tb.po_info + (size_t)f * 8;
    const double* T_in = tb.Tcur + (size_t)f * 16;
    const PoCam cam = {(double)tcam.ffx, (double)tcam.ffy, (double)tcam.fcx, (double)tcam.fcy, (double)tcam.bf};
    const double deltaMono = (double)(float)sqrt(5.991), deltaStereo = (double)(float)sqrt(7.815);   // const float delta = sqrt(..)
    const float chi2Mono = 5.991f, chi2Stereo = 7.815f;
    int my_edges = 0;   // entries of this thread's list (set below, once `match` is final)
    auto load_edges = [&](int k0, PoEdge (&E)[PO_U]) {
      int ic[PO_U];
#pragma unroll
      for (int u = 0; u < PO_U; u++) {
        const int k = k0 + u;
        const bool live = k < my_edges;
        ic[u] = live ? (int)s_idx[min(k, PO_SLOTS - 1)][tid] : 0;   // slot 0 of the arrays always exists: safe dummy address
        E[u].i = ic[u];
      }
#pragma unroll
      for (int u = 0; u < PO_U; u++) {
        const int mm = match[ic[u]];
        E[u].m = (k0 + u < my_edges) ? mm : -1;
        E[u].outl = outl[ic[u]];
      }
      int oct[PO_U];
#pragma unroll
      for (int u = 0; u < PO_U; u++) {
        const int mc = max(E[u].m, 0);
        const sd_keypoint kp = kps[ic[u]];
        E[u].x = kp.x;
        E[u].y = kp.y;
        oct[u] = kp.octave;
        E[u].ur = uright[ic[u]];
        E[u].Xw[0] = XW(mc, 0);
        E[u].Xw[1] = XW(mc, 1);
        E[u].Xw[2] = XW(mc, 2);
      }
#pragma unroll
      for (int u = 0; u < PO_U; u++) E[u].info = (double)inv_sigma2[oct[u]];
    };

    if (source == 2) {
      // mvpMapPoints after Tracking::SearchLocalPoints: a local match replaces whatever the keypoint held (it is only made
      // where the keypoint held nothing or a point without observations, src/ORBmatcher.cc:81-83, :114)
      const int32_t* fm = tb.cur_match + (size_t)f * cap;
      const int32_t* lm = tb.lm_match + (size_t)f * cap;
      for (int i = tid; i < cap; i += PO_THREADS) match[i] = lm[i] >= 0 ? lm[i] + M : fm[i];
      __syncthreads();
    }
    // r3: the edges of a wave's keypoints are dealt to its 64 lanes ROUND-ROBIN (ballot-ordered compaction: edge p of the wave goes to
    // lane p % 64, slot p / 64).  Each lane used to keep the matched ones among ITS keypoints (i = tid + 64 k): with a quarter of the
    // keypoints matched the longest list of a wave was about twice the mean, and the edge loops run to the longest list.
    int nInitial = 0;
    {
      const unsigned long long lt_ = lane == 0 ? 0ull : (~0ull >> (64 - lane));
      int wbase = 0;   // edges of this wave so far
      for (int i0 = 0; i0 < nkp; i0 += PO_THREADS) {
        const int i = i0 + tid;
        const bool has = i < nkp && match[i] >= 0;
        const unsigned long long bal = __ballot(has);
        if (has) {
          const int p = wbase + __popcll(bal & lt_);
          s_idx[p >> 6][(wave << 6) + (p & 63)] = (uint16_t)i;
        }
        wbase += __popcll(bal);
      }
      nInitial = wbase;
      my_edges = (wbase + 63 - lane) >> 6;   // positions p < wbase with p % 64 == lane
      __syncthreads();   // the lists are written across lanes (one wave: a fence would do; with four waves block_count2 syncs anyway)
    }
    int max_edges = my_edges;   // longest list of the workgroup: uniform trip count of the edge loops
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) max_edges = max(max_edges, __shfl_xor(max_edges, o));
    {
      int neg = -max_edges;   // block_count2 sums; a maximum over waves is needed: take it from the per-wave slots below
      block_count2(nInitial, neg, s_red, wave, lane);
      if (W > 1) {
        max_edges = 0;
#pragma unroll
        for (int w = 0; w < W; w++) max_edges = max(max_edges, -s_red.cnt[w][1]);
      }
    }
    for (int i = tid; i < cap; i += PO_THREADS) outl[i] = 0;
    if (tid < 16) T_out[tid] = T_in[tid];
    if (tid == 0) {
      for (int k = 0; k < 8; k++) info[k] = 0;
      info[0] = nInitial;
    }
    // TrackWithMotionModel: "Not enough matches, tracking failed" returns before PoseOptimization (src/Tracking.cc:691-694)
    const int nm_search = min_matches > 0 ? tb.n_matches[f] : 0;
    if (min_matches > 0 && nm_search < min_matches) {
      if (tid == 0) {
        tb.tw_info[(size_t)f * 4 + 0] = 0;
        tb.tw_info[(size_t)f * 4 + 1] = nm_search;
        tb.tw_info[(size_t)f * 4 + 2] = 0;
      }
      continue;
    }
    if (nInitial < 3) {
      if (source == 2) {   // PoseOptimization returns 0 and leaves pose and flags alone; TrackLocalMap still counts
        __syncthreads();
        tlm_tail(tb, f, nkp, match, outl, nInitial, min_inliers, s_red);
      }
      continue;
    }
    double R0[3][3], t0[3];
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) R0[r][c] = T_in[c * 4 + r];
      t0[r] = T_in[12 + r];
    }
    Se3q est = se3_from_Rt(R0, t0), est_err = est;
    int nBad = 0, its_total = 0, trials_total = 0, rounds = 0;
    bool robust = true;
    for (int round = 0; round < 4; round++) {
      est = se3_from_Rt(R0, t0);   // vSE3->setEstimate(Converter::toSE3Quat(pFrame->GetPose()))
      // ---------------- optimizer.optimize(10)
      int n_active = 0;
      for (int i0 = 0; i0 < nkp; i0 += PO_THREADS) {
        const int i = i0 + tid;
        n_active += __popcll(__ballot(i < nkp && match[i] >= 0 && !outl[i]));
      }
      {
        int zero = 0;
        block_count2(n_active, zero, s_red, wave, lane);
      }
      double lambda = -1., ni = 2.;
      int nBadLM = 0;
      for (int it = 0; it < 10 && n_active > 0; it++) {
        // computeActiveErrors + activeRobustChi2 + buildSystem at `est`
        double H[21], b[6], chi = 0;
#pragma unroll
        for (int k = 0; k < 21; k++) H[k] = 0;
#pragma unroll
        for (int k = 0; k < 6; k++) b[k] = 0;
        for (int k0 = 0; k0 < max_edges; k0 += PO_U) {
         PoEdge E[PO_U];
         load_edges(k0, E);
#pragma unroll
         for (int u = 0; u < PO_U; u++) {
          const PoEdge& ed = E[u];
          if (ed.m < 0 || ed.outl) continue;
          const float ur = ed.ur;
          const bool stereo = !(ur < 0);
          const double infoe = ed.info;
          const double* Xw = ed.Xw;
          double e[3], p[3];
          const double c2 = po_error(est, Xw, ed.x, ed.y, ur, stereo, infoe, cam, e, p);
          double rho1 = 1.0;
          if (robust) {
            const double delta = stereo ? deltaStereo : deltaMono, dsqr = delta * delta;
            if (c2 <= dsqr) {
              chi += c2;
            } else {
              const double sqrte = sqrt(c2);
              chi += 2 * sqrte * delta - dsqr;
              rho1 = delta / sqrte;
            }
          } else {
            chi += c2;
          }
          const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
          double J[3][6];
          J[0][0] = x * y * invz_2 * cam.fx;
          J[0][1] = -(1 + (x * x * invz_2)) * cam.fx;
          J[0][2] = y * invz * cam.fx;
          J[0][3] = -invz * cam.fx;
          J[0][4] = 0;
          J[0][5] = x * invz_2 * cam.fx;
          J[1][0] = (1 + y * y * invz_2) * cam.fy;
          J[1][1] = -x * y * invz_2 * cam.fy;
          J[1][2] = -x * invz * cam.fy;
          J[1][3] = 0;
          J[1][4] = -invz * cam.fy;
          J[1][5] = y * invz_2 * cam.fy;
          J[2][0] = stereo ? J[0][0] - cam.bf * y * invz_2 : 0.0;
          J[2][1] = stereo ? J[0][1] + cam.bf * x * invz_2 : 0.0;
          J[2][2] = stereo ? J[0][2] : 0.0;
          J[2][3] = stereo ? J[0][3] : 0.0;
          J[2][4] = 0;
          J[2][5] = stereo ? J[0][5] - cam.bf * invz_2 : 0.0;
          const double wi = rho1 * infoe;
          int q = 0;
#pragma unroll
          for (int a = 0; a < 6; a++) {
            b[a] -= rho1 * ((J[0][a] * (infoe * e[0]) + J[1][a] * (infoe * e[1])) + J[2][a] * (infoe * e[2]));
#pragma unroll
            for (int c = a; c < 6; c++) H[q++] += (J[0][a] * (wi * J[0][c]) + J[1][a] * (wi * J[1][c])) + J[2][a] * (wi * J[2][c]);
          }
         }
        }
        est_err = est;
        {
          double r[28];
#pragma unroll
          for (int k = 0; k < 21; k++) r[k] = H[k];
#pragma unroll
          for (int k = 0; k < 6; k++) r[21 + k] = b[k];
          r[27] = chi;
          block_sum(r, s_red, wave, lane);
#pragma unroll
          for (int k = 0; k < 21; k++) H[k] = r[k];
#pragma unroll
          for (int k = 0; k < 6; k++) b[k] = r[21 + k];
          chi = r[27];
        }
        double currentChi = chi, tempChi = currentChi;
        const double iniChi = currentChi;
        if (it == 0) {
          double maxDiagonal = 0.;
          int q = 0;
          for (int a = 0; a < 6; a++) {
            maxDiagonal = fmax(fabs(H[q]), maxDiagonal);
            q += 6 - a;
          }
          lambda = 1e-5 * maxDiagonal;
          ni = 2;
          nBadLM = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
          const Se3q backup = est;
          __syncthreads();
          if (tid == 0) {
            int q = 0;
            for (int a = 0; a < 6; a++)
              for (int c = a; c < 6; c++) {
                s_A[a * 6 + c] = H[q];
                s_A[c * 6 + a] = H[q];
                q++;
              }
            for (int a = 0; a < 6; a++) {
              s_A[a * 6 + a] += lambda;
              s_b[a] = b[a];
            }
            s_ok = ldlt6_solve_sign(s_A, s_b, s_x);   // on failure s_x keeps the previous solution, like _solver->x()
          }
          __syncthreads();
          const bool ok2 = s_ok != 0;
          double x[6];
          for (int k = 0; k < 6; k++) x[k] = s_x[k];
          est = se3q_mul(se3q_exp(x), est);
          // computeActiveErrors + activeRobustChi2 at the trial estimate
          double c = 0;
          for (int k0 = 0; k0 < max_edges; k0 += PO_U) {
            PoEdge E[PO_U];
            load_edges(k0, E);
#pragma unroll
            for (int u = 0; u < PO_U; u++) {
              const PoEdge& ed = E[u];
              if (ed.m < 0 || ed.outl) continue;
              const bool stereo = !(ed.ur < 0);
              double e[3], p[3];
              const double c2 = po_error(est, ed.Xw, ed.x, ed.y, ed.ur, stereo, ed.info, cam, e, p);
              if (robust) {
                const double delta = stereo ? deltaStereo : deltaMono, dsqr = delta * delta;
                c += (c2 <= dsqr) ? c2 : (2 * sqrt(c2) * delta - dsqr);
              } else {
                c += c2;
              }
            }
          }
          est_err = est;
          {
            double r[1] = {c};
            block_sum(r, s_red, wave, lane);
            tempChi = r[0];
          }
          if (!ok2) tempChi = DBL_MAX;
          rho = (currentChi - tempChi);
          double scale = 0.;
          for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
          scale += 1e-3;
          rho /= scale;
          if (rho > 0 && isfinite(tempChi)) {
            double alpha = 1. - pow((2 * rho - 1), 3.0);
            alpha = fmin(alpha, 2. / 3.);
            const double scaleFactor = fmax(1. / 3., alpha);
            lambda *= scaleFactor;
            ni = 2;
            currentChi = tempChi;
          } else {
            lambda *= ni;
            ni *= 2;
            est = backup;
          }
          qmax++;
          trials_total++;
        } while (rho < 0 && qmax < 10);
        its_total++;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBadLM++;
        else nBadLM = 0;
        if (nBadLM >= 3) break;
      }
      rounds++;
      // ---------------- classify (src/Optimizer.cc:353-398)
      int bad = 0;
      for (int k0 = 0; k0 < max_edges; k0 += PO_U) {   // uniform trip count (ballots inside)
        PoEdge E[PO_U];
        load_edges(k0, E);
#pragma unroll
        for (int u = 0; u < PO_U; u++) {
          const PoEdge& ed = E[u];
          const int i = ed.i;
          bool isbad = false;
          if (ed.m >= 0) {
            const bool stereo = !(ed.ur < 0);
            double e[3], p[3];
            // outliers: e->computeError() at the current estimate; the others keep the errors of the last computeActiveErrors
            const float chi2 = (float)po_error(ed.outl ? est : est_err, ed.Xw, ed.x, ed.y, ed.ur, stereo, ed.info, cam, e, p);
            isbad = chi2 > (stereo ? chi2Stereo : chi2Mono);
            outl[i] = isbad ? 1 : 0;
          }
          bad += __popcll(__ballot(isbad));
        }
      }
      {
        int zero = 0;
        block_count2(bad, zero, s_red, wave, lane);
      }
      nBad = bad;
      if (round == 2) robust = false;
      if (nInitial < 10) break;   // optimizer.edges().size() < 10
    }
    // ---------------- result
    if (tid == 0) {
      const double* q = est.q;
      const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
      const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
      const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
      const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
      const double Ro[3][3] = {{1 - (tyy + tzz), txy - twz, txz + twy}, {txy + twz, 1 - (txx + tzz), tyz - twx}, {txz - twy, tyz + twx, 1 - (txx + tyy)}};
      for (int i = 0; i < 16; i++) T_out[i] = (i % 5 == 0) ? 1.0 : 0.0;
      for (int r = 0; r < 3; r++) {
        for (int c = 0; c < 3; c++) T_out[c * 4 + r] = Ro[r][c];
        T_out[12 + r] = est.t[r];
      }
      info[1] = nBad;
      info[2] = rounds;
      info[3] = its_total;
      info[4] = trials_total;
      info[5] = nInitial - nBad;
    }
    if (source == 2) {
      __syncthreads();
      tlm_tail(tb, f, nkp, match, outl, nInitial, min_inliers, s_red);
      if (tid < 16) tb.Tcur[(size_t)f * 16 + tid] = T_out[tid];
    } else if (min_matches > 0) {
      // "Discard outliers" (src/Tracking.cc:699-711): outliers lose their map point and their flag; the survivors whose
      // point has observations count towards nmatchesMap.  The optimised pose is the frame's pose (pFrame->SetPose).
      __syncthreads();   // T_out written by lane 0
      const int32_t* obs = (source == 0 ? tb.obs : tb.lm_obs) + (size_t)f * M;
      int ndrop = 0, nmap = 0;
      for (int i0 = 0; i0 < nkp; i0 += PO_THREADS) {
        const int i = i0 + tid;
        bool drop = false, inmap = false;
        if (i < nkp && match[i] >= 0) {
          if (outl[i]) {
            drop = true;
            match[i] = -1;
            outl[i] = 0;
          } else {
            inmap = obs[match[i]] > 0;
          }
        }
        ndrop += __popcll(__ballot(drop));
        nmap += __popcll(__ballot(inmap));
      }
      block_count2(ndrop, nmap, s_red, wave, lane);
      const int nmatches = nm_search - ndrop;
      if (tid < 16) tb.Tcur[(size_t)f * 16 + tid] = T_out[tid];
      if (tid == 0) {
        tb.tw_info[(size_t)f * 4 + 0] = nmap >= min_inliers ? 2 : 1;
        tb.tw_info[(size_t)f * 4 + 1] = nmatches;
        tb.tw_info[(size_t)f * 4 + 2] = nmap;
      }
    }
  }
#undef XW
}

int launch_pose_opt(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_inv_sigma2, int source, int n_frames,
                    hipStream_t s, int min_matches, int min_inliers) {
  // waves per frame by batch size: option "track.poseopt_waves" = 1 | 4 overrides (tests, experiments)
  const int forced = opt(OPT_POSEOPT_WAVES);
  const int waves = forced ? forced : (n_frames <= 256 ? 4 : 1);
  const sd_keypoint* kps = cur->have_dist ? cur->d_kps_un : cur->d_kps;
  if (waves == 4)
    hipLaunchKernelGGL(k_pose_opt<4>, dim3(n_frames), dim3(256), 0, s, kps, cur->d_nout, tb, cam, d_inv_sigma2, source, n_frames, min_matches,
                       min_inliers);
  else
    hipLaunchKernelGGL(k_pose_opt<1>, dim3(n_frames), dim3(64), 0, s, kps, cur->d_nout, tb, cam, d_inv_sigma2, source, n_frames, min_matches,
                       min_inliers);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

}  // namespace sd
