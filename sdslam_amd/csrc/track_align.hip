// ImageAlign on MI355X: sparse direct (inverse-compositional) photometric alignment, one
// workgroup per frame pair, persistent over pyramid levels and Gauss-Newton iterations.
//
// Replaces SD_SLAM::ImageAlign::ComputePose / Optimize / ComputeResiduals / PrecomputePatches
// (reference src/ImageAlign.cc:45-421) and Exp/RotationExp (:473-517).
//
// Work split (one thread per point, <= 300 points x 16 patch pixels; details at k_align):
//   * a thread keeps its point's reference patch values and image gradients (dx, dy) in registers for the whole level
//     -- the reference's 230 KB fp64 jacobian_cache_ is never materialised;
//   * H (21 upper entries) and Jres (6) are formed per point from three gradient sums and combined with a fixed-shape
//     wave reduction + LDS tree (deterministic run to run);
//   * chi2 is accumulated in FLOAT in point/pixel order by one wave from per-pixel squares in LDS,
//     i.e. bit-identical to the reference's sequential `chi2 += res*res` (src/ImageAlign.cc:298,341),
//     so the accept / stop decisions (`new_chi2 > chi2_`, `> 0.99*chi2_`) do not flip;
//   * one lane solves the 6x6 system (pivoted LDLT, Eigen 3.3 semantics) beside that chain.
// Reproduced quirks: sticky visibility flags, sticky stop_/chi2_, fx on both
// Jacobian rows, float chi2, double-then-float bilinear weights.  Images are read from the
// padded pyramids the extractor left resident in HBM (levels 4,3,2 only).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "orb_internal.h"
#include "track_internal.h"

namespace sd {

#ifdef SD_PNP_PROF   // phase timers (tools/prof_pnp.py prints them): cycles of thread 0, summed over frames
__device__ unsigned long long g_align_prof[16];
#define APROF_DECL long long _pt = clock64()
#define APROF(i)                                                                               \
  do {                                                                                         \
    long long _n = clock64();                                                                  \
    if (threadIdx.x == 0) atomicAdd(&g_align_prof[i], (unsigned long long)(_n - _pt));         \
    _pt = _n;                                                                                  \
  } while (0)
#else
#define APROF_DECL
#define APROF(i)
#endif

#define AL_MAXP 300

extern "C" __device__ __attribute__((const)) double __ockl_wfred_add_f64(double);
extern "C" __device__ __attribute__((const)) int __ockl_wfred_add_i32(int);

// The serial pieces of an iteration (6 x 6 LDLT solve, Exp, 4 x 4 products) run on ONE lane beside 319 lanes whose point
// data must stay in registers: they work on LDS-resident operands through address-space-3 pointers with rolled loops, so
// they need a handful of VGPRs (unrolled into registers they cost > 128 and forced the point data out to scratch memory --
// 430 MB of spill writes per 1024-frame launch).  Their latency hides under the float chi2 chain of wave 0.
typedef __attribute__((address_space(3))) double ldsd;
typedef __attribute__((address_space(3))) int ldsi;
#define LDSD(x) ((ldsd*)(x))
#define NOUNROLL _Pragma("clang loop unroll(disable)")

// r = a * b, row-major 4 x 4 (r must not alias a or b); each entry is ((a0 b0 + a1 b1) + a2 b2) + a3 b3 from 0.0
__device__ __forceinline__ void m4_mul_lds(const ldsd* a, const ldsd* b, ldsd* r) {
  NOUNROLL for (int i = 0; i < 4; i++) {
    NOUNROLL for (int j = 0; j < 4; j++) {
      double s = 0;
      NOUNROLL for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
      r[i * 4 + j] = s;
    }
  }
}

// the same product by 16 lanes (lane l < 16 computes entry l): every entry is the same four multiply-adds in the same order
__device__ __forceinline__ void m4_mul_lds16(const ldsd* a, const ldsd* b, ldsd* r, int l) {
  if (l < 16) {
    const int i = l >> 2, j = l & 3;
    double s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
    r[l] = s;
  }
}

// Eigen 3.3 LDLT<Matrix6d>::solve (pivoting on the largest |diagonal|, pseudo-inverse of D); A row-major 6 x 6, everything
// in LDS (tmp: 12 doubles, tr: 6 ints), single lane
#define A_(r, c) A[(r) * 6 + (c)]
__device__ __forceinline__ void ldlt_solve6(ldsd* A, const ldsd* b, ldsd* x, ldsd* tmp, ldsi* tr) {
  const int n = 6;
  ldsd* temp = tmp;
  ldsd* d = tmp + 6;
  bool done = false;
  for (int k = 0; k < n && !done; ++k) {
    int big = k;
    double best = fabs(A_(k, k));
    for (int i = k + 1; i < n; i++) {
      const double v = fabs(A_(i, i));
      if (v > best) { best = v; big = i; }
    }
    tr[k] = big;
    if (k != big) {
      int s = n - big - 1;
      for (int j = 0; j < k; j++) { double t = A_(k, j); A_(k, j) = A_(big, j); A_(big, j) = t; }
      for (int i = 0; i < s; i++) { double t = A_(big + 1 + i, k); A_(big + 1 + i, k) = A_(big + 1 + i, big); A_(big + 1 + i, big) = t; }
      { double t = A_(k, k); A_(k, k) = A_(big, big); A_(big, big) = t; }
      for (int i = k + 1; i < big; ++i) { double t = A_(i, k); A_(i, k) = A_(big, i); A_(big, i) = t; }
    }
    int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = A_(j, j) * A_(k, j);
      double s = 0;
      for (int j = 0; j < k; j++) s += A_(k, j) * temp[j];
      A_(k, k) -= s;
      for (int i = 0; i < rs; i++) {
        double t = 0;
        for (int j = 0; j < k; j++) t += A_(k + 1 + i, j) * temp[j];
        A_(k + 1 + i, k) -= t;
      }
    }
    double akk = A_(k, k);
    bool valid = fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      for (int j = 0; j < n; j++) tr[j] = j;
      done = true;
    } else if (rs > 0 && valid) {
      for (int i = 0; i < rs; i++) A_(k + 1 + i, k) /= akk;
    }
  }
  for (int i = 0; i < n; i++) d[i] = b[i];
  for (int k = 0; k < n; k++) {
    const int t_ = tr[k];
    if (t_ != k) { double t = d[k]; d[k] = d[t_]; d[t_] = t; }
  }
  for (int i = 0; i < n; i++) {
    double di = d[i];
    for (int j = 0; j < i; j++) di -= A_(i, j) * d[j];
    d[i] = di;
  }
  const double tol = 2.2250738585072014e-308;
  for (int i = 0; i < n; i++) {
    const double aii = A_(i, i);
    if (fabs(aii) > tol) d[i] /= aii;
    else d[i] = 0;
  }
  for (int i = n - 1; i >= 0; i--) {
    double di = d[i];
    for (int j = i + 1; j < n; j++) di -= A_(j, i) * d[j];
    d[i] = di;
  }
  for (int k = n - 1; k >= 0; k--) {
    const int t_ = tr[k];
    if (t_ != k) { double t = d[k]; d[k] = d[t_]; d[t_] = t; }
  }
  for (int i = 0; i < n; i++) x[i] = d[i];
}
#undef A_

// ImageAlign::Exp (translation-first twist, src/ImageAlign.cc:473-517) -> row-major 4 x 4 in LDS.  upd = {upsilon, omega};
// tmp: 27 doubles of LDS (Omega, Omega^2, V)
__device__ __forceinline__ void se3_exp(const ldsd* upd, ldsd* res, ldsd* tmp) {
  const double o0 = upd[3], o1 = upd[4], o2 = upd[5];
  double theta = sqrt(o0 * o0 + o1 * o1 + o2 * o2);
  double half_theta = 0.5 * theta;
  double imag_factor;
  double real_factor = cos(half_theta);
  if (theta < 1e-10) {
    double theta_sq = theta * theta;
    double theta_po4 = theta_sq * theta_sq;
    imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
  } else {
    imag_factor = sin(half_theta) / theta;
  }
  NOUNROLL for (int i = 0; i < 16; i++) res[i] = (i % 5 == 0) ? 1.0 : 0.0;
  {
    const double qw = real_factor, qx = imag_factor * o0, qy = imag_factor * o1, qz = imag_factor * o2;
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    res[0] = 1 - (tyy + tzz); res[1] = txy - twz; res[2] = txz + twy;
    res[4] = txy + twz; res[5] = 1 - (txx + tzz); res[6] = tyz - twx;
    res[8] = txz - twy; res[9] = tyz + twx; res[10] = 1 - (txx + tyy);
  }
  ldsd* Om = tmp;
  ldsd* Om2 = tmp + 9;
  ldsd* V = tmp + 18;
  Om[0] = 0; Om[1] = -o2; Om[2] = o1;
  Om[3] = o2; Om[4] = 0; Om[5] = -o0;
  Om[6] = -o1; Om[7] = o0; Om[8] = 0;
  if (theta < 1e-10) {
    NOUNROLL for (int i = 0; i < 3; i++) {
      NOUNROLL for (int j = 0; j < 3; j++) V[i * 3 + j] = res[i * 4 + j];
    }
  } else {
    NOUNROLL for (int i = 0; i < 3; i++) {
      NOUNROLL for (int j = 0; j < 3; j++) {
        double s = 0;
        NOUNROLL for (int k = 0; k < 3; k++) s += Om[i * 3 + k] * Om[k * 3 + j];
        Om2[i * 3 + j] = s;
      }
    }
    double theta_sq = theta * theta;
    double c1 = (1 - cos(theta)) / (theta_sq);
    double c2 = (theta - sin(theta)) / (theta_sq * theta);
    NOUNROLL for (int i = 0; i < 3; i++) {
      NOUNROLL for (int j = 0; j < 3; j++) V[i * 3 + j] = ((i == j ? 1.0 : 0.0) + c1 * Om[i * 3 + j]) + c2 * Om2[i * 3 + j];
    }
  }
  const double u0 = upd[0], u1 = upd[1], u2 = upd[2];
  NOUNROLL for (int i = 0; i < 3; i++) res[i * 4 + 3] = V[i * 3] * u0 + V[i * 3 + 1] * u1 + V[i * 3 + 2] * u2;
}

// The serial step of an iteration: x = H.ldlt().solve(Jres), E = Exp(-x); returns AbsMax(x).  A FUNCTION CALL on purpose: its
// register needs (the unrolled LDLT) stay out of the kernel's allocation, and what the caller has to preserve around the call
// is saved inside the one-lane branch the call sits in -- as inline code the same registers were spilled by every lane, every
// iteration (350 MB of scratch writes per 1024-frame launch).
__device__ __noinline__ double solve_and_exp(ldsd* H, const ldsd* b, ldsd* x, ldsd* nx, ldsd* E, ldsd* tmp, ldsi* tr) {
  ldlt_solve6(H, b, x, tmp, tr);
  double mx = -1;
  NOUNROLL for (int i = 0; i < 6; i++) {
    const double xi = x[i];
    nx[i] = -xi;
    if (fabs(xi) > mx) mx = fabs(xi);
  }
  se3_exp(nx, E, tmp);
  return mx;
}

// One THREAD per point (its 16 patch pixels), AL_PT_THREADS = 320 threads per frame pair.
//
// What a thread keeps in registers for a whole level is small: the 7 x 8 BYTES of the reference image around its point
// (14 dwords) with the four bilinear weights -- from which patch value, dx and dy of a pixel are re-derived with the
// reference's own float expressions whenever they are needed --, the point's reference-frame coordinates, and three sums
// a = S dx^2, b = S dx dy, c = S dy^2 over the 16 pixels.  The reference accumulates H += J J^T and Jres -= J res pixel by
// pixel with J = (dx J0 + dy J1) fx scale, where J0 / J1 (the two rows of Jacobian3DToPlane) belong to the POINT; summed
// over a point's pixels that is
//     H_point = (fx scale)^2 (a J0 J0^T + b (J0 J1^T + J1 J0^T) + c J1 J1^T),   Jres_point = -(fx scale)(J0 S dx res + J1 S dy res).
// H_point does not change during a level, so H is reduced ONCE per level over the points with a Jacobian; an iteration
// only reduces Jres (6 sums) and, when a point with a Jacobian has left the image at the trial pose (rare), the H_point
// terms to take out again.  Nothing per pixel lives in fp64, and the kernel's hot loop touches no scratch memory (the
// first layout -- 19 pixel slots per thread, 27 fp64 accumulators, patch / gradient floats in registers -- spilled ~290
// VGPRs: 413 MB of scratch writes per 1024-frame launch).  H and Jres are sums of the same terms as the
// reference's in a different order (fp64 rounding level; the reference's order is per pixel); everything that DECIDES --
// the float residuals, the float chi2 chain in the reference's order, the comparisons -- is operation for operation the
// reference's.
//
// Per iteration: every thread projects its point at the trial pose, fetches the 5 x 5 pixel neighbourhood (10 dwords in
// flight at once), writes its 16 squared residuals to LDS and its Jres contribution into a wave reduction; then wave 0
// runs the float chi2 chain while lane 0 of wave 1 solves the 6 x 6 system and prepares the candidate update, so the
// LDLT + exp latency hides under the chain; thread 0 takes the accept / stop decisions and forms the next trial pose.
#define AL_PT_THREADS 320
#define AL_PT_WAVES (AL_PT_THREADS / 64)
static_assert(AL_PT_THREADS >= AL_MAXP, "one thread per point");

template <int MINW>   // waves per SIMD the register budget is set for (MINW x 4 / 5 workgroups per CU)
__global__ __launch_bounds__(AL_PT_THREADS, MINW) void k_align(const OrbPlan* __restrict__ P, const uint8_t* __restrict__ pyr_cur,
                                                         const uint8_t* __restrict__ pyr_ref, TrackBuffers tb, TrackCam cam,
                                                         const float* __restrict__ inv_sf, const float* __restrict__ sf, int mode,
                                                         int n_frames) {
  __shared__ double s_pts[AL_MAXP * 3];
  __shared__ __attribute__((aligned(16))) float s_chi[AL_MAXP * 16];
  __shared__ double s_red[AL_PT_WAVES][28];   // per wave: [0..20] H terms to take out, [21..26] Jres, per iteration
  __shared__ double s_redH[AL_PT_WAVES][21];  // per wave: H of the level's points with a Jacobian
  __shared__ double s_Hlvl[21];
  __shared__ int s_delta[AL_PT_WAVES];        // wave has H terms to take out this iteration
  __shared__ double s_last[16], s_se3[16], s_pose[16], s_bk[16], s_cand[16], s_posec[16];
  __shared__ double s_H[36], s_b[6], s_x[6], s_nx[6], s_E[16];
  __shared__ double s_tmp[32];   // scratchpad of the serial solver lane
  __shared__ int s_tr[8];
  __shared__ double s_mx;
  __shared__ int s_cnt[AL_PT_WAVES];
  __shared__ int s_ctrl[4];   // [0] break flag, [2] level-loop exit
  __shared__ int s_iters[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f = blockIdx.x;   // one workgroup per frame pair
  if (f >= n_frames) return;
  {
  const int M = tb.max_points;
  const uint8_t* valid = tb.valid + (size_t)f * M;
  const double* Xw = tb.Xw + (size_t)f * M * 3;
  const int n_last = min(tb.n_last[f], M);
  const int max_pts = (mode == 2 || mode == 3) ? 100 : 300;

  APROF_DECL;
  // ---- gather the first max_pts valid world points, in index order (src/ImageAlign.cc:62-72)
  int running = 0;
  for (int base = 0; base < n_last && running < max_pts; base += AL_PT_THREADS) {
    int i = base + tid;
    bool fl = i < n_last && valid[i] != 0;
    unsigned long long m = __ballot(fl);
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; w++) off += s_cnt[w];
    int pos = off + __popcll(m & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
    if (fl && pos < max_pts) {
      s_pts[pos * 3 + 0] = Xw[(size_t)i * 3 + 0];
      s_pts[pos * 3 + 1] = Xw[(size_t)i * 3 + 1];
      s_pts[pos * 3 + 2] = Xw[(size_t)i * 3 + 2];
    }
    for (int w = 0; w < AL_PT_WAVES; w++) running += s_cnt[w];
    __syncthreads();
  }
  const int npts = min(running, max_pts);

  double* out_T = tb.Tcur + (size_t)f * 16;
  const double* prior_T = tb.Tprior + (size_t)f * 16;
  if (P->nlevels <= 4 || npts == 0) {   // "Not enough pyramid levels" / "No points to track!"
    if (tid == 0) {
      for (int i = 0; i < 16; i++) out_T[i] = prior_T[i];
      tb.al_ok[f] = 0;
      tb.al_err[f] = 1e10;
      tb.al_chi2[f] = 1e10;
      for (int l = 0; l < 16; l++) tb.al_iters[(size_t)f * 16 + l] = 0;
    }
    return;
  }
  if (tid == 0) {
    // column-major in HBM (Eigen::Matrix4d::data()) -> row-major working copies (s_E: the prior, s_cand: the inverse)
    NOUNROLL for (int c = 0; c < 4; c++) {
      NOUNROLL for (int r = 0; r < 4; r++) {
        s_last[r * 4 + c] = tb.Tref[(size_t)f * 16 + c * 4 + r];
        s_E[r * 4 + c] = prior_T[c * 4 + r];
      }
    }
    if (mode == 3) {
      NOUNROLL for (int i = 0; i < 16; i++) s_se3[i] = (i % 5 == 0) ? 1.0 : 0.0;
    } else {
      // Frame::GetPoseInverse: [R^T | -R^T t]
      NOUNROLL for (int i = 0; i < 16; i++) s_cand[i] = (i % 5 == 0) ? 1.0 : 0.0;
      NOUNROLL for (int i = 0; i < 3; i++) {
        NOUNROLL for (int j = 0; j < 3; j++) s_cand[i * 4 + j] = s_last[j * 4 + i];
      }
      NOUNROLL for (int i = 0; i < 3; i++) {
        double sacc = 0;
        NOUNROLL for (int k = 0; k < 3; k++) sacc += (-s_cand[i * 4 + k]) * s_last[k * 4 + 3];
        s_cand[i * 4 + 3] = sacc;
      }
      m4_mul_lds(LDSD(s_E), LDSD(s_cand), LDSD(s_se3));
    }
    m4_mul_lds(LDSD(s_se3), LDSD(s_last), LDSD(s_pose));   // trial pose of the first iteration
  }
  __syncthreads();

  // the thread's point (threads >= npts idle through the per-point parts)
  const bool has_pt = tid < npts;   // its world point stays in s_pts (three LDS reads per projection instead of six registers)
  bool vis = false;                         // visible_pts_[pt]: only ever set
  // patch_cache_ row of the point, as what it is computed from: reference pixels (uf-3 .. uf+4) x (vf-3 .. vf+3) and the
  // bilinear weights (survives a level in which the point is clipped, like the reference's cache row)
  uint32_t rlo[7], rhi[7];
  float rw_tl = 0.f, rw_tr = 0.f, rw_bl = 0.f, rw_br = 0.f;
#pragma unroll
  for (int r = 0; r < 7; r++) rlo[r] = rhi[r] = 0u;
  double X = 0, Y = 0, z_inv = 0;           // reference-frame point of the current level's Jacobian
  double sa = 0, sb = 0, sc = 0;            // S dx^2, S dx dy, S dy^2 over the 16 pixels
  bool has_jac = false;                     // jacobian_cache_ columns of the point are non-zero at this level

  // byte c (0..7) of window row r as float: reference pixel (uf - 3 + c, vf - 3 + r) / current pixel (uf - 2 + c, vf - 2 + r)
#define RPX(r, c) ((float)((((c) < 4 ? rlo[r] : rhi[r]) >> (8 * ((c)&3))) & 0xffu))
#define CPX(r, c) ((float)((((c) < 4 ? lo[r] : hi[r]) >> (8 * ((c)&3))) & 0xffu))
  // patch value and gradient of pixel (px, py) of the 4 x 4 patch, the reference's expressions (src/ImageAlign.cc:398-411):
  // rp[-1..2] is window row py+1, rnext[-1..2] row py+2, rprev[0..1] row py, rnext2[0..1] row py+3; rp[0] is column px+1
#define REF_PIXEL(px, py, patch, ddx, ddy)                                                                                        \
  {                                                                                                                               \
    const float m_1 = RPX(py + 1, px), m0 = RPX(py + 1, px + 1), m1 = RPX(py + 1, px + 2), m2 = RPX(py + 1, px + 3);               \
    const float n_1 = RPX(py + 2, px), n0 = RPX(py + 2, px + 1), n1 = RPX(py + 2, px + 2), n2 = RPX(py + 2, px + 3);               \
    const float v0 = RPX(py, px + 1), v1 = RPX(py, px + 2);                                                                        \
    const float x0 = RPX(py + 3, px + 1), x1 = RPX(py + 3, px + 2);                                                                \
    patch = rw_tl * m0 + rw_tr * m1 + rw_bl * n0 + rw_br * n1;                                                                     \
    ddx = 0.5f * ((rw_tl * m1 + rw_tr * m2 + rw_bl * n1 + rw_br * n2) - (rw_tl * m_1 + rw_tr * m0 + rw_bl * n_1 + rw_br * n0));    \
    ddy = 0.5f * ((rw_tl * n0 + rw_tr * n1 + rw_bl * x0 + rw_br * x1) - (rw_tl * v0 + rw_tr * v1 + rw_bl * m0 + rw_br * m1));      \
  }

  // persistent optimisation state (meaningful on thread 0 only)
  double chi2_ = 1e10, error_ = 1e10;
  bool stop_ = false;
  int ok = 1;
  if (tid < 16) s_iters[tid] = 0;

  APROF(0);
  const int lvl_hi = 4, lvl_lo = (mode == 3) ? 4 : 2;
  for (int level = lvl_hi; level >= lvl_lo; level--) {
    const LevelGeom L = P->lv[level];
    const float scale = (mode == 3) ? (float)(1.0 / sf[level]) : inv_sf[level];
    const uint8_t* img_cur = pyr_cur + (size_t)(tb.cur_bcast >= 0 ? tb.cur_bcast : f) * P->pyr_frame_bytes + L.off + (size_t)SD_EDGE * L.pstride + SD_EDGE;
    const uint8_t* img_ref = pyr_ref + (size_t)f * P->pyr_frame_bytes + L.off + (size_t)SD_EDGE * L.pstride + SD_EDGE;
    const int cols = L.w, rows = L.h, step = L.pstride;
    const double fscale = cam.fx * scale;   // cam_fx_*scale
    has_jac = false;                         // jacobian_cache_.setZero()
    if (tid == 0) {
      NOUNROLL for (int i = 0; i < 16; i++) s_bk[i] = s_se3[i];
    }
    bool small = false;

    // ------------------------------------------------ PrecomputePatches (src/ImageAlign.cc:355-421), once per level
    if (has_pt) {
      const double p0 = s_pts[tid * 3], p1 = s_pts[tid * 3 + 1], p2 = s_pts[tid * 3 + 2];
      double xc[3];
      for (int i = 0; i < 3; i++) xc[i] = (s_last[i * 4] * p0 + s_last[i * 4 + 1] * p1 + s_last[i * 4 + 2] * p2) + s_last[i * 4 + 3];
      const double invzc = 1.0 / xc[2];
      if (!(invzc < 0)) {
        const double u2 = cam.fx * xc[0] * invzc + cam.cx;
        const double v2 = cam.fy * xc[1] * invzc + cam.cy;
        const float u_ref = (float)(u2 * scale);
        const float v_ref = (float)(v2 * scale);
        const int uf = (int)floorf(u_ref), vf = (int)floorf(v_ref);
        if (!(uf - 3 < 0 || vf - 3 < 0 || uf + 3 >= cols || vf + 3 >= rows)) {
          vis = true;
          has_jac = true;
          X = xc[0];
          Y = xc[1];
          z_inv = invzc;   // Jacobian3DToPlane needs 1 / z only
          const float su = u_ref - uf, sv = v_ref - vf;
          rw_tl = (float)((1.0 - su) * (1.0 - sv));
          rw_tr = (float)(su * (1.0 - sv));
          rw_bl = (float)((1.0 - su) * sv);
          rw_br = (float)(su * sv);
          // 7 x 2 dwords, all in flight before any is used
          const uint8_t* rp = img_ref + (size_t)(vf - 3) * step + (uf - 3);
#pragma unroll
          for (int r = 0; r < 7; r++) {
            __builtin_memcpy(&rlo[r], rp + (size_t)r * step, 4);
            __builtin_memcpy(&rhi[r], rp + (size_t)r * step + 4, 4);
          }
          sa = sb = sc = 0;
#pragma unroll
          for (int py = 0; py < 4; py++)
#pragma unroll
            for (int px = 0; px < 4; px++) {
              float pv, fdx, fdy;
              REF_PIXEL(px, py, pv, fdx, fdy);
              (void)pv;
              const double ddx = fdx, ddy = fdy;
              sa += ddx * ddx;
              sb += ddx * ddy;
              sc += ddy * ddy;
              if (px == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
      }
    }
    // H of the level: S over the points with a Jacobian of (fx scale)^2 (a J0 J0^T + b (J0 J1^T + J1 J0^T) + c J1 J1^T),
    // reduced one entry at a time over the wave (DPP reductions, a fixed tree); summed over the waves by the solver lane
    {
      const double f2 = fscale * fscale;
      const double ha = has_jac ? sa * f2 : 0.0, hb = has_jac ? sb * f2 : 0.0, hc = has_jac ? sc * f2 : 0.0;
      const double z_inv_2 = z_inv * z_inv;
      double J0[6], J1[6];   // Jacobian3DToPlane at the reference-frame point (src/ImageAlign.cc:423-441)
      J0[0] = -z_inv; J0[1] = 0.0; J0[2] = X * z_inv_2; J0[3] = Y * J0[2]; J0[4] = -(1.0 + X * J0[2]); J0[5] = Y * z_inv;
      J1[0] = 0.0; J1[1] = -z_inv; J1[2] = Y * z_inv_2; J1[3] = 1.0 + Y * J1[2]; J1[4] = -J0[3]; J1[5] = -X * z_inv;
      int q = 0;
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) {
          const double v = __ockl_wfred_add_f64(ha * (J0[a] * J0[b]) + hb * (J0[a] * J1[b] + J1[a] * J0[b]) + hc * (J1[a] * J1[b]));
          if (lane == 0) s_redH[wave][q] = v;
          q++;
        }
    }
    APROF(1);

    for (int it = 0; it < 30; it++) {
      // ------------------------------------------------ ComputeResiduals (src/ImageAlign.cc:281-353) at s_pose
      // patch values and gradients are RE-DERIVED from the reference bytes every iteration on purpose: hoisted out of the loop
      // (which the optimiser would do, they are loop-invariant) they are 48 floats per thread and the kernel spills
#pragma unroll
      for (int r = 0; r < 7; r++) asm volatile("" : "+v"(rlo[r]), "+v"(rhi[r]));
      asm volatile("" : "+v"(rw_tl), "+v"(rw_tr), "+v"(rw_bl), "+v"(rw_br));
      asm volatile("" : "+v"(X), "+v"(Y), "+v"(z_inv));   // likewise the 63 Jacobian products of the level's H
      int nmeas = 0;
      double s1 = 0, s2 = 0;   // S dx res, S dy res over the point's measured pixels
      bool measured = false;
      float4* c4w = (float4*)(s_chi + (has_pt ? tid : 0) * 16);
      if (has_pt && vis) {
        const double p0 = s_pts[tid * 3], p1 = s_pts[tid * 3 + 1], p2 = s_pts[tid * 3 + 2];
        double xc[3];
        for (int i = 0; i < 3; i++) xc[i] = (s_pose[i * 4] * p0 + s_pose[i * 4 + 1] * p1 + s_pose[i * 4 + 2] * p2) + s_pose[i * 4 + 3];
        const double invzc = 1.0 / xc[2];
        if (!(invzc < 0)) {
          const double u2 = cam.fx * xc[0] * invzc + cam.cx;
          const double v2 = cam.fy * xc[1] * invzc + cam.cy;
          const float u_cur = (float)(u2 * scale);
          const float v_cur = (float)(v2 * scale);
          const int uf = (int)floorf(u_cur), vf = (int)floorf(v_cur);
          if (!(uf < 0 || vf < 0 || uf - 3 < 0 || vf - 3 < 0 || uf + 3 >= cols || vf + 3 >= rows)) {
            measured = true;
            const float su = u_cur - uf, sv = v_cur - vf;
            const float w_tl = (float)((1.0 - su) * (1.0 - sv));
            const float w_tr = (float)(su * (1.0 - sv));
            const float w_bl = (float)((1.0 - su) * sv);
            const float w_br = (float)(su * sv);
            // rows vf-2 .. vf+2, columns uf-2 .. uf+5 of the current level
            uint32_t lo[5], hi[5];
            const uint8_t* rp = img_cur + (size_t)(vf - 2) * step + (uf - 2);
#pragma unroll
            for (int r = 0; r < 5; r++) {
              __builtin_memcpy(&lo[r], rp + (size_t)r * step, 4);
              __builtin_memcpy(&hi[r], rp + (size_t)r * step + 4, 4);
            }
#pragma unroll
            for (int py = 0; py < 4; py++) {
              float ch[4];
#pragma unroll
              for (int px = 0; px < 4; px++) {
                float pv, fdx, fdy;
                // (opaque to the optimiser per pixel: sharing the byte -> float conversions of the 4 x 7 window rows between
                // the pixels of a patch row keeps 28 more floats alive than the 96-VGPR budget has room for)
                asm volatile("" : "+v"(rlo[py]), "+v"(rhi[py]), "+v"(rlo[py + 1]), "+v"(rhi[py + 1]), "+v"(rlo[py + 2]), "+v"(rhi[py + 2]),
                             "+v"(rlo[py + 3]), "+v"(rhi[py + 3]));
                REF_PIXEL(px, py, pv, fdx, fdy);
                const float intensity = w_tl * CPX(py, px) + w_tr * CPX(py, px + 1) + w_bl * CPX(py + 1, px) + w_br * CPX(py + 1, px + 1);
                const float res = intensity - pv;
                ch[px] = res * res * 1.0f;
                s1 += (double)fdx * (double)res;
                s2 += (double)fdy * (double)res;
              }
              c4w[py] = make_float4(ch[0], ch[1], ch[2], ch[3]);
              __builtin_amdgcn_sched_barrier(0);   // one patch row at a time: the scheduler must not interleave the 16 pixels
            }
            nmeas = 16;
          }
        }
      }
      if (has_pt && !measured) {   // +0.0f terms: they leave the chi2 chain unchanged
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        c4w[0] = z; c4w[1] = z; c4w[2] = z; c4w[3] = z;
      }
      APROF(3);
      // Jres of the point, reduced one entry at a time over the wave; and the H terms of the points that have a Jacobian
      // but no measurement at this pose (H_ only receives J J^T inside the pixel loop): taken out of the level's H again
      {
        const bool contrib = measured && has_jac;
        const double t1 = contrib ? s1 : 0.0, t2 = contrib ? s2 : 0.0;
        const double z_inv_2 = z_inv * z_inv;
        double J0[6], J1[6];
        J0[0] = -z_inv; J0[1] = 0.0; J0[2] = X * z_inv_2; J0[3] = Y * J0[2]; J0[4] = -(1.0 + X * J0[2]); J0[5] = Y * z_inv;
        J1[0] = 0.0; J1[1] = -z_inv; J1[2] = Y * z_inv_2; J1[3] = 1.0 + Y * J1[2]; J1[4] = -J0[3]; J1[5] = -X * z_inv;
#pragma unroll
        for (int a = 0; a < 6; a++) {
          const double v = __ockl_wfred_add_f64(-(J0[a] * t1 + J1[a] * t2) * fscale);
          if (lane == 0) s_red[wave][21 + a] = v;
        }
        const bool out = has_jac && !measured;
        const bool any_out = __ballot(out) != 0ull;   // wave-uniform
        if (any_out) {
          const double f2 = fscale * fscale;
          const double ha = out ? sa * f2 : 0.0, hb = out ? sb * f2 : 0.0, hc = out ? sc * f2 : 0.0;
          int q = 0;
#pragma unroll
          for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = a; b < 6; b++) {
              const double v = __ockl_wfred_add_f64(ha * (J0[a] * J0[b]) + hb * (J0[a] * J1[b] + J1[a] * J0[b]) + hc * (J1[a] * J1[b]));
              if (lane == 0) s_red[wave][q] = v;
              q++;
            }
        }
        nmeas = __ockl_wfred_add_i32(nmeas);
        if (lane == 0) {
          s_cnt[wave] = nmeas;
          s_delta[wave] = any_out ? 1 : 0;
        }
      }
      __syncthreads();
      APROF(4);
      // ------------------------------------------------ wave 0: float chi2 in the reference's order (src/ImageAlign.cc:298,341):
      // 16 npts dependent adds.  Lane l holds terms 4l .. 4l+3 of a 256-term chunk and the chain runs THROUGH the lanes:
      // d[l] = (((d[l-1] + x[l]) + y[l]) + z[l]) + w[l], d[l-1] from the neighbouring lane by DPP (wave_shr:1; lane 0 takes
      // the carry of the previous chunk).  After s steps lanes 0 .. s-1 hold their final value, so 64 steps of four dependent
      // adds give lane 63 the chunk's sequential sum: the same adds in the same order and rounding as one lane adding them.
      // Terms of points without a measurement are +0.0f (rewritten every iteration) and leave the sum unchanged.
      if (wave == 0) {
        float chi2f = 0.0f;
        const float4* c4 = (const float4*)s_chi;
        const int n4 = npts * 4;
        for (int c0 = 0; c0 < n4; c0 += 64) {
          const int idx = c0 + lane;
          const float4 v = idx < n4 ? c4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
          // one step = four dependent adds: u = d[l-1] + x as ONE v_add_f32 with a DPP source (wave_shr:1); lane 0 has no left
          // neighbour, so the instruction leaves its u alone -- preset to carry + x[0], lane 0's first partial sum in every step.
          // (As v_mov_b32_dpp + v_add_f32 with the carry as the DPP `old` operand the compiler needed seven issue slots per
          // step.)  s_nop 1: the two wait states between the VALU write of d and its DPP read.
          float d = 0.0f, u = chi2f + v.x;
#pragma unroll 8
          for (int st = 0; st < 64; st++) {
            asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(u) : "v"(d), "v"(v.x));
            d = u + v.y;
            d = d + v.z;
            d = d + v.w;
          }
          chi2f = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
        }
        if (lane == 0) s_chi[0] = chi2f;   // the terms are consumed
      }
      // ------------------------------------------------ wave 1, lane 0 (beside the chain): H_.ldlt().solve(Jres_), Exp(-x)
      if (wave == 1) {
        // wave 1 beside the chain: H and Jres over the waves (one lane per entry), the solve and Exp on lane 0, the two 4 x 4
        // products on 16 lanes (single wave: LDS operations complete in order, no barrier needed between the steps)
        if (lane < 21) {
          if (it == 0) {   // the level's H over the waves
            double v = s_redH[0][lane];
#pragma unroll
            for (int w = 1; w < AL_PT_WAVES; w++) v += s_redH[w][lane];
            s_Hlvl[lane] = v;
          }
          double v = s_Hlvl[lane];
#pragma unroll
          for (int w = 0; w < AL_PT_WAVES; w++)
            if (s_delta[w]) v -= s_red[w][lane];
          // entry q = lane of the upper triangle, row-major: (a, bb)
          int a = 0, q = lane;
          while (q >= 6 - a) { q -= 6 - a; a++; }
          const int bb = a + q;
          s_H[a * 6 + bb] = v;
          s_H[bb * 6 + a] = v;
        } else if (lane < 27) {
          const int a = lane - 21;
          double v = s_red[0][21 + a];
#pragma unroll
          for (int w = 1; w < AL_PT_WAVES; w++) v += s_red[w][21 + a];
          s_b[a] = v;
        }
        if (lane == 0) s_mx = solve_and_exp(LDSD(s_H), LDSD(s_b), LDSD(s_x), LDSD(s_nx), LDSD(s_E), LDSD(s_tmp), (ldsi*)s_tr);
        m4_mul_lds16(LDSD(s_se3), LDSD(s_E), LDSD(s_cand), lane);     // se3 * Exp(-x): used only if the step is accepted
        m4_mul_lds16(LDSD(s_cand), LDSD(s_last), LDSD(s_posec), lane);   // ... and the trial pose that goes with it
      }
      __syncthreads();
      APROF(5);
      // ------------------------------------------------ Optimize: the decisions (src/ImageAlign.cc:247-278)
      if (tid == 0) {
        s_iters[level] = it + 1;
        int n_meas = 0;
        for (int w = 0; w < AL_PT_WAVES; w++) n_meas += s_cnt[w];
        const double new_chi2 = (double)(s_chi[0] / (float)n_meas);   // float/size_t -> float, then widened
        if (n_meas == 0) stop_ = true;
        if (isnan(s_x[0])) stop_ = true;
        int brk = 0;
        if ((it > 0 && new_chi2 > chi2_) || stop_) {
          NOUNROLL for (int i = 0; i < 16; i++) s_se3[i] = s_bk[i];
          m4_mul_lds(LDSD(s_se3), LDSD(s_last), LDSD(s_pose));   // rolled back: the next level starts from the last accepted pose
          brk = 1;
        } else {
          if (it > 0 && new_chi2 > chi2_ * 0.99) small = true;
          NOUNROLL for (int i = 0; i < 16; i++) {
            s_bk[i] = s_se3[i];
            s_se3[i] = s_cand[i];
            s_pose[i] = s_posec[i];   // trial pose of the next iteration (of this level or the next)
          }
          chi2_ = new_chi2;
          error_ = s_mx;
          if (error_ <= 1e-10 || small) brk = 1;
        }
        s_ctrl[0] = brk;
        APROF(6);
      }
      __syncthreads();
      APROF(7);
      if (s_ctrl[0]) break;
    }
    // fast mode: "High error in max level means frames are not close, skip other levels"
    if (tid == 0) {
      int fail = 0;
      if (mode == 2 && error_ > 0.01) { error_ = 1e10; ok = 0; fail = 1; }
      if (mode == 3 && error_ > 0.03) { error_ = 1e10; ok = 0; fail = 1; }
      s_ctrl[2] = fail;
    }
    __syncthreads();
    if (s_ctrl[2]) break;
  }
  if (tid == 0) {
    if (ok && mode != 3) {
      m4_mul_lds(LDSD(s_se3), LDSD(s_last), LDSD(s_pose));
      NOUNROLL for (int c = 0; c < 4; c++) {
        NOUNROLL for (int r = 0; r < 4; r++) out_T[c * 4 + r] = s_pose[r * 4 + c];
      }
    } else {
      NOUNROLL for (int i = 0; i < 16; i++) out_T[i] = prior_T[i];   // pose left as it was (src/Tracking.cc:669-672)
    }
    tb.al_ok[f] = ok;
    tb.al_err[f] = error_;
    tb.al_chi2[f] = chi2_;
    NOUNROLL for (int l = 0; l < 16; l++) tb.al_iters[(size_t)f * 16 + l] = s_iters[l];
  }
  }
}

int read_align_prof(unsigned long long* out16, int reset) {
#ifdef SD_PNP_PROF
  SD_HIP_CHECK(hipDeviceSynchronize());
  SD_HIP_CHECK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_align_prof), 16 * sizeof(unsigned long long)));
  if (reset) {
    unsigned long long z[16] = {};
    SD_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_align_prof), z, sizeof(z)));
  }
  return SD_OK;
#else
  set_error("library built without -DSD_PNP_PROF");
  return SD_ERR_INVALID_ARG;
#endif
}

int launch_align(const sd_orb* cur, const sd_orb* ref, const TrackBuffers& tb, const TrackCam& cam, const float* d_inv_sf,
                 const float* d_sf, int n_frames, int mode, hipStream_t s) {
  const int grid = n_frames;
  // register budget: 5 waves per SIMD = four workgroups per CU (1024 frames resident at once); option "track.align_min_waves" = 3 | 4 for experiments
  const int minw = opt(OPT_ALIGN_MIN_WAVES);
  if (minw == 3)
    hipLaunchKernelGGL(k_align<3>, dim3(grid), dim3(AL_PT_THREADS), 0, s, cur->d_plan, cur->d_pyr, ref->d_pyr, tb, cam, d_inv_sf, d_sf, mode, n_frames);
  else if (minw == 4)
    hipLaunchKernelGGL(k_align<4>, dim3(grid), dim3(AL_PT_THREADS), 0, s, cur->d_plan, cur->d_pyr, ref->d_pyr, tb, cam, d_inv_sf, d_sf, mode, n_frames);
  else
    hipLaunchKernelGGL(k_align<5>, dim3(grid), dim3(AL_PT_THREADS), 0, s, cur->d_plan, cur->d_pyr, ref->d_pyr, tb, cam, d_inv_sf, d_sf, mode, n_frames);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

}  // namespace sd
