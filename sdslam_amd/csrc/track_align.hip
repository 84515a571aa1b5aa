// ImageAlign on MI355X: sparse direct (inverse-compositional) photometric alignment, one
// workgroup per frame pair, persistent over pyramid levels and Gauss-Newton iterations.
//
// Replaces SD_SLAM::ImageAlign::ComputePose / Optimize / ComputeResiduals / PrecomputePatches
// (reference src/ImageAlign.cc:45-421) and Exp/RotationExp (:473-517).
//
// Work split (AL_THREADS threads, <= 300 points x 16 patch pixels = 4800 pixel slots, AL_SLOTS per thread):
//   * each thread keeps its pixels' reference patch value and image gradient (dx, dy) in
//     registers for the whole level -- the reference's 230 KB fp64 jacobian_cache_ is never
//     materialised: J = (dx*Jrow0 + dy*Jrow1)*(fx*scale) is recomputed from the point's
//     reference-frame coordinates (LDS) with the same operations, hence the same values;
//   * H (21 upper entries) and Jres (6) are accumulated per thread in fp64 and combined with a
//     fixed-shape wave-shuffle + LDS tree (deterministic run to run);
//   * chi2 is accumulated in FLOAT in point/pixel order by one lane from per-pixel squares in LDS,
//     i.e. bit-identical to the reference's sequential `chi2 += res*res` (src/ImageAlign.cc:298,341),
//     so the accept / stop decisions (`new_chi2 > chi2_`, `> 0.99*chi2_`) do not flip;
//   * lane 0 solves the 6x6 system (pivoted LDLT, Eigen 3.3 semantics) and updates se3.
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
#ifndef AL_THREADS
#define AL_THREADS 256
#endif
#ifndef AL_MIN_WAVES
#define AL_MIN_WAVES 4   // waves per SIMD the register budget is set for (4 workgroups of 256 threads per CU)
#endif
#define AL_WAVES (AL_THREADS / 64)
#define AL_SLOTS ((AL_MAXP * 16 + AL_THREADS - 1) / AL_THREADS)   // pixel slots per thread
static_assert(AL_SLOTS <= 64, "jvalid is a 64-bit mask");
#define AL_PGROUP 10   // pixel slots whose reference-image loads are in flight together in PrecomputePatches

extern "C" __device__ __attribute__((const)) double __ockl_wfred_add_f64(double);
extern "C" __device__ __attribute__((const)) int __ockl_wfred_add_i32(int);

struct Mat4 { double m[4][4]; };

__device__ __forceinline__ void m4_mul(const double* a, const double* b, double* r) {   // row-major 4x4
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += a[i * 4 + k] * b[k * 4 + j];
      r[i * 4 + j] = s;
    }
}

// Eigen 3.3 LDLT<Matrix6d>::solve (pivoting on the largest |diagonal|, pseudo-inverse of D)
// A is a row-major 6x6 in LDS (single lane): private arrays with run-time indices would live in
// scratch memory, whose latency dominated the serial part of every Gauss-Newton iteration.
#define A_(r, c) A[(r) * 6 + (c)]
__device__ void ldlt_solve6(double* A, const double* b, double* x) {
  const int n = 6;
  int tr[6];
  double temp[6];
  for (int k = 0; k < n; ++k) {
    int big = k;
    double best = fabs(A_(k, k));
    for (int i = k + 1; i < n; i++)
      if (fabs(A_(i, i)) > best) { best = fabs(A_(i, i)); big = i; }
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
      break;
    }
    if (rs > 0 && valid)
      for (int i = 0; i < rs; i++) A_(k + 1 + i, k) /= akk;
  }
  double d[6];
  for (int i = 0; i < n; i++) d[i] = b[i];
  for (int k = 0; k < n; k++)
    if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) d[i] -= A_(i, j) * d[j];
  const double tol = 2.2250738585072014e-308;
  for (int i = 0; i < n; i++) {
    if (fabs(A_(i, i)) > tol) d[i] /= A_(i, i);
    else d[i] = 0;
  }
  for (int i = n - 1; i >= 0; i--)
    for (int j = i + 1; j < n; j++) d[i] -= A_(j, i) * d[j];
  for (int k = n - 1; k >= 0; k--)
    if (tr[k] != k) { double t = d[k]; d[k] = d[tr[k]]; d[tr[k]] = t; }
  for (int i = 0; i < n; i++) x[i] = d[i];
}

#undef A_

// ImageAlign::Exp (translation-first twist) -> row-major 4x4
__device__ void se3_exp(const double* update, double* res) {
  const double* upsilon = update;
  const double* omega = update + 3;
  double theta = sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
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
  const double qw = real_factor, qx = imag_factor * omega[0], qy = imag_factor * omega[1], qz = imag_factor * omega[2];
  double rot[3][3];
  {
    const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
    const double twx = tx * qw, twy = ty * qw, twz = tz * qw;
    const double txx = tx * qx, txy = ty * qx, txz = tz * qx;
    const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
    rot[0][0] = 1 - (tyy + tzz); rot[0][1] = txy - twz; rot[0][2] = txz + twy;
    rot[1][0] = txy + twz; rot[1][1] = 1 - (txx + tzz); rot[1][2] = tyz - twx;
    rot[2][0] = txz - twy; rot[2][1] = tyz + twx; rot[2][2] = 1 - (txx + tyy);
  }
  double Om[3][3] = {{0, -omega[2], omega[1]}, {omega[2], 0, -omega[0]}, {-omega[1], omega[0], 0}};
  double V[3][3];
  if (theta < 1e-10) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) V[i][j] = rot[i][j];
  } else {
    double Om2[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += Om[i][k] * Om[k][j];
        Om2[i][j] = s;
      }
    double theta_sq = theta * theta;
    double c1 = (1 - cos(theta)) / (theta_sq);
    double c2 = (theta - sin(theta)) / (theta_sq * theta);
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) V[i][j] = ((i == j ? 1.0 : 0.0) + c1 * Om[i][j]) + c2 * Om2[i][j];
  }
  for (int i = 0; i < 16; i++) res[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) res[i * 4 + j] = rot[i][j];
    res[i * 4 + 3] = V[i][0] * upsilon[0] + V[i][1] * upsilon[1] + V[i][2] * upsilon[2];
  }
}

// MINW = waves per SIMD the register budget is set for: 4 (128 VGPRs, four workgroups per CU: 1024 frames resident at once,
// with spills) for large batches, 2 (256 VGPRs, fewer spills, shorter critical path) when the batch fits 512 slots anyway.
template <int MINW>
__global__ __launch_bounds__(AL_THREADS, MINW) void k_align(const OrbPlan* __restrict__ P, const uint8_t* __restrict__ pyr_cur,
                                               const uint8_t* __restrict__ pyr_ref, TrackBuffers tb, TrackCam cam,
                                               const float* __restrict__ inv_sf, const float* __restrict__ sf, int mode,
                                               int n_frames) {
  __shared__ double s_pts[AL_MAXP * 3];
  __shared__ double s_xyz[AL_MAXP * 3];
  __shared__ uint8_t s_vis[AL_MAXP + 4];
  __shared__ __attribute__((aligned(16))) float4 s_proj[AL_MAXP];   // per point at the trial pose: {ui (< 0: not measured), vi, su, sv}
  __shared__ __attribute__((aligned(16))) float s_chi[AL_MAXP * 16];
  __shared__ double s_red[AL_WAVES][28];
  __shared__ double s_last[16], s_se3[16], s_pose[16], s_bk[16];
  __shared__ double s_H[36], s_b[6], s_x[6];
  __shared__ int s_cnt[AL_WAVES];
  __shared__ int s_ctrl[4];   // [0] break flag, [1] npts
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // persistent workgroups: the grid may be smaller than the batch (launch_align), so that the aligner
  // occupies only part of every CU while the next batch is being extracted beside it
  for (int f = blockIdx.x; f < n_frames; f += gridDim.x) {
  __syncthreads();   // shared state of the previous frame is dead
  const int M = tb.max_points;
  const uint8_t* valid = tb.valid + (size_t)f * M;
  const double* Xw = tb.Xw + (size_t)f * M * 3;
  const int n_last = min(tb.n_last[f], M);
  const int max_pts = (mode == 2 || mode == 3) ? 100 : 300;

  APROF_DECL;
  // ---- gather the first max_pts valid world points, in index order (src/ImageAlign.cc:62-72)
  int running = 0;
  for (int base = 0; base < n_last && running < max_pts; base += AL_THREADS) {
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
    for (int w = 0; w < AL_WAVES; w++) running += s_cnt[w];
    __syncthreads();
  }
  const int npts = min(running, max_pts);
  for (int i = tid; i < AL_MAXP; i += AL_THREADS) s_vis[i] = 0;

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
    continue;
  }
  if (tid == 0) {
    // column-major in HBM (Eigen::Matrix4d::data()) -> row-major working copies
    double last[16], cur[16], inv[16];
    for (int c = 0; c < 4; c++)
      for (int r = 0; r < 4; r++) {
        last[r * 4 + c] = tb.Tref[(size_t)f * 16 + c * 4 + r];
        cur[r * 4 + c] = prior_T[c * 4 + r];
      }
    for (int i = 0; i < 16; i++) s_last[i] = last[i];
    if (mode == 3) {
      for (int i = 0; i < 16; i++) s_se3[i] = (i % 5 == 0) ? 1.0 : 0.0;
    } else {
      // Frame::GetPoseInverse: [R^T | -R^T t]
      for (int i = 0; i < 16; i++) inv[i] = (i % 5 == 0) ? 1.0 : 0.0;
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) inv[i * 4 + j] = last[j * 4 + i];
      for (int i = 0; i < 3; i++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += (-inv[i * 4 + k]) * last[k * 4 + 3];
        inv[i * 4 + 3] = s;
      }
      double se3[16];
      m4_mul(cur, inv, se3);
      for (int i = 0; i < 16; i++) s_se3[i] = se3[i];
    }
  }
  __syncthreads();

  // persistent optimisation state (meaningful on thread 0 only)
  double chi2_ = 1e10, error_ = 1e10;
  bool stop_ = false;
  int ok = 1;
  int iters[16];
  for (int l = 0; l < 16; l++) iters[l] = 0;

  float r_patch[AL_SLOTS], r_dx[AL_SLOTS], r_dy[AL_SLOTS];
#pragma unroll
  for (int k = 0; k < AL_SLOTS; k++) r_patch[k] = r_dx[k] = r_dy[k] = 0.f;
  unsigned long long jvalid = 0;   // bit k: slot k has a Jacobian at this level (AL_SLOTS <= 64)

  APROF(0);
  const int lvl_hi = 4, lvl_lo = (mode == 3) ? 4 : 2;
  for (int level = lvl_hi; level >= lvl_lo; level--) {
    const LevelGeom L = P->lv[level];
    const float scale = (mode == 3) ? (float)(1.0 / sf[level]) : inv_sf[level];
    const uint8_t* img_cur = pyr_cur + (size_t)(tb.cur_bcast >= 0 ? tb.cur_bcast : f) * P->pyr_frame_bytes + L.off + (size_t)SD_EDGE * L.pstride + SD_EDGE;
    const uint8_t* img_ref = pyr_ref + (size_t)f * P->pyr_frame_bytes + L.off + (size_t)SD_EDGE * L.pstride + SD_EDGE;
    const int cols = L.w, rows = L.h, step = L.pstride;
    const double fscale = cam.fx * scale;   // cam_fx_*scale
    jvalid = 0;                              // jacobian_cache_.setZero()
    if (tid == 0)
      for (int i = 0; i < 16; i++) s_bk[i] = s_se3[i];
    bool small = false;

    for (int it = 0; it < 30; it++) {
      // ------------------------------------------------ PrecomputePatches (first iteration of a level)
      if (it == 0) {
        // per point: projection into the reference image at this level, visibility, reference-frame coordinates
        for (int pt = tid; pt < npts; pt += AL_THREADS) {
          int ui = -1, vi = 0;
          float su = 0.f, sv = 0.f;
          const double p0 = s_pts[pt * 3], p1 = s_pts[pt * 3 + 1], p2 = s_pts[pt * 3 + 2];
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
              s_vis[pt] = 1;
              s_xyz[pt * 3] = xc[0];
              s_xyz[pt * 3 + 1] = xc[1];
              s_xyz[pt * 3 + 2] = invzc;   // Jacobian3DToPlane needs 1 / z only
              ui = uf;
              vi = vf;
              su = u_ref - uf;
              sv = v_ref - vf;
            }
          }
          s_proj[pt] = make_float4(__int_as_float(ui), __int_as_float(vi), su, sv);
        }
        __syncthreads();
        // per pixel slot, in two groups: all loads of the group (4 rows of the 4x4 neighbourhood: two dwords, two halfwords,
        // index-clamped) before any is used
#pragma unroll
        for (int g0 = 0; g0 < AL_SLOTS; g0 += AL_PGROUP) {
          uint32_t q_mid[AL_PGROUP], q_next[AL_PGROUP];
          uint16_t q_prev[AL_PGROUP], q_next2[AL_PGROUP];
#pragma unroll
          for (int kk = 0; kk < AL_PGROUP; kk++) {
            const int k = g0 + kk;
            if (k < AL_SLOTS) {
              const int p = tid + AL_THREADS * k;
              const int pt = p >> 4, pix = p & 15;
              int ui = 3, vi = 3;   // slots without a patch load from a fixed in-image address (s_proj is written for pt < npts only)
              if (pt < npts) {
                const float4 pj = s_proj[pt];
                if (__float_as_int(pj.x) >= 0) {
                  ui = __float_as_int(pj.x);
                  vi = __float_as_int(pj.y);
                }
              }
              const uint8_t* rp = img_ref + (size_t)(vi - 2 + (pix >> 2)) * step + (ui - 2 + (pix & 3));
              __builtin_memcpy(&q_mid[kk], rp - 1, 4);
              __builtin_memcpy(&q_next[kk], rp + step - 1, 4);
              __builtin_memcpy(&q_prev[kk], rp - step, 2);
              __builtin_memcpy(&q_next2[kk], rp + 2 * (size_t)step, 2);
            }
          }
#pragma unroll
          for (int kk = 0; kk < AL_PGROUP; kk++) {
            const int k = g0 + kk;
            if (k < AL_SLOTS) {
              const int p = tid + AL_THREADS * k;
              const int pt = p >> 4;
              if (pt < npts) {
                const float4 pj = s_proj[pt];
                if (__float_as_int(pj.x) >= 0) {
                  const float su = pj.z, sv = pj.w;
                  const float w_tl = (float)((1.0 - su) * (1.0 - sv));
                  const float w_tr = (float)(su * (1.0 - sv));
                  const float w_bl = (float)((1.0 - su) * sv);
                  const float w_br = (float)(su * sv);
                  // rp[-1..2], rnext[-1..2], rprev[0..1], rnext2[0..1]
                  const float m_1 = (float)(q_mid[kk] & 0xff), m0 = (float)((q_mid[kk] >> 8) & 0xff), m1 = (float)((q_mid[kk] >> 16) & 0xff),
                              m2 = (float)(q_mid[kk] >> 24);
                  const float n_1 = (float)(q_next[kk] & 0xff), n0 = (float)((q_next[kk] >> 8) & 0xff), n1 = (float)((q_next[kk] >> 16) & 0xff),
                              n2 = (float)(q_next[kk] >> 24);
                  const float v0 = (float)(q_prev[kk] & 0xff), v1 = (float)(q_prev[kk] >> 8);
                  const float x0 = (float)(q_next2[kk] & 0xff), x1 = (float)(q_next2[kk] >> 8);
                  r_patch[k] = w_tl * m0 + w_tr * m1 + w_bl * n0 + w_br * n1;
                  r_dx[k] = 0.5f * ((w_tl * m1 + w_tr * m2 + w_bl * n1 + w_br * n2) - (w_tl * m_1 + w_tr * m0 + w_bl * n_1 + w_br * n0));
                  r_dy[k] = 0.5f * ((w_tl * n0 + w_tr * n1 + w_bl * x0 + w_br * x1) - (w_tl * v0 + w_tr * v1 + w_bl * m0 + w_br * m1));
                  jvalid |= 1ull << k;
                }
              }
            }
          }
        }
      }
      APROF(1);
      if (tid == 0) {
        double pose[16];
        m4_mul(s_se3, s_last, pose);
        for (int i = 0; i < 16; i++) s_pose[i] = pose[i];
      }
      __syncthreads();
      // projection of every visible point at the trial pose, once per point (its 16 pixel slots share it)
      for (int pt = tid; pt < npts; pt += AL_THREADS) {
        int ui = -1, vi = 0;
        float su = 0.f, sv = 0.f;
        if (s_vis[pt]) {
          const double p0 = s_pts[pt * 3], p1 = s_pts[pt * 3 + 1], p2 = s_pts[pt * 3 + 2];
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
              ui = uf;
              vi = vf;
              su = u_cur - uf;
              sv = v_cur - vf;
            }
          }
        }
        s_proj[pt] = make_float4(__int_as_float(ui), __int_as_float(vi), su, sv);
      }
      __syncthreads();
      APROF(2);
      // ------------------------------------------------ ComputeResiduals
      // all image loads of the thread's slots first (index-clamped, straight line): one round trip instead of one per slot
      uint16_t px_top[AL_SLOTS], px_bot[AL_SLOTS];
#pragma unroll
      for (int k = 0; k < AL_SLOTS; k++) {
        const int p = tid + AL_THREADS * k;
        const int pt = p >> 4, pix = p & 15;
        // slots without a measurement (no point, point not visible / out of the image at this pose) load from a fixed
        // in-image address: s_proj is only written for pt < npts, and only measured points carry in-range coordinates
        int ui = 3, vi = 3;
        if (pt < npts) {
          const float4 pj = s_proj[pt];
          if (__float_as_int(pj.x) >= 0) {
            ui = __float_as_int(pj.x);
            vi = __float_as_int(pj.y);
          }
        }
        const uint8_t* rp = img_cur + (size_t)(vi - 2 + (pix >> 2)) * step + (ui - 2 + (pix & 3));
        __builtin_memcpy(&px_top[k], rp, 2);
        __builtin_memcpy(&px_bot[k], rp + step, 2);
      }
      double H[21], Jr[6];
#pragma unroll
      for (int i = 0; i < 21; i++) H[i] = 0;
#pragma unroll
      for (int i = 0; i < 6; i++) Jr[i] = 0;
      int nmeas = 0;
#pragma unroll
      for (int k = 0; k < AL_SLOTS; k++) {
        const int p = tid + AL_THREADS * k;
        const int pt = p >> 4, pix = p & 15;
        float chi = 0.f;
        if (pt < npts) {
          const float4 pj = s_proj[pt];
          const int ui = __float_as_int(pj.x);
          {
            if (ui >= 0) {
              const float su = pj.z, sv = pj.w;
              const float w_tl = (float)((1.0 - su) * (1.0 - sv));
              const float w_tr = (float)(su * (1.0 - sv));
              const float w_bl = (float)((1.0 - su) * sv);
              const float w_br = (float)(su * sv);
              const float intensity = w_tl * (float)(px_top[k] & 0xff) + w_tr * (float)(px_top[k] >> 8) + w_bl * (float)(px_bot[k] & 0xff) +
                                      w_br * (float)(px_bot[k] >> 8);
              const float res = intensity - r_patch[k];
              chi = res * res * 1.0f;
              nmeas++;
              if (jvalid & (1ull << k)) {
                // Jacobian3DToPlane at the reference-frame point, then (dx*row0 + dy*row1)*(fx*scale)
                const double X = s_xyz[pt * 3], Y = s_xyz[pt * 3 + 1];
                const double z_inv = s_xyz[pt * 3 + 2];
                const double z_inv_2 = z_inv * z_inv;
                double J0[6], J1[6];
                J0[0] = -z_inv; J0[1] = 0.0; J0[2] = X * z_inv_2; J0[3] = Y * J0[2]; J0[4] = -(1.0 + X * J0[2]); J0[5] = Y * z_inv;
                J1[0] = 0.0; J1[1] = -z_inv; J1[2] = Y * z_inv_2; J1[3] = 1.0 + Y * J1[2]; J1[4] = -J0[3]; J1[5] = -X * z_inv;
                double J[6];
                const double ddx = r_dx[k], ddy = r_dy[k];
#pragma unroll
                for (int a = 0; a < 6; a++) J[a] = (ddx * J0[a] + ddy * J1[a]) * fscale;
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; a++) {
#pragma unroll
                  for (int b = a; b < 6; b++) H[q++] += J[a] * J[b] * 1.0;
                  Jr[a] -= J[a] * (double)res * 1.0;
                }
              }
            }
          }
        }
        if (p < AL_MAXP * 16) s_chi[p] = chi;
      }
      APROF(3);
      // fixed-shape reduction of the 27 sums + measurement count
      // (device library's DPP reductions: a fixed tree like the shuffle butterfly they replace, every lane gets the sum)
#pragma unroll
      for (int i = 0; i < 21; i++) H[i] = __ockl_wfred_add_f64(H[i]);
#pragma unroll
      for (int i = 0; i < 6; i++) Jr[i] = __ockl_wfred_add_f64(Jr[i]);
      nmeas = __ockl_wfred_add_i32(nmeas);
      if (lane == 0) {
        for (int i = 0; i < 21; i++) s_red[wave][i] = H[i];
        for (int i = 0; i < 6; i++) s_red[wave][21 + i] = Jr[i];
        s_cnt[wave] = nmeas;
      }
      __syncthreads();
      APROF(4);
      // float chi2 in the reference's order (src/ImageAlign.cc:298,341): 4800 dependent adds.  Wave 0 fetches 256 terms per
      // LDS instruction (lane l holds terms 4l .. 4l+3 of the chunk) and runs the chain THROUGH the lanes: one step is
      // d[l] = (((d[l-1] + x[l]) + y[l]) + z[l]) + w[l] with d[l-1] taken from the neighbouring lane by DPP (wave_shr:1; lane 0
      // takes the carry of the previous chunk).  After s steps lanes 0 .. s-1 hold their final value, so 64 steps of four
      // dependent adds give lane 63 the chunk's sequential sum -- the same 256 adds in the same order and rounding as one
      // lane adding them, without LDS or scalar-register latency inside the chain.  Terms of slots without a measurement
      // are +0.0f (rewritten every iteration) and leave the sum unchanged.
      float chi2f = 0.0f;
      if (wave == 0) {
        const float4* c4 = (const float4*)s_chi;
        const int n4 = npts * 4;
        for (int c0 = 0; c0 < n4; c0 += 64) {
          const int idx = c0 + lane;
          const float4 v = idx < AL_MAXP * 4 ? c4[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
          float d = 0.0f;
#pragma unroll 8
          for (int st = 0; st < 64; st++) {
            const float in = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(chi2f), __float_as_int(d), 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
            d = in + v.x;
            d = d + v.y;
            d = d + v.z;
            d = d + v.w;
          }
          chi2f = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63));
        }
      }
      // ------------------------------------------------ Optimize (serial part)
      if (tid == 0) {
        iters[level] = it + 1;
        int q = 0;
        for (int a = 0; a < 6; a++)
          for (int bb = a; bb < 6; bb++) {
            double v = s_red[0][q];
            for (int w = 1; w < AL_WAVES; w++) v += s_red[w][q];
            s_H[a * 6 + bb] = v;
            s_H[bb * 6 + a] = v;
            q++;
          }
        for (int a = 0; a < 6; a++) {
          double v = s_red[0][21 + a];
          for (int w = 1; w < AL_WAVES; w++) v += s_red[w][21 + a];
          s_b[a] = v;
        }
        int n_meas = 0;
        for (int w = 0; w < AL_WAVES; w++) n_meas += s_cnt[w];
        APROF(5);
        const double new_chi2 = (double)(chi2f / (float)n_meas);   // float/size_t -> float, then widened
        if (n_meas == 0) stop_ = true;
        ldlt_solve6(s_H, s_b, s_x);
        double x[6];
        for (int i = 0; i < 6; i++) x[i] = s_x[i];
        if (isnan(x[0])) stop_ = true;
        int brk = 0;
        if ((it > 0 && new_chi2 > chi2_) || stop_) {
          for (int i = 0; i < 16; i++) s_se3[i] = s_bk[i];
          brk = 1;
        } else {
          if (it > 0 && new_chi2 > chi2_ * 0.99) small = true;
          for (int i = 0; i < 16; i++) s_bk[i] = s_se3[i];
          double nx[6], E[16], ns[16];
          for (int i = 0; i < 6; i++) nx[i] = -x[i];
          se3_exp(nx, E);
          m4_mul(s_bk, E, ns);
          for (int i = 0; i < 16; i++) s_se3[i] = ns[i];
          chi2_ = new_chi2;
          double mx = -1;
          for (int i = 0; i < 6; i++)
            if (fabs(x[i]) > mx) mx = fabs(x[i]);
          error_ = mx;
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
      double pose[16];
      m4_mul(s_se3, s_last, pose);
      for (int c = 0; c < 4; c++)
        for (int r = 0; r < 4; r++) out_T[c * 4 + r] = pose[r * 4 + c];
    } else {
      for (int i = 0; i < 16; i++) out_T[i] = prior_T[i];   // pose left as it was (src/Tracking.cc:669-672)
    }
    tb.al_ok[f] = ok;
    tb.al_err[f] = error_;
    tb.al_chi2[f] = chi2_;
    for (int l = 0; l < 16; l++) tb.al_iters[(size_t)f * 16 + l] = iters[l];
  }
  }   // frames of this workgroup
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
  static const int grid_cap = [] { const char* e = getenv("SD_ALIGN_GRID"); return e ? atoi(e) : 0; }();
  const int grid = grid_cap > 0 ? std::min(n_frames, grid_cap) : n_frames;
  if (grid <= 512)
    hipLaunchKernelGGL(k_align<2>, dim3(grid), dim3(AL_THREADS), 0, s, cur->d_plan, cur->d_pyr, ref->d_pyr, tb, cam, d_inv_sf, d_sf, mode,
                       n_frames);
  else
    hipLaunchKernelGGL(k_align<AL_MIN_WAVES>, dim3(grid), dim3(AL_THREADS), 0, s, cur->d_plan, cur->d_pyr, ref->d_pyr, tb, cam, d_inv_sf, d_sf,
                       mode, n_frames);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

}  // namespace sd
