// ORACLE (test infrastructure only -- never linked into the product path).
//
// CPU restatement of SD_SLAM::ImageAlign (sparse direct / inverse-compositional photometric
// alignment), function by function:
//   ComputePose(Frame&, const Frame&)      reference src/ImageAlign.cc:45-104
//   ComputePose(Frame&, KeyFrame*, fast)   src/ImageAlign.cc:106-176
//   ComputePose(KeyFrame*, KeyFrame*)      src/ImageAlign.cc:178-232
//   Optimize                               src/ImageAlign.cc:234-279
//   ComputeResiduals                       src/ImageAlign.cc:281-353
//   PrecomputePatches                      src/ImageAlign.cc:355-421
//   Project / Jacobian3DToPlane            src/ImageAlign.cc:423-455
//   AbsMax / Exp / RotationExp / RotationHat  src/ImageAlign.cc:457-525
// Eigen (>=3.1, unpinned, absent here) is replaced by hand-written fixed-size double math;
// H.ldlt().solve(b) follows Eigen 3.3's pivoted LDLT (LDLT.h: unblocked in-place factorisation,
// pseudo-inverse of D in the solve).  Quirks reproduced on purpose:
// visible flags are never cleared between levels, stop_/chi2_ persist across levels, chi2 is
// accumulated in float in point order, fx scales both Jacobian rows, bilinear weights are
// computed in double and narrowed to float.  The patch cache is zero-initialised (the
// reference leaves cv::Mat memory uninitialised; rows are always written before being read
// for a visible point).  Parity status: UNPINNED (no reference tests); pinned by the
// known-answer tests in tests/test_oracle_align.py.  Built with -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

struct M3 { double m[3][3]; };
struct V3 { double v[3]; };
struct M4 { double m[4][4]; };   // row-major internally

static M4 m4_from_colmajor(const double* p) {
  M4 r;
  for (int c = 0; c < 4; c++)
    for (int q = 0; q < 4; q++) r.m[q][c] = p[c * 4 + q];
  return r;
}
static void m4_to_colmajor(const M4& a, double* p) {
  for (int c = 0; c < 4; c++)
    for (int q = 0; q < 4; q++) p[c * 4 + q] = a.m[q][c];
}
static M4 m4_mul(const M4& a, const M4& b) {
  M4 r;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += a.m[i][k] * b.m[k][j];
      r.m[i][j] = s;
    }
  return r;
}
static M4 m4_identity() {
  M4 r;
  memset(&r, 0, sizeof(r));
  for (int i = 0; i < 4; i++) r.m[i][i] = 1;
  return r;
}
// Frame::GetPoseInverse: Twc = [Rcw^T | -Rcw^T tcw] (reference src/Frame.cc:204-213)
static M4 pose_inverse(const M4& T) {
  M4 r = m4_identity();
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r.m[i][j] = T.m[j][i];
  for (int i = 0; i < 3; i++) {
    double s = 0;
    for (int k = 0; k < 3; k++) s += (-r.m[i][k]) * T.m[k][3];
    r.m[i][3] = s;
  }
  return r;
}

// ---- Eigen 3.3 LDLT<Matrix6d>::solve -----------------------------------------------------
static void ldlt_solve6(const double Hin[6][6], const double b[6], double x[6]) {
  const int n = 6;
  double A[6][6];
  memcpy(A, Hin, sizeof(A));
  int tr[6];
  double temp[6];
  bool zero_diag = false;
  for (int k = 0; k < n; ++k) {
    int big = k;
    double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < n; i++)
      if (std::fabs(A[i][i]) > best) { best = std::fabs(A[i][i]); big = i; }
    tr[k] = big;
    if (k != big) {
      int s = n - big - 1;
      for (int j = 0; j < k; j++) std::swap(A[k][j], A[big][j]);
      for (int i = 0; i < s; i++) std::swap(A[big + 1 + i][k], A[big + 1 + i][big]);
      std::swap(A[k][k], A[big][big]);
      for (int i = k + 1; i < big; ++i) std::swap(A[i][k], A[big][i]);
    }
    int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = A[j][j] * A[k][j];
      double s = 0;
      for (int j = 0; j < k; j++) s += A[k][j] * temp[j];
      A[k][k] -= s;
      for (int i = 0; i < rs; i++) {
        double t = 0;
        for (int j = 0; j < k; j++) t += A[k + 1 + i][j] * temp[j];
        A[k + 1 + i][k] -= t;
      }
    }
    double akk = A[k][k];
    bool valid = std::fabs(akk) > 0.0;
    if (k == 0 && !valid) {
      for (int j = 0; j < n; j++) tr[j] = j;
      zero_diag = true;
      break;
    }
    if (rs > 0 && valid)
      for (int i = 0; i < rs; i++) A[k + 1 + i][k] /= akk;
  }
  (void)zero_diag;
  double d[6];
  for (int i = 0; i < n; i++) d[i] = b[i];
  for (int k = 0; k < n; k++)
    if (tr[k] != k) std::swap(d[k], d[tr[k]]);
  for (int i = 0; i < n; i++)       // L^-1 (unit lower)
    for (int j = 0; j < i; j++) d[i] -= A[i][j] * d[j];
  const double tol = 2.2250738585072014e-308;   // numeric_limits<double>::min()
  for (int i = 0; i < n; i++) {
    if (std::fabs(A[i][i]) > tol) d[i] /= A[i][i];
    else d[i] = 0;
  }
  for (int i = n - 1; i >= 0; i--)  // L^-T
    for (int j = i + 1; j < n; j++) d[i] -= A[j][i] * d[j];
  for (int k = n - 1; k >= 0; k--)
    if (tr[k] != k) std::swap(d[k], d[tr[k]]);
  for (int i = 0; i < n; i++) x[i] = d[i];
}

struct ImgView { const uint8_t* data; int cols, rows, step; const uint8_t* ptr(int y) const { return data + (size_t)y * step; } };

struct ImageAlign {
  int patch_size_ = 4, min_level_ = 2, max_level_ = 4, max_its_ = 30;
  double chi2_ = 1e10;
  size_t n_meas_ = 0;
  bool stop_ = false;
  double error_ = 1e10;
  double cam_fx_, cam_fy_, cam_cx_, cam_cy_;
  std::vector<float> patch_cache_;
  std::vector<bool> visible_pts_;
  std::vector<V3> points_;
  double H_[6][6];
  double Jres_[6];
  std::vector<double> jacobian_cache_;   // 6 x (size*16), column-major
  int iters_per_level[16];

  bool Project(const double R[3][3], const double T[3], const V3& p, double res[2]) const {
    double x3Dc[3];
    for (int i = 0; i < 3; i++) x3Dc[i] = (R[i][0] * p.v[0] + R[i][1] * p.v[1] + R[i][2] * p.v[2]) + T[i];
    const double invzc = 1.0 / x3Dc[2];
    if (invzc < 0) return false;
    res[0] = cam_fx_ * x3Dc[0] * invzc + cam_cx_;
    res[1] = cam_fy_ * x3Dc[1] * invzc + cam_cy_;
    return true;
  }

  static void Jacobian3DToPlane(const double p[3], double J[2][6]) {
    const double x = p[0], y = p[1];
    const double z_inv = 1. / p[2];
    const double z_inv_2 = z_inv * z_inv;
    J[0][0] = -z_inv;
    J[0][1] = 0.0;
    J[0][2] = x * z_inv_2;
    J[0][3] = y * J[0][2];
    J[0][4] = -(1.0 + x * J[0][2]);
    J[0][5] = y * z_inv;
    J[1][0] = 0.0;
    J[1][1] = -z_inv;
    J[1][2] = y * z_inv_2;
    J[1][3] = 1.0 + y * J[1][2];
    J[1][4] = -J[0][3];
    J[1][5] = -x * z_inv;
  }

  static M4 Exp(const double update[6]) {
    const double* upsilon = update;
    const double* omega = update + 3;
    double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
    double half_theta = 0.5 * theta;
    double imag_factor;
    double real_factor = std::cos(half_theta);
    if (theta < 1e-10) {
      double theta_sq = theta * theta;
      double theta_po4 = theta_sq * theta_sq;
      imag_factor = 0.5 - 0.0208333 * theta_sq + 0.000260417 * theta_po4;
    } else {
      imag_factor = std::sin(half_theta) / theta;
    }
    const double qw = real_factor, qx = imag_factor * omega[0], qy = imag_factor * omega[1], qz = imag_factor * omega[2];
    // Eigen::Quaterniond::toRotationMatrix
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
    double Om2[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        double s = 0;
        for (int k = 0; k < 3; k++) s += Om[i][k] * Om[k][j];
        Om2[i][j] = s;
      }
    double V[3][3];
    if (theta < 1e-10) {
      memcpy(V, rot, sizeof(V));
    } else {
      double theta_sq = theta * theta;
      double c1 = (1 - std::cos(theta)) / (theta_sq);
      double c2 = (theta - std::sin(theta)) / (theta_sq * theta);
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) V[i][j] = ((i == j ? 1.0 : 0.0) + c1 * Om[i][j]) + c2 * Om2[i][j];
    }
    M4 res = m4_identity();
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) res.m[i][j] = rot[i][j];
      res.m[i][3] = V[i][0] * upsilon[0] + V[i][1] * upsilon[1] + V[i][2] * upsilon[2];
    }
    return res;
  }

  void PrecomputePatches(const ImgView& src, const M4& pose, float scale) {
    const int half_patch = patch_size_ / 2, patch_area = patch_size_ * patch_size_, border = half_patch + 1;
    double R[3][3], T[3];
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) R[i][j] = pose.m[i][j];
      T[i] = pose.m[i][3];
    }
    double frame_jac[2][6];
    for (size_t counter = 0; counter < points_.size(); counter++) {
      const V3& p = points_[counter];
      double p2d[2];
      if (!Project(R, T, p, p2d)) continue;
      const float u_ref = p2d[0] * scale;
      const float v_ref = p2d[1] * scale;
      const int u_first_i = floorf(u_ref);
      const int v_first_i = floorf(v_ref);
      if (u_first_i - border < 0 || v_first_i - border < 0 || u_first_i + border >= src.cols || v_first_i + border >= src.rows)
        continue;
      visible_pts_[counter] = true;
      double xyz[3];
      for (int i = 0; i < 3; i++) xyz[i] = (R[i][0] * p.v[0] + R[i][1] * p.v[1] + R[i][2] * p.v[2]) + T[i];
      Jacobian3DToPlane(xyz, frame_jac);
      const float subpix_u_ref = u_ref - u_first_i;
      const float subpix_v_ref = v_ref - v_first_i;
      const float w_tl = (1.0 - subpix_u_ref) * (1.0 - subpix_v_ref);
      const float w_tr = subpix_u_ref * (1.0 - subpix_v_ref);
      const float w_bl = (1.0 - subpix_u_ref) * subpix_v_ref;
      const float w_br = subpix_u_ref * subpix_v_ref;
      size_t pixel_counter = 0;
      float* cache_ptr = patch_cache_.data() + patch_area * counter;
      for (int y = v_first_i - half_patch; y < v_first_i + half_patch; y++) {
        const uint8_t* row_ptr = src.ptr(y);
        const uint8_t* row_prev_ptr = src.ptr(y - 1);
        const uint8_t* row_next_ptr = src.ptr(y + 1);
        const uint8_t* row_next2_ptr = src.ptr(y + 2);
        for (int x = u_first_i - half_patch; x < u_first_i + half_patch; x++, cache_ptr++, pixel_counter++) {
          *cache_ptr = w_tl * row_ptr[x] + w_tr * row_ptr[x + 1] + w_bl * row_next_ptr[x] + w_br * row_next_ptr[x + 1];
          float dx = 0.5f * ((w_tl * row_ptr[x + 1] + w_tr * row_ptr[x + 2] + w_bl * row_next_ptr[x + 1] + w_br * row_next_ptr[x + 2]) -
                             (w_tl * row_ptr[x - 1] + w_tr * row_ptr[x] + w_bl * row_next_ptr[x - 1] + w_br * row_next_ptr[x]));
          float dy = 0.5f * ((w_tl * row_next_ptr[x] + w_tr * row_next_ptr[x + 1] + w_bl * row_next2_ptr[x] + w_br * row_next2_ptr[x + 1]) -
                             (w_tl * row_prev_ptr[x] + w_tr * row_prev_ptr[x + 1] + w_bl * row_ptr[x] + w_br * row_ptr[x + 1]));
          double* J = &jacobian_cache_[(counter * patch_area + pixel_counter) * 6];
          const double f = cam_fx_ * scale;
          for (int k = 0; k < 6; k++) J[k] = (dx * frame_jac[0][k] + dy * frame_jac[1][k]) * f;
        }
      }
    }
  }

  double ComputeResiduals(const ImgView& src, const ImgView& last_img, const M4& last_pose, const M4& se3, float scale, bool patches) {
    const int half_patch = patch_size_ / 2, patch_area = patch_size_ * patch_size_, border = half_patch + 1;
    if (patches) PrecomputePatches(last_img, last_pose, scale);
    M4 pose = m4_mul(se3, last_pose);
    double R[3][3], T[3];
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) R[i][j] = pose.m[i][j];
      T[i] = pose.m[i][3];
    }
    float chi2 = 0.0;
    for (size_t counter = 0; counter < points_.size(); counter++) {
      if (!visible_pts_[counter]) continue;
      double p2d[2];
      if (!Project(R, T, points_[counter], p2d)) continue;
      const float u_cur = p2d[0] * scale;
      const float v_cur = p2d[1] * scale;
      const int u_last_i = floorf(u_cur);
      const int v_last_i = floorf(v_cur);
      if (u_last_i < 0 || v_last_i < 0 || u_last_i - border < 0 || v_last_i - border < 0 || u_last_i + border >= src.cols ||
          v_last_i + border >= src.rows)
        continue;
      const float subpix_u_cur = u_cur - u_last_i;
      const float subpix_v_cur = v_cur - v_last_i;
      const float w_tl = (1.0 - subpix_u_cur) * (1.0 - subpix_v_cur);
      const float w_tr = subpix_u_cur * (1.0 - subpix_v_cur);
      const float w_bl = (1.0 - subpix_u_cur) * subpix_v_cur;
      const float w_br = subpix_u_cur * subpix_v_cur;
      const float* patch_cache_ptr = patch_cache_.data() + patch_area * counter;
      size_t pixel_counter = 0;
      for (int y = v_last_i - half_patch; y < v_last_i + half_patch; y++) {
        const uint8_t* row_ptr = src.ptr(y);
        const uint8_t* row_next_ptr = src.ptr(y + 1);
        for (int x = u_last_i - half_patch; x < u_last_i + half_patch; x++, pixel_counter++, patch_cache_ptr++) {
          const float intensity_cur = w_tl * row_ptr[x] + w_tr * row_ptr[x + 1] + w_bl * row_next_ptr[x] + w_br * row_next_ptr[x + 1];
          const float res = intensity_cur - (*patch_cache_ptr);
          float weight = 1.0;
          chi2 += res * res * weight;
          n_meas_++;
          const double* J = &jacobian_cache_[(counter * patch_area + pixel_counter) * 6];
          for (int a = 0; a < 6; a++) {
            for (int b = 0; b < 6; b++) H_[a][b] += J[a] * J[b] * weight;
            Jres_[a] -= J[a] * res * weight;
          }
        }
      }
    }
    return chi2 / n_meas_;
  }

  void Optimize(const ImgView& src, const ImgView& last_img, const M4& last_pose, M4& se3, float scale, int level) {
    double x[6];
    M4 se3_bk = se3;
    bool small = false;
    int its = 0;
    for (int i = 0; i < max_its_; i++) {
      its = i + 1;
      memset(H_, 0, sizeof(H_));
      memset(Jres_, 0, sizeof(Jres_));
      n_meas_ = 0;
      double new_chi2 = ComputeResiduals(src, last_img, last_pose, se3, scale, i == 0);
      if (n_meas_ == 0) stop_ = true;
      ldlt_solve6(H_, Jres_, x);
      if (std::isnan(x[0])) stop_ = true;
      if ((i > 0 && new_chi2 > chi2_) || stop_) {
        se3 = se3_bk;
        break;
      }
      if (i > 0 && new_chi2 > chi2_ * 0.99) small = true;
      se3_bk = se3;
      double nx[6];
      for (int k = 0; k < 6; k++) nx[k] = -x[k];
      se3 = m4_mul(se3, Exp(nx));
      chi2_ = new_chi2;
      double mx = -1;
      for (int k = 0; k < 6; k++)
        if (std::fabs(x[k]) > mx) mx = std::fabs(x[k]);
      error_ = mx;
      if (error_ <= 1e-10 || small) break;
    }
    if (level >= 0 && level < 16) iters_per_level[level] = its;
  }
};

}  // namespace orc

using namespace orc;

extern "C" {

// mode: 0 = (Frame, Frame)   levels 4..2, <=300 points
//       1 = (Frame, KeyFrame)            <=300 points
//       2 = (Frame, KeyFrame, fast)      <=100 points, abort when error_ > 0.01 after a level
//       3 = (KeyFrame, KeyFrame)         level 4 only, <=100 points, identity init, reject error_ > 0.03
// cur_lv/ref_lv: per pyramid level l (0..nlevels-1) pointer to the level image (ROI origin).
// Poses: 16 doubles, column-major (Eigen::Matrix4d::data()).  Returns 1 (true) / 0 (false).
int orc_align(int nlevels, const uint8_t* const* cur_lv, const uint8_t* const* ref_lv, const int* lv_w, const int* lv_h,
              const int* lv_step_cur, const int* lv_step_ref, const float* inv_sf, const float* sf, const double* Xw, int npts,
              const double* Tref_cm, double* Tcur_cm_inout, double fx, double fy, double cx, double cy, int mode, double* error_out,
              int* iters_out /* nlevels, may be NULL */, double* chi2_out /* may be NULL */) {
  ImageAlign A;
  memset(A.iters_per_level, 0, sizeof(A.iters_per_level));
  A.cam_fx_ = fx; A.cam_fy_ = fy; A.cam_cx_ = cx; A.cam_cy_ = cy;
  if (error_out) *error_out = A.error_;
  if (nlevels <= A.max_level_) return 0;   // "Not enough pyramid levels"
  const int max_points = (mode == 2 || mode == 3) ? 100 : 300;
  int counter = 0;
  for (int i = 0; i < npts && counter < max_points; i++, counter++) A.points_.push_back(V3{{Xw[3 * i], Xw[3 * i + 1], Xw[3 * i + 2]}});
  const int size = (int)A.points_.size();
  if (size == 0) return 0;   // "No points to track!"
  A.patch_cache_.assign((size_t)size * 16, 0.f);
  A.visible_pts_.assign(size, false);
  A.jacobian_cache_.assign((size_t)size * 16 * 6, 0.0);
  M4 last_pose = m4_from_colmajor(Tref_cm);
  M4 cur_pose = m4_from_colmajor(Tcur_cm_inout);
  M4 current_se3 = (mode == 3) ? m4_identity() : m4_mul(cur_pose, pose_inverse(last_pose));
  auto view = [&](const uint8_t* const* lv, const int* step, int l) { return ImgView{lv[l], lv_w[l], lv_h[l], step[l]}; };
  int ret = 1;
  if (mode == 3) {
    const int level = A.max_level_;
    std::fill(A.jacobian_cache_.begin(), A.jacobian_cache_.end(), 0.0);
    float scale = 1.0 / sf[level];
    A.Optimize(view(cur_lv, lv_step_cur, level), view(ref_lv, lv_step_ref, level), last_pose, current_se3, scale, level);
    if (A.error_ > 0.03) {
      A.error_ = 1e10;
      ret = 0;
    }
  } else {
    for (int level = A.max_level_; level >= A.min_level_; level--) {
      std::fill(A.jacobian_cache_.begin(), A.jacobian_cache_.end(), 0.0);
      float scale = inv_sf[level];
      A.Optimize(view(cur_lv, lv_step_cur, level), view(ref_lv, lv_step_ref, level), last_pose, current_se3, scale, level);
      if (mode == 2 && A.error_ > 0.01) {
        A.error_ = 1e10;
        ret = 0;
        break;
      }
    }
    if (ret) {
      M4 pose = m4_mul(current_se3, last_pose);
      m4_to_colmajor(pose, Tcur_cm_inout);
    }
  }
  if (error_out) *error_out = A.error_;
  if (iters_out)
    for (int l = 0; l < nlevels && l < 16; l++) iters_out[l] = A.iters_per_level[l];
  if (chi2_out) *chi2_out = A.chi2_;
  return ret;
}

// stage-level helpers for known-answer tests
void orc_se3_exp(const double* update6, double* T_cm) { m4_to_colmajor(ImageAlign::Exp(update6), T_cm); }
void orc_ldlt_solve6(const double* H_rowmajor36, const double* b6, double* x6) {
  double H[6][6];
  memcpy(H, H_rowmajor36, sizeof(H));
  ldlt_solve6(H, b6, x6);
}

}  // extern "C"
