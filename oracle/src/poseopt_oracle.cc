// ORACLE (test infrastructure only -- never linked into the product path).
//
// CPU restatement of Optimizer::PoseOptimization (reference src/Optimizer.cc:221-415), the pose solve the
// reference actually runs on the tracked path, together with the parts of the vendored g2o
// it exercises for a single 6-DoF vertex with unary edges:
//   OptimizationAlgorithmLevenberg::solve / computeLambdaInit / computeScale   src/extra/g2o/core/optimization_algorithm_levenberg.cpp:60-187
//   SparseOptimizer::optimize, activeRobustChi2                                src/extra/g2o/core/sparse_optimizer.cpp:100-114, 354-419
//   BlockSolver::buildSystem / setLambda / restoreDiagonal                     src/extra/g2o/core/block_solver.hpp:502-604
//   BaseUnaryEdge::constructQuadraticForm, BaseEdge::robustInformation         src/extra/g2o/core/base_unary_edge.hpp:43-72, base_edge.h:96-102
//   RobustKernelHuber::robustify                                               src/extra/g2o/core/robust_kernel_impl.cpp:78-91
//   LinearSolverDense::solve (Eigen::LDLT<MatrixXd>, isPositive)               src/extra/g2o/solvers/linear_solver_dense.h:65-116
//   VertexSE3Expmap::oplusImpl, SE3Quat (exp, operator*, map, normalizeRotation)  src/extra/g2o/types/types_six_dof_expmap.h:73-76, se3quat.h
//   EdgeSE3ProjectXYZOnlyPose / EdgeStereoSE3ProjectXYZOnlyPose (computeError, linearizeOplus, cam_project)
//                                                                              src/extra/g2o/types/types_six_dof_expmap.h:143-199, .cpp:266-364
//   Converter::toSE3Quat / toMatrix4d                                          src/Converter.cc:38-42, 105-107
// Eigen 3.3 pieces are restated from their published algorithms (Quaternion from a rotation matrix, quaternion
// product / vector rotation / toRotationMatrix, pivoted LDLT); summation orders of Eigen's reductions are not
// observable at the parity tolerance used for this stage (pose 1e-5, identical outlier flags).
// Quirks kept: every one of the 4 rounds restarts from the frame's INITIAL pose (the frame pose is only written at
// the end); edges classified with the errors of the last computeActiveErrors (a rejected LM trial leaves the errors of
// the rejected estimate behind); robust kernel removed after the third round; float thresholds / Huber deltas.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace {

struct SE3Quat {
  double q[4];   // x, y, z, w
  double t[3];
};

void quat_normalize(double* q) {
  if (q[3] < 0)
    for (int i = 0; i < 4; i++) q[i] *= -1;
  const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int i = 0; i < 4; i++) q[i] /= n;
}

void quat_from_R(const double m[3][3], double* q) {   // Eigen: QuaternionBase::operator=(MatrixBase)
  double t = m[0][0] + m[1][1] + m[2][2];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[2][1] - m[1][2]) * t;
    q[1] = (m[0][2] - m[2][0]) * t;
    q[2] = (m[1][0] - m[0][1]) * t;
  } else {
    int i = 0;
    if (m[1][1] > m[0][0]) i = 1;
    if (m[2][2] > m[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k][j] - m[j][k]) * t;
    q[j] = (m[j][i] + m[i][j]) * t;
    q[k] = (m[k][i] + m[i][k]) * t;
  }
}

void quat_mul(const double* a, const double* b, double* r) {
  const double w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  const double x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  const double y = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  const double z = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  r[0] = x; r[1] = y; r[2] = z; r[3] = w;
}

void quat_rotate(const double* q, const double* v, double* r) {   // Eigen: v + w*uv + vec x uv, uv = 2 (vec x v)
  double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
  for (int i = 0; i < 3; i++) uv[i] += uv[i];
  const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
  for (int i = 0; i < 3; i++) r[i] = v[i] + q[3] * uv[i] + c[i];
}

void quat_to_R(const double* q, double R[3][3]) {
  const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0][0] = 1 - (tyy + tzz); R[0][1] = txy - twz; R[0][2] = txz + twy;
  R[1][0] = txy + twz; R[1][1] = 1 - (txx + tzz); R[1][2] = tyz - twx;
  R[2][0] = txz - twy; R[2][1] = tyz + twx; R[2][2] = 1 - (txx + tyy);
}

SE3Quat se3_from_Rt(const double R[3][3], const double* t) {
  SE3Quat s;
  quat_from_R(R, s.q);
  for (int i = 0; i < 3; i++) s.t[i] = t[i];
  quat_normalize(s.q);
  return s;
}

SE3Quat se3_mul(const SE3Quat& a, const SE3Quat& b) {   // result._t += _r * tr2._t; result._r *= tr2._r; normalize
  SE3Quat r = a;
  double rt[3];
  quat_rotate(a.q, b.t, rt);
  for (int i = 0; i < 3; i++) r.t[i] += rt[i];
  quat_mul(a.q, b.q, r.q);
  quat_normalize(r.q);
  return r;
}

void mat3_mul(const double a[3][3], const double b[3][3], double r[3][3]) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r[i][j] = a[i][0] * b[0][j] + a[i][1] * b[1][j] + a[i][2] * b[2][j];
}

SE3Quat se3_exp(const double* update) {
  const double omega[3] = {update[0], update[1], update[2]}, upsilon[3] = {update[3], update[4], update[5]};
  const double theta = std::sqrt(omega[0] * omega[0] + omega[1] * omega[1] + omega[2] * omega[2]);
  const double Om[3][3] = {{0, -omega[2], omega[1]}, {omega[2], 0, -omega[0]}, {-omega[1], omega[0], 0}};
  double Om2[3][3], R[3][3], V[3][3];
  mat3_mul(Om, Om, Om2);
  if (theta < 0.00001) {
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) V[i][j] = R[i][j] = ((i == j ? 1.0 : 0.0) + Om[i][j]) + Om2[i][j];
  } else {
    const double a = std::sin(theta) / theta, b = (1 - std::cos(theta)) / (theta * theta),
                 c = (theta - std::sin(theta)) / std::pow(theta, 3);
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

// Eigen 3.3 LDLT<MatrixXd>: pivoting on the largest |diagonal|, sign tracking, solve with the pseudo-inverse of D.
// Returns false when !isPositive().
bool ldlt_solve(int n, const double* Ain, const double* b, double* x) {
  std::vector<double> A(Ain, Ain + n * n), temp(n);
  std::vector<int> tr(n);
  auto a = [&](int r, int c) -> double& { return A[r * n + c]; };
  int sign = 0;   // 0 ZeroSign, 1 PositiveSemiDef, -1 NegativeSemiDef, 2 Indefinite
  bool found_zero_pivot = false;
  for (int k = 0; k < n; ++k) {
    int big = k;
    double best = std::fabs(a(k, k));
    for (int i = k + 1; i < n; i++)
      if (std::fabs(a(i, i)) > best) { best = std::fabs(a(i, i)); big = i; }
    tr[k] = big;
    if (k != big) {
      const int s = n - big - 1;
      for (int j = 0; j < k; j++) std::swap(a(k, j), a(big, j));
      for (int i = 0; i < s; i++) std::swap(a(big + 1 + i, k), a(big + 1 + i, big));
      std::swap(a(k, k), a(big, big));
      for (int i = k + 1; i < big; ++i) std::swap(a(i, k), a(big, i));
    }
    const int rs = n - k - 1;
    if (k > 0) {
      for (int j = 0; j < k; j++) temp[j] = a(j, j) * a(k, j);
      double s = 0;
      for (int j = 0; j < k; j++) s += a(k, j) * temp[j];
      a(k, k) -= s;
      for (int i = 0; i < rs; i++) {
        double t = 0;
        for (int j = 0; j < k; j++) t += a(k + 1 + i, j) * temp[j];
        a(k + 1 + i, k) -= t;
      }
    }
    const double akk = a(k, k);
    const bool pivot_is_valid = std::fabs(akk) > 0.0;
    if (k == 0 && !pivot_is_valid) {
      sign = 0;
      for (int j = 0; j < n; j++) tr[j] = j;
      break;
    }
    if (pivot_is_valid) {
      for (int i = 0; i < rs; i++) a(k + 1 + i, k) /= akk;
    } else {
      found_zero_pivot = true;
    }
    if (sign == 1) {
      if (akk < 0) sign = 2;
    } else if (sign == -1) {
      if (akk > 0) sign = 2;
    } else if (sign == 0) {
      if (akk > 0) sign = 1;
      else if (akk < 0) sign = -1;
    }
    if (found_zero_pivot && pivot_is_valid) sign = 2;
  }
  if (!(sign == 1 || sign == 0)) return false;
  std::vector<double> y(b, b + n);
  for (int k = 0; k < n; k++) std::swap(y[k], y[tr[k]]);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) y[i] -= a(i, j) * y[j];
  const double tol = std::numeric_limits<double>::min();
  for (int i = 0; i < n; i++) {
    if (std::fabs(a(i, i)) > tol) y[i] /= a(i, i);
    else y[i] = 0;
  }
  for (int i = n - 1; i >= 0; i--)
    for (int j = i + 1; j < n; j++) y[i] -= a(j, i) * y[j];
  for (int k = n - 1; k >= 0; k--) std::swap(y[k], y[tr[k]]);
  for (int i = 0; i < n; i++) x[i] = y[i];
  return true;
}

struct Edge {
  int kp;            // keypoint index
  bool stereo;
  double obs[3];
  double info;       // invSigma2 (information = I * invSigma2)
  double Xw[3];
  double delta;      // Huber delta (float value widened)
  bool robust;
  int level;
  double err[3];
};

struct Problem {
  double fx, fy, cx, cy, bf;
  std::vector<Edge> edges;
  SE3Quat est;
};

void map_point(const SE3Quat& T, const double* X, double* out) {   // _r * xyz + _t
  double r[3];
  quat_rotate(T.q, X, r);
  for (int i = 0; i < 3; i++) out[i] = r[i] + T.t[i];
}

void compute_error(const Problem& P, Edge& e) {
  double p[3];
  map_point(P.est, e.Xw, p);
  if (!e.stereo) {
    const double px = p[0] / p[2], py = p[1] / p[2];   // project2d
    e.err[0] = e.obs[0] - (px * P.fx + P.cx);
    e.err[1] = e.obs[1] - (py * P.fy + P.cy);
    e.err[2] = 0;
  } else {
    const float invz = 1.0f / p[2];
    const double r0 = p[0] * invz * P.fx + P.cx, r1 = p[1] * invz * P.fy + P.cy, r2 = r0 - P.bf * invz;
    e.err[0] = e.obs[0] - r0;
    e.err[1] = e.obs[1] - r1;
    e.err[2] = e.obs[2] - r2;
  }
}

double edge_chi2(const Edge& e) {   // _error.dot(information() * _error)
  const int D = e.stereo ? 3 : 2;
  double s = 0;
  for (int i = 0; i < D; i++) s += e.err[i] * (e.info * e.err[i]);
  return s;
}

void huber(double e, double delta, double* rho) {
  const double dsqr = delta * delta;
  if (e <= dsqr) {
    rho[0] = e; rho[1] = 1.; rho[2] = 0.;
  } else {
    const double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * delta - dsqr;
    rho[1] = delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

void compute_active_errors(Problem& P) {
  for (Edge& e : P.edges)
    if (e.level == 0) compute_error(P, e);
}

double active_robust_chi2(const Problem& P) {
  double chi = 0;
  for (const Edge& e : P.edges) {
    if (e.level != 0) continue;
    if (e.robust) {
      double rho[3];
      huber(edge_chi2(e), e.delta, rho);
      chi += rho[0];
    } else {
      chi += edge_chi2(e);
    }
  }
  return chi;
}

void build_system(const Problem& P, double H[36], double b[6]) {
  for (int i = 0; i < 36; i++) H[i] = 0;
  for (int i = 0; i < 6; i++) b[i] = 0;
  for (const Edge& e : P.edges) {
    if (e.level != 0) continue;
    double p[3];
    map_point(P.est, e.Xw, p);
    const double x = p[0], y = p[1], invz = 1.0 / p[2], invz_2 = invz * invz;
    double J[3][6];
    J[0][0] = x * y * invz_2 * P.fx;
    J[0][1] = -(1 + (x * x * invz_2)) * P.fx;
    J[0][2] = y * invz * P.fx;
    J[0][3] = -invz * P.fx;
    J[0][4] = 0;
    J[0][5] = x * invz_2 * P.fx;
    J[1][0] = (1 + y * y * invz_2) * P.fy;
    J[1][1] = -x * y * invz_2 * P.fy;
    J[1][2] = -x * invz * P.fy;
    J[1][3] = 0;
    J[1][4] = -invz * P.fy;
    J[1][5] = y * invz_2 * P.fy;
    const int D = e.stereo ? 3 : 2;
    if (e.stereo) {
      J[2][0] = J[0][0] - P.bf * y * invz_2;
      J[2][1] = J[0][1] + P.bf * x * invz_2;
      J[2][2] = J[0][2];
      J[2][3] = J[0][3];
      J[2][4] = 0;
      J[2][5] = J[0][5] - P.bf * invz_2;
    }
    double rho1 = 1.0;
    if (e.robust) {
      double rho[3];
      huber(edge_chi2(e), e.delta, rho);
      rho1 = rho[1];
    }
    // b -= rho1 * A^T * omega * e ; H += A^T * (rho1 * omega) * A
    for (int a = 0; a < 6; a++) {
      double s = 0;
      for (int d = 0; d < D; d++) s += J[d][a] * (e.info * e.err[d]);
      b[a] -= rho1 * s;
      for (int c = 0; c < 6; c++) {
        double h = 0;
        for (int d = 0; d < D; d++) h += J[d][a] * ((rho1 * e.info) * J[d][c]);
        H[a * 6 + c] += h;
      }
    }
  }
}

// SparseOptimizer::optimize(iterations) with OptimizationAlgorithmLevenberg; returns iterations run
int optimize(Problem& P, int iterations, int* lm_trials_total) {
  double lambda = -1., ni = 2.;
  int nBad = 0, done = 0;
  const int maxTrials = 10;
  bool any_active = false;
  for (const Edge& e : P.edges) any_active |= (e.level == 0);
  if (!any_active) return 0;   // g2o would work on an empty system; PoseOptimization never gets here with < 3 edges in practice
  for (int it = 0; it < iterations; it++) {
    compute_active_errors(P);
    double currentChi = active_robust_chi2(P), tempChi = currentChi;
    const double iniChi = currentChi;
    double H[36], b[6];
    static thread_local double x[6] = {0, 0, 0, 0, 0, 0};   // _solver->x() persists across failed solves
    build_system(P, H, b);
    if (it == 0) {
      double maxDiagonal = 0.;
      for (int j = 0; j < 6; j++) maxDiagonal = std::max(std::fabs(H[j * 6 + j]), maxDiagonal);
      lambda = 1e-5 * maxDiagonal;
      ni = 2;
      nBad = 0;
    }
    double rho = 0;
    int qmax = 0;
    do {
      const SE3Quat backup = P.est;   // push
      double Hl[36];
      for (int i = 0; i < 36; i++) Hl[i] = H[i];
      for (int j = 0; j < 6; j++) Hl[j * 6 + j] += lambda;
      const bool ok2 = ldlt_solve(6, Hl, b, x);
      P.est = se3_mul(se3_exp(x), P.est);           // oplusImpl: exp(update) * estimate
      compute_active_errors(P);
      tempChi = active_robust_chi2(P);
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = (currentChi - tempChi);
      double scale = 0.;
      for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        const double scaleFactor = std::max(1. / 3., alpha);
        lambda *= scaleFactor;
        ni = 2;
        currentChi = tempChi;
      } else {
        lambda *= ni;
        ni *= 2;
        P.est = backup;   // pop (the edges keep the errors of the rejected estimate)
      }
      qmax++;
      if (lm_trials_total) (*lm_trials_total)++;
    } while (rho < 0 && qmax < maxTrials);
    done++;
    if (qmax == maxTrials || rho == 0) break;   // Terminate
    if ((iniChi - currentChi) * 1e3 < iniChi) nBad++;
    else nBad = 0;
    if (nBad >= 3) break;
  }
  return done;
}

}  // namespace

extern "C" {

// kp arrays have N entries; has_mp[i] != 0 <=> pFrame->mvpMapPoints[i] != NULL, Xw[i] its world position.
// Poses: 16 doubles column-major.  outlier_out[i] = pFrame->mvbOutlier[i] for keypoints with a map point (others 0).
// info[0] = nInitialCorrespondences, info[1] = nBad, info[2] = rounds run, info[3] = g2o iterations (sum), info[4] = LM trials (sum).
// Returns nInitialCorrespondences - nBad (0 when fewer than 3 correspondences: the pose is then left untouched).
int orc_pose_optimization(int N, const uint8_t* has_mp, const float* kp_xy, const int* kp_octave, const float* uright,
                          const float* inv_level_sigma2, const double* Xw, float fx, float fy, float cx, float cy, float bf,
                          const double* Tcw_in, double* Tcw_out, uint8_t* outlier_out, int* info) {
  Problem P;
  P.fx = fx; P.fy = fy; P.cx = cx; P.cy = cy; P.bf = bf;
  const float deltaMono = std::sqrt(5.991), deltaStereo = std::sqrt(7.815);
  for (int i = 0; i < N; i++) outlier_out[i] = 0;
  for (int i = 0; i < 16; i++) Tcw_out[i] = Tcw_in[i];
  int nInitialCorrespondences = 0;
  for (int i = 0; i < N; i++) {
    if (!has_mp[i]) continue;
    nInitialCorrespondences++;
    Edge e;
    e.kp = i;
    e.stereo = !(uright[i] < 0);
    e.obs[0] = kp_xy[2 * i];
    e.obs[1] = kp_xy[2 * i + 1];
    e.obs[2] = e.stereo ? uright[i] : 0;
    const float invSigma2 = inv_level_sigma2[kp_octave[i]];
    e.info = invSigma2;
    e.delta = e.stereo ? deltaStereo : deltaMono;
    e.robust = true;
    e.level = 0;
    for (int k = 0; k < 3; k++) e.Xw[k] = Xw[3 * i + k];
    e.err[0] = e.err[1] = e.err[2] = 0;
    P.edges.push_back(e);
  }
  for (int k = 0; k < 5; k++) info[k] = 0;
  info[0] = nInitialCorrespondences;
  if (nInitialCorrespondences < 3) return 0;
  double R[3][3], t[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[r][c] = Tcw_in[c * 4 + r];
    t[r] = Tcw_in[12 + r];
  }
  const float chi2Mono[4] = {5.991, 5.991, 5.991, 5.991}, chi2Stereo[4] = {7.815, 7.815, 7.815, 7.815};
  int nBad = 0;
  for (int it = 0; it < 4; it++) {
    P.est = se3_from_Rt(R, t);   // vSE3->setEstimate(Converter::toSE3Quat(pFrame->GetPose()))
    info[3] += optimize(P, 10, &info[4]);
    info[2]++;
    nBad = 0;
    for (Edge& e : P.edges) {   // mono edges then stereo edges in the reference; the two loops are independent
      if (outlier_out[e.kp]) compute_error(P, e);
      const float chi2 = edge_chi2(e);
      const float thr = e.stereo ? chi2Stereo[it] : chi2Mono[it];
      if (chi2 > thr) {
        outlier_out[e.kp] = 1;
        e.level = 1;
        nBad++;
      } else {
        outlier_out[e.kp] = 0;
        e.level = 0;
      }
      if (it == 2) e.robust = false;
    }
    if (P.edges.size() < 10) break;
  }
  double Ro[3][3];
  quat_to_R(P.est.q, Ro);
  for (int i = 0; i < 16; i++) Tcw_out[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) Tcw_out[c * 4 + r] = Ro[r][c];
    Tcw_out[12 + r] = P.est.t[r];
  }
  info[1] = nBad;
  return nInitialCorrespondences - nBad;
}

}  // extern "C"
