// ORACLE (test infrastructure only -- never linked into the product path).
//
// CPU restatement of SD_SLAM::PnPsolver (RANSAC over EPnP), function by function:
//   ctor gather / SetRansacParameters   reference src/PnPsolver.cc:71-110, 120-155
//   find / iterate                      src/PnPsolver.cc:157-244
//   Refine / CheckInliers               src/PnPsolver.cc:246-315
//   EPnP: choose_control_points :348-381, compute_barycentric_coordinates :383-405, fill_M :407-421,
//         compute_ccs/pcs :423-443, compute_pose :445-492, reprojection_error :514-530,
//         estimate_R_and_t :532-589, solve_for_sign :597-609, compute_R_and_t :611-621,
//         find_betas_approx_{1,2,3} :626-714, compute_L_6x10 :716-755, compute_rho :757-764,
//         gauss_newton + qr_solve :766-901
//   SD_SLAM::Random                     src/extra/utils.cc:23-26
// The OpenCV legacy C calls (cvSVD, cvSolve(CV_SVD), cvInvert(CV_SVD), cvMulTransposed) are
// restated from OpenCV 3.2's generic one-sided Jacobi SVD (modules/core/src/lapack.cpp:
// JacobiSVDImpl_, SVBkSb); OpenCV is absent here, so this is "parity unpinned" (SURVEY
// App. A9).  The RANSAC draws consume an explicit stream of raw rand() values (4 per
// iteration) so runs are reproducible; with stream == NULL the libc rand() is called exactly
// as the reference does.  PnPsolver has no call site in the reference.
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace orc {

// ---- OpenCV 3.2 JacobiSVDImpl_<double> ---------------------------------------------------
// At: n x m (rows are the columns of A), Vt: n x n or NULL.  On return rows of At are the left
// singular vectors (first n1 rows normalised), W descending.
static void JacobiSVD(double* At, int astep, double* _W, double* Vt, int vstep, int m, int n, int n1) {
  const double minval = DBL_MIN, eps = DBL_EPSILON * 10;
  std::vector<double> Wb(n);
  double* W = Wb.data();
  int i, j, k, iter, max_iter = std::max(m, 30);
  double c, s, sd;
  for (i = 0; i < n; i++) {
    for (k = 0, sd = 0; k < m; k++) {
      double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = sd;
    if (Vt) {
      for (k = 0; k < n; k++) Vt[i * vstep + k] = 0;
      Vt[i * vstep + i] = 1;
    }
  }
  for (iter = 0; iter < max_iter; iter++) {
    bool changed = false;
    for (i = 0; i < n - 1; i++)
      for (j = i + 1; j < n; j++) {
        double *Ai = At + i * astep, *Aj = At + j * astep;
        double a = W[i], p = 0, b = W[j];
        for (k = 0; k < m; k++) p += Ai[k] * Aj[k];
        if (std::abs(p) <= eps * std::sqrt(a * b)) continue;
        p *= 2;
        double beta = a - b, gamma = hypot(p, beta);
        if (beta < 0) {
          double delta = (gamma - beta) * 0.5;
          s = std::sqrt(delta / gamma);
          c = p / (gamma * s * 2);
        } else {
          c = std::sqrt((gamma + beta) / (gamma * 2));
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
        if (Vt) {
          double *Vi = Vt + i * vstep, *Vj = Vt + j * vstep;
          for (k = 0; k < n; k++) {
            double t0 = c * Vi[k] + s * Vj[k];
            double t1 = -s * Vi[k] + c * Vj[k];
            Vi[k] = t0;
            Vj[k] = t1;
          }
        }
      }
    if (!changed) break;
  }
  for (i = 0; i < n; i++) {
    for (k = 0, sd = 0; k < m; k++) {
      double t = At[i * astep + k];
      sd += t * t;
    }
    W[i] = std::sqrt(sd);
  }
  for (i = 0; i < n - 1; i++) {
    j = i;
    for (k = i + 1; k < n; k++)
      if (W[j] < W[k]) j = k;
    if (i != j) {
      std::swap(W[i], W[j]);
      if (Vt) {
        for (k = 0; k < m; k++) std::swap(At[i * astep + k], At[j * astep + k]);
        for (k = 0; k < n; k++) std::swap(Vt[i * vstep + k], Vt[j * vstep + k]);
      }
    }
  }
  for (i = 0; i < n; i++) _W[i] = W[i];
  if (!Vt) return;
  uint64_t rng = 0x12345678;   // cv::RNG (multiply-with-carry)
  auto rng_next = [&]() -> unsigned {
    rng = (uint64_t)(unsigned)rng * 4164903690U + (unsigned)(rng >> 32);
    return (unsigned)rng;
  };
  for (i = 0; i < n1; i++) {
    sd = i < n ? W[i] : 0;
    for (int ii = 0; ii < 100 && sd <= minval; ii++) {
      // zero singular value: random vector orthogonalised against the previous ones
      const double val0 = 1. / m;
      for (k = 0; k < m; k++) {
        double val = (rng_next() & 256) != 0 ? val0 : -val0;
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
            asum += std::abs(t);
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
      sd = std::sqrt(sd);
    }
    s = sd > minval ? 1 / sd : 0.;
    for (k = 0; k < m; k++) At[i * astep + k] *= s;
  }
}

// SVD of a square n x n matrix A (row-major): Ut rows = left singular vectors, Vt rows = right.
static void svd_square(const double* A, int n, double* W, double* Ut, double* Vt) {
  std::vector<double> at((size_t)n * n), vt((size_t)n * n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) at[i * n + j] = A[j * n + i];   // transpose(src, temp_a)
  JacobiSVD(at.data(), n, W, vt.data(), n, n, n, n);
  if (Ut) memcpy(Ut, at.data(), sizeof(double) * n * n);
  if (Vt) memcpy(Vt, vt.data(), sizeof(double) * n * n);
}

// cvSolve(A (m x n, m >= n), b (m), x (n), CV_SVD): least squares through SVBkSb
static void solve_svd(const double* A, int m, int n, const double* b, double* x) {
  std::vector<double> a((size_t)n * m), v((size_t)n * n), w(n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++) a[i * m + j] = A[j * n + i];
  JacobiSVD(a.data(), m, w.data(), v.data(), n, m, n, n);
  for (int i = 0; i < n; i++) x[i] = 0;
  double threshold = 0;
  for (int i = 0; i < n; i++) threshold += w[i];
  threshold *= DBL_EPSILON * 2;
  for (int i = 0; i < n; i++) {
    double wi = w[i];
    if (std::abs(wi) <= threshold) continue;
    wi = 1 / wi;
    double s = 0;
    for (int j = 0; j < m; j++) s += a[i * m + j] * b[j];
    s *= wi;
    for (int j = 0; j < n; j++) x[j] = x[j] + s * v[i * n + j];
  }
}

// cvInvert(A 3x3, CV_SVD): pseudo-inverse V diag(1/w) U^T
static void invert_svd3(const double* A, double* Ainv) {
  double w[3], ut[9], vt[9];
  svd_square(A, 3, w, ut, vt);
  for (int i = 0; i < 9; i++) Ainv[i] = 0;
  double threshold = (w[0] + w[1] + w[2]) * DBL_EPSILON * 2;
  for (int i = 0; i < 3; i++) {
    double wi = w[i];
    if (std::abs(wi) <= threshold) continue;
    wi = 1 / wi;
    double buffer[3];
    for (int j = 0; j < 3; j++) buffer[j] = ut[i * 3 + j] * wi;   // u[j*ldu + i] of U == Ut[i][j]
    for (int r = 0; r < 3; r++)
      for (int j = 0; j < 3; j++) Ainv[r * 3 + j] += vt[i * 3 + r] * buffer[j];
  }
}

// cvMulTransposed(M (rows x cols), dst, order=1): dst = M^T M, upper triangle mirrored
static void mul_transposed(const double* M, int rows, int cols, double* dst) {
  for (int i = 0; i < cols; i++)
    for (int j = i; j < cols; j++) {
      double s = 0;
      for (int k = 0; k < rows; k++) s += M[k * cols + i] * M[k * cols + j];
      dst[i * cols + j] = s;
    }
  for (int i = 0; i < cols; i++)
    for (int j = 0; j < i; j++) dst[i * cols + j] = dst[j * cols + i];
}

struct PnPsolver {
  // EPnP state
  double uc, vc, fu, fv;
  std::vector<double> pws, us, alphas, pcs;
  int number_of_correspondences = 0;
  double cws[4][3], ccs[4][3];
  // RANSAC state
  std::vector<float> mvP2D;      // 2 per point
  std::vector<float> mvSigma2;
  std::vector<float> mvP3Dw;     // 3 per point (narrowed to float, src/PnPsolver.cc:93)
  std::vector<size_t> mvKeyPointIndices;
  std::vector<size_t> mvAllIndices;
  size_t nMatchesSize = 0;       // mvpMapPointMatches.size()
  double mRi[3][3], mti[3];
  std::vector<bool> mvbInliersi, mvbBestInliers, mvbRefinedInliers;
  int mnInliersi = 0, mnIterations = 0, mnBestInliers = 0, mnRefinedInliers = 0, N = 0;
  float mBestTcw[16], mRefinedTcw[16];   // row-major 4x4 CV_32F
  double mRansacProb;
  int mRansacMinInliers, mRansacMaxIts, mRansacMinSet;
  float mRansacEpsilon;
  std::vector<float> mvMaxError;
  const int* rand_stream = nullptr;
  size_t rand_pos = 0, rand_len = 0;

  int Random(int min, int max) {
    int r;
    if (rand_stream) {
      r = rand_pos < rand_len ? rand_stream[rand_pos] : 0;
      rand_pos++;
    } else {
      r = rand();
    }
    int d = max - min + 1;
    return static_cast<int>(((static_cast<double>(r) / (static_cast<double>(RAND_MAX) + 1.0)) * d) + min);
  }

  void SetRansacParameters(double probability, int minInliers, int maxIterations, int minSet, float epsilon, float th2) {
    mRansacProb = probability;
    mRansacMinInliers = minInliers;
    mRansacMaxIts = maxIterations;
    mRansacEpsilon = epsilon;
    mRansacMinSet = minSet;
    N = mvP2D.size() / 2;
    mvbInliersi.resize(N);
    int nMinInliers = N * mRansacEpsilon;
    if (nMinInliers < mRansacMinInliers) nMinInliers = mRansacMinInliers;
    if (nMinInliers < minSet) nMinInliers = minSet;
    mRansacMinInliers = nMinInliers;
    if (mRansacEpsilon < (float)mRansacMinInliers / N) mRansacEpsilon = (float)mRansacMinInliers / N;
    int nIterations;
    if (mRansacMinInliers == N) nIterations = 1;
    else nIterations = ceil(log(1 - mRansacProb) / log(1 - pow(mRansacEpsilon, 3)));
    mRansacMaxIts = std::max(1, std::min(nIterations, mRansacMaxIts));
    mvMaxError.resize(mvSigma2.size());
    for (size_t i = 0; i < mvSigma2.size(); i++) mvMaxError[i] = mvSigma2[i] * th2;
  }

  void set_max(int n) {
    if ((int)pws.size() < 3 * n) {
      pws.resize(3 * n);
      us.resize(2 * n);
      alphas.resize(4 * n);
      pcs.resize(3 * n);
    }
  }
  void add_correspondence(double X, double Y, double Z, double u, double v) {
    pws[3 * number_of_correspondences] = X;
    pws[3 * number_of_correspondences + 1] = Y;
    pws[3 * number_of_correspondences + 2] = Z;
    us[2 * number_of_correspondences] = u;
    us[2 * number_of_correspondences + 1] = v;
    number_of_correspondences++;
  }

  void choose_control_points() {
    cws[0][0] = cws[0][1] = cws[0][2] = 0;
    for (int i = 0; i < number_of_correspondences; i++)
      for (int j = 0; j < 3; j++) cws[0][j] += pws[3 * i + j];
    for (int j = 0; j < 3; j++) cws[0][j] /= number_of_correspondences;
    std::vector<double> PW0(3 * number_of_correspondences);
    double pw0tpw0[9], dc[3], uct[9];
    for (int i = 0; i < number_of_correspondences; i++)
      for (int j = 0; j < 3; j++) PW0[3 * i + j] = pws[3 * i + j] - cws[0][j];
    mul_transposed(PW0.data(), number_of_correspondences, 3, pw0tpw0);
    svd_square(pw0tpw0, 3, dc, uct, nullptr);
    for (int i = 1; i < 4; i++) {
      double k = sqrt(dc[i - 1] / number_of_correspondences);
      for (int j = 0; j < 3; j++) cws[i][j] = cws[0][j] + k * uct[3 * (i - 1) + j];
    }
  }

  void compute_barycentric_coordinates() {
    double cc[9], cc_inv[9];
    for (int i = 0; i < 3; i++)
      for (int j = 1; j < 4; j++) cc[3 * i + j - 1] = cws[j][i] - cws[0][i];
    invert_svd3(cc, cc_inv);
    double* ci = cc_inv;
    for (int i = 0; i < number_of_correspondences; i++) {
      double* pi = &pws[3 * i];
      double* a = &alphas[4 * i];
      for (int j = 0; j < 3; j++)
        a[1 + j] = ci[3 * j] * (pi[0] - cws[0][0]) + ci[3 * j + 1] * (pi[1] - cws[0][1]) + ci[3 * j + 2] * (pi[2] - cws[0][2]);
      a[0] = 1.0f - a[1] - a[2] - a[3];
    }
  }

  void fill_M(double* M, const int row, const double* as, const double u, const double v) {
    double* M1 = M + row * 12;
    double* M2 = M1 + 12;
    for (int i = 0; i < 4; i++) {
      M1[3 * i] = as[i] * fu;
      M1[3 * i + 1] = 0.0;
      M1[3 * i + 2] = as[i] * (uc - u);
      M2[3 * i] = 0.0;
      M2[3 * i + 1] = as[i] * fv;
      M2[3 * i + 2] = as[i] * (vc - v);
    }
  }

  void compute_ccs(const double* betas, const double* ut) {
    for (int i = 0; i < 4; i++) ccs[i][0] = ccs[i][1] = ccs[i][2] = 0.0f;
    for (int i = 0; i < 4; i++) {
      const double* v = ut + 12 * (11 - i);
      for (int j = 0; j < 4; j++)
        for (int k = 0; k < 3; k++) ccs[j][k] += betas[i] * v[3 * j + k];
    }
  }
  void compute_pcs() {
    for (int i = 0; i < number_of_correspondences; i++) {
      double* a = &alphas[4 * i];
      double* pc = &pcs[3 * i];
      for (int j = 0; j < 3; j++) pc[j] = a[0] * ccs[0][j] + a[1] * ccs[1][j] + a[2] * ccs[2][j] + a[3] * ccs[3][j];
    }
  }
  static double dist2(const double* p1, const double* p2) {
    return (p1[0] - p2[0]) * (p1[0] - p2[0]) + (p1[1] - p2[1]) * (p1[1] - p2[1]) + (p1[2] - p2[2]) * (p1[2] - p2[2]);
  }
  static double dot(const double* v1, const double* v2) { return v1[0] * v2[0] + v1[1] * v2[1] + v1[2] * v2[2]; }

  double reprojection_error(const double R[3][3], const double t[3]) {
    double sum2 = 0.0;
    for (int i = 0; i < number_of_correspondences; i++) {
      double* pw = &pws[3 * i];
      double Xc = dot(R[0], pw) + t[0];
      double Yc = dot(R[1], pw) + t[1];
      double inv_Zc = 1.0 / (dot(R[2], pw) + t[2]);
      double ue = uc + fu * Xc * inv_Zc;
      double ve = vc + fv * Yc * inv_Zc;
      double u = us[2 * i], v = us[2 * i + 1];
      sum2 += sqrt((u - ue) * (u - ue) + (v - ve) * (v - ve));
    }
    return sum2 / number_of_correspondences;
  }

  void estimate_R_and_t(double R[3][3], double t[3]) {
    double pc0[3] = {0, 0, 0}, pw0[3] = {0, 0, 0};
    for (int i = 0; i < number_of_correspondences; i++) {
      const double* pc = &pcs[3 * i];
      const double* pw = &pws[3 * i];
      for (int j = 0; j < 3; j++) {
        pc0[j] += pc[j];
        pw0[j] += pw[j];
      }
    }
    for (int j = 0; j < 3; j++) {
      pc0[j] /= number_of_correspondences;
      pw0[j] /= number_of_correspondences;
    }
    double abt[9], abt_d[3], ut[9], vt[9];
    for (int i = 0; i < 9; i++) abt[i] = 0;
    for (int i = 0; i < number_of_correspondences; i++) {
      double* pc = &pcs[3 * i];
      double* pw = &pws[3 * i];
      for (int j = 0; j < 3; j++) {
        abt[3 * j] += (pc[j] - pc0[j]) * (pw[0] - pw0[0]);
        abt[3 * j + 1] += (pc[j] - pc0[j]) * (pw[1] - pw0[1]);
        abt[3 * j + 2] += (pc[j] - pc0[j]) * (pw[2] - pw0[2]);
      }
    }
    svd_square(abt, 3, abt_d, ut, vt);
    // abt_u = U (columns = left vectors) = ut^T ; abt_v = V = vt^T ; R = U V^T
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) R[i][j] = ut[0 * 3 + i] * vt[0 * 3 + j] + ut[1 * 3 + i] * vt[1 * 3 + j] + ut[2 * 3 + i] * vt[2 * 3 + j];
    const double det = R[0][0] * R[1][1] * R[2][2] + R[0][1] * R[1][2] * R[2][0] + R[0][2] * R[1][0] * R[2][1] -
                       R[0][2] * R[1][1] * R[2][0] - R[0][1] * R[1][0] * R[2][2] - R[0][0] * R[1][2] * R[2][1];
    if (det < 0) {
      R[2][0] = -R[2][0];
      R[2][1] = -R[2][1];
      R[2][2] = -R[2][2];
    }
    t[0] = pc0[0] - dot(R[0], pw0);
    t[1] = pc0[1] - dot(R[1], pw0);
    t[2] = pc0[2] - dot(R[2], pw0);
  }

  void solve_for_sign() {
    if (pcs[2] < 0.0) {
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 3; j++) ccs[i][j] = -ccs[i][j];
      for (int i = 0; i < number_of_correspondences; i++) {
        pcs[3 * i] = -pcs[3 * i];
        pcs[3 * i + 1] = -pcs[3 * i + 1];
        pcs[3 * i + 2] = -pcs[3 * i + 2];
      }
    }
  }
  double compute_R_and_t(const double* ut, const double* betas, double R[3][3], double t[3]) {
    compute_ccs(betas, ut);
    compute_pcs();
    solve_for_sign();
    estimate_R_and_t(R, t);
    return reprojection_error(R, t);
  }

  void find_betas_approx_1(const double* L, const double* rho, double* betas) {
    double l_6x4[24], b4[4];
    for (int i = 0; i < 6; i++) {
      l_6x4[4 * i] = L[10 * i];
      l_6x4[4 * i + 1] = L[10 * i + 1];
      l_6x4[4 * i + 2] = L[10 * i + 3];
      l_6x4[4 * i + 3] = L[10 * i + 6];
    }
    solve_svd(l_6x4, 6, 4, rho, b4);
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
  }
  void find_betas_approx_2(const double* L, const double* rho, double* betas) {
    double l_6x3[18], b3[3];
    for (int i = 0; i < 6; i++) {
      l_6x3[3 * i] = L[10 * i];
      l_6x3[3 * i + 1] = L[10 * i + 1];
      l_6x3[3 * i + 2] = L[10 * i + 2];
    }
    solve_svd(l_6x3, 6, 3, rho, b3);
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
  }
  void find_betas_approx_3(const double* L, const double* rho, double* betas) {
    double l_6x5[30], b5[5];
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 5; j++) l_6x5[5 * i + j] = L[10 * i + j];
    solve_svd(l_6x5, 6, 5, rho, b5);
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

  void compute_L_6x10(const double* ut, double* l_6x10) {
    const double* v[4] = {ut + 12 * 11, ut + 12 * 10, ut + 12 * 9, ut + 12 * 8};
    double dv[4][6][3];
    for (int i = 0; i < 4; i++) {
      int a = 0, b = 1;
      for (int j = 0; j < 6; j++) {
        dv[i][j][0] = v[i][3 * a] - v[i][3 * b];
        dv[i][j][1] = v[i][3 * a + 1] - v[i][3 * b + 1];
        dv[i][j][2] = v[i][3 * a + 2] - v[i][3 * b + 2];
        b++;
        if (b > 3) {
          a++;
          b = a + 1;
        }
      }
    }
    for (int i = 0; i < 6; i++) {
      double* row = l_6x10 + 10 * i;
      row[0] = dot(dv[0][i], dv[0][i]);
      row[1] = 2.0f * dot(dv[0][i], dv[1][i]);
      row[2] = dot(dv[1][i], dv[1][i]);
      row[3] = 2.0f * dot(dv[0][i], dv[2][i]);
      row[4] = 2.0f * dot(dv[1][i], dv[2][i]);
      row[5] = dot(dv[2][i], dv[2][i]);
      row[6] = 2.0f * dot(dv[0][i], dv[3][i]);
      row[7] = 2.0f * dot(dv[1][i], dv[3][i]);
      row[8] = 2.0f * dot(dv[2][i], dv[3][i]);
      row[9] = dot(dv[3][i], dv[3][i]);
    }
  }
  void compute_rho(double* rho) {
    rho[0] = dist2(cws[0], cws[1]);
    rho[1] = dist2(cws[0], cws[2]);
    rho[2] = dist2(cws[0], cws[3]);
    rho[3] = dist2(cws[1], cws[2]);
    rho[4] = dist2(cws[1], cws[3]);
    rho[5] = dist2(cws[2], cws[3]);
  }

  static void qr_solve(double* pA, int nr, int nc, double* pb, double* pX) {
    double A1[6], A2[6];
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
        return;   // "A is singular, this shouldn't happen"
      } else {
        double *ppAik2 = ppAkk, sum = 0.0, inv_eta = 1. / eta;
        for (int i = k; i < nr; i++) {
          *ppAik2 *= inv_eta;
          sum += *ppAik2 * *ppAik2;
          ppAik2 += nc;
        }
        double sigma = sqrt(sum);
        if (*ppAkk < 0) sigma = -sigma;
        *ppAkk += sigma;
        A1[k] = sigma * *ppAkk;
        A2[k] = -eta * sigma;
        for (int j = k + 1; j < nc; j++) {
          double *ppAik3 = ppAkk, sum2 = 0;
          for (int i = k; i < nr; i++) {
            sum2 += *ppAik3 * ppAik3[j - k];
            ppAik3 += nc;
          }
          double tau = sum2 / A1[k];
          ppAik3 = ppAkk;
          for (int i = k; i < nr; i++) {
            ppAik3[j - k] -= tau * *ppAik3;
            ppAik3 += nc;
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

  void gauss_newton(const double* l_6x10, const double* rho, double betas[4]) {
    double a[24], b[6], x[4] = {0, 0, 0, 0};
    for (int k = 0; k < 5; k++) {
      for (int i = 0; i < 6; i++) {
        const double* rowL = l_6x10 + i * 10;
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
      qr_solve(a, 6, 4, b, x);
      for (int i = 0; i < 4; i++) betas[i] += x[i];
    }
  }

  double compute_pose(double R[3][3], double t[3]) {
    choose_control_points();
    compute_barycentric_coordinates();
    std::vector<double> M((size_t)2 * number_of_correspondences * 12);
    for (int i = 0; i < number_of_correspondences; i++) fill_M(M.data(), 2 * i, &alphas[4 * i], us[2 * i], us[2 * i + 1]);
    double mtm[144], d[12], ut[144];
    mul_transposed(M.data(), 2 * number_of_correspondences, 12, mtm);
    svd_square(mtm, 12, d, ut, nullptr);
    double l_6x10[60], rho[6];
    compute_L_6x10(ut, l_6x10);
    compute_rho(rho);
    double Betas[4][4], rep_errors[4];
    double Rs[4][3][3], ts[4][3];
    find_betas_approx_1(l_6x10, rho, Betas[1]);
    gauss_newton(l_6x10, rho, Betas[1]);
    rep_errors[1] = compute_R_and_t(ut, Betas[1], Rs[1], ts[1]);
    find_betas_approx_2(l_6x10, rho, Betas[2]);
    gauss_newton(l_6x10, rho, Betas[2]);
    rep_errors[2] = compute_R_and_t(ut, Betas[2], Rs[2], ts[2]);
    find_betas_approx_3(l_6x10, rho, Betas[3]);
    gauss_newton(l_6x10, rho, Betas[3]);
    rep_errors[3] = compute_R_and_t(ut, Betas[3], Rs[3], ts[3]);
    int Nb = 1;
    if (rep_errors[2] < rep_errors[1]) Nb = 2;
    if (rep_errors[3] < rep_errors[Nb]) Nb = 3;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) R[i][j] = Rs[Nb][i][j];
      t[i] = ts[Nb][i];
    }
    return rep_errors[Nb];
  }

  void CheckInliers() {
    mnInliersi = 0;
    for (int i = 0; i < N; i++) {
      const float* P3Dw = &mvP3Dw[3 * i];
      const float* P2D = &mvP2D[2 * i];
      float Xc = mRi[0][0] * P3Dw[0] + mRi[0][1] * P3Dw[1] + mRi[0][2] * P3Dw[2] + mti[0];
      float Yc = mRi[1][0] * P3Dw[0] + mRi[1][1] * P3Dw[1] + mRi[1][2] * P3Dw[2] + mti[1];
      float invZc = 1 / (mRi[2][0] * P3Dw[0] + mRi[2][1] * P3Dw[1] + mRi[2][2] * P3Dw[2] + mti[2]);
      double ue = uc + fu * Xc * invZc;
      double ve = vc + fv * Yc * invZc;
      float distX = P2D[0] - ue;
      float distY = P2D[1] - ve;
      float error2 = distX * distX + distY * distY;
      if (error2 < mvMaxError[i]) {
        mvbInliersi[i] = true;
        mnInliersi++;
      } else {
        mvbInliersi[i] = false;
      }
    }
  }

  void store_T(float* T) {
    for (int i = 0; i < 16; i++) T[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) T[4 * i + j] = (float)mRi[i][j];
      T[4 * i + 3] = (float)mti[i];
    }
  }

  bool Refine() {
    std::vector<int> vIndices;
    for (size_t i = 0; i < mvbBestInliers.size(); i++)
      if (mvbBestInliers[i]) vIndices.push_back(i);
    set_max(vIndices.size());
    number_of_correspondences = 0;
    for (size_t i = 0; i < vIndices.size(); i++) {
      int idx = vIndices[i];
      add_correspondence(mvP3Dw[3 * idx], mvP3Dw[3 * idx + 1], mvP3Dw[3 * idx + 2], mvP2D[2 * idx], mvP2D[2 * idx + 1]);
    }
    compute_pose(mRi, mti);
    CheckInliers();
    mnRefinedInliers = mnInliersi;
    mvbRefinedInliers = mvbInliersi;
    if (mnInliersi > mRansacMinInliers) {
      store_T(mRefinedTcw);
      return true;
    }
    return false;
  }

  // returns 1 and fills Tcw (row-major 4x4 float) or 0 (empty cv::Mat)
  int iterate(int nIterations, bool& bNoMore, std::vector<bool>& vbInliers, int& nInliers, float* Tcw) {
    bNoMore = false;
    vbInliers.clear();
    nInliers = 0;
    set_max(mRansacMinSet);
    if (N < mRansacMinInliers) {
      bNoMore = true;
      return 0;
    }
    std::vector<size_t> vAvailableIndices;
    int nCurrentIterations = 0;
    while (mnIterations < mRansacMaxIts || nCurrentIterations < nIterations) {
      nCurrentIterations++;
      mnIterations++;
      number_of_correspondences = 0;
      vAvailableIndices = mvAllIndices;
      for (short i = 0; i < mRansacMinSet; ++i) {
        int randi = Random(0, vAvailableIndices.size() - 1);
        int idx = vAvailableIndices[randi];
        add_correspondence(mvP3Dw[3 * idx], mvP3Dw[3 * idx + 1], mvP3Dw[3 * idx + 2], mvP2D[2 * idx], mvP2D[2 * idx + 1]);
        vAvailableIndices[randi] = vAvailableIndices.back();
        vAvailableIndices.pop_back();
      }
      compute_pose(mRi, mti);
      CheckInliers();
      if (mnInliersi >= mRansacMinInliers) {
        if (mnInliersi > mnBestInliers) {
          mvbBestInliers = mvbInliersi;
          mnBestInliers = mnInliersi;
          store_T(mBestTcw);
        }
        if (Refine()) {
          nInliers = mnRefinedInliers;
          vbInliers = std::vector<bool>(nMatchesSize, false);
          for (int i = 0; i < N; i++)
            if (mvbRefinedInliers[i]) vbInliers[mvKeyPointIndices[i]] = true;
          memcpy(Tcw, mRefinedTcw, sizeof(mRefinedTcw));
          return 1;
        }
      }
    }
    if (mnIterations >= mRansacMaxIts) {
      bNoMore = true;
      if (mnBestInliers >= mRansacMinInliers) {
        nInliers = mnBestInliers;
        vbInliers = std::vector<bool>(nMatchesSize, false);
        for (int i = 0; i < N; i++)
          if (mvbBestInliers[i]) vbInliers[mvKeyPointIndices[i]] = true;
        memcpy(Tcw, mBestTcw, sizeof(mBestTcw));
        return 1;
      }
    }
    return 0;
  }
};

}  // namespace orc

using namespace orc;

extern "C" {

// ctor: n_matches = vpMapPointMatches.size(); valid[i] <=> pMP != NULL && !pMP->isBad();
// kp_xy (n_matches x 2 f32) = F.mvKeysUn[i].pt, kp_octave, level_sigma2 = F.mvLevelSigma2,
// Xw (n_matches x 3 f64) = pMP->GetWorldPos().
void* orc_pnp_create(int n_matches, const uint8_t* valid, const float* kp_xy, const int* kp_octave, const float* level_sigma2,
                     const double* Xw, float fx, float fy, float cx, float cy) {
  PnPsolver* s = new PnPsolver();
  s->nMatchesSize = n_matches;
  int idx = 0;
  for (int i = 0; i < n_matches; i++) {
    if (!valid[i]) continue;
    s->mvP2D.push_back(kp_xy[2 * i]);
    s->mvP2D.push_back(kp_xy[2 * i + 1]);
    s->mvSigma2.push_back(level_sigma2[kp_octave[i]]);
    s->mvP3Dw.push_back((float)Xw[3 * i]);
    s->mvP3Dw.push_back((float)Xw[3 * i + 1]);
    s->mvP3Dw.push_back((float)Xw[3 * i + 2]);
    s->mvKeyPointIndices.push_back(i);
    s->mvAllIndices.push_back(idx);
    idx++;
  }
  s->fu = fx; s->fv = fy; s->uc = cx; s->vc = cy;
  s->SetRansacParameters(0.99, 8, 300, 4, 0.4f, 5.991f);
  return s;
}
void orc_pnp_destroy(void* h) { delete (PnPsolver*)h; }
void orc_pnp_set_ransac(void* h, double probability, int minInliers, int maxIterations, int minSet, float epsilon, float th2) {
  ((PnPsolver*)h)->SetRansacParameters(probability, minInliers, maxIterations, minSet, epsilon, th2);
}
void orc_pnp_params(void* h, int* N, int* minInliers, int* maxIts) {
  PnPsolver* s = (PnPsolver*)h;
  *N = s->N; *minInliers = s->mRansacMinInliers; *maxIts = s->mRansacMaxIts;
}
// rand_stream: raw rand() outputs, 4 consumed per iteration (NULL: libc rand()).
// Tcw_out: 16 floats row-major (the CV_32F 4x4 the reference returns).  Returns 1 / 0 (empty Mat).
int orc_pnp_iterate(void* h, int nIterations, const int* rand_stream, int rand_len, float* Tcw_out, uint8_t* inliers_out,
                    int* nInliers, int* noMore, int* iterations_done) {
  PnPsolver* s = (PnPsolver*)h;
  s->rand_stream = rand_stream;
  s->rand_len = rand_len;
  s->rand_pos = 0;
  bool bNoMore;
  std::vector<bool> vb;
  int nI;
  int ret = s->iterate(nIterations, bNoMore, vb, nI, Tcw_out);
  *nInliers = nI;
  *noMore = bNoMore ? 1 : 0;
  if (iterations_done) *iterations_done = s->mnIterations;
  if (inliers_out)
    for (size_t i = 0; i < s->nMatchesSize; i++) inliers_out[i] = (i < vb.size() && vb[i]) ? 1 : 0;
  return ret;
}

// EPnP alone on n correspondences (known-answer tests): returns reprojection error
double orc_epnp(int n, const double* Xw, const double* uv, double fx, double fy, double cx, double cy, double* R9, double* t3) {
  PnPsolver s;
  s.fu = fx; s.fv = fy; s.uc = cx; s.vc = cy;
  s.set_max(n);
  for (int i = 0; i < n; i++) s.add_correspondence(Xw[3 * i], Xw[3 * i + 1], Xw[3 * i + 2], uv[2 * i], uv[2 * i + 1]);
  double R[3][3], t[3];
  double e = s.compute_pose(R, t);
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) R9[3 * i + j] = R[i][j];
    t3[i] = t[i];
  }
  return e;
}

// SVD of a square row-major matrix (known-answer tests)
void orc_svd_square(const double* A, int n, double* W, double* Ut, double* Vt) { svd_square(A, n, W, Ut, Vt); }

}  // extern "C"
