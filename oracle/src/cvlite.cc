// ORACLE (test infrastructure only). See cvlite.h for scope and provenance.
#include "cvlite.h"

namespace orc {

// ------------------------------------------------------------------------------------------
// fastAtan2 -- OpenCV 3.2 modules/core/src/mathfuncs.cpp semantics.
// ------------------------------------------------------------------------------------------
float fastAtan2(float y, float x) {
  static const float p1 = 0.9997878412794807f * (float)(180 / M_PI);
  static const float p3 = -0.3258083974640975f * (float)(180 / M_PI);
  static const float p5 = 0.1555786518463281f * (float)(180 / M_PI);
  static const float p7 = -0.04432655554792128f * (float)(180 / M_PI);
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// ------------------------------------------------------------------------------------------
// FAST-9/16.  Scans rows 3..rows-4, cols 3..cols-4 of the given view;
// score = largest threshold for which the pixel is still a corner, minus 1; 3x3 NMS with
// strict '>' against scores of the same call (pixels outside the scan zone score 0).
// ------------------------------------------------------------------------------------------
static void makeOffsets(int pixel[25], int step) {
  static const int offs[16][2] = {{0, 3},  {1, 3},   {2, 2},   {3, 1},  {3, 0},  {3, -1},
                                  {2, -2}, {1, -3},  {0, -3},  {-1, -3}, {-2, -2}, {-3, -1},
                                  {-3, 0}, {-3, 1},  {-2, 2},  {-1, 3}};
  int k = 0;
  for (; k < 16; k++) pixel[k] = offs[k][0] + offs[k][1] * step;
  for (; k < 25; k++) pixel[k] = pixel[k - 16];
}

static int cornerScore16(const uint8_t* ptr, const int pixel[], int threshold) {
  const int K = 8, N = K * 3 + 1;
  int k, v = ptr[0];
  short d[N];
  for (k = 0; k < N; k++) d[k] = (short)(v - ptr[pixel[k]]);

  int a0 = threshold;
  for (k = 0; k < 16; k += 2) {
    int a = std::min((int)d[k + 1], (int)d[k + 2]);
    a = std::min(a, (int)d[k + 3]);
    if (a <= a0) continue;
    a = std::min(a, (int)d[k + 4]);
    a = std::min(a, (int)d[k + 5]);
    a = std::min(a, (int)d[k + 6]);
    a = std::min(a, (int)d[k + 7]);
    a = std::min(a, (int)d[k + 8]);
    a0 = std::max(a0, std::min(a, (int)d[k]));
    a0 = std::max(a0, std::min(a, (int)d[k + 9]));
  }

  int b0 = -a0;
  for (k = 0; k < 16; k += 2) {
    int b = std::max((int)d[k + 1], (int)d[k + 2]);
    b = std::max(b, (int)d[k + 3]);
    b = std::max(b, (int)d[k + 4]);
    b = std::max(b, (int)d[k + 5]);
    if (b >= b0) continue;
    b = std::max(b, (int)d[k + 6]);
    b = std::max(b, (int)d[k + 7]);
    b = std::max(b, (int)d[k + 8]);
    b0 = std::min(b0, std::max(b, (int)d[k]));
    b0 = std::min(b0, std::max(b, (int)d[k + 9]));
  }
  return -b0 - 1;
}

int fastCornerScore(const uint8_t* ptr, int step, int threshold) {
  int pixel[25];
  makeOffsets(pixel, step);
  return cornerScore16(ptr, pixel, threshold);
}

void FAST(const View& img, std::vector<KeyPoint>& keypoints, int threshold, bool nonmax) {
  const int K = 8, N = 16 + K + 1;
  int i, j, k, pixel[25];
  makeOffsets(pixel, img.step);
  keypoints.clear();
  threshold = std::min(std::max(threshold, 0), 255);

  uint8_t threshold_tab[512];
  for (i = -255; i <= 255; i++)
    threshold_tab[i + 255] = (uint8_t)(i < -threshold ? 1 : i > threshold ? 2 : 0);

  const int cols = std::max(img.cols, 0);
  std::vector<uint8_t> sbuf((size_t)cols * 3 + 16, 0);
  std::vector<int> cbuf((size_t)(cols + 1) * 3 + 16, 0);
  uint8_t* buf[3] = {sbuf.data(), sbuf.data() + cols, sbuf.data() + 2 * cols};
  int* cpbuf[3] = {cbuf.data() + 1, cbuf.data() + 1 + (cols + 1), cbuf.data() + 1 + 2 * (cols + 1)};

  for (i = 3; i < img.rows - 2; i++) {
    const uint8_t* ptr = img.ptr(i) + 3;
    uint8_t* curr = buf[(i - 3) % 3];
    int* cornerpos = cpbuf[(i - 3) % 3];
    memset(curr, 0, cols);
    int ncorners = 0;

    if (i < img.rows - 3) {
      for (j = 3; j < img.cols - 3; j++, ptr++) {
        int v = ptr[0];
        const uint8_t* tab = &threshold_tab[0] - v + 255;
        int d = tab[ptr[pixel[0]]] | tab[ptr[pixel[8]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[2]]] | tab[ptr[pixel[10]]];
        d &= tab[ptr[pixel[4]]] | tab[ptr[pixel[12]]];
        d &= tab[ptr[pixel[6]]] | tab[ptr[pixel[14]]];
        if (d == 0) continue;
        d &= tab[ptr[pixel[1]]] | tab[ptr[pixel[9]]];
        d &= tab[ptr[pixel[3]]] | tab[ptr[pixel[11]]];
        d &= tab[ptr[pixel[5]]] | tab[ptr[pixel[13]]];
        d &= tab[ptr[pixel[7]]] | tab[ptr[pixel[15]]];

        if (d & 1) {
          int vt = v - threshold, count = 0;
          for (k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x < vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
        if (d & 2) {
          int vt = v + threshold, count = 0;
          for (k = 0; k < N; k++) {
            int x = ptr[pixel[k]];
            if (x > vt) {
              if (++count > K) {
                cornerpos[ncorners++] = j;
                if (nonmax) curr[j] = (uint8_t)cornerScore16(ptr, pixel, threshold);
                break;
              }
            } else
              count = 0;
          }
        }
      }
    }
    cornerpos[-1] = ncorners;
    if (i == 3) continue;

    const uint8_t* prev = buf[(i - 4 + 3) % 3];
    const uint8_t* pprev = buf[(i - 5 + 3) % 3];
    cornerpos = cpbuf[(i - 4 + 3) % 3];
    ncorners = cornerpos[-1];
    for (k = 0; k < ncorners; k++) {
      j = cornerpos[k];
      int score = prev[j];
      if (!nonmax || (score > prev[j + 1] && score > prev[j - 1] && score > pprev[j - 1] &&
                      score > pprev[j] && score > pprev[j + 1] && score > curr[j - 1] &&
                      score > curr[j] && score > curr[j + 1])) {
        keypoints.push_back(KeyPoint{(float)j, (float)(i - 1), 7.f, -1.f, (float)score, 0, -1});
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// retainBest.  The order left behind by std::nth_element is whatever
// libstdc++'s introselect produces -- exactly what a build of the reference against this
// toolchain's libstdc++ does, so the oracle calls the real std:: algorithms.
// ------------------------------------------------------------------------------------------
void retainBest(std::vector<KeyPoint>& keypoints, int n_points) {
  if (n_points >= 0 && keypoints.size() > (size_t)n_points) {
    if (n_points == 0) {
      keypoints.clear();
      return;
    }
    std::nth_element(keypoints.begin(), keypoints.begin() + n_points, keypoints.end(),
                     [](const KeyPoint& a, const KeyPoint& b) { return a.response > b.response; });
    float ambiguous = keypoints[n_points - 1].response;
    auto new_end = std::partition(keypoints.begin() + n_points, keypoints.end(),
                                  [ambiguous](const KeyPoint& k) { return k.response >= ambiguous; });
    keypoints.resize(new_end - keypoints.begin());
  }
}

// ------------------------------------------------------------------------------------------
// resize INTER_LINEAR, 8UC1.  Exact 2x2 decimation silently takes the
// INTER_AREA fast path; otherwise 11-bit fixed-point separable bilinear.
// ------------------------------------------------------------------------------------------
static inline short satShort(float v) {
  int iv = cvRound(v);
  return (short)std::min(std::max(iv, -32768), 32767);
}

void resizeLinear(const View& src, const View& dst) {
  const int sw = src.cols, sh = src.rows, dw = dst.cols, dh = dst.rows;
  if (dw <= 0 || dh <= 0) return;
  double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
  double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
  int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);  // saturate_cast<int>
  bool is_area_fast = std::fabs(scale_x - iscale_x) < DBL_EPSILON && std::fabs(scale_y - iscale_y) < DBL_EPSILON;

  if (is_area_fast && iscale_x == 2 && iscale_y == 2) {
    for (int y = 0; y < dh; y++) {
      const uint8_t* S = src.ptr(2 * y);
      const uint8_t* nS = src.ptr(2 * y + 1);
      uint8_t* D = dst.ptr(y);
      for (int x = 0; x < dw; x++) D[x] = (uint8_t)((S[2 * x] + S[2 * x + 1] + nS[2 * x] + nS[2 * x + 1] + 2) >> 2);
    }
    return;
  }

  const int SCALE = 2048;
  std::vector<int> xofs(dw), yofs(dh);
  std::vector<short> ialpha(2 * dw), ibeta(2 * dh);
  int xmin = 0, xmax = dw;
  for (int dx = 0; dx < dw; dx++) {
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = cvFloor(fx);
    fx -= sx;
    if (sx < 0) {
      xmin = dx + 1;
      fx = 0, sx = 0;
    }
    if (sx + 1 >= sw) {
      xmax = std::min(xmax, dx);
      if (sx >= sw - 1) fx = 0, sx = sw - 1;
    }
    xofs[dx] = sx;
    float c0 = 1.f - fx, c1 = fx;
    ialpha[2 * dx] = satShort(c0 * SCALE);
    ialpha[2 * dx + 1] = satShort(c1 * SCALE);
  }
  (void)xmin;
  for (int dy = 0; dy < dh; dy++) {
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = cvFloor(fy);
    fy -= sy;
    yofs[dy] = sy;
    float c0 = 1.f - fy, c1 = fy;
    ibeta[2 * dy] = satShort(c0 * SCALE);
    ibeta[2 * dy + 1] = satShort(c1 * SCALE);
  }

  std::vector<int> row0(dw), row1(dw);
  auto hresize = [&](int sy, std::vector<int>& D) {
    const uint8_t* S = src.ptr(sy);
    int dx = 0;
    for (; dx < xmax; dx++) {
      int sx = xofs[dx];
      D[dx] = S[sx] * ialpha[2 * dx] + S[sx + 1] * ialpha[2 * dx + 1];
    }
    for (; dx < dw; dx++) D[dx] = S[xofs[dx]] * SCALE;
  };
  auto clip = [](int x, int a, int b) { return x >= a ? (x < b ? x : b - 1) : a; };
  for (int dy = 0; dy < dh; dy++) {
    int sy0 = clip(yofs[dy], 0, sh), sy1 = clip(yofs[dy] + 1, 0, sh);
    hresize(sy0, row0);
    hresize(sy1, row1);
    int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
    uint8_t* D = dst.ptr(dy);
    for (int x = 0; x < dw; x++)
      D[x] = (uint8_t)((((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2);
  }
}

// ------------------------------------------------------------------------------------------
// copyMakeBorder REFLECT_101
// ------------------------------------------------------------------------------------------
void copyMakeBorder101(const View& src, const View& dst, int b) {
  const int w = src.cols, h = src.rows;
  // interior first (row by row, memmove: src may alias dst interior)
  for (int y = 0; y < h; y++) {
    uint8_t* d = dst.ptr(y + b) + b;
    const uint8_t* s = src.ptr(y);
    if (d != s) memmove(d, s, w);
  }
  for (int y = 0; y < h; y++) {
    uint8_t* d = dst.ptr(y + b);
    for (int x = 0; x < b; x++) {
      d[x] = d[b + reflect101(x - b, w)];
      d[b + w + x] = d[b + reflect101(w + x, w)];
    }
  }
  for (int y = 0; y < b; y++) {
    memcpy(dst.ptr(y), dst.ptr(b + reflect101(y - b, h)), w + 2 * b);
    memcpy(dst.ptr(b + h + y), dst.ptr(b + reflect101(h + y, h)), w + 2 * b);
  }
}

// ------------------------------------------------------------------------------------------
// GaussianBlur 7x7 sigma 2, 8-bit fixed-point separable path (OpenCV 3.2
// generic C++: taps = cvRound(g*256) as int32, row pass int32, column pass
// (sum + 2^15) >> 16 saturated).  NOTE: the SSE2 column filter of an x86 OpenCV build
// rounds exact .5 ties to even instead of up; this restatement follows the generic path.
// ------------------------------------------------------------------------------------------
void gaussianKernel7Fixed(int k[7]) {
  const int n = 7;
  const double sigma = 2.0;
  float cf[7];
  double scale2X = -0.5 / (sigma * sigma);
  double sum = 0;
  for (int i = 0; i < n; i++) {
    double x = i - (n - 1) * 0.5;
    double t = std::exp(scale2X * x * x);
    cf[i] = (float)t;
    sum += cf[i];
  }
  sum = 1. / sum;
  for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
  for (int i = 0; i < n; i++) k[i] = cvRound(cf[i] * 256.f);
}

void gaussianBlur7(const View& src, const View& dst) {
  int k[7];
  gaussianKernel7Fixed(k);
  const int w = src.cols, h = src.rows;
  std::vector<int> tmp((size_t)w * h);
  for (int y = 0; y < h; y++) {
    const uint8_t* S = src.ptr(y);
    int* T = tmp.data() + (size_t)y * w;
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int i = 0; i < 7; i++) s += k[i] * S[reflect101(x + i - 3, w)];
      T[x] = s;
    }
  }
  for (int y = 0; y < h; y++) {
    uint8_t* D = dst.ptr(y);
    const int* R[7];
    for (int i = 0; i < 7; i++) R[i] = tmp.data() + (size_t)reflect101(y + i - 3, h) * w;
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int i = 0; i < 7; i++) s += k[i] * R[i][x];
      int v = (s + (1 << 15)) >> 16;
      D[x] = (uint8_t)std::min(std::max(v, 0), 255);
    }
  }
}

}  // namespace orc
