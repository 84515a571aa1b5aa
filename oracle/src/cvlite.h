// ORACLE (test infrastructure only -- never linked into the product path).
//
// cvlite.h: CPU restatement of the OpenCV primitives the reference's hot path calls.
// OpenCV is NOT vendored by the reference and is absent from this image, so these follow
// the OpenCV 3.2 *generic C++* code paths as specified in SURVEY.md Appendix A
// ("parity unpinned": no reference test pins results at this boundary).
//
//   cvRound / cvFloor / cvCeil      A6   (round-half-even via lrint)
//   FAST-9/16 + score + 3x3 NMS     A1   (call site: reference src/ORBextractor.cc:536)
//   KeyPointsFilter::retainBest     A2   (call sites: src/ORBextractor.cc:586,602)
//   resize INTER_LINEAR 8UC1        A3   (call site: src/ORBextractor.cc:690)
//   copyMakeBorder REFLECT_101      A7   (call sites: src/ORBextractor.cc:692,695)
//   GaussianBlur 7x7 s=2 8U         A4   (call site: src/ORBextractor.cc:660)
//   fastAtan2                       A5   (call site: src/ORBextractor.cc:101)
#pragma once
#include <cstdint>
#include <cmath>
#include <cfloat>
#include <cstring>
#include <vector>
#include <algorithm>

namespace orc {

struct KeyPoint {  // layout == cv::KeyPoint (28 bytes)
  float x, y, size, angle, response;
  int32_t octave, class_id;
};
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

// 8-bit single-channel image view (cv::Mat ROI semantics: data points at (0,0) of the view)
struct View {
  uint8_t* data;
  int cols, rows, step;
  inline uint8_t* ptr(int y) const { return data + (size_t)y * step; }
  inline View roi(int x0, int y0, int w, int h) const { return View{data + (size_t)y0 * step + x0, w, h, step}; }
};

inline int cvRound(double v) { return (int)lrint(v); }
inline int cvRound(float v) { return (int)lrintf(v); }
inline int cvFloor(double v) { int i = (int)v; return i - (i > v); }
inline int cvCeil(double v) { int i = (int)v; return i + (i < v); }

// ---- cv::borderInterpolate, BORDER_REFLECT_101 -------------------------------------------
inline int reflect101(int p, int len) {
  if ((unsigned)p < (unsigned)len) return p;
  if (len == 1) return 0;
  do {
    if (p < 0) p = -p;              // -p - 1 + delta, delta = 1
    else p = len - 1 - (p - len) - 1;
  } while ((unsigned)p >= (unsigned)len);
  return p;
}

// ---- cv::fastAtan2 (OpenCV 3.x polynomial version), degrees in [0,360) --------------------
float fastAtan2(float y, float x);

// ---- cv::FAST(img, kps, threshold, nonmax=true), TYPE_9_16 --------------------------------
void FAST(const View& img, std::vector<KeyPoint>& keypoints, int threshold, bool nonmax);
// score of one pixel (cornerScore<16>), exposed for known-answer tests
int fastCornerScore(const uint8_t* ptr, int step, int threshold);

// ---- cv::KeyPointsFilter::retainBest (OpenCV 3.2: nth_element at begin+n) -----------------
void retainBest(std::vector<KeyPoint>& keypoints, int n_points);

// ---- cv::resize(src, dst, dsize, 0, 0, INTER_LINEAR) for 8UC1 -----------------------------
void resizeLinear(const View& src, const View& dst);

// ---- cv::copyMakeBorder(src -> dst, b,b,b,b, REFLECT_101 [+ISOLATED]) ---------------------
// dst is (src.cols+2b) x (src.rows+2b); src may alias the interior of dst.
void copyMakeBorder101(const View& src, const View& dst, int b);

// ---- cv::GaussianBlur(img, img, Size(7,7), 2, 2, BORDER_REFLECT_101), 8-bit fixed point ---
void gaussianBlur7(const View& src, const View& dst);
void gaussianKernel7Fixed(int k[7]);  // round(g*256) integer taps (exposed for tests)

}  // namespace orc
