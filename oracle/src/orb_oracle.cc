// ORACLE (test infrastructure only -- never linked into the product path).
//
// CPU restatement of the reference's ORB extractor, function by function:
//   ORBextractor::ORBextractor   reference src/ORBextractor.cc:406-457   (scale tables, quotas, umax)
//   ComputePyramid               src/ORBextractor.cc:680-700
//   ComputeKeyPoints             src/ORBextractor.cc:466-610            (grid FAST, quota loop, retainBest)
//   IC_Angle/computeOrientation  src/ORBextractor.cc:78-102,459-464
//   computeOrbDescriptor         src/ORBextractor.cc:106-143
//   operator()                   src/ORBextractor.cc:620-678
// OpenCV primitives come from cvlite (OpenCV 3.2 generic-C++ semantics).
// Parity status: UNPINNED -- the reference ships no tests/golden vectors and its OpenCV
// dependency is absent here; this file is pinned only by first-principles known-answer
// tests (tests/test_oracle_kat.py).  Floating point: built with -ffp-contract=off.
#include "cvlite.h"
#include "oracle_api.h"
#include <cassert>
#include <chrono>

namespace orc {

static const int PATCH_SIZE = 31;
static const int HALF_PATCH_SIZE = 15;
static const int EDGE_THRESHOLD = 19;

static const int8_t bit_pattern_31_[256 * 4] = {
#include "orb_pattern.inc"
};

struct Point { int x, y; };

struct Level {
  std::vector<uint8_t> buf;  // padded (w+38)x(h+38)
  int w, h, step;
  View whole() { return View{buf.data(), w + 2 * EDGE_THRESHOLD, h + 2 * EDGE_THRESHOLD, step}; }
  View roi() { return View{buf.data() + (size_t)EDGE_THRESHOLD * step + EDGE_THRESHOLD, w, h, step}; }
};

struct ORBextractor {
  int nfeatures;
  double scaleFactor;  // (sic) member is double, ctor arg float: src/ORBextractor.h:38,78
  int nlevels, thFAST;
  std::vector<int> mnFeaturesPerLevel, umax;
  std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
  std::vector<Point> pattern;
  std::vector<Level> pyr;
  // diagnostics for stage-level parity tests
  std::vector<std::vector<KeyPoint>> lastLevelKeypoints;   // after selection, level coords
  std::vector<std::vector<int>> lastCellTotals;            // nTotal per cell (raster) per level
  std::vector<std::vector<uint8_t>> lastBlurred;           // compact blurred level (levels with kps)

  ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _thFAST)
      : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), thFAST(_thFAST) {
    mvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvScaleFactor[0] = 1.0f;
    mvLevelSigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
      mvScaleFactor[i] = mvScaleFactor[i - 1] * scaleFactor;
      mvLevelSigma2[i] = mvScaleFactor[i] * mvScaleFactor[i];
    }
    mvInvScaleFactor.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    for (int i = 0; i < nlevels; i++) {
      mvInvScaleFactor[i] = 1.0f / mvScaleFactor[i];
      mvInvLevelSigma2[i] = 1.0f / mvLevelSigma2[i];
    }
    mnFeaturesPerLevel.resize(nlevels);
    float factor = 1.0f / scaleFactor;
    float nDesiredFeaturesPerScale = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sumFeatures = 0;
    for (int level = 0; level < nlevels - 1; level++) {
      mnFeaturesPerLevel[level] = cvRound(nDesiredFeaturesPerScale);
      sumFeatures += mnFeaturesPerLevel[level];
      nDesiredFeaturesPerScale *= factor;
    }
    mnFeaturesPerLevel[nlevels - 1] = std::max(nfeatures - sumFeatures, 0);

    for (int i = 0; i < 512; i++) pattern.push_back(Point{bit_pattern_31_[2 * i], bit_pattern_31_[2 * i + 1]});

    umax.resize(HALF_PATCH_SIZE + 1);
    int v, v0, vmax = cvFloor(HALF_PATCH_SIZE * sqrt(2.f) / 2 + 1);
    int vmin = cvCeil(HALF_PATCH_SIZE * sqrt(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) umax[v] = cvRound(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
      while (umax[v0] == umax[v0 + 1]) ++v0;
      umax[v] = v0;
      ++v0;
    }
  }

  // src/ORBextractor.cc:680-700
  void ComputePyramid(const View& image) {
    pyr.resize(nlevels);
    for (int level = 0; level < nlevels; ++level) {
      float scale = mvInvScaleFactor[level];
      int sw = cvRound((float)image.cols * scale), sh = cvRound((float)image.rows * scale);
      Level& L = pyr[level];
      L.w = sw;
      L.h = sh;
      L.step = sw + EDGE_THRESHOLD * 2;
      L.buf.assign((size_t)L.step * (sh + EDGE_THRESHOLD * 2), 0);
      if (level != 0) {
        resizeLinear(pyr[level - 1].roi(), L.roi());
        copyMakeBorder101(L.roi(), L.whole(), EDGE_THRESHOLD);
      } else {
        copyMakeBorder101(image, L.whole(), EDGE_THRESHOLD);
      }
    }
  }

  // src/ORBextractor.cc:78-102
  float IC_Angle(const View& image, float ptx, float pty) const {
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = image.ptr(cvRound(pty)) + cvRound(ptx);
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    int step = image.step;
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
      int v_sum = 0;
      int d = umax[v];
      for (int u = -d; u <= d; ++u) {
        int val_plus = center[u + v * step], val_minus = center[u - v * step];
        v_sum += (val_plus - val_minus);
        m_10 += u * (val_plus + val_minus);
      }
      m_01 += v * v_sum;
    }
    return fastAtan2((float)m_01, (float)m_10);
  }

  // src/ORBextractor.cc:466-610
  void ComputeKeyPoints(std::vector<std::vector<KeyPoint>>& allKeypoints) {
    allKeypoints.assign(nlevels, {});
    lastCellTotals.assign(nlevels, {});
    float imageRatio = (float)pyr[0].w / pyr[0].h;

    for (int level = 0; level < nlevels; ++level) {
      const int nDesiredFeatures = mnFeaturesPerLevel[level];
      const int levelCols = sqrt((float)nDesiredFeatures / (5 * imageRatio));
      const int levelRows = imageRatio * levelCols;
      if (levelCols <= 0 || levelRows <= 0) continue;  // reference would divide by zero here

      View img = pyr[level].roi();
      const int minBorderX = EDGE_THRESHOLD;
      const int minBorderY = minBorderX;
      const int maxBorderX = img.cols - EDGE_THRESHOLD;
      const int maxBorderY = img.rows - EDGE_THRESHOLD;

      const int W = maxBorderX - minBorderX;
      const int H = maxBorderY - minBorderY;
      const int cellW = ceil((float)W / levelCols);
      const int cellH = ceil((float)H / levelRows);

      const int nCells = levelRows * levelCols;
      const int nfeaturesCell = ceil((float)nDesiredFeatures / nCells);

      std::vector<std::vector<std::vector<KeyPoint>>> cellKeyPoints(levelRows, std::vector<std::vector<KeyPoint>>(levelCols));
      std::vector<std::vector<int>> nToRetain(levelRows, std::vector<int>(levelCols, 0));
      std::vector<std::vector<int>> nTotal(levelRows, std::vector<int>(levelCols, 0));
      std::vector<std::vector<bool>> bNoMore(levelRows, std::vector<bool>(levelCols, false));
      std::vector<int> iniXCol(levelCols);
      std::vector<int> iniYRow(levelRows);
      int nNoMore = 0;
      int nToDistribute = 0;

      float hY = cellH + 6;
      for (int i = 0; i < levelRows; i++) {
        const float iniY = minBorderY + i * cellH - 3;
        iniYRow[i] = iniY;
        if (i == levelRows - 1) {
          hY = maxBorderY + 3 - iniY;
          if (hY <= 0) continue;
        }
        float hX = cellW + 6;
        for (int j = 0; j < levelCols; j++) {
          float iniX;
          if (i == 0) {
            iniX = minBorderX + j * cellW - 3;
            iniXCol[j] = iniX;
          } else {
            iniX = iniXCol[j];
          }
          if (j == levelCols - 1) {
            hX = maxBorderX + 3 - iniX;
            if (hX <= 0) continue;
          }
          // cv::Mat::rowRange/colRange would assert outside the level image
          int x0 = (int)iniX, x1 = (int)(iniX + hX), y0 = (int)iniY, y1 = (int)(iniY + hY);
          if (x0 < 0 || y0 < 0 || x1 > img.cols || y1 > img.rows || x1 < x0 || y1 < y0) continue;
          View cellImage = img.roi(x0, y0, x1 - x0, y1 - y0);
          cellKeyPoints[i][j].reserve(std::max(nfeaturesCell, 0) * 5);
          const double tf = now_ns();
          FAST(cellImage, cellKeyPoints[i][j], thFAST, true);
          stage_ns[1] += now_ns() - tf;

          const int nKeys = cellKeyPoints[i][j].size();
          nTotal[i][j] = nKeys;
          if (nKeys > nfeaturesCell) {
            nToRetain[i][j] = nfeaturesCell;
            bNoMore[i][j] = false;
          } else {
            nToRetain[i][j] = nKeys;
            nToDistribute += nfeaturesCell - nKeys;
            bNoMore[i][j] = true;
            nNoMore++;
          }
        }
      }
      for (int i = 0; i < levelRows; i++)
        for (int j = 0; j < levelCols; j++) lastCellTotals[level].push_back(nTotal[i][j]);

      while (nToDistribute > 0 && nNoMore < nCells) {
        int nNewFeaturesCell = nfeaturesCell + ceil((float)nToDistribute / (nCells - nNoMore));
        nToDistribute = 0;
        for (int i = 0; i < levelRows; i++) {
          for (int j = 0; j < levelCols; j++) {
            if (!bNoMore[i][j]) {
              if (nTotal[i][j] > nNewFeaturesCell) {
                nToRetain[i][j] = nNewFeaturesCell;
                bNoMore[i][j] = false;
              } else {
                nToRetain[i][j] = nTotal[i][j];
                nToDistribute += nNewFeaturesCell - nTotal[i][j];
                bNoMore[i][j] = true;
                nNoMore++;
              }
            }
          }
        }
      }

      std::vector<KeyPoint>& keypoints = allKeypoints[level];
      keypoints.reserve(nDesiredFeatures * 2);
      const int scaledPatchSize = PATCH_SIZE * mvScaleFactor[level];

      for (int i = 0; i < levelRows; i++) {
        for (int j = 0; j < levelCols; j++) {
          std::vector<KeyPoint>& keysCell = cellKeyPoints[i][j];
          retainBest(keysCell, nToRetain[i][j]);
          if ((int)keysCell.size() > nToRetain[i][j]) keysCell.resize(nToRetain[i][j]);
          for (size_t k = 0, kend = keysCell.size(); k < kend; k++) {
            keysCell[k].x += iniXCol[j];
            keysCell[k].y += iniYRow[i];
            keysCell[k].octave = level;
            keysCell[k].size = scaledPatchSize;
            keypoints.push_back(keysCell[k]);
          }
        }
      }
      if ((int)keypoints.size() > nDesiredFeatures) {
        retainBest(keypoints, nDesiredFeatures);
        keypoints.resize(nDesiredFeatures);
      }
    }
    const double ta = now_ns();
    for (int level = 0; level < nlevels; ++level) {
      View img = pyr[level].roi();
      for (auto& kp : allKeypoints[level]) kp.angle = IC_Angle(img, kp.x, kp.y);
    }
    stage_ns[3] = now_ns() - ta;
  }

  // src/ORBextractor.cc:106-143
  void computeOrbDescriptor(const KeyPoint& kpt, const View& img, const Point* pat, uint8_t* desc) const {
    const float factorPI = (float)(M_PI / 180.f);
    float angle = (float)kpt.angle * factorPI;
    float a = (float)cosf(angle), b = (float)sinf(angle);
    const uint8_t* center = img.ptr(cvRound(kpt.y)) + cvRound(kpt.x);
    const int step = img.step;
#define GET_VALUE(idx) center[cvRound(pat[idx].x * b + pat[idx].y * a) * step + cvRound(pat[idx].x * a - pat[idx].y * b)]
    for (int i = 0; i < 32; ++i, pat += 16) {
      int val = 0;
      for (int k = 0; k < 8; k++) {
        int t0 = GET_VALUE(2 * k), t1 = GET_VALUE(2 * k + 1);
        val |= (t0 < t1) << k;
      }
      desc[i] = (uint8_t)val;
    }
#undef GET_VALUE
  }

  // src/ORBextractor.cc:620-678
  // wall-clock per stage of the last extract() (bench.py's cpu_baseline: SURVEY §8d asks for per-stage medians):
  // 0 pyramid, 1 FAST + NMS, 2 quota loop + retainBest + assembly, 3 IC_Angle, 4 GaussianBlur, 5 rBRIEF
  double stage_ns[6] = {0, 0, 0, 0, 0, 0};
  static double now_ns() {
    return (double)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
  }

  int extract(const View& image, std::vector<KeyPoint>& keypointsOut, std::vector<uint8_t>& descriptors) {
    keypointsOut.clear();
    descriptors.clear();
    for (double& v : stage_ns) v = 0;
    if (image.cols <= 0 || image.rows <= 0) return 0;
    double t0 = now_ns();
    ComputePyramid(image);
    stage_ns[0] = now_ns() - t0;
    std::vector<std::vector<KeyPoint>> allKeypoints;
    t0 = now_ns();
    ComputeKeyPoints(allKeypoints);
    stage_ns[2] = now_ns() - t0 - stage_ns[1] - stage_ns[3];
    lastLevelKeypoints = allKeypoints;
    lastBlurred.assign(nlevels, {});

    int nkeypoints = 0;
    for (int level = 0; level < nlevels; ++level) nkeypoints += (int)allKeypoints[level].size();
    descriptors.assign((size_t)nkeypoints * 32, 0);
    keypointsOut.reserve(nkeypoints);

    int offset = 0;
    for (int level = 0; level < nlevels; ++level) {
      std::vector<KeyPoint>& keypoints = allKeypoints[level];
      int n = (int)keypoints.size();
      if (n == 0) continue;
      Level& L = pyr[level];
      std::vector<uint8_t>& work = lastBlurred[level];
      work.resize((size_t)L.w * L.h);
      View wv{work.data(), L.w, L.h, L.w};
      View src = L.roi();
      for (int y = 0; y < L.h; y++) memcpy(wv.ptr(y), src.ptr(y), L.w);  // clone(): compact
      const double tb = now_ns();
      gaussianBlur7(wv, wv);
      const double td = now_ns();
      for (int i = 0; i < n; i++) computeOrbDescriptor(keypoints[i], wv, pattern.data(), &descriptors[(size_t)(offset + i) * 32]);
      stage_ns[4] += td - tb;
      stage_ns[5] += now_ns() - td;
      offset += n;
      if (level != 0) {
        float scale = mvScaleFactor[level];
        for (auto& kp : keypoints) {
          kp.x *= scale;
          kp.y *= scale;
        }
      }
      keypointsOut.insert(keypointsOut.end(), keypoints.begin(), keypoints.end());
    }
    return nkeypoints;
  }
};

}  // namespace orc

// ------------------------------------------------------------------------------------------
// C entry points (ctypes)
// ------------------------------------------------------------------------------------------
using namespace orc;

extern "C" {

void* orc_orb_create(int nfeatures, float scaleFactor, int nlevels, int thFAST) {
  return new ORBextractor(nfeatures, scaleFactor, nlevels, thFAST);
}
void orc_orb_destroy(void* h) { delete (ORBextractor*)h; }

void orc_orb_tables(void* h, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2, int* quota, int* umax16) {
  ORBextractor* e = (ORBextractor*)h;
  for (int i = 0; i < e->nlevels; i++) {
    if (sf) sf[i] = e->mvScaleFactor[i];
    if (inv_sf) inv_sf[i] = e->mvInvScaleFactor[i];
    if (sigma2) sigma2[i] = e->mvLevelSigma2[i];
    if (inv_sigma2) inv_sigma2[i] = e->mvInvLevelSigma2[i];
    if (quota) quota[i] = e->mnFeaturesPerLevel[i];
  }
  if (umax16)
    for (int i = 0; i < 16; i++) umax16[i] = e->umax[i];
}

int orc_orb_extract(void* h, const uint8_t* img, int w, int hh, int stride, void* kps_out, uint8_t* desc_out, int cap) {
  ORBextractor* e = (ORBextractor*)h;
  std::vector<KeyPoint> kps;
  std::vector<uint8_t> desc;
  int n = e->extract(View{(uint8_t*)img, w, hh, stride}, kps, desc);
  int m = std::min(n, cap);
  if (kps_out && m > 0) memcpy(kps_out, kps.data(), (size_t)m * sizeof(KeyPoint));
  if (desc_out && m > 0) memcpy(desc_out, desc.data(), (size_t)m * 32);
  return n;
}

void orc_orb_stage_ns(void* h, double* out6) {
  for (int i = 0; i < 6; i++) out6[i] = ((ORBextractor*)h)->stage_ns[i];
}

// pyramid level access after extract(): padded=1 -> whole (w+38)x(h+38) buffer
int orc_orb_level_info(void* h, int level, int* w, int* hh) {
  ORBextractor* e = (ORBextractor*)h;
  if (level < 0 || level >= (int)e->pyr.size()) return -1;
  *w = e->pyr[level].w;
  *hh = e->pyr[level].h;
  return 0;
}
int orc_orb_level_copy(void* h, int level, int padded, uint8_t* out, int out_stride) {
  ORBextractor* e = (ORBextractor*)h;
  if (level < 0 || level >= (int)e->pyr.size()) return -1;
  Level& L = e->pyr[level];
  View v = padded ? L.whole() : L.roi();
  for (int y = 0; y < v.rows; y++) memcpy(out + (size_t)y * out_stride, v.ptr(y), v.cols);
  return 0;
}
int orc_orb_blurred_copy(void* h, int level, uint8_t* out, int out_stride) {
  ORBextractor* e = (ORBextractor*)h;
  if (level < 0 || level >= (int)e->lastBlurred.size() || e->lastBlurred[level].empty()) return -1;
  Level& L = e->pyr[level];
  for (int y = 0; y < L.h; y++) memcpy(out + (size_t)y * out_stride, e->lastBlurred[level].data() + (size_t)y * L.w, L.w);
  return 0;
}
int orc_orb_level_keypoints(void* h, int level, void* kps_out, int cap) {
  ORBextractor* e = (ORBextractor*)h;
  if (level < 0 || level >= (int)e->lastLevelKeypoints.size()) return -1;
  auto& v = e->lastLevelKeypoints[level];
  int m = std::min((int)v.size(), cap);
  if (m > 0) memcpy(kps_out, v.data(), (size_t)m * sizeof(KeyPoint));
  return (int)v.size();
}
int orc_orb_cell_totals(void* h, int level, int* out, int cap) {
  ORBextractor* e = (ORBextractor*)h;
  if (level < 0 || level >= (int)e->lastCellTotals.size()) return -1;
  auto& v = e->lastCellTotals[level];
  for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
  return (int)v.size();
}

// ---- stage-level entry points for known-answer tests ----
int orc_fast(const uint8_t* img, int w, int h, int stride, int threshold, int nonmax, void* kps_out, int cap) {
  std::vector<KeyPoint> k;
  FAST(View{(uint8_t*)img, w, h, stride}, k, threshold, nonmax != 0);
  int m = std::min((int)k.size(), cap);
  if (m > 0) memcpy(kps_out, k.data(), (size_t)m * sizeof(KeyPoint));
  return (int)k.size();
}
int orc_fast_score(const uint8_t* center, int stride, int threshold) { return fastCornerScore(center, stride, threshold); }
void orc_resize_linear(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride) {
  resizeLinear(View{(uint8_t*)src, sw, sh, sstride}, View{dst, dw, dh, dstride});
}
void orc_border101(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride, int b) {
  copyMakeBorder101(View{(uint8_t*)src, w, h, sstride}, View{dst, w + 2 * b, h + 2 * b, dstride}, b);
}
void orc_blur7(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride) {
  gaussianBlur7(View{(uint8_t*)src, w, h, sstride}, View{dst, w, h, dstride});
}
void orc_gauss_taps(int* k7) { gaussianKernel7Fixed(k7); }
float orc_fast_atan2(float y, float x) { return fastAtan2(y, x); }
int orc_retain_best(void* kps_inout, int n, int n_points) {
  std::vector<KeyPoint> v((KeyPoint*)kps_inout, (KeyPoint*)kps_inout + n);
  retainBest(v, n_points);
  memcpy(kps_inout, v.data(), v.size() * sizeof(KeyPoint));
  return (int)v.size();
}
float orc_ic_angle(void* h, const uint8_t* img, int stride, float x, float y) {
  ORBextractor* e = (ORBextractor*)h;
  return e->IC_Angle(View{(uint8_t*)img, 0, 0, stride}, x, y);
}
void orc_brief(void* h, const uint8_t* blurred, int stride, float x, float y, float angle, uint8_t* desc32) {
  ORBextractor* e = (ORBextractor*)h;
  KeyPoint k{x, y, 31.f, angle, 0.f, 0, -1};
  e->computeOrbDescriptor(k, View{(uint8_t*)blurred, 0, 0, stride}, e->pattern.data(), desc32);
}

}  // extern "C"
