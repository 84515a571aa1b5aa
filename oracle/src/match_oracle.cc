// ORACLE (test infrastructure only -- never linked into the product path).
//
// CPU restatement of the tracking-side matcher and the Frame grid it queries:
//   Frame::AssignFeaturesToGrid / PosInGrid   reference src/Frame.cc:179-192, 323-332
//   Frame::GetFeaturesInArea                  src/Frame.cc:271-321
//   ORBmatcher::DescriptorDistance            src/ORBmatcher.cc:1459-1473
//   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)   src/ORBmatcher.cc:946-1075
//   ORBmatcher::ComputeThreeMaxima            src/ORBmatcher.cc:1423-1454
//   ORBmatcher::SearchByPoints(KeyFrame*, KeyFrame*, matches)         src/ORBmatcher.cc:1209-1301 (brute-force Hamming)
// MapPoint pointers are flattened to indices into the last frame's arrays: cur_match[i2] = index
// of the last-frame map point assigned to current keypoint i2, or -1 (NULL).
// Quirks kept: histogram factor 1/30, PosInGrid rounds while
// GetFeaturesInArea floors/ceils, first strict minimum wins, later points overwrite earlier
// assignments when the earlier point has Observations() == 0.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

struct KeyPoint { float x, y, size, angle, response; int32_t octave, class_id; };

static const int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64;   // src/Frame.h:34-35
static const int TH_HIGH = 100, TH_LOW = 50, HISTO_LENGTH = 30;   // src/ORBmatcher.cc:36-38

static int DescriptorDistance(const uint8_t* a, const uint8_t* b) {
  const int32_t* pa = (const int32_t*)a;
  const int32_t* pb = (const int32_t*)b;
  int dist = 0;
  for (int i = 0; i < 8; i++, pa++, pb++) {
    unsigned int v = *pa ^ *pb;
    v = v - ((v >> 1) & 0x55555555);
    v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
    dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
  }
  return dist;
}

struct FrameGrid {
  int N;
  const KeyPoint* keysUn;
  float mnMinX, mnMaxX, mnMinY, mnMaxY, invW, invH;
  std::vector<size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];

  void build() {
    invW = static_cast<float>(FRAME_GRID_COLS) / static_cast<float>(mnMaxX - mnMinX);
    invH = static_cast<float>(FRAME_GRID_ROWS) / static_cast<float>(mnMaxY - mnMinY);
    for (int i = 0; i < N; i++) {
      const KeyPoint& kp = keysUn[i];
      int posX = round((kp.x - mnMinX) * invW);
      int posY = round((kp.y - mnMinY) * invH);
      if (posX < 0 || posX >= FRAME_GRID_COLS || posY < 0 || posY >= FRAME_GRID_ROWS) continue;
      mGrid[posX][posY].push_back(i);
    }
  }

  std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel, const int maxLevel) const {
    std::vector<size_t> vIndices;
    const int nMinCellX = std::max(0, static_cast<int>(floor((x - mnMinX - r) * invW)));
    if (nMinCellX >= FRAME_GRID_COLS) return vIndices;
    const int nMaxCellX = std::min(static_cast<int>(FRAME_GRID_COLS - 1), static_cast<int>(ceil((x - mnMinX + r) * invW)));
    if (nMaxCellX < 0) return vIndices;
    const int nMinCellY = std::max(0, static_cast<int>(floor((y - mnMinY - r) * invH)));
    if (nMinCellY >= FRAME_GRID_ROWS) return vIndices;
    const int nMaxCellY = std::min(static_cast<int>(FRAME_GRID_ROWS - 1), static_cast<int>(ceil((y - mnMinY + r) * invH)));
    if (nMaxCellY < 0) return vIndices;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
        const std::vector<size_t>& vCell = mGrid[ix][iy];
        for (size_t j = 0, jend = vCell.size(); j < jend; j++) {
          const KeyPoint& kpUn = keysUn[vCell[j]];
          if (bCheckLevels) {
            if (kpUn.octave < minLevel) continue;
            if (maxLevel >= 0)
              if (kpUn.octave > maxLevel) continue;
          }
          const float distx = kpUn.x - x;
          const float disty = kpUn.y - y;
          if (fabs(distx) < r && fabs(disty) < r) vIndices.push_back(vCell[j]);
        }
      }
    }
    return vIndices;
  }
};

static void ComputeThreeMaxima(std::vector<int>* histo, const int L, int& ind1, int& ind2, int& ind3) {
  int max1 = 0, max2 = 0, max3 = 0;
  for (int i = 0; i < L; i++) {
    const int s = histo[i].size();
    if (s > max1) {
      max3 = max2; max2 = max1; max1 = s;
      ind3 = ind2; ind2 = ind1; ind1 = i;
    } else if (s > max2) {
      max3 = max2; max2 = s;
      ind3 = ind2; ind2 = i;
    } else if (s > max3) {
      max3 = s;
      ind3 = i;
    }
  }
  if (max2 < 0.1f * (float)max1) {
    ind2 = -1;
    ind3 = -1;
  } else if (max3 < 0.1f * (float)max1) {
    ind3 = -1;
  }
}

}  // namespace orc

using namespace orc;

extern "C" {

// Poses: 16 doubles column-major.  valid[i] != 0 <=> LastFrame.mvpMapPoints[i] != NULL && !mvbOutlier[i].
// cur_match (N, in/out): -1 = NULL.  Returns nmatches (may count a keypoint twice exactly like the
// reference does when an assignment is overwritten).
int orc_search_by_projection(int N, const void* keysUn_, const uint8_t* desc, const float* uRight, const float* scaleFactors,
                             float minX, float maxX, float minY, float maxY, float fx, float fy, float cx, float cy, float mbf,
                             float mb, const double* Tcw_cm, const double* Tlw_cm, int M, const uint8_t* valid, const double* Xw,
                             const uint8_t* mp_desc, const int* last_octave, const float* last_angle, const int* mp_obs, float th,
                             int bMono, int checkOri, int* cur_match) {
  const KeyPoint* keysUn = (const KeyPoint*)keysUn_;
  FrameGrid G;
  G.N = N;
  G.keysUn = keysUn;
  G.mnMinX = minX; G.mnMaxX = maxX; G.mnMinY = minY; G.mnMaxY = maxY;
  G.build();

  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  const float factor = 1.0f / HISTO_LENGTH;
  auto at = [](const double* T, int r, int c) { return T[c * 4 + r]; };
  double Rcw[3][3], tcw[3], Rlw[3][3], tlw[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) {
      Rcw[i][j] = at(Tcw_cm, i, j);
      Rlw[i][j] = at(Tlw_cm, i, j);
    }
    tcw[i] = at(Tcw_cm, i, 3);
    tlw[i] = at(Tlw_cm, i, 3);
  }
  double twc[3], tlc[3];
  for (int i = 0; i < 3; i++) twc[i] = (-Rcw[0][i]) * tcw[0] + (-Rcw[1][i]) * tcw[1] + (-Rcw[2][i]) * tcw[2];
  for (int i = 0; i < 3; i++) tlc[i] = (Rlw[i][0] * twc[0] + Rlw[i][1] * twc[1] + Rlw[i][2] * twc[2]) + tlw[i];
  const bool bForward = tlc[2] > mb && !bMono;
  const bool bBackward = -tlc[2] > mb && !bMono;

  for (int i = 0; i < M; i++) {
    if (!valid[i]) continue;
    const double* x3Dw = Xw + 3 * i;
    double x3Dc[3];
    for (int r = 0; r < 3; r++) x3Dc[r] = (Rcw[r][0] * x3Dw[0] + Rcw[r][1] * x3Dw[1] + Rcw[r][2] * x3Dw[2]) + tcw[r];
    const float xc = x3Dc[0];
    const float yc = x3Dc[1];
    const float invzc = 1.0 / x3Dc[2];
    if (invzc < 0) continue;
    float u = fx * xc * invzc + cx;
    float v = fy * yc * invzc + cy;
    if (u < minX || u > maxX) continue;
    if (v < minY || v > maxY) continue;
    int nLastOctave = last_octave[i];
    float radius = th * scaleFactors[nLastOctave];
    std::vector<size_t> vIndices2;
    if (bForward) vIndices2 = G.GetFeaturesInArea(u, v, radius, nLastOctave, -1);
    else if (bBackward) vIndices2 = G.GetFeaturesInArea(u, v, radius, 0, nLastOctave);
    else vIndices2 = G.GetFeaturesInArea(u, v, radius, nLastOctave - 1, nLastOctave + 1);
    if (vIndices2.empty()) continue;
    const uint8_t* dMP = mp_desc + 32 * (size_t)i;
    int bestDist = 256;
    int bestIdx2 = -1;
    for (size_t k = 0; k < vIndices2.size(); k++) {
      const size_t i2 = vIndices2[k];
      if (cur_match[i2] >= 0)
        if (mp_obs[cur_match[i2]] > 0) continue;
      if (uRight && uRight[i2] > 0) {
        const float ur = u - mbf * invzc;
        const float er = fabs(ur - uRight[i2]);
        if (er > radius) continue;
      }
      const int dist = DescriptorDistance(dMP, desc + 32 * i2);
      if (dist < bestDist) {
        bestDist = dist;
        bestIdx2 = i2;
      }
    }
    if (bestDist <= TH_HIGH) {
      cur_match[bestIdx2] = i;
      nmatches++;
      if (checkOri) {
        float rot = last_angle[i] - keysUn[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = round(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        rotHist[bin].push_back(bestIdx2);
      }
    }
  }
  if (checkOri) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i != ind1 && i != ind2 && i != ind3) {
        for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
          cur_match[rotHist[i][j]] = -1;
          nmatches--;
        }
      }
    }
  }
  return nmatches;
}

int orc_descriptor_distance(const uint8_t* a, const uint8_t* b) { return DescriptorDistance(a, b); }

// ORBmatcher::SearchByPoints(currentKF, pKF, matches) (src/ORBmatcher.cc:1209-1301): brute-force Hamming over the map
// points of two keyframes.  has_mp1 / has_mp2 = "GetMapPointMatches()[i] != NULL && !isBad()"; matches12[idx1] = index of
// the pKF keypoint whose map point is assigned (the reference stores vpMapPoints2[bestIdx2]) or -1.
int orc_search_by_points(int N1, const void* keysUn1_, const uint8_t* desc1, const uint8_t* has_mp1, int N2, const void* keysUn2_,
                         const uint8_t* desc2, const uint8_t* has_mp2, float mfNNratio, int mbCheckOrientation, int* matches12) {
  const KeyPoint* vKeysUn1 = (const KeyPoint*)keysUn1_;
  const KeyPoint* vKeysUn2 = (const KeyPoint*)keysUn2_;
  int nmatches = 0;
  std::vector<int> rotHist[HISTO_LENGTH];
  for (int i = 0; i < HISTO_LENGTH; i++) rotHist[i].reserve(500);
  const float factor = 1.0f / HISTO_LENGTH;
  for (int i = 0; i < N1; i++) matches12[i] = -1;
  std::vector<bool> vbMatched2(N2, false);
  for (int idx1 = 0; idx1 < N1; idx1++) {
    if (!has_mp1[idx1]) continue;
    const uint8_t* d1 = desc1 + (size_t)idx1 * 32;
    int bestDist1 = 256;
    int bestIdx2 = -1;
    int bestDist2 = 256;
    for (int idx2 = 0; idx2 < N2; idx2++) {
      if (!has_mp2[idx2] || vbMatched2[idx2]) continue;
      const uint8_t* d2 = desc2 + (size_t)idx2 * 32;
      int dist = DescriptorDistance(d1, d2);
      if (dist < bestDist1) {
        bestDist2 = bestDist1;
        bestDist1 = dist;
        bestIdx2 = idx2;
      } else if (dist < bestDist2) {
        bestDist2 = dist;
      }
    }
    if (bestDist1 < TH_LOW) {
      if (static_cast<float>(bestDist1) < mfNNratio * static_cast<float>(bestDist2)) {
        matches12[idx1] = bestIdx2;
        vbMatched2[bestIdx2] = true;
        if (mbCheckOrientation) {
          float rot = vKeysUn1[idx1].angle - vKeysUn2[bestIdx2].angle;
          if (rot < 0.0) rot += 360.0f;
          int bin = round(rot * factor);
          if (bin == HISTO_LENGTH) bin = 0;
          rotHist[bin].push_back(idx1);
        }
        nmatches++;
      }
    }
  }
  if (mbCheckOrientation) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    ComputeThreeMaxima(rotHist, HISTO_LENGTH, ind1, ind2, ind3);
    for (int i = 0; i < HISTO_LENGTH; i++) {
      if (i == ind1 || i == ind2 || i == ind3) continue;
      for (size_t j = 0, jend = rotHist[i].size(); j < jend; j++) {
        matches12[rotHist[i][j]] = -1;
        nmatches--;
      }
    }
  }
  return nmatches;
}

// Frame::UndistortKeyPoints (reference src/Frame.cc:335-366): cv::undistortPoints(pts, pts, K, dist, Mat(), K)
// restated from OpenCV 3.2 cvUndistortPoints (5 fixed-point iterations, fp64 inside, f32 in/out;
// K arrives as the CV_32F matrix Converter::toCvMat builds; dist = {k1,k2,p1,p2,k3} floats).
// k1 == 0 -> copy (src/Frame.cc:336-339).
void orc_undistort_points(int n, const float* xy_in, float fxf, float fyf, float cxf, float cyf, const float* dist5, float* xy_out) {
  if (dist5[0] == 0.0f) {
    for (int i = 0; i < 2 * n; i++) xy_out[i] = xy_in[i];
    return;
  }
  double k[12] = {dist5[0], dist5[1], dist5[2], dist5[3], dist5[4], 0, 0, 0, 0, 0, 0, 0};
  const double fx = fxf, fy = fyf, cx = cxf, cy = cyf;
  const double ifx = 1. / fx, ify = 1. / fy;
  for (int i = 0; i < n; i++) {
    double x = xy_in[2 * i], y = xy_in[2 * i + 1];
    x = (x - cx) * ifx;
    y = (y - cy) * ify;
    const double x0 = x, y0 = y;
    for (int j = 0; j < 5; j++) {
      double r2 = x * x + y * y;
      double icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2);
      double deltaX = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x) + k[8] * r2 + k[9] * r2 * r2;
      double deltaY = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y + k[10] * r2 + k[11] * r2 * r2;
      x = (x0 - deltaX) * icdist;
      y = (y0 - deltaY) * icdist;
    }
    double xx = fx * x + 0.0 * y + cx;
    double yy = 0.0 * x + fy * y + cy;
    double ww = 1. / (0.0 * x + 0.0 * y + 1.0);
    xy_out[2 * i] = (float)(xx * ww);
    xy_out[2 * i + 1] = (float)(yy * ww);
  }
}

// TrackLocalMap's search: Frame::isInFrustum (src/Frame.cc:215-269) + MapPoint::PredictScale
// (src/MapPoint.cc:371-385) for every candidate local map point, then
// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:43-126) with
// mfNNratio = nnratio.  MapPoints are flattened to arrays of length M in vpMapPoints order:
//   cand[i] != 0      the point reaches isInFrustum (not bad, mnLastFrameSeen != frame id; src/Tracking.cc:916-923)
//   Xw, normal        GetWorldPos(), GetNormal()                       (3 doubles each)
//   min_dist/max_dist GetMinDistanceInvariance() / GetMaxDistanceInvariance()  (0.8 mfMin, 1.2 mfMax: floats)
//   mf_max_dist       mfMaxDistance (PredictScale)
//   mp_desc, mp_obs   GetDescriptor(), Observations()
// kp_claimed[idx] != 0: F.mvpMapPoints[idx] already holds a point with Observations() > 0.
// Unqualified log() on a float resolves to the float overload through libstdc++'s <math.h> (std::log(float)),
// which is how the restatement reads `ceil(log(ratio)/mfLogScaleFactor)` and `mfLogScaleFactor = log(mfScaleFactor)`.
// Outputs: local_match[N] = index of the local point assigned to keypoint idx or -1; in_view[M], proj[M][3] =
// (mTrackProjX, mTrackProjY, mTrackProjXR), level[M], view_cos[M] (only meaningful where in_view).  Returns nmatches.
int orc_search_local_points(int N, const void* keysUn_, const uint8_t* desc, const float* uRight, const float* scaleFactors, int nLevels,
                            float logScaleFactor, float minX, float maxX, float minY, float maxY, float fx, float fy, float cx, float cy,
                            float mbf, const double* Tcw_cm, int M, const uint8_t* cand, const double* Xw, const double* normal,
                            const float* min_dist, const float* max_dist, const float* mf_max_dist, const uint8_t* mp_desc,
                            const int* mp_obs, const uint8_t* kp_claimed, float th, float nnratio, float viewingCosLimit, int* local_match,
                            uint8_t* in_view, float* proj, int* level, float* view_cos) {
  const KeyPoint* keysUn = (const KeyPoint*)keysUn_;
  FrameGrid G;
  G.N = N;
  G.keysUn = keysUn;
  G.mnMinX = minX; G.mnMaxX = maxX; G.mnMinY = minY; G.mnMaxY = maxY;
  G.build();
  auto at = [](const double* T, int r, int c) { return T[c * 4 + r]; };
  double Rcw[3][3], tcw[3], Ow[3];
  for (int i = 0; i < 3; i++) {
    for (int j = 0; j < 3; j++) Rcw[i][j] = at(Tcw_cm, i, j);
    tcw[i] = at(Tcw_cm, i, 3);
  }
  // Frame::UpdatePoseMatrices: mOw = -mRcw.transpose() * mtcw
  for (int i = 0; i < 3; i++) Ow[i] = (-Rcw[0][i]) * tcw[0] + (-Rcw[1][i]) * tcw[1] + (-Rcw[2][i]) * tcw[2];
  for (int i = 0; i < N; i++) local_match[i] = -1;
  // ---- isInFrustum
  for (int i = 0; i < M; i++) {
    in_view[i] = 0;
    proj[3 * i] = proj[3 * i + 1] = proj[3 * i + 2] = 0;
    level[i] = 0;
    view_cos[i] = 0;
    if (!cand[i]) continue;
    const double* P = Xw + 3 * i;
    double Pc[3];
    for (int r = 0; r < 3; r++) Pc[r] = (Rcw[r][0] * P[0] + Rcw[r][1] * P[1] + Rcw[r][2] * P[2]) + tcw[r];
    const double PcX = Pc[0], PcY = Pc[1], PcZ = Pc[2];
    if (PcZ < 0.0) continue;
    const float invz = 1.0f / PcZ;
    const float u = fx * PcX * invz + cx;
    const float v = fy * PcY * invz + cy;
    if (u < minX || u > maxX) continue;
    if (v < minY || v > maxY) continue;
    const float maxDistance = max_dist[i];
    const float minDistance = min_dist[i];
    const double PO[3] = {P[0] - Ow[0], P[1] - Ow[1], P[2] - Ow[2]};
    const float dist = std::sqrt((PO[0] * PO[0] + PO[1] * PO[1]) + PO[2] * PO[2]);
    if (dist < minDistance || dist > maxDistance) continue;
    const double* Pn = normal + 3 * i;
    const float viewCos = ((PO[0] * Pn[0] + PO[1] * Pn[1]) + PO[2] * Pn[2]) / dist;
    if (viewCos < viewingCosLimit) continue;
    // MapPoint::PredictScale
    const float ratio = mf_max_dist[i] / dist;
    int nScale = std::ceil(std::log(ratio) / logScaleFactor);
    if (nScale < 0) nScale = 0;
    else if (nScale >= nLevels) nScale = nLevels - 1;
    in_view[i] = 1;
    proj[3 * i] = u;
    proj[3 * i + 2] = u - mbf * invz;
    proj[3 * i + 1] = v;
    level[i] = nScale;
    view_cos[i] = viewCos;
  }
  // ---- SearchByProjection(F, vpMapPoints, th)
  int nmatches = 0;
  const bool bFactor = th != 1.0;
  for (int iMP = 0; iMP < M; iMP++) {
    if (!in_view[iMP]) continue;   // mbTrackInView (isBad points never reach isInFrustum here)
    const int nPredictedLevel = level[iMP];
    float r = view_cos[iMP] > 0.998 ? 2.5 : 4.0;   // RadiusByViewingCos
    if (bFactor) r *= th;
    const std::vector<size_t> vIndices =
        G.GetFeaturesInArea(proj[3 * iMP], proj[3 * iMP + 1], r * scaleFactors[nPredictedLevel], nPredictedLevel - 1, nPredictedLevel);
    if (vIndices.empty()) continue;
    const uint8_t* MPdescriptor = mp_desc + 32 * iMP;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (size_t k = 0; k < vIndices.size(); k++) {
      const size_t idx = vIndices[k];
      if (local_match[idx] >= 0) {
        if (mp_obs[local_match[idx]] > 0) continue;
      } else if (kp_claimed && kp_claimed[idx]) {
        continue;
      }
      if (uRight && uRight[idx] > 0) {
        const float er = fabs(proj[3 * iMP + 2] - uRight[idx]);
        if (er > r * scaleFactors[nPredictedLevel]) continue;
      }
      const int dist = DescriptorDistance(MPdescriptor, desc + 32 * idx);
      if (dist < bestDist) {
        bestDist2 = bestDist;
        bestDist = dist;
        bestLevel2 = bestLevel;
        bestLevel = keysUn[idx].octave;
        bestIdx = idx;
      } else if (dist < bestDist2) {
        bestLevel2 = keysUn[idx].octave;
        bestDist2 = dist;
      }
    }
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && bestDist > nnratio * bestDist2) continue;
      local_match[bestIdx] = iMP;
      nmatches++;
    }
  }
  return nmatches;
}

// Frame::ComputeStereoFromRGBD -- src/Frame.cc:399-417.  imDepth.at<float>(v, u) takes int arguments:
// the float keypoint coordinates are truncated.  depth: CV_32F image with `stride` floats per row.
void orc_stereo_from_rgbd(int N, const void* keys_, const void* keysUn_, const float* depth, int stride, float mbf, float* uRight,
                          float* mvDepth) {
  const KeyPoint* keys = (const KeyPoint*)keys_;
  const KeyPoint* keysUn = (const KeyPoint*)keysUn_;
  for (int i = 0; i < N; i++) {
    uRight[i] = -1;
    mvDepth[i] = -1;
    const float v = keys[i].y, u = keys[i].x;
    const float d = depth[(size_t)(int)v * stride + (int)u];
    if (d > 0) {
      mvDepth[i] = d;
      uRight[i] = keysUn[i].x - mbf / d;
    }
  }
}

// GetFeaturesInArea exposed for known-answer tests; returns count, indices in reference order
int orc_features_in_area(int N, const void* keysUn, float minX, float maxX, float minY, float maxY, float x, float y, float r,
                         int minLevel, int maxLevel, int* out, int cap) {
  FrameGrid G;
  G.N = N;
  G.keysUn = (const KeyPoint*)keysUn;
  G.mnMinX = minX; G.mnMaxX = maxX; G.mnMinY = minY; G.mnMaxY = maxY;
  G.build();
  std::vector<size_t> v = G.GetFeaturesInArea(x, y, r, minLevel, maxLevel);
  for (size_t i = 0; i < v.size() && (int)i < cap; i++) out[i] = (int)v[i];
  return (int)v.size();
}

}  // extern "C"
