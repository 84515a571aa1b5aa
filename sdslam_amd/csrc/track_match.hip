// Frame bucket grid + ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono) on MI355X.
//
// Replaces, per frame of a batch:
//   Frame::AssignFeaturesToGrid / PosInGrid      reference src/Frame.cc:179-192, 323-332
//   Frame::GetFeaturesInArea                     src/Frame.cc:271-321
//   ORBmatcher::DescriptorDistance               src/ORBmatcher.cc:1459-1473  (__popcll on 4 x u64)
//   ORBmatcher::SearchByProjection(Frame,Frame)  src/ORBmatcher.cc:946-1075
//   ORBmatcher::ComputeThreeMaxima               src/ORBmatcher.cc:1423-1454
//
// One workgroup per frame.  The 64x48 bucket grid is built by sorting (cell, index) keys in LDS
// with cell = ix*48 + iy, so the reference's candidate order (ix outer, iy inner, ascending
// index inside a cell) is, for every grid column ix, one contiguous run of the sorted array.
// The assignment loop is sequential by definition (a keypoint claimed by an earlier map point
// with Observations() > 0 is skipped by later ones; otherwise later points overwrite), so one
// wavefront walks the last frame's points in order and parallelises each point's window:
// lanes take candidates, compute Hamming distances, and a wave-min over (dist, order, index)
// yields the reference's "first strict minimum".  Stereo / RGB-D gates (mvuRight, bForward /
// bBackward octave windows) follow src/ORBmatcher.cc:965-966,999-1004,1020-1025; keypoints are
// the undistorted ones (Frame::mvKeysUn).
#include <hip/hip_runtime.h>

#include "orb_internal.h"
#include "track_internal.h"

namespace sd {

#define MT_MAXKP 2048
#define GRID_COLS 64
#define GRID_ROWS 48
#define TH_HIGH 100
#define HISTO_LENGTH 30

__global__ __launch_bounds__(256) void k_match(const sd_keypoint* __restrict__ kps_all, const uint8_t* __restrict__ desc_all,
                                               const int32_t* __restrict__ nkp_all, TrackBuffers tb, TrackCam cam,
                                               const float* __restrict__ sf, float th, int mono, int check_ori) {
  __shared__ uint32_t s_key[MT_MAXKP];
  __shared__ uint16_t s_cstart[GRID_COLS * GRID_ROWS + 2];
  __shared__ int s_match[MT_MAXKP];
  __shared__ uint32_t s_ev[MT_MAXKP];
  __shared__ int s_hist[HISTO_LENGTH];
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int cap = tb.kp_cap;
  const sd_keypoint* kps = kps_all + (size_t)f * cap;
  const uint8_t* desc = desc_all + (size_t)f * cap * 32;
  const int N = min(nkp_all[f], min(cap, MT_MAXKP));
  const float invW = (float)GRID_COLS / (float)(cam.max_x - cam.min_x);   // mfGridElementWidthInv
  const float invH = (float)GRID_ROWS / (float)(cam.max_y - cam.min_y);

  // ---- AssignFeaturesToGrid: key = cell << 11 | index (PosInGrid uses round())
  for (int i = tid; i < MT_MAXKP; i += 256) {
    uint32_t key = 0xFFFFFFFFu;
    if (i < N) {
      const float x = kps[i].x, y = kps[i].y;
      const int posX = (int)roundf((x - cam.min_x) * invW);
      const int posY = (int)roundf((y - cam.min_y) * invH);
      if (!(posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS)) key = ((uint32_t)(posX * GRID_ROWS + posY) << 11) | (uint32_t)i;
    }
    s_key[i] = key;
    s_match[i] = -1;   // CurrentFrame.mvpMapPoints filled with NULL (src/Tracking.cc:676)
  }
  __syncthreads();
  for (int k = 2; k <= MT_MAXKP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < MT_MAXKP; i += 256) {
        int ixj = i ^ j;
        if (ixj > i) {
          uint32_t a = s_key[i], b = s_key[ixj];
          bool up = (i & k) == 0;
          if ((a > b) == up) {
            s_key[i] = b;
            s_key[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  for (int c = tid; c <= GRID_COLS * GRID_ROWS; c += 256) {
    const uint32_t target = (uint32_t)c << 11;
    int lo = 0, hi = MT_MAXKP;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (s_key[mid] < target) lo = mid + 1;
      else hi = mid;
    }
    s_cstart[c] = (uint16_t)lo;
  }
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  __syncthreads();
  if (tid >= 64) return;   // one wavefront runs the order-dependent assignment loop

  const int M = tb.max_points;
  const uint8_t* valid = tb.valid + (size_t)f * M;
  const double* Xw = tb.Xw + (size_t)f * M * 3;
  const uint8_t* mp_desc = tb.mp_desc + (size_t)f * M * 32;
  const int32_t* l_oct = tb.octave + (size_t)f * M;
  const float* l_ang = tb.angle + (size_t)f * M;
  const int32_t* l_obs = tb.obs + (size_t)f * M;
  const int n_last = min(tb.n_last[f], M);
  const double* Tc = tb.Tcur + (size_t)f * 16;   // column-major
  double R[3][3], t[3];
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) R[r][c] = Tc[c * 4 + r];
    t[r] = Tc[12 + r];
  }
  int nmatches = 0, nev = 0;
  const float factor = 1.0f / HISTO_LENGTH;
  const float* uright = tb.uright + (size_t)f * cap;
  // bForward / bBackward: tlc = Rlw * (-Rcw^T tcw) + tlw compared with the baseline mb
  bool bForward = false, bBackward = false;
  if (!mono) {
    const double* Tl = tb.Tref + (size_t)f * 16;
    double twc[3], tlc2;
    for (int i = 0; i < 3; i++) twc[i] = (-R[0][i]) * t[0] + (-R[1][i]) * t[1] + (-R[2][i]) * t[2];
    tlc2 = (Tl[0 * 4 + 2] * twc[0] + Tl[1 * 4 + 2] * twc[1] + Tl[2 * 4 + 2] * twc[2]) + Tl[12 + 2];
    bForward = tlc2 > cam.mb;
    bBackward = -tlc2 > cam.mb;
  }

  for (int i = 0; i < n_last; i++) {
    if (!valid[i]) continue;
    const double xw = Xw[(size_t)i * 3], yw = Xw[(size_t)i * 3 + 1], zw = Xw[(size_t)i * 3 + 2];
    const double X = (R[0][0] * xw + R[0][1] * yw + R[0][2] * zw) + t[0];
    const double Y = (R[1][0] * xw + R[1][1] * yw + R[1][2] * zw) + t[1];
    const double Z = (R[2][0] * xw + R[2][1] * yw + R[2][2] * zw) + t[2];
    const float xc = (float)X, yc = (float)Y;
    const float invzc = (float)(1.0 / Z);
    if (invzc < 0) continue;
    const float u = cam.ffx * xc * invzc + cam.fcx;
    const float v = cam.ffy * yc * invzc + cam.fcy;
    if (u < cam.min_x || u > cam.max_x) continue;
    if (v < cam.min_y || v > cam.max_y) continue;
    const int nLastOctave = l_oct[i];
    const float radius = th * sf[nLastOctave];
    const int minLevel = bForward ? nLastOctave : (bBackward ? 0 : nLastOctave - 1);
    const int maxLevel = bForward ? -1 : (bBackward ? nLastOctave : nLastOctave + 1);
    // GetFeaturesInArea(u, v, radius, minLevel, maxLevel)
    const int nMinCellX = max(0, (int)floorf((u - cam.min_x - radius) * invW));
    if (nMinCellX >= GRID_COLS) continue;
    const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf((u - cam.min_x + radius) * invW));
    if (nMaxCellX < 0) continue;
    const int nMinCellY = max(0, (int)floorf((v - cam.min_y - radius) * invH));
    if (nMinCellY >= GRID_ROWS) continue;
    const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((v - cam.min_y + radius) * invH));
    if (nMaxCellY < 0) continue;
    const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    const unsigned long long* dm = (const unsigned long long*)(mp_desc + (size_t)i * 32);
    const unsigned long long d0 = dm[0], d1 = dm[1], d2 = dm[2], d3 = dm[3];
    uint32_t best = 0x7FFFFFFFu;
    int seq0 = 0;
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
      const int a = s_cstart[ix * GRID_ROWS + nMinCellY], b = s_cstart[ix * GRID_ROWS + nMaxCellY + 1];
      for (int e0 = a; e0 < b; e0 += 64) {
        const int e = e0 + lane;
        if (e < b) {
          const int idx = s_key[e] & 2047;
          const sd_keypoint kp = kps[idx];
          bool okc = true;
          if (bCheckLevels) {
            if (kp.octave < minLevel) okc = false;
            if (maxLevel >= 0 && kp.octave > maxLevel) okc = false;
          }
          const float distx = kp.x - u, disty = kp.y - v;
          if (!(fabsf(distx) < radius && fabsf(disty) < radius)) okc = false;
          if (okc) {
            const int m = s_match[idx];
            bool claimed = (m >= 0) && (l_obs[m] > 0);
            const float ur2 = uright[idx];
            if (!claimed && ur2 > 0) {   // stereo consistency gate (src/ORBmatcher.cc:1020-1025)
              const float ur = u - cam.bf * invzc;
              const float er = fabsf(ur - ur2);
              if (er > radius) claimed = true;
            }
            if (!claimed) {
              const unsigned long long* dk = (const unsigned long long*)(desc + (size_t)idx * 32);
              const int dist = __popcll(dk[0] ^ d0) + __popcll(dk[1] ^ d1) + __popcll(dk[2] ^ d2) + __popcll(dk[3] ^ d3);
              // NB: candidates failing the window test do not advance the reference's vIndices2
              // order relative to each other, so the sorted-array position is a valid order key
              const uint32_t key = ((uint32_t)dist << 22) | ((uint32_t)(seq0 + (e - a)) << 11) | (uint32_t)idx;
              best = min(best, key);
            }
          }
        }
      }
      seq0 += b - a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o));
    if (best == 0x7FFFFFFFu) continue;
    const int bestDist = best >> 22;
    const int bestIdx2 = best & 2047;
    if (bestDist <= TH_HIGH) {
      if (lane == 0) s_match[bestIdx2] = i;
      nmatches++;
      if (check_ori) {
        float rot = l_ang[i] - kps[bestIdx2].angle;
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        if (lane == 0) {
          s_ev[nev] = ((uint32_t)bin << 16) | (uint32_t)bestIdx2;
          s_hist[bin]++;
        }
        nev++;
      }
    }
  }
  // ---- rotation consistency: keep the three dominant 30-degree bins (10 % rule)
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    int max1 = 0, max2 = 0, max3 = 0;
    for (int b = 0; b < HISTO_LENGTH; b++) {
      const int s = s_hist[b];
      if (s > max1) {
        max3 = max2; max2 = max1; max1 = s;
        ind3 = ind2; ind2 = ind1; ind1 = b;
      } else if (s > max2) {
        max3 = max2; max2 = s;
        ind3 = ind2; ind2 = b;
      } else if (s > max3) {
        max3 = s;
        ind3 = b;
      }
    }
    if (max2 < 0.1f * (float)max1) {
      ind2 = -1;
      ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
      ind3 = -1;
    }
    for (int e = 0; e < nev; e++) {   // uniform loop; lane 0 applies
      const uint32_t ev = s_ev[e];
      const int bin = ev >> 16;
      if (bin != ind1 && bin != ind2 && bin != ind3) {
        if (lane == 0) s_match[ev & 0xffff] = -1;
        nmatches--;
      }
    }
  }
  int32_t* out = tb.cur_match + (size_t)f * cap;
  for (int i = lane; i < cap; i += 64) out[i] = i < MT_MAXKP ? s_match[i] : -1;
  if (lane == 0) tb.n_matches[f] = nmatches;
}

// Frame::ComputeStereoFromRGBD (src/Frame.cc:399-417): d = imDepth.at<float>(kp.pt.y, kp.pt.x) at the
// DISTORTED keypoint (coordinates truncated to int); mvDepth = d, mvuRight = kpU.pt.x - mbf / d if d > 0.
__global__ void k_stereo_from_depth(const sd_keypoint* __restrict__ kps, const sd_keypoint* __restrict__ kps_un,
                                    const int32_t* __restrict__ nkp, TrackBuffers tb, TrackCam cam, const float* __restrict__ depth,
                                    int w, int h, int stride, size_t frame_stride) {
  const int f = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, cap = tb.kp_cap;
  if (i >= cap) return;
  float ur = -1.f, dd = -1.f;
  if (i < nkp[f]) {
    const sd_keypoint kp = kps[(size_t)f * cap + i];
    const int v = (int)kp.y, u = (int)kp.x;
    if (u >= 0 && v >= 0 && u < w && v < h) {
      const float d = depth[(size_t)f * frame_stride + (size_t)v * stride + u];
      if (d > 0) {
        dd = d;
        ur = kps_un[(size_t)f * cap + i].x - cam.bf / d;
      }
    }
  }
  tb.uright[(size_t)f * cap + i] = ur;
  tb.depth[(size_t)f * cap + i] = dd;
}

int launch_stereo_from_depth(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_depth, int w, int h,
                             int stride_elems, size_t frame_stride_elems, int n_frames, hipStream_t s) {
  hipLaunchKernelGGL(k_stereo_from_depth, dim3((tb.kp_cap + 255) / 256, n_frames), dim3(256), 0, s, cur->d_kps,
                     (cur->have_dist ? cur->d_kps_un : cur->d_kps), cur->d_nout, tb, cam, d_depth, w, h, stride_elems, frame_stride_elems);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

int launch_match(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_sf, int n_frames, float th,
                 int mono, int check_ori, hipStream_t s) {
  hipLaunchKernelGGL(k_match, dim3(n_frames), dim3(256), 0, s, (cur->have_dist ? cur->d_kps_un : cur->d_kps), cur->d_desc, cur->d_nout, tb, cam, d_sf, th, mono,
                     check_ori);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

}  // namespace sd
