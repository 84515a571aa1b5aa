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
// Two phases.  (1) Everything that does not depend on earlier assignments -- projection, window,
// level / distance / uRight gates, Hamming distances -- runs for all points in parallel (one wave
// per point, lanes take candidates) and leaves each point's candidate keys
// (dist, order-in-vIndices2, index) in an LDS list.  (2) The assignment loop is sequential by
// definition (a keypoint claimed by an earlier map point with Observations() > 0 is skipped by
// later ones; otherwise later points overwrite): one wavefront walks the points in order, drops
// claimed candidates and takes the wave-min key = the reference's "first strict minimum".  Stereo / RGB-D gates (mvuRight, bForward /
// bBackward octave windows) follow src/ORBmatcher.cc:965-966,999-1004,1020-1025; keypoints are
// the undistorted ones (Frame::mvKeysUn).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "orb_internal.h"
#include "track_internal.h"

namespace sd {

#define MT_MAXKP 2048      // keypoints per frame supported by the 11-bit index fields
#define MT_WAVES 8
#ifndef MT_LIST_CAP
#define MT_LIST_CAP 4096   // candidate keys per frame kept in LDS (more: per-point slow path)
#endif
#define GRID_COLS 64
#define GRID_ROWS 48
#define TH_HIGH 100
#define HISTO_LENGTH 30

struct MatchGeom {   // what every per-point evaluation needs (uniform over the workgroup)
  double R[3][3], t[3];
  float th, invW, invH;
  bool bForward, bBackward;
};

// Candidate keys of last-frame point i: projection, window (GetFeaturesInArea), level / distance / uRight
// gates, Hamming distance.  Everything here is independent of earlier assignments.  All 64 lanes call.
//   mode 0: returns the number of candidates (wave-uniform)
//   mode 1: writes key = dist << 22 | order-in-vIndices2 << 11 | keypoint index to list[0 .. count)
//   mode 2: returns this lane's minimum key over the candidates not claimed by a point with observations
// *seq_total = entries of the searched cells (keys carry 11 bits of order).
// Candidates of one search window: GetFeaturesInArea(u, v, radius, minLevel, maxLevel) (src/Frame.cc:271-321) in
// the reference's order, then the stereo gate |ur - mvuRight[idx]| <= radius for keypoints with a right
// coordinate (src/ORBmatcher.cc:76-80, 1020-1025) and, in MODE 2, the "already holds a map point with
// Observations() > 0" gate.  All 64 lanes call.
//   mode 0: returns the number of candidates (wave-uniform)
//   mode 1: writes key = dist << 22 | order-in-vIndices << 11 | keypoint index to list[0 .. count)
//   mode 2: returns this lane's minimum key over the unclaimed candidates
// *seq_total = entries of the searched cells (keys carry 11 bits of order).
struct MatchLds {
  const uint32_t* s_key;      // sorted (cell << 11 | index)
  const uint16_t* s_cstart;   // first sorted position of every cell
  const int16_t* s_match;     // map point assigned to a keypoint in this call, or -1
  const uint32_t* s_obs;      // bit m: map point m has Observations() > 0
  const uint32_t* s_kclaim;   // bit idx: keypoint idx already held such a point before the call (may be null)
};

// GL = lanes that work on one point: 64 (the whole wave; `lane`, `lt` as usual, gm = ~0) or 32 (a wave handles two points,
// one per half: `lane` = lane within the half, `lt` = the lower lanes OF THE HALF and `gm` = the half's lanes, both as
// bit masks of the 64-bit wave ballot).  The halves diverge freely; a ballot only ever carries the active lanes.
// Minimum over the wave, every lane gets it: the device library's DPP reduction instead of six LDS-crossbar shuffles
// (the serial phase-2 chain of the matchers does one per point: single-frame search 1.25 -> 0.97 ms; A/B on one box at
// 1024 frames: 129.6 k vs 129.2 k frames/s).
extern "C" __device__ __attribute__((const)) unsigned int __ockl_wfred_min_u32(unsigned int);
__device__ __forceinline__ void sdsel_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return __ockl_wfred_min_u32(v); }

template <int MODE, int GL = 64>
__device__ __forceinline__ uint32_t match_window(float u, float v, float radius, int minLevel, int maxLevel, float ur,
                                                 const uint8_t* __restrict__ dmp /* 32-byte map point descriptor */,
                                                 const sd_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                 const float* __restrict__ uright, const MatchLds& S, const TrackCam& cam, float invW,
                                                 float invH, uint32_t* list, int lane, unsigned long long lt, int* seq_total,
                                                 long long above = -1 /* MODE 2: only keys greater than this */,
                                                 unsigned long long gm = ~0ull) {
  uint32_t best = 0x7FFFFFFFu;
  *seq_total = 0;
  const int nMinCellX = max(0, (int)floorf((u - cam.min_x - radius) * invW));
  if (nMinCellX >= GRID_COLS) return MODE == 2 ? best : 0;
  const int nMaxCellX = min(GRID_COLS - 1, (int)ceilf((u - cam.min_x + radius) * invW));
  if (nMaxCellX < 0) return MODE == 2 ? best : 0;
  const int nMinCellY = max(0, (int)floorf((v - cam.min_y - radius) * invH));
  if (nMinCellY >= GRID_ROWS) return MODE == 2 ? best : 0;
  const int nMaxCellY = min(GRID_ROWS - 1, (int)ceilf((v - cam.min_y + radius) * invH));
  if (nMaxCellY < 0) return MODE == 2 ? best : 0;
  const bool bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
  unsigned long long d0 = 0, d1 = 0, d2 = 0, d3 = 0;
  if (MODE != 0) {
    const unsigned long long* dm = (const unsigned long long*)dmp;
    d0 = dm[0]; d1 = dm[1]; d2 = dm[2]; d3 = dm[3];
  }
  int seq0 = 0, w = 0;
  for (int ix = nMinCellX; ix <= nMaxCellX; ix++) {
    const int a = S.s_cstart[ix * GRID_ROWS + nMinCellY], b = S.s_cstart[ix * GRID_ROWS + nMaxCellY + 1];
    for (int e0 = a; e0 < b; e0 += GL) {
      const int e = e0 + lane;
      bool okc = false;
      int idx = 0;
      unsigned long long k0 = 0, k1 = 0, k2 = 0, k3 = 0;
      if (e < b) {
        idx = S.s_key[e] & 2047;
        // everything a candidate needs is fetched at once (position, octave, mvuRight, descriptor): one global round trip per
        // candidate chunk instead of four dependent ones (the gates below used to sit between the loads)
        const float kx = kps[idx].x, ky = kps[idx].y;
        const int koct = kps[idx].octave;
        const float ur2 = uright[idx];
        if (MODE != 0) {
          const unsigned long long* dk = (const unsigned long long*)(desc + (size_t)idx * 32);
          k0 = dk[0]; k1 = dk[1]; k2 = dk[2]; k3 = dk[3];
          asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3));   // keep the loads here (not sunk behind the gates)
        }
        okc = true;
        if (bCheckLevels) {
          if (koct < minLevel) okc = false;
          if (maxLevel >= 0 && koct > maxLevel) okc = false;
        }
        const float distx = kx - u, disty = ky - v;
        if (!(fabsf(distx) < radius && fabsf(disty) < radius)) okc = false;
        if (okc && ur2 > 0) {
          const float er = fabsf(ur - ur2);
          if (er > radius) okc = false;
        }
        if (MODE == 2 && okc) {
          const int mr = S.s_match[idx];
          const int m = mr < 0 ? -1 : (mr & 2047);   // (k_match_assign keeps a flag in bit 14)
          if (m >= 0 && ((S.s_obs[m >> 5] >> (m & 31)) & 1u)) okc = false;
          if (S.s_kclaim && ((S.s_kclaim[idx >> 5] >> (idx & 31)) & 1u) && m < 0) okc = false;
        }
      }
      if (MODE == 0) {
        w += __popcll(__ballot(okc) & gm);
      } else {
        uint32_t key = 0;
        if (okc) {
          const int dist = __popcll(k0 ^ d0) + __popcll(k1 ^ d1) + __popcll(k2 ^ d2) + __popcll(k3 ^ d3);
          // NB: candidates failing the window test do not advance the reference's vIndices order
          // relative to each other, so the sorted-array position is a valid order key
          key = ((uint32_t)dist << 22) | ((uint32_t)(seq0 + (e - a)) << 11) | (uint32_t)idx;
          if (MODE == 2 && (long long)key > above) best = min(best, key);
        }
        if (MODE == 1) {
          const unsigned long long bal = __ballot(okc);
          if (okc) list[w + __popcll(bal & lt)] = key;
          w += __popcll(bal & gm);
        }
      }
    }
    seq0 += b - a;
  }
  *seq_total = seq0;
  return MODE == 2 ? best : (uint32_t)w;
}

// ---- Frame bucket grid (shared by k_match, k_match_local and the debug read-out k_features_in_area) ----
// Frame::PosInGrid (src/Frame.cc:323-332, round()) as a sort key: cell << 11 | keypoint index, cell = posX * 48 + posY;
// 0xFFFFFFFF for a keypoint outside the grid (AssignFeaturesToGrid skips it).
__device__ __forceinline__ uint32_t grid_key(const sd_keypoint& kp, const TrackCam& cam, float invW, float invH, int i) {
  const int posX = (int)roundf((kp.x - cam.min_x) * invW);
  const int posY = (int)roundf((kp.y - cam.min_y) * invH);
  if (posX < 0 || posX >= GRID_COLS || posY < 0 || posY >= GRID_ROWS) return 0xFFFFFFFFu;
  return ((uint32_t)(posX * GRID_ROWS + posY) << 11) | (uint32_t)i;
}
// Frame::AssignFeaturesToGrid (src/Frame.cc:179-192): bitonic sort of the KP2 keys in LDS (mGrid[x][y] in ascending
// keypoint index = one contiguous, ordered run per cell) and the first sorted position of every cell.  All NT threads
// call, after a barrier that makes s_key complete; ends with s_cstart written but NOT yet synchronised.
__device__ __forceinline__ void grid_sort_and_starts(uint32_t* s_key, uint16_t* s_cstart, int KP2, int tid, int NT) {
  for (int k = 2; k <= KP2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < KP2; i += NT) {
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
  for (int c = tid; c <= GRID_COLS * GRID_ROWS; c += NT) {
    const uint32_t target = (uint32_t)c << 11;
    int lo = 0, hi = KP2;
    while (lo < hi) {
      int mid = (lo + hi) >> 1;
      if (s_key[mid] < target) lo = mid + 1;
      else hi = mid;
    }
    s_cstart[c] = (uint16_t)lo;
  }
}

// a16 / a17: projection of last-frame point i (src/ORBmatcher.cc:974-1004) and its window
template <int MODE, int GL = 64>
__device__ __forceinline__ uint32_t match_point(int i, const sd_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                const double* __restrict__ Xw, const uint8_t* __restrict__ mp_desc,
                                                const int32_t* __restrict__ l_oct, const float* __restrict__ uright,
                                                const MatchLds& S, const MatchGeom& G, const TrackCam& cam,
                                                const float* __restrict__ sf, uint32_t* list, int lane, unsigned long long lt,
                                                int* seq_total, unsigned long long gm = ~0ull) {
  const double xw = Xw[(size_t)i * 3], yw = Xw[(size_t)i * 3 + 1], zw = Xw[(size_t)i * 3 + 2];
  int nLastOctave = l_oct[i];
  asm volatile("" : "+v"(nLastOctave));   // fetched with the point, not after the frustum gates (one round trip less per point)
  const double X = (G.R[0][0] * xw + G.R[0][1] * yw + G.R[0][2] * zw) + G.t[0];
  const double Y = (G.R[1][0] * xw + G.R[1][1] * yw + G.R[1][2] * zw) + G.t[1];
  const double Z = (G.R[2][0] * xw + G.R[2][1] * yw + G.R[2][2] * zw) + G.t[2];
  const float xc = (float)X, yc = (float)Y;
  const float invzc = (float)(1.0 / Z);
  *seq_total = 0;
  if (invzc < 0) return MODE == 2 ? 0x7FFFFFFFu : 0;
  const float u = cam.ffx * xc * invzc + cam.fcx;
  const float v = cam.ffy * yc * invzc + cam.fcy;
  if (u < cam.min_x || u > cam.max_x) return MODE == 2 ? 0x7FFFFFFFu : 0;
  if (v < cam.min_y || v > cam.max_y) return MODE == 2 ? 0x7FFFFFFFu : 0;
  const float radius = G.th * sf[nLastOctave];
  const int minLevel = G.bForward ? nLastOctave : (G.bBackward ? 0 : nLastOctave - 1);
  const int maxLevel = G.bForward ? -1 : (G.bBackward ? nLastOctave : nLastOctave + 1);
  const float ur = u - cam.bf * invzc;
  return match_window<MODE, GL>(u, v, radius, minLevel, maxLevel, ur, mp_desc + (size_t)i * 32, kps, desc, uright, S, cam, G.invW, G.invH,
                                list, lane, lt, seq_total, -1, gm);
}

// Dynamic LDS layout (KP2 = power of two >= keypoint capacity, MP = max_points):
//   u32 s_key[KP2] | u32 s_list[MT_LIST_CAP] | u32 s_pt[MP] | f32 s_kang[KP2] | u32 s_obs[(MP+31)/32] | u32 s_valid[(MP+31)/32] |
//   i16 s_match[KP2] | u16 s_ev[max(KP2, MP)] | u16 s_cstart[64*48+2] | int s_hist[30] | int s_nlist
// (s_ev records one entry per ASSIGNMENT -- rotHist[bin].push_back -- and a keypoint may be assigned again by a later
// point, so up to n_last <= MP entries: it is sized by the larger of the two capacities)
__global__ __launch_bounds__(64 * MT_WAVES) void k_match(const sd_keypoint* __restrict__ kps_all, const uint8_t* __restrict__ desc_all,
                                                          const int32_t* __restrict__ nkp_all, TrackBuffers tb, TrackCam cam,
                                                          const float* __restrict__ sf, float th, int mono, int check_ori, int KP2,
                                                          int retry_below) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  // "If few matches, ignores alignment and uses a wider window search" (src/Tracking.cc:684-689): the second search of
  // TrackWithMotionModel runs only for the frames whose first search found < retry_below matches (whole-workgroup exit)
  if (retry_below > 0 && tb.n_matches[blockIdx.x] >= retry_below) return;
  const int MP = tb.max_points;
  uint32_t* s_key = (uint32_t*)smem;
  uint32_t* s_list = s_key + KP2;
  uint32_t* s_pt = s_list + MT_LIST_CAP;
  float* s_kang = (float*)(s_pt + MP);
  uint32_t* s_obs = (uint32_t*)(s_kang + KP2);
  uint32_t* s_valid = s_obs + ((MP + 31) >> 5);
  int16_t* s_match = (int16_t*)(s_valid + ((MP + 31) >> 5));
  uint16_t* s_ev = (uint16_t*)(s_match + KP2);
  uint16_t* s_cstart = s_ev + (KP2 > MP ? KP2 : MP);
  int* s_hist = (int*)(((uintptr_t)(s_cstart + GRID_COLS * GRID_ROWS + 2) + 3) & ~(uintptr_t)3);
  int* s_nlist = s_hist + HISTO_LENGTH;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = 64 * MT_WAVES;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int cap = tb.kp_cap;
  const int fc = tb.cur_bcast >= 0 ? tb.cur_bcast : f;   // current-frame slot (TrackBuffers::cur_bcast)
  const sd_keypoint* kps = kps_all + (size_t)fc * cap;
  const uint8_t* desc = desc_all + (size_t)fc * cap * 32;
  const int N = min(nkp_all[fc], min(cap, KP2));
  MatchGeom G;
  G.th = th;
  G.invW = (float)GRID_COLS / (float)(cam.max_x - cam.min_x);   // mfGridElementWidthInv
  G.invH = (float)GRID_ROWS / (float)(cam.max_y - cam.min_y);
  const int M = MP;
  const uint8_t* valid = tb.valid + (size_t)f * M;
  const double* Xw = tb.Xw + (size_t)f * M * 3;
  const uint8_t* mp_desc = tb.mp_desc + (size_t)f * M * 32;
  const int32_t* l_oct = tb.octave + (size_t)f * M;
  const float* l_ang = tb.angle + (size_t)f * M;
  const int32_t* l_obs = tb.obs + (size_t)f * M;
  const int n_last = min(tb.n_last[f], M);
  const float* uright = tb.uright + (size_t)fc * cap;

  // ---- AssignFeaturesToGrid: key = cell << 11 | index (PosInGrid uses round())
  for (int i = tid; i < KP2; i += NT) {
    uint32_t key = 0xFFFFFFFFu;
    float ang = 0.f;
    if (i < N) {
      const sd_keypoint kp = kps[i];
      ang = kp.angle;
      key = grid_key(kp, cam, G.invW, G.invH, i);
    }
    s_key[i] = key;
    s_kang[i] = ang;
    s_match[i] = -1;   // CurrentFrame.mvpMapPoints filled with NULL (src/Tracking.cc:676)
  }
  // Observations() > 0 and validity flags of the last frame's points as bit masks: one element per thread, wave ballots
  for (int m0 = 0; m0 < ((MP + 63) & ~63); m0 += NT) {
    const int m = m0 + tid;
    const unsigned long long bo = __ballot(m < n_last && l_obs[m] > 0), bv = __ballot(m < n_last && valid[m] != 0);
    const int w = (m0 >> 5) + 2 * wave;
    if (lane == 0 && w < ((MP + 31) >> 5)) {
      s_obs[w] = (uint32_t)bo;
      s_valid[w] = (uint32_t)bv;
      if (w + 1 < ((MP + 31) >> 5)) {
        s_obs[w + 1] = (uint32_t)(bo >> 32);
        s_valid[w + 1] = (uint32_t)(bv >> 32);
      }
    }
  }
  if (tid < HISTO_LENGTH) s_hist[tid] = 0;
  if (tid == 0) *s_nlist = 0;
  __syncthreads();
  grid_sort_and_starts(s_key, s_cstart, KP2, tid, NT);
  {
    // column-major; the retry searches from the predicted pose, which becomes the frame's pose (SetPose(predicted_pose))
    const double* Tc = (retry_below > 0 ? tb.Tprior : tb.Tcur) + (size_t)f * 16;
    if (retry_below > 0 && tid < 16) tb.Tcur[(size_t)f * 16 + tid] = Tc[tid];
    if (retry_below > 0 && tid == 0) tb.tw_info[(size_t)f * 4 + 3] = 1;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) G.R[r][c] = Tc[c * 4 + r];
      G.t[r] = Tc[12 + r];
    }
    // bForward / bBackward: tlc = Rlw * (-Rcw^T tcw) + tlw compared with the baseline mb
    G.bForward = G.bBackward = false;
    if (!mono) {
      const double* Tl = tb.Tref + (size_t)f * 16;
      double twc[3], tlc2;
      for (int i = 0; i < 3; i++) twc[i] = (-G.R[0][i]) * G.t[0] + (-G.R[1][i]) * G.t[1] + (-G.R[2][i]) * G.t[2];
      tlc2 = (Tl[0 * 4 + 2] * twc[0] + Tl[1 * 4 + 2] * twc[1] + Tl[2 * 4 + 2] * twc[2]) + Tl[12 + 2];
      G.bForward = tlc2 > cam.mb;
      G.bBackward = -tlc2 > cam.mb;
    }
  }
  __syncthreads();

  const MatchLds SL = {s_key, s_cstart, s_match, s_obs, nullptr};
  // ---- phase 1 (all waves, one wave per last-frame point): candidate keys into the LDS list.
  // s_pt[i] = 0 (nothing to do) | offset << 16 | count | 0xFFFFFFFF (list full: evaluate in phase 2)
  // Two points per wave, one per 32-lane half: a point's window rarely holds more than 32 candidates, and a wave is one
  // dependent chain of loads per point (map point -> keypoints of the window -> their descriptors), so two chains in flight
  // per wave nearly halve the phase.
  {
    const int half = lane >> 5, glane = lane & 31;
    const unsigned long long gm = 0xFFFFFFFFull << (32 * half);
    const unsigned long long glt = ((1ull << glane) - 1ull) << (32 * half);
    for (int i0 = 0; i0 < n_last; i0 += 2 * MT_WAVES) {
      const int i = i0 + 2 * wave + half;
      if (i < n_last) {
        uint32_t pc = 0;
        if ((s_valid[i >> 5] >> (i & 31)) & 1u) {
          int seq = 0;
          const int cnt = (int)match_point<0, 32>(i, kps, desc, Xw, mp_desc, l_oct, uright, SL, G, cam, sf, nullptr, glane, glt, &seq, gm);
          if (cnt > 0) {
            int off = 0;
            if (glane == 0) off = atomicAdd(s_nlist, cnt);
            off = __shfl(off, 32 * half);
            if (off + cnt > MT_LIST_CAP || seq >= 2048 || cnt > 0xffff) {
              pc = 0xFFFFFFFFu;
            } else {
              match_point<1, 32>(i, kps, desc, Xw, mp_desc, l_oct, uright, SL, G, cam, sf, s_list + off, glane, glt, &seq, gm);
              pc = ((uint32_t)off << 16) | (uint32_t)cnt;
            }
          }
        }
        if (glane == 0) s_pt[i] = pc;
      }
    }
  }
  __syncthreads();
  if (tid >= 64) return;   // one wavefront runs the order-dependent assignment loop

  // ---- phase 2 (points in order): minimum over the candidates not claimed by a point with observations
  // (first strict minimum = smallest key), assignment (later points overwrite), rotation histogram
  int nmatches = 0, nev = 0;
  const float factor = 1.0f / HISTO_LENGTH;
  // s_pt is read 64 entries at a time and only the points with something to do are visited (most slots of the arrays
  // hold no valid point: one LDS round trip per slot was a tenth of the kernel for a single frame)
  for (int base = 0; base < n_last; base += 64) {
   const uint32_t pcv = (base + lane < n_last) ? s_pt[base + lane] : 0u;
   unsigned long long todo = __ballot(pcv != 0);
   while (todo) {
    const int jsel = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int i = base + jsel;
    const uint32_t pc = (uint32_t)__builtin_amdgcn_readlane((int)pcv, jsel);
    uint32_t best = 0x7FFFFFFFu;
    if (pc != 0xFFFFFFFFu) {
      const int off = pc >> 16, cnt = pc & 0xffff;
      for (int j = lane; j < cnt; j += 64) {
        const uint32_t key = s_list[off + j];
        const int m = s_match[key & 2047];
        const bool claimed = (m >= 0) && ((s_obs[m >> 5] >> (m & 31)) & 1u);
        if (!claimed) best = min(best, key);
      }
    } else {
      int seq = 0;
      best = match_point<2>(i, kps, desc, Xw, mp_desc, l_oct, uright, SL, G, cam, sf, nullptr, lane, lt, &seq);
    }
    best = wave_min_u32(best);
    if (best == 0x7FFFFFFFu) continue;
    const int bestDist = best >> 22;
    const int bestIdx2 = best & 2047;
    if (bestDist <= TH_HIGH) {
      if (lane == 0) s_match[bestIdx2] = (int16_t)i;
      nmatches++;
      if (check_ori) {
        float rot = l_ang[i] - s_kang[bestIdx2];
        if (rot < 0.0) rot += 360.0f;
        int bin = (int)roundf(rot * factor);
        if (bin == HISTO_LENGTH) bin = 0;
        if (lane == 0) {
          s_ev[nev] = (uint16_t)((bin << 11) | bestIdx2);
          s_hist[bin]++;
        }
        nev++;
      }
    }
     }
  }
  // ---- rotation consistency: keep the three dominant 30-degree bins (10 % rule)
  if (check_ori) {
    int ind1 = -1, ind2 = -1, ind3 = -1;
    int max1 = 0, max2 = 0, max3 = 0;
    for (int b = 0; b < HISTO_LENGTH; b++) {
      const int sh = s_hist[b];
      if (sh > max1) {
        max3 = max2; max2 = max1; max1 = sh;
        ind3 = ind2; ind2 = ind1; ind1 = b;
      } else if (sh > max2) {
        max3 = max2; max2 = sh;
        ind3 = ind2; ind2 = b;
      } else if (sh > max3) {
        max3 = sh;
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
      const unsigned ev = s_ev[e];
      const int bin = ev >> 11;
      if (bin != ind1 && bin != ind2 && bin != ind3) {
        if (lane == 0) s_match[ev & 2047] = -1;
        nmatches--;
      }
    }
  }
  int32_t* out = tb.cur_match + (size_t)f * cap;
  for (int i = lane; i < cap; i += 64) out[i] = i < KP2 ? (int32_t)s_match[i] : -1;
  if (lane == 0) tb.n_matches[f] = nmatches;
}

// ------------------------------------------------------------------------------------------------
// Split SearchByProjection (the default): the same two phases as k_match, as two launches.
//   k_match_cand    8 waves per frame: grid build + sort, then every point's candidate keys -- counted, offsets by a prefix
//                   sum in POINT ORDER, written to a per-frame list in HBM (MT_HBM_LIST keys) with one packed
//                   offset << 16 | count word per point; the sorted grid goes to HBM too (10 KB).  14 KB of LDS, gone when
//                   the parallel work is done.
//   k_match_assign  1 wave per frame: the order-dependent assignment loop.  Because the list is in point order the wave
//                   streams it front to back in 64-key chunks held in registers (two chunks prefetched); all it keeps in LDS
//                   is CurrentFrame.mvpMapPoints (i16 per keypoint, bit 14 = "that point has Observations() > 0", so the
//                   claimed test is ONE LDS read), the obs bit mask and the rotation events: 6 KB per frame.
// Why: k_match holds 39 KB per frame through its one-wave phase 2 -- four frames per CU keep 154 of the 160 KB while 7 of 8
// wave slots are empty, and FAST workgroups (24-39 KB each) of the next batch cannot use the CU (VERDICT r2 weak #5/#8:
// FAST 2.19 ms alone, 4.11 ms in the pipeline).  The rotation histogram is filled AFTER the loop, in parallel, from the
// recorded (point, keypoint) events: rotHist only ever filters at the end (src/ORBmatcher.cc:1057-1072), and clearing a
// keypoint for each of its events in a losing bin is order-free.
// A point whose keys do not fit (list full, or a window with >= 2048 grid entries: keys carry 11 bits of order) is marked
// 0xFFFFFFFF like in k_match and evaluated inside the assignment loop by the same match_point<2> walk, on the grid copy in
// HBM (rare: th = 64 windows in the tests).
// ------------------------------------------------------------------------------------------------
#define MT_HBM_LIST 8192
#define MT_HBM_CSTART (GRID_COLS * GRID_ROWS + 4)
extern "C" __device__ __attribute__((const)) int __ockl_wfred_add_i32(int);

// pose / direction flags of a search (k_match's preamble): column-major Tcw, bForward / bBackward from tlc = Rlw * (-Rcw^T tcw) + tlw
__device__ __forceinline__ void match_geom(MatchGeom& G, const double* Tc, const double* Tl, const TrackCam& cam, float th, int mono) {
  G.th = th;
  G.invW = (float)GRID_COLS / (float)(cam.max_x - cam.min_x);
  G.invH = (float)GRID_ROWS / (float)(cam.max_y - cam.min_y);
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) G.R[r][c] = Tc[c * 4 + r];
    G.t[r] = Tc[12 + r];
  }
  G.bForward = G.bBackward = false;
  if (!mono) {
    double twc[3], tlc2;
    for (int i = 0; i < 3; i++) twc[i] = (-G.R[0][i]) * G.t[0] + (-G.R[1][i]) * G.t[1] + (-G.R[2][i]) * G.t[2];
    tlc2 = (Tl[0 * 4 + 2] * twc[0] + Tl[1 * 4 + 2] * twc[1] + Tl[2 * 4 + 2] * twc[2]) + Tl[12 + 2];
    G.bForward = tlc2 > cam.mb;
    G.bBackward = -tlc2 > cam.mb;
  }
}

__device__ __forceinline__ void match_cand_frame(const int f, uint8_t* smem, const sd_keypoint* __restrict__ kps_all,
                                                 const uint8_t* __restrict__ desc_all, const int32_t* __restrict__ nkp_all, const TrackBuffers& tb,
                                                 const TrackCam& cam, const float* __restrict__ sf, float th, int mono, int KP2, int retry_below) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = 64 * MT_WAVES;
  const int MP = tb.max_points;
  uint32_t* s_key = (uint32_t*)smem;
  uint32_t* s_off = s_key + KP2;                                  // exclusive prefix of the counts (may exceed the list)
  uint16_t* s_cnt = (uint16_t*)(s_off + MP);                      // 0xFFFF: window too large for the 11-bit order field
  uint32_t* s_valid = (uint32_t*)(s_cnt + MP + (MP & 1));
  uint16_t* s_cstart = (uint16_t*)(s_valid + ((MP + 31) >> 5));
  const int cap = tb.kp_cap;
  const int fc = tb.cur_bcast >= 0 ? tb.cur_bcast : f;
  const sd_keypoint* kps = kps_all + (size_t)fc * cap;
  const uint8_t* desc = desc_all + (size_t)fc * cap * 32;
  const int N = min(nkp_all[fc], min(cap, KP2));
  const int M = MP;
  const uint8_t* valid = tb.valid + (size_t)f * M;
  const double* Xw = tb.Xw + (size_t)f * M * 3;
  const uint8_t* mp_desc = tb.mp_desc + (size_t)f * M * 32;
  const int32_t* l_oct = tb.octave + (size_t)f * M;
  const int n_last = min(tb.n_last[f], M);
  const float* uright = tb.uright + (size_t)fc * cap;
  uint32_t* g_list = tb.mt_list + (size_t)f * MT_HBM_LIST;
  uint32_t* g_pt = tb.mt_pt + (size_t)f * M;
  const float invW = (float)GRID_COLS / (float)(cam.max_x - cam.min_x), invH = (float)GRID_ROWS / (float)(cam.max_y - cam.min_y);

  for (int i = tid; i < KP2; i += NT) s_key[i] = i < N ? grid_key(kps[i], cam, invW, invH, i) : 0xFFFFFFFFu;
  for (int m0 = 0; m0 < ((MP + 63) & ~63); m0 += NT) {
    const int m = m0 + tid;
    const unsigned long long bv = __ballot(m < n_last && valid[m] != 0);
    const int w = (m0 >> 5) + 2 * wave;
    if (lane == 0 && w < ((MP + 31) >> 5)) {
      s_valid[w] = (uint32_t)bv;
      if (w + 1 < ((MP + 31) >> 5)) s_valid[w + 1] = (uint32_t)(bv >> 32);
    }
  }
  __syncthreads();
  grid_sort_and_starts(s_key, s_cstart, KP2, tid, NT);
  MatchGeom G;
  {
    // column-major; the retry searches from the predicted pose, which becomes the frame's pose (SetPose(predicted_pose))
    const double* Tc = (retry_below > 0 ? tb.Tprior : tb.Tcur) + (size_t)f * 16;
    if (retry_below > 0 && tid < 16) tb.Tcur[(size_t)f * 16 + tid] = Tc[tid];
    if (retry_below > 0 && tid == 0) tb.tw_info[(size_t)f * 4 + 3] = 1;
    match_geom(G, Tc, tb.Tref + (size_t)f * 16, cam, th, mono);
  }
  __syncthreads();
  {   // the grid for the assignment kernel's slow path
    uint32_t* g_key = tb.mt_key + (size_t)f * MT_MAXKP;
    uint16_t* g_cs = tb.mt_cstart + (size_t)f * MT_HBM_CSTART;
    for (int i = tid; i < KP2; i += NT) g_key[i] = s_key[i];
    for (int c = tid; c <= GRID_COLS * GRID_ROWS; c += NT) g_cs[c] = s_cstart[c];
  }
  const MatchLds SL = {s_key, s_cstart, nullptr, nullptr, nullptr};
  // r3: FOUR points per wave, one per 16-lane quarter (k_match: two).  A point's window is walked one grid-cell column at a time and a
  // column's run holds 0...3 keypoints of the ~1000 spread over 64 x 48 cells: with 32 lanes per point nine tenths of them idled.
  constexpr int CGL = 8, CPW = 64 / CGL;
  const int half = lane / CGL, glane = lane % CGL;
  const unsigned long long gm = ((1ull << CGL) - 1ull) << (CGL * half);
  const unsigned long long glt = ((1ull << glane) - 1ull) << (CGL * half);
  // ---- pass A: candidates per point
  for (int i0 = 0; i0 < n_last; i0 += CPW * MT_WAVES) {
    const int i = i0 + CPW * wave + half;
    if (i < n_last) {
      int cnt = 0;
      if ((s_valid[i >> 5] >> (i & 31)) & 1u) {
        int seq = 0;
        cnt = (int)match_point<0, CGL>(i, kps, desc, Xw, mp_desc, l_oct, uright, SL, G, cam, sf, nullptr, glane, glt, &seq, gm);
        if (cnt > 0 && seq >= 2048) cnt = 0xFFFF;   // (cnt itself is at most the keypoint count, < 0xFFFF)
      }
      if (glane == 0) s_cnt[i] = (uint16_t)cnt;
    }
  }
  __syncthreads();
  // ---- offsets in point order: wave 0, a run of consecutive points per lane
  if (wave == 0) {
    const int per = (n_last + 63) >> 6;
    const int b = min(lane * per, n_last), e = min(b + per, n_last);
    int sum = 0;
    for (int i = b; i < e; i++) sum += s_cnt[i] == 0xFFFF ? 0 : s_cnt[i];
    int incl = sum;
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    int run = incl - sum;
    for (int i = b; i < e; i++) {
      s_off[i] = (uint32_t)run;
      run += s_cnt[i] == 0xFFFF ? 0 : s_cnt[i];
    }
  }
  __syncthreads();
  // ---- pass B: the keys, straight into the frame's HBM list
  for (int i0 = 0; i0 < n_last; i0 += CPW * MT_WAVES) {
    const int i = i0 + CPW * wave + half;
    if (i < n_last) {
      const int cnt = s_cnt[i];
      const uint32_t off = s_off[i];
      uint32_t pc = 0;
      if (cnt == 0xFFFF || (cnt > 0 && off + (uint32_t)cnt > MT_HBM_LIST)) {
        pc = 0xFFFFFFFFu;   // evaluated by the assignment loop itself
      } else if (cnt > 0) {
        int seq = 0;
        match_point<1, CGL>(i, kps, desc, Xw, mp_desc, l_oct, uright, SL, G, cam, sf, g_list + off, glane, glt, &seq, gm);
        pc = (off << 16) | (uint32_t)cnt;
      }
      if (glane == 0) g_pt[i] = pc;
    }
  }
}

// r3: the retry pass of TrackWithMotionModel (src/Tracking.cc:681-686: search again from the predicted pose with 2 th when fewer
// than 20 matches were found) walks a LIST of the frames that need it -- written by the first pass's assignment kernel -- with a
// small grid of ONE-WAVE workgroups.  It used to be a second full-grid launch of both kernels whose workgroups looked at their
// frame's count and left: eight-wave, 14-KB workgroups still have to be PLACED on a machine the extraction kernels fill -- 0.7-1.0 ms
// of the tracking chain per step in which nothing was computed (k_match_cand 2 x 1.10 ms in the pipeline against 0.33 + 0.002 alone;
// 128 such workgroups instead of 1024 took as long).
#define MT_RETRY_GRID 128
__global__ __launch_bounds__(64 * MT_WAVES) void k_match_cand(const sd_keypoint* __restrict__ kps_all, const uint8_t* __restrict__ desc_all,
                                                               const int32_t* __restrict__ nkp_all, TrackBuffers tb, TrackCam cam,
                                                               const float* __restrict__ sf, float th, int mono, int KP2, int note_below) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  if (note_below > 0 && blockIdx.x == 0 && threadIdx.x == 0) tb.retry_list[0] = 0;   // the assignment kernel behind this launch fills it
  match_cand_frame(blockIdx.x, smem, kps_all, desc_all, nkp_all, tb, cam, sf, th, mono, KP2, 0);
}
// RETRY: TrackWithMotionModel's second search of a frame (from the predicted pose, with the doubled window) done by THIS wave alone:
// every valid point goes through the per-point path that walks its window on the grid copy the first pass left in HBM (the keypoints,
// hence the grid, are the same); no candidate kernel runs for a retry.
template <bool RETRY>
__device__ __forceinline__ void match_assign_frame(const int f, uint8_t* smem, const sd_keypoint* __restrict__ kps_all,
                                                   const uint8_t* __restrict__ desc_all, const int32_t* __restrict__ nkp_all,
                                                   const TrackBuffers& tb, const TrackCam& cam, const float* __restrict__ sf, float th, int mono,
                                                   int check_ori, int KP2, int retry_below, int note_below) {
  const int lane = threadIdx.x;
  const int MP = tb.max_points, cap = tb.kp_cap;
  uint32_t* s_ev = (uint32_t*)smem;                      // one entry per ASSIGNMENT (rotHist[bin].push_back): <= n_last
  uint32_t* s_obs = s_ev + MP;
  int* s_hist = (int*)(s_obs + ((MP + 31) >> 5));
  int16_t* s_match = (int16_t*)(s_hist + HISTO_LENGTH + 2);   // -1 | point index | 0x4000 where that point has observations
  const int fc = tb.cur_bcast >= 0 ? tb.cur_bcast : f;
  const sd_keypoint* kps = kps_all + (size_t)fc * cap;
  const float* l_ang = tb.angle + (size_t)f * MP;
  const int32_t* l_obs = tb.obs + (size_t)f * MP;
  const int n_last = min(tb.n_last[f], MP);
  const uint32_t* g_list = tb.mt_list + (size_t)f * MT_HBM_LIST;
  const uint32_t* g_pt = tb.mt_pt + (size_t)f * MP;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  // the first three chunks of the list and the first group of per-point words are on their way while LDS is initialised
  uint32_t c0 = 0, c1 = 0, c2 = 0;
  if (!RETRY) { c0 = g_list[lane]; c1 = g_list[64 + lane]; c2 = g_list[128 + lane]; }
  const uint8_t* l_valid = tb.valid + (size_t)f * MP;
  auto point_word = [&](int m) -> uint32_t {   // first pass: what the candidate kernel left for point m; retry: "evaluate it here"
    if (m >= n_last) return 0u;
    return RETRY ? (l_valid[m] != 0 ? 0xFFFFFFFFu : 0u) : g_pt[m];
  };
  uint32_t pcv_next = point_word(lane);
  MatchGeom Gr;
  if (RETRY) {   // the retry searches from the predicted pose, which becomes the frame's pose (SetPose(predicted_pose), src/Tracking.cc:683)
    const double* Tp = tb.Tprior + (size_t)f * 16;
    match_geom(Gr, Tp, tb.Tref + (size_t)f * 16, cam, th, mono);
    if (lane < 16) tb.Tcur[(size_t)f * 16 + lane] = Tp[lane];
    if (lane == 0) tb.tw_info[(size_t)f * 4 + 3] = 1;
  }
  for (int i = lane; i < KP2; i += 64) s_match[i] = -1;   // CurrentFrame.mvpMapPoints filled with NULL (src/Tracking.cc:676)
  for (int m0 = 0; m0 < ((MP + 63) & ~63); m0 += 64) {
    const int m = m0 + lane;
    const unsigned long long bo = __ballot(m < n_last && l_obs[m] > 0);
    const int w = m0 >> 5;
    if (lane == 0 && w < ((MP + 31) >> 5)) {
      s_obs[w] = (uint32_t)bo;
      if (w + 1 < ((MP + 31) >> 5)) s_obs[w + 1] = (uint32_t)(bo >> 32);
    }
  }
  if (lane < HISTO_LENGTH) s_hist[lane] = 0;
  sdsel_fence();
  int nmatches = 0, nev = 0, cbase = 0;
  for (int base = 0; base < n_last; base += 64) {
    const uint32_t pcv = pcv_next;
    pcv_next = point_word(base + 64 + lane);
    unsigned long long todo = __ballot(pcv != 0);
    while (todo) {
      const int jsel = __ffsll((long long)todo) - 1;
      todo &= todo - 1;
      const int i = base + jsel;
      const uint32_t pc = (uint32_t)__builtin_amdgcn_readlane((int)pcv, jsel);
      const uint32_t obs_i = (s_obs[i >> 5] >> (i & 31)) & 1u;   // independent of the assignments: issued ahead of the chain
      uint32_t best = 0x7FFFFFFFu;
      if (!RETRY && pc != 0xFFFFFFFFu) {
        int off = pc >> 16, cnt = pc & 0xffff;
        while (cnt > 0) {
          while (off >= cbase + 64) {   // next chunk of the stream (offsets only grow: the list is in point order)
            c0 = c1;
            c1 = c2;
            cbase += 64;
            c2 = (cbase + 128 < MT_HBM_LIST) ? g_list[cbase + 128 + lane] : 0u;
          }
          const int lo = off - cbase, take = min(cnt, 64 - lo);
          if (lane >= lo && lane < lo + take) {
            const int m = s_match[c0 & 2047];
            if ((m & 0xC000) != 0x4000) best = min(best, c0);   // not claimed by a point with Observations() > 0
          }
          off += take;
          cnt -= take;
        }
      } else {   // keys not in the list: walk the window now, against the assignments so far (k_match's slow path)
        MatchGeom G;
        if (RETRY) G = Gr;
        else match_geom(G, tb.Tcur + (size_t)f * 16, tb.Tref + (size_t)f * 16, cam, th, mono);
        const MatchLds SG = {tb.mt_key + (size_t)f * MT_MAXKP, tb.mt_cstart + (size_t)f * MT_HBM_CSTART, s_match, s_obs, nullptr};
        int seq = 0;
        best = match_point<2>(i, kps, desc_all + (size_t)fc * cap * 32, tb.Xw + (size_t)f * MP * 3, tb.mp_desc + (size_t)f * MP * 32,
                              tb.octave + (size_t)f * MP, tb.uright + (size_t)fc * cap, SG, G, cam, sf, nullptr, lane, lt, &seq);
      }
      best = wave_min_u32(best);
      if (best == 0x7FFFFFFFu) continue;
      const int bestDist = best >> 22, bestIdx2 = best & 2047;
      if (bestDist <= TH_HIGH) {
        if (lane == 0) {
          s_match[bestIdx2] = (int16_t)(i | (int)(obs_i << 14));
          s_ev[nev] = ((uint32_t)i << 11) | (uint32_t)bestIdx2;
        }
        nmatches++;
        nev++;
      }
    }
  }
  sdsel_fence();
  // ---- rotation consistency (src/ORBmatcher.cc:1041-1072): bins of all recorded assignments, three dominant bins stay
  if (check_ori) {
    const float factor = 1.0f / HISTO_LENGTH;
    for (int e = lane; e < nev; e += 64) {
      const uint32_t ev = s_ev[e];
      float rot = l_ang[ev >> 11] - kps[ev & 2047].angle;
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == HISTO_LENGTH) bin = 0;
      atomicAdd(&s_hist[bin], 1);
      s_ev[e] = ev | ((uint32_t)bin << 22);
    }
    sdsel_fence();
    int ind1 = -1, ind2 = -1, ind3 = -1;
    int max1 = 0, max2 = 0, max3 = 0;
    for (int b = 0; b < HISTO_LENGTH; b++) {
      const int sh = s_hist[b];
      if (sh > max1) {
        max3 = max2; max2 = max1; max1 = sh;
        ind3 = ind2; ind2 = ind1; ind1 = b;
      } else if (sh > max2) {
        max3 = max2; max2 = sh;
        ind3 = ind2; ind2 = b;
      } else if (sh > max3) {
        max3 = sh;
        ind3 = b;
      }
    }
    if (max2 < 0.1f * (float)max1) {
      ind2 = -1;
      ind3 = -1;
    } else if (max3 < 0.1f * (float)max1) {
      ind3 = -1;
    }
    int bad = 0;
    for (int e = lane; e < nev; e += 64) {
      const uint32_t ev = s_ev[e];
      const int bin = (int)(ev >> 22);
      if (bin != ind1 && bin != ind2 && bin != ind3) {
        s_match[ev & 2047] = -1;   // every event in a losing bin clears its keypoint and counts (also a since-overwritten one)
        bad++;
      }
    }
    nmatches -= __ockl_wfred_add_i32(bad);
    sdsel_fence();
  }
  int32_t* out = tb.cur_match + (size_t)f * cap;
  for (int i = lane; i < cap; i += 64) {
    const int m = i < KP2 ? (int)s_match[i] : -1;
    out[i] = m < 0 ? -1 : (m & 2047);
  }
  if (lane == 0) {
    tb.n_matches[f] = nmatches;
    if (retry_below == 0 && note_below > 0 && nmatches < note_below)   // this frame goes through the retry pass (any order: frames are independent)
      tb.retry_list[1 + atomicAdd(&tb.retry_list[0], 1)] = f;
  }
}

__global__ __launch_bounds__(64) void k_match_assign(const sd_keypoint* __restrict__ kps_all, const uint8_t* __restrict__ desc_all,
                                                     const int32_t* __restrict__ nkp_all, TrackBuffers tb, TrackCam cam,
                                                     const float* __restrict__ sf, float th, int mono, int check_ori, int KP2, int note_below) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  match_assign_frame<false>(blockIdx.x, smem, kps_all, desc_all, nkp_all, tb, cam, sf, th, mono, check_ori, KP2, 0, note_below);
}
__global__ __launch_bounds__(64) void k_match_assign_retry(const sd_keypoint* __restrict__ kps_all, const uint8_t* __restrict__ desc_all,
                                                           const int32_t* __restrict__ nkp_all, TrackBuffers tb, TrackCam cam,
                                                           const float* __restrict__ sf, float th, int mono, int check_ori, int KP2,
                                                           int retry_below) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int cnt = tb.retry_list[0];
  for (int it = blockIdx.x; it < cnt; it += gridDim.x) {
    match_assign_frame<true>(tb.retry_list[1 + it], smem, kps_all, desc_all, nkp_all, tb, cam, sf, th, mono, check_ori, KP2, retry_below, 0);
    __syncthreads();   // (one wave: orders the LDS reuse)
  }
}

// ------------------------------------------------------------------------------------------------
// k_match_local: TrackLocalMap's search for one frame per workgroup.
//   Frame::isInFrustum            src/Frame.cc:215-269   (one thread per local map point)
//   MapPoint::PredictScale        src/MapPoint.cc:371-385: nScale = clamp(ceil(log(ratio) / logScaleFactor)) is a
//                                 monotone step function of the float `ratio`; its breakpoints scale_thr[n] (smallest
//                                 ratio that reaches level n) are found on the host with the host libm, so the device
//                                 needs no log() and agrees with the reference's libm by construction
//   ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)   src/ORBmatcher.cc:43-126
// Same two phases as k_match; the per-point result needs the two smallest keys (best / second best for the
// mfNNratio test), which are the reference's bestDist / bestDist2 because both are updated with strict `<` in
// vIndices order.
// Dynamic LDS: the k_match layout + u32 s_kclaim[KP2/32] + u8 s_koct[KP2].
__global__ __launch_bounds__(64 * MT_WAVES) void k_match_local(const sd_keypoint* __restrict__ kps_all, const uint8_t* __restrict__ desc_all,
                                                                const int32_t* __restrict__ nkp_all, TrackBuffers tb, TrackCam cam,
                                                                const float* __restrict__ sf, const float* __restrict__ scale_thr,
                                                                int nlevels, float th, float nnratio, float cos_limit, int KP2,
                                                                int claim_from_matches, int frustum_given) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  const int MP = tb.max_points;
  uint32_t* s_key = (uint32_t*)smem;
  uint32_t* s_list = s_key + KP2;
  uint32_t* s_pt = s_list + MT_LIST_CAP;
  uint32_t* s_obs = s_pt + MP;
  uint32_t* s_kclaim = s_obs + ((MP + 31) >> 5);
  int16_t* s_match = (int16_t*)(s_kclaim + (KP2 >> 5));
  uint16_t* s_cstart = (uint16_t*)(s_match + KP2);
  uint8_t* s_koct = (uint8_t*)(s_cstart + GRID_COLS * GRID_ROWS + 2);
  int* s_nlist = (int*)(((uintptr_t)(s_koct + KP2) + 3) & ~(uintptr_t)3);
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = 64 * MT_WAVES;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int cap = tb.kp_cap, M = MP;
  const int fc = tb.cur_bcast >= 0 ? tb.cur_bcast : f;   // current-frame slot (TrackBuffers::cur_bcast)
  const sd_keypoint* kps = kps_all + (size_t)fc * cap;
  const uint8_t* desc = desc_all + (size_t)fc * cap * 32;
  const int N = min(nkp_all[fc], min(cap, KP2));
  const float invW = (float)GRID_COLS / (float)(cam.max_x - cam.min_x);
  const float invH = (float)GRID_ROWS / (float)(cam.max_y - cam.min_y);
  const int n_loc = min(tb.lm_n[f], M);
  const uint8_t* cand = tb.lm_cand + (size_t)f * M;
  const double* Xw = tb.lm_Xw + (size_t)f * M * 3;
  const double* nrm = tb.lm_normal + (size_t)f * M * 3;
  const float* dmin = tb.lm_min + (size_t)f * M;
  const float* dmax = tb.lm_max + (size_t)f * M;
  const float* mfmax = tb.lm_mfmax + (size_t)f * M;
  const uint8_t* mp_desc = tb.lm_desc + (size_t)f * M * 32;
  const int32_t* l_obs = tb.lm_obs + (size_t)f * M;
  const uint8_t* kclaim = tb.lm_kclaim + (size_t)f * cap;
  const float* uright = tb.uright + (size_t)fc * cap;
  uint8_t* o_inview = tb.lm_inview + (size_t)f * M;
  float* o_proj = tb.lm_proj + (size_t)f * M * 3;
  int32_t* o_level = tb.lm_level + (size_t)f * M;
  float* o_cos = tb.lm_cos + (size_t)f * M;

  // ---- grid (as in k_match) + claim flags + octaves
  for (int i = tid; i < KP2; i += NT) {
    uint32_t key = 0xFFFFFFFFu;
    int oct = 0;
    if (i < N) {
      const sd_keypoint kp = kps[i];
      oct = kp.octave;
      key = grid_key(kp, cam, invW, invH, i);
    }
    s_key[i] = key;
    s_koct[i] = (uint8_t)oct;
    s_match[i] = -1;
  }
  for (int i0 = 0; i0 < KP2; i0 += NT) {   // claim flags / Observations() > 0 flags as bit masks (wave ballots)
    const int i = i0 + tid;
    bool claimed = false;
    if (i < N) {
      if (claim_from_matches) {   // F.mvpMapPoints[idx] && F.mvpMapPoints[idx]->Observations() > 0 (src/ORBmatcher.cc:81-83)
        const int m = tb.cur_match[(size_t)f * cap + i];
        claimed = m >= 0 && tb.obs[(size_t)f * MP + m] > 0;
      } else {
        claimed = kclaim[i] != 0;
      }
    }
    const unsigned long long bc = __ballot(claimed);
    const int w = (i0 >> 5) + 2 * wave;
    if (lane == 0 && w < (KP2 >> 5)) {
      s_kclaim[w] = (uint32_t)bc;
      if (w + 1 < (KP2 >> 5)) s_kclaim[w + 1] = (uint32_t)(bc >> 32);
    }
  }
  for (int m0 = 0; m0 < ((MP + 63) & ~63); m0 += NT) {
    const int m = m0 + tid;
    const unsigned long long bo = __ballot(m < n_loc && l_obs[m] > 0);
    const int w = (m0 >> 5) + 2 * wave;
    if (lane == 0 && w < ((MP + 31) >> 5)) {
      s_obs[w] = (uint32_t)bo;
      if (w + 1 < ((MP + 31) >> 5)) s_obs[w + 1] = (uint32_t)(bo >> 32);
    }
  }
  if (tid == 0) *s_nlist = 0;
  __syncthreads();
  grid_sort_and_starts(s_key, s_cstart, KP2, tid, NT);
  // ---- isInFrustum, one thread per point (frustum_given: the caller's own Frame::isInFrustum results -- mbTrackInView,
  // mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos, uploaded by sd_track_set_local_view -- are used as they are, which is
  // what ORBmatcher::SearchByProjection(F, vpMapPoints, th) itself reads, src/ORBmatcher.cc:48-60)
  if (!frustum_given) {
    const double* Tc = tb.Tcur + (size_t)f * 16;   // column-major
    double R[3][3], t[3], Ow[3];
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) R[r][c] = Tc[c * 4 + r];
      t[r] = Tc[12 + r];
    }
    for (int i = 0; i < 3; i++) Ow[i] = (-R[0][i]) * t[0] + (-R[1][i]) * t[1] + (-R[2][i]) * t[2];   // mOw = -Rcw^T tcw
    for (int i = tid; i < M; i += NT) {
      uint8_t inview = 0;
      float pu = 0, pv = 0, pxr = 0, vc = 0;
      int lvl = 0;
      if (i < n_loc && cand[i]) {
        const double P0 = Xw[(size_t)i * 3], P1 = Xw[(size_t)i * 3 + 1], P2 = Xw[(size_t)i * 3 + 2];
        const double PcX = (R[0][0] * P0 + R[0][1] * P1 + R[0][2] * P2) + t[0];
        const double PcY = (R[1][0] * P0 + R[1][1] * P1 + R[1][2] * P2) + t[1];
        const double PcZ = (R[2][0] * P0 + R[2][1] * P1 + R[2][2] * P2) + t[2];
        if (!(PcZ < 0.0)) {
          const float invz = (float)(1.0 / PcZ);                                  // 1.0f / PcZ
          const float u = (float)((double)cam.ffx * PcX * (double)invz + (double)cam.fcx);
          const float v = (float)((double)cam.ffy * PcY * (double)invz + (double)cam.fcy);
          if (!(u < cam.min_x || u > cam.max_x) && !(v < cam.min_y || v > cam.max_y)) {
            const double PO0 = P0 - Ow[0], PO1 = P1 - Ow[1], PO2 = P2 - Ow[2];
            const float dist = (float)sqrt((PO0 * PO0 + PO1 * PO1) + PO2 * PO2);
            if (!(dist < dmin[i] || dist > dmax[i])) {
              const float viewCos = (float)(((PO0 * nrm[(size_t)i * 3] + PO1 * nrm[(size_t)i * 3 + 1]) + PO2 * nrm[(size_t)i * 3 + 2]) / (double)dist);
              if (!(viewCos < cos_limit)) {
                const float ratio = mfmax[i] / dist;
                int nScale = 0;
                for (int n = 1; n < nlevels; n++) nScale += (ratio >= scale_thr[n]) ? 1 : 0;   // PredictScale
                inview = 1;
                pu = u;
                pv = v;
                pxr = u - cam.bf * invz;
                lvl = nScale;
                vc = viewCos;
              }
            }
          }
        }
      }
      o_inview[i] = inview;
      o_proj[(size_t)i * 3] = pu;
      o_proj[(size_t)i * 3 + 1] = pv;
      o_proj[(size_t)i * 3 + 2] = pxr;
      o_level[i] = lvl;
      o_cos[i] = vc;
    }
  }
  __syncthreads();   // also makes the o_* stores of this workgroup visible to its own later loads
  const MatchLds SL = {s_key, s_cstart, s_match, s_obs, s_kclaim};
  const bool bFactor = th != 1.0f;
  // ---- phase 1: candidate lists, eight points per wave (one per 8-lane group, as in k_match_cand: a window column's run holds
  // 0...3 keypoints)
  {
    constexpr int CGL = 8, CPW = 64 / CGL;
    const int half = lane / CGL, glane = lane % CGL;
    const unsigned long long gm = ((1ull << CGL) - 1ull) << (CGL * half);
    const unsigned long long glt = ((1ull << glane) - 1ull) << (CGL * half);
    for (int i0 = 0; i0 < n_loc; i0 += CPW * MT_WAVES) {
      const int i = i0 + CPW * wave + half;
      if (i < n_loc) {
        uint32_t pc = 0;
        if (o_inview[i]) {
          float r = o_cos[i] > 0.998 ? 2.5f : 4.0f;   // RadiusByViewingCos
          if (bFactor) r *= th;
          const int lvl = o_level[i];
          const float radius = r * sf[lvl];
          int seq = 0;
          const int cnt = (int)match_window<0, CGL>(o_proj[(size_t)i * 3], o_proj[(size_t)i * 3 + 1], radius, lvl - 1, lvl, o_proj[(size_t)i * 3 + 2],
                                                   mp_desc + (size_t)i * 32, kps, desc, uright, SL, cam, invW, invH, nullptr, glane, glt, &seq, -1, gm);
          if (cnt > 0) {
            int off = 0;
            if (glane == 0) off = atomicAdd(s_nlist, cnt);
            off = __shfl(off, CGL * half);
            if (off + cnt > MT_LIST_CAP || seq >= 2048 || cnt > 0xffff) {
              pc = 0xFFFFFFFFu;
            } else {
              match_window<1, CGL>(o_proj[(size_t)i * 3], o_proj[(size_t)i * 3 + 1], radius, lvl - 1, lvl, o_proj[(size_t)i * 3 + 2],
                                  mp_desc + (size_t)i * 32, kps, desc, uright, SL, cam, invW, invH, s_list + off, glane, glt, &seq, -1, gm);
              pc = ((uint32_t)off << 16) | (uint32_t)cnt;
            }
          }
        }
        if (glane == 0) s_pt[i] = pc;
      }
    }
  }
  __syncthreads();
  if (tid >= 64) return;
  // ---- phase 2: best / second best over the unclaimed candidates, ratio test, assignment
  int nmatches = 0;
  // s_pt is read 64 entries at a time and only the points with something to do are visited (most slots of the arrays
  // hold no valid point: one LDS round trip per slot was a tenth of the kernel for a single frame)
  for (int base = 0; base < n_loc; base += 64) {
   const uint32_t pcv = (base + lane < n_loc) ? s_pt[base + lane] : 0u;
   unsigned long long todo = __ballot(pcv != 0);
   while (todo) {
    const int jsel = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int i = base + jsel;
    const uint32_t pc = (uint32_t)__builtin_amdgcn_readlane((int)pcv, jsel);
    uint32_t gb, gs;   // the two smallest unclaimed keys of the point (wave-uniform)
    if (pc != 0xFFFFFFFFu) {
      uint32_t best = 0x7FFFFFFFu, second = 0x7FFFFFFFu;
      const int off = pc >> 16, cnt = pc & 0xffff;
      for (int j = lane; j < cnt; j += 64) {
        const uint32_t key = s_list[off + j];
        const int idx = key & 2047;
        const int m = s_match[idx];
        const bool claimed = m >= 0 ? (((s_obs[m >> 5] >> (m & 31)) & 1u) != 0) : (((s_kclaim[idx >> 5] >> (idx & 31)) & 1u) != 0);
        if (!claimed) {
          if (key < best) { second = best; best = key; }
          else if (key < second) second = key;
        }
      }
      gb = best;
      gb = wave_min_u32(gb);
      gs = (best == gb) ? second : best;   // the lane that holds the best offers its runner-up, the others their best
      gs = wave_min_u32(gs);
    } else {   // candidate list did not fit in LDS: enumerate the window again (best, then the best above it)
      float r = o_cos[i] > 0.998 ? 2.5f : 4.0f;
      if (bFactor) r *= th;
      const int lvl = o_level[i];
      int seq = 0;
      gb = match_window<2>(o_proj[(size_t)i * 3], o_proj[(size_t)i * 3 + 1], r * sf[lvl], lvl - 1, lvl, o_proj[(size_t)i * 3 + 2],
                           mp_desc + (size_t)i * 32, kps, desc, uright, SL, cam, invW, invH, nullptr, lane, lt, &seq);
      gb = wave_min_u32(gb);
      gs = 0x7FFFFFFFu;
      if (gb != 0x7FFFFFFFu) {
        gs = match_window<2>(o_proj[(size_t)i * 3], o_proj[(size_t)i * 3 + 1], r * sf[lvl], lvl - 1, lvl, o_proj[(size_t)i * 3 + 2],
                             mp_desc + (size_t)i * 32, kps, desc, uright, SL, cam, invW, invH, nullptr, lane, lt, &seq, (long long)gb);
        gs = wave_min_u32(gs);
      }
    }
    if (gb == 0x7FFFFFFFu) continue;
    const int bestDist = gb >> 22, bestIdx = gb & 2047;
    const int bestDist2 = gs == 0x7FFFFFFFu ? 256 : (int)(gs >> 22);
    const int bestLevel = s_koct[bestIdx];
    const int bestLevel2 = gs == 0x7FFFFFFFu ? -1 : (int)s_koct[gs & 2047];
    if (bestDist <= TH_HIGH) {
      if (bestLevel == bestLevel2 && (float)bestDist > nnratio * (float)bestDist2) continue;
      if (lane == 0) s_match[bestIdx] = (int16_t)i;
      nmatches++;
    }
     }
  }
  int32_t* out = tb.lm_match + (size_t)f * cap;
  for (int i = lane; i < cap; i += 64) out[i] = i < KP2 ? (int32_t)s_match[i] : -1;
  if (lane == 0) tb.lm_nmatch[f] = nmatches;
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

// ------------------------------------------------------------------------------------------------
// Debug read-out of the device grid: Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (src/Frame.cc:271-321) for
// one current frame, through the SAME grid build and window walk the matchers use (grid_key / grid_sort_and_starts /
// match_window).  out[0 .. n) = keypoint indices in the reference's vIndices order; grid_cells (may be null) receives
// mGrid's occupancy, [64][48] counts.  Lets the parity tests compare a13 / a14 directly instead of through match vectors.
__global__ __launch_bounds__(64 * MT_WAVES) void k_features_in_area(const sd_keypoint* __restrict__ kps, const uint8_t* __restrict__ desc,
                                                                     const int32_t* __restrict__ nkp, const float* __restrict__ uright,
                                                                     TrackCam cam, int cap, int KP2, float x, float y, float r, int minLevel,
                                                                     int maxLevel, int32_t* __restrict__ out, int out_cap,
                                                                     int32_t* __restrict__ n_out, int32_t* __restrict__ grid_cells) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* s_key = (uint32_t*)smem;
  uint32_t* s_list = s_key + KP2;
  uint16_t* s_cstart = (uint16_t*)(s_list + KP2);
  const int tid = threadIdx.x, lane = tid & 63, NT = 64 * MT_WAVES;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int N = min(nkp[0], min(cap, KP2));
  const float invW = (float)GRID_COLS / (float)(cam.max_x - cam.min_x);
  const float invH = (float)GRID_ROWS / (float)(cam.max_y - cam.min_y);
  __shared__ __attribute__((aligned(16))) uint8_t zero_desc[32];
  if (tid < 32) zero_desc[tid] = 0;
  for (int i = tid; i < KP2; i += NT) s_key[i] = i < N ? grid_key(kps[i], cam, invW, invH, i) : 0xFFFFFFFFu;
  __syncthreads();
  grid_sort_and_starts(s_key, s_cstart, KP2, tid, NT);
  __syncthreads();
  if (grid_cells)
    for (int c = tid; c < GRID_COLS * GRID_ROWS; c += NT) grid_cells[c] = (int)s_cstart[c + 1] - (int)s_cstart[c];
  if (tid >= 64) return;
  const MatchLds SL = {s_key, s_cstart, nullptr, nullptr, nullptr};
  int seq = 0;
  // the matchers' stereo gate (|ur - mvuRight[idx]| > radius rejects) is not part of GetFeaturesInArea: ur = NaN makes the
  // comparison false for every keypoint, whatever mvuRight holds
  const int cnt = (int)match_window<1>(x, y, r, minLevel, maxLevel, __builtin_nanf(""), zero_desc, kps, desc, uright, SL, cam, invW, invH, s_list, lane, lt, &seq);
  // keys are (dist << 22 | order << 11 | idx); emitted in walk order already (list position = order among the survivors)
  for (int j = lane; j < min(cnt, out_cap); j += 64) out[j] = (int32_t)(s_list[j] & 2047);
  if (lane == 0) *n_out = cnt;
}

int launch_features_in_area(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, int frame, float x, float y, float r,
                            int min_level, int max_level, int32_t* d_out, int out_cap, int32_t* d_n, int32_t* d_grid, hipStream_t s) {
  int KP2 = 64;
  while (KP2 < tb.kp_cap) KP2 <<= 1;
  SD_REQUIRE(KP2 <= MT_MAXKP, SD_ERR_CAPACITY, "matcher supports at most 2048 keypoints per frame");
  const size_t lds = (size_t)KP2 * 8 + (GRID_COLS * GRID_ROWS + 2) * 2;
  const sd_keypoint* kps = (cur->have_dist ? cur->d_kps_un : cur->d_kps) + (size_t)frame * tb.kp_cap;
  hipLaunchKernelGGL(k_features_in_area, dim3(1), dim3(64 * MT_WAVES), lds, s, kps, cur->d_desc + (size_t)frame * tb.kp_cap * 32,
                     cur->d_nout + frame, tb.uright + (size_t)frame * tb.kp_cap, cam, tb.kp_cap, KP2, x, y, r, min_level, max_level, d_out,
                     out_cap, d_n, d_grid);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

int launch_match_local(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_sf, const float* d_scale_thr,
                       int nlevels, int n_frames, float th, float nnratio, float cos_limit, hipStream_t s, int claim_from_matches,
                       int frustum_given) {
  int KP2 = 64;
  while (KP2 < tb.kp_cap) KP2 <<= 1;
  SD_REQUIRE(KP2 <= MT_MAXKP && tb.max_points <= 2048, SD_ERR_CAPACITY, "matcher supports at most 2048 keypoints / map points per frame");
  const int MP = tb.max_points;
  const size_t lds = (size_t)KP2 * 4 + MT_LIST_CAP * 4 + (size_t)MP * 4 + (size_t)((MP + 31) >> 5) * 4 + (size_t)(KP2 >> 5) * 4 + (size_t)KP2 * 2 +
                     (GRID_COLS * GRID_ROWS + 2) * 2 + (size_t)KP2 + 4 + 8;
  hipLaunchKernelGGL(k_match_local, dim3(n_frames), dim3(64 * MT_WAVES), lds, s, (cur->have_dist ? cur->d_kps_un : cur->d_kps), cur->d_desc,
                     cur->d_nout, tb, cam, d_sf, d_scale_thr, nlevels, th, nnratio, cos_limit, KP2, claim_from_matches, frustum_given);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

int launch_match(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_sf, int n_frames, float th,
                 int mono, int check_ori, hipStream_t s, int retry_below, int note_below) {
  int KP2 = 64;
  while (KP2 < tb.kp_cap) KP2 <<= 1;
  SD_REQUIRE(KP2 <= MT_MAXKP && tb.max_points <= 2048, SD_ERR_CAPACITY, "matcher supports at most 2048 keypoints / map points per frame");
  const int MP = tb.max_points;
  const sd_keypoint* kps = cur->have_dist ? cur->d_kps_un : cur->d_kps;
  const size_t lds = (size_t)KP2 * 4 + MT_LIST_CAP * 4 + (size_t)MP * 4 + (size_t)KP2 * 4 + (size_t)((MP + 31) >> 5) * 8 + (size_t)KP2 * 2 +
                     (size_t)std::max(KP2, MP) * 2 + (GRID_COLS * GRID_ROWS + 2) * 2 + 4 + (HISTO_LENGTH + 1) * 4;
  if (opt(OPT_MATCH_SPLIT)) {   // candidates -> HBM list -> one-wave assignment
    const size_t lds_c = (size_t)KP2 * 4 + (size_t)MP * 4 + (size_t)(MP + (MP & 1)) * 2 + (size_t)((MP + 31) >> 5) * 4 +
                         (GRID_COLS * GRID_ROWS + 2) * 2 + 8;
    const size_t lds_a = (size_t)MP * 4 + (size_t)((MP + 31) >> 5) * 4 + (HISTO_LENGTH + 2) * 4 + (size_t)KP2 * 2;
    // first pass: one workgroup per frame (note_below > 0: frames with fewer matches are listed for a retry pass);
    // retry pass (retry_below > 0): a small grid walks that list
    if (retry_below > 0) {   // one wave per listed frame does the whole second search (match_assign_frame<true>)
      hipLaunchKernelGGL(k_match_assign_retry, dim3(std::min(n_frames, MT_RETRY_GRID)), dim3(64), lds_a, s, kps, cur->d_desc, cur->d_nout, tb, cam,
                         d_sf, th, mono, check_ori, KP2, retry_below);
    } else {
      hipLaunchKernelGGL(k_match_cand, dim3(n_frames), dim3(64 * MT_WAVES), lds_c, s, kps, cur->d_desc, cur->d_nout, tb, cam, d_sf, th, mono, KP2,
                         note_below);
      hipLaunchKernelGGL(k_match_assign, dim3(n_frames), dim3(64), lds_a, s, kps, cur->d_desc, cur->d_nout, tb, cam, d_sf, th, mono, check_ori,
                         KP2, note_below);
    }
  } else {
    hipLaunchKernelGGL(k_match, dim3(n_frames), dim3(64 * MT_WAVES), lds, s, kps, cur->d_desc, cur->d_nout, tb, cam, d_sf, th, mono, check_ori,
                       KP2, retry_below);
  }
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

}  // namespace sd
