// ORBmatcher::SearchByPoints(KeyFrame* currentKF, KeyFrame* pKF, matches) on MI355X: brute-force Hamming matching of the
// map points of two keyframes (reference src/ORBmatcher.cc:1209-1301; called per loop candidate by
// LoopClosing::ComputeSim3, src/LoopClosing.cc:255).  BASELINE configs[1] ("ORB extract + Hamming match") / SURVEY §8(d)
// C2(ii): 1000 x 1000 descriptors per frame pair.
//
// One workgroup per keyframe pair; slot f pairs frame f (or the broadcast frame) of the `cur` extractor -- currentKF --
// with frame f of the `ref` extractor -- pKF.  The reference loop is sequential in idx1 only through vbMatched2 (a pKF point
// is given away once), so the work splits in two:
//   phase 1 (all lanes, N1 x N2 popcounts): every currentKF point with a map point gets the BF_K smallest keys
//            dist << 11 | idx2 over the pKF points with a map point.  One lane owns BF_PT points (descriptors in
//            registers); the pKF descriptors are staged through LDS in tiles and read as broadcasts, and the loop runs
//            over the SET BITS of the tile's validity mask on the scalar unit, so points without a map point cost nothing.
//   phase 2 (one wave, idx1 ascending): the two smallest keys not yet given away = (bestDist1, bestIdx2) and bestDist2
//            -- both updated with strict `<` in idx2 order in the reference, which is exactly key order -- then the
//            TH_LOW / mfNNratio tests and vbMatched2.  When fewer than two of a point's BF_K keys survive and a further
//            candidate could still matter, the row is recomputed by the whole wave against the live vbMatched2.
//   rotation histogram (ComputeThreeMaxima, src/ORBmatcher.cc:1423-1454) afterwards: the bins never feed back into the loop.
// Integer work only; results are bit-identical to the sequential loop by construction.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "orb_internal.h"
#include "track_internal.h"

namespace sd {

#define BF_THREADS 256
#define BF_K 4          // smallest keys kept per currentKF point
#define BF_PT 2         // currentKF points per lane in phase 1
#define BF_TILE 512     // pKF descriptors per LDS tile (16 KB)
#define BF_TH_LOW 50
#define BF_HISTO 30
#define BF_INF 0xFFFFFFFFu

extern "C" __device__ __attribute__((const)) unsigned int __ockl_wfred_min_u32(unsigned int);

__device__ __forceinline__ int bf_dist(const uint32_t (&a)[8], const uint32_t (&b)[8]) {
  int d = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) d += __popc(a[k] ^ b[k]);   // v_bcnt_u32_b32 accumulates
  return d;
}

__device__ __forceinline__ void bf_insert(uint32_t (&k)[BF_K], uint32_t key) {
  if (key < k[BF_K - 1]) {
#pragma unroll
    for (int q = 0; q < BF_K; q++) {
      const uint32_t lo = min(k[q], key);
      key = max(k[q], key);
      k[q] = lo;
    }
  }
}

// Dynamic LDS: u32 s_tile[BF_TILE * 8] | u32 s_list[cap * BF_K] | u16 s_i1[cap] | i16 s_match[cap] | u32 s_v2[cap/32] |
//              u32 s_m2[cap/32] | int s_hist[32] | int s_n1v
__global__ __launch_bounds__(BF_THREADS) void k_search_points(const sd_keypoint* __restrict__ kps1_all, const uint8_t* __restrict__ desc1_all,
                                                              const int32_t* __restrict__ n1_all, const sd_keypoint* __restrict__ kps2_all,
                                                              const uint8_t* __restrict__ desc2_all, const int32_t* __restrict__ n2_all,
                                                              TrackBuffers tb, float nnratio, int check_ori, int capw /* cap rounded up to 64 */,
                                                              int klist /* keys of a point's list phase 2 may use: BF_K (tests: fewer) */) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* s_tile = (uint32_t*)smem;
  uint32_t* s_list = s_tile + BF_TILE * 8;
  uint16_t* s_i1 = (uint16_t*)(s_list + (size_t)capw * BF_K);
  int16_t* s_match = (int16_t*)(s_i1 + capw);
  uint32_t* s_v2 = (uint32_t*)(s_match + capw);
  uint32_t* s_m2 = s_v2 + (capw >> 5);
  int* s_hist = (int*)(s_m2 + (capw >> 5));
  int* s_n1v = s_hist + 32;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int cap = tb.kp_cap;
  const int fc = tb.cur_bcast >= 0 ? tb.cur_bcast : f;
  const sd_keypoint* kps1 = kps1_all + (size_t)fc * cap;
  const sd_keypoint* kps2 = kps2_all + (size_t)f * cap;
  const uint32_t* desc1 = (const uint32_t*)(desc1_all + (size_t)fc * cap * 32);
  const uint32_t* desc2 = (const uint32_t*)(desc2_all + (size_t)f * cap * 32);
  const int N1 = min(n1_all[fc], cap), N2 = min(n2_all[f], cap);
  const uint8_t* v1 = tb.sp_valid1 + (size_t)f * cap;
  const uint8_t* v2 = tb.sp_valid2 + (size_t)f * cap;

  // ---- validity of the pKF points as bits; matched bits cleared; compacted list of the currentKF points (ascending)
  for (int i0 = 0; i0 < capw; i0 += BF_THREADS) {
    const int i = i0 + tid;
    const unsigned long long b = __ballot(i < N2 && v2[i] != 0);
    if (lane == 0 && i < capw) {
      s_v2[i >> 5] = (uint32_t)b;
      s_v2[(i >> 5) + 1] = (uint32_t)(b >> 32);
      s_m2[i >> 5] = 0;
      s_m2[(i >> 5) + 1] = 0;
    }
  }
  for (int i = tid; i < capw; i += BF_THREADS) s_match[i] = -1;
  if (tid < 32) s_hist[tid] = 0;
  if (wave == 0) {
    int n = 0;
    for (int i0 = 0; i0 < capw; i0 += 64) {
      const int i = i0 + lane;
      const bool ok = i < N1 && v1[i] != 0;
      const unsigned long long b = __ballot(ok);
      if (ok) s_i1[n + __popcll(b & lt)] = (uint16_t)i;
      n += __popcll(b);
    }
    if (lane == 0) *s_n1v = n;
  }
  __syncthreads();
  const int n1v = *s_n1v;
  const int ntiles = (N2 + BF_TILE - 1) / BF_TILE;

  // ---- phase 1
  for (int e0 = 0; e0 < n1v; e0 += BF_THREADS * BF_PT) {
    uint32_t d1[BF_PT][8], top[BF_PT][BF_K];
    int ent[BF_PT];
#pragma unroll
    for (int p = 0; p < BF_PT; p++) {
      ent[p] = e0 + p * BF_THREADS + tid;
      const int i1 = ent[p] < n1v ? s_i1[ent[p]] : 0;
      const uint4 a = ((const uint4*)(desc1 + (size_t)i1 * 8))[0], b = ((const uint4*)(desc1 + (size_t)i1 * 8))[1];
      d1[p][0] = a.x; d1[p][1] = a.y; d1[p][2] = a.z; d1[p][3] = a.w;
      d1[p][4] = b.x; d1[p][5] = b.y; d1[p][6] = b.z; d1[p][7] = b.w;
#pragma unroll
      for (int q = 0; q < BF_K; q++) top[p][q] = BF_INF;
    }
    for (int t = 0; t < ntiles; t++) {
      __syncthreads();   // previous tile fully consumed
      const int j0 = t * BF_TILE, jn = min(BF_TILE, N2 - j0);
      for (int w = tid; w < jn * 2; w += BF_THREADS) ((uint4*)s_tile)[w] = ((const uint4*)(desc2 + (size_t)j0 * 8))[w];
      __syncthreads();
      for (int wd = 0; wd < (jn + 31) >> 5; wd++) {
        uint32_t mask = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_v2[(j0 >> 5) + wd]);   // wave-uniform: scalar loop
        while (mask) {
          const int bit = __ffs((int)mask) - 1;
          mask &= mask - 1;
          const int jl = wd * 32 + bit;
          const uint4 a = ((const uint4*)(s_tile + jl * 8))[0], b = ((const uint4*)(s_tile + jl * 8))[1];   // LDS broadcast
          const uint32_t d2[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
          for (int p = 0; p < BF_PT; p++) bf_insert(top[p], ((uint32_t)bf_dist(d1[p], d2) << 11) | (uint32_t)(j0 + jl));
        }
      }
    }
#pragma unroll
    for (int p = 0; p < BF_PT; p++)
      if (ent[p] < n1v) {
#pragma unroll
        for (int q = 0; q < BF_K; q++) s_list[(size_t)ent[p] * BF_K + q] = top[p][q];
      }
  }
  __syncthreads();
  if (tid >= 64) return;   // one wavefront runs the order-dependent loop

  // ---- phase 2: for (idx1 ascending) best / second best among the pKF points not yet given away
  int nmatches = 0;
  for (int e = 0; e < n1v; e++) {
    uint32_t key = BF_INF;
    if (lane < klist) key = s_list[(size_t)e * BF_K + lane];
    const bool have = key != BF_INF;                         // the list ends where the pKF candidates end
    const bool taken = have && ((s_m2[(key & 2047) >> 5] >> (key & 31)) & 1u);
    const unsigned long long bhave = __ballot(have), bavail = __ballot(have && !taken);
    const int navail = __popcll(bavail);
    uint32_t k1 = BF_INF, k2 = BF_INF;
    if (navail >= 1) k1 = (uint32_t)__builtin_amdgcn_readlane((int)key, __ffsll((long long)bavail) - 1);
    if (navail >= 2) k2 = (uint32_t)__builtin_amdgcn_readlane((int)key, __ffsll((long long)(bavail & (bavail - 1))) - 1);
    const bool complete = __popcll(bhave) < klist;           // fewer than klist candidates exist at all: the list is everything
    if (navail < 2 && !complete) {
      // a candidate beyond the list could be the best or the second best; every such key is >= the list's last one
      const uint32_t klast = (uint32_t)__shfl((int)key, klist - 1);
      const uint32_t lower = navail >= 1 ? k1 : klast;       // smallest key the best can still have
      if ((int)(lower >> 11) < BF_TH_LOW) {
        // whole-wave recomputation of the row against the live vbMatched2
        const int i1 = s_i1[e];
        uint32_t a1[8];
        {
          const uint4 a = ((const uint4*)(desc1 + (size_t)i1 * 8))[0], b = ((const uint4*)(desc1 + (size_t)i1 * 8))[1];
          a1[0] = a.x; a1[1] = a.y; a1[2] = a.z; a1[3] = a.w; a1[4] = b.x; a1[5] = b.y; a1[6] = b.z; a1[7] = b.w;
        }
        uint32_t m0 = BF_INF, m1 = BF_INF;
        for (int j = lane; j < N2; j += 64) {
          if (!((s_v2[j >> 5] >> (j & 31)) & 1u) || ((s_m2[j >> 5] >> (j & 31)) & 1u)) continue;
          const uint4 a = ((const uint4*)(desc2 + (size_t)j * 8))[0], b = ((const uint4*)(desc2 + (size_t)j * 8))[1];
          const uint32_t b2[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
          const uint32_t kk = ((uint32_t)bf_dist(a1, b2) << 11) | (uint32_t)j;
          if (kk < m0) { m1 = m0; m0 = kk; }
          else if (kk < m1) m1 = kk;
        }
        k1 = __ockl_wfred_min_u32(m0);
        if (m0 == k1) m0 = m1;                               // keys are unique: exactly one lane gives up its first
        k2 = __ockl_wfred_min_u32(m0);
      } else {
        k1 = BF_INF;                                         // bestDist1 >= TH_LOW whatever lies beyond the list: no match
      }
    }
    const int bestDist1 = k1 == BF_INF ? 256 : (int)(k1 >> 11);
    const int bestDist2 = k2 == BF_INF ? 256 : (int)(k2 >> 11);
    if (bestDist1 < BF_TH_LOW && (float)bestDist1 < nnratio * (float)bestDist2) {
      const int j = k1 & 2047;
      if (lane == 0) {
        s_match[s_i1[e]] = (int16_t)j;
        s_m2[j >> 5] |= 1u << (j & 31);
      }
      nmatches++;
    }
  }
  // ---- rotation consistency (src/ORBmatcher.cc:1269-1296), after the loop: the bins never influence a match decision
  if (check_ori) {
    const float factor = 1.0f / BF_HISTO;
    for (int e = lane; e < n1v; e += 64) {
      const int i1 = s_i1[e], j = s_match[i1];
      if (j < 0) continue;
      float rot = kps1[i1].angle - kps2[j].angle;
      if (rot < 0.0) rot += 360.0f;
      int bin = (int)roundf(rot * factor);
      if (bin == BF_HISTO) bin = 0;
      atomicAdd(&s_hist[bin], 1);
      s_list[e] = (uint32_t)bin;   // phase 1's lists are dead
    }
    __threadfence_block();
    int ind1 = -1, ind2 = -1, ind3 = -1;
    int max1 = 0, max2 = 0, max3 = 0;
    for (int b = 0; b < BF_HISTO; b++) {
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
    int dropped = 0;
    for (int e0 = 0; e0 < n1v; e0 += 64) {
      const int e = e0 + lane;
      bool drop = false;
      if (e < n1v) {
        const int i1 = s_i1[e];
        if (s_match[i1] >= 0) {
          const int bin = (int)s_list[e];
          if (bin != ind1 && bin != ind2 && bin != ind3) {
            drop = true;
            s_match[i1] = -1;
          }
        }
      }
      dropped += __popcll(__ballot(drop));
    }
    nmatches -= dropped;
  }
  int32_t* out = tb.sp_match + (size_t)f * cap;
  for (int i = lane; i < cap; i += 64) out[i] = (int32_t)s_match[i];
  if (lane == 0) tb.sp_n[f] = nmatches;
}

int launch_search_points(const sd_orb* cur, const sd_orb* ref, const TrackBuffers& tb, int n_frames, float nnratio, int check_ori,
                         hipStream_t s) {
  const int capw = (tb.kp_cap + 63) & ~63;
  SD_REQUIRE(capw <= 2048, SD_ERR_CAPACITY, "SearchByPoints supports at most 2048 keypoints per keyframe");
  // option "track.bf_list_k" = 1..3 (tests): phase 2 sees only the first keys of every list, so the whole-wave recomputation -- rare on real
  // data -- runs for most points
  const int klist = std::min(BF_K, std::max(1, opt(OPT_BF_LIST_K)));
  const size_t lds = (size_t)BF_TILE * 32 + (size_t)capw * BF_K * 4 + (size_t)capw * 2 * 2 + (size_t)(capw >> 5) * 4 * 2 + 8 + 33 * 4 + 16;
  hipLaunchKernelGGL(k_search_points, dim3(n_frames), dim3(BF_THREADS), lds, s, (cur->have_dist ? cur->d_kps_un : cur->d_kps), cur->d_desc,
                     cur->d_nout, (ref->have_dist ? ref->d_kps_un : ref->d_kps), ref->d_desc, ref->d_nout, tb, nnratio, check_ori, capw, klist);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

}  // namespace sd
