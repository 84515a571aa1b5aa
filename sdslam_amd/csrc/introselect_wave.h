// Wavefront-parallel replay of libstdc++'s std::nth_element on packed candidate keys: the same
// permutation as sdsel::nth_element (introselect.h), computed by the 64 lanes of one wave.
//
// Introselect's control flow (median-of-3, cut <= nth ? first = cut : last = cut, depth limit, final
// insertion sort) is kept as is -- it is a short serial chain.  The O(n) part, the unguarded Hoare
// partition, uses the bulk formulation derived in introselect.h: one sweep over the range in chunks of
// 64 finds both stop sets of the ORIGINAL range with ballots; ballots are wave-uniform, so the running
// stop count is a scalar and every stop lane knows its rank (count so far + popcount of the lower
// lanes) and parks its position in a small LDS table tmpL[rank] / tmpR[rank].  Then lane k pairs
// tmpL[k] with tmpR[k], every valid pair (L_k < R_k) is swapped at once, and the cut is
// min(L_K, R_{K-1}).  Tables hold WAVE_SEL_CAP stops per side; if all of them pair up, the partition
// simply continues on the remaining inner range (that is what the sequential loop does, too).
// tests/native/introselect_check.cc checks the formulation against std::nth_element on the host;
// tests/test_orb_gpu.py checks the kernels that use this file bit for bit.
//
// All 64 lanes of the wave must call with identical arguments; `a` may be an LDS (address space 3) or a
// generic pointer to memory that only this wave touches; tmp = 2 * WAVE_SEL_CAP uint16 in LDS.
#pragma once
#include <hip/hip_runtime.h>

#include "introselect.h"

namespace sdsel {

__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); }

#define WAVE_SEL_CAP 256
typedef __attribute__((address_space(3))) uint16_t lds_u16;

template <typename P>
__device__ __forceinline__ void wave_nth_element(P a, int n, int nth, lds_u16* tmp) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int BIG = 0x7fffffff;
  lds_u16* tmpL = tmp;
  lds_u16* tmpR = tmp + WAVE_SEL_CAP;
  if (n <= 0 || nth >= n) return;
  int first = 0, last = n;
  int depth = 2 * (31 - __clz(n));
  while (last - first > 3) {
    if (depth == 0) {   // std::__heap_select fallback (adversarial inputs only): serial
      if (lane == 0) {
        uint32_t* g = (uint32_t*)a;
        heap_select(g + first, nth + 1 - first, last - first);
        swp(g, first, nth);
      }
      wave_fence();
      return;
    }
    --depth;
    const int mid = first + (last - first) / 2;
    // __move_median_to_first(first, first + 1, mid, last - 1): every lane decides, lane 0 writes
    const uint32_t va = a[first + 1], vb = a[mid], vc = a[last - 1], vr = a[first];
    int src;
    if (gt(va, vb)) {
      if (gt(vb, vc)) src = mid;
      else if (gt(va, vc)) src = last - 1;
      else src = first + 1;
    } else if (gt(va, vc)) src = first + 1;
    else if (gt(vb, vc)) src = last - 1;
    else src = mid;
    const uint32_t pivot = src == mid ? vb : (src == first + 1 ? va : vc);
    if (lane == 0) {
      a[first] = pivot;
      a[src] = vr;
    }
    wave_fence();
    // ---- __unguarded_partition(first + 1, last, pivot), bulk form
    int lo0 = first + 1, hi0 = last, lastR = BIG, cut;
    while (true) {
      const int m = hi0 - lo0;
      int cntL = 0, cntR = 0;   // wave-uniform running stop counts
      for (int q0 = 0; q0 < m; q0 += 64) {
        const int q = q0 + lane;
        const bool in = q < m;
        const uint32_t vL = in ? a[lo0 + q] : 0u, vR = in ? a[hi0 - 1 - q] : 0u;
        const bool sL = in && !gt(vL, pivot), sR = in && !gt(pivot, vR);
        const unsigned long long bL = __ballot(sL), bR = __ballot(sR);
        const int rL = cntL + __popcll(bL & lt), rR = cntR + __popcll(bR & lt);
        if (sL && rL < WAVE_SEL_CAP) tmpL[rL] = (uint16_t)q;
        if (sR && rR < WAVE_SEL_CAP) tmpR[rR] = (uint16_t)q;
        cntL += __popcll(bL);
        cntR += __popcll(bR);
        if (cntL >= WAVE_SEL_CAP && cntR >= WAVE_SEL_CAP) break;   // both tables full: later stops pair in the next round
      }
      wave_fence();
      const int capL = min(cntL, WAVE_SEL_CAP), capR = min(cntR, WAVE_SEL_CAP);
      const int kmax = min(capL, capR);
      int K = 0;
      for (int k0 = 0; k0 < kmax; k0 += 64) {
        const int k = k0 + lane;
        const bool have = k < kmax;
        const int Lk = have ? lo0 + (int)tmpL[k] : 0, Rk = have ? hi0 - 1 - (int)tmpR[k] : 0;
        const bool valid = have && Lk < Rk;
        if (valid) {
          const uint32_t x = a[Lk], y = a[Rk];
          a[Lk] = y;
          a[Rk] = x;
        }
        const unsigned long long bv = __ballot(valid);
        K += __popcll(bv);
        if (bv != ~0ull) break;
      }
      wave_fence();
      if (K == WAVE_SEL_CAP) {   // every tabulated pair swapped: go on inside (L_{K-1}, R_{K-1})
        const int nlo = lo0 + (int)tmpL[K - 1] + 1, nhi = hi0 - 1 - (int)tmpR[K - 1];
        lastR = nhi;
        lo0 = nlo;
        hi0 = nhi;
        continue;
      }
      // K < table size on at least one side that is not truncated, or the first invalid pair was seen
      const int LK = K < capL ? lo0 + (int)tmpL[K] : BIG;
      const int Rp = K > 0 ? hi0 - 1 - (int)tmpR[K - 1] : lastR;
      cut = min(LK, Rp);
      break;
    }
    if (cut <= nth) first = cut;
    else last = cut;
  }
  // __insertion_sort on the last <= 3 elements (registers; same moves as sdsel::insertion_sort)
  const int c = last - first;
  if (c >= 2) {
    uint32_t v0 = a[first], v1 = a[first + 1], v2 = c == 3 ? a[first + 2] : 0u;
    if (gt(v1, v0)) {
      const uint32_t t = v0;
      v0 = v1;
      v1 = t;
    }
    if (c == 3) {
      if (gt(v2, v0)) {
        const uint32_t t = v2;
        v2 = v1;
        v1 = v0;
        v0 = t;
      } else if (gt(v2, v1)) {
        const uint32_t t = v2;
        v2 = v1;
        v1 = t;
      }
    }
    if (lane == 0) {
      a[first] = v0;
      a[first + 1] = v1;
      if (c == 3) a[first + 2] = v2;
    }
  }
  wave_fence();
}

}  // namespace sdsel
