// Replay of libstdc++'s std::nth_element (bits/stl_algo.h, GCC 11: __introselect with
// median-of-3 + unguarded Hoare partition, __heap_select fallback at depth 2*floor(log2 n),
// final insertion sort on <= 3 elements) on packed 32-bit candidate keys.
//
// Why: the reference trims keypoints with cv::KeyPointsFilter::retainBest, i.e.
// std::nth_element(begin, begin+n, end, response-greater) followed by resize(n)
// (reference src/ORBextractor.cc:586-588,602-603).  FAST scores are small integers, ties are
// everywhere, so WHICH tied corners survive and the ORDER of the survivors are whatever
// libstdc++'s introselect leaves behind.  To be bit-exact in keypoint identity and order the
// device runs the same sequence of comparisons and moves.  The comparator looks only at the
// response byte (bits 31..24); the low 24 bits (y:12, x:12) ride along like the rest of a
// cv::KeyPoint would.  tests/test_introselect.py checks this file against the real
// std::nth_element on the host for random/tied/adversarial inputs.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SD_HD __host__ __device__ inline
#else
#define SD_HD inline
#endif

namespace sdsel {

// comp(a, b) == (a.response > b.response)
SD_HD bool gt(uint32_t a, uint32_t b) { return (a >> 24) > (b >> 24); }

SD_HD void swp(uint32_t* a, int i, int j) {
  uint32_t t = a[i];
  a[i] = a[j];
  a[j] = t;
}

SD_HD void push_heap(uint32_t* a, int hole, int top, uint32_t value) {
  int parent = (hole - 1) / 2;
  while (hole > top && gt(a[parent], value)) {
    a[hole] = a[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[hole] = value;
}

SD_HD void adjust_heap(uint32_t* a, int hole, int len, uint32_t value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (gt(a[child], a[child - 1])) child--;
    a[hole] = a[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[hole] = a[child - 1];
    hole = child - 1;
  }
  push_heap(a, hole, top, value);
}

// std::__heap_select(first, middle, last)
SD_HD void heap_select(uint32_t* a, int middle, int last) {
#ifdef SDSEL_TRACE_HEAP
  SDSEL_TRACE_HEAP();
#endif
  const int len = middle;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      uint32_t v = a[parent];
      adjust_heap(a, parent, len, v);
      if (parent == 0) break;
      parent--;
    }
  }
  for (int i = middle; i < last; ++i) {
    if (gt(a[i], a[0])) {
      uint32_t v = a[i];
      a[i] = a[0];
      adjust_heap(a, 0, len, v);
    }
  }
}

SD_HD void insertion_sort(uint32_t* a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    uint32_t val = a[i];
    if (gt(val, a[first])) {
      for (int k = i; k > first; --k) a[k] = a[k - 1];
      a[first] = val;
    } else {
      int l = i, nx = i - 1;
      while (gt(val, a[nx])) {
        a[l] = a[nx];
        l = nx;
        --nx;
      }
      a[l] = val;
    }
  }
}

// std::nth_element(a, a+nth, a+n, gt)
SD_HD void nth_element(uint32_t* a, int n, int nth) {
  if (n <= 0 || nth >= n) return;
  int first = 0, last = n;
  int depth = 2 * (31 - __builtin_clz((unsigned)n));
  while (last - first > 3) {
    if (depth == 0) {
      heap_select(a + first, nth + 1 - first, last - first);
      swp(a, first, nth);
      return;
    }
    --depth;
    // __unguarded_partition_pivot
    int mid = first + (last - first) / 2;
    {
      const int r = first, ia = first + 1, ib = mid, ic = last - 1;
      if (gt(a[ia], a[ib])) {
        if (gt(a[ib], a[ic])) swp(a, r, ib);
        else if (gt(a[ia], a[ic])) swp(a, r, ic);
        else swp(a, r, ia);
      } else if (gt(a[ia], a[ic])) swp(a, r, ia);
      else if (gt(a[ib], a[ic])) swp(a, r, ic);
      else swp(a, r, ib);
    }
    int lo = first + 1, hi = last;
    const uint32_t pivot = a[first];  // never moves during the partition
    while (true) {
      while (gt(a[lo], pivot)) ++lo;
      --hi;
      while (gt(pivot, a[hi])) --hi;
      if (!(lo < hi)) break;
      swp(a, lo, hi);
      ++lo;
    }
    const int cut = lo;
    if (cut <= nth) first = cut;
    else last = cut;
  }
  insertion_sort(a, first, last);
}

// ---- bulk (data-parallel) formulation of the unguarded Hoare partition ---------------------------
// The sequential partition loop
//     while (true) { while (gt(a[lo], p)) ++lo;  --hi;  while (gt(p, a[hi])) --hi;
//                    if (!(lo < hi)) return lo;  swap(a[lo], a[hi]);  ++lo; }
// only ever swaps the k-th "left stop" L_k (k-th position from the left, in the ORIGINAL array, whose
// element is not gt the pivot) with the k-th "right stop" R_k (k-th position from the right whose
// element the pivot is not gt), for k = 0 .. K-1 where K is the first k with !(L_k < R_k): both scans
// run over elements no swap has touched yet.  The returned cut is L_K if it lies below R_{K-1}, else
// R_{K-1} (which by then holds a left-stop value); L_0 when K == 0.  So a wavefront can compute both
// stop sets with ballots, pair them by rank and apply all swaps at once (introselect_wave.h).
// tests/native/introselect_check.cc states this formulation on a snapshot of the array and checks it
// against std::nth_element.

}  // namespace sdsel
