// Host-side check of sdslam_amd/csrc/introselect.h against the real std::nth_element
// (the algorithm cv::KeyPointsFilter::retainBest runs in the reference).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
static long g_heap_calls = 0;
#define SDSEL_TRACE_HEAP() (++g_heap_calls)
#include "../../sdslam_amd/csrc/introselect.h"

struct KP { float x, y, size, angle, response; int octave, class_id; };

static bool run_case(const std::vector<uint32_t>& keys, int nth) {
  std::vector<KP> ref(keys.size());
  for (size_t i = 0; i < keys.size(); i++) ref[i] = KP{(float)(keys[i] & 0xfff), (float)((keys[i] >> 12) & 0xfff), 7.f, -1.f, (float)(keys[i] >> 24), 0, -1};
  std::nth_element(ref.begin(), ref.begin() + nth, ref.end(), [](const KP& a, const KP& b) { return a.response > b.response; });
  std::vector<uint32_t> mine = keys;
  sdsel::nth_element(mine.data(), (int)mine.size(), nth);
  for (size_t i = 0; i < keys.size(); i++) {
    uint32_t k = ((uint32_t)ref[i].response << 24) | ((uint32_t)ref[i].y << 12) | (uint32_t)ref[i].x;
    if (k != mine[i]) return false;
  }
  return true;
}

extern "C" long introselect_selftest(int seed, int ncases, int maxn, int resp_span) {
  std::mt19937 rng(seed);
  long bad = 0;
  for (int c = 0; c < ncases; c++) {
    int n = 1 + rng() % maxn;
    std::vector<uint32_t> keys(n);
    int span = 1 + rng() % resp_span;
    for (int i = 0; i < n; i++) keys[i] = ((20u + rng() % span) << 24) | ((uint32_t)(i / 64) << 12) | (uint32_t)(i % 64);
    int mode = rng() % 4;
    if (mode == 1) std::sort(keys.begin(), keys.end());
    if (mode == 2) std::sort(keys.begin(), keys.end(), std::greater<uint32_t>());
    int nth = rng() % n;
    if (!run_case(keys, nth)) bad++;
  }
  return bad;
}

// adversarial search: small distinct-valued permutations that exhaust the depth limit
extern "C" long introselect_heap_cases(int seed, int tries, long* heap_calls_out) {
  std::mt19937 rng(seed);
  long bad = 0;
  g_heap_calls = 0;
  for (int t = 0; t < tries; t++) {
    int n = 8 + rng() % 40;
    std::vector<uint32_t> keys(n);
    // organ-pipe / sawtooth shapes make median-of-3 partitions maximally unbalanced
    int shape = rng() % 3;
    for (int i = 0; i < n; i++) {
      int v = shape == 0 ? (i % 2 ? i : n - i) : shape == 1 ? ((i * 7) % n) : (i < n / 2 ? i : n - i);
      keys[i] = ((uint32_t)(20 + (v % 200)) << 24) | (uint32_t)i;
    }
    for (int s = 0; s < 3; s++) std::swap(keys[rng() % n], keys[rng() % n]);
    int nth = rng() % n;
    if (!run_case(keys, nth)) bad++;
  }
  *heap_calls_out = g_heap_calls;
  return bad;
}

// ---- the bulk (ballot-style) partition formulation of introselect.h / introselect_wave.h ----------
// Stops are taken from a SNAPSHOT of the range, paired by rank, all swaps applied at once.
static int partition_snapshot(uint32_t* a, int lo0, int hi0, uint32_t pivot) {
  std::vector<int> Ls, Rs;
  for (int i = lo0; i < hi0; i++)
    if (!sdsel::gt(a[i], pivot)) Ls.push_back(i);
  for (int j = hi0 - 1; j >= lo0; j--)
    if (!sdsel::gt(pivot, a[j])) Rs.push_back(j);
  size_t K = 0;
  while (K < Ls.size() && K < Rs.size() && Ls[K] < Rs[K]) K++;
  for (size_t k = 0; k < K; k++) std::swap(a[Ls[k]], a[Rs[k]]);
  const int BIG = 0x7fffffff;
  const int LK = K < Ls.size() ? Ls[K] : BIG, Rprev = K > 0 ? Rs[K - 1] : BIG;
  return LK < Rprev ? LK : Rprev;
}

static void nth_element_snapshot(uint32_t* a, int n, int nth) {
  using namespace sdsel;
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
    const int cut = partition_snapshot(a, first + 1, last, a[first]);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  insertion_sort(a, first, last);
}

extern "C" long introselect_snapshot_selftest(int seed, int ncases, int maxn, int resp_span) {
  std::mt19937 rng(seed);
  long bad = 0;
  for (int c = 0; c < ncases; c++) {
    int n = 1 + rng() % maxn;
    std::vector<uint32_t> keys(n);
    int span = 1 + rng() % resp_span;
    for (int i = 0; i < n; i++) keys[i] = ((20u + rng() % span) << 24) | ((uint32_t)(i / 64) << 12) | (uint32_t)(i % 64);
    int mode = rng() % 4;
    if (mode == 1) std::sort(keys.begin(), keys.end());
    if (mode == 2) std::sort(keys.begin(), keys.end(), std::greater<uint32_t>());
    int nth = rng() % n;
    std::vector<uint32_t> a = keys, b = keys;
    sdsel::nth_element(a.data(), n, nth);
    nth_element_snapshot(b.data(), n, nth);
    if (a != b) bad++;
  }
  return bad;
}
