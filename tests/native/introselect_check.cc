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
