// sd_set_option / sd_get_option: the library's tuning and test switches as ONE documented, process-wide table
// (include/sdslam_hip.h lists them).  Nothing in the library reads the environment: a stray variable cannot change LDS
// budgets or stream ordering of a drop-in; a harness that wants a non-default setting says so through the ABI.
#include <atomic>
#include <cstring>

#include "sd_common.h"

namespace sd {

struct OptDef {
  const char* name;
  int def, lo, hi;
};

// order = enum Opt (sd_common.h)
static const OptDef kOpts[OPT_COUNT] = {
    {"extract.fast0_from_frames", 1, 0, 1},
    {"extract.use_graph", 0, 0, 1},
    {"extract.select_small_cap", 0, 0, 1 << 20},
    {"extract.select_big_cap", 0, 0, 1 << 20},
    {"extract.fast_merge_from", 6, 1, 64},
    {"extract.fast_lds_kb", 24, 4, 160},
    {"extract.fast_lds_whole_kb", 40, 4, 160},
    {"track.stream_priority", 2, 0, 2},
    {"track.align_start", 2, 0, 2},
    {"track.align_min_waves", 5, 3, 5},
    {"track.bf_list_k", 4, 1, 4},
    {"track.poseopt_waves", 0, 0, 4},
    {"track.match_split", 1, 0, 1},
    {"extract.fast0_early", 1, 0, 1},
    {"extract.pyr_early", 0, 0, 1},
};

static std::atomic<int> g_val[OPT_COUNT];
static std::atomic<bool> g_init{false};

static void init_once() {
  if (g_init.load(std::memory_order_acquire)) return;
  static std::atomic_flag busy = ATOMIC_FLAG_INIT;
  while (busy.test_and_set(std::memory_order_acquire)) {}
  if (!g_init.load(std::memory_order_relaxed)) {
    for (int i = 0; i < OPT_COUNT; i++) g_val[i].store(kOpts[i].def, std::memory_order_relaxed);
    g_init.store(true, std::memory_order_release);
  }
  busy.clear(std::memory_order_release);
}

int opt(Opt o) {
  init_once();
  return g_val[o].load(std::memory_order_relaxed);
}

}  // namespace sd

using namespace sd;

extern "C" {

int sd_set_option(const char* name, int value) {
  SD_REQUIRE(name, SD_ERR_INVALID_ARG, "option name is NULL");
  init_once();
  for (int i = 0; i < OPT_COUNT; i++)
    if (std::strcmp(name, kOpts[i].name) == 0) {
      SD_REQUIRE(value >= kOpts[i].lo && value <= kOpts[i].hi, SD_ERR_INVALID_ARG, std::string("value out of range for option ") + name);
      g_val[i].store(value, std::memory_order_relaxed);
      return SD_OK;
    }
  set_error(std::string("unknown option: ") + name);
  return SD_ERR_INVALID_ARG;
}

int sd_get_option(const char* name, int* value) {
  SD_REQUIRE(name && value, SD_ERR_INVALID_ARG, "NULL argument");
  init_once();
  for (int i = 0; i < OPT_COUNT; i++)
    if (std::strcmp(name, kOpts[i].name) == 0) {
      *value = g_val[i].load(std::memory_order_relaxed);
      return SD_OK;
    }
  set_error(std::string("unknown option: ") + name);
  return SD_ERR_INVALID_ARG;
}

int sd_option_count(void) { return OPT_COUNT; }
const char* sd_option_name(int index) { return (index >= 0 && index < OPT_COUNT) ? kOpts[index].name : nullptr; }

}  // extern "C"
