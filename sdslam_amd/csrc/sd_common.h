// Shared host-side plumbing for libsdslam_hip.so (error reporting, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/sdslam_hip.h"

namespace sd {

void set_error(const std::string& msg);

#define SD_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      ::sd::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + \
                      ":" + std::to_string(__LINE__) + ")");                            \
      return SD_ERR_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define SD_REQUIRE(cond, code, msg) \
  do {                              \
    if (!(cond)) {                  \
      ::sd::set_error(msg);         \
      return (code);                \
    }                               \
  } while (0)

// Process-wide options (sd_set_option / sd_get_option, sd_options.cpp; documented in include/sdslam_hip.h).
enum Opt {
  OPT_FAST0_FROM_FRAMES, OPT_USE_GRAPH, OPT_SELECT_SMALL_CAP, OPT_SELECT_BIG_CAP, OPT_FAST_MERGE_FROM, OPT_FAST_LDS_KB,
  OPT_FAST_LDS_WHOLE_KB, OPT_TRACK_PRIORITY, OPT_ALIGN_START, OPT_ALIGN_MIN_WAVES, OPT_BF_LIST_K,
  OPT_POSEOPT_WAVES, OPT_MATCH_SPLIT, OPT_FAST0_EARLY, OPT_PYR_EARLY, OPT_COUNT
};
int opt(Opt o);

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace sd
