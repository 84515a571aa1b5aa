// Internal layout of the sd_orb handle (shared by orb.hip and track.hip; not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>

#include "orb_plan.h"
#include "sd_common.h"

enum { ST_PYR = 0, ST_FAST, ST_SELECT, ST_BLUR, ST_DESC, ST_COUNT };

struct sd_orb {
  int nfeatures, nlevels, thFAST;
  float scaleFactor;
  int max_w, max_h, max_batch, device;
  sd::HostPlan hp;
  bool have_geom = false;
  int cur_w = 0, cur_h = 0;
  int last_frames = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  // blur depends only on the pyramid: it runs on aux_stream beside FAST + selection
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_pyr_done = nullptr, ev_blur_done = nullptr, ev_blur_start = nullptr;
  // FAST of level l needs only level l of the pyramid: it runs on fast_stream as soon as that level is complete,
  // beside the (small, dependent) resize launches of the remaining levels
  hipStream_t fast_stream = nullptr;
  hipEvent_t ev_level[SD_MAX_LEVELS] = {};
  hipEvent_t ev_fast_done = nullptr;
  hipEvent_t ev_select_done = nullptr, ev_body_start = nullptr;   // end of a call's selection (d_cand free again) / start of a call on `stream`
  bool select_recorded = false, user_fence_live = false, staging_input = false;
  // Output sets.  What a tracker reads (padded pyramid, keypoints, descriptors, counts) exists once
  // or -- after sd::orb_enable_double_buffer, which sd_track_create calls on its `cur` handle -- twice:
  // extraction alternates between the sets, so batch n+1 is extracted on `stream` while the tracker
  // still works on batch n on its own stream.  d_pyr / d_kps / d_kps_un / d_desc / d_nout below always
  // alias the set of the most recent extraction.  A tracker records ev_set_free[set] after every
  // kernel that reads a set; the extraction that is about to overwrite that set waits for it.
  int nsets = 1, set = 0;
  uint8_t* pyr_set[2] = {nullptr, nullptr};
  sd_keypoint* kps_set[2] = {nullptr, nullptr};
  sd_keypoint* kps_un_set[2] = {nullptr, nullptr};
  uint8_t* desc_set[2] = {nullptr, nullptr};
  int32_t* nout_set[2] = {nullptr, nullptr};
  hipEvent_t ev_set_free[2] = {nullptr, nullptr};
  bool set_busy[2] = {false, false};
  hipEvent_t ev_user_fence[2] = {nullptr, nullptr};   // sd_orb_stream_fence
  hipEvent_t ev_extract_done = nullptr;   // end of the most recent extraction on `stream`
  bool extract_recorded = false;
  unsigned long long extract_serial = 0;  // extractions launched so far (a tracker checks that its inputs have not been replaced)
  bool pyr_event_live = false;            // ev_pyr_done of the most recent extraction is a real record (not a captured graph node)
  // device buffers
  sd::OrbPlan* d_plan = nullptr;
  sd::CellGeom* d_cells = nullptr;
  sd::BlurTile* d_tiles = nullptr;
  int32_t* d_coef = nullptr;
  uint8_t* d_img = nullptr;     // staging for host-input calls
  uint8_t* d_pyr = nullptr;
  uint8_t* d_blur = nullptr;
  uint32_t* d_cand = nullptr;
  uint32_t* d_scratch = nullptr;
  int32_t* d_cell_count = nullptr;
  uint32_t* d_sel = nullptr;
  int32_t* d_sel_count = nullptr;
  int32_t* d_cell_keep = nullptr;   // split selection: kept count / offset in the level list per (frame, cell), list length per (frame, level)
  int32_t* d_cell_off = nullptr;
  int32_t* d_lvl_m = nullptr;
  sd_keypoint* d_kps = nullptr;
  sd_keypoint* d_kps_un = nullptr;   // Frame::mvKeysUn (== d_kps when k1 == 0)
  bool have_dist = false;
  float dist_K[4] = {0, 0, 0, 0};    // fx, fy, cx, cy as CV_32F (Converter::toCvMat)
  float dist[5] = {0, 0, 0, 0, 0};   // k1, k2, p1, p2, k3
  uint8_t* d_desc = nullptr;
  int32_t* d_nout = nullptr;
  size_t cap_pyr = 0, cap_cand = 0, cap_cells = 0, cap_tiles = 0, cap_coef = 0;
  // hipGraph cache of the extraction pipeline (multi-stream fork/join captured once per argument set); opt-in with
  // option "extract.use_graph" -- see launch_pipeline for the measurement that keeps direct launches the default
  struct GraphEntry {
    const void* imgs = nullptr;
    int n = 0, stride = 0, set = -1;
    size_t frame_stride = 0;
    bool dist = false;
    float distv[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    hipGraphExec_t exec = nullptr;
  };
  GraphEntry graphs[4];
  int graph_next = 0;
  bool profiling = false;
  // ring of per-call stage events: the bench reads mean stage times over its whole timed region
  static const int kRing = 128;
  // per call: [0] start, [1] pyramid end, [8]/[2] FAST start/end (fast stream), [9]/[7] select start/end, [3]/[6] blur start/end
  // (aux stream), [4]/[5] descriptor start/end
  hipEvent_t ev[kRing][10] = {};
  // one (start, stop) pair per k_fast_cells launch: the FAST stage is reported as the SUM of its launches' durations (the stream
  // waits for pyramid levels between them; ev[8] .. ev[2] would count those gaps).  Created when profiling is first switched on.
  static const int kFastPairs = SD_MAX_LEVELS + 1;
  hipEvent_t evf[kRing][2 * kFastPairs] = {};
  int evf_n[kRing] = {};
  bool evf_ready = false;
  int ev_calls = 0;   // calls recorded since profiling was (re-)enabled
};

namespace sd {
int orb_enable_double_buffer(sd_orb* h);   // orb.hip; idempotent
}
