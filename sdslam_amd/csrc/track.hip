// sd_track_*: batched TrackWithMotionModel context (ImageAlign -> SearchByProjection -> pose solve)
// over two resident extractor handles (current frames, last frames).  Host side only; the
// kernels live in track_align.hip / track_match.hip / track_pnp.hip / track_poseopt.hip.
// Besides the stage calls: whole-function calls that keep the reference's per-frame decisions on
// the device (sd_track_with_motion_model, sd_track_local_map) and the one-frame-against-all-
// keyframes calls (sd_track_relocalize, sd_track_detect_loop) built on the broadcast current frame.
//
// Call sequence of the reference this mirrors (src/Tracking.cc:654-718):
//   ImageAlign::ComputePose(cur, last)            -> sd_track_align
//   ORBmatcher(0.9,true).SearchByProjection(...)  -> sd_track_match
//   pose solve (PnPsolver per BASELINE; the reference calls Optimizer::PoseOptimization, D1)
//                                                 -> sd_track_pnp
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "orb_internal.h"
#include "track_internal.h"

using namespace sd;

struct sd_track {
  sd_orb* cur = nullptr;
  sd_orb* ref = nullptr;
  int max_points = 0, max_batch = 0, kp_cap = 0, device = 0;
  int rand_per_frame = 0;
  std::vector<int> rand_len;      // rand() values actually supplied per slot (sd_track_set_rand)
  bool have_pnp = false;          // sd_track_pnp has constructed the solvers sd_track_pnp_iterate continues ...
  unsigned long long pnp_serial = 0;   // ... on the keypoints of THIS extraction of `cur` (the reference's solver owns copies of its inputs)
  PnpParams pnp_params{};
  int pnp_frames = 0, pnp_iter_upper = 0;   // slots / upper bound of mnIterations of those solvers
  TrackBuffers tb{};
  TrackCam cam{};
  bool have_cam = false;
  float* d_sf = nullptr;
  float* d_inv_sf = nullptr;
  float* d_sigma2 = nullptr;
  float* d_inv_sigma2 = nullptr;
  float* d_scale_thr = nullptr;   // MapPoint::PredictScale breakpoints (see k_match_local)
  std::vector<void*> allocs;
  // The tracking kernels (align, match, PnP: latency-bound, few waves) run on their own stream, so
  // the extraction of the next batch on cur->stream overlaps them; `cur` is double-buffered
  // (orb_internal.h: output sets) and every tracking launch waits for the extraction it consumes
  // (ev_extract_done) and marks the sets it read (ev_set_free).
  hipStream_t pnp_stream = nullptr;
  bool profiling = false;
  hipEvent_t ev_fence = nullptr;   // sd_track_stream_fence
  static const int kRing = 128;
  hipEvent_t ev[kRing][6] = {};
  int ev_calls[3] = {0, 0, 0};
};

// Per-frame result record (the only data that crosses xGMI in the batched-frames mode), 20 doubles:
// pose 4x4 column-major | ImageAlign ok | nmatches | pose-solver inliers | pose-solver ok
//   source 0: PnPsolver (pose = Tcw it returned, zeros for an empty Mat)   1: PoseOptimization (ok = nGood >= 10)
//   source 2: TrackWithMotionModel (inliers = nmatchesMap, ok = tracked)   3: TrackLocalMap (mnMatchesInliers, tracked)
//   source 4: ImageAlign only (pose = the aligned pose)
__global__ void k_pack_records(TrackBuffers tb, int source, int n_frames, double* __restrict__ out) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  double* r = out + (size_t)f * 20;
  if (source == 0) {
    const float* T = tb.pnp_T + (size_t)f * 16;   // row-major CV_32F
    for (int c = 0; c < 4; c++)
      for (int rr = 0; rr < 4; rr++) r[c * 4 + rr] = (double)T[rr * 4 + c];
  } else {
    const double* T = (source == 4 ? tb.Tcur : tb.po_T) + (size_t)f * 16;
    for (int i = 0; i < 16; i++) r[i] = T[i];
  }
  r[16] = (double)tb.al_ok[f];
  r[17] = (double)tb.n_matches[f];
  double inl = 0, ok = 0;
  if (source == 0) { inl = tb.pnp_info[(size_t)f * 8 + 1]; ok = tb.pnp_info[(size_t)f * 8]; }
  else if (source == 1) { inl = tb.po_info[(size_t)f * 8 + 5]; ok = inl >= 10; }
  else if (source == 2) { inl = tb.tw_info[(size_t)f * 4 + 2]; ok = tb.tw_info[(size_t)f * 4] == 2; }
  else if (source == 3) { inl = tb.tl_info[(size_t)f * 4 + 2]; ok = tb.tl_info[(size_t)f * 4] == 2; }
  else { ok = tb.al_ok[f]; }
  r[18] = inl;
  r[19] = ok;
}

template <typename T>
static int dalloc(sd_track* h, T** p, size_t count) {
  void* q = nullptr;
  SD_HIP_CHECK(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
  SD_HIP_CHECK(hipMemset(q, 0, std::max<size_t>(count, 1) * sizeof(T)));
  h->allocs.push_back(q);
  *p = (T*)q;
  return SD_OK;
}

extern "C" {

int sd_track_create(sd_orb* cur, sd_orb* ref, int max_points, int max_batch, int pnp_max_iterations, sd_track** out) {
  SD_REQUIRE(out, SD_ERR_INVALID_ARG, "out is NULL");
  *out = nullptr;
  SD_REQUIRE(cur && ref, SD_ERR_INVALID_ARG, "extractor handle is NULL");
  SD_REQUIRE(cur->device == ref->device && cur->nlevels == ref->nlevels && cur->scaleFactor == ref->scaleFactor,
             SD_ERR_INVALID_ARG, "cur/ref extractors must share device and pyramid parameters");
  // `cur` may hold fewer frames than the tracker has slots: one current frame against many keyframes
  // (sd_track_set_current_broadcast); slot-paired calls then check their n_frames against what `cur` extracted
  SD_REQUIRE(max_points >= 1 && max_points <= 2048 && max_batch >= 1 && max_batch <= ref->max_batch, SD_ERR_INVALID_ARG,
             "bad capacities (max_points <= 2048, max_batch <= the ref extractor's max_batch)");
  SD_REQUIRE(pnp_max_iterations >= 1 && pnp_max_iterations <= 4096, SD_ERR_INVALID_ARG, "bad pnp_max_iterations");
  int nsel = 0;
  for (int q : cur->hp.quota) nsel += q;
  SD_REQUIRE(nsel >= 1 && nsel <= 2048, SD_ERR_INVALID_ARG, "tracking supports at most 2048 keypoints per frame");
  SD_HIP_CHECK(hipSetDevice(cur->device));
  sd_track* h = new sd_track();
  h->cur = cur;
  h->ref = ref;
  h->max_points = max_points;
  h->max_batch = max_batch;
  h->kp_cap = nsel;
  h->device = cur->device;
  h->rand_per_frame = 4 * pnp_max_iterations;
  h->rand_len.assign((size_t)max_batch, 0);
  const size_t B = max_batch, M = max_points, K = nsel;
  TrackBuffers& tb = h->tb;
  tb.max_points = max_points;
  tb.kp_cap = nsel;
  tb.cur_bcast = -1;
  int rc = SD_OK;
  auto A = [&](int r) { if (rc == SD_OK) rc = r; };
  A(dalloc(h, &tb.valid, B * M));
  A(dalloc(h, &tb.Xw, B * M * 3));
  A(dalloc(h, &tb.mp_desc, B * M * 32));
  A(dalloc(h, &tb.octave, B * M));
  A(dalloc(h, &tb.angle, B * M));
  A(dalloc(h, &tb.obs, B * M));
  A(dalloc(h, &tb.n_last, B));
  A(dalloc(h, &tb.Tref, B * 16));
  A(dalloc(h, &tb.Tprior, B * 16));
  A(dalloc(h, &tb.Tcur, B * 16));
  A(dalloc(h, &tb.al_ok, B));
  A(dalloc(h, &tb.al_err, B));
  A(dalloc(h, &tb.al_chi2, B));
  A(dalloc(h, &tb.al_iters, B * 16));
  A(dalloc(h, &tb.cur_match, B * K));
  A(dalloc(h, &tb.n_matches, B));
  A(dalloc(h, &tb.mt_list, B * 8192));
  A(dalloc(h, &tb.mt_pt, B * M));
  A(dalloc(h, &tb.mt_key, B * 2048));
  A(dalloc(h, &tb.mt_cstart, B * (64 * 48 + 4)));
  A(dalloc(h, &tb.retry_list, B + 1));
  A(dalloc(h, &tb.uright, B * K));
  A(dalloc(h, &tb.depth, B * K));
  A(dalloc(h, &tb.rand_stream, B * (size_t)h->rand_per_frame));
  A(dalloc(h, &tb.pnp_T, B * 16));
  A(dalloc(h, &tb.pnp_inliers, B * K));
  A(dalloc(h, &tb.pnp_info, B * 8));
  A(dalloc(h, &tb.pnp_scratch, B * K * 5));
  A(dalloc(h, &tb.pnp_pts, B * K * 6));
  A(dalloc(h, &tb.pnp_kpidx, B * K));
  A(dalloc(h, &tb.pnp_state, B * 4));
  A(dalloc(h, &tb.pnp_best_mask, B * 32));
  A(dalloc(h, &tb.pnp_best_T, B * 12));
  A(dalloc(h, &tb.sp_valid1, B * K));
  A(dalloc(h, &tb.sp_valid2, B * K));
  A(dalloc(h, &tb.sp_match, B * K));
  A(dalloc(h, &tb.sp_n, B));
  A(dalloc(h, &tb.lm_cand, B * M));
  A(dalloc(h, &tb.lm_Xw, B * M * 3));
  A(dalloc(h, &tb.lm_normal, B * M * 3));
  A(dalloc(h, &tb.lm_min, B * M));
  A(dalloc(h, &tb.lm_max, B * M));
  A(dalloc(h, &tb.lm_mfmax, B * M));
  A(dalloc(h, &tb.lm_desc, B * M * 32));
  A(dalloc(h, &tb.lm_obs, B * M));
  A(dalloc(h, &tb.lm_n, B));
  A(dalloc(h, &tb.lm_kclaim, B * K));
  A(dalloc(h, &tb.lm_inview, B * M));
  A(dalloc(h, &tb.lm_proj, B * M * 3));
  A(dalloc(h, &tb.lm_level, B * M));
  A(dalloc(h, &tb.lm_cos, B * M));
  A(dalloc(h, &tb.lm_match, B * K));
  A(dalloc(h, &tb.lm_nmatch, B));
  A(dalloc(h, &tb.po_T, B * 16));
  A(dalloc(h, &tb.po_outlier, B * K));
  A(dalloc(h, &tb.po_info, B * 8));
  A(dalloc(h, &tb.tw_info, B * 4));
  A(dalloc(h, &tb.un_match, B * K));
  A(dalloc(h, &tb.tl_info, B * 4));
  A(dalloc(h, &h->d_inv_sigma2, (size_t)cur->nlevels));
  A(dalloc(h, &h->d_scale_thr, (size_t)SD_MAX_LEVELS));
  A(dalloc(h, &h->d_sf, (size_t)cur->nlevels));
  A(dalloc(h, &h->d_inv_sf, (size_t)cur->nlevels));
  A(dalloc(h, &h->d_sigma2, (size_t)cur->nlevels));
  if (rc == SD_OK) {
    std::vector<float> neg((size_t)B * K, -1.f);
    hipError_t e0 = hipMemcpy(tb.uright, neg.data(), neg.size() * 4, hipMemcpyHostToDevice);
    if (e0 == hipSuccess) e0 = hipMemcpy(tb.depth, neg.data(), neg.size() * 4, hipMemcpyHostToDevice);
    // "no map point" until a search has run (a TrackLocalMap / PoseOptimization called first must not see index 0 everywhere)
    if (e0 == hipSuccess) e0 = hipMemset(tb.cur_match, 0xFF, (size_t)B * K * 4);
    if (e0 == hipSuccess) e0 = hipMemset(tb.lm_match, 0xFF, (size_t)B * K * 4);
    if (e0 == hipSuccess) e0 = hipMemset(tb.un_match, 0xFF, (size_t)B * K * 4);
    if (e0 != hipSuccess) { set_error(std::string("sd_track_create: ") + hipGetErrorString(e0)); rc = SD_ERR_HIP; }
  }
  if (rc == SD_OK) {
    hipError_t e = hipMemcpy(h->d_sf, cur->hp.sf.data(), cur->nlevels * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_inv_sf, cur->hp.inv_sf.data(), cur->nlevels * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_sigma2, cur->hp.sigma2.data(), cur->nlevels * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->d_inv_sigma2, cur->hp.inv_sigma2.data(), cur->nlevels * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      // MapPoint::PredictScale (src/MapPoint.cc:371-385): nScale = ceil(log(ratio) / mfLogScaleFactor), float overloads,
      // mfLogScaleFactor = log(mfScaleFactor) (src/Frame.cc:80).  thr[n] = smallest float ratio that reaches level n,
      // by bisection over the float bit patterns with THIS host's libm (the function is monotone).
      float thr[SD_MAX_LEVELS];
      const float Lsf = logf(cur->scaleFactor);
      thr[0] = 0.f;
      for (int n = 1; n < SD_MAX_LEVELS; n++) {
        uint32_t lo = 1, hi = 0x7f800000u;   // hi = +inf: never reached
        while (lo < hi) {
          const uint32_t mid = lo + (hi - lo) / 2;
          float r;
          memcpy(&r, &mid, 4);
          const int ns = (int)ceilf(logf(r) / Lsf);
          if (ns >= n) hi = mid;
          else lo = mid + 1;
        }
        memcpy(&thr[n], &lo, 4);
      }
      e = hipMemcpy(h->d_scale_thr, thr, sizeof(thr), hipMemcpyHostToDevice);
    }
    for (int r = 0; r < sd_track::kRing && e == hipSuccess; r++)
      for (int i = 0; i < 6 && e == hipSuccess; i++) e = hipEventCreate(&h->ev[r][i]);
    if (e == hipSuccess) {
      // Priority of the tracking stream.  Round 1 (extraction kernels at 6-7 waves per SIMD): lowest was best (122.3 k vs
      // 119.2 k frames/s at highest) -- the few, long-running, latency-bound tracking workgroups filled what the extraction
      // left free.  Round 2 (FAST / select at 8 waves per SIMD leave nothing free): the chain of tracking kernels starves at
      // the lowest priority and becomes the longest path of the step when it is long (TrackWithMotionModel + TrackLocalMap:
      // low 107.3 k, normal 108.3 k, HIGH 113.3 k frames/s); with the PnP step all three are within 0.5 % (145.7 / 145.4 /
      // 146.1 k).  Default: highest.
      int lo = 0, hi = 0;
      e = hipDeviceGetStreamPriorityRange(&lo, &hi);
      const int pe = opt(OPT_TRACK_PRIORITY);   // option "track.stream_priority": 0 low | 1 normal | 2 high (default)
      const int prio = pe == 0 ? lo : (pe == 1 ? (lo + hi) / 2 : hi);
      if (e == hipSuccess) e = hipStreamCreateWithPriority(&h->pnp_stream, hipStreamNonBlocking, prio);
    }
    if (e != hipSuccess) {
      set_error(std::string("sd_track_create: ") + hipGetErrorString(e));
      rc = SD_ERR_HIP;
    }
    if (rc == SD_OK) rc = orb_enable_double_buffer(cur);
  }
  if (rc != SD_OK) {
    sd_track_destroy(h);
    return rc;
  }
  *out = h;
  return SD_OK;
}

void sd_track_destroy(sd_track* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->pnp_stream) (void)hipStreamSynchronize(h->pnp_stream);
  if (h->cur && h->cur->stream) (void)hipStreamSynchronize(h->cur->stream);
  if (h->pnp_stream) (void)hipStreamDestroy(h->pnp_stream);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->ev_fence) (void)hipEventDestroy(h->ev_fence);
  for (int r = 0; r < sd_track::kRing; r++)
    for (int i = 0; i < 6; i++)
      if (h->ev[r][i]) (void)hipEventDestroy(h->ev[r][i]);
  delete h;
}

int sd_track_set_camera(sd_track* h, float fx, float fy, float cx, float cy, float bf, float min_x, float max_x, float min_y,
                        float max_y) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  SD_REQUIRE(fx > 0 && fy > 0 && max_x > min_x && max_y > min_y, SD_ERR_INVALID_ARG, "bad camera parameters");
  TrackCam& c = h->cam;
  c.ffx = fx; c.ffy = fy; c.fcx = cx; c.fcy = cy;
  c.fx = fx; c.fy = fy; c.cx = cx; c.cy = cy;
  c.min_x = min_x; c.max_x = max_x; c.min_y = min_y; c.max_y = max_y;
  c.bf = bf;
  c.mb = bf / fx;
  h->have_cam = true;
  return SD_OK;
}

#define TRACK_RANGE(h, frame0, n)                                                                        \
  SD_REQUIRE((h), SD_ERR_INVALID_ARG, "handle is NULL");                                                 \
  SD_REQUIRE((frame0) >= 0 && (n) >= 1 && (frame0) + (n) <= (h)->max_batch, SD_ERR_CAPACITY, "frame range exceeds max_batch"); \
  SD_HIP_CHECK(hipSetDevice((h)->device));                                                               \
  SD_HIP_CHECK(hipStreamSynchronize((h)->pnp_stream))

int sd_track_set_last(sd_track* h, int frame0, int n_frames, const int32_t* n_last, const uint8_t* valid, const double* Xw,
                      const uint8_t* desc, const int32_t* octave, const float* angle, const int32_t* obs) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(n_last && valid && Xw && desc && octave && angle && obs, SD_ERR_INVALID_ARG, "NULL argument");
  h->have_pnp = false;   // the solvers' 3-D points are being replaced
  const size_t M = h->max_points, o = (size_t)frame0;
  for (int f = 0; f < n_frames; f++) SD_REQUIRE(n_last[f] >= 0 && n_last[f] <= h->max_points, SD_ERR_CAPACITY, "n_last exceeds max_points");
  // the octave indexes mvScaleFactors / mvLevelSigma2 on the device (search radius, level window)
  for (int f = 0; f < n_frames; f++)
    for (int i = 0; i < n_last[f]; i++)
      SD_REQUIRE(!valid[(size_t)f * M + i] || (octave[(size_t)f * M + i] >= 0 && octave[(size_t)f * M + i] < h->cur->nlevels),
                 SD_ERR_INVALID_ARG, "octave of a valid last-frame point outside [0, nlevels)");
  hipStream_t s = h->cur->stream;
  const TrackBuffers& tb = h->tb;
  SD_HIP_CHECK(hipMemcpyAsync(tb.n_last + o, n_last, (size_t)n_frames * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.valid + o * M, valid, n_frames * M, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.Xw + o * M * 3, Xw, n_frames * M * 3 * 8, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.mp_desc + o * M * 32, desc, n_frames * M * 32, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.octave + o * M, octave, n_frames * M * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.angle + o * M, angle, n_frames * M * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.obs + o * M, obs, n_frames * M * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

int sd_track_set_poses(sd_track* h, int frame0, int n_frames, const double* Tref_cm, const double* Tcur_cm) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(Tref_cm && Tcur_cm, SD_ERR_INVALID_ARG, "NULL argument");
  hipStream_t s = h->cur->stream;
  SD_HIP_CHECK(hipMemcpyAsync(h->tb.Tref + (size_t)frame0 * 16, Tref_cm, (size_t)n_frames * 128, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(h->tb.Tprior + (size_t)frame0 * 16, Tcur_cm, (size_t)n_frames * 128, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(h->tb.Tcur + (size_t)frame0 * 16, Tcur_cm, (size_t)n_frames * 128, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

int sd_track_set_rand(sd_track* h, int frame0, int n_frames, const int32_t* rand_values, int per_frame) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(rand_values && per_frame >= 1 && per_frame <= h->rand_per_frame, SD_ERR_INVALID_ARG, "bad rand stream");
  h->have_pnp = false;   // a resumed iterate() continues at a position of the OLD stream
  SD_HIP_CHECK(hipMemcpy2DAsync(h->tb.rand_stream + (size_t)frame0 * h->rand_per_frame, (size_t)h->rand_per_frame * 4, rand_values,
                                (size_t)per_frame * 4, (size_t)per_frame * 4, n_frames, hipMemcpyHostToDevice, h->cur->stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->cur->stream));
  for (int f = 0; f < n_frames; f++) h->rand_len[(size_t)frame0 + f] = per_frame;
  return SD_OK;
}

// Order the tracking stream behind the extractions it consumes ...
static int wait_inputs(sd_track* h, bool need_ref, bool pyramid_only = false) {
  // ImageAlign reads pyramids only: it may start as soon as the current batch's pyramid exists, beside FAST / selection /
  // descriptors of the same batch (option "track.align_start" = 0 restores the wait for the whole extraction)
  const int align_start = opt(OPT_ALIGN_START);
  const bool early = align_start != 0;
  // ... but not beside FAST: k_align holds 30 KB of LDS per frame (4 frames per CU), FAST wants 24-39 KB per workgroup, while
  // selection + descriptors, which follow FAST, use next to none.  Waiting for the end of the batch's FAST launches instead of
  // its pyramid: full step 175.2 -> 178.8 k frames/s (three alternating runs; k_align 1.38 -> 0.84 ms in the pipeline).
  // "track.align_start" = 1: wait for the pyramid only; 2 (default): for the FAST launches.
  const bool after_fast = align_start == 2;
  if (h->cur->extract_recorded) {
    hipEvent_t ev = h->cur->ev_extract_done;
    if (pyramid_only && early && h->cur->pyr_event_live) ev = after_fast ? h->cur->ev_fast_done : h->cur->ev_pyr_done;
    SD_HIP_CHECK(hipStreamWaitEvent(h->pnp_stream, ev, 0));
  }
  if (need_ref && h->ref->extract_recorded) SD_HIP_CHECK(hipStreamWaitEvent(h->pnp_stream, h->ref->ev_extract_done, 0));
  return SD_OK;
}
// ... and tell the extractors which output sets the kernel just queued is reading.
static int mark_reads(sd_track* h, bool used_ref) {
  sd_orb* c = h->cur;
  SD_HIP_CHECK(hipEventRecord(c->ev_set_free[c->set], h->pnp_stream));
  c->set_busy[c->set] = true;
  if (used_ref && h->ref != h->cur) {
    sd_orb* r = h->ref;
    SD_HIP_CHECK(hipEventRecord(r->ev_set_free[r->set], h->pnp_stream));
    r->set_busy[r->set] = true;
  }
  return SD_OK;
}

// per_cur_frame: the call walks the frames of `cur` themselves (not tracker slots), so the broadcast does not apply
static int check_ready(sd_track* h, int n_frames, bool per_cur_frame = false) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  SD_REQUIRE(h->have_cam, SD_ERR_INVALID_ARG, "sd_track_set_camera has not been called");
  SD_REQUIRE(n_frames >= 1 && n_frames <= h->max_batch, SD_ERR_CAPACITY, "n_frames exceeds max_batch");
  const int need = (h->tb.cur_bcast >= 0 && !per_cur_frame) ? h->tb.cur_bcast + 1 : n_frames;
  SD_REQUIRE(h->cur->have_geom && h->cur->last_frames >= need, SD_ERR_INVALID_ARG, "current frames have not been extracted");
  SD_HIP_CHECK(hipSetDevice(h->device));
  return SD_OK;
}

// One current frame against n keyframes (Tracking::Relocalization, LoopClosing::DetectLoop): slot f of the tracker
// (its map points, Tref/Tprior, the ref extractor's frame f) pairs with frame `cur_frame` of the cur extractor and with
// that frame's mvuRight row.  -1 restores slot f <-> current frame f.
int sd_track_set_current_broadcast(sd_track* h, int cur_frame) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  SD_REQUIRE(cur_frame >= -1 && cur_frame < h->cur->max_batch && cur_frame < h->max_batch, SD_ERR_INVALID_ARG,
             "cur_frame outside the cur extractor / tracker batch");
  h->tb.cur_bcast = cur_frame;   // TrackBuffers travels by value with every launch: queued kernels keep the old setting
  return SD_OK;
}

int sd_track_align(sd_track* h, int n_frames, int mode) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  SD_REQUIRE(mode >= 0 && mode <= 3, SD_ERR_INVALID_ARG, "bad mode");
  SD_REQUIRE(h->ref->have_geom && h->ref->last_frames >= n_frames && h->ref->cur_w == h->cur->cur_w && h->ref->cur_h == h->cur->cur_h,
             SD_ERR_INVALID_ARG, "reference frames not extracted or of different size");
  hipStream_t s = h->pnp_stream;
  rc = wait_inputs(h, true, true);
  if (rc != SD_OK) return rc;
  hipEvent_t* ev = h->ev[h->ev_calls[0] % sd_track::kRing];
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev[0], s));
  rc = launch_align(h->cur, h->ref, h->tb, h->cam, h->d_inv_sf, h->d_sf, n_frames, mode, s);
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev[1], s)); h->ev_calls[0]++; }
  if (rc == SD_OK) rc = mark_reads(h, true);
  return rc;
}

int sd_track_match(sd_track* h, int n_frames, float th, int mono, int check_ori) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  hipStream_t s = h->pnp_stream;
  h->have_pnp = false;   // mvpMapPoints is rewritten: solvers built on the old vector cannot be continued
  rc = wait_inputs(h, false);
  if (rc != SD_OK) return rc;
  hipEvent_t* ev = h->ev[h->ev_calls[1] % sd_track::kRing];
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev[2], s));
  rc = launch_match(h->cur, h->tb, h->cam, h->d_sf, n_frames, th, mono, check_ori, s);
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev[3], s)); h->ev_calls[1]++; }
  if (rc == SD_OK) rc = mark_reads(h, false);
  return rc;
}

// TrackLocalMap's search.  Local map points of every frame, flattened in mvpLocalMapPoints order.
int sd_track_set_local(sd_track* h, int frame0, int n_frames, const int32_t* n_local, const uint8_t* cand, const double* Xw,
                       const double* normal, const float* min_dist, const float* max_dist, const float* mf_max_dist, const uint8_t* desc,
                       const int32_t* obs, const uint8_t* kp_claimed) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(n_local && cand && Xw && normal && min_dist && max_dist && mf_max_dist && desc && obs, SD_ERR_INVALID_ARG, "NULL argument");
  const size_t M = h->max_points, o = (size_t)frame0, K = h->kp_cap;
  for (int f = 0; f < n_frames; f++) SD_REQUIRE(n_local[f] >= 0 && n_local[f] <= h->max_points, SD_ERR_CAPACITY, "n_local exceeds max_points");
  hipStream_t s = h->cur->stream;
  const TrackBuffers& tb = h->tb;
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_n + o, n_local, (size_t)n_frames * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_cand + o * M, cand, n_frames * M, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_Xw + o * M * 3, Xw, n_frames * M * 3 * 8, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_normal + o * M * 3, normal, n_frames * M * 3 * 8, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_min + o * M, min_dist, n_frames * M * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_max + o * M, max_dist, n_frames * M * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_mfmax + o * M, mf_max_dist, n_frames * M * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_desc + o * M * 32, desc, n_frames * M * 32, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_obs + o * M, obs, n_frames * M * 4, hipMemcpyHostToDevice, s));
  if (kp_claimed) SD_HIP_CHECK(hipMemcpyAsync(tb.lm_kclaim + o * K, kp_claimed, n_frames * K, hipMemcpyHostToDevice, s));
  else SD_HIP_CHECK(hipMemsetAsync(tb.lm_kclaim + o * K, 0, n_frames * K, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

// ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>& vpMapPoints, th) on the caller's OWN isInFrustum results
// (reference src/ORBmatcher.cc:43-119 reads pMP->mbTrackInView, mTrackProjX / Y / XR, mnTrackScaleLevel, mTrackViewCos, which
// Frame::isInFrustum left in the MapPoint): sd_track_set_local_view uploads them, sd_track_match_local_view searches.
// in_view[i] = mbTrackInView && !isBad().
int sd_track_set_local_view(sd_track* h, int frame0, int n_frames, const int32_t* n_local, const uint8_t* in_view, const float* proj3,
                            const int32_t* level, const float* view_cos, const uint8_t* desc, const int32_t* obs, const uint8_t* kp_claimed) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(n_local && in_view && proj3 && level && view_cos && desc && obs, SD_ERR_INVALID_ARG, "NULL argument");
  const size_t M = h->max_points, o = (size_t)frame0, K = h->kp_cap;
  for (int f = 0; f < n_frames; f++) {
    SD_REQUIRE(n_local[f] >= 0 && n_local[f] <= h->max_points, SD_ERR_CAPACITY, "n_local exceeds max_points");
    for (int i = 0; i < n_local[f]; i++)
      SD_REQUIRE(!in_view[(size_t)f * M + i] || (level[(size_t)f * M + i] >= 0 && level[(size_t)f * M + i] < h->cur->nlevels), SD_ERR_INVALID_ARG,
                 "mnTrackScaleLevel of an in-view point outside [0, nlevels)");
  }
  hipStream_t s = h->cur->stream;
  const TrackBuffers& tb = h->tb;
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_n + o, n_local, (size_t)n_frames * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_inview + o * M, in_view, n_frames * M, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_proj + o * M * 3, proj3, n_frames * M * 12, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_level + o * M, level, n_frames * M * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_cos + o * M, view_cos, n_frames * M * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_desc + o * M * 32, desc, n_frames * M * 32, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(tb.lm_obs + o * M, obs, n_frames * M * 4, hipMemcpyHostToDevice, s));
  if (kp_claimed) SD_HIP_CHECK(hipMemcpyAsync(tb.lm_kclaim + o * K, kp_claimed, n_frames * K, hipMemcpyHostToDevice, s));
  else SD_HIP_CHECK(hipMemsetAsync(tb.lm_kclaim + o * K, 0, n_frames * K, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

int sd_track_match_local_view(sd_track* h, int n_frames, float th, float nnratio) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  hipStream_t s = h->pnp_stream;
  rc = wait_inputs(h, false);
  if (rc != SD_OK) return rc;
  rc = launch_match_local(h->cur, h->tb, h->cam, h->d_sf, h->d_scale_thr, h->cur->nlevels, n_frames, th, nnratio, 0.f, s, 0, 1);
  if (rc == SD_OK) rc = mark_reads(h, false);
  return rc;
}

// Frame::isInFrustum for every candidate + ORBmatcher::SearchByProjection(F, vpMapPoints, th) with mfNNratio = nnratio,
// at the frames' current poses (sd_track_set_poses / the ImageAlign result)
int sd_track_match_local(sd_track* h, int n_frames, float th, float nnratio, float viewing_cos_limit) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  hipStream_t s = h->pnp_stream;
  rc = wait_inputs(h, false);
  if (rc != SD_OK) return rc;
  rc = launch_match_local(h->cur, h->tb, h->cam, h->d_sf, h->d_scale_thr, h->cur->nlevels, n_frames, th, nnratio, viewing_cos_limit, s);
  if (rc == SD_OK) rc = mark_reads(h, false);
  return rc;
}

int sd_track_get_local(sd_track* h, int frame0, int n_frames, int32_t* local_match, int cap, int32_t* n_matches, uint8_t* in_view,
                       float* proj3, int32_t* level, float* view_cos) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(!local_match || cap >= h->kp_cap, SD_ERR_CAPACITY, "cap smaller than the keypoint capacity");
  hipStream_t s = h->cur->stream;
  const size_t M = h->max_points, o = frame0, n = n_frames;
  const TrackBuffers& tb = h->tb;
  if (local_match)
    SD_HIP_CHECK(hipMemcpy2DAsync(local_match, (size_t)cap * 4, tb.lm_match + o * h->kp_cap, (size_t)h->kp_cap * 4, (size_t)h->kp_cap * 4, n_frames,
                                  hipMemcpyDeviceToHost, s));
  if (n_matches) SD_HIP_CHECK(hipMemcpyAsync(n_matches, tb.lm_nmatch + o, n * 4, hipMemcpyDeviceToHost, s));
  if (in_view) SD_HIP_CHECK(hipMemcpyAsync(in_view, tb.lm_inview + o * M, n * M, hipMemcpyDeviceToHost, s));
  if (proj3) SD_HIP_CHECK(hipMemcpyAsync(proj3, tb.lm_proj + o * M * 3, n * M * 12, hipMemcpyDeviceToHost, s));
  if (level) SD_HIP_CHECK(hipMemcpyAsync(level, tb.lm_level + o * M, n * M * 4, hipMemcpyDeviceToHost, s));
  if (view_cos) SD_HIP_CHECK(hipMemcpyAsync(view_cos, tb.lm_cos + o * M, n * M * 4, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

// Optimizer::PoseOptimization(&CurrentFrame) at the frames' current poses (tb.Tcur: sd_track_set_poses / ImageAlign result).
// source 0: mvpMapPoints = the frame-to-frame matches (sd_track_match); 1: the local-map matches (sd_track_match_local).
int sd_track_pose_opt(sd_track* h, int n_frames, int source) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  SD_REQUIRE(source >= 0 && source <= 2, SD_ERR_INVALID_ARG, "source must be 0 (frame matches), 1 (local-map matches) or 2 (both, as after SearchLocalPoints)");
  hipStream_t s = h->pnp_stream;
  rc = wait_inputs(h, false);
  if (rc != SD_OK) return rc;
  hipEvent_t* ev = h->ev[h->ev_calls[2] % sd_track::kRing];   // timed in the pose-solve slot, like sd_track_pnp
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev[4], s));
  rc = launch_pose_opt(h->cur, h->tb, h->cam, h->d_inv_sigma2, source, n_frames, s);
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev[5], s)); h->ev_calls[2]++; }
  if (rc == SD_OK) rc = mark_reads(h, false);
  return rc;
}

// Tracking::TrackWithMotionModel (reference src/Tracking.cc:654-718) / TrackReferenceKeyFrame (:583-644) for the batch, as
// one queue of launches with the reference's per-frame decisions taken on the device (no host round trip):
//   ImageAlign::ComputePose(cur, last)  [align_mode 0; 1 = (cur, reference keyframe); -1 = align_image_ off]; a failed
//     alignment leaves the predicted pose
//   SearchByProjection(cur, last, th, mono) with orientation check
//   nmatches < min_matches (20): pose := predicted, SearchByProjection again with 2 * th       (k_match, retry_below)
//   nmatches < min_matches: tracking failed (status 0), PoseOptimization is not run             (k_pose_opt gate)
//   PoseOptimization; outliers lose their map point and flag; nmatchesMap = survivors with Observations() > 0;
//   status 2 iff nmatchesMap >= min_inliers (10), else 1
// The frame's pose (sd_track_get_pose_opt / the tracker's current pose), mvpMapPoints (sd_track_get_matches) and the
// counts (sd_track_get_tracked) are what the reference leaves in mCurrentFrame.  TrackReferenceKeyFrame's second search
// goes to mLastFrame, not to the keyframe (src/Tracking.cc:610): with align_mode 1 it equals the reference when the slot's
// points serve both, otherwise run the stages separately.
int sd_track_with_motion_model(sd_track* h, int n_frames, int align_mode, float th, int mono, int min_matches, int min_inliers) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  SD_REQUIRE(align_mode >= -1 && align_mode <= 1, SD_ERR_INVALID_ARG, "align_mode must be -1 (off), 0 (last frame) or 1 (reference keyframe)");
  SD_REQUIRE(min_matches >= 3 && min_inliers >= 0, SD_ERR_INVALID_ARG, "bad gates (reference: 20 matches, 10 inliers)");
  hipStream_t s = h->pnp_stream;
  const TrackBuffers& tb = h->tb;
  h->have_pnp = false;
  const bool use_ref = align_mode >= 0;
  if (use_ref)
    SD_REQUIRE(h->ref->have_geom && h->ref->last_frames >= n_frames && h->ref->cur_w == h->cur->cur_w && h->ref->cur_h == h->cur->cur_h,
               SD_ERR_INVALID_ARG, "reference frames not extracted or of different size");
  rc = wait_inputs(h, use_ref, use_ref);
  if (rc != SD_OK) return rc;
  hipEvent_t* ev = h->ev[h->ev_calls[0] % sd_track::kRing];
  hipEvent_t* ev1 = h->ev[h->ev_calls[1] % sd_track::kRing];
  hipEvent_t* ev2 = h->ev[h->ev_calls[2] % sd_track::kRing];
  SD_HIP_CHECK(hipMemsetAsync(tb.tw_info, 0, (size_t)n_frames * 16, s));
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev[0], s));
  if (use_ref) rc = launch_align(h->cur, h->ref, tb, h->cam, h->d_inv_sf, h->d_sf, n_frames, align_mode, s);
  else SD_HIP_CHECK(hipMemcpyAsync(tb.Tcur, tb.Tprior, (size_t)n_frames * 128, hipMemcpyDeviceToDevice, s));
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev[1], s)); h->ev_calls[0]++; }
  if (rc != SD_OK) return rc;
  if (use_ref) {
    rc = mark_reads(h, true);
    if (rc != SD_OK) return rc;
    rc = wait_inputs(h, false);   // the matcher needs the keypoints, not only the pyramid
    if (rc != SD_OK) return rc;
  }
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev1[2], s));
  rc = launch_match(h->cur, tb, h->cam, h->d_sf, n_frames, th, mono, 1, s, 0, min_matches);   // (lists the frames that need the retry)
  if (rc == SD_OK) rc = launch_match(h->cur, tb, h->cam, h->d_sf, n_frames, 2.f * th, mono, 1, s, min_matches);
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev1[3], s)); h->ev_calls[1]++; }
  if (rc != SD_OK) return rc;
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev2[4], s));
  rc = launch_pose_opt(h->cur, tb, h->cam, h->d_inv_sigma2, 0, n_frames, s, min_matches, min_inliers);
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev2[5], s)); h->ev_calls[2]++; }
  if (rc == SD_OK) rc = mark_reads(h, false);
  return rc;
}

// Tracking::TrackLocalMap (reference src/Tracking.cc:720-751) for the batch, after sd_track_with_motion_model (or any
// sd_track_match) left the frame-to-frame matches and the pose: SearchLocalPoints (:898-939: isInFrustum + the local-map
// SearchByProjection; a keypoint is closed to the search where its frame match has observations) -> PoseOptimization over
// ALL of mvpMapPoints (frame matches and local matches) -> mnMatchesInliers -> tracked iff >= min_inliers (30).
// The local map (sd_track_set_local) is the caller's UpdateLocalMap(); its `cand` flags carry the "already matched in this
// frame / isBad" skips of :916-921, kp_claimed is ignored here.  th: 1, 3 for RGB-D, 5 after a relocalisation (:929-934).
int sd_track_local_map(sd_track* h, int n_frames, float th, float nnratio, float viewing_cos_limit, int min_inliers) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  SD_REQUIRE(min_inliers >= 0, SD_ERR_INVALID_ARG, "bad min_inliers (reference: 30)");
  hipStream_t s = h->pnp_stream;
  rc = wait_inputs(h, false);
  if (rc != SD_OK) return rc;
  hipEvent_t* ev1 = h->ev[h->ev_calls[1] % sd_track::kRing];
  hipEvent_t* ev2 = h->ev[h->ev_calls[2] % sd_track::kRing];
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev1[2], s));
  rc = launch_match_local(h->cur, h->tb, h->cam, h->d_sf, h->d_scale_thr, h->cur->nlevels, n_frames, th, nnratio, viewing_cos_limit, s, 1);
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev1[3], s)); h->ev_calls[1]++; }
  if (rc != SD_OK) return rc;
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev2[4], s));
  rc = launch_pose_opt(h->cur, h->tb, h->cam, h->d_inv_sigma2, 2, n_frames, s, 0, min_inliers);
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev2[5], s)); h->ev_calls[2]++; }
  if (rc == SD_OK) rc = mark_reads(h, false);
  return rc;
}

// map_match (may be NULL): mvpMapPoints after SearchLocalPoints, -1 | v < max_points: last-frame point v | v >= max_points:
// local map point v - max_points.  info4: status (1 failed, 2 tracked), points in mvpMapPoints, mnMatchesInliers, local matches
int sd_track_get_local_map(sd_track* h, int frame0, int n_frames, int32_t* map_match, int cap, int32_t* info4) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(!map_match || cap >= h->kp_cap, SD_ERR_CAPACITY, "cap smaller than the keypoint capacity");
  hipStream_t s = h->cur->stream;
  if (map_match)
    SD_HIP_CHECK(hipMemcpy2DAsync(map_match, (size_t)cap * 4, h->tb.un_match + (size_t)frame0 * h->kp_cap, (size_t)h->kp_cap * 4,
                                  (size_t)h->kp_cap * 4, n_frames, hipMemcpyDeviceToHost, s));
  if (info4) SD_HIP_CHECK(hipMemcpyAsync(info4, h->tb.tl_info + (size_t)frame0 * 4, (size_t)n_frames * 16, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

// info4 per frame: status (0 few matches, 1 few inliers, 2 tracked), nmatches after the discard, nmatchesMap, retried
int sd_track_get_tracked(sd_track* h, int frame0, int n_frames, int32_t* info4) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(info4, SD_ERR_INVALID_ARG, "NULL argument");
  SD_HIP_CHECK(hipMemcpyAsync(info4, h->tb.tw_info + (size_t)frame0 * 4, (size_t)n_frames * 16, hipMemcpyDeviceToHost, h->cur->stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->cur->stream));
  return SD_OK;
}

// Tracking::Relocalization (reference src/Tracking.cc:1064-1097) over all keyframes at once.  The reference walks the
// keyframes newest first and, for each: sets the frame's pose to the keyframe's, ImageAlign::ComputePose(frame, kf, fast)
// [continue on failure], clears mvpMapPoints, SearchByProjection(frame, kf, th, mono) [continue if < 20],
// PoseOptimization [continue if nGood < 10], else returns true.  Nothing an attempt leaves behind is read by the next one
// (pose and mvpMapPoints are reset, mvbOutlier is rewritten per edge), so the attempts are independent: slot i of the
// tracker holds the i-th keyframe TRIED (caller order = kfs.rbegin() ...), all slots run the three stages against the
// broadcast current frame, and the host returns the first slot that passes the three gates -- the keyframe at which the
// sequential loop would have stopped.  Its pose / matches / outlier flags are slot `winner`'s (sd_track_get_pose_opt,
// sd_track_get_matches).  Caller: sd_track_set_last (keyframe points), sd_track_set_poses(Tref = Tprior = kf pose).
int sd_track_relocalize(sd_track* h, int n_keyframes, int cur_frame, float th, int mono, int min_matches, int min_good,
                        int32_t* winner, int32_t* stage3 /* [n][3] align ok, nmatches, nGood; may be NULL */) {
  SD_REQUIRE(h && winner, SD_ERR_INVALID_ARG, "NULL argument");
  *winner = -1;
  SD_REQUIRE(cur_frame >= 0, SD_ERR_INVALID_ARG, "cur_frame must name a frame of the cur extractor");
  const int saved = h->tb.cur_bcast;
  int rc = sd_track_set_current_broadcast(h, cur_frame);
  if (rc != SD_OK) return rc;
  rc = sd_track_align(h, n_keyframes, 2);
  if (rc == SD_OK) rc = sd_track_match(h, n_keyframes, th, mono, 1);   // ORBmatcher matcher(0.75, true)
  if (rc == SD_OK) rc = sd_track_pose_opt(h, n_keyframes, 0);
  h->tb.cur_bcast = saved;
  if (rc != SD_OK) return rc;
  std::vector<int32_t> ok(n_keyframes), nm(n_keyframes), info((size_t)n_keyframes * 8);
  hipStream_t s = h->pnp_stream;
  SD_HIP_CHECK(hipMemcpyAsync(ok.data(), h->tb.al_ok, (size_t)n_keyframes * 4, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipMemcpyAsync(nm.data(), h->tb.n_matches, (size_t)n_keyframes * 4, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipMemcpyAsync(info.data(), h->tb.po_info, (size_t)n_keyframes * 32, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  for (int i = 0; i < n_keyframes; i++) {
    const int good = info[(size_t)i * 8 + 5];
    if (stage3) { stage3[i * 3] = ok[i]; stage3[i * 3 + 1] = nm[i]; stage3[i * 3 + 2] = good; }
    if (*winner < 0 && ok[i] && nm[i] >= min_matches && good >= min_good) *winner = i;
  }
  return SD_OK;
}

// The candidate search of LoopClosing::DetectLoop (reference src/LoopClosing.cc:115-149): ImageAlign::ComputePose(
// mpCurrentKF, kf) -- level 4 only, identity start, rejected above 0.03 -- against every keyframe of the map.  Slot i holds
// kfs[i] (its GetMapPoints() in the caller's order, its pose as Tref, its pyramid in the ref extractor); the current
// keyframe is frame `cur_frame` of the cur extractor.  excluded[i] != 0 marks the keyframes the loop `continue`s over before
// aligning (the current keyframe itself, connected keyframes).  The reference's loop is replayed on the host over the batched
// results, including its `i++` after a failed alignment (the keyframe after a failure is never looked at), then the
// survivors with error < 1.5 * best are returned -- in slot order (the reference iterates a std::map<KeyFrame*, double>,
// i.e. pointer order; only the SET is defined).
int sd_track_detect_loop(sd_track* h, int n_keyframes, int cur_frame, const uint8_t* excluded, int32_t* candidates, int cap,
                         int32_t* n_candidates, double* best_error, double* errors /* [n], 1e10 where rejected; may be NULL */) {
  SD_REQUIRE(h && candidates && n_candidates && cap >= 0, SD_ERR_INVALID_ARG, "NULL argument");
  *n_candidates = 0;
  SD_REQUIRE(cur_frame >= 0, SD_ERR_INVALID_ARG, "cur_frame must name a frame of the cur extractor");
  const int saved = h->tb.cur_bcast;
  int rc = sd_track_set_current_broadcast(h, cur_frame);
  if (rc != SD_OK) return rc;
  rc = sd_track_align(h, n_keyframes, 3);
  h->tb.cur_bcast = saved;
  if (rc != SD_OK) return rc;
  std::vector<int32_t> ok(n_keyframes);
  std::vector<double> err(n_keyframes);
  hipStream_t s = h->pnp_stream;
  SD_HIP_CHECK(hipMemcpyAsync(ok.data(), h->tb.al_ok, (size_t)n_keyframes * 4, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipMemcpyAsync(err.data(), h->tb.al_err, (size_t)n_keyframes * 8, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  if (errors) std::memcpy(errors, err.data(), (size_t)n_keyframes * 8);
  double best = 1e10;
  std::vector<int32_t> kept;
  for (int i = 0; i < n_keyframes; i++) {
    if (excluded && excluded[i]) continue;
    if (!ok[i]) { i++; continue; }   // "Skip some keyframes"
    kept.push_back(i);
    if (err[i] < best) best = err[i];
  }
  int n = 0;
  for (int32_t i : kept)
    if (err[i] < best * 1.5) {
      SD_REQUIRE(n < cap, SD_ERR_CAPACITY, "candidates array too small");
      candidates[n++] = i;
    }
  *n_candidates = n;
  if (best_error) *best_error = best;
  return SD_OK;
}

int sd_track_get_pose_opt(sd_track* h, int frame0, int n_frames, double* Tcw_cm, uint8_t* outlier, int cap, int32_t* info8) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(!outlier || cap >= h->kp_cap, SD_ERR_CAPACITY, "cap smaller than the keypoint capacity");
  hipStream_t s = h->cur->stream;
  if (Tcw_cm) SD_HIP_CHECK(hipMemcpyAsync(Tcw_cm, h->tb.po_T + (size_t)frame0 * 16, (size_t)n_frames * 128, hipMemcpyDeviceToHost, s));
  if (outlier)
    SD_HIP_CHECK(hipMemcpy2DAsync(outlier, cap, h->tb.po_outlier + (size_t)frame0 * h->kp_cap, h->kp_cap, h->kp_cap, n_frames,
                                  hipMemcpyDeviceToHost, s));
  if (info8) SD_HIP_CHECK(hipMemcpyAsync(info8, h->tb.po_info + (size_t)frame0 * 8, (size_t)n_frames * 32, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

// CurrentFrame.mvuRight supplied by the caller (stereo) -- [n_frames][kp_cap] floats, -1 = none
int sd_track_set_uright(sd_track* h, int frame0, int n_frames, const float* uright, int cap) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(uright && cap >= 1 && cap <= h->kp_cap, SD_ERR_INVALID_ARG, "bad uright array");
  SD_HIP_CHECK(hipMemcpy2DAsync(h->tb.uright + (size_t)frame0 * h->kp_cap, (size_t)h->kp_cap * 4, uright, (size_t)cap * 4, (size_t)cap * 4,
                                n_frames, hipMemcpyHostToDevice, h->cur->stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->cur->stream));
  return SD_OK;
}

// Frame::ComputeStereoFromRGBD on the current frames of the batch: depth images (CV_32F, host memory)
int sd_track_stereo_from_depth(sd_track* h, int n_frames, const float* depth, int w, int hgt, int stride_elems, size_t frame_stride_elems) {
  int rc = check_ready(h, n_frames, true);
  if (rc != SD_OK) return rc;
  SD_REQUIRE(depth && w >= 1 && hgt >= 1 && stride_elems >= w, SD_ERR_INVALID_ARG, "bad depth image");
  SD_HIP_CHECK(hipStreamSynchronize(h->pnp_stream));   // a queued match may still read the stereo arrays
  float* d_depth = nullptr;
  const size_t total = (size_t)n_frames * w * hgt;
  SD_HIP_CHECK(hipMalloc(&d_depth, total * 4));
  hipStream_t s = h->cur->stream;
  hipError_t e = hipSuccess;
  for (int f = 0; f < n_frames && e == hipSuccess; f++)
    e = hipMemcpy2DAsync(d_depth + (size_t)f * w * hgt, (size_t)w * 4, depth + (size_t)f * frame_stride_elems, (size_t)stride_elems * 4,
                         (size_t)w * 4, hgt, hipMemcpyHostToDevice, s);
  if (e == hipSuccess) rc = launch_stereo_from_depth(h->cur, h->tb, h->cam, d_depth, w, hgt, w, (size_t)w * hgt, n_frames, s);
  hipError_t e2 = hipStreamSynchronize(s);
  (void)hipFree(d_depth);
  if (e != hipSuccess || e2 != hipSuccess) {
    set_error(std::string("sd_track_stereo_from_depth: ") + hipGetErrorString(e != hipSuccess ? e : e2));
    return SD_ERR_HIP;
  }
  return rc;
}

int sd_track_get_stereo(sd_track* h, int frame0, int n_frames, float* uright, float* depth, int cap) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(cap >= h->kp_cap, SD_ERR_CAPACITY, "cap smaller than the keypoint capacity");
  hipStream_t s = h->cur->stream;
  if (uright) SD_HIP_CHECK(hipMemcpy2DAsync(uright, (size_t)cap * 4, h->tb.uright + (size_t)frame0 * h->kp_cap, (size_t)h->kp_cap * 4, (size_t)h->kp_cap * 4, n_frames, hipMemcpyDeviceToHost, s));
  if (depth) SD_HIP_CHECK(hipMemcpy2DAsync(depth, (size_t)cap * 4, h->tb.depth + (size_t)frame0 * h->kp_cap, (size_t)h->kp_cap * 4, (size_t)h->kp_cap * 4, n_frames, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

// ORBmatcher::SearchByPoints(currentKF, pKF, matches) (reference src/ORBmatcher.cc:1209-1301) for the batch: slot f matches
// the keypoints of current frame f (or the broadcast frame) that hold a map point against those of frame f of the ref
// extractor.  has_mp_* = "GetMapPointMatches()[i] != NULL && !isBad()", [n_frames][cap], rows shorter than the keypoint
// capacity are padded with 0.
int sd_track_set_point_flags(sd_track* h, int frame0, int n_frames, const uint8_t* has_mp_cur, const uint8_t* has_mp_ref, int cap) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(has_mp_cur && has_mp_ref && cap >= 1 && cap <= h->kp_cap, SD_ERR_INVALID_ARG, "bad flag arrays (cap must be 1..keypoint capacity)");
  hipStream_t s = h->cur->stream;
  const size_t K = h->kp_cap, o = (size_t)frame0 * K;
  SD_HIP_CHECK(hipMemsetAsync(h->tb.sp_valid1 + o, 0, (size_t)n_frames * K, s));
  SD_HIP_CHECK(hipMemsetAsync(h->tb.sp_valid2 + o, 0, (size_t)n_frames * K, s));
  SD_HIP_CHECK(hipMemcpy2DAsync(h->tb.sp_valid1 + o, K, has_mp_cur, (size_t)cap, (size_t)cap, n_frames, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpy2DAsync(h->tb.sp_valid2 + o, K, has_mp_ref, (size_t)cap, (size_t)cap, n_frames, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

int sd_track_search_by_points(sd_track* h, int n_frames, float nnratio, int check_ori) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  SD_REQUIRE(h->ref->have_geom && h->ref->last_frames >= n_frames, SD_ERR_INVALID_ARG, "keyframes of the ref extractor have not been extracted");
  int nsel_ref = 0;
  for (int q : h->ref->hp.quota) nsel_ref += q;
  SD_REQUIRE(nsel_ref == h->kp_cap, SD_ERR_INVALID_ARG, "cur / ref extractors must share the keypoint capacity");
  hipStream_t s = h->pnp_stream;
  rc = wait_inputs(h, true);
  if (rc != SD_OK) return rc;
  hipEvent_t* ev = h->ev[h->ev_calls[1] % sd_track::kRing];   // timed in the matcher's slot
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev[2], s));
  rc = launch_search_points(h->cur, h->ref, h->tb, n_frames, nnratio, check_ori, s);
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev[3], s)); h->ev_calls[1]++; }
  if (rc == SD_OK) rc = mark_reads(h, true);
  return rc;
}

int sd_track_get_point_matches(sd_track* h, int frame0, int n_frames, int32_t* matches12, int cap, int32_t* n_matches) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(!matches12 || cap >= h->kp_cap, SD_ERR_CAPACITY, "cap smaller than the keypoint capacity");
  hipStream_t s = h->cur->stream;
  if (matches12)
    SD_HIP_CHECK(hipMemcpy2DAsync(matches12, (size_t)cap * 4, h->tb.sp_match + (size_t)frame0 * h->kp_cap, (size_t)h->kp_cap * 4,
                                  (size_t)h->kp_cap * 4, n_frames, hipMemcpyDeviceToHost, s));
  if (n_matches) SD_HIP_CHECK(hipMemcpyAsync(n_matches, h->tb.sp_n + frame0, (size_t)n_frames * 4, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

// PnPsolver(F, vpMapPointMatches) accepts ANY match vector (reference src/PnPsolver.cc:71-110), and so does
// Optimizer::PoseOptimization through pFrame->mvpMapPoints: this replaces the slot's CurrentFrame.mvpMapPoints (indices into
// the last-frame arrays, -1 = NULL) with the caller's, as if a search had produced them.
int sd_track_set_matches(sd_track* h, int frame0, int n_frames, const int32_t* cur_match, int cap) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(cur_match && cap >= 1 && cap <= h->kp_cap, SD_ERR_INVALID_ARG, "bad match array (cap must be 1..keypoint capacity)");
  h->have_pnp = false;   // the saved best-inlier mask indexes the old correspondence list
  std::vector<int32_t> full((size_t)n_frames * h->kp_cap, -1), cnt((size_t)n_frames, 0);
  for (int f = 0; f < n_frames; f++)
    for (int i = 0; i < cap; i++) {
      const int32_t m = cur_match[(size_t)f * cap + i];
      SD_REQUIRE(m >= -1 && m < h->max_points, SD_ERR_INVALID_ARG, "match index outside [-1, max_points)");
      full[(size_t)f * h->kp_cap + i] = m;
      cnt[f] += m >= 0;
    }
  hipStream_t s = h->cur->stream;
  SD_HIP_CHECK(hipMemcpyAsync(h->tb.cur_match + (size_t)frame0 * h->kp_cap, full.data(), full.size() * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipMemcpyAsync(h->tb.n_matches + frame0, cnt.data(), cnt.size() * 4, hipMemcpyHostToDevice, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

static int run_pnp(sd_track* h, int n_frames, const PnpParams& pp) {
  hipStream_t s = h->pnp_stream;
  int rc = wait_inputs(h, false);
  if (rc != SD_OK) return rc;
  hipEvent_t* ev = h->ev[h->ev_calls[2] % sd_track::kRing];
  if (h->profiling) SD_HIP_CHECK(hipEventRecord(ev[4], s));
  rc = launch_pnp(h->cur, h->tb, h->cam, h->d_sigma2, pp, n_frames, s);
  if (h->profiling) { SD_HIP_CHECK(hipEventRecord(ev[5], s)); h->ev_calls[2]++; }
  if (rc == SD_OK) rc = mark_reads(h, false);
  return rc;
}

// every slot must have been given the rand() values the call can consume: minSet per RANSAC iteration
static int check_rand(sd_track* h, int n_frames, long long need) {
  SD_REQUIRE(need <= h->rand_per_frame, SD_ERR_CAPACITY, "iterations exceed the handle's pnp_max_iterations (x4 rand values per slot)");
  for (int f = 0; f < n_frames; f++)
    SD_REQUIRE(h->rand_len[f] >= need, SD_ERR_INVALID_ARG,
               "sd_track_set_rand supplied fewer rand() values than the iterations can consume (minSet per iteration)");
  return SD_OK;
}

// PnPsolver ctor + SetRansacParameters + iterate(n_iterations) (reference src/PnPsolver.cc:71-244)
int sd_track_pnp(sd_track* h, int n_frames, double probability, int min_inliers, int max_iterations, int min_set, float epsilon,
                 float th2, int n_iterations) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  // the reference's default is 4 (src/PnPsolver.h:74); other sizes go through the general (one hypothesis at a time) path.
  // Fewer than 4 correspondences leave EPnP's 12 x 12 system rank deficient beyond its 4-D null space: the reference's own
  // hypotheses are then rounding noise of its summation order.  minSet 3 is accepted because its OUTCOME is pinned against
  // the oracle (no hypothesis reaches minInliers: empty Mat + bNoMore, tests/pnp_cases.py "minset3_degenerate"); 1 and 2
  // are pinned by nothing and refused.
  SD_REQUIRE(min_set >= 3 && min_set <= 64, SD_ERR_INVALID_ARG, "minSet must be in [3, 64]");
  SD_REQUIRE(max_iterations >= 1 && n_iterations >= 0 && min_inliers >= 0, SD_ERR_INVALID_ARG, "bad RANSAC parameters");
  const long long upper = std::max(max_iterations, n_iterations);
  rc = check_rand(h, n_frames, (long long)min_set * upper);
  if (rc != SD_OK) return rc;
  PnpParams pp;
  pp.probability = probability;
  pp.min_inliers = min_inliers;
  pp.max_iterations = max_iterations;
  pp.min_set = min_set;
  pp.epsilon = epsilon;
  pp.th2 = th2;
  pp.n_iterations = n_iterations;
  pp.rand_per_frame = h->rand_per_frame;
  pp.resume = 0;
  rc = run_pnp(h, n_frames, pp);
  if (rc == SD_OK) {
    h->have_pnp = true;
    h->pnp_serial = h->cur->extract_serial;
    h->pnp_params = pp;
    h->pnp_frames = n_frames;
    h->pnp_iter_upper = (int)upper;
  }
  return rc;
}

// A further PnPsolver::iterate(n_iterations) on the solvers the last sd_track_pnp constructed: mnIterations, the best
// hypothesis so far and the position in the rand() stream carry over (src/PnPsolver.cc:177: the loop runs while
// mnIterations < mRansacMaxIts OR nCurrentIterations < nIterations, so after the first call every call adds exactly
// n_iterations).  The match vector must not have been changed in between (the reference's solver holds its own copy).
int sd_track_pnp_iterate(sd_track* h, int n_frames, int n_iterations) {
  int rc = check_ready(h, n_frames);
  if (rc != SD_OK) return rc;
  SD_REQUIRE(h->have_pnp && n_frames <= h->pnp_frames, SD_ERR_INVALID_ARG,
             "sd_track_pnp has not constructed solvers for these slots (or their matches / map points / rand stream were replaced since)");
  SD_REQUIRE(h->pnp_serial == h->cur->extract_serial, SD_ERR_INVALID_ARG,
             "the current frames were re-extracted since sd_track_pnp: the solvers' keypoints are gone");
  SD_REQUIRE(n_iterations >= 0, SD_ERR_INVALID_ARG, "bad n_iterations");
  PnpParams pp = h->pnp_params;
  const long long upper = std::max<long long>(pp.max_iterations, (long long)h->pnp_iter_upper + n_iterations);
  rc = check_rand(h, n_frames, (long long)pp.min_set * upper);
  if (rc != SD_OK) return rc;
  pp.n_iterations = n_iterations;
  pp.resume = 1;
  rc = run_pnp(h, n_frames, pp);
  if (rc == SD_OK) h->pnp_iter_upper = (int)upper;
  return rc;
}

int sd_track_get_align(sd_track* h, int frame0, int n_frames, double* Tcur_cm, double* error, int32_t* ok, int32_t* iters,
                       double* chi2) {
  TRACK_RANGE(h, frame0, n_frames);
  hipStream_t s = h->cur->stream;
  const size_t o = frame0, n = n_frames;
  if (Tcur_cm) SD_HIP_CHECK(hipMemcpyAsync(Tcur_cm, h->tb.Tcur + o * 16, n * 128, hipMemcpyDeviceToHost, s));
  if (error) SD_HIP_CHECK(hipMemcpyAsync(error, h->tb.al_err + o, n * 8, hipMemcpyDeviceToHost, s));
  if (ok) SD_HIP_CHECK(hipMemcpyAsync(ok, h->tb.al_ok + o, n * 4, hipMemcpyDeviceToHost, s));
  if (iters) SD_HIP_CHECK(hipMemcpyAsync(iters, h->tb.al_iters + o * 16, n * 64, hipMemcpyDeviceToHost, s));
  if (chi2) SD_HIP_CHECK(hipMemcpyAsync(chi2, h->tb.al_chi2 + o, n * 8, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

int sd_track_get_matches(sd_track* h, int frame0, int n_frames, int32_t* cur_match, int cap, int32_t* n_matches) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(!cur_match || cap >= h->kp_cap, SD_ERR_CAPACITY, "cap smaller than the keypoint capacity");
  hipStream_t s = h->cur->stream;
  if (cur_match)
    SD_HIP_CHECK(hipMemcpy2DAsync(cur_match, (size_t)cap * 4, h->tb.cur_match + (size_t)frame0 * h->kp_cap, (size_t)h->kp_cap * 4,
                                  (size_t)h->kp_cap * 4, n_frames, hipMemcpyDeviceToHost, s));
  if (n_matches) SD_HIP_CHECK(hipMemcpyAsync(n_matches, h->tb.n_matches + frame0, (size_t)n_frames * 4, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

int sd_track_get_pnp(sd_track* h, int frame0, int n_frames, float* Tcw_rowmajor, uint8_t* inliers, int cap, int32_t* info8) {
  TRACK_RANGE(h, frame0, n_frames);
  SD_REQUIRE(!inliers || cap >= h->kp_cap, SD_ERR_CAPACITY, "cap smaller than the keypoint capacity");
  hipStream_t s = h->cur->stream;
  if (Tcw_rowmajor) SD_HIP_CHECK(hipMemcpyAsync(Tcw_rowmajor, h->tb.pnp_T + (size_t)frame0 * 16, (size_t)n_frames * 64, hipMemcpyDeviceToHost, s));
  if (inliers)
    SD_HIP_CHECK(hipMemcpy2DAsync(inliers, cap, h->tb.pnp_inliers + (size_t)frame0 * h->kp_cap, h->kp_cap, h->kp_cap, n_frames,
                                  hipMemcpyDeviceToHost, s));
  if (info8) SD_HIP_CHECK(hipMemcpyAsync(info8, h->tb.pnp_info + (size_t)frame0 * 8, (size_t)n_frames * 32, hipMemcpyDeviceToHost, s));
  SD_HIP_CHECK(hipStreamSynchronize(s));
  return SD_OK;
}

// EPnP (compute_pose, src/PnPsolver.cc:445-492) alone on explicit correspondences -- parity diagnostics
int sd_debug_pnp_prof(unsigned long long* out32, int reset) {
  SD_REQUIRE(out32, SD_ERR_INVALID_ARG, "null output");
  return read_pnp_prof(out32, reset);
}

int sd_debug_sel_prof(unsigned long long* out64, int reset) {
  SD_REQUIRE(out64, SD_ERR_INVALID_ARG, "null output");
  return read_sel_prof(out64, reset);
}

int sd_debug_align_prof(unsigned long long* out16, int reset) {
  SD_REQUIRE(out16, SD_ERR_INVALID_ARG, "null output");
  return read_align_prof(out16, reset);
}

int sd_debug_epnp(int n, const double* Xw, const double* uv, double fx, double fy, double cx, double cy, double* R9, double* t3,
                  double* reproj_err) {
  SD_REQUIRE(n >= 4 && Xw && uv && R9 && t3, SD_ERR_INVALID_ARG, "bad arguments");
  return run_epnp_debug(n, Xw, uv, fx, fy, cx, cy, R9, t3, reproj_err);
}

// diagnostics: raw copy of an internal per-frame buffer (0: pnp correspondences f32[kp_cap*6], 1: pnp keypoint indices u16[kp_cap])
int sd_track_debug_read(sd_track* h, int which, int frame, void* out, size_t bytes) {
  SD_REQUIRE(h && out && frame >= 0 && frame < h->max_batch, SD_ERR_INVALID_ARG, "bad arguments");
  SD_HIP_CHECK(hipSetDevice(h->device));
  SD_HIP_CHECK(hipStreamSynchronize(h->pnp_stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->cur->stream));
  const void* src = which == 0 ? (const void*)(h->tb.pnp_pts + (size_t)frame * h->kp_cap * 6)
                               : (const void*)(h->tb.pnp_kpidx + (size_t)frame * h->kp_cap);
  SD_HIP_CHECK(hipMemcpy(out, src, bytes, hipMemcpyDeviceToHost));
  return SD_OK;
}

// Frame::GetFeaturesInArea (src/Frame.cc:271-321) on the device grid of current frame `frame` (the grid k_match builds):
// indices in the reference's order; grid_counts (may be NULL): mGrid[x][y].size() as [64][48] ints.
int sd_track_debug_features_in_area(sd_track* h, int frame, float x, float y, float r, int min_level, int max_level, int32_t* indices,
                                    int cap, int32_t* n_out, int32_t* grid_counts) {
  int rc = check_ready(h, frame + 1, true);
  if (rc != SD_OK) return rc;
  SD_REQUIRE(frame >= 0 && indices && n_out && cap >= 0, SD_ERR_INVALID_ARG, "bad arguments");
  hipStream_t s = h->pnp_stream;
  rc = wait_inputs(h, false);
  if (rc != SD_OK) return rc;
  int32_t* d = nullptr;
  const size_t n_ints = (size_t)h->kp_cap + 1 + 64 * 48;
  SD_HIP_CHECK(hipMalloc(&d, n_ints * 4));
  rc = launch_features_in_area(h->cur, h->tb, h->cam, frame, x, y, r, min_level, max_level, d, h->kp_cap, d + h->kp_cap, d + h->kp_cap + 1, s);
  std::vector<int32_t> host(n_ints);
  hipError_t e = hipMemcpyAsync(host.data(), d, n_ints * 4, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  (void)hipFree(d);
  if (rc != SD_OK) return rc;
  if (e != hipSuccess) { set_error(std::string("sd_track_debug_features_in_area: ") + hipGetErrorString(e)); return SD_ERR_HIP; }
  const int n = host[h->kp_cap];
  *n_out = n;
  SD_REQUIRE(n <= cap, SD_ERR_CAPACITY, "indices array too small");
  std::memcpy(indices, host.data(), (size_t)n * 4);
  if (grid_counts) std::memcpy(grid_counts, host.data() + h->kp_cap + 1, 64 * 48 * 4);
  return SD_OK;
}

// The batch's result records into a caller-owned DEVICE buffer (n_frames x 20 doubles), queued on the tracking stream
// behind the stages that produce them: no host round trip between the last tracking kernel and the collective that
// gathers the records.
int sd_track_pack_records(sd_track* h, int n_frames, int source, void* d_records) {
  SD_REQUIRE(h && d_records, SD_ERR_INVALID_ARG, "NULL argument");
  SD_REQUIRE(n_frames >= 1 && n_frames <= h->max_batch && source >= 0 && source <= 4, SD_ERR_INVALID_ARG, "bad n_frames / source");
  SD_HIP_CHECK(hipSetDevice(h->device));
  hipLaunchKernelGGL(k_pack_records, dim3((n_frames + 255) / 256), dim3(256), 0, h->pnp_stream, h->tb, source, n_frames, (double*)d_records);
  SD_HIP_CHECK(hipGetLastError());
  return SD_OK;
}

// Ordering against a caller's HIP stream (the stream its RCCL collectives run on).  direction 0: `hip_stream` waits for
// everything queued on the tracking stream so far (collective after the records are packed); 1: the tracking stream waits
// for everything queued on `hip_stream` so far (the next pack must not overwrite a buffer a collective is still reading).
int sd_track_stream_fence(sd_track* h, void* hip_stream, int direction) {
  SD_REQUIRE(h && (direction == 0 || direction == 1), SD_ERR_INVALID_ARG, "bad arguments");
  SD_HIP_CHECK(hipSetDevice(h->device));
  hipStream_t ext = (hipStream_t)hip_stream;   // NULL = the legacy default stream
  if (!h->ev_fence) SD_HIP_CHECK(hipEventCreateWithFlags(&h->ev_fence, hipEventDisableTiming));
  if (direction == 0) {
    SD_HIP_CHECK(hipEventRecord(h->ev_fence, h->pnp_stream));
    SD_HIP_CHECK(hipStreamWaitEvent(ext, h->ev_fence, 0));
  } else {
    SD_HIP_CHECK(hipEventRecord(h->ev_fence, ext));
    SD_HIP_CHECK(hipStreamWaitEvent(h->pnp_stream, h->ev_fence, 0));
  }
  return SD_OK;
}

int sd_track_set_profiling(sd_track* h, int on) {
  SD_REQUIRE(h, SD_ERR_INVALID_ARG, "handle is NULL");
  h->profiling = on != 0;
  h->ev_calls[0] = h->ev_calls[1] = h->ev_calls[2] = 0;
  return SD_OK;
}

// mean ms of the align / match / pnp launches since profiling was switched on
int sd_track_stage_ms(sd_track* h, float* ms_out, int cap) {
  SD_REQUIRE(h && ms_out && cap >= 3, SD_ERR_INVALID_ARG, "bad arguments");
  SD_REQUIRE(h->profiling, SD_ERR_INVALID_ARG, "profiling is off");
  SD_HIP_CHECK(hipSetDevice(h->device));
  SD_HIP_CHECK(hipStreamSynchronize(h->pnp_stream));
  SD_HIP_CHECK(hipStreamSynchronize(h->cur->stream));
  for (int k = 0; k < 3; k++) {
    ms_out[k] = 0;
    const int n = std::min(h->ev_calls[k], (int)sd_track::kRing);
    for (int r = 0; r < n; r++) {
      float ms = 0;
      const int slot = (h->ev_calls[k] - 1 - r) % sd_track::kRing;
      SD_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[slot][2 * k], h->ev[slot][2 * k + 1]));
      ms_out[k] += ms / n;
    }
  }
  return SD_OK;
}

}  // extern "C"
