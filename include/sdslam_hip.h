/* sdslam_hip.h -- C ABI of libsdslam_hip.so: the MI355X (gfx950) implementation of SD-SLAM's
 * per-frame tracking hot path.  Plain pointers and sizes only; every entry point returns an
 * int32 status (SD_OK = 0) and never throws.  Opaque handles own device memory and one HIP
 * stream; calls on one handle are serialised by the caller, different handles are independent
 * (re-entrancy contract of SURVEY.md §8b: Tracking thread + LoopClosing thread).
 *
 * Each group cites the reference C++ interface it replaces (paths relative to the reference
 * repository pasensio97/SDslam); INTEGRATION.md shows the reference-side binding.
 */
#ifndef SDSLAM_HIP_H_
#define SDSLAM_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_OK 0
#define SD_ERR_INVALID_ARG 1  /* bad pointer/size/shape                                   */
#define SD_ERR_HIP 2          /* a HIP runtime call failed (see sd_last_error)             */
#define SD_ERR_CAPACITY 3     /* caller buffer / handle capacity too small                  */
#define SD_ERR_NO_DEVICE 4    /* no gfx950 device visible                                   */
#define SD_FALSE 100          /* the reference would have returned `false` / an empty Mat   */

const char* sd_last_error(void);      /* thread-local description of the last failure      */
int sd_device_count(void);            /* number of visible HIP devices (0 on a CPU box)    */
const char* sd_version(void);

/* Process-wide options: the library's tuning and test switches.  The library never reads the environment.  Integer values;
 * unknown names and out-of-range values fail with SD_ERR_INVALID_ARG.  "plan" options are read when a handle (re)builds its
 * geometry (first extraction at a new frame size), "create" options when a handle is created, the others at every call.
 *   extract.fast0_from_frames  1     0: FAST of level 0 reads the padded pyramid copy instead of the caller's frames
 *   extract.use_graph          0     1: replay the extraction pipeline as a captured hipGraph (slower on ROCm 7.2; kept for tests)
 *   extract.select_small_cap   0     >0: cap (entries) of the per-cell selection buffer -- tests force the large-cell paths
 *   extract.select_big_cap     0     >0: cap of the per-level selection buffer -- tests force the serial fallback
 *   extract.fast_merge_from    6     plan: first pyramid level of the merged FAST launch (>= nlevels: one launch per level)
 *   extract.fast_lds_kb        24    plan: LDS budget of a FAST strip
 *   extract.fast_lds_whole_kb  40    plan: LDS budget under which a cell of the unmerged levels is processed as one strip
 *   track.stream_priority      2     create: priority of the tracking stream, 0 lowest / 1 normal / 2 highest
 *   track.align_start          2     ImageAlign of a batch may start 0: after its whole extraction, 1: after its pyramid,
 *                                    2: after its pyramid and FAST launches (beside selection + descriptors)
 *   track.align_min_waves      5     register budget of k_align in waves per SIMD (3, 4, 5)
 *   track.bf_list_k            4     SearchByPoints: keys kept per point, 1..4 -- tests force the whole-row recomputation
 *   track.poseopt_waves        0     k_pose_opt waves per frame: 0 = by batch size (4 up to 256 frames, else 1), 1, 4
 *   track.match_split          1     SearchByProjection(Frame, Frame / KeyFrame) as candidate + one-wave assignment kernels
 *                                    (6 KB of LDS per frame through the serial part); 0: the single 39-KB kernel
 *   extract.fast0_early        1     device-input extractions: FAST of level 0 starts behind the PREVIOUS call's selection (beside its
 *                                    descriptor kernel) instead of behind the whole previous call; 0: as before
 *   extract.pyr_early          0     1: ... and, on an extractor with two output sets (one a tracker is attached to), the resize chain as
 *                                    well, on the auxiliary stream in front of the blur (measured: loses 8 % with the PnP step)
 * Results never depend on an option (each setting is covered by a parity test); only speed does. */
int sd_set_option(const char* name, int value);
int sd_get_option(const char* name, int* value);
int sd_option_count(void);
const char* sd_option_name(int index);

/* cv::KeyPoint, 28 bytes: {pt.x, pt.y, size, angle, response, octave, class_id} */
typedef struct sd_keypoint {
  float x, y, size, angle, response;
  int32_t octave, class_id;
} sd_keypoint;

/* ------------------------------------------------------------------------------------------
 * ORB extractor -- replaces SD_SLAM::ORBextractor
 *   ctor            src/ORBextractor.h:38,   src/ORBextractor.cc:406-457
 *   operator()      src/ORBextractor.h:45-46, src/ORBextractor.cc:620-678
 *   Get*()          src/ORBextractor.h:48-70
 * One handle serves frames up to max_w x max_h, at most max_batch frames per call.
 * A geometry the kernels do not cover is refused with SD_ERR_INVALID_ARG and a message (sd_last_error, "unsupported geometry: ..."): image
 * sides beyond 4095, a pyramid level that collapses to zero size, a grid cell whose FAST zone is one pixel wide (frames a few dozen
 * pixels wide with hundreds of features per level).
 * ------------------------------------------------------------------------------------------ */
typedef struct sd_orb sd_orb;

int sd_orb_create(int nfeatures, float scale_factor, int nlevels, int th_fast,
                  int max_w, int max_h, int max_batch, int device, sd_orb** out);
void sd_orb_destroy(sd_orb* h);

/* GetLevels / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares (arrays of nlevels floats; any pointer may be NULL). */
int sd_orb_levels(const sd_orb* h);
int sd_orb_scale_tables(const sd_orb* h, float* sf, float* inv_sf, float* sigma2, float* inv_sigma2);
int sd_orb_features_per_level(const sd_orb* h, int32_t* quota);

/* Host-only geometry query (works without a GPU): per level {w, h, quota, levelCols, levelRows,
 * cellW, cellH, nfeaturesCell} (src/ORBextractor.cc:472-488,683) and per cell {level, zone x0,
 * y0, w, h, evaluated} (the FAST detection zone of src/ORBextractor.cc:501-536). */
int sd_orb_plan_info(int nfeatures, float scale_factor, int nlevels, int th_fast, int w, int hgt,
                     int32_t* level_info /* nlevels x 8 */, int32_t* cell_zones /* cap x 6, may be NULL */,
                     int cell_cap, int32_t* n_cells, uint64_t* bytes_per_frame);

/* operator()(image, mask(ignored), keypoints, descriptors, pyramid): one 8-bit grey frame in
 * host memory -> keypoints (cap entries), descriptors (cap x 32 bytes), *n_out.  The image
 * pyramid stays resident on the device (sd_orb_level_*).  Empty image => *n_out = 0, SD_OK
 * (src/ORBextractor.cc:622-623). */
int sd_orb_extract(sd_orb* h, const uint8_t* img, int w, int hgt, int stride,
                   sd_keypoint* kps_out, uint8_t* desc_out, int cap, int* n_out);

/* Batched-frames mode: n_frames independent frames of identical size.
 * Host variant copies in/out; device variant takes a device pointer, launches asynchronously
 * on the handle's stream and leaves the results resident (read them with sd_orb_download or
 * chain into sd_match_ / sd_align_ calls on the same handle). */
int sd_orb_extract_batch(sd_orb* h, const uint8_t* imgs, int n_frames, int w, int hgt, int stride,
                         size_t frame_stride, sd_keypoint* kps_out, uint8_t* desc_out,
                         int cap_per_frame, int32_t* n_out);
int sd_orb_extract_batch_device(sd_orb* h, const void* d_imgs, int n_frames, int w, int hgt,
                                int stride, size_t frame_stride);
int sd_orb_download(sd_orb* h, int frame0, int n_frames, sd_keypoint* kps_out, uint8_t* desc_out,
                    int cap_per_frame, int32_t* n_out);

/* Frame::UndistortKeyPoints (src/Frame.cc:335-366): with k1 != 0 every extraction also produces
 * mvKeysUn = cv::undistortPoints(mvKeys, K, {k1,k2,p1,p2,k3}, R=I, P=K); the tracking stages read
 * the undistorted keypoints.  K is the CV_32F camera matrix Converter::toCvMat builds. */
int sd_orb_set_distortion(sd_orb* h, float fx, float fy, float cx, float cy, float k1, float k2,
                          float p1, float p2, float k3);
int sd_orb_download_undistorted(sd_orb* h, int frame0, int n_frames, sd_keypoint* kps_un_out,
                                int cap_per_frame);

/* std::vector<cv::Mat>& imagePyramid of operator(): level geometry and a host copy of one
 * level of one frame of the last batch (padded != 0: including the 19-px REFLECT_101 border
 * the reference keeps around each level, src/ORBextractor.cc:684-697). */
int sd_orb_level_info(const sd_orb* h, int level, int* w, int* hgt);
int sd_orb_level_copy(sd_orb* h, int frame, int level, int padded, uint8_t* out, int out_stride);

/* Diagnostics used by the parity tests (stage outputs of the last batch). */
int sd_orb_debug_blurred(sd_orb* h, int frame, int level, uint8_t* out, int out_stride);
int sd_orb_debug_cell_counts(sd_orb* h, int frame, int level, int32_t* out, int cap, int* n_cells);
int sd_orb_debug_level_keys(sd_orb* h, int frame, int level, uint32_t* keys_out, int cap, int* n);

/* Stream / timing plumbing (bench + rocprof).  sd_orb_set_stream: run on a caller-owned
 * hipStream_t (NULL restores the handle's own stream).  With profiling on, every extract call
 * brackets each stage with HIP events on the launch stream; sd_orb_stage_ms returns the mean
 * elapsed ms per stage over the calls made since profiling was switched on (last 128 at most;
 * names from sd_orb_stage_name). */
int sd_orb_set_stream(sd_orb* h, void* hip_stream);
/* Ordering against a caller-owned hipStream_t (e.g. the stream that uploads the NEXT batch's frames while this batch is being
 * processed): direction 0 = that stream waits for the extractions queued so far (their input frames may then be overwritten),
 * 1 = the extractions queued from now on wait for everything queued on that stream so far (the upload of their frames). */
int sd_orb_stream_fence(sd_orb* h, void* hip_stream, int direction);
int sd_orb_sync(sd_orb* h);
int sd_orb_set_profiling(sd_orb* h, int on);
int sd_orb_num_stages(void);
const char* sd_orb_stage_name(int stage);
int sd_orb_stage_ms(sd_orb* h, float* ms_out, int cap);
/* algorithmic bytes per frame of each stage for the current geometry */
int sd_orb_stage_bytes(const sd_orb* h, double* bytes_out, int cap);

/* device memory helpers for harnesses that have no HIP binding of their own */
int sd_dev_alloc(size_t bytes, void** out);
/* page-locked host buffers (hipHostMalloc): frames handed to sd_orb_extract / sd_orb_extract_batch from such a buffer
 * are copied at the PCIe rate instead of through the driver's pageable-memory staging */
int sd_host_alloc(size_t bytes, void** out);
int sd_host_free(void* p);
int sd_dev_free(void* p);
int sd_dev_upload(void* dst, const void* src, size_t bytes);
int sd_dev_download(void* dst, const void* src, size_t bytes);

/* ------------------------------------------------------------------------------------------
 * Batched TrackWithMotionModel context -- the per-frame sequence of src/Tracking.cc:654-718
 * over two resident extractor handles: `cur` holds the current frames of the batch, `ref` the
 * last frames (same geometry, frame f of one pairs with frame f of the other).
 *
 *   sd_track_align   ImageAlign::ComputePose            src/ImageAlign.h:36-42, src/ImageAlign.cc:45-232
 *                    mode 0 (Frame,Frame) / 1 (Frame,KeyFrame) / 2 (Frame,KeyFrame,fast) / 3 (KF,KF)
 *   sd_track_match   ORBmatcher::SearchByProjection(Frame&, const Frame&, th, bMono)
 *                                                       src/ORBmatcher.h:46, src/ORBmatcher.cc:946-1075
 *                    (+ Frame::AssignFeaturesToGrid / GetFeaturesInArea, src/Frame.cc:179-192,271-332)
 *   sd_track_pnp     PnPsolver ctor + SetRansacParameters + iterate
 *                                                       src/PnPsolver.h:67-76, src/PnPsolver.cc:71-315
 *
 * Last-frame data (sd_track_set_last) flattens LastFrame.mvpMapPoints: for last-frame keypoint
 * i, valid[i] = (pMP != NULL && !mvbOutlier[i]), Xw = pMP->GetWorldPos(), desc =
 * pMP->GetDescriptor(), octave = mvKeys[i].octave, angle = mvKeysUn[i].angle, obs =
 * pMP->Observations().  Arrays are [n_frames][max_points(...)] in host memory.  Poses are 16
 * doubles column-major (Eigen::Matrix4d::data()).  All launches are asynchronous on the `cur`
 * extractor's stream; the sd_track_get_* calls synchronise.
 * ------------------------------------------------------------------------------------------ */
typedef struct sd_track sd_track;

int sd_track_create(sd_orb* cur, sd_orb* ref, int max_points, int max_batch, int pnp_max_iterations,
                    sd_track** out);
void sd_track_destroy(sd_track* h);
/* Frame statics: fx, fy, cx, cy, mbf, mnMinX, mnMaxX, mnMinY, mnMaxY (src/Frame.cc:158-174) */
int sd_track_set_camera(sd_track* h, float fx, float fy, float cx, float cy, float bf,
                        float min_x, float max_x, float min_y, float max_y);
int sd_track_set_last(sd_track* h, int frame0, int n_frames, const int32_t* n_last,
                      const uint8_t* valid, const double* Xw, const uint8_t* desc,
                      const int32_t* octave, const float* angle, const int32_t* obs);
int sd_track_set_poses(sd_track* h, int frame0, int n_frames, const double* Tref_cm,
                       const double* Tcur_cm);
/* raw rand() values consumed by SD_SLAM::Random, 4 per RANSAC iteration (src/extra/utils.cc:23-26,
 * src/PnPsolver.cc:185-194); rand_values is [n_frames][per_frame] */
int sd_track_set_rand(sd_track* h, int frame0, int n_frames, const int32_t* rand_values, int per_frame);

/* Stereo / RGB-D information of the current frames: either mvuRight directly, or
 * Frame::ComputeStereoFromRGBD (src/Frame.cc:399-417) from CV_32F depth images in host memory
 * (mvDepth, mvuRight = kpU.x - mbf / d); bf comes from sd_track_set_camera. */
int sd_track_set_uright(sd_track* h, int frame0, int n_frames, const float* uright, int cap);
int sd_track_stereo_from_depth(sd_track* h, int n_frames, const float* depth, int w, int hgt,
                               int stride_elems, size_t frame_stride_elems);
int sd_track_get_stereo(sd_track* h, int frame0, int n_frames, float* uright, float* depth, int cap);

/* TrackLocalMap's search (reference src/Tracking.cc:898-939): Frame::isInFrustum (src/Frame.cc:215-269, incl.
 * MapPoint::PredictScale src/MapPoint.cc:371-385) for every local map point with cand != 0, then
 * ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th) (src/ORBmatcher.cc:43-126) with
 * mfNNratio = nnratio, at the frames' current poses.  Per-point arrays are [n_frames][max_points]:
 * min_dist / max_dist = GetMin/MaxDistanceInvariance(), mf_max_dist = mfMaxDistance, normal = GetNormal(),
 * obs = Observations(); kp_claimed [n_frames][kp_cap] (may be NULL) marks keypoints that already hold a map
 * point with Observations() > 0.  Results: local_match[kp] = index of the assigned local point or -1,
 * in_view = mbTrackInView, proj3 = {mTrackProjX, mTrackProjY, mTrackProjXR}, level = mnTrackScaleLevel. */
int sd_track_set_local(sd_track* h, int frame0, int n_frames, const int32_t* n_local, const uint8_t* cand, const double* Xw,
                       const double* normal, const float* min_dist, const float* max_dist, const float* mf_max_dist,
                       const uint8_t* desc, const int32_t* obs, const uint8_t* kp_claimed);
int sd_track_match_local(sd_track* h, int n_frames, float th, float nnratio, float viewing_cos_limit);
/* The same search on the CALLER's isInFrustum results -- what ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
 * itself reads from the map points (src/ORBmatcher.cc:48-60): in_view = mbTrackInView && !isBad(), proj3 = {mTrackProjX,
 * mTrackProjY, mTrackProjXR}, level = mnTrackScaleLevel, view_cos = mTrackViewCos; [n_frames][max_points] arrays.  Results
 * through sd_track_get_local. */
int sd_track_set_local_view(sd_track* h, int frame0, int n_frames, const int32_t* n_local, const uint8_t* in_view, const float* proj3,
                            const int32_t* level, const float* view_cos, const uint8_t* desc, const int32_t* obs, const uint8_t* kp_claimed);
int sd_track_match_local_view(sd_track* h, int n_frames, float th, float nnratio);
int sd_track_get_local(sd_track* h, int frame0, int n_frames, int32_t* local_match, int cap, int32_t* n_matches,
                       uint8_t* in_view, float* proj3, int32_t* level, float* view_cos);

/* Optimizer::PoseOptimization(Frame*) (reference src/Optimizer.cc:221-415: g2o Levenberg, Huber kernel, 4 rounds of
 * 10 iterations with outlier re-classification), the pose solve the reference runs after SearchByProjection
 * (src/Tracking.cc:693) and after the local-map search (:729).  Input pose = the frames' current poses; map points =
 * the matches of sd_track_match (source 0) or sd_track_match_local (source 1); stereo observations where
 * mvuRight >= 0.  Results: optimised Tcw (16 doubles column-major), mvbOutlier flags, info8 = {nInitialCorrespondences,
 * nBad, rounds, g2o iterations, LM trials, return value (nInitial - nBad), 0, 0}. */
int sd_track_pose_opt(sd_track* h, int n_frames, int source);
int sd_track_get_pose_opt(sd_track* h, int frame0, int n_frames, double* Tcw_cm, uint8_t* outlier, int cap, int32_t* info8);

/* One current frame against many keyframes.  After sd_track_set_current_broadcast(h, c) with c >= 0, slot f of every
 * later sd_track_align / _match / _pnp / _pose_opt pairs its own map points, poses and ref-extractor frame f with frame c
 * of the cur extractor (and with row c of mvuRight); -1 restores slot f <-> current frame f.  The cur extractor may then
 * hold fewer frames than the tracker has slots.
 *
 * sd_track_relocalize: Tracking::Relocalization (src/Tracking.cc:1064-1097) with every keyframe attempt of its loop run
 * as one batch slot: ImageAlign::ComputePose(frame, kf, fast) -> SearchByProjection(frame, kf, th, mono) with orientation
 * check -> PoseOptimization.  Slots are in the order the reference tries them (newest keyframe first); *winner = first
 * slot with align ok, nmatches >= min_matches (20) and nGood >= min_good (10), or -1 (= Relocalization returns false).
 * stage3 (may be NULL): [n][3] = {align ok, nmatches, nGood} of every slot.
 *
 * sd_track_detect_loop: the candidate search of LoopClosing::DetectLoop (src/LoopClosing.cc:115-149):
 * ImageAlign::ComputePose(mpCurrentKF, kfs[i]) for all i as one batch, then the reference's loop replayed over the
 * results (skip excluded[i] != 0; a failed alignment also skips slot i+1; best error; keep error < 1.5 * best).
 * candidates: slot indices in ascending order (the reference's std::map<KeyFrame*,...> order is pointer order). */
int sd_track_set_current_broadcast(sd_track* h, int cur_frame);
int sd_track_relocalize(sd_track* h, int n_keyframes, int cur_frame, float th, int mono, int min_matches, int min_good,
                        int32_t* winner, int32_t* stage3);
int sd_track_detect_loop(sd_track* h, int n_keyframes, int cur_frame, const uint8_t* excluded, int32_t* candidates, int cap,
                         int32_t* n_candidates, double* best_error, double* errors);

/* Tracking::TrackWithMotionModel (src/Tracking.cc:654-718) for the batch in one call: ImageAlign (align_mode 0: against the
 * last frame; 1: ComputePose(frame, reference keyframe) as in TrackReferenceKeyFrame :583-644; -1: align_image_ off) ->
 * SearchByProjection(th) -> if nmatches < min_matches: pose := predicted and SearchByProjection(2 th) -> if still
 * < min_matches: failed -> PoseOptimization -> outliers discarded, nmatchesMap counted -> tracked iff nmatchesMap >=
 * min_inliers.  The reference's constants are min_matches = 20, min_inliers = 10.  All per-frame decisions are taken on the
 * device.  Afterwards: sd_track_get_tracked (info4 = status 0 few matches / 1 few inliers / 2 tracked, nmatches, nmatchesMap,
 * retried), sd_track_get_pose_opt (the frame's pose; mvbOutlier all false after the discard), sd_track_get_matches
 * (mvpMapPoints after the discard). */
int sd_track_with_motion_model(sd_track* h, int n_frames, int align_mode, float th, int mono, int min_matches, int min_inliers);
int sd_track_get_tracked(sd_track* h, int frame0, int n_frames, int32_t* info4);

/* Tracking::TrackLocalMap (src/Tracking.cc:720-751) for the batch, on top of the frame-to-frame matches and pose that
 * sd_track_with_motion_model (or sd_track_match) left: SearchLocalPoints (:898-939) over the local map of sd_track_set_local
 * (a keypoint is closed to the search where its frame match has Observations() > 0; kp_claimed of sd_track_set_local is not
 * used) -> PoseOptimization over all of mvpMapPoints -> mnMatchesInliers -> tracked iff >= min_inliers (reference: 30).
 * th = 1 (3 for RGB-D, 5 right after a relocalisation), nnratio = 0.8, viewing_cos_limit = 0.5 in the reference.
 * sd_track_get_local_map: map_match[i] = -1 | v < max_points: last-frame point v | v >= max_points: local point v - max_points;
 * info4 = {status 1 failed / 2 tracked, points in mvpMapPoints, mnMatchesInliers, local matches}.  Pose and mvbOutlier:
 * sd_track_get_pose_opt; isInFrustum outputs: sd_track_get_local.  sd_track_pose_opt(h, n, 2) runs the optimisation alone
 * on the same union. */
int sd_track_local_map(sd_track* h, int n_frames, float th, float nnratio, float viewing_cos_limit, int min_inliers);
int sd_track_get_local_map(sd_track* h, int frame0, int n_frames, int32_t* map_match, int cap, int32_t* info4);

/* ORBmatcher::SearchByPoints(KeyFrame* currentKF, KeyFrame* pKF, vector<MapPoint*>& matches)
 *   src/ORBmatcher.h:56, src/ORBmatcher.cc:1209-1301 (caller: LoopClosing::ComputeSim3, src/LoopClosing.cc:255)
 * Brute-force Hamming matching between the map points of two keyframes, for the whole batch: slot f pairs frame f (or the
 * broadcast frame) of the cur extractor with frame f of the ref extractor.  has_mp_cur / has_mp_ref [n_frames][cap]:
 * GetMapPointMatches()[i] != NULL && !isBad().  nnratio = mfNNratio (0.75 in ComputeSim3), check_ori = mbCheckOrientation.
 * matches12[i] = index of the pKF keypoint whose map point the reference stores in matches[i], or -1; *n_matches = return value. */
int sd_track_set_point_flags(sd_track* h, int frame0, int n_frames, const uint8_t* has_mp_cur, const uint8_t* has_mp_ref, int cap);
int sd_track_search_by_points(sd_track* h, int n_frames, float nnratio, int check_ori);
int sd_track_get_point_matches(sd_track* h, int frame0, int n_frames, int32_t* matches12, int cap, int32_t* n_matches);

int sd_track_align(sd_track* h, int n_frames, int mode);
int sd_track_match(sd_track* h, int n_frames, float th, int mono, int check_ori);
/* sd_track_pnp = PnPsolver(CurrentFrame, CurrentFrame.mvpMapPoints) + SetRansacParameters(...) + iterate(n_iterations)
 * (src/PnPsolver.cc:71-244); min_set 4 is the reference's default (src/PnPsolver.h:74); 3..64 are accepted (1 and 2: SD_ERR_INVALID_ARG --
 * EPnP on fewer than 3 points is pinned by nothing; 3 is pinned only as "no hypothesis is ever accepted").
 * sd_track_pnp_iterate = a further iterate(n_iterations) on those solvers: mnIterations, the best hypothesis so far and the
 * position in the rand() stream carry over (src/PnPsolver.cc:177).  The reference's solver owns copies of its inputs; here they
 * stay in the tracker, so anything that replaces them -- a new extraction on `cur`, sd_track_match / _with_motion_model /
 * _relocalize, sd_track_set_matches / _set_last / _set_rand -- ends the solvers' life: the next sd_track_pnp_iterate fails with
 * SD_ERR_INVALID_ARG until sd_track_pnp constructs new ones.  Each RANSAC iteration consumes min_set values of the
 * stream given to sd_track_set_rand; a call that could run past the supplied values fails with SD_ERR_INVALID_ARG.
 * sd_track_set_matches replaces CurrentFrame.mvpMapPoints of the slots by the caller's vector (indices into the
 * last-frame arrays, -1 = NULL; cap entries per frame, the rest NULL): PnPsolver and Optimizer::PoseOptimization take any
 * match vector, not only the one sd_track_match leaves (src/PnPsolver.cc:71-110, src/Optimizer.cc:240-330). */
int sd_track_pnp(sd_track* h, int n_frames, double probability, int min_inliers, int max_iterations,
                 int min_set, float epsilon, float th2, int n_iterations);
int sd_track_pnp_iterate(sd_track* h, int n_frames, int n_iterations);
int sd_track_set_matches(sd_track* h, int frame0, int n_frames, const int32_t* cur_match, int cap);

/* sd_track_set_poses: LastFrame.GetPose() and the current frame's prior pose (motion-model
 * prediction); sd_track_align reads the prior and leaves the aligned pose for the later stages.
 * ImageAlign results: pose written by CurrentFrame.SetPose (= the prior when ok == 0 or mode 3),
 * GetError(), the bool return, iterations per pyramid level (n x 16) and chi2_ */
int sd_track_get_align(sd_track* h, int frame0, int n_frames, double* Tcur_cm, double* error,
                       int32_t* ok, int32_t* iters, double* chi2);
/* CurrentFrame.mvpMapPoints as indices into the last-frame arrays (-1 = NULL), return value */
int sd_track_get_matches(sd_track* h, int frame0, int n_frames, int32_t* cur_match, int cap,
                         int32_t* n_matches);
/* iterate(): Tcw (4x4 CV_32F row-major; all zeros = empty Mat), vbInliers, and per frame
 * info8 = {returned(0/1), nInliers, bNoMore, iterations, N, minInliers, maxIts, refined} */
int sd_track_get_pnp(sd_track* h, int frame0, int n_frames, float* Tcw_rowmajor, uint8_t* inliers,
                     int cap, int32_t* info8);
/* diagnostics: device EPnP (PnPsolver::compute_pose, src/PnPsolver.cc:445-492) on n explicit
 * correspondences; R9 row-major, returns the mean reprojection error in *reproj_err */
/* Stage cycle counters of k_pnp (only in a library built with -DSD_PNP_PROF; tools/prof_pnp.py) */
int sd_debug_pnp_prof(unsigned long long* out32, int reset);
int sd_debug_align_prof(unsigned long long* out16, int reset); /* k_align phases */
int sd_debug_sel_prof(unsigned long long* out64, int reset);   /* out64[8 * i + 7] = k_fast_cells phase i (cycles), rest 0 */
int sd_debug_epnp(int n, const double* Xw, const double* uv, double fx, double fy, double cx, double cy,
                  double* R9, double* t3, double* reproj_err);
int sd_track_debug_read(sd_track* h, int which, int frame, void* out, size_t bytes);
/* Frame::GetFeaturesInArea(x, y, r, minLevel, maxLevel) (src/Frame.cc:271-321) answered by the device-side bucket grid of
 * current frame `frame` (Frame::AssignFeaturesToGrid, src/Frame.cc:179-192 -- the grid the matchers build): keypoint
 * indices in the reference's vIndices order; grid_counts (may be NULL) = mGrid[x][y].size(), [64][48]. */
int sd_track_debug_features_in_area(sd_track* h, int frame, float x, float y, float r, int min_level, int max_level,
                                    int32_t* indices, int cap, int32_t* n_out, int32_t* grid_counts);
/* Batched-frames mode across GPUs: the fixed-size per-frame result records -- the only data that leaves a GPU.
 * sd_track_pack_records queues, behind the tracking stages, a kernel that writes n_frames x 20 doubles {pose 4x4
 * column-major, ImageAlign ok, nmatches, pose-solver inliers, pose-solver ok} into a caller-owned DEVICE buffer; source =
 * 0 PnPsolver / 1 PoseOptimization / 2 TrackWithMotionModel / 3 TrackLocalMap / 4 ImageAlign only.
 * sd_track_stream_fence orders the tracking stream against a caller's hipStream_t (the stream of its RCCL collective):
 * direction 0 = that stream waits for the tracking stream, 1 = the tracking stream waits for that stream. */
int sd_track_pack_records(sd_track* h, int n_frames, int source, void* d_records);
int sd_track_stream_fence(sd_track* h, void* hip_stream, int direction);
int sd_track_set_profiling(sd_track* h, int on);
int sd_track_stage_ms(sd_track* h, float* ms_out /* [0]=align, [1]=match, [2]=pnp */, int cap);

/* ------------------------------------------------------------------------------------------
 * ORBmatcher::DescriptorDistance -- src/ORBmatcher.h:44, src/ORBmatcher.cc:1459-1473
 * (pure host function; kept in the ABI so callers need no second library)
 * ------------------------------------------------------------------------------------------ */
int sd_hamming(const uint8_t* a32, const uint8_t* b32);

#ifdef __cplusplus
}
#endif
#endif /* SDSLAM_HIP_H_ */
