// Internal structures of the sd_track handle (batched TrackWithMotionModel context).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "orb_internal.h"

namespace sd {

// Device-resident per-frame arrays (frame index f selects a slice of each).
struct TrackBuffers {
  int max_points;        // capacity M of the last-frame arrays
  int kp_cap;            // keypoint capacity of the current frame (extractor's nsel)
  int cur_bcast;         // >= 0: every batch slot pairs with THIS frame of the `cur` extractor (one current frame against
                         // many keyframes: Relocalization, DetectLoop); -1: slot f pairs with current frame f
  // last frame (LastFrame.mvpMapPoints flattened; index i == last-frame keypoint index)
  uint8_t* valid;        // [B][M]   pMP != NULL && !mvbOutlier[i]
  double* Xw;            // [B][M][3] pMP->GetWorldPos()
  uint8_t* mp_desc;      // [B][M][32] pMP->GetDescriptor()
  int32_t* octave;       // [B][M]   LastFrame.mvKeys[i].octave
  float* angle;          // [B][M]   LastFrame.mvKeysUn[i].angle
  int32_t* obs;          // [B][M]   pMP->Observations()
  int32_t* n_last;       // [B]
  // poses, 16 doubles column-major (Eigen::Matrix4d::data())
  double* Tref;          // [B][16]  LastFrame.GetPose()
  double* Tprior;        // [B][16]  CurrentFrame pose before alignment (motion-model prediction)
  double* Tcur;          // [B][16]  CurrentFrame pose after ImageAlign (= Tprior when it returns false)
  // ImageAlign outputs
  int32_t* al_ok;        // [B]
  double* al_err;        // [B]  error_
  double* al_chi2;       // [B]  chi2_
  int32_t* al_iters;     // [B][16] iterations run per pyramid level
  // SearchByProjection outputs
  int32_t* cur_match;    // [B][kp_cap]  index into the last-frame arrays or -1
  int32_t* n_matches;    // [B]
  uint32_t* mt_list;     // [B][8192] split matcher: candidate keys of the frame's points, in point order (track_match.hip)
  uint32_t* mt_pt;       // [B][M]    ... offset << 16 | count of every point's keys
  uint32_t* mt_key;      // [B][2048] ... the frame's sorted grid keys and cell starts (slow path of the assignment loop)
  uint16_t* mt_cstart;   // [B][64*48+4]
  int32_t* retry_list;   // [1 + B]: count, then the frames whose first search found too few matches (TrackWithMotionModel's retry)
  float* uright;         // [B][kp_cap]  CurrentFrame.mvuRight (-1: no stereo/depth information)
  float* depth;          // [B][kp_cap]  CurrentFrame.mvDepth
  // PnP
  int32_t* rand_stream;  // [B][4*pnp_max_its] raw rand() values
  float* pnp_T;          // [B][16] row-major CV_32F 4x4
  uint8_t* pnp_inliers;  // [B][kp_cap]
  int32_t* pnp_info;     // [B][8]: ok, nInliers, noMore, iterations, N, minInliers, maxIts, refined
  float* pnp_scratch;    // [B][kp_cap*5] EPnP refit: the best set's correspondences, compacted (pws f32 x 3 | us f32 x 2)
  float* pnp_pts;        // [B][kp_cap][6] gathered correspondences {u, v, X, Y, Z, maxErr}
  uint16_t* pnp_kpidx;   // [B][kp_cap] mvKeyPointIndices
  // PnPsolver members that persist between iterate() calls (sd_track_pnp constructs, sd_track_pnp_iterate continues)
  int32_t* pnp_state;    // [B][4]: mnIterations, mnBestInliers, Refine() outcome for the current best set, 0
  unsigned long long* pnp_best_mask;   // [B][32] mvbBestInliers as bits over the gathered correspondences
  float* pnp_best_T;     // [B][12] mBestTcw (R row-major, t)
  // ORBmatcher::SearchByPoints (brute-force Hamming between two keyframes' map points)
  uint8_t* sp_valid1;    // [B][kp_cap] currentKF keypoint holds a map point that is not bad
  uint8_t* sp_valid2;    // [B][kp_cap] the same for pKF (the `ref` extractor's frame)
  int32_t* sp_match;     // [B][kp_cap] pKF keypoint index assigned to the currentKF keypoint, or -1
  int32_t* sp_n;         // [B]
  // local map (TrackLocalMap's search); capacity M like the last-frame arrays
  uint8_t* lm_cand;      // [B][M]   point reaches isInFrustum (not bad, not already seen in this frame)
  double* lm_Xw;         // [B][M][3]
  double* lm_normal;     // [B][M][3] GetNormal()
  float* lm_min;         // [B][M]   GetMinDistanceInvariance()
  float* lm_max;         // [B][M]   GetMaxDistanceInvariance()
  float* lm_mfmax;       // [B][M]   mfMaxDistance
  uint8_t* lm_desc;      // [B][M][32]
  int32_t* lm_obs;       // [B][M]
  int32_t* lm_n;         // [B]
  uint8_t* lm_kclaim;    // [B][kp_cap] keypoint already holds a point with Observations() > 0
  // outputs
  uint8_t* lm_inview;    // [B][M]   mbTrackInView
  float* lm_proj;        // [B][M][3] mTrackProjX, mTrackProjY, mTrackProjXR
  int32_t* lm_level;     // [B][M]   mnTrackScaleLevel
  float* lm_cos;         // [B][M]   mTrackViewCos
  int32_t* lm_match;     // [B][kp_cap] index into the local-map arrays or -1
  int32_t* lm_nmatch;    // [B]
  // Optimizer::PoseOptimization outputs
  double* po_T;          // [B][16] optimised Tcw, column-major
  uint8_t* po_outlier;   // [B][kp_cap] mvbOutlier
  int32_t* po_info;      // [B][8]: nInitialCorrespondences, nBad, rounds, g2o iterations, LM trials, nInitial - nBad
  // Tracking::TrackWithMotionModel outcome (sd_track_with_motion_model)
  int32_t* tw_info;      // [B][4]: status (0 few matches, 1 few inliers, 2 tracked), nmatches after the outlier discard,
                         //         nmatchesMap, 1 if the wider-window retry ran
  // Tracking::TrackLocalMap (sd_track_local_map): mvpMapPoints after SearchLocalPoints = frame matches + local matches
  int32_t* un_match;     // [B][kp_cap] -1 | v < M: last-frame point v | v >= M: local map point v - M
  int32_t* tl_info;      // [B][4]: status (1 failed, 2 tracked), points in mvpMapPoints, mnMatchesInliers, local matches
};

struct TrackCam {
  double fx, fy, cx, cy;                 // (double)(float) like ImageAlign::cam_fx_
  float ffx, ffy, fcx, fcy;              // Frame::fx ... (static floats)
  float min_x, max_x, min_y, max_y;      // Frame::mnMinX ...
  float bf, mb;                          // Frame::mbf, mb = mbf / fx
};

struct PnpParams {
  double probability;
  int min_inliers, max_iterations, min_set;
  float epsilon, th2;
  int n_iterations;     // argument of iterate()
  int rand_per_frame;   // entries of rand_stream per frame
  int resume;           // 0: freshly constructed solver; 1: a further iterate() on the state the last call left
};

int launch_align(const sd_orb* cur, const sd_orb* ref, const TrackBuffers& tb, const TrackCam& cam, const float* d_inv_sf,
                 const float* d_sf, int n_frames, int mode, hipStream_t s);
// retry_below > 0: only frames whose last search found fewer matches run, from the PRIOR pose (which becomes the frame's pose)
int launch_match(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_sf, int n_frames, float th,
                 int mono, int check_ori, hipStream_t s, int retry_below = 0, int note_below = 0);
// min_matches > 0: the tail of Tracking::TrackWithMotionModel around PoseOptimization (gate, outlier discard, tw_info)
// source 2 (+ min_inliers): TrackLocalMap -- the union of both match vectors, mnMatchesInliers, tl_info
int launch_pose_opt(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_inv_sigma2, int source, int n_frames,
                    hipStream_t s, int min_matches = 0, int min_inliers = 0);
// claim_from_matches: "keypoint already holds a point with Observations() > 0" is read off the frame-to-frame matches
// (tb.cur_match / tb.obs) instead of the caller's lm_kclaim flags
int launch_match_local(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_sf, const float* d_scale_thr,
                       int nlevels, int n_frames, float th, float nnratio, float cos_limit, hipStream_t s, int claim_from_matches = 0,
                       int frustum_given = 0);
int launch_features_in_area(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, int frame, float x, float y, float r,
                            int min_level, int max_level, int32_t* d_out, int out_cap, int32_t* d_n, int32_t* d_grid, hipStream_t s);
int launch_search_points(const sd_orb* cur, const sd_orb* ref, const TrackBuffers& tb, int n_frames, float nnratio, int check_ori,
                         hipStream_t s);
int launch_stereo_from_depth(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_depth, int w, int h,
                             int stride_elems, size_t frame_stride_elems, int n_frames, hipStream_t s);
int read_pnp_prof(unsigned long long* out32, int reset);
int read_sel_prof(unsigned long long* out64, int reset);
int read_align_prof(unsigned long long* out16, int reset);
int launch_pnp(const sd_orb* cur, const TrackBuffers& tb, const TrackCam& cam, const float* d_sigma2, const PnpParams& pp,
               int n_frames, hipStream_t s);

int run_epnp_debug(int n, const double* Xw, const double* uv, double fx, double fy, double cx, double cy, double* R9, double* t3,
                   double* err);

}  // namespace sd
