// sdslam.hpp -- header-only C++ facade over the C ABI (sdslam_hip.h) that re-exposes the
// reference's class and method names for the tracking hot path, over POD types that are
// layout-compatible with what the reference's callers hold:
//   SD_SLAM::KeyPoint   == cv::KeyPoint (28 bytes)
//   descriptors         == N x 32 uint8 row-major (cv::Mat CV_8U)
//   poses               == 16 doubles column-major (Eigen::Matrix4d::data())
// A reference translation unit (Frame.cc / Tracking.cc style) keeps its call sites; see
// INTEGRATION.md for the type-alias header that maps cv::Mat / Eigen arguments onto these.
//
//   ORBextractor   reference src/ORBextractor.h:38-70
//   ORBmatcher     reference src/ORBmatcher.h:40-52  (DescriptorDistance, SearchByProjection(Frame,Frame))
//   ImageAlign     reference src/ImageAlign.h:32-44  (ComputePose, GetError)
//   PnPsolver      reference src/PnPsolver.h:67-76   (SetRansacParameters, iterate)
// ImageAlign / ORBmatcher / PnPsolver operate on a TrackBatch (frames resident on the GPU).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../sdslam_hip.h"

namespace SD_SLAM {

typedef sd_keypoint KeyPoint;
static_assert(sizeof(KeyPoint) == 28, "cv::KeyPoint layout");

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) {
  if (rc != SD_OK) throw Error(rc, sd_last_error());
}

// Image pyramid level as handed back by ORBextractor::operator() (host copy on demand).
struct Mat8 {
  int cols = 0, rows = 0;
  std::vector<uint8_t> data;
  const uint8_t* ptr(int y) const { return data.data() + (size_t)y * cols; }
};

class ORBextractor {
 public:
  enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

  ORBextractor(int nfeatures, float scaleFactor, int nlevels, int thFAST, int max_w = 1280, int max_h = 960, int max_batch = 1,
               int device = 0)
      : nlevels_(nlevels), scaleFactor_(scaleFactor), cap_(nfeatures) {
    check(sd_orb_create(nfeatures, scaleFactor, nlevels, thFAST, max_w, max_h, max_batch, device, &h_));
  }
  ~ORBextractor() { sd_orb_destroy(h_); }
  ORBextractor(const ORBextractor&) = delete;
  ORBextractor& operator=(const ORBextractor&) = delete;

  // operator()(image, mask, keypoints, descriptors, imagePyramid); mask is ignored like in the
  // reference.  `image` is an 8-bit single-channel buffer (cv::Mat::data / step).
  void operator()(const uint8_t* image, int cols, int rows, int step, std::vector<KeyPoint>& keypoints,
                  std::vector<uint8_t>& descriptors, std::vector<Mat8>* imagePyramid = nullptr) {
    keypoints.resize(cap_);
    descriptors.resize((size_t)cap_ * 32);
    int n = 0;
    check(sd_orb_extract(h_, image, cols, rows, step, keypoints.data(), descriptors.data(), cap_, &n));
    keypoints.resize(n);
    descriptors.resize((size_t)n * 32);
    if (imagePyramid && cols > 0 && rows > 0) {
      imagePyramid->resize(nlevels_);
      for (int l = 0; l < nlevels_; l++) {
        Mat8& m = (*imagePyramid)[l];
        check(sd_orb_level_info(h_, l, &m.cols, &m.rows));
        m.data.resize((size_t)m.cols * m.rows);
        check(sd_orb_level_copy(h_, 0, l, 0, m.data.data(), m.cols));
      }
    }
  }

  int GetLevels() { return nlevels_; }
  float GetScaleFactor() { return scaleFactor_; }
  std::vector<float> GetScaleFactors() { return table(0); }
  std::vector<float> GetInverseScaleFactors() { return table(1); }
  std::vector<float> GetScaleSigmaSquares() { return table(2); }
  std::vector<float> GetInverseScaleSigmaSquares() { return table(3); }

  // Frame::UndistortKeyPoints (src/Frame.cc:335-366): K as Converter::toCvMat(K) (CV_32F), DistCoef {k1,k2,p1,p2,k3}.
  // After this, matching / PnP read mvKeysUn; UndistortedKeyPoints() returns them for the frames of the last call.
  void SetDistortion(float fx, float fy, float cx, float cy, float k1, float k2 = 0, float p1 = 0, float p2 = 0, float k3 = 0) {
    check(sd_orb_set_distortion(h_, fx, fy, cx, cy, k1, k2, p1, p2, k3));
  }
  void UndistortedKeyPoints(int frame, std::vector<KeyPoint>& mvKeysUn, int n) {
    std::vector<KeyPoint> all(cap_);
    check(sd_orb_download_undistorted(h_, frame, 1, all.data(), cap_));
    mvKeysUn.assign(all.begin(), all.begin() + (n < cap_ ? n : cap_));
  }

  sd_orb* handle() { return h_; }

 private:
  std::vector<float> table(int which) {
    std::vector<float> t[4];
    for (auto& v : t) v.resize(nlevels_);
    check(sd_orb_scale_tables(h_, t[0].data(), t[1].data(), t[2].data(), t[3].data()));
    return t[which];
  }
  sd_orb* h_ = nullptr;
  int nlevels_;
  float scaleFactor_;
  int cap_;
};

// Flattened view of what TrackWithMotionModel reads from LastFrame (src/Tracking.cc:654-718):
// one entry per last-frame keypoint.
struct LastFrameView {
  std::vector<uint8_t> valid;    // mvpMapPoints[i] != NULL && !mvbOutlier[i]
  std::vector<double> Xw;        // 3 per entry: pMP->GetWorldPos()
  std::vector<uint8_t> desc;     // 32 per entry: pMP->GetDescriptor()
  std::vector<int32_t> octave;   // mvKeys[i].octave
  std::vector<float> angle;      // mvKeysUn[i].angle
  std::vector<int32_t> obs;      // pMP->Observations()
  int size() const { return (int)valid.size(); }
};

// One batch of (current frame, last frame) pairs resident on the GPU.
class TrackBatch {
 public:
  TrackBatch(ORBextractor& cur, ORBextractor& ref, int max_points, int max_batch, int pnp_max_iterations = 300)
      : max_points_(max_points) {
    check(sd_track_create(cur.handle(), ref.handle(), max_points, max_batch, pnp_max_iterations, &h_));
  }
  ~TrackBatch() { sd_track_destroy(h_); }
  TrackBatch(const TrackBatch&) = delete;
  TrackBatch& operator=(const TrackBatch&) = delete;

  // Frame statics (src/Frame.cc:158-174)
  void SetCamera(float fx, float fy, float cx, float cy, float bf, float minX, float maxX, float minY, float maxY) {
    check(sd_track_set_camera(h_, fx, fy, cx, cy, bf, minX, maxX, minY, maxY));
  }
  void SetLastFrame(int frame, const LastFrameView& v) {
    LastFrameView p = v;   // pad to capacity
    const int n = v.size();
    p.valid.resize(max_points_); p.Xw.resize((size_t)max_points_ * 3); p.desc.resize((size_t)max_points_ * 32);
    p.octave.resize(max_points_); p.angle.resize(max_points_); p.obs.resize(max_points_);
    check(sd_track_set_last(h_, frame, 1, &n, p.valid.data(), p.Xw.data(), p.desc.data(), p.octave.data(), p.angle.data(),
                            p.obs.data()));
  }
  // LastFrame.GetPose(), CurrentFrame.GetPose() (prior); 16 doubles column-major each
  void SetPoses(int frame, const double* Tlast, const double* Tcur_prior) { check(sd_track_set_poses(h_, frame, 1, Tlast, Tcur_prior)); }
  // one current frame (frame `cur_frame` of the cur extractor) against the keyframes held in the batch slots; -1 = off
  void SetCurrentBroadcast(int cur_frame) { check(sd_track_set_current_broadcast(h_, cur_frame)); }
  // Frame::ComputeStereoFromRGBD (src/Frame.cc:399-417) on the current frames: depth = CV_32F images, one per frame
  void ComputeStereoFromRGBD(int n_frames, const float* imDepth, int cols, int rows, int step_elems, size_t frame_step_elems) {
    check(sd_track_stereo_from_depth(h_, n_frames, imDepth, cols, rows, step_elems, frame_step_elems));
  }
  // TrackLocalMap: local map points of one frame (src/Tracking.cc:898-939), flattened in mvpLocalMapPoints order
  struct LocalMapView {
    std::vector<uint8_t> cand;                 // reaches isInFrustum: !isBad() && mnLastFrameSeen != frame id
    std::vector<double> Xw, normal;            // GetWorldPos(), GetNormal()  (3 per point)
    std::vector<float> min_dist, max_dist;     // GetMinDistanceInvariance(), GetMaxDistanceInvariance()
    std::vector<float> mf_max_dist;            // mfMaxDistance (MapPoint::PredictScale)
    std::vector<uint8_t> desc;                 // GetDescriptor(), 32 per point
    std::vector<int32_t> obs;                  // Observations()
  };
  void SetLocalMap(int frame, const LocalMapView& v, const uint8_t* kp_claimed /* kp_cap flags or nullptr */) {
    LocalMapView p = v;
    const int n = (int)v.cand.size();
    p.cand.resize(max_points_); p.Xw.resize((size_t)max_points_ * 3); p.normal.resize((size_t)max_points_ * 3);
    p.min_dist.resize(max_points_); p.max_dist.resize(max_points_); p.mf_max_dist.resize(max_points_);
    p.desc.resize((size_t)max_points_ * 32); p.obs.resize(max_points_);
    check(sd_track_set_local(h_, frame, 1, &n, p.cand.data(), p.Xw.data(), p.normal.data(), p.min_dist.data(), p.max_dist.data(),
                             p.mf_max_dist.data(), p.desc.data(), p.obs.data(), kp_claimed));
  }
  // stereo frames: mvuRight computed by the caller (-1 = no match)
  void SetURight(int frame, const float* mvuRight, int n) { check(sd_track_set_uright(h_, frame, 1, mvuRight, n)); }
  sd_track* handle() { return h_; }

 private:
  sd_track* h_ = nullptr;
  int max_points_;
};

class ImageAlign {
 public:
  ImageAlign() {}
  // bool ComputePose(Frame &CurrentFrame, const Frame &LastFrame) for every pair of the batch;
  // per-frame results through Result().
  void ComputePose(TrackBatch& batch, int n_frames, bool keyframe = false, bool fast = false) {
    check(sd_track_align(batch.handle(), n_frames, keyframe ? (fast ? 2 : 1) : 0));
  }
  void ComputePoseKF(TrackBatch& batch, int n_frames) { check(sd_track_align(batch.handle(), n_frames, 3)); }
  // returns the bool of ComputePose; Tcw = pose set by CurrentFrame.SetPose; error = GetError()
  bool Result(TrackBatch& batch, int frame, double Tcw[16], double* error = nullptr) {
    int32_t ok = 0;
    check(sd_track_get_align(batch.handle(), frame, 1, Tcw, error, &ok, nullptr, nullptr));
    return ok != 0;
  }
};

class ORBmatcher {
 public:
  static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;
  ORBmatcher(float nnratio = 0.6f, bool checkOri = true) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}
  static int DescriptorDistance(const uint8_t* a, const uint8_t* b) { return sd_hamming(a, b); }
  // int SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
  void SearchByProjection(TrackBatch& batch, int n_frames, float th, bool bMono) {
    check(sd_track_match(batch.handle(), n_frames, th, bMono ? 1 : 0, mbCheckOrientation ? 1 : 0));
  }
  // CurrentFrame.mvpMapPoints as indices into LastFrame (-1 = NULL); returns nmatches
  // SearchByProjection(Frame&, const vector<MapPoint*>&, th): isInFrustum (viewing cosine limit 0.5) + the local-map search
  void SearchLocalPoints(TrackBatch& batch, int n_frames, float th) {
    check(sd_track_match_local(batch.handle(), n_frames, th, mfNNratio, 0.5f));
  }
  int LocalResult(TrackBatch& batch, int frame, std::vector<int32_t>& local_match, int kp_cap) {
    local_match.resize(kp_cap);
    int32_t n = 0;
    check(sd_track_get_local(batch.handle(), frame, 1, local_match.data(), kp_cap, &n, nullptr, nullptr, nullptr, nullptr));
    return n;
  }
  // int SearchByPoints(KeyFrame* currentKF, KeyFrame* pKF, vector<MapPoint*>& matches) (src/ORBmatcher.h:56, brute-force
  // Hamming; LoopClosing::ComputeSim3): has_mp_* = GetMapPointMatches()[i] != NULL && !isBad(), n flags per keyframe
  void SearchByPoints(TrackBatch& batch, int n_frames, const uint8_t* has_mp_cur, const uint8_t* has_mp_ref, int n) {
    check(sd_track_set_point_flags(batch.handle(), 0, n_frames, has_mp_cur, has_mp_ref, n));
    check(sd_track_search_by_points(batch.handle(), n_frames, mfNNratio, mbCheckOrientation ? 1 : 0));
  }
  int PointsResult(TrackBatch& batch, int frame, std::vector<int32_t>& matches12, int kp_cap) {
    matches12.resize(kp_cap);
    int32_t n = 0;
    check(sd_track_get_point_matches(batch.handle(), frame, 1, matches12.data(), kp_cap, &n));
    return n;
  }
  int Result(TrackBatch& batch, int frame, std::vector<int32_t>& mvpMapPoints, int kp_cap) {
    mvpMapPoints.resize(kp_cap);
    int32_t n = 0;
    check(sd_track_get_matches(batch.handle(), frame, 1, mvpMapPoints.data(), kp_cap, &n));
    return n;
  }

 protected:
  float mfNNratio;
  bool mbCheckOrientation;
};

// Optimizer::PoseOptimization (src/Optimizer.h: static int PoseOptimization(Frame*)), batched.
class Optimizer {
 public:
  // source 0: map points of ORBmatcher::SearchByProjection(batch, ...); 1: of SearchLocalPoints
  static void PoseOptimization(TrackBatch& batch, int n_frames, int source = 0) { check(sd_track_pose_opt(batch.handle(), n_frames, source)); }
  // returns nInitialCorrespondences - nBad; Tcw = 16 doubles column-major; mvbOutlier resized to kp_cap
  static int Result(TrackBatch& batch, int frame, double Tcw[16], std::vector<uint8_t>& mvbOutlier, int kp_cap) {
    mvbOutlier.resize(kp_cap);
    int32_t info[8];
    check(sd_track_get_pose_opt(batch.handle(), frame, 1, Tcw, mvbOutlier.data(), kp_cap, info));
    return info[5];
  }
};

// Tracking::Relocalization (src/Tracking.cc:1064-1097): every keyframe attempt of the loop is one slot of `batch`
// (slot order = the order the reference tries them, kfs.rbegin() first; SetLastFrame = the keyframe's map points,
// SetPoses(kf pose, kf pose)).  Returns the slot at which the reference's loop returns true, or -1.
class Tracking {
 public:
  struct Tracked {
    bool ok;            // return value of TrackWithMotionModel
    int nmatches;       // after "Discard outliers"
    int nmatchesMap;
    bool retried;       // the 2 * threshold search ran
  };
  // Tracking::TrackWithMotionModel (src/Tracking.cc:654-718) for every frame of the batch; the caller has set the last
  // frame (SetLastFrame) and the poses (SetPoses: last pose, motion-model prediction).  referenceKF: ImageAlign against
  // the slot's keyframe as TrackReferenceKeyFrame does (:583-644); align_image = Tracking::align_image_ (:122).
  static void TrackWithMotionModel(TrackBatch& batch, int n_frames, float threshold, bool bMono, bool align_image = true,
                                   bool referenceKF = false) {
    check(sd_track_with_motion_model(batch.handle(), n_frames, align_image ? (referenceKF ? 1 : 0) : -1, threshold, bMono ? 1 : 0, 20, 10));
  }
  // Tracking::TrackLocalMap (src/Tracking.cc:720-751) on top of TrackWithMotionModel's matches and pose; the local map is
  // TrackBatch::SetLocalMap (UpdateLocalMap stays with the caller).  th: 1, 3 for RGB-D, 5 right after a relocalisation.
  static void TrackLocalMap(TrackBatch& batch, int n_frames, float th = 1.f) { check(sd_track_local_map(batch.handle(), n_frames, th, 0.8f, 0.5f, 30)); }
  // returns TrackLocalMap's return value; mnMatchesInliers and mvpMapPoints (v < max_points: last-frame point v, else
  // local map point v - max_points) on request
  static bool LocalMapResult(TrackBatch& batch, int frame, int* mnMatchesInliers = nullptr, std::vector<int32_t>* mvpMapPoints = nullptr,
                             int kp_cap = 0) {
    int32_t i4[4];
    if (mvpMapPoints) mvpMapPoints->resize(kp_cap);
    check(sd_track_get_local_map(batch.handle(), frame, 1, mvpMapPoints ? mvpMapPoints->data() : nullptr, kp_cap, i4));
    if (mnMatchesInliers) *mnMatchesInliers = i4[2];
    return i4[0] == 2;
  }
  static Tracked Result(TrackBatch& batch, int frame) {
    int32_t i4[4];
    check(sd_track_get_tracked(batch.handle(), frame, 1, i4));
    return Tracked{i4[0] == 2, i4[1], i4[2], i4[3] != 0};
  }
  static int Relocalization(TrackBatch& batch, int n_keyframes, int cur_frame, float threshold, bool bMono) {
    int32_t winner = -1;
    check(sd_track_relocalize(batch.handle(), n_keyframes, cur_frame, threshold, bMono ? 1 : 0, 20, 10, &winner, nullptr));
    return winner;
  }
};

// The candidate search of LoopClosing::DetectLoop (src/LoopClosing.cc:115-149): slot i = kfs[i]; excluded[i] marks the
// current keyframe itself and its connected keyframes.  Returns the slots with error < 1.5 * best (vpCandidateKFs as a set).
class LoopClosing {
 public:
  static std::vector<int32_t> DetectLoopCandidates(TrackBatch& batch, int n_keyframes, int cur_frame, const std::vector<uint8_t>& excluded,
                                                   double* best_error = nullptr) {
    if (!excluded.empty() && (int)excluded.size() < n_keyframes) throw Error(SD_ERR_INVALID_ARG, "excluded shorter than n_keyframes");
    std::vector<int32_t> cand(n_keyframes);
    int32_t n = 0;
    check(sd_track_detect_loop(batch.handle(), n_keyframes, cur_frame, excluded.empty() ? nullptr : excluded.data(), cand.data(),
                               n_keyframes, &n, best_error, nullptr));
    cand.resize(n);
    return cand;
  }
};

// ------------------------------------------------------------------------------------------------------------------
// Drop-in overloads on the reference's OWN Frame / MapPoint types: the call sites of
// src/Tracking.cc:668-693 keep their arguments -- `image_align.ComputePose(mCurrentFrame, mLastFrame)`,
// `matcher.SearchByProjection(mCurrentFrame, mLastFrame, th, bMono)`, `Optimizer::PoseOptimization(&mCurrentFrame)` --
// and the SoA flattening of INTEGRATION.md section 3 happens in here.  FrameT / its map-point type only need the reference's
// member names (src/Frame.h, src/MapPoint.h): N, mvpMapPoints, mvbOutlier, mvKeys, mvKeysUn, GetPose(), SetPose(),
// static fx fy cx cy mnMinX mnMaxX mnMinY mnMaxY, mbf; GetWorldPos() (indexable by (k)), GetDescriptor() (with .data),
// Observations(), isBad().  One FrameTracker per camera: `cur` holds the extraction of the current frame, `last` that of the
// previous one (the shimmed ORBextractor of INTEGRATION.md section 3 fills them from Frame's constructor; swap per frame).
class FrameTracker {
 public:
  FrameTracker(ORBextractor& cur, ORBextractor& last, int max_points = 2048, int pnp_max_iterations = 300)
      : batch_(cur, last, max_points, 1, pnp_max_iterations), max_points_(max_points), pnp_max_its_(pnp_max_iterations) {}
  TrackBatch& batch() { return batch_; }
  // What the tracker's map-point arrays hold is remembered between calls: the three calls of one TrackWithMotionModel body
  // (ComputePose, SearchByProjection, PoseOptimization on the same two frames) flatten and upload mLastFrame ONCE.  The key is
  // (object address, mnId, N, the mvpMapPoints pointers and outlier flags); call this when a frame's map points were edited in
  // place some other way (world positions moved by a bundle adjustment between two calls, say).
  void InvalidateCache() { kind_ = NONE; }

  // bool ImageAlign::ComputePose(Frame &CurrentFrame, const Frame &LastFrame)          src/ImageAlign.h:36, src/Tracking.cc:668
  template <class FrameT>
  bool ComputePose(FrameT& CurrentFrame, const FrameT& LastFrame, double* error = nullptr) {
    Upload(CurrentFrame, LastFrame);
    return Align(CurrentFrame, 0, error);
  }
  // bool ImageAlign::ComputePose(Frame &CurrentFrame, KeyFrame *LastKF, bool fast = false)
  //                                                      src/ImageAlign.h:39, src/ImageAlign.cc:106-176, src/Tracking.cc:595,1077
  // The points are LastKF->GetMapPoints() in ITS iteration order (a std::set<MapPoint*> in the reference: pointer order), the
  // first 300 (100 when fast); neither isBad() nor outlier flags are looked at (src/ImageAlign.cc:129-136).  The `last`
  // extractor must hold the keyframe's frame (its pyramid is what the reference reads from LastKF->mvImagePyramid).
  template <class FrameT, class KeyFrameT>
  bool ComputePose(FrameT& CurrentFrame, KeyFrameT* LastKF, bool fast = false, double* error = nullptr) {
    SetCameraOf(CurrentFrame);
    UploadSetPoints(LastKF);
    const auto Tl = LastKF->GetPose();
    const auto Tc = CurrentFrame.GetPose();
    for (int i = 0; i < 16; i++) last_pose_[i] = Tl.data()[i];
    check(sd_track_set_poses(batch_.handle(), 0, 1, Tl.data(), Tc.data()));
    return Align(CurrentFrame, fast ? 2 : 1, error);
  }
  // bool ImageAlign::ComputePose(KeyFrame *CurrentKF, KeyFrame *LastKF)                src/ImageAlign.h:42, src/LoopClosing.cc:133
  // level 4 only, identity start, rejected above 0.03; no pose is written.  The `cur` extractor holds CurrentKF's frame.
  template <class KeyFrameT>
  bool ComputePose(KeyFrameT* CurrentKF, KeyFrameT* LastKF, double* error = nullptr) {
    SetCameraOf(*CurrentKF);
    UploadSetPoints(LastKF);
    const auto Tl = LastKF->GetPose();
    const auto Tc = CurrentKF->GetPose();
    check(sd_track_set_poses(batch_.handle(), 0, 1, Tl.data(), Tc.data()));
    check(sd_track_align(batch_.handle(), 1, 3));
    int32_t ok = 0;
    check(sd_track_get_align(batch_.handle(), 0, 1, nullptr, error, &ok, nullptr, nullptr));
    return ok != 0;
  }
  // int ORBmatcher::SearchByProjection(Frame &CurrentFrame, const Frame &LastFrame, const float th, const bool bMono)
  //                                                                                     src/ORBmatcher.h:46, src/Tracking.cc:677
  template <class FrameT>
  int SearchByProjection(FrameT& CurrentFrame, const FrameT& LastFrame, float th, bool bMono, bool checkOrientation = true) {
    Upload(CurrentFrame, LastFrame);
    return Search(CurrentFrame, LastFrame.mvpMapPoints, th, bMono, checkOrientation);
  }
  // int ORBmatcher::SearchByProjection(Frame &CurrentFrame, KeyFrame *pKF, const float th, const bool bMono)
  //                                                      src/ORBmatcher.h:52, src/ORBmatcher.cc:1077-1207, src/Tracking.cc:603,1083
  // over pKF->GetMapPointMatches() (NULL and isBad() points skipped), octave / angle from pKF->mvKeys / mvKeysUn
  template <class FrameT, class KeyFrameT>
  int SearchByProjection(FrameT& CurrentFrame, KeyFrameT* pKF, float th, bool bMono, bool checkOrientation = true) {
    SetCameraOf(CurrentFrame);
    const auto vpMapPointMatches = pKF->GetMapPointMatches();
    const int n = (int)vpMapPointMatches.size();
    if (!Cached(KF_MATCHES, pKF, IdOf(*pKF, 0), n, vpMapPointMatches, nullptr)) {
      LastFrameView v;
      Blank(v, n);
      for (int i = 0; i < n; i++) {
        v.octave[i] = pKF->mvKeys[i].octave;
        v.angle[i] = pKF->mvKeysUn[i].angle;
        auto* p = vpMapPointMatches[i];
        if (!p || p->isBad()) continue;
        FillPoint(v, i, p);
      }
      batch_.SetLastFrame(0, v);
    }
    const auto Tl = pKF->GetPose();
    const auto Tc = CurrentFrame.GetPose();
    for (int i = 0; i < 16; i++) last_pose_[i] = Tl.data()[i];
    check(sd_track_set_poses(batch_.handle(), 0, 1, Tl.data(), Tc.data()));
    return Search(CurrentFrame, vpMapPointMatches, th, bMono, checkOrientation);
  }
  // int ORBmatcher::SearchByProjection(Frame &F, const vector<MapPoint*> &vpMapPoints, const float th)
  //                                                      src/ORBmatcher.h:47, src/ORBmatcher.cc:43-119, src/Tracking.cc:937
  // on the isInFrustum results the caller's SearchLocalPoints left in the map points (mbTrackInView, mTrackProjX / Y / XR,
  // mnTrackScaleLevel, mTrackViewCos); keypoints of F that already hold a point with Observations() > 0 stay closed.
  template <class FrameT, class MapPointT>
  int SearchByProjection(FrameT& F, const std::vector<MapPointT*>& vpMapPoints, float th, float nnratio = 0.8f) {
    SetCameraOf(F);
    kind_ = NONE;
    const int n = (int)vpMapPoints.size();
    if (n > max_points_) throw Error(SD_ERR_CAPACITY, "more local map points than the tracker's max_points");
    std::vector<uint8_t> in_view((size_t)max_points_, 0), desc((size_t)max_points_ * 32, 0), claimed((size_t)cap(F), 0);
    std::vector<float> proj((size_t)max_points_ * 3, 0.f), vcos((size_t)max_points_, 0.f);
    std::vector<int32_t> level((size_t)max_points_, 0), obs((size_t)max_points_, 0);
    for (int i = 0; i < n; i++) {
      auto* p = vpMapPoints[i];
      if (!p->mbTrackInView || p->isBad()) continue;
      in_view[i] = 1;
      proj[(size_t)i * 3] = p->mTrackProjX; proj[(size_t)i * 3 + 1] = p->mTrackProjY; proj[(size_t)i * 3 + 2] = p->mTrackProjXR;
      level[i] = p->mnTrackScaleLevel;
      vcos[i] = p->mTrackViewCos;
      const auto D = p->GetDescriptor();
      for (int k = 0; k < 32; k++) desc[(size_t)i * 32 + k] = D.data[k];
      obs[i] = p->Observations();
    }
    for (int i = 0; i < F.N; i++) claimed[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0;
    UploadURight(F, 0);
    check(sd_track_set_local_view(batch_.handle(), 0, 1, &n, in_view.data(), proj.data(), level.data(), vcos.data(), desc.data(), obs.data(),
                                  claimed.data()));
    check(sd_track_match_local_view(batch_.handle(), 1, th, nnratio));
    std::vector<int32_t> idx(cap(F));
    int32_t nm = 0;
    check(sd_track_get_local(batch_.handle(), 0, 1, idx.data(), (int)idx.size(), &nm, nullptr, nullptr, nullptr, nullptr));
    for (int i = 0; i < F.N; i++)
      if (idx[i] >= 0) F.mvpMapPoints[i] = vpMapPoints[idx[i]];
    return nm;
  }
  // int Optimizer::PoseOptimization(Frame *pFrame)                                       src/Optimizer.h, src/Tracking.cc:693
  // on whatever pFrame->mvpMapPoints holds.  Right behind a SearchByProjection on the same frame -- mvpMapPoints still exactly what
  // that search assigned -- the matches and map points are already on the device and only the pose travels; otherwise keypoint
  // i <-> map point i is flattened (sd_track_set_matches).
  template <class FrameT>
  int PoseOptimization(FrameT* pFrame) {
    FrameT& F = *pFrame;
    SetCameraOf(F);
    bool resident = searched_frame_ == (const void*)pFrame && searched_id_ == IdOf(F, 0) && (int)assigned_.size() == F.N && kind_ != NONE &&
                    kind_ != CUR_POINTS;
    for (int i = 0; resident && i < F.N; i++) resident = (const void*)F.mvpMapPoints[i] == assigned_[i];
    const auto Tc = F.GetPose();
    if (resident) {
      check(sd_track_set_poses(batch_.handle(), 0, 1, last_pose_, Tc.data()));
    } else {
      LastFrameView v;
      std::vector<int32_t> cm((size_t)F.N, -1);
      Blank(v, F.N);
      for (int i = 0; i < F.N; i++) {
        auto* p = F.mvpMapPoints[i];
        if (!p) continue;
        v.valid[i] = 1;
        cm[i] = i;
        const auto X = p->GetWorldPos();
        for (int k = 0; k < 3; k++) v.Xw[(size_t)i * 3 + k] = X(k);
        v.obs[i] = p->Observations();
      }
      batch_.SetLastFrame(0, v);
      kind_ = CUR_POINTS;
      check(sd_track_set_poses(batch_.handle(), 0, 1, Tc.data(), Tc.data()));
      check(sd_track_set_matches(batch_.handle(), 0, 1, cm.data(), F.N));
    }
    UploadURight(F, 0);                        // stereo edges where mvuRight >= 0 (src/Optimizer.cc:262-300)
    check(sd_track_pose_opt(batch_.handle(), 1, 0));
    std::vector<uint8_t> outl(cap(F));
    double T[16];
    int32_t info[8];
    check(sd_track_get_pose_opt(batch_.handle(), 0, 1, T, outl.data(), (int)outl.size(), info));
    SetPoseOf(F, T);
    for (int i = 0; i < F.N; i++)
      if (F.mvpMapPoints[i]) F.mvbOutlier[i] = outl[i] != 0;
    return info[5];
  }

  // PnPsolver(const Frame &F, const vector<MapPoint*> &vpMapPointMatches) + SetRansacParameters + find / iterate
  //                                                      src/PnPsolver.h:67-76, src/PnPsolver.cc:71-244
  // (dead code in the reference; kept because BASELINE names it).  One solver per FrameTracker at a time.  rand_fn
  // supplies what SD_SLAM::Random would draw from rand() (src/extra/utils.cc:23-26): minSet values per iteration, drawn up
  // front for the iterations the call may run (the reference draws them one iteration at a time).
  template <class FrameT, class MapPointT>
  void PnPsolverConstruct(const FrameT& F, const std::vector<MapPointT*>& vpMapPointMatches) {
    SetCameraOf(F);
    const int n = (int)vpMapPointMatches.size();
    LastFrameView v;
    Blank(v, n);
    std::vector<int32_t> cm((size_t)n, -1);
    for (int i = 0; i < n; i++) {
      auto* p = vpMapPointMatches[i];
      if (!p || p->isBad()) continue;          // src/PnPsolver.cc:84-87
      v.valid[i] = 1;
      cm[i] = i;
      const auto X = p->GetWorldPos();
      for (int k = 0; k < 3; k++) v.Xw[(size_t)i * 3 + k] = X(k);
    }
    batch_.SetLastFrame(0, v);
    kind_ = CUR_POINTS;
    const auto Tc = F.GetPose();
    check(sd_track_set_poses(batch_.handle(), 0, 1, Tc.data(), Tc.data()));
    check(sd_track_set_matches(batch_.handle(), 0, 1, cm.data(), n));
    pnp_n_ = n;
    pnp_started_ = false;
    SetRansacParameters();
  }
  void SetRansacParameters(double probability = 0.99, int minInliers = 8, int maxIterations = 300, int minSet = 4, float epsilon = 0.4f,
                           float th2 = 5.991f) {
    pnp_p_ = probability; pnp_min_inl_ = minInliers; pnp_max_iterations_ = maxIterations; pnp_min_set_ = minSet; pnp_eps_ = epsilon;
    pnp_th2_ = th2;
    pnp_started_ = false;
  }
  // cv::Mat iterate(int nIterations, bool &bNoMore, vector<bool> &vbInliers, int &nInliers): Tcw = 4 x 4 CV_32F row-major;
  // returns false for the reference's empty Mat
  template <class RandFn>
  bool iterate(int nIterations, bool& bNoMore, std::vector<bool>& vbInliers, int& nInliers, float Tcw[16], RandFn rand_fn) {
    if (pnp_n_ < 0) throw Error(SD_ERR_INVALID_ARG, "PnPsolverConstruct has not been called");
    if (!pnp_started_) {
      // the first call may run to max(mRansacMaxIts, nIterations) (the `||` of src/PnPsolver.cc:177); later calls add nIterations each
      const int per_frame = 4 * pnp_max_its_;
      std::vector<int32_t> r((size_t)per_frame);
      for (auto& x : r) x = (int32_t)rand_fn();
      check(sd_track_set_rand(batch_.handle(), 0, 1, r.data(), per_frame));
      check(sd_track_pnp(batch_.handle(), 1, pnp_p_, pnp_min_inl_, pnp_max_iterations_, pnp_min_set_, pnp_eps_, pnp_th2_, nIterations));
      pnp_started_ = true;
    } else {
      check(sd_track_pnp_iterate(batch_.handle(), 1, nIterations));
    }
    std::vector<uint8_t> inl((size_t)(pnp_n_ > max_points_ ? pnp_n_ : max_points_));
    int32_t info[8];
    check(sd_track_get_pnp(batch_.handle(), 0, 1, Tcw, inl.data(), (int)inl.size(), info));
    bNoMore = info[2] != 0;
    nInliers = info[1];
    vbInliers.assign((size_t)pnp_n_, false);
    for (int i = 0; i < pnp_n_; i++) vbInliers[i] = inl[i] != 0;
    return info[0] != 0;
  }
  // cv::Mat find(vector<bool> &vbInliers, int &nInliers) { bool bFlag; return iterate(mRansacMaxIts, bFlag, vbInliers, nInliers); }
  template <class RandFn>
  bool find(std::vector<bool>& vbInliers, int& nInliers, float Tcw[16], RandFn rand_fn) {
    bool bFlag;
    return iterate(pnp_max_iterations_, bFlag, vbInliers, nInliers, Tcw, rand_fn);
  }

 private:
  enum Kind { NONE, FRAME_POINTS, KF_SET_POINTS, KF_MATCHES, CUR_POINTS };
  template <class FrameT>
  int cap(const FrameT& F) { return F.N > max_points_ ? F.N : max_points_; }
  // frame / keyframe identity for the cache: mnId where the type has one (src/Frame.h:121, src/KeyFrame.h:112)
  template <class T>
  static auto IdOf(const T& F, int) -> decltype((unsigned long long)F.mnId) { return (unsigned long long)F.mnId; }
  template <class T>
  static unsigned long long IdOf(const T&, long) { return 0; }
  static void Blank(LastFrameView& v, int n) {
    v.valid.assign(n, 0); v.Xw.assign((size_t)n * 3, 0.0); v.desc.assign((size_t)n * 32, 0);
    v.octave.assign(n, 0); v.angle.assign(n, 0.f); v.obs.assign(n, 0);
  }
  template <class MapPointT>
  static void FillPoint(LastFrameView& v, int i, MapPointT* p) {
    v.valid[i] = 1;
    const auto X = p->GetWorldPos();
    for (int k = 0; k < 3; k++) v.Xw[(size_t)i * 3 + k] = X(k);
    const auto D = p->GetDescriptor();
    for (int k = 0; k < 32; k++) v.desc[(size_t)i * 32 + k] = D.data[k];
    v.obs[i] = p->Observations();
  }
  // is (kind, object, id, the pointer vector [+ outlier flags]) what the tracker's arrays already hold?  Records it otherwise.
  template <class Vec>
  bool Cached(Kind kind, const void* obj, unsigned long long id, int n, const Vec& pts, const std::vector<bool>* outl) {
    bool same = kind_ == kind && key_obj_ == obj && key_id_ == id && (int)key_pts_.size() == n;
    for (int i = 0; same && i < n; i++) same = key_pts_[i] == (const void*)pts[i] && (!outl || key_outl_[i] == (*outl)[i]);
    if (same) return true;
    kind_ = kind; key_obj_ = obj; key_id_ = id;
    key_pts_.resize(n);
    key_outl_.assign(n, false);
    for (int i = 0; i < n; i++) {
      key_pts_[i] = (const void*)pts[i];
      if (outl) key_outl_[i] = (*outl)[i];
    }
    return false;
  }
  // CurrentFrame.mvuRight (filled by Frame::ComputeStereoFromRGBD / ComputeStereoMatches in the reference's constructor)
  template <class FrameT>
  auto UploadURight(const FrameT& F, int) -> decltype((void)F.mvuRight) {
    if (!F.mvuRight.empty()) batch_.SetURight(0, F.mvuRight.data(), (int)F.mvuRight.size());
  }
  template <class FrameT>
  void UploadURight(const FrameT&, long) {}   // frame types without mvuRight (monocular builds)
  // Frame's camera members are static (src/Frame.h:109-112,175-178), KeyFrame's are per-object constants (src/KeyFrame.h:148,169-172):
  // instance access reads both
  template <class FrameT>
  void SetCameraOf(const FrameT& F) {
    batch_.SetCamera(F.fx, F.fy, F.cx, F.cy, F.mbf, (float)F.mnMinX, (float)F.mnMaxX, (float)F.mnMinY, (float)F.mnMaxY);
  }
  template <class FrameT>
  static void SetPoseOf(FrameT& F, const double T[16]) {
    auto M = F.GetPose();                      // Eigen::Matrix4d: 16 doubles column-major behind data()
    for (int i = 0; i < 16; i++) M.data()[i] = T[i];
    F.SetPose(M);
  }
  template <class FrameT>
  bool Align(FrameT& CurrentFrame, int mode, double* error) {
    check(sd_track_align(batch_.handle(), 1, mode));
    double T[16];
    int32_t ok = 0;
    check(sd_track_get_align(batch_.handle(), 0, 1, T, error, &ok, nullptr, nullptr));
    if (ok) SetPoseOf(CurrentFrame, T);
    return ok != 0;
  }
  // the search on what the tracker's arrays hold; `source` = the pointer vector the indices refer to
  template <class FrameT, class Vec>
  int Search(FrameT& CurrentFrame, const Vec& source, float th, bool bMono, bool checkOrientation) {
    if (!bMono) UploadURight(CurrentFrame, 0);   // RGB-D / stereo: the mvuRight gate of src/ORBmatcher.cc:1020-1025
    check(sd_track_match(batch_.handle(), 1, th, bMono ? 1 : 0, checkOrientation ? 1 : 0));
    std::vector<int32_t> idx(CurrentFrame.N > 0 ? cap(CurrentFrame) : 1);
    int32_t n = 0;
    check(sd_track_get_matches(batch_.handle(), 0, 1, idx.data(), (int)idx.size(), &n));
    for (int i = 0; i < CurrentFrame.N; i++)
      if (idx[i] >= 0) CurrentFrame.mvpMapPoints[i] = source[idx[i]];   // the search only ever ASSIGNS
    // remembered for PoseOptimization's resident path: valid when the caller had cleared mvpMapPoints before the search, as
    // every call site of the reference does (src/Tracking.cc:602,676,1080) -- PoseOptimization compares pointer by pointer
    searched_frame_ = (const void*)&CurrentFrame;
    searched_id_ = IdOf(CurrentFrame, 0);
    assigned_.assign((size_t)CurrentFrame.N, nullptr);
    for (int i = 0; i < CurrentFrame.N; i++)
      if (idx[i] >= 0) assigned_[i] = (const void*)source[idx[i]];
    return n;
  }
  // LastKF->GetMapPoints() in iteration order, the first 300 (what ImageAlign's KeyFrame overloads gather)
  template <class KeyFrameT>
  void UploadSetPoints(KeyFrameT* LastKF) {
    const auto mappoints = LastKF->GetMapPoints();
    std::vector<const void*> ptrs;
    for (auto it = mappoints.begin(); it != mappoints.end() && (int)ptrs.size() < 300; ++it) ptrs.push_back((const void*)*it);
    const int n = (int)ptrs.size();
    if (n > max_points_) throw Error(SD_ERR_CAPACITY, "FrameTracker max_points below 300");
    if (Cached(KF_SET_POINTS, LastKF, IdOf(*LastKF, 0), n, ptrs, nullptr)) return;
    LastFrameView v;
    Blank(v, n);
    int i = 0;
    for (auto it = mappoints.begin(); it != mappoints.end() && i < n; ++it, ++i) {
      v.valid[i] = 1;
      const auto X = (*it)->GetWorldPos();
      for (int k = 0; k < 3; k++) v.Xw[(size_t)i * 3 + k] = X(k);
    }
    batch_.SetLastFrame(0, v);
  }
  // what TrackWithMotionModel reads of mLastFrame (src/ImageAlign.cc:64-72, src/ORBmatcher.cc:968-1042) + both poses
  template <class FrameT>
  void Upload(const FrameT& CurrentFrame, const FrameT& LastFrame) {
    SetCameraOf(CurrentFrame);
    const int n = LastFrame.N;
    if (!Cached(FRAME_POINTS, &LastFrame, IdOf(LastFrame, 0), n, LastFrame.mvpMapPoints, &LastFrame.mvbOutlier)) {
      LastFrameView v;
      Blank(v, n);
      for (int i = 0; i < n; i++) {
        v.octave[i] = LastFrame.mvKeys[i].octave;
        v.angle[i] = LastFrame.mvKeysUn[i].angle;
        auto* p = LastFrame.mvpMapPoints[i];
        if (!p || LastFrame.mvbOutlier[i]) continue;
        FillPoint(v, i, p);
      }
      batch_.SetLastFrame(0, v);
    }
    const auto Tl = LastFrame.GetPose();
    const auto Tc = CurrentFrame.GetPose();
    for (int i = 0; i < 16; i++) last_pose_[i] = Tl.data()[i];
    check(sd_track_set_poses(batch_.handle(), 0, 1, Tl.data(), Tc.data()));
  }
  TrackBatch batch_;
  int max_points_, pnp_max_its_;
  Kind kind_ = NONE;
  const void* key_obj_ = nullptr;
  unsigned long long key_id_ = 0;
  std::vector<const void*> key_pts_;
  std::vector<bool> key_outl_;
  const void* searched_frame_ = nullptr;
  unsigned long long searched_id_ = 0;
  std::vector<const void*> assigned_;
  double last_pose_[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  int pnp_n_ = -1;
  bool pnp_started_ = false;
  double pnp_p_ = 0.99;
  int pnp_min_inl_ = 8, pnp_max_iterations_ = 300, pnp_min_set_ = 4;
  float pnp_eps_ = 0.4f, pnp_th2_ = 5.991f;
};

class PnPsolver {
 public:
  PnPsolver() { SetRansacParameters(); }
  void SetRansacParameters(double probability = 0.99, int minInliers = 8, int maxIterations = 300, int minSet = 4, float epsilon = 0.4f,
                           float th2 = 5.991f) {
    p_ = probability; minInl_ = minInliers; maxIts_ = maxIterations; minSet_ = minSet; eps_ = epsilon; th2_ = th2;
  }
  // cv::Mat iterate(int nIterations, bool &bNoMore, vector<bool> &vbInliers, int &nInliers) for the batch;
  // rand_values: 4 raw rand() values per iteration and frame (what SD_SLAM::Random would draw)
  void iterate(TrackBatch& batch, int n_frames, int nIterations, const int32_t* rand_values, int per_frame) {
    check(sd_track_set_rand(batch.handle(), 0, n_frames, rand_values, per_frame));
    check(sd_track_pnp(batch.handle(), n_frames, p_, minInl_, maxIts_, minSet_, eps_, th2_, nIterations));
  }
  // a further iterate(nIterations) on the same solvers (mnIterations and the best hypothesis carry over, src/PnPsolver.cc:177)
  void iterateAgain(TrackBatch& batch, int n_frames, int nIterations) { check(sd_track_pnp_iterate(batch.handle(), n_frames, nIterations)); }
  // PnPsolver(F, vpMapPointMatches) on a match vector of the caller's (indices into the last-frame arrays, -1 = NULL)
  static void SetMatches(TrackBatch& batch, int frame, const int32_t* vpMapPointMatches, int n) {
    check(sd_track_set_matches(batch.handle(), frame, 1, vpMapPointMatches, n));
  }
  // Tcw: 4x4 CV_32F row-major; returns false for the reference's empty cv::Mat
  bool Result(TrackBatch& batch, int frame, float Tcw[16], bool& bNoMore, std::vector<uint8_t>& vbInliers, int& nInliers, int kp_cap) {
    int32_t info[8];
    vbInliers.resize(kp_cap);
    check(sd_track_get_pnp(batch.handle(), frame, 1, Tcw, vbInliers.data(), kp_cap, info));
    bNoMore = info[2] != 0;
    nInliers = info[1];
    return info[0] != 0;
  }

 private:
  double p_;
  int minInl_, maxIts_, minSet_;
  float eps_, th2_;
};

}  // namespace SD_SLAM
