// The drop-in overloads of include/sdslam/sdslam.hpp on stand-ins for the reference's own types: Frame / KeyFrame / MapPoint
// with the member names of src/Frame.h, src/KeyFrame.h and src/MapPoint.h, an Eigen-like 4 x 4 / 3-vector and a cv::Mat-like
// descriptor.  Without RUN_ON_GPU it only instantiates every template (compile + link check).  With -DRUN_ON_GPU it runs, on
// two 640 x 480 frames read from a raw file, the bodies of
//   Tracking::TrackWithMotionModel      src/Tracking.cc:668-693   (ComputePose(F, F), SearchByProjection(F, F), PoseOptimization)
//   Tracking::TrackReferenceKeyFrame    src/Tracking.cc:583-644   (ComputePose(F, KF), SearchByProjection(F, KF), PoseOptimization, discard)
//   Tracking::Relocalization, one turn  src/Tracking.cc:1069-1092 (ComputePose(F, KF, fast), SearchByProjection(F, KF), PoseOptimization)
//   LoopClosing::DetectLoop, one turn   src/LoopClosing.cc:132-134 (ComputePose(KF, KF))
//   Tracking::SearchLocalPoints' search src/Tracking.cc:937       (SearchByProjection(F, vpMapPoints, th))
//   PnPsolver(F, matches) + find()      src/PnPsolver.h:67-76
// and prints what tests/test_robustness_gpu.py compares with the oracle.
#include <sdslam/sdslam.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <set>
#include <vector>

namespace ref {
struct Matrix4d {
  double d[16];
  double* data() { return d; }
  const double* data() const { return d; }
};
struct Vector3d {
  double v[3];
  double operator()(int i) const { return v[i]; }
};
struct Mat {
  const unsigned char* data;
};
struct MapPoint {
  Vector3d X;
  unsigned char desc[32];
  int nobs = 1;
  bool bad = false;
  // Frame::isInFrustum's outputs (src/MapPoint.h:92-97)
  float mTrackProjX = 0, mTrackProjY = 0, mTrackProjXR = 0, mTrackViewCos = 0;
  bool mbTrackInView = false;
  int mnTrackScaleLevel = 0;
  Vector3d GetWorldPos() { return X; }
  Mat GetDescriptor() { return Mat{desc}; }
  int Observations() { return nobs; }
  bool isBad() { return bad; }
};
struct Frame {
  static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY;
  float mbf = 0.f;
  int N = 0;
  unsigned long mnId = 0;
  std::vector<SD_SLAM::KeyPoint> mvKeys, mvKeysUn;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  Matrix4d Tcw;
  Matrix4d GetPose() const { return Tcw; }
  void SetPose(const Matrix4d& T) { Tcw = T; }
};
float Frame::fx = 500.f, Frame::fy = 500.f, Frame::cx = 320.f, Frame::cy = 240.f;
float Frame::mnMinX = 0.f, Frame::mnMaxX = 640.f, Frame::mnMinY = 0.f, Frame::mnMaxY = 480.f;
struct KeyFrame {   // camera members are per-object constants here (src/KeyFrame.h:148,169-172)
  const float fx = 500.f, fy = 500.f, cx = 320.f, cy = 240.f, mbf = 0.f;
  const int mnMinX = 0, mnMaxX = 640, mnMinY = 0, mnMaxY = 480;
  unsigned long mnId = 0;
  int N = 0;
  std::vector<SD_SLAM::KeyPoint> mvKeys, mvKeysUn;
  std::vector<MapPoint*> matches;
  Matrix4d Tcw;
  Matrix4d GetPose() const { return Tcw; }
  std::set<MapPoint*> GetMapPoints() {   // src/KeyFrame.cc: every non-null, non-bad match, as a set (pointer order)
    std::set<MapPoint*> s;
    for (MapPoint* p : matches)
      if (p && !p->isBad()) s.insert(p);
    return s;
  }
  std::vector<MapPoint*> GetMapPointMatches() { return matches; }
};
}  // namespace ref

#ifdef RUN_ON_GPU
static void print_pose(const ref::Matrix4d& T) {
  for (int i = 0; i < 16; i++) std::printf(" %.17g", T.d[i]);
}
static void print_matches(const char* tag, const ref::Frame& Cur, const std::vector<ref::MapPoint*>& source) {
  std::printf("%s", tag);
  for (int i = 0; i < Cur.N; i++) {
    int m = -1;
    if (Cur.mvpMapPoints[i])
      for (size_t j = 0; j < source.size(); j++)
        if (source[j] == Cur.mvpMapPoints[i]) { m = (int)j; break; }
    std::printf(" %d", m);
  }
  std::printf("\n");
}
#endif

int main(int argc, char** argv) {
  // instantiate every template (compile + link check; runs nothing without RUN_ON_GPU)
  using FT = SD_SLAM::FrameTracker;
  bool (FT::*f1)(ref::Frame&, const ref::Frame&, double*) = &FT::ComputePose<ref::Frame>;
  int (FT::*f2)(ref::Frame&, const ref::Frame&, float, bool, bool) = &FT::SearchByProjection<ref::Frame>;
  int (FT::*f3)(ref::Frame*) = &FT::PoseOptimization<ref::Frame>;
  bool (FT::*f4)(ref::Frame&, ref::KeyFrame*, bool, double*) = &FT::ComputePose<ref::Frame, ref::KeyFrame>;
  bool (FT::*f5)(ref::KeyFrame*, ref::KeyFrame*, double*) = &FT::ComputePose<ref::KeyFrame>;
  int (FT::*f6)(ref::Frame&, ref::KeyFrame*, float, bool, bool) = &FT::SearchByProjection<ref::Frame, ref::KeyFrame>;
  int (FT::*f7)(ref::Frame&, const std::vector<ref::MapPoint*>&, float, float) = &FT::SearchByProjection<ref::Frame, ref::MapPoint>;
  void (FT::*f8)(const ref::Frame&, const std::vector<ref::MapPoint*>&) = &FT::PnPsolverConstruct<ref::Frame, ref::MapPoint>;
  if (!f1 || !f2 || !f3 || !f4 || !f5 || !f6 || !f7 || !f8) return 1;
#ifdef RUN_ON_GPU
  if (argc < 2) return 2;
  // input file: u8 cur[480*640], u8 ref[480*640], f64 Tref[16], f64 Tprior[16] (column-major), i32 npts, then per point
  // {i32 ref keypoint index, f64 X, Y, Z}; i32 nlocal, then per local map point {f64 X, Y, Z, u8 desc[32], i32 obs,
  // i32 mbTrackInView, f32 mTrackProjX, mTrackProjY, mTrackProjXR, i32 mnTrackScaleLevel, f32 mTrackViewCos}
  FILE* fp = std::fopen(argv[1], "rb");
  if (!fp) return 3;
  std::vector<unsigned char> cur(640 * 480), rf(640 * 480);
  ref::Frame Cur, Last;
  Cur.mnId = 2;
  Last.mnId = 1;
  int npts = 0;
  if (std::fread(cur.data(), 1, cur.size(), fp) != cur.size() || std::fread(rf.data(), 1, rf.size(), fp) != rf.size() ||
      std::fread(Last.Tcw.d, 8, 16, fp) != 16 || std::fread(Cur.Tcw.d, 8, 16, fp) != 16 || std::fread(&npts, 4, 1, fp) != 1)
    return 4;
  SD_SLAM::ORBextractor ecur(1000, 1.2f, 8, 20, 640, 480), elast(1000, 1.2f, 8, 20, 640, 480);
  std::vector<unsigned char> dcur, dlast;
  ecur(cur.data(), 640, 480, 640, Cur.mvKeys, dcur);       // Frame::Frame -> ORBextractor::operator()
  elast(rf.data(), 640, 480, 640, Last.mvKeys, dlast);
  Cur.mvKeysUn = Cur.mvKeys; Last.mvKeysUn = Last.mvKeys;  // k1 == 0
  Cur.N = (int)Cur.mvKeys.size(); Last.N = (int)Last.mvKeys.size();
  Cur.mvpMapPoints.assign(Cur.N, nullptr); Cur.mvbOutlier.assign(Cur.N, false);
  Last.mvpMapPoints.assign(Last.N, nullptr); Last.mvbOutlier.assign(Last.N, false);
  std::vector<std::unique_ptr<ref::MapPoint>> pool;
  for (int k = 0; k < npts; k++) {
    int idx;
    double X[3];
    if (std::fread(&idx, 4, 1, fp) != 1 || std::fread(X, 8, 3, fp) != 3) return 5;
    pool.emplace_back(new ref::MapPoint());
    ref::MapPoint* p = pool.back().get();
    for (int c = 0; c < 3; c++) p->X.v[c] = X[c];
    std::memcpy(p->desc, dlast.data() + (size_t)idx * 32, 32);
    Last.mvpMapPoints[idx] = p;
  }
  int nlocal = 0;
  if (std::fread(&nlocal, 4, 1, fp) != 1) return 6;
  std::vector<ref::MapPoint*> vpLocal;
  for (int k = 0; k < nlocal; k++) {
    pool.emplace_back(new ref::MapPoint());
    ref::MapPoint* p = pool.back().get();
    int inview = 0;
    if (std::fread(p->X.v, 8, 3, fp) != 3 || std::fread(p->desc, 1, 32, fp) != 32 || std::fread(&p->nobs, 4, 1, fp) != 1 ||
        std::fread(&inview, 4, 1, fp) != 1 || std::fread(&p->mTrackProjX, 4, 1, fp) != 1 || std::fread(&p->mTrackProjY, 4, 1, fp) != 1 ||
        std::fread(&p->mTrackProjXR, 4, 1, fp) != 1 || std::fread(&p->mnTrackScaleLevel, 4, 1, fp) != 1 ||
        std::fread(&p->mTrackViewCos, 4, 1, fp) != 1)
      return 7;
    p->mbTrackInView = inview != 0;
    vpLocal.push_back(p);
  }
  std::fclose(fp);
  SD_SLAM::FrameTracker trk(ecur, elast, 1000);
  const ref::Matrix4d predicted = Cur.Tcw;

  // ---- Tracking::TrackWithMotionModel, src/Tracking.cc:668-693
  {
    double err = 0;
    if (!trk.ComputePose(Cur, Last, &err)) Cur.SetPose(predicted);
    std::fill(Cur.mvpMapPoints.begin(), Cur.mvpMapPoints.end(), nullptr);
    int nmatches = trk.SearchByProjection(Cur, Last, 8.f, true);
    int ngood = trk.PoseOptimization(&Cur);
    int nout = 0;
    for (int i = 0; i < Cur.N; i++) nout += Cur.mvpMapPoints[i] && Cur.mvbOutlier[i];
    std::printf("RESULT %d %d %d %d", Cur.N, nmatches, ngood, nout);
    print_pose(Cur.Tcw);
    std::printf("\n");
    print_matches("MATCH", Cur, Last.mvpMapPoints);
  }
  // ---- the local-map search on the frame as TrackWithMotionModel left it (outliers discarded first, src/Tracking.cc:696-710)
  {
    for (int i = 0; i < Cur.N; i++)
      if (Cur.mvpMapPoints[i] && Cur.mvbOutlier[i]) { Cur.mvpMapPoints[i] = nullptr; Cur.mvbOutlier[i] = false; }
    const std::vector<ref::MapPoint*> before = Cur.mvpMapPoints;
    const int nloc = trk.SearchByProjection(Cur, vpLocal, 1.f);
    std::printf("RESULTLOCAL %d\nMATCHLOCAL", nloc);
    for (int i = 0; i < Cur.N; i++) {
      int m = -1;
      if (Cur.mvpMapPoints[i] != before[i])
        for (size_t j = 0; j < vpLocal.size(); j++)
          if (vpLocal[j] == Cur.mvpMapPoints[i]) { m = (int)j; break; }
      std::printf(" %d", m);
    }
    std::printf("\n");
  }
  // ---- PnPsolver(F, mvpMapPoints) + find() on the frame-to-frame matches
  {
    std::vector<ref::MapPoint*> fm(Cur.N, nullptr);
    for (int i = 0; i < Cur.N; i++)
      for (int j = 0; j < Last.N && !fm[i]; j++)
        if (Cur.mvpMapPoints[i] && Last.mvpMapPoints[j] == Cur.mvpMapPoints[i]) fm[i] = Cur.mvpMapPoints[i];
    trk.PnPsolverConstruct(Cur, fm);
    trk.SetRansacParameters(0.99, 10, 200, 4, 0.28f, 5.991f);
    std::vector<bool> inl;
    int ninl = 0;
    float T[16];
    std::srand(1);   // glibc's default state (the reference never seeds); something in this process has drawn from rand() before
    const bool ok = trk.find(inl, ninl, T, [] { return std::rand(); });
    std::printf("RESULTPNP %d %d", ok ? 1 : 0, ninl);
    for (int i = 0; i < 16; i++) std::printf(" %.9g", T[i]);
    std::printf("\nINLPNP");
    for (int i = 0; i < Cur.N; i++) std::printf(" %d", inl[i] ? 1 : 0);
    std::printf("\nPNPMATCH");
    for (int i = 0; i < Cur.N; i++) std::printf(" %d", fm[i] ? 1 : 0);
    std::printf("\n");
  }
  // the reference keyframe = the last frame promoted (src/KeyFrame.cc: copies mvKeys / mvKeysUn / mvpMapPoints / pose)
  ref::KeyFrame KF;
  KF.mnId = 7;
  KF.N = Last.N;
  KF.mvKeys = Last.mvKeys;
  KF.mvKeysUn = Last.mvKeysUn;
  KF.matches = Last.mvpMapPoints;
  KF.Tcw = Last.Tcw;
  {
    const std::set<ref::MapPoint*> s = KF.GetMapPoints();
    std::printf("SETORDER");
    for (ref::MapPoint* p : s)
      for (int j = 0; j < Last.N; j++)
        if (Last.mvpMapPoints[j] == p) { std::printf(" %d", j); break; }
    std::printf("\n");
  }
  // ---- Tracking::TrackReferenceKeyFrame, src/Tracking.cc:583-644 (threshold_ = 8, monocular)
  {
    const ref::Matrix4d last_pose = Last.GetPose();
    Cur.SetPose(last_pose);
    if (!trk.ComputePose(Cur, &KF)) Cur.SetPose(last_pose);
    const ref::Matrix4d aligned = Cur.Tcw;
    std::fill(Cur.mvpMapPoints.begin(), Cur.mvpMapPoints.end(), nullptr);
    int nmatches = trk.SearchByProjection(Cur, &KF, 8.f, true);
    int retried = 0;
    if (nmatches < 20) {
      retried = 1;
      Cur.SetPose(last_pose);
      std::fill(Cur.mvpMapPoints.begin(), Cur.mvpMapPoints.end(), nullptr);
      nmatches = trk.SearchByProjection(Cur, Last, 16.f, true);
    }
    print_matches("MATCHKF", Cur, KF.matches);
    int ngood = -1, nmatchesMap = 0;
    if (nmatches >= 20) {
      ngood = trk.PoseOptimization(&Cur);
      for (int i = 0; i < Cur.N; i++)
        if (Cur.mvpMapPoints[i]) {
          if (Cur.mvbOutlier[i]) { Cur.mvpMapPoints[i] = nullptr; Cur.mvbOutlier[i] = false; nmatches--; }
          else if (Cur.mvpMapPoints[i]->Observations() > 0) nmatchesMap++;
        }
    }
    std::printf("RESULTKF %d %d %d %d", nmatches, ngood, nmatchesMap, retried);
    print_pose(aligned);
    print_pose(Cur.Tcw);
    std::printf("\n");
  }
  // ---- one turn of Tracking::Relocalization's loop, src/Tracking.cc:1069-1092
  {
    Cur.SetPose(KF.GetPose());
    double err = 0;
    const bool ok = trk.ComputePose(Cur, &KF, true, &err);
    const ref::Matrix4d aligned = Cur.Tcw;
    int nmatches = -1, ngood = -1;
    if (ok) {
      std::fill(Cur.mvpMapPoints.begin(), Cur.mvpMapPoints.end(), nullptr);
      nmatches = trk.SearchByProjection(Cur, &KF, 8.f, true);
      if (nmatches >= 20) ngood = trk.PoseOptimization(&Cur);
    }
    std::printf("RESULTRELOC %d %d %d %.17g", ok ? 1 : 0, nmatches, ngood, err);
    print_pose(aligned);
    print_pose(Cur.Tcw);
    std::printf("\n");
  }
  // ---- one turn of LoopClosing::DetectLoop's loop, src/LoopClosing.cc:132-134: the current frame promoted to a keyframe
  {
    ref::KeyFrame CurKF;
    CurKF.mnId = 8;
    CurKF.N = Cur.N;
    CurKF.mvKeys = Cur.mvKeys;
    CurKF.mvKeysUn = Cur.mvKeysUn;
    CurKF.matches = Cur.mvpMapPoints;
    CurKF.Tcw = Cur.Tcw;
    double err = 0;
    const bool ok = trk.ComputePose(&CurKF, &KF, &err);
    std::printf("RESULTLOOP %d %.17g\n", ok ? 1 : 0, err);
  }
#else
  (void)argc; (void)argv;
#endif
  std::printf("facade frame ok\n");
  return 0;
}
