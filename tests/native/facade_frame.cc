// The drop-in overloads of include/sdslam/sdslam.hpp on stand-ins for the reference's own types: Frame / MapPoint with the
// member names of src/Frame.h and src/MapPoint.h, an Eigen-like 4 x 4 / 3-vector and a cv::Mat-like descriptor.  With
// -DRUN_ON_GPU it runs one TrackWithMotionModel body (src/Tracking.cc:668-693: ComputePose, SearchByProjection,
// PoseOptimization) on two 640 x 480 frames read from a raw file and prints what the Python tests compare.
#include <sdslam/sdslam.hpp>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
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
  Vector3d GetWorldPos() { return X; }
  Mat GetDescriptor() { return Mat{desc}; }
  int Observations() { return nobs; }
  bool isBad() { return false; }
};
struct Frame {
  static float fx, fy, cx, cy, mnMinX, mnMaxX, mnMinY, mnMaxY;
  float mbf = 0.f;
  int N = 0;
  std::vector<SD_SLAM::KeyPoint> mvKeys, mvKeysUn;
  std::vector<MapPoint*> mvpMapPoints;
  std::vector<bool> mvbOutlier;
  Matrix4d Tcw;
  Matrix4d GetPose() const { return Tcw; }
  void SetPose(const Matrix4d& T) { Tcw = T; }
};
float Frame::fx = 500.f, Frame::fy = 500.f, Frame::cx = 320.f, Frame::cy = 240.f;
float Frame::mnMinX = 0.f, Frame::mnMaxX = 640.f, Frame::mnMinY = 0.f, Frame::mnMaxY = 480.f;
}  // namespace ref

int main(int argc, char** argv) {
  // instantiate every template (compile + link check; runs nothing without RUN_ON_GPU)
  auto f1 = &SD_SLAM::FrameTracker::ComputePose<ref::Frame>;
  auto f2 = &SD_SLAM::FrameTracker::SearchByProjection<ref::Frame>;
  auto f3 = &SD_SLAM::FrameTracker::PoseOptimization<ref::Frame>;
  if (!f1 || !f2 || !f3) return 1;
#ifdef RUN_ON_GPU
  if (argc < 2) return 2;
  // input file: u8 cur[480*640], u8 ref[480*640], f64 Tref[16], f64 Tprior[16] (column-major), i32 npts, then per point
  // {i32 ref keypoint index, f64 X, Y, Z}
  FILE* fp = std::fopen(argv[1], "rb");
  if (!fp) return 3;
  std::vector<unsigned char> cur(640 * 480), rf(640 * 480);
  ref::Frame Cur, Last;
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
  std::fclose(fp);
  SD_SLAM::FrameTracker trk(ecur, elast, 1000);
  // src/Tracking.cc:668-693
  const ref::Matrix4d predicted = Cur.Tcw;
  double err = 0;
  if (!trk.ComputePose(Cur, Last, &err)) Cur.SetPose(predicted);
  int nmatches = trk.SearchByProjection(Cur, Last, 8.f, true);
  int ngood = trk.PoseOptimization(&Cur);
  int nout = 0;
  for (int i = 0; i < Cur.N; i++) nout += Cur.mvpMapPoints[i] && Cur.mvbOutlier[i];
  std::printf("RESULT %d %d %d %d", Cur.N, nmatches, ngood, nout);
  for (int i = 0; i < 16; i++) std::printf(" %.17g", Cur.Tcw.d[i]);
  std::printf("\n");
  std::printf("MATCH");
  for (int i = 0; i < Cur.N; i++) {
    int m = -1;
    if (Cur.mvpMapPoints[i])
      for (int j = 0; j < Last.N; j++)
        if (Last.mvpMapPoints[j] == Cur.mvpMapPoints[i]) { m = j; break; }
    std::printf(" %d", m);
  }
  std::printf("\n");
#else
  (void)argc; (void)argv;
#endif
  std::printf("facade frame ok\n");
  return 0;
}
