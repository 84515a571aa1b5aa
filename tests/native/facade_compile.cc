// Compile/link check of the C++ facade against libsdslam_hip.so (no GPU work is issued).
#include <sdslam/sdslam.hpp>
#include <cstdio>
int main() {
  uint8_t a[32] = {0}, b[32] = {0};
  b[3] = 0x81;
  if (SD_SLAM::ORBmatcher::DescriptorDistance(a, b) != 2) return 1;
  if (sd_device_count() == 0) {
    try {
      SD_SLAM::ORBextractor e(1000, 1.2f, 8, 20);
      return 2;                       // must not succeed without a GPU
    } catch (const SD_SLAM::Error& err) {
      if (err.code != SD_ERR_NO_DEVICE && err.code != SD_ERR_HIP) return 3;
    }
  }
  SD_SLAM::PnPsolver p;
  p.SetRansacParameters(0.99, 10, 200, 4, 0.28f, 5.991f);
  // instantiate the batched Relocalization / DetectLoop entry points (link check; they need a GPU to run)
  auto reloc = &SD_SLAM::Tracking::Relocalization;
  auto loop = &SD_SLAM::LoopClosing::DetectLoopCandidates;
  auto popt = &SD_SLAM::Optimizer::PoseOptimization;
  auto twmm = &SD_SLAM::Tracking::TrackWithMotionModel;
  auto twres = &SD_SLAM::Tracking::Result;
  auto tlm = &SD_SLAM::Tracking::TrackLocalMap;
  auto tlmres = &SD_SLAM::Tracking::LocalMapResult;
  if (!tlm || !tlmres) return 5;
  if (!reloc || !loop || !popt || !twmm || !twres) return 4;
  std::printf("facade ok\n");
  return 0;
}
