// NormalMapEstimator.hpp — normal maps of filtered depth maps.  Restates depthMap/NormalMapEstimator.{hpp,cpp}
// (compute :30-100): the IGPUJob that aliceVision_depthMapFiltering --computeNormalMaps runs through computeOnMultiGPUs
// (main_depthMapFiltering.cpp:142-155).
#pragma once

#include "DepthMapEstimator.hpp"
#include "MultiViewParams.hpp"

#include <vector>

namespace avdm_host {

class NormalMapEstimator : public IGPUJob
{
  public:
    explicit NormalMapEstimator(const MultiViewParams& mp) : _mp(mp) {}
    NormalMapEstimator(const NormalMapEstimator&) = delete;
    void operator=(const NormalMapEstimator&) = delete;

    // NormalMapEstimator.cpp:30-100
    void compute(int deviceId, const std::vector<int>& cams) override;

  private:
    const MultiViewParams& _mp;
};

} // namespace avdm_host
