// DepthMapEstimator.hpp — tiles -> batches -> streams: the per-device job that computes and writes the depth / similarity
// maps of a list of R cameras.  Restates depthMap/DepthMapEstimator.{hpp,cpp} and the IGPUJob interface of
// depthMap/computeOnMultiGPUs.hpp:18-27.
#pragma once

#include "MultiViewParams.hpp"
#include "params.hpp"

#include <string>
#include <vector>

namespace avdm_host {

// computeOnMultiGPUs.hpp:18-27
struct IGPUJob
{
    virtual void compute(int deviceId, const std::vector<int>& cams) = 0;
    virtual ~IGPUJob() = default;
};

// the plan of one tile as the estimator will compute it (also what --dryRun prints)
struct TilePlan
{
    Tile tile;
    std::vector<float> depths;
    std::vector<Pixel> depthsTcLimits;
};

class DepthMapEstimator : public IGPUJob
{
  public:
    // DepthMapEstimator.cpp:27-55
    DepthMapEstimator(const MultiViewParams& mp, const TileParams& tileParams, const DepthMapParams& depthMapParams, const SgmParams& sgmParams,
                      const RefineParams& refineParams);

    // DepthMapEstimator.cpp:57-175: needs a current device (reads its free memory)
    int getNbSimultaneousTiles() const;
    // DepthMapEstimator.cpp:177-222
    void getTilesList(const std::vector<int>& cams, std::vector<Tile>& tiles) const;
    // DepthMapEstimator.cpp:224-512
    void compute(int deviceId, const std::vector<int>& cams) override;

    // CPU-only part of compute(): tiles, T cameras and depth lists, no device needed (hidden CLI switch --dryRun)
    void plan(const std::vector<int>& cams, std::vector<TilePlan>& out) const;
    const std::vector<ROI>& tileRoiList() const { return _tileRoiList; }

  private:
    const MultiViewParams& _mp;
    const TileParams& _tileParams;
    const DepthMapParams& _depthMapParams;
    const SgmParams& _sgmParams;
    const RefineParams& _refineParams;
    std::vector<ROI> _tileRoiList;
};

// computeOnMultiGPUs.cpp:15-69: one host thread per device, contiguous chunks of the camera list
void computeOnMultiGPUs(const std::vector<int>& cams, IGPUJob& gpujob, int nbGPUsToUse);

} // namespace avdm_host
