// DepthMapEstimator.hpp — tiles -> batches -> streams: the per-device job that computes and writes the depth / similarity
// maps of a list of R cameras.  Restates depthMap/DepthMapEstimator.{hpp,cpp} and the IGPUJob interface of
// depthMap/computeOnMultiGPUs.hpp:18-27.
#pragma once

#include "MultiViewParams.hpp"
#include "params.hpp"

#include <string>
#include <vector>

namespace avdm_host {

class PyramidExchange;

// computeOnMultiGPUs.hpp:18-27, plus the hooks of the multi-GPU pyramid exchange (device.hpp: PyramidExchange); a job that does not
// override them runs exactly like the reference's (every worker loads what it needs by itself)
struct IGPUJob
{
    virtual void compute(int deviceId, const std::vector<int>& cams) = 0;
    // the views (camera indices) the job reads for these R cameras; empty = the job does not take part in the exchange
    virtual std::vector<int> viewsNeeded(const std::vector<int>& cams) const { return {}; }
    // worker `worker` of exchange.nbWorkers(): compute `cams` on `deviceId`, sharing pyramids through `exchange`;
    // `allViews` = viewsNeeded(every camera of the job)
    virtual void computeShared(int worker, int deviceId, const std::vector<int>& cams, const std::vector<int>& allViews, PyramidExchange& exchange)
    {
        compute(deviceId, cams);
    }
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
    // multi-GPU form (see IGPUJob): R, SGM-T and Refine-T cameras of every tile of `cams`
    std::vector<int> viewsNeeded(const std::vector<int>& cams) const override;
    void computeShared(int worker, int deviceId, const std::vector<int>& cams, const std::vector<int>& allViews, PyramidExchange& exchange) override;

    // CPU-only part of compute(): tiles, T cameras and depth lists, no device needed (hidden CLI switch --dryRun)
    void plan(const std::vector<int>& cams, std::vector<TilePlan>& out) const;
    const std::vector<ROI>& tileRoiList() const { return _tileRoiList; }

  private:
    void computeImpl(int deviceId, const std::vector<int>& cams, int worker, const std::vector<int>* allViews, PyramidExchange* exchange);

    const MultiViewParams& _mp;
    const TileParams& _tileParams;
    const DepthMapParams& _depthMapParams;
    const SgmParams& _sgmParams;
    const RefineParams& _refineParams;
    std::vector<ROI> _tileRoiList;
};

// computeOnMultiGPUs.cpp:15-69 re-designed: one host thread per device, R cameras dealt round-robin, pyramids shared through a
// PyramidExchange when the job takes part in it (computeOnMultiGPUs.cpp of this directory)
void computeOnMultiGPUs(const std::vector<int>& cams, IGPUJob& gpujob, int nbGPUsToUse);

} // namespace avdm_host
