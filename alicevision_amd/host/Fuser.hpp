// Fuser.hpp — depth-map filtering: for every camera, count in how many neighbour cameras each depth is confirmed, then drop the depths
// without enough support.  Restates the two filtering members of fuseCut::Fuser (fuseCut/Fuser.{hpp,cpp}: filterGroups :124-141,
// filterGroupsRC :144-231, filterDepthMaps :234-247, filterDepthMapsRC :250-304) with the per-pixel work on the GPU (include/avdm_fuse.h)
// instead of one OpenMP thread per camera; files in and out are the reference's (depth / similarity EXR maps, `_nmodMap.png`).
#pragma once

#include "MultiViewParams.hpp"
#include "depthMapUtils.hpp"
#include "device.hpp"

#include <list>
#include <map>
#include <memory>
#include <vector>

namespace avdm_host {

class Fuser
{
  public:
    // maxDeviceMaps: depth maps kept in HBM (least recently used first out); 0 = $AVDM_FUSE_CACHE or 64
    explicit Fuser(const MultiViewParams& mp, int deviceId = 0, int maxDeviceMaps = 0);
    ~Fuser();
    Fuser(const Fuser&) = delete;
    Fuser& operator=(const Fuser&) = delete;

    void filterGroups(const std::vector<int>& cams, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, int nNearestCams);
    bool filterGroupsRC(int rc, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, int nNearestCams);
    void filterDepthMaps(const std::vector<int>& cams, int minNumOfModals, int minNumOfModalsWSP2SSP);
    bool filterDepthMapsRC(int rc, int minNumOfModals, int minNumOfModalsWSP2SSP);
    // filterGroups followed by filterDepthMaps (main_depthMapFiltering.cpp:137-138) in ONE pass over the cameras: a camera's second
    // pass only needs its own modal counts, so its maps are filtered while they are still in HBM instead of being decoded again.
    // Same files, same contents as the two calls.
    void filterGroupsAndDepthMaps(const std::vector<int>& cams, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, int nNearestCams,
                                  int minNumOfModals, int minNumOfModalsWSP2SSP);

  private:
    struct DeviceMap
    {
        DeviceBuffer buf;
        int width = 0, height = 0;
    };
    // depth map of a camera in HBM (decoded and uploaded on first use)
    std::shared_ptr<DeviceMap> deviceDepthMap(int cam);
    // decode the depth maps of `cams` that are not in HBM yet on the host cores, then upload them
    void prefetch(const std::vector<int>& cams);
    void upload(int cam, const FloatMap& map);
    // modal counts of `rc` into _nmod (device); simMap is uploaded into _sim
    void runGroupsKernel(int rc, const FloatMap& simMap, const std::vector<int>& tcams, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP);

    const MultiViewParams& _mp;
    int _deviceId;
    size_t _maxDeviceMaps;
    hipStream_t _stream = nullptr;
    std::map<int, std::pair<std::shared_ptr<DeviceMap>, std::list<int>::iterator>> _cache;
    std::list<int> _lru;
    DeviceBuffer _scratch, _sim, _nmod, _depthTmp;
};

} // namespace avdm_host
