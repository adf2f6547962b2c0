// Sgm.hpp — SGM stage of one tile: similarity volumes (best / second best over the T cameras), 4-path aggregation,
// winner-take-all depth + thickness.  Restates depthMap/Sgm.{hpp,cpp} of the reference on top of the avdm C ABI.
//
// Difference of design: the path aggregation only parallelises over volume columns, so the estimator
// aggregates ALL tiles of a batch in one launch per path (avdm_volume_optimize_tiles).  sgmRc() is therefore split in
// three steps — computeVolumes(), [batched optimise by the caller through sgmTileDescriptor()], finish() — and sgmRc()
// itself remains as the one-tile form.
#pragma once

#include "SgmDepthList.hpp"
#include "device.hpp"
#include "params.hpp"

namespace avdm_host {

class Sgm
{
  public:
    // Sgm.cpp:23-76
    Sgm(const MultiViewParams& mp, const TileParams& tileParams, const SgmParams& sgmParams, bool computeDepthSimMap, bool computeNormalMap,
        DeviceCache& deviceCache, hipStream_t stream);

    // Sgm.cpp:78-114: device bytes (MB) one Sgm object allocates; static so that the scheduler can size a tile without
    // allocating one (the reference builds a throw-away object, DepthMapEstimator.cpp:81-89)
    static double deviceMemoryConsumption(const TileParams& tileParams, const SgmParams& sgmParams, bool computeDepthSimMap, bool computeNormalMap);
    double getDeviceMemoryConsumption() const { return deviceMemoryConsumption(_tileParams, _sgmParams, _computeDepthSimMap, _computeNormalMap); }
    double getDeviceMemoryConsumptionUnpadded() const { return getDeviceMemoryConsumption(); }

    float* getDeviceDepthThicknessMap() const { return _depthThicknessMap.as<float>(); }
    int getDepthThicknessMapPitch() const { return _mapPitch; }
    float* getDeviceDepthSimMap() const { return _depthSimMap.as<float>(); }
    float* getDeviceNormalMap() const { return _normalMap.as<float>(); }
    int getMapWidth() const { return _mapWidth; }   // allocated map width (max tile width at SGM resolution)
    int getMapHeight() const { return _mapHeight; }
    hipStream_t getStream() const { return _stream; }

    // Sgm.cpp:117-188
    void sgmRc(const Tile& tile, const SgmDepthList& tileDepthList);
    // the three steps of sgmRc
    void computeVolumes(const Tile& tile, const SgmDepthList& tileDepthList);   // depth upload + Sgm.cpp:203-279
    avdm_sgm_tile_t sgmTileDescriptor(const Tile& tile, const SgmDepthList& tileDepthList) const; // arguments of Sgm.cpp:281-304 for a batched launch
    avdm_sgm_tile_t layoutAndDescribe(const Tile& tile, const SgmDepthList& tileDepthList);       // the same BEFORE computeVolumes() (lays the volumes out for the tile): for avdm_volume_optimize_prepare
    size_t optimizeScratchBytes(const Tile& tile, const SgmDepthList& tileDepthList) const;
    // Sgm.cpp:327-396 (the 9-point CSV part; the Alembic exports are not built)
    void exportVolumeInformation(const Tile& tile, const SgmDepthList& tileDepthList, const DeviceBuffer& volume, const std::string& name) const;
    void optimizeDisabledCopy();                                                // Sgm.cpp:147-151
    void finish(const Tile& tile, const SgmDepthList& tileDepthList);           // Sgm.cpp:306-329 + exports + normals
    // Sgm.cpp:190-201
    void smoothThicknessMap(const Tile& tile, const RefineParams& refineParams);

  private:
    void layoutFor(const Tile& tile, int nbDepths);

    const MultiViewParams& _mp;
    const TileParams& _tileParams;
    const SgmParams& _sgmParams;
    const bool _computeDepthSimMap;
    const bool _computeNormalMap;
    DeviceCache& _deviceCache;
    hipStream_t _stream;

    int _mapWidth = 0, _mapHeight = 0, _mapPitch = 0; // float2 rows
    PinnedBuffer<float> _depths_h;
    DeviceBuffer _depths_d;
    DeviceBuffer _depthThicknessMap, _depthSimMap, _normalMap;
    DeviceBuffer _volumeBestSim, _volumeSecBestSim, _optimizeScratch;
    // layout of the current tile inside the volume allocations (z-fastest, include/avdm.h)
    int _volX = 0, _volY = 0, _volZ = 0, _pitchX = 0;
    long long _pitchY = 0;
};

} // namespace avdm_host
