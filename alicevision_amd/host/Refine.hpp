// Refine.hpp — Refine stage of one tile: SGM depth upscale, 31-plane fp16 re-sweep summed over the T cameras, sliding-Gaussian
// sub-sample arg-min, colour-guided optimisation.  Restates depthMap/Refine.{hpp,cpp} of the reference on the avdm C ABI.
#pragma once

#include "Sgm.hpp"
#include "device.hpp"
#include "params.hpp"

namespace avdm_host {

class Refine
{
  public:
    // Refine.cpp:23-62
    Refine(const MultiViewParams& mp, const TileParams& tileParams, const RefineParams& refineParams, DeviceCache& deviceCache, hipStream_t stream);

    // Refine.cpp:64-96 (static: see Sgm::deviceMemoryConsumption)
    static double deviceMemoryConsumption(const TileParams& tileParams, const RefineParams& refineParams);
    double getDeviceMemoryConsumption() const { return deviceMemoryConsumption(_tileParams, _refineParams); }
    double getDeviceMemoryConsumptionUnpadded() const { return getDeviceMemoryConsumption(); }

    float* getDeviceDepthSimMap() const { return _optimizedDepthSimMap.as<float>(); }
    int getMapPitch() const { return _mapPitch; }
    int getMapWidth() const { return _mapWidth; }
    int getMapHeight() const { return _mapHeight; }

    // Refine.cpp:97-176: in_sgmDepthThicknessMap / in_sgmNormalMap are the device maps of the tile's Sgm object
    void refineRc(const Tile& tile, const Sgm& sgm);

  private:
    void refineAndFuseDepthSimMap(const Tile& tile);  // Refine.cpp:178-246
    void optimizeDepthSimMap(const Tile& tile);       // Refine.cpp:248-276
    void computeAndWriteNormalMap(const Tile& tile, const float* in_depthSimMap, const std::string& name = ""); // Refine.cpp:278-294

    const MultiViewParams& _mp;
    const TileParams& _tileParams;
    const RefineParams& _refineParams;
    DeviceCache& _deviceCache;
    hipStream_t _stream;

    int _mapWidth = 0, _mapHeight = 0, _mapPitch = 0; // float2 maps
    int _volZ = 0, _volPitchX = 0;                    // refine volume: planes, bytes per pixel (fp16, z-fastest, padded to 16 B)
    DeviceBuffer _sgmDepthPixSizeMap, _refinedDepthSimMap, _optimizedDepthSimMap;
    DeviceBuffer _sgmNormalMap, _normalMap;
    DeviceBuffer _volumeRefineSim;
    DeviceBuffer _optTmpDepthMap, _optImgVariance;
};

} // namespace avdm_host
