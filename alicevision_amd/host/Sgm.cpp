// Sgm.cpp — see Sgm.hpp.
#include "Sgm.hpp"

#include <cstdlib>

#include "depthMapUtils.hpp"
#include "log.hpp"

#include <algorithm>

namespace avdm_host {

Sgm::Sgm(const MultiViewParams& mp, const TileParams& tileParams, const SgmParams& sgmParams, bool computeDepthSimMap, bool computeNormalMap,
         DeviceCache& deviceCache, hipStream_t stream)
  : _mp(mp),
    _tileParams(tileParams),
    _sgmParams(sgmParams),
    _computeDepthSimMap(computeDepthSimMap || sgmParams.exportIntermediateDepthSimMaps),
    _computeNormalMap(computeNormalMap || sgmParams.exportIntermediateNormalMaps),
    _deviceCache(deviceCache),
    _stream(stream)
{
    const int downscale = _sgmParams.scale * _sgmParams.stepXY;
    _mapWidth = divideRoundUp(tileParams.bufferWidth, downscale);
    _mapHeight = divideRoundUp(tileParams.bufferHeight, downscale);
    _mapPitch = _mapWidth * 8;
    const size_t maxDepths = (size_t)std::max(_sgmParams.maxDepths, 1);
    _depths_h.allocate(maxDepths);
    _depths_d.allocate(maxDepths * sizeof(float));
    const size_t mapBytes = (size_t)_mapPitch * _mapHeight;
    _depthThicknessMap.allocate(mapBytes);
    if(_computeDepthSimMap)
        _depthSimMap.allocate(mapBytes);
    if(_computeNormalMap)
        _normalMap.allocate((size_t)_mapWidth * 12 * _mapHeight);
    const size_t volBytes = (size_t)_mapWidth * _mapHeight * (size_t)(divideRoundUp((int)maxDepths, 4) * 4);
    _volumeBestSim.allocate(volBytes);
    _volumeSecBestSim.allocate(volBytes);
    if(sgmParams.doSgmOptimizeVolume)
        _optimizeScratch.allocate(avdm_volume_optimize_scratch_bytes(_mapWidth, _mapHeight, (int)maxDepths));
}

double Sgm::deviceMemoryConsumption(const TileParams& tileParams, const SgmParams& sgmParams, bool computeDepthSimMap, bool computeNormalMap)
{
    // same terms as Sgm.cpp:78-95 with this implementation's layouts (uint8 z-fastest volumes, P2-map scratch instead of the
    // uint32 slice buffers)
    computeDepthSimMap = computeDepthSimMap || sgmParams.exportIntermediateDepthSimMaps;
    computeNormalMap = computeNormalMap || sgmParams.exportIntermediateNormalMaps;
    const int downscale = sgmParams.scale * sgmParams.stepXY;
    const int mapWidth = divideRoundUp(tileParams.bufferWidth, downscale), mapHeight = divideRoundUp(tileParams.bufferHeight, downscale);
    const size_t maxDepths = (size_t)std::max(sgmParams.maxDepths, 1);
    size_t bytes = maxDepths * sizeof(float);
    const size_t mapBytes = (size_t)mapWidth * 8 * mapHeight;
    bytes += mapBytes;
    if(computeDepthSimMap)
        bytes += mapBytes;
    if(computeNormalMap)
        bytes += (size_t)mapWidth * 12 * mapHeight;
    bytes += 2 * (size_t)mapWidth * mapHeight * (size_t)(divideRoundUp((int)maxDepths, 4) * 4);
    if(sgmParams.doSgmOptimizeVolume)
        bytes += avdm_volume_optimize_scratch_bytes(mapWidth, mapHeight, (int)maxDepths);
    return double(bytes) / (1024.0 * 1024.0);
}

// The SGM volumes are laid out for the tile BUFFER and the path aggregation walks that extent, like the reference: its
// cuda_volumeAggregatePath takes X / Y from the allocated volume (deviceSimilarityVolume.cu:278-283; Sgm.cpp:37-72 allocates for the
// buffer), so the reverse paths cross the 255-filled remainder of the buffer before they enter the tile.  The tile's ROI lives in the
// corner of the volume, every other kernel keeps the ROI.  Differs from an aggregation over the ROI only for tiles that do not start at
// the image origin.  AVDM_SGM_BUFFER_EXTENT=0 is the A/B switch back to the ROI extent (rounds 1-2).
static bool sgmBufferExtent()
{
    const char* e = std::getenv("AVDM_SGM_BUFFER_EXTENT");
    return !(e != nullptr && e[0] == '0');
}

void Sgm::layoutFor(const Tile& tile, int nbDepths)
{
    const ROI roi = downscaleROI(tile.roi, float(_sgmParams.scale * _sgmParams.stepXY));
    _volX = (int)roi.width();
    _volY = (int)roi.height();
    if(sgmBufferExtent() && _volX <= _mapWidth && _volY <= _mapHeight)
    {
        _volX = _mapWidth;
        _volY = _mapHeight;
    }
    _volZ = nbDepths;
    _pitchX = divideRoundUp(nbDepths, 4) * 4;
    _pitchY = (long long)_volX * _pitchX;
    if(_volX > _mapWidth || _volY > _mapHeight || nbDepths > _sgmParams.maxDepths)
        AVDM_THROW_ERROR(tile << "tile does not fit the SGM buffers (" << _volX << "x" << _volY << "x" << nbDepths << " > " << _mapWidth << "x" << _mapHeight
                              << "x" << _sgmParams.maxDepths << ").");
}

avdm_sgm_tile_t Sgm::layoutAndDescribe(const Tile& tile, const SgmDepthList& tileDepthList)
{
    layoutFor(tile, (int)tileDepthList.getDepths().size());
    return sgmTileDescriptor(tile, tileDepthList);
}

void Sgm::sgmRc(const Tile& tile, const SgmDepthList& tileDepthList)
{
    const avdm_sgm_params_t sp = _sgmParams.toAvdm();
    avdm_sgm_tile_t t;
    if(_sgmParams.doSgmOptimizeVolume)
    {
        // the adaptive-P2 maps depend only on the R pyramid: ahead of the sweep, so that the aggregation is the path launches alone
        t = layoutAndDescribe(tile, tileDepthList);
        avdmCheck(avdm_volume_optimize_prepare(1, &t, _optimizeScratch.ptr(), &sp, _stream), "avdm_volume_optimize_prepare");
    }
    computeVolumes(tile, tileDepthList);
    if(_sgmParams.doSgmOptimizeVolume)
    {
        AVDM_LOG_INFO(tile << "SGM Optimizing volume (filtering axes: " << _sgmParams.filteringAxes << ").");
        avdmCheck(avdm_volume_optimize_tiles_prepared(1, &t, _optimizeScratch.ptr(), &sp, _stream), "avdm_volume_optimize_tiles_prepared");
        AVDM_LOG_INFO(tile << "SGM Optimizing volume done.");
    }
    else
        optimizeDisabledCopy();
    finish(tile, tileDepthList);
}

void Sgm::computeVolumes(const Tile& tile, const SgmDepthList& tileDepthList)
{
    const IndexT viewId = _mp.getViewId(tile.rc);
    AVDM_LOG_INFO(tile << "SGM depth/thickness map of view id: " << viewId << ", rc: " << tile.rc << " (" << (tile.rc + 1) << " / " << _mp.ncams << ").");
    if(tile.sgmTCams.empty() || tileDepthList.getDepths().empty())
        AVDM_THROW_ERROR(tile << "Cannot compute Semi-Global Matching, no depths or no T cameras (viewId: " << viewId << ").");

    const std::vector<float>& depths = tileDepthList.getDepths();
    layoutFor(tile, (int)depths.size());
    std::copy(depths.begin(), depths.end(), _depths_h.data());
    AVDM_HIP_CHECK(hipMemcpyAsync(_depths_d.ptr(), _depths_h.data(), depths.size() * sizeof(float), hipMemcpyHostToDevice, _stream));

    AVDM_LOG_INFO(tile << "SGM Compute similarity volume.");
    const ROI downscaledRoi = downscaleROI(tile.roi, float(_sgmParams.scale * _sgmParams.stepXY));
    const avdm_roi_t roi = {{downscaledRoi.x.begin, downscaledRoi.x.end}, {downscaledRoi.y.begin, downscaledRoi.y.end}};
    const avdm_sgm_params_t sp = _sgmParams.toAvdm();

    avdmCheck(avdm_volume_initialize_u8(_volumeBestSim.as<uint8_t>(), _pitchY, _pitchX, _volX, _volY, _pitchX, 255, _stream), "avdm_volume_initialize_u8");
    avdmCheck(avdm_volume_initialize_u8(_volumeSecBestSim.as<uint8_t>(), _pitchY, _pitchX, _volX, _volY, _pitchX, 255, _stream), "avdm_volume_initialize_u8");

    const avdm_camera_t& rcCam = _deviceCache.requestCameraParams(tile.rc, _sgmParams.scale, _mp);
    const DeviceMipmapImage& rcMip = _deviceCache.requestMipmapImage(tile.rc, _mp);

    for(std::size_t tci = 0; tci < tile.sgmTCams.size(); ++tci)
    {
        const int tc = tile.sgmTCams.at(tci);
        const int firstDepth = tileDepthList.getDepthsTcLimits()[tci].x;
        const int lastDepth = firstDepth + tileDepthList.getDepthsTcLimits()[tci].y;
        const avdm_camera_t& tcCam = _deviceCache.requestCameraParams(tc, _sgmParams.scale, _mp);
        const DeviceMipmapImage& tcMip = _deviceCache.requestMipmapImage(tc, _mp);
        AVDM_LOG_DEBUG(tile << "Compute similarity volume:" << std::endl
                            << "\t- rc: " << tile.rc << std::endl
                            << "\t- tc: " << tc << " (" << (tci + 1) << "/" << tile.sgmTCams.size() << ")" << std::endl
                            << "\t- tc first depth: " << firstDepth << std::endl
                            << "\t- tc last depth: " << lastDepth << std::endl
                            << "\t- tile range x: [" << downscaledRoi.x.begin << " - " << downscaledRoi.x.end << "]" << std::endl
                            << "\t- tile range y: [" << downscaledRoi.y.begin << " - " << downscaledRoi.y.end << "]" << std::endl);
        const avdm_range_t depthRange = {(unsigned)firstDepth, (unsigned)lastDepth};
        avdmCheck(avdm_volume_compute_similarity(_volumeBestSim.as<uint8_t>(), _volumeSecBestSim.as<uint8_t>(), _pitchY, _pitchX, _depths_d.as<float>(), &rcCam,
                                                 &tcCam, &rcMip.pyramid(), &tcMip.pyramid(), &sp, depthRange, roi, _stream),
                  "avdm_volume_compute_similarity");
    }
    if(_sgmParams.updateUninitializedSim)
    {
        AVDM_LOG_DEBUG(tile << "SGM Update uninitialized similarity volume values from best similarity volume.");
        avdmCheck(avdm_volume_update_uninitialized(_volumeBestSim.as<uint8_t>(), _volumeSecBestSim.as<uint8_t>(), _pitchY, _pitchX, _volX, _volY, _volZ, _stream),
                  "avdm_volume_update_uninitialized");
    }
    AVDM_LOG_INFO(tile << "SGM Compute similarity volume done.");
    exportVolumeInformation(tile, tileDepthList, _volumeSecBestSim, "beforeFiltering"); // Sgm.cpp:139
}

void Sgm::exportVolumeInformation(const Tile& tile, const SgmDepthList& tileDepthList, const DeviceBuffer& volume, const std::string& name) const
{
    // Sgm.cpp:327-396.  (The reference's early return does not look at exportIntermediateTopographicCutVolumes: asked for alone, it
    // exports nothing — kept.)
    if(!_sgmParams.exportIntermediateVolumes && !_sgmParams.exportIntermediateCrossVolumes && !_sgmParams.exportIntermediateVolume9pCsv)
        return;
    const int tileBeginX = tile.nbTiles > 1 ? (int)tile.roi.x.begin : -1, tileBeginY = tile.nbTiles > 1 ? (int)tile.roi.y.begin : -1;
    const int nbPlanes = (int)tileDepthList.getDepths().size();
    if(_sgmParams.exportIntermediateVolumes || _sgmParams.exportIntermediateCrossVolumes || _sgmParams.exportIntermediateTopographicCutVolumes)
    {
        // the reference samples the ALLOCATED volume (in_volume_dmp.getSize()): the laid-out extent here
        const HostVolume vol = downloadVolume(volume.ptr(), false, _pitchY, _pitchX, _volX, _volY, std::min(nbPlanes, _volZ), _stream);
        if(_sgmParams.exportIntermediateVolumes)
        {
            AVDM_LOG_INFO(tile << "Export similarity volume (" << name << ").");
            exportSimilarityVolume(vol, tileDepthList.getDepths(), _mp, tile.rc, _sgmParams,
                                   getFileNameFromIndex(_mp, tile.rc, EFileType::volume, "_" + name, tileBeginX, tileBeginY), tile.roi);
        }
        if(_sgmParams.exportIntermediateCrossVolumes)
        {
            AVDM_LOG_INFO(tile << "Export similarity volume cross (" << name << ").");
            exportSimilarityVolumeCross(vol, tileDepthList.getDepths(), _mp, tile.rc, _sgmParams,
                                        getFileNameFromIndex(_mp, tile.rc, EFileType::volumeCross, "_" + name, tileBeginX, tileBeginY), tile.roi);
        }
        if(_sgmParams.exportIntermediateTopographicCutVolumes)
        {
            AVDM_LOG_INFO(tile << "Export similarity volume topographic cut (" << name << ").");
            exportSimilarityVolumeTopographicCut(vol, tileDepthList.getDepths(), _mp, tile.rc, _sgmParams,
                                                 getFileNameFromIndex(_mp, tile.rc, EFileType::volumeTopographicCut, "_" + name, tileBeginX, tileBeginY),
                                                 tile.roi);
        }
    }
    if(_sgmParams.exportIntermediateVolume9pCsv)
    {
        AVDM_LOG_INFO(tile << "Export similarity volume 9 points CSV (" << name << ").");
        const std::string stats9Path = getFileNameFromIndex(_mp, tile.rc, EFileType::stats9p, "_sgm", tileBeginX, tileBeginY);
        const ROI r = downscaleROI(tile.roi, float(_sgmParams.scale * _sgmParams.stepXY));
        exportSimilaritySamplesCSV(volume.ptr(), false, _pitchY, _pitchX, nbPlanes, (int)r.width(), (int)r.height(), name, stats9Path, _stream);
    }
}

avdm_sgm_tile_t Sgm::sgmTileDescriptor(const Tile& tile, const SgmDepthList& tileDepthList) const
{
    const ROI downscaledRoi = downscaleROI(tile.roi, float(_sgmParams.scale * _sgmParams.stepXY));
    avdm_sgm_tile_t t;
    t.out_vol = _volumeBestSim.as<uint8_t>(); // the best-sim volume is reused for the optimised similarity (Sgm.cpp:292)
    t.in_vol = _volumeSecBestSim.as<uint8_t>();
    t.pitch_y = _pitchY;
    t.pitch_x = _pitchX;
    t.last_depth_index = (int)tileDepthList.getDepths().size();
    t.roi = {{downscaledRoi.x.begin, downscaledRoi.x.end}, {downscaledRoi.y.begin, downscaledRoi.y.end}};
    if(sgmBufferExtent()) // the aggregation walks the laid-out extent (layoutFor), image coordinates still start at the ROI's begin
        t.roi = {{downscaledRoi.x.begin, downscaledRoi.x.begin + (unsigned)_volX}, {downscaledRoi.y.begin, downscaledRoi.y.begin + (unsigned)_volY}};
    t.rc_pyr = &_deviceCache.requestMipmapImage(tile.rc, _mp).pyramid();
    return t;
}

size_t Sgm::optimizeScratchBytes(const Tile& tile, const SgmDepthList& tileDepthList) const
{
    const ROI r = downscaleROI(tile.roi, float(_sgmParams.scale * _sgmParams.stepXY));
    if(sgmBufferExtent())
        return avdm_volume_optimize_scratch_bytes(_mapWidth, _mapHeight, (int)tileDepthList.getDepths().size());
    return avdm_volume_optimize_scratch_bytes((int)r.width(), (int)r.height(), (int)tileDepthList.getDepths().size());
}

void Sgm::optimizeDisabledCopy()
{
    AVDM_HIP_CHECK(hipMemcpyAsync(_volumeBestSim.ptr(), _volumeSecBestSim.ptr(), (size_t)_pitchY * _volY, hipMemcpyDeviceToDevice, _stream));
}

void Sgm::finish(const Tile& tile, const SgmDepthList& tileDepthList)
{
    exportVolumeInformation(tile, tileDepthList, _volumeBestSim, "afterFiltering"); // Sgm.cpp:155
    AVDM_LOG_INFO(tile << "SGM Retrieve best depth in volume.");
    const ROI downscaledRoi = downscaleROI(tile.roi, float(_sgmParams.scale * _sgmParams.stepXY));
    const avdm_roi_t roi = {{downscaledRoi.x.begin, downscaledRoi.x.end}, {downscaledRoi.y.begin, downscaledRoi.y.end}};
    const avdm_sgm_params_t sp = _sgmParams.toAvdm();
    const avdm_range_t depthRange = {0u, (unsigned)tileDepthList.getDepths().size()};
    const avdm_camera_t& rcCam1 = _deviceCache.requestCameraParams(tile.rc, 1, _mp);
    avdmCheck(avdm_volume_retrieve_best_depth(_depthThicknessMap.as<float>(), _mapPitch, _depthSimMap.as<float>(), _mapPitch, _depths_d.as<float>(),
                                              _volumeBestSim.as<uint8_t>(), _pitchY, _pitchX, _volZ, &rcCam1, &sp, depthRange, roi, _stream),
              "avdm_volume_retrieve_best_depth");
    AVDM_LOG_INFO(tile << "SGM Retrieve best depth in volume done.");

    if(_sgmParams.exportIntermediateDepthSimMaps)
        writeDepthSimMap(tile.rc, _mp, _tileParams, tile.roi, _depthSimMap.as<float>(), _mapPitch, _sgmParams.scale, _sgmParams.stepXY, "sgm", _stream);

    if(_computeNormalMap)
    {
        AVDM_LOG_INFO(tile << "SGM compute normal map of view id: " << _mp.getViewId(tile.rc) << ", rc: " << tile.rc << " (" << (tile.rc + 1) << " / "
                           << _mp.ncams << ").");
        const avdm_camera_t& rcCam = _deviceCache.requestCameraParams(tile.rc, _sgmParams.scale, _mp);
        avdmCheck(avdm_depth_sim_map_compute_normal(_normalMap.as<float>(), _mapWidth * 12, _depthSimMap.as<float>(), _mapPitch, &rcCam, _sgmParams.stepXY, roi,
                                                    _stream),
                  "avdm_depth_sim_map_compute_normal");
        if(_sgmParams.exportIntermediateNormalMaps)
            writeNormalMap(tile.rc, _mp, _tileParams, tile.roi, _normalMap.as<float>(), _mapWidth * 12, _sgmParams.scale, _sgmParams.stepXY, "sgm", _stream);
    }
    AVDM_LOG_INFO(tile << "SGM depth/thickness map done.");
}

void Sgm::smoothThicknessMap(const Tile& tile, const RefineParams& refineParams)
{
    AVDM_LOG_INFO(tile << "SGM Smooth thickness map.");
    const ROI downscaledRoi = downscaleROI(tile.roi, float(_sgmParams.scale * _sgmParams.stepXY));
    const avdm_roi_t roi = {{downscaledRoi.x.begin, downscaledRoi.x.end}, {downscaledRoi.y.begin, downscaledRoi.y.end}};
    const avdm_sgm_params_t sp = _sgmParams.toAvdm();
    const avdm_refine_params_t rp = refineParams.toAvdm();
    avdmCheck(avdm_depth_thickness_smooth_thickness(_depthThicknessMap.as<float>(), _mapPitch, &sp, &rp, roi, _stream), "avdm_depth_thickness_smooth_thickness");
    AVDM_LOG_INFO(tile << "SGM Smooth thickness map done.");
}

} // namespace avdm_host
