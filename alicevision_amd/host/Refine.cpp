// Refine.cpp — see Refine.hpp.
#include "Refine.hpp"

#include "depthMapUtils.hpp"
#include "log.hpp"

namespace avdm_host {

Refine::Refine(const MultiViewParams& mp, const TileParams& tileParams, const RefineParams& refineParams, DeviceCache& deviceCache, hipStream_t stream)
  : _mp(mp),
    _tileParams(tileParams),
    _refineParams(refineParams),
    _deviceCache(deviceCache),
    _stream(stream)
{
    const int downscale = _refineParams.scale * _refineParams.stepXY;
    _mapWidth = divideRoundUp(tileParams.bufferWidth, downscale);
    _mapHeight = divideRoundUp(tileParams.bufferHeight, downscale);
    _mapPitch = _mapWidth * 8;
    _volZ = _refineParams.halfNbDepths * 2 + 1;
    _volPitchX = divideRoundUp(_volZ, 8) * 8 * 2;
    const size_t mapBytes = (size_t)_mapPitch * _mapHeight;
    _sgmDepthPixSizeMap.allocate(mapBytes);
    _refinedDepthSimMap.allocate(mapBytes);
    _optimizedDepthSimMap.allocate(mapBytes);
    if(_refineParams.useSgmNormalMap)
        _sgmNormalMap.allocate((size_t)_mapWidth * 12 * _mapHeight);
    if(_refineParams.exportIntermediateNormalMaps)
        _normalMap.allocate((size_t)_mapWidth * 12 * _mapHeight);
    _volumeRefineSim.allocate((size_t)_mapWidth * _mapHeight * _volPitchX);
    if(_refineParams.useColorOptimization)
    {
        _optTmpDepthMap.allocate((size_t)_mapWidth * 4 * _mapHeight);
        _optImgVariance.allocate((size_t)_mapWidth * 4 * _mapHeight);
    }
}

double Refine::deviceMemoryConsumption(const TileParams& tileParams, const RefineParams& refineParams)
{
    const int downscale = refineParams.scale * refineParams.stepXY;
    const size_t mapWidth = (size_t)divideRoundUp(tileParams.bufferWidth, downscale), mapHeight = (size_t)divideRoundUp(tileParams.bufferHeight, downscale);
    const size_t volPitchX = (size_t)divideRoundUp(refineParams.halfNbDepths * 2 + 1, 8) * 8 * 2;
    size_t bytes = 3 * mapWidth * 8 * mapHeight;
    if(refineParams.useSgmNormalMap)
        bytes += mapWidth * 12 * mapHeight;
    if(refineParams.exportIntermediateNormalMaps)
        bytes += mapWidth * 12 * mapHeight;
    bytes += mapWidth * mapHeight * volPitchX;
    if(refineParams.useColorOptimization)
        bytes += 2 * mapWidth * 4 * mapHeight;
    return double(bytes) / (1024.0 * 1024.0);
}

void Refine::refineRc(const Tile& tile, const Sgm& sgm)
{
    const IndexT viewId = _mp.getViewId(tile.rc);
    AVDM_LOG_INFO(tile << "Refine depth/sim map of view id: " << viewId << ", rc: " << tile.rc << " (" << (tile.rc + 1) << " / " << _mp.ncams << ").");

    const ROI downscaledRoi = downscaleROI(tile.roi, float(_refineParams.scale * _refineParams.stepXY));
    const avdm_roi_t roi = {{downscaledRoi.x.begin, downscaledRoi.x.end}, {downscaledRoi.y.begin, downscaledRoi.y.end}};
    const avdm_refine_params_t rp = _refineParams.toAvdm();
    {
        const avdm_camera_t& rcCam = _deviceCache.requestCameraParams(tile.rc, _refineParams.scale, _mp);
        const DeviceMipmapImage& rcMip = _deviceCache.requestMipmapImage(tile.rc, _mp);
        // upscale ratio from the ALLOCATED map widths, like deviceDepthSimilarityMap.cu:116-118
        const float ratio = float(sgm.getMapWidth()) / float(_mapWidth);
        avdmCheck(avdm_compute_sgm_upscaled_depth_pixsize_map(_sgmDepthPixSizeMap.as<float>(), _mapPitch, sgm.getDeviceDepthThicknessMap(),
                                                              sgm.getDepthThicknessMapPitch(), &rcCam, &rcMip.pyramid(), &rp, ratio, roi, _stream),
                  "avdm_compute_sgm_upscaled_depth_pixsize_map");
        if(_refineParams.exportIntermediateDepthSimMaps)
            writeDepthPixSizeMap(tile.rc, _mp, _tileParams, tile.roi, _sgmDepthPixSizeMap.as<float>(), _mapPitch, _refineParams.scale, _refineParams.stepXY,
                                 "sgmUpscaled", _stream);
        if(_refineParams.useSgmNormalMap && sgm.getDeviceNormalMap() != nullptr)
            avdmCheck(avdm_normal_map_upscale(_sgmNormalMap.as<float>(), _mapWidth * 12, sgm.getDeviceNormalMap(), sgm.getMapWidth() * 12, ratio, roi, _stream),
                      "avdm_normal_map_upscale");
    }

    if(_refineParams.useRefineFuse)
        refineAndFuseDepthSimMap(tile);
    else
    {
        AVDM_LOG_INFO(tile << "Refine and fuse depth/sim map volume disabled.");
        // the reference copies the whole allocated map (deviceDepthSimilarityMap.cu:24-44)
        avdmCheck(avdm_depth_sim_map_copy_depth_only(_refinedDepthSimMap.as<float>(), _mapPitch, _sgmDepthPixSizeMap.as<float>(), _mapPitch,
                                                     (int)downscaledRoi.width(), (int)downscaledRoi.height(), 1.0f, _stream),
                  "avdm_depth_sim_map_copy_depth_only");
    }
    if(_refineParams.exportIntermediateDepthSimMaps)
        writeDepthSimMap(tile.rc, _mp, _tileParams, tile.roi, _refinedDepthSimMap.as<float>(), _mapPitch, _refineParams.scale, _refineParams.stepXY,
                         "refinedFused", _stream);
    if(_refineParams.exportIntermediateNormalMaps)
        computeAndWriteNormalMap(tile, _refinedDepthSimMap.as<float>(), "refinedFused");

    if(_refineParams.useColorOptimization && _refineParams.optimizationNbIterations > 0)
        optimizeDepthSimMap(tile);
    else
    {
        AVDM_LOG_INFO(tile << "Color optimize depth/sim map disabled.");
        AVDM_HIP_CHECK(hipMemcpyAsync(_optimizedDepthSimMap.ptr(), _refinedDepthSimMap.ptr(), (size_t)_mapPitch * _mapHeight, hipMemcpyDeviceToDevice, _stream));
    }
    if(_refineParams.exportIntermediateNormalMaps)
        computeAndWriteNormalMap(tile, _optimizedDepthSimMap.as<float>());
    AVDM_LOG_INFO(tile << "Refine depth/sim map done.");
}

void Refine::refineAndFuseDepthSimMap(const Tile& tile)
{
    AVDM_LOG_INFO(tile << "Refine and fuse depth/sim map volume.");
    const ROI downscaledRoi = downscaleROI(tile.roi, float(_refineParams.scale * _refineParams.stepXY));
    const avdm_roi_t roi = {{downscaledRoi.x.begin, downscaledRoi.x.end}, {downscaledRoi.y.begin, downscaledRoi.y.end}};
    const avdm_refine_params_t rp = _refineParams.toAvdm();
    const avdm_range_t depthRange = {0u, (unsigned)_volZ};
    const int X = (int)downscaledRoi.width(), Y = (int)downscaledRoi.height();
    if(X > _mapWidth || Y > _mapHeight)
        AVDM_THROW_ERROR(tile << "tile does not fit the Refine buffers.");
    const long long pitchY = (long long)X * _volPitchX;

    avdmCheck(avdm_volume_initialize_f16(_volumeRefineSim.ptr(), pitchY, _volPitchX, X, Y, _volPitchX / 2, 0.f, _stream), "avdm_volume_initialize_f16");

    const avdm_camera_t& rcCam = _deviceCache.requestCameraParams(tile.rc, _refineParams.scale, _mp);
    const DeviceMipmapImage& rcMip = _deviceCache.requestMipmapImage(tile.rc, _mp);
    for(std::size_t tci = 0; tci < tile.refineTCams.size(); ++tci)
    {
        const int tc = tile.refineTCams.at(tci);
        const avdm_camera_t& tcCam = _deviceCache.requestCameraParams(tc, _refineParams.scale, _mp);
        const DeviceMipmapImage& tcMip = _deviceCache.requestMipmapImage(tc, _mp);
        AVDM_LOG_DEBUG(tile << "Refine similarity volume:" << std::endl
                            << "\t- rc: " << tile.rc << std::endl
                            << "\t- tc: " << tc << " (" << (tci + 1) << "/" << tile.refineTCams.size() << ")" << std::endl
                            << "\t- tile range x: [" << downscaledRoi.x.begin << " - " << downscaledRoi.x.end << "]" << std::endl
                            << "\t- tile range y: [" << downscaledRoi.y.begin << " - " << downscaledRoi.y.end << "]" << std::endl);
        avdmCheck(avdm_volume_refine_similarity(_volumeRefineSim.ptr(), pitchY, _volPitchX, _volZ, _sgmDepthPixSizeMap.as<float>(), _mapPitch,
                                                _refineParams.useSgmNormalMap ? _sgmNormalMap.as<float>() : nullptr, _mapWidth * 12, &rcCam, &tcCam,
                                                &rcMip.pyramid(), &tcMip.pyramid(), &rp, depthRange, roi, _stream),
                  "avdm_volume_refine_similarity");
    }
    avdmCheck(avdm_volume_refine_best_depth(_refinedDepthSimMap.as<float>(), _mapPitch, _sgmDepthPixSizeMap.as<float>(), _mapPitch, _volumeRefineSim.ptr(),
                                            pitchY, _volPitchX, _volZ, &rp, roi, _stream),
              "avdm_volume_refine_best_depth");
    if(_refineParams.exportIntermediateCrossVolumes || _refineParams.exportIntermediateVolume9pCsv)
    { // Refine.cpp:235, :291-352 (exportVolumeInformation: its early return, too, leaves the topographic cut out)
        const int tileBeginX = tile.nbTiles > 1 ? (int)tile.roi.x.begin : -1, tileBeginY = tile.nbTiles > 1 ? (int)tile.roi.y.begin : -1;
        if(_refineParams.exportIntermediateCrossVolumes || _refineParams.exportIntermediateTopographicCutVolumes)
        {
            // the tile's extent (the reference takes the centre of the ALLOCATED volume, which is the tile's for every full-size tile; for
            // a smaller border tile it samples cells no kernel of the tile wrote)
            const HostVolume vol = downloadVolume(_volumeRefineSim.ptr(), true, pitchY, _volPitchX, X, Y, _volZ, _stream);
            Float2Tile dps;
            dps.allocate(X, Y);
            AVDM_HIP_CHECK(hipMemcpy2DAsync(dps.data.data(), (size_t)X * 8, _sgmDepthPixSizeMap.ptr(), (size_t)_mapPitch, (size_t)X * 8, (size_t)Y,
                                            hipMemcpyDeviceToHost, _stream));
            AVDM_HIP_CHECK(hipStreamSynchronize(_stream));
            if(_refineParams.exportIntermediateCrossVolumes)
            {
                AVDM_LOG_INFO(tile << "Export similarity volume cross (afterRefine).");
                exportSimilarityVolumeCross(vol, dps, _mp, tile.rc, _refineParams,
                                            getFileNameFromIndex(_mp, tile.rc, EFileType::volumeCross, "_afterRefine", tileBeginX, tileBeginY), tile.roi);
            }
            if(_refineParams.exportIntermediateTopographicCutVolumes)
            {
                AVDM_LOG_INFO(tile << "Export similarity volume topographic cut (afterRefine).");
                exportSimilarityVolumeTopographicCut(vol, dps, _mp, tile.rc, _refineParams,
                                                     getFileNameFromIndex(_mp, tile.rc, EFileType::volumeTopographicCut, "_afterRefine", tileBeginX, tileBeginY),
                                                     tile.roi);
            }
        }
        if(_refineParams.exportIntermediateVolume9pCsv)
        {
            AVDM_LOG_INFO(tile << "Export similarity volume 9 points CSV (afterRefine).");
            exportSimilaritySamplesCSV(_volumeRefineSim.ptr(), true, pitchY, _volPitchX, _volZ, X, Y, "afterRefine",
                                       getFileNameFromIndex(_mp, tile.rc, EFileType::stats9p, "_refine", tileBeginX, tileBeginY), _stream);
        }
    }
    AVDM_LOG_INFO(tile << "Refine and fuse depth/sim map volume done.");
}

void Refine::optimizeDepthSimMap(const Tile& tile)
{
    AVDM_LOG_INFO(tile << "Color optimize depth/sim map.");
    const ROI downscaledRoi = downscaleROI(tile.roi, float(_refineParams.scale * _refineParams.stepXY));
    const avdm_roi_t roi = {{downscaledRoi.x.begin, downscaledRoi.x.end}, {downscaledRoi.y.begin, downscaledRoi.y.end}};
    const avdm_refine_params_t rp = _refineParams.toAvdm();
    const avdm_camera_t& rcCam = _deviceCache.requestCameraParams(tile.rc, _refineParams.scale, _mp);
    const DeviceMipmapImage& rcMip = _deviceCache.requestMipmapImage(tile.rc, _mp);
    // the temporary depth map is clamped to the TILE extent: the reference binds the whole allocated buffer as a texture
    // (deviceDepthSimilarityMap.cu:228-229), so its border pixels read texels no kernel of this tile wrote
    avdmCheck(avdm_depth_sim_map_optimize_gradient_descent(_optimizedDepthSimMap.as<float>(), _mapPitch, _optImgVariance.as<float>(), _mapWidth * 4,
                                                           _optTmpDepthMap.as<float>(), _mapWidth * 4, (int)downscaledRoi.width(), (int)downscaledRoi.height(), _sgmDepthPixSizeMap.as<float>(),
                                                           _mapPitch, _refinedDepthSimMap.as<float>(), _mapPitch, &rcCam, &rcMip.pyramid(), &rp, roi, _stream),
              "avdm_depth_sim_map_optimize_gradient_descent");
    AVDM_LOG_INFO(tile << "Color optimize depth/sim map done.");
}

void Refine::computeAndWriteNormalMap(const Tile& tile, const float* in_depthSimMap, const std::string& name)
{
    const ROI downscaledRoi = downscaleROI(tile.roi, float(_refineParams.scale * _refineParams.stepXY));
    const avdm_roi_t roi = {{downscaledRoi.x.begin, downscaledRoi.x.end}, {downscaledRoi.y.begin, downscaledRoi.y.end}};
    const avdm_camera_t& rcCam = _deviceCache.requestCameraParams(tile.rc, _refineParams.scale, _mp);
    AVDM_LOG_INFO(tile << "Refine compute normal map of view id: " << _mp.getViewId(tile.rc) << ", rc: " << tile.rc << " (" << (tile.rc + 1) << " / "
                       << _mp.ncams << ").");
    avdmCheck(avdm_depth_sim_map_compute_normal(_normalMap.as<float>(), _mapWidth * 12, in_depthSimMap, _mapPitch, &rcCam, _refineParams.stepXY, roi, _stream),
              "avdm_depth_sim_map_compute_normal");
    writeNormalMap(tile.rc, _mp, _tileParams, tile.roi, _normalMap.as<float>(), _mapWidth * 12, _refineParams.scale, _refineParams.stepXY, name, _stream);
}

} // namespace avdm_host
