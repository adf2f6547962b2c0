// NormalMapEstimator.cpp — see NormalMapEstimator.hpp.
#include "NormalMapEstimator.hpp"

#include "depthMapUtils.hpp"
#include "device.hpp"
#include "log.hpp"

#include <chrono>
#include <sys/stat.h>

namespace avdm_host {

void NormalMapEstimator::compute(int deviceId, const std::vector<int>& cams)
{
    AVDM_HIP_CHECK(hipSetDevice(deviceId));
    DeviceCache deviceCache(0, 1, AVDM_FILTER_CUDA_FIXED8); // 0 mipmap image, 1 camera parameters (NormalMapEstimator.cpp:37)
    hipStream_t stream = nullptr;                          // the reference uses the default stream (:87)
    DeviceBuffer depthSimMap_d, normalMap_d;

    for(const int rc : cams)
    {
        const std::string normalMapFilepath = getFileNameFromIndex(_mp, rc, EFileType::normalMapFiltered);
        struct stat st;
        if(::stat(normalMapFilepath.c_str(), &st) == 0)
            continue;
        const auto t0 = std::chrono::steady_clock::now();
        AVDM_LOG_INFO("Compute normal map (rc: " << rc << ")");

        // R camera parameters, no additional downscale: we are working at input depth map resolution (:50-55)
        deviceCache.addCameraParams(rc, 1, _mp);
        const avdm_camera_t& rcCamera = deviceCache.requestCameraParams(rc, 1, _mp);

        FloatMap in_depthMap;
        readMap(rc, _mp, EFileType::depthMapFiltered, in_depthMap, 1, 1);
        const int width = in_depthMap.width, height = in_depthMap.height;
        if(width <= 0 || height <= 0)
            AVDM_THROW_ERROR("Cannot read the filtered depth map of camera " << _mp.getViewId(rc));

        const TileParams tileParams; // default tile parameters, no tiles
        const ROI roi(0, _mp.getWidth(rc), 0, _mp.getHeight(rc));
        if((int)roi.width() != width || (int)roi.height() != height)
            AVDM_THROW_ERROR("Filtered depth map of camera " << _mp.getViewId(rc) << " is " << width << "x" << height << ", expected " << roi.width() << "x"
                                                             << roi.height());

        // depth map -> depth/sim map in device memory; the similarity is not used by the normal computation (:72-83)
        std::vector<float> depthSim((size_t)width * height * 2);
        for(size_t i = 0; i < (size_t)width * height; ++i)
        {
            depthSim[2 * i] = in_depthMap.data[i];
            depthSim[2 * i + 1] = 1.f;
        }
        const int inPitch = width * 8, outPitch = width * 12;
        if(depthSimMap_d.bytes() < depthSim.size() * sizeof(float))
            depthSimMap_d.allocate(depthSim.size() * sizeof(float));
        if(normalMap_d.bytes() < (size_t)outPitch * height)
            normalMap_d.allocate((size_t)outPitch * height);
        AVDM_HIP_CHECK(hipMemcpyAsync(depthSimMap_d.ptr(), depthSim.data(), depthSim.size() * sizeof(float), hipMemcpyHostToDevice, stream));

        avdm_roi_t aroi;
        aroi.x.begin = roi.x.begin, aroi.x.end = roi.x.end, aroi.y.begin = roi.y.begin, aroi.y.end = roi.y.end;
        avdmCheck(avdm_depth_sim_map_compute_normal(normalMap_d.as<float>(), outPitch, depthSimMap_d.as<float>(), inPitch, &rcCamera, 1 /*step*/, aroi, stream),
                  "avdm_depth_sim_map_compute_normal");

        // writeNormalMapFiltered (depthMapUtils.cpp:185-195)
        std::vector<float> rgb((size_t)width * height * 3);
        AVDM_HIP_CHECK(hipMemcpyAsync(rgb.data(), normalMap_d.ptr(), rgb.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
        AVDM_HIP_CHECK(hipStreamSynchronize(stream));
        writeMap3(rc, _mp, EFileType::normalMapFiltered, tileParams, roi, rgb, width, height, 1, 1);

        const double elapsedMs = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() * 1e-3;
        AVDM_LOG_INFO("Compute normal map (rc: " << rc << ") done in: " << elapsedMs << " ms.");
    }
}

} // namespace avdm_host
