// NormalMapEstimator.cpp — see NormalMapEstimator.hpp.
#include "NormalMapEstimator.hpp"

#include "depthMapUtils.hpp"
#include "device.hpp"
#include "log.hpp"

#include <algorithm>
#include <chrono>
#include <exception>
#include <sys/stat.h>

namespace avdm_host {

void NormalMapEstimator::compute(int deviceId, const std::vector<int>& cams)
{
    AVDM_HIP_CHECK(hipSetDevice(deviceId));
    DeviceCache deviceCache(0, 1, AVDM_FILTER_CUDA_FIXED8); // 0 mipmap image, 1 camera parameters (NormalMapEstimator.cpp:37)
    hipStream_t stream = nullptr;                          // the reference uses the default stream (:87)
    DeviceBuffer depthSimMap_d, normalMap_d;

    // cameras whose normal map is missing (:44-46), a few at a time; the EXR codec spreads the blocks of a file over the host cores
    std::vector<int> todo;
    for(const int rc : cams)
    {
        struct stat st;
        if(::stat(getFileNameFromIndex(_mp, rc, EFileType::normalMapFiltered).c_str(), &st) != 0)
            todo.push_back(rc);
    }
    const int kChunk = 8;
    for(size_t c0 = 0; c0 < todo.size(); c0 += kChunk)
    {
        const int n = (int)(std::min(todo.size(), c0 + kChunk) - c0);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<FloatMap> depthMaps(n);
        std::vector<std::vector<float>> normals(n);
        for(int i = 0; i < n; ++i)
            readMap(todo[c0 + i], _mp, EFileType::depthMapFiltered, depthMaps[i], 1, 1); // read input depth map (:58-60)

        for(int i = 0; i < n; ++i)
        {
            const int rc = todo[c0 + i];
            AVDM_LOG_INFO("Compute normal map (rc: " << rc << ")");
            // R camera parameters, no additional downscale: we are working at input depth map resolution (:50-55)
            deviceCache.addCameraParams(rc, 1, _mp);
            const avdm_camera_t& rcCamera = deviceCache.requestCameraParams(rc, 1, _mp);

            const FloatMap& in_depthMap = depthMaps[i];
            const int width = in_depthMap.width, height = in_depthMap.height;
            if(width <= 0 || height <= 0)
                AVDM_THROW_ERROR("Cannot read the filtered depth map of camera " << _mp.getViewId(rc));
            const ROI roi(0, _mp.getWidth(rc), 0, _mp.getHeight(rc)); // fullsize roi
            if((int)roi.width() != width || (int)roi.height() != height)
                AVDM_THROW_ERROR("Filtered depth map of camera " << _mp.getViewId(rc) << " is " << width << "x" << height << ", expected " << roi.width()
                                                                 << "x" << roi.height());

            // depth map -> depth/sim map in device memory; the similarity is not used by the normal computation (:72-83)
            std::vector<float> depthSim((size_t)width * height * 2);
            for(size_t k = 0; k < (size_t)width * height; ++k)
            {
                depthSim[2 * k] = in_depthMap.data[k];
                depthSim[2 * k + 1] = 1.f;
            }
            const int inPitch = width * 8, outPitch = width * 12;
            if(depthSimMap_d.bytes() < depthSim.size() * sizeof(float))
                depthSimMap_d.allocate(depthSim.size() * sizeof(float));
            if(normalMap_d.bytes() < (size_t)outPitch * height)
                normalMap_d.allocate((size_t)outPitch * height);
            AVDM_HIP_CHECK(hipMemcpyAsync(depthSimMap_d.ptr(), depthSim.data(), depthSim.size() * sizeof(float), hipMemcpyHostToDevice, stream));

            avdm_roi_t aroi;
            aroi.x.begin = roi.x.begin, aroi.x.end = roi.x.end, aroi.y.begin = roi.y.begin, aroi.y.end = roi.y.end;
            avdmCheck(avdm_depth_sim_map_compute_normal(normalMap_d.as<float>(), outPitch, depthSimMap_d.as<float>(), inPitch, &rcCamera, 1 /*step*/, aroi,
                                                        stream),
                      "avdm_depth_sim_map_compute_normal");
            normals[i].resize((size_t)width * height * 3);
            AVDM_HIP_CHECK(hipMemcpyAsync(normals[i].data(), normalMap_d.ptr(), normals[i].size() * sizeof(float), hipMemcpyDeviceToHost, stream));
            AVDM_HIP_CHECK(hipStreamSynchronize(stream));
        }

        // writeNormalMapFiltered (depthMapUtils.cpp:185-195)
        for(int i = 0; i < n; ++i)
        {
            const int rc = todo[c0 + i];
            const TileParams tileParams; // default tile parameters, no tiles
            const ROI roi(0, _mp.getWidth(rc), 0, _mp.getHeight(rc));
            writeMap3(rc, _mp, EFileType::normalMapFiltered, tileParams, roi, normals[i], depthMaps[i].width, depthMaps[i].height, 1, 1);
        }
        const double elapsedMs = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() * 1e-3;
        AVDM_LOG_INFO("Compute normal maps of " << n << " camera(s) done in: " << elapsedMs << " ms.");
    }
}

} // namespace avdm_host
