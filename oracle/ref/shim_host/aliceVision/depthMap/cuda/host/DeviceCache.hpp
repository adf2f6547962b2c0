// oracle/_ref: STAND-IN for aliceVision/depthMap/cuda/host/DeviceCache.hpp — the singleton Sgm.cpp / Refine.cpp ask for an image's mip-map
// pyramid and for the constant-memory slot of a camera at a downscale.  The reference's class loads and converts images through the
// image cache (OpenImageIO); here the test registers pyramids it built with the reference's own DeviceMipmapImage::fill and camera
// blocks it put into the constant-memory slots (tile_driver.cpp).  Test infrastructure only.
#pragma once
#include <map>
#include <stdexcept>
#include <utility>

#include <aliceVision/mvsUtils/MultiViewParams.hpp>
#include <aliceVision/depthMap/cuda/host/DeviceMipmapImage.hpp>
#include <aliceVision/depthMap/cuda/device/DeviceCameraParams.hpp>

namespace aliceVision {
namespace depthMap {
class DeviceCache
{
  public:
    static DeviceCache& getInstance()
    {
        static DeviceCache instance;
        return instance;
    }
    const DeviceMipmapImage& requestMipmapImage(int camId, const mvsUtils::MultiViewParams&)
    {
        const auto it = images.find(camId);
        if(it == images.end())
            throw std::runtime_error("DeviceCache stand-in: no mipmap image registered for this camera");
        return *it->second;
    }
    const int requestCameraParamsId(int camId, int downscale, const mvsUtils::MultiViewParams&)
    {
        const auto it = slots.find({camId, downscale});
        if(it == slots.end())
            throw std::runtime_error("DeviceCache stand-in: no camera block registered for this camera and downscale");
        return it->second;
    }
    std::map<int, const DeviceMipmapImage*> images;
    std::map<std::pair<int, int>, int> slots;
};
} // namespace depthMap
} // namespace aliceVision
