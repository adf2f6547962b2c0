// device.cpp — see device.hpp.
#include "device.hpp"

#include "log.hpp"

namespace avdm_host {

DeviceStreamManager::DeviceStreamManager(int nbStreams)
{
    if(nbStreams < 1)
        nbStreams = 1;
    _streams.resize(nbStreams);
    for(auto& s : _streams)
        AVDM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
}
DeviceStreamManager::~DeviceStreamManager()
{
    for(auto& s : _streams)
        (void)hipStreamDestroy(s);
}

void DeviceMipmapImage::fill(const HostImage& img, int minDownscale, int maxDownscale, int filterMode, hipStream_t stream)
{
    avdm_pyramid_t p;
    avdmCheck(avdm_pyramid_layout(&p, img.width, img.height, minDownscale, maxDownscale, filterMode), "avdm_pyramid_layout");
    if(_buf.bytes() != (size_t)p.bytes)
        _buf.allocate((size_t)p.bytes);
    p.base = _buf.ptr();
    _pyr = p;
    const size_t imgBytes = (size_t)img.width * img.height * 16;
    DeviceBuffer rgba(imgBytes), scratch;
    if(minDownscale > 1)
        scratch.allocate((size_t)img.width * img.height * 8);
    AVDM_HIP_CHECK(hipMemcpyAsync(rgba.ptr(), img.rgba.data(), imgBytes, hipMemcpyHostToDevice, stream));
    avdmCheck(avdm_pyramid_fill(&_pyr, rgba.as<float>(), img.width * 16, scratch.ptr(), stream), "avdm_pyramid_fill");
    AVDM_HIP_CHECK(hipStreamSynchronize(stream)); // the temporaries die here
}

DeviceCache::DeviceCache(int maxMipmapImages, int maxCameraParams, int filterMode)
  : _filterMode(filterMode),
    _mipmapCache(maxMipmapImages),
    _cameraParamCache(maxCameraParams)
{
    int dev = 0;
    AVDM_HIP_CHECK(hipGetDevice(&dev));
    AVDM_LOG_TRACE("Initialize device cache (device id: " << dev << "):" << std::endl
                                                          << "\t - # mipmap images: " << maxMipmapImages << std::endl
                                                          << "\t - # cameras parameters: " << maxCameraParams);
    _mipmaps.reserve(maxMipmapImages);
    for(int i = 0; i < maxMipmapImages; ++i)
        _mipmaps.push_back(std::make_unique<DeviceMipmapImage>());
    _cameraParams.resize(maxCameraParams);
}

void DeviceCache::addMipmapImage(int camId, int minDownscale, int maxDownscale, ImagesCache& imageCache, const MultiViewParams& mp, hipStream_t stream)
{
    int slot;
    if(!_mipmapCache.insert(camId, &slot))
    {
        AVDM_LOG_TRACE("Add mipmap image on device cache: already on cache (id: " << camId << ", view id: " << mp.getViewId(camId) << ").");
        return;
    }
    AVDM_LOG_TRACE("Add mipmap image on device cache (id: " << camId << ", view id: " << mp.getViewId(camId) << ").");
    const std::shared_ptr<const HostImage> img = imageCache.getImg_sync(camId);
    _mipmaps.at(slot)->fill(*img, minDownscale, maxDownscale, _filterMode, stream);
}

void DeviceCache::addCameraParams(int camId, int downscale, const MultiViewParams& mp)
{
    int slot;
    if(!_cameraParamCache.insert({camId, downscale}, &slot))
        return;
    AVDM_LOG_TRACE("Add camera parameters on device cache (id: " << camId << ", view id: " << mp.getViewId(camId) << ", downscale: " << downscale << ").");
    avdm_camera_fill(&_cameraParams.at(slot), mp.KArr[camId].m, mp.RArr[camId].m, &mp.CArr[camId].x, downscale);
}

const DeviceMipmapImage& DeviceCache::requestMipmapImage(int camId, const MultiViewParams& mp) const
{
    int slot;
    if(!_mipmapCache.find(camId, &slot))
        AVDM_THROW_ERROR("Request mipmap image on device cache: Not found (id: " << camId << ", view id: " << mp.getViewId(camId) << ").");
    return *_mipmaps.at(slot);
}

const avdm_camera_t& DeviceCache::requestCameraParams(int camId, int downscale, const MultiViewParams& mp) const
{
    int slot;
    if(!_cameraParamCache.find({camId, downscale}, &slot))
        AVDM_THROW_ERROR("Request camera parameters on device cache: Not found (id: " << camId << ", view id: " << mp.getViewId(camId)
                                                                                      << ", downscale: " << downscale << ").");
    return _cameraParams.at(slot);
}

void getDeviceMemoryInfo(double& availableMB, double& usedMB, double& totalMB)
{
    size_t iavail = 0, itotal = 0;
    AVDM_HIP_CHECK(hipMemGetInfo(&iavail, &itotal));
    availableMB = double(iavail) / (1024.0 * 1024.0);
    totalMB = double(itotal) / (1024.0 * 1024.0);
    usedMB = double(itotal - iavail) / (1024.0 * 1024.0);
}

void logDeviceMemoryInfo()
{
    double availableMB, usedMB, totalMB;
    getDeviceMemoryInfo(availableMB, usedMB, totalMB);
    int dev = 0;
    (void)hipGetDevice(&dev);
    AVDM_LOG_INFO("Device memory (device id: " << dev << "):" << std::endl
                                               << "\t- used: " << usedMB << " MB" << std::endl
                                               << "\t- available: " << availableMB << " MB" << std::endl
                                               << "\t- total: " << totalMB << " MB");
}

} // namespace avdm_host
