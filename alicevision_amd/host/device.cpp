// device.cpp — see device.hpp.
#include "device.hpp"

#include <algorithm>
#include <cstdlib>
#include <chrono>

#include <omp.h>

#include "jpeg.hpp"
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
DeviceStreamManager::~DeviceStreamManager() { destroy(); }
void DeviceStreamManager::destroy()
{
    for(auto& s : _streams)
    {
        (void)avdm_stream_release(s); // the library's per-stream block (optimisation point maps, resize tap tables) goes with the stream
        (void)hipStreamDestroy(s);
    }
    _streams.clear();
}

void decodeJpegToLinearRgba(const JpegImage& jpeg, float* rgba_d, hipStream_t stream)
{
    const int nc = (int)jpeg.components.size();
    if(jpeg.exifOrientation == 2 || jpeg.exifOrientation == 4 || jpeg.exifOrientation == 5 || jpeg.exifOrientation == 7)
        AVDM_LOG_WARNING("JPEG with a mirrored EXIF orientation (" << jpeg.exifOrientation << "): the reference mirrors it back (image/io.cpp:752-761); not done here.");
    avdm_jpeg_component_t comps[3] = {};
    std::vector<DeviceBuffer> coefs((size_t)nc);
    for(int i = 0; i < nc; ++i)
    {
        const JpegComponent& c = jpeg.components[(size_t)i];
        coefs[(size_t)i].allocate(c.coef.size() * sizeof(int16_t));
        AVDM_HIP_CHECK(hipMemcpyAsync(coefs[(size_t)i].ptr(), c.coef.data(), c.coef.size() * sizeof(int16_t), hipMemcpyHostToDevice, stream));
        comps[i].coef = coefs[(size_t)i].as<int16_t>();
        comps[i].blocks_w = c.blocksW, comps[i].blocks_h = c.blocksH, comps[i].width = c.width, comps[i].height = c.height;
        comps[i].h_samp = c.h, comps[i].v_samp = c.v;
        for(int k = 0; k < 64; ++k)
            comps[i].quant[k] = jpeg.quant[c.tq][k];
    }
    DeviceBuffer scratch(avdm_image_decode_jpeg_scratch_bytes(comps, nc)), rgb((size_t)jpeg.width * jpeg.height * 3);
    avdmCheck(avdm_image_decode_jpeg(rgb.as<uint8_t>(), jpeg.width * 3, jpeg.width, jpeg.height, comps, nc, jpeg.hmax, jpeg.vmax, jpeg.storedAsRgb() ? 0 : 1,
                                     scratch.ptr(), stream),
              "avdm_image_decode_jpeg");
    avdmCheck(avdm_image_decode_integer(rgba_d, jpeg.width * 16, rgb.ptr(), jpeg.width * 3, jpeg.width, jpeg.height, 3, 8, 1, stream), "avdm_image_decode_integer");
    AVDM_HIP_CHECK(hipStreamSynchronize(stream));
}

void DeviceMipmapImage::fill(const HostImage& img, int minDownscale, int maxDownscale, int filterMode, hipStream_t stream, DeviceBuffer* staging)
{
    avdm_pyramid_t p;
    avdmCheck(avdm_pyramid_layout(&p, img.width, img.height, minDownscale, maxDownscale, filterMode), "avdm_pyramid_layout");
    if(_buf.bytes() != (size_t)p.bytes)
        _buf.allocate((size_t)p.bytes);
    p.base = _buf.ptr();
    _pyr = p;
    const int srcW = img.srcWidth > 0 ? img.srcWidth : img.width, srcH = img.srcHeight > 0 ? img.srcHeight : img.height;
    const size_t srcBytes = (size_t)srcW * srcH * 16, imgBytes = (size_t)img.width * img.height * 16;
    DeviceBuffer rgbaOwn, resized, scratch, samples;
    if(staging == nullptr)
        rgbaOwn.allocate(srcBytes);
    else if(staging->bytes() < srcBytes)
        staging->allocate(srcBytes);
    const DeviceBuffer& rgba = staging != nullptr ? *staging : rgbaOwn;
    if(minDownscale > 1)
        scratch.allocate((size_t)img.width * img.height * 8);
    if(img.jpeg)
        decodeJpegToLinearRgba(*img.jpeg, rgba.as<float>(), stream);
    else if(img.exrLines)
    {
        // an OpenEXR file: its scan lines as stored (mapped file or inflated blocks) go up as they are; float RGBA is made on the device
        const ExrLines& x = *img.exrLines;
        samples.allocate(x.bytes);
        AVDM_HIP_CHECK(hipMemcpyAsync(samples.ptr(), x.lines, x.bytes, hipMemcpyHostToDevice, stream));
        avdmCheck(avdm_image_decode_exr_lines(rgba.as<float>(), srcW * 16, samples.ptr(), x.lineStride, srcW, srcH, x.chanOffset, x.chanType, stream),
                  "avdm_image_decode_exr_lines");
    }
    else if(!img.raw.empty())
    {
        // an integer file (PNG): upload the decoder's samples, make the linear float RGBA image on the device
        samples.allocate(img.raw.size());
        AVDM_HIP_CHECK(hipMemcpyAsync(samples.ptr(), img.raw.data(), img.raw.size(), hipMemcpyHostToDevice, stream));
        avdmCheck(avdm_image_decode_integer(rgba.as<float>(), srcW * 16, samples.ptr(), srcW * img.rawChannels * (img.rawBits / 8), srcW, srcH, img.rawChannels,
                                            img.rawBits, 1, stream),
                  "avdm_image_decode_integer");
    }
    else
        AVDM_HIP_CHECK(hipMemcpyAsync(rgba.ptr(), img.rgba.data(), srcBytes, hipMemcpyHostToDevice, stream));
    const float* processImage = rgba.as<float>();
    if(srcW != img.width || srcH != img.height)
    {
        // --downscale (fileIO.cpp:432-441): OpenImageIO's default resize, on the device
        resized.allocate(imgBytes);
        avdmCheck(avdm_image_resize(resized.as<float>(), img.width * 16, img.width, img.height, rgba.as<float>(), srcW * 16, srcW, srcH, stream),
                  "avdm_image_resize");
        processImage = resized.as<float>();
    }
    avdmCheck(avdm_pyramid_fill(&_pyr, processImage, img.width * 16, scratch.ptr(), stream), "avdm_pyramid_fill");
    AVDM_HIP_CHECK(hipStreamSynchronize(stream)); // the temporaries die here
}

void DeviceMipmapImage::copyFromPeer(const DeviceMipmapImage& src, int srcDevice, int dstDevice, hipStream_t stream)
{
    if(_buf.bytes() != src.bytes())
        _buf.allocate(src.bytes());
    _pyr = src.pyramid();
    _pyr.base = _buf.ptr();
    AVDM_HIP_CHECK(hipMemcpyPeerAsync(_buf.ptr(), dstDevice, src.pyramid().base, srcDevice, src.bytes(), stream));
}

bool PyramidExchange::publish(int camId, std::shared_ptr<const DeviceMipmapImage> img)
{
    {
        std::lock_guard<std::mutex> lock(_mutex);
        const int owner = ownerOf(camId);
        if(_residentBytes.at(owner) + img->bytes() > _budget)
        {
            _declined.insert(camId);
            ++nbDeclined;
            _published.notify_all();
            return false;
        }
        _residentBytes.at(owner) += img->bytes();
        _resident[camId] = std::move(img);
    }
    ++nbBuilt;
    _published.notify_all();
    return true;
}

void PyramidExchange::decline(int camId)
{
    {
        std::lock_guard<std::mutex> lock(_mutex);
        if(_resident.count(camId) != 0 || !_declined.insert(camId).second)
            return;
    }
    ++nbDeclined;
    _published.notify_all();
}

bool PyramidExchange::isDeclined(int camId)
{
    std::lock_guard<std::mutex> lock(_mutex);
    return _declined.count(camId) != 0;
}

size_t PyramidExchange::residentBytes(int worker)
{
    std::lock_guard<std::mutex> lock(_mutex);
    return _residentBytes.at(worker);
}

std::shared_ptr<const DeviceMipmapImage> PyramidExchange::find(int camId)
{
    std::lock_guard<std::mutex> lock(_mutex);
    const auto it = _resident.find(camId);
    return it == _resident.end() ? nullptr : it->second;
}

std::shared_ptr<const DeviceMipmapImage> PyramidExchange::await(int camId)
{
    std::unique_lock<std::mutex> lock(_mutex);
    _published.wait(lock, [&] { return _failure || _resident.count(camId) != 0 || _declined.count(camId) != 0; });
    const auto it = _resident.find(camId);
    if(it != _resident.end())
        return it->second;
    if(_declined.count(camId) != 0)
        return nullptr;
    std::rethrow_exception(_failure);
}

void PyramidExchange::fail(std::exception_ptr e)
{
    {
        std::lock_guard<std::mutex> lock(_mutex);
        if(!_failure)
            _failure = e;
    }
    _published.notify_all();
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
    _mipmaps.resize(maxMipmapImages);
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
    if(_exchange != nullptr)
    {
        const int owner = _exchange->ownerOf(camId);
        if(owner == _worker)
        {
            // mine: built once (normally by the pre-pass of DepthMapEstimator::compute), resident for the whole job; the slot aliases it
            if(!_exchange->find(camId) && !_exchange->isDeclined(camId))
                buildOwnedView(camId, minDownscale, maxDownscale, imageCache, mp, stream);
            if(const std::shared_ptr<const DeviceMipmapImage> mine = _exchange->find(camId))
            {
                _mipmaps.at(slot) = mine;
                return;
            }
            // declined (the exchange's residency budget is spent): an ordinary LRU entry of this device, below
        }
        else
        {
            const auto tA = std::chrono::steady_clock::now();
            const std::shared_ptr<const DeviceMipmapImage> src = _exchange->await(camId);
            const auto tB = std::chrono::steady_clock::now();
            _times.awaitOwner += std::chrono::duration<double>(tB - tA).count();
            if(src)
            {
                // another worker's: its pyramid copied over the fabric
                auto copy = std::make_shared<DeviceMipmapImage>();
                copy->copyFromPeer(*src, _exchange->deviceOf(owner), _exchange->deviceOf(_worker), stream);
                AVDM_HIP_CHECK(hipStreamSynchronize(stream)); // `src` may be released by its owner's exchange only after the copy has read it
                _times.peerCopy += std::chrono::duration<double>(std::chrono::steady_clock::now() - tB).count();
                ++_times.received;
                _times.bytesReceived += (long long)src->bytes();
                ++_exchange->nbCopied;
                _exchange->bytesCopied += (long long)src->bytes();
                _mipmaps.at(slot) = copy;
                return;
            }
        }
        // declined by its owner: decode and convert it here, like the reference does for every neighbour on every device
    }
    const auto tL = std::chrono::steady_clock::now();
    const std::shared_ptr<const HostImage> img = imageCache.getImg_sync(camId);
    auto own = std::make_shared<DeviceMipmapImage>();
    own->fill(*img, minDownscale, maxDownscale, _filterMode, stream);
    _mipmaps.at(slot) = own;
    _times.localBuild += std::chrono::duration<double>(std::chrono::steady_clock::now() - tL).count();
    ++_times.built;
}

void DeviceCache::addMipmapImages(const std::vector<int>& camIds, int minDownscale, int maxDownscale, ImagesCache& imageCache, const MultiViewParams& mp)
{
    struct Job
    {
        int camId, slot;
    };
    std::vector<Job> jobs;
    if(_exchange != nullptr || camIds.size() > _mipmaps.size())
    { // (more views than slots: a later view would evict an earlier one of the same list)
        hipStream_t s = nullptr;
        AVDM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        try
        {
            for(const int c : camIds)
                addMipmapImage(c, minDownscale, maxDownscale, imageCache, mp, s);
        }
        catch(...)
        {
            (void)avdm_stream_release(s);
            (void)hipStreamDestroy(s);
            throw;
        }
        (void)avdm_stream_release(s);
        (void)hipStreamDestroy(s);
        return;
    }
    for(const int c : camIds)
    {
        int slot;
        if(_mipmapCache.insert(c, &slot))
            jobs.push_back({c, slot});
    }
    if(jobs.empty())
        return;
    const auto tL = std::chrono::steady_clock::now();
    int dev = 0;
    AVDM_HIP_CHECK(hipGetDevice(&dev));
    // (measured, session r06_j/k: an uncompressed EXR is mapped and uploaded as it lies — 13 ms per 12 MP view from ONE thread; a team of 11
    // was slower than that thread alone: the uploads share one staging path and the page faults of the mappings one address space)
    int teamSize = 2;
    if(const char* e = getenv("AVDM_HOST_INGEST_THREADS"))
        teamSize = std::max(1, atoi(e));
    const int nbThreads = (int)std::min<size_t>(jobs.size(), (size_t)teamSize);
    while((int)_staging.size() < nbThreads)
        _staging.push_back(std::make_unique<DeviceBuffer>());
    std::exception_ptr error;
#pragma omp parallel num_threads(nbThreads)
    {
        const int tid = omp_get_thread_num();
        hipStream_t s = nullptr;
        bool ok = hipSetDevice(dev) == hipSuccess && hipStreamCreateWithFlags(&s, hipStreamNonBlocking) == hipSuccess;
        if(!ok)
        {
#pragma omp critical
            error = std::make_exception_ptr(std::runtime_error("addMipmapImages: cannot create a stream on the device"));
        }
#pragma omp for schedule(dynamic, 1)
        for(int k = 0; k < (int)jobs.size(); ++k)
        {
            if(!ok)
                continue;
            try
            {
                const Job& j = jobs[(size_t)k];
                AVDM_LOG_TRACE("Add mipmap image on device cache (id: " << j.camId << ", view id: " << mp.getViewId(j.camId) << ").");
                const std::shared_ptr<const HostImage> img = imageCache.getImg_sync(j.camId);
                // the pyramid the slot held is reused when nobody else holds it (same image size: no allocation at all)
                std::shared_ptr<DeviceMipmapImage> own;
                if(_mipmaps.at((size_t)j.slot) && _mipmaps.at((size_t)j.slot).use_count() == 1)
                    own = std::const_pointer_cast<DeviceMipmapImage>(_mipmaps.at((size_t)j.slot));
                else
                    own = std::make_shared<DeviceMipmapImage>();
                _mipmaps.at((size_t)j.slot).reset();
                own->fill(*img, minDownscale, maxDownscale, _filterMode, s, _staging.at((size_t)tid).get());
                _mipmaps.at((size_t)j.slot) = own; // (every job has a slot of its own)
            }
            catch(...)
            {
#pragma omp critical
                error = std::current_exception();
            }
        }
        if(s != nullptr)
        {
            (void)avdm_stream_release(s);
            (void)hipStreamDestroy(s);
        }
    }
    if(error)
        std::rethrow_exception(error);
    _times.localBuild += std::chrono::duration<double>(std::chrono::steady_clock::now() - tL).count();
    _times.built += (int)jobs.size();
}

void DeviceCache::buildOwnedView(int camId, int minDownscale, int maxDownscale, ImagesCache& imageCache, const MultiViewParams& mp, hipStream_t stream)
{
    if(_exchange == nullptr || _exchange->find(camId) || _exchange->isDeclined(camId))
        return;
    AVDM_LOG_TRACE("Build the pyramid of an owned view for the exchange (id: " << camId << ", view id: " << mp.getViewId(camId) << ").");
    const std::shared_ptr<const HostImage> img = imageCache.getImg_sync(camId);
    // its size is known before anything is allocated: a view that does not fit the residency budget is declined without being built
    avdm_pyramid_t layout;
    avdmCheck(avdm_pyramid_layout(&layout, img->width, img->height, minDownscale, maxDownscale, _filterMode), "avdm_pyramid_layout");
    if(_exchange->residentBytes(_worker) + (size_t)layout.bytes > _exchange->budgetBytes())
    {
        _exchange->decline(camId);
        return;
    }
    auto own = std::make_shared<DeviceMipmapImage>();
    own->fill(*img, minDownscale, maxDownscale, _filterMode, stream);
    (void)_exchange->publish(camId, own);
}

void DeviceCache::releaseImages()
{
    _mipmapCache = LRUCache<int>((int)_mipmaps.size());
    for(auto& m : _mipmaps)
        m.reset();
    _staging.clear();
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
