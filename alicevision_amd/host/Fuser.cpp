// Fuser.cpp — see Fuser.hpp.  The reference decodes every neighbour depth map again for every camera (Fuser.cpp:182-187) and walks
// the pixels on one core per camera; here decoded maps stay in HBM (LRU), the host cores only decode / encode files, chunk by chunk.
#include "Fuser.hpp"

#include "log.hpp"
#include "png.hpp"

#include <avdm_fuse.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <exception>
#include <set>
#include <sys/stat.h>

namespace avdm_host {

namespace {
bool fileExists(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
void avdmFuseCheck(int status, const char* what)
{
    if(status != 0)
        throw std::runtime_error(std::string(what) + " failed (" + std::to_string(status) + "): " + avdm_last_error());
}
avdm_fuse_camera_t fuseCamera(const MultiViewParams& mp, int c)
{
    avdm_fuse_camera_t cam;
    std::copy_n(mp.camArr.at(c).m, 12, cam.P);
    std::copy_n(mp.iCamArr.at(c).m, 9, cam.iP);
    cam.C[0] = mp.CArr.at(c).x, cam.C[1] = mp.CArr.at(c).y, cam.C[2] = mp.CArr.at(c).z;
    cam.width = mp.getWidth(c), cam.height = mp.getHeight(c);
    return cam;
}
const int kChunk = 8; // reference cameras whose files are decoded / encoded together on the host cores
double secondsSince(std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); }
} // namespace

Fuser::Fuser(const MultiViewParams& mp, int deviceId, int maxDeviceMaps)
  : _mp(mp),
    _deviceId(deviceId)
{
    if(maxDeviceMaps <= 0)
    {
        const char* e = std::getenv("AVDM_FUSE_CACHE");
        maxDeviceMaps = e ? std::max(2, std::atoi(e)) : 64;
    }
    _maxDeviceMaps = (size_t)maxDeviceMaps;
    AVDM_HIP_CHECK(hipSetDevice(_deviceId));
    AVDM_HIP_CHECK(hipStreamCreateWithFlags(&_stream, hipStreamNonBlocking));
}

Fuser::~Fuser()
{
    if(_stream)
        (void)hipStreamDestroy(_stream);
}

void Fuser::upload(int cam, const FloatMap& map)
{
    auto it = _cache.find(cam);
    if(it != _cache.end())
        return;
    while(_cache.size() >= _maxDeviceMaps)
    {
        const int victim = _lru.back();
        _lru.pop_back();
        _cache.erase(victim);
    }
    auto dm = std::make_shared<DeviceMap>();
    dm->width = map.width, dm->height = map.height;
    if(!map.data.empty())
    {
        dm->buf.allocate(map.data.size() * sizeof(float));
        AVDM_HIP_CHECK(hipMemcpyAsync(dm->buf.ptr(), map.data.data(), map.data.size() * sizeof(float), hipMemcpyHostToDevice, _stream));
        AVDM_HIP_CHECK(hipStreamSynchronize(_stream)); // `map` is pageable and goes away
    }
    _lru.push_front(cam);
    _cache[cam] = {dm, _lru.begin()};
}

void Fuser::prefetch(const std::vector<int>& cams)
{
    std::vector<int> missing;
    for(const int c : cams)
        if(_cache.find(c) == _cache.end() && std::find(missing.begin(), missing.end(), c) == missing.end())
            missing.push_back(c);
    if(missing.empty())
        return;
    // one file after the other: the EXR codec spreads the blocks of a file over the host cores itself
    for(const int c : missing)
    {
        FloatMap map;
        readMap(c, _mp, EFileType::depthMap, map, 1, 1); // read depth map from the depthMapEstimation folder (Fuser.cpp:159, :186)
        upload(c, map);
    }
}

std::shared_ptr<Fuser::DeviceMap> Fuser::deviceDepthMap(int cam)
{
    auto it = _cache.find(cam);
    if(it == _cache.end())
    {
        prefetch({cam});
        it = _cache.find(cam);
    }
    _lru.splice(_lru.begin(), _lru, it->second.second);
    return it->second.first;
}

// Fuser.cpp:124-141
void Fuser::filterGroups(const std::vector<int>& cams, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, int nNearestCams)
{
    AVDM_LOG_INFO("Precomputing groups.");
    const auto t0 = std::chrono::steady_clock::now();
    AVDM_HIP_CHECK(hipSetDevice(_deviceId));
    for(size_t c0 = 0; c0 < cams.size(); c0 += kChunk)
    {
        const size_t c1 = std::min(cams.size(), c0 + kChunk);
        // depth maps this chunk touches, as far as the device cache holds them at once
        std::vector<int> needed;
        for(size_t c = c0; c < c1; ++c)
        {
            const int rc = cams[c];
            if(fileExists(getFileNameFromIndex(_mp, rc, EFileType::nmodMap)))
                continue;
            needed.push_back(rc);
            for(const int tc : _mp.findNearestCamsFromLandmarks(rc, nNearestCams))
                needed.push_back(tc);
        }
        std::sort(needed.begin(), needed.end());
        needed.erase(std::unique(needed.begin(), needed.end()), needed.end());
        if(needed.size() > _maxDeviceMaps)
            needed.resize(_maxDeviceMaps);
        prefetch(needed);
        for(size_t c = c0; c < c1; ++c)
            filterGroupsRC(cams[c], pixToleranceFactor, pixSizeBall, pixSizeBallWSP, nNearestCams);
    }
    AVDM_LOG_INFO("Groups of " << cams.size() << " camera(s) computed in " << secondsSince(t0) << " s.");
}

void Fuser::runGroupsKernel(int rc, const FloatMap& simMap, const std::vector<int>& tcams, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP)
{
    const int w = _mp.getWidth(rc), h = _mp.getHeight(rc);
    const std::shared_ptr<DeviceMap> depthMap = deviceDepthMap(rc);
    if(depthMap->width * depthMap->height != w * h || simMap.width * simMap.height != w * h)
        AVDM_THROW_ERROR("filterGroupsRC: bad image dimension for camera: " << _mp.getViewId(rc) << "\n"
                                                                            << "depthMap size: " << depthMap->width * depthMap->height
                                                                            << ", simMap size: " << simMap.width * simMap.height << ", width: " << w
                                                                            << ", height: " << h);
    if(_sim.bytes() < (size_t)w * h * sizeof(float))
        _sim.allocate((size_t)w * h * sizeof(float));
    if(_nmod.bytes() < (size_t)w * h)
        _nmod.allocate((size_t)w * h);
    const size_t scratchBytes = avdm_fuse_filter_groups_scratch_bytes(w, h);
    if(_scratch.bytes() < scratchBytes)
        _scratch.allocate(scratchBytes);
    AVDM_HIP_CHECK(hipMemcpyAsync(_sim.ptr(), simMap.data.data(), (size_t)w * h * sizeof(float), hipMemcpyHostToDevice, _stream));

    // the T cameras' depth maps: hold the shared pointers until the kernels are done (the cache may evict)
    std::vector<std::shared_ptr<DeviceMap>> held;
    std::vector<avdm_fuse_tc_t> tcs(tcams.size());
    for(size_t c = 0; c < tcams.size(); ++c)
    {
        const int tc = tcams[c];
        std::shared_ptr<DeviceMap> m = deviceDepthMap(tc);
        held.push_back(m);
        avdm_fuse_tc_t& t = tcs[c];
        t.cam = fuseCamera(_mp, tc);
        t.reserved = 0;
        // the map decides the loop bounds (Fuser.cpp:189-193: tcdepthMap.height() / width())
        t.cam.width = m->width, t.cam.height = m->height;
        t.depth = (m->width > 0 && m->height > 0) ? m->buf.as<float>() : nullptr;
        t.depth_pitch = m->width * (int)sizeof(float);
        if(t.depth != nullptr && (m->width != _mp.getWidth(tc) || m->height != _mp.getHeight(tc)))
            AVDM_THROW_ERROR("filterGroupsRC: depth map of camera " << _mp.getViewId(tc) << " is " << m->width << "x" << m->height << ", expected "
                                                                    << _mp.getWidth(tc) << "x" << _mp.getHeight(tc));
    }
    const avdm_fuse_camera_t rcCam = fuseCamera(_mp, rc);
    avdmFuseCheck(avdm_fuse_filter_groups(_nmod.as<unsigned char>(), w, depthMap->buf.as<float>(), w * (int)sizeof(float), _sim.as<float>(),
                                          w * (int)sizeof(float), &rcCam, (int)tcs.size(), tcs.data(), pixToleranceFactor, pixSizeBall, pixSizeBallWSP,
                                          _scratch.ptr(), _stream),
                  "avdm_fuse_filter_groups");
    AVDM_HIP_CHECK(hipStreamSynchronize(_stream)); // `held` and `simMap` may go away
}

// Fuser.cpp:144-231
bool Fuser::filterGroupsRC(int rc, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, int nNearestCams)
{
    const std::string nmodPath = getFileNameFromIndex(_mp, rc, EFileType::nmodMap);
    if(fileExists(nmodPath))
        return true;
    AVDM_HIP_CHECK(hipSetDevice(_deviceId));
    const int w = _mp.getWidth(rc), h = _mp.getHeight(rc);
    const std::vector<int> tcams = _mp.findNearestCamsFromLandmarks(rc, nNearestCams);
    {
        std::vector<int> all(tcams);
        all.push_back(rc);
        prefetch(all);
    }
    // read the similarity map from the depthMapEstimation folder
    FloatMap simMap;
    readMap(rc, _mp, EFileType::simMap, simMap, 1, 1);
    runGroupsKernel(rc, simMap, tcams, pixToleranceFactor, pixSizeBall, pixSizeBallWSP);
    std::vector<unsigned char> numOfModalsMap((size_t)w * h);
    AVDM_HIP_CHECK(hipMemcpyAsync(numOfModalsMap.data(), _nmod.ptr(), numOfModalsMap.size(), hipMemcpyDeviceToHost, _stream));
    AVDM_HIP_CHECK(hipStreamSynchronize(_stream));
    writePngGray8(nmodPath, w, h, numOfModalsMap.data());
    AVDM_LOG_DEBUG(rc << " solved.");
    return true;
}

void Fuser::filterGroupsAndDepthMaps(const std::vector<int>& cams, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, int nNearestCams,
                                     int minNumOfModals, int minNumOfModalsWSP2SSP)
{
    AVDM_LOG_INFO("Precomputing groups and filtering depth maps.");
    const auto t0 = std::chrono::steady_clock::now();
    AVDM_HIP_CHECK(hipSetDevice(_deviceId));
    struct Item
    {
        std::vector<int> tcams;
        FloatMap depth, sim;
        std::vector<unsigned char> nmod;
        bool nmodFromFile = false;
        int nw = 0, nh = 0;
    };
    for(size_t c0 = 0; c0 < cams.size(); c0 += kChunk)
    {
        const int n = (int)(std::min(cams.size(), c0 + kChunk) - c0);
        std::vector<Item> items(n);
        std::vector<int> needed;
        for(int i = 0; i < n; ++i)
        {
            const int rc = cams[c0 + i];
            items[i].nmodFromFile = fileExists(getFileNameFromIndex(_mp, rc, EFileType::nmodMap));
            needed.push_back(rc);
            if(!items[i].nmodFromFile)
            {
                items[i].tcams = _mp.findNearestCamsFromLandmarks(rc, nNearestCams);
                needed.insert(needed.end(), items[i].tcams.begin(), items[i].tcams.end());
            }
        }
        std::sort(needed.begin(), needed.end());
        needed.erase(std::unique(needed.begin(), needed.end()), needed.end());
        if(needed.size() > _maxDeviceMaps)
            needed.resize(_maxDeviceMaps);
        prefetch(needed);

        // similarity maps (and modal counts left by an earlier run) of the chunk; file after file, the EXR codec is parallel inside
        for(int i = 0; i < n; ++i)
        {
            const int rc = cams[c0 + i];
            readMap(rc, _mp, EFileType::simMap, items[i].sim, 1, 1);
            if(items[i].nmodFromFile)
                readPngGray8(getFileNameFromIndex(_mp, rc, EFileType::nmodMap), items[i].nw, items[i].nh, items[i].nmod);
        }

        for(int i = 0; i < n; ++i)
        {
            Item& it = items[i];
            const int rc = cams[c0 + i];
            const int w = _mp.getWidth(rc), h = _mp.getHeight(rc);
            const size_t bytes = (size_t)w * h * sizeof(float);
            if(!it.nmodFromFile)
            {
                runGroupsKernel(rc, it.sim, it.tcams, pixToleranceFactor, pixSizeBall, pixSizeBallWSP);
                it.nmod.resize((size_t)w * h);
                AVDM_HIP_CHECK(hipMemcpyAsync(it.nmod.data(), _nmod.ptr(), it.nmod.size(), hipMemcpyDeviceToHost, _stream));
            }
            else
            {
                const std::shared_ptr<DeviceMap> dm = deviceDepthMap(rc);
                if(dm->width != it.sim.width || dm->width != it.nw || dm->height != it.sim.height || dm->height != it.nh || dm->width != w || dm->height != h)
                    throw std::invalid_argument("depthMap, simMap and numOfModalsMap must have same size");
                if(_sim.bytes() < bytes)
                    _sim.allocate(bytes);
                if(_nmod.bytes() < (size_t)w * h)
                    _nmod.allocate((size_t)w * h);
                AVDM_HIP_CHECK(hipMemcpyAsync(_sim.ptr(), it.sim.data.data(), bytes, hipMemcpyHostToDevice, _stream));
                AVDM_HIP_CHECK(hipMemcpyAsync(_nmod.ptr(), it.nmod.data(), (size_t)w * h, hipMemcpyHostToDevice, _stream));
            }
            // second pass on a copy of the depth map: the cached one stays unfiltered for the other cameras
            const std::shared_ptr<DeviceMap> depthMap = deviceDepthMap(rc);
            if(_depthTmp.bytes() < bytes)
                _depthTmp.allocate(bytes);
            AVDM_HIP_CHECK(hipMemcpyAsync(_depthTmp.ptr(), depthMap->buf.ptr(), bytes, hipMemcpyDeviceToDevice, _stream));
            avdmFuseCheck(avdm_fuse_filter_depth_maps(_depthTmp.as<float>(), w * (int)sizeof(float), _sim.as<float>(), w * (int)sizeof(float),
                                                      _nmod.as<unsigned char>(), w, w, h, minNumOfModals, minNumOfModalsWSP2SSP, _stream),
                          "avdm_fuse_filter_depth_maps");
            it.depth.reshape(w, h);
            AVDM_HIP_CHECK(hipMemcpyAsync(it.depth.data.data(), _depthTmp.ptr(), bytes, hipMemcpyDeviceToHost, _stream));
            AVDM_HIP_CHECK(hipMemcpyAsync(it.sim.data.data(), _sim.ptr(), bytes, hipMemcpyDeviceToHost, _stream));
            AVDM_HIP_CHECK(hipStreamSynchronize(_stream));
        }

        // modal counts (Fuser.cpp:220-223) of the chunk: one deflate stream per file, so one file per core
        std::exception_ptr error;
#pragma omp parallel for schedule(dynamic, 1)
        for(int i = 0; i < n; ++i)
        {
            try
            {
                const int rc = cams[c0 + i];
                if(!items[i].nmodFromFile)
                    writePngGray8(getFileNameFromIndex(_mp, rc, EFileType::nmodMap), _mp.getWidth(rc), _mp.getHeight(rc), items[i].nmod.data());
            }
            catch(...)
            {
#pragma omp critical
                error = std::current_exception();
            }
        }
        if(error)
            std::rethrow_exception(error);
        // filtered maps (:296-297)
        for(int i = 0; i < n; ++i)
        {
            const int rc = cams[c0 + i];
            const ROI fullRoi(0, _mp.getWidth(rc), 0, _mp.getHeight(rc));
            const TileParams defaultTileParams;
            writeMap(rc, _mp, EFileType::depthMapFiltered, defaultTileParams, fullRoi, items[i].depth, 1, 1);
            writeMap(rc, _mp, EFileType::simMapFiltered, defaultTileParams, fullRoi, items[i].sim, 1, 1);
        }
    }
    AVDM_LOG_INFO("Groups computed and depth maps filtered for " << cams.size() << " camera(s) in " << secondsSince(t0) << " s.");
}

// Fuser.cpp:234-247
void Fuser::filterDepthMaps(const std::vector<int>& cams, int minNumOfModals, int minNumOfModalsWSP2SSP)
{
    AVDM_LOG_INFO("Filtering depth maps.");
    const auto t0 = std::chrono::steady_clock::now();
    AVDM_HIP_CHECK(hipSetDevice(_deviceId));
    struct Item
    {
        FloatMap depth, sim;
        std::vector<unsigned char> nmod;
        int nw = 0, nh = 0;
    };
    for(size_t c0 = 0; c0 < cams.size(); c0 += kChunk)
    {
        const int n = (int)(std::min(cams.size(), c0 + kChunk) - c0);
        std::vector<Item> items(n);
        // read depth / sim maps from the depthMapEstimation folder and the modal counts of the first pass (Fuser.cpp:258-264)
        for(int i = 0; i < n; ++i)
        {
            const int rc = cams[c0 + i];
            readMap(rc, _mp, EFileType::depthMap, items[i].depth, 1, 1);
            readMap(rc, _mp, EFileType::simMap, items[i].sim, 1, 1);
            readPngGray8(getFileNameFromIndex(_mp, rc, EFileType::nmodMap), items[i].nw, items[i].nh, items[i].nmod);
        }
        for(int i = 0; i < n; ++i)
        {
            Item& it = items[i];
            if(it.depth.width != it.sim.width || it.depth.width != it.nw || it.depth.height != it.sim.height || it.depth.height != it.nh)
                throw std::invalid_argument("depthMap, simMap and numOfModalsMap must have same size");
            const int w = it.depth.width, h = it.depth.height;
            const size_t bytes = (size_t)w * h * sizeof(float);
            if(_sim.bytes() < bytes)
                _sim.allocate(bytes);
            if(_scratch.bytes() < bytes)
                _scratch.allocate(bytes); // depth goes here: the cached copy must stay unfiltered for the other cameras
            if(_nmod.bytes() < (size_t)w * h)
                _nmod.allocate((size_t)w * h);
            AVDM_HIP_CHECK(hipMemcpyAsync(_scratch.ptr(), it.depth.data.data(), bytes, hipMemcpyHostToDevice, _stream));
            AVDM_HIP_CHECK(hipMemcpyAsync(_sim.ptr(), it.sim.data.data(), bytes, hipMemcpyHostToDevice, _stream));
            AVDM_HIP_CHECK(hipMemcpyAsync(_nmod.ptr(), it.nmod.data(), (size_t)w * h, hipMemcpyHostToDevice, _stream));
            avdmFuseCheck(avdm_fuse_filter_depth_maps(_scratch.as<float>(), w * (int)sizeof(float), _sim.as<float>(), w * (int)sizeof(float),
                                                      _nmod.as<unsigned char>(), w, w, h, minNumOfModals, minNumOfModalsWSP2SSP, _stream),
                          "avdm_fuse_filter_depth_maps");
            AVDM_HIP_CHECK(hipMemcpyAsync(it.depth.data.data(), _scratch.ptr(), bytes, hipMemcpyDeviceToHost, _stream));
            AVDM_HIP_CHECK(hipMemcpyAsync(it.sim.data.data(), _sim.ptr(), bytes, hipMemcpyDeviceToHost, _stream));
            AVDM_HIP_CHECK(hipStreamSynchronize(_stream));
        }
        // Fuser.cpp:296-297
        for(int i = 0; i < n; ++i)
        {
            const int rc = cams[c0 + i];
            const ROI fullRoi(0, _mp.getWidth(rc), 0, _mp.getHeight(rc));
            const TileParams defaultTileParams;
            writeMap(rc, _mp, EFileType::depthMapFiltered, defaultTileParams, fullRoi, items[i].depth, 1, 1);
            writeMap(rc, _mp, EFileType::simMapFiltered, defaultTileParams, fullRoi, items[i].sim, 1, 1);
        }
    }
    AVDM_LOG_INFO("Depth maps of " << cams.size() << " camera(s) filtered in " << secondsSince(t0) << " s.");
}

// Fuser.cpp:250-304
bool Fuser::filterDepthMapsRC(int rc, int minNumOfModals, int minNumOfModalsWSP2SSP)
{
    filterDepthMaps({rc}, minNumOfModals, minNumOfModalsWSP2SSP);
    return true;
}

} // namespace avdm_host
