// DepthMapEstimator.cpp — see DepthMapEstimator.hpp.
//
// Scheduling differences from the reference, results unchanged:
//   * the tiles of a batch are processed in groups of `nbStreams` tiles, each tile on its own stream with its own Sgm / Refine
//     buffers; inside a group the work is phased — (0) depth lists of all tiles on the host cores in parallel (the reference
//     computes them one by one between asynchronous launches, SgmDepthList.cpp), (A) similarity volumes per stream,
//     (B) ONE batched path-aggregation launch per path over all tiles of the group (the recurrence only parallelises over
//     columns), (C) best depth, Refine and the device-to-host copy per stream;
//   * no 100-slot constant-memory limit on camera parameters (they are kernel arguments).
#include "DepthMapEstimator.hpp"

#include "Refine.hpp"
#include "Sgm.hpp"
#include "SgmDepthList.hpp"
#include "depthMapUtils.hpp"
#include "device.hpp"
#include "log.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <future>
#include <memory>

namespace avdm_host {

// mvsUtils/TileParams.cpp:15-61
void getTileRoiList(const TileParams& tileParams, int imageWidth, int imageHeight, int maxDownscale, std::vector<ROI>& out_tileRoiList)
{
    if(hasOnlyOneTile(tileParams, imageWidth, imageHeight))
    {
        out_tileRoiList.emplace_back(0, imageWidth, 0, imageHeight);
        return;
    }
    const int maxEffectiveTileWidth = tileParams.bufferWidth - 2 * tileParams.padding;
    const int maxEffectiveTileHeight = tileParams.bufferHeight - 2 * tileParams.padding;
    const int nbTileSideX = divideRoundUp(imageWidth, maxEffectiveTileWidth);
    const int nbTileSideY = divideRoundUp(imageHeight, maxEffectiveTileHeight);
    out_tileRoiList.resize((size_t)nbTileSideX * nbTileSideY);
    const int downscaledImageWidth = divideRoundUp(imageWidth, maxDownscale);
    const int downscaledImageHeight = divideRoundUp(imageHeight, maxDownscale);
    const int effectiveTileWidth = divideRoundUp(downscaledImageWidth, nbTileSideX) * maxDownscale;
    const int effectiveTileHeight = divideRoundUp(downscaledImageHeight, nbTileSideY) * maxDownscale;
    for(int i = 0; i < nbTileSideX; ++i)
    {
        const int beginX = i * effectiveTileWidth;
        const int endX = std::min((i + 1) * effectiveTileWidth + tileParams.padding, imageWidth);
        for(int j = 0; j < nbTileSideY; ++j)
        {
            const int beginY = j * effectiveTileHeight;
            const int endY = std::min((j + 1) * effectiveTileHeight + tileParams.padding, imageHeight);
            out_tileRoiList.at((size_t)i * nbTileSideY + j) = ROI(beginX, endX, beginY, endY);
        }
    }
}

namespace {

// TileParams.cpp:63-118
void logTileRoiList(const TileParams& tileParams, int imageWidth, int imageHeight, int maxDownscale, const std::vector<ROI>& in_tileRoiList)
{
    std::ostringstream ostr;
    ostr << "Tiling information: " << std::endl
         << "\t- parameters: " << std::endl
         << "\t      - buffer width:  " << tileParams.bufferWidth << " px" << std::endl
         << "\t      - buffer height: " << tileParams.bufferHeight << " px" << std::endl
         << "\t      - padding: " << tileParams.padding << " px" << std::endl
         << "\t- maximum downscale:  " << maxDownscale << std::endl
         << "\t- maximum image width:  " << imageWidth << " px" << std::endl
         << "\t- maximum image height: " << imageHeight << " px" << std::endl;
    if(hasOnlyOneTile(tileParams, imageWidth, imageHeight))
    {
        AVDM_LOG_INFO(ostr.str());
        AVDM_LOG_INFO("Maximum image size is smaller than one tile, use only one tile.");
        return;
    }
    ostr << "\t- tile list: " << std::endl;
    for(size_t i = 0; i < in_tileRoiList.size(); ++i)
    {
        const ROI& roi = in_tileRoiList.at(i);
        ostr << "\t   - tile (" << (i + 1) << "/" << in_tileRoiList.size() << ") "
             << "size: " << roi.width() << "x" << roi.height() << " px, roi: [" << roi << "]" << std::endl;
    }
    AVDM_LOG_INFO(ostr.str());
}

int filterModeFromEnv()
{
    const char* e = std::getenv("AVDM_FILTER");
    if(e && std::string(e) == "exact")
        return AVDM_FILTER_EXACT;
    return AVDM_FILTER_CUDA_FIXED8; // the arithmetic of the CUDA texture unit the reference samples through
}

int maxStreamsFromEnv()
{
    const char* e = std::getenv("AVDM_MAX_STREAMS");
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : 24; // = AVDM_SGM_MAX_TILES: one batched aggregation launch covers a whole group
}

} // namespace

DepthMapEstimator::DepthMapEstimator(const MultiViewParams& mp, const TileParams& tileParams, const DepthMapParams& depthMapParams, const SgmParams& sgmParams,
                                     const RefineParams& refineParams)
  : _mp(mp),
    _tileParams(tileParams),
    _depthMapParams(depthMapParams),
    _sgmParams(sgmParams),
    _refineParams(refineParams)
{
    const int maxDownscale = std::max(_sgmParams.scale * _sgmParams.stepXY, _refineParams.scale * _refineParams.stepXY);
    getTileRoiList(_tileParams, _mp.getMaxImageWidth(), _mp.getMaxImageHeight(), maxDownscale, _tileRoiList);
    logTileRoiList(_tileParams, _mp.getMaxImageWidth(), _mp.getMaxImageHeight(), maxDownscale, _tileRoiList);
    AVDM_LOG_INFO("SGM parameters:" << std::endl << "\t- scale: " << _sgmParams.scale << std::endl << "\t- stepXY: " << _sgmParams.stepXY);
    AVDM_LOG_INFO("Refine parameters:" << std::endl << "\t- scale: " << _refineParams.scale << std::endl << "\t- stepXY: " << _refineParams.stepXY);
}

int DepthMapEstimator::getNbSimultaneousTiles() const
{
    // How many tiles may be in flight on this device — the question of the reference's getNbSimultaneousTiles (DepthMapEstimator.cpp:57-135),
    // answered from what THIS implementation allocates on an MI355X rather than from the CUDA-era estimate (80 % of the free memory, 1.5 x
    // the image for a mip pyramid, nothing for the scratch blocks):
    //   per tile slot (= stream)   the Sgm / Refine device buffers of the slot (sizes only: deviceMemoryConsumption);
    //                              the stream's scratch block of the library (avdm::stream_scratch): the two point maps of the colour
    //                              optimisation (2 x 16 B per pixel of the tile buffer) or the Refine outlier list (8 B per 4 (pixel, chunk)
    //                              pairs), whichever is larger — they follow each other on the stream;
    //                              its share of the batched aggregation's scratch (adaptive-P2 maps of the group);
    //   per camera of a batch      the Lab pyramids of R + maxTCams views at their EXACT size (avdm_pyramid_layout);
    //   what is free               as the device reports it NOW — after the multi-GPU pre-pass, i.e. without the pyramids this worker
    //                              publishes to the others (PyramidExchange, bounded by its own budget) — minus 10 % (at least 1 GiB) for code
    //                              objects, the runtime's own pools and the rounding of the allocations.
    // A batch holds whole cameras, so n tiles need the images of ceil(n / tiles per camera) cameras.  With 288 GB the answer is "as many as
    // one aggregation launch takes" (AVDM_SGM_MAX_TILES, applied by the caller) except for very large images or a nearly full device.
    const int nbTilesPerCamera = (int)_tileRoiList.size();
    const int minDs = std::min(_refineParams.scale, _sgmParams.scale), maxDs = std::max(_refineParams.scale, _sgmParams.scale) * 64;
    avdm_pyramid_t layout;
    avdmCheck(avdm_pyramid_layout(&layout, _mp.getMaxImageWidth(), _mp.getMaxImageHeight(), minDs, maxDs, filterModeFromEnv()), "avdm_pyramid_layout");
    const double MB = 1024.0 * 1024.0;
    const double mipmapCostMB = (double)layout.bytes / MB;
    const double rcCamsCostMB = (1 + _depthMapParams.maxTCams) * mipmapCostMB;

    const double sgmTileCostMB = Sgm::deviceMemoryConsumption(_tileParams, _sgmParams, !_depthMapParams.useRefine, _refineParams.useSgmNormalMap);
    const double refineTileCostMB = _depthMapParams.useRefine ? Refine::deviceMemoryConsumption(_tileParams, _refineParams) : 0.0;
    double scratchCostMB = 0.0;
    if(_depthMapParams.useRefine)
    {
        const int ds = _refineParams.scale * _refineParams.stepXY;
        const double px = (double)divideRoundUp(_tileParams.bufferWidth, ds) * divideRoundUp(_tileParams.bufferHeight, ds);
        const double pointMaps = _refineParams.useColorOptimization ? 2.0 * 16.0 * px : 0.0;
        // (the library's own figure: the capacity formula lives in ONE place)
        const double outlierList = (double)avdm_refine_similarity_scratch_bytes((size_t)px, _refineParams.halfNbDepths * 2 + 1);
        scratchCostMB = std::max(pointMaps, outlierList) / MB;
    }
    const double tileCostMB = sgmTileCostMB + refineTileCostMB + scratchCostMB; // (the aggregation's scratch is inside the Sgm figure)

    double availableMB, usedMB, totalMB;
    getDeviceMemoryInfo(availableMB, usedMB, totalMB);
    // a PROPORTIONAL margin with a floor (the reference keeps 20 %, DepthMapEstimator.cpp:112): 10 % of what is free, at least 1 GiB —
    // the allocations round up and the scratch blocks grow by reallocating, on a nearly full device a fixed 1 GiB admitted one tile too many
    const double deviceMemoryMB = std::min(availableMB * 0.9, availableMB - 1024.0);
    const double rcMinCostMB = rcCamsCostMB + tileCostMB;
    const int cap = maxStreamsFromEnv() * 64; // far beyond what the caller will take: the search below is bounded
    int out_nbSimultaneousTiles = 0;
    for(int n = 1; n <= cap; ++n)
    {
        if(n * tileCostMB + divideRoundUp(n, nbTilesPerCamera) * rcCamsCostMB > deviceMemoryMB)
            break;
        out_nbSimultaneousTiles = n;
    }

    AVDM_LOG_INFO("Device memory:" << std::endl
                                   << "\t- available: " << deviceMemoryMB << " MB (free now, minus 10 % / at least 1024 MB kept for the runtime)" << std::endl
                                   << "\t- requirement for the first tile: " << rcMinCostMB << " MB" << std::endl
                                   << "\t- # computation buffers per tile: " << tileCostMB << " MB"
                                   << " (Sgm: " << sgmTileCostMB << " MB"
                                   << ", Refine: " << refineTileCostMB << " MB, stream scratch: " << scratchCostMB << " MB)" << std::endl
                                   << "\t- # input images (R + " << _depthMapParams.maxTCams << " Ts): " << rcCamsCostMB
                                   << " MB (single mipmap image size: " << mipmapCostMB << " MB)");
    AVDM_LOG_INFO("Parallelization:" << std::endl
                                     << "\t- # tiles per image: " << nbTilesPerCamera << std::endl
                                     << "\t- # simultaneous depth maps computation: " << divideRoundUp(out_nbSimultaneousTiles, nbTilesPerCamera) << std::endl
                                     << "\t- # simultaneous tiles computation: " << out_nbSimultaneousTiles << (out_nbSimultaneousTiles == cap ? " (or more)" : ""));
    if(out_nbSimultaneousTiles < 1)
        AVDM_THROW_ERROR("Not enough GPU memory to compute a single tile.");
    return out_nbSimultaneousTiles;
}

void DepthMapEstimator::getTilesList(const std::vector<int>& cams, std::vector<Tile>& tiles) const
{
    const int nbTilesPerCamera = (int)_tileRoiList.size();
    tiles.reserve(cams.size() * nbTilesPerCamera);
    for(const int rc : cams)
    {
        const std::vector<int> tCams = _mp.findNearestCamsFromLandmarks(rc, _depthMapParams.maxTCams);
        const ROI rcImageRoi(Range(0, _mp.getWidth(rc)), Range(0, _mp.getHeight(rc)));
        for(int i = 0; i < nbTilesPerCamera; ++i)
        {
            Tile t;
            t.id = i;
            t.nbTiles = nbTilesPerCamera;
            t.rc = rc;
            t.roi = intersect(_tileRoiList.at(i), rcImageRoi);
            if(t.roi.isEmpty())
            {
                // this ROI cannot intersect the R camera ROI
            }
            else if(_depthMapParams.chooseTCamsPerTile)
            {
                t.sgmTCams = _mp.findTileNearestCams(rc, _sgmParams.maxTCamsPerTile, tCams, t.roi);
                if(_depthMapParams.useRefine)
                    t.refineTCams = _mp.findTileNearestCams(rc, _refineParams.maxTCamsPerTile, tCams, t.roi);
            }
            else
            {
                t.sgmTCams = tCams;
                t.refineTCams = tCams;
            }
            tiles.push_back(t);
        }
    }
}

void DepthMapEstimator::plan(const std::vector<int>& cams, std::vector<TilePlan>& out) const
{
    std::vector<Tile> tiles;
    getTilesList(cams, tiles);
    out.resize(tiles.size());
    std::exception_ptr error;
#pragma omp parallel for schedule(dynamic)
    for(int i = 0; i < (int)tiles.size(); ++i)
    {
        try
        {
            TilePlan& p = out[i];
            p.tile = tiles[i];
            if(p.tile.roi.isEmpty() || p.tile.sgmTCams.empty() || (_depthMapParams.useRefine && p.tile.refineTCams.empty()))
                continue;
            SgmDepthList dl(_mp, _sgmParams, p.tile);
            dl.computeListRc();
            if(dl.getDepths().empty())
                continue;
            dl.removeTcWithNoDepth(p.tile);
            p.depths = dl.getDepths();
            p.depthsTcLimits = dl.getDepthsTcLimits();
        }
        catch(...)
        {
#pragma omp critical
            error = std::current_exception();
        }
    }
    if(error)
        std::rethrow_exception(error);
}

std::vector<int> DepthMapEstimator::viewsNeeded(const std::vector<int>& cams) const
{
    std::vector<Tile> tiles;
    getTilesList(cams, tiles);
    std::vector<int> views;
    for(const Tile& t : tiles)
    {
        if(t.roi.isEmpty())
            continue;
        views.push_back(t.rc);
        views.insert(views.end(), t.sgmTCams.begin(), t.sgmTCams.end());
        views.insert(views.end(), t.refineTCams.begin(), t.refineTCams.end());
    }
    std::sort(views.begin(), views.end());
    views.erase(std::unique(views.begin(), views.end()), views.end());
    return views;
}

void DepthMapEstimator::compute(int deviceId, const std::vector<int>& cams) { computeImpl(deviceId, cams, 0, nullptr, nullptr); }

void DepthMapEstimator::computeShared(int worker, int deviceId, const std::vector<int>& cams, const std::vector<int>& allViews, PyramidExchange& exchange)
{
    computeImpl(deviceId, cams, worker, &allViews, &exchange);
}

void DepthMapEstimator::computeImpl(int deviceId, const std::vector<int>& cams, int worker, const std::vector<int>* allViews, PyramidExchange* exchange)
{
    AVDM_HIP_CHECK(hipSetDevice(deviceId));
    const auto tCompute0 = std::chrono::steady_clock::now();

    ImagesCache ic(_mp);

    const int minMipmapDownscale = std::min(_refineParams.scale, _sgmParams.scale);
    const int maxMipmapDownscale = std::max(_refineParams.scale, _sgmParams.scale) * (int)std::pow(2, 6); // 6 more levels

    // multi-GPU pre-pass: decode, convert and publish the views of the job this worker owns, before anything can wait for them (no
    // worker waits inside its own pre-pass, so the exchange cannot deadlock); a worker without R cameras still serves its views
    if(exchange != nullptr && allViews != nullptr)
    {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<int> mine;
        for(const int v : *allViews)
            if(exchange->ownerOf(v) == worker)
                mine.push_back(v);
        DeviceCache publisher(1, 1, filterModeFromEnv());
        publisher.setExchange(exchange, worker);
        hipStream_t s0;
        AVDM_HIP_CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
        try
        {
            // decode a few images ahead on the host cores, then convert them in order
            const int chunk = 8;
            for(size_t i0 = 0; i0 < mine.size(); i0 += chunk)
            {
                const int n = (int)std::min(mine.size() - i0, (size_t)chunk);
                std::exception_ptr loadError;
#pragma omp parallel for schedule(dynamic, 1)
                for(int k = 0; k < n; ++k)
                {
                    try
                    {
                        ic.getImg_sync(mine[i0 + k]);
                    }
                    catch(...)
                    {
#pragma omp critical
                        loadError = std::current_exception();
                    }
                }
                if(loadError)
                    std::rethrow_exception(loadError);
                for(int k = 0; k < n; ++k)
                    publisher.buildOwnedView(mine[i0 + k], minMipmapDownscale, maxMipmapDownscale, ic, _mp, s0);
            }
        }
        catch(...)
        {
            (void)avdm_stream_release(s0);
            (void)hipStreamDestroy(s0);
            throw;
        }
        (void)avdm_stream_release(s0);
        (void)hipStreamDestroy(s0);
        // whatever this worker owns and did not publish will never come: nobody may wait for it
        for(const int v : mine)
            if(!exchange->find(v))
                exchange->decline(v);
        AVDM_LOG_INFO("Worker " << worker << " (device " << deviceId << "): published " << mine.size() << " of the job's " << allViews->size() << " views in "
                                << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s ("
                                << (exchange->residentBytes(worker) >> 20) << " MB resident of a budget of " << (exchange->budgetBytes() >> 20)
                                << " MB; views declined so far: " << exchange->nbDeclined.load() << ").");
    }

    std::vector<Tile> tiles;
    getTilesList(cams, tiles);
    if(tiles.empty())
        return;

    const int nbStreams = std::min({getNbSimultaneousTiles(), static_cast<int>(tiles.size()), maxStreamsFromEnv()});

    const int nbTilesPerCamera = static_cast<int>(_tileRoiList.size());
    const int nbRcPerBatch = divideRoundUp(nbStreams, nbTilesPerCamera);
    const int nbTilesPerBatch = nbRcPerBatch * nbTilesPerCamera;
    const int nbMipmapImagesPerBatch = nbRcPerBatch * (1 + _depthMapParams.maxTCams);
    const int nbCamerasParamsPerBatch = nbMipmapImagesPerBatch * 3;

    struct PinnedRegistrations
    {
        std::vector<void*> ptrs;
        void add(void* p, size_t bytes)
        {
            if(hipHostRegister(p, bytes, hipHostRegisterDefault) == hipSuccess) // unpinned tiles still work (synchronous copies)
                ptrs.push_back(p);
            else
                (void)hipGetLastError();
        }
        void releaseAll()
        {
            for(void* p : ptrs)
                (void)hipHostUnregister(p);
            ptrs.clear();
        }
        ~PinnedRegistrations() { releaseAll(); }
    };
    // final depth/similarity map tiles in host memory, per camera of a batch (the map size of the last stage: Refine.cpp:17-18 / Sgm.cpp:22-23)
    const int finalDownscale = _depthMapParams.useRefine ? _refineParams.scale * _refineParams.stepXY : _sgmParams.scale * _sgmParams.stepXY;
    const int finalMapW = divideRoundUp(_tileParams.bufferWidth, finalDownscale), finalMapH = divideRoundUp(_tileParams.bufferHeight, finalDownscale);
    // two sets: while the tiles of batch b are merged and written by a background task, batch b + 1 computes into the other set
    // (the reference merges and writes between the batches, DepthMapEstimator.cpp:446-466, with the device idle)
    const int nbHostSets = 2;
    std::vector<std::vector<Float2Tile>> depthSimMapTileSets[nbHostSets];
    PinnedRegistrations pinned; // declared after the tile sets: unregistered before the vectors are freed
    std::vector<std::vector<std::pair<float, float>>> depthMinMaxTileSets[nbHostSets];
    // allocated (huge pages, depthMapUtils.hpp) and page-locked by a background task that starts HERE, beside the rest of the set-up and the first
    // batch's images; set 0 first — the first batch waits for it alone, set 1 is ready long before the second batch asks
    std::promise<void> hostSetPromise[nbHostSets];
    std::future<void> hostSetReady[nbHostSets];
    for(int s = 0; s < nbHostSets; ++s)
        hostSetReady[s] = hostSetPromise[s].get_future();
    std::future<void> hostTilesTask = std::async(std::launch::async, [&]() {
        int s = 0;
        try
        {
            AVDM_HIP_CHECK(hipSetDevice(deviceId));
            for(; s < nbHostSets; ++s)
            {
                const auto tSet0 = std::chrono::steady_clock::now();
                depthSimMapTileSets[s].resize(nbRcPerBatch);
                depthMinMaxTileSets[s].resize(nbRcPerBatch);
                for(int i = 0; i < nbRcPerBatch; ++i)
                {
                    depthSimMapTileSets[s][i].resize(nbTilesPerCamera);
                    depthMinMaxTileSets[s][i].resize(nbTilesPerCamera);
                    for(int j = 0; j < nbTilesPerCamera; ++j)
                    {
                        depthSimMapTileSets[s][i][j].allocate(finalMapW, finalMapH);
                        // page-lock the result tiles (CudaHostMemoryHeap is pinned memory in the reference): the device-to-host copies of a
                        // group then run asynchronously on the tile streams instead of blocking the host thread that feeds them
                        auto& v = depthSimMapTileSets[s][i][j].data;
                        pinned.add(v.data(), v.size() * sizeof(float));
                    }
                }
                hostSetPromise[s].set_value();
                AVDM_LOG_INFO("Worker " << worker << " (device " << deviceId << "): result tiles of set " << s << " (" << nbRcPerBatch * nbTilesPerCamera << " x "
                                        << ((size_t)finalMapW * finalMapH * 8 >> 10) << " KiB) allocated and page-locked in "
                                        << std::chrono::duration<double>(std::chrono::steady_clock::now() - tSet0).count() << " s (background).");
            }
        }
        catch(...)
        {
            for(; s < nbHostSets; ++s)
                hostSetPromise[s].set_exception(std::current_exception());
        }
    });

    DeviceCache deviceCache(nbMipmapImagesPerBatch, nbCamerasParamsPerBatch, filterModeFromEnv());
    deviceCache.setExchange(exchange, worker);

    // the views of batch b, in the order its tiles first use them
    auto viewsOfBatch = [&](int b) {
        std::vector<int> views;
        auto want = [&](int c) {
            if(std::find(views.begin(), views.end(), c) == views.end())
                views.push_back(c);
        };
        const int i1 = std::min((b + 1) * nbTilesPerBatch, static_cast<int>(tiles.size()));
        for(int i = b * nbTilesPerBatch; i < i1; ++i)
        {
            const Tile& tile = tiles.at(i);
            if(tile.roi.isEmpty())
                continue;
            want(tile.rc);
            for(const int tc : tile.sgmTCams)
                want(tc);
            if(_depthMapParams.useRefine)
                for(const int tc : tile.refineTCams)
                    want(tc);
        }
        return views;
    };
    // AVDM_HOST_INGEST=serial: rounds 1-5's form (parallel decode, then one view after the other uploaded and converted by this thread)
    const bool teamIngest = [] {
        const char* e = getenv("AVDM_HOST_INGEST");
        return !(e != nullptr && std::string(e) == "serial");
    }();
    // the first batch's views are decoded, uploaded and converted (DeviceCache::addMipmapImages: a team of host threads, one view each on a
    // stream of its own) BESIDE the rest of the set-up — streams, the tile slots' device buffers, the page-locked result tiles
    std::future<void> firstIngest;
    const auto tIngest0 = std::chrono::steady_clock::now();
    if(exchange == nullptr && teamIngest)
        firstIngest = std::async(std::launch::async, [&]() {
            AVDM_HIP_CHECK(hipSetDevice(deviceId));
            deviceCache.addMipmapImages(viewsOfBatch(0), minMipmapDownscale, maxMipmapDownscale, ic, _mp);
        });

    // (after the two background tasks have been started: creating 24 streams — a hardware queue each — takes ~0.1 s)
    DeviceStreamManager deviceStreamManager(nbStreams);

    // build the custom patch pattern (DepthMapEstimator.cpp:272-274; library state like the reference's constant memory)
    if(_sgmParams.useCustomPatchPattern || _refineParams.useCustomPatchPattern)
    {
        std::vector<avdm_patch_subpart_params_t> sub;
        for(const auto& sp : _depthMapParams.customPatchPattern.subpartsParams)
        {
            avdm_patch_subpart_params_t a;
            a.isCircle = sp.isCircle ? 1 : 0, a.level = sp.level, a.nbCoordinates = sp.nbCoordinates, a.radius = sp.radius, a.weight = sp.weight;
            sub.push_back(a);
        }
        avdmCheck(avdm_build_custom_patch_pattern((int)sub.size(), sub.data(), _depthMapParams.customPatchPattern.groupSubpartsPerLevel ? 1 : 0, nullptr),
                  "Cannot build custom patch pattern");
    }

    // ONE device allocation for the fixed-size buffers of all tile slots (DeviceArena, device.hpp): ~12 buffers per slot used to be ~500 hipMalloc
    // calls at set-up and ~500 hipFree calls — each a device-wide wait — at the end (AVDM_HOST_ARENA=0: the A/B).  Declared before its users.
    const auto tSetup0 = std::chrono::steady_clock::now();
    size_t arenaBytes = 0;
    {
        const char* e = getenv("AVDM_HOST_ARENA");
        if(!(e != nullptr && e[0] == '0'))
        {
            const double MB = 1024.0 * 1024.0;
            const double perSlotMB = Sgm::deviceMemoryConsumption(_tileParams, _sgmParams, !_depthMapParams.useRefine, _refineParams.useSgmNormalMap) +
                                     (_depthMapParams.useRefine ? Refine::deviceMemoryConsumption(_tileParams, _refineParams) : 0.0);
            // + a guard page and the 4 KiB rounding per buffer, and the group's aggregation scratch (bounded by one more slot's Sgm figure)
            arenaBytes = (size_t)((double)nbStreams * (perSlotMB * MB * 1.02 + 32.0 * 8192.0) +
                                  Sgm::deviceMemoryConsumption(_tileParams, _sgmParams, false, false) * MB * (double)nbStreams * 0.25 + 16.0 * MB);
        }
    }
    DeviceArena arena(arenaBytes);
    std::vector<std::unique_ptr<Sgm>> sgmPerStream;
    std::vector<std::unique_ptr<Refine>> refinePerStream;
    DeviceBuffer groupScratch;
    {
        const DeviceArena::Scope arenaScope(&arena);
        const bool sgmComputeDepthSimMap = !_depthMapParams.useRefine;
        const bool sgmComputeNormalMap = _refineParams.useSgmNormalMap;
        for(int i = 0; i < nbStreams; ++i)
            sgmPerStream.push_back(std::make_unique<Sgm>(_mp, _tileParams, _sgmParams, sgmComputeDepthSimMap, sgmComputeNormalMap, deviceCache,
                                                         deviceStreamManager.getStream(i)));
        if(_depthMapParams.useRefine)
            for(int i = 0; i < nbStreams; ++i)
                refinePerStream.push_back(std::make_unique<Refine>(_mp, _tileParams, _refineParams, deviceCache, deviceStreamManager.getStream(i)));
        // scratch of the batched aggregation: the sum over a group is bounded by nbStreams maximum-size tiles
        if(_sgmParams.doSgmOptimizeVolume)
            groupScratch.allocate((size_t)nbStreams * avdm_volume_optimize_scratch_bytes(sgmPerStream.front()->getMapWidth(), sgmPerStream.front()->getMapHeight(),
                                                                                         std::max(_sgmParams.maxDepths, 1)));
    }
    const double secondsDeviceBuffers = std::chrono::duration<double>(std::chrono::steady_clock::now() - tSetup0).count();
    // one event per stream (group fan-in) + one for the aggregation (fan-out); RAII holders: an exception anywhere below (a tile that does
    // not fit, a decoding error, a failed launch) must not leak events or free page-locked vectors while they are still registered
    struct EventSet
    {
        std::vector<hipEvent_t> ev;
        explicit EventSet(size_t n)
        {
            ev.reserve(n);
            for(size_t i = 0; i < n; ++i)
            {
                hipEvent_t e = nullptr;
                AVDM_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                ev.push_back(e);
            }
        }
        ~EventSet()
        {
            for(hipEvent_t e : ev)
                (void)hipEventDestroy(e);
        }
        EventSet(const EventSet&) = delete;
        EventSet& operator=(const EventSet&) = delete;
    };
    EventSet events((size_t)nbStreams + 1);
    std::vector<hipEvent_t> volumeDone(events.ev.begin(), events.ev.begin() + nbStreams);
    const hipEvent_t aggregationDone = events.ev.back();

    std::future<void> pendingWrite;
    // the background writer reads the host tile sets: on unwinding it must have finished before they go away (its own exception, if any, is
    // dropped then — the one in flight is reported)
    struct WriterGuard
    {
        std::future<void>& f;
        ~WriterGuard()
        {
            if(f.valid())
            {
                try
                {
                    f.get();
                }
                catch(...)
                {
                }
            }
        }
    } writerGuard{pendingWrite};
    logDeviceMemoryInfo();
    AVDM_LOG_INFO("Worker " << worker << " (device " << deviceId << "): set-up (streams, per-stream device buffers; the result tiles are page-locked and the first batch's images "
                            << "uploaded beside it) in " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tCompute0).count() << " s ("
                            << secondsDeviceBuffers << " s for the device buffers of " << nbStreams << " tile slot(s): "
                            << (arena.bytes() ? "one arena of " + std::to_string(arena.bytes() >> 20) + " MB, " + std::to_string(arena.used() >> 20) + " MB used"
                                              : std::string("one allocation per buffer"))
                            << ").");

    const int nbBatches = divideRoundUp(static_cast<int>(tiles.size()), nbTilesPerBatch);
    double tilesSeconds = 0.0; // from "the batch's images are on the device" to "its tiles are computed", summed over the batches
    const int finalScaleStep = _depthMapParams.useRefine ? _refineParams.scale * _refineParams.stepXY : _sgmParams.scale * _sgmParams.stepXY;

    // camera index inside a batch: the reference uses rc % nbRcPerBatch (:390), which only separates the cameras of a batch
    // when they are consecutive; the position in the batch's camera list is used instead
    for(int b = 0; b < nbBatches; ++b)
    {
        const int firstTileIndex = b * nbTilesPerBatch;
        const int lastTileIndex = std::min((b + 1) * nbTilesPerBatch, static_cast<int>(tiles.size()));
        auto batchCamIndexOf = [&](int tileIndex) { return (tileIndex - firstTileIndex) / nbTilesPerCamera; };

        std::vector<std::vector<Float2Tile>>& depthSimMapTilePerCam = depthSimMapTileSets[b % nbHostSets];
        std::vector<std::vector<std::pair<float, float>>>& depthMinMaxTilePerCam = depthMinMaxTileSets[b % nbHostSets];

        const auto tBatch0 = std::chrono::steady_clock::now();
        auto secondsSince = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
        // the views of the batch: decoded, uploaded and converted to pyramids by a team of host threads, one view each on its own stream
        // (DeviceCache::addMipmapImages; the first batch's team has been at it since before the set-up) — the reference reads and converts them
        // one by one inside addMipmapImage (DepthMapEstimator.cpp:224-232), and so did rounds 1-5 after a parallel decode;
        // with the multi-GPU exchange nothing is decoded here: own views were published by the pre-pass, the others arrive as pyramids
        if(exchange == nullptr)
        {
            if(firstIngest.valid())
                firstIngest.get(); // (rethrows)
            else if(teamIngest)
                deviceCache.addMipmapImages(viewsOfBatch(b), minMipmapDownscale, maxMipmapDownscale, ic, _mp);
            else
            {
                const std::vector<int> camsOfBatch = viewsOfBatch(b);
                std::exception_ptr loadError;
#pragma omp parallel for schedule(dynamic, 1)
                for(int k = 0; k < (int)camsOfBatch.size(); ++k)
                {
                    try
                    {
                        ic.getImg_sync(camsOfBatch[k]);
                    }
                    catch(...)
                    {
#pragma omp critical
                        loadError = std::current_exception();
                    }
                }
                if(loadError)
                    std::rethrow_exception(loadError);
            }
        }
        // load the R and T cameras of the batch in the device cache
        for(int i = firstTileIndex; i < lastTileIndex; ++i)
        {
            const Tile& tile = tiles.at(i);
            hipStream_t s0 = deviceStreamManager.getStream(0);
            deviceCache.addMipmapImage(tile.rc, minMipmapDownscale, maxMipmapDownscale, ic, _mp, s0);
            deviceCache.addCameraParams(tile.rc, _sgmParams.scale, _mp);
            for(const int tc : tile.sgmTCams)
            {
                deviceCache.addMipmapImage(tc, minMipmapDownscale, maxMipmapDownscale, ic, _mp, s0);
                deviceCache.addCameraParams(tc, _sgmParams.scale, _mp);
            }
            if(_depthMapParams.useRefine)
            {
                deviceCache.addCameraParams(tile.rc, _refineParams.scale, _mp);
                for(const int tc : tile.refineTCams)
                {
                    deviceCache.addMipmapImage(tc, minMipmapDownscale, maxMipmapDownscale, ic, _mp, s0);
                    deviceCache.addCameraParams(tc, _refineParams.scale, _mp);
                }
            }
            deviceCache.addCameraParams(tile.rc, 1, _mp); // retrieveBestDepth always asks for downscale 1 (Sgm.cpp:316)
        }
        AVDM_HIP_CHECK(hipDeviceSynchronize());
        AVDM_LOG_INFO("Batch " << (b + 1) << "/" << nbBatches << ": images decoded, uploaded and converted to pyramids in " << secondsSince(tBatch0) << " s"
                               << (b == 0 && teamIngest && exchange == nullptr ? " after the set-up (" + std::to_string(secondsSince(tIngest0)) + " s since its team started, beside the set-up)" : std::string())
                               << ".");
        if(hostSetReady[b % nbHostSets].valid())
        {
            const auto th0 = std::chrono::steady_clock::now();
            hostSetReady[b % nbHostSets].get(); // (rethrows the task's failure)
            AVDM_LOG_INFO("Worker " << worker << " (device " << deviceId << "): waited " << secondsSince(th0) << " s more for the page-locked result tiles of set "
                                    << (b % nbHostSets) << ".");
        }
        const auto tTiles0 = std::chrono::steady_clock::now();

        // what the batch actually sweeps, in the kernels' work unit (voxel x T camera): the depth lists are capped per tile and every T camera has
        // its own plane range, so this is NOT tiles x maxDepths x maxTCams — logged so that the program's rate can be compared with a sweep of known size
        long long workSgmVoxelT = 0, workRefineVoxelT = 0, workPlanes = 0, workSgmT = 0, workRefineT = 0;
        int workTiles = 0;
        // groups of nbStreams tiles
        for(int g0 = firstTileIndex; g0 < lastTileIndex; g0 += nbStreams)
        {
            const int g1 = std::min(g0 + nbStreams, lastTileIndex);
            const int n = g1 - g0;
            std::vector<std::unique_ptr<SgmDepthList>> depthLists(n);
            std::vector<char> active(n, 0);

            // (0) depth lists on the host cores
            std::exception_ptr error;
#pragma omp parallel for schedule(dynamic)
            for(int k = 0; k < n; ++k)
            {
                try
                {
                    Tile& tile = tiles.at(g0 + k);
                    if(tile.roi.isEmpty())
                        continue;
                    Float2Tile& hostTile = depthSimMapTilePerCam.at(batchCamIndexOf(g0 + k)).at(tile.id);
                    if(tile.sgmTCams.empty() || (_depthMapParams.useRefine && tile.refineTCams.empty()))
                    {
                        resetDepthSimMap(hostTile);
                        continue;
                    }
                    depthLists[k] = std::make_unique<SgmDepthList>(_mp, _sgmParams, tile);
                    depthLists[k]->computeListRc();
                    if(depthLists[k]->getDepths().empty())
                    {
                        resetDepthSimMap(hostTile);
                        depthMinMaxTilePerCam.at(batchCamIndexOf(g0 + k)).at(tile.id) = {0.f, 0.f};
                        continue;
                    }
                    depthLists[k]->removeTcWithNoDepth(tile);
                    depthMinMaxTilePerCam.at(batchCamIndexOf(g0 + k)).at(tile.id) = depthLists[k]->getMinMaxDepths();
                    active[k] = 1;
                }
                catch(...)
                {
#pragma omp critical
                    error = std::current_exception();
                }
            }
            if(error)
                std::rethrow_exception(error);

            for(int k = 0; k < n; ++k)
            {
                if(!active[k])
                    continue;
                const Tile& tile = tiles.at(g0 + k);
                const ROI rs = downscaleROI(tile.roi, (float)(_sgmParams.scale * _sgmParams.stepXY));
                const ROI rr = downscaleROI(tile.roi, (float)(_refineParams.scale * _refineParams.stepXY));
                long long planesT = 0;
                for(const Pixel& lim : depthLists[k]->getDepthsTcLimits())
                    planesT += lim.y;
                workSgmVoxelT += (long long)rs.width() * rs.height() * planesT;
                if(_depthMapParams.useRefine)
                    workRefineVoxelT += (long long)rr.width() * rr.height() * (2 * _refineParams.halfNbDepths + 1) * (long long)tile.refineTCams.size();
                workPlanes += (long long)depthLists[k]->getDepths().size();
                workSgmT += (long long)tile.sgmTCams.size();
                workRefineT += (long long)tile.refineTCams.size();
                ++workTiles;
            }

            // (A0) the adaptive-P2 maps of the group's aggregation: they depend on nothing but the R pyramids, so they are evaluated on the
            // aggregation stream BEFORE the sweeps (avdm_volume_optimize_prepare) and step (B) is the path launches alone
            std::vector<avdm_sgm_tile_t> aggTiles;
            const avdm_sgm_params_t sp = _sgmParams.toAvdm();
            hipStream_t aggStream = deviceStreamManager.getStream(0);
            if(_sgmParams.doSgmOptimizeVolume)
            {
                for(int k = 0; k < n; ++k)
                    if(active[k])
                        aggTiles.push_back(sgmPerStream.at(k)->layoutAndDescribe(tiles.at(g0 + k), *depthLists[k]));
                if(!aggTiles.empty())
                    avdmCheck(avdm_volume_optimize_prepare((int)aggTiles.size(), aggTiles.data(), groupScratch.ptr(), &sp, aggStream), "avdm_volume_optimize_prepare");
            }

            // (A) similarity volumes, one stream per tile
            for(int k = 0; k < n; ++k)
            {
                if(!active[k])
                    continue;
                const Tile& tile = tiles.at(g0 + k);
                depthLists[k]->logRcTcDepthInformation();
                depthLists[k]->checkStartingAndStoppingDepth();
                Sgm& sgm = *sgmPerStream.at(k);
                sgm.computeVolumes(tile, *depthLists[k]);
                if(_sgmParams.doSgmOptimizeVolume)
                    AVDM_HIP_CHECK(hipEventRecord(volumeDone[k], sgm.getStream()));
                else
                    sgm.optimizeDisabledCopy();
            }

            // (B) one batched aggregation for the group, on stream 0 after every tile's volumes
            if(!aggTiles.empty())
            {
                for(int k = 0; k < n; ++k)
                    if(active[k])
                        AVDM_HIP_CHECK(hipStreamWaitEvent(aggStream, volumeDone[k], 0));
                AVDM_LOG_INFO("SGM Optimizing volume of " << aggTiles.size() << " tile(s) in one batch (filtering axes: " << _sgmParams.filteringAxes << ").");
                avdmCheck(avdm_volume_optimize_tiles_prepared((int)aggTiles.size(), aggTiles.data(), groupScratch.ptr(), &sp, aggStream),
                          "avdm_volume_optimize_tiles_prepared");
                AVDM_HIP_CHECK(hipEventRecord(aggregationDone, aggStream));
            }

            // (C) best depth, Refine, copy back
            for(int k = 0; k < n; ++k)
            {
                if(!active[k])
                    continue;
                Tile& tile = tiles.at(g0 + k);
                Sgm& sgm = *sgmPerStream.at(k);
                hipStream_t stream = sgm.getStream();
                if(!aggTiles.empty())
                    AVDM_HIP_CHECK(hipStreamWaitEvent(stream, aggregationDone, 0));
                sgm.finish(tile, *depthLists[k]);
                Float2Tile& hostTile = depthSimMapTilePerCam.at(batchCamIndexOf(g0 + k)).at(tile.id);
                const ROI r = downscaleROI(tile.roi, (float)finalScaleStep);
                const float* src;
                int srcPitch;
                if(_depthMapParams.useRefine)
                {
                    sgm.smoothThicknessMap(tile, _refineParams);
                    Refine& refine = *refinePerStream.at(k);
                    refine.refineRc(tile, sgm);
                    src = refine.getDeviceDepthSimMap();
                    srcPitch = refine.getMapPitch();
                }
                else
                {
                    src = sgm.getDeviceDepthSimMap();
                    srcPitch = sgm.getDepthThicknessMapPitch();
                }
                AVDM_HIP_CHECK(hipMemcpy2DAsync(hostTile.data.data(), (size_t)hostTile.width * 8, src, (size_t)srcPitch, (size_t)r.width() * 8, (size_t)r.height(),
                                                hipMemcpyDeviceToHost, stream));
            }
            // the Sgm / Refine buffers of a slot are reused by the next group
            AVDM_HIP_CHECK(hipDeviceSynchronize());
            AVDM_LOG_INFO("Batch " << (b + 1) << "/" << nbBatches << ": " << n << " tile(s) computed, " << secondsSince(tBatch0) << " s since the batch started.");
        }

        tilesSeconds += secondsSince(tTiles0);
        if(workTiles > 0)
            AVDM_LOG_INFO("Batch " << (b + 1) << "/" << nbBatches << ": swept " << workTiles << " tile(s): " << workSgmVoxelT << " SGM voxel-T, " << workRefineVoxelT
                                   << " Refine voxel-T; per tile on average " << (double)workPlanes / workTiles << " planes, " << (double)workSgmT / workTiles
                                   << " SGM T cameras, " << (double)workRefineT / workTiles << " Refine T cameras.");
        // write the finished cameras of the batch, in the background: the previous batch's task must be done first (it owns the other set)
        if(pendingWrite.valid())
        {
            const auto tw0 = std::chrono::steady_clock::now();
            pendingWrite.get();
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count();
            if(waited > 0.01)
                AVDM_LOG_INFO("Batch " << (b + 1) << "/" << nbBatches << ": waited " << waited << " s for the previous batch's maps to be written.");
        }
        std::vector<int> batchCams;
        for(int ci = 0; ci * nbTilesPerCamera + firstTileIndex < lastTileIndex; ++ci)
            batchCams.push_back(tiles.at(firstTileIndex + ci * nbTilesPerCamera).rc);
        pendingWrite = std::async(std::launch::async, [this, batchCams, b, nbBatches, tBatch0, &depthSimMapTilePerCam, &depthMinMaxTilePerCam]() {
            for(size_t ci = 0; ci < batchCams.size(); ++ci)
            {
                const int c = batchCams[ci];
                if(_depthMapParams.useRefine)
                    writeDepthSimMapFromTileList(c, _mp, _tileParams, _tileRoiList, depthSimMapTilePerCam.at(ci), _refineParams.scale, _refineParams.stepXY);
                else
                    writeDepthSimMapFromTileList(c, _mp, _tileParams, _tileRoiList, depthSimMapTilePerCam.at(ci), _sgmParams.scale, _sgmParams.stepXY);
                if(_depthMapParams.exportTilePattern)
                    exportDepthSimMapTilePatternObj(c, _mp, _tileRoiList, depthMinMaxTilePerCam.at(ci));
            }
            AVDM_LOG_INFO("Batch " << (b + 1) << "/" << nbBatches << ": depth / similarity maps merged and written, "
                                   << std::chrono::duration<double>(std::chrono::steady_clock::now() - tBatch0).count() << " s since the batch started.");
        });
    }
    {
        const auto tTail0 = std::chrono::steady_clock::now();
        // While the background task merges and writes the last batch's maps (it reads host memory only), this thread gives the device side back:
        // the tile slots' buffers, the page-locks of the result tiles (unpinning leaves the host bytes where they are), the pyramids, the streams
        // with the library's scratch blocks — ~0.4 s of hipFree / hipHostUnregister calls that used to follow the wait.
        {
            auto lap = [last = tTail0]() mutable {
                const auto now = std::chrono::steady_clock::now();
                const double d = std::chrono::duration<double>(now - last).count();
                last = now;
                return d;
            };
            refinePerStream.clear();
            sgmPerStream.clear();
            groupScratch.release();
            hostTilesTask.wait();
            arena.release();
            const double tBuffers = lap();
            // (the page-locks on a second thread: 80 hipHostUnregister calls beside the frees of this one)
            std::future<void> unpin = std::async(std::launch::async, [&]() {
                (void)hipSetDevice(deviceId);
                pinned.releaseAll();
            });
            deviceCache.releaseImages();
            const double tImages = lap();
            deviceStreamManager.destroy();
            const double tStreams = lap();
            unpin.wait();
            const double tUnpin = lap();
            AVDM_LOG_INFO("Worker " << worker << " (device " << deviceId << "): device buffers, pyramids, streams and page-locks released in "
                                    << std::chrono::duration<double>(std::chrono::steady_clock::now() - tTail0).count() << " s, beside the last batch's merge + write (tile slots "
                                    << tBuffers << ", pyramids " << tImages << ", streams + scratch blocks " << tStreams << ", waiting for the page-locks " << tUnpin << " s).");
        }
        if(pendingWrite.valid())
            pendingWrite.get();
        AVDM_LOG_INFO("Worker " << worker << " (device " << deviceId << "): waited " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tTail0).count()
                                << " s for the last batch's maps to be merged and written; " << std::chrono::duration<double>(std::chrono::steady_clock::now() - tCompute0).count()
                                << " s for " << cams.size() << " camera(s) in all.");
        // the Refine sweep's outlier lists (library scratch): a list that was ever full put waves on the slower per-plane path — same results, but
        // an undersized capacity must not go unnoticed (0 in every run measured)
        unsigned refused = 0;
        avdmCheck(avdm_refine_outlier_refused(&refused), "avdm_refine_outlier_refused");
        if(refused > 0)
            AVDM_LOG_WARNING("Worker " << worker << " (device " << deviceId << "): " << refused << " Refine outlier-list unit(s) found their list full and ran on the per-plane path.");
        else
            AVDM_LOG_INFO("Worker " << worker << " (device " << deviceId << "): Refine outlier lists: no unit refused.");
        const DeviceCache::ImageTimes& it = deviceCache.imageTimes();
        AVDM_LOG_INFO("Worker " << worker << " (device " << deviceId << "): images of the tiles' batches — " << it.received << " pyramid(s) received from their owners ("
                                << (it.bytesReceived >> 20) << " MB; " << it.awaitOwner << " s waiting for the owner, " << it.peerCopy << " s copying), " << it.built
                                << " built here (" << it.localBuild << " s: decode unless cached, upload, pyramid); tiles: " << tilesSeconds << " s in " << nbBatches
                                << " batch(es).");
    }

    {
        const auto tFree0 = std::chrono::steady_clock::now();
        for(int s = 0; s < nbHostSets; ++s)
        {
            depthSimMapTileSets[s].clear();
            depthSimMapTileSets[s].shrink_to_fit();
        }
        AVDM_LOG_INFO("Worker " << worker << " (device " << deviceId << "): host result tiles freed in "
                                << std::chrono::duration<double>(std::chrono::steady_clock::now() - tFree0).count() << " s.");
    }

    // merge intermediate result tiles (:470-505)
    if(tiles.size() > cams.size())
        for(const int rc : cams)
        {
            if(_sgmParams.exportIntermediateDepthSimMaps)
                mergeDepthSimMapTiles(rc, _mp, _sgmParams.scale, _sgmParams.stepXY, "sgm");
            if(_sgmParams.exportIntermediateNormalMaps)
                mergeNormalMapTiles(rc, _mp, _sgmParams.scale, _sgmParams.stepXY, "sgm");
            if(_depthMapParams.useRefine)
            {
                if(_refineParams.exportIntermediateDepthSimMaps)
                {
                    mergeDepthPixSizeMapTiles(rc, _mp, _refineParams.scale, _refineParams.stepXY, "sgmUpscaled");
                    mergeDepthSimMapTiles(rc, _mp, _refineParams.scale, _refineParams.stepXY, "refinedFused");
                }
                if(_refineParams.exportIntermediateNormalMaps)
                {
                    mergeNormalMapTiles(rc, _mp, _refineParams.scale, _refineParams.stepXY, "refinedFused");
                    mergeNormalMapTiles(rc, _mp, _refineParams.scale, _refineParams.stepXY);
                }
            }
        }

}

} // namespace avdm_host
