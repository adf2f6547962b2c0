// oracle/_ref: one tile through the reference's OWN host classes — depthMap/Sgm.cpp and depthMap/Refine.cpp compiled WHOLE and unchanged
// (constructors with their buffer sizes, Sgm::sgmRc, Sgm::smoothThicknessMap, Refine::refineRc and everything they call, over the
// kernel-launch layer that libavdm_ref.so already is).  Test infrastructure only: tests/test_oracle_ref.py holds the oracle's
// per-tile control flow (oracle/oracle.py: OracleDepthMap.run_sgm / run_refine) against this, map for map.
// Stand-ins: DeviceCache (shim_host/: answers from what the test registered — pyramids built with the reference's own
// DeviceMipmapImage::fill, camera blocks in the constant-memory slots), MultiViewParams (only getViewId is asked of it here), and
// no-op bodies for the export functions of depthMapUtils.hpp / volumeIO.hpp (OpenImageIO / Alembic; their switches stay off).
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <aliceVision/mvsData/Pixel.hpp>
#include <aliceVision/mvsUtils/MultiViewParams.hpp>
#include <aliceVision/depthMap/SgmParams.hpp>
#include <aliceVision/depthMap/Tile.hpp>
// SgmDepthList has no setter for a given list — the test hands over the planes and limits it also gives the oracle — so its two result
// members are written directly (everything SgmDepthList.hpp includes has been included above, with its access specifiers intact)
#define private public
#include <aliceVision/depthMap/SgmDepthList.hpp>
#undef private
#include <aliceVision/depthMap/Sgm.hpp>
#include <aliceVision/depthMap/Refine.hpp>
#include <aliceVision/depthMap/depthMapUtils.hpp>
#include <aliceVision/depthMap/volumeIO.hpp>
#include <aliceVision/depthMap/cuda/host/DeviceCache.hpp>

#include "avdm.h"

namespace aliceVision {
namespace depthMap {
// ---- exports: declared by the reference's headers, never reached (export switches off) ----
void writeNormalMap(int, const mvsUtils::MultiViewParams&, const mvsUtils::TileParams&, const ROI&, const CudaDeviceMemoryPitched<float3, 2>&, int, int,
                    const std::string&) {}
void writeDepthPixSizeMap(int, const mvsUtils::MultiViewParams&, const mvsUtils::TileParams&, const ROI&, const CudaDeviceMemoryPitched<float2, 2>&, int, int,
                          const std::string&) {}
void writeDepthSimMap(int, const mvsUtils::MultiViewParams&, const mvsUtils::TileParams&, const ROI&, const CudaDeviceMemoryPitched<float2, 2>&, int, int,
                      const std::string&) {}
void exportSimilaritySamplesCSV(const CudaHostMemoryHeap<TSim, 3>&, const std::vector<float>&, const std::string&, const SgmParams&, const std::string&,
                                const ROI&) {}
void exportSimilaritySamplesCSV(const CudaHostMemoryHeap<TSimRefine, 3>&, const std::string&, const RefineParams&, const std::string&, const ROI&) {}
void exportSimilarityVolume(const CudaHostMemoryHeap<TSim, 3>&, const std::vector<float>&, const mvsUtils::MultiViewParams&, int, const SgmParams&,
                            const std::string&, const ROI&) {}
void exportSimilarityVolumeCross(const CudaHostMemoryHeap<TSim, 3>&, const std::vector<float>&, const mvsUtils::MultiViewParams&, int, const SgmParams&,
                                 const std::string&, const ROI&) {}
void exportSimilarityVolumeCross(const CudaHostMemoryHeap<TSimRefine, 3>&, const CudaHostMemoryHeap<float2, 2>&, const mvsUtils::MultiViewParams&, int,
                                 const RefineParams&, const std::string&, const ROI&) {}
void exportSimilarityVolumeTopographicCut(const CudaHostMemoryHeap<TSim, 3>&, const std::vector<float>&, const mvsUtils::MultiViewParams&, int, const SgmParams&,
                                          const std::string&, const ROI&) {}
void exportSimilarityVolumeTopographicCut(const CudaHostMemoryHeap<TSimRefine, 3>&, const CudaHostMemoryHeap<float2, 2>&, const mvsUtils::MultiViewParams&, int,
                                          const RefineParams&, const std::string&, const ROI&) {}
} // namespace depthMap
} // namespace aliceVision

using namespace aliceVision;
using namespace aliceVision::depthMap;

namespace {
template <class T>
void map_out2(void* m, int pitch, const CudaDeviceMemoryPitched<T, 2>& dmp, int W, int H)
{
    for(int y = 0; y < H; ++y)
        std::memcpy((char*)m + (long long)y * pitch, (const char*)dmp.getBuffer() + y * dmp.getPitch(), (size_t)W * sizeof(T));
}
SgmParams to_ref(const avdm_sgm_params_t& p, int maxDepths)
{
    SgmParams s;
    s.scale = p.scale, s.stepXY = p.stepXY, s.wsh = p.wsh, s.maxDepths = maxDepths;
    s.gammaC = p.gammaC, s.gammaP = p.gammaP, s.p1 = p.p1, s.p2Weighting = p.p2Weighting;
    s.maxSimilarity = p.maxSimilarity, s.depthThicknessInflate = p.depthThicknessInflate;
    s.filteringAxes = p.filteringAxes;
    s.useConsistentScale = p.useConsistentScale != 0, s.useCustomPatchPattern = p.useCustomPatchPattern != 0;
    return s;
}
RefineParams to_ref(const avdm_refine_params_t& p)
{
    RefineParams r;
    r.scale = p.scale, r.stepXY = p.stepXY, r.wsh = p.wsh, r.halfNbDepths = p.halfNbDepths, r.nbSubsamples = p.nbSubsamples;
    r.optimizationNbIterations = p.optimizationNbIterations;
    r.sigma = p.sigma, r.gammaC = p.gammaC, r.gammaP = p.gammaP;
    r.interpolateMiddleDepth = p.interpolateMiddleDepth != 0;
    r.useConsistentScale = p.useConsistentScale != 0, r.useCustomPatchPattern = p.useCustomPatchPattern != 0;
    return r;
}
} // namespace

extern "C" {

// what DeviceCache::requestMipmapImage / requestCameraParamsId answer with
void avr_cache_clear()
{
    DeviceCache::getInstance().images.clear();
    DeviceCache::getInstance().slots.clear();
}
void avr_cache_register_image(int camId, void* image) { DeviceCache::getInstance().images[camId] = (const DeviceMipmapImage*)image; }
void avr_cache_register_camera(int camId, int downscale, int slot) { DeviceCache::getInstance().slots[{camId, downscale}] = slot; }

// Sgm(mp, tileParams, sgmParams, computeDepthSimMap = 1, computeNormalMap) ; sgm.sgmRc(tile, depth list) ; then, like
// DepthMapEstimator::compute (DepthMapEstimator.cpp:411-433), sgm.smoothThicknessMap(tile, refineParams) ; Refine(mp, tileParams,
// refineParams) ; refine.refineRc(tile, sgm depth/thickness map, sgm normal map).
// roi = x0, x1, y0, y1 at process resolution; limits = first index and count per T camera (SgmDepthList::getDepthsTcLimits).
// Outputs (row pitch = width * sizeof(pixel)): SGM depth/thickness BEFORE smoothing (sgmW x sgmH float2), SGM depth/sim, the smoothed
// depth/thickness, SGM normals (float3, may be NULL), and the Refine result (refW x refH float2; NULL: SGM only).  0 on success.
int avr_tile_run(int tile_buffer_w, int tile_buffer_h, const int roi[4], int rc, int n_tc, const int* tcams, const avdm_sgm_params_t* sp, int max_depths,
                 const avdm_refine_params_t* rp, int use_refine_fuse, int use_color_optimization, const float* depths, int n_depths, const int* limits,
                 int compute_normal, float* out_dt, float* out_ds, float* out_dt_smooth, float* out_normal, float* out_refined)
{
    try
    {
        mvsUtils::MultiViewParams mp;
        mvsUtils::TileParams tp;
        tp.bufferWidth = tile_buffer_w;
        tp.bufferHeight = tile_buffer_h;
        const SgmParams sgmParams = to_ref(*sp, max_depths);
        RefineParams refineParams = to_ref(*rp);
        refineParams.useRefineFuse = use_refine_fuse != 0;
        refineParams.useColorOptimization = use_color_optimization != 0;

        Tile tile;
        tile.id = 0, tile.nbTiles = 1, tile.rc = rc;
        tile.sgmTCams.assign(tcams, tcams + n_tc);
        tile.refineTCams = tile.sgmTCams;
        tile.roi = ROI((unsigned)roi[0], (unsigned)roi[1], (unsigned)roi[2], (unsigned)roi[3]);

        SgmDepthList dl(mp, sgmParams, tile);
        dl._depths.assign(depths, depths + n_depths);
        for(int c = 0; c < n_tc; ++c)
            dl._depthsTcLimits.push_back(Pixel(limits[2 * c], limits[2 * c + 1]));

        Sgm sgm(mp, tp, sgmParams, true, compute_normal != 0, nullptr);
        sgm.sgmRc(tile, dl);
        const ROI rs = downscaleROI(tile.roi, sgmParams.scale * sgmParams.stepXY);
        const int sw = (int)rs.width(), sh = (int)rs.height();
        map_out2(out_dt, sw * 8, sgm.getDeviceDepthThicknessMap(), sw, sh);
        map_out2(out_ds, sw * 8, sgm.getDeviceDepthSimMap(), sw, sh);
        if(compute_normal && out_normal != nullptr)
            map_out2(out_normal, sw * 12, sgm.getDeviceNormalMap(), sw, sh);
        if(out_refined == nullptr)
            return 0;
        sgm.smoothThicknessMap(tile, refineParams);
        map_out2(out_dt_smooth, sw * 8, sgm.getDeviceDepthThicknessMap(), sw, sh);
        Refine refine(mp, tp, refineParams, nullptr);
        refine.refineRc(tile, sgm.getDeviceDepthThicknessMap(), sgm.getDeviceNormalMap());
        const ROI rr = downscaleROI(tile.roi, refineParams.scale * refineParams.stepXY);
        map_out2(out_refined, (int)rr.width() * 8, refine.getDeviceDepthSimMap(), (int)rr.width(), (int)rr.height());
        return 0;
    }
    catch(const std::exception& e)
    {
        std::cerr << "[ref] avr_tile_run: " << e.what() << std::endl;
        return 2;
    }
}

} // extern "C"
