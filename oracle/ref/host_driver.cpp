// oracle/_ref (host): C entry point around the reference's OWN depth-plane list — depthMap/SgmDepthList.cpp compiled WHOLE and unchanged
// (computeListRc and everything below it: getMinMaxMidNbDepthFromSfM, getRcTcDepthRangeFromSfM, computeRcTcDepths,
// computePixelSizeDepths, computeRcDepthList, indexOfNearestSorted), with MultiViewParams' projection / pixel-size methods and
// common.cpp's epipolar helpers from the reference's text (gen_extract.py) and mvsData as it lies.  Test infrastructure only:
// tests/test_host_ref.py holds oracle/host_oracle.py (which the C++ host equals, tests/test_host_cpu.py) against it.
// Stand-ins: shim_host/ (MultiViewParams as plain arrays, the landmark containers, Boost's tail quantile — the one unpinned piece).
#include <aliceVision/depthMap/DepthMapParams.hpp>
#include <aliceVision/depthMap/SgmDepthList.hpp>
#include <aliceVision/mvsUtils/TileParams.hpp>

#include "fuse_standin.hpp" // image::Image<T>

#include <algorithm>
#include <sstream>

namespace aliceVision {
namespace mvsUtils {
// mvsUtils/mapIO.cpp:170-311 (file-local templates there), from the reference's text: gen/host_mapIO.cpp instantiates them for float
void addSingleTileMapWeightedFloat(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, int downscale,
                                   image::Image<float>& in_tileMap, image::Image<float>& inout_map);
} // namespace mvsUtils
} // namespace aliceVision

using namespace aliceVision;

extern "C" {

// Cameras: n full-resolution projection matrices (row-major 3x4); like MultiViewParams::loadMatricesFromRawProjectionMatrix
// (MultiViewParams.cpp:283-297) the first two rows are divided by the process downscale, then K, R, C = decomposeProjectionMatrix,
// iK, iR, iCamArr = iR * iK.  widths / heights at process resolution.
// Landmarks: points X (n x 3) with observations obs_view / obs_xy (full-resolution pixel coordinates), CSR offsets obs_begin (n + 1).
// Returns 0; out_depths (<= cap entries, *out_n of them) and out_limits (first index, count per T camera of `tcams`, computeListRc's
// _depthsTcLimits before removeTcWithNoDepth).  2 = the reference threw (message on stderr).
int avref_sgm_depth_list(int n_cams, const double* P, const int* widths, const int* heights, int process_downscale, float min_view_angle,
                         float max_view_angle, int n_landmarks, const double* X, const int* obs_begin, const int* obs_view, const double* obs_xy,
                         int rc, int n_tc, const int* tcams, const int roi[4], int sgm_scale, int max_depths, int step_z,
                         double seeds_range_inflate, int use_sfm_seeds, int depth_list_per_tile, float* out_depths, int cap, int* out_n,
                         int* out_limits)
{
    try
    {
        mvsUtils::MultiViewParams mp;
        mp.processDownscale = process_downscale;
        mp._minViewAngle = min_view_angle;
        mp._maxViewAngle = max_view_angle;
        for(int i = 0; i < n_cams; ++i)
        {
            Matrix3x4 pMatrix;
            std::copy_n(P + 12 * i, 12, pMatrix.m);
            const double imgScale = double(process_downscale);
            for(int k = 0; k < 8; ++k)
                pMatrix.m[k] /= imgScale;
            Matrix3x3 K, R;
            Point3d C;
            pMatrix.decomposeProjectionMatrix(K, R, C);
            mp.camArr.push_back(pMatrix);
            mp.KArr.push_back(K);
            mp.RArr.push_back(R);
            mp.CArr.push_back(C);
            mp.iKArr.push_back(K.inverse());
            mp.iRArr.push_back(R.inverse());
            mp.iCamArr.push_back(mp.iRArr.back() * mp.iKArr.back());
            mp.widths.push_back(widths[i]);
            mp.heights.push_back(heights[i]);
        }
        for(int l = 0; l < n_landmarks; ++l)
        {
            sfmData::Landmark lm;
            lm.X.v[0] = X[3 * l], lm.X.v[1] = X[3 * l + 1], lm.X.v[2] = X[3 * l + 2];
            for(int o = obs_begin[l]; o < obs_begin[l + 1]; ++o)
                lm.observations[(IndexT)obs_view[o]] = sfmData::Observation{Vec2(obs_xy[2 * o], obs_xy[2 * o + 1])};
            mp.sfm.landmarks[(IndexT)l] = lm;
        }
        depthMap::SgmParams sp;
        sp.scale = sgm_scale;
        sp.maxDepths = max_depths;
        sp.stepZ = step_z;
        sp.seedsRangeInflate = seeds_range_inflate;
        sp.useSfmSeeds = use_sfm_seeds != 0;
        sp.depthListPerTile = depth_list_per_tile != 0;
        depthMap::Tile tile;
        tile.id = 0;
        tile.nbTiles = 1;
        tile.rc = rc;
        tile.sgmTCams.assign(tcams, tcams + n_tc);
        tile.refineTCams = tile.sgmTCams;
        tile.roi = ROI((unsigned)roi[0], (unsigned)roi[1], (unsigned)roi[2], (unsigned)roi[3]);

        depthMap::SgmDepthList dl(mp, sp, tile);
        dl.computeListRc();
        const std::vector<float>& d = dl.getDepths();
        *out_n = (int)d.size();
        std::copy_n(d.data(), std::min<size_t>(d.size(), (size_t)cap), out_depths);
        const std::vector<Pixel>& lim = dl.getDepthsTcLimits();
        for(size_t c = 0; c < lim.size() && c < (size_t)n_tc; ++c)
        {
            out_limits[2 * c] = lim[c].x;
            out_limits[2 * c + 1] = lim[c].y;
        }
        return 0;
    }
    catch(const std::exception& e)
    {
        std::cerr << "[ref] avref_sgm_depth_list: " << e.what() << std::endl;
        return 2;
    }
}

// MultiViewParams::findNearestCamsFromLandmarks (MultiViewParams.cpp:519-575) and findTileNearestCams (:577-667), from the reference's
// text.  Cameras: pinhole K = (fx, fy, cx, cy) per view at full resolution and world -> camera rotations R (row-major); landmarks as in
// avref_sgm_depth_list.  tcams_in / roi only for the tile form (n_tc_in < 0: the whole-image ranking).  Returns the number of cameras.
int avref_nearest_cams(int n_cams, const double* K4, const double* R, int process_downscale, float min_view_angle, float max_view_angle, int n_landmarks,
                       const int* obs_begin, const int* obs_view, const double* obs_xy, int rc, int nb_nearest, int n_tc_in, const int* tcams_in,
                       const int roi[4], int* out)
{
    try
    {
        mvsUtils::MultiViewParams mp;
        mp.processDownscale = process_downscale;
        mp._minViewAngle = min_view_angle;
        mp._maxViewAngle = max_view_angle;
        for(int i = 0; i < n_cams; ++i)
        {
            mp.camArr.push_back(Matrix3x4());
            auto v = std::make_shared<sfmData::View>();
            v->intrinsicId = (IndexT)i;
            std::copy_n(R + 9 * i, 9, v->pose.R);
            mp.sfm.views[(IndexT)i] = v;
            mp.sfm.intrinsics[(IndexT)i] = camera::IntrinsicBase{K4[4 * i], K4[4 * i + 1], K4[4 * i + 2], K4[4 * i + 3]};
        }
        for(int l = 0; l < n_landmarks; ++l)
        {
            sfmData::Landmark lm;
            
            for(int o = obs_begin[l]; o < obs_begin[l + 1]; ++o)
                lm.observations[(IndexT)obs_view[o]] = sfmData::Observation{Vec2(obs_xy[2 * o], obs_xy[2 * o + 1])};
            mp.sfm.landmarks[(IndexT)l] = lm;
        }
        int n = 0;
        if(n_tc_in < 0)
        {
            const StaticVector<int> got = mp.findNearestCamsFromLandmarksRef(rc, nb_nearest);
            for(int i = 0; i < got.size(); ++i)
                out[n++] = got[i];
        }
        else
        {
            const std::vector<int> tc(tcams_in, tcams_in + n_tc_in);
            const ROI r((unsigned)roi[0], (unsigned)roi[1], (unsigned)roi[2], (unsigned)roi[3]);
            for(const int c : mp.findTileNearestCams(rc, nb_nearest, tc, r))
                out[n++] = c;
        }
        return n;
    }
    catch(const std::exception& e)
    {
        std::cerr << "[ref] avref_nearest_cams: " << e.what() << std::endl;
        return -1;
    }
}

// The default value of every parameter of the stage, read from the reference's own headers (depthMap/SgmParams.hpp, RefineParams.hpp,
// DepthMapParams.hpp, mvsUtils/TileParams.hpp): "group.name=value" lines.  Returns the length written (0 if cap is too small).
int avref_default_params(char* out, int cap)
{
    const depthMap::SgmParams sgm{};
    const depthMap::RefineParams refine{};
    const depthMap::DepthMapParams dm{};
    const mvsUtils::TileParams tile{};
    std::ostringstream os;
    os.precision(17);
    os << "sgm.scale=" << sgm.scale << "\n";
    os << "sgm.stepXY=" << sgm.stepXY << "\n";
    os << "sgm.stepZ=" << sgm.stepZ << "\n";
    os << "sgm.wsh=" << sgm.wsh << "\n";
    os << "sgm.maxDepths=" << sgm.maxDepths << "\n";
    os << "sgm.maxTCamsPerTile=" << sgm.maxTCamsPerTile << "\n";
    os << "sgm.seedsRangeInflate=" << sgm.seedsRangeInflate << "\n";
    os << "sgm.depthThicknessInflate=" << sgm.depthThicknessInflate << "\n";
    os << "sgm.maxSimilarity=" << sgm.maxSimilarity << "\n";
    os << "sgm.gammaC=" << sgm.gammaC << "\n";
    os << "sgm.gammaP=" << sgm.gammaP << "\n";
    os << "sgm.p1=" << sgm.p1 << "\n";
    os << "sgm.p2Weighting=" << sgm.p2Weighting << "\n";
    os << "sgm.filteringAxes=" << sgm.filteringAxes << "\n";
    os << "sgm.useSfmSeeds=" << sgm.useSfmSeeds << "\n";
    os << "sgm.depthListPerTile=" << sgm.depthListPerTile << "\n";
    os << "sgm.useConsistentScale=" << sgm.useConsistentScale << "\n";
    os << "sgm.useCustomPatchPattern=" << sgm.useCustomPatchPattern << "\n";
    os << "sgm.exportIntermediateDepthSimMaps=" << sgm.exportIntermediateDepthSimMaps << "\n";
    os << "sgm.exportIntermediateNormalMaps=" << sgm.exportIntermediateNormalMaps << "\n";
    os << "sgm.exportIntermediateVolumes=" << sgm.exportIntermediateVolumes << "\n";
    os << "sgm.exportIntermediateCrossVolumes=" << sgm.exportIntermediateCrossVolumes << "\n";
    os << "sgm.exportIntermediateTopographicCutVolumes=" << sgm.exportIntermediateTopographicCutVolumes << "\n";
    os << "sgm.exportIntermediateVolume9pCsv=" << sgm.exportIntermediateVolume9pCsv << "\n";
    os << "sgm.exportDepthsTxtFiles=" << sgm.exportDepthsTxtFiles << "\n";
    os << "sgm.updateUninitializedSim=" << sgm.updateUninitializedSim << "\n";
    os << "sgm.prematchingMaxDepthScale=" << sgm.prematchingMaxDepthScale << "\n";
    os << "sgm.seedsRangePercentile=" << sgm.seedsRangePercentile << "\n";
    os << "sgm.doSgmOptimizeVolume=" << sgm.doSgmOptimizeVolume << "\n";
    os << "refine.scale=" << refine.scale << "\n";
    os << "refine.stepXY=" << refine.stepXY << "\n";
    os << "refine.wsh=" << refine.wsh << "\n";
    os << "refine.halfNbDepths=" << refine.halfNbDepths << "\n";
    os << "refine.nbSubsamples=" << refine.nbSubsamples << "\n";
    os << "refine.maxTCamsPerTile=" << refine.maxTCamsPerTile << "\n";
    os << "refine.optimizationNbIterations=" << refine.optimizationNbIterations << "\n";
    os << "refine.sigma=" << refine.sigma << "\n";
    os << "refine.gammaC=" << refine.gammaC << "\n";
    os << "refine.gammaP=" << refine.gammaP << "\n";
    os << "refine.interpolateMiddleDepth=" << refine.interpolateMiddleDepth << "\n";
    os << "refine.useConsistentScale=" << refine.useConsistentScale << "\n";
    os << "refine.useCustomPatchPattern=" << refine.useCustomPatchPattern << "\n";
    os << "refine.useRefineFuse=" << refine.useRefineFuse << "\n";
    os << "refine.useColorOptimization=" << refine.useColorOptimization << "\n";
    os << "refine.exportIntermediateDepthSimMaps=" << refine.exportIntermediateDepthSimMaps << "\n";
    os << "refine.exportIntermediateNormalMaps=" << refine.exportIntermediateNormalMaps << "\n";
    os << "refine.exportIntermediateCrossVolumes=" << refine.exportIntermediateCrossVolumes << "\n";
    os << "refine.exportIntermediateTopographicCutVolumes=" << refine.exportIntermediateTopographicCutVolumes << "\n";
    os << "refine.exportIntermediateVolume9pCsv=" << refine.exportIntermediateVolume9pCsv << "\n";
    os << "refine.useSgmNormalMap=" << refine.useSgmNormalMap << "\n";
    os << "tile.bufferWidth=" << tile.bufferWidth << "\n";
    os << "tile.bufferHeight=" << tile.bufferHeight << "\n";
    os << "tile.padding=" << tile.padding << "\n";
    os << "depthMap.maxTCams=" << dm.maxTCams << "\n";
    os << "depthMap.chooseTCamsPerTile=" << dm.chooseTCamsPerTile << "\n";
    os << "depthMap.exportTilePattern=" << dm.exportTilePattern << "\n";
    os << "depthMap.autoAdjustSmallImage=" << dm.autoAdjustSmallImage << "\n";
    os << "depthMap.useRefine=" << dm.useRefine << "\n";
    const std::string text = os.str();
    if((int)text.size() + 1 > cap)
        return 0;
    std::copy(text.begin(), text.end(), out);
    out[text.size()] = 0;
    return (int)text.size();
}

// mvsUtils::getTileRoiList (TileParams.cpp:15-61, compiled whole): out = x0, x1, y0, y1 per tile; returns the number of tiles
int avref_tile_roi_list(int buffer_w, int buffer_h, int padding, int image_w, int image_h, int max_downscale, int* out, int cap)
{
    mvsUtils::TileParams tp;
    tp.bufferWidth = buffer_w;
    tp.bufferHeight = buffer_h;
    tp.padding = padding;
    std::vector<ROI> rois;
    mvsUtils::getTileRoiList(tp, image_w, image_h, max_downscale, rois);
    for(size_t i = 0; i < rois.size() && (int)i < cap; ++i)
    {
        out[4 * i] = (int)rois[i].x.begin, out[4 * i + 1] = (int)rois[i].x.end;
        out[4 * i + 2] = (int)rois[i].y.begin, out[4 * i + 3] = (int)rois[i].y.end;
    }
    return (int)rois.size();
}

// addSingleTileMapWeighted (mapIO.cpp:206-311) on a tile of ones: the weight every pixel of the tile is multiplied with before it is
// added to the full map.  out_w: (roi / downscale) pixels, row-major; out_sum: the full map after the addition ((image / downscale) pixels).
int avref_tile_weight_map(int image_w, int image_h, const int roi[4], int padding, int downscale, float* out_w, float* out_sum)
{
    try
    {
        mvsUtils::MultiViewParams mp;
        mp.widths.push_back(image_w);
        mp.heights.push_back(image_h);
        mvsUtils::TileParams tp;
        tp.padding = padding;
        const ROI r((unsigned)roi[0], (unsigned)roi[1], (unsigned)roi[2], (unsigned)roi[3]);
        const ROI d = downscaleROI(r, downscale);
        image::Image<float> tile((int)d.width(), (int)d.height(), true, 1.0f);
        const int fw = (image_w + downscale - 1) / downscale, fh = (image_h + downscale - 1) / downscale;
        image::Image<float> full(fw, fh, true, 0.0f);
        mvsUtils::addSingleTileMapWeightedFloat(0, mp, tp, r, downscale, tile, full);
        std::copy_n(tile.data(), (size_t)tile.size(), out_w);
        if(out_sum != nullptr)
            std::copy_n(full.data(), (size_t)full.size(), out_sum);
        return 0;
    }
    catch(const std::exception& e)
    {
        std::cerr << "[ref] avref_tile_weight_map: " << e.what() << std::endl;
        return 2;
    }
}

} // extern "C"
