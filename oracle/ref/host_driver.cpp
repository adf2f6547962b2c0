// oracle/_ref (host): C entry point around the reference's OWN depth-plane list — depthMap/SgmDepthList.cpp compiled WHOLE and unchanged
// (computeListRc and everything below it: getMinMaxMidNbDepthFromSfM, getRcTcDepthRangeFromSfM, computeRcTcDepths,
// computePixelSizeDepths, computeRcDepthList, indexOfNearestSorted), with MultiViewParams' projection / pixel-size methods and
// common.cpp's epipolar helpers from the reference's text (gen_extract.py) and mvsData as it lies.  Test infrastructure only:
// tests/test_host_ref.py holds oracle/host_oracle.py (which the C++ host equals, tests/test_host_cpu.py) against it.
// Stand-ins: shim_host/ (MultiViewParams as plain arrays, the landmark containers, Boost's tail quantile — the one unpinned piece).
#include <aliceVision/depthMap/SgmDepthList.hpp>

#include <algorithm>

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
        mp.minViewAngle = min_view_angle;
        mp.maxViewAngle = max_view_angle;
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
            lm.X = Vec3{{X[3 * l], X[3 * l + 1], X[3 * l + 2]}};
            for(int o = obs_begin[l]; o < obs_begin[l + 1]; ++o)
                lm.observations[(IndexT)obs_view[o]] = sfmData::Observation{Vec2{{obs_xy[2 * o], obs_xy[2 * o + 1]}}};
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

} // extern "C"
