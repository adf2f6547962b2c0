// oracle/_ref (fuse): C entry points around the reference's OWN depth-map filtering functions (fuseCut/Fuser.cpp:66-304 and the helpers
// they call, compiled from the reference's text — see fuse_standin.hpp / gen_extract.py).  Test infrastructure only: tests/test_fuse_ref.py
// holds oracle/avdm_fuse_oracle.c against these, call for call, on the same arrays.  Same argument lists as the oracle's
// avo_fuse_filter_groups_rc / avo_fuse_filter_depth_maps_rc / avo_fuse_pixel_size_plane_sweep_alpha.
#include "fuse_standin.hpp"

#include <algorithm>

namespace aliceVision {
namespace mvsUtils {
MapStore& store()
{
    static thread_local MapStore s;
    return s;
}
} // namespace mvsUtils
} // namespace aliceVision

using namespace aliceVision;

extern "C" {

struct avref_fuse_cam_t // = avo_fuse_cam_t
{
    double P[12], iP[9], C[3];
    int width, height;
};

static void add_camera(mvsUtils::MultiViewParams& mp, const avref_fuse_cam_t& c)
{
    Matrix3x4 P;
    std::copy_n(c.P, 12, P.m);
    Matrix3x3 iP;
    std::copy_n(c.iP, 9, iP.m);
    mp.camArr.push_back(P);
    mp.iCamArr.push_back(iP);
    mp.CArr.push_back(Point3d(c.C[0], c.C[1], c.C[2]));
    mp.widths.push_back(c.width);
    mp.heights.push_back(c.height);
}

static image::Image<float> to_image(const float* p, int w, int h)
{
    image::Image<float> im(w, h);
    std::copy_n(p, (size_t)w * h, im.data());
    return im;
}

// Fuser::filterGroupsRC(rc = 0, ...) with the T cameras 1..n_tc; tc_depth[c] == NULL: that camera has no depth map on disk
int avref_fuse_filter_groups_rc(unsigned char* nmod, const float* depth, const float* sim, const avref_fuse_cam_t* rc, int n_tc, const avref_fuse_cam_t* tcs,
                                const float* const* tc_depth, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP)
{
    try
    {
        mvsUtils::MultiViewParams mp;
        add_camera(mp, *rc);
        mvsUtils::MapStore& st = mvsUtils::store();
        st.f32.clear();
        st.u8.clear();
        st.f32[{0, (int)mvsUtils::EFileType::depthMap}] = to_image(depth, rc->width, rc->height);
        st.f32[{0, (int)mvsUtils::EFileType::simMap}] = to_image(sim, rc->width, rc->height);
        for(int c = 0; c < n_tc; ++c)
        {
            add_camera(mp, tcs[c]);
            mp.nearest.push_back(c + 1);
            if(tc_depth[c] != nullptr)
                st.f32[{c + 1, (int)mvsUtils::EFileType::depthMap}] = to_image(tc_depth[c], tcs[c].width, tcs[c].height);
        }
        fuseCut::Fuser fuser(mp);
        fuser.filterGroupsRC(0, pixToleranceFactor, pixSizeBall, pixSizeBallWSP, n_tc);
        const image::Image<unsigned char>& out = st.u8.at(mvsUtils::getFileNameFromIndex(mp, 0, mvsUtils::EFileType::nmodMap));
        std::copy_n(out.data(), (size_t)rc->width * rc->height, nmod);
        return 0;
    }
    catch(const std::exception& e)
    {
        std::cerr << "[ref] avref_fuse_filter_groups_rc: " << e.what() << std::endl;
        return 1;
    }
}

// Fuser::filterDepthMapsRC on maps of n pixels (the function is pixel-wise: the maps are handed over as n x 1 images)
int avref_fuse_filter_depth_maps_rc(float* depthMap, float* simMap, const unsigned char* numOfModalsMap, size_t n, int minNumOfModals,
                                    int minNumOfModalsWSP2SSP)
{
    try
    {
        mvsUtils::MultiViewParams mp;
        avref_fuse_cam_t cam{};
        cam.width = (int)n;
        cam.height = 1;
        add_camera(mp, cam);
        mvsUtils::MapStore& st = mvsUtils::store();
        st.f32.clear();
        st.u8.clear();
        st.f32[{0, (int)mvsUtils::EFileType::depthMap}] = to_image(depthMap, (int)n, 1);
        st.f32[{0, (int)mvsUtils::EFileType::simMap}] = to_image(simMap, (int)n, 1);
        image::Image<unsigned char> nm((int)n, 1);
        std::copy_n(numOfModalsMap, n, nm.data());
        st.u8[mvsUtils::getFileNameFromIndex(mp, 0, mvsUtils::EFileType::nmodMap)] = nm;
        fuseCut::Fuser fuser(mp);
        fuser.filterDepthMapsRC(0, minNumOfModals, minNumOfModalsWSP2SSP);
        const image::Image<float>& d = st.f32.at({0, (int)mvsUtils::EFileType::depthMapFiltered});
        const image::Image<float>& s = st.f32.at({0, (int)mvsUtils::EFileType::simMapFiltered});
        std::copy_n(d.data(), n, depthMap);
        std::copy_n(s.data(), n, simMap);
        return 0;
    }
    catch(const std::exception& e)
    {
        std::cerr << "[ref] avref_fuse_filter_depth_maps_rc: " << e.what() << std::endl;
        return 1;
    }
}

// The camera arrays as the reference derives them from a projection matrix (MultiViewParams::loadMatricesFromRawProjectionMatrix,
// MultiViewParams.cpp:293-296): K, R, C = decomposeProjectionMatrix(P), iCamArr = R^-1 * K^-1.  The epipolar helpers decompose P again
// (common.cpp:119-137) and expect to find these very values; a test that fed both sides arrays formed in numpy would compare two
// different roundings of the same camera.
void avref_fuse_camera_from_projection(const double P[12], int width, int height, avref_fuse_cam_t* out)
{
    Matrix3x4 pMatrix;
    std::copy_n(P, 12, pMatrix.m);
    Matrix3x3 K, R;
    Point3d Cc;
    pMatrix.decomposeProjectionMatrix(K, R, Cc);
    const Matrix3x3 iK = K.inverse();
    const Matrix3x3 iR = R.inverse();
    const Matrix3x3 iCam = iR * iK;
    std::copy_n(pMatrix.m, 12, out->P);
    std::copy_n(iCam.m, 9, out->iP);
    out->C[0] = Cc.x, out->C[1] = Cc.y, out->C[2] = Cc.z;
    out->width = width;
    out->height = height;
}

// MultiViewParams::getCamPixelSizePlaneSweepAlpha(p, rc, tc, scale = 1, step = 1), the tolerance of updateInSurr (Fuser.cpp:101)
double avref_fuse_pixel_size_plane_sweep_alpha(const double p[3], const avref_fuse_cam_t* rc, const avref_fuse_cam_t* tc)
{
    mvsUtils::MultiViewParams mp;
    add_camera(mp, *rc);
    add_camera(mp, *tc);
    return mp.getCamPixelSizePlaneSweepAlpha(Point3d(p[0], p[1], p[2]), 0, 1, 1, 1);
}

} // extern "C"
