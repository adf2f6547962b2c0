// oracle/_ref (host): C entry point around the reference's OWN undistortion (see undistort_standin.hpp for what is compiled from where).
// Test infrastructure only: tests/test_host_ref.py holds oracle/avdm_oracle.c's avo_image_undistort against it, pixel for pixel.
#include "undistort_standin.hpp"

#include "avdm.h"

#include <algorithm>
#include <iostream>

namespace aliceVision {
namespace camera {
// camera/cameraUndistortImage.hpp:81-139, from the reference's text (gen/host_undistort_image.cpp)
template <typename T>
void UndistortImage(const image::Image<T>& imageIn, const camera::IntrinsicBase* intrinsicPtr, image::Image<T>& image_ud, T fillcolor,
                    bool correctPrincipalPoint, const oiio::ROI& roi);
void undistortRgbaFloat(const image::Image<image::RGBAfColor>& in, const IntrinsicBase* cam, image::Image<image::RGBAfColor>& out, image::RGBAfColor fill);
} // namespace camera
} // namespace aliceVision

using namespace aliceVision;

extern "C" int avref_image_undistort(float* dst, int dst_pitch, const float* src, int src_pitch, const avdm_intrinsic_t* cam, const float fill[4])
{
    try
    {
        camera::Pinhole k;
        k._w = (unsigned)cam->width, k._h = (unsigned)cam->height;
        k._scale = Vec2(cam->scale_x, cam->scale_y);
        k._offset = Vec2(cam->offset_x, cam->offset_y);
        std::shared_ptr<camera::Distortion> d;
        switch(cam->distortion_model)
        {
            case AVDM_DISTORTION_NONE: break;
            case AVDM_DISTORTION_RADIALK1: d = std::make_shared<camera::DistortionRadialK1>(); d->_distortionParams = {cam->k[0]}; break;
            case AVDM_DISTORTION_RADIALK3: d = std::make_shared<camera::DistortionRadialK3>(); d->_distortionParams = {cam->k[0], cam->k[1], cam->k[2]}; break;
            case AVDM_DISTORTION_RADIALK3PT: d = std::make_shared<camera::DistortionRadialK3PT>(); d->_distortionParams = {cam->k[0], cam->k[1], cam->k[2]}; break;
            default: return 1;
        }
        k._pDistortion = d;
        const int w = cam->width, h = cam->height;
        image::Image<image::RGBAfColor> in(w, h), out;
        for(int y = 0; y < h; ++y)
        {
            const float* row = (const float*)((const char*)src + (long long)y * src_pitch);
            for(int x = 0; x < w; ++x)
                in(y, x) = image::RGBAfColor(row[4 * x], row[4 * x + 1], row[4 * x + 2], row[4 * x + 3]);
        }
        camera::undistortRgbaFloat(in, &k, out, image::RGBAfColor(fill[0], fill[1], fill[2], fill[3]));
        for(int y = 0; y < h; ++y)
        {
            float* row = (float*)((char*)dst + (long long)y * dst_pitch);
            for(int x = 0; x < w; ++x)
                for(int c = 0; c < 4; ++c)
                    row[4 * x + c] = out(y, x).v[c];
        }
        return 0;
    }
    catch(const std::exception& e)
    {
        std::cerr << "[ref] avref_image_undistort: " << e.what() << std::endl;
        return 2;
    }
}
