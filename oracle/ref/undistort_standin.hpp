// oracle/_ref (host): STAND-IN declarations around the reference's undistortion code — camera::UndistortImage
// (camera/cameraUndistortImage.hpp:81-139), IntrinsicScaleOffsetDisto::getDistortedPixel (IntrinsicScaleOffsetDisto.cpp:80),
// IntrinsicScaleOffset::cam2ima / ima2cam (IntrinsicScaleOffset.cpp:31-66), DistortionRadialK1 / K3 / K3PT::addDistortion
// (DistortionRadial.cpp:18-24, 110-124, 262-277), all compiled from the reference's text (gen_extract.py), and the bilinear sampler of
// image/Sampler.hpp, included as it lies.  Declared here: the class skeletons those methods belong to (members as in the reference's
// headers; the two header-inline one-liners getPrincipalPoint, IntrinsicScaleOffset.hpp:44-51, and the addDistortion forwarder,
// IntrinsicScaleOffsetDisto.hpp:75-86, are restated), the ROI argument that prepareDenseScene leaves undefined, vectors and pixels
// (shim_host/).  Test infrastructure only.
#pragma once
#include <cmath>
#include <memory>
#include <vector>

#include <aliceVision/system/Logger.hpp>
#include <aliceVision/numeric/standin_vec.hpp>
#include <aliceVision/image/Image.hpp>
#include <aliceVision/image/Sampler.hpp> // the reference's own

namespace oiio {
struct ROI
{
    int xbegin = 0, ybegin = 0, w = 0, h = 0;
    bool defined() const { return false; }
    int width() const { return w; }
    int height() const { return h; }
};
} // namespace oiio

namespace aliceVision {
namespace camera {

enum EINTRINSIC { PINHOLE_CAMERA_RADIAL3 };
inline bool isPinhole(EINTRINSIC) { return true; }

class Distortion
{
  public:
    virtual ~Distortion() = default;
    virtual Vec2 addDistortion(const Vec2& p) const { return p; }
    std::vector<double> _distortionParams;
};
class DistortionRadialK1 : public Distortion
{
  public:
    Vec2 addDistortion(const Vec2& p) const override;
};
class DistortionRadialK3 : public Distortion
{
  public:
    Vec2 addDistortion(const Vec2& p) const override;
};
class DistortionRadialK3PT : public Distortion
{
  public:
    Vec2 addDistortion(const Vec2& p) const override;
};

class IntrinsicBase
{
  public:
    virtual ~IntrinsicBase() = default;
    virtual bool hasDistortion() const { return false; }
    virtual Vec2 getDistortedPixel(const Vec2& p) const = 0;
    EINTRINSIC getType() const { return PINHOLE_CAMERA_RADIAL3; }
    unsigned int _w = 0, _h = 0;
};
class IntrinsicScaleOffset : public IntrinsicBase
{
  public:
    Vec2 cam2ima(const Vec2& p) const;
    Vec2 ima2cam(const Vec2& p) const;
    const Vec2 getPrincipalPoint() const // IntrinsicScaleOffset.hpp:44-51
    {
        Vec2 ret = _offset;
        ret(0) += static_cast<double>(_w) * 0.5;
        ret(1) += static_cast<double>(_h) * 0.5;
        return ret;
    }
    Vec2 _scale{1.0, 1.0}, _offset{0.0, 0.0};
};
class IntrinsicScaleOffsetDisto : public IntrinsicScaleOffset
{
  public:
    bool hasDistortion() const override { return _pDistortion != nullptr; }
    Vec2 addDistortion(const Vec2& p) const // IntrinsicScaleOffsetDisto.hpp:75-86 (no undistortion object here)
    {
        if(_pDistortion)
            return _pDistortion->addDistortion(p);
        return p;
    }
    Vec2 getDistortedPixel(const Vec2& p) const override;
    std::shared_ptr<Distortion> _pDistortion;
};
class Pinhole : public IntrinsicScaleOffsetDisto
{
};

} // namespace camera
} // namespace aliceVision
