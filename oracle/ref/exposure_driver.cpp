// oracle/_ref: the reference's own sfmData/ExposureSetting.hpp (header-only, included where it lies) behind a C entry point — the
// exposure of a view from its shutter / aperture / ISO, as aliceVision_prepareDenseScene's AliceVision:EV / EVComp metadata need it.
// Test infrastructure (tests/test_host_ref.py); nothing in the product links this.
#include <aliceVision/sfmData/ExposureSetting.hpp>

extern "C" double avr_exposure(double shutter, double fnumber, double iso)
{
    return aliceVision::sfmData::ExposureSetting(shutter, fnumber, iso).getExposure();
}
extern "C" int avr_exposure_partially_defined(double shutter, double fnumber, double iso)
{
    return aliceVision::sfmData::ExposureSetting(shutter, fnumber, iso).isPartiallyDefined() ? 1 : 0;
}
