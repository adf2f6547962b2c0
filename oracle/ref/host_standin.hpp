// oracle/_ref (host): what the reference's own .cpp files include at their top, for the translation units gen_extract.py assembles from
// their function definitions: the reference's own headers where they are self-contained (mvsData, mvsUtils/common.hpp), stand-ins from
// shim_host/ (searched first) where they are not (MultiViewParams.hpp, SfMData.hpp, Logger.hpp).  Test infrastructure only.
#pragma once
#include <cmath>
#include <string>

#include <aliceVision/system/Logger.hpp>
#include <aliceVision/mvsData/geometry.hpp>
#include <aliceVision/mvsData/Matrix3x3.hpp>
#include <aliceVision/mvsData/Matrix3x4.hpp>
#include <aliceVision/mvsData/OrientedPoint.hpp>
#include <aliceVision/mvsData/Pixel.hpp>
#include <aliceVision/mvsUtils/common.hpp>
