// oracle/_ref (host): what mvsUtils/mapIO.cpp includes at its top, for the translation unit gen_extract.py assembles from its two
// tile-merge templates (weightTileBorder, addSingleTileMapWeighted): the reference's TileParams.hpp / ROI.hpp / Point2d.hpp, the numeric
// stand-in (clamp), the stand-in image class and MultiViewParams.  Test infrastructure only.
#pragma once
#include "fuse_standin.hpp"

#include <aliceVision/mvsData/Point2d.hpp>
#include <aliceVision/mvsData/ROI.hpp>
#include <aliceVision/mvsUtils/TileParams.hpp>
#include <aliceVision/numeric/numeric.hpp>
