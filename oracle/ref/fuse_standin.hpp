// oracle/_ref (fuse): STAND-IN declarations for what the reference's depth-map filtering functions use from the rest of AliceVision.
// Test infrastructure only.  The functions under test — Fuser::updateInSurr / filterGroupsRC / filterDepthMapsRC (fuseCut/Fuser.cpp),
// MultiViewParams::getPixelFor3DPoint / getCamPixelSize* / isPixelInImage / decomposeProjectionMatrix (mvsUtils/MultiViewParams.cpp),
// get2dLineImageIntersection / getTarEpipolarDirectedLine / triangulateMatch (mvsUtils/common.cpp) — are compiled from the reference's
// own text (gen_extract.py), and mvsData (Point3d, Point2d, Pixel, Matrix3x3, Matrix3x4, StaticVector, geometry.cpp) is the reference's
// own code included where it lies.  What is declared HERE (and in shim_host/aliceVision/mvsUtils/MultiViewParams.hpp) is only their
// environment, none of which computes anything:
//   * MultiViewParams (shim_host/): the camera arrays as plain members, image sizes, g_border = 2, and findNearestCamsFromLandmarks
//     answering with the T-camera list of the test;
//   * image::Image<T>: row-major pixels with the accessors the functions use (image/Image.hpp is an Eigen matrix);
//   * readMap / writeMap / readImage / writeImageWithFloat / utils::exists: an in-memory store instead of EXR files.
#pragma once

#include <cmath>
#include <cstring>
#include <ctime>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "host_standin.hpp"
#include <aliceVision/image/Image.hpp>   // stand-in (shim_host/)
#include <aliceVision/mvsUtils/fileIO.hpp> // stand-in: getFileNameFromIndex, the key of the in-memory store // the reference's own mvsData / common.hpp declarations + the stand-in MultiViewParams (shim_host/)

namespace aliceVision {


namespace mvsUtils {

// ---- the in-memory map store (fuse_driver.cpp) ----
struct MapStore
{
    std::map<std::pair<int, int>, image::Image<float>> f32;
    std::map<std::string, image::Image<unsigned char>> u8;
};
MapStore& store();
inline void readMap(int rc, const MultiViewParams&, const EFileType fileType, image::Image<float>& out, int = 1, int = 1, const std::string& = "")
{
    const auto it = store().f32.find({rc, (int)fileType});
    out = it == store().f32.end() ? image::Image<float>() : it->second; // a camera without a map reads as an empty image (mapIO.cpp)
}
inline void writeMap(int rc, const MultiViewParams&, const EFileType fileType, const image::Image<float>& in, int = 1, int = 1, const std::string& = "")
{
    store().f32[{rc, (int)fileType}] = in;
}
} // namespace mvsUtils

namespace image {
inline void writeImageWithFloat(const std::string& path, const Image<unsigned char>& im, const ImageWriteOptions&) { mvsUtils::store().u8[path] = im; }
inline void readImage(const std::string& path, Image<unsigned char>& im, EImageColorSpace) { im = mvsUtils::store().u8.at(path); }
} // namespace image

namespace utils {
inline bool exists(const std::string&) { return false; } // no cached nmodMap: filterGroupsRC always computes (Fuser.cpp:146-149)
} // namespace utils

namespace fuseCut {
// fuseCut/Fuser.hpp:21-54, the members the filtering step uses
class Fuser
{
  public:
    const mvsUtils::MultiViewParams& _mp;
    explicit Fuser(const mvsUtils::MultiViewParams& mp) : _mp(mp) {}
    bool filterGroupsRC(int rc, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, int nNearestCams);
    bool filterDepthMapsRC(int rc, int minNumOfModals, int minNumOfModalsWSP2SSP);

  private:
    bool updateInSurr(float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, Point3d& p, int rc, int tc, StaticVector<int>* numOfPtsMap,
                      const image::Image<float>& depthMap, const image::Image<float>& simMap, int scale);
};
} // namespace fuseCut

} // namespace aliceVision
