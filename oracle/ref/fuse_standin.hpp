// oracle/_ref (fuse): STAND-IN declarations for what the reference's depth-map filtering functions use from the rest of AliceVision.
// Test infrastructure only.  The functions under test — Fuser::updateInSurr / filterGroupsRC / filterDepthMapsRC (fuseCut/Fuser.cpp),
// MultiViewParams::getPixelFor3DPoint / getCamPixelSize* / isPixelInImage / decomposeProjectionMatrix (mvsUtils/MultiViewParams.cpp),
// get2dLineImageIntersection / getTarEpipolarDirectedLine / triangulateMatch (mvsUtils/common.cpp) — are compiled from the reference's
// own text (gen_extract.py), and mvsData (Point3d, Point2d, Pixel, Matrix3x3, Matrix3x4, StaticVector, geometry.cpp) is the reference's
// own code included where it lies.  What is declared HERE is only their environment, none of which computes anything:
//   * MultiViewParams: the camera arrays as plain members (the reference fills them in loadMatricesFromRawProjectionMatrix,
//     MultiViewParams.cpp:283-297; the driver fills them from the arrays the test hands to BOTH sides), image sizes, g_border = 2
//     (MultiViewParams.hpp:111), and findNearestCamsFromLandmarks answering with the T-camera list of the test;
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

#include <aliceVision/system/Logger.hpp>
#include <aliceVision/mvsData/Point2d.hpp>
#include <aliceVision/mvsData/Point3d.hpp>
#include <aliceVision/mvsData/Pixel.hpp>
#include <aliceVision/mvsData/Matrix3x3.hpp>
#include <aliceVision/mvsData/Matrix3x4.hpp>
#include <aliceVision/mvsData/StaticVector.hpp>
#include <aliceVision/mvsData/geometry.hpp>

namespace aliceVision {

namespace image {
template <class T>
class Image
{
  public:
    Image() = default;
    Image(int width, int height, bool fInit = false, const T val = T()) : _w(width), _h(height), _d((size_t)width * height, fInit ? val : T()) {}
    int width() const { return _w; }
    int height() const { return _h; }
    int size() const { return _w * _h; }
    T& operator()(int y, int x) { return _d[(size_t)y * _w + x]; }
    const T& operator()(int y, int x) const { return _d[(size_t)y * _w + x]; }
    T& operator()(int i) { return _d[i]; }
    const T& operator()(int i) const { return _d[i]; }
    T* data() { return _d.data(); }
    const T* data() const { return _d.data(); }

  private:
    int _w = 0, _h = 0;
    std::vector<T> _d;
};
enum class EImageColorSpace { LINEAR, NO_CONVERSION };
enum class EStorageDataType { Float };
struct ImageWriteOptions
{
    ImageWriteOptions& toColorSpace(EImageColorSpace) { return *this; }
    ImageWriteOptions& storageDataType(EStorageDataType) { return *this; }
};
} // namespace image

namespace mvsUtils {

enum class EFileType { depthMap, simMap, nmodMap, depthMapFiltered, simMapFiltered };

class MultiViewParams
{
  public:
    std::vector<Matrix3x4> camArr;
    std::vector<Matrix3x3> iCamArr;
    std::vector<Point3d> CArr;
    std::vector<int> widths, heights;
    std::vector<int> nearest; // answer of findNearestCamsFromLandmarks
    int g_border = 2;         // MultiViewParams.hpp:111

    int getViewId(int index) const { return index; }
    int getWidth(int index) const { return widths.at(index); }
    int getHeight(int index) const { return heights.at(index); }
    StaticVector<int> findNearestCamsFromLandmarks(int, int) const
    {
        StaticVector<int> out;
        for(int c : nearest)
            out.push_back(c);
        return out;
    }

    // defined by the reference's own text (gen/fuse_MultiViewParams.cpp)
    void getPixelFor3DPoint(Point2d* out, const Point3d& X, const Matrix3x4& P) const;
    void getPixelFor3DPoint(Point2d* out, const Point3d& X, int rc) const;
    void getPixelFor3DPoint(Pixel* out, const Point3d& X, int rc) const;
    double getCamPixelSize(const Point3d& x0, int cam) const;
    double getCamPixelSize(const Point3d& x0, int cam, float d) const;
    double getCamPixelSizeRcTc(const Point3d& p, int rc, int tc, float d) const;
    double getCamPixelSizePlaneSweepAlpha(const Point3d& p, int rc, int tc, int scale, int step) const;
    double getCamPixelSizePlaneSweepAlpha(const Point3d& p, int rc, StaticVector<int>* tcams, int scale, int step) const;
    bool isPixelInImage(const Pixel& pix, int camId, int margin) const;
    bool isPixelInImage(const Pixel& pix, int camId) const;
    bool isPixelInImage(const Point2d& pix, int camId) const;
    bool isPixelInImage(const Point2d& pix, int camId, int margin) const;
    void decomposeProjectionMatrix(Point3d& Co, Matrix3x3& Ro, Matrix3x3& iRo, Matrix3x3& Ko, Matrix3x3& iKo, Matrix3x3& iPo, const Matrix3x4& P) const;
};

// defined by the reference's own text (gen/fuse_common.cpp)
bool get2dLineImageIntersection(Point2d* pFrom, Point2d* pTo, Point2d linePoint1, Point2d linePoint2, const MultiViewParams& mp, int camId);
bool getTarEpipolarDirectedLine(Point2d* pFromTar, Point2d* pToTar, Point2d refpix, int refCam, int tarCam, const MultiViewParams& mp);
bool triangulateMatch(Point3d& out, const Point2d& refpix, const Point2d& tarpix, int refCam, int tarCam, const MultiViewParams& mp);

// ---- the in-memory map store (fuse_driver.cpp) ----
struct MapStore
{
    std::map<std::pair<int, int>, image::Image<float>> f32;
    std::map<std::string, image::Image<unsigned char>> u8;
};
MapStore& store();
inline std::string getFileNameFromIndex(const MultiViewParams&, int index, EFileType fileType, const std::string& = "", int = -1, int = -1)
{
    return std::to_string(index) + ":" + std::to_string((int)fileType);
}
inline void readMap(int rc, const MultiViewParams&, const EFileType fileType, image::Image<float>& out, int = 1, int = 1, const std::string& = "")
{
    const auto it = store().f32.find({rc, (int)fileType});
    out = it == store().f32.end() ? image::Image<float>() : it->second; // a camera without a map reads as an empty image (mapIO.cpp)
}
inline void writeMap(int rc, const MultiViewParams&, const EFileType fileType, const image::Image<float>& in, int = 1, int = 1, const std::string& = "")
{
    store().f32[{rc, (int)fileType}] = in;
}
inline void printfElapsedTime(long, const std::string& = "") {}
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
