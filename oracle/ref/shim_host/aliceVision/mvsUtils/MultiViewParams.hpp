// oracle/_ref (host): STAND-IN for aliceVision/mvsUtils/MultiViewParams.hpp.  Test infrastructure only.
// The class as the code under test sees it: the camera arrays as plain members (the reference fills them in
// loadMatricesFromRawProjectionMatrix, MultiViewParams.cpp:283-297; the drivers fill them the same way, with the reference's own
// Matrix3x4::decomposeProjectionMatrix), image sizes, the view-angle limits, the landmarks, g_border = 2 (MultiViewParams.hpp:111).
// The METHODS that compute — getPixelFor3DPoint, getCamPixelSize*, isPixelInImage, decomposeProjectionMatrix — are only declared here:
// their definitions are the reference's own text (gen_extract.py -> oracle/_ref/gen/fuse_MultiViewParams.cpp).
#pragma once

#include <string>
#include <vector>

#include <aliceVision/mvsData/Matrix3x3.hpp>
#include <aliceVision/mvsData/Matrix3x4.hpp>
#include <aliceVision/mvsData/Pixel.hpp>
#include <aliceVision/mvsData/Point2d.hpp>
#include <aliceVision/mvsData/Point3d.hpp>
#include <aliceVision/mvsData/ROI.hpp>
#include <aliceVision/mvsData/StaticVector.hpp>
#include <aliceVision/mvsData/structures.hpp>
#include <aliceVision/sfmData/SfMData.hpp>

namespace aliceVision {
namespace mvsUtils {

// the file kinds the code under test names (MultiViewParams.hpp:33-84 of the reference; only ever compared or passed on here)
enum class EFileType
{
    depthMap, depthMapFiltered, simMap, simMapFiltered, normalMap, normalMapFiltered, thicknessMap, pixSizeMap, nmodMap, volume, volumeCross,
    volumeTopographicCut, stats9p, tilePattern, none
};

class MultiViewParams
{
  public:
    std::vector<Matrix3x4> camArr;
    std::vector<Matrix3x3> KArr, iKArr, RArr, iRArr, iCamArr;
    std::vector<Point3d> CArr;
    std::vector<int> widths, heights, viewIds;
    std::vector<int> nearest; // answer of findNearestCamsFromLandmarks (fuse driver)
    sfmData::SfMData sfm;
    const sfmData::SfMData& _sfmData = sfm; // the member name the reference's methods use
    int processDownscale = 1;
    float _minViewAngle = 2.0f, _maxViewAngle = 70.0f;
    int g_border = 2; // MultiViewParams.hpp:111
    MultiViewParams() = default;
    MultiViewParams(const MultiViewParams&) = delete;

    int getNbCameras() const { return (int)camArr.size(); }
    int getViewId(int index) const { return viewIds.empty() ? index : viewIds.at(index); }
    int getWidth(int index) const { return widths.at(index); }
    int getHeight(int index) const { return heights.at(index); }
    int getProcessDownscale() const { return processDownscale; }
    float getMinViewAngle() const { return _minViewAngle; }
    float getMaxViewAngle() const { return _maxViewAngle; }
    int getIndexFromViewId(IndexT viewId) const { return (int)viewId; }
    const sfmData::SfMData& getInputSfMData() const { return sfm; }
    std::string getDepthMapsFolder() const { return "/tmp/"; }
    StaticVector<int> findNearestCamsFromLandmarks(int, int) const
    {
        StaticVector<int> out;
        for(int c : nearest)
            out.push_back(c);
        return out;
    }

    // defined by the reference's own text; the landmark ranking under another name (Makefile: -D on that one generated file) because the
    // filtering driver answers findNearestCamsFromLandmarks with the list of its test
    StaticVector<int> findNearestCamsFromLandmarksRef(int rc, int nbNearestCams) const;
    std::vector<int> findTileNearestCams(int rc, int nbNearestCams, const std::vector<int>& tCams, const ROI& roi) const;
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

} // namespace mvsUtils
} // namespace aliceVision
