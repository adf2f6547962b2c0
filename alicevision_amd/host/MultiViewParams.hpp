// MultiViewParams.hpp — cameras of the scene at process resolution, file naming, neighbour-camera selection and the RAM image
// loader.  Restates mvsUtils/MultiViewParams.{hpp,cpp} (constructor :37-246, matrices :278-319, pixel size :375-400,
// findNearestCamsFromLandmarks :519-575, findTileNearestCams :577-667), mvsUtils/fileIO.cpp:18-358 (file names),
// :389-443 (loadImage) and mvsUtils/common.cpp:23-169 (epipolar-line / triangulation helpers).
#pragma once

#include "exr.hpp"
#include "jpeg.hpp"
#include "mvsData.hpp"
#include "sfmData.hpp"

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace avdm_host {

enum class EFileType
{
    none, // image files (the reference's default, EFileType::img)
    depthMap,
    simMap,
    normalMap,
    thicknessMap,
    pixSizeMap,
    tilePattern,
    depthMapFiltered,
    simMapFiltered,
    normalMapFiltered,
    nmodMap,
    volume,
    volumeCross,
    volumeTopographicCut,
    stats9p,
    P,
    D
};

struct ImageParams
{
    IndexT viewId = 0;
    int width = 0, height = 0;
    std::string path;
};

class MultiViewParams
{
  public:
    std::vector<Matrix3x4> camArr;   // P at process resolution
    std::vector<Matrix3x3> KArr, iKArr, RArr, iRArr, iCamArr;
    std::vector<Point3d> CArr;
    int ncams = 0;
    int g_border = 2;

    // MultiViewParams.cpp:37-246.  fileType == depthMap: the per-view file whose metadata (downscale, P) is read is the depth map of
    // the view — from depthMapsFolder, or the filtered one from depthMapsFilterFolder when depthMapsFolder is empty (:66-78)
    MultiViewParams(const SfMData& sfmData, const std::string& imagesFolder, const std::string& depthMapsFolder, const std::string& depthMapsFilterFolder,
                    EFileType fileType, int downscale = 1);
    MultiViewParams(const SfMData& sfmData, const std::string& imagesFolder, const std::string& depthMapsFolder, int downscale)
      : MultiViewParams(sfmData, imagesFolder, depthMapsFolder, "", EFileType::none, downscale)
    {}

    const SfMData& getInputSfMData() const { return _sfmData; }
    const std::string& getImagesFolder() const { return _imagesFolder; }
    const std::string& getDepthMapsFolder() const { return _depthMapsFolder; }
    const std::string& getDepthMapsFilterFolder() const { return _depthMapsFilterFolder; }
    const std::string& getImagePath(int index) const { return _imagesParams.at(index).path; }
    IndexT getViewId(int index) const { return _imagesParams.at(index).viewId; }
    int getOriginalWidth(int index) const { return _imagesParams.at(index).width; }
    int getOriginalHeight(int index) const { return _imagesParams.at(index).height; }
    int getWidth(int index) const { return _imagesParams.at(index).width / getDownscaleFactor(index); }
    int getHeight(int index) const { return _imagesParams.at(index).height / getDownscaleFactor(index); }
    int getDownscaleFactor(int index) const { return _imagesScale.at(index) * _processDownscale; }
    int getProcessDownscale() const { return _processDownscale; }
    int getMaxImageWidth() const { return _maxImageWidth / _processDownscale; }
    int getMaxImageHeight() const { return _maxImageHeight / _processDownscale; }
    int getNbCameras() const { return (int)_imagesParams.size(); }
    // unique per constructed object (process-wide counter): caches keyed by the object's address compare it too
    unsigned long long generation() const { return _generation; }
    int getIndexFromViewId(IndexT viewId) const { return _imageIdsPerViewId.at(viewId); }
    float getMinViewAngle() const { return _minViewAngle; }
    float getMaxViewAngle() const { return _maxViewAngle; }
    void setMinViewAngle(float a) { _minViewAngle = a; }
    void setMaxViewAngle(float a) { _maxViewAngle = a; }
    const std::map<std::string, std::string>& getMetadata(int index) const { return _sfmData.views.at(getViewId(index)).metadata; }
    // MultiViewParams.hpp:170-182: P at scale 1 as a 4x4 (row-major, last row 0 0 0 1)
    std::vector<double> getOriginalP(int index) const;

    void getPixelFor3DPoint(Point2d* out, const Point3d& X, const Matrix3x4& P) const;
    void getPixelFor3DPoint(Point2d* out, const Point3d& X, int rc) const { getPixelFor3DPoint(out, X, camArr[rc]); }
    double getCamPixelSize(const Point3d& x0, int cam, float d) const;
    bool isPixelInImage(const Pixel& pix, int camId, int margin) const
    {
        return (pix.x >= margin) && (pix.x < getWidth(camId) - margin) && (pix.y >= margin) && (pix.y < getHeight(camId) - margin);
    }
    bool isPixelInImage(const Point2d& pix, int camId) const { return isPixelInImage(Pixel(pix), camId, g_border); }
    void decomposeProjectionMatrix(Point3d& Co, Matrix3x3& Ro, Matrix3x3& iRo, Matrix3x3& Ko, Matrix3x3& iKo, Matrix3x3& iPo, const Matrix3x4& P) const;

    std::vector<int> findNearestCamsFromLandmarks(int rc, int nbNearestCams) const;
    std::vector<int> findTileNearestCams(int rc, int nbNearestCams, const std::vector<int>& tCams, const ROI& roi) const;

  private:
    const SfMData& _sfmData;
    std::string _imagesFolder, _depthMapsFolder, _depthMapsFilterFolder;
    int _processDownscale = 1;
    unsigned long long _generation = nextGeneration();
    static unsigned long long nextGeneration();
    float _minViewAngle = 2.0f, _maxViewAngle = 70.0f;
    std::vector<ImageParams> _imagesParams;
    std::vector<int> _imagesScale;
    std::map<IndexT, int> _imageIdsPerViewId;
    int _maxImageWidth = 0, _maxImageHeight = 0;

    void loadMatricesFromRawProjectionMatrix(int index, const double* rawProjMatrix);
    void loadMatricesFromSfM(int index);
};

// mvsUtils/fileIO.cpp:18-358
std::string getFileNameFromViewId(const MultiViewParams& mp, IndexT viewId, EFileType fileType, const std::string& customSuffix = "", int tileBeginX = -1,
                                  int tileBeginY = -1);
inline std::string getFileNameFromIndex(const MultiViewParams& mp, int index, EFileType fileType, const std::string& customSuffix = "", int tileBeginX = -1,
                                        int tileBeginY = -1)
{
    return getFileNameFromViewId(mp, mp.getViewId(index), fileType, customSuffix, tileBeginX, tileBeginY);
}

// mvsUtils/common.cpp:23-116, :155-169
bool get2dLineImageIntersection(Point2d* pFrom, Point2d* pTo, Point2d linePoint1, Point2d linePoint2, const MultiViewParams& mp, int camId);
bool triangulateMatch(Point3d& out, const Point2d& refpix, const Point2d& tarpix, int refCam, int tarCam, const MultiViewParams& mp);

// linear RGBA float image at process resolution (image::Image<RGBAfColor>), interleaved
struct HostImage
{
    int width = 0, height = 0;       // size the process works at: source size / --downscale (integer division, imageAlgo.cpp:326-368)
    int srcWidth = 0, srcHeight = 0; // size of `rgba` as decoded; the --downscale resize runs on the device (avdm_image_resize)
    std::vector<float> rgba;
    // an integer file (PNG): the decoder's samples as stored; DeviceMipmapImage::fill uploads THESE (a quarter / half of the float RGBA bytes)
    // and converts them to linear float RGBA on the device (avdm_image_decode_integer).  `rgba` is empty then.
    std::vector<unsigned char> raw;
    int rawChannels = 0, rawBits = 0;
    // a JPEG file: the entropy-decoded coefficients (host/jpeg.cpp); the inverse DCT, up-sampling, colour conversion and the sRGB decoding run
    // on the device (avdm_image_decode_jpeg + avdm_image_decode_integer).  `rgba` and `raw` are empty then.
    std::shared_ptr<const JpegImage> jpeg;
    // an OpenEXR file: its scan lines as stored (exr.hpp ExrLines: mapped when uncompressed, inflated otherwise); the de-interleave to float
    // RGBA runs on the device (avdm_image_decode_exr_lines).  `rgba` is empty then.  AVDM_HOST_EXR=host: rounds 1-5's form (`rgba` filled here).
    std::shared_ptr<const ExrLines> exrLines;
};
// mvsUtils/fileIO.cpp:389-443 loadImage + mvsUtils/ImagesCache.hpp: a small thread-safe RAM cache keyed by camera index
class ImagesCache
{
  public:
    explicit ImagesCache(const MultiViewParams& mp, size_t maxImages = 32) : _mp(mp), _max(maxImages) {}
    std::shared_ptr<const HostImage> getImg_sync(int camId);

  private:
    const MultiViewParams& _mp;
    size_t _max;
    std::mutex _mutex;
    std::map<int, std::pair<long, std::shared_ptr<const HostImage>>> _cache;
    long _tick = 0;
};

} // namespace avdm_host
