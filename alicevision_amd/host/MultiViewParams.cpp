// MultiViewParams.cpp — see MultiViewParams.hpp for the reference lines restated.
#include "MultiViewParams.hpp"
#include "png.hpp"
#include "tiff.hpp"

#include <cctype>

#include "log.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cfloat>
#include <cmath>
#include <set>

namespace avdm_host {

namespace {
// both paths exist: do they name the same file (device + inode)?
bool sameFile(const std::string& a, const std::string& b)
{
    struct stat sa, sb;
    return ::stat(a.c_str(), &sa) == 0 && ::stat(b.c_str(), &sb) == 0 && sa.st_dev == sb.st_dev && sa.st_ino == sb.st_ino;
}
} // namespace


namespace {
bool fileExists(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
bool dirExists(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
struct SortedId
{
    int id;
    float value;
};
std::string lowerExtension(const std::string& path)
{
    const size_t dot = path.rfind('.');
    std::string e = dot == std::string::npos ? "" : path.substr(dot);
    for(char& c : e)
        c = (char)std::tolower((unsigned char)c);
    return e;
}
bool isPngPath(const std::string& path) { return lowerExtension(path) == ".png"; }
bool isTiffPath(const std::string& path)
{
    const std::string e = lowerExtension(path);
    return e == ".tif" || e == ".tiff";
}
bool isJpegPath(const std::string& path)
{
    const std::string e = lowerExtension(path);
    return e == ".jpg" || e == ".jpeg";
}
} // namespace

unsigned long long MultiViewParams::nextGeneration()
{
    static std::atomic<unsigned long long> counter{0};
    return ++counter;
}

MultiViewParams::MultiViewParams(const SfMData& sfmData, const std::string& imagesFolder, const std::string& depthMapsFolder,
                                 const std::string& depthMapsFilterFolder, EFileType fileType, int downscale)
  : _sfmData(sfmData),
    _imagesFolder(imagesFolder + "/"),
    _depthMapsFolder(depthMapsFolder + "/"),
    _depthMapsFilterFolder(depthMapsFilterFolder + "/"),
    _processDownscale(downscale)
{
    // image uid, path and dimensions (MultiViewParams.cpp:52-110)
    {
        std::set<std::pair<int, int>> dimensions;
        int i = 0;
        for(const auto& viewPair : sfmData.views)
        {
            const View& view = viewPair.second;
            if(!sfmData.isPoseAndIntrinsicDefined(view))
                continue;
            std::string path = view.path;
            if(fileType == EFileType::depthMap)
                path = getFileNameFromViewId(*this, view.viewId, depthMapsFolder.empty() ? EFileType::depthMapFiltered : EFileType::depthMap);
            else if(fileType == EFileType::normalMap)
                path = getFileNameFromViewId(*this, view.viewId, EFileType::normalMap);
            else if(_imagesFolder != "/" && dirExists(_imagesFolder))
            {
                // one file per view named <viewId>.<ext> (MultiViewParams.cpp:83-103: exactly one file with a supported extension); this build
                // decodes OpenEXR (PrepareDenseScene's output format), PNG, JPEG and TIFF
                std::string candidate;
                // (each extension in lower and upper case, the spellings isPngPath / isJpegPath / isTiffPath and prepareDenseScene accept)
                for(const char* ext : {".exr", ".EXR", ".png", ".PNG", ".jpg", ".JPG", ".jpeg", ".JPEG", ".tif", ".TIF", ".tiff", ".TIFF"})
                {
                    const std::string c = _imagesFolder + std::to_string(view.viewId) + ext;
                    if(!fileExists(c))
                        continue;
                    // (two spellings that name ONE file — a case-insensitive or case-folding directory — are not an ambiguity)
                    if(!candidate.empty() && !sameFile(candidate, c))
                        throw std::runtime_error("Ambiguous case: Multiple image file found for the view '" + std::to_string(view.viewId) + "' in folder '" +
                                                 _imagesFolder + "'.");
                    if(candidate.empty())
                        candidate = c;
                }
                if(candidate.empty())
                    throw std::runtime_error("Cannot find image file coresponding to the view '" + std::to_string(view.viewId) + "' in folder '" +
                                             _imagesFolder + "' (expected " + std::to_string(view.viewId) + ".exr, .png, .jpg or .tif).");
                path = candidate;
            }
            dimensions.emplace(view.width, view.height);
            ImageParams ip;
            ip.viewId = view.viewId;
            ip.width = view.width;
            ip.height = view.height;
            ip.path = path;
            _imagesParams.push_back(ip);
            _imageIdsPerViewId[view.viewId] = i;
            ++i;
        }
        AVDM_LOG_INFO("Found " << dimensions.size() << " image dimension(s): ");
        for(const auto& dim : dimensions)
            AVDM_LOG_INFO("\t- [" << dim.first << "x" << dim.second << "]");
    }

    ncams = getNbCameras();
    camArr.resize(ncams);
    KArr.resize(ncams);
    iKArr.resize(ncams);
    RArr.resize(ncams);
    iRArr.resize(ncams);
    iCamArr.resize(ncams);
    CArr.resize(ncams);
    _imagesScale.assign(ncams, 1);

    for(int i = 0; i < ncams; ++i)
    {
        const ImageParams& imgParams = _imagesParams.at(i);
        ExrImage header;
        bool exists = fileExists(imgParams.path);
        if(exists && isPngPath(imgParams.path))
        {
            // a PNG carries no AliceVision metadata here: its size gives the scale, the camera comes from the SfMData
            PngImage png;
            readPng(imgParams.path, png, true);
            header.width = png.width;
            header.height = png.height;
        }
        else if(exists && isTiffPath(imgParams.path))
        {
            TiffImage tiff;
            readTiff(imgParams.path, tiff, true);
            header.width = tiff.width;
            header.height = tiff.height;
        }
        else if(exists && isJpegPath(imgParams.path))
        {
            JpegImage jpeg;
            readJpeg(imgParams.path, jpeg, true);
            header.width = jpeg.width;
            header.height = jpeg.height;
        }
        else if(exists)
        {
            try
            {
                readExr(imgParams.path, header, true);
            }
            catch(const std::exception&)
            {
                exists = false; // not an EXR (e.g. the original JPEG of the view): fall back to the SfMData like a missing file
            }
        }
        int scaleMeta = 0;
        double rawP[16];
        if(exists && header.attributes.getInt("AliceVision:downscale", scaleMeta))
            _imagesScale.at(i) = scaleMeta;
        else if(exists)
        {
            const int widthScale = imgParams.width / header.width, heightScale = imgParams.height / header.height;
            if(widthScale != heightScale)
                throw std::runtime_error("Scale of file: '" + imgParams.path + "' is not uniform, check image dimension ratio.");
            _imagesScale.at(i) = widthScale;
        }
        if(exists && header.attributes.getM44d("AliceVision:P", rawP))
        {
            AVDM_LOG_DEBUG("Reading view " << getViewId(i) << " projection matrix from image metadata.");
            loadMatricesFromRawProjectionMatrix(i, rawP);
        }
        else
        {
            AVDM_LOG_DEBUG("Reading view " << getViewId(i) << " projection matrix from SfMData.");
            loadMatricesFromSfM(i);
        }
        if(KArr[i](0, 0) > (float)(getWidth(i) * 100))
            throw std::runtime_error("Camera " + std::to_string(i) + " at infinity."); // the reference zeroes such cameras (:183-235); refuse instead
        _maxImageWidth = std::max(_maxImageWidth, imgParams.width / _imagesScale.at(i));
        _maxImageHeight = std::max(_maxImageHeight, imgParams.height / _imagesScale.at(i));
    }
    AVDM_LOG_INFO("Overall maximum dimension: [" << _maxImageWidth << "x" << _maxImageHeight << "]");
}

// MultiViewParams.cpp:283-298
void MultiViewParams::loadMatricesFromRawProjectionMatrix(int index, const double* rawProjMatrix)
{
    Matrix3x4& P = camArr.at(index);
    std::copy_n(rawProjMatrix, 12, P.m);
    const double imgScale = double(_imagesScale.at(index) * _processDownscale);
    for(int i = 0; i < 8; ++i)
        P.m[i] /= imgScale;
    P.decomposeProjectionMatrix(KArr.at(index), RArr.at(index), CArr.at(index));
    iKArr.at(index) = KArr.at(index).inverse();
    iRArr.at(index) = RArr.at(index).inverse();
    iCamArr.at(index) = iRArr.at(index) * iKArr.at(index);
}

// MultiViewParams.cpp:300-319: P = K [R | -R C] for pinhole intrinsics (Pinhole::getProjectiveEquivalent, camera/Pinhole.cpp:277-283)
void MultiViewParams::loadMatricesFromSfM(int index)
{
    const View& view = _sfmData.views.at(getViewId(index));
    const Intrinsic& intr = _sfmData.getIntrinsic(view);
    const Pose pose = _sfmData.getPose(view);
    const Point3d t = (pose.rotation * pose.center) * -1.0;
    Matrix3x4 P;
    if(intr.isPinhole)
        P = composeP(intr.K(), pose.rotation, t);
    else
    {
        Matrix3x3 I = Matrix3x3::diag(1, 1, 1);
        P = composeP(I, pose.rotation, t);
    }
    loadMatricesFromRawProjectionMatrix(index, P.m);
}

std::vector<double> MultiViewParams::getOriginalP(int index) const
{
    const Matrix3x4& p34 = camArr.at(index);
    const int downscale = getDownscaleFactor(index);
    std::vector<double> p44(p34.m, p34.m + 12);
    for(int i = 0; i < 8; ++i)
        p44[i] *= downscale;
    p44.push_back(0);
    p44.push_back(0);
    p44.push_back(0);
    p44.push_back(1);
    return p44;
}

// MultiViewParams.cpp:335-349
void MultiViewParams::getPixelFor3DPoint(Point2d* out, const Point3d& X, const Matrix3x4& P) const
{
    const Point3d XT = P * X;
    if(XT.z <= 0)
    {
        out->x = -1.0;
        out->y = -1.0;
    }
    else
    {
        out->x = XT.x / XT.z;
        out->y = XT.y / XT.z;
    }
}

// MultiViewParams.cpp:385-400
double MultiViewParams::getCamPixelSize(const Point3d& x0, int cam, float d) const
{
    if(d == 0.0f)
        return 0.0f;
    Point2d pix;
    getPixelFor3DPoint(&pix, x0, cam);
    pix.x = pix.x + d;
    const Point3d vect = (iCamArr[cam] * pix).normalize();
    return pointLineDistance3D(x0, CArr[cam], vect);
}

// MultiViewParams.cpp:505-517
void MultiViewParams::decomposeProjectionMatrix(Point3d& Co, Matrix3x3& Ro, Matrix3x3& iRo, Matrix3x3& Ko, Matrix3x3& iKo, Matrix3x3& iPo,
                                                const Matrix3x4& P) const
{
    P.decomposeProjectionMatrix(Ko, Ro, Co);
    iKo = Ko.inverse();
    iRo = Ro.inverse();
    iPo = iRo * iKo;
}

// MultiViewParams.cpp:519-575.  Scores = number of common landmarks whose two rays form an angle inside [min, max] view
// angle; cameras sorted by descending score with the C library's qsort and a comparator that, like the reference's
// (mvsData/structures.cpp:37-47), never reports equality — so equal scores come out in the order the reference's own call produces on
// the same C library (glibc: a merge sort that takes the element of the SECOND half on a tie; pinned by tests/test_host_ref.py) —
// at least 21 common landmarks required.
namespace {
int bySortedIdValueDescending(const void* pa, const void* pb)
{
    const SortedId& a = *static_cast<const SortedId*>(pa);
    const SortedId& b = *static_cast<const SortedId*>(pb);
    return a.value > b.value ? -1 : 1; // never 0, like the reference's comparator
}
} // namespace

std::vector<int> MultiViewParams::findNearestCamsFromLandmarks(int rc, int nbNearestCams) const
{
    std::vector<int> out;
    std::vector<SortedId> ids;
    ids.reserve(getNbCameras());
    for(int tc = 0; tc < getNbCameras(); ++tc)
        ids.push_back({tc, 0.f});

    const IndexT viewId = getViewId(rc);
    const View& view = _sfmData.views.at(viewId);
    const Pose pose = _sfmData.getPose(view);
    const Intrinsic& intr = _sfmData.getIntrinsic(view);

    for(const auto& landmarkPair : _sfmData.landmarks)
    {
        const auto& observations = landmarkPair.second.observations;
        const auto viewObsIt = observations.find(viewId);
        if(viewObsIt == observations.end())
            continue;
        for(const auto& observationPair : observations)
        {
            const IndexT otherViewId = observationPair.first;
            if(otherViewId == viewId)
                continue;
            const auto idIt = _imageIdsPerViewId.find(otherViewId);
            if(idIt == _imageIdsPerViewId.end())
                continue; // observation of a view without pose / intrinsic
            const View& otherView = _sfmData.views.at(otherViewId);
            const double angle = angleBetweenRays(pose, intr, _sfmData.getPose(otherView), _sfmData.getIntrinsic(otherView),
                                                  Point2d(viewObsIt->second.x, viewObsIt->second.y), Point2d(observationPair.second.x, observationPair.second.y));
            if(angle < _minViewAngle || angle > _maxViewAngle)
                continue;
            ++ids.at(idIt->second).value;
        }
    }
    if(!ids.empty())
        std::qsort(ids.data(), ids.size(), sizeof(SortedId), bySortedIdValueDescending);
    const int maxTc = std::min({getNbCameras(), nbNearestCams, static_cast<int>(ids.size())});
    out.reserve(maxTc);
    for(int i = 0; i < maxTc; ++i)
        if(ids[i].value > (10 * 2))
            out.push_back(ids[i].id);
    if((int)out.size() < nbNearestCams)
        AVDM_LOG_INFO("Found only " << out.size() << "/" << nbNearestCams << " nearest cameras for view id: " << getViewId(rc));
    return out;
}

// MultiViewParams.cpp:577-667
std::vector<int> MultiViewParams::findTileNearestCams(int rc, int nbNearestCams, const std::vector<int>& tCams, const ROI& roi) const
{
    auto plateauFunction = [](int a, int b, int c, int d, int x) {
        if(x > a && x <= b)
            return (float(x - a) / float(b - a));
        if(x > b && x <= c)
            return 1.0f;
        if(x > c && x <= d)
            return 1.0f - (float(x - c) / float(d - c));
        return 0.f;
    };
    std::vector<int> out;
    std::map<int, float> tcScore;
    for(const int tc : tCams)
        tcScore[tc] = 0.0f;

    const IndexT viewId = getViewId(rc);
    const View& view = _sfmData.views.at(viewId);
    const Pose pose = _sfmData.getPose(view);
    const Intrinsic& intr = _sfmData.getIntrinsic(view);
    const ROI fullsizeRoi = upscaleROI(roi, (float)getProcessDownscale());

    for(const auto& landmarkPair : _sfmData.landmarks)
    {
        const auto& observations = landmarkPair.second.observations;
        const auto viewObsIt = observations.find(viewId);
        if(viewObsIt == observations.end())
            continue;
        // ROI::contains takes unsigned ints: the double coordinates are converted (truncated) at the call (ROI.hpp:119)
        if(!fullsizeRoi.contains((unsigned int)viewObsIt->second.x, (unsigned int)viewObsIt->second.y))
            continue;
        for(const auto& observationPair : observations)
        {
            const IndexT otherViewId = observationPair.first;
            if(otherViewId == viewId)
                continue;
            const auto idIt = _imageIdsPerViewId.find(otherViewId);
            if(idIt == _imageIdsPerViewId.end())
                continue;
            const int tc = idIt->second;
            if(tcScore.find(tc) == tcScore.end())
                continue;
            const View& otherView = _sfmData.views.at(otherViewId);
            const double angle = angleBetweenRays(pose, intr, _sfmData.getPose(otherView), _sfmData.getIntrinsic(otherView),
                                                  Point2d(viewObsIt->second.x, viewObsIt->second.y), Point2d(observationPair.second.x, observationPair.second.y));
            tcScore[tc] += plateauFunction(1, 10, 50, 150, (int)angle); // the lambda takes an int: the angle is truncated
        }
    }
    std::vector<SortedId> ids;
    for(const auto& p : tcScore)
        if(p.second > 0.0f)
            ids.push_back({p.first, p.second});
    if(!ids.empty())
        std::qsort(ids.data(), ids.size(), sizeof(SortedId), bySortedIdValueDescending);
    const int maxTc = std::min(std::min(getNbCameras(), nbNearestCams), static_cast<int>(ids.size()));
    for(int i = 0; i < maxTc; ++i)
        out.push_back(ids[i].id);
    return out;
}

// mvsUtils/fileIO.cpp:18-358 (only the file types this stage produces or reads)
std::string getFileNameFromViewId(const MultiViewParams& mp, IndexT viewId, EFileType fileType, const std::string& customSuffix, int tileBeginX, int tileBeginY)
{
    std::string folder = mp.getImagesFolder(), suffix, tileSuffix, ext = "exr";
    if(tileBeginX >= 0 && tileBeginY >= 0)
        tileSuffix = "_" + std::to_string(tileBeginX) + "_" + std::to_string(tileBeginY);
    switch(fileType)
    {
        case EFileType::P: suffix = "_P", ext = "txt"; break;
        case EFileType::D: suffix = "_D", ext = "txt"; break;
        case EFileType::depthMap: folder = mp.getDepthMapsFolder(), suffix = "_depthMap"; break;
        case EFileType::simMap: folder = mp.getDepthMapsFolder(), suffix = "_simMap"; break;
        case EFileType::normalMap: folder = mp.getDepthMapsFolder(), suffix = "_normalMap"; break;
        case EFileType::thicknessMap: folder = mp.getDepthMapsFolder(), suffix = "_thicknessMap"; break;
        case EFileType::pixSizeMap: folder = mp.getDepthMapsFolder(), suffix = "_pixSizeMap"; break;
        case EFileType::tilePattern: folder = mp.getDepthMapsFolder(), suffix = "_tilePattern", ext = "obj"; break;
        case EFileType::depthMapFiltered: folder = mp.getDepthMapsFilterFolder(), suffix = "_depthMap"; break;
        case EFileType::simMapFiltered: folder = mp.getDepthMapsFilterFolder(), suffix = "_simMap"; break;
        case EFileType::normalMapFiltered: folder = mp.getDepthMapsFilterFolder(), suffix = "_normalMap"; break;
        case EFileType::nmodMap: folder = mp.getDepthMapsFilterFolder(), suffix = "_nmodMap", ext = "png"; break;
        case EFileType::volume: folder = mp.getDepthMapsFolder(), suffix = "_volume", ext = "abc"; break;
        case EFileType::volumeCross: folder = mp.getDepthMapsFolder(), suffix = "_volumeCross", ext = "abc"; break;
        case EFileType::volumeTopographicCut: folder = mp.getDepthMapsFolder(), suffix = "_volumeTopographicCut", ext = "abc"; break;
        case EFileType::stats9p: folder = mp.getDepthMapsFolder(), suffix = "_9p", ext = "csv"; break;
        case EFileType::none: break;
    }
    return folder + std::to_string(viewId) + suffix + customSuffix + tileSuffix + "." + ext;
}

// mvsUtils/common.cpp:23-116
bool get2dLineImageIntersection(Point2d* pFrom, Point2d* pTo, Point2d linePoint1, Point2d linePoint2, const MultiViewParams& mp, int camId)
{
    Point2d v = linePoint2 - linePoint1;
    if(v.size() < FLT_EPSILON)
        return false;
    v = v.normalize();
    const double a = -v.y, b = v.x, c = -a * linePoint1.x - b * linePoint1.y;
    int intersections = 0;
    const double rw = (double)mp.getWidth(camId), rh = (double)mp.getHeight(camId);
    auto add = [&](double x, double y) {
        if(intersections == 0)
            *pFrom = Point2d(x, y);
        else
            *pTo = Point2d(x, y);
        intersections++;
    };
    double x = 0, y = -c / b;
    if((y >= 0) && (y < rh))
        add(x, y);
    x = rw;
    y = (-c - a * rw) / b;
    if((y >= 0) && (y < rh))
        add(x, y);
    x = -c / a;
    y = 0;
    if((x >= 0) && (x < rw))
        add(x, y);
    x = (-c - b * rh) / a;
    y = rh;
    if((x >= 0) && (x < rw))
        add(x, y);
    if(intersections == 2)
    {
        if((linePoint1 - *pFrom).size() > (linePoint1 - *pTo).size())
            std::swap(*pFrom, *pTo);
        return true;
    }
    return false;
}

// mvsUtils/common.cpp:155-169
bool triangulateMatch(Point3d& out, const Point2d& refpix, const Point2d& tarpix, int refCam, int tarCam, const MultiViewParams& mp)
{
    const Point3d refvect = (mp.iCamArr[refCam] * refpix).normalize();
    const Point3d refpoint = refvect + mp.CArr[refCam];
    const Point3d tarvect = (mp.iCamArr[tarCam] * tarpix).normalize();
    const Point3d tarpoint = tarvect + mp.CArr[tarCam];
    return lineLineIntersect(out, mp.CArr[refCam], refpoint, mp.CArr[tarCam], tarpoint);
}

// mvsUtils/fileIO.cpp:389-443.  OpenEXR is decoded on the host; PNG and JPEG leave it as integer samples / DCT coefficients and become
// linear float RGBA on the device; the --downscale resize happens on the device (see below).
std::shared_ptr<const HostImage> ImagesCache::getImg_sync(int camId)
{
    {
        std::lock_guard<std::mutex> lock(_mutex);
        auto it = _cache.find(camId);
        if(it != _cache.end())
        {
            it->second.first = ++_tick;
            return it->second.second;
        }
    }
    // decode outside the lock: several images can be read at once (DepthMapEstimator pre-warms the cache of a batch in parallel)
    const std::string& path = _mp.getImagePath(camId);
    if(isPngPath(path) || isTiffPath(path))
    {
        // the container's own decoding on the host (inflate + scan-line filters / strips, LZW, predictor: sequential by nature); the
        // integer samples go to the device as they are and become linear float RGBA there (avdm_image_decode_integer:
        // image::readImage(..., LINEAR) for an 8- / 16-bit sRGB file)
        int w = 0, h = 0, channels = 0, bits = 0;
        std::vector<unsigned char> samples;
        if(isPngPath(path))
        {
            PngImage png;
            readPng(path, png);
            w = png.width, h = png.height, channels = png.channels, bits = png.bits;
            samples.swap(png.samples);
        }
        else
        {
            TiffImage tiff;
            readTiff(path, tiff);
            w = tiff.width, h = tiff.height, channels = tiff.channels, bits = tiff.bits;
            samples.swap(tiff.samples);
        }
        if(_mp.getOriginalWidth(camId) != w || _mp.getOriginalHeight(camId) != h)
            throw std::runtime_error("Bad image dimension for camera : " + std::to_string(camId) + "\n\t- image path : " + path + "\n\t- expected dimension : " +
                                     std::to_string(_mp.getOriginalWidth(camId)) + "x" + std::to_string(_mp.getOriginalHeight(camId)) +
                                     "\n\t- real dimension : " + std::to_string(w) + "x" + std::to_string(h));
        auto full = std::make_shared<HostImage>();
        full->raw.swap(samples);
        full->rawChannels = channels;
        full->rawBits = bits;
        const int s = _mp.getProcessDownscale();
        full->srcWidth = w;
        full->srcHeight = h;
        full->width = s > 1 ? w / s : w;
        full->height = s > 1 ? h / s : h;
        std::shared_ptr<const HostImage> result = full;
        std::lock_guard<std::mutex> lock(_mutex);
        if(_cache.size() >= _max)
        {
            auto oldest = _cache.begin();
            for(auto i = _cache.begin(); i != _cache.end(); ++i)
                if(i->second.first < oldest->second.first)
                    oldest = i;
            _cache.erase(oldest);
        }
        _cache[camId] = {++_tick, result};
        return result;
    }
    if(isJpegPath(path))
    {
        // markers and Huffman decoding on the host (sequential by nature); the coefficients go to the device, which does the rest of the decode
        auto jpeg = std::make_shared<JpegImage>();
        readJpeg(path, *jpeg);
        if(_mp.getOriginalWidth(camId) != jpeg->width || _mp.getOriginalHeight(camId) != jpeg->height)
            throw std::runtime_error("Bad image dimension for camera : " + std::to_string(camId) + "\n\t- image path : " + path + "\n\t- expected dimension : " +
                                     std::to_string(_mp.getOriginalWidth(camId)) + "x" + std::to_string(_mp.getOriginalHeight(camId)) +
                                     "\n\t- real dimension : " + std::to_string(jpeg->width) + "x" + std::to_string(jpeg->height));
        auto full = std::make_shared<HostImage>();
        const int s = _mp.getProcessDownscale();
        full->srcWidth = jpeg->width;
        full->srcHeight = jpeg->height;
        full->width = s > 1 ? jpeg->width / s : jpeg->width;
        full->height = s > 1 ? jpeg->height / s : jpeg->height;
        full->jpeg = jpeg;
        std::shared_ptr<const HostImage> result = full;
        std::lock_guard<std::mutex> lock(_mutex);
        if(_cache.size() >= _max)
        {
            auto oldest = _cache.begin();
            for(auto i = _cache.begin(); i != _cache.end(); ++i)
                if(i->second.first < oldest->second.first)
                    oldest = i;
            _cache.erase(oldest);
        }
        _cache[camId] = {++_tick, result};
        return result;
    }
    {
        // the scan lines as stored: no host pass over the samples at all (an uncompressed file is mapped and uploaded as it lies), the
        // de-interleave to float RGBA runs on the device.  Rounds 1-5 decoded to four channel planes and interleaved them here: three passes
        // over 192 MB per 12 MP view and ~150 000 page faults, 0.55 s of the 11-view job even with one host thread per view.
        const char* e = getenv("AVDM_HOST_EXR");
        auto lines = std::make_shared<ExrLines>();
        if(!(e != nullptr && std::string(e) == "host") && readExrLines(path, *lines))
        {
            if(_mp.getOriginalWidth(camId) != lines->width || _mp.getOriginalHeight(camId) != lines->height)
                throw std::runtime_error("Bad image dimension for camera : " + std::to_string(camId) + "\n\t- image path : " + path + "\n\t- expected dimension : " +
                                         std::to_string(_mp.getOriginalWidth(camId)) + "x" + std::to_string(_mp.getOriginalHeight(camId)) +
                                         "\n\t- real dimension : " + std::to_string(lines->width) + "x" + std::to_string(lines->height));
            auto full = std::make_shared<HostImage>();
            const int s = _mp.getProcessDownscale();
            full->srcWidth = lines->width;
            full->srcHeight = lines->height;
            full->width = s > 1 ? lines->width / s : lines->width;
            full->height = s > 1 ? lines->height / s : lines->height;
            full->exrLines = lines;
            // a MAPPED file is not kept in this cache: mapping it again costs nothing (the pages stay in the page cache), and un-mapping 11 x
            // 192 MB at the end of a job was 0.1 s of its tear-down — the mapping goes when its upload is done, on the thread that did it
            if(lines->mapBase != nullptr)
                return full;
            std::shared_ptr<const HostImage> result = full;
            std::lock_guard<std::mutex> lock(_mutex);
            if(_cache.size() >= _max)
            {
                auto oldest = _cache.begin();
                for(auto i = _cache.begin(); i != _cache.end(); ++i)
                    if(i->second.first < oldest->second.first)
                        oldest = i;
                _cache.erase(oldest);
            }
            _cache[camId] = {++_tick, result};
            return result;
        }
    }
    ExrImage exr;
    readExr(path, exr);
    if(_mp.getOriginalWidth(camId) != exr.width || _mp.getOriginalHeight(camId) != exr.height)
        throw std::runtime_error("Bad image dimension for camera : " + std::to_string(camId) + "\n\t- image path : " + path + "\n\t- expected dimension : " +
                                 std::to_string(_mp.getOriginalWidth(camId)) + "x" + std::to_string(_mp.getOriginalHeight(camId)) +
                                 "\n\t- real dimension : " + std::to_string(exr.width) + "x" + std::to_string(exr.height));
    const int iR = exr.channelIndex("R"), iG = exr.channelIndex("G"), iB = exr.channelIndex("B"), iA = exr.channelIndex("A"), iY = exr.channelIndex("Y");
    if(!((iR >= 0 && iG >= 0 && iB >= 0) || iY >= 0))
        throw std::runtime_error("image '" + path + "' has neither R,G,B nor Y channels");
    const size_t n = (size_t)exr.width * exr.height;
    auto full = std::make_shared<HostImage>();
    full->rgba.resize(n * 4);
    const float* r = exr.channels[iR >= 0 ? iR : iY].data();
    const float* g = exr.channels[iG >= 0 ? iG : iY].data();
    const float* b = exr.channels[iB >= 0 ? iB : iY].data();
    const float* a = iA >= 0 ? exr.channels[iA].data() : nullptr;
#pragma omp parallel for
    for(long long i = 0; i < (long long)n; ++i)
    {
        full->rgba[4 * i + 0] = r[i];
        full->rgba[4 * i + 1] = g[i];
        full->rgba[4 * i + 2] = b[i];
        full->rgba[4 * i + 3] = a ? a[i] : 1.0f;
    }
    // --downscale: the image stays at its decoded size here; DeviceMipmapImage::fill resizes it on the device with OpenImageIO's default
    // filter restated (avdm_image_resize <-> imageAlgo::resizeImage, fileIO.cpp:432-441) instead of a host loop over 12-24 M pixels
    const int s = _mp.getProcessDownscale();
    full->srcWidth = exr.width;
    full->srcHeight = exr.height;
    full->width = s > 1 ? exr.width / s : exr.width;
    full->height = s > 1 ? exr.height / s : exr.height;
    if(s > 1)
        AVDM_LOG_DEBUG("Downscale (x" << s << ") image: " << _mp.getViewId(camId) << ".");
    std::shared_ptr<const HostImage> result = full;
    std::lock_guard<std::mutex> lock(_mutex);
    if(_cache.size() >= _max)
    {
        auto oldest = _cache.begin();
        for(auto i = _cache.begin(); i != _cache.end(); ++i)
            if(i->second.first < oldest->second.first)
                oldest = i;
        _cache.erase(oldest);
    }
    _cache[camId] = {++_tick, result};
    return result;
}

} // namespace avdm_host
