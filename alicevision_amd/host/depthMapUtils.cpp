// depthMapUtils.cpp — see depthMapUtils.hpp.
#include "depthMapUtils.hpp"

#include "alembic.hpp"
#include "sfmData.hpp"

#include "device.hpp"
#include "exr.hpp"
#include "log.hpp"

#include <sys/mman.h>

#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <fstream>
#include <future>
#include <sstream>
#include <cstdio>
#include <fstream>
#include <limits>
#include <regex>

namespace avdm_host {

void* hugePageAlloc(size_t bytes)
{
    const size_t huge = (size_t)2 << 20;
    void* p = nullptr;
    if(bytes >= huge)
    {
        const size_t rounded = (bytes + huge - 1) & ~(huge - 1);
        if(posix_memalign(&p, huge, rounded) != 0)
            throw std::bad_alloc();
        (void)madvise(p, rounded, MADV_HUGEPAGE); // (advice: a kernel without transparent huge pages serves 4 KiB pages)
        return p;
    }
    p = std::malloc(bytes ? bytes : 1);
    if(p == nullptr)
        throw std::bad_alloc();
    return p;
}
void hugePageFree(void* p) { std::free(p); }

void resetDepthSimMap(Float2Tile& inout, float depth, float sim)
{
    for(size_t i = 0; i < (size_t)inout.width * inout.height; ++i)
    {
        inout.data[2 * i] = depth;
        inout.data[2 * i + 1] = sim;
    }
}

namespace {

inline double clampd(double v, double lo, double hi) { return std::min(std::max(v, lo), hi); }

// mapIO.cpp:170-202
void weightTileBorder(int a, int b, int c, int d, int borderWidth, int borderHeight, const Point2d& lu, FloatMap& in_tileMap)
{
    const Point2d rd = lu + Point2d(borderWidth, borderHeight);
    const int endX = std::min(int(rd.x), in_tileMap.width);
    const int endY = std::min(int(rd.y), in_tileMap.height);
    static const double margin = 2.0;
    const Point2d lu_m(lu.x + margin, lu.y + margin);
    const Point2d rd_m(rd.x - margin, rd.y - margin);
    const double borderWidth_m = borderWidth - 2.0 * margin;
    const double borderHeight_m = borderHeight - 2.0 * margin;
    // row-major maps: y outer, x inner (the reference iterates column by column; the weight of a pixel does not depend on the order)
    for(int y = (int)lu.y; y < endY; ++y)
        for(int x = (int)lu.x; x < endX; ++x)
        {
            if(x < 0 || y < 0)
                continue; // a tile narrower than the padding: the reference would index out of bounds
            const float r_x = (float)clampd((rd_m.x - x) / borderWidth_m, 0.0, 1.0);
            const float r_y = (float)clampd((rd_m.y - y) / borderHeight_m, 0.0, 1.0);
            const float l_x = (float)clampd((x - lu_m.x) / borderWidth_m, 0.0, 1.0);
            const float l_y = (float)clampd((y - lu_m.y) / borderHeight_m, 0.0, 1.0);
            const float weight = r_y * (r_x * a + l_x * b) + l_y * (r_x * d + l_x * c);
            in_tileMap(y, x) *= weight;
        }
}

bool fileExists(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// mapIO.cpp:136-155
void getTilePathList(int rc, const MultiViewParams& mp, EFileType fileType, const std::string& customSuffix, std::vector<std::string>& out)
{
    const std::string mapPath = getFileNameFromIndex(mp, rc, fileType, customSuffix);
    const size_t slash = mapPath.rfind('/');
    const std::string dir = slash == std::string::npos ? "." : mapPath.substr(0, slash);
    const std::string file = slash == std::string::npos ? mapPath : mapPath.substr(slash + 1);
    const size_t dot = file.rfind('.');
    const std::string stem = file.substr(0, dot), ext = file.substr(dot);
    const std::regex pattern(stem + "_\\d+_\\d+\\" + ext);
    DIR* d = ::opendir(dir.c_str());
    if(!d)
        AVDM_THROW_ERROR("Cannot find map directory (rc: " << rc << ").");
    while(dirent* e = ::readdir(d))
        if(std::regex_match(std::string(e->d_name), pattern))
            out.push_back(dir + "/" + e->d_name);
    ::closedir(d);
    std::sort(out.begin(), out.end());
}

// device float2 map (tile ROI) -> two host float maps (depthMapUtils.cpp:22-60)
void copyFloat2MapFromDevice(FloatMap& outX, FloatMap& outY, const float* map_d, int pitch, const ROI& roi, int downscale, hipStream_t stream)
{
    const ROI r = downscaleROI(roi, (float)downscale);
    const int w = (int)r.width(), h = (int)r.height();
    std::vector<float> tmp((size_t)w * h * 2);
    AVDM_HIP_CHECK(hipMemcpy2DAsync(tmp.data(), (size_t)w * 8, map_d, (size_t)pitch, (size_t)w * 8, (size_t)h, hipMemcpyDeviceToHost, stream));
    AVDM_HIP_CHECK(hipStreamSynchronize(stream));
    outX = FloatMap(w, h);
    outY = FloatMap(w, h);
    for(size_t i = 0; i < (size_t)w * h; ++i)
    {
        outX.data[i] = tmp[2 * i];
        outY.data[i] = tmp[2 * i + 1];
    }
}

void writeFloat2MapFromDevice(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, const float* map_d, int pitch, EFileType typeX,
                              EFileType typeY, int scale, int step, const std::string& name, hipStream_t stream)
{
    const std::string customSuffix = name.empty() ? "" : "_" + name;
    FloatMap mapX, mapY;
    copyFloat2MapFromDevice(mapX, mapY, map_d, pitch, roi, scale * step, stream);
    writeMap(rc, mp, typeX, tileParams, roi, mapX, scale, step, customSuffix);
    writeMap(rc, mp, typeY, tileParams, roi, mapY, scale, step, customSuffix);
}

// mapIO.cpp:402-540 common part: path, display / pixel windows and metadata
struct MapFileInfo
{
    std::string path;
    int imageWidth, imageHeight;
    ROI downscaledROI;
    ExrAttributes metadata;
};

MapFileInfo prepareMapFile(int rc, const MultiViewParams& mp, EFileType fileType, const TileParams& tileParams, const ROI& roi, int scale, int step,
                           const std::string& customSuffix)
{
    MapFileInfo f;
    const int scaleStep = scale * step;
    f.imageWidth = divideRoundUp(mp.getWidth(rc), scaleStep);
    f.imageHeight = divideRoundUp(mp.getHeight(rc), scaleStep);
    f.downscaledROI = downscaleROI(roi, (float)scaleStep);
    if((int)f.downscaledROI.width() != f.imageWidth || (int)f.downscaledROI.height() != f.imageHeight)
        f.path = getFileNameFromIndex(mp, rc, fileType, customSuffix, (int)roi.x.begin, (int)roi.y.begin); // tile
    else
        f.path = getFileNameFromIndex(mp, rc, fileType, customSuffix);

    // original picture metadata (strings), then the AliceVision entries
    for(const auto& kv : mp.getMetadata(rc))
        f.metadata.setString(kv.first, kv.second);
    f.metadata.setInt("AliceVision:downscale", mp.getDownscaleFactor(rc) * scaleStep);
    f.metadata.setInt("AliceVision:roiBeginX", int(roi.x.begin));
    f.metadata.setInt("AliceVision:roiBeginY", int(roi.y.begin));
    f.metadata.setInt("AliceVision:roiEndX", int(roi.x.end));
    f.metadata.setInt("AliceVision:roiEndY", int(roi.y.end));
    f.metadata.setInt("AliceVision:tileBufferWidth", tileParams.bufferWidth);
    f.metadata.setInt("AliceVision:tileBufferHeight", tileParams.bufferHeight);
    f.metadata.setInt("AliceVision:tilePadding", tileParams.padding);
    {
        const std::vector<double> matrixP = mp.getOriginalP(rc);
        f.metadata.setM44d("AliceVision:P", matrixP.data());
    }
    {
        Point3d C = mp.CArr[rc];
        Matrix3x3 iP = mp.iCamArr[rc];
        if(scaleStep > 1)
        {
            Matrix3x4 P = mp.camArr[rc];
            for(int i = 0; i < 8; ++i)
                P.m[i] /= double(scaleStep);
            Matrix3x3 K, R;
            P.decomposeProjectionMatrix(K, R, C);
            iP = R.inverse() * K.inverse();
        }
        const double c[3] = {C.x, C.y, C.z};
        f.metadata.setV3d("AliceVision:CArr", c);
        f.metadata.setM33d("AliceVision:iCamArr", iP.m);
    }
    return f;
}

} // namespace

void addTileMapWeighted(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, int downscale, FloatMap& in_tileMap, FloatMap& inout_map)
{
    const ROI downscaledRoi = downscaleROI(roi, (float)downscale);
    const int tileWidth = (int)downscaledRoi.width();
    const int tileHeight = (int)downscaledRoi.height();
    const int tilePadding = tileParams.padding / downscale;

    const bool firstColumn = (roi.x.begin == 0);
    const bool lastColumn = ((int)roi.x.end == mp.getWidth(rc));
    const bool firstRow = (roi.y.begin == 0);
    const bool lastRow = ((int)roi.y.end == mp.getHeight(rc));

    if(!firstColumn || !firstRow) // top left corner
        weightTileBorder(0, firstRow ? 1 : 0, 1, firstColumn ? 1 : 0, tilePadding, tilePadding, Point2d(0, 0), in_tileMap);
    if(!firstColumn || !lastRow) // bottom left corner
        weightTileBorder(firstColumn ? 1 : 0, 1, lastRow ? 1 : 0, 0, tilePadding, tilePadding, Point2d(0, tileHeight - tilePadding), in_tileMap);
    if(!lastColumn || !firstRow) // top right corner
        weightTileBorder(firstRow ? 1 : 0, 0, lastColumn ? 1 : 0, 1, tilePadding, tilePadding, Point2d(tileWidth - tilePadding, 0), in_tileMap);
    if(!lastColumn || !lastRow) // bottom right corner
        weightTileBorder(1, lastColumn ? 1 : 0, 0, lastRow ? 1 : 0, tilePadding, tilePadding, Point2d(tileWidth - tilePadding, tileHeight - tilePadding), in_tileMap);
    if(!firstRow) // top border
        weightTileBorder(0, 0, 1, 1, tileWidth - 2 * tilePadding, tilePadding, Point2d(tilePadding, 0), in_tileMap);
    if(!lastRow) // bottom border
        weightTileBorder(1, 1, 0, 0, tileWidth - 2 * tilePadding, tilePadding, Point2d(tilePadding, tileHeight - tilePadding), in_tileMap);
    if(!firstColumn) // left border
        weightTileBorder(0, 1, 1, 0, tilePadding, tileHeight - 2 * tilePadding, Point2d(0, tilePadding), in_tileMap);
    if(!lastColumn) // right border
        weightTileBorder(1, 0, 0, 1, tilePadding, tileHeight - 2 * tilePadding, Point2d(tileWidth - tilePadding, tilePadding), in_tileMap);

    const int xEnd = std::min((int)downscaledRoi.x.end, inout_map.width), yEnd = std::min((int)downscaledRoi.y.end, inout_map.height);
#pragma omp parallel for schedule(static) num_threads(8)
    for(int y = (int)downscaledRoi.y.begin; y < yEnd; ++y)
        for(int x = (int)downscaledRoi.x.begin; x < xEnd; ++x)
            inout_map(y, x) += in_tileMap(y - (int)downscaledRoi.y.begin, x - (int)downscaledRoi.x.begin);
}

void writeMap(int rc, const MultiViewParams& mp, EFileType fileType, const TileParams& tileParams, const ROI& roi, const FloatMap& in_map, int scale, int step,
              const std::string& customSuffix)
{
    MapFileInfo f = prepareMapFile(rc, mp, fileType, tileParams, roi, scale, step, customSuffix);
    if(in_map.width != (int)f.downscaledROI.width() || in_map.height != (int)f.downscaledROI.height())
        AVDM_THROW_ERROR("writeMap: map size " << in_map.width << "x" << in_map.height << " does not match the ROI " << f.downscaledROI);
    if(fileType == EFileType::depthMap || fileType == EFileType::depthMapFiltered)
    {
        // mapIO.cpp:493-511
        // (one pass, a team of 8: a count and two order-independent extrema — the same values as the reference's three sequential passes)
        long long nbDepthValuesL = 0;
        float maxDepth = -1.0f;
        float minDepth = std::numeric_limits<float>::max();
        const float* const dv = in_map.data.data();
        const long long nv = (long long)in_map.data.size();
#pragma omp parallel for schedule(static) num_threads(8) reduction(+ : nbDepthValuesL) reduction(max : maxDepth) reduction(min : minDepth)
        for(long long i = 0; i < nv; ++i)
        {
            const float depth = dv[i];
            if(depth > 0.0f)
                ++nbDepthValuesL;
            if(depth <= -1.0f)
                continue;
            maxDepth = std::max(maxDepth, depth);
            minDepth = std::min(minDepth, depth);
        }
        const int nbDepthValues = (int)nbDepthValuesL;
        f.metadata.setInt("AliceVision:nbDepthValues", nbDepthValues);
        f.metadata.setFloat("AliceVision:minDepth", minDepth);
        f.metadata.setFloat("AliceVision:maxDepth", maxDepth);
    }
    // depth maps are stored as float, every other map as half (mapIO.cpp:517-526); one channel is named "Y" by OpenImageIO
    const bool storeHalf = (fileType != EFileType::depthMap && fileType != EFileType::depthMapFiltered);
    writeExr(f.path, in_map.width, in_map.height, {{"Y", in_map.data.data()}}, storeHalf, f.metadata, (int)f.downscaledROI.x.begin, (int)f.downscaledROI.y.begin,
             f.imageWidth, f.imageHeight);
}

void writeMap3(int rc, const MultiViewParams& mp, EFileType fileType, const TileParams& tileParams, const ROI& roi, const std::vector<float>& rgb, int width,
               int height, int scale, int step, const std::string& customSuffix)
{
    MapFileInfo f = prepareMapFile(rc, mp, fileType, tileParams, roi, scale, step, customSuffix);
    std::vector<float> r((size_t)width * height), g(r.size()), b(r.size());
    for(size_t i = 0; i < r.size(); ++i)
    {
        r[i] = rgb[3 * i];
        g[i] = rgb[3 * i + 1];
        b[i] = rgb[3 * i + 2];
    }
    writeExr(f.path, width, height, {{"R", r.data()}, {"G", g.data()}, {"B", b.data()}}, true, f.metadata, (int)f.downscaledROI.x.begin,
             (int)f.downscaledROI.y.begin, f.imageWidth, f.imageHeight);
}

void readMap(int rc, const MultiViewParams& mp, EFileType fileType, FloatMap& out_map, int scale, int step, const std::string& customSuffix)
{
    const std::string mapPath = getFileNameFromIndex(mp, rc, fileType, customSuffix);
    if(fileExists(mapPath))
    {
        ExrImage img;
        readExr(mapPath, img);
        out_map = FloatMap(img.width, img.height);
        out_map.data = img.channels.at(0);
        return;
    }
    const ROI imageRoi(Range(0, mp.getWidth(rc)), Range(0, mp.getHeight(rc)));
    const int scaleStep = scale * step;
    out_map = FloatMap(divideRoundUp(mp.getWidth(rc), scaleStep), divideRoundUp(mp.getHeight(rc), scaleStep), 0.f);

    std::vector<std::string> mapTilePathList;
    getTilePathList(rc, mp, fileType, customSuffix, mapTilePathList);
    if(mapTilePathList.empty())
    {
        AVDM_LOG_INFO("Cannot find any map tile file (rc: " << rc << ").");
        return;
    }
    // tile parameters from the first tile, ROI from each tile (mapIO.cpp:64-103, 360-372)
    TileParams tileParams;
    std::vector<ROI> tileRoiList(mapTilePathList.size());
    for(size_t i = 0; i < mapTilePathList.size(); ++i)
    {
        ExrImage header;
        readExr(mapTilePathList[i], header, true);
        int bx = -1, by = -1, ex = -1, ey = -1;
        header.attributes.getInt("AliceVision:roiBeginX", bx);
        header.attributes.getInt("AliceVision:roiBeginY", by);
        header.attributes.getInt("AliceVision:roiEndX", ex);
        header.attributes.getInt("AliceVision:roiEndY", ey);
        if(bx < 0 || by < 0 || ex <= 0 || ey <= 0)
            AVDM_THROW_ERROR("Cannot find ROI information in file: " << mapTilePathList[i]);
        tileRoiList[i] = ROI(bx, ex, by, ey);
        if(i == 0)
        {
            header.attributes.getInt("AliceVision:tileBufferWidth", tileParams.bufferWidth);
            header.attributes.getInt("AliceVision:tileBufferHeight", tileParams.bufferHeight);
            header.attributes.getInt("AliceVision:tilePadding", tileParams.padding);
        }
    }
    for(size_t i = 0; i < tileRoiList.size(); ++i)
    {
        const ROI roi = intersect(tileRoiList.at(i), imageRoi);
        if(roi.isEmpty())
            continue;
        const std::string mapTilePath = getFileNameFromIndex(mp, rc, fileType, customSuffix, (int)roi.x.begin, (int)roi.y.begin);
        try
        {
            ExrImage img;
            readExr(mapTilePath, img);
            FloatMap tileMap(img.width, img.height);
            tileMap.data = img.channels.at(0);
            addTileMapWeighted(rc, mp, tileParams, roi, scaleStep, tileMap, out_map);
        }
        catch(const std::exception&)
        {
            AVDM_LOG_WARNING("Cannot find map (rc: " << rc << "): " << mapTilePath);
        }
    }
}

void deleteMapTiles(int rc, const MultiViewParams& mp, EFileType fileType, const std::string& customSuffix)
{
    std::vector<std::string> mapTilePathList;
    getTilePathList(rc, mp, fileType, customSuffix, mapTilePathList);
    if(mapTilePathList.empty())
        AVDM_LOG_INFO("Cannot find any map tile file to delete (rc: " << rc << ").");
    for(const std::string& p : mapTilePathList)
        if(std::remove(p.c_str()) != 0)
            AVDM_LOG_WARNING("Cannot delete map tile file (rc: " << rc << "): " << p);
}

void writeDepthSimMap(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, const float* map_d, int pitch, int scale, int step,
                      const std::string& name, hipStream_t stream)
{
    writeFloat2MapFromDevice(rc, mp, tileParams, roi, map_d, pitch, EFileType::depthMap, EFileType::simMap, scale, step, name, stream);
}
void writeDepthPixSizeMap(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, const float* map_d, int pitch, int scale, int step,
                          const std::string& name, hipStream_t stream)
{
    writeFloat2MapFromDevice(rc, mp, tileParams, roi, map_d, pitch, EFileType::depthMap, EFileType::pixSizeMap, scale, step, name, stream);
}
void writeNormalMap(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, const float* map_d, int pitch, int scale, int step,
                    const std::string& name, hipStream_t stream)
{
    const ROI r = downscaleROI(roi, float(scale * step));
    const int w = (int)r.width(), h = (int)r.height();
    std::vector<float> rgb((size_t)w * h * 3);
    AVDM_HIP_CHECK(hipMemcpy2DAsync(rgb.data(), (size_t)w * 12, map_d, (size_t)pitch, (size_t)w * 12, (size_t)h, hipMemcpyDeviceToHost, stream));
    AVDM_HIP_CHECK(hipStreamSynchronize(stream));
    writeMap3(rc, mp, EFileType::normalMap, tileParams, roi, rgb, w, h, scale, step, name.empty() ? "" : "_" + name);
}

void exportSimilaritySamplesCSV(const void* volume_d, bool halfFloat, long long pitchY, int pitchX, int nbPlanes, int width, int height, const std::string& name,
                                const std::string& filepath, hipStream_t stream)
{
    const int sampleSize = 3;
    const int xOffset = (int)std::floor(width / (sampleSize + 1.0f));
    const int yOffset = (int)std::floor(height / (sampleSize + 1.0f));
    const size_t elem = halfFloat ? 2 : 1;
    std::vector<unsigned char> column((size_t)nbPlanes * elem);
    std::stringstream ss;
    ss << name << "\n";
    for(int iy = 0; iy < sampleSize; ++iy)
        for(int ix = 0; ix < sampleSize; ++ix)
        {
            const int x = (ix + 1) * xOffset, y = (iy + 1) * yOffset;
            AVDM_HIP_CHECK(hipMemcpyAsync(column.data(), (const char*)volume_d + (long long)y * pitchY + (long long)x * pitchX, column.size(), hipMemcpyDeviceToHost,
                                          stream));
            AVDM_HIP_CHECK(hipStreamSynchronize(stream));
            ss << "p" << (iy * sampleSize + ix + 1) << " (x: " << x << ", y: " << y << ");";
            for(int iz = 0; iz < nbPlanes; ++iz)
            {
                const float simValue = halfFloat ? halfToFloat(reinterpret_cast<const uint16_t*>(column.data())[iz]) : (float)column[iz];
                ss << simValue << ";";
            }
            ss << "\n";
        }
    std::ofstream file(filepath, std::ios_base::app);
    if(file.is_open())
        file << ss.str();
}

// ---- similarity volumes as point clouds ---------------------------------------------------------------------------------------------
float HostVolume::at(int x, int y, int z) const
{
    const unsigned char* p = bytes.data() + (long long)y * pitchY + (long long)x * pitchX;
    return halfFloat ? halfToFloat(reinterpret_cast<const uint16_t*>(p)[z]) : (float)p[z];
}

HostVolume downloadVolume(const void* volume_d, bool halfFloat, long long pitchY, int pitchX, int X, int Y, int Z, hipStream_t stream)
{
    HostVolume v;
    v.pitchY = pitchY, v.pitchX = pitchX, v.X = X, v.Y = Y, v.Z = Z, v.halfFloat = halfFloat;
    v.bytes.resize((size_t)pitchY * Y);
    AVDM_HIP_CHECK(hipMemcpyAsync(v.bytes.data(), volume_d, v.bytes.size(), hipMemcpyDeviceToHost, stream));
    AVDM_HIP_CHECK(hipStreamSynchronize(stream));
    return v;
}

void jetColor(float value, unsigned char rgb[3])
{
    // jet(64): with u = { 1/16 .. 16/16, fifteen ones, 16/16 .. 1/16 } (47 values), green = u placed at entries 8 .. 54, red the same
    // ramp 16 entries later (cut at entry 63), blue 16 entries earlier (starting inside the ramp) — the table of image/jetColorMap.cpp
    auto u = [](int k) -> float { return k < 0 || k > 46 ? 0.0f : (k < 16 ? (k + 1) / 16.0f : (k < 31 ? 1.0f : (47 - k) / 16.0f)); };
    auto jet = [&](int i, int channel) -> float { return u(i - (channel == 0 ? 24 : (channel == 1 ? 8 : -8))); };
    float c[3];
    if(value <= 0.0f)
        c[0] = c[1] = c[2] = 0.0f;
    else if(value >= 1.0f)
        c[0] = c[1] = c[2] = 1.0f;
    else
    {
        const float idx_f = value * 63.0f;
        float integral;
        const float fractB = std::modf(idx_f, &integral);
        const float fractA = 1.0f - fractB;
        const int idx = (int)integral;
        for(int k = 0; k < 3; ++k)
            c[k] = jet(idx, k) * fractA + jet(idx + 1, k) * fractB;
    }
    for(int k = 0; k < 3; ++k)
        rgb[k] = (unsigned char)(c[k] * 255.0f);
}

namespace {

// mvsData/geometry.cpp:156-177
Point3d linePlaneIntersect(const Point3d& linePoint, const Point3d& lineVect, const Point3d& planePoint, const Point3d& planeNormal)
{
    const double k = (dot(planePoint, planeNormal) - dot(planeNormal, linePoint)) / dot(planeNormal, lineVect);
    return linePoint + lineVect * k;
}

void addPoint(SfMData& cloud, IndexT& id, const Point3d& p, float colourValue)
{
    Landmark L;
    L.X = p;
    jetColor(colourValue, L.rgb);
    cloud.landmarks[id++] = L;
}

void savePointCloud(const SfMData& cloud, const std::string& filepath)
{
    saveSfMDataAlembic(cloud, filepath, /*withViews*/ false, /*withObservations*/ false); // ESfMData::STRUCTURE
}

// the point of pixel (x, y) on the fronto-parallel plane at `planeDepth` (volumeIO.cpp:176-180)
Point3d planePoint(const MultiViewParams& mp, int camIndex, double x, double y, double planeDepth)
{
    const Point3d planen = (mp.iRArr[camIndex] * Point3d(0.0, 0.0, 1.0)).normalize();
    const Point3d planep = mp.CArr[camIndex] + planen * planeDepth;
    const Point3d v = (mp.iCamArr[camIndex] * Point2d(x, y)).normalize();
    return linePlaneIntersect(mp.CArr[camIndex], v, planep, planen);
}

} // namespace

void exportSimilarityVolume(const HostVolume& vol, const std::vector<float>& depths, const MultiViewParams& mp, int camIndex, const SgmParams& sgmParams,
                            const std::string& filepath, const ROI& roi)
{
    SfMData cloud;
    const int xyStep = 10;
    IndexT landmarkId = 0; // (uninitialised in the reference, volumeIO.cpp:159)
    const int nz = std::min<int>((int)depths.size(), vol.Z);
    for(int vy = 0; vy < vol.Y; vy += xyStep)
        for(int vx = 0; vx < vol.X; vx += xyStep)
        {
            const double x = roi.x.begin + (vx * sgmParams.scale * sgmParams.stepXY);
            const double y = roi.y.begin + (vy * sgmParams.scale * sgmParams.stepXY);
            for(int vz = 0; vz < nz; ++vz)
            {
                const float maxValue = 80.f, simValue = vol.at(vx, vy, vz);
                if(simValue > maxValue)
                    continue;
                addPoint(cloud, landmarkId, planePoint(mp, camIndex, x, y, depths[vz]), simValue / maxValue);
            }
        }
    savePointCloud(cloud, filepath);
}

void exportSimilarityVolumeCross(const HostVolume& vol, const std::vector<float>& depths, const MultiViewParams& mp, int camIndex,
                                 const SgmParams& sgmParams, const std::string& filepath, const ROI& roi)
{
    SfMData cloud;
    IndexT landmarkId = 0; // (uninitialised in the reference, volumeIO.cpp:206)
    const int nz = std::min<int>((int)depths.size(), vol.Z);
    for(int vz = 0; vz < nz; ++vz)
        for(int vy = 0; vy < vol.Y; ++vy)
        {
            const bool vyCenter = (vy >= vol.Y / 2) && ((vy - 1) < vol.Y / 2);
            const int xIdxStart = vyCenter ? 0 : (vol.X / 2);
            const int xIdxStop = vyCenter ? vol.X : (xIdxStart + 1);
            for(int vx = xIdxStart; vx < xIdxStop; ++vx)
            {
                const double x = roi.x.begin + (vx * sgmParams.scale * sgmParams.stepXY);
                const double y = roi.y.begin + (vy * sgmParams.scale * sgmParams.stepXY);
                const float maxValue = 80.f, simValue = vol.at(vx, vy, vz);
                if(simValue > maxValue)
                    continue;
                addPoint(cloud, landmarkId, planePoint(mp, camIndex, x, y, depths[vz]), simValue / maxValue);
            }
        }
    savePointCloud(cloud, filepath);
}

void exportSimilarityVolumeCross(const HostVolume& vol, const Float2Tile& dps, const MultiViewParams& mp, int camIndex, const RefineParams& refineParams,
                                 const std::string& filepath, const ROI& roi)
{
    SfMData cloud;
    IndexT landmarkId = 0;
    for(int vy = 0; vy < vol.Y; ++vy)
    {
        const bool vyCenter = (vy * 2) == vol.Y;
        const int xIdxStart = vyCenter ? 0 : (vol.X / 2);
        const int xIdxStop = vyCenter ? vol.X : (xIdxStart + 1);
        for(int vx = xIdxStart; vx < xIdxStop; ++vx)
        {
            const int x = (int)(roi.x.begin + (double(vx) * refineParams.scale * refineParams.stepXY));
            const int y = (int)(roi.y.begin + (double(vy) * refineParams.scale * refineParams.stepXY));
            const Point2d pix(x, y);
            const float depth0 = dps.data[((size_t)vy * dps.width + vx) * 2], pixSize = dps.data[((size_t)vy * dps.width + vx) * 2 + 1];
            if(depth0 < 0.0f) // original depth invalid or masked
                continue;
            for(int vz = 0; vz < vol.Z; ++vz)
            {
                const float simValue = vol.at(vx, vy, vz), maxValue = 10.f; // sum of similarity between 0 and 1
                if(simValue > maxValue)
                    continue;
                const int relativeDepthIndexOffset = vz - refineParams.halfNbDepths;
                const double depth = depth0 + (relativeDepthIndexOffset * pixSize);
                addPoint(cloud, landmarkId, mp.CArr[camIndex] + (mp.iCamArr[camIndex] * pix).normalize() * depth, simValue / maxValue);
            }
        }
    }
    savePointCloud(cloud, filepath);
}

void exportSimilarityVolumeTopographicCut(const HostVolume& vol, const std::vector<float>& depths, const MultiViewParams& mp, int camIndex,
                                          const SgmParams& sgmParams, const std::string& filepath, const ROI& roi)
{
    SfMData cloud;
    const int vy = divideRoundUp(vol.Y, 2); // centre only
    if(vy >= vol.Y)
        return savePointCloud(cloud, filepath);
    const int nz = std::min<int>((int)depths.size(), vol.Z);
    float minSim = std::numeric_limits<float>::max(), maxSim = std::numeric_limits<float>::min();
    for(int vx = 0; vx < vol.X; ++vx)
        for(int vz = 0; vz < nz; ++vz) // (the reference also scans the planes past the depth list: 255, which it skips)
        {
            const float simValue = vol.at(vx, vy, vz);
            if(simValue > 254.f) // invalid similarity
                continue;
            maxSim = std::max(maxSim, simValue);
            minSim = std::min(minSim, simValue);
        }
    const float simNorm = (maxSim == minSim) ? 0.f : (1.f / (maxSim - minSim));
    const Point3d planen = (mp.iRArr[camIndex] * Point3d(0.0, 0.0, 1.0)).normalize();
    IndexT landmarkId = 0;
    for(int vx = 0; vx < vol.X; ++vx)
    {
        const double x = roi.x.begin + (vx * sgmParams.scale * sgmParams.stepXY);
        const double y = roi.y.begin + (vy * sgmParams.scale * sgmParams.stepXY);
        for(int vz = 0; vz < nz; ++vz)
        {
            const float simValue = vol.at(vx, vy, vz);
            if(simValue > 254.f)
                continue;
            const float simValueNorm = (simValue - minSim) * simNorm;
            const Point3d planep = mp.CArr[camIndex] + planen * (double)depths[vz];
            const Point3d v = (mp.iCamArr[camIndex] * Point2d(x, y + simValueNorm * 15.0)).normalize();
            addPoint(cloud, landmarkId, linePlaneIntersect(mp.CArr[camIndex], v, planep, planen), simValueNorm);
        }
    }
    savePointCloud(cloud, filepath);
}

void exportSimilarityVolumeTopographicCut(const HostVolume& vol, const Float2Tile& dps, const MultiViewParams& mp, int camIndex,
                                          const RefineParams& refineParams, const std::string& filepath, const ROI& roi)
{
    SfMData cloud;
    const int vy = divideRoundUp(vol.Y, 2); // centre only
    if(vy >= vol.Y)
        return savePointCloud(cloud, filepath);
    const float minSim = 0.f;
    float maxSim = std::numeric_limits<float>::epsilon();
    for(int vx = 0; vx < vol.X; ++vx)
        for(int vz = 0; vz < vol.Z; ++vz)
            maxSim = std::max(maxSim, vol.at(vx, vy, vz));
    IndexT landmarkId = 0;
    for(int vx = 0; vx < vol.X; ++vx)
    {
        const double x = roi.x.begin + (vx * refineParams.scale * refineParams.stepXY);
        const double y = roi.y.begin + (vy * refineParams.scale * refineParams.stepXY);
        const float depth0 = dps.data[((size_t)vy * dps.width + vx) * 2], pixSize = dps.data[((size_t)vy * dps.width + vx) * 2 + 1];
        if(depth0 < 0.0f) // middle depth (SGM) invalid or masked
            continue;
        for(int vz = 0; vz < vol.Z; ++vz)
        {
            const float simValue = vol.at(vx, vy, vz);
            const float simValueNorm = (simValue - minSim) / (maxSim - minSim);
            const float simValueColor = 1 - simValueNorm; // best similarity value is 0, worst value is 1
            const int relativeDepthIndexOffset = vz - refineParams.halfNbDepths;
            const double depth = depth0 + (relativeDepthIndexOffset * pixSize);
            const Point3d p = mp.CArr[camIndex] + (mp.iCamArr[camIndex] * Point2d(x, y - simValueNorm * 15.0)).normalize() * depth;
            addPoint(cloud, landmarkId, p, simValueColor);
        }
    }
    savePointCloud(cloud, filepath);
}

void writeDepthSimMapFromTileList(int rc, const MultiViewParams& mp, const TileParams& tileParams, const std::vector<ROI>& tileRoiList,
                                  const std::vector<Float2Tile>& in_depthSimMapTiles, int scale, int step, const std::string& name)
{
    AVDM_LOG_TRACE("Merge and write depth/similarity map tiles (rc: " << rc << ", view id: " << mp.getViewId(rc) << ").");
    const std::string customSuffix = name.empty() ? "" : "_" + name;
    const ROI imageRoi(Range(0, mp.getWidth(rc)), Range(0, mp.getHeight(rc)));
    const int scaleStep = scale * step;
    const int width = divideRoundUp(mp.getWidth(rc), scaleStep);
    const int height = divideRoundUp(mp.getHeight(rc), scaleStep);
    // the two channels are independent from the tile buffers to the files: merged and written by two threads (the last batch of a job has
    // nothing to hide behind — its merge + write is on the critical path)
    const TileParams defaultTileParams; // the merged maps are written with DEFAULT tile parameters and the full-size ROI (mapIO.hpp:118-130)
    const ROI fullRoi(0, mp.getWidth(rc), 0, mp.getHeight(rc));
    auto channel = [&](int c, EFileType fileType) {
        const auto tC0 = std::chrono::steady_clock::now();
        FloatMap map(width, height, 0.0f);
        // the tile map is allocated once (largest tile) and reshaped per tile: allocating and freeing multi-megabyte vectors per
        // tile means mmap / munmap each time, and every munmap interrupts all the cores the OpenMP team runs on
        FloatMap tileMap;
        {
            size_t maxPixels = 0;
            for(const ROI& t : tileRoiList)
            {
                const ROI r = downscaleROI(intersect(t, imageRoi), (float)scaleStep);
                maxPixels = std::max(maxPixels, (size_t)r.width() * r.height());
            }
            tileMap.data.reserve(maxPixels);
        }
        for(size_t i = 0; i < tileRoiList.size(); ++i)
        {
            const ROI roi = intersect(tileRoiList.at(i), imageRoi);
            if(roi.isEmpty())
                continue;
            const ROI r = downscaleROI(roi, (float)scaleStep);
            const int w = (int)r.width(), h = (int)r.height();
            const Float2Tile& t = in_depthSimMapTiles.at(i);
            tileMap.reshape(w, h);
#pragma omp parallel for schedule(static) num_threads(8)
            for(int y = 0; y < h; ++y)
                for(int x = 0; x < w; ++x)
                    tileMap(y, x) = t.data[((size_t)y * t.width + x) * 2 + c];
            addTileMapWeighted(rc, mp, tileParams, roi, scaleStep, tileMap, map);
        }
        const auto tW0 = std::chrono::steady_clock::now();
        writeMap(rc, mp, fileType, defaultTileParams, fullRoi, map, scale, step, customSuffix);
        AVDM_LOG_DEBUG("Map of rc " << rc << " (channel " << c << "): tiles merged in " << std::chrono::duration<double>(tW0 - tC0).count() << " s, written in "
                                    << std::chrono::duration<double>(std::chrono::steady_clock::now() - tW0).count() << " s.");
    };
    std::future<void> sim = std::async(std::launch::async, channel, 1, EFileType::simMap);
    try
    {
        channel(0, EFileType::depthMap);
    }
    catch(...)
    {
        sim.wait();
        throw;
    }
    sim.get();
}

namespace {
void mergeFloatMapTiles(int rc, const MultiViewParams& mp, EFileType fileType, int scale, int step, const std::string& name)
{
    const std::string customSuffix = name.empty() ? "" : "_" + name;
    FloatMap map;
    readMap(rc, mp, fileType, map, scale, step, customSuffix);
    const TileParams defaultTileParams;
    writeMap(rc, mp, fileType, defaultTileParams, ROI(0, mp.getWidth(rc), 0, mp.getHeight(rc)), map, scale, step, customSuffix);
    deleteMapTiles(rc, mp, fileType, customSuffix);
}
} // namespace

void mergeDepthSimMapTiles(int rc, const MultiViewParams& mp, int scale, int step, const std::string& name)
{
    mergeFloatMapTiles(rc, mp, EFileType::depthMap, scale, step, name);
    mergeFloatMapTiles(rc, mp, EFileType::simMap, scale, step, name);
}
void mergeDepthPixSizeMapTiles(int rc, const MultiViewParams& mp, int scale, int step, const std::string& name)
{
    mergeFloatMapTiles(rc, mp, EFileType::depthMap, scale, step, name);
    mergeFloatMapTiles(rc, mp, EFileType::pixSizeMap, scale, step, name);
}
void mergeNormalMapTiles(int rc, const MultiViewParams& mp, int scale, int step, const std::string& name)
{
    // three-channel variant of mergeFloatMapTiles: read every tile, weight and add per channel
    const std::string customSuffix = name.empty() ? "" : "_" + name;
    std::vector<std::string> tiles;
    getTilePathList(rc, mp, EFileType::normalMap, customSuffix, tiles);
    if(tiles.empty())
        return;
    const int scaleStep = scale * step;
    const int width = divideRoundUp(mp.getWidth(rc), scaleStep), height = divideRoundUp(mp.getHeight(rc), scaleStep);
    FloatMap acc[3] = {FloatMap(width, height, 0.f), FloatMap(width, height, 0.f), FloatMap(width, height, 0.f)};
    TileParams tileParams;
    bool first = true;
    for(const std::string& p : tiles)
    {
        ExrImage img;
        readExr(p, img);
        int bx = 0, by = 0, ex = 0, ey = 0;
        img.attributes.getInt("AliceVision:roiBeginX", bx);
        img.attributes.getInt("AliceVision:roiBeginY", by);
        img.attributes.getInt("AliceVision:roiEndX", ex);
        img.attributes.getInt("AliceVision:roiEndY", ey);
        if(first)
        {
            img.attributes.getInt("AliceVision:tileBufferWidth", tileParams.bufferWidth);
            img.attributes.getInt("AliceVision:tileBufferHeight", tileParams.bufferHeight);
            img.attributes.getInt("AliceVision:tilePadding", tileParams.padding);
            first = false;
        }
        const char* names[3] = {"R", "G", "B"};
        for(int c = 0; c < 3; ++c)
        {
            const int ci = img.channelIndex(names[c]);
            if(ci < 0)
                continue;
            FloatMap t(img.width, img.height);
            t.data = img.channels[ci];
            addTileMapWeighted(rc, mp, tileParams, ROI(bx, ex, by, ey), scaleStep, t, acc[c]);
        }
    }
    std::vector<float> rgb((size_t)width * height * 3);
    for(size_t i = 0; i < (size_t)width * height; ++i)
        for(int c = 0; c < 3; ++c)
            rgb[3 * i + c] = acc[c].data[i];
    const TileParams defaultTileParams;
    writeMap3(rc, mp, EFileType::normalMap, defaultTileParams, ROI(0, mp.getWidth(rc), 0, mp.getHeight(rc)), rgb, width, height, scale, step, customSuffix);
    deleteMapTiles(rc, mp, EFileType::normalMap, customSuffix);
}

void exportDepthSimMapTilePatternObj(int rc, const MultiViewParams& mp, const std::vector<ROI>& tileRoiList,
                                     const std::vector<std::pair<float, float>>& tileMinMaxDepthsList)
{
    // same vertices / faces as depthMapUtils.cpp:342-452 (6 bevel vertices and 4 faces per ROI corner + first / last depth
    // faces), written directly as Wavefront OBJ with per-vertex colours (the reference goes through assimp's "objnomtl")
    const std::string filepath = getFileNameFromIndex(mp, rc, EFileType::tilePattern);
    std::ofstream f(filepath);
    if(!f)
        AVDM_THROW_ERROR("cannot write '" << filepath << "'");
    const double colors[6][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {0, 1, 1}, {1, 0, 1}};
    const double cornerPixSize = tileRoiList.front().x.size() / 5;
    const Point2d offs[4][2] = {{{cornerPixSize, 0.0}, {0.0, cornerPixSize}},
                                {{cornerPixSize, 0.0}, {0.0, -cornerPixSize}},
                                {{-cornerPixSize, 0.0}, {0.0, cornerPixSize}},
                                {{-cornerPixSize, 0.0}, {0.0, -cornerPixSize}}};
    auto linePlane = [](const Point3d& lp, const Point3d& lv, const Point3d& pp, const Point3d& pn) {
        const double k = (dot(pp, pn) - dot(pn, lp)) / dot(pn, lv);
        return lp + lv * k;
    };
    std::vector<Point3d> vertices;
    std::vector<int> faces;
    for(std::size_t ri = 0; ri < tileRoiList.size(); ++ri)
    {
        const ROI& roi = tileRoiList.at(ri);
        const auto& mm = tileMinMaxDepthsList.at(ri);
        const Point3d planeN = (mp.iRArr[rc] * Point3d(0.0, 0.0, 1.0)).normalize();
        const Point3d firstPlaneP = mp.CArr[rc] + planeN * mm.first, lastPlaneP = mp.CArr[rc] + planeN * mm.second;
        const Point2d corners[4] = {{double(roi.x.begin), double(roi.y.begin)},
                                    {double(roi.x.begin), double(roi.y.end)},
                                    {double(roi.x.end), double(roi.y.begin)},
                                    {double(roi.x.end), double(roi.y.end)}};
        const int v0 = (int)vertices.size();
        for(int ci = 0; ci < 4; ++ci)
        {
            const int vs = (int)vertices.size();
            const Point2d pts[3] = {corners[ci], corners[ci] + offs[ci][0], corners[ci] + offs[ci][1]};
            for(const Point2d& p : pts)
            {
                const Point3d dir = (mp.iCamArr[rc] * p).normalize();
                vertices.push_back(linePlane(mp.CArr[rc], dir, firstPlaneP, planeN));
                vertices.push_back(linePlane(mp.CArr[rc], dir, lastPlaneP, planeN));
            }
            const int fs[4][3] = {{vs, vs + 1, vs + 2}, {vs + 1, vs + 2, vs + 3}, {vs, vs + 1, vs + 4}, {vs + 1, vs + 4, vs + 5}};
            for(const auto& t : fs)
                faces.insert(faces.end(), t, t + 3);
        }
        const int fl[2][3] = {{v0, v0 + 6, v0 + 12}, {v0 + 7, v0 + 13, v0 + 19}};
        for(const auto& t : fl)
            faces.insert(faces.end(), t, t + 3);
    }
    for(size_t i = 0; i < vertices.size(); ++i)
    {
        const double* c = colors[(i / 24) % 6];
        f << "v " << vertices[i].x << " " << -vertices[i].y << " " << -vertices[i].z << " " << c[0] << " " << c[1] << " " << c[2] << "\n";
    }
    for(size_t i = 0; i < faces.size(); i += 3)
        f << "f " << faces[i] + 1 << " " << faces[i + 1] + 1 << " " << faces[i + 2] + 1 << "\n";
    AVDM_LOG_INFO("Save debug tiles pattern obj (rc: " << rc << ", view id: " << mp.getViewId(rc) << ") done.");
}

} // namespace avdm_host
