// depthMapUtils.hpp — device map -> host copies, tile weighting / merging and EXR output with the AliceVision metadata.
// Restates depthMap/depthMapUtils.cpp:22-330 (copyFloat2Map, writeFloat2Map, write*Map, writeDepthSimMapFromTileList,
// resetDepthSimMap, merge*MapTiles) and mvsUtils/mapIO.cpp:170-311 (weightTileBorder, addSingleTileMapWeighted), :313-400
// (readMapFromFileOrTiles), :402-540 (writeMapToFileOrTile), :640-687 (deleteMapTiles).
#pragma once

#include "MultiViewParams.hpp"
#include "params.hpp"

#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

namespace avdm_host {

// image::Image<float>: row-major
struct FloatMap
{
    int width = 0, height = 0;
    std::vector<float> data;
    FloatMap() = default;
    FloatMap(int w, int h, float v = 0.f) : width(w), height(h), data((size_t)w * h, v) {}
    // change the shape, keeping the allocation when it is large enough (contents unspecified)
    void reshape(int w, int h)
    {
        width = w, height = h;
        data.resize((size_t)w * h);
    }
    float& operator()(int y, int x) { return data[(size_t)y * width + x]; }
    float operator()(int y, int x) const { return data[(size_t)y * width + x]; }
};

// host copy of a tile's float2 map: interleaved (x, y) pairs, row-major, `width` pairs per row (CudaHostMemoryHeap<float2, 2>)
// Storage of a result tile: 2 MiB-aligned and marked for transparent huge pages (madvise).  A tile is several MB that is zero-filled, page-
// locked (hipHostRegister) and written by the device: with 4 KiB pages that is ~1 400 page faults and as many pages to pin per tile — the
// page-locking of a 12 MP job's 80 tiles took 0.25 s; with 2 MiB pages it is three of each.
template <typename T>
struct HugePageAllocator
{
    using value_type = T;
    HugePageAllocator() = default;
    template <typename U>
    HugePageAllocator(const HugePageAllocator<U>&) {}
    T* allocate(size_t n);
    void deallocate(T* p, size_t) noexcept;
    template <typename U>
    bool operator==(const HugePageAllocator<U>&) const { return true; }
    template <typename U>
    bool operator!=(const HugePageAllocator<U>&) const { return false; }
};
void* hugePageAlloc(size_t bytes);
void hugePageFree(void* p);
template <typename T>
T* HugePageAllocator<T>::allocate(size_t n) { return static_cast<T*>(hugePageAlloc(n * sizeof(T))); }
template <typename T>
void HugePageAllocator<T>::deallocate(T* p, size_t) noexcept { hugePageFree(p); }

struct Float2Tile
{
    int width = 0, height = 0;
    std::vector<float, HugePageAllocator<float>> data;
    void allocate(int w, int h)
    {
        width = w, height = h;
        data.assign((size_t)w * h * 2, 0.f);
    }
};

// depthMapUtils.cpp:279-293
void resetDepthSimMap(Float2Tile& inout, float depth = -1.f, float sim = 1.f);

// mapIO.cpp:213-311
void addTileMapWeighted(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, int downscale, FloatMap& in_tileMap, FloatMap& inout_map);

// mapIO.cpp:402-540: nbChannels = 1 (float maps) or 3 (normal maps: interleaved RGB)
void writeMap(int rc, const MultiViewParams& mp, EFileType fileType, const TileParams& tileParams, const ROI& roi, const FloatMap& in_map, int scale, int step,
              const std::string& customSuffix = "");
void writeMap3(int rc, const MultiViewParams& mp, EFileType fileType, const TileParams& tileParams, const ROI& roi, const std::vector<float>& rgb, int width,
               int height, int scale, int step, const std::string& customSuffix = "");
// mapIO.cpp:313-400
void readMap(int rc, const MultiViewParams& mp, EFileType fileType, FloatMap& out_map, int scale, int step, const std::string& customSuffix = "");
void deleteMapTiles(int rc, const MultiViewParams& mp, EFileType fileType, const std::string& customSuffix = "");

// depthMapUtils.cpp:196-238 for device maps (float2 rows, pitch in bytes): copies the tile ROI to the host on `stream` (synchronises it)
void writeDepthSimMap(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, const float* map_d, int pitch, int scale, int step,
                      const std::string& name, hipStream_t stream);
void writeDepthPixSizeMap(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, const float* map_d, int pitch, int scale, int step,
                          const std::string& name, hipStream_t stream);
void writeNormalMap(int rc, const MultiViewParams& mp, const TileParams& tileParams, const ROI& roi, const float* map_d, int pitch, int scale, int step,
                    const std::string& name, hipStream_t stream);

// volumeIO.cpp:28-146 (exportSimilaritySamplesCSV): the similarity over the planes at a 3 x 3 grid of pixels of the tile, appended to `filepath`.
// The volume is z-fastest here: a sample is one contiguous run of nbPlanes elements (uint8 for the SGM volumes, fp16 for the Refine volume).
void exportSimilaritySamplesCSV(const void* volume_d, bool halfFloat, long long pitchY, int pitchX, int nbPlanes, int width, int height, const std::string& name,
                                const std::string& filepath, hipStream_t stream);

// ---- debug exports of the similarity volumes as coloured point clouds (depthMap/volumeIO.cpp:148-441), saved as Alembic archives like
// the reference's sfmDataIO::save(pointCloud, path, ESfMData::STRUCTURE).  A volume is copied to the host once and sampled there.
struct HostVolume
{
    std::vector<unsigned char> bytes;
    long long pitchY = 0;
    int pitchX = 0, X = 0, Y = 0, Z = 0;
    bool halfFloat = false;
    float at(int x, int y, int z) const;
};
// X, Y: the extent the volume is laid out for (the reference's volDim: the allocation, not the tile's ROI); Z: number of planes held
HostVolume downloadVolume(const void* volume_d, bool halfFloat, long long pitchY, int pitchX, int X, int Y, int Z, hipStream_t stream);
// volumeIO.cpp:148-194: every 10th voxel column of the SGM volume, jet-coloured by similarity / 80 (similarities above 80 skipped)
void exportSimilarityVolume(const HostVolume& vol, const std::vector<float>& depths, const MultiViewParams& mp, int camIndex, const SgmParams& sgmParams,
                            const std::string& filepath, const ROI& roi);
// volumeIO.cpp:196-246 (SGM volume) and :248-304 (Refine volume around the up-scaled SGM depth / pixel-size map): the centre row and column
void exportSimilarityVolumeCross(const HostVolume& vol, const std::vector<float>& depths, const MultiViewParams& mp, int camIndex,
                                 const SgmParams& sgmParams, const std::string& filepath, const ROI& roi);
void exportSimilarityVolumeCross(const HostVolume& vol, const Float2Tile& depthPixSizeMapSgmUpscale, const MultiViewParams& mp, int camIndex,
                                 const RefineParams& refineParams, const std::string& filepath, const ROI& roi);
// volumeIO.cpp:306-376 and :378-441: the centre row, every voxel lifted by its normalised similarity (a profile of the cost per plane)
void exportSimilarityVolumeTopographicCut(const HostVolume& vol, const std::vector<float>& depths, const MultiViewParams& mp, int camIndex,
                                          const SgmParams& sgmParams, const std::string& filepath, const ROI& roi);
void exportSimilarityVolumeTopographicCut(const HostVolume& vol, const Float2Tile& depthPixSizeMapSgmUpscale, const MultiViewParams& mp, int camIndex,
                                          const RefineParams& refineParams, const std::string& filepath, const ROI& roi);
// image/jetColorMap.cpp:34-53 (getRGBFromJetColorMap): MATLAB's jet(64), linearly interpolated; <= 0 black, >= 1 white
void jetColor(float value, unsigned char rgb[3]);

// depthMapUtils.cpp:240-277
void writeDepthSimMapFromTileList(int rc, const MultiViewParams& mp, const TileParams& tileParams, const std::vector<ROI>& tileRoiList,
                                  const std::vector<Float2Tile>& in_depthSimMapTiles, int scale, int step, const std::string& name = "");

// depthMapUtils.cpp:295-340
void mergeDepthSimMapTiles(int rc, const MultiViewParams& mp, int scale, int step, const std::string& name = "");
void mergeDepthPixSizeMapTiles(int rc, const MultiViewParams& mp, int scale, int step, const std::string& name = "");
void mergeNormalMapTiles(int rc, const MultiViewParams& mp, int scale, int step, const std::string& name = "");

// depthMapUtils.cpp:342-490: Wavefront OBJ of the tile frusta
void exportDepthSimMapTilePatternObj(int rc, const MultiViewParams& mp, const std::vector<ROI>& tileRoiList, const std::vector<std::pair<float, float>>& tileMinMaxDepthsList);

} // namespace avdm_host
