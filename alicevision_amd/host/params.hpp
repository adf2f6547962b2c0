// params.hpp — user / constant parameters of the depth-map stage and the tile descriptor.
// Same names and defaults as the reference: depthMap/SgmParams.hpp:21-55, depthMap/RefineParams.hpp:19-45,
// depthMap/DepthMapParams.hpp:21-35, mvsUtils/TileParams.hpp:19-27, depthMap/Tile.hpp:21-29.
#pragma once

#include "mvsData.hpp"

#include <avdm.h>

#include <cstring>
#include <ostream>
#include <string>
#include <vector>

namespace avdm_host {

struct SgmParams
{
    // user parameters
    int scale = 2;
    int stepXY = 2;
    int stepZ = -1;
    int wsh = 4;
    int maxDepths = 1500;
    int maxTCamsPerTile = 4;
    double seedsRangeInflate = 0.2;
    double depthThicknessInflate = 0.0;
    double maxSimilarity = 1.0;
    double gammaC = 5.5;
    double gammaP = 8.0;
    double p1 = 10;
    double p2Weighting = 100.0;
    std::string filteringAxes = "YX";
    bool useSfmSeeds = true;
    bool depthListPerTile = false;
    bool useConsistentScale = false;
    bool useCustomPatchPattern = false;
    // not a parameter of the reference: --sgmReferenceArithmetic 1 runs the similarity sweep in the reference's arithmetic as written
    // (avdm_sgm_params_t::referenceArithmetic: volumes equal to the reference's own code compiled for the CPU bit for bit, ~6 x the sweep's cost)
    bool referenceArithmetic = false;
    bool exportIntermediateDepthSimMaps = false;
    bool exportIntermediateNormalMaps = false;
    bool exportIntermediateVolumes = false;
    bool exportIntermediateCrossVolumes = false;
    bool exportIntermediateTopographicCutVolumes = false;
    bool exportIntermediateVolume9pCsv = false;
    bool exportDepthsTxtFiles = false; // const false in the reference (SgmParams.hpp:47)

    // constant parameters of the reference; the two marked (*) are exposed as hidden CLI switches
    bool updateUninitializedSim = true;
    bool doSgmOptimizeVolume = true; // (*) --sgmOptimizeVolume
    double prematchingMaxDepthScale = 1.5;
    double seedsRangePercentile = 0.999;

    avdm_sgm_params_t toAvdm() const
    {
        avdm_sgm_params_t p;
        std::memset(&p, 0, sizeof(p));
        p.scale = scale, p.stepXY = stepXY, p.wsh = wsh;
        p.gammaC = gammaC, p.gammaP = gammaP, p.p1 = p1, p.p2Weighting = p2Weighting;
        p.maxSimilarity = maxSimilarity, p.depthThicknessInflate = depthThicknessInflate;
        std::strncpy(p.filteringAxes, filteringAxes.c_str(), sizeof(p.filteringAxes) - 1);
        p.useConsistentScale = useConsistentScale ? 1 : 0;
        p.strictRoiQuirk = 1;
        p.useCustomPatchPattern = useCustomPatchPattern ? 1 : 0;
        p.referenceArithmetic = referenceArithmetic ? 1 : 0;
        return p;
    }
};

struct RefineParams
{
    int scale = 1;
    int stepXY = 1;
    int wsh = 3;
    int halfNbDepths = 15;
    int nbSubsamples = 10;
    int maxTCamsPerTile = 4;
    int optimizationNbIterations = 100;
    double sigma = 15.0;
    double gammaC = 15.5;
    double gammaP = 8.0;
    bool interpolateMiddleDepth = false;
    bool useConsistentScale = false;
    bool useCustomPatchPattern = false;
    bool referenceArithmetic = false; // --refineReferenceArithmetic (see SgmParams::referenceArithmetic)
    bool useRefineFuse = true;
    bool useColorOptimization = true;
    bool useSgmNormalMap = false; // const false in the reference (RefineParams.hpp:44)
    bool exportIntermediateDepthSimMaps = false;
    bool exportIntermediateNormalMaps = false;
    bool exportIntermediateCrossVolumes = false;
    bool exportIntermediateTopographicCutVolumes = false;
    bool exportIntermediateVolume9pCsv = false;

    avdm_refine_params_t toAvdm() const
    {
        avdm_refine_params_t p;
        std::memset(&p, 0, sizeof(p));
        p.scale = scale, p.stepXY = stepXY, p.wsh = wsh, p.halfNbDepths = halfNbDepths, p.nbSubsamples = nbSubsamples;
        p.optimizationNbIterations = optimizationNbIterations;
        p.sigma = sigma, p.gammaC = gammaC, p.gammaP = gammaP;
        p.interpolateMiddleDepth = interpolateMiddleDepth ? 1 : 0;
        p.useConsistentScale = useConsistentScale ? 1 : 0;
        p.useCustomPatchPattern = useCustomPatchPattern ? 1 : 0;
        p.referenceArithmetic = referenceArithmetic ? 1 : 0;
        return p;
    }
};

// DM/CustomPatchPatternParams.hpp:19-36
struct CustomPatchPatternParams
{
    struct SubpartParams
    {
        bool isCircle = false;
        int level = 0;
        int nbCoordinates = 0;
        float radius = 0.f;
        float weight = 0.f;
    };
    std::vector<SubpartParams> subpartsParams;
    bool groupSubpartsPerLevel = false;
};

struct DepthMapParams
{
    int maxTCams = 10;
    bool chooseTCamsPerTile = true;
    bool exportTilePattern = false;
    bool autoAdjustSmallImage = true;
    CustomPatchPatternParams customPatchPattern; // DepthMapParams.hpp:31
    bool useRefine = true; // const true in the reference (DepthMapParams.hpp:34); hidden CLI switch --useRefine
};

struct TileParams
{
    int bufferWidth = 1024;
    int bufferHeight = 1024;
    int padding = 64;
};
// mvsUtils/TileParams.hpp:35-38 — note: bufferHeight is tested against BOTH image dimensions, as in the reference
inline bool hasOnlyOneTile(const TileParams& tp, int imageWidth, int imageHeight) { return tp.bufferHeight >= imageWidth && tp.bufferHeight >= imageHeight; }
// mvsUtils/TileParams.cpp:15-61
void getTileRoiList(const TileParams& tileParams, int imageWidth, int imageHeight, int maxDownscale, std::vector<ROI>& out_tileRoiList);

struct Tile
{
    int id = 0;
    int nbTiles = 0;
    int rc = 0;
    std::vector<int> sgmTCams;
    std::vector<int> refineTCams;
    ROI roi;
};
// depthMap/Tile.hpp:31-36
inline std::ostream& operator<<(std::ostream& os, const Tile& tile)
{
    os << "(rc: " << tile.rc << ", tile: " << (tile.id + 1) << "/" << tile.nbTiles << ") ";
    return os;
}

} // namespace avdm_host
