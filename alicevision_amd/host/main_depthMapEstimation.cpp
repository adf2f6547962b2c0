// aliceVision_depthMapEstimation — the process boundary Meshroom sees: same flags, defaults, checks and
// parameter adjustments as software/pipeline/main_depthMapEstimation.cpp:50-435 of the reference (+ the common options of
// cmdline.cpp:10-26), SfMData in, <viewId>_depthMap.exr / <viewId>_simMap.exr out.
// Hidden switches that the reference only has as compile-time constants: --sgmOptimizeVolume, --useRefine;
// and --dryRun 1 (print the tile / T-camera / depth-plane plan as JSON on stdout and stop: no GPU needed).
#include "DepthMapEstimator.hpp"
#include "MultiViewParams.hpp"
#include "cmdline.hpp"
#include "log.hpp"
#include "params.hpp"
#include "sfmData.hpp"

#include <avdm.h>
#include <omp.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <iomanip>
#include <iostream>
#include <map>
#include <sstream>
#include <algorithm>
#include <string>
#include <vector>

using namespace avdm_host;

namespace {

// main_depthMapEstimation.cpp:29-48
int computeDownscale(const MultiViewParams& mp, int scale, int maxWidth, int maxHeight)
{
    const int maxImageWidth = mp.getMaxImageWidth() / scale;
    const int maxImageHeight = mp.getMaxImageHeight() / scale;
    int downscale = 1;
    int downscaleWidth = mp.getMaxImageWidth() / scale;
    int downscaleHeight = mp.getMaxImageHeight() / scale;
    while((downscaleWidth > maxWidth) || (downscaleHeight > maxHeight))
    {
        downscale++;
        downscaleWidth = maxImageWidth / downscale;
        downscaleHeight = maxImageHeight / downscale;
    }
    return downscale;
}

bool dirExists(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}

void printPlanJson(const MultiViewParams& mp, const std::vector<TilePlan>& plans, const SgmParams& sgm, const RefineParams& refine, const TileParams& tp)
{
    std::cout << std::setprecision(9);
    std::cout << "{\"sgmScale\": " << sgm.scale << ", \"sgmStepXY\": " << sgm.stepXY << ", \"sgmMaxTCamsPerTile\": " << sgm.maxTCamsPerTile
              << ", \"refineMaxTCamsPerTile\": " << refine.maxTCamsPerTile << ", \"tileBufferWidth\": " << tp.bufferWidth << ", \"tileBufferHeight\": "
              << tp.bufferHeight << ", \"tilePadding\": " << tp.padding << ", \"tiles\": [";
    for(size_t i = 0; i < plans.size(); ++i)
    {
        const TilePlan& p = plans[i];
        std::cout << (i ? ", " : "") << "{\"rc\": " << p.tile.rc << ", \"viewId\": " << mp.getViewId(p.tile.rc) << ", \"id\": " << p.tile.id
                  << ", \"nbTiles\": " << p.tile.nbTiles << ", \"roi\": [" << p.tile.roi.x.begin << ", " << p.tile.roi.x.end << ", " << p.tile.roi.y.begin << ", "
                  << p.tile.roi.y.end << "], \"sgmTCams\": [";
        for(size_t k = 0; k < p.tile.sgmTCams.size(); ++k)
            std::cout << (k ? ", " : "") << p.tile.sgmTCams[k];
        std::cout << "], \"refineTCams\": [";
        for(size_t k = 0; k < p.tile.refineTCams.size(); ++k)
            std::cout << (k ? ", " : "") << p.tile.refineTCams[k];
        std::cout << "], \"depthsTcLimits\": [";
        for(size_t k = 0; k < p.depthsTcLimits.size(); ++k)
            std::cout << (k ? ", " : "") << "[" << p.depthsTcLimits[k].x << ", " << p.depthsTcLimits[k].y << "]";
        std::cout << "], \"depths\": [";
        for(size_t k = 0; k < p.depths.size(); ++k)
            std::cout << (k ? ", " : "") << p.depths[k];
        std::cout << "]}";
    }
    std::cout << "]}" << std::endl;
}

int aliceVision_main(int argc, char* argv[])
{
    const auto startTime = std::chrono::steady_clock::now();
    std::string sfmDataFilename, outputFolder, imagesFolder, verboseLevel = "info";
    int rangeStart = -1, rangeSize = -1;
    int downscale = 2;
    float minViewAngle = 2.0f, maxViewAngle = 70.0f;
    TileParams tileParams;
    DepthMapParams depthMapParams;
    SgmParams sgmParams;
    RefineParams refineParams;
    bool exportIntermediateDepthSimMaps = false, exportIntermediateNormalMaps = false, exportIntermediateVolumes = false;
    bool exportIntermediateCrossVolumes = false, exportIntermediateTopographicCutVolumes = false, exportIntermediateVolume9pCsv = false;
    int nbGPUs = 0;
    int maxMemoryAvailable = 0, maxCoresAvailable = 0; // cmdline.cpp:10-26 hardware limits
    bool dryRun = false;
    std::vector<std::string> customPatchPatternSubparts; // tokens `type:radius:nbCoords:level:weight` (CustomPatchPatternParams.cpp:16-41)

    CmdLine cmdline("Dense Reconstruction.\n"
                    "This program estimate a depth map for each input calibrated camera using Plane Sweeping, a multi-view stereo algorithm notable "
                    "for its efficiency on modern graphics hardware (GPU).\n"
                    "AliceVision depthMapEstimation");
    // required
    cmdline.add("input", &sfmDataFilename, "SfMData file.", true, 'i');
    cmdline.add("imagesFolder", &imagesFolder, "Images folder. Filename should be the image uid.", true);
    cmdline.add("output", &outputFolder, "Output folder for generated depth maps.", true, 'o');
    // optional
    cmdline.add("rangeStart", &rangeStart, "Compute a sub-range of images from index rangeStart to rangeStart+rangeSize.");
    cmdline.add("rangeSize", &rangeSize, "Compute a sub-range of N images (N=rangeSize).");
    cmdline.add("downscale", &downscale, "Downscale the input images to compute the depth map.");
    cmdline.add("minViewAngle", &minViewAngle, "Minimum angle between two views (select the neighbouring cameras, select depth planes from epipolar segment point).");
    cmdline.add("maxViewAngle", &maxViewAngle, "Maximum angle between two views (select the neighbouring cameras, select depth planes from epipolar segment point).");
    cmdline.add("tileBufferWidth", &tileParams.bufferWidth, "Maximum tile buffer width.");
    cmdline.add("tileBufferHeight", &tileParams.bufferHeight, "Maximum tile buffer height.");
    cmdline.add("tilePadding", &tileParams.padding, "Buffer padding for overlapping tiles.");
    cmdline.add("chooseTCamsPerTile", &depthMapParams.chooseTCamsPerTile, "Choose neighbour cameras per tile or globally to the image.");
    cmdline.add("maxTCams", &depthMapParams.maxTCams, "Maximum number of neighbour cameras per image.");
    cmdline.add("sgmScale", &sgmParams.scale, "Semi Global Matching: Downscale factor applied on source images for the SGM step (in addition to the global downscale).");
    cmdline.add("sgmStepXY", &sgmParams.stepXY, "Semi Global Matching: Step is used to compute the similarity volume for one pixel over N (in the XY image plane).");
    cmdline.add("sgmStepZ", &sgmParams.stepZ, "Semi Global Matching: Initial step used to compute the similarity volume on Z axis (every N pixels on the epilolar line). -1 means automatic estimation.");
    cmdline.add("sgmMaxTCamsPerTile", &sgmParams.maxTCamsPerTile, "Semi Global Matching: Maximum number of neighbour cameras used per tile.");
    cmdline.add("sgmWSH", &sgmParams.wsh, "Semi Global Matching: Half-size of the patch used to compute the similarity. Patch width is wsh*2+1.");
    cmdline.add("sgmUseSfmSeeds", &sgmParams.useSfmSeeds, "Semi Global Matching: Use landmarks from Structure-from-Motion as input seeds to define min/max depth ranges.");
    cmdline.add("sgmSeedsRangeInflate", &sgmParams.seedsRangeInflate, "Semi Global Matching: Inflate factor to add margins around SfM seeds.");
    cmdline.add("sgmDepthThicknessInflate", &sgmParams.depthThicknessInflate, "Semi Global Matching: Inflate factor to add margins to the depth thickness.");
    cmdline.add("sgmMaxSimilarity", &sgmParams.maxSimilarity, "Semi Global Matching: Maximum similarity threshold (between 0 and 1) used to filter out poorly supported depth values.");
    cmdline.add("sgmGammaC", &sgmParams.gammaC, "Semi Global Matching: GammaC threshold used for similarity computation, strength of grouping by color similarity.");
    cmdline.add("sgmGammaP", &sgmParams.gammaP, "Semi Global Matching: GammaP threshold used for similarity computation, strength of grouping by proximity.");
    cmdline.add("sgmP1", &sgmParams.p1, "Semi Global Matching: P1 parameter for SGM filtering.");
    cmdline.add("sgmP2Weighting", &sgmParams.p2Weighting, "Semi Global Matching: P2 weighting parameter for SGM filtering.");
    cmdline.add("sgmMaxDepths", &sgmParams.maxDepths, "Semi Global Matching: Maximum number of depths in the similarity volume.");
    cmdline.add("sgmFilteringAxes", &sgmParams.filteringAxes, "Semi Global Matching: Define axes for the filtering of the similarity volume.");
    cmdline.add("sgmDepthListPerTile", &sgmParams.depthListPerTile, "Semi Global Matching: Select the list of depth planes per tile or globally to the image.");
    cmdline.add("sgmUseConsistentScale", &sgmParams.useConsistentScale, "Semi Global Matching: Compare patch with consistent scale for similarity volume computation.");
    cmdline.add("sgmUseCustomPatchPattern", &sgmParams.useCustomPatchPattern, "Semi Global Matching: Use user custom patch pattern for similarity volume computation.");
    // (not a flag of the reference) the parity mode: the reference's similarity arithmetic as written, bit-equal volumes, ~6 x the sweep's cost
    cmdline.add("sgmReferenceArithmetic", &sgmParams.referenceArithmetic, "Semi Global Matching: Evaluate the similarity volume in the reference implementation's arithmetic as written (bit-equal volumes; slower).");
    cmdline.add("refineReferenceArithmetic", &refineParams.referenceArithmetic, "Refine: Evaluate the refine similarity volume in the reference implementation's arithmetic as written (bit-equal volumes; slower).");
    cmdline.add("refineScale", &refineParams.scale, "Refine: Downscale factor applied on source images for the Refine step (in addition to the global downscale).");
    cmdline.add("refineStepXY", &refineParams.stepXY, "Refine: Step is used to compute the refine volume for one pixel over N (in the XY image plane).");
    cmdline.add("refineMaxTCamsPerTile", &refineParams.maxTCamsPerTile, "Refine: Maximum number of neighbour cameras used per tile.");
    cmdline.add("refineHalfNbDepths", &refineParams.halfNbDepths, "Refine: The thickness of the refine area around the initial depth map.");
    cmdline.add("refineSubsampling", &refineParams.nbSubsamples, "Refine: Number of subsamples used to extract the best depth from the refine volume (sliding gaussian window precision).");
    cmdline.add("refineWSH", &refineParams.wsh, "Refine: Half-size of the patch used to compute the similarity. Patch width is wsh*2+1.");
    cmdline.add("refineSigma", &refineParams.sigma, "Refine: Sigma (2*sigma^2) of the gaussian filter used to extract the best depth from the refine volume.");
    cmdline.add("refineGammaC", &refineParams.gammaC, "Refine: GammaC threshold used for similarity computation.");
    cmdline.add("refineGammaP", &refineParams.gammaP, "Refine: GammaP threshold used for similarity computation.");
    cmdline.add("refineInterpolateMiddleDepth", &refineParams.interpolateMiddleDepth, "Refine: Enable/Disable middle depth bilinear interpolation for the refinement process.");
    cmdline.add("refineUseConsistentScale", &refineParams.useConsistentScale, "Refine: Compare patch with consistent scale for similarity volume computation.");
    cmdline.add("refineUseCustomPatchPattern", &refineParams.useCustomPatchPattern, "Refine: Use user custom patch pattern for similarity volume computation.");
    cmdline.add("colorOptimizationNbIterations", &refineParams.optimizationNbIterations, "Color Optimization: Number of iterations of the optimization.");
    cmdline.add("refineEnabled", &refineParams.useRefineFuse, "Enable/Disable depth/similarity map refinement process.");
    cmdline.add("colorOptimizationEnabled", &refineParams.useColorOptimization, "Enable/Disable depth/similarity map post-process color optimization.");
    cmdline.add("autoAdjustSmallImage", &depthMapParams.autoAdjustSmallImage, "Automatically adjust depth map parameters if images are smaller than one tile (maxTCamsPerTile=maxTCams, adjust step if needed).");
    cmdline.addMultitoken("customPatchPatternSubparts", &customPatchPatternSubparts, "User custom patch pattern subparts for similarity volume computation.");
    cmdline.add("customPatchPatternGroupSubpartsPerLevel", &depthMapParams.customPatchPattern.groupSubpartsPerLevel, "Group all custom patch pattern subparts with the same image level.");
    cmdline.add("exportIntermediateDepthSimMaps", &exportIntermediateDepthSimMaps, "Export intermediate depth/similarity maps from the SGM and Refine steps.");
    cmdline.add("exportIntermediateNormalMaps", &exportIntermediateNormalMaps, "Export intermediate normal maps from the SGM and Refine steps.");
    cmdline.add("exportIntermediateVolumes", &exportIntermediateVolumes, "Export intermediate full similarity volumes from the SGM and Refine steps.");
    cmdline.add("exportIntermediateCrossVolumes", &exportIntermediateCrossVolumes, "Export intermediate similarity cross volumes from the SGM and Refine steps.");
    cmdline.add("exportIntermediateTopographicCutVolumes", &exportIntermediateTopographicCutVolumes, "Export intermediate similarity topographic cut volumes from the SGM and Refine steps.");
    cmdline.add("exportIntermediateVolume9pCsv", &exportIntermediateVolume9pCsv, "Export intermediate volumes 9 points from the SGM and Refine steps in CSV files.");
    cmdline.add("exportTilePattern", &depthMapParams.exportTilePattern, "Export workflow tile pattern.");
    cmdline.add("nbGPUs", &nbGPUs, "Number of GPUs to use (0 means use all GPUs).");
    // common options (cmdline.cpp:10-26)
    cmdline.add("verboseLevel", &verboseLevel, "verbosity level (fatal, error, warning, info, debug, trace).", false, 'v');
    cmdline.add("maxMemoryAvailable", &maxMemoryAvailable, "User specified available RAM");
    cmdline.add("maxCoresAvailable", &maxCoresAvailable, "User specified available number of cores");
    // hidden
    cmdline.add("sgmOptimizeVolume", &sgmParams.doSgmOptimizeVolume, "", false, 0, true);
    cmdline.add("useRefine", &depthMapParams.useRefine, "", false, 0, true);
    cmdline.add("dryRun", &dryRun, "", false, 0, true);

    bool cmdError = false;
    if(!cmdline.execute(argc, argv, cmdError))
        return cmdError ? EXIT_FAILURE : EXIT_SUCCESS;
    if(!Logger::setLevel(verboseLevel))
    {
        std::cerr << "ERROR: invalid verboseLevel '" << verboseLevel << "'" << std::endl;
        return EXIT_FAILURE;
    }
    if(maxCoresAvailable > 0)
        omp_set_num_threads(maxCoresAvailable);
    else if(omp_get_max_threads() > 32)
        omp_set_num_threads(32); // the host work (EXR blocks, depth lists, image decode) is short: on a 256-thread box starting and joining
                                 // the full team costs more per parallel region (~0.2 s measured) than the region's work

    sgmParams.exportIntermediateDepthSimMaps = exportIntermediateDepthSimMaps;
    sgmParams.exportIntermediateNormalMaps = exportIntermediateNormalMaps;
    sgmParams.exportIntermediateVolumes = exportIntermediateVolumes;
    sgmParams.exportIntermediateCrossVolumes = exportIntermediateCrossVolumes;
    sgmParams.exportIntermediateTopographicCutVolumes = exportIntermediateTopographicCutVolumes;
    sgmParams.exportIntermediateVolume9pCsv = exportIntermediateVolume9pCsv;
    refineParams.exportIntermediateDepthSimMaps = exportIntermediateDepthSimMaps;
    refineParams.exportIntermediateNormalMaps = exportIntermediateNormalMaps;
    refineParams.exportIntermediateCrossVolumes = exportIntermediateCrossVolumes;
    refineParams.exportIntermediateTopographicCutVolumes = exportIntermediateTopographicCutVolumes;
    refineParams.exportIntermediateVolume9pCsv = exportIntermediateVolume9pCsv;

    // CustomPatchPatternParams.cpp:16-41 (operator>>): `circle|full:radius:nbCoordinates:level:weight`
    for(const std::string& token : customPatchPatternSubparts)
    {
        std::vector<std::string> parts;
        size_t b = 0;
        while(true)
        {
            const size_t e = token.find(':', b);
            parts.push_back(token.substr(b, e == std::string::npos ? std::string::npos : e - b));
            if(e == std::string::npos)
                break;
            b = e + 1;
        }
        CustomPatchPatternParams::SubpartParams sp;
        try
        {
            if(parts.size() != 5)
                throw std::invalid_argument(token);
            std::string type = parts[0];
            for(char& c : type)
                c = (char)std::tolower((unsigned char)c);
            sp.isCircle = (type == "circle");
            sp.radius = std::stof(parts[1]);
            sp.nbCoordinates = std::stoi(parts[2]);
            sp.level = std::stoi(parts[3]);
            sp.weight = std::stof(parts[4]);
        }
        catch(const std::exception&)
        {
            std::cerr << "ERROR: Failed to parse CustomPatchPatternParams::SubpartParams from: " << token << std::endl;
            return EXIT_FAILURE;
        }
        depthMapParams.customPatchPattern.subpartsParams.push_back(sp);
    }
    if((sgmParams.useCustomPatchPattern || refineParams.useCustomPatchPattern) && depthMapParams.customPatchPattern.subpartsParams.empty())
    {
        // buildCustomPatchPattern would throw at the start of the computation (patchPattern.cpp:20-24); say so before loading anything
        AVDM_LOG_ERROR("Cannot build custom patch pattern: No patch pattern subpart given (--customPatchPatternSubparts).");
        return EXIT_FAILURE;
    }
    if(!dryRun)
    {
        // gpu::gpuInformationCUDA / gpuSupportCUDA (main_depthMapEstimation.cpp:246-255)
        const int nbDevices = avdm_device_count();
        for(int d = 0; d < nbDevices; ++d)
        {
            char info[1024];
            if(avdm_device_info(d, info, sizeof(info)) == 0)
                AVDM_LOG_INFO(info);
        }
        if(nbDevices < 1)
        {
            AVDM_LOG_ERROR("This program needs a HIP-enabled GPU (gfx950).");
            return EXIT_FAILURE;
        }
    }
    if(downscale < 1)
    {
        AVDM_LOG_ERROR("Invalid value for downscale parameter. Should be at least 1.");
        return EXIT_FAILURE;
    }
    if(depthMapParams.useRefine && sgmParams.scale != -1 && sgmParams.stepXY != -1)
    {
        const int sgmScaleStep = sgmParams.scale * sgmParams.stepXY;
        const int refineScaleStep = refineParams.scale * refineParams.stepXY;
        if(sgmScaleStep < refineScaleStep)
        {
            AVDM_LOG_ERROR("SGM downscale (scale x step) should be greater or equal to the Refine downscale (scale x step).");
            return EXIT_FAILURE;
        }
        if(sgmScaleStep % refineScaleStep != 0)
        {
            AVDM_LOG_ERROR("SGM downscale (scale x step) should be a multiple of the Refine downscale (scale x step).");
            return EXIT_FAILURE;
        }
    }
    // Mip levels.  The reference samples its images at the level log2(scale / min(sgmScale, refineScale)) with a mip-linear texture
    // (deviceMipmappedArray.cu:348, DeviceMipmapImage::getLevel).  Scales that are a power-of-two multiple of each other (every combination
    // Meshroom's defaults produce) give integral levels and run the LDS-staged kernels; any other combination (e.g. --sgmScale 3
    // --refineScale 1) gives a FRACTIONAL level: the stage's similarity volume then comes from the plain trilinear kernel (every tap blends
    // two levels through the software texture unit, ~10 x slower) and the map kernels blend two levels.  Said here once, not per tile.
    if(sgmParams.scale > 0 && refineParams.scale > 0)
    {
        const int lo = std::min(sgmParams.scale, refineParams.scale), hi = std::max(sgmParams.scale, refineParams.scale);
        const int ratio = (hi % lo == 0) ? hi / lo : 0;
        if(ratio == 0 || (ratio & (ratio - 1)) != 0)
            AVDM_LOG_WARNING("sgmScale (" << sgmParams.scale << ") and refineScale (" << refineParams.scale
                                          << ") are not a power-of-two multiple of each other: the coarser stage samples a fractional mip level "
                                             "(trilinear taps, the slow similarity kernel).");
    }
    // filtering axes: the reference maps every character through a table of {X, Y} (std::map::at throws on anything else) and runs two
    // paths per character; this implementation runs one or two axes.
    {
        const std::string& axes = sgmParams.filteringAxes;
        if(axes.empty() || axes.size() > 2 || axes.find_first_not_of("XY") != std::string::npos)
        {
            AVDM_LOG_ERROR("Invalid value for sgmFilteringAxes ('" << axes << "'): one or two characters out of 'X' and 'Y' (e.g. \"YX\").");
            return EXIT_FAILURE;
        }
    }
    if(minViewAngle < 0.f || minViewAngle > 360.f || maxViewAngle < 0.f || maxViewAngle > 360.f || minViewAngle > maxViewAngle)
    {
        AVDM_LOG_ERROR("Invalid value for minViewAngle/maxViewAngle parameter(s). Should be between 0 and 360.");
        return EXIT_FAILURE;
    }

    SfMData sfmData;
    try
    {
        loadSfMData(sfmData, sfmDataFilename);
    }
    catch(const std::exception& e)
    {
        AVDM_LOG_ERROR("The input SfMData file '" << sfmDataFilename << "' cannot be read (" << e.what() << ").");
        return EXIT_FAILURE;
    }
    if(!dirExists(outputFolder))
        ::mkdir(outputFolder.c_str(), 0755);

    MultiViewParams mp(sfmData, imagesFolder, outputFolder, downscale);
    mp.setMinViewAngle(minViewAngle);
    mp.setMaxViewAngle(maxViewAngle);

    if(tileParams.bufferWidth <= 0 || tileParams.bufferHeight <= 0)
    {
        tileParams.bufferWidth = mp.getMaxImageWidth();
        tileParams.bufferHeight = mp.getMaxImageHeight();
    }
    if(tileParams.padding < 0 && tileParams.padding * 2 < tileParams.bufferWidth && tileParams.padding * 2 < tileParams.bufferHeight)
    {
        AVDM_LOG_ERROR("Invalid value for tilePadding parameter. Should be at least 0 and not exceed half buffer width and height.");
        return EXIT_FAILURE;
    }
    if(tileParams.bufferWidth > mp.getMaxImageWidth() || tileParams.bufferHeight > mp.getMaxImageHeight())
        AVDM_LOG_WARNING("Tile buffer size (width: " << tileParams.bufferWidth << ", height: " << tileParams.bufferHeight
                                                     << ") is larger than the maximum image size (width: " << mp.getMaxImageWidth()
                                                     << ", height: " << mp.getMaxImageHeight() << ").");

    bool autoSgmScaleStep = false;
    if(sgmParams.scale == -1 || sgmParams.stepXY == -1)
    {
        const int fileScale = 1;
        const int maxSideXY = 700 / mp.getProcessDownscale();
        const int maxImageW = mp.getMaxImageWidth();
        const int maxImageH = mp.getMaxImageHeight();
        int maxW = maxSideXY;
        int maxH = int(maxSideXY * 0.8);
        if(maxImageW < maxImageH)
            std::swap(maxW, maxH);
        if(sgmParams.scale == -1)
        {
            const int scaleTmp = computeDownscale(mp, fileScale, maxW, maxH);
            sgmParams.scale = std::min(2, scaleTmp);
        }
        if(sgmParams.stepXY == -1)
            sgmParams.stepXY = computeDownscale(mp, fileScale * sgmParams.scale, maxW, maxH);
        autoSgmScaleStep = true;
    }

    if(depthMapParams.autoAdjustSmallImage && hasOnlyOneTile(tileParams, mp.getMaxImageWidth(), mp.getMaxImageHeight()))
    {
        if(sgmParams.maxTCamsPerTile < depthMapParams.maxTCams)
        {
            AVDM_LOG_WARNING("Single tile computation, override SGM maximum number of T cameras per tile (before: " << sgmParams.maxTCamsPerTile
                                                                                                                   << ", now: " << depthMapParams.maxTCams << ").");
            sgmParams.maxTCamsPerTile = depthMapParams.maxTCams;
        }
        if(refineParams.maxTCamsPerTile < depthMapParams.maxTCams)
        {
            AVDM_LOG_WARNING("Single tile computation, override Refine maximum number of T cameras per tile (before: " << refineParams.maxTCamsPerTile
                                                                                                                      << ", now: " << depthMapParams.maxTCams << ").");
            refineParams.maxTCamsPerTile = depthMapParams.maxTCams;
        }
        const int maxSgmBufferWidth = divideRoundUp(mp.getMaxImageWidth(), sgmParams.scale * sgmParams.stepXY);
        const int maxSgmBufferHeight = divideRoundUp(mp.getMaxImageHeight(), sgmParams.scale * sgmParams.stepXY);
        if(!autoSgmScaleStep && (sgmParams.stepXY == 2) && (maxSgmBufferWidth < tileParams.bufferWidth * 0.5) && (maxSgmBufferHeight < tileParams.bufferHeight * 0.5))
        {
            AVDM_LOG_WARNING("Single tile computation, override SGM step XY (before: " << sgmParams.stepXY << ", now: 1).");
            sgmParams.stepXY = 1;
        }
    }

    const int maxDownscale = std::max(sgmParams.scale * sgmParams.stepXY, refineParams.scale * refineParams.stepXY);
    if(tileParams.padding % maxDownscale != 0)
    {
        const int padding = divideRoundUp(tileParams.padding, maxDownscale) * maxDownscale;
        AVDM_LOG_WARNING("Override tiling padding parameter (before: " << tileParams.padding << ", now: " << padding << ").");
        tileParams.padding = padding;
    }

    std::vector<int> cams;
    cams.reserve(mp.ncams);
    if(rangeSize == -1)
    {
        for(int rc = 0; rc < mp.ncams; ++rc)
            cams.push_back(rc);
    }
    else
    {
        if(rangeStart < 0)
        {
            AVDM_LOG_ERROR("invalid subrange of cameras to process.");
            return EXIT_FAILURE;
        }
        for(int rc = rangeStart; rc < std::min(rangeStart + rangeSize, mp.ncams); ++rc)
            cams.push_back(rc);
        if(cams.empty())
        {
            AVDM_LOG_INFO("No camera to process.");
            return EXIT_SUCCESS;
        }
    }

    DepthMapEstimator depthMapEstimator(mp, tileParams, depthMapParams, sgmParams, refineParams);
    if(dryRun)
    {
        std::vector<TilePlan> plans;
        depthMapEstimator.plan(cams, plans);
        printPlanJson(mp, plans, sgmParams, refineParams, tileParams);
        return EXIT_SUCCESS;
    }
    computeOnMultiGPUs(cams, depthMapEstimator, nbGPUs);

    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - startTime).count();
    AVDM_LOG_INFO("Task done in (s): " << std::fixed << std::setprecision(6) << sec);
    // Every output file is written and closed and every device resource this program took has been given back (DepthMapEstimator.cpp): what is
    // left is the HIP runtime's own exit handlers (code objects, queues, signal pools: 0.14 s measured on MI355X, session r06_k) and the static
    // destructors — work whose only effect is to return to the system what the system takes back anyway when the process ends.
    // AVDM_HOST_EXIT=normal keeps the ordinary return path.
    {
        const char* e = getenv("AVDM_HOST_EXIT");
        if(!(e != nullptr && std::string(e) == "normal"))
        {
            std::cout.flush();
            std::cerr.flush();
            std::fflush(nullptr);
            std::_Exit(EXIT_SUCCESS);
        }
    }
    return EXIT_SUCCESS;
}

} // namespace

int main(int argc, char* argv[])
{
    // cmdline.hpp:29-44: errors between ==== banners, exit code 1
    try
    {
        return aliceVision_main(argc, argv);
    }
    catch(const std::exception& e)
    {
        std::cerr << "================================================================================" << std::endl
                  << "====================== Command line failed with an error =======================" << std::endl
                  << "================================================================================" << std::endl
                  << e.what() << std::endl
                  << "================================================================================" << std::endl
                  << std::endl;
        return EXIT_FAILURE;
    }
    catch(...)
    {
        std::cerr << "================================================================================" << std::endl
                  << "============== Command line failed with an unrecoginzed exception ==============" << std::endl
                  << "================================================================================" << std::endl
                  << std::endl;
        return EXIT_FAILURE;
    }
}
