// aliceVision_depthMapFiltering — the step Meshroom runs right after depth-map estimation: same flags, defaults
// and flow as software/pipeline/main_depthMapFiltering.cpp:33-160 of the reference.  Reads <viewId>_depthMap.exr / _simMap.exr from
// --depthMapsFolder, writes <viewId>_nmodMap.png, <viewId>_depthMap.exr and <viewId>_simMap.exr (filtered) and, with
// --computeNormalMaps 1, <viewId>_normalMap.exr into --output.  The per-pixel work runs on the GPU (include/avdm_fuse.h, avdm.h); there
// is no CPU path.
#include "Fuser.hpp"
#include "MultiViewParams.hpp"
#include "NormalMapEstimator.hpp"
#include "cmdline.hpp"
#include "log.hpp"
#include "sfmData.hpp"

#include <avdm.h>
#include <omp.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

using namespace avdm_host;

static int aliceVision_main(int argc, char* argv[])
{
    const auto startTime = std::chrono::steady_clock::now();
    std::string sfmDataFilename, depthMapsFolder, outputFolder, verboseLevel = "info";
    int rangeStart = -1, rangeSize = -1;
    float minViewAngle = 2.0f, maxViewAngle = 70.0f;
    int minNumOfConsistentCams = 3;
    int minNumOfConsistentCamsWithLowSimilarity = 4;
    float pixToleranceFactor = 2.0f;
    int pixSizeBall = 0;
    int pixSizeBallWithLowSimilarity = 0;
    int nNearestCams = 10;
    bool computeNormalMaps = false;
    int maxMemoryAvailable = 0, maxCoresAvailable = 0;
    int nbGPUs = 0; // hidden: the reference hard-codes 0 = all devices for the normal maps (:146)
    bool twoPasses = false; // hidden: run the two filtering passes one after the other over all cameras, like the reference

    CmdLine cmdline("This program filters depth maps to remove values that are not consistent with other depth maps.\n"
                    "AliceVision depthMapFiltering");
    cmdline.add("input", &sfmDataFilename, "SfMData file.", true, 'i');
    cmdline.add("depthMapsFolder", &depthMapsFolder, "Input depth map folder.", true);
    cmdline.add("output", &outputFolder, "Output folder for filtered depth maps.", true, 'o');
    cmdline.add("rangeStart", &rangeStart, "Compute only a sub-range of images from index rangeStart to rangeStart+rangeSize.");
    cmdline.add("rangeSize", &rangeSize, "Compute only a sub-range of N images (N=rangeSize).");
    cmdline.add("minViewAngle", &minViewAngle, "Minimum angle between two views.");
    cmdline.add("maxViewAngle", &maxViewAngle, "Maximum angle between two views.");
    cmdline.add("minNumOfConsistentCams", &minNumOfConsistentCams, "Minimal number of consistent cameras to consider the pixel.");
    cmdline.add("minNumOfConsistentCamsWithLowSimilarity", &minNumOfConsistentCamsWithLowSimilarity,
                "Minimal number of consistent cameras to consider the pixel when the similarity is weak or ambiguous.");
    cmdline.add("pixToleranceFactor", &pixToleranceFactor, "Filtering tolerance size factor (in px).");
    cmdline.add("pixSizeBall", &pixSizeBall, "Filter ball size (in px).");
    cmdline.add("pixSizeBallWithLowSimilarity", &pixSizeBallWithLowSimilarity, "Filter ball size (in px) when the similarity is weak or ambiguous.");
    cmdline.add("nNearestCams", &nNearestCams, "Number of nearest cameras.");
    cmdline.add("computeNormalMaps", &computeNormalMaps, "Compute normal maps per depth map.");
    cmdline.add("verboseLevel", &verboseLevel, "verbosity level (fatal, error, warning, info, debug, trace).", false, 'v');
    cmdline.add("maxMemoryAvailable", &maxMemoryAvailable, "User specified available RAM");
    cmdline.add("maxCoresAvailable", &maxCoresAvailable, "User specified available number of cores");
    cmdline.add("nbGPUs", &nbGPUs, "", false, 0, true);
    cmdline.add("twoPasses", &twoPasses, "", false, 0, true);

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
        omp_set_num_threads(32); // file decode / encode only: see main_depthMapEstimation.cpp

    if(pixSizeBall < 0 || pixSizeBallWithLowSimilarity < 0)
    {
        AVDM_LOG_ERROR("Invalid value for pixSizeBall / pixSizeBallWithLowSimilarity: must not be negative.");
        return EXIT_FAILURE;
    }

    // read the input SfM scene
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
    {
        struct stat st;
        if(::stat(outputFolder.c_str(), &st) != 0)
            ::mkdir(outputFolder.c_str(), 0755);
    }

    // initialization (main_depthMapFiltering.cpp:103-107): cameras and sizes come from the depth maps' own metadata
    MultiViewParams mp(sfmData, "", depthMapsFolder, outputFolder, EFileType::depthMap);
    mp.setMinViewAngle(minViewAngle);
    mp.setMaxViewAngle(maxViewAngle);

    std::vector<int> cams;
    cams.reserve(mp.ncams);
    if(rangeSize == -1)
    {
        for(int rc = 0; rc < mp.ncams; rc++) // process all cameras
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

    if(avdm_device_count() <= 0)
    {
        AVDM_LOG_ERROR("This program needs a HIP-enabled GPU (gfx950).");
        return EXIT_FAILURE;
    }

    AVDM_LOG_INFO("Filter depth maps.");
    {
        // fs.filterGroups(...) then fs.filterDepthMaps(...) in the reference (:137-138); one pass here, same files (Fuser.hpp)
        Fuser fs(mp);
        if(twoPasses)
        {
            fs.filterGroups(cams, pixToleranceFactor, pixSizeBall, pixSizeBallWithLowSimilarity, nNearestCams);
            fs.filterDepthMaps(cams, minNumOfConsistentCams, minNumOfConsistentCamsWithLowSimilarity);
        }
        else
            fs.filterGroupsAndDepthMaps(cams, pixToleranceFactor, pixSizeBall, pixSizeBallWithLowSimilarity, nNearestCams, minNumOfConsistentCams,
                                        minNumOfConsistentCamsWithLowSimilarity);
    }

    if(computeNormalMaps)
    {
        NormalMapEstimator normalMapEstimator(mp);
        computeOnMultiGPUs(cams, normalMapEstimator, nbGPUs);
    }

    AVDM_LOG_INFO("Task done in (s): " << std::to_string(std::chrono::duration<double>(std::chrono::steady_clock::now() - startTime).count()));
    return EXIT_SUCCESS;
}

int main(int argc, char* argv[])
{
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
