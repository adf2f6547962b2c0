// aliceVision_prepareDenseScene — the step Meshroom runs right BEFORE depth-map estimation: same flags and flow as
// software/pipeline/main_prepareDenseScene.cpp:34-420 of the reference.  For every view with a pose and an intrinsic it writes
// <viewId>.exr into --output — the source image undistorted (camera::UndistortImage, on the GPU: avdm_image_undistort) when the camera has
// a distortion model, copied otherwise — with the camera in the image metadata (AliceVision:downscale / P / K / R / t, which
// aliceVision_depthMapEstimation prefers over the SfMData: mvsUtils/MultiViewParams.cpp:164-186) and, with --saveMatricesTxtFiles 1,
// <viewId>_P.txt / <viewId>_KRt.txt.
// Sources: OpenEXR, PNG, TIFF and JPEG (PNG / TIFF / JPEG become linear float RGBA on the device); masks (--masksFolders, PNG or TIFF).
// Exposure: AliceVision:EV / AliceVision:EVComp from the views' EXIF metadata, --evCorrection scales the colours (sfmData.cpp: ExposureSetting).
// Not built: output formats other than OpenEXR (asking for one is an error, not a silent no-op).
// The range is taken over the views in id order (the reference iterates its hash container's order: chunk MEMBERSHIP may differ, the union
// over all chunks does not).
#include "cmdline.hpp"
#include "device.hpp"
#include "exr.hpp"
#include "log.hpp"
#include "mvsData.hpp"
#include "sfmData.hpp"
#include "png.hpp"
#include "tiff.hpp"
#include "jpeg.hpp"

#include <avdm.h>
#include <hip/hip_runtime.h>
#include <omp.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <chrono>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <set>
#include <string>
#include <vector>

using namespace avdm_host;

namespace {

bool fileExists(const std::string& p)
{
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

// sfmDataIO::viewPathsFromFolders (sfmDataIO/viewIO.cpp): <folder>/<viewId>.<ext> or <folder>/<stem of the view's path>.<ext>
std::vector<std::string> viewPathsFromFolders(const View& view, const std::vector<std::string>& folders)
{
    std::string stem = view.path;
    const size_t slash = stem.find_last_of('/');
    if(slash != std::string::npos)
        stem = stem.substr(slash + 1);
    const size_t dot = stem.find_last_of('.');
    if(dot != std::string::npos)
        stem = stem.substr(0, dot);
    std::vector<std::string> out;
    for(const std::string& folder : folders)
        for(const std::string& base : {std::to_string(view.viewId), stem})
            for(const char* ext : {".exr", ".png", ".jpg", ".jpeg", ".tif", ".tiff", ".JPG", ".JPEG", ".PNG", ".TIF", ".TIFF"}) // the formats this build decodes
            {
                const std::string p = folder + "/" + base + ext;
                if(fileExists(p) && std::find(out.begin(), out.end(), p) == out.end())
                    out.push_back(p);
            }
    return out;
}

// image::tryLoadMask (image/io.cpp:1357-1386): <folder>/<viewId>.<ext>, else <folder>/<file name of the source image with that extension>; the
// mask is read as ONE 8-bit channel (readImageNoFloat, io.cpp:955-984: any other channel count is an error; 16-bit samples are converted the
// way OpenImageIO converts types: v / 65535 * 255, rounded).  PNG and TIFF masks are decoded here.
bool tryLoadMask(std::vector<unsigned char>& mask, int& mw, int& mh, const std::vector<std::string>& masksFolders, IndexT viewId, const std::string& srcImage,
                 const std::string& fileExtension)
{
    std::string ext = fileExtension;
    if(!ext.empty() && ext[0] != '.')
        ext = "." + ext;
    std::string name = srcImage;
    const size_t slash = name.find_last_of('/');
    if(slash != std::string::npos)
        name = name.substr(slash + 1);
    const size_t dot = name.find_last_of('.');
    if(dot != std::string::npos)
        name = name.substr(0, dot);
    for(const std::string& folder : masksFolders)
    {
        if(folder.empty())
            continue;
        for(const std::string& candidate : {folder + "/" + std::to_string(viewId) + ext, folder + "/" + name + ext})
        {
            if(!fileExists(candidate))
                continue;
            std::string e = ext;
            for(char& ch : e)
                ch = (char)std::tolower((unsigned char)ch);
            int channels = 0, bits = 0;
            std::vector<unsigned char> samples;
            if(e == ".png")
            {
                PngImage png;
                readPng(candidate, png);
                mw = png.width, mh = png.height, channels = png.channels, bits = png.bits;
                samples.swap(png.samples);
            }
            else if(e == ".tif" || e == ".tiff")
            {
                TiffImage tiff;
                readTiff(candidate, tiff);
                mw = tiff.width, mh = tiff.height, channels = tiff.channels, bits = tiff.bits;
                samples.swap(tiff.samples);
            }
            else
                throw std::runtime_error("mask '" + candidate + "': only .png and .tif masks are decoded by this build");
            if(channels != 1)
                throw std::runtime_error("Can't load channels of image file '" + candidate + "'.");
            mask.resize((size_t)mw * mh);
            if(bits == 8)
                mask = samples;
            else
                for(size_t i = 0; i < mask.size(); ++i)
                    mask[i] = (unsigned char)((float)reinterpret_cast<const uint16_t*>(samples.data())[i] / 65535.0f * 255.0f + 0.5f);
            return true;
        }
    }
    return false;
}

int distortionModelOf(const Intrinsic& I)
{
    if(I.distortionType == "none" || I.distortionParams.empty())
        return AVDM_DISTORTION_NONE;
    if(I.distortionType == "radialk1")
        return AVDM_DISTORTION_RADIALK1;
    if(I.distortionType == "radialk3")
        return AVDM_DISTORTION_RADIALK3;
    if(I.distortionType == "radialk3pt")
        return AVDM_DISTORTION_RADIALK3PT;
    return -1;
}

} // namespace

static int aliceVision_main(int argc, char* argv[])
{
    const auto startTime = std::chrono::steady_clock::now();
    std::string sfmDataFilename, outFolder, outImageFileTypeName = "exr", maskExtension = "png", verboseLevel = "info";
    std::vector<std::string> imagesFolders, masksFolders;
    int rangeStart = -1, rangeSize = 1; // main_prepareDenseScene.cpp:297-298
    bool saveMetadata = true, saveMatricesTxtFiles = false, evCorrection = false;
    int maxMemoryAvailable = 0, maxCoresAvailable = 0;

    CmdLine cmdline("AliceVision prepareDenseScene");
    cmdline.add("input", &sfmDataFilename, "SfMData file.", true, 'i');
    cmdline.add("output", &outFolder, "Output folder.", true, 'o');
    cmdline.addMultitoken("imagesFolders", &imagesFolders,
                     "Use images from specific folder(s) instead of those specify in the SfMData file.\nFilename should be the image uid.");
    cmdline.addMultitoken("masksFolders", &masksFolders, "Use masks from specific folder(s).\nFilename should be the same or the image uid.");
    cmdline.add("maskExtension", &maskExtension, "File extension of the masks to use.");
    cmdline.add("outputFileType", &outImageFileTypeName, "Output file type: exr.");
    cmdline.add("saveMetadata", &saveMetadata, "Save projections and intrinsics information in images metadata.");
    cmdline.add("saveMatricesTxtFiles", &saveMatricesTxtFiles, "Save projections and intrinsics information in text files.");
    cmdline.add("rangeStart", &rangeStart, "Range image index start.");
    cmdline.add("rangeSize", &rangeSize, "Range size.");
    cmdline.add("evCorrection", &evCorrection, "Correct exposure value.");
    cmdline.add("verboseLevel", &verboseLevel, "verbosity level (fatal, error, warning, info, debug, trace).", false, 'v');
    cmdline.add("maxMemoryAvailable", &maxMemoryAvailable, "User specified available RAM");
    cmdline.add("maxCoresAvailable", &maxCoresAvailable, "User specified available number of cores");

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
        omp_set_num_threads(32);
    if(outImageFileTypeName != "exr" && outImageFileTypeName != "EXR")
    {
        AVDM_LOG_ERROR("outputFileType '" << outImageFileTypeName << "' is not supported by this implementation: exr only.");
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

    // main_prepareDenseScene.cpp:394-412: the range is over the views in container order
    int rangeEnd = (int)sfmData.views.size();
    if(rangeStart != -1)
    {
        if(rangeStart < 0 || rangeSize < 0 || rangeStart > (int)sfmData.views.size())
        {
            AVDM_LOG_ERROR("Range is incorrect");
            return EXIT_FAILURE;
        }
        if(rangeStart + rangeSize > (int)sfmData.views.size())
            rangeSize = (int)sfmData.views.size() - rangeStart;
        rangeEnd = rangeStart + rangeSize;
        if(rangeSize <= 0)
        {
            AVDM_LOG_WARNING("Nothing to compute.");
            return EXIT_SUCCESS;
        }
    }
    else
        rangeStart = 0;
    ::mkdir(outFolder.c_str(), 0777);

    std::vector<const View*> todo;
    {
        int i = 0;
        for(const auto& kv : sfmData.views)
        {
            if(i >= rangeStart && i < rangeEnd && sfmData.isPoseAndIntrinsicDefined(kv.second))
                todo.push_back(&kv.second);
            ++i;
        }
    }
    if(avdm_device_count() < 1)
    {
        AVDM_LOG_ERROR("This program needs a HIP-enabled GPU (gfx950).");
        return EXIT_FAILURE;
    }
    AVDM_HIP_CHECK(hipSetDevice(0));
    hipStream_t stream = nullptr;
    AVDM_HIP_CHECK(hipStreamCreate(&stream));
    AVDM_LOG_INFO("Exporting Scene Undistorted Images: " << todo.size() << " view(s).");

    int nbUndistorted = 0;
    // for the exposure metadata / correction (main_prepareDenseScene.cpp:127-129)
    const double medianCameraExposure = sfmData.medianCameraExposure();
    AVDM_LOG_INFO("Median Camera Exposure: " << medianCameraExposure << ", Median EV: " << std::log2(1.0 / medianCameraExposure));
    try
    {
        for(const View* view : todo)
        {
            const Intrinsic& intr = sfmData.getIntrinsic(*view);
            const Pose pose = sfmData.getPose(*view);
            const std::string baseFilename = std::to_string(view->viewId);
            if(!intr.isPinhole)
            {
                AVDM_LOG_ERROR("Camera is not pinhole in filter");
                continue;
            }
            // source image: from --imagesFolders if given, else the path of the view
            std::string srcImage = view->path;
            if(!imagesFolders.empty())
            {
                const std::vector<std::string> paths = viewPathsFromFolders(*view, imagesFolders);
                if(paths.empty())
                    throw std::runtime_error("Cannot find view '" + baseFilename + "' image file in given folder(s)");
                if(paths.size() > 1)
                    throw std::runtime_error("Ambiguous case: Multiple source image files found in given folder(s) for the view '" + baseFilename + "'.");
                srcImage = paths.front();
            }
            // image::readImage(srcImage, image, LINEAR) (main_prepareDenseScene.cpp:55): OpenEXR is decoded on the host; a PNG or a JPEG
            // leaves the host as integer samples / DCT coefficients and becomes linear float RGBA on the device
            ExrImage exr; // (size and metadata of the source; pixels only for an OpenEXR source)
            std::vector<float> rgba;
            DeviceBuffer deviceSource;
            std::string ext = srcImage.rfind('.') == std::string::npos ? "" : srcImage.substr(srcImage.rfind('.'));
            for(char& ch : ext)
                ch = (char)std::tolower((unsigned char)ch);
            if(ext == ".jpg" || ext == ".jpeg")
            {
                JpegImage jpeg;
                readJpeg(srcImage, jpeg);
                exr.width = jpeg.width, exr.height = jpeg.height;
                deviceSource.allocate((size_t)jpeg.width * jpeg.height * 16);
                decodeJpegToLinearRgba(jpeg, deviceSource.as<float>(), stream);
            }
            else if(ext == ".png" || ext == ".tif" || ext == ".tiff")
            {
                int w = 0, h = 0, channels = 0, bits = 0;
                std::vector<unsigned char> samplesHost;
                if(ext == ".png")
                {
                    PngImage png;
                    readPng(srcImage, png);
                    w = png.width, h = png.height, channels = png.channels, bits = png.bits;
                    samplesHost.swap(png.samples);
                }
                else
                {
                    TiffImage tiff;
                    readTiff(srcImage, tiff);
                    w = tiff.width, h = tiff.height, channels = tiff.channels, bits = tiff.bits;
                    samplesHost.swap(tiff.samples);
                }
                exr.width = w, exr.height = h;
                DeviceBuffer samples(samplesHost.size());
                deviceSource.allocate((size_t)w * h * 16);
                AVDM_HIP_CHECK(hipMemcpyAsync(samples.ptr(), samplesHost.data(), samplesHost.size(), hipMemcpyHostToDevice, stream));
                avdmCheck(avdm_image_decode_integer(deviceSource.as<float>(), w * 16, samples.ptr(), w * channels * (bits / 8), w, h, channels, bits, 1, stream),
                          "avdm_image_decode_integer");
                AVDM_HIP_CHECK(hipStreamSynchronize(stream));
            }
            else
                readExr(srcImage, exr);
            if(exr.width != intr.width || exr.height != intr.height)
                throw std::runtime_error("image '" + srcImage + "' is " + std::to_string(exr.width) + "x" + std::to_string(exr.height) + ", its intrinsic " +
                                         std::to_string(intr.width) + "x" + std::to_string(intr.height));
            const size_t n = (size_t)exr.width * exr.height;
            rgba.resize(n * 4);
            if(deviceSource.bytes() == 0)
            {
                const int iR = exr.channelIndex("R"), iG = exr.channelIndex("G"), iB = exr.channelIndex("B"), iA = exr.channelIndex("A"), iY = exr.channelIndex("Y");
                if(!((iR >= 0 && iG >= 0 && iB >= 0) || iY >= 0))
                    throw std::runtime_error("image '" + srcImage + "' has neither R,G,B nor Y channels");
                const float* r = exr.channels[iR >= 0 ? iR : iY].data();
                const float* g = exr.channels[iG >= 0 ? iG : iY].data();
                const float* b = exr.channels[iB >= 0 ? iB : iY].data();
                const float* a = iA >= 0 ? exr.channels[iA].data() : nullptr;
#pragma omp parallel for
                for(long long i = 0; i < (long long)n; ++i)
                {
                    rgba[4 * i + 0] = r[i];
                    rgba[4 * i + 1] = g[i];
                    rgba[4 * i + 2] = b[i];
                    rgba[4 * i + 3] = a ? a[i] : 1.0f;
                }
            }

            // exposure (main_prepareDenseScene.cpp:241-252, 58-66): EV and the compensation towards the scene's median exposure go into the
            // metadata; with --evCorrection the colours are scaled by the compensation (before the mask and the undistortion)
            const double cameraExposure = cameraExposureSetting(view->metadata).getExposure();
            const double ev = std::log2(1.0 / cameraExposure);
            const float exposureCompensation = float(medianCameraExposure / cameraExposure);
            if(evCorrection)
            {
                AVDM_LOG_INFO("image " << view->viewId << ", exposure: " << cameraExposure << ", Ev " << ev << " Ev compensation: " << exposureCompensation);
                if(deviceSource.bytes())
                { // (like the mask: a device-decoded source comes back for this optional step)
                    AVDM_HIP_CHECK(hipMemcpyAsync(rgba.data(), deviceSource.ptr(), n * 16, hipMemcpyDeviceToHost, stream));
                    AVDM_HIP_CHECK(hipStreamSynchronize(stream));
                    deviceSource.release();
                }
#pragma omp parallel for
                for(long long i = 0; i < (long long)n; ++i)
                {
                    rgba[4 * i + 0] *= exposureCompensation;
                    rgba[4 * i + 1] *= exposureCompensation;
                    rgba[4 * i + 2] *= exposureCompensation;
                }
            }

            // mask (main_prepareDenseScene.cpp:68-69, 255-273): BEFORE the undistortion, alpha = 0 where the mask is 0 and 1 elsewhere
            {
                std::vector<unsigned char> mask;
                int mw = 0, mh = 0;
                if(!masksFolders.empty() && tryLoadMask(mask, mw, mh, masksFolders, view->viewId, srcImage, maskExtension))
                {
                    if((size_t)mw * mh != n)
                        AVDM_LOG_WARNING("Invalid image mask size: mask is ignored.");
                    else
                    {
                        if(deviceSource.bytes())
                        { // (a source decoded on the device comes back for the mask; masks are the exception, not the hot path)
                            AVDM_HIP_CHECK(hipMemcpyAsync(rgba.data(), deviceSource.ptr(), n * 16, hipMemcpyDeviceToHost, stream));
                            AVDM_HIP_CHECK(hipStreamSynchronize(stream));
                            deviceSource.release();
                        }
#pragma omp parallel for
                        for(long long i = 0; i < (long long)n; ++i)
                            rgba[4 * i + 3] = mask[(size_t)i] == 0 ? 0.f : 1.f;
                    }
                }
            }

            // undistort (main_prepareDenseScene.cpp:71-83): on the device when the camera has a distortion model
            const int model = distortionModelOf(intr);
            if(model < 0)
                throw std::runtime_error("distortion model '" + intr.distortionType + "' of intrinsic " + std::to_string(intr.intrinsicId) +
                                         " is not supported (none, radialk1, radialk3, radialk3pt)");
            // cam->isValid() && cam->hasDistortion() (main_prepareDenseScene.cpp:72; Pinhole::isValid, camera/Pinhole.hpp:56: focal lengths > 0 and a
            // non-empty image): an invalid camera's image is copied through
            const bool camValid = intr.scaleX > 0.0 && intr.scaleY > 0.0 && intr.width != 0 && intr.height != 0;
            if(camValid && model != AVDM_DISTORTION_NONE)
            {
                avdm_intrinsic_t cam{};
                cam.width = intr.width;
                cam.height = intr.height;
                cam.scale_x = intr.scaleX;
                cam.scale_y = intr.scaleY;
                cam.offset_x = intr.offsetX;
                cam.offset_y = intr.offsetY;
                cam.distortion_model = model;
                for(int k = 0; k < 3; ++k)
                    cam.k[k] = k < (int)intr.distortionParams.size() ? intr.distortionParams[k] : 0.0;
                DeviceBuffer src, dst(n * 16);
                if(deviceSource.bytes() == 0)
                {
                    src.allocate(n * 16);
                    AVDM_HIP_CHECK(hipMemcpyAsync(src.ptr(), rgba.data(), n * 16, hipMemcpyHostToDevice, stream));
                }
                const float* source = deviceSource.bytes() ? deviceSource.as<float>() : src.as<float>();
                const float fill[4] = {0.f, 0.f, 0.f, 0.f}; // Pix::Zero()
                avdmCheck(avdm_image_undistort(dst.as<float>(), exr.width * 16, source, exr.width * 16, &cam, fill, stream), "avdm_image_undistort");
                AVDM_HIP_CHECK(hipMemcpyAsync(rgba.data(), dst.ptr(), n * 16, hipMemcpyDeviceToHost, stream));
                AVDM_HIP_CHECK(hipStreamSynchronize(stream));
                ++nbUndistorted;
            }
            else if(deviceSource.bytes())
            {
                AVDM_HIP_CHECK(hipMemcpyAsync(rgba.data(), deviceSource.ptr(), n * 16, hipMemcpyDeviceToHost, stream));
                AVDM_HIP_CHECK(hipStreamSynchronize(stream));
            }

            // camera: Pinhole::getProjectiveEquivalent(pose) = K [R | t], t = -R C (camera/Pinhole.cpp:277-283)
            const Matrix3x3 K = intr.K();
            const Matrix3x3& R = pose.rotation;
            const Point3d t = (pose.rotation * pose.center) * -1.0;
            const Matrix3x4 P = composeP(K, R, t);
            ExrAttributes metadata = exr.attributes; // the source image's own metadata travel with it (main_prepareDenseScene.cpp:146-147)
            // the exposure values are written whether or not --saveMetadata is set (main_prepareDenseScene.cpp:241-246)
            metadata.setFloat("AliceVision:EV", float(ev));
            metadata.setFloat("AliceVision:EVComp", exposureCompensation);
            if(saveMetadata)
            {
                double vP[16] = {P(0, 0), P(0, 1), P(0, 2), P(0, 3), P(1, 0), P(1, 1), P(1, 2), P(1, 3), P(2, 0), P(2, 1), P(2, 2), P(2, 3), 0, 0, 0, 1};
                double vK[9], vR[9], vt[3] = {t.x, t.y, t.z};
                for(int r = 0; r < 3; ++r)
                    for(int c = 0; c < 3; ++c)
                    {
                        vK[3 * r + c] = K(r, c);
                        vR[3 * r + c] = R(r, c);
                    }
                metadata.setInt("AliceVision:downscale", 1);
                metadata.setM44d("AliceVision:P", vP);
                metadata.setM33d("AliceVision:K", vK);
                metadata.setM33d("AliceVision:R", vR);
                metadata.setV3d("AliceVision:t", vt);
            }
            if(saveMatricesTxtFiles)
            {
                std::ofstream fileP(outFolder + "/" + baseFilename + "_P.txt");
                fileP << std::setprecision(10) << P(0, 0) << " " << P(0, 1) << " " << P(0, 2) << " " << P(0, 3) << "\n"
                      << P(1, 0) << " " << P(1, 1) << " " << P(1, 2) << " " << P(1, 3) << "\n"
                      << P(2, 0) << " " << P(2, 1) << " " << P(2, 2) << " " << P(2, 3) << "\n";
                std::ofstream fileKRt(outFolder + "/" + baseFilename + "_KRt.txt");
                fileKRt << std::setprecision(10);
                for(int r = 0; r < 3; ++r)
                    fileKRt << K(r, 0) << " " << K(r, 1) << " " << K(r, 2) << "\n";
                fileKRt << "\n";
                for(int r = 0; r < 3; ++r)
                    fileKRt << R(r, 0) << " " << R(r, 1) << " " << R(r, 2) << "\n";
                fileKRt << "\n" << t.x << " " << t.y << " " << t.z << "\n";
            }

            // <viewId>.exr, float RGBA
            std::vector<std::vector<float>> planes(4, std::vector<float>(n));
#pragma omp parallel for
            for(long long i = 0; i < (long long)n; ++i)
                for(int c = 0; c < 4; ++c)
                    planes[c][i] = rgba[4 * i + c];
            const std::vector<ExrChannelIn> channels = {{"A", planes[3].data()}, {"B", planes[2].data()}, {"G", planes[1].data()}, {"R", planes[0].data()}};
            writeExr(outFolder + "/" + baseFilename + ".exr", exr.width, exr.height, channels, false, metadata, 0, 0, exr.width, exr.height);
        }
    }
    catch(const std::exception& e)
    {
        AVDM_LOG_ERROR(e.what());
        (void)hipStreamDestroy(stream);
        return EXIT_FAILURE;
    }
    (void)hipStreamDestroy(stream);
    AVDM_LOG_INFO(todo.size() << " view(s) exported, " << nbUndistorted << " undistorted on the device.");
    AVDM_LOG_INFO("Task done in (s): " << std::chrono::duration<double>(std::chrono::steady_clock::now() - startTime).count());
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
        std::cerr << "================================================================================\n"
                  << "====================== Command line failed with an error =======================\n"
                  << e.what() << "\n"
                  << "================================================================================" << std::endl;
        return EXIT_FAILURE;
    }
}
