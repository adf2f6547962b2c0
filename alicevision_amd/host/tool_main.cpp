// avdm_host_tool — small command-line probes of the host library for the CPU test-suite (no GPU needed):
//   tiles W H bufW bufH padding maxDownscale           print the tile ROI list (getTileRoiList)
//   exr-copy in.exr out.exr half(0|1)                   decode an EXR and write it back (reader + writer round trip)
//   exr-info in.exr                                     print size, windows, channels and attribute names / types
//   merge-ones sfm imagesFolder outFolder downscale bufW bufH padding scaleStep
//                                                       merge all-ones tiles of camera 0 with addTileMapWeighted and write the sum
//   quantile left|right p v0 v1 ...                     boost-style tail quantile used by SgmDepthList
//   png-copy in.png out.png                             decode an 8-bit PNG (first channel) and write it back as greyscale
//   fuse-cameras sfm depthMapsFolder filterFolder n     cameras as aliceVision_depthMapFiltering sees them (from the depth maps' metadata)
//                                                       and the n nearest cameras of each, as JSON with round-trip precision
//   sfm-dump scene.(sfm|json|abc)                      the loaded SfMData as JSON with round-trip precision (views, intrinsics, poses, landmarks)
//   tiff-dump in.tif out.raw                           decode a TIFF to its interleaved integer samples (host/tiff.cpp)
//   jpeg-dump in.jpg out.bin                           entropy-decode a JPEG (host/jpeg.cpp) and dump geometry, tables and coefficients
//   jet v0 v1 ...                                      the reference's jet colour map (debug volume exports) at the given values
//   exposures scene.(sfm|abc)                          every view's exposure setting from its metadata and the scene's median exposure
//   sfm-to-abc scene.(sfm|json|abc) out.abc            write the loaded SfMData as an Alembic archive (AlembicExporter's layout)
#include "DepthMapEstimator.hpp"
#include "MultiViewParams.hpp"
#include "depthMapUtils.hpp"
#include "exr.hpp"
#include "log.hpp"
#include "params.hpp"
#include "png.hpp"
#include "sfmData.hpp"
#include "alembic.hpp"
#include "jpeg.hpp"
#include "tiff.hpp"

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <iterator>
#include <fstream>
#include <string>

using namespace avdm_host;

static int usage()
{
    std::cerr << "usage: avdm_host_tool params|tiles|exr-copy|exr-info|merge-ones ..." << std::endl;
    return 2;
}

int main(int argc, char** argv)
{
    try
    {
        if(argc < 2)
            return usage();
        const std::string cmd = argv[1];
        Logger::setLevel("error");
        if(cmd == "params" && argc == 2)
        {
            // the default value of every parameter of the stage as this host holds it ("group.name=value"): compared with the reference's
            // own headers by tests/test_host_ref.py
            const SgmParams sgm;
            const RefineParams refine;
            const DepthMapParams dm;
            const TileParams tile;
            std::ostream& os = std::cout;
            os.precision(17);
            os << "sgm.scale=" << sgm.scale << "\n";
            os << "sgm.stepXY=" << sgm.stepXY << "\n";
            os << "sgm.stepZ=" << sgm.stepZ << "\n";
            os << "sgm.wsh=" << sgm.wsh << "\n";
            os << "sgm.maxDepths=" << sgm.maxDepths << "\n";
            os << "sgm.maxTCamsPerTile=" << sgm.maxTCamsPerTile << "\n";
            os << "sgm.seedsRangeInflate=" << sgm.seedsRangeInflate << "\n";
            os << "sgm.depthThicknessInflate=" << sgm.depthThicknessInflate << "\n";
            os << "sgm.maxSimilarity=" << sgm.maxSimilarity << "\n";
            os << "sgm.gammaC=" << sgm.gammaC << "\n";
            os << "sgm.gammaP=" << sgm.gammaP << "\n";
            os << "sgm.p1=" << sgm.p1 << "\n";
            os << "sgm.p2Weighting=" << sgm.p2Weighting << "\n";
            os << "sgm.filteringAxes=" << sgm.filteringAxes << "\n";
            os << "sgm.useSfmSeeds=" << sgm.useSfmSeeds << "\n";
            os << "sgm.depthListPerTile=" << sgm.depthListPerTile << "\n";
            os << "sgm.useConsistentScale=" << sgm.useConsistentScale << "\n";
            os << "sgm.useCustomPatchPattern=" << sgm.useCustomPatchPattern << "\n";
            os << "sgm.exportIntermediateDepthSimMaps=" << sgm.exportIntermediateDepthSimMaps << "\n";
            os << "sgm.exportIntermediateNormalMaps=" << sgm.exportIntermediateNormalMaps << "\n";
            os << "sgm.exportIntermediateVolumes=" << sgm.exportIntermediateVolumes << "\n";
            os << "sgm.exportIntermediateCrossVolumes=" << sgm.exportIntermediateCrossVolumes << "\n";
            os << "sgm.exportIntermediateTopographicCutVolumes=" << sgm.exportIntermediateTopographicCutVolumes << "\n";
            os << "sgm.exportIntermediateVolume9pCsv=" << sgm.exportIntermediateVolume9pCsv << "\n";
            os << "sgm.exportDepthsTxtFiles=" << sgm.exportDepthsTxtFiles << "\n";
            os << "sgm.updateUninitializedSim=" << sgm.updateUninitializedSim << "\n";
            os << "sgm.prematchingMaxDepthScale=" << sgm.prematchingMaxDepthScale << "\n";
            os << "sgm.seedsRangePercentile=" << sgm.seedsRangePercentile << "\n";
            os << "sgm.doSgmOptimizeVolume=" << sgm.doSgmOptimizeVolume << "\n";
            os << "refine.scale=" << refine.scale << "\n";
            os << "refine.stepXY=" << refine.stepXY << "\n";
            os << "refine.wsh=" << refine.wsh << "\n";
            os << "refine.halfNbDepths=" << refine.halfNbDepths << "\n";
            os << "refine.nbSubsamples=" << refine.nbSubsamples << "\n";
            os << "refine.maxTCamsPerTile=" << refine.maxTCamsPerTile << "\n";
            os << "refine.optimizationNbIterations=" << refine.optimizationNbIterations << "\n";
            os << "refine.sigma=" << refine.sigma << "\n";
            os << "refine.gammaC=" << refine.gammaC << "\n";
            os << "refine.gammaP=" << refine.gammaP << "\n";
            os << "refine.interpolateMiddleDepth=" << refine.interpolateMiddleDepth << "\n";
            os << "refine.useConsistentScale=" << refine.useConsistentScale << "\n";
            os << "refine.useCustomPatchPattern=" << refine.useCustomPatchPattern << "\n";
            os << "refine.useRefineFuse=" << refine.useRefineFuse << "\n";
            os << "refine.useColorOptimization=" << refine.useColorOptimization << "\n";
            os << "refine.exportIntermediateDepthSimMaps=" << refine.exportIntermediateDepthSimMaps << "\n";
            os << "refine.exportIntermediateNormalMaps=" << refine.exportIntermediateNormalMaps << "\n";
            os << "refine.exportIntermediateCrossVolumes=" << refine.exportIntermediateCrossVolumes << "\n";
            os << "refine.exportIntermediateTopographicCutVolumes=" << refine.exportIntermediateTopographicCutVolumes << "\n";
            os << "refine.exportIntermediateVolume9pCsv=" << refine.exportIntermediateVolume9pCsv << "\n";
            os << "refine.useSgmNormalMap=" << refine.useSgmNormalMap << "\n";
            os << "tile.bufferWidth=" << tile.bufferWidth << "\n";
            os << "tile.bufferHeight=" << tile.bufferHeight << "\n";
            os << "tile.padding=" << tile.padding << "\n";
            os << "depthMap.maxTCams=" << dm.maxTCams << "\n";
            os << "depthMap.chooseTCamsPerTile=" << dm.chooseTCamsPerTile << "\n";
            os << "depthMap.exportTilePattern=" << dm.exportTilePattern << "\n";
            os << "depthMap.autoAdjustSmallImage=" << dm.autoAdjustSmallImage << "\n";
            os << "depthMap.useRefine=" << dm.useRefine << "\n";
            return 0;
        }
        if(cmd == "tiles" && argc == 8)
        {
            TileParams tp;
            const int W = std::atoi(argv[2]), H = std::atoi(argv[3]);
            tp.bufferWidth = std::atoi(argv[4]);
            tp.bufferHeight = std::atoi(argv[5]);
            tp.padding = std::atoi(argv[6]);
            std::vector<ROI> rois;
            getTileRoiList(tp, W, H, std::atoi(argv[7]), rois);
            for(const ROI& r : rois)
                std::cout << r.x.begin << " " << r.x.end << " " << r.y.begin << " " << r.y.end << "\n";
            return 0;
        }
        if(cmd == "exr-copy" && argc == 5)
        {
            ExrImage img;
            readExr(argv[2], img);
            std::vector<ExrChannelIn> ch;
            for(size_t i = 0; i < img.channelNames.size(); ++i)
                ch.push_back({img.channelNames[i], img.channels[i].data()});
            writeExr(argv[3], img.width, img.height, ch, std::atoi(argv[4]) != 0, img.attributes, img.dataX0, img.dataY0, img.displayW, img.displayH);
            return 0;
        }
        if(cmd == "exr-lines-dump" && argc == 4)
        { // the scan lines as readExrLines hands them to the device: "<w> <h> <stride> <bytes> <mapped> <offR> <offG> <offB> <offA> <types...>" to stdout,
          // the line bytes to a raw file ("refused" when the layout is not taken)
            ExrLines x;
            if(!readExrLines(argv[2], x))
            {
                std::cout << "refused" << std::endl;
                return 0;
            }
            std::ofstream f(argv[3], std::ios::binary);
            f.write((const char*)x.lines, (std::streamsize)x.bytes);
            std::cout << x.width << " " << x.height << " " << x.lineStride << " " << x.bytes << " " << (x.mapBase != nullptr ? 1 : 0);
            for(int k = 0; k < 4; ++k)
                std::cout << " " << x.chanOffset[k];
            for(int k = 0; k < 4; ++k)
                std::cout << " " << x.chanType[k];
            std::cout << std::endl;
            return 0;
        }
        if(cmd == "png-dump" && argc == 4)
        { // decode a PNG, write "<w> <h> <channels> <bits>" to stdout and the samples (host byte order) to a raw file
            PngImage img;
            readPng(argv[2], img);
            std::cout << img.width << " " << img.height << " " << img.channels << " " << img.bits << "\n";
            std::ofstream f(argv[3], std::ios::binary);
            f.write((const char*)img.samples.data(), (std::streamsize)img.samples.size());
            return f ? 0 : 1;
        }
        if(cmd == "png-write" && argc == 8)
        { // raw samples (host byte order) -> PNG with every scan-line filter type in turn
            const int w = std::atoi(argv[4]), h = std::atoi(argv[5]), c = std::atoi(argv[6]), bits = std::atoi(argv[7]);
            std::ifstream f(argv[2], std::ios::binary);
            std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
            if(raw.size() != (size_t)w * h * c * (bits / 8))
                throw std::runtime_error("png-write: raw size does not match");
            writePng(argv[3], w, h, c, bits, raw.data());
            return 0;
        }
        if(cmd == "exr-info" && argc == 3)
        {
            ExrImage img;
            readExr(argv[2], img, true);
            std::cout << "size " << img.width << " " << img.height << "\norigin " << img.dataX0 << " " << img.dataY0 << "\ndisplay " << img.displayW << " "
                      << img.displayH << "\nchannels";
            for(const auto& n : img.channelNames)
                std::cout << " " << n;
            std::cout << "\n";
            for(const auto& a : img.attributes.list)
                std::cout << "attr " << a.name << " " << a.type << " " << a.data.size() << "\n";
            return 0;
        }
        if(cmd == "merge-ones" && argc == 10)
        {
            SfMData sfm;
            loadSfMData(sfm, argv[2]);
            MultiViewParams mp(sfm, argv[3], argv[4], std::atoi(argv[5]));
            TileParams tp;
            tp.bufferWidth = std::atoi(argv[6]);
            tp.bufferHeight = std::atoi(argv[7]);
            tp.padding = std::atoi(argv[8]);
            const int scaleStep = std::atoi(argv[9]);
            std::vector<ROI> rois;
            getTileRoiList(tp, mp.getMaxImageWidth(), mp.getMaxImageHeight(), scaleStep, rois);
            std::vector<Float2Tile> tiles(rois.size());
            const int tw = divideRoundUp(tp.bufferWidth, scaleStep), th = divideRoundUp(tp.bufferHeight, scaleStep);
            for(size_t i = 0; i < rois.size(); ++i)
            {
                tiles[i].allocate(rois.size() > 1 ? tw : divideRoundUp(mp.getWidth(0), scaleStep), rois.size() > 1 ? th : divideRoundUp(mp.getHeight(0), scaleStep));
                for(size_t k = 0; k < tiles[i].data.size(); k += 2)
                {
                    tiles[i].data[k] = 1.0f;       // "depth"
                    tiles[i].data[k + 1] = 0.5f;   // "sim"
                }
            }
            const auto t0 = std::chrono::steady_clock::now();
            writeDepthSimMapFromTileList(0, mp, tp, rois, tiles, scaleStep, 1);
            std::cout << "merge+write " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " s" << std::endl;
            return 0;
        }
        if(cmd == "png-copy" && argc == 4)
        {
            int w = 0, h = 0;
            std::vector<unsigned char> data;
            readPngGray8(argv[2], w, h, data);
            writePngGray8(argv[3], w, h, data.data());
            std::cout << w << " " << h << std::endl;
            return 0;
        }
        if(cmd == "fuse-cameras" && argc == 6)
        {
            SfMData sfm;
            loadSfMData(sfm, argv[2]);
            MultiViewParams mp(sfm, "", argv[3], argv[4], EFileType::depthMap);
            const int n = std::atoi(argv[5]);
            std::cout << std::setprecision(17) << "{\"cams\": [";
            for(int c = 0; c < mp.ncams; ++c)
            {
                std::cout << (c ? ", " : "") << "{\"viewId\": " << mp.getViewId(c) << ", \"width\": " << mp.getWidth(c) << ", \"height\": " << mp.getHeight(c)
                          << ", \"P\": [";
                for(int i = 0; i < 12; ++i)
                    std::cout << (i ? ", " : "") << mp.camArr[c].m[i];
                std::cout << "], \"iP\": [";
                for(int i = 0; i < 9; ++i)
                    std::cout << (i ? ", " : "") << mp.iCamArr[c].m[i];
                std::cout << "], \"C\": [" << mp.CArr[c].x << ", " << mp.CArr[c].y << ", " << mp.CArr[c].z << "], \"tcams\": [";
                const std::vector<int> t = mp.findNearestCamsFromLandmarks(c, n);
                for(size_t i = 0; i < t.size(); ++i)
                    std::cout << (i ? ", " : "") << t[i];
                std::cout << "]}";
            }
            std::cout << "]}" << std::endl;
            return 0;
        }
        if(cmd == "tiff-dump" && argc == 4)
        { // decode a TIFF to its integer samples: prints "width height channels bits orientation", writes the samples (host byte order)
            TiffImage t;
            readTiff(argv[2], t);
            std::ofstream f(argv[3], std::ios::binary);
            f.write(reinterpret_cast<const char*>(t.samples.data()), (std::streamsize)t.samples.size());
            std::cout << t.width << " " << t.height << " " << t.channels << " " << t.bits << " " << t.orientation << std::endl;
            return 0;
        }
        if(cmd == "jpeg-dump" && argc == 4)
        { // the entropy-decoded image: int32 header { width, height, components, hmax, vmax, storedAsRgb, progressive, exifOrientation },
          // then per component int32 { h, v, blocksW, blocksH, width, height }, uint16 quant[64], int16 coefficients
            JpegImage j;
            readJpeg(argv[2], j);
            std::ofstream f(argv[3], std::ios::binary);
            const int32_t hdr[8] = {j.width, j.height, (int32_t)j.components.size(), j.hmax, j.vmax, j.storedAsRgb() ? 1 : 0, j.progressive ? 1 : 0,
                                    j.exifOrientation};
            f.write(reinterpret_cast<const char*>(hdr), sizeof(hdr));
            for(const JpegComponent& c : j.components)
            {
                const int32_t ch[6] = {c.h, c.v, c.blocksW, c.blocksH, c.width, c.height};
                f.write(reinterpret_cast<const char*>(ch), sizeof(ch));
                f.write(reinterpret_cast<const char*>(j.quant[c.tq]), 128);
                f.write(reinterpret_cast<const char*>(c.coef.data()), (std::streamsize)(c.coef.size() * 2));
            }
            std::cout << j.width << " " << j.height << " " << j.components.size() << std::endl;
            return 0;
        }
        if(cmd == "jet" && argc >= 3)
        { // getRGBFromJetColorMap of every value given
            for(int i = 2; i < argc; ++i)
            {
                unsigned char c[3];
                jetColor((float)std::atof(argv[i]), c);
                std::cout << (int)c[0] << " " << (int)c[1] << " " << (int)c[2] << "\n";
            }
            return 0;
        }
        if(cmd == "exposure-of" && argc >= 5 && (argc - 2) % 3 == 0)
        { // ExposureSetting(shutter, fnumber, iso).getExposure() for every triple given
            std::cout << std::setprecision(17);
            for(int i = 2; i + 2 < argc; i += 3)
            {
                ExposureSetting e;
                e.shutter = std::atof(argv[i]), e.fnumber = std::atof(argv[i + 1]), e.iso = std::atof(argv[i + 2]);
                std::cout << e.getExposure() << " " << (e.isPartiallyDefined() ? 1 : 0) << "\n";
            }
            return 0;
        }
        if(cmd == "exposures" && argc == 3)
        { // per view: id, shutter, fnumber, iso, exposure; last line: the median exposure (ExposureSetting / getMedianCameraExposureSetting)
            SfMData sfm;
            loadSfMData(sfm, argv[2]);
            std::cout << std::setprecision(17);
            for(const auto& kv : sfm.views)
            {
                const ExposureSetting e = cameraExposureSetting(kv.second.metadata);
                std::cout << kv.first << " " << e.shutter << " " << e.fnumber << " " << e.iso << " " << e.getExposure() << "\n";
            }
            std::cout << "median " << sfm.medianCameraExposure() << std::endl;
            return 0;
        }
        if(cmd == "sfm-to-abc" && argc == 4)
        {
            SfMData sfm;
            loadSfMData(sfm, argv[2]);
            saveSfMDataAlembic(sfm, argv[3]);
            return 0;
        }
        if(cmd == "sfm-dump" && argc == 3)
        {
            SfMData sfm;
            loadSfMData(sfm, argv[2]);
            std::ostream& os = std::cout;
            os << std::setprecision(17) << "{\"views\": [";
            bool first = true;
            auto str = [](const std::string& t) {
                std::string o = "\"";
                for(char c : t)
                {
                    if(c == '"' || c == '\\')
                        o += '\\';
                    o += c;
                }
                return o + "\"";
            };
            for(const auto& kv : sfm.views)
            {
                const View& v = kv.second;
                os << (first ? "" : ", ") << "{\"viewId\": " << v.viewId << ", \"poseId\": " << v.poseId << ", \"intrinsicId\": " << v.intrinsicId
                   << ", \"path\": " << str(v.path) << ", \"width\": " << v.width << ", \"height\": " << v.height << ", \"metadata\": {";
                bool f2 = true;
                for(const auto& m : v.metadata)
                {
                    os << (f2 ? "" : ", ") << str(m.first) << ": " << str(m.second);
                    f2 = false;
                }
                os << "}, \"rigId\": " << (long long)(v.isPartOfRig() ? (long long)v.rigId : -1) << ", \"subPoseId\": "
                   << (long long)(v.isPartOfRig() ? (long long)v.subPoseId : -1) << ", \"independantPose\": " << (v.isPoseIndependant() ? 1 : 0);
                if(sfm.isPoseAndIntrinsicDefined(v))
                {
                    const Pose ap = sfm.getPose(v);
                    os << ", \"absRotation\": [";
                    for(int i = 0; i < 9; ++i)
                        os << (i ? ", " : "") << ap.rotation.m[i];
                    os << "], \"absCenter\": [" << ap.center.x << ", " << ap.center.y << ", " << ap.center.z << "]";
                }
                os << "}";
                first = false;
            }
            os << "], \"intrinsics\": [";
            first = true;
            for(const auto& kv : sfm.intrinsics)
            {
                const Intrinsic& I = kv.second;
                os << (first ? "" : ", ") << "{\"intrinsicId\": " << I.intrinsicId << ", \"type\": " << str(I.type) << ", \"distortionType\": "
                   << str(I.distortionType) << ", \"width\": " << I.width << ", \"height\": " << I.height << ", \"sensorWidth\": " << I.sensorWidth
                   << ", \"sensorHeight\": " << I.sensorHeight << ", \"scale\": [" << I.scaleX << ", " << I.scaleY << "], \"offset\": [" << I.offsetX
                   << ", " << I.offsetY << "], \"isPinhole\": " << (I.isPinhole ? 1 : 0) << ", \"distortionParams\": [";
                for(size_t i = 0; i < I.distortionParams.size(); ++i)
                    os << (i ? ", " : "") << I.distortionParams[i];
                os << "]}";
                first = false;
            }
            os << "], \"poses\": [";
            first = true;
            for(const auto& kv : sfm.poses)
            {
                os << (first ? "" : ", ") << "{\"poseId\": " << kv.first << ", \"rotation\": [";
                for(int i = 0; i < 9; ++i)
                    os << (i ? ", " : "") << kv.second.rotation.m[i];
                os << "], \"center\": [" << kv.second.center.x << ", " << kv.second.center.y << ", " << kv.second.center.z << "]}";
                first = false;
            }
            os << "], \"landmarks\": [";
            first = true;
            for(const auto& kv : sfm.landmarks)
            {
                os << (first ? "" : ", ") << "{\"id\": " << kv.first << ", \"X\": [" << kv.second.X.x << ", " << kv.second.X.y << ", " << kv.second.X.z
                   << "], \"rgb\": [" << (int)kv.second.rgb[0] << ", " << (int)kv.second.rgb[1] << ", " << (int)kv.second.rgb[2] << "], \"obs\": [";
                bool f2 = true;
                for(const auto& ob : kv.second.observations)
                {
                    os << (f2 ? "" : ", ") << "[" << ob.first << ", " << ob.second.x << ", " << ob.second.y << "]";
                    f2 = false;
                }
                os << "]}";
                first = false;
            }
            os << "]}" << std::endl;
            return 0;
        }
        return usage();
    }
    catch(const std::exception& e)
    {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
}
