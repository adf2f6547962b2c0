// ref_driver.cpp — C entry points over the REFERENCE's own kernel-launch layer compiled for the CPU (oracle/_ref/libavdm_ref.so).
//
// TEST INFRASTRUCTURE ONLY.  What is linked behind these entry points is the reference's code, unchanged, from where it lies under
// /root/reference/src/aliceVision/depthMap: the 19 `cuda_*` host wrappers (cuda/planeSweeping/deviceSimilarityVolume.cu,
// deviceDepthSimilarityMap.cu, cuda/imageProcessing/*.cu), every kernel and device helper they launch (cuda/planeSweeping/*.cuh,
// cuda/device/*.cuh), DeviceMipmapImage (cuda/host/DeviceMipmapImage.cpp), the pitched-memory classes (cuda/host/memory.hpp) and
// buildCustomPatchPattern (cuda/host/patchPattern.cpp).  Underneath sits the stand-in CUDA runtime of oracle/ref/shim (device =
// host memory, kernels = loops, texture unit restated from the CUDA programming guide).  This file only marshals: volumes arrive
// z-fastest (include/avdm.h) and are re-laid x-fastest into the reference's CudaDeviceMemoryPitched buffers, cameras are copied into
// the reference's constant-memory array, images are built by DeviceMipmapImage::fill from float RGBA.
//
// Used by tests/ to pin oracle/avdm_oracle.c (and through it the HIP kernels) to the reference's code, and by
// tests/golden/make_ref_golden.py to generate the committed vectors.  Nothing under alicevision_amd/ or include/ touches it.
#include <aliceVision/depthMap/cuda/planeSweeping/deviceDepthSimilarityMap.hpp>
#include <aliceVision/depthMap/cuda/planeSweeping/deviceSimilarityVolume.hpp>
#include <aliceVision/depthMap/cuda/imageProcessing/deviceColorConversion.hpp>
#include <aliceVision/depthMap/cuda/imageProcessing/deviceGaussianFilter.hpp>
#include <aliceVision/depthMap/cuda/imageProcessing/deviceMipmappedArray.hpp>
#include <aliceVision/depthMap/cuda/host/DeviceMipmapImage.hpp>
#include <aliceVision/depthMap/cuda/host/patchPattern.hpp>
#include <aliceVision/depthMap/cuda/device/DeviceCameraParams.hpp>
#include <aliceVision/depthMap/cuda/device/DevicePatchPattern.hpp>
#include <aliceVision/depthMap/cuda/device/color.cuh>
#include <aliceVision/depthMap/cuda/device/matrix.cuh>
#include <aliceVision/depthMap/cuda/device/SimStat.cuh>
#include <aliceVision/depthMap/cuda/device/eig33.cuh>

#include "avdm.h"

#include <string>
#include <vector>

// the constant-memory symbols (camera blocks, patch pattern) are the reference's own: cuda/device/DeviceCameraParams.cu, DevicePatchPattern.cu
using namespace aliceVision;
using namespace aliceVision::depthMap;

namespace {
bool g_gaussReady = false;
void ensure_gauss()
{
    if(!g_gaussReady)
    {
        cuda_createConstantGaussianArray(0, 8 /* DEVICE_MAX_DOWNSCALE, DeviceCache.hpp */);
        g_gaussReady = true;
    }
}
SgmParams to_ref(const avdm_sgm_params_t* p)
{
    SgmParams s;
    s.scale = p->scale;
    s.stepXY = p->stepXY;
    s.wsh = p->wsh;
    s.gammaC = p->gammaC;
    s.gammaP = p->gammaP;
    s.p1 = p->p1;
    s.p2Weighting = p->p2Weighting;
    s.maxSimilarity = p->maxSimilarity;
    s.depthThicknessInflate = p->depthThicknessInflate;
    s.filteringAxes = std::string(p->filteringAxes, strnlen(p->filteringAxes, sizeof(p->filteringAxes)));
    s.useConsistentScale = p->useConsistentScale != 0;
    s.useCustomPatchPattern = p->useCustomPatchPattern != 0;
    return s;
}
RefineParams to_ref(const avdm_refine_params_t* p)
{
    RefineParams r;
    r.scale = p->scale;
    r.stepXY = p->stepXY;
    r.wsh = p->wsh;
    r.halfNbDepths = p->halfNbDepths;
    r.nbSubsamples = p->nbSubsamples;
    r.optimizationNbIterations = p->optimizationNbIterations;
    r.sigma = p->sigma;
    r.gammaC = p->gammaC;
    r.gammaP = p->gammaP;
    r.interpolateMiddleDepth = p->interpolateMiddleDepth != 0;
    r.useConsistentScale = p->useConsistentScale != 0;
    r.useCustomPatchPattern = p->useCustomPatchPattern != 0;
    return r;
}
ROI to_ref(avdm_roi_t r) { return ROI(Range(r.x.begin, r.x.end), Range(r.y.begin, r.y.end)); }
Range to_ref(avdm_range_t r) { return Range(r.begin, r.end); }

// z-fastest host volume  <->  the reference's x-fastest pitched 3-D buffer
template <class TRef, class THost>
void vol_in(CudaDeviceMemoryPitched<TRef, 3>& dmp, const THost* v, long long pitch_y, int pitch_x, int X, int Y, int Z)
{
    const size_t p = dmp.getBytesPaddedUpToDim(0), s = dmp.getBytesPaddedUpToDim(1);
    for(int z = 0; z < Z; ++z)
        for(int y = 0; y < Y; ++y)
        {
            TRef* row = (TRef*)((char*)dmp.getBuffer() + z * s + y * p);
            for(int x = 0; x < X; ++x)
                memcpy(&row[x], (const char*)v + (long long)y * pitch_y + (long long)x * pitch_x + (long long)z * sizeof(THost), sizeof(THost));
        }
}
template <class TRef, class THost>
void vol_out(THost* v, long long pitch_y, int pitch_x, const CudaDeviceMemoryPitched<TRef, 3>& dmp, int X, int Y, int Z)
{
    const size_t p = dmp.getBytesPaddedUpToDim(0), s = dmp.getBytesPaddedUpToDim(1);
    for(int z = 0; z < Z; ++z)
        for(int y = 0; y < Y; ++y)
        {
            const TRef* row = (const TRef*)((const char*)dmp.getBuffer() + z * s + y * p);
            for(int x = 0; x < X; ++x)
                memcpy((char*)v + (long long)y * pitch_y + (long long)x * pitch_x + (long long)z * sizeof(THost), &row[x], sizeof(THost));
        }
}
template <class T>
void map_in(CudaDeviceMemoryPitched<T, 2>& dmp, const void* m, int pitch, int W, int H)
{
    for(int y = 0; y < H; ++y)
        memcpy((char*)dmp.getBuffer() + y * dmp.getPitch(), (const char*)m + (long long)y * pitch, (size_t)W * sizeof(T));
}
template <class T>
void map_out(void* m, int pitch, const CudaDeviceMemoryPitched<T, 2>& dmp, int W, int H)
{
    for(int y = 0; y < H; ++y)
        memcpy((char*)m + (long long)y * pitch, (const char*)dmp.getBuffer() + y * dmp.getPitch(), (size_t)W * sizeof(T));
}
}

extern "C" {

// 1 = linear-filter weights of the stand-in texture unit in 1.8 fixed point (AVDM_FILTER_CUDA_FIXED8), 0 = fp32 (AVDM_FILTER_EXACT)
void avr_set_filter_mode(int filter_mode) { shim::g_fixed8 = filter_mode == AVDM_FILTER_CUDA_FIXED8; }

// ---- cameras: constant-memory slots (DeviceCameraParams.hpp:16-34); avdm_camera_t has the same layout ----
int avr_camera_set(int slot, const avdm_camera_t* cam)
{
    static_assert(sizeof(avdm_camera_t) == sizeof(DeviceCameraParams), "camera block layouts differ");
    if(slot < 0 || slot >= ALICEVISION_DEVICE_MAX_CONSTANT_CAMERA_PARAM_SETS)
        return 1;
    memcpy(&constantCameraParametersArray_d[slot], cam, sizeof(DeviceCameraParams));
    return 0;
}

// ---- images: DeviceCache::addMipmapImage's conversion loop (cuda/host/DeviceCache.cpp:249-280: float RGBA * 255 -> half, the one
//      statement restated here) + DeviceMipmapImage::fill (the reference's) ----
void* avr_image_create(const float* rgba, int in_pitch, int width, int height, int min_downscale, int max_downscale)
{
    ensure_gauss();
    CudaHostMemoryHeap<CudaRGBA, 2> img_hmh(CudaSize<2>(width, height));
    for(int y = 0; y < height; ++y)
    {
        const float* s = (const float*)((const char*)rgba + (long long)y * in_pitch);
        for(int x = 0; x < width; ++x)
        {
            CudaRGBA& c = img_hmh(x, y);
            c.x = __float2half(s[4 * x + 0] * 255.0f);
            c.y = __float2half(s[4 * x + 1] * 255.0f);
            c.z = __float2half(s[4 * x + 2] * 255.0f);
            c.w = __float2half(s[4 * x + 3] * 255.0f);
        }
    }
    auto* img = new DeviceMipmapImage;
    img->fill(img_hmh, min_downscale, max_downscale);
    return img;
}
void avr_image_destroy(void* h) { delete(DeviceMipmapImage*)h; }
// texture probe: out[i] = tex2DLod<float4>(image texture, u, v, lod)
void avr_image_tex2dlod(void* h, const float* uvl, int n, float* out4)
{
    const cudaTextureObject_t t = ((DeviceMipmapImage*)h)->getTextureObject();
    for(int i = 0; i < n; ++i)
    {
        const float4 c = tex2DLod<float4>(t, uvl[3 * i], uvl[3 * i + 1], uvl[3 * i + 2]);
        out4[4 * i] = c.x; out4[4 * i + 1] = c.y; out4[4 * i + 2] = c.z; out4[4 * i + 3] = c.w;
    }
}
// texel (x, y) of mip level l = the texture sampled at the texel centre with point-exact coordinates (bilinear weights are 0 there)
void avr_image_read_level(void* h, int level, int w, int h_, float* out4)
{
    const cudaTextureObject_t t = ((DeviceMipmapImage*)h)->getTextureObject();
    const bool keep = shim::g_fixed8;
    shim::g_fixed8 = true; // fixed-point weights are exactly 0 at a texel centre; fp32 weights are only nearly so ((x + 0.5) / w * w)
    for(int y = 0; y < h_; ++y)
        for(int x = 0; x < w; ++x)
        {
            const float4 c = tex2DLod<float4>(t, (x + 0.5f) / float(w), (y + 0.5f) / float(h_), float(level));
            float* o = out4 + 4 * ((size_t)y * w + x);
            o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
        }
    shim::g_fixed8 = keep;
}
float avr_image_level(void* h, int downscale) { return ((DeviceMipmapImage*)h)->getLevel(downscale); }
void avr_image_dimensions(void* h, int downscale, int* w, int* h_)
{
    const CudaSize<2> d = ((DeviceMipmapImage*)h)->getDimensions(downscale);
    *w = (int)d.x();
    *h_ = (int)d.y();
}

// ---- custom patch pattern: the reference's builder (cuda/host/patchPattern.cpp:18-251) ----
int avr_build_custom_patch_pattern(int n, const avdm_patch_subpart_params_t* sub, int group, avdm_patch_pattern_t* out)
{
    CustomPatchPatternParams pp;
    pp.groupSubpartsPerLevel = group != 0;
    for(int i = 0; i < n; ++i)
        pp.subpartsParams.push_back({sub[i].isCircle != 0, sub[i].level, sub[i].nbCoordinates, sub[i].radius, sub[i].weight});
    try
    {
        buildCustomPatchPattern(pp);
    }
    catch(const std::exception&)
    {
        return 1;
    }
    if(out)
    {
        memset(out, 0, sizeof(*out));
        out->nbSubparts = constantPatchPattern_d.nbSubparts;
        for(int i = 0; i < ALICEVISION_DEVICE_PATCH_MAX_SUBPARTS; ++i)
        {
            const DevicePatchPatternSubpart& s = constantPatchPattern_d.subparts[i];
            avdm_patch_pattern_subpart_t& d = out->subparts[i];
            for(int c = 0; c < ALICEVISION_DEVICE_PATCH_MAX_COORDS_PER_SUBPARTS; ++c)
            {
                d.coordinates[c][0] = s.coordinates[c].x;
                d.coordinates[c][1] = s.coordinates[c].y;
            }
            d.nbCoordinates = s.nbCoordinates;
            d.level = s.level;
            d.downscale = s.downscale;
            d.weight = s.weight;
            d.isCircle = s.isCircle ? 1 : 0;
            d.wsh = s.wsh;
        }
    }
    return 0;
}

// ---- similarity volumes (deviceSimilarityVolume.hpp) ----
void avr_volume_initialize_u8(uint8_t* vol, long long pitch_y, int pitch_x, int X, int Y, int Z, uint8_t value)
{
    CudaDeviceMemoryPitched<TSim, 3> v(CudaSize<3>(X, Y, Z));
    cuda_volumeInitialize(v, value, 0);
    vol_out(vol, pitch_y, pitch_x, v, X, Y, Z);
}
void avr_volume_update_uninitialized(const uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, int X, int Y, int Z)
{
    CudaDeviceMemoryPitched<TSim, 3> b(CudaSize<3>(X, Y, Z)), s(CudaSize<3>(X, Y, Z));
    vol_in(b, best, pitch_y, pitch_x, X, Y, Z);
    vol_in(s, second, pitch_y, pitch_x, X, Y, Z);
    cuda_volumeUpdateUninitializedSimilarity(b, s, 0);
    vol_out(second, pitch_y, pitch_x, s, X, Y, Z);
}
// volumes are X x Y x volZ (volZ = allocated depth, >= the depth range)
void avr_volume_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, int X, int Y, int volZ, const float* depths,
                                   int nDepths, int rcSlot, int tcSlot, void* rcImg, void* tcImg, const avdm_sgm_params_t* sp,
                                   avdm_range_t depthRange, avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<TSim, 3> b(CudaSize<3>(X, Y, volZ)), s(CudaSize<3>(X, Y, volZ));
    vol_in(b, best, pitch_y, pitch_x, X, Y, volZ);
    vol_in(s, second, pitch_y, pitch_x, X, Y, volZ);
    CudaDeviceMemoryPitched<float, 2> d(CudaSize<2>(nDepths, 1)); // Sgm.hpp: _depths_dmp is (maxDepths, 1)
    map_in(d, depths, nDepths * 4, nDepths, 1);
    const SgmParams p = to_ref(sp);
    cuda_volumeComputeSimilarity(b, s, d, rcSlot, tcSlot, *(DeviceMipmapImage*)rcImg, *(DeviceMipmapImage*)tcImg, p, to_ref(depthRange), to_ref(roi), 0);
    vol_out(best, pitch_y, pitch_x, b, X, Y, volZ);
    vol_out(second, pitch_y, pitch_x, s, X, Y, volZ);
}
void avr_volume_refine_similarity(uint16_t* vol, long long pitch_y, int pitch_x, int X, int Y, int volZ, const float* sgmDepthPixSize, int map_pitch,
                                  const float* sgmNormal, int normal_pitch, int rcSlot, int tcSlot, void* rcImg, void* tcImg,
                                  const avdm_refine_params_t* rp, avdm_range_t depthRange, avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<TSimRefine, 3> v(CudaSize<3>(X, Y, volZ));
    vol_in(v, vol, pitch_y, pitch_x, X, Y, volZ);
    CudaDeviceMemoryPitched<float2, 2> m(CudaSize<2>(X, Y));
    map_in(m, sgmDepthPixSize, map_pitch, X, Y);
    CudaDeviceMemoryPitched<float3, 2> nm(CudaSize<2>(X, Y));
    if(sgmNormal)
        map_in(nm, sgmNormal, normal_pitch, X, Y);
    const RefineParams p = to_ref(rp);
    cuda_volumeRefineSimilarity(v, m, sgmNormal ? &nm : nullptr, rcSlot, tcSlot, *(DeviceMipmapImage*)rcImg, *(DeviceMipmapImage*)tcImg, p,
                                to_ref(depthRange), to_ref(roi), 0);
    vol_out(vol, pitch_y, pitch_x, v, X, Y, volZ);
}
// `out` must hold what the reference's output volume holds on entry (it is read-modify-written by paths 1..3 only)
void avr_volume_optimize(uint8_t* out, const uint8_t* in, long long pitch_y, int pitch_x, int X, int Y, int volZ, void* rcImg,
                         const avdm_sgm_params_t* sp, int lastDepthIndex, avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<TSim, 3> o(CudaSize<3>(X, Y, volZ)), i(CudaSize<3>(X, Y, volZ));
    vol_in(o, out, pitch_y, pitch_x, X, Y, volZ);
    vol_in(i, in, pitch_y, pitch_x, X, Y, volZ);
    // Sgm.cpp:59-62: slices are (maxTileSide, maxDepths), the axis accumulator (maxTileSide, 1)
    const size_t side = X > Y ? X : Y;
    CudaDeviceMemoryPitched<TSimAcc, 2> a(CudaSize<2>(side, volZ)), b(CudaSize<2>(side, volZ)), acc(CudaSize<2>(side, 1));
    const SgmParams p = to_ref(sp);
    cuda_volumeOptimize(o, a, b, acc, i, *(DeviceMipmapImage*)rcImg, p, lastDepthIndex, to_ref(roi), 0);
    vol_out(out, pitch_y, pitch_x, o, X, Y, volZ);
}
void avr_volume_retrieve_best_depth(float* outDT, int dt_pitch, float* outDS, int ds_pitch, const float* depths, int nDepths, const uint8_t* vol,
                                    long long pitch_y, int pitch_x, int X, int Y, int volZ, int rcSlot, const avdm_sgm_params_t* sp,
                                    avdm_range_t depthRange, avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<TSim, 3> v(CudaSize<3>(X, Y, volZ));
    vol_in(v, vol, pitch_y, pitch_x, X, Y, volZ);
    CudaDeviceMemoryPitched<float, 2> d(CudaSize<2>(nDepths, 1));
    map_in(d, depths, nDepths * 4, nDepths, 1);
    CudaDeviceMemoryPitched<float2, 2> dt(CudaSize<2>(X, Y)), ds(CudaSize<2>(X, Y));
    const SgmParams p = to_ref(sp);
    cuda_volumeRetrieveBestDepth(dt, ds, d, v, rcSlot, p, to_ref(depthRange), to_ref(roi), 0);
    map_out(outDT, dt_pitch, dt, X, Y);
    if(outDS)
        map_out(outDS, ds_pitch, ds, X, Y);
}
void avr_volume_refine_best_depth(float* out, int out_pitch, const float* sgmDepthPixSize, int map_pitch, const uint16_t* vol, long long pitch_y,
                                  int pitch_x, int X, int Y, int volZ, const avdm_refine_params_t* rp, avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<TSimRefine, 3> v(CudaSize<3>(X, Y, volZ));
    vol_in(v, vol, pitch_y, pitch_x, X, Y, volZ);
    CudaDeviceMemoryPitched<float2, 2> m(CudaSize<2>(X, Y)), o(CudaSize<2>(X, Y));
    map_in(m, sgmDepthPixSize, map_pitch, X, Y);
    const RefineParams p = to_ref(rp);
    cuda_volumeRefineBestDepth(o, m, v, p, to_ref(roi), 0);
    map_out(out, out_pitch, o, X, Y);
}

// ---- depth / similarity maps (deviceDepthSimilarityMap.hpp); W x H = allocated extent of the named map ----
void avr_depth_sim_map_copy_depth_only(float* out, int out_pitch, const float* in, int in_pitch, int W, int H, float defaultSim)
{
    CudaDeviceMemoryPitched<float2, 2> o(CudaSize<2>(W, H)), i(CudaSize<2>(W, H));
    map_in(i, in, in_pitch, W, H);
    cuda_depthSimMapCopyDepthOnly(o, i, defaultSim, 0);
    map_out(out, out_pitch, o, W, H);
}
void avr_normal_map_upscale(float* out, int out_pitch, int outW, int outH, const float* in, int in_pitch, int inW, int inH, avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<float3, 2> o(CudaSize<2>(outW, outH)), i(CudaSize<2>(inW, inH));
    map_in(i, in, in_pitch, inW, inH);
    cuda_normalMapUpscale(o, i, to_ref(roi), 0);
    map_out(out, out_pitch, o, outW, outH);
}
void avr_depth_thickness_smooth_thickness(float* map, int pitch, int W, int H, const avdm_sgm_params_t* sp, const avdm_refine_params_t* rp,
                                          avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<float2, 2> m(CudaSize<2>(W, H));
    map_in(m, map, pitch, W, H);
    const SgmParams s = to_ref(sp);
    const RefineParams r = to_ref(rp);
    cuda_depthThicknessSmoothThickness(m, s, r, to_ref(roi), 0);
    map_out(map, pitch, m, W, H);
}
void avr_compute_sgm_upscaled_depth_pixsize_map(float* out, int out_pitch, int outW, int outH, const float* in, int in_pitch, int inW, int inH,
                                                int rcSlot, void* rcImg, const avdm_refine_params_t* rp, avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<float2, 2> o(CudaSize<2>(outW, outH)), i(CudaSize<2>(inW, inH));
    map_in(i, in, in_pitch, inW, inH);
    const RefineParams r = to_ref(rp);
    cuda_computeSgmUpscaledDepthPixSizeMap(o, i, rcSlot, *(DeviceMipmapImage*)rcImg, r, to_ref(roi), 0);
    map_out(out, out_pitch, o, outW, outH);
}
void avr_depth_sim_map_compute_normal(float* out, int out_pitch, const float* depthSim, int in_pitch, int W, int H, int rcSlot, int stepXY,
                                      avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<float3, 2> o(CudaSize<2>(W, H));
    CudaDeviceMemoryPitched<float2, 2> i(CudaSize<2>(W, H));
    map_in(i, depthSim, in_pitch, W, H);
    cuda_depthSimMapComputeNormal(o, i, rcSlot, stepXY, to_ref(roi), 0);
    map_out(out, out_pitch, o, W, H);
}
// tmpDepth arrives with its current content (the reference binds the whole allocated buffer as a texture, Map.cu:228-229)
void avr_depth_sim_map_optimize_gradient_descent(float* outOpt, int out_pitch, float* imgVariance, int var_pitch, float* tmpDepth, int tmp_pitch,
                                                 int W, int H, const float* sgmDepthPixSize, int sgm_pitch, const float* refineDepthSim,
                                                 int ref_pitch, int rcSlot, void* rcImg, const avdm_refine_params_t* rp, avdm_roi_t roi)
{
    CudaDeviceMemoryPitched<float2, 2> o(CudaSize<2>(W, H)), s(CudaSize<2>(W, H)), r(CudaSize<2>(W, H));
    CudaDeviceMemoryPitched<float, 2> var(CudaSize<2>(W, H)), tmp(CudaSize<2>(W, H));
    map_in(s, sgmDepthPixSize, sgm_pitch, W, H);
    map_in(r, refineDepthSim, ref_pitch, W, H);
    map_in(var, imgVariance, var_pitch, W, H);
    map_in(tmp, tmpDepth, tmp_pitch, W, H);
    const RefineParams p = to_ref(rp);
    cuda_depthSimMapOptimizeGradientDescent(o, var, tmp, s, r, rcSlot, *(DeviceMipmapImage*)rcImg, p, to_ref(roi), 0);
    map_out(outOpt, out_pitch, o, W, H);
    map_out(imgVariance, var_pitch, var, W, H);
    map_out(tmpDepth, tmp_pitch, tmp, W, H);
}

// ---- device helpers called directly (golden vectors for the oracle's restatements) ----
void avr_rgb2lab(const float* rgb01, int n, float* lab) // color.cuh:65-70,124-141: xyz2lab(rgb2xyz(c)), c in [0, 1]
{
    for(int i = 0; i < n; ++i)
    {
        const float3 l = xyz2lab(rgb2xyz(make_float3(rgb01[3 * i], rgb01[3 * i + 1], rgb01[3 * i + 2])));
        lab[3 * i] = l.x; lab[3 * i + 1] = l.y; lab[3 * i + 2] = l.z;
    }
}
void avr_cost_yk_from_lab(const int* dxdy, const float* c1c2, int n, float invGammaC, float invGammaP, float* out) // color.cuh:167-210
{
    for(int i = 0; i < n; ++i)
    {
        const float* c = c1c2 + 8 * i;
        out[i] = CostYKfromLab(dxdy[2 * i], dxdy[2 * i + 1], make_float4(c[0], c[1], c[2], c[3]), make_float4(c[4], c[5], c[6], c[7]), invGammaC, invGammaP);
    }
}
// SimStat.cuh:102-155: weighted NCC of m samples (gx, gy, w) per case
void avr_sim_stat_wsim(const float* gxgyw, int m, int n, float* out)
{
    for(int i = 0; i < n; ++i)
    {
        simStat s;
        for(int k = 0; k < m; ++k)
        {
            const float* g = gxgyw + 3 * ((size_t)i * m + k);
            s.update(g[0], g[1], g[2]);
        }
        out[i] = s.computeWSim();
    }
}
void avr_sigmoid(const float* zv, int n, float zeroVal, float endVal, float sigwidth, float sigMid, float* out, float* out2) // matrix.cuh:334-346
{
    for(int i = 0; i < n; ++i)
    {
        out[i] = sigmoid(zeroVal, endVal, sigwidth, sigMid, zv[i]);
        out2[i] = sigmoid2(zeroVal, endVal, sigwidth, sigMid, zv[i]);
    }
}
// eig33.cuh:351-445: plane through m weighted points per case; out = (p, n) or NaNs when computePlaneByPCA refuses
void avr_stat3d_plane(const float* pts_w, int m, int n, float* out6, int* ok)
{
    for(int i = 0; i < n; ++i)
    {
        cuda_stat3d s;
        for(int k = 0; k < m; ++k)
        {
            const float* q = pts_w + 4 * ((size_t)i * m + k);
            s.update(make_float3(q[0], q[1], q[2]), q[3]);
        }
        float3 p, nn;
        ok[i] = s.computePlaneByPCA(p, nn) ? 1 : 0;
        out6[6 * i] = p.x; out6[6 * i + 1] = p.y; out6[6 * i + 2] = p.z;
        out6[6 * i + 3] = nn.x; out6[6 * i + 4] = nn.y; out6[6 * i + 5] = nn.z;
    }
}
void avr_project3d(const float* P12, const float* pts, int n, float* out2) // matrix.cuh:117-126
{
    for(int i = 0; i < n; ++i)
    {
        const float2 r = project3DPoint(P12, make_float3(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]));
        out2[2 * i] = r.x; out2[2 * i + 1] = r.y;
    }
}

} // extern "C"
