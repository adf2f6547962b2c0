/*
 * avdm.h — C ABI of the MI355X-native depth-map estimation hot path
 * (plane-sweep similarity volume -> 4-path SGM aggregation -> WTA depth ->
 *  Refine re-sweep -> sliding-Gaussian sub-sample arg-min -> colour-guided
 *  optimisation), the drop-in for AliceVision's `src/aliceVision/depthMap`
 *  kernel-launch layer.
 *
 * Every entry point replaces one host wrapper of the reference (cited per
 * function as file:line relative to /root/reference/src/aliceVision/depthMap).
 * Differences from the reference wrappers, all forced by the C ABI:
 *   - `int` status (0 = ok, otherwise a hipError_t value; text through
 *     avdm_last_error()) instead of C++ exceptions (cuda/host/utils.hpp:10-40);
 *   - camera parameters passed by pointer to a host-side avdm_camera_t (copied
 *     into kernel arguments) instead of a slot id into __constant__ memory
 *     (cuda/device/DeviceCameraParams.hpp:32-34);
 *   - images passed as avdm_pyramid_t (fp16 RGBA Lab pyramid in plain device
 *     memory, filtered in ALU) instead of a cudaTextureObject_t over a
 *     mipmapped array (cuda/host/DeviceMipmapImage.cpp:28-90);
 *   - buffers as (pointer, pitches) instead of CudaDeviceMemoryPitched<T,N>.
 *
 * Volume layout (ours): z-fastest.  A voxel (x,y,z) of a uint8
 * similarity volume lives at  base + y*pitch_y + x*pitch_x + z  (bytes);
 * of the fp16 refine volume at  base + y*pitch_y + x*pitch_x + 2*z.
 * pitch_x must be a multiple of 4 (u8) / 16 (fp16 volume) and >= Z (resp. 2*Z).
 *
 * All functions are asynchronous on `stream` (a hipStream_t passed as void*),
 * borrow caller-allocated device buffers and never allocate device memory,
 * except avdm_scratch_* helpers which are explicit.
 */
#ifndef AVDM_H
#define AVDM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AVDM_MAX_LEVELS 8

/* texture filter arithmetic (see DESIGN.md "Texture unit restatement") */
#define AVDM_FILTER_EXACT 0      /* fp32 bilinear / mip weights                         */
#define AVDM_FILTER_CUDA_FIXED8 1 /* weights quantised to 1.8 fixed point like the CUDA texture unit */

/* mirrors DeviceCameraParams (cuda/device/DeviceCameraParams.hpp:16-28); all matrices column-major */
typedef struct avdm_camera
{
    float P[12];
    float iP[9];
    float R[9];
    float iR[9];
    float K[9];
    float iK[9];
    float C[3];
    float XVect[3];
    float YVect[3];
    float ZVect[3];
} avdm_camera_t;

/* mirrors ROI / Range (mvsData/ROI.hpp:35-126): half-open ranges, already divided by scale*stepXY */
typedef struct avdm_range { unsigned int begin, end; } avdm_range_t;
typedef struct avdm_roi { avdm_range_t x, y; } avdm_roi_t;

/* replaces DeviceMipmapImage (cuda/host/DeviceMipmapImage.hpp:25-84): fp16 (L,a,b,alpha) texels, 8 B each */
typedef struct avdm_pyramid
{
    void* base;                        /* device (or host, for the oracle) pointer to level 0 */
    int levels;                        /* number of levels actually built                     */
    int filter_mode;                   /* AVDM_FILTER_*                                        */
    int min_downscale;                 /* downscale of level 0 w.r.t. the process image        */
    int width0, height0;               /* size of the process image (before min_downscale)    */
    int width[AVDM_MAX_LEVELS];        /* texels per row of each level (floor halving, like cudaMallocMipmappedArray) */
    int height[AVDM_MAX_LEVELS];
    int pitch[AVDM_MAX_LEVELS];        /* bytes per row                                        */
    long long offset[AVDM_MAX_LEVELS]; /* byte offset of each level from base                  */
    long long bytes;                   /* total allocation size                                */
} avdm_pyramid_t;

/* subset of SgmParams used by the kernels (SgmParams.hpp:21-55) */
typedef struct avdm_sgm_params
{
    int scale;
    int stepXY;
    int wsh;
    double gammaC;
    double gammaP;
    double p1;
    double p2Weighting;
    double maxSimilarity;
    double depthThicknessInflate;
    char filteringAxes[8];    /* ordered axis string, e.g. "YX" (SgmParams.hpp:34) */
    int useConsistentScale;
    int strictRoiQuirk;       /* 1 = replicate the begin-x/begin-y swap of deviceSimilarityVolumeKernels.cuh:688-709 */
    int useCustomPatchPattern; /* SgmParams.hpp:51: compare with the pattern of avdm_build_custom_patch_pattern instead of the wsh square */
    int referenceArithmetic;  /* avdm_volume_compute_similarity: 1 = the reference's arithmetic AS WRITTEN (compNCCby3DptsYK, Patch.cuh:466-572;
                                 simStat, SimStat.cuh:72-153; CostYKfromLab, color.cuh:167-210: operation for operation, IEEE division / square
                                 root, no contraction, expf to the bits of the pinned reference build's C library) — similarity volumes equal
                                 to the reference's own code compiled for the CPU bit for bit, at ~6 x the instructions of the default sweep;
                                 0 (default) = the packed multi-plane kernels (tolerance class) */
} avdm_sgm_params_t;

/* subset of RefineParams used by the kernels (RefineParams.hpp:19-45) */
typedef struct avdm_refine_params
{
    int scale;
    int stepXY;
    int wsh;
    int halfNbDepths;
    int nbSubsamples;
    int optimizationNbIterations;
    double sigma;
    double gammaC;
    double gammaP;
    int interpolateMiddleDepth;
    int useConsistentScale;
    int useCustomPatchPattern; /* RefineParams.hpp:42 */
    int referenceArithmetic;   /* avdm_volume_refine_similarity: as avdm_sgm_params_t::referenceArithmetic */
} avdm_refine_params_t;

/* ---- custom patch pattern (cuda/device/DevicePatchPattern.hpp:11-54) ---- */
#define AVDM_PATCH_MAX_SUBPARTS 4
#define AVDM_PATCH_MAX_COORDS_PER_SUBPART 24
typedef struct avdm_patch_pattern_subpart
{
    float coordinates[AVDM_PATCH_MAX_COORDS_PER_SUBPART][2]; /* circle(s): patch-relative sample positions */
    int nbCoordinates;
    float level;     /* mipmap level added to the stage's level (>= 0) */
    float downscale; /* 2^level */
    float weight;
    int isCircle;
    int wsh;         /* half-width of a full subpart */
} avdm_patch_pattern_subpart_t;
typedef struct avdm_patch_pattern
{
    avdm_patch_pattern_subpart_t subparts[AVDM_PATCH_MAX_SUBPARTS]; /* one similarity per subpart */
    int nbSubparts;
} avdm_patch_pattern_t;
/* CustomPatchPatternParams::SubpartParams (CustomPatchPatternParams.hpp:26-33) */
typedef struct avdm_patch_subpart_params
{
    int isCircle;
    int level;
    int nbCoordinates;
    float radius;
    float weight;
} avdm_patch_subpart_params_t;

/* ---- library ---- */
const char* avdm_last_error(void);
int avdm_version(void);
/* gpu/gpu.cpp:15-66 (gpuSupportCUDA / gpuInformationCUDA) */
int avdm_device_count(void);
int avdm_device_info(int device, char* out, size_t out_len);
/* Gives back what the library keeps per stream on the current device: the temporary maps of avdm_depth_sim_map_optimize_gradient_descent and
 * the tap tables of avdm_image_resize live in one block per (device, stream) that is reused by later calls on that stream (the reference
 * passes a pre-allocated CudaDeviceMemoryPitched for the former, Refine.cpp:60-66; the wrapper signature has no room for the point maps
 * this implementation iterates on).  Call it before hipStreamDestroy; waits for the stream.  Returns 2 (and frees nothing) while another
 * host thread is inside an entry point on that stream. */
int avdm_stream_release(void* stream);

/* buildCustomPatchPattern (cuda/host/patchPattern.cpp:18-251): validates the subparts, builds the pattern and makes it the one the
 * similarity entry points use when their parameters say useCustomPatchPattern (the reference keeps it in constant memory,
 * DevicePatchPattern.hpp:51).  `out` (optional) receives a copy.  In the non-grouped form the reference reads the circle's number
 * of coordinates from uninitialised memory (:196-201); the value of the subpart's parameters is used here. */
int avdm_build_custom_patch_pattern(int n_subparts, const avdm_patch_subpart_params_t* subparts, int group_subparts_per_level,
                                    avdm_patch_pattern_t* out);

/* ---- image side ---- */
/* host-only: fills width/height/pitch/offset/bytes for an image of w x h process pixels.
 * DeviceMipmapImage.cpp:28-35,92-108 (levels = log2(maxDs/minDs)+1, level dims) */
int avdm_pyramid_layout(avdm_pyramid_t* pyr, int width, int height, int min_downscale, int max_downscale, int filter_mode);
/* float RGBA (0..1, device) * 255 -> fp16 texels.  cuda/host/DeviceCache.cpp:249-280 */
int avdm_image_rgba_f32_to_f16x255(void* out_h4, int out_pitch, const float* in_rgba, int in_pitch, int width, int height, void* stream);
/* in-place linear RGB(0..255) -> CIELAB*2.55.  imageProcessing/deviceColorConversion.cu:16-59 (cuda_rgb2lab) */
int avdm_rgb2lab(void* inout_h4, int pitch, int width, int height, void* stream);
/* (2r+1)^2 Gaussian downscale.  imageProcessing/deviceGaussianFilter.cu:44-80,268-287 (cuda_downscaleWithGaussianBlur) */
int avdm_downscale_with_gaussian_blur(void* out_h4, int out_pitch, int out_w, int out_h, const void* in_h4, int in_pitch, int in_w, int in_h,
                                      int downscale, int gauss_radius, int filter_mode, void* stream);
/* levels 1..n-1 from level 0.  imageProcessing/deviceMipmappedArray.cu:20-92,222-327 (cuda_createMipmappedArrayFromImage) */
int avdm_pyramid_build_levels(const avdm_pyramid_t* pyr, void* stream);
/* probe of the software texture unit: out[i] = tex2DLod<float4>(pyr, uvl[3i], uvl[3i+1], uvl[3i+2]) for n device-resident samples.
 * Restates the texture object of deviceMipmappedArray.cu:329-351 (normalised coords, linear + mip-linear, clamp). */
int avdm_tex2dlod(float* out4, const avdm_pyramid_t* pyr, const float* uvl, int n, void* stream);
/* The --downscale resize of the input images (first slice of the GPU image ingest):
 * imageAlgo::resizeImage(downscale, in, out) (image/imageAlgo.cpp:220-235, 357-368) as mvsUtils/fileIO.cpp:432-441 (loadImage) calls it,
 * i.e. oiio::ImageBufAlgo::resize(out, in, "", 0) — OpenImageIO's default filter: a separable 6-pixel lanczos3 when shrinking.  Float
 * RGBA, device memory, dst no larger than src (enlarging would select blackman-harris: refused).  dst_w = src_w / downscale and
 * dst_h = src_h / downscale (integer division) are the caller's, like in the reference. */
int avdm_image_resize(float* dst_rgba, int dst_pitch, int dst_w, int dst_h, const float* src_rgba, int src_pitch, int src_w, int src_h, void* stream);
/* Integer image samples -> linear float RGBA on the device: what image::readImage(path, img, LINEAR) hands mvsUtils::loadImage
 * (mvsUtils/fileIO.cpp:386-446) for an 8- / 16-bit file.  v / 255 resp. v / 65535; colour channels through OpenImageIO's sRGB decoding when
 * `srgb_to_linear` (x <= 0.04045 ? x / 12.92 : ((x + 0.055) / 1.055) ^ 2.4), alpha never; one channel replicated, missing alpha = 1.
 * `src`: `channels` (1 Y, 2 YA, 3 RGB, 4 RGBA) interleaved samples of `bits` (8, or 16 in host byte order), device memory.  The curve is a
 * host-evaluated table (C library powf): the device result is a look-up, bit-identical to the oracle's. */
int avdm_image_decode_integer(float* dst_rgba, int dst_pitch, const void* src, int src_pitch, int width, int height, int channels, int bits,
                              int srgb_to_linear, void* stream);
/* OpenEXR scan lines -> linear float RGBA on the device: image::readImage(path, img, LINEAR) as mvsUtils/fileIO.cpp:386-446
 * calls it for an .exr.  `lines` = the scan lines as stored (per line the channels one after the other, `width` samples each), line y at
 * lines + y * line_stride.  chan_offset[k] / chan_type[k]: byte offset in a line and pixel type (0 UINT, 1 HALF, 2 FLOAT) of R, G, B, A
 * (A offset -1: absent -> 1; Y-only: Y three times).  Unaligned samples are read byte by byte.  Bit-identical to host/exr.cpp readExr. */
int avdm_image_decode_exr_lines(float* dst_rgba, int dst_pitch, const void* lines, long long line_stride, int width, int height,
                                const long long chan_offset[4], const int chan_type[4], void* stream);
/* one component of a JPEG frame as the host's entropy decoder leaves it (avdm_image_decode_jpeg below) */
typedef struct avdm_jpeg_component
{
    const int16_t* coef; /* device memory: blocks_h x blocks_w blocks of 64 quantised coefficients, natural (row-major) order */
    int blocks_w, blocks_h, width, height, h_samp, v_samp; /* width / height in samples: ceil(image * samp / max samp) */
    uint16_t quant[64]; /* natural order */
} avdm_jpeg_component_t;
/* JPEG (image ingest): quantised DCT coefficients -> 8-bit RGB on the device, exactly as libjpeg(-turbo) decodes with its
 * defaults (JDCT_ISLOW inverse DCT, fancy up-sampling, YCbCr -> RGB): what OpenImageIO's JPEG reader hands image::readImage
 * (image/io.cpp:571-760) of the reference.  Entropy decoding stays on the host (host/jpeg.cpp).  4:4:4, 4:2:2 (h2v1), 4:2:0 (h2v2).
 * `scratch`: avdm_image_decode_jpeg_scratch_bytes bytes of device memory. */
size_t avdm_image_decode_jpeg_scratch_bytes(const avdm_jpeg_component_t* comps, int n_comps);
int avdm_image_decode_jpeg(uint8_t* dst_rgb, int dst_pitch, int width, int height, const avdm_jpeg_component_t* comps, int n_comps, int hmax, int vmax,
                           int ycc_to_rgb, void* scratch, void* stream);
/* Undistortion of an input image (second slice of the image ingest): camera::UndistortImage(imageIn, intrinsic, image_ud,
 * fillcolor) (camera/cameraUndistortImage.hpp:81-139) as software/pipeline/main_prepareDenseScene.cpp:71-79 calls it — for every pixel of
 * the undistorted image the distorted position  cam2ima(addDistortion(ima2cam(p)))  (camera/IntrinsicScaleOffsetDisto.cpp:80,
 * IntrinsicScaleOffset.cpp:31-66, DistortionRadial.cpp:18-24, 110-124, 262-277) in double precision, then the bilinear sampler of
 * image/Sampler.hpp:377-477 (out-of-range neighbours dropped and the weights renormalised, nearest pixel when less than 0.2 of the weight is
 * left) if the position lies in the image, else the fill colour.  Float RGBA, device memory, same size in and out. */
#define AVDM_DISTORTION_NONE 0
#define AVDM_DISTORTION_RADIALK1 1
#define AVDM_DISTORTION_RADIALK3 2
#define AVDM_DISTORTION_RADIALK3PT 3
typedef struct avdm_intrinsic
{
    int width, height;          /* camera::IntrinsicBase::_w, _h                                            */
    double scale_x, scale_y;    /* IntrinsicScaleOffset::_scale: focal length in pixels                      */
    double offset_x, offset_y;  /* IntrinsicScaleOffset::_offset: principal point - image centre             */
    int distortion_model;       /* AVDM_DISTORTION_*                                                          */
    double k[3];                /* distortion parameters                                                      */
} avdm_intrinsic_t;
/* camera::UndistortImage (camera/cameraUndistortImage.hpp:81-139), as described above the model constants */
int avdm_image_undistort(float* dst_rgba, int dst_pitch, const float* src_rgba, int src_pitch, const avdm_intrinsic_t* cam, const float fill_rgba[4],
                         void* stream);
/* convenience: the whole of DeviceCache::addMipmapImage (cuda/host/DeviceCache.cpp:222-281) + DeviceMipmapImage::fill
 * (cuda/host/DeviceMipmapImage.cpp:28-90) for an image already on the device.
 * `scratch_h4` must hold width*height fp16x4 texels when min_downscale > 1 (may be NULL otherwise). */
int avdm_pyramid_fill(const avdm_pyramid_t* pyr, const float* in_rgba, int in_pitch, void* scratch_h4, void* stream);

/* ---- similarity volume (planeSweeping/deviceSimilarityVolume.hpp) ---- */
/* cuda_volumeInitialize(TSim) :25 */
int avdm_volume_initialize_u8(uint8_t* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, uint8_t value, void* stream);
/* cuda_volumeInitialize(TSimRefine) :33 */
int avdm_volume_initialize_f16(void* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, float value, void* stream);
/* cuda_volumeAdd :41-43 */
int avdm_volume_add_f16(void* inout_vol, const void* in_vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, void* stream);
/* cuda_volumeUpdateUninitializedSimilarity :51-53 */
int avdm_volume_update_uninitialized(const uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, void* stream);
/* cuda_volumeComputeSimilarity :69-79 */
int avdm_volume_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x,
                                   const float* depths, const avdm_camera_t* rc, const avdm_camera_t* tc,
                                   const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr,
                                   const avdm_sgm_params_t* params, avdm_range_t depth_range, avdm_roi_t roi, void* stream);
/* cuda_volumeRefineSimilarity :95-105 (normal map optional, may be NULL) */
/* (no counterpart in the reference: its kernels take no library-owned scratch) what avdm_volume_refine_similarity draws from the stream's scratch
 * block for a sweep of n_pixels pixels x n_planes planes, and the number of list units that found it full since the last call (they ran on
 * the slower path; results unchanged) — for the scheduler that sizes the tile slots, host/DepthMapEstimator.cpp; reference: the per-tile
 * memory estimate of DepthMapEstimator.cpp:57-135 */
size_t avdm_refine_similarity_scratch_bytes(size_t n_pixels, int n_planes);
int avdm_refine_outlier_refused(unsigned int* out);
int avdm_volume_refine_similarity(void* vol_f16, long long pitch_y, int pitch_x, int dimZ,
                                  const float* sgm_depth_pixsize, int map_pitch, const float* sgm_normal, int normal_pitch,
                                  const avdm_camera_t* rc, const avdm_camera_t* tc,
                                  const avdm_pyramid_t* rc_pyr, const avdm_pyramid_t* tc_pyr,
                                  const avdm_refine_params_t* params, avdm_range_t depth_range, avdm_roi_t roi, void* stream);
/* cuda_volumeOptimize :120-129.  `scratch` must hold avdm_volume_optimize_scratch_bytes() bytes (device memory, 4-byte aligned).
 * The paths are walked over roi.width() x roi.height() columns starting at the volume's origin; image coordinates (adaptive P2) start at
 * roi's begin.  The reference walks the ALLOCATED volume (deviceSimilarityVolume.cu:278-283: the tile in the corner of a buffer-sized
 * volume whose remainder holds 255): a caller that wants its bytes for tiles with an offset hands over roi.end = roi.begin + that
 * extent. */
size_t avdm_volume_optimize_scratch_bytes(int dimX, int dimY, int dimZ);
int avdm_volume_optimize(uint8_t* out_vol, const uint8_t* in_vol, long long pitch_y, int pitch_x, void* scratch,
                         const avdm_pyramid_t* rc_pyr, const avdm_sgm_params_t* params, int last_depth_index, avdm_roi_t roi, void* stream);
/* cuda_volumeOptimize :120-129 for ALL tiles of a batch in one launch per path (the reference runs one tile per CUDA stream,
 * DepthMapEstimator.cpp:375-444; the path recurrence only parallelises over columns, so tiles are batched instead).
 * `scratch` must hold the SUM of avdm_volume_optimize_scratch_bytes() over the tiles.  Tiles may belong to different R cameras. */
typedef struct avdm_sgm_tile
{
    uint8_t* out_vol;
    const uint8_t* in_vol;
    long long pitch_y;
    int pitch_x;
    int last_depth_index;
    avdm_roi_t roi;
    const avdm_pyramid_t* rc_pyr;
} avdm_sgm_tile_t;
int avdm_volume_optimize_tiles(int n_tiles, const avdm_sgm_tile_t* tiles, void* scratch, const avdm_sgm_params_t* params, void* stream);
/* cuda_volumeOptimize :120-129 in two halves.  The adaptive P2 (deviceSimilarityVolumeKernels.cuh:696-720: the colour step of the R image
 * between neighbouring stage pixels) depends on nothing but the R pyramid, the ROI and the parameters: _prepare evaluates those maps into
 * `scratch` as soon as the R pyramid exists (the tiles' volumes are not touched and may be NULL), e.g. on a side stream beside the
 * similarity sweep.  The caller orders the two halves (same stream, or an event). */
int avdm_volume_optimize_prepare(int n_tiles, const avdm_sgm_tile_t* tiles, void* scratch, const avdm_sgm_params_t* params, void* stream);
/* ... and the path launches alone (deviceSimilarityVolume.cu:376-425) on the SAME tiles / scratch / params after avdm_volume_optimize_prepare:
 * prepare + prepared == avdm_volume_optimize_tiles, byte for byte. */
int avdm_volume_optimize_tiles_prepared(int n_tiles, const avdm_sgm_tile_t* tiles, void* scratch, const avdm_sgm_params_t* params, void* stream);
/* cuda_volumeRetrieveBestDepth :143-151 (out_depth_sim may be NULL); vol_dimZ = allocated depth of the volume (kernels.cuh:459) */
int avdm_volume_retrieve_best_depth(float* out_depth_thickness, int dt_pitch, float* out_depth_sim, int ds_pitch,
                                    const float* depths, const uint8_t* vol, long long pitch_y, int pitch_x, int vol_dimZ,
                                    const avdm_camera_t* rc_scale1, const avdm_sgm_params_t* params, avdm_range_t depth_range, avdm_roi_t roi,
                                    void* stream);
/* cuda_volumeRefineBestDepth :163-168 */
int avdm_volume_refine_best_depth(float* out_depth_sim, int out_pitch, const float* sgm_depth_pixsize, int map_pitch,
                                  const void* vol_f16, long long pitch_y, int pitch_x, int dimZ,
                                  const avdm_refine_params_t* params, avdm_roi_t roi, void* stream);

/* ---- depth/sim maps (planeSweeping/deviceDepthSimilarityMap.hpp); maps are float2 rows, pitch in bytes ---- */
/* cuda_depthSimMapCopyDepthOnly :25-28 */
int avdm_depth_sim_map_copy_depth_only(float* out_map, int out_pitch, const float* in_map, int in_pitch, int width, int height, float default_sim,
                                       void* stream);
/* cuda_normalMapUpscale :37-40 (float3 maps) */
int avdm_normal_map_upscale(float* out_map, int out_pitch, const float* in_map, int in_pitch, float ratio, avdm_roi_t roi, void* stream);
/* cuda_depthThicknessSmoothThickness :50-54 */
int avdm_depth_thickness_smooth_thickness(float* inout_map, int pitch, const avdm_sgm_params_t* sgm, const avdm_refine_params_t* refine,
                                          avdm_roi_t roi, void* stream);
/* cuda_computeSgmUpscaledDepthPixSizeMap :66-72; ratio = allocated SGM map width / allocated Refine map width (Map.cu:116-118) */
int avdm_compute_sgm_upscaled_depth_pixsize_map(float* out_map, int out_pitch, const float* in_sgm_depth_thickness, int in_pitch,
                                                const avdm_camera_t* rc, const avdm_pyramid_t* rc_pyr, const avdm_refine_params_t* params,
                                                float ratio, avdm_roi_t roi, void* stream);
/* cuda_depthSimMapComputeNormal :83-88 (float3 out) */
int avdm_depth_sim_map_compute_normal(float* out_normal, int out_pitch, const float* in_depth_sim, int in_pitch, const avdm_camera_t* rc,
                                      int stepXY, avdm_roi_t roi, void* stream);
/* cuda_depthSimMapOptimizeGradientDescent :103-112.  tmp_depth/img_variance are float maps of at least roi size;
 * tmp_depth_w/h = allocated extent of tmp_depth (the reference binds the whole buffer as a clamped texture, Map.cu:228-229). */
int avdm_depth_sim_map_optimize_gradient_descent(float* out_opt_depth_sim, int out_pitch, float* img_variance, int var_pitch,
                                                 float* tmp_depth, int tmp_pitch, int tmp_w, int tmp_h,
                                                 const float* sgm_depth_pixsize, int sgm_pitch, const float* refine_depth_sim, int ref_pitch,
                                                 const avdm_camera_t* rc, const avdm_pyramid_t* rc_pyr, const avdm_refine_params_t* params,
                                                 avdm_roi_t roi, void* stream);

/* ---- host helper: fillHostCameraParameters (cuda/host/DeviceCache.cpp:41-134) ---- */
/* fillHostCameraParameters (cuda/host/DeviceCache.cpp:41-134): K,R row-major 3x3 doubles, C 3 doubles; downscale >= 1 */
void avdm_camera_fill(avdm_camera_t* out, const double K[9], const double R[9], const double C[3], int downscale);

#ifdef __cplusplus
}
#endif
#endif /* AVDM_H */
