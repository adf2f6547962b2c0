/*
 * avdm_oracle.h — CPU parity oracle (TEST INFRASTRUCTURE ONLY; pinned to the reference's own code through oracle/_ref — see avdm_oracle.c header).
 * Types are shared with the product ABI (include/avdm.h); all pointers are HOST pointers here.
 */
#ifndef AVDM_ORACLE_H
#define AVDM_ORACLE_H

#include "../include/avdm.h"

#ifdef __cplusplus
extern "C" {
#endif

uint16_t avo_float_to_half(float f);
float avo_half_to_float(uint16_t h);
float avo_exp_p2(float x);
void avo_set_exact_rc_pixel(int on);  /* 0 = literal reprojected R pixel in the border test (default), 1 = exact pixel */
void avo_set_ncc_precision(int f64); /* 0 = fp32 sums like the reference (default), 1 = double-precision sums */
void avo_tex2dlod(const avdm_pyramid_t* p, float u, float v, float lod, float out[4]);

int avo_pyramid_layout(avdm_pyramid_t* p, int width, int height, int min_downscale, int max_downscale, int filter_mode);
void avo_image_rgba_f32_to_f16x255(uint16_t* out, int out_pitch, const float* in, int in_pitch, int width, int height);
void avo_rgb2lab(uint16_t* img, int pitch, int width, int height);
void avo_downscale_with_gaussian_blur(uint16_t* out, int out_pitch, int out_w, int out_h, const uint16_t* in, int in_pitch, int in_w, int in_h,
                                      int downscale, int gaussRadius, int filter_mode);
void avo_pyramid_build_levels(const avdm_pyramid_t* p);
int avo_pyramid_fill(const avdm_pyramid_t* p, const float* rgba, int in_pitch);
/* imageAlgo::resizeImage(downscale, in, out) -> oiio::ImageBufAlgo::resize with the default filter (see avdm_oracle.c) */
int avo_image_resize(float* dst, int dst_pitch, int dst_w, int dst_h, const float* src, int src_pitch, int src_w, int src_h, int nchannels);
int avo_image_decode_integer(float* dst, int dst_pitch, const void* src, int src_pitch, int width, int height, int channels, int bits, int srgb_to_linear);
/* libjpeg(-turbo)'s default decode of one image from its quantised coefficients (host memory): jidctint.c, jdsample.c, jdcolor.c */
int avo_image_decode_jpeg(uint8_t* dst_rgb, int dst_pitch, int width, int height, const avdm_jpeg_component_t* comps, int n_comps, int hmax, int vmax,
                          int ycc_to_rgb);
int avo_image_resize_taps(int dst_n, int src_n, float* weights /* [dst_n][taps] or NULL */, int* first /* [dst_n] or NULL */);
/* camera::UndistortImage (camera/cameraUndistortImage.hpp:81-139) */
int avo_image_undistort(float* dst, int dst_pitch, const float* src, int src_pitch, const avdm_intrinsic_t* cam, const float fill[4]);
void avo_camera_fill(avdm_camera_t* out, const double K[9], const double R[9], const double C[3], int downscale);

void avo_volume_initialize_u8(uint8_t* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, uint8_t value);
void avo_volume_initialize_f16(uint16_t* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, float value);
void avo_volume_add_f16(uint16_t* inout, const uint16_t* in, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ);
void avo_volume_update_uninitialized(const uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ);
void avo_volume_compute_similarity(uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, const float* depths, const avdm_camera_t* rc,
                                   const avdm_camera_t* tc, const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr,
                                   const avdm_sgm_params_t* sp, avdm_range_t depthRange, avdm_roi_t roi);
void avo_volume_refine_similarity(uint16_t* vol, long long pitch_y, int pitch_x, int volDimZ, const float* sgmDepthPixSize, int map_pitch,
                                  const float* sgmNormal, int normal_pitch, const avdm_camera_t* rc, const avdm_camera_t* tc,
                                  const avdm_pyramid_t* rcPyr, const avdm_pyramid_t* tcPyr, const avdm_refine_params_t* rp, avdm_range_t depthRange,
                                  avdm_roi_t roi);
void avo_volume_optimize(uint8_t* out, const uint8_t* in, long long pitch_y, int pitch_x, int dimX, int dimY, const avdm_pyramid_t* rcPyr,
                         const avdm_sgm_params_t* sp, int lastDepthIndex, avdm_roi_t roi);
void avo_volume_retrieve_best_depth(float* outDT, int dt_pitch, float* outDS, int ds_pitch, const float* depths, const uint8_t* vol,
                                    long long pitch_y, int pitch_x, int volDimZ, const avdm_camera_t* rc1, const avdm_sgm_params_t* sp,
                                    avdm_range_t depthRange, avdm_roi_t roi);
void avo_volume_refine_best_depth(float* out, int out_pitch, const float* sgmDepthPixSize, int map_pitch, const uint16_t* vol, long long pitch_y,
                                  int pitch_x, int volDimZ, const avdm_refine_params_t* rp, avdm_roi_t roi);

void avo_depth_sim_map_copy_depth_only(float* out, int out_pitch, const float* in, int in_pitch, int width, int height, float defaultSim);
void avo_normal_map_upscale(float* out, int out_pitch, const float* in, int in_pitch, float ratio, avdm_roi_t roi);
void avo_depth_sim_map_compute_normal(float* out, int out_pitch, const float* depthSim, int in_pitch, const avdm_camera_t* rc, int stepXY, avdm_roi_t roi);
void avo_depth_thickness_smooth_thickness(float* map, int pitch, const avdm_sgm_params_t* sp, const avdm_refine_params_t* rp, avdm_roi_t roi);
void avo_compute_sgm_upscaled_depth_pixsize_map(float* out, int out_pitch, const float* in, int in_pitch, const avdm_camera_t* rc,
                                                const avdm_pyramid_t* rcPyr, const avdm_refine_params_t* rp, float ratio, avdm_roi_t roi);
void avo_depth_sim_map_optimize_gradient_descent(float* outOpt, int out_pitch, float* imgVariance, int var_pitch, float* tmpDepth, int tmp_pitch,
                                                 int tmpW, int tmpH, const float* sgmDepthPixSize, int sgm_pitch, const float* refineDepthSim,
                                                 int ref_pitch, const avdm_camera_t* rc, const avdm_pyramid_t* rcPyr,
                                                 const avdm_refine_params_t* rp, avdm_roi_t roi);

/* custom patch pattern (patchPattern.cpp:18-251): builds the pattern and makes it the oracle's current one */
int avo_build_custom_patch_pattern(int n_subparts, const avdm_patch_subpart_params_t* subparts, int group, avdm_patch_pattern_t* out);

/* ---- depth-map filtering (avdm_fuse_oracle.c; fuseCut/Fuser.cpp:66-304) ---- */
/* camArr (3x4), iCamArr (3x3), CArr of a camera, row-major, and the image size MultiViewParams reports for it */
typedef struct
{
    double P[12];
    double iP[9];
    double C[3];
    int width, height;
} avo_fuse_cam_t;

int avo_fuse_filter_groups_rc(unsigned char* nmod, const float* depth, const float* sim, const avo_fuse_cam_t* rc, int n_tc, const avo_fuse_cam_t* tcs,
                              const float* const* tc_depth, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP);
void avo_fuse_filter_depth_maps_rc(float* depthMap, float* simMap, const unsigned char* numOfModalsMap, size_t n, int minNumOfModals,
                                   int minNumOfModalsWSP2SSP);
double avo_fuse_pixel_size_plane_sweep_alpha(const double p[3], const avo_fuse_cam_t* rc, const avo_fuse_cam_t* tc);

#ifdef __cplusplus
}
#endif
#endif
