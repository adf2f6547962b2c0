/* avdm_fuse.h — C ABI of the depth-map filtering step that follows depth-map estimation, gfx950.
 *
 * The reference runs this step on the CPU (aliceVision_depthMapFiltering -> fuseCut::Fuser); its interface is C++:
 *   fuseCut/Fuser.hpp:36-40   bool Fuser::filterGroupsRC(int rc, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, int nNearestCams)
 *   fuseCut/Fuser.hpp:41-42   bool Fuser::filterDepthMapsRC(int rc, int minNumOfModals, int minNumOfModalsWSP2SSP)
 * The entry points below are those two functions without the file I/O and the camera ranking (host side, see
 * alicevision_amd/host/Fuser.cpp): plain pointers to DEVICE buffers, sizes, a HIP stream.  All maps are row-major with a row pitch
 * in bytes.  Return 0 on success; avdm_last_error() (avdm.h) holds the message otherwise.
 *
 * Arithmetic: the reference's double / float expressions in their order (Fuser.cpp:66-121, MultiViewParams.cpp:337-448,
 * common.cpp:23-170, geometry.cpp:14-146), IEEE division and square root, no FMA contraction: the modal-count map is bit-exact
 * against the CPU restatement (oracle/avdm_fuse_oracle.c).
 */
#ifndef AVDM_FUSE_H
#define AVDM_FUSE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* mvsUtils::MultiViewParams camArr[c] (3x4), iCamArr[c] (3x3), CArr[c] — row-major — and getWidth(c) / getHeight(c)
 * (MultiViewParams.hpp:53-66) */
typedef struct avdm_fuse_camera
{
    double P[12];
    double iP[9];
    double C[3];
    int width, height;
} avdm_fuse_camera_t;

/* one T camera of Fuser::filterGroupsRC (Fuser.cpp:178-218): its depth map (cam.width x cam.height floats, device) or NULL when it
 * has none (the camera is skipped, :189) */
typedef struct avdm_fuse_tc
{
    const float* depth;
    int depth_pitch;
    int reserved;
    avdm_fuse_camera_t cam;
} avdm_fuse_tc_t;

/* bytes of device scratch avdm_fuse_filter_groups needs for a width x height reference camera (stands for the hit counters
 * numOfPtsMap, Fuser.cpp:172-174) */
size_t avdm_fuse_filter_groups_scratch_bytes(int width, int height);

/* Fuser::filterGroupsRC (Fuser.cpp:144-231): out_nmod[y][x] = number of T cameras in which the pixel of the reference camera finds a
 * consistent depth (with the reference's carry-over of the hit counters from one T camera to the next).
 * rc_depth / rc_sim: the reference camera's depth and similarity maps (rc->width x rc->height); tcs in ranking order. */
int avdm_fuse_filter_groups(unsigned char* out_nmod, int nmod_pitch, const float* rc_depth, int depth_pitch, const float* rc_sim, int sim_pitch,
                            const avdm_fuse_camera_t* rc, int n_tc, const avdm_fuse_tc_t* tcs, float pixToleranceFactor, int pixSizeBall,
                            int pixSizeBallWSP, void* scratch, void* stream);

/* Fuser::filterDepthMapsRC (Fuser.cpp:250-304), in place on depth / sim */
int avdm_fuse_filter_depth_maps(float* depth, int depth_pitch, float* sim, int sim_pitch, const unsigned char* nmod, int nmod_pitch, int width,
                                int height, int minNumOfModals, int minNumOfModalsWSP2SSP, void* stream);

#ifdef __cplusplus
}
#endif
#endif
