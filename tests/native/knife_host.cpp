// knife_host.cpp — TEST INFRASTRUCTURE: alicevision_amd/csrc/avdm_knife.h (the knife-edge border test the similarity kernels run on the device)
// compiled for the HOST, text unchanged, so that tests/test_oracle.py can hold it to the oracle's literal evaluation voxel by voxel without a GPU.
//   g++ -O2 -ffp-contract=off -shared -fPIC -I include tests/native/knife_host.cpp -o <tmp>/libknife_host.so
#include <math.h>

#include "avdm.h"

namespace avdm {
#include "../../alicevision_amd/csrc/avdm_knife.h"
}

extern "C" {
// one byte per (pixel, plane): 1 = the reference's R-side border test passes.  Pixel coordinates are the stage's image coordinates.
void knife_sgm_mask(const avdm_camera_t* rc, const float* xs, const float* ys, int nPix, const float* depths, int nPlanes, float dd, float W1, float H1,
                    unsigned char* out)
{
    for(int i = 0; i < nPix; ++i)
        for(int z = 0; z < nPlanes; ++z)
            out[(long)i * nPlanes + z] = avdm::knife::sgm_r_inside(rc->P, rc->iP, rc->C, rc->ZVect, xs[i], ys[i], depths[z], dd, W1, H1) ? 1 : 0;
}
void knife_refine_mask(const avdm_camera_t* rc, const float* xs, const float* ys, const float* depth, const float* pixSize, int nPix, int nPlanes, float dd,
                       float W1, float H1, unsigned char* out)
{
    for(int i = 0; i < nPix; ++i)
        for(int z = 0; z < nPlanes; ++z)
            out[(long)i * nPlanes + z] =
              avdm::knife::refine_r_inside(rc->P, rc->iP, rc->C, xs[i], ys[i], depth[i], pixSize[i], z - (nPlanes - 1) / 2, dd, W1, H1) ? 1 : 0;
}
}
