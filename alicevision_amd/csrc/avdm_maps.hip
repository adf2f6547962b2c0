// avdm_maps.hip — streaming volume helpers, Refine sub-sample arg-min and the depth/similarity-map kernels for gfx950.
//   avdm_volume_initialize_u8/f16, _add_f16, _update_uninitialized  <-> deviceSimilarityVolume.cu:57-153 (kernels.cuh:48-107)
//   avdm_volume_refine_best_depth                                  <-> cuda_volumeRefineBestDepth (.cu:469-502, kernels.cuh:515-594)
//   avdm_depth_sim_map_* / avdm_compute_sgm_upscaled_*             <-> planeSweeping/deviceDepthSimilarityMap.cu + ...MapKernels.cuh
// Compiled with -ffp-contract=off (operation order of the reference kept; see DESIGN.md "parity classes").
#include "avdm_device.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>
#include <vector>

namespace avdm {

// ---------------------------------------------------------------------------------------------
// streaming volume kernels: z-fastest rows of `rowBytes` bytes, 16 B per lane
// ---------------------------------------------------------------------------------------------
// Each (x, y) column is a contiguous run of dimZ elements; rows of a y-slice are pitch_x apart.
template <typename F>
__device__ __forceinline__ void for_each_word(long long pitch_y, int pitch_x, int dimX, int dimY, int wordsPerCol, F f)
{
    // grid-stride over (y, x, word)
    const long long total = (long long)dimX * dimY * wordsPerCol;
    for(long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x)
    {
        const int w = (int)(i % wordsPerCol);
        const long long c = i / wordsPerCol;
        const int x = (int)(c % dimX);
        const int y = (int)(c / dimX);
        f((long long)y * pitch_y + (long long)x * pitch_x, w);
    }
}

__global__ void __launch_bounds__(256) volume_init_u8_kernel(uint8_t* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, unsigned v4)
{
    const int words = (dimZ + 3) / 4;
    for_each_word(pitch_y, pitch_x, dimX, dimY, words, [&](long long off, int w) {
        if(4 * w + 3 < dimZ)
            *reinterpret_cast<unsigned*>(vol + off + 4 * w) = v4;
        else
            for(int j = 4 * w; j < dimZ; ++j)
                vol[off + j] = (uint8_t)v4;
    });
}

__global__ void __launch_bounds__(256)
  volume_init_f16_kernel(__half* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, unsigned short h)
{
    for_each_word(pitch_y, pitch_x, dimX, dimY, dimZ, [&](long long off, int z) { *reinterpret_cast<unsigned short*>((char*)vol + off + 2 * z) = h; });
}

__global__ void __launch_bounds__(256)
  volume_add_f16_kernel(__half* inout, const __half* in, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ)
{
    for_each_word(pitch_y, pitch_x, dimX, dimY, dimZ, [&](long long off, int z) {
        __half* p = (__half*)((char*)inout + off) + z;
        const __half* q = (const __half*)((const char*)in + off) + z;
        *p = __float2half(__half2float(*p) + __half2float(*q));
    });
}

__global__ void __launch_bounds__(256)
  volume_update_uninit_kernel(const uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ)
{
    const int words = (dimZ + 3) / 4;
    for_each_word(pitch_y, pitch_x, dimX, dimY, words, [&](long long off, int w) {
        if(4 * w + 3 < dimZ)
        {
            const unsigned b = *reinterpret_cast<const unsigned*>(best + off + 4 * w);
            unsigned s = *reinterpret_cast<const unsigned*>(second + off + 4 * w);
            unsigned r = 0;
#pragma unroll
            for(int j = 0; j < 4; ++j)
            {
                const unsigned sj = (s >> (8 * j)) & 0xffu, bj = (b >> (8 * j)) & 0xffu;
                r |= (sj >= 255u ? bj : sj) << (8 * j);
            }
            *reinterpret_cast<unsigned*>(second + off + 4 * w) = r;
        }
        else
            for(int j = 4 * w; j < dimZ; ++j)
                if(second[off + j] >= 255)
                    second[off + j] = best[off + j];
    });
}

// ---------------------------------------------------------------------------------------------
// Refine best depth: sliding Gaussian over 2*halfNbSamples+1 sub-samples (kernels.cuh:515-594)
// The weight depends only on the integer |zs - sample| -> table in LDS (double-precision exp rounded to fp32 == libm expf).
// ---------------------------------------------------------------------------------------------
#define RBD_MAXZ 64
template <int NZ>
__global__ void __launch_bounds__(256)
  refine_best_depth_kernel(float2* out, int out_pitch, const float2* __restrict__ sgmDepthPixSize, int map_pitch, const __half* __restrict__ vol,
                           long long pitch_y, int pitch_x, int volDimZ, int samplesPerPixSize, int halfNbSamples, int halfNbDepths,
                           float twoTimesSigmaPowerTwo, int tableSize, avdm_roi_t roi)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gauss = reinterpret_cast<float*>(smem); // gauss[d] = expf(-(d*d) / tt), d = |zs - sample|
    for(int d = threadIdx.x; d < tableSize; d += blockDim.x)
    {
        const float arg = -(float)(d * d) / twoTimesSigmaPowerTwo;
        gauss[d] = (float)exp((double)arg);
    }
    __syncthreads();

    const unsigned vx = blockIdx.x * 64 + (threadIdx.x & 63);
    const unsigned vy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(vx >= roi.x.end - roi.x.begin || vy >= roi.y.end - roi.y.begin)
        return;
    const float2 dps = *((const float2*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + vx);
    float2* o = (float2*)((char*)out + (long long)vy * out_pitch) + vx;
    if(dps.x <= 0.0f)
    {
        *o = make_float2(dps.x, 1.0f);
        return;
    }
    const __half* v = (const __half*)((const char*)vol + (long long)vy * pitch_y + (long long)vx * pitch_x);
    float simSum[NZ];
#pragma unroll
    for(int c = 0; c < NZ / 8; ++c)
    {
        if(8 * c < volDimZ)
        {
            const uint4 q = *reinterpret_cast<const uint4*>(v + 8 * c);
            const __half* hq = reinterpret_cast<const __half*>(&q);
#pragma unroll
            for(int j = 0; j < 8; ++j)
                simSum[8 * c + j] = -__half2float(hq[j]);
        }
    }

    float bestSampleSim = 0.f;
    int bestSampleOffsetIndex = 0;
    for(int sample = -halfNbSamples; sample <= halfNbSamples; ++sample)
    {
        float sampleSim = 0.f;
#pragma unroll
        for(int vz = 0; vz < NZ; ++vz)
        {
            if(vz < volDimZ)
            {
                const int zs = (vz - halfNbDepths) * samplesPerPixSize;
                const int d = zs - sample;
                sampleSim += simSum[vz] * gauss[d < 0 ? -d : d];
            }
        }
        if(sampleSim < bestSampleSim)
        {
            bestSampleOffsetIndex = sample;
            bestSampleSim = sampleSim;
        }
    }
    const float sampleSize = dps.y / (float)samplesPerPixSize;
    const float sampleSizeOffset = (float)bestSampleOffsetIndex * sampleSize;
    *o = make_float2(dps.x + sampleSizeOffset, bestSampleSim);
}

// Same computation with the Gaussian table as a kernel argument.  (sample, plane) is uniform over the wave: the weights of 8 consecutive
// sub-samples of a plane are 8 consecutive entries of the mirrored table gm[j] = gauss[|j - off|], j = sample - zs + off, fetched with
// scalar loads from the kernarg segment, and each term is a VALU multiply + add with an SGPR operand — no LDS traffic.  Every sample's
// sum still runs over the planes in ascending order (same floats as the one-sample-at-a-time loop); planes past volDimZ carry a zero
// weight-free term (x + 0 * g == x).  The table is filled on the host with the reference's expression (expf, kernels.cuh:566-570).
#define RBD_TAB 640
#define RBD_GROUP 8
struct GaussTable
{
    float g[RBD_TAB];
};
template <int NZ>
__global__ void __launch_bounds__(256)
  refine_best_depth_ktab_kernel(float2* out, int out_pitch, const float2* __restrict__ sgmDepthPixSize, int map_pitch, const __half* __restrict__ vol,
                                long long pitch_y, int pitch_x, int volDimZ, int samplesPerPixSize, int halfNbSamples, int halfNbDepths, int off,
                                const GaussTable gm, avdm_roi_t roi)
{
    const unsigned vx = blockIdx.x * 64 + (threadIdx.x & 63);
    const unsigned vy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(vx >= roi.x.end - roi.x.begin || vy >= roi.y.end - roi.y.begin)
        return;
    const float2 dps = *((const float2*)((const char*)sgmDepthPixSize + (long long)vy * map_pitch) + vx);
    float2* o = (float2*)((char*)out + (long long)vy * out_pitch) + vx;
    if(dps.x <= 0.0f)
    {
        *o = make_float2(dps.x, 1.0f);
        return;
    }
    const __half* v = (const __half*)((const char*)vol + (long long)vy * pitch_y + (long long)vx * pitch_x);
    float simSum[NZ];
#pragma unroll
    for(int c = 0; c < NZ / 8; ++c)
    {
        uint4 q = make_uint4(0u, 0u, 0u, 0u);
        if(8 * c < volDimZ)
            q = *reinterpret_cast<const uint4*>(v + 8 * c);
        const __half* hq = reinterpret_cast<const __half*>(&q);
#pragma unroll
        for(int j = 0; j < 8; ++j)
            simSum[8 * c + j] = (8 * c + j < volDimZ) ? -__half2float(hq[j]) : 0.0f;
    }

    float bestSampleSim = 0.f;
    int bestSampleOffsetIndex = 0;
#pragma unroll 1
    for(int s0 = -halfNbSamples; s0 <= halfNbSamples; s0 += RBD_GROUP)
    {
        float acc[RBD_GROUP];
#pragma unroll
        for(int t = 0; t < RBD_GROUP; ++t)
            acc[t] = 0.f;
#pragma unroll
        for(int vz = 0; vz < NZ; ++vz)
        {
            // planes past the volume: any in-range index, their simSum is 0
            const int zs = ((vz < volDimZ ? vz : 0) - halfNbDepths) * samplesPerPixSize;
            const float* g = gm.g + (s0 - zs + off);
#pragma unroll
            for(int t = 0; t < RBD_GROUP; ++t)
                acc[t] += simSum[vz] * g[t];
        }
#pragma unroll
        for(int t = 0; t < RBD_GROUP; ++t)
        {
            if(s0 + t <= halfNbSamples && acc[t] < bestSampleSim)
            {
                bestSampleOffsetIndex = s0 + t;
                bestSampleSim = acc[t];
            }
        }
    }
    const float sampleSize = dps.y / (float)samplesPerPixSize;
    const float sampleSizeOffset = (float)bestSampleOffsetIndex * sampleSize;
    *o = make_float2(dps.x + sampleSizeOffset, bestSampleSim);
}

// ---------------------------------------------------------------------------------------------
// depth/sim map kernels
// ---------------------------------------------------------------------------------------------
#define MAP_XY()                                                                                                                                      \
    const unsigned roiX = blockIdx.x * 64 + (threadIdx.x & 63);                                                                                       \
    const unsigned roiY = blockIdx.y * 4 + (threadIdx.x >> 6);

__global__ void __launch_bounds__(256)
  copy_depth_only_kernel(float2* out, int out_pitch, const float2* in, int in_pitch, unsigned width, unsigned height, float defaultSim)
{
    MAP_XY();
    if(roiX >= width || roiY >= height)
        return;
    const float d = ((const float2*)((const char*)in + (long long)roiY * in_pitch) + roiX)->x;
    *((float2*)((char*)out + (long long)roiY * out_pitch) + roiX) = make_float2(d, defaultSim);
}

__global__ void __launch_bounds__(256) normal_upscale_kernel(float* out, int out_pitch, const float* in, int in_pitch, float ratio, avdm_roi_t roi)
{
    MAP_XY();
    const unsigned roiW = roi.x.end - roi.x.begin, roiH = roi.y.end - roi.y.begin;
    if(roiX >= roiW || roiY >= roiH)
        return;
    const float ox = ((float)roiX - 0.5f) * ratio;
    const float oy = ((float)roiY - 0.5f) * ratio;
    const int xp = min((int)floor((double)ox + 0.5), (int)((float)roiW * ratio) - 1);
    const int yp = min((int)floor((double)oy + 0.5), (int)((float)roiH * ratio) - 1);
    const float* s = (const float*)((const char*)in + (long long)yp * in_pitch) + 3 * xp;
    float* d = (float*)((char*)out + (long long)roiY * out_pitch) + 3 * roiX;
    d[0] = s[0];
    d[1] = s[1];
    d[2] = s[2];
}

// depthSimMapComputeNormal_kernel (mapKernels.cuh:393-477, wsh = 3) + cuda_stat3d (cuda/device/eig33.cuh:351-445): plane fit by PCA of
// the 7 x 7 neighbourhood's 3-D points, normal = eigenvector of the smallest eigenvalue of the (double precision) covariance matrix,
// oriented towards the camera.  The reference solves the symmetric 3 x 3 problem with tred2 / tql2; any accurate solver yields the same
// vector up to rounding: here cyclic Jacobi rotations in double (tolerance class).  Neighbours outside the ROI are skipped
// on all four sides — the reference only checks the lower bounds and reads whatever the allocated map holds beyond the tile.
__device__ __forceinline__ void jacobi_rotate(double (&A)[3][3], double (&V)[3][3], int p, int q)
{
    if(fabs(A[p][q]) < 1e-300)
        return;
    const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
    for(int k = 0; k < 3; ++k)
    {
        const double akp = A[k][p], akq = A[k][q];
        A[k][p] = c * akp - sn * akq;
        A[k][q] = sn * akp + c * akq;
    }
#pragma unroll
    for(int k = 0; k < 3; ++k)
    {
        const double apk = A[p][k], aqk = A[q][k];
        A[p][k] = c * apk - sn * aqk;
        A[q][k] = sn * apk + c * aqk;
    }
#pragma unroll
    for(int k = 0; k < 3; ++k)
    {
        const double vkp = V[k][p], vkq = V[k][q];
        V[k][p] = c * vkp - sn * vkq;
        V[k][q] = sn * vkp + c * vkq;
    }
}

__global__ void __launch_bounds__(256)
  compute_normal_kernel(float* out, int out_pitch, const float2* __restrict__ depthSim, int in_pitch, avdm_camera_t rc, int stepXY, avdm_roi_t roi)
{
    MAP_XY();
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
    if((int)roiX >= roiW || (int)roiY >= roiH)
        return;
    float* o = (float*)((char*)out + (long long)roiY * out_pitch) + 3 * roiX;
    const unsigned x = (roi.x.begin + roiX) * (unsigned)stepXY, y = (roi.y.begin + roiY) * (unsigned)stepXY;
    const float in_depth = ((const float2*)((const char*)depthSim + (long long)roiY * in_pitch))[roiX].x;
    if(in_depth <= 0.0f)
    {
        o[0] = o[1] = o[2] = -1.f;
        return;
    }
    const f3 p = get3DPointForPixelAndDepthFromRC(rc, (float)x, (float)y, in_depth);
    const float pixSize = size(p - get3DPointForPixelAndDepthFromRC(rc, (float)(x + 1), (float)y, in_depth));
    double xs = 0, ys = 0, zs = 0, xx = 0, yy = 0, zz = 0, xy = 0, xz = 0, yz = 0, count = 0;
    for(int yp = -3; yp <= 3; ++yp)
    {
        const int ry = (int)roiY + yp;
        if(ry < 0 || ry >= roiH)
            continue;
        for(int xp = -3; xp <= 3; ++xp)
        {
            const int rx = (int)roiX + xp;
            if(rx < 0 || rx >= roiW)
                continue;
            const float depthP = ((const float2*)((const char*)depthSim + (long long)ry * in_pitch))[rx].x;
            if((depthP > 0.0f) && (fabsf(depthP - in_depth) < 30.0f * pixSize))
            {
                const f3 q = get3DPointForPixelAndDepthFromRC(rc, (float)((int)x + xp), (float)((int)y + yp), depthP);
                xx += (double)q.x * (double)q.x;
                yy += (double)q.y * (double)q.y;
                zz += (double)q.z * (double)q.z;
                xy += (double)q.x * (double)q.y;
                xz += (double)q.x * (double)q.z;
                yz += (double)q.y * (double)q.z;
                xs += (double)q.x;
                ys += (double)q.y;
                zs += (double)q.z;
                count += 1.0;
            }
        }
    }
    if(count < 3.0)
    {
        o[0] = o[1] = o[2] = -1.f;
        return;
    }
    const double xm = xs / count, ym = ys / count, zm = zs / count;
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    A[0][0] = (xx - xs * xm - xs * xm + xm * xm * count) / count;
    A[0][1] = A[1][0] = (xy - ys * xm - xs * ym + xm * ym * count) / count;
    A[0][2] = A[2][0] = (xz - zs * xm - xs * zm + xm * zm * count) / count;
    A[1][1] = (yy - ys * ym - ys * ym + ym * ym * count) / count;
    A[1][2] = A[2][1] = (yz - zs * ym - ys * zm + ym * zm * count) / count;
    A[2][2] = (zz - zs * zm - zs * zm + zm * zm * count) / count;
    for(int sweep = 0; sweep < 8; ++sweep)
    {
        jacobi_rotate(A, V, 0, 1);
        jacobi_rotate(A, V, 0, 2);
        jacobi_rotate(A, V, 1, 2);
    }
    int k = 0;
    if(A[1][1] < A[k][k])
        k = 1;
    if(A[2][2] < A[k][k])
        k = 2;
    f3 nn = normalize(f3{(float)V[0][k], (float)V[1][k], (float)V[2][k]});
    const f3 pp = f3{(float)xm, (float)ym, (float)zm};
    const f3 nc = normalize(ld3(rc.C) - p);
    if((dot(pp + nn, nc) - dot(pp, nc)) < 0.0f)
        nn = f3{-nn.x, -nn.y, -nn.z};
    o[0] = nn.x;
    o[1] = nn.y;
    o[2] = nn.z;
}

__global__ void __launch_bounds__(256)
  smooth_thickness_kernel(float2* map, int pitch, float minThicknessInflate, float maxThicknessInflate, avdm_roi_t roi)
{
    MAP_XY();
    const int roiW = (int)(roi.x.end - roi.x.begin), roiH = (int)(roi.y.end - roi.y.begin);
    if((int)roiX >= roiW || (int)roiY >= roiH)
        return;
    float2* dtp = (float2*)((char*)map + (long long)roiY * pitch) + roiX;
    const float2 dt = *dtp;
    if(dt.x <= 0.0f)
        return;
    const float minThickness = minThicknessInflate * dt.y;
    const float maxThickness = maxThicknessInflate * dt.y;
    float sumCenterDepthDist = 0.f;
    int nbValidPatchPixels = 0;
#pragma unroll
    for(int yp = -1; yp <= 1; ++yp)
#pragma unroll
        for(int xp = -1; xp <= 1; ++xp)
        {
            const int roiXp = (int)roiX + xp, roiYp = (int)roiY + yp;
            if((xp == 0 && yp == 0) || roiXp < 0 || roiXp >= roiW || roiYp < 0 || roiYp >= roiH)
                continue;
            // only .x of the neighbours is read and only .y of the own pixel is written: no hazard (mapKernels.cuh:151-211)
            const float pd = ((const float2*)((const char*)map + (long long)roiYp * pitch) + roiXp)->x;
            if(pd > 0.0f)
            {
                const float depthDistance = fabsf(dt.x - pd);
                sumCenterDepthDist += fmaxf(minThickness, fminf(maxThickness, depthDistance));
                ++nbValidPatchPixels;
            }
        }
    if(nbValidPatchPixels < 3)
        return;
    dtp->y = sumCenterDepthDist / (float)nbValidPatchPixels;
}

template <bool FIXED8>
__global__ void __launch_bounds__(256) upscale_depth_pixsize_kernel(float2* out, int out_pitch, const float2* in, int in_pitch, TexLod L, float sxN,
                                                                    float syN, int stepXY, int halfNbDepths, float ratio, int bilinear, avdm_roi_t roi)
{
    MAP_XY();
    const unsigned roiW = roi.x.end - roi.x.begin, roiH = roi.y.end - roi.y.begin;
    if(roiX >= roiW || roiY >= roiH)
        return;
    const unsigned x = (roi.x.begin + roiX) * (unsigned)stepXY;
    const unsigned y = (roi.y.begin + roiY) * (unsigned)stepXY;
    float2* o = (float2*)((char*)out + (long long)roiY * out_pitch) + roiX;
    // sxN = nominal level width (getDimensions), syN = height
    const float alpha = tex2D_lod<FIXED8>(L, ((float)x + 0.5f) / sxN, ((float)y + 0.5f) / syN).w;
    const float oy = ((float)roiY - 0.5f) * ratio;
    const float ox = ((float)roiX - 0.5f) * ratio;
    float2 dT;
    if(!bilinear)
    {
        if(alpha < 0.9f) // sic (mapKernels.cuh:238)
        {
            *o = make_float2(-2.f, 0.f);
            return;
        }
        int xp = (int)floor((double)ox + 0.5);
        int yp = (int)floor((double)oy + 0.5);
        xp = min(xp, (int)((float)roiW * ratio) - 1);
        yp = min(yp, (int)((float)roiH * ratio) - 1);
        dT = *((const float2*)((const char*)in + (long long)yp * in_pitch) + xp);
    }
    else
    {
        if(alpha < (255.f * 0.9f))
        {
            *o = make_float2(-2.f, 0.f);
            return;
        }
        int xp = (int)floorf(ox);
        int yp = (int)floorf(oy);
        xp = max(min(xp, (int)((float)roiW * ratio) - 2), 0); // max(.., 0): deviation, the reference reads out of bounds at roiX == 0
        yp = max(min(yp, (int)((float)roiH * ratio) - 2), 0);
        const float2 lu = *((const float2*)((const char*)in + (long long)yp * in_pitch) + xp);
        const float2 ru = *((const float2*)((const char*)in + (long long)yp * in_pitch) + xp + 1);
        const float2 rd = *((const float2*)((const char*)in + (long long)(yp + 1) * in_pitch) + xp + 1);
        const float2 ld = *((const float2*)((const char*)in + (long long)(yp + 1) * in_pitch) + xp);
        if(lu.x <= 0.0f || ru.x <= 0.0f || rd.x <= 0.0f || ld.x <= 0.0f)
        {
            float sx = 0.f, sy = 0.f;
            int count = 0;
            if(lu.x > 0.0f) { sx = sx + lu.x; sy = sy + lu.y; ++count; }
            if(ru.x > 0.0f) { sx = sx + ru.x; sy = sy + ru.y; ++count; }
            if(rd.x > 0.0f) { sx = sx + rd.x; sy = sy + rd.y; ++count; }
            if(ld.x > 0.0f) { sx = sx + ld.x; sy = sy + ld.y; ++count; }
            if(count != 0)
                dT = make_float2(sx / (float)count, sy / (float)count);
            else
            {
                *o = make_float2(-1.0f, 1.0f);
                return;
            }
        }
        else
        {
            const float ui = ox - (float)xp;
            const float vi = oy - (float)yp;
            const float ux = lu.x + (ru.x - lu.x) * ui, uy = lu.y + (ru.y - lu.y) * ui;
            const float dx = ld.x + (rd.x - ld.x) * ui, dy = ld.y + (rd.y - ld.y) * ui;
            dT = make_float2(ux + (dx - ux) * vi, uy + (dy - uy) * vi);
        }
    }
    *o = make_float2(dT.x, dT.y / (float)halfNbDepths);
}

// ---- colour-guided optimisation (mapKernels.cuh:25-101, 479-608; Map.cu:193-263) ----
template <bool FIXED8>
__global__ void __launch_bounds__(256)
  var_L_kernel(float* out, int out_pitch, TexLod L, float wN, float hN, int stepXY, avdm_roi_t roi)
{
    MAP_XY();
    if(roiX >= roi.x.end - roi.x.begin || roiY >= roi.y.end - roi.y.begin)
        return;
    const float x = (float)(roi.x.begin + roiX) * (float)stepXY;
    const float y = (float)(roi.y.begin + roiY) * (float)stepXY;
    const float iw = 1.f / wN, ih = 1.f / hN;
    const float xM1 = tex2D_lod<FIXED8>(L, ((x - 1.f) + 0.5f) * iw, ((y + 0.f) + 0.5f) * ih).x;
    const float xP1 = tex2D_lod<FIXED8>(L, ((x + 1.f) + 0.5f) * iw, ((y + 0.f) + 0.5f) * ih).x;
    const float yM1 = tex2D_lod<FIXED8>(L, ((x + 0.f) + 0.5f) * iw, ((y - 1.f) + 0.5f) * ih).x;
    const float yP1 = tex2D_lod<FIXED8>(L, ((x + 0.f) + 0.5f) * iw, ((y + 1.f) + 0.5f) * ih).x;
    const float gx = xM1 - xP1, gy = yM1 - yP1;
    *((float*)((char*)out + (long long)roiY * out_pitch) + roiX) = sqrtf(gx * gx + gy * gy);
}

__global__ void __launch_bounds__(256) copy_rows_kernel(float2* out, int out_pitch, const float2* in, int in_pitch, unsigned w, unsigned h)
{
    MAP_XY();
    if(roiX >= w || roiY >= h)
        return;
    *((float2*)((char*)out + (long long)roiY * out_pitch) + roiX) = *((const float2*)((const char*)in + (long long)roiY * in_pitch) + roiX);
}

__global__ void __launch_bounds__(256) extract_depth_kernel(float* tmp, int tmp_pitch, const float2* opt, int opt_pitch, avdm_roi_t roi)
{
    MAP_XY();
    if(roiX >= roi.x.end - roi.x.begin || roiY >= roi.y.end - roi.y.begin)
        return;
    *((float*)((char*)tmp + (long long)roiY * tmp_pitch) + roiX) = ((const float2*)((const char*)opt + (long long)roiY * opt_pitch) + roiX)->x;
}

__device__ __forceinline__ float tex_point(const float* buf, int pitch, int W, int H, int x, int y)
{
    x = min(max(x, 0), W - 1);
    y = min(max(y, 0), H - 1);
    return *((const float*)((const char*)buf + (long long)y * pitch) + x);
}

// AVDM_OPT_FAST=1 (compile time, the default since round 4; 0 = the IEEE form of rounds 1-3 for an A/B): the colour
// optimisation's IEEE divisions and square roots — 24 + 14 per pixel and iteration, ~10 instructions each, more than half of the kernel's 943
// VALU instructions — as the hardware's v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 (1 ulp) and its four sigmoids through v_exp_f32: 440 instructions,
// 20.7 -> 14 ms per 12 MP depth map (profiles/r04_e_ab.txt).  The two unit vectors of the smoothness angle take the fast reciprocal square
// root too (normalize_exact; keeping them IEEE was measured and dropped: see angleBetwABandAC).  Both forms of the stage (point map / depth
// map) use the same helpers, so they stay bit-identical to each other; against the oracle the stage is in the tolerance class either way
// (acosf / expf of another library): |d sim| max 1.43e-2 with the fast forms (session r04_g; 1.2e-2 with the IEEE forms), asserted < 2e-2.
#ifndef AVDM_OPT_FAST
#define AVDM_OPT_FAST 1
#endif
__device__ __forceinline__ float opt_rcp(float x) { return AVDM_OPT_FAST ? __builtin_amdgcn_rcpf(x) : 1.0f / x; }
__device__ __forceinline__ float opt_div(float a, float b) { return AVDM_OPT_FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
__device__ __forceinline__ float opt_size(f3 a) { return AVDM_OPT_FAST ? __builtin_amdgcn_sqrtf(dot(a, a)) : size(a); }
// sigmoid / sigmoid2 of avdm_device.h (matrix.cuh:334-346) with a constant width: 1 / (1 + exp(10 (x - mid) / width))
__device__ __forceinline__ float opt_sigmoid(float zeroVal, float endVal, float sigwidth, float sigMid, float xval)
{
#if AVDM_OPT_FAST
    const float k = 10.0f * 1.44269504088896340736f / sigwidth; // folded at compile time: the widths are literals
    return zeroVal + (endVal - zeroVal) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(k * (xval - sigMid)));
#else
    return sigmoid(zeroVal, endVal, sigwidth, sigMid, xval);
#endif
}
__device__ __forceinline__ float opt_sigmoid2(float zeroVal, float endVal, float sigwidth, float sigMid, float xval)
{
#if AVDM_OPT_FAST
    const float k = 10.0f * 1.44269504088896340736f / sigwidth;
    return zeroVal + (endVal - zeroVal) * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(k * (sigMid - xval)));
#else
    return sigmoid2(zeroVal, endVal, sigwidth, sigMid, xval);
#endif
}
__device__ __forceinline__ f3 normalize_ieee(f3 a)
{
    const float dInv = 1.0f / sqrtf(dot(a, a));
    return f3{a.x * dInv, a.y * dInv, a.z * dInv};
}
__device__ __forceinline__ f3 normalize_exact(f3 a)
{
#if AVDM_OPT_FAST
    const float dInv = __builtin_amdgcn_rsqf(dot(a, a));
    return f3{a.x * dInv, a.y * dInv, a.z * dInv};
#else
    return normalize_ieee(a);
#endif
}
__device__ __forceinline__ f3 point_at_depth(const avdm_camera_t& cam, float px, float py, float depth)
{
    const f3 rpv = normalize_exact(M3x3mulV2(cam.iP, px, py));
    return ld3(cam.C) + rpv * depth;
}
__device__ __forceinline__ float angleBetwABandAC(f3 A, f3 B, f3 C)
{
    // (measured, r04_f: IEEE unit vectors here — normalize_ieee — cost 2.4 ms per 12 MP depth map and do not bring |d sim| max against the
    // oracle back under 1e-2: for a neighbourhood 0.5 degrees from flat the last bit of the two vectors moves the energy by 8e-4 degrees, 4e-5 in the
    // similarity; for a flatter one the fp32 dot product itself cannot resolve 1 - |cos| and the energy is rounding noise in ANY fp32
    // evaluation, the reference's included)
    const f3 V1 = normalize_exact(B - A);
    const f3 V2 = normalize_exact(C - A);
    // The reference evaluates acos in double precision (matrix.cuh:306-322: the fp32 dot product converted, `acos`, then back to float).
    // Two fp64 acos per pixel and iteration were ~60 % of this kernel's instruction slots (fp64 runs at half rate); the single-precision
    // acos of the device library is within 2 ulp of the same value — 4e-5 degrees at 180 — which moves the energy of a flat neighbourhood
    // (180 - angle, a fraction of a degree) by 1e-4 relative and the optimised depth by far less than the parity class of this stage
    // allows (tests/test_gpu_parity.py::test_optimize_parity: depth RMSE < 1e-3 pixSize and |d sim| < 1e-2 against the double-precision
    // oracle).  AVDM_OPT_ACOS_F64 restores the double evaluation for A/B.
    const float xf = V1.x * V2.x + V1.y * V2.y + V1.z * V2.z;
#ifdef AVDM_OPT_ACOS_F64
    double a = acos((double)xf);
    a = isinf(a) ? 0.0 : a;
    return (float)(fabs(a) / (3.14159265358979323846 / 180.0));
#else
    float a = acosf(xf);
    a = isinf(a) ? 0.0f : a;
    return fabsf(a) * 57.29577951308232f;
#endif
}

__global__ void __launch_bounds__(256)
  optimize_step_kernel(float2* outOpt, int out_pitch, const float2* __restrict__ sgmDepthPixSize, int sgm_pitch,
                       const float2* __restrict__ refineDepthSim, int ref_pitch, const float* __restrict__ imgVariance, int var_pitch,
                       const float* __restrict__ depthTex, int tex_pitch, int texW, int texH, avdm_camera_t rc, int iter, avdm_roi_t roi)
{
    MAP_XY();
    if(roiX >= roi.x.end - roi.x.begin || roiY >= roi.y.end - roi.y.begin)
        return;
    const float2 sgm = *((const float2*)((const char*)sgmDepthPixSize + (long long)roiY * sgm_pitch) + roiX);
    const float sgmDepth = sgm.x, sgmPixSize = sgm.y;
    const float2 rf = *((const float2*)((const char*)refineDepthSim + (long long)roiY * ref_pitch) + roiX);
    const float refineDepth = rf.x, refineSim = rf.y;
    float2* op = (float2*)((char*)outOpt + (long long)roiY * out_pitch) + roiX;
    float2 outDS = (iter == 0) ? make_float2(sgmDepth, refineSim) : *op;
    const float depthOpt = outDS.x;
    if(depthOpt > 0.0f)
    {
        // getCellSmoothStepEnergy
        float smoothStep = 0.0f, energy = 180.0f;
        const int cx = (int)roiX, cy = (int)roiY;
        const float d0 = tex_point(depthTex, tex_pitch, texW, texH, cx, cy);
        if(d0 > 0.0f)
        {
            const float offx = (float)roi.x.begin, offy = (float)roi.y.begin;
            // cellL = cell0 + (0,-1), cellR = cell0 + (0,1), cellU = cell0 + (-1,0), cellB = cell0 + (1,0)
            const float dL = tex_point(depthTex, tex_pitch, texW, texH, cx, cy - 1);
            const float dR = tex_point(depthTex, tex_pitch, texW, texH, cx, cy + 1);
            const float dU = tex_point(depthTex, tex_pitch, texW, texH, cx - 1, cy);
            const float dB = tex_point(depthTex, tex_pitch, texW, texH, cx + 1, cy);
            const float fx = (float)roiX, fy = (float)roiY;
            const f3 p0 = point_at_depth(rc, fx + offx, fy + offy, d0);
            const f3 pL = point_at_depth(rc, (fx + 0.f) + offx, (fy + -1.f) + offy, dL);
            const f3 pR = point_at_depth(rc, (fx + 0.f) + offx, (fy + 1.f) + offy, dR);
            const f3 pU = point_at_depth(rc, (fx + -1.f) + offx, (fy + 0.f) + offy, dU);
            const f3 pB = point_at_depth(rc, (fx + 1.f) + offx, (fy + 0.f) + offy, dB);
            f3 cg = f3{0.f, 0.f, 0.f};
            float n = 0.0f;
            if(dL > 0.0f) { cg = cg + pL; n++; }
            if(dR > 0.0f) { cg = cg + pR; n++; }
            if(dU > 0.0f) { cg = cg + pU; n++; }
            if(dB > 0.0f) { cg = cg + pB; n++; }
            if(n > 1.0f)
            {
                {
                    const float invN = opt_rcp(n); // (exact form: three divisions by n; x / n == x * (1 / n) only up to rounding, hence the switch)
                    cg = AVDM_OPT_FAST ? f3{cg.x * invN, cg.y * invN, cg.z * invN} : f3{cg.x / n, cg.y / n, cg.z / n};
                }
                const f3 vcn = normalize_exact(ld3(rc.C) - p0);
                const f3 pS = closestPointToLine3D(cg, p0, vcn);
                smoothStep = opt_size(ld3(rc.C) - pS) - d0;
            }
            float e = 0.0f;
            n = 0.0f;
            if(dL > 0.0f && dR > 0.0f)
            {
                e = fmaxf(e, (180.0f - angleBetwABandAC(p0, pL, pR)));
                n++;
            }
            if(dU > 0.0f && dB > 0.0f)
            {
                e = fmaxf(e, (180.0f - angleBetwABandAC(p0, pU, pB)));
                n++;
            }
            if(n > 0.0f)
                energy = e;
        }
        float stepToSmoothDepth = smoothStep;
        stepToSmoothDepth = copysignf(fminf(fabsf(stepToSmoothDepth), (AVDM_OPT_FAST ? sgmPixSize * 0.1f : sgmPixSize / 10.0f)), stepToSmoothDepth);
        const float depthEnergy = energy;
        float stepToFineDM = refineDepth - depthOpt;
        stepToFineDM = copysignf(fminf(fabsf(stepToFineDM), (AVDM_OPT_FAST ? sgmPixSize * 0.1f : sgmPixSize / 10.0f)), stepToFineDM);
        const float stepToRoughDM = sgmDepth - depthOpt;
        const float imgColorVariance = *((const float*)((const char*)imgVariance + (long long)roiY * var_pitch) + roiX);
        const float weightedColorVariance = opt_sigmoid2(5.0f, 30.0f, 40.0f, 20.0f, imgColorVariance);
        const float fineSimWeight = opt_sigmoid(0.0f, 1.0f, 0.7f, -0.7f, refineSim);
        const float energyLowerThanVarianceWeight = opt_sigmoid(0.0f, 1.0f, 30.0f, weightedColorVariance, depthEnergy);
        const float closeToRoughWeight = 1.0f - opt_sigmoid(0.0f, 1.0f, 10.0f, 17.0f, fabsf(opt_div(stepToRoughDM, sgmPixSize)));
        const float depthOptStep = closeToRoughWeight * stepToRoughDM +
                                   (1.0f - closeToRoughWeight) * (energyLowerThanVarianceWeight * fineSimWeight * stepToFineDM +
                                                                  (1.0f - energyLowerThanVarianceWeight) * stepToSmoothDepth);
        outDS.x = depthOpt + depthOptStep;
        outDS.y = (1.0f - closeToRoughWeight) *
                  (energyLowerThanVarianceWeight * fineSimWeight * refineSim + (1.0f - energyLowerThanVarianceWeight) * (AVDM_OPT_FAST ? depthEnergy * 0.05f : depthEnergy / 20.0f));
    }
    *op = outDS;
}

// ---- the same iteration over a POINT map ----------------------------------------------------------------------------------------
// optimize_step_kernel evaluates five pixel rays (3x3 product, IEEE square root and division) per pixel and iteration to turn its own
// depth and its four neighbours' depths into 3-D points.  A pixel's point only depends on its own ray and depth, and every neighbour
// computes it with the very same expression — so each pixel keeps {p.x, p.y, p.z, depth} in a float4 map: the owner evaluates its ray
// ONCE per iteration (for the depth it has just stepped to) and the four neighbours read the finished point.  Same operations on the same
// operands, hence the same bits as the depth-map form; the copy of the depth map per iteration (kernel 19 of the reference) becomes the
// ping-pong of two point maps, and the (depth, sim) map is only written by the last iteration (sim is recomputed from scratch by every
// iteration, mapKernels.cuh:596-604).  Neighbours outside the tile keep the reference's clamp semantics: the CLAMPED cell's depth on the
// UNCLAMPED pixel's ray (mapKernels.cuh:41-51 sample the depth texture with clamp addressing but build the point from the cell index).
__device__ __forceinline__ float4 opt_point(const avdm_camera_t& rc, float px, float py, float depth)
{
    const f3 p = point_at_depth(rc, px, py, depth);
    return make_float4(p.x, p.y, p.z, depth);
}

__global__ void __launch_bounds__(256)
  optimize_init_points_kernel(float4* __restrict__ pts, int pts_pitch, const float2* __restrict__ sgmDepthPixSize, int sgm_pitch, avdm_camera_t rc, avdm_roi_t roi)
{
    MAP_XY();
    if(roiX >= roi.x.end - roi.x.begin || roiY >= roi.y.end - roi.y.begin)
        return;
    const float d = ((const float2*)((const char*)sgmDepthPixSize + (long long)roiY * sgm_pitch) + roiX)->x;
    *((float4*)((char*)pts + (long long)roiY * pts_pitch) + roiX) = opt_point(rc, (float)roiX + (float)roi.x.begin, (float)roiY + (float)roi.y.begin, d);
}

template <bool LAST>
__global__ void __launch_bounds__(256)
  optimize_step_points_kernel(float4* __restrict__ ptsOut, const float4* __restrict__ ptsIn, int pts_pitch, float2* __restrict__ outOpt, int out_pitch,
                              const float2* __restrict__ sgmDepthPixSize, int sgm_pitch, const float2* __restrict__ refineDepthSim, int ref_pitch,
                              const float* __restrict__ imgVariance, int var_pitch, int texW, int texH, avdm_camera_t rc, avdm_roi_t roi)
{
    MAP_XY();
    if(roiX >= roi.x.end - roi.x.begin || roiY >= roi.y.end - roi.y.begin)
        return;
    const float2 sgm = *((const float2*)((const char*)sgmDepthPixSize + (long long)roiY * sgm_pitch) + roiX);
    const float sgmDepth = sgm.x, sgmPixSize = sgm.y;
    const float2 rf = *((const float2*)((const char*)refineDepthSim + (long long)roiY * ref_pitch) + roiX);
    const float refineDepth = rf.x, refineSim = rf.y;
    const int cx = (int)roiX, cy = (int)roiY;
    auto cell = [&](int x, int y) __attribute__((always_inline)) -> float4 {
        return *((const float4*)((const char*)ptsIn + (long long)y * pts_pitch) + x);
    };
    const float fx = (float)roiX, fy = (float)roiY;
    const float offx = (float)roi.x.begin, offy = (float)roi.y.begin;
    // a neighbour cell: its finished point, or — outside the depth texture — the clamped cell's depth on the unclamped pixel's ray
    auto neighbour = [&](int dx, int dy, f3& pt, float& d) __attribute__((always_inline)) {
        const int x = cx + dx, y = cy + dy;
        const int xc = min(max(x, 0), texW - 1), yc = min(max(y, 0), texH - 1);
        const float4 c = cell(xc, yc);
        d = c.w;
        if(xc == x && yc == y)
            pt = f3{c.x, c.y, c.z};
        else
            pt = point_at_depth(rc, (fx + (float)dx) + offx, (fy + (float)dy) + offy, d);
    };
    // my own state; the reference reads its own cell of the depth texture with clamp addressing too (a tile larger than the texture is
    // not a case the callers produce, but the semantics are kept)
    const float4 own = cell(cx, cy);
    float4 c0 = own;
    if(cx >= texW || cy >= texH)
        c0 = opt_point(rc, fx + offx, fy + offy, cell(min(cx, texW - 1), min(cy, texH - 1)).w);
    const float depthOpt = own.w;
    float2 outDS = make_float2(depthOpt, refineSim); // pixels that never move keep (sgmDepth, refineSim) of iteration 0
    float4 outPt = own;
    if(depthOpt > 0.0f)
    {
        float smoothStep = 0.0f, energy = 180.0f;
        const float d0 = c0.w;
        if(d0 > 0.0f)
        {
            const f3 p0 = f3{c0.x, c0.y, c0.z};
            f3 pL, pR, pU, pB;
            float dL, dR, dU, dB;
            neighbour(0, -1, pL, dL);
            neighbour(0, 1, pR, dR);
            neighbour(-1, 0, pU, dU);
            neighbour(1, 0, pB, dB);
            f3 cg = f3{0.f, 0.f, 0.f};
            float n = 0.0f;
            if(dL > 0.0f) { cg = cg + pL; n++; }
            if(dR > 0.0f) { cg = cg + pR; n++; }
            if(dU > 0.0f) { cg = cg + pU; n++; }
            if(dB > 0.0f) { cg = cg + pB; n++; }
            if(n > 1.0f)
            {
                {
                    const float invN = opt_rcp(n); // (exact form: three divisions by n; x / n == x * (1 / n) only up to rounding, hence the switch)
                    cg = AVDM_OPT_FAST ? f3{cg.x * invN, cg.y * invN, cg.z * invN} : f3{cg.x / n, cg.y / n, cg.z / n};
                }
                const f3 vcn = normalize_exact(ld3(rc.C) - p0);
                const f3 pS = closestPointToLine3D(cg, p0, vcn);
                smoothStep = opt_size(ld3(rc.C) - pS) - d0;
            }
            float e = 0.0f;
            n = 0.0f;
            if(dL > 0.0f && dR > 0.0f)
            {
                e = fmaxf(e, (180.0f - angleBetwABandAC(p0, pL, pR)));
                n++;
            }
            if(dU > 0.0f && dB > 0.0f)
            {
                e = fmaxf(e, (180.0f - angleBetwABandAC(p0, pU, pB)));
                n++;
            }
            if(n > 0.0f)
                energy = e;
        }
        float stepToSmoothDepth = smoothStep;
        stepToSmoothDepth = copysignf(fminf(fabsf(stepToSmoothDepth), (AVDM_OPT_FAST ? sgmPixSize * 0.1f : sgmPixSize / 10.0f)), stepToSmoothDepth);
        const float depthEnergy = energy;
        float stepToFineDM = refineDepth - depthOpt;
        stepToFineDM = copysignf(fminf(fabsf(stepToFineDM), (AVDM_OPT_FAST ? sgmPixSize * 0.1f : sgmPixSize / 10.0f)), stepToFineDM);
        const float stepToRoughDM = sgmDepth - depthOpt;
        const float imgColorVariance = *((const float*)((const char*)imgVariance + (long long)roiY * var_pitch) + roiX);
        const float weightedColorVariance = opt_sigmoid2(5.0f, 30.0f, 40.0f, 20.0f, imgColorVariance);
        const float fineSimWeight = opt_sigmoid(0.0f, 1.0f, 0.7f, -0.7f, refineSim);
        const float energyLowerThanVarianceWeight = opt_sigmoid(0.0f, 1.0f, 30.0f, weightedColorVariance, depthEnergy);
        const float closeToRoughWeight = 1.0f - opt_sigmoid(0.0f, 1.0f, 10.0f, 17.0f, fabsf(opt_div(stepToRoughDM, sgmPixSize)));
        const float depthOptStep = closeToRoughWeight * stepToRoughDM +
                                   (1.0f - closeToRoughWeight) * (energyLowerThanVarianceWeight * fineSimWeight * stepToFineDM +
                                                                  (1.0f - energyLowerThanVarianceWeight) * stepToSmoothDepth);
        outDS.x = depthOpt + depthOptStep;
        outDS.y = (1.0f - closeToRoughWeight) *
                  (energyLowerThanVarianceWeight * fineSimWeight * refineSim + (1.0f - energyLowerThanVarianceWeight) * (AVDM_OPT_FAST ? depthEnergy * 0.05f : depthEnergy / 20.0f));
        if(!LAST)
            outPt = opt_point(rc, fx + offx, fy + offy, outDS.x);
    }
    if(LAST)
        *((float2*)((char*)outOpt + (long long)roiY * out_pitch) + roiX) = outDS;
    else
        *((float4*)((char*)ptsOut + (long long)roiY * pts_pitch) + roiX) = outPt;
}

// ---- the --downscale resize of the input images (imageAlgo::resizeImage -> OpenImageIO's ImageBufAlgo::resize, default filter) ----------
// One lane per destination pixel, float4 = RGBA.  The tap tables (one row of normalised weights per destination column / row) are built on
// the host by avdm_image_resize below; the double loop keeps OpenImageIO's order — rows outer, columns inner, the product wy * wx formed
// first, zero products skipped — and this file is compiled without FMA contraction, so the sums round like the scalar code.
__global__ void __launch_bounds__(256)
  image_resize_kernel(float4* __restrict__ dst, int dst_pitch, int dstW, int dstH, const float4* __restrict__ src, int src_pitch, int srcW, int srcH,
                      const float* __restrict__ wx, const int* __restrict__ fx, int xtaps, const float* __restrict__ wy, const int* __restrict__ fy, int ytaps)
{
    MAP_XY();
    if(roiX >= (unsigned)dstW || roiY >= (unsigned)dstH)
        return;
    const float* xw = wx + (size_t)roiX * xtaps;
    const float* yw = wy + (size_t)roiY * ytaps;
    const int x0 = fx[roiX], y0 = fy[roiY];
    float4 pel = make_float4(0.f, 0.f, 0.f, 0.f);
    float totalx = 0.0f;
    for(int i = 0; i < xtaps; ++i)
        totalx += xw[i];
    if(totalx != 0.0f)
        for(int j = 0; j < ytaps; ++j)
        {
            const float wyj = yw[j];
            if(wyj == 0.0f)
                continue;
            const int sy = min(max(y0 + j, 0), srcH - 1);
            const float4* srow = (const float4*)((const char*)src + (long long)sy * src_pitch);
            for(int i = 0; i < xtaps; ++i)
            {
                const float w = wyj * xw[i];
                if(w != 0.0f)
                {
                    const float4 p = srow[min(max(x0 + i, 0), srcW - 1)];
                    pel.x += w * p.x;
                    pel.y += w * p.y;
                    pel.z += w * p.z;
                    pel.w += w * p.w;
                }
            }
        }
    *((float4*)((char*)dst + (long long)roiY * dst_pitch) + roiX) = pel;
}

// ---- undistortion of an input image (camera::UndistortImage, camera/cameraUndistortImage.hpp:81-139) -------------------------------------
// One lane per pixel of the undistorted image; the geometry in double like the reference (this file is compiled without contraction), the
// sample position converted to float before the sampler like its `operator()(src, float y, float x)`, the sampler's accumulation in double.
__device__ __forceinline__ void add_distortion(const avdm_intrinsic_t& c, double& x, double& y)
{
    // DistortionRadialK1 / K3 / K3PT::addDistortion (camera/DistortionRadial.cpp:18-24, 110-124, 262-277)
    double coeff = 1.0;
    if(c.distortion_model == AVDM_DISTORTION_RADIALK1)
    {
        const double r2 = x * x + y * y;
        coeff = (1. + c.k[0] * r2);
    }
    else if(c.distortion_model == AVDM_DISTORTION_RADIALK3 || c.distortion_model == AVDM_DISTORTION_RADIALK3PT)
    {
        const double r = sqrt(x * x + y * y);
        const double r2 = r * r;
        const double r4 = r2 * r2;
        const double r6 = r4 * r2;
        coeff = (1. + c.k[0] * r2 + c.k[1] * r4 + c.k[2] * r6);
        if(c.distortion_model == AVDM_DISTORTION_RADIALK3PT)
            coeff = coeff / (1.0 + c.k[0] + c.k[1] + c.k[2]);
    }
    x = x * coeff;
    y = y * coeff;
}

__global__ void __launch_bounds__(256)
  image_undistort_kernel(float4* __restrict__ dst, int dst_pitch, const float4* __restrict__ src, int src_pitch, avdm_intrinsic_t cam, float4 fill)
{
    MAP_XY();
    const int W = cam.width, H = cam.height;
    if(roiX >= (unsigned)W || roiY >= (unsigned)H)
        return;
    // getDistortedPixel(p) = cam2ima(addDistortion(ima2cam(p))) (IntrinsicScaleOffsetDisto.cpp:80; IntrinsicScaleOffset.cpp:31, 55-66)
    const double ppx = cam.offset_x + (double)W * 0.5, ppy = cam.offset_y + (double)H * 0.5;
    double cx = ((double)roiX - ppx) / cam.scale_x, cy = ((double)roiY - ppy) / cam.scale_y;
    add_distortion(cam, cx, cy);
    const double dxp = cx * cam.scale_x + ppx, dyp = cy * cam.scale_y + ppy;
    float4 out = fill;
    // imageIn.contains(disto_pix(1), disto_pix(0)): the doubles convert to int by truncation (image/Image.hpp:178)
    const int ix = (int)dxp, iy = (int)dyp;
    if(0 <= ix && ix < W && 0 <= iy && iy < H)
    {
        // Sampler2d<SamplerLinear>::operator()(src, float y, float x) (image/Sampler.hpp:391-474)
        const float x = (float)dxp, y = (float)dyp;
        const double fxl = floor((double)x), fyl = floor((double)y);
        const double dx = (double)x - fxl, dy = (double)y - fyl;
        const double coefsX[2] = {1.0 - dx, dx}, coefsY[2] = {1.0 - dy, dy};
        const int gridX = (int)fxl, gridY = (int)fyl;
        double r = 0.0, g = 0.0, b = 0.0, a = 0.0, totalWeight = 0.0;
        for(int i = 0; i < 2; ++i)
        {
            const int iCurrent = gridY + 1 + i - 1;
            if(iCurrent < 0 || iCurrent >= H)
                continue;
            const float4* row = (const float4*)((const char*)src + (long long)iCurrent * src_pitch);
            for(int j = 0; j < 2; ++j)
            {
                const int jCurrent = gridX + 1 + j - 1;
                if(jCurrent < 0 || jCurrent >= W)
                    continue;
                const double w = coefsX[j] * coefsY[i];
                const float4 p = row[jCurrent];
                r += (double)p.x * w;
                g += (double)p.y * w;
                b += (double)p.z * w;
                a += (double)p.w * w;
                totalWeight += w;
            }
        }
        if(totalWeight <= 0.2)
        {
            int row = (int)floor((double)y), col = (int)floor((double)x);
            row = row < 0 ? 0 : (row >= H ? H - 1 : row);
            col = col < 0 ? 0 : (col >= W ? W - 1 : col);
            out = *((const float4*)((const char*)src + (long long)row * src_pitch) + col);
        }
        else
        {
            if(totalWeight != 1.0)
            {
                r /= totalWeight;
                g /= totalWeight;
                b /= totalWeight;
                a /= totalWeight;
            }
            out = make_float4((float)r, (float)g, (float)b, (float)a);
        }
    }
    *((float4*)((char*)dst + (long long)roiY * dst_pitch) + roiX) = out;
}

// libutil/filter.cpp, FilterLanczos3_1D::lanczos3 (OpenImageIO 2.x): one sinf, sin(pi x) through the triple-angle identity
static float oiio_lanczos3(float x)
{
    const float a = 3.0f;
    const float ainv = 1.0f / a;
    const float m_pi = (float)3.14159265358979323846;
    x = fabsf(x);
    if(x > a)
        return 0.0f;
    if(x < 0.0001f)
        return 1.0f;
    const float s1 = sinf(x * ainv * m_pi);
    const float s3 = (-4.0f * s1 * s1 + 3.0f) * s1;
    return a / (x * x * (m_pi * m_pi)) * s3 * s1;
}
// imagebufalgo_xform.cpp, resize_(): the normalised tap weights of every destination pixel of one axis, and the source index of tap 0
static int oiio_resize_taps(int dstN, int srcN, std::vector<float>& weights, std::vector<int>& first)
{
    const float srcf = (float)srcN, dstf = (float)dstN;
    const float ratio = dstf / srcf;
    const float dstpixel = 1.0f / dstf;
    const float width = 6.0f * std::max(1.0f, ratio); // get_resize_filter(): fd.width * max(1, ratio)
    const float filterrad = width / 2.0f;
    const float wscale = 6.0f / width;                // FilterLanczos3_2D::m_wscale
    const int rad = (int)ceilf(filterrad / ratio);
    const int taps = 2 * rad + 1;
    weights.resize((size_t)taps * dstN);
    first.resize(dstN);
    for(int d = 0; d < dstN; ++d)
    {
        const float s = ((float)d - 0.0f + 0.5f) * dstpixel;
        const float src_f = 0.0f + s * srcf;
        const float fl = floorf(src_f);
        const float frac = src_f - fl;
        float total = 0.0f;
        float* w = weights.data() + (size_t)d * taps;
        for(int i = 0; i < taps; ++i)
        {
            w[i] = oiio_lanczos3((ratio * ((float)(i - rad) - (frac - 0.5f))) * wscale);
            total += w[i];
        }
        if(total != 0.0f)
            for(int i = 0; i < taps; ++i)
                w[i] /= total;
        first[d] = (int)fl - rad;
    }
    return taps;
}

// avdm_image_decode_integer: one lane per pixel; the transfer curve is a table (host-evaluated, see the entry point), alpha is linear
template <typename T>
__global__ void __launch_bounds__(256) decode_integer_kernel(float4* dst, int dst_pitch, const T* src, int src_pitch, int width, int height, int channels,
                                                             const float* __restrict__ lut, float inv)
{
    MAP_XY();
    if(roiX >= (unsigned)width || roiY >= (unsigned)height)
        return;
    const T* p = (const T*)((const char*)src + (long long)roiY * src_pitch) + (size_t)roiX * channels;
    float4 o;
    if(channels >= 3)
    {
        o.x = lut[p[0]];
        o.y = lut[p[1]];
        o.z = lut[p[2]];
        o.w = channels == 4 ? (float)p[3] * inv : 1.0f;
    }
    else
    {
        o.x = o.y = o.z = lut[p[0]];
        o.w = channels == 2 ? (float)p[1] * inv : 1.0f;
    }
    *((float4*)((char*)dst + (long long)roiY * dst_pitch) + roiX) = o;
}

// avdm_image_decode_exr_lines: one lane per pixel; per channel the lanes of a wave read consecutive samples of one line (coalesced), the four
// values leave as one 16-byte store.  HALF -> float is exact, UINT -> float is OpenEXR's / OpenImageIO's conversion ((float)u).
struct ExrLineLayout
{
    long long lineStride;
    long long off[4];
    int type[4]; // 0 UINT, 1 HALF, 2 FLOAT; off < 0: absent
};
// ALIGNED = false: OpenEXR aligns nothing (the header has any length, a line of HALF channels of odd width any parity): bytes one by one then
template <bool ALIGNED>
__device__ __forceinline__ float exr_sample(const unsigned char* line, long long off, int type, unsigned x)
{
    if(type == 1)
    {
        const unsigned char* p = line + off + 2ll * x;
        const unsigned short h = ALIGNED ? *(const unsigned short*)p : (unsigned short)(p[0] | (p[1] << 8));
        return __half2float(__ushort_as_half(h));
    }
    const unsigned char* p = line + off + 4ll * x;
    const unsigned u = ALIGNED ? *(const unsigned*)p : ((unsigned)p[0] | ((unsigned)p[1] << 8) | ((unsigned)p[2] << 16) | ((unsigned)p[3] << 24));
    return type == 2 ? __uint_as_float(u) : (float)u;
}
template <bool ALIGNED>
__global__ void __launch_bounds__(256) decode_exr_lines_kernel(float4* dst, int dst_pitch, const unsigned char* __restrict__ lines, ExrLineLayout L, int width,
                                                               int height)
{
    MAP_XY();
    if(roiX >= (unsigned)width || roiY >= (unsigned)height)
        return;
    const unsigned char* line = lines + (long long)roiY * L.lineStride;
    float4 o;
    o.x = exr_sample<ALIGNED>(line, L.off[0], L.type[0], roiX);
    o.y = exr_sample<ALIGNED>(line, L.off[1], L.type[1], roiX);
    o.z = exr_sample<ALIGNED>(line, L.off[2], L.type[2], roiX);
    o.w = L.off[3] >= 0 ? exr_sample<ALIGNED>(line, L.off[3], L.type[3], roiX) : 1.0f;
    *((float4*)((char*)dst + (long long)roiY * dst_pitch) + roiX) = o;
}

static inline dim3 map_grid(unsigned w, unsigned h) { return dim3(divUp(w, 64), divUp(h, 4)); }
static inline int stream_blocks(long long total)
{
    long long b = (total + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 256 * 16 ? 256 * 16 : b));
}

} // namespace avdm

using namespace avdm;

extern "C" {

int avdm_volume_initialize_u8(uint8_t* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, uint8_t value, void* stream)
{
    if(dimX <= 0 || dimY <= 0 || dimZ <= 0)
        return 0;
    if((pitch_x & 3) || (pitch_y & 3) || ((uintptr_t)vol & 3))
        return set_error_msg(1, "avdm_volume_initialize_u8: base / pitches must be multiples of 4");
    const unsigned v4 = value * 0x01010101u;
    const long long total = (long long)dimX * dimY * ((dimZ + 3) / 4);
    hipLaunchKernelGGL(volume_init_u8_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, vol, pitch_y, pitch_x, dimX, dimY, dimZ, v4);
    AVDM_LAUNCH_CHECK("avdm_volume_initialize_u8");
}

int avdm_volume_initialize_f16(void* vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, float value, void* stream)
{
    if(dimX <= 0 || dimY <= 0 || dimZ <= 0)
        return 0;
    const __half h = __float2half(value);
    const unsigned short hs = *reinterpret_cast<const unsigned short*>(&h);
    const long long total = (long long)dimX * dimY * dimZ;
    hipLaunchKernelGGL(volume_init_f16_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, (__half*)vol, pitch_y, pitch_x, dimX, dimY,
                       dimZ, hs);
    AVDM_LAUNCH_CHECK("avdm_volume_initialize_f16");
}

int avdm_volume_add_f16(void* inout_vol, const void* in_vol, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, void* stream)
{
    if(dimX <= 0 || dimY <= 0 || dimZ <= 0)
        return 0;
    const long long total = (long long)dimX * dimY * dimZ;
    hipLaunchKernelGGL(volume_add_f16_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, (__half*)inout_vol, (const __half*)in_vol,
                       pitch_y, pitch_x, dimX, dimY, dimZ);
    AVDM_LAUNCH_CHECK("avdm_volume_add_f16");
}

int avdm_volume_update_uninitialized(const uint8_t* best, uint8_t* second, long long pitch_y, int pitch_x, int dimX, int dimY, int dimZ, void* stream)
{
    if(dimX <= 0 || dimY <= 0 || dimZ <= 0)
        return 0;
    if((pitch_x & 3) || (pitch_y & 3) || ((uintptr_t)best & 3) || ((uintptr_t)second & 3))
        return set_error_msg(1, "avdm_volume_update_uninitialized: base / pitches must be multiples of 4");
    const long long total = (long long)dimX * dimY * ((dimZ + 3) / 4);
    hipLaunchKernelGGL(volume_update_uninit_kernel, dim3(stream_blocks(total)), dim3(256), 0, (hipStream_t)stream, best, second, pitch_y, pitch_x, dimX,
                       dimY, dimZ);
    AVDM_LAUNCH_CHECK("avdm_volume_update_uninitialized");
}

int avdm_volume_refine_best_depth(float* out_depth_sim, int out_pitch, const float* sgm_depth_pixsize, int map_pitch, const void* vol_f16,
                                  long long pitch_y, int pitch_x, int dimZ, const avdm_refine_params_t* rp, avdm_roi_t roi, void* stream)
{
    const unsigned roiW = roi.x.end - roi.x.begin, roiH = roi.y.end - roi.y.begin;
    if(roiW == 0 || roiH == 0)
        return 0;
    if(dimZ > RBD_MAXZ)
        return set_error_msg(1, "avdm_volume_refine_best_depth: more than 64 refine planes are not supported");
    if((pitch_x & 15) || (pitch_y & 15) || ((uintptr_t)vol_f16 & 15) || pitch_x < ((dimZ + 7) & ~7) * 2)
        return set_error_msg(1, "avdm_volume_refine_best_depth: volume base / pitches must be multiples of 16 and cover 8-aligned planes");
    const int halfNbSamples = rp->nbSubsamples * rp->halfNbDepths;
    const float tt = (float)(2.0 * rp->sigma * rp->sigma);
    // largest |zs - sample|
    const int zsMax = (dimZ - 1 - rp->halfNbDepths) > rp->halfNbDepths ? (dimZ - 1 - rp->halfNbDepths) : rp->halfNbDepths;
    const int tableSize = zsMax * rp->nbSubsamples + halfNbSamples + 1;
    const size_t lds = (size_t)((tableSize * 4 + 15) & ~15);
    if(lds > 64 * 1024)
        return set_error_msg(1, "avdm_volume_refine_best_depth: Gaussian table too large");
    // mirrored table gm[j] = gauss[|j - off|], padded for the last group of sub-samples
    const int off = tableSize - 1;
    if(2 * off + 1 + RBD_GROUP <= RBD_TAB)
    {
        GaussTable gt;
        for(int j = 0; j < RBD_TAB; ++j)
        {
            const int d = j - off < 0 ? off - j : j - off;
            gt.g[j] = d < tableSize ? expf(-(float)(d * d) / tt) : 0.0f;
        }
        if(dimZ <= 32)
            hipLaunchKernelGGL(refine_best_depth_ktab_kernel<32>, map_grid(roiW, roiH), dim3(256), 0, (hipStream_t)stream, (float2*)out_depth_sim,
                               out_pitch, (const float2*)sgm_depth_pixsize, map_pitch, (const __half*)vol_f16, pitch_y, pitch_x, dimZ, rp->nbSubsamples,
                               halfNbSamples, rp->halfNbDepths, off, gt, roi);
        else
            hipLaunchKernelGGL(refine_best_depth_ktab_kernel<64>, map_grid(roiW, roiH), dim3(256), 0, (hipStream_t)stream, (float2*)out_depth_sim,
                               out_pitch, (const float2*)sgm_depth_pixsize, map_pitch, (const __half*)vol_f16, pitch_y, pitch_x, dimZ, rp->nbSubsamples,
                               halfNbSamples, rp->halfNbDepths, off, gt, roi);
        AVDM_LAUNCH_CHECK("avdm_volume_refine_best_depth");
    }
    if(dimZ <= 32)
        hipLaunchKernelGGL(refine_best_depth_kernel<32>, map_grid(roiW, roiH), dim3(256), lds, (hipStream_t)stream, (float2*)out_depth_sim, out_pitch,
                           (const float2*)sgm_depth_pixsize, map_pitch, (const __half*)vol_f16, pitch_y, pitch_x, dimZ, rp->nbSubsamples,
                           halfNbSamples, rp->halfNbDepths, tt, tableSize, roi);
    else
        hipLaunchKernelGGL(refine_best_depth_kernel<64>, map_grid(roiW, roiH), dim3(256), lds, (hipStream_t)stream, (float2*)out_depth_sim, out_pitch,
                           (const float2*)sgm_depth_pixsize, map_pitch, (const __half*)vol_f16, pitch_y, pitch_x, dimZ, rp->nbSubsamples,
                           halfNbSamples, rp->halfNbDepths, tt, tableSize, roi);
    AVDM_LAUNCH_CHECK("avdm_volume_refine_best_depth");
}

int avdm_depth_sim_map_copy_depth_only(float* out_map, int out_pitch, const float* in_map, int in_pitch, int width, int height, float default_sim,
                                       void* stream)
{
    if(width <= 0 || height <= 0)
        return 0;
    hipLaunchKernelGGL(copy_depth_only_kernel, map_grid(width, height), dim3(256), 0, (hipStream_t)stream, (float2*)out_map, out_pitch,
                       (const float2*)in_map, in_pitch, (unsigned)width, (unsigned)height, default_sim);
    AVDM_LAUNCH_CHECK("avdm_depth_sim_map_copy_depth_only");
}

int avdm_normal_map_upscale(float* out_map, int out_pitch, const float* in_map, int in_pitch, float ratio, avdm_roi_t roi, void* stream)
{
    const unsigned roiW = roi.x.end - roi.x.begin, roiH = roi.y.end - roi.y.begin;
    if(roiW == 0 || roiH == 0)
        return 0;
    hipLaunchKernelGGL(normal_upscale_kernel, map_grid(roiW, roiH), dim3(256), 0, (hipStream_t)stream, out_map, out_pitch, in_map, in_pitch, ratio, roi);
    AVDM_LAUNCH_CHECK("avdm_normal_map_upscale");
}

int avdm_depth_thickness_smooth_thickness(float* inout_map, int pitch, const avdm_sgm_params_t* sp, const avdm_refine_params_t* rp, avdm_roi_t roi,
                                          void* stream)
{
    const unsigned roiW = roi.x.end - roi.x.begin, roiH = roi.y.end - roi.y.begin;
    if(roiW == 0 || roiH == 0)
        return 0;
    const int sgmScaleStep = sp->scale * sp->stepXY;
    const int refineScaleStep = rp->scale * rp->stepXY;
    const float minNbRefineSamples = 2.f;
    const float q = (float)sgmScaleStep / (float)refineScaleStep;
    const float maxNbRefineSamples = q > minNbRefineSamples ? q : minNbRefineSamples;
    const float minThicknessInflate = (float)rp->halfNbDepths / maxNbRefineSamples;
    const float maxThicknessInflate = (float)rp->halfNbDepths / minNbRefineSamples;
    hipLaunchKernelGGL(smooth_thickness_kernel, map_grid(roiW, roiH), dim3(256), 0, (hipStream_t)stream, (float2*)inout_map, pitch, minThicknessInflate,
                       maxThicknessInflate, roi);
    AVDM_LAUNCH_CHECK("avdm_depth_thickness_smooth_thickness");
}

int avdm_compute_sgm_upscaled_depth_pixsize_map(float* out_map, int out_pitch, const float* in_sgm_depth_thickness, int in_pitch,
                                                const avdm_camera_t* rc, const avdm_pyramid_t* rc_pyr, const avdm_refine_params_t* rp, float ratio,
                                                avdm_roi_t roi, void* stream)
{
    (void)rc;
    const unsigned roiW = roi.x.end - roi.x.begin, roiH = roi.y.end - roi.y.begin;
    if(roiW == 0 || roiH == 0)
        return 0;
    const TexLod lodTex = make_tex_lod(rc_pyr, rp->scale); // the Refine stage's level of the R image (fractional levels blend two)
    const float wN = (float)tex_dim_w(rc_pyr, rp->scale), hN = (float)tex_dim_h(rc_pyr, rp->scale);
    if(rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8)
        hipLaunchKernelGGL(upscale_depth_pixsize_kernel<true>, map_grid(roiW, roiH), dim3(256), 0, (hipStream_t)stream, (float2*)out_map, out_pitch,
                           (const float2*)in_sgm_depth_thickness, in_pitch, lodTex, wN, hN, rp->stepXY, rp->halfNbDepths, ratio,
                           rp->interpolateMiddleDepth, roi);
    else
        hipLaunchKernelGGL(upscale_depth_pixsize_kernel<false>, map_grid(roiW, roiH), dim3(256), 0, (hipStream_t)stream, (float2*)out_map, out_pitch,
                           (const float2*)in_sgm_depth_thickness, in_pitch, lodTex, wN, hN, rp->stepXY, rp->halfNbDepths, ratio,
                           rp->interpolateMiddleDepth, roi);
    AVDM_LAUNCH_CHECK("avdm_compute_sgm_upscaled_depth_pixsize_map");
}

int avdm_image_decode_integer(float* dst_rgba, int dst_pitch, const void* src, int src_pitch, int width, int height, int channels, int bits,
                              int srgb_to_linear, void* stream)
{
    if(width <= 0 || height <= 0)
        return set_error_msg(1, "avdm_image_decode_integer: empty image");
    if(channels < 1 || channels > 4 || (bits != 8 && bits != 16))
        return set_error_msg(1, "avdm_image_decode_integer: 1-4 channels of 8 or 16 bits");
    hipStream_t st = (hipStream_t)stream;
    const int n = bits == 8 ? 256 : 65536;
    std::vector<float> lut((size_t)n);
    const float inv = 1.0f / (float)(n - 1);
    for(int v = 0; v < n; ++v)
    {
        const float x = (float)v * inv;
        lut[(size_t)v] = !srgb_to_linear ? x : (x <= 0.04045f ? x * (1.0f / 12.92f) : powf((x + 0.055f) * (1.0f / 1.055f), 2.4f));
    }
    const StreamScratch lease(st, (size_t)n * sizeof(float));
    float* dlut = (float*)lease.ptr();
    if(dlut == nullptr)
        return set_error_msg(2, "avdm_image_decode_integer: scratch allocation failed");
    const hipError_t e = hipMemcpyAsync(dlut, lut.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice, st);
    if(e != hipSuccess)
        return ::avdm::set_error(e, "avdm_image_decode_integer");
    if(bits == 8)
        hipLaunchKernelGGL(decode_integer_kernel<uint8_t>, map_grid((unsigned)width, (unsigned)height), dim3(256), 0, st, (float4*)dst_rgba, dst_pitch,
                           (const uint8_t*)src, src_pitch, width, height, channels, dlut, inv);
    else
        hipLaunchKernelGGL(decode_integer_kernel<uint16_t>, map_grid((unsigned)width, (unsigned)height), dim3(256), 0, st, (float4*)dst_rgba, dst_pitch,
                           (const uint16_t*)src, src_pitch, width, height, channels, dlut, inv);
    AVDM_LAUNCH_CHECK("avdm_image_decode_integer");
}

int avdm_image_decode_exr_lines(float* dst_rgba, int dst_pitch, const void* lines, long long line_stride, int width, int height,
                                const long long chan_offset[4], const int chan_type[4], void* stream)
{
    if(width <= 0 || height <= 0)
        return set_error_msg(1, "avdm_image_decode_exr_lines: empty image");
    if(dst_rgba == nullptr || lines == nullptr || chan_offset == nullptr || chan_type == nullptr)
        return set_error_msg(1, "avdm_image_decode_exr_lines: null argument");
    ExrLineLayout L;
    L.lineStride = line_stride;
    bool aligned = true;
    for(int k = 0; k < 4; ++k)
    {
        L.off[k] = chan_offset[k];
        L.type[k] = chan_type[k];
        if(k < 3 && chan_offset[k] < 0)
            return set_error_msg(1, "avdm_image_decode_exr_lines: the R, G and B offsets are required (a Y-only image passes Y three times)");
        if(chan_offset[k] >= 0)
        {
            if(chan_type[k] < 0 || chan_type[k] > 2)
                return set_error_msg(1, "avdm_image_decode_exr_lines: pixel type must be 0 (UINT), 1 (HALF) or 2 (FLOAT)");
            const long long size = chan_type[k] == 1 ? 2 : 4;
            if(chan_offset[k] + size * width > (line_stride < 0 ? -line_stride : line_stride))
                return set_error_msg(1, "avdm_image_decode_exr_lines: a channel does not fit the line stride");
            if((((uintptr_t)lines + (uintptr_t)chan_offset[k]) % (uintptr_t)size) || (line_stride % size))
                aligned = false;
        }
    }
    if((dst_pitch & 15) || ((uintptr_t)dst_rgba & 15))
        return set_error_msg(1, "avdm_image_decode_exr_lines: destination base / pitch must be multiples of 16 bytes");
    if(aligned)
        hipLaunchKernelGGL(decode_exr_lines_kernel<true>, map_grid((unsigned)width, (unsigned)height), dim3(256), 0, (hipStream_t)stream, (float4*)dst_rgba,
                           dst_pitch, (const unsigned char*)lines, L, width, height);
    else
        hipLaunchKernelGGL(decode_exr_lines_kernel<false>, map_grid((unsigned)width, (unsigned)height), dim3(256), 0, (hipStream_t)stream, (float4*)dst_rgba,
                           dst_pitch, (const unsigned char*)lines, L, width, height);
    AVDM_LAUNCH_CHECK("avdm_image_decode_exr_lines");
}

int avdm_image_undistort(float* dst_rgba, int dst_pitch, const float* src_rgba, int src_pitch, const avdm_intrinsic_t* cam, const float fill_rgba[4],
                         void* stream)
{
    if(cam == nullptr || cam->width <= 0 || cam->height <= 0)
        return set_error_msg(1, "avdm_image_undistort: empty image");
    if(cam->distortion_model < AVDM_DISTORTION_NONE || cam->distortion_model > AVDM_DISTORTION_RADIALK3PT)
        return set_error_msg(1, "avdm_image_undistort: unknown distortion model");
    hipStream_t st = (hipStream_t)stream;
    const bool hasDistortion = cam->distortion_model != AVDM_DISTORTION_NONE; // IntrinsicScaleOffsetDisto::hasDistortion(): a distortion object exists
    if(!hasDistortion)
    { // cameraUndistortImage.hpp:89-93: no distortion, a direct copy
        const hipError_t e = hipMemcpy2DAsync(dst_rgba, (size_t)dst_pitch, src_rgba, (size_t)src_pitch, (size_t)cam->width * 16, (size_t)cam->height,
                                              hipMemcpyDeviceToDevice, st);
        return ::avdm::set_error(e, "avdm_image_undistort");
    }
    const float4 fill = make_float4(fill_rgba[0], fill_rgba[1], fill_rgba[2], fill_rgba[3]);
    hipLaunchKernelGGL(image_undistort_kernel, map_grid((unsigned)cam->width, (unsigned)cam->height), dim3(256), 0, st, (float4*)dst_rgba, dst_pitch,
                       (const float4*)src_rgba, src_pitch, *cam, fill);
    AVDM_LAUNCH_CHECK("avdm_image_undistort");
}

int avdm_image_resize(float* dst_rgba, int dst_pitch, int dst_w, int dst_h, const float* src_rgba, int src_pitch, int src_w, int src_h, void* stream)
{
    if(dst_w <= 0 || dst_h <= 0 || src_w <= 0 || src_h <= 0)
        return set_error_msg(1, "avdm_image_resize: empty image");
    if(dst_w > src_w || dst_h > src_h)
        return set_error_msg(1, "avdm_image_resize: enlarging is not supported (OpenImageIO would select blackman-harris)");
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> wx, wy;
    std::vector<int> fx, fy;
    const int xtaps = oiio_resize_taps(dst_w, src_w, wx, fx), ytaps = oiio_resize_taps(dst_h, src_h, wy, fy);
    const size_t bwx = (wx.size() * sizeof(float) + 255) & ~(size_t)255, bwy = (wy.size() * sizeof(float) + 255) & ~(size_t)255;
    const size_t bfx = (fx.size() * sizeof(int) + 255) & ~(size_t)255, bfy = (fy.size() * sizeof(int) + 255) & ~(size_t)255;
    const StreamScratch lease(st, bwx + bwy + bfx + bfy);
    char* tab = (char*)lease.ptr();
    if(tab == nullptr)
        return set_error_msg(2, "avdm_image_resize: scratch allocation failed");
    hipError_t e = hipMemcpyAsync(tab, wx.data(), wx.size() * sizeof(float), hipMemcpyHostToDevice, st);
    if(e == hipSuccess)
        e = hipMemcpyAsync(tab + bwx, wy.data(), wy.size() * sizeof(float), hipMemcpyHostToDevice, st);
    if(e == hipSuccess)
        e = hipMemcpyAsync(tab + bwx + bwy, fx.data(), fx.size() * sizeof(int), hipMemcpyHostToDevice, st);
    if(e == hipSuccess)
        e = hipMemcpyAsync(tab + bwx + bwy + bfx, fy.data(), fy.size() * sizeof(int), hipMemcpyHostToDevice, st);
    if(e == hipSuccess)
    {
        hipLaunchKernelGGL(image_resize_kernel, map_grid((unsigned)dst_w, (unsigned)dst_h), dim3(256), 0, st, (float4*)dst_rgba, dst_pitch, dst_w, dst_h,
                           (const float4*)src_rgba, src_pitch, src_w, src_h, (const float*)tab, (const int*)(tab + bwx + bwy), xtaps,
                           (const float*)(tab + bwx), (const int*)(tab + bwx + bwy + bfx), ytaps);
        e = hipGetLastError();
    }
    // the tap tables live in pageable host memory: they must outlive the copies
    const hipError_t es = hipStreamSynchronize(st);
    return ::avdm::set_error(e != hipSuccess ? e : es, "avdm_image_resize");
}

int avdm_depth_sim_map_compute_normal(float* out_normal, int out_pitch, const float* in_depth_sim, int in_pitch, const avdm_camera_t* rc, int stepXY,
                                      avdm_roi_t roi, void* stream)
{
    const unsigned roiW = roi.x.end - roi.x.begin, roiH = roi.y.end - roi.y.begin;
    if(roiW == 0 || roiH == 0)
        return 0;
    hipLaunchKernelGGL(compute_normal_kernel, map_grid(roiW, roiH), dim3(256), 0, (hipStream_t)stream, out_normal, out_pitch, (const float2*)in_depth_sim,
                       in_pitch, *rc, stepXY, roi);
    AVDM_LAUNCH_CHECK("avdm_depth_sim_map_compute_normal");
}

int avdm_depth_sim_map_optimize_gradient_descent(float* out_opt_depth_sim, int out_pitch, float* img_variance, int var_pitch, float* tmp_depth,
                                                 int tmp_pitch, int tmp_w, int tmp_h, const float* sgm_depth_pixsize, int sgm_pitch,
                                                 const float* refine_depth_sim, int ref_pitch, const avdm_camera_t* rc, const avdm_pyramid_t* rc_pyr,
                                                 const avdm_refine_params_t* rp, avdm_roi_t roi, void* stream)
{
    const unsigned roiW = roi.x.end - roi.x.begin, roiH = roi.y.end - roi.y.begin;
    if(roiW == 0 || roiH == 0)
        return 0;
    const TexLod lodTex = make_tex_lod(rc_pyr, rp->scale);
    const float wN = (float)tex_dim_w(rc_pyr, rp->scale), hN = (float)tex_dim_h(rc_pyr, rp->scale);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid = map_grid(roiW, roiH);

    hipLaunchKernelGGL(copy_rows_kernel, grid, dim3(256), 0, st, (float2*)out_opt_depth_sim, out_pitch, (const float2*)sgm_depth_pixsize, sgm_pitch, roiW,
                       roiH);
    if(rc_pyr->filter_mode == AVDM_FILTER_CUDA_FIXED8)
        hipLaunchKernelGGL(var_L_kernel<true>, grid, dim3(256), 0, st, img_variance, var_pitch, lodTex, wN, hN, rp->stepXY, roi);
    else
        hipLaunchKernelGGL(var_L_kernel<false>, grid, dim3(256), 0, st, img_variance, var_pitch, lodTex, wN, hN, rp->stepXY, roi);

    const int nIter = rp->optimizationNbIterations;
    const char* legacy = getenv("AVDM_OPT_DEPTH_MAP_FORM"); // A/B: the two-launches-per-iteration depth-map form
    if(legacy && legacy[0] == '1')
    {
        for(int iter = 0; iter < nIter; ++iter)
        {
            hipLaunchKernelGGL(extract_depth_kernel, grid, dim3(256), 0, st, tmp_depth, tmp_pitch, (const float2*)out_opt_depth_sim, out_pitch, roi);
            hipLaunchKernelGGL(optimize_step_kernel, grid, dim3(256), 0, st, (float2*)out_opt_depth_sim, out_pitch, (const float2*)sgm_depth_pixsize,
                               sgm_pitch, (const float2*)refine_depth_sim, ref_pitch, img_variance, var_pitch, tmp_depth, tmp_pitch, tmp_w, tmp_h, *rc,
                               iter, roi);
        }
        AVDM_LAUNCH_CHECK("avdm_depth_sim_map_optimize_gradient_descent");
    }
    if(nIter <= 0)
        return 0;
    // point-map form (see optimize_step_points_kernel): two float4 maps of the tile in the stream's scratch block (several tiles may be in
    // flight on different streams); `tmp_depth` — the reference's copy of the depth map — is not needed.  The depth texture is the tile
    // itself (the reference binds the whole allocated buffer and reads cells no kernel of the tile wrote).
    const int texW = std::min<int>(tmp_w, (int)roiW), texH = std::min<int>(tmp_h, (int)roiH);
    const int ptsPitch = (int)(((size_t)roiW * sizeof(float4) + 255) & ~(size_t)255);
    const size_t mapBytes = (size_t)ptsPitch * roiH;
    const StreamScratch lease(st, 2 * mapBytes);
    char* scratch = (char*)lease.ptr();
    if(scratch == nullptr)
        return set_error_msg(2, "avdm_depth_sim_map_optimize_gradient_descent: scratch allocation failed");
    float4* pts[2] = {(float4*)scratch, (float4*)(scratch + mapBytes)};
    hipLaunchKernelGGL(optimize_init_points_kernel, grid, dim3(256), 0, st, pts[0], ptsPitch, (const float2*)sgm_depth_pixsize, sgm_pitch, *rc, roi);
    for(int iter = 0; iter < nIter; ++iter)
    {
        if(iter == nIter - 1)
            hipLaunchKernelGGL(optimize_step_points_kernel<true>, grid, dim3(256), 0, st, pts[(iter + 1) & 1], pts[iter & 1], ptsPitch,
                               (float2*)out_opt_depth_sim, out_pitch, (const float2*)sgm_depth_pixsize, sgm_pitch, (const float2*)refine_depth_sim,
                               ref_pitch, img_variance, var_pitch, texW, texH, *rc, roi);
        else
            hipLaunchKernelGGL(optimize_step_points_kernel<false>, grid, dim3(256), 0, st, pts[(iter + 1) & 1], pts[iter & 1], ptsPitch,
                               (float2*)out_opt_depth_sim, out_pitch, (const float2*)sgm_depth_pixsize, sgm_pitch, (const float2*)refine_depth_sim,
                               ref_pitch, img_variance, var_pitch, texW, texH, *rc, roi);
    }
    return ::avdm::set_error(hipGetLastError(), "avdm_depth_sim_map_optimize_gradient_descent");
}

} // extern "C"
