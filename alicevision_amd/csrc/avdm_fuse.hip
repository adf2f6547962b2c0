// avdm_fuse.hip — depth-map filtering after depth-map estimation, gfx950.  Compiled with -ffp-contract=off.
//
// The reference runs fuseCut::Fuser::filterGroupsRC / filterDepthMapsRC on the host cores, one reference camera per OpenMP thread
// (fuseCut/Fuser.cpp:124-141, 234-247): for every pixel of every T camera's depth map the 3-D point is projected into the reference
// camera, and if the reference depth map agrees within a tolerance derived from the pixel footprint of both cameras, the pixel of the
// reference camera counts one more "modal" camera.  That is a scatter with independent items:
//
//   fuse_groups_kernel       one lane per T-camera pixel; double arithmetic exactly as written in the reference (IEEE div / sqrt, no
//                            contraction), float where the reference narrows to float.  A hit is atomicMin(first[cell], c) with c the
//                            index of the T camera: the reference never resets its hit counters between T cameras
//                            (StaticVector::resize_with on an unchanged size, mvsData/StaticVector.hpp:70), so what a pixel ends up
//                            with is "number of T cameras from the first hit onwards" = nT - first.
//   fuse_nmod_kernel         first -> modal count (uint8, wraps like the reference's unsigned char)
//   fuse_filter_maps_kernel  Fuser.cpp:250-304, elementwise
//
// Bound: the double-precision VALU rate (≈ 30 divisions / square roots per item; 4 B read + a gather of 8 B per item from HBM).
#include "avdm_device.h"

#include "../../include/avdm_fuse.h"

#include <float.h>

namespace avdm {

struct d2
{
    double x, y;
};
struct d3
{
    double x, y, z;
};
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return d3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return d3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ d3 operator*(d3 a, double d) { return d3{a.x * d, a.y * d, a.z * d}; }
// Point3d.hpp:100-113
__device__ __forceinline__ d3 normalize(d3 a)
{
    const double d = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return d3{a.x / d, a.y / d, a.z / d};
}
__device__ __forceinline__ double size(d3 a)
{
    const double d = a.x * a.x + a.y * a.y + a.z * a.z;
    return d == 0.0 ? 0.0 : sqrt(d);
}
__device__ __forceinline__ d3 cross(d3 a, d3 b) { return d3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// Point2d.hpp:61-67
__device__ __forceinline__ double size(d2 a) { return sqrt(a.x * a.x + a.y * a.y); }
__device__ __forceinline__ d2 normalize(d2 a)
{
    const double d = sqrt(a.x * a.x + a.y * a.y);
    return d2{a.x / d, a.y / d};
}
__device__ __forceinline__ d3 ldC(const avdm_fuse_camera_t& c) { return d3{c.C[0], c.C[1], c.C[2]}; }
// Matrix3x3.hpp:127-134, Matrix3x4.hpp:45-49 (row-major)
__device__ __forceinline__ d3 iPmul(const avdm_fuse_camera_t& c, d2 p)
{
    const double* m = c.iP;
    return d3{m[0] * p.x + m[1] * p.y + m[2], m[3] * p.x + m[4] * p.y + m[5], m[6] * p.x + m[7] * p.y + m[8]};
}
__device__ __forceinline__ d3 Pmul(const avdm_fuse_camera_t& c, d3 p)
{
    const double* m = c.P;
    return d3{m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
}
// MultiViewParams.cpp:337-351
__device__ __forceinline__ d2 project2d(const avdm_fuse_camera_t& c, d3 X)
{
    const d3 XT = Pmul(c, X);
    if(XT.z <= 0)
        return d2{-1.0, -1.0};
    return d2{XT.x / XT.z, XT.y / XT.z};
}

// geometry.cpp:50-146 (the midpoint only)
__device__ __forceinline__ bool lineLineIntersect(d3& llis, d3 p1, d3 p2, d3 p3, d3 p4)
{
    const double p13x = p1.x - p3.x, p13y = p1.y - p3.y, p13z = p1.z - p3.z;
    const double p43x = p4.x - p3.x, p43y = p4.y - p3.y, p43z = p4.z - p3.z;
    if((fabs(p43x) < FLT_EPSILON) && (fabs(p43y) < FLT_EPSILON) && (fabs(p43z) < FLT_EPSILON))
        return false;
    const double p21x = p2.x - p1.x, p21y = p2.y - p1.y, p21z = p2.z - p1.z;
    if((fabs(p21x) < FLT_EPSILON) && (fabs(p21y) < FLT_EPSILON) && (fabs(p21z) < FLT_EPSILON))
        return false;
    const double d1343 = p13x * p43x + p13y * p43y + p13z * p43z;
    const double d4321 = p43x * p21x + p43y * p21y + p43z * p21z;
    const double d1321 = p13x * p21x + p13y * p21y + p13z * p21z;
    const double d4343 = p43x * p43x + p43y * p43y + p43z * p43z;
    const double d2121 = p21x * p21x + p21y * p21y + p21z * p21z;
    const double denom = d2121 * d4343 - d4321 * d4321;
    if(fabs(denom) < FLT_EPSILON)
        return false;
    const double numer = d1343 * d4321 - d1321 * d4343;
    const double mua = numer / denom;
    const double mub = (d1343 + d4321 * mua) / d4343;
    const double pax = p1.x + mua * p21x, pay = p1.y + mua * p21y, paz = p1.z + mua * p21z;
    const double pbx = p3.x + mub * p43x, pby = p3.y + mub * p43y, pbz = p3.z + mub * p43z;
    llis.x = (pax + pbx) / 2.0;
    llis.y = (pay + pby) / 2.0;
    llis.z = (paz + pbz) / 2.0;
    return true;
}

// common.cpp:23-117; pFrom / pTo enter as (0, 0) (default-constructed Point2d at the call site) and keep that unless set
__device__ __forceinline__ bool get2dLineImageIntersection(d2& pFrom, d2& pTo, d2 linePoint1, d2 linePoint2, int width, int height)
{
    d2 v{linePoint2.x - linePoint1.x, linePoint2.y - linePoint1.y};
    if(size(v) < FLT_EPSILON)
        return false;
    v = normalize(v);
    const double a = -v.y;
    const double b = v.x;
    const double c = -a * linePoint1.x - b * linePoint1.y;
    int intersections = 0;
    const double rw = (double)width;
    const double rh = (double)height;

    double x = 0;
    double y = -c / b;
    if((y >= 0) && (y < rh))
    {
        pFrom = d2{x, y};
        intersections++;
    }
    x = rw;
    y = (-c - a * rw) / b;
    if((y >= 0) && (y < rh))
    {
        if(intersections == 0)
            pFrom = d2{x, y};
        else
            pTo = d2{x, y};
        intersections++;
    }
    x = -c / a;
    y = 0;
    if((x >= 0) && (x < rw))
    {
        if(intersections == 0)
            pFrom = d2{x, y};
        else
            pTo = d2{x, y};
        intersections++;
    }
    x = (-c - b * rh) / a;
    y = rh;
    if((x >= 0) && (x < rw))
    {
        if(intersections == 0)
            pFrom = d2{x, y};
        else
            pTo = d2{x, y};
        intersections++;
    }
    if(intersections == 2)
    {
        if(size(d2{linePoint1.x - pFrom.x, linePoint1.y - pFrom.y}) > size(d2{linePoint1.x - pTo.x, linePoint1.y - pTo.y}))
        {
            const d2 t = pFrom;
            pFrom = pTo;
            pTo = t;
        }
        return true;
    }
    return false;
}

// MultiViewParams.cpp:386-401 (pointLineDistance3D: geometry.cpp:14-17)
__device__ __forceinline__ double getCamPixelSize(d3 x0, const avdm_fuse_camera_t& cam, float d)
{
    if(d == 0.0f)
        return 0.0f;
    d2 pix = project2d(cam, x0);
    pix.x = pix.x + d;
    const d3 vect = normalize(iPmul(cam, pix));
    return size(cross(vect, ldC(cam) - x0));
}

// MultiViewParams.cpp:406-435 with getTarEpipolarDirectedLine (common.cpp:119-153) and triangulateMatch (:155-170) in line.
// dRcTc = (float)|C_rc - C_tc| (common.cpp:143), the same for every point of a camera pair: computed once on the host.
__device__ __forceinline__ double getCamPixelSizeRcTc(d3 p, const avdm_fuse_camera_t& rc, const avdm_fuse_camera_t& tc, float d, float dRcTc)
{
    if(d == 0.0f)
        return 0.0f;
    const d3 rC = ldC(rc), tC = ldC(tc);
    d3 p1 = rC + (p - rC) * (double)0.1f;
    const d2 rpix = project2d(rc, p);

    // getTarEpipolarDirectedLine; its result flag is ignored by the caller (:418)
    const d3 refvect = normalize(iPmul(rc, rpix));
    d3 X = refvect * (double)dRcTc + rC;
    const d2 tarpix1 = project2d(tc, X);
    X = (refvect * (double)dRcTc) * 500.0 + rC;
    const d2 tarpix2 = project2d(tc, X);
    d2 pFromTar{0.0, 0.0}, pToTar{0.0, 0.0};
    get2dLineImageIntersection(pFromTar, pToTar, tarpix1, tarpix2, tc.width, tc.height);

    const d2 n = normalize(d2{pToTar.x - pFromTar.x, pToTar.y - pFromTar.y});
    const d2 pixelVect{n.x * d, n.y * d};
    const d2 tpix = project2d(tc, p);
    const d2 tpix1{tpix.x + pixelVect.x * d, tpix.y + pixelVect.y * d};

    // triangulateMatch(p1, rpix, tpix1, rc, tc): refvect is the same normalised ray as above (iCamArr[rc] * rpix)
    const d3 refpoint = refvect + rC;
    const d3 tarvect = normalize(iPmul(tc, tpix1));
    const d3 tarpoint = tarvect + tC;
    if(!lineLineIntersect(p1, rC, refpoint, tC, tarpoint))
        return getCamPixelSize(p, rc, d);
    return size(p - p1);
}

struct FuseGroupsArgs
{
    int* first;           // per reference pixel: index of the first T camera with a hit (>= nT: none)
    const float* rcDepth; // reference camera maps
    const float* rcSim;
    const float* tcDepth;
    int rcDepthPitch, rcSimPitch, tcDepthPitch;
    int c;                // index of this T camera among the T cameras with a depth map
    float pixToleranceFactor;
    int pixSizeBall, pixSizeBallWSP;
    float dRcTc;
    avdm_fuse_camera_t rc, tc;
};

__global__ __launch_bounds__(256) void fuse_groups_kernel(const FuseGroupsArgs A)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(x >= A.tc.width || y >= A.tc.height)
        return;
    const float depthT = *(const float*)((const char*)A.tcDepth + (size_t)y * A.tcDepthPitch + (size_t)x * 4);
    if(!(depthT > 0.0f))
        return;
    // Fuser.cpp:199
    const d3 p = ldC(A.tc) + normalize(iPmul(A.tc, d2{(double)(float)x, (double)(float)y})) * (double)depthT;

    // updateInSurr (Fuser.cpp:66-121), scale 1
    const int w = A.rc.width, h = A.rc.height;
    const d3 XT = Pmul(A.rc, p);
    if(XT.z <= 0)
        return; // pixel (-1, -1): outside
    const int px = (int)floor(XT.x / XT.z + 0.5);
    const int py = (int)floor(XT.y / XT.z + 0.5);
    if(!((px >= 2) && (px < w - 2) && (py >= 2) && (py < h - 2))) // g_border = 2 (MultiViewParams.hpp:111)
        return;
    const float pixDepth = (float)size(ldC(A.rc) - p);
    int d = A.pixSizeBall;
    const float sim = *(const float*)((const char*)A.rcSim + (size_t)py * A.rcSimPitch + (size_t)px * 4);
    if(sim >= 1.0f)
        d = A.pixSizeBallWSP;
    // getCamPixelSizePlaneSweepAlpha(p, rc, tc, 1, 1) (MultiViewParams.cpp:437-448)
    const double avRcTc = getCamPixelSizeRcTc(p, A.rc, A.tc, 1.0f, A.dRcTc);
    const double avRc = getCamPixelSize(p, A.rc, 1.0f);
    const float pixSize = (float)((double)A.pixToleranceFactor * ((avRcTc + avRc) * 0.5));

    const int x0 = max(0, px - d), x1 = min(w - 1, px + d);
    const int y0 = max(0, py - d), y1 = min(h - 1, py + d);
    for(int ny = y0; ny <= y1; ny++)
        for(int nx = x0; nx <= x1; nx++)
        {
            const float depth = *(const float*)((const char*)A.rcDepth + (size_t)ny * A.rcDepthPitch + (size_t)nx * 4);
            if(fabsf(pixDepth - depth) < pixSize)
            {
                int* f = A.first + (size_t)ny * w + nx;
                if(*f > A.c) // monotone: a stale larger value only costs the atomic
                    atomicMin(f, A.c);
            }
        }
}

__global__ __launch_bounds__(256) void fuse_nmod_kernel(unsigned char* nmod, int nmodPitch, const int* first, int w, int h, int nT)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(x >= w || y >= h)
        return;
    const int f = first[(size_t)y * w + x];
    nmod[(size_t)y * nmodPitch + x] = (unsigned char)(f < nT ? nT - f : 0);
}

__global__ __launch_bounds__(256) void fuse_filter_maps_kernel(float* depth, int depthPitch, float* sim, int simPitch, const unsigned char* nmod,
                                                               int nmodPitch, int w, int h, int minNumOfModals, int minNumOfModalsWSP2SSP)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(x >= w || y >= h)
        return;
    float* dp = (float*)((char*)depth + (size_t)y * depthPitch) + x;
    float* sp = (float*)((char*)sim + (size_t)y * simPitch) + x;
    float dv = *dp, sv = *sp;
    const int n = nmod[(size_t)y * nmodPitch + x];
    // the point is part of a mask (alpha): untouched (Fuser.cpp:274-275)
    if(dv <= -2.0f)
        return;
    // consistent in enough T cameras and weakly supported: make it strongly supported
    if((n >= minNumOfModalsWSP2SSP - 1) && (sv >= 1.0f))
        sv = sv - 2.0f;
    // weakly supported points must be consistent in at least two cameras
    if((n <= 1) && (sv >= 1.0f))
    {
        dv = -1.0f;
        sv = 1.0f;
    }
    // strongly supported but not consistent in the minimal number of cameras
    if((n < minNumOfModals - 1) && (sv < 1.0f))
    {
        dv = -1.0f;
        sv = 1.0f;
    }
    *dp = dv;
    *sp = sv;
}

} // namespace avdm

using namespace avdm;

extern "C" {

size_t avdm_fuse_filter_groups_scratch_bytes(int width, int height)
{
    if(width <= 0 || height <= 0)
        return 0;
    return (size_t)width * height * sizeof(int);
}

int avdm_fuse_filter_groups(unsigned char* out_nmod, int nmod_pitch, const float* rc_depth, int depth_pitch, const float* rc_sim, int sim_pitch,
                            const avdm_fuse_camera_t* rc, int n_tc, const avdm_fuse_tc_t* tcs, float pixToleranceFactor, int pixSizeBall,
                            int pixSizeBallWSP, void* scratch, void* stream)
{
    if(out_nmod == nullptr || rc_depth == nullptr || rc_sim == nullptr || rc == nullptr || scratch == nullptr || (n_tc > 0 && tcs == nullptr))
        return set_error_msg(1, "avdm_fuse_filter_groups: null argument");
    const int w = rc->width, h = rc->height;
    if(w <= 0 || h <= 0 || nmod_pitch < w || depth_pitch < w * 4 || sim_pitch < w * 4 || n_tc < 0)
        return set_error_msg(1, "avdm_fuse_filter_groups: bad map size or pitch");
    if(pixSizeBall < 0 || pixSizeBallWSP < 0)
        return set_error_msg(1, "avdm_fuse_filter_groups: negative ball size");
    hipStream_t st = (hipStream_t)stream;
    int* first = (int*)scratch;
    hipError_t e = hipMemsetAsync(first, 0x7f, (size_t)w * h * sizeof(int), st); // 0x7f7f7f7f: no hit
    if(e != hipSuccess)
        return set_error(e, "avdm_fuse_filter_groups(memset)");
    int nT = 0;
    for(int c = 0; c < n_tc; ++c)
    {
        const avdm_fuse_tc_t& t = tcs[c];
        if(t.depth == nullptr || t.cam.width <= 0 || t.cam.height <= 0)
            continue; // Fuser.cpp:189
        if(t.depth_pitch < t.cam.width * 4)
            return set_error_msg(1, "avdm_fuse_filter_groups: bad T depth map pitch");
        FuseGroupsArgs A;
        A.first = first;
        A.rcDepth = rc_depth, A.rcSim = rc_sim, A.tcDepth = t.depth;
        A.rcDepthPitch = depth_pitch, A.rcSimPitch = sim_pitch, A.tcDepthPitch = t.depth_pitch;
        A.c = nT;
        A.pixToleranceFactor = pixToleranceFactor;
        A.pixSizeBall = pixSizeBall, A.pixSizeBallWSP = pixSizeBallWSP;
        {
            // common.cpp:143: float d = (rC - tC).size()  (Point3d.hpp:106-113)
            const double dx = rc->C[0] - t.cam.C[0], dy = rc->C[1] - t.cam.C[1], dz = rc->C[2] - t.cam.C[2];
            const double s = dx * dx + dy * dy + dz * dz;
            A.dRcTc = (float)(s == 0.0 ? 0.0 : sqrt(s));
        }
        A.rc = *rc, A.tc = t.cam;
        hipLaunchKernelGGL(fuse_groups_kernel, dim3(divUp(t.cam.width, 64), divUp(t.cam.height, 4)), dim3(256), 0, st, A);
        ++nT;
    }
    hipLaunchKernelGGL(fuse_nmod_kernel, dim3(divUp(w, 64), divUp(h, 4)), dim3(256), 0, st, out_nmod, nmod_pitch, first, w, h, nT);
    AVDM_LAUNCH_CHECK("avdm_fuse_filter_groups");
}

int avdm_fuse_filter_depth_maps(float* depth, int depth_pitch, float* sim, int sim_pitch, const unsigned char* nmod, int nmod_pitch, int width,
                                int height, int minNumOfModals, int minNumOfModalsWSP2SSP, void* stream)
{
    if(depth == nullptr || sim == nullptr || nmod == nullptr)
        return set_error_msg(1, "avdm_fuse_filter_depth_maps: null argument");
    if(width <= 0 || height <= 0 || depth_pitch < width * 4 || sim_pitch < width * 4 || nmod_pitch < width)
        return set_error_msg(1, "avdm_fuse_filter_depth_maps: bad map size or pitch");
    hipLaunchKernelGGL(fuse_filter_maps_kernel, dim3(divUp(width, 64), divUp(height, 4)), dim3(256), 0, (hipStream_t)stream, depth, depth_pitch, sim,
                       sim_pitch, nmod, nmod_pitch, width, height, minNumOfModals, minNumOfModalsWSP2SSP);
    AVDM_LAUNCH_CHECK("avdm_fuse_filter_depth_maps");
}

} // extern "C"
