/* avdm_fuse_oracle.c — CPU restatement of the depth-map filtering step that follows depth-map estimation.
 *
 * TEST INFRASTRUCTURE ONLY: nothing under alicevision_amd/ may call into this file.  The reference holds no golden vectors or
 * known-answer tests for fuseCut::Fuser and its translation units cannot be built whole here (OpenImageIO, Boost, Eigen,
 * SfMData); PINNED instead against the reference's OWN functions: oracle/_ref/libavdm_host_ref.so is every function listed below compiled
 * from the reference's text where it lies (oracle/ref/Makefile: gen_extract.py pulls the definitions out of Fuser.cpp,
 * MultiViewParams.cpp and common.cpp; mvsData is included unchanged), and tests/test_fuse_ref.py holds this file against it with ==
 * (pixel size incl. its NaN paths, modal-count maps, filtered maps).
 *
 * Restates, in the reference's double / float arithmetic and expression order (compiled with -ffp-contract=off):
 *   fuseCut/Fuser.cpp:66-121   Fuser::updateInSurr
 *   fuseCut/Fuser.cpp:144-231  Fuser::filterGroupsRC   (without the file I/O; the camera ranking is an input)
 *   fuseCut/Fuser.cpp:250-304  Fuser::filterDepthMapsRC
 *   mvsUtils/MultiViewParams.cpp:337-369  getPixelFor3DPoint (Point2d and Pixel forms)
 *   mvsUtils/MultiViewParams.cpp:386-448  getCamPixelSize(d), getCamPixelSizeRcTc, getCamPixelSizePlaneSweepAlpha
 *   mvsUtils/common.cpp:23-117, 119-153, 155-170  get2dLineImageIntersection, getTarEpipolarDirectedLine, triangulateMatch
 *   mvsData/geometry.cpp:14-17, 50-146    pointLineDistance3D, lineLineIntersect
 * Quirk kept: the per-T-camera hit counters are NOT reset between T cameras (StaticVector::resize_with on an unchanged size is a
 * no-op, mvsData/StaticVector.hpp:70), so a pixel's modal count is the number of T cameras from its first hit onwards.
 */
#include "avdm_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double x, y; } p2;
typedef struct { double x, y, z; } p3;

static p3 p3_sub(p3 a, p3 b) { p3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static p3 p3_add(p3 a, p3 b) { p3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static p3 p3_mul(p3 a, double d) { p3 r = {a.x * d, a.y * d, a.z * d}; return r; }
/* Point3d.hpp:100-113 */
static p3 p3_normalize(p3 a) { double d = sqrt(a.x * a.x + a.y * a.y + a.z * a.z); p3 r = {a.x / d, a.y / d, a.z / d}; return r; }
static double p3_size(p3 a) { double d = a.x * a.x + a.y * a.y + a.z * a.z; if(d == 0.0) return 0.0; return sqrt(d); }
static p3 p3_cross(p3 a, p3 b) { p3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; return r; }
/* Point2d.hpp:61-67 */
static double p2_size(p2 a) { return sqrt(a.x * a.x + a.y * a.y); }
static p2 p2_normalize(p2 a) { double d = sqrt(a.x * a.x + a.y * a.y); p2 r = {a.x / d, a.y / d}; return r; }

/* Matrix3x3 * Point2d (Matrix3x3.hpp:127-134), row-major m[9] */
static p3 m33_mul_p2(const double* m, p2 p)
{
    p3 r = {m[0] * p.x + m[1] * p.y + m[2], m[3] * p.x + m[4] * p.y + m[5], m[6] * p.x + m[7] * p.y + m[8]};
    return r;
}
/* Matrix3x4 * Point3d (Matrix3x4.hpp:45-49), row-major m[12] */
static p3 m34_mul_p3(const double* m, p3 p)
{
    p3 r = {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
    return r;
}

/* MultiViewParams.cpp:337-351 */
static p2 project2d(const double* P, p3 X)
{
    p3 XT = m34_mul_p3(P, X);
    p2 out;
    if(XT.z <= 0)
    {
        out.x = -1.0;
        out.y = -1.0;
    }
    else
    {
        out.x = XT.x / XT.z;
        out.y = XT.y / XT.z;
    }
    return out;
}

/* geometry.cpp:50-146; only the midpoint is used by the callers here */
static int line_line_intersect(p3* llis, p3 p1, p3 p2_, p3 p3_, p3 p4)
{
    double d1343, d4321, d1321, d4343, d2121, denom, numer, p13[3], p43[3], p21[3], pa[3], pb[3], muab[2];
    p13[0] = p1.x - p3_.x;
    p13[1] = p1.y - p3_.y;
    p13[2] = p1.z - p3_.z;
    p43[0] = p4.x - p3_.x;
    p43[1] = p4.y - p3_.y;
    p43[2] = p4.z - p3_.z;
    if((fabs(p43[0]) < FLT_EPSILON) && (fabs(p43[1]) < FLT_EPSILON) && (fabs(p43[2]) < FLT_EPSILON))
        return 0;
    p21[0] = p2_.x - p1.x;
    p21[1] = p2_.y - p1.y;
    p21[2] = p2_.z - p1.z;
    if((fabs(p21[0]) < FLT_EPSILON) && (fabs(p21[1]) < FLT_EPSILON) && (fabs(p21[2]) < FLT_EPSILON))
        return 0;
    d1343 = p13[0] * p43[0] + p13[1] * p43[1] + p13[2] * p43[2];
    d4321 = p43[0] * p21[0] + p43[1] * p21[1] + p43[2] * p21[2];
    d1321 = p13[0] * p21[0] + p13[1] * p21[1] + p13[2] * p21[2];
    d4343 = p43[0] * p43[0] + p43[1] * p43[1] + p43[2] * p43[2];
    d2121 = p21[0] * p21[0] + p21[1] * p21[1] + p21[2] * p21[2];
    denom = d2121 * d4343 - d4321 * d4321;
    if(fabs(denom) < FLT_EPSILON)
        return 0;
    numer = d1343 * d4321 - d1321 * d4343;
    muab[0] = numer / denom;
    muab[1] = (d1343 + d4321 * muab[0]) / d4343;
    pa[0] = p1.x + muab[0] * p21[0];
    pa[1] = p1.y + muab[0] * p21[1];
    pa[2] = p1.z + muab[0] * p21[2];
    pb[0] = p3_.x + muab[1] * p43[0];
    pb[1] = p3_.y + muab[1] * p43[1];
    pb[2] = p3_.z + muab[1] * p43[2];
    llis->x = (pa[0] + pb[0]) / 2.0;
    llis->y = (pa[1] + pb[1]) / 2.0;
    llis->z = (pa[2] + pb[2]) / 2.0;
    return 1;
}

/* common.cpp:23-117; pFrom / pTo keep their incoming values (default-constructed Point2d = (0,0) at the call site) unless set */
static int line_image_intersection(p2* pFrom, p2* pTo, p2 linePoint1, p2 linePoint2, int width, int height)
{
    p2 v = {linePoint2.x - linePoint1.x, linePoint2.y - linePoint1.y};
    if(p2_size(v) < FLT_EPSILON)
        return 0;
    v = p2_normalize(v);
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
        pFrom->x = x, pFrom->y = y;
        intersections++;
    }
    x = rw;
    y = (-c - a * rw) / b;
    if((y >= 0) && (y < rh))
    {
        if(intersections == 0)
            pFrom->x = x, pFrom->y = y;
        else
            pTo->x = x, pTo->y = y;
        intersections++;
    }
    x = -c / a;
    y = 0;
    if((x >= 0) && (x < rw))
    {
        if(intersections == 0)
            pFrom->x = x, pFrom->y = y;
        else
            pTo->x = x, pTo->y = y;
        intersections++;
    }
    x = (-c - b * rh) / a;
    y = rh;
    if((x >= 0) && (x < rw))
    {
        if(intersections == 0)
            pFrom->x = x, pFrom->y = y;
        else
            pTo->x = x, pTo->y = y;
        intersections++;
    }
    if(intersections == 2)
    {
        const p2 dF = {linePoint1.x - pFrom->x, linePoint1.y - pFrom->y}, dT = {linePoint1.x - pTo->x, linePoint1.y - pTo->y};
        if(p2_size(dF) > p2_size(dT))
        {
            const p2 t = *pFrom;
            *pFrom = *pTo;
            *pTo = t;
        }
        return 1;
    }
    return 0;
}

/* common.cpp:119-153; decomposeProjectionMatrix(P) reproduces CArr / iCamArr (the same function filled them, MultiViewParams.cpp:170-172) */
static int tar_epipolar_directed_line(p2* pFromTar, p2* pToTar, p2 refpix, const avo_fuse_cam_t* rc, const avo_fuse_cam_t* tc)
{
    const p3 rC = {rc->C[0], rc->C[1], rc->C[2]}, tC = {tc->C[0], tc->C[1], tc->C[2]};
    p3 refvect = m33_mul_p2(rc->iP, refpix);
    refvect = p3_normalize(refvect);
    const float d = (float)p3_size(p3_sub(rC, tC));
    p3 X = p3_add(p3_mul(refvect, (double)d), rC);
    const p2 tarpix1 = project2d(tc->P, X);
    X = p3_add(p3_mul(p3_mul(refvect, (double)d), 500.0), rC);
    const p2 tarpix2 = project2d(tc->P, X);
    return line_image_intersection(pFromTar, pToTar, tarpix1, tarpix2, tc->width, tc->height);
}

/* common.cpp:155-170 */
static int triangulate_match(p3* out, p2 refpix, p2 tarpix, const avo_fuse_cam_t* rc, const avo_fuse_cam_t* tc)
{
    const p3 rC = {rc->C[0], rc->C[1], rc->C[2]}, tC = {tc->C[0], tc->C[1], tc->C[2]};
    p3 refvect = m33_mul_p2(rc->iP, refpix);
    refvect = p3_normalize(refvect);
    const p3 refpoint = p3_add(refvect, rC);
    p3 tarvect = m33_mul_p2(tc->iP, tarpix);
    tarvect = p3_normalize(tarvect);
    const p3 tarpoint = p3_add(tarvect, tC);
    return line_line_intersect(out, rC, refpoint, tC, tarpoint);
}

/* MultiViewParams.cpp:386-401 */
static double cam_pixel_size(p3 x0, const avo_fuse_cam_t* cam, float d)
{
    if(d == 0.0f)
        return 0.0f;
    p2 pix = project2d(cam->P, x0);
    pix.x = pix.x + d;
    p3 vect = m33_mul_p2(cam->iP, pix);
    vect = p3_normalize(vect);
    const p3 C = {cam->C[0], cam->C[1], cam->C[2]};
    return p3_size(p3_cross(vect, p3_sub(C, x0))); /* geometry.cpp:14-17 */
}

/* MultiViewParams.cpp:406-435 */
static double cam_pixel_size_rc_tc(p3 p, const avo_fuse_cam_t* rc, const avo_fuse_cam_t* tc, float d)
{
    if(d == 0.0f)
        return 0.0f;
    const p3 rC = {rc->C[0], rc->C[1], rc->C[2]};
    p3 p1 = p3_add(rC, p3_mul(p3_sub(p, rC), 0.1f));
    const p2 rpix = project2d(rc->P, p);
    p2 pFromTar = {0.0, 0.0}, pToTar = {0.0, 0.0};
    tar_epipolar_directed_line(&pFromTar, &pToTar, rpix, rc, tc);
    const p2 dir = {pToTar.x - pFromTar.x, pToTar.y - pFromTar.y};
    const p2 n = p2_normalize(dir);
    const p2 pixelVect = {n.x * d, n.y * d};
    const p2 tpix = project2d(tc->P, p);
    const p2 tpix1 = {tpix.x + pixelVect.x * d, tpix.y + pixelVect.y * d};
    if(!triangulate_match(&p1, rpix, tpix1, rc, tc))
        return cam_pixel_size(p, rc, d);
    return p3_size(p3_sub(p, p1));
}

/* MultiViewParams.cpp:437-448 */
static double cam_pixel_size_plane_sweep_alpha(p3 p, const avo_fuse_cam_t* rc, const avo_fuse_cam_t* tc, int scale, int step)
{
    const double splaneSeweepAlpha = (double)(scale * step);
    const double avRcTc = cam_pixel_size_rc_tc(p, rc, tc, (float)splaneSeweepAlpha);
    const double avRc = cam_pixel_size(p, rc, (float)splaneSeweepAlpha);
    return (avRcTc + avRc) * 0.5;
}

double avo_fuse_pixel_size_plane_sweep_alpha(const double p[3], const avo_fuse_cam_t* rc, const avo_fuse_cam_t* tc)
{
    const p3 q = {p[0], p[1], p[2]};
    return cam_pixel_size_plane_sweep_alpha(q, rc, tc, 1, 1);
}

/* Fuser.cpp:66-121 with scale = 1 */
static int update_in_surr(float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP, p3 p, const avo_fuse_cam_t* rc, const avo_fuse_cam_t* tc,
                          int* numOfPtsMap, const float* depthMap, const float* simMap)
{
    const int w = rc->width, h = rc->height;
    /* MultiViewParams.cpp:353-369 (Pixel form), :495-499 with g_border = 2 (MultiViewParams.hpp:111) */
    const p3 XT = m34_mul_p3(rc->P, p);
    int px, py;
    if(XT.z <= 0)
        px = -1, py = -1;
    else
    {
        px = (int)floor(XT.x / XT.z + 0.5);
        py = (int)floor(XT.y / XT.z + 0.5);
    }
    if(!((px >= 2) && (px < w - 2) && (py >= 2) && (py < h - 2)))
        return 0;
    const p3 rC = {rc->C[0], rc->C[1], rc->C[2]};
    const float pixDepth = (float)p3_size(p3_sub(rC, p));
    int d = pixSizeBall;
    const float sim = simMap[(size_t)py * w + px];
    if(sim >= 1.0f)
        d = pixSizeBallWSP;
    const float pixSize = (float)(pixToleranceFactor * cam_pixel_size_plane_sweep_alpha(p, rc, tc, 1, 1));
    const int x0 = px - d > 0 ? px - d : 0, x1 = px + d < w - 1 ? px + d : w - 1;
    const int y0 = py - d > 0 ? py - d : 0, y1 = py + d < h - 1 ? py + d : h - 1;
    for(int nx = x0; nx <= x1; nx++)
        for(int ny = y0; ny <= y1; ny++)
        {
            const float depth = depthMap[(size_t)ny * w + nx];
            if(fabs(pixDepth - depth) < pixSize)
                numOfPtsMap[(size_t)ny * w + nx]++;
        }
    return 1;
}

/* Fuser.cpp:144-231.  tc_depth[c] == NULL stands for a T camera without a depth map (skipped, :189). */
int avo_fuse_filter_groups_rc(unsigned char* nmod, const float* depth, const float* sim, const avo_fuse_cam_t* rc, int n_tc, const avo_fuse_cam_t* tcs,
                              const float* const* tc_depth, float pixToleranceFactor, int pixSizeBall, int pixSizeBallWSP)
{
    const int w = rc->width, h = rc->height;
    int* numOfPtsMap = (int*)calloc((size_t)w * h, sizeof(int));
    if(numOfPtsMap == NULL)
        return 1;
    memset(nmod, 0, (size_t)w * h);
    for(int c = 0; c < n_tc; c++)
    {
        const avo_fuse_cam_t* tc = &tcs[c];
        const float* tcdepthMap = tc_depth[c];
        if(tcdepthMap == NULL || tc->width <= 0 || tc->height <= 0)
            continue;
        const p3 tC = {tc->C[0], tc->C[1], tc->C[2]};
        for(int y = 0; y < tc->height; ++y)
            for(int x = 0; x < tc->width; ++x)
            {
                const float dpt = tcdepthMap[(size_t)y * tc->width + x];
                if(dpt > 0.0f)
                {
                    const p2 pix = {(double)(float)x, (double)(float)y};
                    const p3 p = p3_add(tC, p3_mul(p3_normalize(m33_mul_p2(tc->iP, pix)), (double)dpt));
                    update_in_surr(pixToleranceFactor, pixSizeBall, pixSizeBallWSP, p, rc, tc, numOfPtsMap, depth, sim);
                }
            }
        for(size_t i = 0; i < (size_t)w * h; i++)
            nmod[i] = (unsigned char)(nmod[i] + (numOfPtsMap[i] > 0 ? 1 : 0));
    }
    free(numOfPtsMap);
    return 0;
}

/* Fuser.cpp:250-304 */
void avo_fuse_filter_depth_maps_rc(float* depthMap, float* simMap, const unsigned char* numOfModalsMap, size_t n, int minNumOfModals,
                                   int minNumOfModalsWSP2SSP)
{
    for(size_t i = 0; i < n; i++)
    {
        if(depthMap[i] <= -2.0f)
            continue;
        if((numOfModalsMap[i] >= minNumOfModalsWSP2SSP - 1) && (simMap[i] >= 1.0f))
            simMap[i] = simMap[i] - 2.0f;
        if((numOfModalsMap[i] <= 1) && (simMap[i] >= 1.0f))
        {
            depthMap[i] = -1.0f;
            simMap[i] = 1.0f;
        }
        if((numOfModalsMap[i] < minNumOfModals - 1) && (simMap[i] < 1.0f))
        {
            depthMap[i] = -1.0f;
            simMap[i] = 1.0f;
        }
    }
}
