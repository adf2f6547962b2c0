// avdm_knife.h — the reference's R-side border test AS WRITTEN, for the knife-edge rows (see avdm_similarity.hip "knife-edge rows").
//
// Plain C++ on purpose (no HIP types): avdm_similarity.hip includes it for the device under `#pragma clang fp contract(off)`, and
// tests/test_oracle.py compiles the same text for the host with `g++ -ffp-contract=off` and compares it, voxel for voxel, with the oracle's
// literal evaluation (which equals the reference's own kernels compiled for the CPU).  Every operation below is one fp32 operation of the
// reference in the reference's order:
//   get3DPointForPixelAndFrontoParellePlaneRC   Patch.cuh:157-163     (SGM: the pixel's ray cut with the fronto-parallel plane)
//   get3DPointForPixelAndDepthFromRC            Patch.cuh:165-170     (Refine: the point at the SGM depth on the ray)
//   move3DPointByRcPixSize                      kernels.cuh:17-24     (Refine: moved by `rel` pixel sizes along the ray)
//   normalize = a * __fdividef(1, sqrtf(dot))   matrix.cuh:66-75      (IEEE division and square root, like the reference compiled for the CPU)
//   linePlaneIntersect                          matrix.cuh:182-189
//   project3DPoint                              matrix.cuh:117-126
//   the border test                             Patch.cuh:486-496
// Matrices are column-major like DeviceCameraParams (P 3 x 4, iP 3 x 3).
#pragma once

#ifndef AVDM_KNIFE_FN
#define AVDM_KNIFE_FN inline
#endif

// (included INSIDE namespace avdm by both users)
namespace knife {

struct v3
{
    float x, y, z;
};

AVDM_KNIFE_FN v3 nrm(v3 a)
{
    const float dInv = 1.0f / sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return v3{a.x * dInv, a.y * dInv, a.z * dInv};
}
AVDM_KNIFE_FN float dot3(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
AVDM_KNIFE_FN v3 iPmul(const float* M, float vx, float vy)
{
    return v3{M[0] * vx + M[3] * vy + M[6], M[1] * vx + M[4] * vy + M[7], M[2] * vx + M[5] * vy + M[8]};
}
AVDM_KNIFE_FN bool inside(const float* P, v3 V, float dd, float W1, float H1)
{
    const float qx = P[0] * V.x + P[3] * V.y + P[6] * V.z + P[9], qy = P[1] * V.x + P[4] * V.y + P[7] * V.z + P[10],
                qz = P[2] * V.x + P[5] * V.y + P[8] * V.z + P[11];
    const float inv = 1.0f / qz;
    const float rx = qx * inv, ry = qy * inv;
    return !((rx < dd) || (rx > W1 - dd) || (ry < dd) || (ry > H1 - dd));
}
// SGM: the ray of pixel (x, y) cut with the fronto-parallel plane at `depth`
AVDM_KNIFE_FN bool sgm_r_inside(const float* P, const float* iP, const float* Cc, const float* Zv, float x, float y, float depth, float dd, float W1,
                                float H1)
{
    const v3 C = v3{Cc[0], Cc[1], Cc[2]}, Z = v3{Zv[0], Zv[1], Zv[2]};
    const v3 planep = v3{C.x + Z.x * depth, C.y + Z.y * depth, C.z + Z.z * depth};
    const v3 v = nrm(iPmul(iP, x, y));
    const float k = (dot3(planep, Z) - dot3(Z, C)) / dot3(Z, v); // linePlaneIntersect
    const v3 p = v3{C.x + v.x * k, C.y + v.y * k, C.z + v.z * k};
    return inside(P, p, dd, W1, H1);
}
// Refine: the point at the SGM depth on the ray, moved by rel pixel sizes along it
AVDM_KNIFE_FN bool refine_r_inside(const float* P, const float* iP, const float* Cc, float x, float y, float depth, float pixSize, int rel, float dd,
                                   float W1, float H1)
{
    const v3 C = v3{Cc[0], Cc[1], Cc[2]};
    const v3 v = nrm(iPmul(iP, x, y));
    v3 p = v3{C.x + v.x * depth, C.y + v.y * depth, C.z + v.z * depth};
    if(rel != 0)
    {
        const v3 d = nrm(v3{p.x - C.x, p.y - C.y, p.z - C.z});
        const float m = (float)rel * pixSize;
        p = v3{p.x + d.x * m, p.y + d.y * m, p.z + d.z * m};
    }
    return inside(P, p, dd, W1, H1);
}

} // namespace knife
