// mvsData.hpp — the small double-precision geometry vocabulary of the host side: points, 3x3 / 3x4 matrices, pixel,
// Range / ROI, and the handful of geometric predicates the depth-list and camera-selection code uses.
// Restates (behaviour, not code) mvsData/{Point2d,Point3d,Matrix3x3,Matrix3x4,Pixel,ROI,geometry}.hpp|cpp of the reference;
// every function cites the lines it follows (paths relative to /root/reference/src/aliceVision).
#pragma once

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <ostream>
#include <stdexcept>

namespace avdm_host {

struct Point2d
{
    double x = 0.0, y = 0.0;
    Point2d() = default;
    Point2d(double x_, double y_) : x(x_), y(y_) {}
    Point2d operator+(const Point2d& o) const { return {x + o.x, y + o.y}; }
    Point2d operator-(const Point2d& o) const { return {x - o.x, y - o.y}; }
    Point2d operator*(double s) const { return {x * s, y * s}; }
    double size() const { return std::sqrt(x * x + y * y); }
    Point2d normalize() const
    {
        const double d = size();
        return {x / d, y / d};
    }
};

struct Point3d
{
    double x = 0.0, y = 0.0, z = 0.0;
    Point3d() = default;
    Point3d(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    Point3d operator+(const Point3d& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Point3d operator-(const Point3d& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Point3d operator*(double s) const { return {x * s, y * s, z * s}; }
    Point3d operator/(double s) const { return {x / s, y / s, z / s}; }
    double size() const { return std::sqrt(x * x + y * y + z * z); }
    Point3d normalize() const
    {
        const double d = size();
        return {x / d, y / d, z / d};
    }
};
inline double dot(const Point3d& a, const Point3d& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Point3d cross(const Point3d& a, const Point3d& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
// mvsData/Point3d.hpp:146
inline Point3d proj(const Point3d& e, const Point3d& a) { return e * (dot(e, a) / dot(e, e)); }

// mvsData/Pixel.hpp: integer pixel; construction from a Point2d truncates (static_cast<int>)
struct Pixel
{
    int x = 0, y = 0;
    Pixel() = default;
    Pixel(int x_, int y_) : x(x_), y(y_) {}
    // rounds half up like the reference (mvsData/Pixel.hpp:30-34) — found by the pin against the reference's own SgmDepthList.cpp
    // (tests/test_host_ref.py): truncation kept / dropped other epipolar samples at the border of the T image
    explicit Pixel(const Point2d& p) : x(static_cast<int>(std::floor(p.x + 0.5))), y(static_cast<int>(std::floor(p.y + 0.5))) {}
};

// row-major 3x3, m[3*r + c]
struct Matrix3x3
{
    double m[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double& operator()(int r, int c) { return m[3 * r + c]; }
    double operator()(int r, int c) const { return m[3 * r + c]; }

    static Matrix3x3 diag(double a, double b, double c)
    {
        Matrix3x3 d;
        d(0, 0) = a;
        d(1, 1) = b;
        d(2, 2) = c;
        return d;
    }
    Matrix3x3 operator*(const Matrix3x3& o) const
    {
        Matrix3x3 r;
        for(int i = 0; i < 3; ++i)
            for(int j = 0; j < 3; ++j)
                r(i, j) = (*this)(i, 0) * o(0, j) + (*this)(i, 1) * o(1, j) + (*this)(i, 2) * o(2, j);
        return r;
    }
    Matrix3x3 operator/(double s) const
    {
        Matrix3x3 r;
        for(int i = 0; i < 9; ++i)
            r.m[i] = m[i] / s;
        return r;
    }
    Matrix3x3 operator-() const
    {
        Matrix3x3 r;
        for(int i = 0; i < 9; ++i)
            r.m[i] = -m[i];
        return r;
    }
    Point3d operator*(const Point3d& p) const
    {
        return {m[0] * p.x + m[1] * p.y + m[2] * p.z, m[3] * p.x + m[4] * p.y + m[5] * p.z, m[6] * p.x + m[7] * p.y + m[8] * p.z};
    }
    // homogeneous pixel (x, y, 1)
    Point3d operator*(const Point2d& p) const { return {m[0] * p.x + m[1] * p.y + m[2], m[3] * p.x + m[4] * p.y + m[5], m[6] * p.x + m[7] * p.y + m[8]}; }

    double det() const
    {
        return m[0] * (m[8] * m[4] - m[7] * m[5]) - m[3] * (m[8] * m[1] - m[7] * m[2]) + m[6] * (m[5] * m[1] - m[4] * m[2]);
    }
    // mvsData/Matrix3x3.hpp:268-287 (adjugate / determinant; a singular matrix yields a zero matrix like the reference's default-constructed result)
    Matrix3x3 inverse() const
    {
        Matrix3x3 o;
        const double dt = det();
        if(std::fabs(dt) < 0.00000001f || std::isnan(dt))
            return o;
        const double m11 = m[0], m12 = m[1], m13 = m[2], m21 = m[3], m22 = m[4], m23 = m[5], m31 = m[6], m32 = m[7], m33 = m[8];
        o.m[0] = (m33 * m22 - m32 * m23) / dt;
        o.m[1] = -(m33 * m12 - m32 * m13) / dt;
        o.m[2] = (m23 * m12 - m22 * m13) / dt;
        o.m[3] = -(m33 * m21 - m31 * m23) / dt;
        o.m[4] = (m33 * m11 - m31 * m13) / dt;
        o.m[5] = -(m23 * m11 - m21 * m13) / dt;
        o.m[6] = (m32 * m21 - m31 * m22) / dt;
        o.m[7] = -(m32 * m11 - m31 * m12) / dt;
        o.m[8] = (m22 * m11 - m21 * m12) / dt;
        return o;
    }
    // mvsData/Matrix3x3.hpp:193-265: RQ by Gram-Schmidt on the rows taken bottom-up (upper-triangular R, orthonormal Q)
    void RQ(Matrix3x3& R, Matrix3x3& Q) const
    {
        const Point3d a1(m[6], m[7], m[8]), a2(m[3], m[4], m[5]), a3(m[0], m[1], m[2]);
        const Point3d e1 = a1.normalize();
        const Point3d e2 = (a2 - proj(e1, a2)).normalize();
        const Point3d e3 = (a3 - proj(e1, a3) - proj(e2, a3)).normalize();
        Q.m[0] = e3.x, Q.m[1] = e3.y, Q.m[2] = e3.z;
        Q.m[3] = e2.x, Q.m[4] = e2.y, Q.m[5] = e2.z;
        Q.m[6] = e1.x, Q.m[7] = e1.y, Q.m[8] = e1.z;
        // the triangular factor of the flipped problem, flipped back
        R.m[0] = dot(e3, a3), R.m[1] = dot(e2, a3), R.m[2] = dot(e1, a3);
        R.m[3] = 0.0, R.m[4] = dot(e2, a2), R.m[5] = dot(e1, a2);
        R.m[6] = 0.0, R.m[7] = 0.0, R.m[8] = dot(e1, a1);
    }
};

// row-major 3x4, m[4*r + c].  NOTE the reference's Matrix3x4::m is laid out m11,m12,m13,m14,m21,... too (mvsData/Matrix3x4.hpp:22-37),
// which is what makes "for i < 8: m[i] /= scale" scale the first two ROWS (MultiViewParams.cpp:281-283).
struct Matrix3x4
{
    double m[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double& operator()(int r, int c) { return m[4 * r + c]; }
    double operator()(int r, int c) const { return m[4 * r + c]; }
    Matrix3x3 sub3x3() const
    {
        Matrix3x3 s;
        for(int r = 0; r < 3; ++r)
            for(int c = 0; c < 3; ++c)
                s(r, c) = (*this)(r, c);
        return s;
    }
    Point3d lastColumn() const { return {m[3], m[7], m[11]}; }
    Point3d operator*(const Point3d& p) const
    {
        return {m[0] * p.x + m[1] * p.y + m[2] * p.z + m[3], m[4] * p.x + m[5] * p.y + m[6] * p.z + m[7], m[8] * p.x + m[9] * p.y + m[10] * p.z + m[11]};
    }
    // mvsData/Matrix3x4.hpp:80-114
    void decomposeProjectionMatrix(Matrix3x3& K, Matrix3x3& R, Point3d& C) const
    {
        const Matrix3x3 H = sub3x3();
        H.RQ(K, R);
        if(K(2, 2) == 0)
            throw std::runtime_error("Matrix3x4::decomposeProjectionMatrix: affine camera.");
        K = K / std::fabs(K(2, 2));
        if(K(0, 0) < 0.0)
        {
            const Matrix3x3 D = Matrix3x3::diag(-1.0, -1.0, 1.0);
            K = K * D;
            R = D * R;
        }
        if(K(1, 1) < 0.0)
        {
            const Matrix3x3 D = Matrix3x3::diag(1.0, -1.0, -1.0);
            K = K * D;
            R = D * R;
        }
        const Matrix3x3 nH = -sub3x3();
        if(std::fabs(nH.det()) < 0.00000001f || std::isnan(nH.det()))
            throw std::runtime_error("Matrix is singular.");
        C = nH.inverse() * lastColumn();
    }
};

// K * [R | t]  (mvsData/Matrix3x4.hpp:117-155)
inline Matrix3x4 composeP(const Matrix3x3& K, const Matrix3x3& R, const Point3d& t)
{
    Matrix3x4 P;
    const Matrix3x3 KR = K * R;
    const Point3d Kt = K * t;
    for(int r = 0; r < 3; ++r)
        for(int c = 0; c < 3; ++c)
            P(r, c) = KR(r, c);
    P(0, 3) = Kt.x, P(1, 3) = Kt.y, P(2, 3) = Kt.z;
    return P;
}

// ---- Range / ROI (mvsData/ROI.hpp:35-200): half-open unsigned ranges ----
struct Range
{
    unsigned int begin = 0, end = 0;
    Range() = default;
    Range(unsigned int b, unsigned int e) : begin(b), end(e) {}
    unsigned int size() const { return end - begin; }
    bool isEmpty() const { return begin >= end; }
    bool contains(unsigned int i) const { return begin <= i && end > i; }
};
struct ROI
{
    Range x, y;
    ROI() = default;
    ROI(unsigned int bx, unsigned int ex, unsigned int by, unsigned int ey) : x(bx, ex), y(by, ey) {}
    ROI(const Range& rx, const Range& ry) : x(rx), y(ry) {}
    unsigned int width() const { return x.size(); }
    unsigned int height() const { return y.size(); }
    bool isEmpty() const { return x.isEmpty() || y.isEmpty(); }
    bool contains(unsigned int px, unsigned int py) const { return x.contains(px) && y.contains(py); }
};
inline Range intersect(const Range& a, const Range& b) { return Range(std::max(a.begin, b.begin), std::min(a.end, b.end)); }
inline ROI intersect(const ROI& a, const ROI& b) { return ROI(intersect(a.x, b.x), intersect(a.y, b.y)); }
// ROI.hpp:147-161: float division, floor / ceil
inline Range downscaleRange(const Range& r, float d) { return Range((unsigned)std::floor(r.begin / d), (unsigned)std::ceil(r.end / d)); }
inline Range upscaleRange(const Range& r, float u) { return Range((unsigned)std::floor(r.begin * u), (unsigned)std::ceil(r.end * u)); }
inline ROI downscaleROI(const ROI& r, float d) { return ROI(downscaleRange(r.x, d), downscaleRange(r.y, d)); }
inline ROI upscaleROI(const ROI& r, float u) { return ROI(upscaleRange(r.x, u), upscaleRange(r.y, u)); }
inline std::ostream& operator<<(std::ostream& os, const Range& r) { return os << r.begin << "-" << r.end; }
inline std::ostream& operator<<(std::ostream& os, const ROI& r) { return os << "x: " << r.x << ", y: " << r.y; }

inline int divideRoundUp(int x, int n) { return (x + n - 1) / n; }

// ---- geometry (mvsData/geometry.cpp) ----
// :14-17
inline double pointLineDistance3D(const Point3d& point, const Point3d& linePoint, const Point3d& lineVectNormalized)
{
    return cross(lineVectNormalized, linePoint - point).size();
}
// :24-27
inline double pointPlaneDistance(const Point3d& point, const Point3d& planePoint, const Point3d& planeNormal)
{
    return std::fabs(dot(point, planeNormal) - dot(planePoint, planeNormal)) / std::sqrt(dot(planeNormal, planeNormal));
}
// :29-32
inline double orientedPointPlaneDistance(const Point3d& point, const Point3d& planePoint, const Point3d& planeNormal)
{
    return (dot(point, planeNormal) - dot(planePoint, planeNormal)) / std::sqrt(dot(planeNormal, planeNormal));
}
// :203-213 (degrees)
inline double angleBetwV1andV2(const Point3d& iV1, const Point3d& iV2)
{
    const Point3d V1 = iV1.normalize(), V2 = iV2.normalize();
    const double a = std::acos(V1.x * V2.x + V1.y * V2.y + V1.z * V2.z);
    if(std::isnan(a))
        return 0.0;
    return std::fabs(a / (M_PI / 180.0));
}
// :50-145: midpoint of the shortest segment between lines p1p2 and p3p4 (Bourke)
inline bool lineLineIntersect(Point3d& out, const Point3d& p1, const Point3d& p2, const Point3d& p3, const Point3d& p4)
{
    const Point3d p13 = p1 - p3, p43 = p4 - p3, p21 = p2 - p1;
    if(std::fabs(p43.x) < FLT_EPSILON && std::fabs(p43.y) < FLT_EPSILON && std::fabs(p43.z) < FLT_EPSILON)
        return false;
    if(std::fabs(p21.x) < FLT_EPSILON && std::fabs(p21.y) < FLT_EPSILON && std::fabs(p21.z) < FLT_EPSILON)
        return false;
    const double d1343 = dot(p13, p43), d4321 = dot(p43, p21), d1321 = dot(p13, p21), d4343 = dot(p43, p43), d2121 = dot(p21, p21);
    const double denom = d2121 * d4343 - d4321 * d4321;
    if(std::fabs(denom) < FLT_EPSILON)
        return false;
    const double numer = d1343 * d4321 - d1321 * d4343;
    const double mua = numer / denom;
    const double mub = (d1343 + d4321 * mua) / d4343;
    const Point3d pa = p1 + p21 * mua, pb = p3 + p43 * mub;
    out = Point3d((pa.x + pb.x) / 2.0, (pa.y + pb.y) / 2.0, (pa.z + pb.z) / 2.0);
    return true;
}

} // namespace avdm_host
