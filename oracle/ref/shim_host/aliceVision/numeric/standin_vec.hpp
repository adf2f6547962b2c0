// oracle/_ref (host): stand-ins for the Eigen vectors (Vec2 / Vec3 of aliceVision/numeric/numeric.hpp) as the code under test uses them:
// element access, sum, difference, product with a scalar, component-wise product.  One IEEE operation per component each, like the
// reference's Eigen expressions compiled without contraction.  Test infrastructure only.
#pragma once
namespace aliceVision {
struct Vec2
{
    double v[2] = {0.0, 0.0};
    Vec2() = default;
    Vec2(double a, double b) : v{a, b} {}
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double operator()(int i) const { return v[i]; }
    double& operator()(int i) { return v[i]; }
    Vec2 operator+(const Vec2& o) const { return Vec2(v[0] + o.v[0], v[1] + o.v[1]); }
    Vec2 operator-(const Vec2& o) const { return Vec2(v[0] - o.v[0], v[1] - o.v[1]); }
    Vec2 operator*(double s) const { return Vec2(v[0] * s, v[1] * s); }
    Vec2 cwiseProduct(const Vec2& o) const { return Vec2(v[0] * o.v[0], v[1] * o.v[1]); }
};
inline Vec2 operator*(double s, const Vec2& p) { return Vec2(s * p.v[0], s * p.v[1]); }
struct Vec3
{
    double v[3] = {0.0, 0.0, 0.0};
    double operator()(int i) const { return v[i]; }
};
} // namespace aliceVision
