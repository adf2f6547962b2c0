// oracle/_ref (host): STAND-IN for aliceVision/image/pixelTypes.hpp (Eigen vectors in the reference): RGB / RGBA pixels with the few
// operations the reference's image/Sampler.hpp performs on them — conversion to double, product with a scalar weight, accumulation,
// division by the total weight — one IEEE operation per channel each, like the Eigen expressions.  Test infrastructure only.
#pragma once
namespace aliceVision {
namespace image {
template <class T>
struct Rgb
{
    T v[3] = {T(), T(), T()};
    Rgb() = default;
    Rgb(T r, T g, T b) : v{r, g, b} {}
    static Rgb Zero() { return Rgb(); }
    T r() const { return v[0]; }
    T g() const { return v[1]; }
    T b() const { return v[2]; }
    template <class U>
    Rgb<U> cast() const { return Rgb<U>((U)v[0], (U)v[1], (U)v[2]); }
    Rgb operator*(double w) const { return Rgb((T)(v[0] * w), (T)(v[1] * w), (T)(v[2] * w)); }
    Rgb& operator+=(const Rgb& o) { v[0] += o.v[0], v[1] += o.v[1], v[2] += o.v[2]; return *this; }
    Rgb& operator/=(double d) { v[0] /= d, v[1] /= d, v[2] /= d; return *this; }
};
template <class T>
struct Rgba
{
    T v[4] = {T(), T(), T(), T()};
    Rgba() = default;
    Rgba(T r, T g, T b, T a) : v{r, g, b, a} {}
    static Rgba Zero() { return Rgba(); }
    T r() const { return v[0]; }
    T g() const { return v[1]; }
    T b() const { return v[2]; }
    T a() const { return v[3]; }
    template <class U>
    Rgba<U> cast() const { return Rgba<U>((U)v[0], (U)v[1], (U)v[2], (U)v[3]); }
    Rgba operator*(double w) const { return Rgba((T)(v[0] * w), (T)(v[1] * w), (T)(v[2] * w), (T)(v[3] * w)); }
    Rgba& operator+=(const Rgba& o) { v[0] += o.v[0], v[1] += o.v[1], v[2] += o.v[2], v[3] += o.v[3]; return *this; }
    Rgba& operator/=(double d) { v[0] /= d, v[1] /= d, v[2] /= d, v[3] /= d; return *this; }
};
using RGBfColor = Rgb<float>;
using RGBAfColor = Rgba<float>;
} // namespace image
} // namespace aliceVision
