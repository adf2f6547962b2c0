// oracle/_ref (host): STAND-IN for aliceVision/image/Image.hpp (an Eigen matrix in the reference): row-major pixels with the accessors the
// code under test uses.  Test infrastructure only.
#pragma once
#include <cstddef>
#include <vector>

#include <aliceVision/numeric/numeric.hpp> // image/Image.hpp:10 of the reference includes it too (divideRoundUp, clamp reach Sgm.cpp this way)

namespace aliceVision {
namespace image {
template <class T>
class Image
{
  public:
    Image() = default;
    Image(int width, int height, bool fInit = false, const T val = T()) : _w(width), _h(height), _d((size_t)width * height, fInit ? val : T()) {}
    int width() const { return _w; }
    int height() const { return _h; }
    int size() const { return _w * _h; }
    T& operator()(int y, int x) { return _d[(size_t)y * _w + x]; }
    const T& operator()(int y, int x) const { return _d[(size_t)y * _w + x]; }
    T& operator()(int i) { return _d[i]; }
    const T& operator()(int i) const { return _d[i]; }
    T* data() { return _d.data(); }
    const T* data() const { return _d.data(); }

  private:
    int _w = 0, _h = 0;
    std::vector<T> _d;
};
enum class EImageColorSpace { LINEAR, NO_CONVERSION };
enum class EStorageDataType { Float };
struct ImageWriteOptions
{
    ImageWriteOptions& toColorSpace(EImageColorSpace) { return *this; }
    ImageWriteOptions& storageDataType(EStorageDataType) { return *this; }
};
struct RGBfColor
{
    float v[3] = {0.f, 0.f, 0.f};
    float& r() { return v[0]; }
    float& g() { return v[1]; }
    float& b() { return v[2]; }
};
struct RGBAfColor
{
    float v[4] = {0.f, 0.f, 0.f, 0.f};
};
} // namespace image
} // namespace aliceVision
