// oracle/_ref (host): STAND-IN for aliceVision/image/Image.hpp (an Eigen matrix in the reference): row-major pixels with the accessors the
// code under test uses.  Test infrastructure only.
#pragma once
#include <cstddef>
#include <vector>

#include <aliceVision/image/pixelTypes.hpp>
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
    bool contains(int y, int x) const { return 0 <= x && x < _w && 0 <= y && y < _h; } // image/Image.hpp:178
    void resize(int width, int height, bool fInit = true, const T& val = T())
    {
        _w = width, _h = height;
        _d.assign((size_t)width * height, fInit ? val : T());
    }
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
} // namespace image
} // namespace aliceVision
