/* STAND-IN for aliceVision/numeric/numeric.hpp (Eigen-based in the reference): only what DeviceMipmapImage.cpp, TileParams.cpp and the
 * tile-merge functions of mapIO.cpp use.  divideRoundUp restates numeric.hpp:487-505 for the positive operands those files pass
 * (ceiling division), clamp restates numeric.hpp:137-142.  oracle/_ref test infrastructure only. */
#pragma once
#include <algorithm>
namespace aliceVision {
template <typename T>
inline T divideRoundUp(T x, T y) { return x / y + T((x % y) != 0); }
template <typename T>
inline T clamp(const T& val, const T& min, const T& max) { return std::max(min, std::min(val, max)); }
}
