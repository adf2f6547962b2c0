/* STAND-IN for aliceVision/numeric/numeric.hpp (Eigen-based in the reference): only what DeviceMipmapImage.cpp uses.
 * divideRoundUp restates numeric.hpp:487-505 for the positive operands that file passes (ceiling division).
 * oracle/_ref test infrastructure only. */
#pragma once
namespace aliceVision {
template <typename T>
inline T divideRoundUp(T x, T y) { return x / y + T((x % y) != 0); }
}
