// oracle/_ref (fuse): stand-in for <boost/math/constants/constants.hpp> (mvsData/Matrix3x3.hpp uses pi<double>() in its Euler-angle helpers)
#pragma once
namespace boost { namespace math { namespace constants {
template <class T>
constexpr T pi() { return static_cast<T>(3.141592653589793238462643383279502884L); }
}}} // namespace boost::math::constants
