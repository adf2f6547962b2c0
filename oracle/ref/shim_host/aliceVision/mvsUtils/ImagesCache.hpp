// oracle/_ref (host): stand-in for aliceVision/mvsUtils/ImagesCache.hpp (only named in declarations the code under test includes)
#pragma once
namespace aliceVision { namespace mvsUtils {
template <class T>
class ImagesCache;
}} // namespace aliceVision::mvsUtils
