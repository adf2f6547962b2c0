/* STAND-IN for CUDA's math_constants.h (oracle/_ref test infrastructure; see cuda_runtime.h in this directory) */
#ifndef AVDM_REF_SHIM_MATH_CONSTANTS_H
#define AVDM_REF_SHIM_MATH_CONSTANTS_H
#include <cmath>
#include <limits>
#define CUDART_INF_F (std::numeric_limits<float>::infinity())
#define CUDART_NAN_F (std::numeric_limits<float>::quiet_NaN())
#define CUDART_PI_F 3.141592654f
#define CUDART_PI 3.1415926535897931e+0
#define CUDART_INF (std::numeric_limits<double>::infinity())
#endif
