/*
 * cuda_runtime.h — STAND-IN for the CUDA runtime header, so that the reference's depthMap kernel-launch layer
 * (/root/reference/src/aliceVision/depthMap/cuda: device/*.cuh, planeSweeping/*.cuh|.cu, imageProcessing/*.cu,
 * host/DeviceMipmapImage.cpp, host/memory.hpp) compiles UNCHANGED with g++ and runs on the CPU.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/_ref): nothing under alicevision_amd/ or include/ uses this.  Written from the public CUDA
 * runtime API / programming guide; it contains no reference code.  What it provides:
 *   - the execution model: __global__ functions are plain functions; shim::launch() runs them for every (block, thread) with
 *     thread-local blockIdx / threadIdx (the reference's kernels use neither shared memory nor __syncthreads);
 *   - device memory = host memory (cudaMalloc* -> aligned_alloc, cudaMemcpy* -> memcpy);
 *   - vector types and the handful of math intrinsics the kernels call (fast-math intrinsics evaluate as the exact operation);
 *   - CUDA arrays / mip-mapped arrays / texture + surface objects with the texture unit's documented filtering
 *     (CUDA C Programming Guide, "Texture Fetching": point and linear filtering, normalized coordinates, clamp addressing,
 *     mip-linear; linear weights optionally stored in 9-bit fixed point with 8 fractional bits: shim::g_fixed8).
 * The texture unit and the intrinsics are the third-party arithmetic SURVEY.md §8(c) names: restated, not pinned.
 */
#ifndef AVDM_REF_SHIM_CUDA_RUNTIME_H
#define AVDM_REF_SHIM_CUDA_RUNTIME_H

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#define __host__
#define __device__
#define __global__
#define __constant__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))

/* ---- vector types ---- */
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct __attribute__((aligned(16))) float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct uchar4 { unsigned char x, y, z, w; };
struct ushort4 { unsigned short x, y, z, w; };
struct double3 { double x, y, z; };
struct dim3
{
    unsigned int x, y, z;
    constexpr dim3(unsigned int vx = 1, unsigned int vy = 1, unsigned int vz = 1) : x(vx), y(vy), z(vz) {}
};
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uchar4 make_uchar4(unsigned char x, unsigned char y, unsigned char z, unsigned char w) { return uchar4{x, y, z, w}; }
static inline double3 make_double3(double x, double y, double z) { return double3{x, y, z}; }

/* ---- execution model ---- */
namespace shim {
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern bool g_fixed8; /* texture linear-filter weights quantised to 1.8 fixed point (like the hardware) or kept in fp32 */
}
#define threadIdx (::shim::t_threadIdx)
#define blockIdx (::shim::t_blockIdx)
#define blockDim (::shim::t_blockDim)
#define gridDim (::shim::t_gridDim)

/* ---- math the kernels call (CUDA's global-namespace overloads) ---- */
using std::isfinite;
using std::isinf;
using std::isnan;
#ifdef AVDM_SHIM_CUDA_FASTMATH
/* Second evaluation mode of the stand-in (oracle/_ref/libavdm_ref_fm.so, libavdm_ref_cuda.so): the fast intrinsics with the ERROR MODEL the
 * CUDA C Programming Guide documents for them ("Intrinsic Functions": __expf(x) = ex2.approx(x * log2(e)), __fdividef(x, y) = x * rcp.approx(y),
 * __powf(x, y) = ex2.approx(y * lg2.approx(x))), each approximate instruction evaluated as the correctly rounded fp32 operation on its fp32
 * operand (the hardware's own 1-2 ulp are not reproducible here) with denormal results flushed like the .ftz forms.  Not "the" CUDA result:
 * a second FAITHFUL evaluation of the reference's source, next to the exact-operation mode above; the distance between the two is the
 * platform spread of the reference itself (tests/test_platform_spread.py). */
static inline float shim_ftz(float v) { return std::fabs(v) < 1.17549435e-38f ? (v < 0.f ? -0.f : 0.f) : v; }
static inline float shim_ex2_approx(float t) { return shim_ftz(exp2f(t)); }
static inline float shim_lg2_approx(float v) { return log2f(v); }
static inline float shim_rcp_approx(float v) { return shim_ftz(1.0f / v); }
static inline float shim_fast_expf(float x)
{
    volatile float t = x * 1.44269504088896340736f; /* the product is rounded to fp32 before ex2 (volatile: never fused into what follows) */
    return shim_ex2_approx(t);
}
static inline float shim_fast_powf(float x, float y)
{
    volatile float t = y * shim_lg2_approx(x);
    return shim_ex2_approx(t);
}
#define __expf(x) shim_fast_expf(x)
#define __powf(x, y) shim_fast_powf((x), (y))
static inline float __fdividef(float a, float b) { return a * shim_rcp_approx(b); }
#else
/* glibc's math.h already declares __expf / __powf (its internal aliases of expf / powf): same functions, map by macro */
#define __expf(x) expf(x)
#define __powf(x, y) powf((x), (y))
static inline float __fdividef(float a, float b) { return a / b; }
#endif
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
static inline float __saturatef(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float norm3df(float a, float b, float c) { return sqrtf(a * a + b * b + c * c); }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
static inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
static inline long long max(long long a, long long b) { return a > b ? a : b; }
static inline long long min(long long a, long long b) { return a < b ? a : b; }
static inline size_t max(size_t a, size_t b) { return a > b ? a : b; }
static inline size_t min(size_t a, size_t b) { return a < b ? a : b; }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline float min(float a, float b) { return fminf(a, b); }
static inline double max(double a, double b) { return fmax(a, b); }
static inline double min(double a, double b) { return fmin(a, b); }
static inline float max(float a, int b) { return fmaxf(a, (float)b); }
static inline float max(int a, float b) { return fmaxf((float)a, b); }
static inline float min(float a, int b) { return fminf(a, (float)b); }
static inline float min(int a, float b) { return fminf((float)a, b); }

/* ---- runtime API ---- */
typedef int cudaError_t;
enum { cudaSuccess = 0, cudaErrorMemoryAllocation = 2, cudaErrorInvalidValue = 11 };
typedef struct shimStream* cudaStream_t;
enum cudaMemcpyKind { cudaMemcpyHostToHost = 0, cudaMemcpyHostToDevice = 1, cudaMemcpyDeviceToHost = 2, cudaMemcpyDeviceToDevice = 3, cudaMemcpyDefault = 4 };
const char* cudaGetErrorString(cudaError_t e);
cudaError_t cudaGetLastError();
cudaError_t cudaDeviceSynchronize();
cudaError_t cudaGetDevice(int* dev);
cudaError_t cudaSetDevice(int dev);
cudaError_t cudaGetDeviceCount(int* n);
cudaError_t cudaMemGetInfo(size_t* freeB, size_t* totalB);
cudaError_t shimMalloc(void** p, size_t bytes);
cudaError_t cudaFree(void* p);
cudaError_t cudaFreeHost(void* p);
template <class T> static inline cudaError_t cudaMalloc(T** p, size_t bytes) { return shimMalloc((void**)p, bytes); }
template <class T> static inline cudaError_t cudaMallocHost(T** p, size_t bytes) { return shimMalloc((void**)p, bytes); }
cudaError_t shimMallocPitch(void** p, size_t* pitch, size_t widthBytes, size_t height);
template <class T> static inline cudaError_t cudaMallocPitch(T** p, size_t* pitch, size_t w, size_t h) { return shimMallocPitch((void**)p, pitch, w, h); }
struct cudaExtent { size_t width, height, depth; };
static inline cudaExtent make_cudaExtent(size_t w, size_t h, size_t d) { return cudaExtent{w, h, d}; }
struct cudaPitchedPtr { void* ptr; size_t pitch, xsize, ysize; };
struct cudaPos { size_t x, y, z; };
cudaError_t cudaMalloc3D(cudaPitchedPtr* p, cudaExtent e);
cudaError_t cudaMemcpy(void* dst, const void* src, size_t n, cudaMemcpyKind k);
cudaError_t cudaMemcpyAsync(void* dst, const void* src, size_t n, cudaMemcpyKind k, cudaStream_t s = 0);
cudaError_t cudaMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t widthBytes, size_t height, cudaMemcpyKind k);
cudaError_t cudaMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t widthBytes, size_t height, cudaMemcpyKind k, cudaStream_t s = 0);
template <class T>
static inline cudaError_t cudaMemcpyToSymbol(T&& symbol, const void* src, size_t count, size_t offset = 0, cudaMemcpyKind = cudaMemcpyHostToDevice)
{
    /* the symbol itself (array or struct) or its address: both forms are accepted by the runtime */
    if constexpr(std::is_pointer<typename std::remove_reference<T>::type>::value)
        memcpy((char*)symbol + offset, src, count);
    else
        memcpy((char*)&symbol + offset, src, count);
    return cudaSuccess;
}
template <class T>
static inline cudaError_t cudaOccupancyMaxPotentialBlockSize(int* minGridSize, int* blockSize, T, size_t = 0, int = 0)
{
    *minGridSize = 1;
    *blockSize = 256;
    return cudaSuccess;
}

/* ---- arrays, textures, surfaces ---- */
enum cudaChannelFormatKind { cudaChannelFormatKindSigned = 0, cudaChannelFormatKindUnsigned = 1, cudaChannelFormatKindFloat = 2, cudaChannelFormatKindNone = 3 };
struct cudaChannelFormatDesc { int x, y, z, w; cudaChannelFormatKind f; };
static inline cudaChannelFormatDesc cudaCreateChannelDescHalf4() { return cudaChannelFormatDesc{16, 16, 16, 16, cudaChannelFormatKindFloat}; }
template <class T> static inline cudaChannelFormatDesc cudaCreateChannelDesc();
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<float>() { return cudaChannelFormatDesc{32, 0, 0, 0, cudaChannelFormatKindFloat}; }
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<float2>() { return cudaChannelFormatDesc{32, 32, 0, 0, cudaChannelFormatKindFloat}; }
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<float4>() { return cudaChannelFormatDesc{32, 32, 32, 32, cudaChannelFormatKindFloat}; }
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<uchar4>() { return cudaChannelFormatDesc{8, 8, 8, 8, cudaChannelFormatKindUnsigned}; }
template <> inline cudaChannelFormatDesc cudaCreateChannelDesc<unsigned char>() { return cudaChannelFormatDesc{8, 0, 0, 0, cudaChannelFormatKindUnsigned}; }

struct cudaArray;           /* one 2-D level: owned storage, tightly packed rows */
struct cudaMipmappedArray;  /* levels with floor-halved sizes */
typedef cudaArray* cudaArray_t;
typedef const cudaArray* cudaArray_const_t;
typedef cudaMipmappedArray* cudaMipmappedArray_t;
typedef const cudaMipmappedArray* cudaMipmappedArray_const_t;
cudaError_t cudaMallocMipmappedArray(cudaMipmappedArray_t* out, const cudaChannelFormatDesc* desc, cudaExtent extent, unsigned int numLevels, unsigned int flags = 0);
cudaError_t cudaFreeMipmappedArray(cudaMipmappedArray_t a);
cudaError_t cudaGetMipmappedArrayLevel(cudaArray_t* level, cudaMipmappedArray_const_t a, unsigned int l);
cudaError_t cudaArrayGetInfo(cudaChannelFormatDesc* desc, cudaExtent* extent, unsigned int* flags, cudaArray_t a);
struct cudaMemcpy3DParms
{
    cudaArray_t srcArray;
    cudaPos srcPos;
    cudaPitchedPtr srcPtr;
    cudaArray_t dstArray;
    cudaPos dstPos;
    cudaPitchedPtr dstPtr;
    cudaExtent extent;
    cudaMemcpyKind kind;
};
cudaError_t cudaMemcpy3D(const cudaMemcpy3DParms* p);

enum cudaResourceType { cudaResourceTypeArray = 0, cudaResourceTypeMipmappedArray = 1, cudaResourceTypeLinear = 2, cudaResourceTypePitch2D = 3 };
enum cudaTextureAddressMode { cudaAddressModeWrap = 0, cudaAddressModeClamp = 1, cudaAddressModeMirror = 2, cudaAddressModeBorder = 3 };
enum cudaTextureFilterMode { cudaFilterModePoint = 0, cudaFilterModeLinear = 1 };
enum cudaTextureReadMode { cudaReadModeElementType = 0, cudaReadModeNormalizedFloat = 1 };
struct cudaResourceDesc
{
    cudaResourceType resType;
    union
    {
        struct { cudaArray_t array; } array;
        struct { cudaMipmappedArray_t mipmap; } mipmap;
        struct { void* devPtr; cudaChannelFormatDesc desc; size_t sizeInBytes; } linear;
        struct { void* devPtr; cudaChannelFormatDesc desc; size_t width, height, pitchInBytes; } pitch2D;
    } res;
};
struct cudaTextureDesc
{
    cudaTextureAddressMode addressMode[3];
    cudaTextureFilterMode filterMode;
    cudaTextureReadMode readMode;
    int sRGB;
    float borderColor[4];
    int normalizedCoords;
    unsigned int maxAnisotropy;
    cudaTextureFilterMode mipmapFilterMode;
    float mipmapLevelBias, minMipmapLevelClamp, maxMipmapLevelClamp;
    int disableTrilinearOptimization, seamlessCubemap;
};
struct cudaResourceViewDesc;
typedef unsigned long long cudaTextureObject_t;
typedef unsigned long long cudaSurfaceObject_t;
cudaError_t cudaCreateTextureObject(cudaTextureObject_t* out, const cudaResourceDesc* res, const cudaTextureDesc* tex, const cudaResourceViewDesc* view);
cudaError_t cudaDestroyTextureObject(cudaTextureObject_t t);
cudaError_t cudaCreateSurfaceObject(cudaSurfaceObject_t* out, const cudaResourceDesc* res);
cudaError_t cudaDestroySurfaceObject(cudaSurfaceObject_t s);

namespace shim {
float4 tex_fetch(cudaTextureObject_t t, float x, float y, float lod);      /* all channels, as float */
void surf_write(cudaSurfaceObject_t s, const void* texel, size_t bytes, int xBytes, int y);
template <class T> struct TexRet;
template <> struct TexRet<float> { static float get(const float4& c) { return c.x; } };
template <> struct TexRet<float4> { static float4 get(const float4& c) { return c; } };
template <> struct TexRet<uchar4>
{
    static uchar4 get(const float4& c) { return uchar4{(unsigned char)c.x, (unsigned char)c.y, (unsigned char)c.z, (unsigned char)c.w}; }
};
}
template <class T> static inline T tex2D(cudaTextureObject_t t, float x, float y) { return shim::TexRet<T>::get(shim::tex_fetch(t, x, y, 0.0f)); }
template <class T> static inline T tex2DLod(cudaTextureObject_t t, float x, float y, float lod) { return shim::TexRet<T>::get(shim::tex_fetch(t, x, y, lod)); }
template <class T> static inline void surf2Dwrite(T v, cudaSurfaceObject_t s, int xBytes, int y) { shim::surf_write(s, &v, sizeof(T), xBytes, y); }

/* ---- kernel launch: the recipe rewrites  k<<<grid, block[, shmem, stream]>>>(args...)  into  shim::launch(grid, block, k, args...) ---- */
namespace shim {
template <class... P, class... A>
static inline void launch(dim3 grid, dim3 block, void (*kernel)(P...), A&&... args)
{
    const long long nBlocks = (long long)grid.x * grid.y * grid.z;
#pragma omp parallel for schedule(dynamic, 4)
    for(long long b = 0; b < nBlocks; ++b)
    {
        t_gridDim = grid;
        t_blockDim = block;
        t_blockIdx = dim3((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long long)grid.x * grid.y)));
        for(unsigned tz = 0; tz < block.z; ++tz)
            for(unsigned ty = 0; ty < block.y; ++ty)
                for(unsigned tx = 0; tx < block.x; ++tx)
                {
                    t_threadIdx = dim3(tx, ty, tz);
                    kernel(static_cast<P>(args)...);
                }
    }
}
}
#endif
