// shim_runtime.cpp — the stand-in CUDA runtime behind oracle/ref/shim/cuda_runtime.h (TEST INFRASTRUCTURE ONLY, oracle/_ref).
// Device memory is host memory; arrays / mip-mapped arrays are plain buffers; the texture unit is restated from the CUDA C
// Programming Guide ("Texture Fetching"): clamp addressing, point filtering tex(x) = T[floor(x)], linear filtering
//     xB = x - 0.5, i = floor(xB), alpha = frac(xB):  tex = (1-a)(1-b) T[i,j] + a(1-b) T[i+1,j] + (1-a)b T[i,j+1] + ab T[i+1,j+1],
// alpha / beta (and the mip-level fraction) stored in 9-bit fixed point with 8 fractional bits when shim::g_fixed8 is set,
// normalized coordinates scaled by the level size, mip level = clamp(lod, min, max) with linear blending of two levels.
#include <cuda_fp16.h>
#include <cuda_runtime.h>

#include <vector>

namespace shim {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
bool g_fixed8 = true;
}

const char* cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : (e == cudaErrorMemoryAllocation ? "out of memory" : "invalid value"); }
cudaError_t cudaGetLastError() { return cudaSuccess; }
cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
cudaError_t cudaGetDevice(int* dev) { *dev = 0; return cudaSuccess; }
cudaError_t cudaSetDevice(int) { return cudaSuccess; }
cudaError_t cudaGetDeviceCount(int* n) { *n = 1; return cudaSuccess; }
cudaError_t cudaMemGetInfo(size_t* f, size_t* t) { *f = *t = (size_t)1 << 34; return cudaSuccess; }
cudaError_t shimMalloc(void** p, size_t bytes)
{
    *p = aligned_alloc(512, (bytes + 511) / 512 * 512 + 512);
    return *p ? cudaSuccess : cudaErrorMemoryAllocation;
}
cudaError_t cudaFree(void* p) { free(p); return cudaSuccess; }
cudaError_t cudaFreeHost(void* p) { free(p); return cudaSuccess; }
cudaError_t shimMallocPitch(void** p, size_t* pitch, size_t widthBytes, size_t height)
{
    *pitch = (widthBytes + 511) / 512 * 512; // cudaMallocPitch pads rows (the value only affects addresses)
    return shimMalloc(p, *pitch * (height ? height : 1));
}
cudaError_t cudaMalloc3D(cudaPitchedPtr* p, cudaExtent e)
{
    p->xsize = e.width;
    p->ysize = e.height;
    return shimMallocPitch(&p->ptr, &p->pitch, e.width, e.height * (e.depth ? e.depth : 1));
}
cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
cudaError_t cudaMemcpyAsync(void* d, const void* s, size_t n, cudaMemcpyKind, cudaStream_t) { memcpy(d, s, n); return cudaSuccess; }
cudaError_t cudaMemcpy2D(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, cudaMemcpyKind)
{
    for(size_t y = 0; y < h; ++y)
        memcpy((char*)d + y * dp, (const char*)s + y * sp, w);
    return cudaSuccess;
}
cudaError_t cudaMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, cudaMemcpyKind k, cudaStream_t) { return cudaMemcpy2D(d, dp, s, sp, w, h, k); }

// ---- arrays ----
struct cudaArray
{
    cudaChannelFormatDesc desc;
    size_t width = 0, height = 0, texelBytes = 0;
    std::vector<unsigned char> data;
};
struct cudaMipmappedArray
{
    std::vector<cudaArray> levels;
};
static size_t texel_bytes(const cudaChannelFormatDesc& d) { return (size_t)(d.x + d.y + d.z + d.w) / 8; }

cudaError_t cudaMallocMipmappedArray(cudaMipmappedArray_t* out, const cudaChannelFormatDesc* desc, cudaExtent extent, unsigned int numLevels, unsigned int)
{
    auto* m = new cudaMipmappedArray;
    size_t w = extent.width, h = extent.height;
    for(unsigned l = 0; l < numLevels; ++l)
    {
        // level sizes: floor halving, never below 1 (CUDA driver API, cuMipmappedArrayCreate)
        cudaArray a;
        a.desc = *desc;
        a.width = w ? w : 1;
        a.height = h ? h : 1;
        a.texelBytes = texel_bytes(*desc);
        a.data.assign(a.width * a.height * a.texelBytes, 0);
        m->levels.push_back(std::move(a));
        w /= 2;
        h /= 2;
    }
    *out = m;
    return cudaSuccess;
}
cudaError_t cudaFreeMipmappedArray(cudaMipmappedArray_t a) { delete a; return cudaSuccess; }
cudaError_t cudaGetMipmappedArrayLevel(cudaArray_t* level, cudaMipmappedArray_const_t a, unsigned int l)
{
    if(l >= a->levels.size())
        return cudaErrorInvalidValue;
    *level = const_cast<cudaArray*>(&a->levels[l]);
    return cudaSuccess;
}
cudaError_t cudaArrayGetInfo(cudaChannelFormatDesc* desc, cudaExtent* extent, unsigned int* flags, cudaArray_t a)
{
    if(desc)
        *desc = a->desc;
    if(extent)
        *extent = cudaExtent{a->width, a->height, 0};
    if(flags)
        *flags = 0;
    return cudaSuccess;
}
cudaError_t cudaMemcpy3D(const cudaMemcpy3DParms* p)
{
    // the one form the reference uses: pitched linear memory -> array (extent.width in ELEMENTS when an array takes part)
    if(!p->dstArray || p->srcArray || !p->srcPtr.ptr)
        return cudaErrorInvalidValue;
    cudaArray* a = p->dstArray;
    const size_t rowBytes = p->extent.width * a->texelBytes;
    if(p->extent.width > a->width || p->extent.height > a->height)
        return cudaErrorInvalidValue;
    for(size_t y = 0; y < p->extent.height; ++y)
        memcpy(a->data.data() + y * a->width * a->texelBytes, (const char*)p->srcPtr.ptr + y * p->srcPtr.pitch, rowBytes);
    return cudaSuccess;
}

// ---- texture / surface objects ----
namespace {
struct Level
{
    const unsigned char* base;
    size_t width, height, pitch;
};
struct TexObj
{
    cudaChannelFormatDesc desc;
    std::vector<Level> levels;
    cudaTextureDesc td;
};
struct SurfObj
{
    cudaArray* a;
};
inline float4 read_texel(const TexObj& t, const Level& L, long x, long y)
{
    // clamp addressing
    x = x < 0 ? 0 : (x > (long)L.width - 1 ? (long)L.width - 1 : x);
    y = y < 0 ? 0 : (y > (long)L.height - 1 ? (long)L.height - 1 : y);
    const int nch = (t.desc.x ? 1 : 0) + (t.desc.y ? 1 : 0) + (t.desc.z ? 1 : 0) + (t.desc.w ? 1 : 0);
    const unsigned char* p = L.base + (size_t)y * L.pitch + (size_t)x * (size_t)(t.desc.x / 8) * nch;
    float c[4] = {0.f, 0.f, 0.f, 0.f};
    for(int i = 0; i < nch; ++i)
    {
        if(t.desc.f == cudaChannelFormatKindFloat && t.desc.x == 32)
            memcpy(&c[i], p + 4 * i, 4);
        else if(t.desc.f == cudaChannelFormatKindFloat && t.desc.x == 16)
        {
            uint16_t h;
            memcpy(&h, p + 2 * i, 2);
            c[i] = shim::f16_to_f32(h);
        }
        else // 8-bit unsigned: element type, or [0, 1] when read as normalized float
            c[i] = (t.td.readMode == cudaReadModeNormalizedFloat) ? (float)p[i] / 255.0f : (float)p[i];
    }
    return make_float4(c[0], c[1], c[2], c[3]);
}
inline float quant8(float a) { return floorf(a * 256.0f + 0.5f) * (1.0f / 256.0f); }
float4 fetch_level(const TexObj& t, const Level& L, float x, float y)
{
    if(t.td.normalizedCoords)
    {
        x *= (float)L.width;
        y *= (float)L.height;
    }
    if(t.td.filterMode == cudaFilterModePoint)
        return read_texel(t, L, (long)floorf(x), (long)floorf(y));
    const float xb = x - 0.5f, yb = y - 0.5f;
    const float fx = floorf(xb), fy = floorf(yb);
    float a = xb - fx, b = yb - fy;
    if(shim::g_fixed8)
    {
        a = quant8(a);
        b = quant8(b);
    }
    const long i = (long)fx, j = (long)fy;
    const float4 t00 = read_texel(t, L, i, j), t10 = read_texel(t, L, i + 1, j), t01 = read_texel(t, L, i, j + 1), t11 = read_texel(t, L, i + 1, j + 1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    return make_float4(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x, w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
                       w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z, w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w);
}
}

cudaError_t cudaCreateTextureObject(cudaTextureObject_t* out, const cudaResourceDesc* res, const cudaTextureDesc* tex, const cudaResourceViewDesc*)
{
    auto* t = new TexObj;
    t->td = *tex;
    if(res->resType == cudaResourceTypePitch2D)
    {
        t->desc = res->res.pitch2D.desc;
        t->levels.push_back(Level{(const unsigned char*)res->res.pitch2D.devPtr, res->res.pitch2D.width, res->res.pitch2D.height, res->res.pitch2D.pitchInBytes});
    }
    else if(res->resType == cudaResourceTypeArray)
    {
        const cudaArray* a = res->res.array.array;
        t->desc = a->desc;
        t->levels.push_back(Level{a->data.data(), a->width, a->height, a->width * a->texelBytes});
    }
    else if(res->resType == cudaResourceTypeMipmappedArray)
    {
        const cudaMipmappedArray* m = res->res.mipmap.mipmap;
        t->desc = m->levels[0].desc;
        for(const cudaArray& a : m->levels)
            t->levels.push_back(Level{a.data.data(), a.width, a.height, a.width * a.texelBytes});
    }
    else
    {
        delete t;
        return cudaErrorInvalidValue;
    }
    *out = (cudaTextureObject_t)(uintptr_t)t;
    return cudaSuccess;
}
cudaError_t cudaDestroyTextureObject(cudaTextureObject_t t) { delete(TexObj*)(uintptr_t)t; return cudaSuccess; }
cudaError_t cudaCreateSurfaceObject(cudaSurfaceObject_t* out, const cudaResourceDesc* res)
{
    if(res->resType != cudaResourceTypeArray)
        return cudaErrorInvalidValue;
    *out = (cudaSurfaceObject_t)(uintptr_t) new SurfObj{res->res.array.array};
    return cudaSuccess;
}
cudaError_t cudaDestroySurfaceObject(cudaSurfaceObject_t s) { delete(SurfObj*)(uintptr_t)s; return cudaSuccess; }

namespace shim {
float4 tex_fetch(cudaTextureObject_t obj, float x, float y, float lod)
{
    const TexObj& t = *(const TexObj*)(uintptr_t)obj;
    if(t.levels.size() == 1)
        return fetch_level(t, t.levels[0], x, y);
    // mip-mapped: level = clamp(lod + bias, [minClamp, maxClamp]) and never past the last level
    float maxl = (float)(t.levels.size() - 1);
    if(t.td.maxMipmapLevelClamp < maxl)
        maxl = t.td.maxMipmapLevelClamp;
    float l = lod + t.td.mipmapLevelBias;
    if(!(l > t.td.minMipmapLevelClamp))
        l = t.td.minMipmapLevelClamp;
    if(l > maxl)
        l = maxl;
    if(t.td.mipmapFilterMode == cudaFilterModePoint)
        return fetch_level(t, t.levels[(size_t)floorf(l + 0.5f)], x, y);
    const float fl = floorf(l);
    float g = l - fl;
    if(g_fixed8)
        g = quant8(g);
    const size_t l0 = (size_t)fl;
    const float4 c0 = fetch_level(t, t.levels[l0], x, y);
    if(g == 0.0f || l0 + 1 >= t.levels.size())
        return c0;
    const float4 c1 = fetch_level(t, t.levels[l0 + 1], x, y);
    return make_float4((1.0f - g) * c0.x + g * c1.x, (1.0f - g) * c0.y + g * c1.y, (1.0f - g) * c0.z + g * c1.z, (1.0f - g) * c0.w + g * c1.w);
}
void surf_write(cudaSurfaceObject_t s, const void* texel, size_t bytes, int xBytes, int y)
{
    cudaArray* a = ((SurfObj*)(uintptr_t)s)->a;
    if(xBytes < 0 || y < 0 || (size_t)xBytes + bytes > a->width * a->texelBytes || (size_t)y >= a->height)
        return; // out-of-range surface writes are dropped (cudaBoundaryModeZero/Trap aside, the reference never issues one)
    memcpy(a->data.data() + (size_t)y * a->width * a->texelBytes + (size_t)xBytes, texel, bytes);
}
}
