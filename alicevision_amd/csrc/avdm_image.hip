// avdm_image.hip — library plumbing + image side of the hot path:
//   float RGBA -> fp16x255 -> CIELAB -> Gaussian mip pyramid in plain HBM (no texture hardware).
// Replaces cuda/host/DeviceCache.cpp:222-281, cuda/host/DeviceMipmapImage.cpp:28-90 and
// cuda/imageProcessing/{deviceColorConversion,deviceGaussianFilter,deviceMipmappedArray}.cu of the reference.
#include "avdm_device.h"
#include "avdm_libm.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <map>
#include <mutex>
#include <utility>

namespace avdm {

static thread_local char g_err[512] = "";

int set_error(hipError_t e, const char* where)
{
    if(e == hipSuccess)
        return 0;
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
    return (int)e;
}
int set_error_msg(int code, const char* msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

namespace {
struct ScratchBlock
{
    void* ptr = nullptr;
    size_t bytes = 0;
    unsigned long long lastUse = 0;
    int inUse = 0; // entry points currently enqueueing work that reads / writes the block (StreamScratch objects alive)
};
std::mutex g_scratchMutex;
std::map<std::pair<int, hipStream_t>, ScratchBlock> g_scratch;
unsigned long long g_scratchTick = 0;
// Blocks belong to a (device, stream) and are given back by avdm_stream_release when the stream goes away (host/device.cpp does that for
// every stream it creates).  The cap is a safety net for callers that never release: PER DEVICE (a host thread per device opens up to 24
// tile streams + the pre-pass stream; a process-wide cap made three devices evict each other's live blocks), and a block that an entry
// point is using right now is never the victim.
constexpr size_t kMaxScratchBlocksPerDevice = 64;

void free_block_locked(std::map<std::pair<int, hipStream_t>, ScratchBlock>::iterator it, int currentDevice)
{
    if(it->first.first != currentDevice)
        (void)hipSetDevice(it->first.first);
    (void)hipFree(it->second.ptr); // waits for the device: no kernel of any stream, live or destroyed, still touches the block
    if(it->first.first != currentDevice)
        (void)hipSetDevice(currentDevice);
    g_scratch.erase(it);
}
} // namespace

StreamScratch::StreamScratch(hipStream_t st, size_t bytes) : _st(st)
{
    if(hipGetDevice(&_dev) != hipSuccess)
        return;
    bytes = (bytes + 255) & ~(size_t)255;
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    const auto key = std::make_pair(_dev, st);
    auto it = g_scratch.find(key);
    ++g_scratchTick;
    if(it != g_scratch.end() && it->second.bytes >= bytes)
    {
        it->second.lastUse = g_scratchTick;
        it->second.inUse += 1;
        _ptr = it->second.ptr;
        return;
    }
    if(it == g_scratch.end())
    {
        size_t onDevice = 0;
        auto oldest = g_scratch.end();
        for(auto i = g_scratch.begin(); i != g_scratch.end(); ++i)
            if(i->first.first == _dev)
            {
                ++onDevice;
                if(i->second.inUse == 0 && (oldest == g_scratch.end() || i->second.lastUse < oldest->second.lastUse))
                    oldest = i;
            }
        if(onDevice >= kMaxScratchBlocksPerDevice && oldest != g_scratch.end())
            free_block_locked(oldest, _dev); // idle longest, nobody enqueueing on it: its stream is most likely gone
    }
    ScratchBlock& b = g_scratch[key];
    if(b.ptr != nullptr)
    {
        // a larger block for the same stream.  Two host threads that share a stream (stream 0) may both be here: the one that still
        // enqueues on the old block keeps it alive through inUse — wait for that, not just for the stream
        if(b.inUse > 0)
        { // cannot grow under a concurrent user: hand out a private allocation instead (freed by the destructor)
            void* priv = nullptr;
            if(hipMalloc(&priv, bytes) != hipSuccess)
            {
                (void)hipGetLastError();
                return;
            }
            _ptr = priv;
            _private = true;
            return;
        }
        (void)hipStreamSynchronize(st); // earlier calls on this stream may still use the smaller block
        (void)hipFree(b.ptr);
        b = ScratchBlock{};
    }
    if(hipMalloc(&b.ptr, bytes) != hipSuccess)
    {
        (void)hipGetLastError();
        g_scratch.erase(key);
        return;
    }
    b.bytes = bytes;
    b.lastUse = g_scratchTick;
    b.inUse = 1;
    _ptr = b.ptr;
}

StreamScratch::~StreamScratch()
{
    if(_ptr == nullptr)
        return;
    if(_private)
    {
        (void)hipStreamSynchronize(_st);
        (void)hipFree(_ptr);
        return;
    }
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    auto it = g_scratch.find(std::make_pair(_dev, _st));
    if(it != g_scratch.end() && it->second.ptr == _ptr && it->second.inUse > 0)
        it->second.inUse -= 1;
}

int stream_scratch_release(hipStream_t st)
{
    int dev = 0;
    if(hipGetDevice(&dev) != hipSuccess)
        return 1;
    std::lock_guard<std::mutex> lock(g_scratchMutex);
    auto it = g_scratch.find(std::make_pair(dev, st));
    if(it == g_scratch.end())
        return 0;
    if(it->second.inUse > 0)
        return 2; // an entry point is still enqueueing on this stream: the caller's bug, nothing is freed
    (void)hipStreamSynchronize(st);
    free_block_locked(it, dev);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// one thread per texel, 256 threads along x: 16 B in / 8 B out per lane, fully coalesced
__global__ void __launch_bounds__(256) rgba_f32_to_f16x255_kernel(uint2* out, int out_pitch, const float4* in, int in_pitch, int width, int height)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if(x >= width)
        return;
    const float4 c = *((const float4*)((const char*)in + (long long)y * in_pitch) + x);
    *((uint2*)((char*)out + (long long)y * out_pitch) + x) = pack_h4(make_float4(c.x * 255.0f, c.y * 255.0f, c.z * 255.0f, c.w * 255.0f));
}

// color.cuh:65-70,124-141 of the reference restated.  The cube root is the C library's of the pinned reference build, to its bits
// (avdm_libm.h: glibc's s_cbrtf.c restated; the device library's cbrtf is within 1 ulp of it, and that ulp moved 1-5 % of the fp16 texels of
// every level by a quantum — rounds 1-5).  With it and without FMA contraction (this file is compiled -ffp-contract=off since round 6) the
// Lab pyramid equals the one the reference's own code builds on the CPU texel for texel: tests/test_gpu_parity.py::test_pyramid_parity.
__device__ __forceinline__ float lab_f(float r) { return r > 216.0f / 24389.0f ? glibc::cbrtf_pos(r) : (24389.0f / 27.0f * r + 16.0f) / 116.0f; }

__global__ void __launch_bounds__(256) rgb2lab_kernel(uint2* img, int pitch, int width, int height)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    const int y = blockIdx.y;
    if(x >= width)
        return;
    uint2* t = (uint2*)((char*)img + (long long)y * pitch) + x;
    float4 c = unpack_h4(*t);
    constexpr float d = 1 / 255.f;
    const float r = c.x * d, g = c.y * d, b = c.z * d;
    const float X = 0.4124564f * r + 0.3575761f * g + 0.1804375f * b;
    const float Y = 0.2126729f * r + 0.7151522f * g + 0.0721750f * b;
    const float Z = 0.0193339f * r + 0.1191920f * g + 0.9503041f * b;
    const float fx = lab_f(X / 0.95047f), fy = lab_f(Y), fz = lab_f(Z / 1.08883f);
    c.x = (116.0f * fy - 16.0f) * 2.55f;
    c.y = (500.0f * (fx - fy)) * 2.55f;
    c.z = (200.0f * (fy - fz)) * 2.55f;
    *t = pack_h4(c);
}

struct GaussTaps
{
    float g[21]; // radius <= 10
};

template <bool FIXED8>
__global__ void __launch_bounds__(256)
  downscale_gauss_kernel(uint2* out, int out_pitch, int out_w, int out_h, TexLevel in, int downscale, int radius, GaussTaps taps)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(x >= out_w || y >= out_h)
        return;
    const float s = (float)downscale * 0.5f;
    float4 acc = make_float4(0, 0, 0, 0);
    float sumFactor = 0.0f;
    for(int i = -radius; i <= radius; i++)
        for(int j = -radius; j <= radius; j++)
        {
            // unnormalised linear texture: texel-space coordinate = coord - 0.5.
            // `float(x * downscale + j)` is evaluated in UNSIGNED arithmetic in the reference (deviceGaussianFilter.cu:54-55,65): the taps
            // left of / above the image wrap to ~4.29e9 and the clamp addressing sends them to the RIGHT / BOTTOM edge (found by
            // oracle/_ref).  min(., 1e9): any coordinate beyond the image reads the edge texel with weight 1, and the int conversion stays exact
            const float cx = fminf((float)((unsigned)x * (unsigned)downscale + (unsigned)j), 1.0e9f);
            const float cy = fminf((float)((unsigned)y * (unsigned)downscale + (unsigned)i), 1.0e9f);
            const float4 c = tex_bilinear_px<FIXED8>(in, (cx + s) - 0.5f, (cy + s) - 0.5f);
            const float factor = taps.g[i + radius] * taps.g[j + radius];
            acc.x = acc.x + c.x * factor;
            acc.y = acc.y + c.y * factor;
            acc.z = acc.z + c.z * factor;
            acc.w = acc.w + c.w * factor;
            sumFactor += factor;
        }
    *((uint2*)((char*)out + (long long)y * out_pitch) + x) =
      pack_h4(make_float4(acc.x / sumFactor, acc.y / sumFactor, acc.z / sumFactor, acc.w / sumFactor));
}

// createMipmappedArrayLevel_kernel<2>: every tap samples the previous level half-way between 4 texels
// (u = (x + j + .5)/w  ->  2(x+j) + .5 in previous-level texel space when the previous width is even).
// A 64x4 output tile per block; the overlapping 5x5 footprints of neighbouring lanes are served by L1/L2
// (one-off per image, < 1 % of a depth map's time).
template <bool FIXED8>
__global__ void __launch_bounds__(256) mip_level_kernel(uint2* out, int out_pitch, int width, int height, TexLevel prev, GaussTaps taps)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if(x >= width || y >= height)
        return;
    const float px = 1.f / (float)width;
    const float py = 1.f / (float)height;
    float4 sum = make_float4(0, 0, 0, 0);
    float sumFactor = 0.0f;
#pragma unroll
    for(int i = -2; i <= 2; i++)
    {
#pragma unroll
        for(int j = -2; j <= 2; j++)
        {
            const float factor = taps.g[i + 2] * taps.g[j + 2];
            // `(x + j + 0.5f)` with x unsigned in the reference (deviceMipmappedArray.cu:28-29,52-53): the taps left of / above the image wrap
            // to ~4.29e9 and clamp to the RIGHT / BOTTOM edge texel — the first two rows / columns of every level carry that quirk
            // (found by oracle/_ref).  min(., 4e8): the same edge texel, u * W stays inside the int range — and far enough out (>= 2^24 texels)
            // that the filter coordinate is integer valued like the reference's 8.6e9, i.e. the blend fraction is exactly 0: rounds 1-5 clamped
            // to 1e4 image widths, where the fraction was 0.5 — the same two edge texels, but (1 - a) t + a t rounds differently from t, which
            // moved a quantum on ~0.1 % of the texels of every level's first two rows / columns (session r06_a)
            const float u = fminf((float)((unsigned)(x + j)), 4.0e8f) + 0.5f;
            const float v = fminf((float)((unsigned)(y + i)), 4.0e8f) + 0.5f;
            const float4 c = tex2D_level<FIXED8>(prev, u * px, v * py);
            sum.x = sum.x + c.x * factor;
            sum.y = sum.y + c.y * factor;
            sum.z = sum.z + c.z * factor;
            sum.w = sum.w + c.w * factor;
            sumFactor += factor;
        }
    }
    *((uint2*)((char*)out + (long long)y * out_pitch) + x) =
      pack_h4(make_float4(sum.x / sumFactor, sum.y / sumFactor, sum.z / sumFactor, sum.w / sumFactor));
}

__global__ void __launch_bounds__(256) tex2dlod_probe_kernel(float4* out, Tex T, const float* uvl, int n)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if(i < n)
        out[i] = tex2DLod(T, uvl[3 * i], uvl[3 * i + 1], uvl[3 * i + 2]);
}

static GaussTaps make_taps(int scale)
{
    // deviceGaussianFilter.cu:240-252: radius = scale + 1, delta = 1
    GaussTaps t;
    memset(&t, 0, sizeof(t));
    const int radius = scale + 1;
    for(int idx = 0; idx < 2 * radius + 1 && idx < 21; ++idx)
    {
        const int x = idx - radius;
        t.g[idx] = expf(-(x * x) / (2 * 1.0f * 1.0f));
    }
    return t;
}

} // namespace avdm

using namespace avdm;

extern "C" {

const char* avdm_last_error(void) { return g_err; }
int avdm_version(void) { return 100; }

int avdm_device_count(void)
{
    int n = 0;
    if(hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

int avdm_stream_release(void* stream)
{
    const int rc = stream_scratch_release((hipStream_t)stream);
    return rc == 0 ? 0 : set_error_msg(rc, rc == 2 ? "avdm_stream_release: the stream's block is in use by an entry point" : "avdm_stream_release: no current device");
}

int avdm_device_info(int device, char* out, size_t out_len)
{
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if(e != hipSuccess)
        return set_error(e, "avdm_device_info");
    size_t freeB = 0, totalB = 0;
    int cur = 0;
    hipGetDevice(&cur);
    hipSetDevice(device);
    hipMemGetInfo(&freeB, &totalB);
    hipSetDevice(cur);
    snprintf(out, out_len,
             "Device information:\n\t- id: %d\n\t- name: %s\n\t- arch: %s\n\t- compute units: %d\n\t- wavefront: %d\n\t- clock: %d kHz\n"
             "\t- global memory: %.1f MB (free %.1f MB)\n\t- LDS per block: %zu B\n",
             device, prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.warpSize, prop.clockRate, totalB / 1048576.0, freeB / 1048576.0,
             (size_t)prop.sharedMemPerBlock);
    return 0;
}

int avdm_pyramid_layout(avdm_pyramid_t* p, int width, int height, int min_downscale, int max_downscale, int filter_mode)
{
    if(!p || width <= 0 || height <= 0 || min_downscale < 1 || max_downscale < min_downscale)
        return set_error_msg(1, "avdm_pyramid_layout: invalid arguments");
    memset(p, 0, sizeof(*p));
    p->filter_mode = filter_mode;
    p->min_downscale = min_downscale;
    p->width0 = width;
    p->height0 = height;
    int levels = (int)log2((double)(max_downscale / min_downscale)) + 1;
    if(levels > AVDM_MAX_LEVELS)
        levels = AVDM_MAX_LEVELS;
    int w = (width + min_downscale - 1) / min_downscale, h = (height + min_downscale - 1) / min_downscale;
    long long off = 0;
    int l = 0;
    for(; l < levels && w > 0 && h > 0; ++l)
    {
        p->width[l] = w;
        p->height[l] = h;
        p->pitch[l] = ((w * 8 + 127) / 128) * 128;
        p->offset[l] = off;
        off += (long long)p->pitch[l] * h;
        w /= 2;
        h /= 2;
    }
    p->levels = l;
    p->bytes = off;
    return 0;
}

int avdm_image_rgba_f32_to_f16x255(void* out_h4, int out_pitch, const float* in_rgba, int in_pitch, int width, int height, void* stream)
{
    if(width <= 0 || height <= 0)
        return 0;
    dim3 grid(divUp(width, 256), height);
    hipLaunchKernelGGL(rgba_f32_to_f16x255_kernel, grid, dim3(256), 0, (hipStream_t)stream, (uint2*)out_h4, out_pitch, (const float4*)in_rgba, in_pitch,
                       width, height);
    AVDM_LAUNCH_CHECK("avdm_image_rgba_f32_to_f16x255");
}

int avdm_rgb2lab(void* inout_h4, int pitch, int width, int height, void* stream)
{
    if(width <= 0 || height <= 0)
        return 0;
    dim3 grid(divUp(width, 256), height);
    hipLaunchKernelGGL(rgb2lab_kernel, grid, dim3(256), 0, (hipStream_t)stream, (uint2*)inout_h4, pitch, width, height);
    AVDM_LAUNCH_CHECK("avdm_rgb2lab");
}

int avdm_downscale_with_gaussian_blur(void* out_h4, int out_pitch, int out_w, int out_h, const void* in_h4, int in_pitch, int in_w, int in_h,
                                      int downscale, int gauss_radius, int filter_mode, void* stream)
{
    if(gauss_radius > 10 || gauss_radius != downscale)
        return set_error_msg(1, "avdm_downscale_with_gaussian_blur: radius must equal downscale and be <= 10");
    TexLevel in{(const uint2*)in_h4, in_w, in_h, in_pitch / 8};
    const GaussTaps taps = make_taps(downscale - 1);
    dim3 grid(divUp(out_w, 64), divUp(out_h, 4));
    if(filter_mode == AVDM_FILTER_CUDA_FIXED8)
        hipLaunchKernelGGL(downscale_gauss_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (uint2*)out_h4, out_pitch, out_w, out_h, in, downscale,
                           gauss_radius, taps);
    else
        hipLaunchKernelGGL(downscale_gauss_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (uint2*)out_h4, out_pitch, out_w, out_h, in,
                           downscale, gauss_radius, taps);
    AVDM_LAUNCH_CHECK("avdm_downscale_with_gaussian_blur");
}

int avdm_pyramid_build_levels(const avdm_pyramid_t* p, void* stream)
{
    const Tex t = make_tex(p);
    const GaussTaps taps = make_taps(1);
    for(int l = 1; l < p->levels; ++l)
    {
        dim3 grid(divUp(p->width[l], 64), divUp(p->height[l], 4));
        uint2* out = (uint2*)((char*)p->base + p->offset[l]);
        if(p->filter_mode == AVDM_FILTER_CUDA_FIXED8)
            hipLaunchKernelGGL(mip_level_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, out, p->pitch[l], p->width[l], p->height[l], t.lv[l - 1],
                               taps);
        else
            hipLaunchKernelGGL(mip_level_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, out, p->pitch[l], p->width[l], p->height[l],
                               t.lv[l - 1], taps);
    }
    AVDM_LAUNCH_CHECK("avdm_pyramid_build_levels");
}

int avdm_tex2dlod(float* out4, const avdm_pyramid_t* pyr, const float* uvl, int n, void* stream)
{
    if(n <= 0)
        return 0;
    hipLaunchKernelGGL(tex2dlod_probe_kernel, dim3(divUp(n, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)out4, make_tex(pyr), uvl, n);
    AVDM_LAUNCH_CHECK("avdm_tex2dlod");
}

int avdm_pyramid_fill(const avdm_pyramid_t* p, const float* in_rgba, int in_pitch, void* scratch_h4, void* stream)
{
    int rc;
    if(p->min_downscale > 1)
    {
        if(!scratch_h4)
            return set_error_msg(1, "avdm_pyramid_fill: scratch required when min_downscale > 1");
        const int pitch = p->width0 * 8;
        if((rc = avdm_image_rgba_f32_to_f16x255(scratch_h4, pitch, in_rgba, in_pitch, p->width0, p->height0, stream)))
            return rc;
        if((rc = avdm_downscale_with_gaussian_blur(p->base, p->pitch[0], p->width[0], p->height[0], scratch_h4, pitch, p->width0, p->height0,
                                                   p->min_downscale, p->min_downscale, p->filter_mode, stream)))
            return rc;
    }
    else if((rc = avdm_image_rgba_f32_to_f16x255(p->base, p->pitch[0], in_rgba, in_pitch, p->width0, p->height0, stream)))
        return rc;
    if((rc = avdm_rgb2lab(p->base, p->pitch[0], p->width[0], p->height[0], stream)))
        return rc;
    return avdm_pyramid_build_levels(p, stream);
}

// fillHostCameraParameters (cuda/host/DeviceCache.cpp:41-134), Matrix3x3::inverse (mvsData/Matrix3x3.hpp:268-287)
static void inv3(const double* m, double* o)
{
    const double m11 = m[0], m12 = m[1], m13 = m[2], m21 = m[3], m22 = m[4], m23 = m[5], m31 = m[6], m32 = m[7], m33 = m[8];
    const double dt = m11 * (m33 * m22 - m32 * m23) - m21 * (m33 * m12 - m32 * m13) + m31 * (m23 * m12 - m22 * m13);
    o[0] = (m33 * m22 - m32 * m23) / dt;
    o[1] = -(m33 * m12 - m32 * m13) / dt;
    o[2] = (m23 * m12 - m22 * m13) / dt;
    o[3] = -(m33 * m21 - m31 * m23) / dt;
    o[4] = (m33 * m11 - m31 * m13) / dt;
    o[5] = -(m23 * m11 - m21 * m13) / dt;
    o[6] = (m32 * m21 - m31 * m22) / dt;
    o[7] = -(m32 * m11 - m31 * m12) / dt;
    o[8] = (m22 * m11 - m21 * m12) / dt;
}

void avdm_camera_fill(avdm_camera_t* out, const double Kin[9], const double R[9], const double C[3], int downscale)
{
    const double s = 1.0 / (float)downscale;
    double K[9], iK[9], iR[9], iP[9], P[12], t[3];
    for(int c = 0; c < 3; ++c)
    {
        K[c] = s * Kin[c];
        K[3 + c] = s * Kin[3 + c];
        K[6 + c] = Kin[6 + c];
    }
    inv3(K, iK);
    inv3(R, iR);
    for(int r = 0; r < 3; ++r)
        t[r] = 0.0 - (R[3 * r] * C[0] + R[3 * r + 1] * C[1] + R[3 * r + 2] * C[2]);
    for(int r = 0; r < 3; ++r)
    {
        for(int c = 0; c < 3; ++c)
            P[4 * r + c] = K[3 * r] * R[c] + K[3 * r + 1] * R[3 + c] + K[3 * r + 2] * R[6 + c];
        P[4 * r + 3] = K[3 * r] * t[0] + K[3 * r + 1] * t[1] + K[3 * r + 2] * t[2];
    }
    for(int r = 0; r < 3; ++r)
        for(int c = 0; c < 3; ++c)
            iP[3 * r + c] = iR[3 * r] * iK[c] + iR[3 * r + 1] * iK[3 + c] + iR[3 * r + 2] * iK[6 + c];
    for(int c = 0; c < 4; ++c)
        for(int r = 0; r < 3; ++r)
            out->P[3 * c + r] = (float)P[4 * r + c];
    for(int c = 0; c < 3; ++c)
        for(int r = 0; r < 3; ++r)
        {
            out->iP[3 * c + r] = (float)iP[3 * r + c];
            out->R[3 * c + r] = (float)R[3 * r + c];
            out->iR[3 * c + r] = (float)iR[3 * r + c];
            out->K[3 * c + r] = (float)K[3 * r + c];
            out->iK[3 * c + r] = (float)iK[3 * r + c];
        }
    out->C[0] = (float)C[0];
    out->C[1] = (float)C[1];
    out->C[2] = (float)C[2];
    float* dst[3] = {out->XVect, out->YVect, out->ZVect};
    for(int k = 0; k < 3; ++k)
    {
        // column k of iR (column-major storage) = iR * e_k
        const float vx = out->iR[3 * k + 0], vy = out->iR[3 * k + 1], vz = out->iR[3 * k + 2];
        const float d = sqrtf(vx * vx + vy * vy + vz * vz);
        dst[k][0] = vx / d;
        dst[k][1] = vy / d;
        dst[k][2] = vz / d;
    }
}

} // extern "C"
