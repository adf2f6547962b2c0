// hbm_probe — what HBM bandwidth do the access shapes of the SGM aggregation reach on this box?
//   build: hipcc --offload-arch=gfx950 -O3 scripts/probes/hbm_probe.hip -o scripts/probes/hbm_probe
// Prints GB/s (bytes read + written) for: float4 copy, dword copy, dword 2-read-1-write, and the column walk of the SGM kernel
// (one wave per column, 256-B reads of two volumes + one 256-B write per step, slice stride = X * 256 B).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                                                                                                 \
    do                                                                                                                                                        \
    {                                                                                                                                                         \
        hipError_t e = (x);                                                                                                                                   \
        if(e != hipSuccess)                                                                                                                                   \
        {                                                                                                                                                     \
            printf("%s: %s\n", #x, hipGetErrorString(e));                                                                                                     \
            return 1;                                                                                                                                         \
        }                                                                                                                                                     \
    } while(0)

__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n)
{
    for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = a[i];
}
__global__ void copy1(const unsigned* __restrict__ a, unsigned* __restrict__ b, size_t n)
{
    for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = a[i];
}
__global__ void rrw1(const unsigned* __restrict__ a, const unsigned* __restrict__ c, unsigned* __restrict__ b, size_t n)
{
    for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        b[i] = a[i] + c[i];
}
// column walk: wave (blockIdx.x * 4 + wave) owns column col; step s touches byte offset (s * X + col) * 256 + lane * 4
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) colwalk(const unsigned* __restrict__ in, const unsigned* __restrict__ old, unsigned* __restrict__ out, int X, int Y)
{
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if(col >= X)
        return;
    size_t off = (size_t)col * 64 + lane;
    const size_t stride = (size_t)X * 64;
    unsigned acc = 0;
    for(int s = 0; s < Y; s += UNROLL)
    {
        unsigned a[UNROLL], o[UNROLL];
#pragma unroll
        for(int u = 0; u < UNROLL; ++u)
        {
            a[u] = in[off + (size_t)u * stride];
            o[u] = NT ? __builtin_nontemporal_load(old + off + (size_t)u * stride) : old[off + (size_t)u * stride];
        }
#pragma unroll
        for(int u = 0; u < UNROLL; ++u)
        {
            acc = acc * 3 + a[u] + o[u];
            if(NT)
                __builtin_nontemporal_store(acc, out + off + (size_t)u * stride);
            else
                out[off + (size_t)u * stride] = acc;
        }
        off += (size_t)UNROLL * stride;
    }
}

int main()
{
    const int X = 1000, Y = 752, NV = 4; // NV volumes side by side -> 4000 columns
    const size_t bytes = (size_t)X * NV * Y * 256;
    unsigned *a, *b, *c;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&c, bytes));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    CK(hipMemset(c, 3, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, double moved, auto launch) {
        for(int i = 0; i < 2; ++i)
            launch();
        hipEventRecord(e0);
        for(int i = 0; i < 5; ++i)
            launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.3f ms  %7.0f GB/s\n", name, ms / 5, moved / (ms / 5) / 1e6);
    };
    const size_t n4 = bytes / 16, n1 = bytes / 4;
    timeit("float4 copy (1R 1W)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy4, dim3(256 * 16), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); });
    timeit("dword copy (1R 1W)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy1, dim3(256 * 16), dim3(256), 0, 0, a, b, n1); });
    timeit("dword 2R 1W", 3.0 * bytes, [&] { hipLaunchKernelGGL(rrw1, dim3(256 * 16), dim3(256), 0, 0, a, c, b, n1); });
    for(int nv = 1; nv <= NV; nv *= 2)
    {
        char name[96];
        const int cols = X * nv;
        const double moved = 3.0 * (double)cols * Y * 256;
        snprintf(name, sizeof(name), "column walk 2R 1W, %d columns, unroll 8", cols);
        timeit(name, moved, [&] { hipLaunchKernelGGL((colwalk<8, false>), dim3((cols + 3) / 4), dim3(256), 0, 0, a, c, b, cols, Y); });
        snprintf(name, sizeof(name), "column walk 2R 1W, %d columns, unroll 16", cols);
        timeit(name, moved, [&] { hipLaunchKernelGGL((colwalk<16, false>), dim3((cols + 3) / 4), dim3(256), 0, 0, a, c, b, cols, Y); });
        snprintf(name, sizeof(name), "column walk 2R 1W nt, %d columns, unroll 16", cols);
        timeit(name, moved, [&] { hipLaunchKernelGGL((colwalk<16, true>), dim3((cols + 3) / 4), dim3(256), 0, 0, a, c, b, cols, Y); });
    }
    return 0;
}
