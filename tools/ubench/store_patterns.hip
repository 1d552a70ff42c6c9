// store_patterns.hip — how many bytes reach the memory side (rocprofv3 WRITE_SIZE) for the store patterns the per-Gaussian
// kernels use: one lane per Gaussian writing 4-, 12-, 16- or 64-byte rows with scalar or 16-byte stores, against the same
// bytes written with consecutive lanes on consecutive addresses.  Every kernel writes N rows of a buffer far larger than the
// 256 MiB Infinity Cache; expected bytes = N x row size.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/store_patterns.hip -o tools/ubench/store_patterns
//   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d out -o p -- tools/ubench/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int N = 16 * 1024 * 1024;   // rows

__global__ __launch_bounds__(256) void rows12_scalar(float *o) { const size_t g = blockIdx.x * 256ull + threadIdx.x; o[3 * g] = 1.f; o[3 * g + 1] = 2.f; o[3 * g + 2] = 3.f; }
__global__ __launch_bounds__(256) void rows16_scalar(float *o) { const size_t g = blockIdx.x * 256ull + threadIdx.x; o[4 * g] = 1.f; o[4 * g + 1] = 2.f; o[4 * g + 2] = 3.f; o[4 * g + 3] = 4.f; }
__global__ __launch_bounds__(256) void rows16_float4(float4 *o) { const size_t g = blockIdx.x * 256ull + threadIdx.x; o[g] = make_float4(1.f, 2.f, 3.f, 4.f); }
__global__ __launch_bounds__(256) void rows4_scalar(float *o) { const size_t g = blockIdx.x * 256ull + threadIdx.x; o[g] = 1.f; }
__global__ __launch_bounds__(256) void rows64_4xfloat4(float4 *o)
{
    const size_t g = blockIdx.x * 256ull + threadIdx.x;
    o[4 * g] = make_float4(1.f, 2.f, 3.f, 4.f); o[4 * g + 1] = make_float4(5.f, 6.f, 7.f, 8.f);
    o[4 * g + 2] = make_float4(1.f, 2.f, 3.f, 4.f); o[4 * g + 3] = make_float4(5.f, 6.f, 7.f, 8.f);
}
// the same 64-byte rows, transposed through LDS so that consecutive lanes store consecutive 16-byte pieces
__global__ __launch_bounds__(256) void rows64_staged(float4 *o)
{
    __shared__ float4 s[256 * 4 + 4];
    const size_t g0 = blockIdx.x * 256ull;
    for (int k = 0; k < 4; ++k) s[threadIdx.x * 4 + k] = make_float4(1.f + k, 2.f, 3.f, (float)threadIdx.x);
    __syncthreads();
    for (int k = 0; k < 4; ++k) o[4 * g0 + k * 256 + threadIdx.x] = s[k * 256 + threadIdx.x];
}
// 12-byte rows transposed through LDS: the block's 768 floats leave as 192 float4 stores of consecutive lanes
__global__ __launch_bounds__(256) void rows12_staged(float *o)
{
    __shared__ float s[256 * 3];
    const size_t g0 = blockIdx.x * 256ull;
    s[threadIdx.x * 3] = 1.f; s[threadIdx.x * 3 + 1] = 2.f; s[threadIdx.x * 3 + 2] = (float)threadIdx.x;
    __syncthreads();
    float4 *o4 = reinterpret_cast<float4 *>(o + 3 * g0);
    if (threadIdx.x < 192) o4[threadIdx.x] = make_float4(s[4 * threadIdx.x], s[4 * threadIdx.x + 1], s[4 * threadIdx.x + 2], s[4 * threadIdx.x + 3]);
}

int main()
{
    float *buf;
    if (hipMalloc(&buf, (size_t)N * 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const dim3 grid(N / 256), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(rows4_scalar, grid, block, 0, 0, buf);
        hipLaunchKernelGGL(rows12_scalar, grid, block, 0, 0, buf);
        hipLaunchKernelGGL(rows12_staged, grid, block, 0, 0, buf);
        hipLaunchKernelGGL(rows16_scalar, grid, block, 0, 0, buf);
        hipLaunchKernelGGL(rows16_float4, grid, block, 0, 0, (float4 *)buf);
        hipLaunchKernelGGL(rows64_4xfloat4, grid, block, 0, 0, (float4 *)buf);
        hipLaunchKernelGGL(rows64_staged, grid, block, 0, 0, (float4 *)buf);
    }
    hipDeviceSynchronize();
    printf("rows = %d: expected KiB  rows4 %d  rows12 %d  rows16 %d  rows64 %d\n", N, N * 4 / 1024, N * 12 / 1024, N * 16 / 1024, N / 1024 * 64);
    return 0;
}
