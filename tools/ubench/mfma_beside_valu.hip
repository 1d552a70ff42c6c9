// Does exact-fp32 MFMA work issue beside packed-fp32 VALU work on gfx950?  ("MFMA for the channel contractions")
//   hipcc --offload-arch=gfx950 -O3 mfma_beside_valu.hip -o mfma_beside_valu && ./mfma_beside_valu
// Per loop iteration and wave: V x v_pk_fma_f32 (independent accumulators) and M x v_mfma_f32_16x16x4_f32 (independent
// accumulators), 4 waves per SIMD.  Prints nominal cycles per iteration per SIMD for (V, 0), (0, M) and (V, M): if the two pipes
// overlap, the mixed loop costs about max() of the pure ones, if they share the issue port, about their sum.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int V, int M>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    f2 acc[16];
    f4 macc[4];
    for (int i = 0; i < 16; ++i) acc[i] = f2{seed + i, seed - i};
    for (int i = 0; i < 4; ++i) macc[i] = f4{seed, seed, seed, seed};
    const f2 a = {0.999f, 0.998f}, b = {1e-3f, 2e-3f};
    const float ma = seed * 1e-3f + threadIdx.x * 1e-6f, mb = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < V; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i & 15]) : "v"(a), "v"(b));
#pragma unroll
        for (int i = 0; i < M; ++i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(macc[i & 3]) : "v"(ma), "v"(mb));
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    for (int i = 0; i < 4; ++i) s += macc[i].x + macc[i].y + macc[i].z + macc[i].w;
    if (s == 12345.678f) out[0] = s;
}

template <int V, int M>
double run(int waves_per_simd)
{
    float *d; hipMalloc(&d, 4);
    const int iters = 4096, blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<V, M><<<blocks, 256>>>(d, 16, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<V, M><<<blocks, 256>>>(d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return ms * 1e-3 * 2.4e9 / ((double)waves_per_simd * iters);     // nominal 2.4 GHz cycles per wave-iteration per SIMD
}

int main()
{
    for (int w : {4}) {
        const double v = run<16, 0>(w), m = run<0, 2>(w), vm = run<16, 2>(w), v2 = run<16, 4>(w), m4 = run<0, 4>(w);
        printf("waves/SIMD %d   16 pk_fma: %.1f   2 mfma16x16x4f32: %.1f   both: %.1f (sum %.1f, max %.1f)   4 mfma: %.1f   16 pk_fma + 4 mfma: %.1f\n",
               w, v, m, vm, v + m, v > m ? v : m, m4, v2);
    }
    return 0;
}
