// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for a few opcodes.
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed)
{
    float a[8];
    float2_ p[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x * 1e-3f; p[i] = float2_{a[i], a[i] * 0.5f}; }
    const float m = 0.999f, c = 1e-3f;
    const float2_ m2 = {0.999f, 0.998f}, c2 = {1e-3f, 2e-3f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(m2), "v"(c2));
            if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(m2));
            if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(m));
            if (OP == 7) asm volatile("s_nop 0\n v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (OP == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            if (OP == 10) asm volatile("v_cmp_le_f32 vcc, %0, %1" :: "v"(a[i]), "v"(c) : "vcc");
            if (OP == 11) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 12) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[10:11]" : "+v"(a[i]) : "v"(m) : "s10", "s11");
            if (OP == 13) asm volatile("v_cmp_le_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(c), "v"(m) : "vcc");
            if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(a[(i + 1) & 7]));
            if (OP == 15) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (OP == 16) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
            if (OP == 17) asm volatile("v_cmp_le_f32 s[10:11], %0, %1" :: "v"(a[i]), "v"(c) : "s10", "s11");
            if (OP == 18) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            if (OP == 20) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(m2), "v"(c2));
            if (OP == 21) asm volatile("v_pk_mul_f32 %0, %0, %1 op_sel_hi:[1,0]" : "+v"(p[i]) : "v"(m2));
            if (OP == 22) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(p[i]) : "v"(m2), "v"(c2));
            if (OP == 23) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
            if (OP == 24) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            if (OP == 25) asm volatile("v_pk_mul_f32 %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
            if (OP == 26) asm volatile("v_mul_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]));
            if (OP == 27) asm volatile("v_fma_f32 %0, %1, %2, %3" : "+v"(a[i]) : "v"(a[(i + 1) & 7]), "v"(a[(i + 2) & 7]), "v"(a[(i + 3) & 7]));
            if (OP == 28) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]), "v"(p[(i + 3) & 7]));
            if (OP == 29) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
            if (OP == 30) asm volatile("v_pk_add_f32 %0, %1, %2" : "+v"(p[i]) : "v"(p[(i + 1) & 7]), "v"(p[(i + 2) & 7]));
            if (OP == 19) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
void run(const char *name, int waves_per_simd)
{
    float *d; hipMalloc(&d, 4);
    const int iters = 4096;
    const int blocks = 256 * waves_per_simd;  // 256-thread blocks: 4 waves -> one per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 16, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD = waves_per_simd * iters * 8 ; cycles = ms * clock
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    double cycles = ms * 1e-3 * clk_khz * 1e3;
    double insts = (double)waves_per_simd * iters * 8;
    printf("%-14s waves/SIMD=%d  %.3f ms  cycles/inst/SIMD = %.2f (clock attr %d MHz)\n", name, waves_per_simd, ms, cycles / insts, clk_khz / 1000);
    hipFree(d);
}

int main()
{
    for (int w : {4, 8}) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<4>("v_mul_f32", w); run<5>("v_pk_mul_f32", w);
        run<8>("v_add_f32", w); run<9>("v_pk_add_f32", w);
        run<2>("v_exp_f32", w); run<3>("v_rcp_f32", w); run<6>("v_cndmask_b32", w); run<7>("nop+mov_dpp", w);
        run<10>("v_cmp_le_f32", w); run<11>("v_min_f32", w); run<12>("cndmask_e64_sgpr", w); run<13>("cmp+cndmask", w);
        run<14>("v_mov_b32", w); run<15>("v_and_b32", w); run<16>("v_cvt_ubyte0", w); run<17>("v_cmp->sgpr", w); run<18>("v_fmac_f32", w); run<19>("v_max3_f32", w);
        run<23>("pk_fma 3reg", w); run<24>("fmac 3reg", w); run<25>("pk_mul 3reg", w); run<26>("mul 3reg", w); run<27>("fma 4reg", w); run<28>("pk_fma 4reg", w); run<29>("pk_fma 3reg opsel", w); run<30>("pk_add 3reg", w);
        run<20>("pk_fma opselhi", w); run<21>("pk_mul opselhi", w); run<22>("pk_fma opsel", w);
    }
    return 0;
}
