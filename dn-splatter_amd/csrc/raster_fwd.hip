// raster_fwd.hip — per-tile front-to-back alpha compositing, forward (stage 3 of include/dnsplat.h).
//
// Replaces gsplat 1.0.0 rasterize_to_pixels_fwd (call site dn_splatter/dn_model.py:495) and the
// legacy rasterize_forward / nd_rasterize_forward behind gsplat.rasterize_gaussians
// (dn_model.py:564); rule set in SURVEY.md Appendix A.6.  D feature channels are composited in ONE
// pass (colour | expected depth | normal), which is what lets dn-splatter's second, "20 % slower"
// (README.md:60) normal pass disappear.
//
// CDNA4 mapping (not a 16x16-threads-per-tile CUDA layout):
//   * the unit of work is one wave64 = one 16x8 half tile, two pixels per lane (rows r and r+4 of
//     the half).  Two pixels per lane halve the LDS broadcast traffic per blended sample — with one
//     pixel per lane the 4 x ds_read_b128 per splat would cost as many LDS cycles as the VALU work.
//   * waves never synchronise with each other: each wave stages its own 64-splat batches
//     (coalesced 64-byte record gathers -> its private 4 KiB LDS slice) and leaves as soon as its
//     own 128 pixels are saturated (wave-level ballot), so there is no __syncthreads in the kernel.
//   * the gather for batch b+1 is issued before batch b is consumed; the loads stay in flight
//     behind the LDS/VALU loop (vmcnt is only waited on when the registers are written to LDS).
//   * workgroup = the two half tiles of one tile (shared L1 lines for the record gather);
//   * the per-pixel predicates (alive, inside the splat's support, saturating) live as explicit wave
//     masks in scalar registers and selections take the mask as the v_cndmask operand: written with
//     bools the loop issued as many scalar as vector instructions and was bound by both.

#include "splat_common.h"

#ifndef DNS_FWD_PAIR_OPSEL
#define DNS_FWD_PAIR_OPSEL 1
#endif
// 1: the splat loop takes TWO kept splats per trip — both records read and both exponents / alphas evaluated before the first is
// blended (the evaluation does not depend on the running transmittance), then blended in list order with the same per-pixel
// operation sequence as the one-splat loop (bit-identical images); an odd last splat pairs with itself under an empty mask.
#ifndef DNS_FWD_UNROLL2
#define DNS_FWD_UNROLL2 0
#endif
// 1: the lane's TWO PIXELS as packed fp32 pairs wherever gfx950 has a packed instruction (v_pk_mul / v_pk_add / v_pk_fma / v_pk_mov):
// (na, nb) dx, nc dy + b dx, the exponent's last FMA, opacity x exp, 1 - alpha, T (1 - alpha), alpha T, the odd seventh channel and
// the hand-over T <- T' are one instruction for both pixels instead of two (left to itself hipcc packs only along the channels), and
// the weights are broadcast out of their pair by operand selection: 32 instead of 39 vector instructions per splat (SQ_INSTS_VALU
// 274 M -> 229 M per C2 frame), same operations on the same operands, bit-identical images, 64 instead of 66 VGPRs.  MEASURED
// (profiles/r06s_fwd_sensitivity.txt): the kernel takes the same number of cycles (GRBM_GUI_ACTIVE 8.77 M vs 8.76 M), -1.0 % ... +1.3 %
// in paired replays — this loop is not short of vector issue slots.  N extra operations per splat (DNS_FWD_X_*) cost: a vector
// instruction 1.4 %, a scalar one 0.8 %, a taken branch 1.7 %, a v_exp 5 %, an LDS read 3 %; one wave per SIMD less 3 %, one MORE
// (64 VGPRs) +2-4 %: a chain of dependent vector -> scalar -> vector hand-overs per splat whose every link costs, at its balance
// point with seven waves.  Off (the long-tested form stays the default); kept as a switch with the experiment.
#ifndef DNS_FWD_PACKED
#define DNS_FWD_PACKED 0
#endif
#ifndef DNS_FWD_X_VALU
#define DNS_FWD_X_VALU 0
#endif
#ifndef DNS_FWD_X_EXP
#define DNS_FWD_X_EXP 0
#endif
#ifndef DNS_FWD_X_SALU
#define DNS_FWD_X_SALU 0
#endif
#ifndef DNS_FWD_X_BR
#define DNS_FWD_X_BR 0
#endif
#ifndef DNS_FWD_X_LDS
#define DNS_FWD_X_LDS 0
#endif
#ifndef DNS_FWD_EXP_NOBRANCH
#define DNS_FWD_EXP_NOBRANCH 0
#endif

namespace {

constexpr int TILE = 16;
constexpr int FWD_WAVES = 2;  // waves per workgroup == half tiles per tile
constexpr int FWD_THREADS = FWD_WAVES * DNS_WAVE;

struct FwdArgs {
    int width, height, tw, n_tiles;            // n_tiles: per camera; the launch covers n_tiles x cameras stacked tile grids
    unsigned long long *counters;              // measurement instantiation only (dnsplat_raster_args.pair_counters)
    unsigned long long *keep_masks;            // or NULL: per (half tile, 64-entry batch) ballot of the rectangle test, for the backward
    long long keep_mask_stride;
    const float4 *__restrict__ splats;
    const int32_t *__restrict__ flatten_ids;
    const int32_t *__restrict__ tile_offsets;
    const int32_t *__restrict__ tile_ends;     // or NULL: the list of tile t ends at tile_offsets[t + 1]
    const float *__restrict__ background;
    int ed_channel;
    float *__restrict__ render;
    float *__restrict__ alphas;
    int32_t *__restrict__ last_ids;
    // fused dn-splatter epilogue (DN instantiation only)
    const float *__restrict__ bg_rgb;
    float *__restrict__ dn_rgb;
    float *__restrict__ dn_depth;
    float *__restrict__ dn_normal;
    float *__restrict__ dn_depth_max;
    float4 *zero_fill;                         // dnsplat_raster_args.zero_fill: cleared by this launch, zero_fill_vec4 16-byte pieces
    long long zero_fill_vec4;
};

// The dn-splatter per-pixel post-ops (dn_model.py:526-528, 577-578) applied to one finished pixel.  bg_rgb: the background colour,
// loaded once by the caller (read here per pixel and channel, each read was a load followed by its own wait).
__device__ __forceinline__ float dn_epilogue(const FwdArgs &a, size_t pid, const float *raw, float al, const float *bg_rgb)
{
    const float one_minus = 1.f - al;   // the reference computes (1 - alpha) from the rounded alpha image
#pragma unroll
    for (int c = 0; c < 3; ++c)
        a.dn_rgb[pid * 3 + c] = fminf(fmaxf(raw[c] + one_minus * bg_rgb[c], 0.f), 1.f);
    const float nx = raw[4], ny = raw[5], nz = raw[6];
    const float nrm = sqrtf(nx * nx + ny * ny + nz * nz);
    a.dn_normal[pid * 3 + 0] = (nx / nrm + 1.f) / 2.f;
    a.dn_normal[pid * 3 + 1] = (ny / nrm + 1.f) / 2.f;
    a.dn_normal[pid * 3 + 2] = (nz / nrm + 1.f) / 2.f;
    a.dn_depth[pid] = raw[3];
    return raw[3];
}

// Register budget: hipcc lands on 65 VGPRs, one too many for the eighth wave per SIMD.  Capped at 64 the spills stay outside
// the per-splat loop, yet the kernel gets slower (0.573 -> 0.627 ms): the SIMDs are 98 % busy with seven waves (PMC), an eighth
// buys nothing.  DNS_FWD_WAVES_PER_EU = 0 leaves the choice to the compiler.
#ifndef DNS_FWD_WAVES_PER_EU
#define DNS_FWD_WAVES_PER_EU 0
#endif
#if DNS_FWD_WAVES_PER_EU > 0
#define DNS_FWD_OCCUPANCY __attribute__((amdgpu_waves_per_eu(DNS_FWD_WAVES_PER_EU, DNS_FWD_WAVES_PER_EU)))
#else
#define DNS_FWD_OCCUPANCY
#endif

// lane-wise select on a wave mask held in scalar registers: mask bit set ? a : b  (sel0: b = 0)
__device__ __forceinline__ float sel(uint64_t mask, float a, float b)
{
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(mask));
    return r;
}
__device__ __forceinline__ float sel0(uint64_t mask, float a)
{
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(a), "s"(mask));
    return r;
}

// COUNT: measurement build of the fused pass — also tallies list entries examined, splats walked, live (pixel, splat) pairs
// evaluated and pairs blended into a.counters (bench.py's VALU roofline); never the instantiation that is timed.
// (A clamp-free twin as in raster_bwd.hip was measured here too: the two `min` it drops from 43 vector instructions per splat
// bought nothing — 0.562 vs 0.561 ms — and its differently scheduled code broke the bit-identity of the raw composite between
// the instantiations, which tests/test_reference_golden.py relies on.  Removed.)
template <int D, bool DN, bool COUNT = false>
__global__ __launch_bounds__(FWD_THREADS) DNS_FWD_OCCUPANCY void raster_fwd_kernel(FwdArgs a)
{
    // one 64-record slice per wave: [wave][splat][4 x float4]
    __shared__ float4 lds[FWD_WAVES][DNS_WAVE][4];
    // The backward's accumulation buffer (one 64-byte gradient record per splat, added to with atomics) has to start at zero.
    // This kernel is bound by vector issue and leaves the memory system idle: every workgroup clears its share on the way
    // (4 stores per lane at C2) instead of a 10 us fill launch in front of the backward.
    if (a.zero_fill) {
        const long long per = (a.zero_fill_vec4 + gridDim.x - 1) / gridDim.x;
        const long long lo = (long long)blockIdx.x * per, hi = min(lo + per, a.zero_fill_vec4);
        for (long long i = lo + threadIdx.x; i < hi; i += FWD_THREADS) a.zero_fill[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // batch of cameras: block b works on tile b % n_tiles of camera b / n_tiles; its lists are tile_offsets[b .. b + 1] and its
    // pixels live in image `cam` of the stacked [C,H,W,.] outputs
    const int cam = blockIdx.x / a.n_tiles;
    const int tile = dns_tile_of_block(blockIdx.x - cam * a.n_tiles, a.n_tiles, a.tw);
    const size_t img = (size_t)cam * a.width * a.height;
    const int list = cam * a.n_tiles + tile;
    const int wave = threadIdx.x / DNS_WAVE;
    [[maybe_unused]] unsigned long long n_entries = 0, n_walked = 0, n_live = 0, n_blend = 0;
    const int lane = threadIdx.x & (DNS_WAVE - 1);
    const int tile_x = tile % a.tw, tile_y = tile / a.tw;
    const int px_i = tile_x * TILE + (lane & 15);
    const int py_i0 = tile_y * TILE + wave * 8 + (lane >> 4);
    const int py_i1 = py_i0 + 4;
    const float px = (float)px_i + 0.5f;
    const float py0 = (float)py_i0 + 0.5f, py1 = (float)py_i1 + 0.5f;
    const bool in0 = px_i < a.width && py_i0 < a.height;
    const bool in1 = px_i < a.width && py_i1 < a.height;

    const int range_start = a.tile_offsets[list];
    const int range_end = a.tile_ends ? a.tile_ends[list] : a.tile_offsets[list + 1];

    float T0 = 1.f, T1 = 1.f;
    float acc0[D], acc1[D];
#pragma unroll
    for (int k = 0; k < D; ++k) { acc0[k] = 0.f; acc1[k] = 0.f; }
    typedef float f2p __attribute__((ext_vector_type(2)));
    [[maybe_unused]] f2p Tp = {1.f, 1.f};            // DNS_FWD_PACKED: (T0, T1) as a register pair, carried through the splat loop
    [[maybe_unused]] f2p acc_odd = {0.f, 0.f};       // DNS_FWD_PACKED, D odd: channel D - 1 of both pixels as a pair
    [[maybe_unused]] f2p accp0[4], accp1[4];         // DNS_FWD_PACKED: channels (2 k, 2 k + 1) of pixel 0 / pixel 1 as pairs
#pragma unroll
    for (int k = 0; k < 4; ++k) { accp0[k] = f2p{0.f, 0.f}; accp1[k] = f2p{0.f, 0.f}; }
    // last_ids: for a pixel that saturated the index before the saturating entry, else the end of the last batch it blended anything
    // of; in both cases no entry in (last blended, last_ids] applies to the pixel.  Raw bits (sel() moves floats).
    float last0 = 0.f, last1 = 0.f;
    uint64_t any0 = 0ull, any1 = 0ull;               // pixels that blended an entry of the current batch
    uint64_t done0 = ~dns_ballot(in0), done1 = ~dns_ballot(in1);   // wave masks: pixel saturated (or outside the image)

    float4(*my)[4] = lds[wave];
    // pixel-centre rectangle of this wave's 16x8 half tile, for the per-splat cull test
    const float rxl = (float)(tile_x * TILE) + 0.5f, rxh = rxl + 15.f;
    const float ryl = (float)(tile_y * TILE + wave * 8) + 0.5f, ryh = ryl + 7.f;

    // prefetch batch 0
    // D == 7 needs one float of the record's last quarter (channel 6): prefetched as one dword, three VGPRs less across the splat loop
    float4 r0, r1, r2;
    [[maybe_unused]] float4 r3;
    [[maybe_unused]] float r3x = 0.f;
    {
        const int idx = range_start + lane;
        if (idx < range_end) {
            const int g = a.flatten_ids[idx];
            const float4 *rec = a.splats + (size_t)g * 4;
            r0 = rec[0]; r1 = rec[1];
            if (D > 2) r2 = rec[2];
            if (D == 7) r3x = reinterpret_cast<const float *>(rec + 3)[0];
            else if (D > 6) r3 = rec[3];
        }
    }

    for (int batch_start = range_start; batch_start < range_end; batch_start += DNS_WAVE) {
        if ((done0 & done1) == ~0ull) break;
        // Splats that cannot reach alpha >= 1/255 anywhere in the half tile are dropped here, once per
        // (wave, splat), instead of being rejected 128 times by the per-pixel test.
        const bool keep = (batch_start + lane < range_end) &&
                          !dns_cull_rect(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, rxl, rxh, ryl, ryh);
        uint64_t todo = dns_ballot(keep);
        if (a.keep_masks && lane == 0)
            a.keep_masks[(size_t)wave * a.keep_mask_stride + (range_start >> 6) + list + ((batch_start - range_start) >> 6)] = todo;
        if (COUNT) n_entries += min(DNS_WAVE, range_end - batch_start);
        // stage the prefetched records (waits for the gather here) with the conic pre-scaled for exp2,
        // then start the next gather
        {
            const DnsConicE q = dns_conic_e(r0.z, r0.w, r1.x);
            my[lane][0] = make_float4(r0.x, r0.y, q.na, q.nb);
            my[lane][1] = make_float4(q.nc, r1.y, r1.z, r1.w);
            if (D > 2) my[lane][2] = r2;
            if (D == 7) reinterpret_cast<float *>(&my[lane][3])[0] = r3x;
            else if (D > 6) my[lane][3] = r3;
        }
        {
            const int idx = batch_start + DNS_WAVE + lane;
            if (idx < range_end) {
                const int g = a.flatten_ids[idx];
                const float4 *rec = a.splats + (size_t)g * 4;
                r0 = rec[0]; r1 = rec[1];
                if (D > 2) r2 = rec[2];
                if (D == 7) r3x = reinterpret_cast<const float *>(rec + 3)[0];
                else if (D > 6) r3 = rec[3];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // The per-pixel predicates are kept as explicit wave masks (uint64_t in scalar registers) and combined with
        // scalar and/or; selections take the mask as the v_cndmask operand (sel / sel0).  Written with bools, hipcc merges
        // every predicate that is live across a branch with three scalar instructions per merge and rebuilds masks through
        // v_cndmask + v_cmp: the loop then issued as many scalar as vector instructions and was bound by both.
        // A kept splat nearly always hits some pixel of the strip, so the blend is unconditional (weight 0 for a skipped pair).
#if DNS_FWD_UNROLL2
        // what of a splat does not depend on the pixels' running state: exponents, capped alphas, channels
        auto eval = [&](int t, float &e0, float &e1, float &alpha0, float &alpha1, float (&ch)[8]) __attribute__((always_inline)) {
            const float4 g0 = my[t][0];  // x y na nb
            const float4 g1 = my[t][1];  // nc opac ch0 ch1
            float4 g2, g3;
            if (D > 2) g2 = my[t][2];
            if (D > 6) g3 = my[t][3];
            DnsConicE q;
            q.na = g0.z; q.nb = g0.w; q.nc = g1.x;
            const float dx = g0.x - px;
            const float dy0 = g0.y - py0, dy1 = g0.y - py1;
#if DNS_EXP_SYM
            const float adx = q.na * dx, bdx = q.nb * dx;
#if DNS_FWD_PAIR_OPSEL
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 an = {adx, q.nb}, dyv = {dy0, dy1};
            f2 hu;
            asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(hu) : "v"(an), "v"(dyv));
            e0 = __builtin_fmaf(dx, hu.x, dy0 * __builtin_fmaf(q.nc, dy0, bdx));
            e1 = __builtin_fmaf(dx, hu.y, dy1 * __builtin_fmaf(q.nc, dy1, bdx));
#else
            e0 = __builtin_fmaf(dx, __builtin_fmaf(q.nb, dy0, adx), dy0 * __builtin_fmaf(q.nc, dy0, bdx));
            e1 = __builtin_fmaf(dx, __builtin_fmaf(q.nb, dy1, adx), dy1 * __builtin_fmaf(q.nc, dy1, bdx));
#endif
#else
            e0 = dns_exponent(q, dx, dy0);
            e1 = dns_exponent(q, dx, dy1);
#endif
            alpha0 = fminf((float)DNS_ALPHA_MAX, g1.y * dns_exp2(e0));
            alpha1 = fminf((float)DNS_ALPHA_MAX, g1.y * dns_exp2(e1));
            ch[0] = g1.z; ch[1] = g1.w;
            if (D > 2) { ch[2] = g2.x; ch[3] = g2.y; ch[4] = g2.z; ch[5] = g2.w; }
            if (D > 6) { ch[6] = g3.x; ch[7] = g3.y; }
        };
        // the order-dependent part, exactly the one-splat loop's: `has` = ~0 (a list entry) or 0 (the filler of an odd trip)
        auto blend = [&](int t, uint64_t has, float e0, float e1, float alpha0, float alpha1, const float (&ch)[8]) __attribute__((always_inline)) {
            float nT0, nT1, v0, v1;
            const uint64_t valid0 = dns_ballot(e0 <= 0.f) & dns_ballot(alpha0 >= (float)DNS_ALPHA_MIN) & ~done0 & has;
            const uint64_t valid1 = dns_ballot(e1 <= 0.f) & dns_ballot(alpha1 >= (float)DNS_ALPHA_MIN) & ~done1 & has;
            const float a0 = sel0(valid0, alpha0), a1 = sel0(valid1, alpha1);
            nT0 = T0 * (1.f - a0); nT1 = T1 * (1.f - a1);
            const uint64_t stop0 = dns_ballot(nT0 <= (float)DNS_T_MIN), stop1 = dns_ballot(nT1 <= (float)DNS_T_MIN);
            v0 = a0 * T0; v1 = a1 * T1;
            any0 |= valid0; any1 |= valid1;
            if (COUNT && has) {
                n_walked += 1;
                n_live += __popcll(~done0) + __popcll(~done1);
                n_blend += __popcll(valid0 & ~stop0) + __popcll(valid1 & ~stop1);
            }
            const int before = batch_start + t - 1;
            float tmp;
            asm volatile(
                "s_or_b64 vcc, %9, %10\n\t"
                "s_cmp_eq_u64 vcc, 0\n\t"
                "s_cbranch_scc1 1f\n\t"
                "v_mov_b32_e32 %8, %13\n\t"
                "v_cndmask_b32_e64 %0, %0, 0, %9\n\t"
                "v_cndmask_b32_e64 %1, %1, 0, %10\n\t"
                "v_cndmask_b32_e64 %2, %2, %11, %9\n\t"
                "v_cndmask_b32_e64 %3, %3, %12, %10\n\t"
                "v_cndmask_b32_e64 %4, %4, %8, %9\n\t"
                "v_cndmask_b32_e64 %5, %5, %8, %10\n\t"
                "s_or_b64 %6, %6, %9\n\t"
                "s_or_b64 %7, %7, %10\n\t"
                "1:"
                : "+v"(v0), "+v"(v1), "+v"(nT0), "+v"(nT1), "+v"(last0), "+v"(last1), "+s"(done0), "+s"(done1), "=&v"(tmp)
                : "s"(stop0), "s"(stop1), "v"(T0), "v"(T1), "s"(before)
                : "vcc", "scc");
#pragma unroll
            for (int k = 0; k < D; ++k) { acc0[k] += ch[k] * v0; acc1[k] += ch[k] * v1; }
            T0 = nT0; T1 = nT1;
        };
        while (todo) {
            const int tA = __ffsll((unsigned long long)todo) - 1;
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(tA));
            const uint64_t hasB = todo ? ~0ull : 0ull;
            const int tB = todo ? __ffsll((unsigned long long)todo) - 1 : tA;
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(tB));       // no second splat: clears bit tA again
            float eA0, eA1, aA0, aA1, chA[8], eB0, eB1, aB0, aB1, chB[8];
            eval(tA, eA0, eA1, aA0, aA1, chA);
            eval(tB, eB0, eB1, aB0, aB1, chB);
            blend(tA, ~0ull, eA0, eA1, aA0, aA1, chA);
            blend(tB, hasB, eB0, eB1, aB0, aB1, chB);
            {
                [[maybe_unused]] uint64_t tmp_s;
                asm("s_and_b64 %1, %2, %3\n\ts_cmp_eq_u64 %1, -1\n\ts_cselect_b64 %0, 0, %0"
                    : "+s"(todo), "=&s"(tmp_s) : "s"(done0), "s"(done1) : "scc");
            }
        }
#else
#if DNS_FWD_PACKED
        while (todo) {
            const int t = __ffsll((unsigned long long)todo) - 1;
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(t));      // todo &= todo - 1 as one scalar instruction instead of three
            const float4 g0 = my[t][0];  // x y na nb
            const float4 g1 = my[t][1];  // nc opac ch0 ch1
            float4 g2, g3;
            if (D > 2) g2 = my[t][2];
            if (D > 6) g3 = my[t][3];
            DnsConicE q;
            q.na = g0.z; q.nb = g0.w; q.nc = g1.x;
            const float dx = g0.x - px;
            const float dy0 = g0.y - py0, dy1 = g0.y - py1;
#if DNS_EXP_SYM && DNS_FWD_PAIR_OPSEL
            // dns_exponent() for both pixels of the lane: (na, nb) x dx as ONE packed multiply, then nb dy + adx and nc dy + bdx as two
            // packed FMAs whose broadcast operands (nb, adx, nc, bdx) are picked out of the pairs they sit in by operand selection
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 nab = {q.na, q.nb}, ncp = {g1.x, g1.y}, dyv = {dy0, dy1};
            const f2 abdx = nab * dx;
            f2 hu, hw;
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(hu) : "v"(nab), "v"(dyv), "v"(abdx));
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,1,1]" : "=v"(hw) : "v"(ncp), "v"(dyv), "v"(abdx));
            const f2 ev = __builtin_elementwise_fma(f2{dx, dx}, hu, dyv * hw);
            const float e0 = ev.x, e1 = ev.y;
#elif DNS_EXP_SYM
            // dns_exponent() for both pixels of the lane, with the two products that only depend on dx formed once
            const float adx = q.na * dx, bdx = q.nb * dx;
#if DNS_FWD_PAIR_OPSEL
            // nb * dy + adx for both pixels as ONE packed FMA on the register pair (adx, nb) the record's (na, nb) turns into:
            // nb is the pair's high half broadcast as the multiplier, adx its low half broadcast as the addend.  hipcc only finds
            // low-half broadcasts and copies nb into a fresh pair first.
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 an = {adx, q.nb}, dyv = {dy0, dy1};
            f2 hu;
            asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(hu) : "v"(an), "v"(dyv));
            const float e0 = __builtin_fmaf(dx, hu.x, dy0 * __builtin_fmaf(q.nc, dy0, bdx));
            const float e1 = __builtin_fmaf(dx, hu.y, dy1 * __builtin_fmaf(q.nc, dy1, bdx));
#else
            const float e0 = __builtin_fmaf(dx, __builtin_fmaf(q.nb, dy0, adx), dy0 * __builtin_fmaf(q.nc, dy0, bdx));
            const float e1 = __builtin_fmaf(dx, __builtin_fmaf(q.nb, dy1, adx), dy1 * __builtin_fmaf(q.nc, dy1, bdx));
#endif
#else
            const float e0 = dns_exponent(q, dx, dy0);
            const float e1 = dns_exponent(q, dx, dy1);
#endif
            // opacity x exp for both pixels in one packed multiply
            f2p al = f2p{dns_exp2(e0), dns_exp2(e1)} * g1.y;
            const float alpha0 = fminf((float)DNS_ALPHA_MAX, al.x), alpha1 = fminf((float)DNS_ALPHA_MAX, al.y);
            float ch[8];
            ch[0] = g1.z; ch[1] = g1.w;
            if (D > 2) { ch[2] = g2.x; ch[3] = g2.y; ch[4] = g2.z; ch[5] = g2.w; }
            if (D > 6) { ch[6] = g3.x; ch[7] = g3.y; }
            const uint64_t valid0 = dns_ballot(e0 <= 0.f) & dns_ballot(alpha0 >= (float)DNS_ALPHA_MIN) & ~done0;
            const uint64_t valid1 = dns_ballot(e1 <= 0.f) & dns_ballot(alpha1 >= (float)DNS_ALPHA_MIN) & ~done1;
            const f2p a = {sel0(valid0, alpha0), sel0(valid1, alpha1)};
            const f2p nT = Tp * (1.f - a);                           // v_pk_add (1 - alpha), v_pk_mul
            const uint64_t stop0 = dns_ballot(nT.x <= (float)DNS_T_MIN), stop1 = dns_ballot(nT.y <= (float)DNS_T_MIN);
            const f2p vw = a * Tp;
            any0 |= valid0; any1 |= valid1;
            if (COUNT) {
                n_walked += 1;
                n_live += __popcll(~done0) + __popcll(~done1);
                n_blend += __popcll(valid0 & ~stop0) + __popcll(valid1 & ~stop1);
            }
            float v0 = vw.x, v1 = vw.y, nT0 = nT.x, nT1 = nT.y;
            {
                const int before = batch_start + t - 1;
                const float T0c = Tp.x, T1c = Tp.y;
                float tmp;
                [[maybe_unused]] uint64_t tmp_s;
                asm volatile(
#if !DNS_FWD_EXP_NOBRANCH  // experiment: the six selects unconditionally, no branch around them (same images)
                    "s_or_b64 vcc, %9, %10\n\t"
                    "s_cmp_eq_u64 vcc, 0\n\t"
                    "s_cbranch_scc1 1f\n\t"
#endif
                    "v_mov_b32_e32 %8, %13\n\t"
                    "v_cndmask_b32_e64 %0, %0, 0, %9\n\t"
                    "v_cndmask_b32_e64 %1, %1, 0, %10\n\t"
                    "v_cndmask_b32_e64 %2, %2, %11, %9\n\t"
                    "v_cndmask_b32_e64 %3, %3, %12, %10\n\t"
                    "v_cndmask_b32_e64 %4, %4, %8, %9\n\t"
                    "v_cndmask_b32_e64 %5, %5, %8, %10\n\t"
                    "s_or_b64 %6, %6, %9\n\t"
                    "s_or_b64 %7, %7, %10\n\t"
                    "1:"
                    : "+v"(v0), "+v"(v1), "+v"(nT0), "+v"(nT1), "+v"(last0), "+v"(last1), "+s"(done0), "+s"(done1), "=&v"(tmp)
                    : "s"(stop0), "s"(stop1), "v"(T0c), "v"(T1c), "s"(before)
                    : "vcc", "scc");
                asm("s_and_b64 %1, %2, %3\n\ts_cmp_eq_u64 %1, -1\n\ts_cselect_b64 %0, 0, %0"
                    : "+s"(todo), "=&s"(tmp_s) : "s"(done0), "s"(done1) : "scc");
            }
            {
                // channel pairs x the pixel's weight, the weight broadcast from the LOW (pixel 0) or the HIGH (pixel 1) half of the
                // (v0, v1) pair by operand selection: hipcc only finds low-half broadcasts and copies v1 into a fresh pair first
                const f2p vv = {v0, v1};
                const f2p chp[4] = {f2p{ch[0], ch[1]}, f2p{ch[2], ch[3]}, f2p{ch[4], ch[5]}, f2p{ch[6], ch[7]}};
#pragma unroll
                for (int kp = 0; kp < D / 2; ++kp) {
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(accp0[kp]) : "v"(chp[kp]), "v"(vv));
                    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(accp1[kp]) : "v"(chp[kp]), "v"(vv));
                }
                if (D & 1) acc_odd += vv * ch[D - 1];               // the odd channel: both pixels in one packed FMA
                // T <- T' for both pixels in one instruction
                const f2p nTp = {nT0, nT1};
                asm("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(Tp) : "v"(nTp));
            }
            // ---- sensitivity experiments (tools/r06s_fwd_sensitivity.sh): N extra operations of ONE kind per splat, results unused, the
            // dynamic work unchanged; the slope of the kernel time in N says which resource the loop is short of
#if DNS_FWD_X_VALU > 0
            { float d = px;
#pragma unroll
              for (int x = 0; x < DNS_FWD_X_VALU; ++x) asm volatile("v_mov_b32_e32 %0, %0" : "+v"(d)); }
#endif
#if DNS_FWD_X_EXP > 0
            { float d = px;
#pragma unroll
              for (int x = 0; x < DNS_FWD_X_EXP; ++x) asm volatile("v_exp_f32_e32 %0, %0" : "+v"(d)); }
#endif
#if DNS_FWD_X_SALU > 0
            { uint32_t d = 0;
#pragma unroll
              for (int x = 0; x < DNS_FWD_X_SALU; ++x) asm volatile("s_add_u32 %0, %0, 1" : "+s"(d) : : "scc"); }
#endif
#if DNS_FWD_X_BR > 0
#pragma unroll
            for (int x = 0; x < DNS_FWD_X_BR; ++x) asm volatile("s_cmp_eq_u32 0, 0\n\ts_cbranch_scc1 2f\n\ts_nop 0\n\t2:" : : : "scc");
#endif
#if DNS_FWD_X_LDS > 0
            { float4 d;
#pragma unroll
              for (int x = 0; x < DNS_FWD_X_LDS; ++x) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"((uint32_t)(uintptr_t)&my[t][0]) : "memory"); }
#endif
        }
#else
        while (todo) {
            const int t = __ffsll((unsigned long long)todo) - 1;
            asm("s_bitset0_b64 %0, %1" : "+s"(todo) : "s"(t));      // todo &= todo - 1 as one scalar instruction instead of three
            const float4 g0 = my[t][0];  // x y na nb
            const float4 g1 = my[t][1];  // nc opac ch0 ch1
            float4 g2, g3;
            if (D > 2) g2 = my[t][2];
            if (D > 6) g3 = my[t][3];
            DnsConicE q;
            q.na = g0.z; q.nb = g0.w; q.nc = g1.x;
            const float dx = g0.x - px;
            const float dy0 = g0.y - py0, dy1 = g0.y - py1;
#if DNS_EXP_SYM
            // dns_exponent() for both pixels of the lane, with the two products that only depend on dx formed once
            const float adx = q.na * dx, bdx = q.nb * dx;
#if DNS_FWD_PAIR_OPSEL
            // nb * dy + adx for both pixels as ONE packed FMA on the register pair (adx, nb) the record's (na, nb) turns into:
            // nb is the pair's high half broadcast as the multiplier, adx its low half broadcast as the addend.  hipcc only finds
            // low-half broadcasts and copies nb into a fresh pair first.
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 an = {adx, q.nb}, dyv = {dy0, dy1};
            f2 hu;
            asm("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(hu) : "v"(an), "v"(dyv));
            const float e0 = __builtin_fmaf(dx, hu.x, dy0 * __builtin_fmaf(q.nc, dy0, bdx));
            const float e1 = __builtin_fmaf(dx, hu.y, dy1 * __builtin_fmaf(q.nc, dy1, bdx));
#else
            const float e0 = __builtin_fmaf(dx, __builtin_fmaf(q.nb, dy0, adx), dy0 * __builtin_fmaf(q.nc, dy0, bdx));
            const float e1 = __builtin_fmaf(dx, __builtin_fmaf(q.nb, dy1, adx), dy1 * __builtin_fmaf(q.nc, dy1, bdx));
#endif
#else
            const float e0 = dns_exponent(q, dx, dy0);
            const float e1 = dns_exponent(q, dx, dy1);
#endif
            const float alpha0 = fminf((float)DNS_ALPHA_MAX, g1.y * dns_exp2(e0));
            const float alpha1 = fminf((float)DNS_ALPHA_MAX, g1.y * dns_exp2(e1));
            float ch[8];
            ch[0] = g1.z; ch[1] = g1.w;
            if (D > 2) { ch[2] = g2.x; ch[3] = g2.y; ch[4] = g2.z; ch[5] = g2.w; }
            if (D > 6) { ch[6] = g3.x; ch[7] = g3.y; }
            float nT0, nT1, v0, v1;
            const uint64_t valid0 = dns_ballot(e0 <= 0.f) & dns_ballot(alpha0 >= (float)DNS_ALPHA_MIN) & ~done0;
            const uint64_t valid1 = dns_ballot(e1 <= 0.f) & dns_ballot(alpha1 >= (float)DNS_ALPHA_MIN) & ~done1;
            // a skipped pair takes alpha = 0: its weight is 0 and T x (1 - 0) is T itself, so the blend and the transmittance update
            // below are unconditional.  Only a valid pair can lower T, so "T' <= 1e-4" alone says "this pair saturates the pixel".
            const float a0 = sel0(valid0, alpha0), a1 = sel0(valid1, alpha1);
            // (measured and not kept, round 4: T' = T - alpha T, i.e. one v_fma per pixel instead of (1 - alpha) and a multiply — two
            // vector instructions of 39 less per splat and no faster, 0.493 vs 0.490 ms paired; the reference's T (1 - alpha) stays)
            nT0 = T0 * (1.f - a0); nT1 = T1 * (1.f - a1);
            const uint64_t stop0 = dns_ballot(nT0 <= (float)DNS_T_MIN), stop1 = dns_ballot(nT1 <= (float)DNS_T_MIN);
            v0 = a0 * T0; v1 = a1 * T1;
            any0 |= valid0; any1 |= valid1;
            if (COUNT) {
                n_walked += 1;
                n_live += __popcll(~done0) + __popcll(~done1);            // pixels still open when the splat arrived
                n_blend += __popcll(valid0 & ~stop0) + __popcll(valid1 & ~stop1);
            }
            // Some pixel saturates at this splat only about one splat in four: then the splat is NOT applied to it (weight 0), its T
            // stays, and the last index the backward has to visit for it is the one before (exactly: this entry must not be replayed).
            // The six selects sit behind a wave-uniform branch INSIDE one asm statement: written as a C++ `if`, hipcc copies every
            // accumulator where the two paths meet (14 v_mov per splat), as a straight line they cost 30 cycles per splat.
            {
                const int before = batch_start + t - 1;
                float tmp;
                [[maybe_unused]] uint64_t tmp_s;
                asm volatile(
                    "s_or_b64 vcc, %9, %10\n\t"
                    "s_cmp_eq_u64 vcc, 0\n\t"
                    "s_cbranch_scc1 1f\n\t"
                    "v_mov_b32_e32 %8, %13\n\t"
                    "v_cndmask_b32_e64 %0, %0, 0, %9\n\t"
                    "v_cndmask_b32_e64 %1, %1, 0, %10\n\t"
                    "v_cndmask_b32_e64 %2, %2, %11, %9\n\t"
                    "v_cndmask_b32_e64 %3, %3, %12, %10\n\t"
                    "v_cndmask_b32_e64 %4, %4, %8, %9\n\t"
                    "v_cndmask_b32_e64 %5, %5, %8, %10\n\t"
                    "s_or_b64 %6, %6, %9\n\t"
                    "s_or_b64 %7, %7, %10\n\t"
                    "1:"
                    : "+v"(v0), "+v"(v1), "+v"(nT0), "+v"(nT1), "+v"(last0), "+v"(last1), "+s"(done0), "+s"(done1), "=&v"(tmp)
                    : "s"(stop0), "s"(stop1), "v"(T0), "v"(T1), "s"(before)
                    : "vcc", "scc");
                // all 128 pixels saturated: nothing left to do (not a `break`: a second loop exit makes hipcc copy every accumulator at
                // the latch).  As a data select on `todo`: written as an `if`, hipcc folds it into the loop condition as two
                // compare-to-mask pairs, an and with them, one with exec and a branch on vcc — with the select (and s_bitset0 for
                // todo &= todo - 1) the loop's exit is a plain scalar compare-and-branch: -7.2 % paired (C2), -6.2 % (C5).
                // Measured and not kept: ten more scalar instructions taken out of the common path (per-lane alpha thresholds that
                // turn +inf when a pixel saturates, the stop test as one v_min + v_cmp into vcc, done masks / all-done test /
                // index only behind the branch): 7 instead of 17 scalar instructions per splat and no faster (-7.4 % vs -7.7 %) —
                // it was the shape of the loop's exit, not the number of scalar instructions.
                asm("s_and_b64 %1, %2, %3\n\ts_cmp_eq_u64 %1, -1\n\ts_cselect_b64 %0, 0, %0"
                    : "+s"(todo), "=&s"(tmp_s) : "s"(done0), "s"(done1) : "scc");
            }
#pragma unroll
            for (int k = 0; k < D; ++k) { acc0[k] += ch[k] * v0; acc1[k] += ch[k] * v1; }
            T0 = nT0; T1 = nT1;
        }
#endif
#endif
        // A pixel that blended anything in this batch and is still open: every entry of the batch behind its last blended one was
        // skipped for it, so the END of the batch serves as its "last index" — the backward replays a few no-ops more, and the loop
        // above does not have to remember the index splat by splat (two selects per splat).
        {
            const float bend = __int_as_float(min(batch_start + DNS_WAVE, range_end) - 1);
            last0 = sel(any0 & ~done0, bend, last0);
            last1 = sel(any1 & ~done1, bend, last1);
            any0 = 0ull; any1 = 0ull;
        }
        __builtin_amdgcn_wave_barrier();
    }

#if DNS_FWD_PACKED && !DNS_FWD_UNROLL2
    T0 = Tp.x; T1 = Tp.y;
#pragma unroll
    for (int k = 0; k < (D & ~1); ++k) { acc0[k] = (k & 1) ? accp0[k >> 1].y : accp0[k >> 1].x; acc1[k] = (k & 1) ? accp1[k >> 1].y : accp1[k >> 1].x; }
    if (D & 1) { acc0[D - 1] = acc_odd.x; acc1[D - 1] = acc_odd.y; }
#endif
    // epilogue: background, expected-depth normalisation, stores.  The (wave-uniform) background values are requested together, once
    float dmax = 0.f;
    float bgk[D], bgc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < D; ++k) bgk[k] = a.background ? a.background[k] : 0.f;
    if (DN) {
#pragma unroll
        for (int c = 0; c < 3; ++c) bgc[c] = a.bg_rgb[c];
    }
    if (in0) {
        const size_t pid = img + (size_t)py_i0 * a.width + px_i;
        const float al = 1.f - T0;
        a.alphas[pid] = al;
        a.last_ids[pid] = __float_as_int(last0);
        float raw[8];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float v = acc0[k];
            if (a.background) v += T0 * bgk[k];
            if (k == a.ed_channel) v = v / fmaxf(al, (float)DNS_ED_ALPHA_FLOOR);
            a.render[pid * D + k] = v;
            raw[k] = v;
        }
        if (DN) dmax = fmaxf(dmax, dn_epilogue(a, pid, raw, al, bgc));
    }
    if (in1) {
        const size_t pid = img + (size_t)py_i1 * a.width + px_i;
        const float al = 1.f - T1;
        a.alphas[pid] = al;
        a.last_ids[pid] = __float_as_int(last1);
        float raw[8];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float v = acc1[k];
            if (a.background) v += T1 * bgk[k];
            if (k == a.ed_channel) v = v / fmaxf(al, (float)DNS_ED_ALPHA_FLOOR);
            a.render[pid * D + k] = v;
            raw[k] = v;
        }
        if (DN) dmax = fmaxf(dmax, dn_epilogue(a, pid, raw, al, bgc));
    }
    if (DN) {
        // image-wide max of the expected depth (dn_model.py:535 `depth_im.detach().max()`): depths are >= 0,
        // so their bit patterns order like the floats and one integer atomicMax per wave suffices
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) dmax = fmaxf(dmax, __shfl_xor(dmax, off, DNS_WAVE));
        if (lane == 0 && dmax > 0.f) atomicMax(reinterpret_cast<int *>(a.dn_depth_max) + cam, __float_as_int(dmax));
    }
    if (COUNT && lane == 0 && a.counters) {
        atomicAdd(a.counters + 0, n_entries); atomicAdd(a.counters + 1, n_walked);
        atomicAdd(a.counters + 2, n_live); atomicAdd(a.counters + 3, n_blend);
    }
}

template <int D, bool DN = false, bool COUNT = false>
int launch_fwd(const FwdArgs &fa, int n_cameras, hipStream_t stream)
{
    hipLaunchKernelGGL((raster_fwd_kernel<D, DN, COUNT>), dim3(fa.n_tiles * n_cameras), dim3(FWD_THREADS), 0, stream, fa);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

}  // namespace

extern "C" int dnsplat_raster_fwd(const dnsplat_raster_args *a, dnsplat_stream_t stream_)
{
    if (!a) return DNSPLAT_ERR_INVALID_ARG;
    if (a->tile_size != TILE) return DNSPLAT_ERR_UNSUPPORTED;
    if (a->D < 1 || a->D > DNSPLAT_MAX_CHANNELS) return DNSPLAT_ERR_UNSUPPORTED;
    if (a->width <= 0 || a->height <= 0) return DNSPLAT_ERR_INVALID_ARG;
    // splats / flatten_ids are only dereferenced for list entries: both may be NULL when every tile list is empty (N == 0)
    if (!a->tile_offsets || !a->render || !a->alphas || !a->last_ids) return DNSPLAT_ERR_INVALID_ARG;
    if (a->ed_channel >= a->D) return DNSPLAT_ERR_INVALID_ARG;
    FwdArgs fa;
    fa.width = a->width; fa.height = a->height;
    fa.tw = dns_tiles_w(a->width, TILE);
    fa.n_tiles = fa.tw * dns_tiles_h(a->height, TILE);
    fa.splats = reinterpret_cast<const float4 *>(a->splats);
    fa.flatten_ids = a->flatten_ids;
    fa.tile_offsets = a->tile_offsets;
    fa.tile_ends = a->tile_ends;
    fa.background = a->background;
    fa.ed_channel = a->ed_channel;
    fa.render = a->render; fa.alphas = a->alphas; fa.last_ids = a->last_ids;
    fa.bg_rgb = nullptr; fa.dn_rgb = fa.dn_depth = fa.dn_normal = fa.dn_depth_max = nullptr;
    fa.counters = reinterpret_cast<unsigned long long *>(a->pair_counters);
    fa.keep_masks = reinterpret_cast<unsigned long long *>(a->keep_masks);
    if (a->zero_fill_bytes < 0 || (a->zero_fill_bytes & 15) || (a->zero_fill_bytes > 0 && !a->zero_fill)) return DNSPLAT_ERR_INVALID_ARG;
    fa.zero_fill = a->zero_fill_bytes > 0 ? reinterpret_cast<float4 *>(a->zero_fill) : nullptr;
    fa.zero_fill_vec4 = a->zero_fill_bytes >> 4;
    fa.keep_mask_stride = a->keep_mask_stride;
    if (a->n_cameras < 0) return DNSPLAT_ERR_INVALID_ARG;
    const int C = a->n_cameras > 1 ? a->n_cameras : 1;
    hipStream_t stream = (hipStream_t)stream_;
    if (a->dn) {
        const dnsplat_dn_post *dn = a->dn;
        if (a->D != 7 || a->ed_channel != 3 || !a->background) return DNSPLAT_ERR_UNSUPPORTED;
        if (!dn->background_rgb || !dn->rgb || !dn->depth || !dn->normal || !dn->depth_max) return DNSPLAT_ERR_INVALID_ARG;
        fa.bg_rgb = dn->background_rgb; fa.dn_rgb = dn->rgb; fa.dn_depth = dn->depth; fa.dn_normal = dn->normal;
        fa.dn_depth_max = dn->depth_max;
        if (fa.counters) return launch_fwd<7, true, true>(fa, C, stream);
        return launch_fwd<7, true>(fa, C, stream);
    }
    switch (a->D) {
        case 1: return launch_fwd<1>(fa, C, stream);
        case 2: return launch_fwd<2>(fa, C, stream);
        case 3: return launch_fwd<3>(fa, C, stream);
        case 4: return launch_fwd<4>(fa, C, stream);
        case 5: return launch_fwd<5>(fa, C, stream);
        case 6: return launch_fwd<6>(fa, C, stream);
        case 7: return launch_fwd<7>(fa, C, stream);
        case 8: return launch_fwd<8>(fa, C, stream);
    }
    return DNSPLAT_ERR_UNSUPPORTED;
}
