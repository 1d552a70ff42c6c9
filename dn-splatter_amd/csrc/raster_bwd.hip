// raster_bwd.hip — per-tile alpha compositing, backward (stage 4 of include/dnsplat.h).
//
// Replaces gsplat 1.0.0 rasterize_to_pixels_bwd and the legacy rasterize_backward /
// nd_rasterize_backward (call sites dn_splatter/dn_model.py:495, :564; rule set SURVEY.md A.7).
//
// The reference design is pixel-parallel: every thread owns a pixel, walks the tile list back to
// front, and each of the ~15 per-splat partial gradients is warp-reduced and atomically added once
// per (warp, splat).  On a 64-wide wave that reduction (15 values x 6 DPP steps) costs more than
// the gradient math itself.  This kernel turns the problem by 90 degrees — a wave64 SYSTOLIC pass:
//
//   * the unit of work is a 16 x 8 HALF tile = one wave = one workgroup (DNS_BWD_ROWS; whole tiles made the units twice as
//     long and the launch's drain with them); lane l owns TWO SPLATS of the current bucket of 128 (lane 0 = back-most),
//     carried as packed fp32 pairs, and keeps their 2 x 16 partial gradients in VGPRs;
//   * the 128 pixels stream through the lanes, back to front: one pixel per lane per step, lane l one step behind lane
//     l-1.  What travels with a pixel is only its running state (T, S_a, S_b), where
//     S = sum_k buffer_k * v_k collapses the reference's per-channel `buffer` into one scalar per gradient group
//     (v_alpha = T*(c.v) - S/(1-alpha));
//   * the state travels THROUGH THE PIXEL'S LDS ROW (DNS_BWD_LDS_STATE): every lane reads its pixel's whole row at the start of a
//     step anyway, so the lane before only has to have written the three state words there at the end of its step — a
//     ds_write2_b32 + ds_write_b32 of the registers where they are, LDS operations of a wave complete in order.  (Until round 3: three v_mov_dpp row_shr:1 per step inside
//     rows of 16 lanes, LDS only between the rows; -3.1 % paired.)  The switch of splats still happens in groups of 16 lanes;
//   * per-pixel constants (upstream gradient, last contributing index) sit in a 6 KiB LDS table (128 rows of 48 bytes), read
//     with conflict-free ds_read_b128 (48-byte stride over consecutive lanes), issued before and awaited after the
//     row-independent arithmetic of the step;
//   * state leaving lane 63 is parked back in the pixel's LDS row and picked up by lane 0 in the next (nearer) bucket — the
//     arithmetic order per pixel is exactly the reference's back-to-front replay (T *= 1/(1-alpha); buffer += c*alpha*T);
//   * a lane in an idle slot reads and writes a DUMMY row of the pixel table (zero cotangents, last index -1): no pair of it is ever
//     valid and its state store needs no narrowed exec (round 4, DNS_BWD_DUMMY_ROW);
//   * the stream is CONTINUOUS over the buckets of a unit: 15 idle slots separate two buckets, which lets the 16 lanes of
//     a row change their splats at the same wave-uniform step (four staggered group switches per bucket) instead of
//     draining and refilling the whole array — 143 steps per bucket instead of 128 + 63;
//   * the last, partly filled bucket of a unit is FOLDED (DNS_BWD_FOLD): with <= 64 (<= 32) splats the four rows are re-cut
//     into two (four) arrays that each hold all of its splats and stream a share of the pixels: 111 (79) steps;
//   * a splat's 16 partials are summed over all 128 pixels in registers: NO cross-lane reduction, and ONE atomic row per
//     (half tile, splat).  The flush is transposed through LDS so that each global_atomic_add_f32 instruction covers whole
//     64-byte gradient records (16 lanes per record).
//
// Bucket compaction.  A tile list holds every splat whose tile box touches the tile; on the benchmark scenes a good part of
// those cannot reach alpha >= 1/255 at any pixel centre of the half tile.  Only contributing splats enter a bucket: the
// benchmark instantiation (MASKS) takes the forward's rectangle-test ballots (dnsplat_raster_args.keep_masks, one 64-bit word
// per 64 list entries), the others test 64 list entries at once against the half tile's rectangle themselves (dns_cull_rect:
// exact minimum of the quadratic over the rectangle, one lane per entry) and ballot-compact the survivors into an LDS queue.
// Culled entries are exactly those the per-pixel test would skip for all 128 pixels: results are unchanged.
//
// Channel groups: channels >= SPLIT (the normal channels of the fused pass) are rendered by the
// reference with xys.detach() (dn_model.py:562), so their share of d/d(alpha) must not reach
// v_xy / |v_xy| while it does reach v_conic and v_opacity.  Hence two S states.  SPLIT is a
// compile-time constant for the two shapes that occur (SPLIT == D: everything feeds xy; D == 7,
// SPLIT == 4: the fused colour+depth | normal pass); other splits take the generic instantiation.

#include <type_traits>

#include "splat_common.h"

namespace {

constexpr int TILE = 16;
// rows of a tile one wave owns.  With whole tiles a work unit took ~570 us of a 2 ms launch and the last third of the launch was a
// drain at 39 % occupancy (tools/bwd_timeline.sh): the end of the kernel lasts about one unit.  Half tiles make the units half as
// long (and the rectangle cull and the per-pixel list bound tighter), at the price of 15 idle slots per 128 instead of per 256
// stream steps and two atomic rows per (tile, splat).
// What is left of that drain (tools/bwd_timeline.sh, C2: 16200 units of 501 +- 6 % steps, ~4 per resident wave; the chip is full
// for 1.34 ms and empties over the last 0.38 ms) is NOT idle capacity: a unit that runs with one or two waves on its SIMD takes
// ~200 us instead of ~380 us, i.e. two waves already keep a SIMD's vector unit about as busy as four.  Two attempts to "repair"
// the tail confirmed it and were removed again: (a) as many persistent waves as the chip holds, each walking the units with a
// fixed stride: +10 % (the waves of a SIMD are not served evenly — some finish four units while others finish two — so a fixed
// assignment only moves the imbalance); (b) cutting the last 10-27 % of the units into 2-3 pieces over pixel windows so that the
// drain lasts one piece: -0.5 % ... +5 % (a piece costs max(64, pixels + 15) steps per bucket).
#ifndef DNS_BWD_ROWS
#define DNS_BWD_ROWS 8
#endif
// Register budget.  The benchmark instantiation (fused pass, forward's keep masks) sits right at the 128-VGPR line that
// separates 4 from 3 waves per SIMD, and hipcc's allocation lands on 120 ... 130 depending on details of the control flow
// around the step loop.  That instantiation is therefore PINNED to 4 waves (amdgpu_waves_per_eu): what does not fit (two
// values at the time of writing) is spilled to scratch outside the step loop — stored once per work unit, reloaded once
// per bucket (tools/check_asm_hazards.py checks that the step loop itself stays scratch-free).  The other instantiations
// are left alone: forcing the same budget on the instantiation that re-derives the masks (164 VGPRs) put nine
// loop-invariant splat parameters into scratch that were re-read every step: measured 1.68 -> 2.74 ms.
#ifndef DNS_BWD_WAVES_PER_EU
#define DNS_BWD_WAVES_PER_EU 4
#endif
#define DNS_BWD_OCCUPANCY(pinned) __attribute__((amdgpu_waves_per_eu((pinned) ? DNS_BWD_WAVES_PER_EU : 1, (pinned) ? DNS_BWD_WAVES_PER_EU : 8)))
// pixel coordinates from an LDS table read one step ahead (1) or recomputed from the pixel counter every step (0).
// Measured (paired A/B, tools/ab_kernels.py): the table is not faster — the six integer / convert instructions it removes were
// not on the kernel's critical resource.
#ifndef DNS_BWD_COORD_TABLE
#define DNS_BWD_COORD_TABLE 0
#endif
// "did this lane's splat meet any valid pixel" as two per-step bool accumulations (1), or derived at the flush from the
// partial sums themselves (0): a record whose 16 sums are all exactly zero need not be added to anything.  The bools are
// live across the step's one branch (the park store of lane 63), where hipcc merges each with three scalar instructions
// per step; the scalar unit is shared by the 12 waves of a CU.
#ifndef DNS_BWD_TOUCH_FLAGS
#define DNS_BWD_TOUCH_FLAGS 0
#endif
// wave-uniform loop bounds moved to scalar registers explicitly (readfirstlane): left alone, hipcc keeps the step counter in a
// VGPR with a per-lane exit mask (v_add, v_cmp, s_or, s_andn2 exec per step)
// 0: two instantiations of the step loop, the clamp-free one for groups that hold no splat with opacity > 0.999 (one compare,
//    one select and the min per splat less); 1: only the clamping loop.  With both loops in the kernel hipcc needs ~34 more
//    VGPRs (154 vs 120 with the forward's keep masks): mode 1 is what lets a fourth wave per SIMD in.
#ifndef DNS_BWD_CLAMP_MODE
#define DNS_BWD_CLAMP_MODE 1
#endif
#ifndef DNS_BWD_SCALAR_LOOP
#define DNS_BWD_SCALAR_LOOP 1
#endif
// Folding of the last, partly filled bucket of a work unit (1) or the plain array (0).
// The 64 lanes are four ROWS OF 16 linked through LDS instead of one 64-lane shift register: the pixel state moves by
// row_shr:1 inside a row, the last lane of a row parks it in the pixel's LDS row and the first lane of the next row picks it
// up from there one step later — for free, every lane reads its pixel's row (for bin_final) anyway, and the park is the
// store lane 63 already did.  A bucket costs ~143 steps however few splats it holds, and the last bucket of a unit holds
// 64 on average: with <= 64 (<= 32) splats the four rows are re-cut into two (four) independent arrays that each hold
// ALL of the bucket's splats and stream a share of the pixels — 80 + 48 (48 + 48 + 16 + 16; 56 + 40 + 24 + 8 before the dummy
// pixel row asked for windows at multiples of 16, DNS_BWD_DUMMY_ROW) of the 128, unequal because a later row is still busy with
// the previous bucket for 16 more steps per row — and the bucket ends after 111 (79; 71) steps.
// State slots of a pixel row ordered (S_a, T, S_b, bin_final) instead of (T, S_a, S_b, bin_final): S_a and S_b arrive in the
// LOW halves of the two aligned register pairs of the row's third ds_read_b128, the value the step computes next for each chain
// (S after splat A) goes into the high half once T / bin_final are consumed, and the packed operand (S before A, S before B) is
// that pair as it stands — no v_mov to build it.  The park store writes 12 bytes and leaves bin_final where it is.
#ifndef DNS_BWD_PAIR_STATE
#define DNS_BWD_PAIR_STATE 1
#endif
// The per-pixel running state (T, S_a, S_b) moves from lane to lane THROUGH the pixel's LDS row instead of three v_mov_dpp: every
// lane reads its pixel's whole row at the start of a step anyway (the state slots came along unused except in the first lane of
// a DPP row), so the lane before only has to have written them — two small LDS stores per step, LDS operations of a wave are in order.
// Needs the (S_a, T, S_b, bin_final) slot order and the folded layout's rule that every lane may park.
#ifndef DNS_BWD_LDS_STATE
#define DNS_BWD_LDS_STATE 1
#endif
// measured and not kept: deciding "not an idle slot, sigma >= 0, alpha >= 1/255" ahead of the row wait and only "entry <= the pixel's
// last index" behind it (two selects more, four compares earlier): +0.4 % at C2, +1.7 % at C5 — the step is bound by issue, not by
// the latency of its tail
#ifndef DNS_BWD_PREVALID
#define DNS_BWD_PREVALID 0
#endif
// D < 8: the eighth cotangent slot of a pixel row is free and carries the x coordinate of the NEXT pixel's centre (the lane's pixel
// of the next step), which saves that step an and, a convert and an add
#ifndef DNS_BWD_PX_SLOT
#define DNS_BWD_PX_SLOT 1
#endif
#ifndef DNS_BWD_FLUSH_ALL
#define DNS_BWD_FLUSH_ALL 1
#endif
// round 4, from the ISA of the step: the magnitude sums of the mean gradient as v_fma_f32 |a|, |b|, acc (the products are no longer
// formed twice) and the second channel group's share of d/d(alpha) as two chained packed FMAs: 42 -> 39 packed instructions per step
// Idle slots read and write a DUMMY pixel row (row NPIX: zero cotangents, last index -1, x of column 0) instead of aliasing a live
// one: "this lane is between two buckets" then needs no term in the pair's validity (no list index is <= -1), the state store
// needs no narrowed exec, and the row index is a select instead of a mask — four scalar instructions per step less (s_and x 2,
// s_and_saveexec, s_or), the vector count unchanged.  The folded arrays' pixel windows start at multiples of 16 for it (the dummy
// row's "x of the next pixel" slot can only name one column): 48 + 48 + 16 + 16 instead of 56 + 40 + 24 + 8 pixels for a four-fold
// bucket, which then lasts 79 instead of 71 steps.  Needs DNS_BWD_LDS_STATE, DNS_BWD_PX_SLOT, DNS_BWD_FOLD, no coordinate table.
#ifndef DNS_BWD_DUMMY_ROW
#define DNS_BWD_DUMMY_ROW 1
#endif
// the group switch requests its two list entries together and its two records together (two memory round trips instead of four)
// dx of the next step formed at the end of the current one (see the step loop)
#ifndef DNS_BWD_DX_CARRY
#define DNS_BWD_DX_CARRY 1
#endif
#ifndef DNS_BWD_BATCHED_SWITCH
#define DNS_BWD_BATCHED_SWITCH 1
#endif
#ifndef DNS_BWD_ABS_FMA
#define DNS_BWD_ABS_FMA 1
#endif
#ifndef DNS_BWD_VA_CHAIN
#define DNS_BWD_VA_CHAIN 1
#endif
#ifndef DNS_BWD_FOLD
#define DNS_BWD_FOLD 1
#endif
constexpr int ROWS = DNS_BWD_ROWS;
constexpr int PARTS = TILE / ROWS;              // waves (workgroups) per tile
constexpr int NPIX = TILE * ROWS;
constexpr int BUCKET = 2 * DNS_WAVE;   // splats in the array at a time: two per lane, processed as packed fp32 pairs
#ifndef DNS_BWD_GROUP
#define DNS_BWD_GROUP 16
#endif
constexpr int GROUP = DNS_BWD_GROUP;            // lanes that change splats at the same step
constexpr int NGROUP = DNS_WAVE / GROUP;
static_assert(!DNS_BWD_FOLD || GROUP == 16, "the folded arrays start at the DPP rows' (16 lanes) switch steps");
#ifndef DNS_BWD_FLUSH_REC
#define DNS_BWD_FLUSH_REC 16
#endif
// The group's flush as ONE pass over both splats of every lane (1) instead of an A pass and a B pass (0).  From the ISA of the two
// passes (round 6): per 16-record round a ds_bpermute for the record's Gaussian id behind a computed lane address, a 64-bit shift of
// the group's mask and a compare for "was this slot filled", a sign extension + 64-bit shift + 64-bit add for the record's address
// — ten vector instructions per atomic instruction, eight atomic instructions per switch — and fourteen v_mul / v_mov per pass to
// line the 16 sums of ONE splat up for four ds_write_b128.  Merged: the partial sums are parked as the (A, B) register pairs they
// already are (ds_write_b64, the scale factors as packed multiplies on both splats at once), the 16 owning lanes also park the two
// records' ADDRESSES (64-bit, formed once per record instead of once per (record, column); an unfilled slot parks -1), and every
// lane of a round reads its (A, B) values with one ds_read_b64 and the two addresses with one ds_read_b128: a sign test and one
// 64-bit add per atomic.
#ifndef DNS_BWD_MERGED_FLUSH
#define DNS_BWD_MERGED_FLUSH 1
#endif
[[maybe_unused]] constexpr int FLUSH_REC = DNS_BWD_FLUSH_REC;    // gradient records staged in LDS per round of the transposed flush
constexpr int PERIOD = NPIX + GROUP - 1;        // steps from one bucket to the next: 256 pixels + GROUP - 1 idle slots
// the step loop accumulates the mean gradient in units of the half-gradients of the exponent (splat_common.h)
#if DNS_EXP_SYM
constexpr float XY_SCALE = -2.f / DNS_LOG2E;
#else
constexpr float XY_SCALE = 1.f;
#endif


struct BwdArgs {
    int width, height, tw, n_tiles;            // n_tiles: per camera; the launch covers n_tiles x cameras stacked tile grids
    unsigned long long *counters;              // measurement instantiation only (dnsplat_raster_args.pair_counters)
    int n_cameras;
    const unsigned long long *__restrict__ keep_masks;   // MASKS instantiations: the forward's rectangle-test ballots (dnsplat.h)
    long long keep_mask_stride;
    const uint32_t *sat_flag;                  // dnsplat_raster_args.saturation_flag: the launch runs as two kernels, one of which leaves at once
    const float4 *__restrict__ splats;
    const int32_t *__restrict__ flatten_ids;
    const int32_t *__restrict__ tile_offsets;
    const int32_t *__restrict__ tile_ends;     // or NULL: the list of tile t ends at tile_offsets[t + 1]
    const float *__restrict__ background;
    int ed_channel;
    const float *__restrict__ render;
    const float *__restrict__ alphas;
    const int32_t *__restrict__ last_ids;
    const float *__restrict__ v_render;
    const float *__restrict__ v_alphas;
    int xy_split;
    float *__restrict__ v_splats;
    // DET instantiations (dnsplat_raster_args.det_partials): the record of (half tile `part`, list entry idx) is STORED at
    // det + (part * det_cap + idx) * 16 instead of being added atomically to v_splats[gaussian]
    float *__restrict__ det;
    long long det_cap;
    // fused dn-splatter epilogue (DN instantiation only): cotangents of rgb / filled depth / normal / accumulation
    const float *__restrict__ bg_rgb;
    const float *__restrict__ dn_v_rgb;
    const float *__restrict__ dn_v_depth;
    const float *__restrict__ dn_v_normal;
    const float *__restrict__ dn_v_acc;
};



typedef float v4f __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 zero2_init() { f2 z = {0.f, 0.f}; return z; }

// One pixel row (48 bytes) of the LDS table as three ds_read_b128.  Left to itself hipcc scalarises the
// row (its fields are carried across the loop back-edge one by one) and re-merges it into
// ds_read_b32 / ds_read2_b32, which at a 48-byte lane stride are 4-way bank conflicted (measured: 63 % of
// all LDS cycles of this kernel were conflict cycles).  ds_read_b128 at that stride is conflict-free
// (MI355X_MICROARCH.md, LDS lane groups).  The loads are issued here and waited for in row_wait(), which
// ties the destination registers to the s_waitcnt so that no use can be scheduled above it.
// `token` is any value the row-independent arithmetic of the step is derived from: passing it through the asm
// pins that arithmetic BEHIND the loads in program order (otherwise hipcc hoists it above them and the wait
// follows the loads immediately).
__device__ __forceinline__ void row_issue(uint32_t byte_addr, v4f &r0, v4f &r1, v4f &r2, int &token)
{
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2), "+v"(token)
                 : "v"(byte_addr)
                 : "memory");
}
// `done` is the last value of the row-independent arithmetic: routing it through the asm keeps that arithmetic
// AHEAD of the wait (hipcc is otherwise free to sink it below).
__device__ __forceinline__ void row_wait(v4f &r0, v4f &r1, v4f &r2, f2 &done)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(done) : : "memory");
}

// Pixel-centre coordinates of the NEXT step's pixel from the per-wave table (one ds_read_b64, conflict-free at an 8-byte lane
// stride), requested together with the row and completed by the same wait: replaces and / cvt / add / shift / cvt / add per step.
__device__ __forceinline__ void coord_issue(uint32_t byte_addr, f2 &nxt, int &token)
{
    asm volatile("ds_read_b64 %0, %2" : "=&v"(nxt), "+v"(token) : "v"(byte_addr) : "memory");
}
__device__ __forceinline__ void row_wait(v4f &r0, v4f &r1, v4f &r2, f2 &nxt, f2 &done)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(nxt), "+v"(done) : : "memory");
}

// acc += a * (b.x or b.y broadcast to both halves): v_pk_fma_f32 with op_sel picking one half of the b pair
__device__ __forceinline__ void pk_fma_bcast(f2 &acc, f2 a, f2 b, int hi)
{
    if (hi) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(a), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
}

// a * (b.x or b.y broadcast to both halves): the first term of such a sum
__device__ __forceinline__ f2 pk_mul_bcast(f2 a, f2 b, int hi)
{
    f2 r;
    if (hi) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float dpp_wave_shr1(float from_prev, float lane0_value)
{
#if DNS_BWD_FOLD
    // lane l receives `from_prev` of lane l-1 of its row of 16; the first lane of a row (no source) keeps `lane0_value`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0_value), __float_as_int(from_prev),
                                                      0x111 /* row_shr:1 */, 0xf, 0xf, false));
#else
    // lane l receives `from_prev` of lane l-1; lane 0 (no source) keeps `lane0_value`
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(lane0_value), __float_as_int(from_prev),
                                                      0x138 /* wave_shr:1 */, 0xf, 0xf, false));
#endif
}

// SPLIT >= 0: compile-time split; SPLIT < 0: run-time a.xy_split
// COUNT: measurement build of the fused pass (bench.py's VALU roofline): tallies the (pixel, splat) slots the stream issues and
// the pairs it replays into a.counters[4..5]; never the instantiation that is timed.
// MASKS: which list entries can contribute to this half tile comes from the forward (a.keep_masks) instead of being re-derived
// with dns_cull_rect: same decisions (the forward ran the same test on the same records), no record gathers for rejected
// entries, and the test's registers (the kernel's VGPR peak) are gone.
// CLAMP_LOOP = false: the step loop without the alpha cap (no min, no "gradient only where not clamped" compare and select: 82 instead
// of 86 vector instructions per step, -3.6 % measured).  Only right when no splat of the launch has an opacity above the cap, which
// the projection reports in a device word (dnsplat_proj_out.saturation_flag).  Nothing on the host knows the answer without a
// sync, so both instantiations are launched and the one that does not apply leaves at its first instruction (a few us).  Two loop
// bodies inside ONE kernel were tried in round 1 / 2: 154 VGPRs, or 36 spilled values under the 128-VGPR pin (+28 %).
// DET: deterministic gradient scatter (DNSPLAT_DETERMINISTIC, a test / debug mode).  The sums of a (half tile, splat) are the same
// numbers in every run — they are formed in registers in stream order — but the order in which the atomic rows of the ~14 tiles x 2
// halves a Gaussian touches reach its gradient record is not, and fp32 addition does not commute with that.  DET stores every
// row in its own slot of a buffer indexed by (half, sorted list index) instead; dnsplat_det_reduce adds up each Gaussian's slots in
// list order.  The last bucket is not folded (a folded splat sits in 2 or 4 lanes, i.e. would have 2 or 4 rows for one slot).
template <int D, int SPLIT, bool DN, bool COUNT = false, bool MASKS = false, bool CLAMP_LOOP = true, bool DET = false>
__global__ __launch_bounds__(DNS_WAVE) DNS_BWD_OCCUPANCY(DN && MASKS && !COUNT && !DET) void raster_bwd_kernel(BwdArgs a)
{
    if (a.sat_flag && (*a.sat_flag != 0u) != CLAMP_LOOP) return;
    __shared__ float4 pix[NPIX + (DNS_BWD_DUMMY_ROW ? 1 : 0)][3];   // [p][0..1] = v_k, [p][2] = (S_a, T, S_b, bin_final) (DNS_BWD_PAIR_STATE) or (T, S_a, S_b, bin_final)
    // compacted list indices waiting for a bucket (never more than 127 + 64) + the 1 KiB staging area of the transposed flush, which
    // ALIASES queue entries >= 64: while a pass is flushed only the < 64 left-over entries at the front of the queue are
    // live.  13.5 KiB per wave = 12 tiles in flight per CU (3 waves per SIMD, the VGPR limit) instead of 11.
    __shared__ int32_t queue[BUCKET + DNS_WAVE];
#if DNS_BWD_MERGED_FLUSH
    __shared__ f2 flush2[GROUP][DNS_REC];             // [record][column] = (splat A's sum, splat B's sum)
    __shared__ ulonglong2 flush_addr[GROUP];          // [record] = addresses of A's and B's gradient record (or -1: slot not filled)
#else
    __shared__ float4 flush[FLUSH_REC][4];
#endif
#if DNS_BWD_COORD_TABLE
    __shared__ float2 coord[NPIX];             // pixel-centre coordinates of the half tile, row-major
#endif

    // batch of cameras: unit u = blockIdx / PARTS works on tile u % n_tiles of camera u / n_tiles (stacked images and lists)
    const int unit = blockIdx.x / PARTS;
    const int cam = unit / a.n_tiles;
    const int tile = dns_tile_of_block(unit - cam * a.n_tiles, a.n_tiles, a.tw);
    const size_t img = (size_t)cam * a.width * a.height;
    const int part = blockIdx.x % PARTS;
    const int lane = threadIdx.x;
    const int range_start = a.tile_offsets[cam * a.n_tiles + tile];
    const int range_end = a.tile_ends ? a.tile_ends[cam * a.n_tiles + tile] : a.tile_offsets[cam * a.n_tiles + tile + 1];
    if (range_end <= range_start) return;
    [[maybe_unused]] unsigned long long n_slots = 0, n_pairs = 0;
#ifdef DNS_BWD_TIMELINE
    // instrumented build (tools/bwd_timeline.sh): per work unit (start, end) of the 100 MHz wall clock, its work and where it ran, written
    // through the v_alphas pointer, which the fused (DN) pass does not use
    unsigned long long *dbg_tl = DN ? (unsigned long long *)a.v_alphas : nullptr;
    const unsigned long long tl_t0 = wall_clock64();
    unsigned long long tl_steps = 0, tl_splats = 0;
#endif
    const int tile_x0 = (tile % a.tw) * TILE, tile_y0 = (tile / a.tw) * TILE + part * ROWS;
    const int split = SPLIT >= 0 ? SPLIT : a.xy_split;

    // ---- prologue: per-pixel table --------------------------------------------------------------
    int hi = -1;
#pragma unroll
    for (int j = 0; j < NPIX / DNS_WAVE; ++j) {
        const int p = lane + DNS_WAVE * j;
        const int xi = tile_x0 + (p & 15), yi = tile_y0 + (p >> 4);
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
        float T_final = 1.f, sa = 0.f, sb = 0.f;
        int bin_final = -1;
        if (DN && xi < a.width && yi < a.height) {
            // cotangents of the dn-splatter images -> cotangents of the raw composite (backward of
            // dn_model.py:526-537, 577-578; forward twin: dn_epilogue in raster_fwd.hip)
            const size_t pid = img + (size_t)yi * a.width + xi;
            const float al = a.alphas[pid];
            T_final = 1.f - al;
            bin_final = a.last_ids[pid];
            float raw[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) raw[k] = a.render[pid * 7 + k];
            float va = a.dn_v_acc ? a.dn_v_acc[pid] : 0.f;
            const float one_minus = 1.f - al;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float bg = a.bg_rgb[c];
                const float pre = raw[c] + one_minus * bg;
                const float g = (pre >= 0.f && pre <= 1.f) ? a.dn_v_rgb[pid * 3 + c] : 0.f;   // clamp(0,1)
                v[c] = g;
                va -= bg * g;                                                                 // (1 - alpha) * background
            }
            {
                const float vd = al > 0.f ? a.dn_v_depth[pid] : 0.f;                          // where(alpha > 0, depth, max)
                const float inv = 1.f / fmaxf(al, (float)DNS_ED_ALPHA_FLOOR);
                v[3] = vd * inv;
                if (al >= (float)DNS_ED_ALPHA_FLOOR) va -= raw[3] * vd * inv;
            }
            {
                const float nx = raw[4], ny = raw[5], nz = raw[6];
                const float inrm = 1.f / sqrtf(nx * nx + ny * ny + nz * nz);
                const float hx = nx * inrm, hy = ny * inrm, hz = nz * inrm;
                const float gx = 0.5f * a.dn_v_normal[pid * 3], gy = 0.5f * a.dn_v_normal[pid * 3 + 1],
                            gz = 0.5f * a.dn_v_normal[pid * 3 + 2];                            // (n_hat + 1) / 2
                const float dot = hx * gx + hy * gy + hz * gz;
                v[4] = (gx - hx * dot) * inrm; v[5] = (gy - hy * dot) * inrm; v[6] = (gz - hz * dot) * inrm;
            }
            sa = -T_final * va;                       // rgb/depth carry no kernel-level background
            sb = T_final * (v[4] + v[5] + v[6]);      // the legacy normal pass composites over ones
            if (al <= 0.f) bin_final = -1;
        } else if (xi < a.width && yi < a.height) {
            const size_t pid = img + (size_t)yi * a.width + xi;
#pragma unroll
            for (int k = 0; k < D; ++k) v[k] = a.v_render[pid * D + k];
            const float al = a.alphas[pid];
            float va = a.v_alphas ? a.v_alphas[pid] : 0.f;
            T_final = 1.f - al;
            bin_final = a.last_ids[pid];
            if (a.ed_channel >= 0) {
                // out_ed = acc_ed / max(alpha, 1e-10)
#pragma unroll
                for (int k = 0; k < D; ++k)
                    if (k == a.ed_channel) {
                        const float inv = 1.f / fmaxf(al, (float)DNS_ED_ALPHA_FLOOR);
                        const float vd = v[k];
                        v[k] = vd * inv;
                        if (al >= (float)DNS_ED_ALPHA_FLOOR) va -= a.render[pid * D + k] * vd * inv;
                    }
            }
            float bga = 0.f, bgb = 0.f;
            if (a.background) {
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    const float t = a.background[k] * v[k];
                    if (k < split) bga += t; else bgb += t;
                }
            }
            // S starts at -(T_final * d/d(alpha_img) share) so that v_alpha = T*cv - ra*S needs no extra term
            sa = -T_final * (va - bga);
            sb = T_final * bgb;
            // a pixel nothing was blended into walks no list entry at all
            if (al <= 0.f) bin_final = -1;
        }
#if DNS_BWD_COORD_TABLE
        coord[p] = make_float2((float)xi + 0.5f, (float)yi + 0.5f);
#endif
        pix[p][0] = make_float4(v[0], v[1], v[2], v[3]);
        if (DNS_BWD_PX_SLOT && D < 8) v[7] = (float)tile_x0 + 0.5f + (float)((p + 1) & 15);   // see DNS_BWD_PX_SLOT
        pix[p][1] = make_float4(v[4], v[5], v[6], v[7]);
#if DNS_BWD_PAIR_STATE
        pix[p][2] = make_float4(sa, T_final, sb, __int_as_float(bin_final));
#else
        pix[p][2] = make_float4(T_final, sa, sb, __int_as_float(bin_final));
#endif
        hi = max(hi, bin_final);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) hi = max(hi, __shfl_xor(hi, off, DNS_WAVE));
    // the same number in every lane: say so, or everything derived from it (the batch counter, the forward's keep masks, the number
    // of queued entries, the bucket's size and fold) is computed by vector instructions on 64 copies
    hi = __builtin_amdgcn_readfirstlane(hi);
    hi = min(hi, range_end - 1);
    if (hi < range_start) return;
#if DNS_BWD_DUMMY_ROW
    static_assert(DNS_BWD_LDS_STATE && DNS_BWD_PAIR_STATE && DNS_BWD_PX_SLOT && DNS_BWD_FOLD && !DNS_BWD_COORD_TABLE && !DNS_BWD_PREVALID && GROUP == 16,
                  "the dummy row replaces the exec-narrowed state store of the LDS hand-off");
    if (lane == 0) {     // what a lane in an idle slot reads: nothing to add, no entry to replay, column 0 next; its state store lands here too
        pix[NPIX][0] = make_float4(0.f, 0.f, 0.f, 0.f);
        pix[NPIX][1] = make_float4(0.f, 0.f, 0.f, (float)tile_x0 + 0.5f);
        pix[NPIX][2] = make_float4(0.f, 1.f, 0.f, __int_as_float(-1));
    }
#endif
    __builtin_amdgcn_wave_barrier();

    const float fx0 = (float)tile_x0 + 0.5f, fy0 = (float)tile_y0 + 0.5f;
    const float rxh = fx0 + 15.f, ryh = fy0 + (float)(ROWS - 1);
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    int cursor = hi;  // next (highest) list index not yet examined
    // MASKS: entries are taken in the forward's batches of 64 (aligned to the start of the list), highest batch first
    [[maybe_unused]] int batch = (hi - range_start) >> 6;
    [[maybe_unused]] const unsigned long long *masks =
        MASKS ? a.keep_masks + (size_t)part * a.keep_mask_stride + (range_start >> 6) + cam * a.n_tiles + tile : nullptr;
    [[maybe_unused]] const uint64_t gt_mask = (lane == 63) ? 0ull : (~0ull << (lane + 1));
    int qn = 0;       // entries waiting in the queue beyond the current bucket (wave-uniform)
    const uint32_t pix_base = (uint32_t)(uintptr_t)&pix[0][0];   // LDS byte address of the table
#if DNS_BWD_COORD_TABLE
    const uint32_t coord_base = (uint32_t)(uintptr_t)&coord[0];
    f2 pxy = zero2_init();                                       // centre of the pixel the lane works on in the coming step
#endif

    // ---- per-lane state of the stream: two splats (x = A, farther; y = B, nearer), their partial sums, the
    // pixel counter of the lane's current bucket and the pixel state handed to the next lane --------------------
    const f2 zero2 = {0.f, 0.f};
    int gid_a = 0, gid_b = 0;
    int cmp_a = 0x7fffffff, cmp_b = 0x7fffffff;   // list index of the lane's splats; "none" lies above every bin_final
    [[maybe_unused]] f2 ca = zero2, cb = zero2, cc = zero2;      // the unscaled conic: only the DNS_EXP_SYM = 0 loop reads it
    f2 sx = zero2, sy = zero2, opac = zero2, na = zero2, nb = zero2, nc = zero2;
    f2 ch[8];
    f2 g_x = zero2, g_y = zero2, g_ca = zero2, g_cb = zero2, g_cc = zero2, g_o = zero2, g_ax = zero2, g_ay = zero2;
    f2 g_ch[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { ch[k] = zero2; g_ch[k] = zero2; }
    [[maybe_unused]] bool touched_a = false, touched_b = false;
    int p = -(1 << 20);                           // negative = not started; a group's switch sets it to -(lane % GROUP)
    // DNS_BWD_FOLD: the lane's array streams pixels [p_first, p_first + p_count) of the half tile (all of them unless folded)
    [[maybe_unused]] int p_first = 0, p_count = NPIX;
    [[maybe_unused]] bool folded = false;          // the bucket in the lanes is a folded (hence the last) one (wave-uniform)
    [[maybe_unused]] float T_out = 0.f, SA_out = 0.f, SB_out = 0.f;
    [[maybe_unused]] int qs = -(1 << 24), qs_lim = 0;   // DNS_BWD_DUMMY_ROW: 16 x (position in the lane's pixel window), 16 x (its length)
    [[maybe_unused]] float fy_arr = 0.f;          // DNS_BWD_DUMMY_ROW: y of the centre of the window's first pixel row
    [[maybe_unused]] uint32_t row_run = 0;        // DNS_BWD_DUMMY_ROW: LDS address of the row of the lane's pixel counter
    [[maybe_unused]] f2 dx_cur = zero2;           // DNS_BWD_DX_CARRY: sx - (x of the pixel of the coming step), formed at the end of the step before
    [[maybe_unused]] float px_cur = 0.f;          // DNS_BWD_PX_SLOT: x of the centre of the pixel the lane works on in the coming step

    const int col = lane & 15;
    const bool col_used = col < REC_CH0 + D || col >= REC_ABSX;
#if !DNS_BWD_MERGED_FLUSH
    const float *fl = reinterpret_cast<const float *>(&flush[0][0]);
#endif
    int prev_take = 0;

    for (;;) {
        // ==== bucket boundary: bring the left-over queue entries to the front, then test list entries (64 at a
        // time, one per lane) until a full bucket of contributing splats is waiting or the list is exhausted ====
        if (qn > 0) {
            const int moved = lane < qn ? queue[BUCKET + lane] : 0;
            __builtin_amdgcn_wave_barrier();
            if (lane < qn) queue[lane] = moved;
        }
        if (MASKS) {
            while (qn < BUCKET && batch >= 0) {
                uint64_t m = masks[batch];                                    // wave-uniform
                const int bstart = range_start + (batch << 6);
                const int top = hi - bstart;                                  // entries above `hi` were never blended by any pixel here
                if (top < 63) m &= (2ull << top) - 1ull;
                // lane l looks at entry bstart + l; the queue is filled in descending list order
                if ((m >> lane) & 1ull) queue[qn + __popcll(m & gt_mask)] = bstart + lane;
                qn += __popcll(m);
                --batch;
            }
            cursor = batch >= 0 ? range_start : range_start - 1;             // "list exhausted" test below
        }
        while (!MASKS && qn < BUCKET && cursor >= range_start) {
            const int idx = cursor - lane;
            bool keep = false;
            if (idx >= range_start) {
                const int g = a.flatten_ids[idx];
                const float4 r0 = a.splats[(size_t)g * 4], r1 = a.splats[(size_t)g * 4 + 1];
                keep = !dns_cull_rect(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, fx0, rxh, fy0, ryh);
            }
            const uint64_t m = dns_ballot(keep);
            if (keep) queue[qn + __popcll(m & lt_mask)] = idx;
            qn += __popcll(m);
            cursor -= DNS_WAVE;
        }
        __builtin_amdgcn_wave_barrier();
        const int take = min(qn, BUCKET);     // entries [0, take) are this bucket; lane l owns 2l (A) and 2l+1 (B)
        qn -= take;                           // what is left sits at [BUCKET, BUCKET + qn) until the next boundary
        const bool last = take == 0;          // nothing new: only drain what is still in the lanes
#if DNS_BWD_FOLD
        // take < BUCKET only when the list is exhausted: this is the unit's last bucket.  fold = number of independent arrays
        const int fold = DET ? 1 : (take > 0 && take <= BUCKET / 4) ? 4 : (take > 0 && take <= BUCKET / 2) ? 2 : 1;
        const int fold_lanes = DNS_WAVE / fold;                 // lanes per array
        // after a folded bucket every lane is finished when its steps end: the closing pass only flushes (all four groups)
        const bool flush_only = last && folded;
        if (flush_only) prev_take = BUCKET;
#else
        constexpr int fold = 1;
        constexpr bool flush_only = false;
#endif

#pragma nounroll
        for (int grp = 0; grp < NGROUP; ++grp) {
            if (last && 2 * GROUP * grp >= prev_take) break;          // the remaining groups hold no splats
            // ==== group switch: GROUP lanes finish their old splats together and take new ones ====================
            // Lane l meets pixel p of a bucket at step (bucket start) + l + p, so lanes reach the end of a bucket one
            // step apart.  GROUP - 1 idle slots between the buckets of the pixel stream let GROUP neighbouring lanes
            // change splats at the same (wave-uniform) step: the first lane of the group is about to start pixel 0,
            // lane i of it is i slots before pixel 0.
            const bool mine = (lane / GROUP) == grp;
            const uint64_t gmask = (GROUP == 64 ? ~0ull : ((1ull << GROUP) - 1ull)) << (GROUP * grp);
            // -- flush: transpose through LDS, one atomic row per touched splat (A rows, then B rows): the group's
            //    lanes park their 16 partial sums, then the whole wave adds those records to global memory, each
            //    atomic instruction covering 4 complete 64-byte records.
#if DNS_BWD_MERGED_FLUSH
            {
                static_assert(DNS_BWD_FLUSH_ALL, "the merged flush writes every filled slot");
                const uint64_t tm = dns_ballot(cmp_a != 0x7fffffff || cmp_b != 0x7fffffff) & gmask;
                if (tm != 0) {                                                   // wave-uniform
                    if (mine) {
                        const int r = lane % GROUP;
                        // g_o / opacity through v_rcp_f32: 1 ulp on a sum whose order the atomics do not fix anyway
                        const f2 inv_o = {__builtin_amdgcn_rcpf(opac.x), __builtin_amdgcn_rcpf(opac.y)};
                        f2 *row = &flush2[r][0];
                        row[0] = XY_SCALE * g_x; row[1] = XY_SCALE * g_y; row[2] = 0.5f * g_ca; row[3] = g_cb;
                        row[4] = 0.5f * g_cc; row[5] = g_o * inv_o;
#pragma unroll
                        for (int k = 0; k < 8; ++k) row[REC_CH0 + k] = g_ch[k];
                        row[REC_ABSX] = -XY_SCALE * g_ax; row[REC_ABSX + 1] = -XY_SCALE * g_ay;
                        // DET: the row's own slot, keyed by the splat's list index (every (half tile, entry) is flushed exactly once)
                        float *base = DET ? a.det + (size_t)part * (size_t)a.det_cap * DNS_REC : a.v_splats;
                        ulonglong2 ad;
                        ad.x = cmp_a != 0x7fffffff ? (unsigned long long)(uintptr_t)(base + (size_t)(DET ? cmp_a : gid_a) * DNS_REC) : ~0ull;
                        ad.y = cmp_b != 0x7fffffff ? (unsigned long long)(uintptr_t)(base + (size_t)(DET ? cmp_b : gid_b) * DNS_REC) : ~0ull;
                        flush_addr[r] = ad;
                    }
                    __builtin_amdgcn_wave_barrier();
                    // byte offset of the lane's column as a 64-bit register pair, formed ONCE per flush (left alone, hipcc rebuilds the pair —
                    // a shift and a copy of a zero — in front of each of the eight atomics)
                    unsigned long long colb = (unsigned long long)(col * 4);
                    asm volatile("" : "+v"(colb));
#pragma unroll
                    for (int j = 0; j < GROUP / 4; ++j) {
                        const int rec = j * 4 + (lane >> 4);                     // record within the group: 16 lanes per record
                        if (((tm >> (GROUP * grp + j * 4)) & 0xfull) == 0) continue;   // wave-uniform: none of the round's four slots is filled
                        const ulonglong2 ad = flush_addr[rec];
                        const f2 val = flush2[rec][col];
                        // a device address has a clear top bit; -1 marks a slot that holds no splat
                        const bool has_a = (int)(ad.x >> 32) >= 0, has_b = (int)(ad.y >> 32) >= 0;
                        if (DET) {
                            typedef __attribute__((address_space(1))) float gfloat;
                            if (has_a) reinterpret_cast<gfloat *>((uintptr_t)ad.x)[col] = col_used ? val.x : 0.f;
                            if (has_b) reinterpret_cast<gfloat *>((uintptr_t)ad.y)[col] = col_used ? val.y : 0.f;
                        } else {
                            // the addresses come back from LDS as integers: say "global memory", or the atomics are issued as flat_atomic
                            typedef __attribute__((address_space(1))) float gfloat;
                            if (has_a && col_used) __builtin_amdgcn_global_atomic_fadd_f32(reinterpret_cast<gfloat *>((uintptr_t)(ad.x + colb)), val.x);
                            if (has_b && col_used) __builtin_amdgcn_global_atomic_fadd_f32(reinterpret_cast<gfloat *>((uintptr_t)(ad.y + colb)), val.y);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
#else
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#if DNS_BWD_FLUSH_ALL
                // every splat the group holds is flushed: with the forward's keep masks 99 % of them met a pixel, and finding the others
                // (an OR over the 16 partial sums per splat) cost more than their rows of zeros
                const uint64_t tmask = dns_ballot((half ? cmp_b : cmp_a) != 0x7fffffff) & gmask;
#elif DNS_BWD_TOUCH_FLAGS
                const uint64_t tmask = dns_ballot(half ? touched_b : touched_a) & gmask;
#else
                uint32_t any_bits = 0;      // OR of the 16 partial sums' bit patterns; << 1 drops the sign of a -0.0
#define BITS(v) (uint32_t)__float_as_int(half ? (v).y : (v).x)
                any_bits = BITS(g_x) | BITS(g_y) | BITS(g_ca) | BITS(g_cb) | BITS(g_cc) | BITS(g_o) | BITS(g_ax) | BITS(g_ay);
#pragma unroll
                for (int k = 0; k < 8; ++k) any_bits |= BITS(g_ch[k]);
#undef BITS
                const uint64_t tmask = dns_ballot((any_bits << 1) != 0u) & gmask;
#endif
                if (tmask == 0) continue;                                    // wave-uniform
                const int gid = half ? gid_b : gid_a;
#define SEL(v) (half ? (v).y : (v).x)
#pragma unroll
                for (int sub = 0; sub < GROUP / FLUSH_REC; ++sub) {
                    if (((tmask >> (GROUP * grp + FLUSH_REC * sub)) & ((1ull << FLUSH_REC) - 1ull)) == 0) continue;   // wave-uniform
                    if (mine && (lane % GROUP) / FLUSH_REC == sub) {
                        const int r = lane % FLUSH_REC;
                        flush[r][0] = make_float4(XY_SCALE * SEL(g_x), XY_SCALE * SEL(g_y), 0.5f * SEL(g_ca), SEL(g_cb));
                        // g_o / opacity through v_rcp_f32: 1 ulp on a sum whose order the atomics do not fix anyway, ten instructions less than an IEEE division
                        flush[r][1] = make_float4(0.5f * SEL(g_cc), SEL(g_o) * __builtin_amdgcn_rcpf(SEL(opac)), SEL(g_ch[0]), SEL(g_ch[1]));
                        flush[r][2] = make_float4(SEL(g_ch[2]), SEL(g_ch[3]), SEL(g_ch[4]), SEL(g_ch[5]));
                        flush[r][3] = make_float4(SEL(g_ch[6]), SEL(g_ch[7]), -XY_SCALE * SEL(g_ax), -XY_SCALE * SEL(g_ay));
                    }
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int j = 0; j < FLUSH_REC / 4; ++j) {
                        const int rec = j * 4 + (lane >> 4);                         // record within the round
                        const int src = GROUP * grp + FLUSH_REC * sub + rec;         // the lane that owns that splat
                        // DET: the row's own slot, keyed by the splat's list index (every (half tile, entry) is flushed exactly once)
                        const int rgid = __shfl(DET ? (half ? cmp_b : cmp_a) : gid, src, DNS_WAVE);
                        const float val = fl[j * 64 + lane];
                        if (DET) {
                            if ((tmask >> src) & 1) a.det[((size_t)part * (size_t)a.det_cap + (size_t)rgid) * DNS_REC + col] = col_used ? val : 0.f;
                        } else if (((tmask >> src) & 1) && col_used) unsafeAtomicAdd(a.v_splats + (size_t)rgid * DNS_REC + col, val);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
#undef SEL
            }
#endif
            // -- the group's new splats, as packed pairs
            if (mine) {
#if DNS_BWD_FOLD
                const int ll = lane & (fold_lanes - 1);          // position in the lane's array
#else
                const int ll = lane;
#endif
                const int idx_a = 2 * ll < take ? queue[2 * ll] : -1;
                const int idx_b = 2 * ll + 1 < take ? queue[2 * ll + 1] : -1;
                float4 ra0 = make_float4(0.f, 0.f, 0.f, 0.f), ra1 = ra0, ra2 = ra0, ra3 = ra0;
                float4 rb0 = ra0, rb1 = ra0, rb2 = ra0, rb3 = ra0;
#if DNS_BWD_BATCHED_SWITCH
                // Both list entries, then both records, each pair of requests in flight together and no branch around them: written as
                // `if (idx >= 0) { gid = ids[idx]; rec = splats[gid]; }` twice, the group's lanes went through FOUR memory round trips in a
                // row per switch (id A, record A, id B, record B; from the ISA).  An empty slot re-reads the list's first entry — a real
                // record, so its arithmetic stays finite — and is kept out of every pair by its "no entry" index (cmp = 0x7fffffff is
                // above every pixel's last index, and such a slot is not flushed).
                gid_a = a.flatten_ids[max(idx_a, range_start)];
                gid_b = a.flatten_ids[max(idx_b, range_start)];
                {
                    const float4 *reca = a.splats + (size_t)gid_a * 4, *recb = a.splats + (size_t)gid_b * 4;
                    ra0 = reca[0]; ra1 = reca[1]; rb0 = recb[0]; rb1 = recb[1];
                    if (D > 2) { ra2 = reca[2]; rb2 = recb[2]; }
                    if (D > 6) { ra3 = reca[3]; rb3 = recb[3]; }
                }
#else
                gid_a = 0; gid_b = 0;
                if (idx_a >= 0) {
                    gid_a = a.flatten_ids[idx_a];
                    const float4 *rec = a.splats + (size_t)gid_a * 4;
                    ra0 = rec[0]; ra1 = rec[1];
                    if (D > 2) ra2 = rec[2];
                    if (D > 6) ra3 = rec[3];
                }
                if (idx_b >= 0) {
                    gid_b = a.flatten_ids[idx_b];
                    const float4 *rec = a.splats + (size_t)gid_b * 4;
                    rb0 = rec[0]; rb1 = rec[1];
                    if (D > 2) rb2 = rec[2];
                    if (D > 6) rb3 = rec[3];
                }
#endif
                sx = f2{ra0.x, rb0.x}; sy = f2{ra0.y, rb0.y};
                ca = f2{ra0.z, rb0.z}; cb = f2{ra0.w, rb0.w}; cc = f2{ra1.x, rb1.x}; opac = f2{ra1.y, rb1.y};
                const DnsConicE qa = dns_conic_e(ra0.z, ra0.w, ra1.x), qb = dns_conic_e(rb0.z, rb0.w, rb1.x);
                na = f2{qa.na, qb.na}; nb = f2{qa.nb, qb.nb}; nc = f2{qa.nc, qb.nc};
                ch[0] = f2{ra1.z, rb1.z}; ch[1] = f2{ra1.w, rb1.w}; ch[2] = f2{ra2.x, rb2.x}; ch[3] = f2{ra2.y, rb2.y};
                ch[4] = f2{ra2.z, rb2.z}; ch[5] = f2{ra2.w, rb2.w}; ch[6] = f2{ra3.x, rb3.x}; ch[7] = f2{ra3.y, rb3.y};
                cmp_a = idx_a >= 0 ? idx_a : 0x7fffffff;
                cmp_b = idx_b >= 0 ? idx_b : 0x7fffffff;
                g_x = zero2; g_y = zero2; g_ca = zero2; g_cb = zero2; g_cc = zero2; g_o = zero2; g_ax = zero2; g_ay = zero2;
#pragma unroll
                for (int k = 0; k < 8; ++k) g_ch[k] = zero2;
                touched_a = false; touched_b = false;
#if DNS_BWD_FOLD
                // pixel shares: array j starts 16 j (fold 4) / 32 j (fold 2) steps after array 0 and all end together
                const int arr = lane / fold_lanes;
#if DNS_BWD_DUMMY_ROW
                // windows that start at column 0 (see DNS_BWD_DUMMY_ROW): 48 | 48 | 16 | 16.  Array j starts 16 j steps after array 0 and
                // may only start at a pixel the PREVIOUS bucket has finished with: lane 63 of the full array works on pixel p at step
                // p - 80 of the new bucket, so p_first <= 16 j + 79 (95, 111, 127) — 64 | 48 | 16 | 0 would start array 2 at pixel
                // 112 in the very step lane 63 still holds it (found as a wrong gradient by tests/test_gpu_determinism.py)
                p_first = fold == 4 ? (arr == 0 ? 0 : arr == 1 ? 48 : arr == 2 ? 96 : 112) * NPIX / 128
                        : fold == 2 ? (arr == 0 ? 0 : 80) * NPIX / 128 : 0;
                p_count = fold == 4 ? (arr == 0 ? 48 : arr == 1 ? 48 : arr == 2 ? 16 : 16) * NPIX / 128
                        : fold == 2 ? (arr == 0 ? 80 : 48) * NPIX / 128 : NPIX;
#else
                p_first = fold == 4 ? (arr == 0 ? 0 : arr == 1 ? 56 : arr == 2 ? 96 : 120) * NPIX / 128
                        : fold == 2 ? (arr == 0 ? 0 : 80) * NPIX / 128 : 0;
                p_count = fold == 4 ? (arr == 0 ? 56 : arr == 1 ? 40 : arr == 2 ? 24 : 8) * NPIX / 128
                        : fold == 2 ? (arr == 0 ? 80 : 48) * NPIX / 128 : NPIX;
#endif
                p = p_first - (lane % GROUP);
#else
                p = -(lane % GROUP);
#endif
                px_cur = fx0 + (float)(p & 15);
                if (DNS_BWD_DX_CARRY && D < 8) dx_cur = sx - px_cur;
#if DNS_BWD_DUMMY_ROW
                row_run = pix_base + (uint32_t)(p * 48);
                qs = -16 * (lane % GROUP); qs_lim = 16 * p_count;
                fy_arr = fy0 + (float)(p_first >> 4);
#endif
#if DNS_BWD_COORD_TABLE
                { const float2 c = coord[p & (NPIX - 1)]; pxy = f2{c.x, c.y}; }
#endif
            }
            int nsteps = grp < NGROUP - 1 ? GROUP : PERIOD - GROUP * (NGROUP - 1);
#if DNS_BWD_FOLD
            // a folded bucket: every array ends (its start) + (its lanes - 1) + (its pixels) steps after the bucket's start
            if (grp == NGROUP - 1 && fold > 1) nsteps = (fold == 4 ? 15 + (DNS_BWD_DUMMY_ROW ? 64 : 56) * NPIX / 128 : 31 + 80 * NPIX / 128) - GROUP * (NGROUP - 1);
            if (grp == NGROUP - 1) folded = fold > 1;
            if (flush_only) continue;                                     // nothing left to stream
#endif
            if (last) {
                if (2 * GROUP * (grp + 1) >= prev_take) break;           // that was the last group with anything to flush
                nsteps = GROUP;
            }
#if DNS_BWD_SCALAR_LOOP
            nsteps = __builtin_amdgcn_readfirstlane(nsteps);
#endif

            // ==== the stream: nsteps steps, one pixel per lane per step ===========================================
            // The pixel row of the step is requested first and waited for only after the row-independent part
            // (pixel coordinates, exponents, exp2) has been issued, which covers the LDS latency.
            // Settle every LDS result the switch left pending (bpermutes of a skipped flush round, queue reads) HERE:
            // otherwise hipcc's wait insertion may find one of their registers overwritten in the loop and put an
            // s_waitcnt lgkmcnt(0) right behind the row loads of every step (tools/check_asm_hazards.py: "early wait").
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0), vmcnt / expcnt untouched
            // Two instantiations of the step loop.  alpha = min(0.999, opacity x vis) can only clamp for a splat whose
            // opacity exceeds 0.999 (vis <= 1 for a valid pair); while no lane holds such a splat — random initialisation,
            // most of training — the clamp, the "gradient only where not clamped" compare and one of the two selects per
            // splat drop out of the step (alpha and its gradient weight are then the same number).
            auto step_loop = [&](auto clamp_tag) {
            constexpr bool CLAMP = decltype(clamp_tag)::value;
            for (int s = 0; s < nsteps; ++s) {
#if DNS_BWD_DUMMY_ROW
                // the lane's position in its pixel window, times 16 (qs): "active" is one unsigned compare, and the pixel row of the window is
                // byte 1 of the counter, which v_cvt_f32_ubyte1 converts without a shift (the windows start at multiples of 16 pixels)
                const bool active = (unsigned)qs < (unsigned)qs_lim;
                v4f c0, c1, cst;
                const uint32_t row_addr = active ? row_run : pix_base + NPIX * 48;   // an idle slot reads (and writes back) the dummy row
                row_run += 48;                                         // LDS address of the row of the lane's pixel, carried along
                row_issue(row_addr, c0, c1, cst, qs);                  // the counter itself is the token: no copy of it
                [[maybe_unused]] const int pcur = qs >> 4;             // D == 8 only (no "x of the next pixel" slot): column = pcur & 15
#elif DNS_BWD_FOLD
                const bool active = (unsigned)(p - p_first) < (unsigned)p_count;
#else
                const bool active = (unsigned)p < (unsigned)NPIX;
#endif
#if DNS_BWD_DUMMY_ROW
#else
                int pcur = p & (NPIX - 1);
                v4f c0, c1, cst;
                const uint32_t row_addr = pix_base + pcur * 48;        // LDS byte address of the pixel's row (read now, state written back at the end)
                row_issue(row_addr, c0, c1, cst, pcur);
#endif
#if DNS_BWD_COORD_TABLE
                f2 pxy_next;
                coord_issue(coord_base + (((p + 1) & (NPIX - 1)) << 3), pxy_next, pcur);
                const float px = pxy.x, py = pxy.y;
#else
#if DNS_BWD_DUMMY_ROW
                const float px = (DNS_BWD_PX_SLOT && D < 8) ? px_cur : fx0 + (float)(pcur & 15);
                const float py = fy_arr + (float)(((uint32_t)qs >> 8) & 0xffu);
#else
                const float px = (DNS_BWD_PX_SLOT && D < 8) ? px_cur : fx0 + (float)(pcur & 15), py = fy0 + (float)(pcur >> 4);
#endif
#endif
                const f2 dx = (DNS_BWD_DX_CARRY && DNS_BWD_PX_SLOT && D < 8) ? dx_cur : sx - px, dy = sy - py;
                // same fused-multiply-add sequence as dns_exponent(), two splats at a time (v_pk_*_f32)
#if DNS_EXP_SYM
                const f2 hu = __builtin_elementwise_fma(nb, dy, na * dx);     // -log2e/2 d sigma / d dx
                const f2 hw = __builtin_elementwise_fma(nc, dy, nb * dx);     // -log2e/2 d sigma / d dy
                const f2 e = __builtin_elementwise_fma(dx, hu, dy * hw);
#else
                const f2 e = __builtin_elementwise_fma(dx, __builtin_elementwise_fma(na, dx, nb * dy), (nc * dy) * dy);
#endif
                f2 vis = {dns_exp2(e.x), dns_exp2(e.y)};
#if DNS_BWD_PREVALID
                // everything of the pair's validity that does not need the pixel's row — the lane is not in an idle slot, sigma >= 0,
                // alpha >= 1/255 — is decided while the row is on its way; behind the wait only "entry <= the pixel's last index" is left
                const f2 ov = opac * vis;
                const float al_a = CLAMP ? fminf((float)DNS_ALPHA_MAX, ov.x) : ov.x;
                const float al_b = CLAMP ? fminf((float)DNS_ALPHA_MAX, ov.y) : ov.y;
                const bool pre_a = active && e.x <= 0.f && al_a >= (float)DNS_ALPHA_MIN;
                const bool pre_b = active && e.y <= 0.f && al_b >= (float)DNS_ALPHA_MIN;
                f2 alpha_pre = {pre_a ? al_a : 0.f, pre_b ? al_b : 0.f};
                row_wait(c0, c1, cst, alpha_pre);
#else
#if DNS_BWD_COORD_TABLE
                row_wait(c0, c1, cst, pxy_next, vis);
                pxy = pxy_next;
#else
                row_wait(c0, c1, cst, vis);
#endif
                const f2 ov = opac * vis;
                const float al_a = CLAMP ? fminf((float)DNS_ALPHA_MAX, ov.x) : ov.x;
                const float al_b = CLAMP ? fminf((float)DNS_ALPHA_MAX, ov.y) : ov.y;
#endif
                // state arrives from the previous lane; lane 0 takes it from the pixel's LDS row
#if DNS_BWD_LDS_STATE
                // the pixel's state is what the row read of this step delivered: the lane before wrote it there at the end of its step
                float T = cst.y;
                float SA = cst.x, SB = cst.z;
#elif DNS_BWD_PAIR_STATE
                float T = dpp_wave_shr1(T_out, cst.y);
                cst.x = dpp_wave_shr1(SA_out, cst.x);
                cst.z = dpp_wave_shr1(SB_out, cst.z);
                float SA = cst.x, SB = cst.z;
#else
                float T = dpp_wave_shr1(T_out, cst.x);
                float SA = dpp_wave_shr1(SA_out, cst.y);
                float SB = dpp_wave_shr1(SB_out, cst.z);
#endif
                const int bin_final = __float_as_int(cst.w);
#if DNS_BWD_DX_CARRY
                (void)px;
#else
                if (DNS_BWD_PX_SLOT && D < 8) px_cur = c1.w;
#endif
#if DNS_BWD_PREVALID
                const bool valid_a = cmp_a <= bin_final && (COUNT || CLAMP ? pre_a : true);
                const bool valid_b = cmp_b <= bin_final && (COUNT || CLAMP ? pre_b : true);
#else
                // DNS_BWD_DUMMY_ROW: an idle slot has read bin_final = -1, below every list index
                const bool valid_a = (DNS_BWD_DUMMY_ROW || active) && cmp_a <= bin_final && e.x <= 0.f && al_a >= (float)DNS_ALPHA_MIN;
                const bool valid_b = (DNS_BWD_DUMMY_ROW || active) && cmp_b <= bin_final && e.y <= 0.f && al_b >= (float)DNS_ALPHA_MIN;
#endif
                if (COUNT) n_pairs += __popcll(dns_ballot(valid_a)) + __popcll(dns_ballot(valid_b));
                {   // straight-line: an idle step costs the same as a busy one, but no phi copies at a join
#if DNS_BWD_TOUCH_FLAGS
                    touched_a |= valid_a; touched_b |= valid_b;
#endif
                    // an invalid pair takes alpha = 0 (=> 1/(1-alpha) = 1, weight 0: state and sums unchanged) and m = 0
#if DNS_BWD_PREVALID
                    const f2 alpha = {valid_a ? alpha_pre.x : 0.f, valid_b ? alpha_pre.y : 0.f};
#else
                    const f2 alpha = {valid_a ? al_a : 0.f, valid_b ? al_b : 0.f};
#endif
                    // opacity x vis where the pair is valid and alpha is not clamped, else 0: the weight of d/d(sigma)
                    // and, divided by the opacity again at the flush, of d/d(opacity)
                    const f2 ovm = CLAMP ? f2{(valid_a && ov.x <= (float)DNS_ALPHA_MAX) ? ov.x : 0.f,
                                              (valid_b && ov.y <= (float)DNS_ALPHA_MAX) ? ov.y : 0.f}
                                         : alpha;
                    const f2 om = 1.f - alpha;
                    const f2 ra = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};
                    const float T1 = T * ra.x, T2 = T1 * ra.y;          // the pixel meets A, then B
                    const f2 Tv = {T1, T2};
                    const f2 fac = alpha * Tv;
                    // the pixel's cotangents as aligned register pairs; a channel is broadcast to both splats by
                    // selecting one half of its pair (op_sel), not by copying it
                    const f2 pp[4] = {__builtin_shufflevector(c0, c0, 0, 1), __builtin_shufflevector(c0, c0, 2, 3),
                                      __builtin_shufflevector(c1, c1, 0, 1), __builtin_shufflevector(c1, c1, 2, 3)};
                    f2 cva = zero2, cvb = zero2;
#pragma unroll
                    for (int k = 0; k < D; ++k) {
                        [[maybe_unused]] const f2 vk = (k & 1) ? __builtin_shufflevector(pp[k >> 1], pp[k >> 1], 1, 1)
                                                               : __builtin_shufflevector(pp[k >> 1], pp[k >> 1], 0, 0);
                        // one packed FMA with the cotangent broadcast by operand selection (hipcc would copy the
                        // broadcast pair into registers first, hence the inline instruction)
                        pk_fma_bcast(g_ch[k], fac, pp[k >> 1], k & 1);
#if DNS_BWD_PAIR_STATE
                        // the same broadcast for the two chains of channel sums: left to the compiler, one of the seven becomes a
                        // v_mov of the channel into the low half of a fresh pair
                        if (k < split) {
                            if (k == 0) cva = pk_mul_bcast(ch[k], pp[0], 0);
                            else pk_fma_bcast(cva, ch[k], pp[k >> 1], k & 1);
                        } else {
                            if (k == split) cvb = pk_mul_bcast(ch[k], pp[k >> 1], k & 1);
                            else pk_fma_bcast(cvb, ch[k], pp[k >> 1], k & 1);
                        }
#else
                        if (k < split) cva = __builtin_elementwise_fma(ch[k], vk, cva);
                        else cvb = __builtin_elementwise_fma(ch[k], vk, cvb);
#endif
                    }
                    const float SA1 = __builtin_fmaf(fac.x, cva.x, SA);
#if DNS_BWD_PAIR_STATE
                    cst.y = SA1;                                         // T has been consumed: (S_a, S_a after A) is a register pair
                    const f2 SAv = __builtin_shufflevector(cst, cst, 0, 1);
#else
                    const f2 SAv = {SA, SA1};
#endif
                    const f2 va_a = Tv * cva - ra * SAv;
                    SA = __builtin_fmaf(fac.y, cva.y, SA1);
                    f2 va = va_a;
                    if (SPLIT != D) {
                        const float SB1 = __builtin_fmaf(fac.x, cvb.x, SB);
#if DNS_BWD_PAIR_STATE
                        cst.w = SB1;                                     // bin_final has been consumed
                        const f2 SBv = __builtin_shufflevector(cst, cst, 2, 3);
#else
                        const f2 SBv = {SB, SB1};
#endif
#if DNS_BWD_VA_CHAIN
                        // the second group's share joins as two chained packed FMAs (written as a sum of two differences it was a
                        // multiply, an FMA and an add: one packed instruction more per step)
                        va = __builtin_elementwise_fma(Tv, cvb, va_a);
                        va = __builtin_elementwise_fma(-ra, SBv, va);
#else
                        va += Tv * cvb - ra * SBv;
#endif
                        SB = __builtin_fmaf(fac.y, cvb.y, SB1);
                    }
                    const f2 vs = -ovm * va, vs_a = -ovm * va_a;
                    const f2 hx = vs * dx, hy = vs * dy;
                    g_ca = __builtin_elementwise_fma(hx, dx, g_ca);      // x 1/2 at the flush
                    g_cb = __builtin_elementwise_fma(hx, dy, g_cb);
                    g_cc = __builtin_elementwise_fma(hy, dy, g_cc);
#if DNS_EXP_SYM
                    const f2 gx = vs_a * hu, gy = vs_a * hw;             // x -2 / log2e at the flush
#else
                    const f2 gx = vs_a * (ca * dx + cb * dy);
                    const f2 gy = vs_a * (cb * dx + cc * dy);
#endif
#if DNS_BWD_ABS_FMA && DNS_EXP_SYM
                    // sum and sum of magnitudes of vs_a x (hu, hw) WITHOUT forming the products as values: the plain sums are two packed
                    // FMAs, the magnitudes four v_fma_f32 with |.| on both factors (|a b| = |a| |b|; source modifiers are free).  With
                    // the products formed first hipcc issued them twice — a packed multiply for the |.| adds and a packed FMA for the
                    // sums: two packed instructions more per step.
                    (void)gx; (void)gy;
                    g_x = __builtin_elementwise_fma(vs_a, hu, g_x); g_y = __builtin_elementwise_fma(vs_a, hw, g_y);
                    g_ax.x = __builtin_fmaf(__builtin_fabsf(vs_a.x), __builtin_fabsf(hu.x), g_ax.x);
                    g_ax.y = __builtin_fmaf(__builtin_fabsf(vs_a.y), __builtin_fabsf(hu.y), g_ax.y);
                    g_ay.x = __builtin_fmaf(__builtin_fabsf(vs_a.x), __builtin_fabsf(hw.x), g_ay.x);
                    g_ay.y = __builtin_fmaf(__builtin_fabsf(vs_a.y), __builtin_fabsf(hw.y), g_ay.y);
#else
                    g_x += gx; g_y += gy;
                    // |.| as a source modifier of a plain add: cheaper than masking the sign bits and a packed add
                    g_ax.x += __builtin_fabsf(gx.x); g_ax.y += __builtin_fabsf(gx.y);
                    g_ay.x += __builtin_fabsf(gy.x); g_ay.y += __builtin_fabsf(gy.y);
#endif
                    g_o = __builtin_elementwise_fma(ovm, va, g_o);             // / opacity at the flush
                    T = T2;
                }
#if DNS_BWD_LDS_STATE
                {   // every lane hands its pixel on through the pixel's LDS row (the next lane reads the row anyway); lanes in an idle slot
                    // must not write (their slot aliases a live pixel's row).  exec is narrowed and restored by hand: left to hipcc the
                    // store sits behind a branch per step
                    // two stores of loose registers (S_a and T, then S_b) rather than one ds_write_b96: the three values end the step in
                    // the high halves of three different register pairs and a 96-bit operand would cost two copies
#if DNS_BWD_DUMMY_ROW
                    asm volatile("ds_write2_b32 %0, %1, %2 offset0:8 offset1:9\n\tds_write_b32 %0, %3 offset:40"
                                 : : "v"(row_addr), "v"(SA), "v"(T), "v"(SB) : "memory");
#else
                    const uint64_t act = dns_ballot(active);
                    uint64_t saved;
                    asm volatile("s_and_saveexec_b64 %0, %1\n\tds_write2_b32 %2, %3, %4 offset0:8 offset1:9\n\tds_write_b32 %2, %5 offset:40\n\t"
                                 "s_or_b64 exec, exec, %0"
                                 : "=&s"(saved) : "s"(act), "v"(row_addr), "v"(SA), "v"(T), "v"(SB) : "memory", "scc");
#endif
                }
#else
                T_out = T; SA_out = SA; SB_out = SB;
#endif
                // park the state of the pixel leaving the array for the next (nearer) bucket
#if DNS_BWD_LDS_STATE
#elif DNS_BWD_FOLD && DNS_BWD_PAIR_STATE
                if ((lane & 15) == 15 && active) {                        // to the next row / bucket; bin_final stays where it is
                    float *st = reinterpret_cast<float *>(&pix[pcur][2]);
                    typedef float v3f __attribute__((ext_vector_type(3)));
                    *reinterpret_cast<v3f *>(st) = v3f{SA, T, SB};
                }
#elif DNS_BWD_FOLD
                if ((lane & 15) == 15 && active) pix[pcur][2] = make_float4(T, SA, SB, cst.w);   // to the next row / bucket
#else
                if (lane == DNS_WAVE - 1 && active) pix[pcur][2] = make_float4(T, SA, SB, cst.w);
#endif
#if DNS_BWD_DX_CARRY
                // the NEXT step's dx straight from the row's "x of the next pixel" slot, in place of this step's (dead by now): the slot
                // need not be copied out of the row's registers before the next row load overwrites them (one v_mov per step less)
                if (DNS_BWD_PX_SLOT && D < 8) {
                    // sx - (high half of the row's last register pair, broadcast): hipcc only finds low-half broadcasts and would copy the slot first
                    const f2 tail = __builtin_shufflevector(c1, c1, 2, 3);
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(dx_cur) : "v"(sx), "v"(tail));
                }
#endif
#if DNS_BWD_DUMMY_ROW
                qs += 16;
#else
                ++p;
#endif
            }
            };
            if (COUNT) n_slots += (unsigned long long)nsteps * BUCKET;
#ifdef DNS_BWD_TIMELINE
            tl_steps += nsteps; if (grp == 0) tl_splats += take;
#endif
#if DNS_BWD_CLAMP_MODE == 1
            step_loop(std::integral_constant<bool, CLAMP_LOOP>{});
#else
            if (dns_ballot(opac.x > (float)DNS_ALPHA_MAX || opac.y > (float)DNS_ALPHA_MAX) != 0ull) step_loop(std::true_type{});
            else step_loop(std::false_type{});
#endif
        }
        if (last) break;
        prev_take = take;
    }
    if (COUNT && lane == 0 && a.counters) { atomicAdd(a.counters + 4, n_slots); atomicAdd(a.counters + 5, n_pairs); }
#ifdef DNS_BWD_TIMELINE
    if (dbg_tl && lane == 0) {
        dbg_tl[4 * blockIdx.x] = tl_t0;
        dbg_tl[4 * blockIdx.x + 1] = wall_clock64();
        // list depth (20 bits) | steps streamed (20 bits) | splats in the buckets (20 bits)
        dbg_tl[4 * blockIdx.x + 2] = (unsigned long long)(hi - range_start + 1) | (tl_steps << 20) | (tl_splats << 40);
        // where it ran: HW_ID (wave, simd, pipe, cu, sh, se) and the XCC
        dbg_tl[4 * blockIdx.x + 3] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);
    }
#endif
}

template <int D, int SPLIT, bool DN = false, bool COUNT = false, bool MASKS = false>
int launch_bwd(const BwdArgs &ba, hipStream_t stream)
{
    if constexpr (!COUNT) if (ba.det) {                  // deterministic scatter: always the clamping loop, never folded
        hipLaunchKernelGGL((raster_bwd_kernel<D, SPLIT, DN, false, MASKS, true, true>), dim3(ba.n_tiles * ba.n_cameras * PARTS), dim3(DNS_WAVE), 0, stream, ba);
        DNS_CHECK_LAUNCH();
        return DNSPLAT_OK;
    }
    hipLaunchKernelGGL((raster_bwd_kernel<D, SPLIT, DN, COUNT, MASKS, true>), dim3(ba.n_tiles * ba.n_cameras * PARTS), dim3(DNS_WAVE), 0, stream, ba);
    DNS_CHECK_LAUNCH();
    if constexpr (DN && !COUNT) if (ba.sat_flag) {       // its clamp-free twin; exactly one of the two does the work
        hipLaunchKernelGGL((raster_bwd_kernel<D, SPLIT, DN, COUNT, MASKS, false>), dim3(ba.n_tiles * ba.n_cameras * PARTS), dim3(DNS_WAVE), 0, stream, ba);
        DNS_CHECK_LAUNCH();
    }
    return DNSPLAT_OK;
}

template <int D>
int dispatch_split(const BwdArgs &ba, hipStream_t stream)
{
    if (ba.xy_split == D) return launch_bwd<D, D>(ba, stream);
    if (D == 7 && ba.xy_split == 4) return launch_bwd<D, (D == 7 ? 4 : D)>(ba, stream);
    return launch_bwd<D, -1>(ba, stream);
}

}  // namespace

extern "C" int dnsplat_raster_bwd(const dnsplat_raster_args *a, dnsplat_stream_t stream_)
{
    if (!a) return DNSPLAT_ERR_INVALID_ARG;
    if (a->tile_size != TILE) return DNSPLAT_ERR_UNSUPPORTED;
    if (a->D < 1 || a->D > DNSPLAT_MAX_CHANNELS) return DNSPLAT_ERR_UNSUPPORTED;
    if (a->width <= 0 || a->height <= 0) return DNSPLAT_ERR_INVALID_ARG;
    // splats / flatten_ids / v_splats are only touched for list entries: NULL is fine when every list is empty (N == 0)
    if (!a->tile_offsets || !a->alphas || !a->last_ids || (!a->v_render && !a->dn)) return DNSPLAT_ERR_INVALID_ARG;
    if (a->ed_channel >= a->D || (a->ed_channel >= 0 && !a->render)) return DNSPLAT_ERR_INVALID_ARG;
    if (a->xy_split < 0 || a->xy_split > a->D) return DNSPLAT_ERR_INVALID_ARG;
    BwdArgs ba;
    ba.width = a->width; ba.height = a->height;
    ba.tw = dns_tiles_w(a->width, TILE);
    ba.n_tiles = ba.tw * dns_tiles_h(a->height, TILE);
    ba.splats = reinterpret_cast<const float4 *>(a->splats);
    ba.flatten_ids = a->flatten_ids;
    ba.tile_offsets = a->tile_offsets;
    ba.tile_ends = a->tile_ends;
    ba.background = a->background;
    ba.ed_channel = a->ed_channel;
    ba.render = a->render; ba.alphas = a->alphas; ba.last_ids = a->last_ids;
    ba.v_render = a->v_render; ba.v_alphas = a->v_alphas;
    ba.xy_split = a->xy_split;
    ba.v_splats = a->v_splats;
    ba.det = a->det_partials; ba.det_cap = a->det_capacity;
    if (ba.det && (a->det_capacity <= 0 || a->pair_counters)) return DNSPLAT_ERR_INVALID_ARG;
    ba.bg_rgb = ba.dn_v_rgb = ba.dn_v_depth = ba.dn_v_normal = ba.dn_v_acc = nullptr;
    if (a->n_cameras < 0) return DNSPLAT_ERR_INVALID_ARG;
    ba.n_cameras = a->n_cameras > 1 ? a->n_cameras : 1;
    ba.counters = reinterpret_cast<unsigned long long *>(a->pair_counters);
    ba.keep_masks = reinterpret_cast<const unsigned long long *>(a->keep_masks);
    ba.keep_mask_stride = a->keep_mask_stride;
    // the fused pass only (the generic and the counting instantiations have no clamp-free twin and always clamp)
    ba.sat_flag = (a->dn && !a->pair_counters && !a->det_partials) ? a->saturation_flag : nullptr;
    if (ba.keep_masks && a->keep_mask_stride <= 0) return DNSPLAT_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (a->dn) {
        const dnsplat_dn_post *dn = a->dn;
        if (a->D != 7 || a->ed_channel != 3 || a->xy_split != 4 || !a->background) return DNSPLAT_ERR_UNSUPPORTED;
        if (!dn->background_rgb || !dn->v_rgb || !dn->v_depth || !dn->v_normal || !a->render) return DNSPLAT_ERR_INVALID_ARG;
        ba.bg_rgb = dn->background_rgb; ba.dn_v_rgb = dn->v_rgb; ba.dn_v_depth = dn->v_depth;
        ba.dn_v_normal = dn->v_normal; ba.dn_v_acc = dn->v_accumulation;
        if (ba.counters) return launch_bwd<7, 4, true, true>(ba, stream);
        if (ba.keep_masks) return launch_bwd<7, 4, true, false, true>(ba, stream);
        return launch_bwd<7, 4, true>(ba, stream);
    }
    switch (a->D) {
        case 1: return dispatch_split<1>(ba, stream);
        case 2: return dispatch_split<2>(ba, stream);
        case 3: return dispatch_split<3>(ba, stream);
        case 4: return dispatch_split<4>(ba, stream);
        case 5: return dispatch_split<5>(ba, stream);
        case 6: return dispatch_split<6>(ba, stream);
        case 7: return dispatch_split<7>(ba, stream);
        case 8: return dispatch_split<8>(ba, stream);
    }
    return DNSPLAT_ERR_UNSUPPORTED;
}
