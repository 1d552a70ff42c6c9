// binning.hip — tile binning (stage 2 of include/dnsplat.h).
//
// Replaces gsplat 1.0.0 isect_tiles (count + emit), cub::DeviceRadixSort::SortPairs over 64-bit
// (tile | depth) keys, and isect_offset_encode; for the legacy
// normal pass also map_gaussian_to_intersects + torch.sort + get_tile_bin_edges (A8).
//
// Same result, different route.  The reference sorts I = n_isects 12-byte pairs on ~45 key bits
// (6 radix passes over I).  Here:
//   1. the N Gaussians are stably radix-sorted once by their 32 depth bits (4 passes over N, N << I);
//   2. (tile, gaussian) pairs are emitted in that depth order — one wave per 64 Gaussians, lanes
//      write a Gaussian's tiles cooperatively so stores are coalesced;
//   3. the pairs are stably radix-sorted on the tile id alone (ceil(log2(T)/8) = 2 passes over I),
//      with 16-bit tile keys; the last pass stores no keys and yields the tile offsets on the way.
// A stable sort by tile of a depth-ordered stream is exactly the (tile, depth, emission index)
// order the reference's stable 64-bit sort yields, so flatten_ids / tile offsets are bit-identical
// while the I-sized traffic drops from 6 passes x 12 B to 2 passes x 6 B (4 B in the last one).
//
// All ranking inside a radix pass is done with wave64 ballots (match-by-digit) and LDS counters:
// no atomics on the data path (the tile offsets are an atomicMin per (chunk, tile) run), fully
// deterministic.  Every kernel takes its element count from a
// device word, so the whole stage can be enqueued without a host round-trip on n_isects (the
// caller bounds it with `isect_capacity`).

#include "splat_common.h"

namespace {

constexpr int RS_THREADS = 256;                  // 4 waves per workgroup
#ifndef DNS_RS_ITEMS_I
#define DNS_RS_ITEMS_I 16
#endif
constexpr int RS_ITEMS_I = DNS_RS_ITEMS_I;       // keys per lane in the I-sized tile passes: 4096 per workgroup (paired A/B at C2 / C3:
                                                 // 8 -> +7 % / +2 %, 32 -> +29 % / +24 % of the emit + sort time)
#ifndef DNS_RS_ITEMS_N
#define DNS_RS_ITEMS_N 8
#endif
constexpr int RS_ITEMS_N = DNS_RS_ITEMS_N;                    // ... in the N-sized depth passes: 1 M keys are only 245 chunks of 4096, less than
                                                 // one workgroup per CU and a 16-round ranking chain each; 2048-key chunks fill the chip (measured best of 2/4/8/16)
constexpr int RS_WAVES = RS_THREADS / DNS_WAVE;
constexpr int RS_DIGITS = 256;

constexpr int SC_THREADS = 256;
constexpr int SC_ITEMS = 8;
constexpr int SC_CHUNK = SC_THREADS * SC_ITEMS;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & (DNS_WAVE - 1); }

// ------------------------------------------------------------------------------------------------
// wave / block scan helpers (wave64 DPP-free version via __shfl_up; these kernels are tiny)
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
#pragma unroll
    for (int off = 1; off < DNS_WAVE; off <<= 1) {
        uint32_t t = __shfl_up(v, off, DNS_WAVE);
        if ((int)lane_id() >= off) v += t;
    }
    return v;
}

// inclusive scan over a 256-thread block; returns inclusive value, total in `total`
__device__ __forceinline__ uint32_t block_incl_scan_256(uint32_t v, uint32_t *lds_wave /*[4]*/, uint32_t &total)
{
    const int w = threadIdx.x / DNS_WAVE;
    uint32_t inc = wave_incl_scan(v);
    if (lane_id() == DNS_WAVE - 1) lds_wave[w] = inc;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t s = lds_wave[i];
        if (i < w) base += s;
    }
    total = lds_wave[0] + lds_wave[1] + lds_wave[2] + lds_wave[3];
    __syncthreads();
    return inc + base;
}

// ------------------------------------------------------------------------------------------------
// 2. one LSD radix pass = histogram, per-digit scan, stable scatter
// K = key type: uint32_t for the depth keys, uint16_t for tile ids (any image up to 65536 tiles), which halves the
// key traffic of the two I-sized passes.
// Depth key of entry g: visible Gaussians have depth >= near_plane > 0, so the raw bits order like the floats; culled ones
// sort to the very end and emit nothing.  The first depth pass reads (radii, depths) directly — key = depth_key(g), value = g —
// instead of a key / value pair a separate kernel would have to write first.
__device__ __forceinline__ uint32_t depth_key(const int32_t *__restrict__ radii, const float *__restrict__ depths, uint32_t g)
{
    return radii[g] > 0 ? __float_as_uint(depths[g]) : 0xFFFFFFFFu;
}

// FIRST = first pass of the depth sort (keys synthesised from radii / depths, n given by value: n_ptr may be NULL)
template <typename K, int ITEMS, bool FIRST = false>
__global__ __launch_bounds__(RS_THREADS) void radix_hist_kernel(const K *__restrict__ keys,
                                                                const uint32_t *__restrict__ n_ptr, uint32_t n_cap,
                                                                int shift, uint32_t mask, uint32_t *__restrict__ table,
                                                                int nb, const int32_t *__restrict__ radii = nullptr,
                                                                const float *__restrict__ depths = nullptr,
                                                                int32_t *__restrict__ tile_first = nullptr, int n_tiles = 0)
{
    __shared__ uint32_t hist[RS_DIGITS];
    const uint32_t n = n_ptr ? min(*n_ptr, n_cap) : n_cap;
    // first pass of the tile sort: also presets the tile offsets to n (the last scatter pass lowers the non-empty tiles'
    // entries with atomicMin, tile_offsets_fill gives the empty ones the offset of the next non-empty tile)
    if (tile_first)
        for (int i = blockIdx.x * RS_THREADS + threadIdx.x; i <= n_tiles; i += gridDim.x * RS_THREADS) tile_first[i] = (int32_t)n;
    hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (RS_THREADS * ITEMS);
    if (base < n) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            uint32_t idx = base + i * RS_THREADS + threadIdx.x;
            if (idx < n) {
                const uint32_t k = FIRST ? depth_key(radii, depths, idx) : (uint32_t)keys[idx];
                atomicAdd(&hist[(k >> shift) & mask], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x <= mask) table[(size_t)threadIdx.x * nb + blockIdx.x] = hist[threadIdx.x];
}

// one workgroup per digit: exclusive scan of table[d][0..nb) in place, totals[d] = row sum.  Eight consecutive
// counters per thread and round: the tile passes scan ~5 k counters per digit, which is 3 rounds instead of 21.
__global__ __launch_bounds__(SC_THREADS) void radix_scan_kernel(uint32_t *__restrict__ table, int nb,
                                                                uint32_t *__restrict__ totals)
{
    __shared__ uint32_t lds_wave[4];
    uint32_t *row = table + (size_t)blockIdx.x * nb;
    uint32_t carry = 0;
    for (int start = 0; start < nb; start += SC_CHUNK) {
        const int i0 = start + threadIdx.x * SC_ITEMS;
        uint32_t v[SC_ITEMS];
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k) {
            v[k] = (i0 + k < nb) ? row[i0 + k] : 0u;
            s += v[k];
        }
        uint32_t tot;
        const uint32_t inc = block_incl_scan_256(s, lds_wave, tot);
        uint32_t run = carry + inc - s;
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k) {
            if (i0 + k < nb) row[i0 + k] = run;
            run += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// ------------------------------------------------------------------------------------------------
// Decoupled look-back (DNS_BIN_LOOKBACK, the I-sized tile passes).  The table-based pass needs, per radix pass, a histogram
// launch that reads every key again and a scan launch over (digits x chunks) counters before the scatter can place anything.
// With look-back a chunk publishes its own per-digit counts ("aggregate"), then walks back over the chunks before it,
// adding their aggregates until it meets one that already knows its inclusive prefix, and publishes its own inclusive prefix:
// the prefix over chunks is computed INSIDE the scatter launch.  What it needs from outside are only the GLOBAL digit totals
// of every pass, which one launch computes for all passes at once (tile_digit_totals_kernel).
//   * A chunk's number is a ticket drawn at workgroup start, so every chunk a workgroup waits for has started before it:
//     waiting never depends on a workgroup that is not resident yet, whatever order the dispatcher uses.
//   * A descriptor is ONE 64-bit word (2 status bits + count) written and read with agent-scope atomics — it is its own
//     payload, nothing else has to become visible with it (the per-XCD L2s are not coherent with each other).
//   * Every spin is bounded by the 100 MHz wall clock; on expiry the chunk records it in *fail and carries on with what it
//     has (wrong lists, but the launch always ends).
constexpr unsigned long long LB_AGG = 1ull << 62, LB_INC = 2ull << 62, LB_VAL = (1ull << 62) - 1ull;
constexpr unsigned long long LB_SPIN_TICKS = 20ull * 100000ull;      // 20 ms

__device__ __forceinline__ void lb_store(unsigned long long *p, unsigned long long v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long lb_load(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Global digit totals of up to three passes over the emitted tile keys (one read of the keys), and the zero state of the
// look-back descriptors and tickets of those passes.  totals: [3][256], tickets: [3], desc: [n_desc] words.
template <typename K, int ITEMS>
__global__ __launch_bounds__(RS_THREADS) void tile_digit_totals_kernel(const K *__restrict__ keys, const uint32_t *__restrict__ n_ptr,
                                                                       uint32_t n_cap, int passes, int shift1, int shift2,
                                                                       uint32_t mask0, uint32_t mask1, uint32_t mask2,
                                                                       uint32_t *__restrict__ totals,
                                                                       unsigned long long *__restrict__ desc, size_t n_desc,
                                                                       int32_t *__restrict__ tile_first, int n_tiles)
{
    __shared__ uint32_t hist[3][RS_DIGITS];
    const uint32_t n = min(*n_ptr, n_cap);
    for (size_t i = (size_t)blockIdx.x * RS_THREADS + threadIdx.x; i < n_desc; i += (size_t)gridDim.x * RS_THREADS) desc[i] = 0ull;
    // the tile offsets start at n (the last scatter pass lowers the non-empty tiles' entries with atomicMin)
    for (int i = blockIdx.x * RS_THREADS + threadIdx.x; i <= n_tiles; i += gridDim.x * RS_THREADS) tile_first[i] = (int32_t)n;
    hist[0][threadIdx.x] = 0; hist[1][threadIdx.x] = 0; hist[2][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (RS_THREADS * ITEMS);
    if (base < n) {
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = base + i * RS_THREADS + threadIdx.x;
            if (idx < n) {
                const uint32_t k = (uint32_t)keys[idx];
                atomicAdd(&hist[0][k & mask0], 1u);
                if (passes > 1) {
                    // consecutive entries share their high digits almost always: count once per run of equal digits in the wave
                    const uint32_t d1 = (k >> shift1) & mask1;
                    const uint32_t prev1 = __shfl_up(d1, 1, DNS_WAVE);
                    const bool head1 = lane_id() == 0 || prev1 != d1;
                    const uint64_t heads = dns_ballot(head1);
                    const int n_act = __popcll(dns_ballot(true));                           // the active lanes are lanes 0 .. n_act-1
                    if (head1) {
                        const uint64_t after = heads & ~((2ull << lane_id()) - 1ull);       // next run head above this lane
                        const int end = after ? __ffsll((unsigned long long)after) - 1 : n_act;
                        atomicAdd(&hist[1][d1], (uint32_t)(end - (int)lane_id()));
                    }
                }
                if (passes > 2) atomicAdd(&hist[2][(k >> shift2) & mask2], 1u);
            }
        }
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p) {
        const uint32_t c = hist[p][threadIdx.x];
        if (c) atomicAdd(&totals[p * RS_DIGITS + threadIdx.x], c);
    }
}

// DBITS = digit width of this pass (<= 8): the tile passes split their 13 bits 7 + 6 instead of 8 + 8 — fewer
// ballots per key and longer per-digit runs for the coalesced run stores.
//
// LAST = the final pass of the tile sort: the sorted keys themselves are not needed any more (no key store), but where
// a tile's entries start is.  Inside one digit's run of the LDS-sorted chunk the keys are non-decreasing (the stream
// was already sorted on the lower bits and the pass is stable), so "key differs from its left neighbour" marks the
// chunk-local first entry of a tile; the minimum of those positions over the chunks is the tile's offset.
// LOOKBACK: the count of the chunks before this one comes from the descriptors of those chunks (see above) instead of a
// precomputed table; `totals` are then the pass's global digit totals from tile_digit_totals_kernel.
template <typename K, bool LAST, int DBITS, int ITEMS, bool FIRST = false, bool LOOKBACK = false>
__global__ __launch_bounds__(RS_THREADS) void radix_scatter_kernel(
    const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, K *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ n_ptr, uint32_t n_cap, int shift,
    const uint32_t *__restrict__ table, const uint32_t *__restrict__ totals, int nb, int32_t *__restrict__ tile_first,
    const int32_t *__restrict__ radii = nullptr, const float *__restrict__ depths = nullptr,
    unsigned long long *__restrict__ desc = nullptr, uint32_t *__restrict__ ticket = nullptr, uint32_t *__restrict__ fail = nullptr)
{
    // The chunk is first sorted by digit INSIDE LDS (stable), then written out run by run: consecutive lanes
    // store to consecutive addresses of one digit's run, so the stores coalesce.  A direct scatter from the
    // ranking registers sends the 64 lanes of one store instruction to up to 64 different cache lines and
    // ran at a quarter of this version's rate on the 21 M-entry tile passes.
    __shared__ uint32_t wave_cnt[RS_WAVES][RS_DIGITS];
    __shared__ uint32_t wave_loc[RS_WAVES][RS_DIGITS];   // chunk-local position of a (wave, digit) run
    __shared__ uint32_t dstart[RS_DIGITS];               // chunk-local start of a digit's run
    __shared__ uint32_t gbase[RS_DIGITS];                // global start of this chunk's run of a digit
    constexpr int CHUNK = RS_THREADS * ITEMS;
    __shared__ K keys_s[CHUNK];
    __shared__ uint32_t vals_s[CHUNK];
    __shared__ uint32_t lds_wave[4];
    constexpr uint32_t DMASK = (1u << DBITS) - 1u;
    const uint32_t n = n_ptr ? min(*n_ptr, n_cap) : n_cap;
    uint32_t chunk = blockIdx.x;
    if (LOOKBACK) {
        __shared__ uint32_t ticket_s;
        if (threadIdx.x == 0) ticket_s = atomicAdd(ticket, 1u);
        __syncthreads();
        chunk = ticket_s;
    }
    const uint32_t base = chunk * CHUNK;
    if (base >= n) return;
    const uint32_t n_valid = min((uint32_t)CHUNK, n - base);
    const int w = threadIdx.x / DNS_WAVE;
    const uint32_t lane = lane_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // this thread's digit: its global total and the count of the chunks before this one — requested now, needed only
    // after the ranking (two loads that depend on nothing and otherwise sit exposed between two barriers)
    const bool is_digit = threadIdx.x <= DMASK;
    const uint32_t pre_tot = is_digit ? totals[threadIdx.x] : 0u;
    const uint32_t pre_tab = (is_digit && !LOOKBACK) ? table[(size_t)threadIdx.x * nb + blockIdx.x] : 0u;

#pragma unroll
    for (int i = 0; i < RS_WAVES; ++i) wave_cnt[i][threadIdx.x] = 0;
    __syncthreads();

    uint32_t key[ITEMS], val[ITEMS], rnk[ITEMS];
    const uint32_t wave_start = base + w * (DNS_WAVE * ITEMS);
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t idx = wave_start + r * DNS_WAVE + lane;
        const bool valid = idx < n;
        if (FIRST) {
            key[r] = valid ? depth_key(radii, depths, idx) : 0u;
            val[r] = idx;
        } else {
            key[r] = valid ? (uint32_t)keys_in[idx] : 0u;
            val[r] = valid ? vals_in[idx] : 0u;
        }
        const uint32_t d = (key[r] >> shift) & DMASK;
        // match-any by digit: DBITS ballots partition the wave into equal-digit lane sets
        uint64_t m = dns_ballot(valid);
#pragma unroll
        for (int bit = 0; bit < DBITS; ++bit) {
            const bool b = (d >> bit) & 1;
            const uint64_t bal = dns_ballot(b);
            m &= b ? bal : ~bal;
        }
        const uint32_t prior = wave_cnt[w][d];            // same address across the set -> LDS broadcast
        const uint32_t below = __popcll(m & lt_mask);
        rnk[r] = prior + below;
        if (valid && below == 0) wave_cnt[w][d] = prior + __popcll(m);  // set leader bumps the counter
    }
    __syncthreads();
    {
        // digit d = threadIdx.x
        const uint32_t c0 = wave_cnt[0][threadIdx.x], c1 = wave_cnt[1][threadIdx.x], c2 = wave_cnt[2][threadIdx.x],
                       c3 = wave_cnt[3][threadIdx.x];
        const uint32_t cnt = c0 + c1 + c2 + c3;
        uint32_t t2;
        const uint32_t linc = block_incl_scan_256(cnt, lds_wave, t2);     // chunk-local exclusive start
        const uint32_t ls = linc - cnt;
        dstart[threadIdx.x] = ls;
        wave_loc[0][threadIdx.x] = ls;
        wave_loc[1][threadIdx.x] = ls + c0;
        wave_loc[2][threadIdx.x] = ls + c0 + c1;
        wave_loc[3][threadIdx.x] = ls + c0 + c1 + c2;
        // global base = (#keys with smaller digit) + (#same digit in earlier chunks)
        const uint32_t tot = pre_tot;
        const uint32_t ginc = block_incl_scan_256(tot, lds_wave, t2);
        uint32_t before = pre_tab;
        if (LOOKBACK && is_digit) {
            constexpr int DIGITS = 1 << DBITS;
            unsigned long long *mine = desc + (size_t)chunk * DIGITS + threadIdx.x;
            if (chunk == 0) {
                lb_store(mine, LB_INC | cnt);
            } else {
                lb_store(mine, LB_AGG | cnt);
                unsigned long long sum = 0;
                const unsigned long long t_end = wall_clock64() + LB_SPIN_TICKS;
                for (int c = (int)chunk - 1; c >= 0; --c) {
                    const unsigned long long *theirs = desc + (size_t)c * DIGITS + threadIdx.x;
                    unsigned long long v = lb_load(theirs);
                    while ((v >> 62) == 0ull) {
                        if (wall_clock64() > t_end) { atomicOr(fail, 1u); v = LB_INC; break; }
                        __builtin_amdgcn_s_sleep(1);
                        v = lb_load(theirs);
                    }
                    sum += v & LB_VAL;
                    if ((v >> 62) == 2ull) break;
                }
                before = (uint32_t)sum;
                lb_store(mine, LB_INC | (sum + cnt));
            }
        }
        gbase[threadIdx.x] = is_digit ? (ginc - tot) + before : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t idx = wave_start + r * DNS_WAVE + lane;
        if (idx < n) {
            const uint32_t d = (key[r] >> shift) & DMASK;
            const uint32_t lpos = wave_loc[w][d] + rnk[r];
            keys_s[lpos] = (K)key[r];
            vals_s[lpos] = val[r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = r * RS_THREADS + threadIdx.x;
        if (i < n_valid) {
            const uint32_t k = keys_s[i];
            const uint32_t d = (k >> shift) & DMASK;
            const uint32_t dst = gbase[d] + (i - dstart[d]);
            if (!LAST) keys_out[dst] = (K)k;
            vals_out[dst] = vals_s[i];
            if (LAST && (i == 0 || (uint32_t)keys_s[i - 1] != k)) atomicMin(&tile_first[k], (int32_t)dst);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 3. inclusive scan of tiles_per_gauss gathered in depth order  -> cum[N], total
__global__ __launch_bounds__(SC_THREADS) void scan_sums_kernel(int N, const uint32_t *__restrict__ order,
                                                               const int32_t *__restrict__ tiles,
                                                               uint32_t *__restrict__ sums)
{
    __shared__ uint32_t lds_wave[4];
    const int base = blockIdx.x * SC_CHUNK + threadIdx.x * SC_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        int j = base + i;
        if (j < N) s += (uint32_t)tiles[order[j]];
    }
    uint32_t tot;
    block_incl_scan_256(s, lds_wave, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SC_THREADS) void scan_final_kernel(int N, const uint32_t *__restrict__ order,
                                                                const int32_t *__restrict__ tiles,
                                                                const uint32_t *__restrict__ sums,
                                                                uint32_t *__restrict__ cum, uint32_t *__restrict__ total_u32,
                                                                int64_t *__restrict__ total_i64)
{
    __shared__ uint32_t lds_wave[4];
    const int base = blockIdx.x * SC_CHUNK + threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        int j = base + i;
        v[i] = (j < N) ? (uint32_t)tiles[order[j]] : 0u;
        s += v[i];
    }
    // exclusive prefix of the chunk sums: every workgroup adds up the (few hundred) sums of the chunks before it itself,
    // which is cheaper than a separate single-workgroup scan launch between the two passes
    uint32_t before = 0;
    for (int c = threadIdx.x; c < (int)blockIdx.x; c += SC_THREADS) before += sums[c];
    uint32_t tot_before;
    block_incl_scan_256(before, lds_wave, tot_before);
    uint32_t tot;
    uint32_t inc = block_incl_scan_256(s, lds_wave, tot);
    uint32_t run = tot_before + inc - s;
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i) {
        int j = base + i;
        run += v[i];
        if (j < N) {
            cum[j] = run;
            if (j == N - 1) { *total_u32 = run; *total_i64 = (int64_t)run; }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 4. emission in depth order.  One wave owns 64 consecutive sorted Gaussians; those that hit tiles are taken two at
// a time, each half of the wave writing one Gaussian's (tile, gaussian) pairs side by side (row-major over its tile
// bbox, the reference's emission order).  The kernel is instruction-bound (a visible Gaussian covers ~30 tiles, less
// than a wave), hence two per round and a float reciprocal instead of the integer division for (row, column):
// floor((t + 0.5) / bw) is exact in fp32 for t < 2^16 tiles and bw <= 256 tile columns, far inside the 0.5 / bw margin.
template <typename K>
__global__ __launch_bounds__(256) void emit_kernel(int N, int n_per_cam, const uint32_t *__restrict__ order,
                                                   const uint32_t *__restrict__ cum,
                                                   const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                                                   int tile_size, int tw, int th, uint32_t cap,
                                                   K *__restrict__ tkeys, uint32_t *__restrict__ tvals,
                                                   uint32_t *__restrict__ lb_ctl = nullptr, int lb_ctl_words = 0,
                                                   const float4 *__restrict__ splats = nullptr)
{
    // the look-back control words (digit totals, tickets, failure flag) of the tile passes that follow start at zero
    if (lb_ctl && blockIdx.x == 0)
        for (int i = threadIdx.x; i < lb_ctl_words; i += blockDim.x) lb_ctl[i] = 0u;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lane = lane_id();
    uint32_t gid = 0, end = 0, start = 0;
    int x0 = 0, y0 = 0, bw = 1;
    if (j < N) {
        gid = order[j];
        end = cum[j];
        start = (j == 0) ? 0u : cum[j - 1];
        if (end > start) {
            int x1, y1;
            if (splats) {      // tight tile boxes (dnsplat_bin_args.tight_tiles): the box the projection kernel counted
                const float4 r0 = splats[(size_t)gid * 4], r1 = splats[(size_t)gid * 4 + 1];
                dns_snug_tile_bbox(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, (float)radii[gid], tile_size, tw, th, x0, y0, x1, y1);
            } else
                dns_tile_bbox(means2d[2 * gid], means2d[2 * gid + 1], (float)radii[gid], tile_size, tw, th, x0, y0, x1, y1);
            bw = x1 - x0;
            // batch of cameras: entry gid belongs to camera gid / n_per_cam, whose tile grid is stacked below the previous
            // cameras' (tile id = camera * tw * th + row * tw + column) — folded into the first tile row of the box
            if (n_per_cam < N) y0 += (int)(gid / (uint32_t)n_per_cam) * th;
        }
    }
    const bool exact = tw <= 256 && (N / n_per_cam) * tw * th <= 65536;      // the fp32 reciprocal route is exact
    uint64_t todo = dns_ballot(end > start);
    const uint32_t half = lane >> 5, hl = lane & 31;
    while (todo) {
        const int src0 = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        int src1 = -1;
        if (todo) { src1 = __ffsll((unsigned long long)todo) - 1; todo &= todo - 1; }
        const int src = half ? src1 : src0;
        const int from = src < 0 ? 0 : src;
        const uint32_t s_gid = __shfl(gid, from, DNS_WAVE);
        const uint32_t s_start = __shfl(start, from, DNS_WAVE);
        // every shuffle is executed by the whole wave: under a lane-dependent branch the lanes that skip it are inactive
        // SOURCES as well, and ds_bpermute returns 0 for them
        const uint32_t s_end = __shfl(end, from, DNS_WAVE);
        const uint32_t s_cnt = src < 0 ? 0u : s_end - s_start;
        const int s_x0 = __shfl(x0, from, DNS_WAVE), s_y0 = __shfl(y0, from, DNS_WAVE), s_bw = __shfl(bw, from, DNS_WAVE);
        const float inv_bw = 1.f / (float)s_bw;
        for (uint32_t t = hl; t < s_cnt; t += 32) {
            const uint32_t dst = s_start + t;
            if (dst < cap) {
                int row;
                if (exact) row = (int)(((float)t + 0.5f) * inv_bw);
                else row = (int)(t / (uint32_t)s_bw);
                const int colm = (int)t - row * s_bw;
                tkeys[dst] = (K)((s_y0 + row) * tw + s_x0 + colm);
                tvals[dst] = s_gid;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 5. tile offsets (gsplat isect_offset_encode), T+1 entries: offsets[t] = number of entries with tile id < t.
// The first histogram pass of the tile sort fills the array with n; the last scatter pass lowers the entries of the non-empty tiles to the
// position of their first entry (atomicMin); tile_offsets_fill gives every empty tile the offset of the next
// non-empty one (a suffix minimum), which is the same number.
constexpr int TO_THREADS = 1024;
__global__ __launch_bounds__(TO_THREADS) void tile_offsets_fill_kernel(int n_tiles, int32_t *__restrict__ offsets)
{
    __shared__ int32_t seg_min[TO_THREADS];
    const int total = n_tiles + 1;
    const int per = (total + TO_THREADS - 1) / TO_THREADS;
    const int lo = threadIdx.x * per, hi = min(lo + per, total);
    int32_t m = 0x7fffffff;
    for (int i = hi - 1; i >= lo; --i) m = min(m, offsets[i]);
    seg_min[threadIdx.x] = m;
    __syncthreads();
    // suffix minimum over the per-thread segments (Hillis-Steele, log2(1024) rounds)
    for (int off = 1; off < TO_THREADS; off <<= 1) {
        const int32_t other = threadIdx.x + off < TO_THREADS ? seg_min[threadIdx.x + off] : 0x7fffffff;
        __syncthreads();
        seg_min[threadIdx.x] = min(seg_min[threadIdx.x], other);
        __syncthreads();
    }
    int32_t run = threadIdx.x + 1 < TO_THREADS ? seg_min[threadIdx.x + 1] : 0x7fffffff;   // minimum of everything to the right
    for (int i = hi - 1; i >= lo; --i) {
        run = min(run, offsets[i]);
        offsets[i] = run;
    }
}

__global__ __launch_bounds__(256) void isect_ids_kernel(int n_tiles, int tile_bits, const int32_t *__restrict__ offsets,
                                                        const int32_t *__restrict__ flatten_ids,
                                                        const float *__restrict__ depths, int64_t *__restrict__ isect_ids,
                                                        int64_t cap)
{
    // one workgroup per (camera, tile); gsplat key = camera << (32 + tile_bits) | tile << 32 | depth bits
    const int t = blockIdx.x;
    const int64_t cam = t / n_tiles, local = t % n_tiles;
    const int64_t hi = (cam << (32 + tile_bits)) | (local << 32);
    const int s = offsets[t], e = offsets[t + 1];
    for (int i = s + threadIdx.x; i < e && i < cap; i += blockDim.x) {
        const uint32_t bits = __float_as_uint(depths[flatten_ids[i]]);
        isect_ids[i] = hi | (int64_t)bits;
    }
}

// ------------------------------------------------------------------------------------------------
constexpr int LB_CTL_WORDS = 3 * RS_DIGITS + 4;

// Tile sort by decoupled look-back (1) or with per-pass histogram + scan launches (0).  MEASURED AND REJECTED (paired A/B of
// dnsplat_bin_emit_sort, bit-identical lists, no wait ever hit its bound): C2 0.319 -> 0.475 ms, C5 0.82 -> 1.13 ms.  ~1000
// chunks run concurrently and all start together, so at the start of a pass the look-back of chunk c walks back through up
// to c descriptors one at a time, each an agent-scope (sc1) load of ~1 us across the non-coherent L2s: a serial chain that
// costs more than the two launches (histogram 22 us + scan 10 us) it removes.  The classic remedy — a whole wave inspecting
// 32-64 predecessors per step — does not fit a layout in which every lane owns a digit.  Kept compiled out as the record.
#ifndef DNS_BIN_LOOKBACK
#define DNS_BIN_LOOKBACK 0
#endif

struct BinWs {
    uint32_t *key_a, *key_b, *val_a, *val_b;  // [N]
    uint32_t *cum;                            // [N]
    uint32_t *tab_n;                          // [256 * nb_n]
    uint32_t *totals;                         // [256]
    uint32_t *sums;                           // [nb_scan]
    uint32_t *n_gauss;                        // [1] = N (device copy so the radix kernels are generic)
    uint32_t *total;                          // [1] n_isects as u32
    uint32_t *tkey_a, *tkey_b, *tval_a, *tval_b;  // [cap]
    uint32_t *tab_i;                          // [256 * nb_i]
    uint32_t *lb_ctl;                         // look-back control: [3][256] digit totals | [3] tickets | [1] failure flag
    unsigned long long *lb_desc;              // look-back descriptors, [3][nb_i][256] at most
    int nb_n, nb_i, nb_scan;
    size_t bytes;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

BinWs carve(void *ws, int N, int64_t cap)
{
    BinWs b{};
    b.nb_n = (N + RS_THREADS * RS_ITEMS_N - 1) / (RS_THREADS * RS_ITEMS_N);
    b.nb_i = (int)((cap + RS_THREADS * RS_ITEMS_I - 1) / (RS_THREADS * RS_ITEMS_I));
    b.nb_scan = (N + SC_CHUNK - 1) / SC_CHUNK;
    if (b.nb_n < 1) b.nb_n = 1;
    if (b.nb_i < 1) b.nb_i = 1;
    if (b.nb_scan < 1) b.nb_scan = 1;
    size_t off = 0;
    char *base = (char *)ws;
    auto take = [&](size_t elems) {
        uint32_t *p = (uint32_t *)(base + off);
        off += align_up(elems * sizeof(uint32_t), 256);
        return p;
    };
    size_t n = (size_t)(N > 0 ? N : 1), c = (size_t)(cap > 0 ? cap : 1);
    b.key_a = take(n); b.key_b = take(n); b.val_a = take(n); b.val_b = take(n);
    b.cum = take(n);
    b.tab_n = take((size_t)RS_DIGITS * b.nb_n);
    b.totals = take(RS_DIGITS);
    b.sums = take(b.nb_scan);
    b.n_gauss = take(1);
    b.total = take(1);
    b.tkey_a = take(c); b.tkey_b = take(c); b.tval_a = take(c); b.tval_b = take(c);
    b.tab_i = take((size_t)RS_DIGITS * b.nb_i);
    b.lb_ctl = take(LB_CTL_WORDS);
    b.lb_desc = reinterpret_cast<unsigned long long *>(take((size_t)3 * RS_DIGITS * b.nb_i * 2));
    b.bytes = off;
    return b;
}

int tile_bits(int n_tiles);

// one LSD pass over `dbits` bits at `shift`; tile_first != nullptr marks the last pass of the tile sort
template <typename K, int ITEMS>
void radix_pass(hipStream_t stream, const K *ka, const uint32_t *va, K *kb, uint32_t *vb, const uint32_t *n_ptr,
                uint32_t n_cap, int shift, int dbits, uint32_t *table, uint32_t *totals, int nb, int32_t *tile_first = nullptr,
                const int32_t *radii = nullptr, const float *depths = nullptr, int32_t *init_offsets = nullptr, int n_tiles = 0)
{
    const uint32_t mask = (1u << dbits) - 1u;
    if (radii) {   // first pass of the depth sort: 8-bit digit, keys synthesised from (radii, depths)
        hipLaunchKernelGGL((radix_hist_kernel<K, ITEMS, true>), dim3(nb), dim3(RS_THREADS), 0, stream, ka, n_ptr, n_cap, shift, mask, table, nb, radii, depths);
        hipLaunchKernelGGL(radix_scan_kernel, dim3(1 << dbits), dim3(SC_THREADS), 0, stream, table, nb, totals);
        hipLaunchKernelGGL((radix_scatter_kernel<K, false, 8, ITEMS, true>), dim3(nb), dim3(RS_THREADS), 0, stream, ka, va, kb, vb,
                           n_ptr, n_cap, shift, table, totals, nb, tile_first, radii, depths);
        return;
    }
    hipLaunchKernelGGL((radix_hist_kernel<K, ITEMS>), dim3(nb), dim3(RS_THREADS), 0, stream, ka, n_ptr, n_cap, shift, mask, table, nb,
                       (const int32_t *)nullptr, (const float *)nullptr, init_offsets, n_tiles);
    hipLaunchKernelGGL(radix_scan_kernel, dim3(1 << dbits), dim3(SC_THREADS), 0, stream, table, nb, totals);
#define DNS_SCATTER(B)                                                                                                      \
    do {                                                                                                                    \
        if (tile_first)                                                                                                     \
            hipLaunchKernelGGL((radix_scatter_kernel<K, true, B, ITEMS>), dim3(nb), dim3(RS_THREADS), 0, stream, ka, va, kb, vb,   \
                               n_ptr, n_cap, shift, table, totals, nb, tile_first);                                         \
        else                                                                                                                \
            hipLaunchKernelGGL((radix_scatter_kernel<K, false, B, ITEMS>), dim3(nb), dim3(RS_THREADS), 0, stream, ka, va, kb, vb,  \
                               n_ptr, n_cap, shift, table, totals, nb, tile_first);                                         \
    } while (0)
    switch (dbits) {
        case 1: DNS_SCATTER(1); break;
        case 2: DNS_SCATTER(2); break;
        case 3: DNS_SCATTER(3); break;
        case 4: DNS_SCATTER(4); break;
        case 5: DNS_SCATTER(5); break;
        case 6: DNS_SCATTER(6); break;
        case 7: DNS_SCATTER(7); break;
        default: DNS_SCATTER(8); break;
    }
#undef DNS_SCATTER
}

// emission + stable sort of the (tile, gaussian) pairs by tile id + tile offsets
// n_tiles = tiles of the whole batch (cameras x tiles per image)
template <typename K>
void emit_and_sort(hipStream_t stream, const dnsplat_bin_args *a, const BinWs &w, int tw, int th, int n_tiles, uint32_t cap)
{
    K *ka = reinterpret_cast<K *>(w.tkey_a), *kb = reinterpret_cast<K *>(w.tkey_b);
    uint32_t *va = w.tval_a, *vb = w.tval_b;
    const int n_cam = a->n_cameras > 1 ? a->n_cameras : 1;
    const int bits = tile_bits(n_tiles);
    const int passes = (bits + 7) / 8;
#if DNS_BIN_LOOKBACK
    {
        hipLaunchKernelGGL(emit_kernel<K>, dim3((a->N + 255) / 256), dim3(256), 0, stream, a->N, a->N / n_cam, w.val_a, w.cum,
                           a->means2d, a->radii, a->tile_size, tw, th, cap, ka, va, w.lb_ctl, LB_CTL_WORDS,
                           a->tight_tiles ? reinterpret_cast<const float4 *>(a->splats) : nullptr);
        int dbits[3] = {0, 0, 0}, shifts[3] = {0, 0, 0};
        size_t desc_off[4] = {0, 0, 0, 0};
        for (int pass = 0, sh = 0; pass < passes; ++pass) {
            dbits[pass] = (bits - sh + (passes - pass) - 1) / (passes - pass);       // 13 bits -> 7 + 6
            shifts[pass] = sh;
            sh += dbits[pass];
            desc_off[pass + 1] = desc_off[pass] + ((size_t)w.nb_i << dbits[pass]);
        }
        uint32_t *totals = w.lb_ctl, *tickets = w.lb_ctl + 3 * RS_DIGITS, *fail = tickets + 3;
        hipLaunchKernelGGL((tile_digit_totals_kernel<K, RS_ITEMS_I>), dim3(w.nb_i), dim3(RS_THREADS), 0, stream, ka, w.total, cap, passes,
                           shifts[1], shifts[2], (1u << dbits[0]) - 1u, (1u << dbits[1]) - 1u, (1u << dbits[2]) - 1u, totals,
                           w.lb_desc, desc_off[passes], a->tile_offsets, n_tiles);
        for (int pass = 0; pass < passes; ++pass) {
            const bool last = pass == passes - 1;
            uint32_t *vout = last ? (uint32_t *)a->flatten_ids : vb;
#define DNS_LB(B, L)                                                                                                              \
            hipLaunchKernelGGL((radix_scatter_kernel<K, L, B, RS_ITEMS_I, false, true>), dim3(w.nb_i), dim3(RS_THREADS), 0, stream, ka, va, \
                               kb, vout, w.total, cap, shifts[pass], (const uint32_t *)nullptr, totals + pass * RS_DIGITS, w.nb_i,  \
                               last ? a->tile_offsets : (int32_t *)nullptr, (const int32_t *)nullptr, (const float *)nullptr,      \
                               w.lb_desc + desc_off[pass], tickets + pass, fail)
#define DNS_LB2(B) do { if (last) DNS_LB(B, true); else DNS_LB(B, false); } while (0)
            switch (dbits[pass]) {
                case 1: DNS_LB2(1); break;
                case 2: DNS_LB2(2); break;
                case 3: DNS_LB2(3); break;
                case 4: DNS_LB2(4); break;
                case 5: DNS_LB2(5); break;
                case 6: DNS_LB2(6); break;
                case 7: DNS_LB2(7); break;
                default: DNS_LB2(8); break;
            }
#undef DNS_LB2
#undef DNS_LB
            K *t = ka; ka = kb; kb = t;
            uint32_t *u = va; va = vb; vb = u;
        }
        hipLaunchKernelGGL(tile_offsets_fill_kernel, dim3(1), dim3(TO_THREADS), 0, stream, n_tiles, a->tile_offsets);
        return;
    }
#endif
    hipLaunchKernelGGL(emit_kernel<K>, dim3((a->N + 255) / 256), dim3(256), 0, stream, a->N, a->N / n_cam, w.val_a, w.cum,
                       a->means2d, a->radii, a->tile_size, tw, th, cap, ka, va, w.lb_ctl, LB_CTL_WORDS,
                       a->tight_tiles ? reinterpret_cast<const float4 *>(a->splats) : nullptr);   // status word := 0
    int shift = 0;
    for (int pass = 0; pass < passes; ++pass) {
        const int dbits = (bits - shift + (passes - pass) - 1) / (passes - pass);   // 13 bits -> 7 + 6
        const bool last = pass == passes - 1;
        uint32_t *vout = last ? (uint32_t *)a->flatten_ids : vb;
        radix_pass<K, RS_ITEMS_I>(stream, ka, va, kb, vout, w.total, cap, shift, dbits, w.tab_i, w.totals, w.nb_i,
                      last ? a->tile_offsets : nullptr, nullptr, nullptr, pass == 0 ? a->tile_offsets : nullptr, n_tiles);
        shift += dbits;
        K *t = ka; ka = kb; kb = t;
        uint32_t *u = va; va = vb; vb = u;
    }
    hipLaunchKernelGGL(tile_offsets_fill_kernel, dim3(1), dim3(TO_THREADS), 0, stream, n_tiles, a->tile_offsets);
}

int tile_bits(int n_tiles)
{
    int bits = 1;
    while ((1 << bits) < n_tiles) ++bits;
    return bits;
}

}  // namespace

extern "C" size_t dnsplat_bin_workspace_bytes(int32_t N, int64_t isect_capacity, int32_t n_tiles)
{
    (void)n_tiles;
    if (N < 0 || isect_capacity < 0) return 0;
    return carve(nullptr, N, isect_capacity).bytes;
}

extern "C" size_t dnsplat_bin_status_offset(int32_t N, int64_t isect_capacity)
{
    if (N < 0 || isect_capacity < 0) return 0;
    const BinWs w = carve(nullptr, N, isect_capacity);
    return (size_t)((char *)(w.lb_ctl + 3 * RS_DIGITS + 3) - (char *)nullptr);
}

static int check_bin(const dnsplat_bin_args *a)
{
    if (!a) return DNSPLAT_ERR_INVALID_ARG;
    if (a->N < 0 || a->width <= 0 || a->height <= 0 || a->tile_size <= 0) return DNSPLAT_ERR_INVALID_ARG;
    if (a->n_cameras < 0 || (a->n_cameras > 1 && a->N % a->n_cameras != 0)) return DNSPLAT_ERR_INVALID_ARG;
    if (a->isect_capacity < 0 || a->isect_capacity > 0x7fffffffLL) return DNSPLAT_ERR_INVALID_ARG;
    if (!a->n_isects || !a->workspace) return DNSPLAT_ERR_INVALID_ARG;
    if (a->N > 0 && (!a->means2d || !a->radii || !a->depths || !a->tiles_per_gauss)) return DNSPLAT_ERR_INVALID_ARG;
    if (a->workspace_bytes < carve(nullptr, a->N, a->isect_capacity).bytes) return DNSPLAT_ERR_WORKSPACE;
    return DNSPLAT_OK;
}

extern "C" int dnsplat_bin_prepare(const dnsplat_bin_args *a, dnsplat_stream_t stream_)
{
    int rc = check_bin(a);
    if (rc != DNSPLAT_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    BinWs w = carve(a->workspace, a->N, a->isect_capacity);
    const int N = a->N;
    if (N == 0) {
        if (hipMemsetAsync(a->n_isects, 0, sizeof(int64_t), stream) != hipSuccess) return DNSPLAT_ERR_LAUNCH;
        if (hipMemsetAsync(w.total, 0, sizeof(uint32_t), stream) != hipSuccess) return DNSPLAT_ERR_LAUNCH;
    } else {
        const uint32_t n_u32 = (uint32_t)N;
        uint32_t *ka = w.key_a, *kb = w.key_b, *va = w.val_a, *vb = w.val_b;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 8 * pass;
            // the element count is known on the host here (n_ptr = NULL); pass 0 reads (radii, depths) instead of a key / value pair
            radix_pass<uint32_t, RS_ITEMS_N>(stream, ka, va, kb, vb, nullptr, n_u32, shift, 8, w.tab_n, w.totals, w.nb_n, nullptr,
                                             pass == 0 ? a->radii : nullptr, pass == 0 ? a->depths : nullptr);
            uint32_t *t = ka; ka = kb; kb = t;
            t = va; va = vb; vb = t;
        }
        // after 4 passes the sorted order is back in val_a
        hipLaunchKernelGGL(scan_sums_kernel, dim3(w.nb_scan), dim3(SC_THREADS), 0, stream, N, w.val_a, a->tiles_per_gauss,
                           w.sums);
        hipLaunchKernelGGL(scan_final_kernel, dim3(w.nb_scan), dim3(SC_THREADS), 0, stream, N, w.val_a,
                           a->tiles_per_gauss, w.sums, w.cum, w.total, a->n_isects);
        DNS_CHECK_LAUNCH();
    }
    if (a->n_isects_host) {
        if (hipMemcpyAsync(a->n_isects_host, a->n_isects, sizeof(int64_t), hipMemcpyDeviceToHost, stream) != hipSuccess)
            return DNSPLAT_ERR_LAUNCH;
    }
    return DNSPLAT_OK;
}

extern "C" int dnsplat_bin_emit_sort(const dnsplat_bin_args *a, dnsplat_stream_t stream_)
{
    int rc = check_bin(a);
    if (rc != DNSPLAT_OK) return rc;
    if (!a->tile_offsets || (a->isect_capacity > 0 && !a->flatten_ids)) return DNSPLAT_ERR_INVALID_ARG;
    if (a->tight_tiles && !a->splats && a->N > 0) return DNSPLAT_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    BinWs w = carve(a->workspace, a->N, a->isect_capacity);
    const int tw = dns_tiles_w(a->width, a->tile_size), th = dns_tiles_h(a->height, a->tile_size);
    const int64_t n_tiles64 = (int64_t)tw * th * (a->n_cameras > 1 ? a->n_cameras : 1);
    if (n_tiles64 > 0x7fffffffLL) return DNSPLAT_ERR_UNSUPPORTED;
    const int n_tiles = (int)n_tiles64;
    const uint32_t cap = (uint32_t)a->isect_capacity;
    if (a->N == 0 || cap == 0) {
        if (hipMemsetAsync(a->tile_offsets, 0, sizeof(int32_t) * (size_t)(n_tiles + 1), stream) != hipSuccess)
            return DNSPLAT_ERR_LAUNCH;
        return DNSPLAT_OK;
    }
    if (n_tiles <= 0x10000) emit_and_sort<uint16_t>(stream, a, w, tw, th, n_tiles, cap);
    else emit_and_sort<uint32_t>(stream, a, w, tw, th, n_tiles, cap);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_bin_isect_ids(int32_t n_tiles, int32_t n_cameras, const int32_t *tile_offsets,
                                     const int32_t *flatten_ids, const float *depths, int64_t *isect_ids, int64_t capacity,
                                     dnsplat_stream_t stream)
{
    if (n_tiles <= 0 || n_cameras < 0 || !tile_offsets || !isect_ids) return DNSPLAT_ERR_INVALID_ARG;
    if (capacity == 0) return DNSPLAT_OK;
    if (!flatten_ids || !depths) return DNSPLAT_ERR_INVALID_ARG;
    const int n_cam = n_cameras > 1 ? n_cameras : 1;
    int tb = 0;                                  // floor(log2(n_tiles)) + 1
    while ((n_tiles >> tb) != 0) ++tb;
    hipLaunchKernelGGL(isect_ids_kernel, dim3(n_tiles * n_cam), dim3(256), 0, (hipStream_t)stream, n_tiles, tb, tile_offsets,
                       flatten_ids, depths, isect_ids, capacity);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}
