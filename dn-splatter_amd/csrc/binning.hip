// binning.hip — tile binning (stage 2 of include/dnsplat.h).
//
// Replaces gsplat 1.0.0 isect_tiles (count + emit), cub::DeviceRadixSort::SortPairs over 64-bit
// (tile | depth) keys, and isect_offset_encode; for the legacy
// normal pass also map_gaussian_to_intersects + torch.sort + get_tile_bin_edges (A8).
//
// Same result, different route.  The reference sorts I = n_isects 12-byte pairs on ~45 key bits
// (6 radix passes over I).  Here:
//   1. the N Gaussians are stably radix-sorted once by their 32 depth bits (4 passes over N, N << I); the first pass
//      drops the culled ones (they emit nothing), the other three and the scans run over the visible Gaussians only;
//   2. the (tile, gaussian) pairs of that depth order are never written out as such: the FIRST pass of the tile
//      sort generates them on the fly — once to count its digits, once to place them (each workgroup owns 4096
//      consecutive pairs of the emission order and finds the Gaussians they belong to through a 16-byte record
//      per depth-ordered Gaussian and an LDS prefix maximum) — so the emission costs neither a launch nor the
//      12 B/pair round trip through HBM;
//   3. the pairs are stably radix-sorted on the tile id alone (ceil(log2(T)/8) = 2 passes over I),
//      with 16-bit tile keys; the last pass stores no keys and yields the tile offsets on the way.
// A stable sort by tile of a depth-ordered stream is exactly the (tile, depth, emission index)
// order the reference's stable 64-bit sort yields, so flatten_ids / tile offsets are bit-identical
// while the I-sized traffic drops from 6 passes x 12 B to one 6-byte store, one 6-byte load and a 4-byte store.
//
// All ranking inside a radix pass is done with wave64 ballots (match-by-digit) and LDS counters:
// no atomics on the data path (the tile offsets are an atomicMin per (chunk, tile) run), fully
// deterministic.  Every kernel takes its element count from a
// device word, so the whole stage can be enqueued without a host round-trip on n_isects (the
// caller bounds it with `isect_capacity`).

#include "splat_common.h"
#include "bin_ranges.h"

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
constexpr int RS_ITEMS_N = DNS_RS_ITEMS_N;       // ... in the N-sized depth passes: 1 M keys are only 245 chunks of 4096, less than
                                                 // one workgroup per CU and a 16-round ranking chain each; 2048-key chunks fill the chip (measured best of 2/4/8/16)
// ... and twice that from 4 M entries on: with 2048-key chunks a 5 M-Gaussian depth sort is 2400 chunks per pass — the chip is full
// either way, and 4096-key chunks halve the per-chunk tables the scans walk and double the scatter's run length (paired, round 4:
// binning 0.659 -> 0.634 ms at 5 M, +-0 at 3 M; 8 stays the better choice at 1 M)
constexpr int RS_ITEMS_N_LARGE = 2 * DNS_RS_ITEMS_N;
constexpr int RS_LARGE_N = 1 << 22;
inline int rs_items_n(int N) { return N >= RS_LARGE_N ? RS_ITEMS_N_LARGE : RS_ITEMS_N; }
// the I-sized tile passes: the same 4096-key chunk as RS_THREADS x RS_ITEMS_I, cut into TI_THREADS x TI_ITEMS.  With 512 threads
// the ranking chain of a wave (one LDS counter round trip per 64 keys) is 8 rounds long instead of 16 and a CU holds 24 instead
// of 16 waves of this latency-bound kernel
#ifndef DNS_TI_THREADS
#define DNS_TI_THREADS 512
#endif
constexpr int TI_THREADS = DNS_TI_THREADS;
constexpr int TI_ITEMS = RS_THREADS * RS_ITEMS_I / TI_THREADS;
static_assert(TI_THREADS * TI_ITEMS == RS_THREADS * RS_ITEMS_I && TI_THREADS % DNS_WAVE == 0 && TI_THREADS >= 256, "tile-pass chunk");
constexpr int RS_DIGITS = 256;
// Measured and removed (round 4): building the per-chunk digit counts of depth passes 1-3 with global atomics in the scatter kernel of
// the pass before (nine launches instead of twelve).  Every lane of such an atomic hits another cache line (256 next digits x the
// destination chunks), and scattered 4-byte atomics complete at ~12 G/s on this chip: 0.72 M of them cost 60 us per pass at C2 (a
// histogram kernel: 6 us), 3.6 M cost 290 us at C5 (13 us) — binning 0.29 -> 1.16 ms.  The compositing backward's atomics are fast
// because 16 neighbouring lanes share one 64-byte record; a histogram's do not.  LDS counters + one coalesced table row it stays.

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

// inclusive scan over a workgroup of NW waves; returns inclusive value, total in `total`
template <int NW = 4>
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t *lds_wave /*[NW]*/, uint32_t &total)
{
    const int w = threadIdx.x / DNS_WAVE;
    uint32_t inc = wave_incl_scan(v);
    if (lane_id() == DNS_WAVE - 1) lds_wave[w] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        uint32_t s = lds_wave[i];
        if (i < w) base += s;
        tot += s;
    }
    total = tot;
    __syncthreads();
    return inc + base;
}
__device__ __forceinline__ uint32_t block_incl_scan_256(uint32_t v, uint32_t *lds_wave /*[4]*/, uint32_t &total)
{
    return block_incl_scan<4>(v, lds_wave, total);
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

// Everything a workgroup needs to re-create "its" 4096 consecutive pairs of the emission order (GEN instantiations of the
// histogram and scatter kernels = the first pass of the tile sort).
struct __align__(16) EmitRec {
    uint32_t gid;        // entry index (camera * N + gaussian): the pair's value
    uint32_t start;      // position of the Gaussian's first pair in the emission order (= cum[j - 1])
    uint32_t base_tile;  // tile id of the first tile of its box (row-major over the stacked tile grids of the batch)
    uint32_t bw;         // box width in tiles
};
struct GenArgs {
    const EmitRec *__restrict__ jrec;        // [N] by depth rank j
    const uint32_t *__restrict__ cum;        // [N] inclusive pair counts by depth rank
    const uint32_t *__restrict__ chunk_first;   // [nb] depth rank of the Gaussian that holds pair chunk * CHUNK
    int N, tw;
    int exact;                               // floor((t + 0.5) / bw) through an fp32 reciprocal is exact (see gen_pair)
    const uint32_t *__restrict__ n_ranked;   // device word: how many depth ranks exist (the visible Gaussians); jrec / cum end there
};

constexpr int gen_pad(int i) { return i + (i >> 5); }      // one pad word per 32: a thread's 16 consecutive slots stay conflict-free

// owner[gen_pad(i)], i < CHUNK := 1 + (depth rank - j0) of the Gaussian that emits pair q0 + i.  Every Gaussian marks the slot of
// its first pair inside the chunk, a prefix maximum spreads the mark over its pairs.  Returns j0.
template <int ITEMS, int TH>
__device__ __forceinline__ uint32_t gen_owners(uint32_t *owner, uint32_t *lds_wave /*[TH / 64]*/, const GenArgs &g, uint32_t chunk,
                                               uint32_t q0, uint32_t n_valid)
{
    constexpr int CHUNK = TH * ITEMS;
    for (int i = threadIdx.x; i < gen_pad(CHUNK); i += TH) owner[i] = 0u;
    const uint32_t j0 = g.chunk_first[chunk];
    const uint32_t n_ranked = min(*g.n_ranked, (uint32_t)g.N);
    __syncthreads();
    for (uint32_t jb = j0;; jb += TH) {
        const uint32_t jj = jb + threadIdx.x;
        bool inside = false;       // this Gaussian starts before the end of the chunk (cum is non-decreasing: so do all before it)
        if (jj < n_ranked) {
            const uint32_t e = g.cum[jj], s = jj ? g.cum[jj - 1] : 0u;
            inside = s < q0 + n_valid;
            if (inside && e > s && e > q0) owner[gen_pad((int)(max(s, q0) - q0))] = jj - j0 + 1u;
        }
        // another round only if the last Gaussian of this one still started inside the chunk
        if (!__syncthreads_or(inside && threadIdx.x == TH - 1)) break;
    }
    // prefix maximum: thread t owns slots [t * ITEMS, (t + 1) * ITEMS)
    uint32_t loc[ITEMS];
    uint32_t m = 0u;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        m = max(m, owner[gen_pad(threadIdx.x * ITEMS + i)]);
        loc[i] = m;
    }
    uint32_t inc = m;
#pragma unroll
    for (int off = 1; off < DNS_WAVE; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off, DNS_WAVE);
        if ((int)lane_id() >= off) inc = max(inc, t);
    }
    const int w = threadIdx.x / DNS_WAVE;
    if (lane_id() == DNS_WAVE - 1) lds_wave[w] = inc;
    __syncthreads();
    uint32_t before = __shfl_up(inc, 1, DNS_WAVE);
    if (lane_id() == 0) before = 0u;
#pragma unroll
    for (int i = 0; i < TH / DNS_WAVE; ++i)
        if (i < w) before = max(before, lds_wave[i]);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) owner[gen_pad(threadIdx.x * ITEMS + i)] = max(loc[i], before);
    __syncthreads();
    return j0;
}

// pair q0 + i of the emission order: (tile id, entry).  The Gaussian's tiles are emitted row-major over its box, the
// reference's order; floor((t + 0.5) / bw) through an fp32 reciprocal is exact for t < 2^16 tiles and bw <= 256 tile columns
// (far inside the 0.5 / bw margin), otherwise the integer division is used.
__device__ __forceinline__ void gen_pair(const uint32_t *owner, const GenArgs &g, uint32_t j0, uint32_t q0, uint32_t i,
                                         uint32_t &key, uint32_t &val)
{
    const uint32_t o = owner[gen_pad((int)i)];
    const uint4 r = *reinterpret_cast<const uint4 *>(g.jrec + (j0 + o - 1u));
    const uint32_t t = q0 + i - r.y;
    uint32_t row;
    if (g.exact) row = (uint32_t)(((float)t + 0.5f) * (1.f / (float)r.w));
    else row = t / r.w;
    key = r.z + row * (uint32_t)g.tw + (t - row * r.w);
    val = r.x;
}

// gen_pair for ITEMS pairs at once, in three phases — all owner reads (LDS), all record loads, then the arithmetic — so that the
// record loads of a thread are in flight together (one by one, each was an LDS read, a wait, a 16-byte load and another wait).
// i[r] = pair index within the chunk (callers clamp out-of-range items to a valid one).
template <int ITEMS>
__device__ __forceinline__ void gen_pairs(const uint32_t *owner, const GenArgs &g, uint32_t j0, uint32_t q0, const uint32_t (&i)[ITEMS],
                                          uint32_t (&key)[ITEMS], uint32_t (&val)[ITEMS])
{
    uint32_t o[ITEMS];
    uint4 rec[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) o[r] = owner[gen_pad((int)i[r])];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) rec[r] = *reinterpret_cast<const uint4 *>(g.jrec + (j0 + o[r] - 1u));
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t t = q0 + i[r] - rec[r].y;
        uint32_t row;
        if (g.exact) row = (uint32_t)(((float)t + 0.5f) * (1.f / (float)rec[r].w));
        else row = t / rec[r].w;
        key[r] = rec[r].z + row * (uint32_t)g.tw + (t - row * rec[r].w);
        val[r] = rec[r].x;
    }
}

// FIRST = first pass of the depth sort (keys synthesised from radii / depths, n given by value: n_ptr may be NULL)
// GEN = first pass of the tile sort: the keys are the tile ids of the pairs the workgroup re-creates (see GenArgs)
// TH = threads per workgroup (chunk = TH x ITEMS keys): 256 x 8 for the N-sized depth passes, 512 x 8 for the tile passes
template <typename K, int ITEMS, bool FIRST = false, bool GEN = false, int TH = RS_THREADS>
__global__ __launch_bounds__(TH) void radix_hist_kernel(const K *__restrict__ keys,
                                                                const uint32_t *__restrict__ n_ptr, uint32_t n_cap,
                                                                int shift, uint32_t mask, uint32_t *__restrict__ table,
                                                                int nb, const int32_t *__restrict__ radii = nullptr,
                                                                const float *__restrict__ depths = nullptr,
                                                                int32_t *__restrict__ tile_first = nullptr, int n_tiles = 0,
                                                                int32_t *__restrict__ tile_end = nullptr, GenArgs gen = GenArgs{},
                                                                uint32_t *__restrict__ status = nullptr)
{
    __shared__ uint32_t hist[RS_DIGITS];
    __shared__ uint32_t owner[GEN ? gen_pad(TH * ITEMS) : 1];
    __shared__ uint32_t lds_wave[TH / DNS_WAVE];
    const uint32_t n = n_ptr ? min(*n_ptr, n_cap) : n_cap;
    // first pass of the tile sort: also presets the tile offsets to n (the last scatter pass lowers the non-empty tiles'
    // entries with atomicMin, tile_offsets_fill gives the empty ones the offset of the next non-empty tile) and the tile ends to 0
    if (tile_first)
        for (int i = blockIdx.x * TH + threadIdx.x; i <= n_tiles; i += gridDim.x * TH) tile_first[i] = (int32_t)n;
    if (tile_end)
        for (int i = blockIdx.x * TH + threadIdx.x; i < n_tiles; i += gridDim.x * TH) tile_end[i] = 0;
    if (status && blockIdx.x == 0 && threadIdx.x == 0) *status = 0u;
    if (threadIdx.x < RS_DIGITS) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t base = blockIdx.x * (TH * ITEMS);
    if (base < n) {
        uint32_t j0 = 0;
        if (GEN) j0 = gen_owners<ITEMS, TH>(owner, lds_wave, gen, blockIdx.x, base, min((uint32_t)(TH * ITEMS), n - base));
        // Two phases, and NO branch around the loads (out-of-range items re-read the last valid one): written as
        // `if (idx < n) { k = keys[idx]; atomicAdd(...) }` per item, every item became its own basic block with
        // load -> s_waitcnt vmcnt(0) -> ds_add, i.e. ITEMS memory round trips in a row per thread (round 4, from the ISA).
        uint32_t k[ITEMS];
        bool ok[ITEMS];
        if (GEN) {
            uint32_t ii[ITEMS], vv[ITEMS];
#pragma unroll
            for (int i = 0; i < ITEMS; ++i) ii[i] = min(base + i * TH + threadIdx.x, n - 1u) - base;
            gen_pairs<ITEMS>(owner, gen, j0, base, ii, k, vv);
        }
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint32_t idx = base + i * TH + threadIdx.x;
            const uint32_t idc = min(idx, n - 1u);
            ok[i] = idx < n;
            if (GEN) {}
            else if (FIRST) {
                // the first depth pass drops the culled Gaussians: they emit nothing, so nothing downstream needs their rank
                const int32_t r = radii[idc];
                k[i] = __float_as_uint(depths[idc]);
                ok[i] = ok[i] && r > 0;
            } else k[i] = (uint32_t)keys[idc];
        }
        // (measured and not kept, round 6: counting the lanes that share the first lane's digit with one ballot + ONE ds_add — the top byte of
        // the depth keys takes two or three values per frame, 64 same-address LDS atomics serialise — changed nothing: binning 0.275 vs 0.274 ms
        // at C2, 0.609 vs 0.610 at C5, gpurun_out/r06k)
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if (ok[i]) atomicAdd(&hist[(k[i] >> shift) & mask], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= mask) table[(size_t)threadIdx.x * nb + blockIdx.x] = hist[threadIdx.x];
}

// The GEN histogram without generating a single pair (round 4; bin_ranges.h): every record that has pairs inside the workgroup's
// chunk adds its box rows as cyclic digit RANGES to a difference array in LDS (two or three LDS atomics per box row instead of an
// owner search, a 16-byte record gather and an LDS atomic per pair), a prefix sum turns the differences into the counts.  One thread
// per record: the chunk's ~200 (C2) records are one round of coalesced 16-byte loads.  Same table as radix_hist_kernel<.., GEN>.
// DNS_GEN_HIST_RANGES = 0 keeps the pair-generating histogram (A/B).
#ifndef DNS_GEN_HIST_RANGES
#define DNS_GEN_HIST_RANGES 1
#endif
template <int ITEMS, int TH>
__global__ __launch_bounds__(TH) void radix_hist_ranges_kernel(const uint32_t *__restrict__ n_ptr, uint32_t n_cap, int dbits,
                                                               uint32_t *__restrict__ table, int nb, int32_t *__restrict__ tile_first,
                                                               int n_tiles, int32_t *__restrict__ tile_end, GenArgs gen,
                                                               uint32_t *__restrict__ status)
{
    __shared__ uint32_t diff[RS_DIGITS + 1];       // diff[ND] is a sink for ranges that end at the last digit
    __shared__ uint32_t all_s;                     // what every digit receives (rows longer than 2^dbits tiles)
    __shared__ uint32_t lds_wave[TH / DNS_WAVE];
    static_assert(TH > RS_DIGITS, "one thread per difference slot");
    const uint32_t n = n_ptr ? min(*n_ptr, n_cap) : n_cap;
    // the duties of the first histogram pass of the tile sort (see radix_hist_kernel): tile offsets preset to n, tile ends to 0
    if (tile_first)
        for (int i = blockIdx.x * TH + threadIdx.x; i <= n_tiles; i += gridDim.x * TH) tile_first[i] = (int32_t)n;
    if (tile_end)
        for (int i = blockIdx.x * TH + threadIdx.x; i < n_tiles; i += gridDim.x * TH) tile_end[i] = 0;
    if (status && blockIdx.x == 0 && threadIdx.x == 0) *status = 0u;
    const uint32_t ND = 1u << dbits;
    if (threadIdx.x <= ND) diff[threadIdx.x] = 0u;
    if (threadIdx.x == 0) all_s = 0u;
    __syncthreads();
    const uint32_t q0 = blockIdx.x * (uint32_t)(TH * ITEMS);
    if (q0 < n) {
        const uint32_t q1 = q0 + min((uint32_t)(TH * ITEMS), n - q0);
        const uint32_t j0 = gen.chunk_first[blockIdx.x];
        const uint32_t n_ranked = min(*gen.n_ranked, (uint32_t)gen.N);
        uint32_t all = 0u;
        for (uint32_t jb = j0;; jb += TH) {
            const uint32_t jj = jb + threadIdx.x;
            bool inside = false;       // this record starts before the end of the chunk (cum is non-decreasing: so do all before it)
            if (jj < n_ranked) {
                const uint32_t e = gen.cum[jj], s = jj ? gen.cum[jj - 1] : 0u;
                inside = s < q1;
                if (inside && e > s && e > q0) {        // records without pairs are never written (emit_prep_kernel): not read either
                    const uint4 r = *reinterpret_cast<const uint4 *>(gen.jrec + jj);
                    all += dns_record_digit_ranges(r.z, r.w, (uint32_t)gen.tw, max(s, q0) - s, min(e, q1) - s, dbits,
                                                   [&](uint32_t d0, uint32_t len) {
                                                       atomicAdd(&diff[d0], 1u);
                                                       const uint32_t end = d0 + len;
                                                       if (end <= ND) atomicAdd(&diff[end], 0xFFFFFFFFu);
                                                       else { atomicAdd(&diff[0], 1u); atomicAdd(&diff[end - ND], 0xFFFFFFFFu); }
                                                   });
                }
            }
            // another round only if the last record of this one still started inside the chunk
            if (!__syncthreads_or(inside && threadIdx.x == TH - 1)) break;
        }
        if (all) atomicAdd(&all_s, all);
    }
    __syncthreads();
    const uint32_t v = threadIdx.x < ND ? diff[threadIdx.x] : 0u;      // differences modulo 2^32: the prefix sums are the true counts
    uint32_t tot;
    const uint32_t inc = block_incl_scan<TH / DNS_WAVE>(v, lds_wave, tot);
    if (threadIdx.x < ND) table[(size_t)threadIdx.x * nb + blockIdx.x] = inc + all_s;
}

// Exclusive scan of every digit's row table[d][0..nb) in place, in SEGMENTS (round 6): workgroup (d, s) scans chunks [s x seg_len,
// (s + 1) x seg_len) of row d on its own and leaves the segment's sum in segsum[d x segs + s]; nobody waits for anybody — the scatter
// kernel adds the sums of the segments in front of its chunk itself (<= 7 words per digit thread, requested together with its table entry)
// and forms the digit's total from all of them.  One workgroup per digit walked the ~11 k chunk counters of a 46 M-pair tile pass in six
// sequential rounds with 128 workgroups on the chip: 13.6 us, twice per frame at C5.
constexpr int SC_MAX_SEGS = 8;
__global__ __launch_bounds__(SC_THREADS) void radix_scan_kernel(uint32_t *__restrict__ table, int nb, uint32_t *__restrict__ segsum,
                                                                int segs, int seg_len)
{
    __shared__ uint32_t lds_wave[4];
    const int d = blockIdx.x / segs, sgm = blockIdx.x - d * segs;
    uint32_t *row = table + (size_t)d * nb;
    const int lo = sgm * seg_len, hi = min(nb, lo + seg_len);
    uint32_t carry = 0;
    for (int start = lo; start < hi; start += SC_CHUNK) {
        const int i0 = start + threadIdx.x * SC_ITEMS;
        uint32_t v[SC_ITEMS];
        uint32_t s = 0;
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k) {
            v[k] = (i0 + k < hi) ? row[i0 + k] : 0u;
            s += v[k];
        }
        uint32_t tot;
        const uint32_t inc = block_incl_scan_256(s, lds_wave, tot);
        uint32_t run = carry + inc - s;
#pragma unroll
        for (int k = 0; k < SC_ITEMS; ++k) {
            if (i0 + k < hi) row[i0 + k] = run;
            run += v[k];
        }
        carry += tot;
    }
    if (threadIdx.x == 0) segsum[blockIdx.x] = carry;
}
// segments for a row of nb chunk counters: one per 2048 counters (one round of the scan), at most SC_MAX_SEGS
inline void scan_segments(int nb, int &segs, int &seg_len)
{
    segs = (nb + SC_CHUNK - 1) / SC_CHUNK;
    if (segs < 1) segs = 1;
    if (segs > SC_MAX_SEGS) segs = SC_MAX_SEGS;
    seg_len = (nb + segs - 1) / segs;
    if (seg_len < 1) seg_len = 1;
}

// DBITS = digit width of this pass (<= 8): the tile passes split their 13 bits 7 + 6 instead of 8 + 8 — fewer
// ballots per key and longer per-digit runs for the coalesced run stores.
//
// LAST = the final pass of the tile sort: the sorted keys themselves are not needed any more (no key store), but where
// a tile's entries start is.  Inside one digit's run of the LDS-sorted chunk the keys are non-decreasing (the stream
// was already sorted on the lower bits and the pass is stable), so "key differs from its left neighbour" marks the
// chunk-local first entry of a tile; the minimum of those positions over the chunks is the tile's offset.
// GEN: the first pass of the tile sort — keys / values are the pairs the workgroup re-creates (GenArgs), nothing is read.
// tile_end (LAST, optional): one past the last entry of every non-empty tile, so that a consumer that takes both arrays needs
// no suffix-minimum fill of the offsets of the empty tiles.
// DNS_TI_WAVES_PER_EU: register budget of the 512-thread tile-pass instantiations.  8 waves per SIMD = 64 VGPRs would admit a fourth
// workgroup per CU (hipcc's own choice: 72 VGPRs = three; the LDS tables, 30 KB since round 4, no longer stand in the way), but the
// nine values that no longer fit are spilled inside the ranking loop: measured +10 % on the binning stage at C2 and +7 % at C5.
// 1 = leave the choice to the compiler.
#ifndef DNS_TI_WAVES_PER_EU
#define DNS_TI_WAVES_PER_EU 1
#endif
#define DNS_TI_OCCUPANCY(th, K) __attribute__((amdgpu_waves_per_eu(((th) == 512 && sizeof(K) == 2) ? DNS_TI_WAVES_PER_EU : 1, 8)))
// BOXG (round 6: the LAST pass of the depth sort): every value is a Gaussian id whose final depth rank `dst` is known here — its tile
// box (dnsplat_proj_out.tile_boxes: one random 8-byte record) is gathered by this kernel and left behind in depth order together with
// the tile count, instead of by scan_sums_kernel<true> afterwards: the 3.6 M random line fetches of a 5 M-Gaussian frame (92 us as a
// kernel of their own) travel beside the scatter's own streams.
template <typename K, bool LAST, int DBITS, int ITEMS, bool FIRST = false, bool GEN = false, int TH = RS_THREADS, bool BOXG = false>
__global__ __launch_bounds__(TH) DNS_TI_OCCUPANCY(TH, K) void radix_scatter_kernel(
    const K *__restrict__ keys_in, const uint32_t *__restrict__ vals_in, K *__restrict__ keys_out,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ n_ptr, uint32_t n_cap, int shift,
    const uint32_t *__restrict__ table, const uint32_t *__restrict__ totals, int nb, int32_t *__restrict__ tile_first,
    const int32_t *__restrict__ radii = nullptr, const float *__restrict__ depths = nullptr,
    int32_t *__restrict__ tile_end = nullptr, GenArgs gen = GenArgs{}, uint32_t *__restrict__ count_out = nullptr,
    const int2 *__restrict__ box_in = nullptr, int2 *__restrict__ box_out = nullptr, uint32_t *__restrict__ tiles_out = nullptr,
    int segs = 1, int seg_len = 0x7fffffff)
{
    // The chunk is first sorted by digit INSIDE LDS (stable), then written out run by run: consecutive lanes
    // store to consecutive addresses of one digit's run, so the stores coalesce.  A direct scatter from the
    // ranking registers sends the 64 lanes of one store instruction to up to 64 different cache lines and
    // ran at a quarter of this version's rate on the 21 M-entry tile passes.
    constexpr int NW = TH / DNS_WAVE;
    constexpr int ND = 1 << DBITS;                       // digits of this pass: the LDS tables are sized for them, not for 256
    __shared__ uint32_t wave_cnt[NW][ND];
    // chunk-local position of a (wave, digit) run: written over the counts it is computed from (the thread of digit d reads its
    // column of counts into registers, then writes the positions) — together with the ND sizing 43 -> 30 KB of LDS per workgroup
    // of the tile passes (measured: +-0 at C2, -0.4 % at C5 — the register count, not LDS, holds these kernels at three workgroups per CU)
    uint32_t (*wave_loc)[ND] = wave_cnt;
    __shared__ uint32_t dstart[ND];                      // chunk-local start of a digit's run
    __shared__ uint32_t gbase[ND];                       // global start of this chunk's run of a digit
    constexpr int CHUNK = TH * ITEMS;
    __shared__ K keys_s[CHUNK];
    // GEN: the owner table (dead once the pairs sit in registers) shares the memory of the value staging area
    __shared__ uint32_t vals_s[GEN ? gen_pad(CHUNK) : CHUNK];
    __shared__ uint32_t lds_wave[NW];
    constexpr uint32_t DMASK = (1u << DBITS) - 1u;
    const uint32_t n = n_ptr ? min(*n_ptr, n_cap) : n_cap;
    const uint32_t chunk = blockIdx.x;
    const uint32_t base = chunk * CHUNK;
    if (base >= n) return;
    const uint32_t n_valid = min((uint32_t)CHUNK, n - base);
    const int w = threadIdx.x / DNS_WAVE;
    const uint32_t lane = lane_id();
    const uint64_t lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // this thread's digit: its global total and the count of the chunks before this one — requested now, needed only
    // after the ranking (two loads that depend on nothing and otherwise sit exposed between two barriers)
    const bool is_digit = threadIdx.x <= DMASK;
    // segmented scan (radix_scan_kernel): the digit's total is the sum of its segment sums, and the chunk's table entry is relative to its
    // segment — the sums of the segments in front of it are added here
    uint32_t pre_tot = 0u, pre_tab = is_digit ? table[(size_t)threadIdx.x * nb + blockIdx.x] : 0u;
    {
        const int my_seg = (int)blockIdx.x / seg_len;
        uint32_t sg[SC_MAX_SEGS];
#pragma unroll
        for (int q = 0; q < SC_MAX_SEGS; ++q) sg[q] = (is_digit && q < segs) ? totals[threadIdx.x * segs + q] : 0u;
#pragma unroll
        for (int q = 0; q < SC_MAX_SEGS; ++q) {
            pre_tot += sg[q];
            if (q < my_seg) pre_tab += sg[q];
        }
    }

    if (threadIdx.x < ND) {
#pragma unroll
        for (int i = 0; i < NW; ++i) wave_cnt[i][threadIdx.x] = 0;
    }
    uint32_t j0 = 0;
    if (GEN) j0 = gen_owners<ITEMS, TH>(vals_s, lds_wave, gen, chunk, base, n_valid);
    else __syncthreads();

    uint32_t key[ITEMS], val[ITEMS], rnk[ITEMS];
    uint32_t live = 0u;                                   // FIRST: bit r = item r is a visible Gaussian (the others are dropped here)
    const uint32_t wave_start = base + w * (DNS_WAVE * ITEMS);
    // every item's key / value is requested up front, without a branch around the loads (out-of-range items re-read the last
    // valid one): inside the ranking loop each load sat between two rounds of ballots and was waited for before the next was issued
    [[maybe_unused]] int32_t rad[ITEMS];
    if (GEN) {
        // all pairs first: the owner table is overwritten by the staging stores below
        uint32_t ii[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) ii[r] = min(wave_start + r * DNS_WAVE + lane, n - 1u) - base;
        gen_pairs<ITEMS>(vals_s, gen, j0, base, ii, key, val);
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t idx = wave_start + r * DNS_WAVE + lane;
        const uint32_t idc = min(idx, n - 1u);
        if (GEN) {
        } else if (FIRST) {
            rad[r] = radii[idc];
            key[r] = __float_as_uint(depths[idc]);
            val[r] = idx;
        } else {
            key[r] = (uint32_t)keys_in[idc];               // plain loads: non-temporal ones measured +7 % on this pass (2- and
            val[r] = vals_in[idc];                         // 4-byte accesses, and the histogram kernel has just read the same keys)
        }
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t idx = wave_start + r * DNS_WAVE + lane;
        bool valid = idx < n;
        if (FIRST) {
            valid = valid && rad[r] > 0;
            live |= (valid ? 1u : 0u) << r;
        }
        if (!valid) { key[r] = 0u; if (!FIRST) val[r] = 0u; }
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
    uint32_t n_out = n_valid;
    {
        // digit d = threadIdx.x (threads beyond the 256 digits only take part in the scans)
        const bool has_digit = threadIdx.x < ND;
        uint32_t cw[NW];
        uint32_t cnt = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            cw[i] = has_digit ? wave_cnt[i][threadIdx.x] : 0u;
            cnt += cw[i];
        }
        uint32_t t2;
        const uint32_t linc = block_incl_scan<NW>(cnt, lds_wave, t2);     // chunk-local exclusive start
        const uint32_t ls = linc - cnt;
        if (FIRST) n_out = t2;                                            // keys this chunk keeps (the same number in every thread)
        if (has_digit) {
            dstart[threadIdx.x] = ls;
            uint32_t run = ls;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                wave_loc[i][threadIdx.x] = run;
                run += cw[i];
            }
        }
        // global base = (#keys with smaller digit) + (#same digit in earlier chunks)
        const uint32_t tot = pre_tot;
        const uint32_t ginc = block_incl_scan<NW>(tot, lds_wave, t2);
        if (has_digit) gbase[threadIdx.x] = is_digit ? (ginc - tot) + pre_tab : 0u;
        // FIRST: the number of keys the sort goes on with (every later pass and the scans read it)
        if (FIRST && count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = t2;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t idx = wave_start + r * DNS_WAVE + lane;
        if (FIRST ? ((live >> r) & 1u) != 0u : idx < n) {
            const uint32_t d = (key[r] >> shift) & DMASK;
            const uint32_t lpos = wave_loc[w][d] + rnk[r];
            keys_s[lpos] = (K)key[r];
            vals_s[lpos] = val[r];
        }
    }
    __syncthreads();
    if constexpr (BOXG) {
        // all gathers of a thread in flight together, then the stores (keys are not needed after the last pass)
        uint32_t dsts[ITEMS], vs[ITEMS];
        int2 bx[ITEMS];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t i = min((uint32_t)(r * TH + threadIdx.x), n_out - 1u);
            const uint32_t d = ((uint32_t)keys_s[i] >> shift) & DMASK;
            dsts[r] = gbase[d] + (i - dstart[d]);
            vs[r] = vals_s[i];
        }
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) bx[r] = box_in[vs[r]];
#pragma unroll
        for (int r = 0; r < ITEMS; ++r)
            if ((uint32_t)(r * TH + threadIdx.x) < n_out) {
                vals_out[dsts[r]] = vs[r];
                box_out[dsts[r]] = bx[r];
                tiles_out[dsts[r]] = ((uint32_t)bx[r].y & 0xffffu) * ((uint32_t)bx[r].y >> 16);
            }
        return;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const uint32_t i = r * TH + threadIdx.x;
        if (i < n_out) {
            const uint32_t k = keys_s[i];
            const uint32_t d = (k >> shift) & DMASK;
            const uint32_t dst = gbase[d] + (i - dstart[d]);
            if (!LAST) keys_out[dst] = (K)k;
            vals_out[dst] = vals_s[i];
            if (LAST && (i == 0 || (uint32_t)keys_s[i - 1] != k)) atomicMin(&tile_first[k], (int32_t)dst);
            if (LAST && tile_end && (i + 1 == n_valid || (uint32_t)keys_s[i + 1] != k)) atomicMax(&tile_end[k], (int32_t)dst + 1);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 3. inclusive scan of tiles_per_gauss gathered in depth order  -> cum[n], total; n = the number of depth ranks (visible Gaussians),
// a device word the first depth pass leaves behind
// Also leaves the gathered counts in depth order (tiles_sorted): scan_final_kernel reads them coalesced instead of repeating the
// random gather through `order` (at 5 M Gaussians: 57 -> 18 us).  Gathering the tile boxes here as well, for emit_prep_kernel, was
// measured too: this kernel 54 -> 154 us for the 73 -> 19 us it saved there — a second random line per Gaussian costs the same
// wherever it is fetched; left where it was.
// BOXES (round 4): the projection's tile boxes carry the count as width x height (dnsplat_proj_out.tile_boxes, ABI 13), so ONE random
// 8-byte gather per Gaussian yields both the count and the box; the box is left behind in depth order as well (boxes_sorted) and
// emit_prep_kernel reads it coalesced instead of gathering it a second time through `order` (its random line per Gaussian was the
// whole cost of that kernel: 73 us at 5 M Gaussians).  Two separate arrays gathered here cost 2 x a gather; one record does not.
template <bool BOXES>
__global__ __launch_bounds__(SC_THREADS) void scan_sums_kernel(int N, const uint32_t *__restrict__ n_ptr,
                                                               const uint32_t *__restrict__ order,
                                                               const int32_t *__restrict__ tiles,
                                                               uint32_t *__restrict__ sums, uint32_t *__restrict__ tiles_sorted,
                                                               const int2 *__restrict__ boxes = nullptr,
                                                               int2 *__restrict__ boxes_sorted = nullptr)
{
    __shared__ uint32_t lds_wave[4];
    const int n = (int)min(*n_ptr, (uint32_t)N);
    const int base = blockIdx.x * SC_CHUNK + threadIdx.x * SC_ITEMS;
    uint32_t s = 0;
    if (blockIdx.x * SC_CHUNK < n) {
        // the eight (order -> tiles) gathers of a thread are issued together: first all ranks, then all counts (no branch around the
        // loads: a rank past the end re-reads the last valid one), instead of eight dependent round-trip pairs in a row
        uint32_t o[SC_ITEMS], t[SC_ITEMS];
        [[maybe_unused]] int2 bx[SC_ITEMS];
#pragma unroll
        for (int i = 0; i < SC_ITEMS; ++i) o[i] = order[min(base + i, n - 1)];
#pragma unroll
        for (int i = 0; i < SC_ITEMS; ++i) {
            if (BOXES) {
                bx[i] = boxes[o[i]];
                t[i] = ((uint32_t)bx[i].y & 0xffffu) * ((uint32_t)bx[i].y >> 16);
            } else t[i] = (uint32_t)tiles[o[i]];
        }
#pragma unroll
        for (int i = 0; i < SC_ITEMS; ++i)
            if (base + i < n) {
                s += t[i];
                tiles_sorted[base + i] = t[i];
                if (BOXES) boxes_sorted[base + i] = bx[i];
            }
    }
    uint32_t tot;
    block_incl_scan_256(s, lds_wave, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// scan_sums_kernel when the last depth pass has already left the counts in depth order (radix_scatter_kernel<..., BOXG>): a coalesced sum
__global__ __launch_bounds__(SC_THREADS) void scan_sums_sorted_kernel(int N, const uint32_t *__restrict__ n_ptr,
                                                                      const uint32_t *__restrict__ tiles_sorted, uint32_t *__restrict__ sums)
{
    __shared__ uint32_t lds_wave[4];
    const int n = (int)min(*n_ptr, (uint32_t)N);
    const int base = blockIdx.x * SC_CHUNK + threadIdx.x * SC_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SC_ITEMS; ++i)
        if (base + i < n) s += tiles_sorted[base + i];
    uint32_t tot;
    block_incl_scan_256(s, lds_wave, tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// What the pair generators of the tile sort's first pass need per Gaussian (EmitRec, by depth rank) and per 4096-pair chunk
// (the depth rank of the Gaussian that holds the chunk's first pair).
struct EmitPrep {
    EmitRec *__restrict__ jrec;
    uint32_t *__restrict__ chunk_first;
    int nb_chunks, chunk;                // chunk = pairs per workgroup of the tile passes
    int n_per_cam, tile_size, tw, th;
    const float *__restrict__ means2d;
    const int32_t *__restrict__ radii;
    const float4 *__restrict__ splats;   // tight tile boxes (dnsplat_bin_args.tight_tiles): the box the projection kernel counted
    const int2 *__restrict__ boxes;      // or NULL: (first tile id within the camera's grid, width | height << 16) straight from the projection
    const int2 *__restrict__ boxes_sorted;   // the same records in depth order (scan_sums_kernel<true> left them): read by rank, no gather
};

// depth rank j = entry gid owns the pairs [start, end) of the emission order, end > start
__device__ __forceinline__ void emit_record(const EmitPrep &ep, int N, uint32_t j, uint32_t gid, uint32_t start, uint32_t end)
{
    EmitRec r;
    r.gid = gid; r.start = start;
    // batch of cameras: entry gid belongs to camera gid / n_per_cam, whose tile grid is stacked below the previous
    // cameras' (tile id = camera * tw * th + row * tw + column)
    const uint32_t cam_tiles = (ep.n_per_cam < N) ? (gid / (uint32_t)ep.n_per_cam) * (uint32_t)(ep.tw * ep.th) : 0u;
    if (ep.boxes) {
        const int2 b = ep.boxes_sorted ? ep.boxes_sorted[j] : ep.boxes[gid];
        r.bw = (uint32_t)b.y & 0xffffu;
        r.base_tile = (uint32_t)b.x + cam_tiles;
    } else {
        int x0, y0, x1, y1;
        if (ep.splats) {
            const float4 r0 = ep.splats[(size_t)gid * 4], r1 = ep.splats[(size_t)gid * 4 + 1];
            dns_snug_tile_bbox(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, (float)ep.radii[gid], ep.tile_size, ep.tw, ep.th, x0, y0, x1, y1);
        } else
            dns_tile_bbox(ep.means2d[2 * gid], ep.means2d[2 * gid + 1], (float)ep.radii[gid], ep.tile_size, ep.tw, ep.th, x0, y0, x1, y1);
        r.base_tile = (uint32_t)(y0 * ep.tw + x0) + cam_tiles;
        r.bw = (uint32_t)(x1 - x0);
    }
    ep.jrec[j] = r;
    const uint32_t c_lo = (start + (uint32_t)ep.chunk - 1u) / (uint32_t)ep.chunk, c_hi = (end - 1u) / (uint32_t)ep.chunk;
    for (uint32_t c = c_lo; c <= c_hi && c < (uint32_t)ep.nb_chunks; ++c) ep.chunk_first[c] = j;
}

__global__ __launch_bounds__(SC_THREADS) void scan_final_kernel(int N, const uint32_t *__restrict__ n_ptr,
                                                                const uint32_t *__restrict__ order,
                                                                const int32_t *__restrict__ tiles,
                                                                const uint32_t *__restrict__ sums,
                                                                uint32_t *__restrict__ cum, uint32_t *__restrict__ total_u32,
                                                                int64_t *__restrict__ total_i64, int64_t *__restrict__ total_max)
{
    __shared__ uint32_t lds_wave[4];
    const int n = (int)min(*n_ptr, (uint32_t)N);
    if (n == 0 && blockIdx.x == 0 && threadIdx.x == 0) { *total_u32 = 0u; *total_i64 = 0; }
    if (blockIdx.x * SC_CHUNK >= n) return;
    const int base = blockIdx.x * SC_CHUNK + threadIdx.x * SC_ITEMS;
    uint32_t v[SC_ITEMS];
    uint32_t s = 0;
    {
        // the counts in depth order, as scan_sums_kernel left them (`tiles` here IS that array: no gather through `order`)
#pragma unroll
        for (int i = 0; i < SC_ITEMS; ++i) v[i] = (uint32_t)tiles[min(base + i, n - 1)];
#pragma unroll
        for (int i = 0; i < SC_ITEMS; ++i) {
            if (base + i >= n) v[i] = 0u;
            s += v[i];
        }
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
        if (j < n) {
            cum[j] = run;
            if (j == n - 1) {
                *total_u32 = run; *total_i64 = (int64_t)run;
                // sticky maximum over the frames since the caller last cleared it: lets a host that never waits for a single
                // frame's count (captured HIP graphs) still find out, later, whether any frame exceeded its capacity
                if (total_max) atomicMax(reinterpret_cast<unsigned long long *>(total_max), (unsigned long long)run);
            }
        }
    }
}

// one thread per depth rank j: the Gaussian's EmitRec and the heads of the pair chunks that start inside its pairs.  A kernel of its
// own: folded into scan_final_kernel (eight consecutive ranks per thread) it cost more than the launch it saved (paired: +2.5 % on
// dnsplat_bin_prepare at C2, +11 % at C5)
__global__ __launch_bounds__(256) void emit_prep_kernel(int N, const uint32_t *__restrict__ n_ptr, const uint32_t *__restrict__ order,
                                                        const uint32_t *__restrict__ cum, EmitPrep ep)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= min(*n_ptr, (uint32_t)N)) return;
    const uint32_t end = cum[j], start = j ? cum[j - 1] : 0u;
    if (end > start) emit_record(ep, N, j, order[j], start, end);
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
// chunks of the largest intersection count the 32-bit positions allow: the per-chunk head table is sized for it so that the
// N-sized front of the workspace (written by dnsplat_bin_prepare) does not depend on the capacity guess
constexpr int MAX_CHUNKS = (int)((0x80000000ull + RS_THREADS * RS_ITEMS_I - 1) / (RS_THREADS * RS_ITEMS_I));

struct BinWs {
    uint32_t *key_a, *key_b, *val_a, *val_b;  // [N]
    uint32_t *cum;                            // [N]
    uint32_t *tiles_sorted;                   // [N] tile counts in depth order
    int2 *boxes_sorted;                       // [N] tile boxes in depth order (only written when the caller passes tile_boxes)
    EmitRec *jrec;                            // [N]
    uint32_t *chunk_first;                    // [MAX_CHUNKS]
    uint32_t *tab_n;                          // [256 * nb_n]
    uint32_t *totals;                         // [256 x SC_MAX_SEGS] segment sums of the digit scans
    uint32_t *sums;                           // [nb_scan]
    uint32_t *total;                          // [1] n_isects as u32
    uint32_t *status;                         // [1] reserved status word (dnsplat_bin_status_offset): 0
    uint32_t *n_ranked;                       // [1] Gaussians the depth sort kept (radii > 0) = number of depth ranks
    uint32_t *tkey_b, *tval_b;                // [cap] pairs between the two tile passes
    uint32_t *tkey_c, *tval_c;                // [cap] only with three tile passes (> 65536 tiles)
    uint32_t *tab_i;                          // [256 * nb_i]
    int nb_n, nb_i, nb_scan;
    size_t bytes;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

BinWs carve(void *ws, int N, int64_t cap)
{
    BinWs b{};
    b.nb_n = (N + RS_THREADS * rs_items_n(N) - 1) / (RS_THREADS * rs_items_n(N));
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
    // N-sized front: same layout for every capacity
    b.key_a = take(n); b.key_b = take(n); b.val_a = take(n); b.val_b = take(n);
    b.cum = take(n);
    b.tiles_sorted = take(n);
    b.boxes_sorted = reinterpret_cast<int2 *>(take(2 * n));
    b.jrec = reinterpret_cast<EmitRec *>(take(4 * n));
    b.chunk_first = take(MAX_CHUNKS);
    b.tab_n = take((size_t)RS_DIGITS * b.nb_n);
    b.totals = take((size_t)RS_DIGITS * SC_MAX_SEGS);
    b.sums = take(b.nb_scan);
    b.total = take(1);
    b.status = take(1);
    b.n_ranked = take(1);
    // capacity-sized part
    b.tkey_b = take(c); b.tval_b = take(c);
    b.tkey_c = take(c); b.tval_c = take(c);
    b.tab_i = take((size_t)RS_DIGITS * b.nb_i);
    b.bytes = off;
    return b;
}

int tile_bits(int n_tiles);

// one LSD pass over `dbits` bits at `shift`; tile_first != nullptr marks the last pass of the tile sort;
// gen != nullptr: the first pass of the tile sort, whose input is generated (no ka / va)
template <typename K, int ITEMS, int TH = RS_THREADS>
void radix_pass(hipStream_t stream, const K *ka, const uint32_t *va, K *kb, uint32_t *vb, const uint32_t *n_ptr,
                uint32_t n_cap, int shift, int dbits, uint32_t *table, uint32_t *totals, int nb, int32_t *tile_first = nullptr,
                const int32_t *radii = nullptr, const float *depths = nullptr, int32_t *init_offsets = nullptr, int n_tiles = 0,
                int32_t *tile_end = nullptr, const GenArgs *gen = nullptr, uint32_t *status = nullptr, uint32_t *count_out = nullptr,
                const int2 *box_in = nullptr, int2 *box_out = nullptr, uint32_t *tiles_out = nullptr)
{
    const uint32_t mask = (1u << dbits) - 1u;
    int segs, seg_len;
    scan_segments(nb, segs, seg_len);
    if (radii) {   // first pass of the depth sort: 8-bit digit, keys synthesised from (radii, depths)
        hipLaunchKernelGGL((radix_hist_kernel<K, ITEMS, true, false, TH>), dim3(nb), dim3(TH), 0, stream, ka, n_ptr, n_cap, shift, mask, table, nb, radii, depths);
        hipLaunchKernelGGL(radix_scan_kernel, dim3((1 << dbits) * segs), dim3(SC_THREADS), 0, stream, table, nb, totals, segs, seg_len);
        hipLaunchKernelGGL((radix_scatter_kernel<K, false, 8, ITEMS, true, false, TH>), dim3(nb), dim3(TH), 0, stream, ka, va, kb, vb,
                           n_ptr, n_cap, shift, table, totals, nb, tile_first, radii, depths, (int32_t *)nullptr, GenArgs{}, count_out,
                           (const int2 *)nullptr, (int2 *)nullptr, (uint32_t *)nullptr, segs, seg_len);
        return;
    }
    const GenArgs g = gen ? *gen : GenArgs{};
    if (gen && DNS_GEN_HIST_RANGES && shift == 0 && TH > RS_DIGITS)
        hipLaunchKernelGGL((radix_hist_ranges_kernel<ITEMS, (TH > RS_DIGITS ? TH : 2 * RS_DIGITS)>), dim3(nb), dim3(TH), 0, stream, n_ptr, n_cap, dbits,
                           table, nb, init_offsets, n_tiles, init_offsets ? tile_end : nullptr, g, status);
    else if (gen)
        hipLaunchKernelGGL((radix_hist_kernel<K, ITEMS, false, true, TH>), dim3(nb), dim3(TH), 0, stream, ka, n_ptr, n_cap, shift, mask,
                           table, nb, (const int32_t *)nullptr, (const float *)nullptr, init_offsets, n_tiles, init_offsets ? tile_end : nullptr, g, status);
    else
        hipLaunchKernelGGL((radix_hist_kernel<K, ITEMS, false, false, TH>), dim3(nb), dim3(TH), 0, stream, ka, n_ptr, n_cap, shift, mask, table, nb,
                           (const int32_t *)nullptr, (const float *)nullptr, init_offsets, n_tiles, init_offsets ? tile_end : nullptr, g, status);
    hipLaunchKernelGGL(radix_scan_kernel, dim3((1 << dbits) * segs), dim3(SC_THREADS), 0, stream, table, nb, totals, segs, seg_len);
    if constexpr (sizeof(K) == 4) {
        if (box_in && !gen && !tile_first && dbits == 8) {      // last depth pass with the box gather folded in
            hipLaunchKernelGGL((radix_scatter_kernel<K, false, 8, ITEMS, false, false, TH, true>), dim3(nb), dim3(TH), 0, stream, ka, va, kb, vb,
                               n_ptr, n_cap, shift, table, totals, nb, tile_first, (const int32_t *)nullptr, (const float *)nullptr,
                               (int32_t *)nullptr, GenArgs{}, (uint32_t *)nullptr, box_in, box_out, tiles_out, segs, seg_len);
            return;
        }
    }
#define DNS_SCATTER3(B, L, G)                                                                                               \
    hipLaunchKernelGGL((radix_scatter_kernel<K, L, B, ITEMS, false, G, TH>), dim3(nb), dim3(TH), 0, stream, ka, va, kb, vb,    \
                       n_ptr, n_cap, shift, table, totals, nb, tile_first, (const int32_t *)nullptr, (const float *)nullptr, \
                       tile_end, g, (uint32_t *)nullptr, (const int2 *)nullptr, (int2 *)nullptr, (uint32_t *)nullptr, segs, seg_len)
#define DNS_SCATTER(B)                                                                                                      \
    do {                                                                                                                    \
        if (tile_first) { if (gen) DNS_SCATTER3(B, true, true); else DNS_SCATTER3(B, true, false); }                        \
        else { if (gen) DNS_SCATTER3(B, false, true); else DNS_SCATTER3(B, false, false); }                                 \
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
#undef DNS_SCATTER3
}

// stable sort of the (tile, gaussian) pairs of the emission order by tile id + tile offsets; the first pass generates the pairs
// n_tiles = tiles of the whole batch (cameras x tiles per image)
template <typename K>
void emit_and_sort(hipStream_t stream, const dnsplat_bin_args *a, const BinWs &w, int tw, int th, int n_tiles, uint32_t cap)
{
    K *kb = reinterpret_cast<K *>(w.tkey_b), *kc = reinterpret_cast<K *>(w.tkey_c);
    uint32_t *vb = w.tval_b, *vc = w.tval_c;
    const int n_cam = a->n_cameras > 1 ? a->n_cameras : 1;
    const int bits = tile_bits(n_tiles);
    const int passes = (bits + 7) / 8;
    GenArgs g;
    g.jrec = w.jrec; g.cum = w.cum; g.chunk_first = w.chunk_first;
    g.N = a->N; g.tw = tw; g.n_ranked = w.n_ranked;
    g.exact = (tw <= 256 && (int64_t)n_cam * tw * th <= 65536) ? 1 : 0;
    const K *ka = nullptr;
    const uint32_t *va = nullptr;
    int shift = 0;
    for (int pass = 0; pass < passes; ++pass) {
        const int dbits = (bits - shift + (passes - pass) - 1) / (passes - pass);   // 13 bits -> 7 + 6
        const bool last = pass == passes - 1;
        K *kout = (pass & 1) ? kc : kb;
        uint32_t *vout = last ? (uint32_t *)a->flatten_ids : ((pass & 1) ? vc : vb);
        radix_pass<K, TI_ITEMS, TI_THREADS>(stream, ka, va, kout, vout, w.total, cap, shift, dbits, w.tab_i, w.totals, w.nb_i,
                                  last ? a->tile_offsets : nullptr, nullptr, nullptr, pass == 0 ? a->tile_offsets : nullptr, n_tiles,
                                  a->tile_ends, pass == 0 ? &g : nullptr, pass == 0 ? w.status : nullptr);
        shift += dbits;
        ka = kout; va = vout;
    }
    // the offsets of the empty tiles (gsplat's isect_offsets): a consumer that takes tile_ends as well does not need them
    if (!(a->tile_ends && a->skip_offsets_fill))
        hipLaunchKernelGGL(tile_offsets_fill_kernel, dim3(1), dim3(TO_THREADS), 0, stream, n_tiles, a->tile_offsets);
}

int tile_bits(int n_tiles)
{
    int bits = 1;
    while ((1 << bits) < n_tiles) ++bits;
    return bits;
}

// ------------------------------------------------------------------------------------------------
// Deterministic gradient reduction (dnsplat_det_reduce): the sorted list is re-sorted by record id (stable, so a record's entries
// stay in list order = ascending tile id) and every record's rows are added up by ONE group of 16 lanes in that order, in double.
struct DetWs {
    uint32_t *key_a, *key_b, *val_a, *val_b;   // [cap]
    uint32_t *table;                           // [256 * nb]
    uint32_t *totals;                          // [256 x SC_MAX_SEGS]
    uint32_t *count;                           // [1]
    int nb;
    size_t bytes;
};

DetWs det_carve(void *ws, int64_t cap)
{
    DetWs d{};
    d.nb = (int)((cap + RS_THREADS * RS_ITEMS_N - 1) / (RS_THREADS * RS_ITEMS_N));
    if (d.nb < 1) d.nb = 1;
    size_t off = 0;
    char *base = (char *)ws;
    auto take = [&](size_t elems) {
        uint32_t *p = (uint32_t *)(base + off);
        off += align_up(elems * sizeof(uint32_t), 256);
        return p;
    };
    const size_t c = (size_t)(cap > 0 ? cap : 1);
    d.key_a = take(c); d.key_b = take(c); d.val_a = take(c); d.val_b = take(c);
    d.table = take((size_t)RS_DIGITS * d.nb);
    d.totals = take((size_t)RS_DIGITS * SC_MAX_SEGS);
    d.count = take(1);
    d.bytes = off;
    return d;
}

__global__ __launch_bounds__(256) void det_iota_kernel(const int64_t *__restrict__ n_isects, uint32_t cap, uint32_t *__restrict__ vals,
                                                       uint32_t *__restrict__ count)
{
    const uint32_t n = (uint32_t)min((int64_t)cap, max((int64_t)0, *n_isects));
    if (blockIdx.x == 0 && threadIdx.x == 0) *count = n;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) vals[i] = i;
}

// 16 lanes per entry of the record-sorted list (lane & 15 = column of the gradient record); the group whose entry is the first of
// its record adds up the record's rows: for each of its entries in list order, half 0 then half 1 — a fixed order, in double.
__global__ __launch_bounds__(256) void det_reduce_kernel(const uint32_t *__restrict__ count, const uint32_t *__restrict__ keys,
                                                         const uint32_t *__restrict__ vals, const float *__restrict__ partials,
                                                         size_t cap, int n_records, float *__restrict__ v_splats)
{
    const uint32_t n = *count;
    const uint32_t i = (uint32_t)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4);
    const int col = threadIdx.x & 15;
    if (i >= n) return;
    const uint32_t g = keys[i];
    if ((i > 0 && keys[i - 1] == g) || g >= (uint32_t)n_records) return;
    double acc = 0.0;
    for (uint32_t j = i; j < n && keys[j] == g; ++j) {
        const size_t e = vals[j];
        acc += (double)partials[e * DNS_REC + col];
        acc += (double)partials[(cap + e) * DNS_REC + col];
    }
    v_splats[(size_t)g * DNS_REC + col] = (float)acc;
}

}  // namespace

extern "C" size_t dnsplat_det_workspace_bytes(int64_t capacity)
{
    if (capacity < 0) return 0;
    return det_carve(nullptr, capacity).bytes;
}

extern "C" int dnsplat_det_reduce(const dnsplat_det_args *a, dnsplat_stream_t stream_)
{
    if (!a || a->n_records < 0 || a->capacity < 0 || a->capacity > 0x7fffffffLL) return DNSPLAT_ERR_INVALID_ARG;
    if (a->capacity == 0 || a->n_records == 0) return DNSPLAT_OK;
    if (!a->n_isects || !a->flatten_ids || !a->partials || !a->v_splats || !a->workspace) return DNSPLAT_ERR_INVALID_ARG;
    if (a->workspace_bytes < det_carve(nullptr, a->capacity).bytes) return DNSPLAT_ERR_WORKSPACE;
    hipStream_t stream = (hipStream_t)stream_;
    const DetWs w = det_carve(a->workspace, a->capacity);
    const uint32_t cap = (uint32_t)a->capacity;
    hipLaunchKernelGGL(det_iota_kernel, dim3(1024), dim3(256), 0, stream, a->n_isects, cap, w.val_a, w.count);
    int bits = 1;
    while ((1ll << bits) < (long long)a->n_records) ++bits;
    const int passes = (bits + 7) / 8;
    const uint32_t *ka = reinterpret_cast<const uint32_t *>(a->flatten_ids);
    uint32_t *va = w.val_a, *kb = w.key_a, *vb = w.val_b;
    for (int pass = 0; pass < passes; ++pass) {
        radix_pass<uint32_t, RS_ITEMS_N>(stream, ka, va, kb, vb, w.count, cap, 8 * pass, 8, w.table, w.totals, w.nb);
        ka = kb; va = vb;
        kb = (kb == w.key_a) ? w.key_b : w.key_a;
        vb = (vb == w.val_b) ? w.val_a : w.val_b;
    }
    const size_t groups = (size_t)cap;
    hipLaunchKernelGGL(det_reduce_kernel, dim3((unsigned)((groups * 16 + 255) / 256)), dim3(256), 0, stream, w.count, ka, va, a->partials,
                       (size_t)a->capacity, a->n_records, a->v_splats);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

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
    return (size_t)((char *)w.status - (char *)nullptr);
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
    if (a->tight_tiles && !a->splats && !a->tile_boxes && N > 0) return DNSPLAT_ERR_INVALID_ARG;
    if (N == 0) {
        if (hipMemsetAsync(a->n_isects, 0, sizeof(int64_t), stream) != hipSuccess) return DNSPLAT_ERR_LAUNCH;
        if (hipMemsetAsync(w.total, 0, sizeof(uint32_t), stream) != hipSuccess) return DNSPLAT_ERR_LAUNCH;
    } else {
        const uint32_t n_u32 = (uint32_t)N;
        uint32_t *ka = w.key_a, *kb = w.key_b, *va = w.val_a, *vb = w.val_b;
        const int2 *boxes = reinterpret_cast<const int2 *>(a->tile_boxes);
        // DNSPLAT_BIN_BOX_GATHER=0: the round-3 route (counts gathered here, boxes gathered again by emit_prep_kernel), for A/B runs
        static const bool box_gather = [] { const char *e = getenv("DNSPLAT_BIN_BOX_GATHER"); return !(e && e[0] == '0'); }();
        const bool sorted_boxes = boxes && box_gather;
        // DNSPLAT_BIN_BOX_FOLD=0: the box gather as scan_sums_kernel<true> behind the depth sort (round 4 / 5) instead of inside its last pass
        static const bool box_fold = [] { const char *e = getenv("DNSPLAT_BIN_BOX_FOLD"); return !(e && e[0] == '0'); }();
        const bool fold = sorted_boxes && box_fold;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 8 * pass;
            const bool fold_here = fold && pass == 3;
            // pass 0 knows its element count on the host (n_ptr = NULL), reads (radii, depths) instead of a key / value pair and drops
            // the culled Gaussians; it leaves the number it kept in w.n_ranked, which bounds every later pass and the scans
            if (rs_items_n(N) == RS_ITEMS_N_LARGE)
                radix_pass<uint32_t, RS_ITEMS_N_LARGE>(stream, ka, va, kb, vb, pass == 0 ? nullptr : w.n_ranked, n_u32, shift, 8, w.tab_n, w.totals,
                                                       w.nb_n, nullptr, pass == 0 ? a->radii : nullptr, pass == 0 ? a->depths : nullptr, nullptr, 0,
                                                       nullptr, nullptr, nullptr, pass == 0 ? w.n_ranked : nullptr,
                                                       fold_here ? boxes : nullptr, w.boxes_sorted, w.tiles_sorted);
            else
                radix_pass<uint32_t, RS_ITEMS_N>(stream, ka, va, kb, vb, pass == 0 ? nullptr : w.n_ranked, n_u32, shift, 8, w.tab_n, w.totals,
                                                 w.nb_n, nullptr, pass == 0 ? a->radii : nullptr, pass == 0 ? a->depths : nullptr, nullptr, 0,
                                                 nullptr, nullptr, nullptr, pass == 0 ? w.n_ranked : nullptr,
                                                 fold_here ? boxes : nullptr, w.boxes_sorted, w.tiles_sorted);
            uint32_t *t = ka; ka = kb; kb = t;
            t = va; va = vb; vb = t;
        }
        // after 4 passes the sorted order is back in val_a
        if (fold)
            hipLaunchKernelGGL(scan_sums_sorted_kernel, dim3(w.nb_scan), dim3(SC_THREADS), 0, stream, N, w.n_ranked, w.tiles_sorted, w.sums);
        else if (sorted_boxes)
            hipLaunchKernelGGL(scan_sums_kernel<true>, dim3(w.nb_scan), dim3(SC_THREADS), 0, stream, N, w.n_ranked, w.val_a, a->tiles_per_gauss,
                               w.sums, w.tiles_sorted, boxes, w.boxes_sorted);
        else
            hipLaunchKernelGGL(scan_sums_kernel<false>, dim3(w.nb_scan), dim3(SC_THREADS), 0, stream, N, w.n_ranked, w.val_a, a->tiles_per_gauss,
                               w.sums, w.tiles_sorted);
        EmitPrep ep;
        ep.jrec = w.jrec; ep.chunk_first = w.chunk_first; ep.nb_chunks = MAX_CHUNKS; ep.chunk = RS_THREADS * RS_ITEMS_I;
        ep.n_per_cam = N / (a->n_cameras > 1 ? a->n_cameras : 1);
        ep.tile_size = a->tile_size;
        ep.tw = dns_tiles_w(a->width, a->tile_size); ep.th = dns_tiles_h(a->height, a->tile_size);
        ep.means2d = a->means2d; ep.radii = a->radii;
        ep.splats = a->tight_tiles ? reinterpret_cast<const float4 *>(a->splats) : nullptr;
        ep.boxes = boxes;
        ep.boxes_sorted = sorted_boxes ? w.boxes_sorted : nullptr;
        hipLaunchKernelGGL(scan_final_kernel, dim3(w.nb_scan), dim3(SC_THREADS), 0, stream, N, w.n_ranked, w.val_a,
                           (const int32_t *)w.tiles_sorted, w.sums, w.cum, w.total, a->n_isects, a->n_isects_max);
        hipLaunchKernelGGL(emit_prep_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, w.n_ranked, w.val_a, w.cum, ep);
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
    if (a->tight_tiles && !a->splats && !a->tile_boxes && a->N > 0) return DNSPLAT_ERR_INVALID_ARG;
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
        // every list is [0, 0): the compositing kernels of the fused path read tile_ends, which must not stay uninitialised
        if (a->tile_ends && hipMemsetAsync(a->tile_ends, 0, sizeof(int32_t) * (size_t)n_tiles, stream) != hipSuccess)
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
