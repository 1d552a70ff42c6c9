// bin_ranges.h — digit counts of a run of (tile, gaussian) pairs WITHOUT generating the pairs.
//
// The first pass of the tile sort (binning.hip, GEN instantiations) works on pairs that exist only as a 16-byte record per
// Gaussian: its pairs are the tiles of a box, row-major — pair k of the record has tile id  base + (k / w) * tw + k % w.  The
// histogram kernel of that pass only needs, per 4096-pair chunk, how many pairs fall on each value of the pass's digit, and the
// digit is the LOW dbits bits of the tile id (shift 0).  Within one box row consecutive pairs have consecutive tile ids, hence
// consecutive digits modulo 2^dbits: a row contributes +1 to a cyclic RANGE of digits (plus a constant to all of them for every
// full 2^dbits it spans).  So a record's share of a chunk is a handful of range increments — one per box row it has inside the
// chunk — instead of one LDS atomic per pair behind an owner search and a record gather per pair.
//
// Plain C++ on purpose (no HIP types): tests/test_host_logic.py compiles this header with g++ and checks it against the
// pair-by-pair count on random boxes, chunk cuts and digit widths.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define DNS_BR_HD __host__ __device__ __forceinline__
#else
#define DNS_BR_HD inline
#endif

// Pairs [lo, hi) (0 <= lo < hi <= rows * w) of a record whose box starts at tile id `base`, is `w` tiles wide and lies in a grid of
// `tw` tile columns.  Calls  range(d0, len)  for every cyclic digit range [d0, d0 + len) (0 < len < 2^dbits, d0 < 2^dbits; the
// range may wrap past 2^dbits - 1) that receives +1, and returns the constant every digit receives on top.
template <typename RangeFn>
DNS_BR_HD uint32_t dns_record_digit_ranges(uint32_t base, uint32_t w, uint32_t tw, uint32_t lo, uint32_t hi, int dbits, RangeFn &&range)
{
    const uint32_t mask = (1u << dbits) - 1u;
    const uint32_t r0 = lo / w, c0 = lo - r0 * w;
    const uint32_t last = hi - 1u;
    const uint32_t r1 = last / w, c1 = last - r1 * w;
    uint32_t all = 0u;
    uint32_t t_row = base + r0 * tw;                       // tile id of column 0 of the box in row r
    for (uint32_t r = r0; r <= r1; ++r, t_row += tw) {
        const uint32_t a = (r == r0) ? c0 : 0u;
        const uint32_t b = (r == r1) ? c1 : w - 1u;
        const uint32_t len = b - a + 1u;
        all += len >> dbits;
        const uint32_t rem = len & mask;
        if (rem) range((t_row + a) & mask, rem);
    }
    return all;
}
