// splat_common.h — shared device-side helpers for the gfx950 kernels of libdnsplat.
// Wave = 64 lanes everywhere in this library (CDNA4); nothing here is written for 32-wide warps.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dnsplat.h"
#include "../../include/dnsplat_constants.h"

#define DNS_WAVE 64
#define DNS_REC DNSPLAT_RECORD_FLOATS  // floats per splat / gradient record

// Record slots
#define REC_X 0
#define REC_Y 1
#define REC_CA 2
#define REC_CB 3
#define REC_CC 4
#define REC_OPAC 5
#define REC_CH0 6
#define REC_ABSX 14
#define REC_ABSY 15

#define DNS_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return DNSPLAT_ERR_LAUNCH;    \
    } while (0)

// wave64 ballot straight from the predicate: HIP's __ballot(int) widens the bool first, and hipcc then rebuilds the mask
// with a v_cndmask + v_cmp pair when the predicate already lives in scalar registers
__device__ __forceinline__ uint64_t dns_ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }

static inline int dns_tiles_w(int width, int tile) { return (width + tile - 1) / tile; }
static inline int dns_tiles_h(int height, int tile) { return (height + tile - 1) / tile; }

// A.3 tile bounding box [min,max) in tile units, clamped to the grid.  Evaluated in the same
// operation order as the oracle's tile_bbox (oracle/oracle_impl.inc) — all divisions are by a
// power of two and floor/ceil are exact, so both sides agree bit for bit.
__device__ __forceinline__ void dns_tile_bbox(float mx, float my, float radius, int tile_size, int tw, int th,
                                              int &x0, int &y0, int &x1, int &y1)
{
    float ts = (float)tile_size;
    float tr = radius / ts;
    float tcx = mx / ts, tcy = my / ts;
    float fx0 = floorf(tcx - tr), fy0 = floorf(tcy - tr);
    float fx1 = ceilf(tcx + tr), fy1 = ceilf(tcy + tr);
    fx0 = fminf(fmaxf(fx0, 0.f), (float)tw);
    fy0 = fminf(fmaxf(fy0, 0.f), (float)th);
    fx1 = fminf(fmaxf(fx1, 0.f), (float)tw);
    fy1 = fminf(fmaxf(fy1, 0.f), (float)th);
    x0 = (int)fx0; y0 = (int)fy0; x1 = (int)fx1; y1 = (int)fy1;
}

// Tile box of the part of a splat that can reach alpha >= 1/255 at some pixel centre (the "SnugBox" of Speedy-Splat): the
// axis-aligned box of the ellipse  sigma(d) = 1/2 d^T conic d <= ln(255 opacity), widened by a rounding margin and clipped to
// gsplat's 3-sigma box (dns_tile_bbox).  Every (tile, splat) pair it leaves out fails the per-pixel alpha test at all 256 pixel
// centres of the tile, so the images and gradients composited from the shorter lists are the same numbers; the tile lists
// themselves are NOT gsplat's any more, which is why only the fused get_outputs path (whose lists are internal) asks for it.
// Evaluated by the projection kernel (tile counts) and by the emit kernel (tile pairs) from the same record floats: contraction
// is pinned off so that both translation units round identically.
__device__ __forceinline__ void dns_snug_tile_bbox(float mx, float my, float ca, float cb, float cc, float opac, float radius,
                                                   int tile_size, int tw, int th, int &x0, int &y0, int &x1, int &y1)
{
#pragma clang fp contract(off)
    dns_tile_bbox(mx, my, radius, tile_size, tw, th, x0, y0, x1, y1);
    const float tau = logf(255.f * opac);          // alpha = opacity exp(-sigma) >= 1/255  <=>  sigma <= tau
    if (!(tau > 0.f)) { x1 = x0; y1 = y0; return; }   // opacity <= 1/255: never composited
    // opacity within 0.2 % of 1/255: the ellipse is a fraction of a pixel wide and the fp32 error of the compositing alpha test
    // (relative ~1e-6 in opacity x vis) is no longer small against the margins below: keep gsplat's box
    if (tau < 2e-3f) return;
    const float det = ca * cc - cb * cb;
    if (!(det > 0.f)) return;                       // degenerate conic: keep gsplat's box
    // half widths sqrt(2 tau Sigma_xx), sqrt(2 tau Sigma_yy) with Sigma = conic^-1; `rel` covers the cancellation in det
    const float s = 2.f * tau / det;
    const float rel = 1e-4f + 2.4e-7f * ((ca * cc + cb * cb) / det);
    const float hx = sqrtf(s * cc) * (1.f + rel) + 0.01f, hy = sqrtf(s * ca) * (1.f + rel) + 0.01f;
    const float ts = (float)tile_size;
    // tile column t holds the pixel centres t ts + 0.5 ... t ts + ts - 0.5
    const float fx0 = fminf(fmaxf(ceilf((mx - hx - (ts - 0.5f)) / ts), 0.f), (float)tw);
    const float fx1 = fminf(fmaxf(floorf((mx + hx - 0.5f) / ts) + 1.f, 0.f), (float)tw);
    const float fy0 = fminf(fmaxf(ceilf((my - hy - (ts - 0.5f)) / ts), 0.f), (float)th);
    const float fy1 = fminf(fmaxf(floorf((my + hy - 0.5f) / ts) + 1.f, 0.f), (float)th);
    x0 = max(x0, (int)fx0); x1 = max(min(x1, (int)fx1), x0);
    y0 = max(y0, (int)fy0); y1 = max(min(y1, (int)fy1), y0);
}

// XCD-aware block remap (guide §5.5 T1): hardware places block b on XCD b % 8.  Give every XCD one
// contiguous band of work ids so that neighbouring tiles — which share most of their splats —
// hit the same 4 MiB L2.  Bijective for any grid size.
__device__ __forceinline__ int dns_xcd_remap(int b, int n)
{
    const int nx = 8;
    int q = n / nx, r = n % nx;
    int xcd = b % nx, k = b / nx;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// blockIdx -> tile for the compositing kernels.  Measured on the 1080p / 1 M benchmark frame (raster_bwd, ms):
//   0  one band of consecutive tile rows per XCD (dns_xcd_remap: neighbouring tiles share an L2)      1.89
//   1  identity: consecutive tiles round-robin over the XCDs                                           1.87
//   2  bands over a row-interleaved image (every XCD gets rows from the whole height)                  1.91
// Both kernels are bound by vector instructions, not by L2 misses (0.2 TB/s of HBM traffic), so the L2 locality of the
// bands buys nothing here and the finer interleave of the identity order balances the XCDs slightly better.
#ifndef DNS_TILE_ORDER
#define DNS_TILE_ORDER 1
#endif
__device__ __forceinline__ int dns_tile_of_block(int b, int n_tiles, int tw)
{
#if DNS_TILE_ORDER == 0
    return dns_xcd_remap(b, n_tiles);
#elif DNS_TILE_ORDER == 1
    return b;
#else
    const int w = dns_xcd_remap(b, n_tiles);
    const int th = n_tiles / tw;
    int rp = w / tw;
    const int col = w - rp * tw;
    int row = rp;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int cnt = (th - c + 7) >> 3;      // image rows with row % 8 == c
        if (rp >= 0 && rp < cnt) row = c + 8 * rp;
        rp -= cnt;                              // negative once the class is found: no later class matches
        if (rp < 0) rp = -0x40000000;
    }
    return row * tw + col;
#endif
}

// ------------------------------------------------------------------------------------------------
// Compositing math shared by the forward and backward kernels, so that both take bit-identical
// decisions on (pixel, splat) pairs:
//     sigma = 1/2 (a dx^2 + c dy^2) + b dx dy,  vis = exp(-sigma),  alpha = min(0.999, o vis),
//     pair skipped if sigma < 0 or alpha < 1/255.
// The exponent is evaluated as exp2 of a pre-scaled quadratic form: e = -log2(e) * sigma with the
// scaling folded into the three conic coefficients once per splat (3 multiplies per splat instead of
// one per pair).  The fused-multiply-add sequence is spelled out so that compiler contraction
// cannot make the two kernels round differently.
#define DNS_LOG2E 1.4426950408889634f

// DNS_EXP_SYM = 1: the quadratic form is evaluated through its two half-gradients,
//     u = na dx + hb dy,  w = hb dx + nc dy,  e = dx u + dy w        (na, hb, nc = -log2e/2 * (a, b, c)),
// which are exactly what the backward needs for d sigma / d(dx, dy) = -(2 / log2e) (u, w): the mean gradient costs one
// multiply per component instead of rebuilding a dx + b dy from the unscaled conic (3 packed instructions less per
// step, and the unscaled conic leaves the backward's registers).  The forward shares na dx and hb dx between the two
// pixels of a lane, so it pays nothing for the sixth operation.  DNS_EXP_SYM = 0 keeps the 5-operation Horner form.
#ifndef DNS_EXP_SYM
#define DNS_EXP_SYM 1
#endif

struct DnsConicE {
    float na, nb, nc;  // -log2e/2 * a, [SYM: -log2e/2 * b (= hb) | else: -log2e * b], -log2e/2 * c
};

__device__ __forceinline__ DnsConicE dns_conic_e(float ca, float cb, float cc)
{
    DnsConicE q;
    q.na = (-0.5f * DNS_LOG2E) * ca;
#if DNS_EXP_SYM
    q.nb = (-0.5f * DNS_LOG2E) * cb;
#else
    q.nb = (-DNS_LOG2E) * cb;
#endif
    q.nc = (-0.5f * DNS_LOG2E) * cc;
    return q;
}

// e = -log2e * sigma  (<= 0 for a valid pair).  The operation sequence below is THE definition both compositing kernels
// follow instruction for instruction (the backward in packed form), so that they take bit-identical skip decisions.
#if DNS_EXP_SYM
__device__ __forceinline__ float dns_half_grad_u(const DnsConicE &q, float dx, float dy) { return __builtin_fmaf(q.nb, dy, q.na * dx); }
__device__ __forceinline__ float dns_half_grad_w(const DnsConicE &q, float dx, float dy) { return __builtin_fmaf(q.nc, dy, q.nb * dx); }
__device__ __forceinline__ float dns_exponent(const DnsConicE &q, float dx, float dy)
{
    const float u = dns_half_grad_u(q, dx, dy), w = dns_half_grad_w(q, dx, dy);
    return __builtin_fmaf(dx, u, dy * w);
}
#else
__device__ __forceinline__ float dns_exponent(const DnsConicE &q, float dx, float dy)
{
    const float u = __builtin_fmaf(q.na, dx, q.nb * dy);
    return __builtin_fmaf(dx, u, (q.nc * dy) * dy);
}
#endif

__device__ __forceinline__ float dns_exp2(float e) { return __builtin_amdgcn_exp2f(e); }

// Conservative test "no pixel centre of the rectangle [xl,xh] x [yl,yh] can pass alpha >= 1/255 for this
// splat".  The exact minimum of the quadratic form over the rectangle (it lies on the edge(s) nearest to
// the mean when the mean is outside) is compared with ln(255 o) plus a safety margin that dominates the
// fp32 rounding of the per-pixel evaluation, so a culled splat is one the per-pixel test would have
// skipped for every pixel: skipping it changes no result bit.  NaNs compare false => keep.
// The minimiser along an edge is located with v_rcp_f32 (1 ulp) instead of an IEEE division: an error of the location enters the
// value at second order (~1e-13 relative), eleven orders below the margin, and saves ~16 instructions per test.
__device__ __forceinline__ bool dns_cull_rect(float sx, float sy, float ca, float cb, float cc, float opac,
                                              float xl, float xh, float yl, float yh)
{
    const float dxl = sx - xh, dxh = sx - xl, dyl = sy - yh, dyh = sy - yl;
    const float X = fminf(fmaxf(0.f, dxl), dxh);  // point of [dxl,dxh] nearest to 0
    const float Y = fminf(fmaxf(0.f, dyl), dyh);
    float smin = 0.f, pos = 0.f;
    if (X != 0.f || Y != 0.f) {
        float sxe = 3.0e38f, sye = 3.0e38f, pxe = 0.f, pye = 0.f;
        if (X != 0.f) {
            const float dy = fminf(fmaxf(-cb * X * __builtin_amdgcn_rcpf(cc), dyl), dyh);
            pxe = 0.5f * (ca * X * X + cc * dy * dy);
            sxe = pxe + cb * X * dy;
        }
        if (Y != 0.f) {
            const float dx = fminf(fmaxf(-cb * Y * __builtin_amdgcn_rcpf(ca), dxl), dxh);
            pye = 0.5f * (ca * dx * dx + cc * Y * Y);
            sye = pye + cb * dx * Y;
        }
        if (sxe < sye) { smin = sxe; pos = pxe; } else { smin = sye; pos = pye; }
    }
    const float tau = __logf(255.f * opac);
    return smin > tau + 0.02f + 1e-5f * pos;
}
