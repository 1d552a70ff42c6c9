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
