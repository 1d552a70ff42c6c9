/*
 * dnsplat_oracle.c — CPU oracle for the dn-splatter rendering hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library.  The product (dn-splatter_amd/) never
 * imports, links or calls it.
 *
 * PARITY UNPINNED: the reference (maturk/dn-splatter) ships no tests, golden
 * vectors or fixtures for this path, and the arithmetic lives in the
 * un-vendored third-party dependency gsplat==1.0.0 (pyproject.toml:8), which
 * cannot be imported or built here (CUDA-only, no network).  This file restates
 * the published algorithm at the reference's call sites
 * dn_splatter/dn_model.py:495-516 (rasterization) and :564-575
 * (rasterize_gaussians).  It is pinned instead by closed-form known-answer
 * tests and by torch-autograd (fp64) checks of its backward passes.
 *
 * Build: see oracle/Makefile  ->  oracle/_build/libdnsplat_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/dnsplat_constants.h"

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

/* Gradient scatter of the compositing backward (orc_rasterize_bwd): 0 = every term is added to the fp32 output arrays with omp
 * atomics, in whatever order the threads arrive (what gsplat's CUDA atomics do: the fp32 result varies from run to run on
 * ill-conditioned entries); 1 = the terms — still evaluated in REAL arithmetic — are accumulated in double and rounded once, so the
 * result does not depend on the order (to the last bit except where the sum sits within 1e-12 of a rounding boundary).  The
 * deterministic-mode tests of the HIP path (tests/test_gpu_determinism.py, tools/parity_seed_sweep.py) compare against that. */
int orc_exact_accum = 0;
void orc_set_exact_accum(int on) { orc_exact_accum = on; }
int orc_get_exact_accum(void) { return orc_exact_accum; }
/* orc_project_bwd's rotation -> quaternion stage evaluated in magnitudes (oracle.py ConditionTrace's Jacobian sweeps only). */
int orc_cond_quat_abs = 0;
void orc_set_cond_quat_abs(int on) { orc_cond_quat_abs = on; }

/* ---- fp32 instantiation ---- */
#define REAL float
#define FN(name) CAT(name, _f32)
#define SQRT sqrtf
#define EXP expf
#define CEIL ceilf
#define FLOOR floorf
#define FABS fabsf
#define ORC_ULP_BELOW_ONE 5.9604644775390625e-8 /* 2^-24: spacing of floats in [0.5, 1) */
#include "oracle_impl.inc"
#undef REAL
#undef FN
#undef SQRT
#undef EXP
#undef CEIL
#undef FLOOR
#undef FABS
#undef ORC_ULP_BELOW_ONE

/* ---- fp64 instantiation ---- */
#define REAL double
#define FN(name) CAT(name, _f64)
#define SQRT sqrt
#define EXP exp
#define CEIL ceil
#define FLOOR floor
#define FABS fabs
#define ORC_ULP_BELOW_ONE 1.1102230246251565e-16 /* 2^-53 */
#include "oracle_impl.inc"
#undef REAL
#undef FN

/* ------------------------------------------------ precision-independent */

/* A.3: stable ascending sort of (isect_id, flatten_id) pairs on the full key.
 * LSD radix, 8 bits x 8 passes; stability is what makes the order of equal
 * (tile, depth) keys deterministic (emission order = Gaussian index). */
void orc_sort_isects(int64_t n, int64_t *keys, int32_t *vals)
{
    if (n <= 1) return;
    int64_t *k2 = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    int32_t *v2 = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    int64_t *ka = keys, *kb = k2;
    int32_t *va = vals, *vb = v2;
    for (int pass = 0; pass < 8; ++pass) {
        int shift = pass * 8;
        size_t count[257];
        memset(count, 0, sizeof(count));
        for (int64_t i = 0; i < n; ++i) count[(((uint64_t)ka[i]) >> shift & 0xff) + 1]++;
        int trivial = 0;
        for (int d = 0; d < 256; ++d) if (count[d + 1] == (size_t)n) trivial = 1;
        if (trivial) continue;
        for (int d = 0; d < 256; ++d) count[d + 1] += count[d];
        for (int64_t i = 0; i < n; ++i) {
            size_t dst = count[((uint64_t)ka[i]) >> shift & 0xff]++;
            kb[dst] = ka[i]; vb[dst] = va[i];
        }
        int64_t *tk = ka; ka = kb; kb = tk;
        int32_t *tv = va; va = vb; vb = tv;
    }
    if (ka != keys) {
        memcpy(keys, ka, (size_t)n * sizeof(int64_t));
        memcpy(vals, va, (size_t)n * sizeof(int32_t));
    }
    free(k2); free(v2);
}

/* A.3: first sorted index of every tile; tiles without intersections point at
 * the start of the next non-empty tile (or n at the end). */
void orc_isect_offsets(int64_t n, const int64_t *sorted_keys, int n_tiles, int32_t *offsets)
{
    int64_t i = 0;
    for (int t = 0; t < n_tiles; ++t) {
        while (i < n && (int)(sorted_keys[i] >> 32) < t) ++i;
        offsets[t] = (int32_t)i;
    }
}

int orc_max_channels(void) { return DNS_MAX_CH; }
