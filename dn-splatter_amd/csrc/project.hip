// project.hip — per-Gaussian front end and back end (stage 1 and stage 5 of include/dnsplat.h).
//
// Replaces, fused into one kernel each way:
//   A0  activations               dn_splatter/dn_model.py:497-499
//   A1  fully_fused_projection    gsplat 1.0.0 (call site dn_model.py:495; math SURVEY.md A.2)
//   A2a tile count                gsplat isect_tiles pass 1
//   A5  spherical harmonics       gsplat spherical_harmonics + clamp_min(c+0.5,0)
//   A7  per-Gaussian normals      dn_model.py:543-560
//   A10 projection backward       SURVEY.md A.8
//
// THIS FILE IS COMPILED WITH -ffp-contract=off.  The integer products of this stage (radii,
// tiles_per_gauss, and the depth bits that order the tile lists) must be bit-identical to the CPU
// oracle, so the geometry is evaluated with IEEE +,-,*,/,sqrt in the very order
// oracle/oracle_impl.inc spells out, without FMA contraction.  The kernel is bandwidth-trivial
// (one thread per Gaussian, ~0.3 KB each way), so nothing is lost.
//
// Mapping: one lane per Gaussian, one wave per workgroup (DNS_PROJ_THREADS = 64: paired A/B against 128 / 256 threads at 1 M
// Gaussians: project_fwd -2 %, project_bwd -4 % — more workgroups per CU overlap their load / compute / store phases).  Camera
// constants are read through wave-uniform pointers, i.e. they live in SGPRs.

#include "splat_common.h"

namespace {

struct Cam {
    float Rv[9], t[3];
    float fx, fy, cx, cy;
    float pos[3];  // camera centre in world = -Rv^T t
};

__device__ __forceinline__ Cam load_cam(const float *__restrict__ vm, const float *__restrict__ K)
{
    Cam c;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) c.Rv[3 * i + j] = vm[4 * i + j];
        c.t[i] = vm[4 * i + 3];
    }
    c.fx = K[0]; c.fy = K[4]; c.cx = K[2]; c.cy = K[5];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        c.pos[i] = -(c.Rv[0 + i] * c.t[0] + c.Rv[3 + i] * c.t[1] + c.Rv[6 + i] * c.t[2]);
    return c;
}

struct Proj {
    float Rq[9], qn[4], inv_norm;
    float M[9];
    float mean_c[3], covar_c[9];
    float rz, rz2, tx, ty;
    bool x_in, y_in;
    float J[6];
    float cov2d[3], cov2d_blur[3], det_orig, det_blur;
    float conic[3], compensation, mean2d[2], radius;
};

__device__ __forceinline__ void quat_to_rotmat(const float *q, float *R, float *qn, float &inv)
{
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float s = x * x + y * y + z * z + w * w;
    inv = 1.f / sqrtf(s);
    w *= inv; x *= inv; y *= inv; z *= inv;
    float x2 = x * x, y2 = y * y, z2 = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    R[0] = 1.f - 2.f * (y2 + z2);
    R[1] = 2.f * (xy - wz);
    R[2] = 2.f * (xz + wy);
    R[3] = 2.f * (xy + wz);
    R[4] = 1.f - 2.f * (x2 + z2);
    R[5] = 2.f * (yz - wx);
    R[6] = 2.f * (xz - wy);
    R[7] = 2.f * (yz + wx);
    R[8] = 1.f - 2.f * (x2 + y2);
    qn[0] = w; qn[1] = x; qn[2] = y; qn[3] = z;
}

__device__ __forceinline__ void mm3(const float *A, const float *B, float *C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i + 0] * B[0 + j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void mm3_abt(const float *A, const float *B, float *C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[3 * i + 0] * B[3 * j + 0] + A[3 * i + 1] * B[3 * j + 1] + A[3 * i + 2] * B[3 * j + 2];
}
__device__ __forceinline__ void mm3_atb(const float *A, const float *B, float *C)
{
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[3 * i + j] = A[0 + i] * B[0 + j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}

// SURVEY.md A.2 steps 1-6; same expression tree as oracle project_one().
__device__ __forceinline__ bool project_one(const float *mean, const float *quat, const float *scale, const Cam &c,
                                            int W, int H, float eps2d, float near_plane, float far_plane,
                                            float radius_clip, Proj &st)
{
    st.radius = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        st.mean_c[i] = c.Rv[3 * i + 0] * mean[0] + c.Rv[3 * i + 1] * mean[1] + c.Rv[3 * i + 2] * mean[2] + c.t[i];
    if (st.mean_c[2] < near_plane || st.mean_c[2] > far_plane) return false;

    quat_to_rotmat(quat, st.Rq, st.qn, st.inv_norm);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) st.M[3 * i + j] = st.Rq[3 * i + j] * scale[j];
    float covar[9], tmp[9];
    mm3_abt(st.M, st.M, covar);
    mm3(c.Rv, covar, tmp);
    mm3_abt(tmp, c.Rv, st.covar_c);

    float x = st.mean_c[0], y = st.mean_c[1], z = st.mean_c[2];
    float tan_fovx = 0.5f * (float)W / c.fx;
    float tan_fovy = 0.5f * (float)H / c.fy;
    float lim_x = (float)DNS_FOV_CLAMP * tan_fovx;
    float lim_y = (float)DNS_FOV_CLAMP * tan_fovy;
    float rz = 1.f / z;
    float rz2 = rz * rz;
    float xz = x * rz, yz = y * rz;
    st.x_in = (xz <= lim_x && xz >= -lim_x);
    st.y_in = (yz <= lim_y && yz >= -lim_y);
    float tx = z * fminf(lim_x, fmaxf(-lim_x, xz));
    float ty = z * fminf(lim_y, fmaxf(-lim_y, yz));
    st.rz = rz; st.rz2 = rz2; st.tx = tx; st.ty = ty;
    float *J = st.J;
    J[0] = c.fx * rz; J[1] = 0.f; J[2] = -c.fx * tx * rz2;
    J[3] = 0.f; J[4] = c.fy * rz; J[5] = -c.fy * ty * rz2;
    const float *Cc = st.covar_c;
    float JC[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        JC[j] = J[0] * Cc[0 + j] + J[2] * Cc[6 + j];
        JC[3 + j] = J[4] * Cc[3 + j] + J[5] * Cc[6 + j];
    }
    st.cov2d[0] = JC[0] * J[0] + JC[2] * J[2];
    st.cov2d[1] = JC[1] * J[4] + JC[2] * J[5];
    st.cov2d[2] = JC[4] * J[4] + JC[5] * J[5];
    st.mean2d[0] = c.fx * x * rz + c.cx;
    st.mean2d[1] = c.fy * y * rz + c.cy;

    st.det_orig = st.cov2d[0] * st.cov2d[2] - st.cov2d[1] * st.cov2d[1];
    st.cov2d_blur[0] = st.cov2d[0] + eps2d;
    st.cov2d_blur[1] = st.cov2d[1];
    st.cov2d_blur[2] = st.cov2d[2] + eps2d;
    st.det_blur = st.cov2d_blur[0] * st.cov2d_blur[2] - st.cov2d_blur[1] * st.cov2d_blur[1];
    st.compensation = sqrtf(fmaxf(0.f, st.det_orig / st.det_blur));
    if (st.det_blur <= 0.f) return false;

    float inv_det = 1.f / st.det_blur;
    st.conic[0] = st.cov2d_blur[2] * inv_det;
    st.conic[1] = -st.cov2d_blur[1] * inv_det;
    st.conic[2] = st.cov2d_blur[0] * inv_det;

    float b = 0.5f * (st.cov2d_blur[0] + st.cov2d_blur[2]);
    float v1 = b + sqrtf(fmaxf((float)DNS_RADIUS_DISC_FLOOR, b * b - st.det_blur));
    float radius = ceilf((float)DNS_RADIUS_SIGMAS * sqrtf(v1));
    if (radius <= radius_clip) return false;
    if (st.mean2d[0] + radius <= 0.f || st.mean2d[0] - radius >= (float)W ||
        st.mean2d[1] + radius <= 0.f || st.mean2d[1] - radius >= (float)H)
        return false;
    st.radius = radius;
    return true;
}

// SURVEY.md A.5 real SH basis (Sloan "fast" form), degree <= 3.
__device__ __forceinline__ void sh_basis(int degree, float x, float y, float z, float *bas)
{
    bas[0] = 0.2820947917738781f;
    if (degree < 1) return;
    bas[1] = -0.48860251190292f * y;
    bas[2] = 0.48860251190292f * z;
    bas[3] = -0.48860251190292f * x;
    if (degree < 2) return;
    float z2 = z * z;
    float fTmp0B = -1.092548430592079f * z;
    float fC1 = x * x - y * y;
    float fS1 = 2.f * x * y;
    bas[4] = 0.5462742152960395f * fS1;
    bas[5] = fTmp0B * y;
    bas[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    bas[7] = fTmp0B * x;
    bas[8] = 0.5462742152960395f * fC1;
    if (degree < 3) return;
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    float fTmp1B = 1.445305721320277f * z;
    float fC2 = x * fC1 - y * fS1;
    float fS2 = x * fS1 + y * fC1;
    bas[9] = -0.5900435899266435f * fS2;
    bas[10] = fTmp1B * fS1;
    bas[11] = fTmp0C * y;
    bas[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
    bas[13] = fTmp0C * x;
    bas[14] = fTmp1B * fC1;
    bas[15] = -0.5900435899266435f * fC2;
}

__device__ __forceinline__ void sh_basis_grad(int degree, float x, float y, float z, float *dx, float *dy, float *dz)
{
#pragma unroll
    for (int k = 0; k < 16; ++k) dx[k] = dy[k] = dz[k] = 0.f;
    if (degree < 1) return;
    dy[1] = -0.48860251190292f;
    dz[2] = 0.48860251190292f;
    dx[3] = -0.48860251190292f;
    if (degree < 2) return;
    const float c2 = 0.5462742152960395f, c1 = 1.092548430592079f, c0 = 0.9461746957575601f;
    dx[4] = c2 * 2.f * y; dy[4] = c2 * 2.f * x;
    dy[5] = -c1 * z;      dz[5] = -c1 * y;
    dz[6] = 2.f * c0 * z;
    dx[7] = -c1 * z;      dz[7] = -c1 * x;
    dx[8] = c2 * 2.f * x; dy[8] = -c2 * 2.f * y;
    if (degree < 3) return;
    float z2 = z * z;
    float fC1 = x * x - y * y, fS1 = 2.f * x * y;
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    float fTmp1B = 1.445305721320277f * z;
    const float c3 = 0.5900435899266435f;
    float dS2dx = 6.f * x * y, dS2dy = 3.f * x * x - 3.f * y * y;
    float dC2dx = 3.f * x * x - 3.f * y * y, dC2dy = -6.f * x * y;
    dx[9] = -c3 * dS2dx; dy[9] = -c3 * dS2dy;
    dx[10] = fTmp1B * 2.f * y; dy[10] = fTmp1B * 2.f * x; dz[10] = 1.445305721320277f * fS1;
    dy[11] = fTmp0C; dz[11] = -2.f * 2.285228997322329f * z * y;
    dz[12] = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
    dx[13] = fTmp0C; dz[13] = -2.f * 2.285228997322329f * z * x;
    dx[14] = fTmp1B * 2.f * x; dy[14] = -fTmp1B * 2.f * y; dz[14] = 1.445305721320277f * fC1;
    dx[15] = -c3 * dC2dx; dy[15] = -c3 * dC2dy;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// dn_model.py:543-556: normal = +-R(q) e_argmin(scales), facing the camera.  Returns the world
// normal in n[], the axis in k, the column norm and the flip sign.
__device__ __forceinline__ void gaussian_normal(const float *Rq, const float *scale_raw, const float *mean,
                                                const float *campos, float *n, int &k, float &nrm, float &sgn)
{
    k = 0;
    float m = scale_raw[0];
    if (scale_raw[1] < m) { m = scale_raw[1]; k = 1; }
    if (scale_raw[2] < m) { m = scale_raw[2]; k = 2; }
    // Column k of R(q) by selection among REGISTER values, not by a run-time index: an indexed read — or a select between
    // loads, which LLVM folds back into a load through a selected pointer — keeps the whole projection state (the struct Rq
    // lives in) in scratch memory: 248 bytes per lane written and re-read per launch.  The empty asm pins each candidate in a
    // VGPR before the select.
    float r[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { r[i] = Rq[i]; asm volatile("" : "+v"(r[i])); }
    const float c0 = k == 0 ? r[0] : (k == 1 ? r[1] : r[2]);
    const float c1 = k == 0 ? r[3] : (k == 1 ? r[4] : r[5]);
    const float c2 = k == 0 ? r[6] : (k == 1 ? r[7] : r[8]);
    nrm = fmaxf(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), 1e-12f);
    n[0] = c0 / nrm; n[1] = c1 / nrm; n[2] = c2 / nrm;
    float vx = campos[0] - mean[0], vy = campos[1] - mean[1], vz = campos[2] - mean[2];
    float dot = n[0] * vx + n[1] * vy + n[2] * vz;
    sgn = (dot < 0.f) ? -1.f : 1.f;
    n[0] *= sgn; n[1] *= sgn; n[2] *= sgn;
}

// ------------------------------------------------------------------------------------------------
// Coalesced access to the SH coefficient rows.  A lane's 45 (or 48) coefficient floats are contiguous but
// 180 bytes away from its neighbour's, so a per-lane `row[j]` access makes every wave instruction touch 64
// different cache lines (measured: project_bwd 0.38 ms for 540 B/Gaussian = 1.0 TB/s).  The staged kernels
// copy the block's whole coefficient span global <-> LDS with 16-byte accesses by consecutive lanes and let
// each lane walk its own LDS row (odd row stride => conflict-free ds_read_b32).  Two layouts:
//   SPLIT  features_dc [N,3] + features_rest [N,15,3]   rows of 45 floats, span = features_rest
//   CAT    colours [N,16,3] (gsplat layout)               rows of 48 floats incl. band 0, LDS stride 49
#ifndef DNS_PROJ_THREADS
#define DNS_PROJ_THREADS 64
#endif
constexpr int SH_STAGE_THREADS = DNS_PROJ_THREADS;      // threads (= Gaussians) per workgroup of the staged kernels
enum ShLayout { SH_DIRECT = 0, SH_SPLIT = 1, SH_CAT = 2 };
template <int L> struct ShRowTraits { static constexpr int ROW = 45, LDS_ROW = 45; };
template <> struct ShRowTraits<SH_CAT> { static constexpr int ROW = 48, LDS_ROW = 49; };

template <int L>
__device__ __forceinline__ int sh_lds_index(int e)
{
    return L == SH_CAT ? e + e / ShRowTraits<L>::ROW : e;
}

// DNS_PROJ_NT: the coefficient rows are read once and their gradients written once per frame, as one coalesced stream per
// workgroup: 1 = non-temporal loads (they skip the CU's vector L1), 2 = non-temporal stores, 3 = both.  Inside the C2 frame
// (library variants interleaved in whole bench runs): project_fwd -10 % with 3, -6 % with 1; project_bwd +-0 with 3, -3 % with 1.
#ifndef DNS_PROJ_NT
#define DNS_PROJ_NT 3
#endif
#ifndef DNS_PROJ_STAGED_REC
#define DNS_PROJ_STAGED_REC 1
#endif
typedef float dns_v4f __attribute__((ext_vector_type(4)));

// DNS_PROJ_STAGE_UNROLL: all 16-byte loads of a workgroup's coefficient span are ISSUED before the first one is waited for.
// Round 4 find, from the ISA: the rolled loop below compiled to  global_load_dwordx4 -> s_waitcnt vmcnt(0) -> ds_write_b128  per
// iteration — ONE kilobyte in flight per wave, twelve HBM round trips in a row per workgroup, 12 KB in flight per CU with the 12
// waves the kernels' registers allow: about 3 MB on the whole chip where HBM needs bandwidth x latency = 6-8 MB.  That, not the
// occupancy, is what held the per-Gaussian kernels at 3.7-4.3 TB/s.  Unrolled, a wave has its whole 11.5 KB span in flight.
#ifndef DNS_PROJ_STAGE_UNROLL
#define DNS_PROJ_STAGE_UNROLL 1
#endif
template <int L>
__device__ __forceinline__ void sh_stage_in(const float *__restrict__ gbase, int nfloats, float *lds)
{
    const dns_v4f *g4 = reinterpret_cast<const dns_v4f *>(gbase);
    const int n4 = nfloats >> 2;
#if DNS_PROJ_STAGE_UNROLL
    constexpr int MAXIT = (SH_STAGE_THREADS * ShRowTraits<L>::ROW / 4 + SH_STAGE_THREADS - 1) / SH_STAGE_THREADS;
    dns_v4f v[MAXIT];
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        const int i = (int)threadIdx.x + k * SH_STAGE_THREADS;
        if (i < n4) {
#if DNS_PROJ_NT & 1
            v[k] = __builtin_nontemporal_load(g4 + i);
#else
            v[k] = g4[i];
#endif
        }
    }
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        const int i = (int)threadIdx.x + k * SH_STAGE_THREADS;
        if (i < n4) {
            const int o = sh_lds_index<L>(4 * i);   // ROW % 4 == 0 in the padded layout: the 4 floats share a row
            lds[o] = v[k].x; lds[o + 1] = v[k].y; lds[o + 2] = v[k].z; lds[o + 3] = v[k].w;
        }
    }
#else
    for (int i = threadIdx.x; i < n4; i += SH_STAGE_THREADS) {
#if DNS_PROJ_NT & 1
        const dns_v4f v = __builtin_nontemporal_load(g4 + i);
#else
        const dns_v4f v = g4[i];
#endif
        const int o = sh_lds_index<L>(4 * i);   // ROW % 4 == 0 in the padded layout: the 4 floats share a row
        lds[o] = v.x; lds[o + 1] = v.y; lds[o + 2] = v.z; lds[o + 3] = v.w;
    }
#endif
    for (int e = (n4 << 2) + threadIdx.x; e < nfloats; e += SH_STAGE_THREADS) lds[sh_lds_index<L>(e)] = gbase[e];
}

template <int L>
__device__ __forceinline__ void sh_stage_out(float *__restrict__ gbase, int nfloats, const float *lds)
{
    dns_v4f *g4 = reinterpret_cast<dns_v4f *>(gbase);
    const int n4 = nfloats >> 2;
    for (int i = threadIdx.x; i < n4; i += SH_STAGE_THREADS) {
        const int o = sh_lds_index<L>(4 * i);
        const dns_v4f v = {lds[o], lds[o + 1], lds[o + 2], lds[o + 3]};
#if DNS_PROJ_NT & 2
        __builtin_nontemporal_store(v, g4 + i);
#else
        g4[i] = v;
#endif
    }
    for (int e = (n4 << 2) + threadIdx.x; e < nfloats; e += SH_STAGE_THREADS) gbase[e] = lds[sh_lds_index<L>(e)];
}

// Row-selective staging (round 6): `rows` = one bit per Gaussian of the workgroup (a wave: SH_STAGE_THREADS ==
// 64).  A 16-byte piece is moved only if a selected row owns one of its floats — the coefficient rows of culled Gaussians (28 % on
// the benchmark scenes) are neither read nor, once known to be zero, written again.  Rows are 180 bytes at a 180-byte stride, so a
// skipped row saves the lines it does not share with a selected neighbour.  DNS_PROJ_VISIBLE_ROWS=0 restores whole-span staging.
#ifndef DNS_PROJ_VISIBLE_ROWS
#define DNS_PROJ_VISIBLE_ROWS 1
#endif
template <int L>
__device__ __forceinline__ bool sh_piece_selected(int piece, uint64_t rows)
{
    constexpr int ROW = ShRowTraits<L>::ROW;
    const int r0 = (4 * piece) / ROW, r1 = (4 * piece + 3) / ROW;
    return ((rows >> r0) | (rows >> r1)) & 1ull;
}
template <int L>
__device__ __forceinline__ void sh_stage_in_rows(const float *__restrict__ gbase, int nfloats, float *lds, uint64_t rows)
{
    const dns_v4f *g4 = reinterpret_cast<const dns_v4f *>(gbase);
    const int n4 = nfloats >> 2;
    constexpr int MAXIT = (SH_STAGE_THREADS * ShRowTraits<L>::ROW / 4 + SH_STAGE_THREADS - 1) / SH_STAGE_THREADS;
    dns_v4f v[MAXIT];
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        const int i = (int)threadIdx.x + k * SH_STAGE_THREADS;
        if (i < n4 && sh_piece_selected<L>(i, rows)) {
#if DNS_PROJ_NT & 1
            v[k] = __builtin_nontemporal_load(g4 + i);
#else
            v[k] = g4[i];
#endif
        }
    }
#pragma unroll
    for (int k = 0; k < MAXIT; ++k) {
        const int i = (int)threadIdx.x + k * SH_STAGE_THREADS;
        if (i < n4 && sh_piece_selected<L>(i, rows)) {
            const int o = sh_lds_index<L>(4 * i);
            lds[o] = v[k].x; lds[o + 1] = v[k].y; lds[o + 2] = v[k].z; lds[o + 3] = v[k].w;
        }
    }
    // the span of a block is a multiple of 16 bytes unless it is the scene's last, partial one
    for (int e = (n4 << 2) + threadIdx.x; e < nfloats; e += SH_STAGE_THREADS) lds[sh_lds_index<L>(e)] = gbase[e];
}
template <int L>
__device__ __forceinline__ void sh_stage_out_rows(float *__restrict__ gbase, int nfloats, const float *lds, uint64_t rows)
{
    dns_v4f *g4 = reinterpret_cast<dns_v4f *>(gbase);
    const int n4 = nfloats >> 2;
    for (int i = threadIdx.x; i < n4; i += SH_STAGE_THREADS) {
        if (!sh_piece_selected<L>(i, rows)) continue;
        const int o = sh_lds_index<L>(4 * i);
        const dns_v4f v = {lds[o], lds[o + 1], lds[o + 2], lds[o + 3]};
#if DNS_PROJ_NT & 2
        __builtin_nontemporal_store(v, g4 + i);
#else
        g4[i] = v;
#endif
    }
    for (int e = (n4 << 2) + threadIdx.x; e < nfloats; e += SH_STAGE_THREADS) gbase[e] = lds[sh_lds_index<L>(e)];
}

// Feature channel `ch` (run-time: 3 colours or n_colors direct ones, then depth, then the normal) of a record held in
// registers.  A plain r[REC_CH0 + ch] with a run-time index sends the whole array — and every other privately indexed array
// of the kernel — to scratch memory: 248 bytes per lane written out and read back per launch, which showed up as ~250 B per
// Gaussian of extra WRITE_SIZE in project_bwd (2.0x its algorithmic bytes) and ~55 B in project_fwd.  A select chain over
// the 8 possible positions keeps everything in VGPRs.
__device__ __forceinline__ void rec_set_ch(float *r, int ch, float v)
{
#pragma unroll
    for (int k = 0; k < DNS_MAX_CH; ++k) r[REC_CH0 + k] = (k == ch) ? v : r[REC_CH0 + k];   // value selects (a conditional store
                                                                                          // becomes a store through a selected pointer)
}
__device__ __forceinline__ float rec_get_ch(const float *r, int ch)
{
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < DNS_MAX_CH; ++k) v = (k == ch) ? r[REC_CH0 + k] : v;
    return v;
}

// Packed colour-gradient slab (include/dnsplat.h, dnsplat_visible_index), in 4-byte words: header 8 | masks 2 nb | offsets nb | pad to 4 |
// rows 3 capacity | pad to 4;  nb = ceil(N / 64).
__host__ __device__ __forceinline__ size_t dns_packed_rows_offset(int nb) { return ((size_t)8 + 3 * (size_t)nb + 3) & ~(size_t)3; }
__host__ __device__ __forceinline__ size_t dns_packed_slab_words(int nb, int capacity)
{
    return (dns_packed_rows_offset(nb) + 3 * (size_t)capacity + 3) & ~(size_t)3;
}

struct FwdParams {
    dnsplat_scene s;
    dnsplat_camera c;
    dnsplat_proj_out o;
    int blocks;          // blocks of 256 Gaussians (phase 2 walks them with fewer workgroups)
};

// PHASE (dnsplat_proj_out.phase): 0 = everything in one launch; 1 = geometry only (the SH colour channels of the records stay
// zero: no coefficient is read, L is ignored); 2 = the SH colours of the Gaussians phase 1 found visible, written into their
// records (channels 0-2).  1 + 2 on two streams let the bandwidth-bound colour half (228 of the 304 B a visible Gaussian costs)
// run beside the latency-bound binning kernels, which need nothing but phase 1's outputs.
template <int L, int PHASE = 0>
__global__ __launch_bounds__(SH_STAGE_THREADS) void project_fwd_kernel(FwdParams p)
{
    __shared__ float sh_lds[(L == SH_DIRECT || PHASE == 1) ? 1 : SH_STAGE_THREADS * ShRowTraits<L>::LDS_ROW];
    // PHASE 2 is launched with a FEW resident workgroups that walk the blocks of 256 Gaussians (p.blocks of them): it is meant to
    // run beside other kernels and must leave them most of the wave slots.  The other phases: one block per workgroup.
    // The per-block work as a lambda: PHASE 2 calls it in a loop, the others exactly once — written as a loop for all of them, LLVM kept
    // every loop-invariant of the body (the staging's per-lane row indices and LDS addresses: ~70 VGPRs) live across the projection.
    auto block_body = [&](const int vb) {
    const int g = vb * blockDim.x + threadIdx.x;
    // The lane's own parameters are requested BEFORE the coefficient rows are staged (and waited for after): one memory round trip
    // for everything the Gaussian needs instead of three in a row (rows | camera | parameters).  A lane beyond N re-reads the last
    // Gaussian and drops it below.
    float mean[3], quat[4], sc_raw[3], opac_raw;
    {
        const int gc = min(g, p.s.N - 1);
#pragma unroll
        for (int i = 0; i < 3; ++i) mean[i] = p.s.means[3 * gc + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) quat[i] = p.s.quats[4 * gc + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) sc_raw[i] = p.s.scales[3 * gc + i];
        opac_raw = p.s.opacities[gc];
    }
    // VISIBLE_ROWS (one-launch kernel, staged layouts): the lane projects its Gaussian FIRST and only the coefficient rows of the
    // Gaussians that survive the culling are staged — a dependent round trip (parameters, then rows) that the other waves of the CU
    // cover, for 28 % fewer row bytes on the benchmark scenes.
    constexpr bool VISIBLE_ROWS = DNS_PROJ_VISIBLE_ROWS && PHASE == 0 && L != SH_DIRECT && SH_STAGE_THREADS == DNS_WAVE;
    Cam cam_pre;
    Proj st_pre;
    float sc_pre[3] = {0.f, 0.f, 0.f};
    bool ok_pre = false;
    if (VISIBLE_ROWS) {
        // every lane, no branch around it (a lane beyond N holds the last Gaussian's parameters): the camera stays in SGPRs
        cam_pre = load_cam(p.c.viewmat, p.c.K);
#pragma unroll
        for (int i = 0; i < 3; ++i) sc_pre[i] = p.s.scales_are_log ? expf(sc_raw[i]) : sc_raw[i];
        ok_pre = project_one(mean, quat, sc_pre, cam_pre, p.c.width, p.c.height, p.c.eps2d, p.c.near_plane, p.c.far_plane,
                             p.c.radius_clip, st_pre) && g < p.s.N;
    }
    if (VISIBLE_ROWS) __builtin_amdgcn_sched_barrier(0);      // the projection is finished (registers released) before the rows are requested
    const uint64_t vis_rows = VISIBLE_ROWS ? __ballot(ok_pre) : ~0ull;
    if (L != SH_DIRECT && PHASE != 1) {
        const int g0 = vb * SH_STAGE_THREADS;
        const int nG = min(SH_STAGE_THREADS, p.s.N - g0);
        const float *base = (L == SH_CAT ? p.s.sh0 : p.s.shN) + (size_t)g0 * ShRowTraits<L>::ROW;
        if (VISIBLE_ROWS) sh_stage_in_rows<L>(base, nG * ShRowTraits<L>::ROW, sh_lds, vis_rows);
        else sh_stage_in<L>(base, nG * ShRowTraits<L>::ROW, sh_lds);
        __syncthreads();
        if (VISIBLE_ROWS) __builtin_amdgcn_sched_barrier(0);
    }
    if (PHASE == 2 && (g >= p.s.N || p.o.radii[g] <= 0)) return;
    // STAGED: the 64-byte records of the workgroup's Gaussians leave through LDS as one coalesced stream (the coefficient rows are
    // dead by then) instead of four 16-byte pieces per lane at a 64-byte stride, which is 4x the write requests for the same lines.
    // Every lane has to reach that store, so the per-Gaussian work sits in a do { } while (false) whose `break`s replace the early returns.
    constexpr bool STAGED = DNS_PROJ_STAGED_REC && PHASE == 0 && L != SH_DIRECT && SH_STAGE_THREADS == DNS_WAVE;
    float r[DNS_REC];
#pragma unroll
    for (int i = 0; i < DNS_REC; ++i) r[i] = 0.f;
    bool phase2_done = false;
    do {                                       // `break` = this Gaussian is finished (beyond N, culled, or its record is in r[])
    if (g >= p.s.N) break;
    const Cam cam = VISIBLE_ROWS ? cam_pre : load_cam(p.c.viewmat, p.c.K);

    float sc[3];
    if (PHASE == 2) {
        // colours only: same arithmetic, in the same order, as the one-launch kernel below
        float dx = mean[0] - cam.pos[0], dy = mean[1] - cam.pos[1], dz = mean[2] - cam.pos[2];
        float inorm = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
        dx *= inorm; dy *= inorm; dz *= inorm;
        float bas[16];
        sh_basis(p.s.sh_degree, dx, dy, dz, bas);
        const int nb = (p.s.sh_degree + 1) * (p.s.sh_degree + 1);
        float col[3];
        const float *row = sh_lds + threadIdx.x * ShRowTraits<L>::LDS_ROW;
        const float *c0 = (L == SH_CAT) ? row : p.s.sh0 + (size_t)g * p.s.sh0_stride;
        const float *cN = (L == SH_DIRECT) ? p.s.shN + (size_t)g * p.s.shN_stride : (L == SH_CAT ? row + 3 : row);
        col[0] = bas[0] * c0[0]; col[1] = bas[0] * c0[1]; col[2] = bas[0] * c0[2];
#pragma unroll
        for (int k = 1; k < 16; ++k)
            if (k < nb) {
                col[0] += bas[k] * cN[3 * (k - 1) + 0];
                col[1] += bas[k] * cN[3 * (k - 1) + 1];
                col[2] += bas[k] * cN[3 * (k - 1) + 2];
            }
        float *rec = p.o.splats + (size_t)g * DNS_REC;
        rec[REC_CH0 + 0] = fmaxf(col[0] + 0.5f, 0.f);
        rec[REC_CH0 + 1] = fmaxf(col[1] + 0.5f, 0.f);
        rec[REC_CH0 + 2] = fmaxf(col[2] + 0.5f, 0.f);
        phase2_done = true;
        break;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) sc[i] = VISIBLE_ROWS ? sc_pre[i] : (p.s.scales_are_log ? expf(sc_raw[i]) : sc_raw[i]);
    float opac = opac_raw;
    if (p.s.opacities_are_logit) opac = sigmoidf(opac);

    Proj st;
    bool ok;
    if (VISIBLE_ROWS) { st = st_pre; ok = ok_pre; }
    else ok = project_one(mean, quat, sc, cam, p.c.width, p.c.height, p.c.eps2d, p.c.near_plane, p.c.far_plane, p.c.radius_clip, st);

    float *rec = p.o.splats + (size_t)g * DNS_REC;
    float4 *rec4 = reinterpret_cast<float4 *>(rec);
    if (!ok) {
        p.o.radii[g] = 0;
        p.o.means2d[2 * g] = 0.f; p.o.means2d[2 * g + 1] = 0.f;
        p.o.depths[g] = 0.f;
        p.o.conics[3 * g] = 0.f; p.o.conics[3 * g + 1] = 0.f; p.o.conics[3 * g + 2] = 0.f;
        if (p.o.compensations) p.o.compensations[g] = 0.f;
        p.o.tiles_per_gauss[g] = 0;
        if (p.c.tight_tiles) p.o.tiles_bin[g] = 0;
        if (p.o.tile_boxes) reinterpret_cast<int2 *>(p.o.tile_boxes)[g] = make_int2(0, 0);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!STAGED && !p.o.skip_culled_records) { rec4[0] = z4; rec4[1] = z4; rec4[2] = z4; rec4[3] = z4; }
        if (p.o.normals_world) {
            // the reference computes the normal for every Gaussian, visible or not (dn_model.py:544-558)
            float Rq[9], qn[4], inv, n[3], nrm, sgn; int k;
            quat_to_rotmat(quat, Rq, qn, inv);
            const float *np = p.c.normal_frame ? p.c.normal_frame + 9 : cam.pos;
            float campos[3] = {np[0], np[1], np[2]};
            gaussian_normal(Rq, sc_raw, mean, campos, n, k, nrm, sgn);
            p.o.normals_world[3 * g] = n[0]; p.o.normals_world[3 * g + 1] = n[1]; p.o.normals_world[3 * g + 2] = n[2];
        }
        break;
    }

    const int tw = (p.c.width + p.c.tile_size - 1) / p.c.tile_size;
    const int th = (p.c.height + p.c.tile_size - 1) / p.c.tile_size;
    if (p.c.antialiased) opac *= st.compensation;
    // alpha = min(0.999, opacity x vis) can only clamp for a splat whose opacity exceeds the cap (vis <= 1): tell the compositing
    // backward whether this frame holds one at all (dnsplat_proj_out.saturation_flag; many lanes may store the same 1)
    if (p.o.saturation_flag && opac > (float)DNS_ALPHA_MAX) *p.o.saturation_flag = 1u;
    int x0, y0, x1, y1;
    dns_tile_bbox(st.mean2d[0], st.mean2d[1], st.radius, p.c.tile_size, tw, th, x0, y0, x1, y1);
    const int tiles_ref = (y1 - y0) * (x1 - x0);          // gsplat's count (A.3): what info["tiles_per_gauss"] / num_tiles_hit report
    if (p.c.tight_tiles) {
        dns_snug_tile_bbox(st.mean2d[0], st.mean2d[1], st.conic[0], st.conic[1], st.conic[2], opac, st.radius, p.c.tile_size, tw, th,
                           x0, y0, x1, y1);
        p.o.tiles_bin[g] = (y1 - y0) * (x1 - x0);          // what the binning of the fused path walks
    }
    // the box the binning will walk, ready-made (dnsplat_bin_args.tile_boxes): first tile id, then width | height << 16 — the
    // tile count is their product, so the binning's scan needs this one 8-byte record per Gaussian and not the count array as well
    if (p.o.tile_boxes) reinterpret_cast<int2 *>(p.o.tile_boxes)[g] = make_int2(y0 * tw + x0, (x1 - x0) | ((y1 - y0) << 16));

    p.o.radii[g] = (int32_t)st.radius;
    p.o.means2d[2 * g] = st.mean2d[0]; p.o.means2d[2 * g + 1] = st.mean2d[1];
    p.o.depths[g] = st.mean_c[2];
    p.o.conics[3 * g] = st.conic[0]; p.o.conics[3 * g + 1] = st.conic[1]; p.o.conics[3 * g + 2] = st.conic[2];
    if (p.o.compensations) p.o.compensations[g] = st.compensation;
    p.o.tiles_per_gauss[g] = tiles_ref;

    r[REC_X] = st.mean2d[0]; r[REC_Y] = st.mean2d[1];
    r[REC_CA] = st.conic[0]; r[REC_CB] = st.conic[1]; r[REC_CC] = st.conic[2];
    r[REC_OPAC] = opac;
    int ch = 0;
    if (PHASE == 1) {
        ch = 3;                      // the three colour channels are phase 2's
    } else if (p.s.sh_degree >= 0) {
        float dx = mean[0] - cam.pos[0], dy = mean[1] - cam.pos[1], dz = mean[2] - cam.pos[2];
        float inorm = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
        dx *= inorm; dy *= inorm; dz *= inorm;
        float bas[16];
        sh_basis(p.s.sh_degree, dx, dy, dz, bas);
        const int nb = (p.s.sh_degree + 1) * (p.s.sh_degree + 1);
        float col[3];
        if (L == SH_DIRECT) {
            const float *c0 = p.s.sh0 + (size_t)g * p.s.sh0_stride;
            const float *cN = p.s.shN + (size_t)g * p.s.shN_stride;
            col[0] = bas[0] * c0[0]; col[1] = bas[0] * c0[1]; col[2] = bas[0] * c0[2];
#pragma unroll
            for (int k = 1; k < 16; ++k)      // static indices (bas[] stays in registers), run-time band count
                if (k < nb) {
                    col[0] += bas[k] * cN[3 * (k - 1) + 0];
                    col[1] += bas[k] * cN[3 * (k - 1) + 1];
                    col[2] += bas[k] * cN[3 * (k - 1) + 2];
                }
        } else {
            const float *row = sh_lds + threadIdx.x * ShRowTraits<L>::LDS_ROW;   // this lane's LDS row
            const float *cN = L == SH_CAT ? row + 3 : row;
            if (L == SH_CAT) {
                col[0] = bas[0] * row[0]; col[1] = bas[0] * row[1]; col[2] = bas[0] * row[2];
            } else {
                const float *c0 = p.s.sh0 + (size_t)g * p.s.sh0_stride;
                col[0] = bas[0] * c0[0]; col[1] = bas[0] * c0[1]; col[2] = bas[0] * c0[2];
            }
#pragma unroll
            for (int k = 1; k < 16; ++k)      // static indices (bas[] stays in registers), run-time band count
                if (k < nb) {
                    col[0] += bas[k] * cN[3 * (k - 1) + 0];
                    col[1] += bas[k] * cN[3 * (k - 1) + 1];
                    col[2] += bas[k] * cN[3 * (k - 1) + 2];
                }
        }
        r[REC_CH0 + 0] = fmaxf(col[0] + 0.5f, 0.f);
        r[REC_CH0 + 1] = fmaxf(col[1] + 0.5f, 0.f);
        r[REC_CH0 + 2] = fmaxf(col[2] + 0.5f, 0.f);
        ch = 3;
    } else {
#pragma unroll
        for (int k = 0; k < DNS_MAX_CH; ++k)
            if (k < p.s.n_colors) {
                const float cv = p.s.colors[(size_t)g * p.s.n_colors + k];
                r[REC_CH0 + k] = p.s.colors_are_logit ? sigmoidf(cv) : cv;      // dn_model.py:491-492: sigmoid(colours), sh_degree = None
            }
        ch = p.s.n_colors;
    }
    if (p.o.with_depth_channel) { rec_set_ch(r, ch, st.mean_c[2]); ch += 1; }
    if (p.o.with_normal_channels || p.o.normals_world) {
        const float *np = p.c.normal_frame ? p.c.normal_frame + 9 : cam.pos;
        float campos[3] = {np[0], np[1], np[2]};
        float n[3], nrm, sgn; int k;
        gaussian_normal(st.Rq, sc_raw, mean, campos, n, k, nrm, sgn);
        if (p.o.normals_world) {
            p.o.normals_world[3 * g] = n[0]; p.o.normals_world[3 * g + 1] = n[1]; p.o.normals_world[3 * g + 2] = n[2];
        }
        if (p.o.with_normal_channels) {
            const float *Mn = p.c.normal_frame;
#pragma unroll
            for (int i = 0; i < 3; ++i)
                rec_set_ch(r, ch + i, Mn[3 * i + 0] * n[0] + Mn[3 * i + 1] * n[1] + Mn[3 * i + 2] * n[2]);
        }
    }
    if (!STAGED) {
        rec4[0] = make_float4(r[0], r[1], r[2], r[3]);
        rec4[1] = make_float4(r[4], r[5], r[6], r[7]);
        rec4[2] = make_float4(r[8], r[9], r[10], r[11]);
        rec4[3] = make_float4(r[12], r[13], r[14], r[15]);
    }
    } while (false);
    if (PHASE == 2 && phase2_done) return;
    if constexpr (STAGED) {
        // One wave per workgroup: LDS operations complete in program order, so every lane's coefficient reads are behind us.
        // Lane l parks its record at a 20-float stride (conflict-free ds_write_b128), then the wave writes the block's records
        // as 16-byte pieces in address order: piece q = 64 k + lane belongs to the record of lane q / 4.
        __builtin_amdgcn_wave_barrier();
        // skip_culled_records: the record of a culled Gaussian is all zeros (r[] was never filled) and nobody reads it
        const bool skip_culled = p.o.skip_culled_records != 0;
        const uint64_t rec_rows = __ballot(r[REC_OPAC] != 0.f || r[REC_CA] != 0.f);
        dns_v4f *park = reinterpret_cast<dns_v4f *>(sh_lds) + threadIdx.x * 5;
        park[0] = dns_v4f{r[0], r[1], r[2], r[3]};
        park[1] = dns_v4f{r[4], r[5], r[6], r[7]};
        park[2] = dns_v4f{r[8], r[9], r[10], r[11]};
        park[3] = dns_v4f{r[12], r[13], r[14], r[15]};
        __builtin_amdgcn_wave_barrier();
        const int g0 = vb * SH_STAGE_THREADS;
        const int pieces = 4 * min(SH_STAGE_THREADS, p.s.N - g0);
        dns_v4f *out = reinterpret_cast<dns_v4f *>(p.o.splats + (size_t)g0 * DNS_REC);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int q = k * SH_STAGE_THREADS + (int)threadIdx.x;
            const dns_v4f v = reinterpret_cast<const dns_v4f *>(sh_lds)[(q >> 2) * 5 + (q & 3)];
            if (q < pieces && (!skip_culled || ((rec_rows >> (q >> 2)) & 1ull))) {
#if DNS_PROJ_NT & 2
                __builtin_nontemporal_store(v, out + q);
#else
                out[q] = v;
#endif
            }
        }
    }
    };      // block_body
    if constexpr (PHASE == 2) {
        for (int vb = blockIdx.x; vb < p.blocks; vb += gridDim.x) {
            if (vb != (int)blockIdx.x) __syncthreads();      // the previous block's rows are still being read
            block_body(vb);
        }
    } else {
        block_body((int)blockIdx.x);
    }
}

struct BwdParams {
    dnsplat_scene s;
    dnsplat_camera c;
    dnsplat_proj_out o;
    dnsplat_proj_grads g;
};

// 169 VGPRs as hipcc allocates them freely = 2 waves / SIMD, one register over the 168 that allow 3: pinned to 3 (no scratch;
// DNS_PROJ_BWD_WAVES=0 lets the compiler choose).
#ifndef DNS_PROJ_BWD_WAVES
#define DNS_PROJ_BWD_WAVES 3
#endif
#if DNS_PROJ_BWD_WAVES
#define DNS_PROJ_BWD_OCC __attribute__((amdgpu_waves_per_eu(DNS_PROJ_BWD_WAVES, DNS_PROJ_BWD_WAVES)))
#else
#define DNS_PROJ_BWD_OCC
#endif
template <int L>
__global__ __launch_bounds__(SH_STAGE_THREADS) DNS_PROJ_BWD_OCC void project_bwd_kernel(BwdParams p)
{
    __shared__ float sh_lds[L == SH_DIRECT ? 1 : SH_STAGE_THREADS * ShRowTraits<L>::LDS_ROW];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int g0 = blockIdx.x * SH_STAGE_THREADS;
    const int nG = min(SH_STAGE_THREADS, p.s.N - g0);
    // as in the forward: the lane's radius and parameters are requested before the coefficient rows are staged — one round trip
    // instead of (rows | radius | parameters) in a row; the gradient record follows as soon as the radius says the Gaussian is
    // visible, and travels while project_one() re-derives the projection
    int32_t radius_g;
    float mean[3], quat[4], sc_raw[3], opac_in;
    {
        const int gc = min(g, p.s.N - 1);
        radius_g = p.g.radii[gc];
#pragma unroll
        for (int i = 0; i < 3; ++i) mean[i] = p.s.means[3 * gc + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) quat[i] = p.s.quats[4 * gc + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) sc_raw[i] = p.s.scales[3 * gc + i];
        opac_in = p.s.opacities[gc];
    }
    // the coefficient gradients are produced elsewhere (multi-view data parallelism): from the colour gradients dnsplat_sh_factors
    // took ahead of this launch, all-gathered and turned into rows by dnsplat_sh_grads_from_factors
    const bool sh_elsewhere = p.g.sh_grads_skip;
    // Row-selective staging (see sh_stage_in_rows): only the coefficient rows of the visible Gaussians are read, and — with
    // dnsplat_proj_grads.sh_zero_state — the zero gradient rows of Gaussians that were culled before and are culled again are not
    // written a second time.  One 64-bit word per workgroup (a wave) on each side.
    constexpr bool VISIBLE_ROWS = DNS_PROJ_VISIBLE_ROWS && L != SH_DIRECT && SH_STAGE_THREADS == DNS_WAVE;
    // DNS_PROJ_BWD_ROWS_IN: 1 = READ only the visible Gaussians' coefficient rows (the staging then waits for the radii: a dependent
    // round trip), 0 = the whole span is requested together with the radii, as before round 6.  Measured inside whole bench runs
    // (profiles/r06_ab_per_gaussian.txt): at 2 waves / SIMD the round trip costs more than the bytes save (+5 %), at 3 waves it pays
    // (C5: 0.474 -> 0.451 ms).
#ifndef DNS_PROJ_BWD_ROWS_IN
#define DNS_PROJ_BWD_ROWS_IN 1
#endif
    const uint64_t vis_rows = __ballot(g < p.s.N && radius_g > 0);
    uint64_t zero_known = 0ull;
    const bool track_zero = SH_STAGE_THREADS == DNS_WAVE && p.g.sh_zero_state != nullptr && !sh_elsewhere && p.s.sh_degree >= 0;
    if (track_zero) zero_known = p.g.sh_zero_state[blockIdx.x];
#ifndef DNS_PROJ_ZERO_ROWS
#define DNS_PROJ_ZERO_ROWS 0
#endif
    // every row of the workgroup is culled now and zero in memory already (wave-uniform): nothing of it needs writing
    const bool wg_stays_zero = track_zero && (vis_rows | ~zero_known) == 0ull;
    // DNS_PROJ_ZERO_ROWS = 1 (round 6, first form): rows skipped one by one — the workgroup's one streaming store becomes 16-byte
    // pieces with holes (partial lines) and the kernel gets 5-10 % SLOWER for 10 % fewer bytes (profiles/r06_ab_per_gaussian.txt).
    // 0: all or nothing per workgroup.  In a random row order every 64 rows hold a visible one and nothing is ever skipped; along a
    // Morton curve (densify.spatial_order) a camera's culled Gaussians are whole workgroups.
    const bool my_row_stays_zero = DNS_PROJ_ZERO_ROWS ? (((zero_known & ~vis_rows) >> threadIdx.x) & 1ull) != 0ull : wg_stays_zero;
    if (L != SH_DIRECT) {
        const float *base = (L == SH_CAT ? p.s.sh0 : p.s.shN) + (size_t)g0 * ShRowTraits<L>::ROW;
        if (VISIBLE_ROWS && DNS_PROJ_BWD_ROWS_IN) sh_stage_in_rows<L>(base, nG * ShRowTraits<L>::ROW, sh_lds, vis_rows);
        else sh_stage_in<L>(base, nG * ShRowTraits<L>::ROW, sh_lds);
        __syncthreads();
    }
    // in the staged layouts a lane turns its LDS row of coefficients into its row of coefficient gradients in
    // place; the block then stores the whole span with coalesced 16-byte writes
    float *lrow = sh_lds + (L == SH_DIRECT ? 0 : threadIdx.x * ShRowTraits<L>::LDS_ROW);
    float *lN = L == SH_CAT ? lrow + 3 : lrow;
    const float shs = (p.g.sh_grad_scale != 0.f) ? p.g.sh_grad_scale : 1.f;      // own-camera rows of a data-parallel step: x 1 / world
    if (g < p.s.N) {
    const int nbK = (p.s.sh_degree >= 0) ? (p.s.sh_degree + 1) * (p.s.sh_degree + 1) : 0;

    float v_mean[3] = {0.f, 0.f, 0.f}, v_quat[4] = {0.f, 0.f, 0.f, 0.f}, v_scale[3] = {0.f, 0.f, 0.f}, v_opac = 0.f;
    float fac3[3] = {0.f, 0.f, 0.f};          // dnsplat_proj_grads.sh_factors: the colour gradients behind the clamp
    const bool visible = radius_g > 0;
    // the gradient record of a visible Gaussian: requested here, used after project_one()
    float4 vrec[4];
    if (visible) {
        const float4 *vr4 = reinterpret_cast<const float4 *>(p.g.v_splats + (size_t)g * DNS_REC);
        vrec[0] = vr4[0]; vrec[1] = vr4[1]; vrec[2] = vr4[2]; vrec[3] = vr4[3];
    }

    // Gradient rows of culled Gaussians are zero; SH rows beyond the active degree are zero too.
    float *vsh0 = p.g.v_sh0 ? p.g.v_sh0 + (size_t)g * p.g.v_sh0_stride : nullptr;
    float *vshN = p.g.v_shN ? p.g.v_shN + (size_t)g * p.g.v_shN_stride : nullptr;
    const int restK = (p.g.v_shN && p.s.sh_K > 1) ? p.s.sh_K - 1 : 0;  // higher-band bases stored per Gaussian

    Cam cam;
    Proj st;
    float sc[3];
    bool ok = false;
    if (visible) {
        cam = load_cam(p.c.viewmat, p.c.K);
#pragma unroll
        for (int i = 0; i < 3; ++i) sc[i] = p.s.scales_are_log ? expf(sc_raw[i]) : sc_raw[i];
        ok = project_one(mean, quat, sc, cam, p.c.width, p.c.height, p.c.eps2d, p.c.near_plane, p.c.far_plane,
                         p.c.radius_clip, st);
    }

    if (!ok) {
        if (L == SH_CAT) { lrow[0] = 0.f; lrow[1] = 0.f; lrow[2] = 0.f; }
        else if (vsh0 && !sh_elsewhere && !my_row_stays_zero) { vsh0[0] = 0.f; vsh0[1] = 0.f; vsh0[2] = 0.f; }
        if (L != SH_DIRECT) { for (int k = 0; k < 45; ++k) lN[k] = 0.f; }
        else if (vshN && !sh_elsewhere && !my_row_stays_zero) for (int k = 0; k < 3 * restK; ++k) vshN[k] = 0.f;
        if (p.g.v_colors) for (int k = 0; k < p.s.n_colors; ++k) p.g.v_colors[(size_t)g * p.s.n_colors + k] = 0.f;
    } else {
        float vr[DNS_REC];
        {
            const float4 a = vrec[0], b = vrec[1], c = vrec[2], d = vrec[3];
            vr[0] = a.x; vr[1] = a.y; vr[2] = a.z; vr[3] = a.w;
            vr[4] = b.x; vr[5] = b.y; vr[6] = b.z; vr[7] = b.w;
            vr[8] = c.x; vr[9] = c.y; vr[10] = c.z; vr[11] = c.w;
            vr[12] = d.x; vr[13] = d.y; vr[14] = d.z; vr[15] = d.w;
        }
        float vx2 = vr[REC_X], vy2 = vr[REC_Y];
        if (p.g.v_means2d) { vx2 = p.g.v_means2d[2 * g]; vy2 = p.g.v_means2d[2 * g + 1]; }
        float vc[3] = {vr[REC_CA], vr[REC_CB], vr[REC_CC]};
        if (p.g.v_conics) { vc[0] += p.g.v_conics[3 * g]; vc[1] += p.g.v_conics[3 * g + 1]; vc[2] += p.g.v_conics[3 * g + 2]; }
        float v_depth = p.g.v_depths ? p.g.v_depths[g] : 0.f;
        float v_comp = p.g.v_compensations ? p.g.v_compensations[g] : 0.f;

        // ---- opacity (A0 / antialiasing)
        float opac_act = p.s.opacities_are_logit ? sigmoidf(opac_in) : opac_in;
        float v_o = vr[REC_OPAC];
        if (p.c.antialiased) { v_comp += v_o * opac_act; v_o *= st.compensation; }
        v_opac = p.s.opacities_are_logit ? v_o * opac_act * (1.f - opac_act) : v_o;

        // ---- feature channels
        int ch = 0;
        float v_R[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) v_R[i] = 0.f;
        if (p.s.sh_degree >= 0) {
            float dx = mean[0] - cam.pos[0], dy = mean[1] - cam.pos[1], dz = mean[2] - cam.pos[2];
            float inorm = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
            dx *= inorm; dy *= inorm; dz *= inorm;
            float bas[16];
            sh_basis(p.s.sh_degree, dx, dy, dz, bas);
            float vdn[3] = {0.f, 0.f, 0.f};
            float bx[16], by[16], bz[16];
            if (p.s.sh_degree >= 1) sh_basis_grad(p.s.sh_degree, dx, dy, dz, bx, by, bz);
            if (L == SH_DIRECT) {
                const float *c0 = p.s.sh0 + (size_t)g * p.s.sh0_stride;
                const float *cN = p.s.shN + (size_t)g * p.s.shN_stride;
                float col[3] = {bas[0] * c0[0], bas[0] * c0[1], bas[0] * c0[2]};
#pragma unroll
                for (int k = 1; k < 16; ++k)          // static indices keep bas[] / bx[] / by[] / bz[] in registers
                    if (k < nbK) {
                        col[0] += bas[k] * cN[3 * (k - 1) + 0];
                        col[1] += bas[k] * cN[3 * (k - 1) + 1];
                        col[2] += bas[k] * cN[3 * (k - 1) + 2];
                    }
                // clamp_min(c + 0.5, 0): gradient passes where c + 0.5 >= 0
                float vcol[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) vcol[i] = (col[i] + 0.5f >= 0.f) ? vr[REC_CH0 + i] : 0.f;
                fac3[0] = vcol[0]; fac3[1] = vcol[1]; fac3[2] = vcol[2];
                const float vcs[3] = {vcol[0] * shs, vcol[1] * shs, vcol[2] * shs};
                if (vsh0 && !sh_elsewhere) { vsh0[0] = bas[0] * vcs[0]; vsh0[1] = bas[0] * vcs[1]; vsh0[2] = bas[0] * vcs[2]; }
                if (vshN && !sh_elsewhere) {
#pragma unroll
                    for (int k = 1; k < 16; ++k)
                        if (k < nbK) {
                            vshN[3 * (k - 1) + 0] = bas[k] * vcs[0];
                            vshN[3 * (k - 1) + 1] = bas[k] * vcs[1];
                            vshN[3 * (k - 1) + 2] = bas[k] * vcs[2];
                        }
                    for (int k = 3 * (nbK - 1); k < 3 * restK; ++k) vshN[k] = 0.f;
                }
#pragma unroll
                for (int k = 1; k < 16; ++k)
                    if (k < nbK) {
                        float s = cN[3 * (k - 1)] * vcol[0] + cN[3 * (k - 1) + 1] * vcol[1] + cN[3 * (k - 1) + 2] * vcol[2];
                        vdn[0] += bx[k] * s; vdn[1] += by[k] * s; vdn[2] += bz[k] * s;
                    }
            } else {
                float c00, c01, c02;
                if (L == SH_CAT) { c00 = lrow[0]; c01 = lrow[1]; c02 = lrow[2]; }
                else { const float *c0 = p.s.sh0 + (size_t)g * p.s.sh0_stride; c00 = c0[0]; c01 = c0[1]; c02 = c0[2]; }
                float col[3] = {bas[0] * c00, bas[0] * c01, bas[0] * c02};
#pragma unroll
                for (int k = 1; k < 16; ++k)
                    if (k < nbK) {
                        col[0] += bas[k] * lN[3 * (k - 1) + 0];
                        col[1] += bas[k] * lN[3 * (k - 1) + 1];
                        col[2] += bas[k] * lN[3 * (k - 1) + 2];
                    }
                float vcol[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) vcol[i] = (col[i] + 0.5f >= 0.f) ? vr[REC_CH0 + i] : 0.f;
                fac3[0] = vcol[0]; fac3[1] = vcol[1]; fac3[2] = vcol[2];
                const float vcs[3] = {vcol[0] * shs, vcol[1] * shs, vcol[2] * shs};
                if (L == SH_CAT) { lrow[0] = bas[0] * vcs[0]; lrow[1] = bas[0] * vcs[1]; lrow[2] = bas[0] * vcs[2]; }
                else if (vsh0 && !sh_elsewhere) { vsh0[0] = bas[0] * vcs[0]; vsh0[1] = bas[0] * vcs[1]; vsh0[2] = bas[0] * vcs[2]; }
#pragma unroll
                for (int k = 1; k < 16; ++k)     // read the coefficient, then overwrite it with its gradient
                    if (k < nbK) {
                        const float a0 = lN[3 * (k - 1)], a1 = lN[3 * (k - 1) + 1], a2 = lN[3 * (k - 1) + 2];
                        const float s = a0 * vcol[0] + a1 * vcol[1] + a2 * vcol[2];
                        vdn[0] += bx[k] * s; vdn[1] += by[k] * s; vdn[2] += bz[k] * s;
                        lN[3 * (k - 1) + 0] = bas[k] * vcs[0];
                        lN[3 * (k - 1) + 1] = bas[k] * vcs[1];
                        lN[3 * (k - 1) + 2] = bas[k] * vcs[2];
                    }
                for (int k = 3 * (nbK - 1); k < 45; ++k) lN[k] = 0.f;
            }
            if (p.s.sh_degree >= 1) {
                float dot = vdn[0] * dx + vdn[1] * dy + vdn[2] * dz;
                v_mean[0] += (vdn[0] - dot * dx) * inorm;
                v_mean[1] += (vdn[1] - dot * dy) * inorm;
                v_mean[2] += (vdn[2] - dot * dz) * inorm;
            }
            ch = 3;
        } else {
            if (p.g.v_colors)
#pragma unroll
                for (int k = 0; k < DNS_MAX_CH; ++k)
                    if (k < p.s.n_colors) {
                        float vck = vr[REC_CH0 + k];
                        if (p.s.colors_are_logit) {      // through sigmoid(): s (1 - s), s re-derived from the logit as the forward did
                            const float sg = sigmoidf(p.s.colors[(size_t)g * p.s.n_colors + k]);
                            vck = vck * sg * (1.f - sg);
                        }
                        p.g.v_colors[(size_t)g * p.s.n_colors + k] = vck;
                    }
            ch = p.s.n_colors;
        }
        if (p.o.with_depth_channel) { v_depth += rec_get_ch(vr, ch); ch += 1; }
        if (p.o.with_normal_channels) {
            // n_cam = Mn * (sgn * col/|col|), col = Rq[:,k]  ->  only the quaternion receives gradient
            const float *Mn = p.c.normal_frame;
            float campos[3] = {Mn[9], Mn[10], Mn[11]};
            float n[3], nrm, sgn; int k;
            gaussian_normal(st.Rq, sc_raw, mean, campos, n, k, nrm, sgn);
            float vn[3];
#pragma unroll
            for (int i = 0; i < 3; ++i)
                vn[i] = sgn * (Mn[0 + i] * rec_get_ch(vr, ch) + Mn[3 + i] * rec_get_ch(vr, ch + 1) + Mn[6 + i] * rec_get_ch(vr, ch + 2));
            float u[3] = {n[0] * sgn, n[1] * sgn, n[2] * sgn};
            float d = u[0] * vn[0] + u[1] * vn[1] + u[2] * vn[2];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const float add = (vn[i] - u[i] * d) / nrm;
#pragma unroll
                for (int j = 0; j < 3; ++j) v_R[3 * i + j] += (j == k) ? add : 0.f;      // static indices: v_R stays in registers
            }
        }

        // ---- conic -> cov2d (A.8)
        float a = st.conic[0], b = st.conic[1], c = st.conic[2];
        float ga = vc[0], gb = 0.5f * vc[1], gc = vc[2];
        float t00 = a * ga + b * gb, t01 = a * gb + b * gc;
        float t10 = b * ga + c * gb, t11 = b * gb + c * gc;
        float G2[4];
        G2[0] = -(t00 * a + t01 * b);
        G2[1] = -(t00 * b + t01 * c);
        G2[2] = -(t10 * a + t11 * b);
        G2[3] = -(t10 * b + t11 * c);
        if (v_comp != 0.f && st.compensation > 0.f) {
            float comp = st.compensation;
            float inv_db = 1.f / st.det_blur;
            float kk = v_comp * 0.5f / comp;
            float one_minus = 1.f - comp * comp;
            float d00 = (st.cov2d[2] - comp * comp * st.cov2d_blur[2]) * inv_db;
            float d11 = (st.cov2d[0] - comp * comp * st.cov2d_blur[0]) * inv_db;
            float d01 = (-2.f * st.cov2d[1] * one_minus) * inv_db;
            G2[0] += kk * d00; G2[3] += kk * d11;
            G2[1] += 0.5f * kk * d01; G2[2] += 0.5f * kk * d01;
        }
        const float *J = st.J; const float *Cc = st.covar_c;
        float GJ[6], GtJ[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            GJ[j] = G2[0] * J[j] + G2[1] * J[3 + j];
            GJ[3 + j] = G2[2] * J[j] + G2[3] * J[3 + j];
            GtJ[j] = G2[0] * J[j] + G2[2] * J[3 + j];
            GtJ[3 + j] = G2[1] * J[j] + G2[3] * J[3 + j];
        }
        float v_Cc[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) v_Cc[3 * i + j] = J[i] * GJ[j] + J[3 + i] * GJ[3 + j];
        float v_J[6];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    s1 += GJ[3 * r + k] * Cc[3 * j + k];
                    s2 += GtJ[3 * r + k] * Cc[3 * k + j];
                }
                v_J[3 * r + j] = s1 + s2;
            }
        float fx = cam.fx, fy = cam.fy, rz = st.rz, rz2 = st.rz2, rz3 = st.rz2 * st.rz;
        float x = st.mean_c[0], y = st.mean_c[1];
        float v_mc[3];
        v_mc[0] = fx * rz * vx2;
        v_mc[1] = fy * rz * vy2;
        v_mc[2] = -(fx * x * vx2 + fy * y * vy2) * rz2;
        if (st.x_in) v_mc[0] += -fx * rz2 * v_J[2];
        else v_mc[2] += -fx * rz3 * v_J[2] * st.tx;
        if (st.y_in) v_mc[1] += -fy * rz2 * v_J[5];
        else v_mc[2] += -fy * rz3 * v_J[5] * st.ty;
        v_mc[2] += -fx * rz2 * v_J[0] - fy * rz2 * v_J[4] + 2.f * fx * st.tx * rz3 * v_J[2] + 2.f * fy * st.ty * rz3 * v_J[5];
        v_mc[2] += v_depth;

        const float *Rv = cam.Rv;
#pragma unroll
        for (int i = 0; i < 3; ++i) v_mean[i] += Rv[0 + i] * v_mc[0] + Rv[3 + i] * v_mc[1] + Rv[6 + i] * v_mc[2];
        float tmp[9], v_cov[9];
        mm3_atb(Rv, v_Cc, tmp);
        mm3(tmp, Rv, v_cov);
        float sym[9], v_M[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) sym[3 * i + j] = v_cov[3 * i + j] + v_cov[3 * j + i];
        mm3(sym, st.M, v_M);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) v_R[3 * i + j] += v_M[3 * i + j] * sc[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float vs = st.Rq[0 + j] * v_M[0 + j] + st.Rq[3 + j] * v_M[3 + j] + st.Rq[6 + j] * v_M[6 + j];
            v_scale[j] = p.s.scales_are_log ? vs * sc[j] : vs;
        }
        float w = st.qn[0], qx = st.qn[1], qy = st.qn[2], qz = st.qn[3];
        float vqn[4];
        vqn[0] = 2.f * (qx * (v_R[7] - v_R[5]) + qy * (v_R[2] - v_R[6]) + qz * (v_R[3] - v_R[1]));
        vqn[1] = 2.f * (-2.f * qx * (v_R[4] + v_R[8]) + qy * (v_R[3] + v_R[1]) + qz * (v_R[6] + v_R[2]) + w * (v_R[7] - v_R[5]));
        vqn[2] = 2.f * (qx * (v_R[3] + v_R[1]) - 2.f * qy * (v_R[0] + v_R[8]) + qz * (v_R[7] + v_R[5]) + w * (v_R[2] - v_R[6]));
        vqn[3] = 2.f * (qx * (v_R[6] + v_R[2]) + qy * (v_R[7] + v_R[5]) - 2.f * qz * (v_R[0] + v_R[4]) + w * (v_R[3] - v_R[1]));
        float dot = vqn[0] * w + vqn[1] * qx + vqn[2] * qy + vqn[3] * qz;
#pragma unroll
        for (int i = 0; i < 4; ++i) v_quat[i] = (vqn[i] - dot * st.qn[i]) * st.inv_norm;
    }

    if (!(wg_stays_zero && p.g.zero_state_geometry)) {     // dnsplat_proj_grads.zero_state_geometry: these rows are tracked too
    p.g.v_means[3 * g] = v_mean[0]; p.g.v_means[3 * g + 1] = v_mean[1]; p.g.v_means[3 * g + 2] = v_mean[2];
    p.g.v_quats[4 * g] = v_quat[0]; p.g.v_quats[4 * g + 1] = v_quat[1]; p.g.v_quats[4 * g + 2] = v_quat[2]; p.g.v_quats[4 * g + 3] = v_quat[3];
    p.g.v_scales[3 * g] = v_scale[0]; p.g.v_scales[3 * g + 1] = v_scale[1]; p.g.v_scales[3 * g + 2] = v_scale[2];
    p.g.v_opacities[g] = v_opac;
    }
    if (p.g.sh_packed) {
        // packed slab (dnsplat_visible_index wrote header, masks and offsets from the same radii): rows of the visible Gaussians only
        const uint32_t *hdr = reinterpret_cast<const uint32_t *>(p.g.sh_packed);
        const uint32_t cap = hdr[4];
        const int nb = (p.s.N + 63) >> 6;
        const uint32_t *offs = hdr + 8 + 2 * nb;
        float *rows = p.g.sh_packed + dns_packed_rows_offset(nb);
        if (visible) {
            const uint32_t k = offs[blockIdx.x] + (uint32_t)__popcll(vis_rows & ((1ull << threadIdx.x) - 1ull));
            if (k < cap) { rows[3 * (size_t)k] = fac3[0]; rows[3 * (size_t)k + 1] = fac3[1]; rows[3 * (size_t)k + 2] = fac3[2]; }
        }
    }
    if (p.g.sh_factors) {
        // the slab of dnsplat_sh_factors, from what this lane holds anyway (one launch and two record lines per Gaussian less)
        float *f = p.g.sh_factors + 3 * (size_t)g;
        f[0] = fac3[0]; f[1] = fac3[1]; f[2] = fac3[2];
        if (g == 0) {
            const Cam cc = load_cam(p.c.viewmat, p.c.K);
            float *tail = p.g.sh_factors + 3 * (size_t)p.s.N;
            tail[0] = cc.pos[0]; tail[1] = cc.pos[1]; tail[2] = cc.pos[2]; tail[3] = 0.f;
        }
    }
    }  // g < N
    if (L != SH_DIRECT && !p.g.sh_grads_skip) {
        __syncthreads();
        float *base = (L == SH_CAT ? p.g.v_sh0 : p.g.v_shN) + (size_t)g0 * ShRowTraits<L>::ROW;
        // rows to store: the visible ones and the culled ones whose memory is not known to be zero (see wg_stays_zero)
        if (VISIBLE_ROWS && track_zero && DNS_PROJ_ZERO_ROWS) sh_stage_out_rows<L>(base, nG * ShRowTraits<L>::ROW, sh_lds, vis_rows | ~zero_known);
        else if (VISIBLE_ROWS && wg_stays_zero) { /* nothing */ }
        else sh_stage_out<L>(base, nG * ShRowTraits<L>::ROW, sh_lds);
    }
    // what memory holds now: zero rows exactly where the Gaussian is culled (lanes beyond N: bits unused)
    if (track_zero && threadIdx.x == 0) p.g.sh_zero_state[blockIdx.x] = ~vis_rows;
}

__global__ __launch_bounds__(256) void pack_splats_kernel(int N, const float *__restrict__ means2d,
                                                          const float *__restrict__ conics,
                                                          const float *__restrict__ opacities,
                                                          const float *__restrict__ colors, int C,
                                                          float *__restrict__ splats)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    float r[DNS_REC];
#pragma unroll
    for (int i = 0; i < DNS_REC; ++i) r[i] = 0.f;
    r[REC_X] = means2d[2 * g]; r[REC_Y] = means2d[2 * g + 1];
    r[REC_CA] = conics[3 * g]; r[REC_CB] = conics[3 * g + 1]; r[REC_CC] = conics[3 * g + 2];
    r[REC_OPAC] = opacities[g];
    for (int k = 0; k < C; ++k) r[REC_CH0 + k] = colors[(size_t)g * C + k];
    float4 *rec4 = reinterpret_cast<float4 *>(splats + (size_t)g * DNS_REC);
    rec4[0] = make_float4(r[0], r[1], r[2], r[3]);
    rec4[1] = make_float4(r[4], r[5], r[6], r[7]);
    rec4[2] = make_float4(r[8], r[9], r[10], r[11]);
    rec4[3] = make_float4(r[12], r[13], r[14], r[15]);
}

// v_coeff = scale * sum over views of basis(dir_view) (x) v_colour_view  — see dnsplat_sh_grads_from_factors.
// PACKED: the views' slabs hold the rows of their visible Gaussians only (dnsplat_visible_index); `slab_words` apart.
// ADD: the rows already hold view `skip_view`'s (pre-scaled) share, the other views are added in place (dnsplat_sh_grads_add_factors).
template <int L, bool PACKED, bool ADD>
__global__ __launch_bounds__(SH_STAGE_THREADS) void sh_from_factors_kernel(int N, int n_views, int skip_view, const float *__restrict__ factors,
                                                              size_t slab_words, const float *__restrict__ means, int degree,
                                                              float scale, float *__restrict__ v_sh0, int s0,
                                                              float *__restrict__ v_shN, int sN, int restK)
{
    __shared__ float sh_lds[L == SH_DIRECT ? 1 : SH_STAGE_THREADS * ShRowTraits<L>::LDS_ROW];
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int g0 = blockIdx.x * SH_STAGE_THREADS;
    const int nG = min(SH_STAGE_THREADS, N - g0);
    const int nb = (degree + 1) * (degree + 1);
    const int nblk = (N + 63) >> 6;
    float acc[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) acc[i] = 0.f;
    bool any = false;
    if (g < N) {
        const float mx = means[3 * g], my = means[3 * g + 1], mz = means[3 * g + 2];
        for (int v = 0; v < n_views; ++v) {
            if (v == skip_view) continue;
            const float *f = factors + (size_t)v * slab_words;
            float c0, c1, c2;
            const float *pos;
            if (PACKED) {
                const uint32_t *hdr = reinterpret_cast<const uint32_t *>(f);
                const uint64_t word = reinterpret_cast<const uint64_t *>(hdr + 8)[blockIdx.x];
                if (!((word >> threadIdx.x) & 1ull)) continue;
                const uint32_t k = (hdr + 8 + 2 * nblk)[blockIdx.x] + (uint32_t)__popcll(word & ((1ull << threadIdx.x) - 1ull));
                if (k >= hdr[4]) continue;                     // beyond the slab's capacity: dropped by the sender (it reports the overflow)
                const float *row = f + dns_packed_rows_offset(nblk) + 3 * (size_t)k;
                c0 = row[0]; c1 = row[1]; c2 = row[2];
                pos = f + 1;
            } else {
                c0 = f[3 * (size_t)g]; c1 = f[3 * (size_t)g + 1]; c2 = f[3 * (size_t)g + 2];
                pos = f + (size_t)3 * N;
            }
            if (c0 == 0.f && c1 == 0.f && c2 == 0.f) continue;
            any = true;
            // the view direction of camera v, re-derived as the projection kernels of rank v derived it (same operations)
            float dx = mx - pos[0], dy = my - pos[1], dz = mz - pos[2];
            const float inorm = 1.f / sqrtf(dx * dx + dy * dy + dz * dz);
            dx *= inorm; dy *= inorm; dz *= inorm;
            float bas[16];
            sh_basis(degree, dx, dy, dz, bas);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < nb) {
                    acc[3 * k + 0] += bas[k] * c0;
                    acc[3 * k + 1] += bas[k] * c1;
                    acc[3 * k + 2] += bas[k] * c2;
                }
        }
    }
    if (L == SH_DIRECT) {
        if (g >= N || (ADD && !any)) return;
        float *r0 = v_sh0 + (size_t)g * s0;
        if (ADD) { r0[0] += scale * acc[0]; r0[1] += scale * acc[1]; r0[2] += scale * acc[2]; }
        else { r0[0] = scale * acc[0]; r0[1] = scale * acc[1]; r0[2] = scale * acc[2]; }
        if (v_shN) {
            float *rN = v_shN + (size_t)g * sN;
            if (ADD) { for (int k = 0; k < 3 * restK && k < 45; ++k) rN[k] += scale * acc[3 + k]; }
            else for (int k = 0; k < 3 * restK; ++k) rN[k] = k < 45 ? scale * acc[3 + k] : 0.f;
        }
        return;
    }
    float *base = (L == SH_CAT ? v_sh0 : v_shN) + (size_t)g0 * ShRowTraits<L>::ROW;
    const uint64_t touched = __ballot(any);
    if (ADD) {
        if (touched == 0ull) return;                         // nobody else saw this block's Gaussians: the own rows stand (wave-uniform)
        sh_stage_in_rows<L>(base, nG * ShRowTraits<L>::ROW, sh_lds, touched);
        __syncthreads();
    }
    if (g < N) {
        float *lrow = sh_lds + threadIdx.x * ShRowTraits<L>::LDS_ROW;
        if (L == SH_CAT) {
            if (ADD) {
                if (any) {
#pragma unroll
                    for (int k = 0; k < 48; ++k) lrow[k] += scale * acc[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 48; ++k) lrow[k] = scale * acc[k];
            }
        } else {
            float *r0 = v_sh0 + (size_t)g * s0;
            if (ADD) {
                if (any) {
                    r0[0] += scale * acc[0]; r0[1] += scale * acc[1]; r0[2] += scale * acc[2];
#pragma unroll
                    for (int k = 0; k < 45; ++k) lrow[k] += scale * acc[3 + k];
                }
            } else {
                r0[0] = scale * acc[0]; r0[1] = scale * acc[1]; r0[2] = scale * acc[2];
#pragma unroll
                for (int k = 0; k < 45; ++k) lrow[k] = scale * acc[3 + k];
            }
        }
    }
    __syncthreads();
    if (ADD) sh_stage_out_rows<L>(base, nG * ShRowTraits<L>::ROW, sh_lds, touched);
    else sh_stage_out<L>(base, nG * ShRowTraits<L>::ROW, sh_lds);
}

// dnsplat_visible_index, launch 1: one wave per block of 64 Gaussians -> mask word + popcount; thread 0 of the grid writes the header
__global__ __launch_bounds__(256) void visible_mask_kernel(int N, uint32_t capacity, const int32_t *__restrict__ radii,
                                                           const float *__restrict__ viewmat, float *__restrict__ slab,
                                                           uint32_t *__restrict__ counts)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int nblk = (N + 63) >> 6;
    const bool vis = g < N && radii[g] > 0;
    const uint64_t word = __ballot(vis);
    const int b = g >> 6;
    if ((threadIdx.x & 63) == 0 && b < nblk) {
        reinterpret_cast<uint64_t *>(reinterpret_cast<uint32_t *>(slab) + 8)[b] = word;
        counts[b] = (uint32_t)__popcll(word);
    }
    if (g == 0) {
        float Rv[9], t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) Rv[3 * i + j] = viewmat[4 * i + j];
            t[i] = viewmat[4 * i + 3];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) slab[1 + i] = -(Rv[0 + i] * t[0] + Rv[3 + i] * t[1] + Rv[6 + i] * t[2]);   // = load_cam().pos
        uint32_t *hdr = reinterpret_cast<uint32_t *>(slab);
        hdr[4] = capacity; hdr[5] = (uint32_t)N; hdr[6] = 0u; hdr[7] = 0u;
    }
}

// launch 2: exclusive prefix sum of the block popcounts (one workgroup; 78 k words at 5 M Gaussians) + the total
__global__ __launch_bounds__(1024) void visible_scan_kernel(int nblk, const uint32_t *__restrict__ counts, float *__restrict__ slab)
{
    __shared__ uint32_t part[1024];
    uint32_t *hdr = reinterpret_cast<uint32_t *>(slab);
    uint32_t *offs = hdr + 8 + 2 * nblk;
    const int per = (nblk + 1023) / 1024;
    const int b0 = threadIdx.x * per, b1 = min(nblk, b0 + per);
    uint32_t sum = 0;
    for (int b = b0; b < b1; ++b) sum += counts[b];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const uint32_t v = threadIdx.x >= (unsigned)d ? part[threadIdx.x - d] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (int b = b0; b < b1; ++b) { offs[b] = run; run += counts[b]; }
    if (threadIdx.x == 1023) hdr[0] = part[1023];
}

}  // namespace

// Which coefficient layout the (band-0, higher-bands) pointer pair describes; SH_DIRECT = anything the staged
// kernels do not cover (no SH, K != 16, unusual strides or an unaligned base).
static int sh_layout(const dnsplat_scene *s, const float *b0, int s0, const float *bN, int sN)
{
    if (s->sh_degree < 0 || s->sh_K != 16 || !b0 || !bN) return SH_DIRECT;
    if (bN == b0 + 3 && s0 == 48 && sN == 48 && ((uintptr_t)b0 & 15) == 0) return SH_CAT;
    if (s0 == 3 && sN == 45 && ((uintptr_t)bN & 15) == 0) return SH_SPLIT;
    return SH_DIRECT;
}

static int check_scene(const dnsplat_scene *s, const dnsplat_camera *c, const dnsplat_proj_out *o)
{
    if (!s || !c || !o) return DNSPLAT_ERR_INVALID_ARG;
    if (s->N < 0) return DNSPLAT_ERR_INVALID_ARG;
    if (s->N == 0) return DNSPLAT_OK;
    if (!s->means || !s->quats || !s->scales || !s->opacities) return DNSPLAT_ERR_INVALID_ARG;
    if (!c->viewmat || !c->K || c->width <= 0 || c->height <= 0 || c->tile_size <= 0) return DNSPLAT_ERR_INVALID_ARG;
    if (s->sh_degree > 3) return DNSPLAT_ERR_UNSUPPORTED;
    int ch;
    if (s->sh_degree >= 0) {
        if (!s->sh0 || (s->sh_degree > 0 && !s->shN)) return DNSPLAT_ERR_INVALID_ARG;
        if (s->sh_K < (s->sh_degree + 1) * (s->sh_degree + 1)) return DNSPLAT_ERR_INVALID_ARG;
        ch = 3;
    } else {
        if (s->n_colors < 0 || (s->n_colors > 0 && !s->colors)) return DNSPLAT_ERR_INVALID_ARG;
        ch = s->n_colors;
    }
    ch += (o->with_depth_channel ? 1 : 0) + (o->with_normal_channels ? 3 : 0);
    if (ch > DNSPLAT_MAX_CHANNELS) return DNSPLAT_ERR_UNSUPPORTED;
    if (o->with_normal_channels && !c->normal_frame) return DNSPLAT_ERR_INVALID_ARG;
    if (!o->radii || !o->means2d || !o->depths || !o->conics || !o->tiles_per_gauss || !o->splats)
        return DNSPLAT_ERR_INVALID_ARG;
    if (c->tight_tiles && !o->tiles_bin) return DNSPLAT_ERR_INVALID_ARG;
    return DNSPLAT_OK;
}

extern "C" int dnsplat_project_fwd(const dnsplat_scene *scene, const dnsplat_camera *cam,
                                   const dnsplat_proj_out *out, dnsplat_stream_t stream)
{
    int rc = check_scene(scene, cam, out);
    if (rc != DNSPLAT_OK) return rc;
    if (scene->N == 0) return DNSPLAT_OK;
    // tile_boxes packs the box as width | height << 16: a tile grid beyond 65535 x 65535 tiles (a megapixel-wide image) does not fit
    if (out->tile_boxes && cam->tile_size > 0 &&
        ((cam->width + cam->tile_size - 1) / cam->tile_size > 0xffff || (cam->height + cam->tile_size - 1) / cam->tile_size > 0xffff))
        return DNSPLAT_ERR_UNSUPPORTED;
    FwdParams p{*scene, *cam, *out, (scene->N + SH_STAGE_THREADS - 1) / SH_STAGE_THREADS};
    dim3 block(SH_STAGE_THREADS), grid((scene->N + SH_STAGE_THREADS - 1) / SH_STAGE_THREADS);
    if (out->phase < 0 || out->phase > 2 || (out->phase != 0 && scene->sh_degree < 0)) return DNSPLAT_ERR_INVALID_ARG;
    if (out->phase == 1) {
        hipLaunchKernelGGL((project_fwd_kernel<SH_DIRECT, 1>), grid, block, 0, (hipStream_t)stream, p);
    } else if (out->phase == 2) {
        const char *e = getenv("DNSPLAT_COLOUR_WGS");
        grid = dim3(min((int)grid.x, e ? atoi(e) : 512));
        switch (sh_layout(scene, scene->sh0, scene->sh0_stride, scene->shN, scene->shN_stride)) {
            case SH_SPLIT: hipLaunchKernelGGL((project_fwd_kernel<SH_SPLIT, 2>), grid, block, 0, (hipStream_t)stream, p); break;
            case SH_CAT: hipLaunchKernelGGL((project_fwd_kernel<SH_CAT, 2>), grid, block, 0, (hipStream_t)stream, p); break;
            default: hipLaunchKernelGGL((project_fwd_kernel<SH_DIRECT, 2>), grid, block, 0, (hipStream_t)stream, p);
        }
    } else {
        switch (sh_layout(scene, scene->sh0, scene->sh0_stride, scene->shN, scene->shN_stride)) {
            case SH_SPLIT: hipLaunchKernelGGL(project_fwd_kernel<SH_SPLIT>, grid, block, 0, (hipStream_t)stream, p); break;
            case SH_CAT: hipLaunchKernelGGL(project_fwd_kernel<SH_CAT>, grid, block, 0, (hipStream_t)stream, p); break;
            default: hipLaunchKernelGGL(project_fwd_kernel<SH_DIRECT>, grid, block, 0, (hipStream_t)stream, p);
        }
    }
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_project_bwd(const dnsplat_scene *scene, const dnsplat_camera *cam,
                                   const dnsplat_proj_out *fwd, const dnsplat_proj_grads *grads,
                                   dnsplat_stream_t stream)
{
    if (!scene || !cam || !fwd || !grads) return DNSPLAT_ERR_INVALID_ARG;
    if (scene->N == 0) return DNSPLAT_OK;
    if (!grads->radii || !grads->v_splats || !grads->v_means || !grads->v_quats || !grads->v_scales ||
        !grads->v_opacities)
        return DNSPLAT_ERR_INVALID_ARG;
    if (scene->sh_degree > 3) return DNSPLAT_ERR_UNSUPPORTED;
    if (fwd->with_normal_channels && !cam->normal_frame) return DNSPLAT_ERR_INVALID_ARG;
    BwdParams p{*scene, *cam, *fwd, *grads};
    dim3 block(SH_STAGE_THREADS), grid((scene->N + SH_STAGE_THREADS - 1) / SH_STAGE_THREADS);
    // the staged kernels need the gradient tensors in the same layout as the coefficients
    int layout = sh_layout(scene, scene->sh0, scene->sh0_stride, scene->shN, scene->shN_stride);
    if (layout != SH_DIRECT && layout != sh_layout(scene, grads->v_sh0, grads->v_sh0_stride, grads->v_shN, grads->v_shN_stride))
        layout = SH_DIRECT;
    switch (layout) {
        case SH_SPLIT: hipLaunchKernelGGL(project_bwd_kernel<SH_SPLIT>, grid, block, 0, (hipStream_t)stream, p); break;
        case SH_CAT: hipLaunchKernelGGL(project_bwd_kernel<SH_CAT>, grid, block, 0, (hipStream_t)stream, p); break;
        default: hipLaunchKernelGGL(project_bwd_kernel<SH_DIRECT>, grid, block, 0, (hipStream_t)stream, p);
    }
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

namespace {
// The part of a camera's SH-coefficient gradient that has to travel (dnsplat.h): the clamp-masked colour gradient of every
// Gaussian, from the forward's records ("the clamped colour in the record is positive") and the gradient records, followed by
// the camera position the receiving ranks re-derive the view directions from.
__global__ __launch_bounds__(256) void sh_factors_kernel(int N, const int32_t *__restrict__ radii, const float *__restrict__ viewmat,
                                                         const float *__restrict__ splats, const float *__restrict__ v_splats,
                                                         float *__restrict__ factors)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g == 0) {
        // camera centre = -R^T t of the world->camera matrix: exactly what load_cam() hands the projection kernels
        float Rv[9], t[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) Rv[3 * i + j] = viewmat[4 * i + j];
            t[i] = viewmat[4 * i + 3];
        }
        float *tail = factors + (size_t)3 * N;
#pragma unroll
        for (int i = 0; i < 3; ++i) tail[i] = -(Rv[0 + i] * t[0] + Rv[3 + i] * t[1] + Rv[6 + i] * t[2]);
        tail[3] = 0.f;
    }
    if (g >= N) return;
    float f[3] = {0.f, 0.f, 0.f};
    if (radii[g] > 0) {
        const float *rec = splats + (size_t)g * DNS_REC + REC_CH0;
        const float *vr = v_splats + (size_t)g * DNS_REC + REC_CH0;
#pragma unroll
        for (int i = 0; i < 3; ++i) f[i] = rec[i] > 0.f ? vr[i] : 0.f;
    }
    factors[3 * (size_t)g] = f[0]; factors[3 * (size_t)g + 1] = f[1]; factors[3 * (size_t)g + 2] = f[2];
}
}  // namespace

extern "C" int dnsplat_sh_factors(int32_t N, const int32_t *radii, const float *viewmat, const float *splats,
                                  const float *v_splats, float *factors, dnsplat_stream_t stream)
{
    if (N < 0) return DNSPLAT_ERR_INVALID_ARG;
    if (!viewmat || !factors) return DNSPLAT_ERR_INVALID_ARG;
    if (N > 0 && (!radii || !splats || !v_splats)) return DNSPLAT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sh_factors_kernel, dim3((N + 256) / 256), dim3(256), 0, (hipStream_t)stream, N, radii, viewmat, splats,
                       v_splats, factors);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

template <bool PACKED, bool ADD>
static int launch_sh_from_factors(int32_t N, int32_t n_views, int32_t skip_view, const float *factors, size_t slab_words, const float *means,
                                  int32_t sh_degree, int32_t sh_K, float scale, float *v_sh0, int32_t v_sh0_stride, float *v_shN,
                                  int32_t v_shN_stride, dnsplat_stream_t stream)
{
    if (N < 0 || n_views < 1 || sh_degree < 0 || sh_degree > 3 || sh_K < (sh_degree + 1) * (sh_degree + 1)) return DNSPLAT_ERR_INVALID_ARG;
    if (ADD && (skip_view < 0 || skip_view >= n_views)) return DNSPLAT_ERR_INVALID_ARG;
    if (N == 0 || (ADD && n_views == 1)) return DNSPLAT_OK;         // a single view: its rows are complete, nothing is launched
    if (!factors || !means || !v_sh0 || (sh_K > 1 && !v_shN)) return DNSPLAT_ERR_INVALID_ARG;
    dnsplat_scene fake{};
    fake.sh_degree = sh_degree; fake.sh_K = sh_K;
    const int layout = sh_layout(&fake, v_sh0, v_sh0_stride, v_shN, v_shN_stride);
    dim3 block(SH_STAGE_THREADS), grid((N + SH_STAGE_THREADS - 1) / SH_STAGE_THREADS);
    const int restK = sh_K - 1;
    const int skip = ADD ? skip_view : -1;
    switch (layout) {
        case SH_SPLIT:
            hipLaunchKernelGGL((sh_from_factors_kernel<SH_SPLIT, PACKED, ADD>), grid, block, 0, (hipStream_t)stream, N, n_views, skip, factors,
                               slab_words, means, sh_degree, scale, v_sh0, v_sh0_stride, v_shN, v_shN_stride, restK);
            break;
        case SH_CAT:
            hipLaunchKernelGGL((sh_from_factors_kernel<SH_CAT, PACKED, ADD>), grid, block, 0, (hipStream_t)stream, N, n_views, skip, factors,
                               slab_words, means, sh_degree, scale, v_sh0, v_sh0_stride, v_shN, v_shN_stride, restK);
            break;
        default:
            hipLaunchKernelGGL((sh_from_factors_kernel<SH_DIRECT, PACKED, ADD>), grid, block, 0, (hipStream_t)stream, N, n_views, skip, factors,
                               slab_words, means, sh_degree, scale, v_sh0, v_sh0_stride, v_shN, v_shN_stride, restK);
    }
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_sh_grads_from_factors(int32_t N, int32_t n_views, const float *factors, const float *means,
                                             int32_t sh_degree, int32_t sh_K,
                                             float scale, float *v_sh0, int32_t v_sh0_stride, float *v_shN,
                                             int32_t v_shN_stride, dnsplat_stream_t stream)
{
    return launch_sh_from_factors<false, false>(N, n_views, -1, factors, (size_t)3 * (size_t)(N > 0 ? N : 0) + 4, means, sh_degree, sh_K, scale,
                                                v_sh0, v_sh0_stride, v_shN, v_shN_stride, stream);
}

extern "C" int dnsplat_sh_grads_add_factors(int32_t N, int32_t n_views, int32_t skip_view, const float *factors, const float *means,
                                            int32_t sh_degree, int32_t sh_K, float scale, float *v_sh0, int32_t v_sh0_stride,
                                            float *v_shN, int32_t v_shN_stride, dnsplat_stream_t stream)
{
    return launch_sh_from_factors<false, true>(N, n_views, skip_view, factors, (size_t)3 * (size_t)(N > 0 ? N : 0) + 4, means, sh_degree, sh_K,
                                               scale, v_sh0, v_sh0_stride, v_shN, v_shN_stride, stream);
}

extern "C" size_t dnsplat_packed_slab_floats(int32_t N, int32_t capacity)
{
    if (N < 0 || capacity < 0) return 0;
    return dns_packed_slab_words((N + 63) >> 6, capacity);
}

extern "C" int dnsplat_visible_index(int32_t N, int32_t capacity, const int32_t *radii, const float *viewmat, float *slab,
                                     uint32_t *scratch, dnsplat_stream_t stream)
{
    if (N < 0 || capacity < 0 || !viewmat || !slab) return DNSPLAT_ERR_INVALID_ARG;
    if (N > 0 && (!radii || !scratch)) return DNSPLAT_ERR_INVALID_ARG;
    const int nblk = (N + 63) >> 6;
    hipLaunchKernelGGL(visible_mask_kernel, dim3((N + 256) / 256), dim3(256), 0, (hipStream_t)stream, N, (uint32_t)capacity, radii, viewmat,
                       slab, scratch);
    DNS_CHECK_LAUNCH();
    hipLaunchKernelGGL(visible_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nblk, scratch, slab);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_sh_grads_from_packed(int32_t N, int32_t capacity, int32_t n_views, int32_t skip_view, const float *slabs,
                                            const float *means, int32_t sh_degree, int32_t sh_K, float scale, float *v_sh0,
                                            int32_t v_sh0_stride, float *v_shN, int32_t v_shN_stride, dnsplat_stream_t stream)
{
    if (capacity < 0) return DNSPLAT_ERR_INVALID_ARG;
    const size_t words = dns_packed_slab_words(((N > 0 ? N : 0) + 63) >> 6, capacity);
    if (skip_view >= 0)
        return launch_sh_from_factors<true, true>(N, n_views, skip_view, slabs, words, means, sh_degree, sh_K, scale, v_sh0, v_sh0_stride, v_shN,
                                                  v_shN_stride, stream);
    return launch_sh_from_factors<true, false>(N, n_views, -1, slabs, words, means, sh_degree, sh_K, scale, v_sh0, v_sh0_stride, v_shN,
                                               v_shN_stride, stream);
}

extern "C" int dnsplat_pack_splats(int32_t N, const float *means2d, const float *conics, const float *opacities,
                                   const float *colors, int32_t C, float *splats, dnsplat_stream_t stream)
{
    if (N < 0 || C < 0 || C > DNSPLAT_MAX_CHANNELS) return DNSPLAT_ERR_INVALID_ARG;
    if (N == 0) return DNSPLAT_OK;
    if (!means2d || !conics || !opacities || !splats || (C > 0 && !colors)) return DNSPLAT_ERR_INVALID_ARG;
    dim3 block(256), grid((N + 255) / 256);
    hipLaunchKernelGGL(pack_splats_kernel, grid, block, 0, (hipStream_t)stream, N, means2d, conics, opacities,
                       colors, C, splats);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}
