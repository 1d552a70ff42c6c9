// losses.hip — dn-splatter's per-pixel training losses and their gradients in two launches
// (restates, fused: dn_splatter/dn_model.py:614-729 for regularization_strategy ==
// "dn-splatter" with mono depth + mono normal supervision, regularization_strategy.py:146-199,
// losses.py:154-224 (L1 / LogL1 / EdgeAwareLogL1), :279-295 (TVLoss), and the inherited nerfstudio RGB term
// (1 - l) L1 + l (1 - SSIM), pytorch_msssim SSIM: 11-tap Gaussian sigma 1.5, valid padding, data range 1).
//
// The PyTorch loss stack is ~120 kernels and 5.9 ms per 1080p frame on MI355X — more than the whole renderer
// (4.0 ms).  Here:
//   dn_ssim_stats_kernel   valid-window SSIM statistics per channel (separable blur of x, y, xx, yy, xy staged in
//                          LDS) -> sum of SSIM and the three sensitivity maps a = ds/dmu_x, b = ds/dE[xx],
//                          c = ds/dE[xy];
//   dn_loss_grad_kernel    transposed blur of (a, b, c) -> d(1-SSIM)/d(rgb); plus the L1, EdgeAwareLogL1, normal L1
//                          and TV terms (3-point stencils straight from global memory) -> cotangents of rgb, depth,
//                          normal and the seven partial sums of the loss value.
// Streaming, HBM/LDS-bound kernels (about 60 floats of traffic per pixel in total).  The min-scale term is per
// Gaussian, not per pixel, and stays in torch.

#include "splat_common.h"

namespace {

constexpr int LS_TW = 32, LS_TH = 16;          // output tile of one 256-thread workgroup
constexpr int LS_K = 11, LS_R = LS_K - 1;      // window, halo
constexpr int LS_IW = LS_TW + LS_R, LS_IH = LS_TH + LS_R;
constexpr int LS_THREADS = 256;

struct Gauss11 { float g[LS_K]; };

struct LossArgs {
    int W, H;
    const float *__restrict__ rgb;        // prediction [H,W,3]
    const float *__restrict__ depth;      // [H,W]
    const float *__restrict__ normal;     // [H,W,3]
    const float *__restrict__ gt_rgb;     // [H,W,3]
    const float *__restrict__ gt_depth;   // [H,W] or null
    const float *__restrict__ gt_normal;  // [H,W,3] or null
    const float *__restrict__ counts;     // device [2]: #valid depth pixels in columns < W-1, in rows < H-1
    float ssim_lambda, depth_weight, depth_tolerance;
    float *__restrict__ maps;             // [3][H,W,3] sensitivity maps a, b, c (only the valid-window region is written and read)
    float *__restrict__ v_rgb, *__restrict__ v_depth, *__restrict__ v_normal;
    float *__restrict__ sums;             // [8]: ssim, l1, ea_x, ea_y, n_l1, tv_h, tv_w, (unused)
    Gauss11 win;
};

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__device__ __forceinline__ float block_sum(float v, float *red)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, DNS_WAVE);
    const int w = threadIdx.x / DNS_WAVE;
    __syncthreads();
    if ((threadIdx.x & (DNS_WAVE - 1)) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// Separable 11x11 blur of NQ quantities over the workgroup's tile.  in_tile: NQ planes [LS_IH][LS_IW] already in
// LDS; h: scratch NQ planes [LS_IH][LS_TW]; result for this thread's two outputs (rows ty and ty + 8) in out[NQ][2].
template <int NQ>
__device__ __forceinline__ void blur_tile(const float (*in_tile)[LS_IH][LS_IW], float (*h)[LS_IH][LS_TW], const Gauss11 &win,
                                          int tx, int ty, float out[NQ][2])
{
    for (int e = threadIdx.x; e < LS_IH * LS_TW; e += LS_THREADS) {
        const int r = e / LS_TW, j = e - r * LS_TW;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < LS_K; ++t) s += win.g[t] * in_tile[q][r][j + t];
            h[q][r][j] = s;
        }
    }
    __syncthreads();
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int i = ty + 8 * o;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < LS_K; ++t) s += win.g[t] * h[q][i + t][tx];
            out[q][o] = s;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(LS_THREADS) void dn_ssim_stats_kernel(LossArgs a)
{
    __shared__ float in_tile[5][LS_IH][LS_IW];   // x, y, xx, yy, xy of one channel
    __shared__ float h[5][LS_IH][LS_TW];
    __shared__ float red[4];
    const int ox = blockIdx.x * LS_TW, oy = blockIdx.y * LS_TH;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8 threads, two output rows each
    const int VW = a.W - LS_R, VH = a.H - LS_R;               // valid window positions
    const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
    float ssim_sum = 0.f;
    for (int c = 0; c < 3; ++c) {
        for (int e = threadIdx.x; e < LS_IH * LS_IW; e += LS_THREADS) {
            const int r = e / LS_IW, j = e - r * LS_IW;
            const int yy = oy + r, xx = ox + j;
            float x = 0.f, y = 0.f;
            if (yy < a.H && xx < a.W) {
                const size_t p = ((size_t)yy * a.W + xx) * 3 + c;
                x = a.rgb[p]; y = a.gt_rgb[p];
            }
            in_tile[0][r][j] = x; in_tile[1][r][j] = y; in_tile[2][r][j] = x * x; in_tile[3][r][j] = y * y;
            in_tile[4][r][j] = x * y;
        }
        __syncthreads();
        float o[5][2];
        blur_tile<5>(in_tile, h, a.win, tx, ty, o);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = oy + ty + 8 * k, j = ox + tx;
            if (i < VH && j < VW) {
                const float mx = o[0][k], my = o[1][k];
                const float sxx = o[2][k] - mx * mx, syy = o[3][k] - my * my, sxy = o[4][k] - mx * my;
                const float A1 = 2.f * mx * my + c1, A2 = 2.f * sxy + c2;
                const float B1 = mx * mx + my * my + c1, B2 = sxx + syy + c2;
                const float iB = 1.f / (B1 * B2);
                const float s = A1 * A2 * iB;
                ssim_sum += s;
                const size_t p = ((size_t)i * a.W + j) * 3 + c;
                const size_t plane = (size_t)a.W * a.H * 3;
                a.maps[p] = 2.f * my * (A2 - A1) * iB - 2.f * mx * s * (1.f / B1 - 1.f / B2);   // ds/dmu_x
                a.maps[plane + p] = -s / B2;                                                     // ds/dE[xx]
                a.maps[2 * plane + p] = 2.f * A1 * iB;                                           // ds/dE[xy]
            }
        }
    }
    const float tot = block_sum(ssim_sum, red);
    if (threadIdx.x == 0) atomicAdd(a.sums + 0, tot);
}

__global__ __launch_bounds__(LS_THREADS) void dn_loss_grad_kernel(LossArgs a)
{
    __shared__ float in_tile[3][LS_IH][LS_IW];   // a, b, c maps of one channel
    __shared__ float h[3][LS_IH][LS_TW];
    __shared__ float red[4];
    const int ox = blockIdx.x * LS_TW, oy = blockIdx.y * LS_TH;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int W = a.W, H = a.H, VW = W - LS_R, VH = H - LS_R;
    const size_t plane = (size_t)W * H * 3;
    const float P = (float)W * (float)H;
    const float M = 3.f * (float)VW * (float)VH;              // SSIM windows x channels
    const float w_l1 = (1.f - a.ssim_lambda) / (3.f * P), w_ss = -a.ssim_lambda / M;   // loss has + l (1 - mean s)
    float s_l1 = 0.f, s_eax = 0.f, s_eay = 0.f, s_nl1 = 0.f, s_tvh = 0.f, s_tvw = 0.f;

    // ---- RGB: L1 + transposed blur of the SSIM sensitivities
    for (int c = 0; c < 3; ++c) {
        for (int e = threadIdx.x; e < LS_IH * LS_IW; e += LS_THREADS) {
            const int r = e / LS_IW, j = e - r * LS_IW;
            const int yy = oy + r - LS_R, xx = ox + j - LS_R;   // transposed window: input origin shifted by the halo
            float va = 0.f, vb = 0.f, vc = 0.f;
            if (yy >= 0 && xx >= 0 && yy < VH && xx < VW) {
                const size_t p = ((size_t)yy * W + xx) * 3 + c;
                va = a.maps[p]; vb = a.maps[plane + p]; vc = a.maps[2 * plane + p];
            }
            in_tile[0][r][j] = va; in_tile[1][r][j] = vb; in_tile[2][r][j] = vc;
        }
        __syncthreads();
        float o[3][2];
        blur_tile<3>(in_tile, h, a.win, tx, ty, o);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = oy + ty + 8 * k, j = ox + tx;
            if (i < H && j < W) {
                const size_t p = ((size_t)i * W + j) * 3 + c;
                const float x = a.rgb[p], y = a.gt_rgb[p];
                const float d = x - y;
                s_l1 += fabsf(d);
                a.v_rgb[p] = w_l1 * sgn(d) + w_ss * (o[0][k] + 2.f * x * o[1][k] + y * o[2][k]);
            }
        }
    }

    // ---- depth (EdgeAwareLogL1) and normal (L1 + TV): 3-point stencils
    const float n_x = a.gt_depth ? a.counts[0] : 1.f, n_y = a.gt_depth ? a.counts[1] : 1.f;
    const float w_tvh = 1.f / (3.f * (float)H * (float)(W - 1)), w_tvw = 1.f / (3.f * (float)(H - 1) * (float)W);
    const float w_nl1 = 1.f / (3.f * P);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = oy + ty + 8 * k, j = ox + tx;
        if (i >= H || j >= W) continue;
        const size_t px = (size_t)i * W + j;
        if (a.gt_depth) {
            const float gd = a.gt_depth[px];
            float vd = 0.f;
            if (gd > a.depth_tolerance) {
                const float d = a.depth[px] - gd;
                const float l = logf(1.f + fabsf(d));
                const float dl = sgn(d) / (1.f + fabsf(d));
                // edge weights from the ground-truth image clamped at 10/255 (dn_model.py:633)
                float g0[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) g0[c] = fmaxf(a.gt_rgb[px * 3 + c], 10.f / 255.f);
                if (j < W - 1) {
                    float m = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) m += fabsf(g0[c] - fmaxf(a.gt_rgb[(px + 1) * 3 + c], 10.f / 255.f));
                    const float lam = expf(-m / 3.f);
                    s_eax += lam * l;
                    vd += lam / n_x;
                }
                if (i < H - 1) {
                    float m = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) m += fabsf(g0[c] - fmaxf(a.gt_rgb[(px + W) * 3 + c], 10.f / 255.f));
                    const float lam = expf(-m / 3.f);
                    s_eay += lam * l;
                    vd += lam / n_y;
                }
                vd *= a.depth_weight * dl;
            }
            a.v_depth[px] = vd;
        } else {
            a.v_depth[px] = 0.f;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const size_t p = px * 3 + c;
            const float n = a.normal[p];
            float v = 0.f;
            if (a.gt_normal) {
                const float d = n - a.gt_normal[p];
                s_nl1 += fabsf(d);
                v += w_nl1 * sgn(d);
                // TVLoss (losses.py:279-295)
                if (j < W - 1) { const float t = n - a.normal[p + 3]; s_tvh += fabsf(t); v += w_tvh * sgn(t); }
                if (j > 0) v -= w_tvh * sgn(a.normal[p - 3] - n);
                if (i < H - 1) { const float t = n - a.normal[p + (size_t)3 * W]; s_tvw += fabsf(t); v += w_tvw * sgn(t); }
                if (i > 0) v -= w_tvw * sgn(a.normal[p - (size_t)3 * W] - n);
            }
            a.v_normal[p] = v;
        }
    }
    float part[6] = {s_l1, s_eax, s_eay, s_nl1, s_tvh, s_tvw};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float tot = block_sum(part[q], red);
        if (threadIdx.x == 0 && tot != 0.f) atomicAdd(a.sums + 1 + q, tot);
    }
}

}  // namespace

extern "C" int dnsplat_dn_loss(const dnsplat_dn_loss_args *u, dnsplat_stream_t stream_)
{
    if (!u) return DNSPLAT_ERR_INVALID_ARG;
    if (u->width <= LS_R || u->height <= LS_R) return DNSPLAT_ERR_UNSUPPORTED;     // SSIM needs an 11x11 window
    if (!u->rgb || !u->depth || !u->normal || !u->gt_rgb || !u->maps || !u->v_rgb || !u->v_depth || !u->v_normal || !u->sums)
        return DNSPLAT_ERR_INVALID_ARG;
    if (u->gt_depth && !u->depth_counts) return DNSPLAT_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    LossArgs a;
    a.W = u->width; a.H = u->height;
    a.rgb = u->rgb; a.depth = u->depth; a.normal = u->normal;
    a.gt_rgb = u->gt_rgb; a.gt_depth = u->gt_depth; a.gt_normal = u->gt_normal; a.counts = u->depth_counts;
    a.ssim_lambda = u->ssim_lambda; a.depth_weight = u->depth_weight; a.depth_tolerance = u->depth_tolerance;
    a.maps = u->maps; a.v_rgb = u->v_rgb; a.v_depth = u->v_depth; a.v_normal = u->v_normal; a.sums = u->sums;
    // pytorch_msssim window: exp(-(x - 5)^2 / (2 * 1.5^2)), normalised
    double g[LS_K], tot = 0.0;
    for (int t = 0; t < LS_K; ++t) { const double x = t - LS_K / 2; g[t] = exp(-(x * x) / (2.0 * 1.5 * 1.5)); tot += g[t]; }
    for (int t = 0; t < LS_K; ++t) a.win.g[t] = (float)(g[t] / tot);
    if (hipMemsetAsync(u->sums, 0, 8 * sizeof(float), stream) != hipSuccess) return DNSPLAT_ERR_LAUNCH;
    dim3 grid_v((a.W - LS_R + LS_TW - 1) / LS_TW, (a.H - LS_R + LS_TH - 1) / LS_TH);
    dim3 grid((a.W + LS_TW - 1) / LS_TW, (a.H + LS_TH - 1) / LS_TH);
    hipLaunchKernelGGL(dn_ssim_stats_kernel, grid_v, dim3(LS_THREADS), 0, stream, a);
    hipLaunchKernelGGL(dn_loss_grad_kernel, grid, dim3(LS_THREADS), 0, stream, a);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}
