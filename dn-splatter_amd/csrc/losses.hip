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
constexpr int LS_SLOTS = 64;

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
    float *__restrict__ slots;            // [LS_SLOTS][8] partial sums: a workgroup adds into copy blockIdx % LS_SLOTS (thousands of
                                          // atomics on ONE address queue up behind each other), dn_loss_fold_kernel adds the copies up
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
    // The workgroup's whole input tile — all three channels of prediction and ground truth — is requested up front, one batch of
    // loads per thread (out-of-image elements re-read a clamped pixel and are zeroed afterwards).  Loaded channel by channel inside
    // the loop, every element was a load followed by its own wait: thirteen memory round trips in a row per workgroup (round 4).
    constexpr int LD_IT = (LS_IH * LS_IW + LS_THREADS - 1) / LS_THREADS;
    float px_[LD_IT][3], py_[LD_IT][3];
    uint32_t in_mask = 0u;
#pragma unroll
    for (int it = 0; it < LD_IT; ++it) {
        const int e = min((int)threadIdx.x + it * LS_THREADS, LS_IH * LS_IW - 1);
        const int r = e / LS_IW, j = e - r * LS_IW;
        const int yy = oy + r, xx = ox + j;
        const bool in = yy < a.H && xx < a.W;
        const size_t p = ((size_t)min(yy, a.H - 1) * a.W + min(xx, a.W - 1)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) { px_[it][c] = a.rgb[p + c]; py_[it][c] = a.gt_rgb[p + c]; }
        in_mask |= (in ? 1u : 0u) << it;
    }
#pragma unroll
    for (int it = 0; it < LD_IT; ++it)       // masked in a loop of its own: a use right behind each load would be waited for there
        if (!((in_mask >> it) & 1u)) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { px_[it][c] = 0.f; py_[it][c] = 0.f; }
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int e = (int)threadIdx.x + it * LS_THREADS;
            if (e < LS_IH * LS_IW) {
                const int r = e / LS_IW, j = e - r * LS_IW;
                const float x = px_[it][c], y = py_[it][c];
                in_tile[0][r][j] = x; in_tile[1][r][j] = y; in_tile[2][r][j] = x * x; in_tile[3][r][j] = y * y;
                in_tile[4][r][j] = x * y;
            }
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
    if (threadIdx.x == 0) atomicAdd(a.slots + 8 * ((blockIdx.y * gridDim.x + blockIdx.x) % LS_SLOTS) + 0, tot);
}

// SSIM_ONLY (dnsplat_ssim): only v_rgb = d(mean SSIM)/d(rgb) is produced — no L1 term, no depth / normal section, no partial sums.
template <bool SSIM_ONLY>
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
    const float w_l1 = SSIM_ONLY ? 0.f : (1.f - a.ssim_lambda) / (3.f * P);
    const float w_ss = SSIM_ONLY ? 1.f / M : -a.ssim_lambda / M;   // loss has + l (1 - mean s)
    float s_l1 = 0.f, s_eax = 0.f, s_eay = 0.f, s_nl1 = 0.f, s_tvh = 0.f, s_tvw = 0.f;

    // ---- RGB: L1 + transposed blur of the SSIM sensitivities.  As in the statistics kernel, everything the workgroup reads for the
    // three channels (the sensitivity maps of its halo tile, prediction and ground truth of its own pixels) is requested up front.
    constexpr int LD_IT = (LS_IH * LS_IW + LS_THREADS - 1) / LS_THREADS;
    float ma[LD_IT][3], mb[LD_IT][3], mc[LD_IT][3];
    uint32_t in_mask = 0u;
#pragma unroll
    for (int it = 0; it < LD_IT; ++it) {
        const int e = min((int)threadIdx.x + it * LS_THREADS, LS_IH * LS_IW - 1);
        const int r = e / LS_IW, j = e - r * LS_IW;
        const int yy = oy + r - LS_R, xx = ox + j - LS_R;   // transposed window: input origin shifted by the halo
        const bool in = yy >= 0 && xx >= 0 && yy < VH && xx < VW;
        const size_t p = ((size_t)min(max(yy, 0), VH - 1) * W + min(max(xx, 0), VW - 1)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) { ma[it][c] = a.maps[p + c]; mb[it][c] = a.maps[plane + p + c]; mc[it][c] = a.maps[2 * plane + p + c]; }
        in_mask |= (in ? 1u : 0u) << it;
    }
    float xk[2][3], yk[2][3];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = min(oy + ty + 8 * k, H - 1), j = min(ox + tx, W - 1);
        const size_t p = ((size_t)i * W + j) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) { xk[k][c] = a.rgb[p + c]; yk[k][c] = a.gt_rgb[p + c]; }
    }
#pragma unroll
    for (int it = 0; it < LD_IT; ++it)       // masked only now: a use right behind each load would be waited for there
        if (!((in_mask >> it) & 1u)) {
#pragma unroll
            for (int c = 0; c < 3; ++c) { ma[it][c] = 0.f; mb[it][c] = 0.f; mc[it][c] = 0.f; }
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int it = 0; it < LD_IT; ++it) {
            const int e = (int)threadIdx.x + it * LS_THREADS;
            if (e < LS_IH * LS_IW) {
                const int r = e / LS_IW, j = e - r * LS_IW;
                in_tile[0][r][j] = ma[it][c]; in_tile[1][r][j] = mb[it][c]; in_tile[2][r][j] = mc[it][c];
            }
        }
        __syncthreads();
        float o[3][2];
        blur_tile<3>(in_tile, h, a.win, tx, ty, o);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int i = oy + ty + 8 * k, j = ox + tx;
            if (i < H && j < W) {
                const size_t p = ((size_t)i * W + j) * 3 + c;
                const float x = xk[k][c], y = yk[k][c];
                const float d = x - y;
                s_l1 += fabsf(d);
                a.v_rgb[p] = w_l1 * sgn(d) + w_ss * (o[0][k] + 2.f * x * o[1][k] + y * o[2][k]);
            }
        }
    }
    if (SSIM_ONLY) return;

    // ---- depth (EdgeAwareLogL1) and normal (L1 + TV): 3-point stencils
    const float n_x = a.gt_depth ? a.counts[0] : 1.f, n_y = a.gt_depth ? a.counts[1] : 1.f;
    const float w_tvh = 1.f / (3.f * (float)H * (float)(W - 1)), w_tvw = 1.f / (3.f * (float)(H - 1) * (float)W);
    const float w_nl1 = 1.f / (3.f * P);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = oy + ty + 8 * k, j = ox + tx;
        if (i >= H || j >= W) continue;
        const size_t px = (size_t)i * W + j;
        // every value the pixel's stencils read, requested together: neighbours beyond the image are clamped to the pixel itself and
        // masked by the same conditions as before (the branchy form compiled to ~25 dependent round trips per pixel)
        const bool has_r = j < W - 1, has_l = j > 0, has_b = i < H - 1, has_t = i > 0;
        const size_t pr = has_r ? px + 1 : px, pl = has_l ? px - 1 : px, pb = has_b ? px + W : px, pt = has_t ? px - W : px;
        float gd = 0.f, dd = 0.f, g0[3], gr[3], gb[3];
        if (a.gt_depth) {
            gd = a.gt_depth[px]; dd = a.depth[px];
#pragma unroll
            for (int c = 0; c < 3; ++c) { g0[c] = yk[k][c]; gr[c] = a.gt_rgb[pr * 3 + c]; gb[c] = a.gt_rgb[pb * 3 + c]; }
        }
        float nc[3], nr[3], nl[3], nb[3], nt[3], gn[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            nc[c] = a.normal[px * 3 + c];
            if (a.gt_normal) {
                gn[c] = a.gt_normal[px * 3 + c];
                nr[c] = a.normal[pr * 3 + c]; nl[c] = a.normal[pl * 3 + c]; nb[c] = a.normal[pb * 3 + c]; nt[c] = a.normal[pt * 3 + c];
            }
        }
        if (a.gt_depth) {
            float vd = 0.f;
            if (gd > a.depth_tolerance) {
                const float d = dd - gd;
                const float l = logf(1.f + fabsf(d));
                const float dl = sgn(d) / (1.f + fabsf(d));
                // edge weights from the ground-truth image clamped at 10/255 (dn_model.py:633)
#pragma unroll
                for (int c = 0; c < 3; ++c) g0[c] = fmaxf(g0[c], 10.f / 255.f);
                if (has_r) {
                    float m = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) m += fabsf(g0[c] - fmaxf(gr[c], 10.f / 255.f));
                    const float lam = expf(-m / 3.f);
                    s_eax += lam * l;
                    vd += lam / n_x;
                }
                if (has_b) {
                    float m = 0.f;
#pragma unroll
                    for (int c = 0; c < 3; ++c) m += fabsf(g0[c] - fmaxf(gb[c], 10.f / 255.f));
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
            const float n = nc[c];
            float v = 0.f;
            if (a.gt_normal) {
                const float d = n - gn[c];
                s_nl1 += fabsf(d);
                v += w_nl1 * sgn(d);
                // TVLoss (losses.py:279-295)
                if (has_r) { const float t = n - nr[c]; s_tvh += fabsf(t); v += w_tvh * sgn(t); }
                if (has_l) v -= w_tvh * sgn(nl[c] - n);
                if (has_b) { const float t = n - nb[c]; s_tvw += fabsf(t); v += w_tvw * sgn(t); }
                if (has_t) v -= w_tvw * sgn(nt[c] - n);
            }
            a.v_normal[p] = v;
        }
    }
    float part[6] = {s_l1, s_eax, s_eay, s_nl1, s_tvh, s_tvw};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float tot = block_sum(part[q], red);
        if (threadIdx.x == 0 && tot != 0.f) atomicAdd(a.slots + 8 * ((blockIdx.y * gridDim.x + blockIdx.x) % LS_SLOTS) + 1 + q, tot);
    }
}

// sums[q] = sum over the LS_SLOTS copies (one wave)
__global__ __launch_bounds__(DNS_WAVE) void dn_loss_fold_kernel(const float *__restrict__ slots, float *__restrict__ sums)
{
    const int q = threadIdx.x & 7, part = threadIdx.x >> 3;      // 8 lanes per quantity... 8 partial sums of LS_SLOTS / 8 copies each
    float v = 0.f;
    for (int s = part; s < LS_SLOTS; s += 8) v += slots[8 * s + q];
#pragma unroll
    for (int off = 32; off >= 8; off >>= 1) v += __shfl_xor(v, off, DNS_WAVE);
    if (threadIdx.x < 8) sums[q] = v;
}

// dn-splatter's per-Gaussian scale regulariser (regularization_strategy.py:195-199): mean over the Gaussians of the SMALLEST
// activated scale, min_k exp(s_k).  One lane per Gaussian: adds its term to *sum and writes the gradient row
// d/d(s_k) = [k == argmin] exp(s_k) * weight  (weight = 1 / N for the mean; ties -> the first minimal component, as torch.min).
__global__ __launch_bounds__(256) void scale_reg_kernel(int N, const float *__restrict__ scales, float weight,
                                                        float *__restrict__ v_scales, float *__restrict__ sum)
{
    __shared__ float red[4];
    float term_sum = 0.f;
    // grid-stride: a few hundred workgroups, ONE atomic on the loss word each.  With a workgroup per 256 Gaussians the 19 500
    // same-address atomics of a 5 M-Gaussian scene queued up behind each other: 253 us for a kernel that moves 120 MB (round 4).
    for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < N; g += gridDim.x * blockDim.x) {
        const float e0 = expf(scales[3 * g]), e1 = expf(scales[3 * g + 1]), e2 = expf(scales[3 * g + 2]);
        const int k = (e1 < e0) ? ((e2 < e1) ? 2 : 1) : ((e2 < e0) ? 2 : 0);
        const float m = k == 0 ? e0 : (k == 1 ? e1 : e2);
        const float term = m * weight;
        term_sum += term;
        v_scales[3 * g] = k == 0 ? term : 0.f;
        v_scales[3 * g + 1] = k == 1 ? term : 0.f;
        v_scales[3 * g + 2] = k == 2 ? term : 0.f;
    }
    const float tot = block_sum(term_sum, red);
    if (threadIdx.x == 0 && tot != 0.f) atomicAdd(sum, tot);
}


// ---- the reference's loss MODULES one by one (ABI 15), for a loss stack that otherwise stays in PyTorch: same arithmetic as the
// stencils of dn_loss_grad_kernel, as separate entry points behind drop-in nn.Modules (fused_loss.EdgeAwareLogL1 / TVLoss).

// EdgeAwareLogL1, "scalar" implementation (losses.py:187-224).  One thread per pixel: sums[0] += lambda_x log(1 + |d|), sums[1] +=
// lambda_y log(1 + |d|), sums[2] / sums[3] += 1 for every pixel that counts in the x / y mean (mask, and a right / lower neighbour);
// v_x / v_y = d/d(pred) of the two UNNORMALISED sums (the caller divides by the counts: they are only known when the launch ends).
__global__ __launch_bounds__(LS_THREADS) void edge_aware_logl1_kernel(int W, int H, const float *__restrict__ pred, const float *__restrict__ gt,
                                                                       const float *__restrict__ rgb, const uint8_t *__restrict__ mask,
                                                                       float *__restrict__ v_x, float *__restrict__ v_y, float *__restrict__ slots)
{
    __shared__ float red[4];
    const long long P = (long long)W * H;
    float s_x = 0.f, s_y = 0.f, c_x = 0.f, c_y = 0.f;
    for (long long px = (long long)blockIdx.x * LS_THREADS + threadIdx.x; px < P; px += (long long)gridDim.x * LS_THREADS) {
        const int i = (int)(px / W), j = (int)(px - (long long)i * W);
        const bool has_r = j < W - 1, has_b = i < H - 1;
        const long long pr = has_r ? px + 1 : px, pb = has_b ? px + W : px;
        const float d = pred[px] - gt[px];
        float g0[3], gr[3], gb[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { g0[c] = rgb[px * 3 + c]; gr[c] = rgb[pr * 3 + c]; gb[c] = rgb[pb * 3 + c]; }
        const bool m = mask ? mask[px] != 0 : true;
        const float l = logf(1.f + fabsf(d)), dl = sgn(d) / (1.f + fabsf(d));
        float vx = 0.f, vy = 0.f;
        if (m && has_r) {
            const float lam = expf(-(fabsf(g0[0] - gr[0]) + fabsf(g0[1] - gr[1]) + fabsf(g0[2] - gr[2])) / 3.f);
            s_x += lam * l; c_x += 1.f; vx = lam * dl;
        }
        if (m && has_b) {
            const float lam = expf(-(fabsf(g0[0] - gb[0]) + fabsf(g0[1] - gb[1]) + fabsf(g0[2] - gb[2])) / 3.f);
            s_y += lam * l; c_y += 1.f; vy = lam * dl;
        }
        if (v_x) { v_x[px] = vx; v_y[px] = vy; }
    }
    float part[4] = {s_x, s_y, c_x, c_y};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float tot = block_sum(part[q], red);
        if (threadIdx.x == 0 && tot != 0.f) atomicAdd(slots + 8 * (blockIdx.x % LS_SLOTS) + q, tot);
    }
}

// TVLoss (losses.py:279-295) of an [H,W,C] image: sums[0] += |p - right|, sums[1] += |p - lower|; v = d/d(pred) of
// sums[0] / (H (W-1) C) + sums[1] / ((H-1) W C), final (the normalisers are constants).
__global__ __launch_bounds__(LS_THREADS) void tv_loss_kernel(int W, int H, int C, const float *__restrict__ pred, float *__restrict__ v,
                                                              float *__restrict__ slots)
{
    __shared__ float red[4];
    const long long P = (long long)W * H;
    const float w_h = 1.f / ((float)C * (float)H * (float)(W - 1)), w_w = 1.f / ((float)C * (float)(H - 1) * (float)W);
    float s_h = 0.f, s_w = 0.f;
    for (long long px = (long long)blockIdx.x * LS_THREADS + threadIdx.x; px < P; px += (long long)gridDim.x * LS_THREADS) {
        const int i = (int)(px / W), j = (int)(px - (long long)i * W);
        const bool has_r = j < W - 1, has_l = j > 0, has_b = i < H - 1, has_t = i > 0;
        const long long pr = has_r ? px + 1 : px, pl = has_l ? px - 1 : px, pb = has_b ? px + W : px, pt = has_t ? px - W : px;
        for (int c = 0; c < C; ++c) {
            const float n = pred[px * C + c], nr = pred[pr * C + c], nl = pred[pl * C + c], nb = pred[pb * C + c], nt = pred[pt * C + c];
            float g = 0.f;
            if (has_r) { const float t = n - nr; s_h += fabsf(t); g += w_h * sgn(t); }
            if (has_l) g -= w_h * sgn(nl - n);
            if (has_b) { const float t = n - nb; s_w += fabsf(t); g += w_w * sgn(t); }
            if (has_t) g -= w_w * sgn(nt - n);
            if (v) v[px * C + c] = g;
        }
    }
    float part[2] = {s_h, s_w};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float tot = block_sum(part[q], red);
        if (threadIdx.x == 0 && tot != 0.f) atomicAdd(slots + 8 * (blockIdx.x % LS_SLOTS) + q, tot);
    }
}

}  // namespace

extern "C" int dnsplat_scale_reg(int32_t N, const float *scales_log, float weight, float *v_scales, float *sum, dnsplat_stream_t stream)
{
    if (N < 0 || (N > 0 && (!scales_log || !v_scales)) || !sum) return DNSPLAT_ERR_INVALID_ARG;
    if (N == 0) return DNSPLAT_OK;
    const int blocks = (N + 255) / 256 < 1024 ? (N + 255) / 256 : 1024;
    hipLaunchKernelGGL(scale_reg_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, N, scales_log, weight, v_scales, sum);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

static void ssim_window(Gauss11 &win)
{
    // pytorch_msssim window: exp(-(x - 5)^2 / (2 * 1.5^2)), normalised
    double g[LS_K], tot = 0.0;
    for (int t = 0; t < LS_K; ++t) { const double x = t - LS_K / 2; g[t] = exp(-(x * x) / (2.0 * 1.5 * 1.5)); tot += g[t]; }
    for (int t = 0; t < LS_K; ++t) win.g[t] = (float)(g[t] / tot);
}

extern "C" int dnsplat_ssim(int32_t width, int32_t height, const float *x, const float *y, float *maps, float *v_x, float *sums,
                            dnsplat_stream_t stream_)
{
    if (width <= LS_R || height <= LS_R) return DNSPLAT_ERR_UNSUPPORTED;           // SSIM needs an 11x11 window
    if (!x || !y || !maps || !sums) return DNSPLAT_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    LossArgs a = {};
    a.W = width; a.H = height;
    a.rgb = x; a.gt_rgb = y; a.v_rgb = v_x; a.sums = sums;
    ssim_window(a.win);
    a.slots = maps;
    a.maps = maps + LS_SLOTS * 8;
    if (hipMemsetAsync(a.slots, 0, LS_SLOTS * 8 * sizeof(float), stream) != hipSuccess) return DNSPLAT_ERR_LAUNCH;
    dim3 grid_v((a.W - LS_R + LS_TW - 1) / LS_TW, (a.H - LS_R + LS_TH - 1) / LS_TH);
    dim3 grid((a.W + LS_TW - 1) / LS_TW, (a.H + LS_TH - 1) / LS_TH);
    hipLaunchKernelGGL(dn_ssim_stats_kernel, grid_v, dim3(LS_THREADS), 0, stream, a);
    if (v_x) hipLaunchKernelGGL(dn_loss_grad_kernel<true>, grid, dim3(LS_THREADS), 0, stream, a);
    hipLaunchKernelGGL(dn_loss_fold_kernel, dim3(1), dim3(DNS_WAVE), 0, stream, (const float *)a.slots, a.sums);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_edge_aware_logl1(int32_t width, int32_t height, const float *pred, const float *gt, const float *rgb,
                                        const uint8_t *mask, float *v_x, float *v_y, float *scratch, float *sums, dnsplat_stream_t stream_)
{
    if (width < 2 || height < 2) return DNSPLAT_ERR_UNSUPPORTED;
    if (!pred || !gt || !rgb || !scratch || !sums || ((v_x == nullptr) != (v_y == nullptr))) return DNSPLAT_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (hipMemsetAsync(scratch, 0, LS_SLOTS * 8 * sizeof(float), stream) != hipSuccess) return DNSPLAT_ERR_LAUNCH;
    const long long P = (long long)width * height;
    const int blocks = (int)((P + LS_THREADS - 1) / LS_THREADS < 4096 ? (P + LS_THREADS - 1) / LS_THREADS : 4096);
    hipLaunchKernelGGL(edge_aware_logl1_kernel, dim3(blocks), dim3(LS_THREADS), 0, stream, width, height, pred, gt, rgb, mask, v_x, v_y, scratch);
    hipLaunchKernelGGL(dn_loss_fold_kernel, dim3(1), dim3(DNS_WAVE), 0, stream, (const float *)scratch, sums);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_tv_loss(int32_t width, int32_t height, int32_t channels, const float *pred, float *v_pred, float *scratch, float *sums,
                               dnsplat_stream_t stream_)
{
    if (width < 2 || height < 2 || channels < 1) return DNSPLAT_ERR_UNSUPPORTED;
    if (!pred || !scratch || !sums) return DNSPLAT_ERR_INVALID_ARG;
    hipStream_t stream = (hipStream_t)stream_;
    if (hipMemsetAsync(scratch, 0, LS_SLOTS * 8 * sizeof(float), stream) != hipSuccess) return DNSPLAT_ERR_LAUNCH;
    const long long P = (long long)width * height;
    const int blocks = (int)((P + LS_THREADS - 1) / LS_THREADS < 4096 ? (P + LS_THREADS - 1) / LS_THREADS : 4096);
    hipLaunchKernelGGL(tv_loss_kernel, dim3(blocks), dim3(LS_THREADS), 0, stream, width, height, channels, pred, v_pred, scratch);
    hipLaunchKernelGGL(dn_loss_fold_kernel, dim3(1), dim3(DNS_WAVE), 0, stream, (const float *)scratch, sums);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

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
    ssim_window(a.win);
    // the partial sums live at the front of the scratch `maps` (9 H W floats, of which the kernels use the valid-window part): the
    // first LS_SLOTS x 8 floats are NOT map entries — see LossArgs::maps below
    a.slots = u->maps;
    a.maps = u->maps + LS_SLOTS * 8;
    if (hipMemsetAsync(a.slots, 0, LS_SLOTS * 8 * sizeof(float), stream) != hipSuccess) return DNSPLAT_ERR_LAUNCH;
    dim3 grid_v((a.W - LS_R + LS_TW - 1) / LS_TW, (a.H - LS_R + LS_TH - 1) / LS_TH);
    dim3 grid((a.W + LS_TW - 1) / LS_TW, (a.H + LS_TH - 1) / LS_TH);
    hipLaunchKernelGGL(dn_ssim_stats_kernel, grid_v, dim3(LS_THREADS), 0, stream, a);
    hipLaunchKernelGGL(dn_loss_grad_kernel<false>, grid, dim3(LS_THREADS), 0, stream, a);
    hipLaunchKernelGGL(dn_loss_fold_kernel, dim3(1), dim3(DNS_WAVE), 0, stream, (const float *)a.slots, a.sums);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}
