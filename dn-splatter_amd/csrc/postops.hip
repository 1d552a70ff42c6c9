// postops.hip — the small per-frame kernels around the compositing pass of the fused dn-splatter path
// (SURVEY.md 8(f) row N1).
//
//   dnsplat_dn_depth_normals   dn_model.py:533-537 (alpha == 0 depth fill with the image-wide maximum) and
//                              dn_model.py:589-603 -> utils/normal_utils.py:9-48 + utils/camera_utils.py:92-144
//                              (depth -> back-projected points -> 4-neighbour cross product -> normal image).
//                              The reference spends ~25 torch kernels and two [P,3]x[3,3] GEMMs on it per frame;
//                              it carries no gradient (depth.detach(), dn_model.py:590).
//   dnsplat_camera_prepare     nerfstudio get_viewmat + intrinsics matrix + the normal frame (dn_model.py:475-479,
//                              550-560) from the camera-to-world matrix, one thread, one launch.
//
// Memory-bound streaming kernels: one lane per pixel, row-major, consecutive lanes on consecutive pixels
// (the 5-point depth stencil re-reads come from L1/L2).

#include "splat_common.h"

namespace {

__global__ __launch_bounds__(256) void dn_depth_normals_kernel(int W, int H, float fx, float fy, float cx, float cy,
                                                               const float *__restrict__ depth,
                                                               const float *__restrict__ alphas,
                                                               const float *__restrict__ depth_max,
                                                               float *__restrict__ depth_out,
                                                               float *__restrict__ surface_normal)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= W * H) return;
    const int i = idx / W, j = idx - i * W;
    // all eleven loads of the pixel are issued together (neighbours clamped into the image, used only away from the border):
    // `alpha > 0 ? depth[..] : dmax` per stencil point compiled to ten dependent round trips in a row
    const bool inner = i >= 1 && i < H - 1 && j >= 1 && j < W - 1;
    const int il = inner ? idx - 1 : idx, ir = inner ? idx + 1 : idx, it = inner ? idx - W : idx, ib = inner ? idx + W : idx;
    const float dmax = *depth_max;
    const float a_c = alphas[idx], a_l = alphas[il], a_r = alphas[ir], a_t = alphas[it], a_b = alphas[ib];
    const float d_c = depth[idx], d_l = depth[il], d_r = depth[ir], d_t = depth[it], d_b = depth[ib];
    depth_out[idx] = a_c > 0.f ? d_c : dmax;

    float n0 = 0.f, n1 = 0.f, n2 = 0.f;
    if (inner) {
        // back-projection with pixel centres at +0.5 (camera_utils.py:92-144), c2w = identity
        const float dl = a_l > 0.f ? d_l : dmax, dr = a_r > 0.f ? d_r : dmax;
        const float dt = a_t > 0.f ? d_t : dmax, db = a_b > 0.f ? d_b : dmax;
        // two reciprocals instead of eight divisions (the kernel is bound by them, not by its 49 MB): within 1-2 ulp of the
        // reference's (x - cx) * d / fx, against a test tolerance of 5e-6 on the [0, 1] normal image
        const float x = (float)j + 0.5f, y = (float)i + 0.5f, ifx = 1.f / fx, ify = 1.f / fy;
        const float lx = (x - 1.f - cx) * dl * ifx, ly = (y - cy) * dl * ify, lz = dl;
        const float rx = (x + 1.f - cx) * dr * ifx, ry = (y - cy) * dr * ify, rz = dr;
        const float tx = (x - cx) * dt * ifx, ty = (y - 1.f - cy) * dt * ify, tz = dt;
        const float bx = (x - cx) * db * ifx, by = (y + 1.f - cy) * db * ify, bz = db;
        const float ax = rx - lx, ay = ry - ly, az = rz - lz;     // left_to_right
        const float ux = tx - bx, uy = ty - by, uz = tz - bz;     // bottom_to_top
        float c0 = ay * uz - az * uy, c1 = az * ux - ax * uz, c2 = ax * uy - ay * ux;
        const float inrm = 1.f / fmaxf(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), 1e-12f);   // F.normalize eps
        n0 = c0 * inrm; n1 = c1 * inrm; n2 = c2 * inrm;
    }
    // dn_model.py:599-603: @ diag(1,-1,-1), then (1 + n) / 2; the zero-padded border becomes 0.5
    surface_normal[3 * idx + 0] = (1.f + n0) / 2.f;
    surface_normal[3 * idx + 1] = (1.f - n1) / 2.f;
    surface_normal[3 * idx + 2] = (1.f - n2) / 2.f;
}

__global__ void camera_prepare_kernel(const float *__restrict__ c2w, float fx, float fy, float cx, float cy,
                                      float *__restrict__ viewmat, float *__restrict__ K, float *__restrict__ nf,
                                      uint32_t *__restrict__ zero_word, int n_zero_words)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (zero_word)
        for (int i = 0; i < n_zero_words; ++i) zero_word[i] = 0u;
    // c2w [3,4] nerfstudio/OpenGL.  get_viewmat: flip the y and z camera axes, then invert analytically.
    float R[9], t[3];
    for (int r = 0; r < 3; ++r) {
        R[3 * r + 0] = c2w[4 * r + 0];
        R[3 * r + 1] = -c2w[4 * r + 1];
        R[3 * r + 2] = -c2w[4 * r + 2];
        t[r] = c2w[4 * r + 3];
    }
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) viewmat[4 * r + c] = R[3 * c + r];               // R^T
        viewmat[4 * r + 3] = -(R[0 + r] * t[0] + R[3 + r] * t[1] + R[6 + r] * t[2]);   // -R^T t
    }
    viewmat[12] = 0.f; viewmat[13] = 0.f; viewmat[14] = 0.f; viewmat[15] = 1.f;
    K[0] = fx; K[1] = 0.f; K[2] = cx; K[3] = 0.f; K[4] = fy; K[5] = cy; K[6] = 0.f; K[7] = 0.f; K[8] = 1.f;
    if (nf) {
        // n_cam = M n_world with M = c2w[:3,:3]^T (dn_model.py:560), camera centre = c2w[:3,3] (dn_model.py:550)
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) nf[3 * r + c] = c2w[4 * c + r];
        nf[9] = t[0]; nf[10] = t[1]; nf[11] = t[2];
    }
}

__global__ __launch_bounds__(256) void densify_stats_kernel(int N, const int32_t *__restrict__ radii,
                                                            const float *__restrict__ xy_grads, int stride, float inv_max_size,
                                                            float *__restrict__ grad_norm, float *__restrict__ vis,
                                                            float *__restrict__ max2d)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    const int r = radii[g];
    if (r <= 0) return;
    const float gx = xy_grads[(size_t)g * stride], gy = xy_grads[(size_t)g * stride + 1];
    grad_norm[g] += sqrtf(gx * gx + gy * gy);
    vis[g] += 1.f;
    // torch divides a tensor by a python scalar as a multiply by its fp32 reciprocal; do the same
    max2d[g] = fmaxf(max2d[g], (float)r * inv_max_size);
}

// ---- densification decisions, one lane per Gaussian --------------------------------------------------
// refinement_after (dn_model.py:271-386) with the helpers it inherits from nerfstudio's SplatfactoModel (cull_gaussians):
// which Gaussians are split, duplicated, and which of {original, split child, duplicate} survive the cull.  Every rank of a
// data-parallel job evaluates this kernel on identical (all-reduced) statistics and identical parameters, so all replicas take
// the same decisions bit for bit.
__global__ __launch_bounds__(256) void densify_classify_kernel(dnsplat_densify_args a)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.N) return;
    const float s0 = a.scales[3 * g], s1 = a.scales[3 * g + 1], s2 = a.scales[3 * g + 2];
    const float smax = fmaxf(fmaxf(expf(s0), expf(s1)), expf(s2));                       // scales.exp().max(dim=-1)
    const float z = a.max_2Dsize ? a.max_2Dsize[g] : 0.f;
    // a split child holds log(exp(s) / 1.6) (split_gaussians); whoever looks at it later exponentiates that again
    const float c0 = expf(logf(expf(s0) / 1.6f)), c1 = expf(logf(expf(s1) / 1.6f)), c2 = expf(logf(expf(s2) / 1.6f));
    const float smax_child = fmaxf(fmaxf(c0, c1), c2);
    float smax_dup = smax;
    uint8_t f = 0;
    if (a.do_densify) {
        // avg_grad_norm = (xys_grad_norm / vis_counts) * 0.5 * max(H, W)                  dn_model.py:293-297
        const float avg = (a.xys_grad_norm[g] / a.vis_counts[g]) * 0.5f * a.max_image_side;
        const bool high = avg > a.densify_grad_thresh;
        bool split = smax > a.densify_size_thresh;
        if (a.screen_rules) split = split || (z > a.split_screen_size);
        split = split && high;
        // split_gaussians shrinks the parents' scales IN PLACE before the duplicate mask is formed (dn_model.py:309-315): a
        // split parent whose shrunk scale falls under the threshold is duplicated as well, with the shrunk scale
        if (split) smax_dup = smax_child;
        const bool dup = (smax_dup <= a.densify_size_thresh) && high;
        if (split) f |= DNSPLAT_DENSIFY_SPLIT;
        if (dup) f |= DNSPLAT_DENSIFY_DUP;
    }
    // cull_gaussians: sigmoid(opacity) < cull_alpha_thresh, plus "too big" once past the first opacity reset
    const float alpha = 1.f / (1.f + expf(-a.opacities[g]));
    const bool low = alpha < a.cull_alpha_thresh;
    bool big = false, big_child = false, big_dup = false;
    if (a.cull_big) {
        big = smax > a.cull_scale_thresh;
        big_dup = smax_dup > a.cull_scale_thresh;           // appended entries start with max_2Dsize = 0 (dn_model.py:324-332)
        if (a.screen_rules && a.max_2Dsize) big = big || (z > a.cull_screen_size);
        big_child = smax_child > a.cull_scale_thresh;
    }
    if ((f & DNSPLAT_DENSIFY_SPLIT) || low || big) f |= DNSPLAT_DENSIFY_CULL;      // a split parent is pruned (dn_model.py:338-352)
    if (low || big_child) f |= DNSPLAT_DENSIFY_CULL_CHILD;
    if (low || big_dup) f |= DNSPLAT_DENSIFY_CULL_DUP;
    a.flags[g] = f;
}

// New means / scales of the split children (nerfstudio split_gaussians): child j of parent p = parents[j % n_parents] takes
//     mean  = mean_p + R(q_p / |q_p|) (exp(scale_p) * noise_j),    scale = log(exp(scale_p) / 1.6)
// noise [n_children,3] ~ N(0,1) is drawn by the caller (identically on every rank).
__global__ __launch_bounds__(256) void densify_split_kernel(int n_children, int n_parents, const int32_t *__restrict__ parents,
                                                            const float *__restrict__ noise, const float *__restrict__ means,
                                                            const float *__restrict__ scales, const float *__restrict__ quats,
                                                            float *__restrict__ new_means, float *__restrict__ new_scales)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_children) return;
    const int p = parents[j % n_parents];
    float q[4];
    float n2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { q[i] = quats[4 * p + i]; n2 += q[i] * q[i]; }
    const float inv = 1.f / sqrtf(n2);
    const float w = q[0] * inv, x = q[1] * inv, y = q[2] * inv, z = q[3] * inv;
    const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y),
                        2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
                        2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)};
    float v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float e = expf(scales[3 * p + i]);
        v[i] = e * noise[3 * j + i];
        new_scales[3 * j + i] = logf(e / 1.6f);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
        new_means[3 * j + i] = (R[3 * i] * v[0] + R[3 * i + 1] * v[1] + R[3 * i + 2] * v[2]) + means[3 * p + i];
}

}  // namespace

extern "C" int dnsplat_densify_classify(const dnsplat_densify_args *a, dnsplat_stream_t stream)
{
    if (!a || a->N < 0) return DNSPLAT_ERR_INVALID_ARG;
    if (a->N == 0) return DNSPLAT_OK;
    if (!a->scales || !a->opacities || !a->flags) return DNSPLAT_ERR_INVALID_ARG;
    if (a->do_densify && (!a->xys_grad_norm || !a->vis_counts)) return DNSPLAT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(densify_classify_kernel, dim3((a->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, *a);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_densify_split(int32_t n_children, int32_t n_parents, const int32_t *parents, const float *noise,
                                     const float *means, const float *scales, const float *quats, float *new_means,
                                     float *new_scales, dnsplat_stream_t stream)
{
    if (n_children < 0 || n_parents < 0) return DNSPLAT_ERR_INVALID_ARG;
    if (n_children == 0) return DNSPLAT_OK;
    if (n_parents == 0 || !parents || !noise || !means || !scales || !quats || !new_means || !new_scales) return DNSPLAT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(densify_split_kernel, dim3((n_children + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_children,
                       n_parents, parents, noise, means, scales, quats, new_means, new_scales);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_densify_stats(int32_t N, const int32_t *radii, const float *xy_grads, int32_t grad_stride,
                                     float inv_max_size, float *xys_grad_norm, float *vis_counts, float *max_2Dsize,
                                     dnsplat_stream_t stream)
{
    if (N < 0 || grad_stride < 2) return DNSPLAT_ERR_INVALID_ARG;
    if (N == 0) return DNSPLAT_OK;
    if (!radii || !xy_grads || !xys_grad_norm || !vis_counts || !max_2Dsize) return DNSPLAT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(densify_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, N, radii, xy_grads,
                       grad_stride, inv_max_size, xys_grad_norm, vis_counts, max_2Dsize);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_dn_depth_normals(int32_t width, int32_t height, float fx, float fy, float cx, float cy,
                                        const float *depth, const float *alphas, const float *depth_max,
                                        float *depth_out, float *surface_normal, dnsplat_stream_t stream)
{
    if (width <= 0 || height <= 0 || !depth || !alphas || !depth_max || !depth_out || !surface_normal)
        return DNSPLAT_ERR_INVALID_ARG;
    const int P = width * height;
    hipLaunchKernelGGL(dn_depth_normals_kernel, dim3((P + 255) / 256), dim3(256), 0, (hipStream_t)stream, width, height,
                       fx, fy, cx, cy, depth, alphas, depth_max, depth_out, surface_normal);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}

extern "C" int dnsplat_camera_prepare(const float *c2w, float fx, float fy, float cx, float cy, float *viewmat, float *K,
                                      float *normal_frame, uint32_t *zero_word, int32_t n_zero_words, dnsplat_stream_t stream)
{
    if (!c2w || !viewmat || !K) return DNSPLAT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(camera_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c2w, fx, fy, cx, cy, viewmat, K,
                       normal_frame, zero_word, zero_word ? (n_zero_words > 0 ? n_zero_words : 1) : 0);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}
