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

__device__ __forceinline__ float filled_depth(const float *__restrict__ depth, const float *__restrict__ alphas,
                                              float dmax, int idx)
{
    return alphas[idx] > 0.f ? depth[idx] : dmax;
}

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
    const float dmax = *depth_max;
    depth_out[idx] = filled_depth(depth, alphas, dmax, idx);

    float n0 = 0.f, n1 = 0.f, n2 = 0.f;
    if (i >= 1 && i < H - 1 && j >= 1 && j < W - 1) {
        // back-projection with pixel centres at +0.5 (camera_utils.py:92-144), c2w = identity
        const float dl = filled_depth(depth, alphas, dmax, idx - 1), dr = filled_depth(depth, alphas, dmax, idx + 1);
        const float dt = filled_depth(depth, alphas, dmax, idx - W), db = filled_depth(depth, alphas, dmax, idx + W);
        const float x = (float)j + 0.5f, y = (float)i + 0.5f;
        const float lx = (x - 1.f - cx) * dl / fx, ly = (y - cy) * dl / fy, lz = dl;
        const float rx = (x + 1.f - cx) * dr / fx, ry = (y - cy) * dr / fy, rz = dr;
        const float tx = (x - cx) * dt / fx, ty = (y - 1.f - cy) * dt / fy, tz = dt;
        const float bx = (x - cx) * db / fx, by = (y + 1.f - cy) * db / fy, bz = db;
        const float ax = rx - lx, ay = ry - ly, az = rz - lz;     // left_to_right
        const float ux = tx - bx, uy = ty - by, uz = tz - bz;     // bottom_to_top
        float c0 = ay * uz - az * uy, c1 = az * ux - ax * uz, c2 = ax * uy - ay * ux;
        const float nrm = fmaxf(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), 1e-12f);   // F.normalize eps
        n0 = c0 / nrm; n1 = c1 / nrm; n2 = c2 / nrm;
    }
    // dn_model.py:599-603: @ diag(1,-1,-1), then (1 + n) / 2; the zero-padded border becomes 0.5
    surface_normal[3 * idx + 0] = (1.f + n0) / 2.f;
    surface_normal[3 * idx + 1] = (1.f - n1) / 2.f;
    surface_normal[3 * idx + 2] = (1.f - n2) / 2.f;
}

__global__ void camera_prepare_kernel(const float *__restrict__ c2w, float fx, float fy, float cx, float cy,
                                      float *__restrict__ viewmat, float *__restrict__ K, float *__restrict__ nf)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
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

}  // namespace

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
                                      float *normal_frame, dnsplat_stream_t stream)
{
    if (!c2w || !viewmat || !K) return DNSPLAT_ERR_INVALID_ARG;
    hipLaunchKernelGGL(camera_prepare_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, c2w, fx, fy, cx, cy, viewmat, K,
                       normal_frame);
    DNS_CHECK_LAUNCH();
    return DNSPLAT_OK;
}
