/*
 * dnsplat.h — C ABI of libdnsplat.so, the MI355X (gfx950) renderer that drops in
 * behind dn-splatter's DNSplatterModel.get_outputs().
 *
 * What it replaces.  The reference has no native code; its hot path is two calls
 * into the third-party CUDA package gsplat==1.0.0 (reference pin pyproject.toml:8):
 *   - gsplat.rendering.rasterization(...)   dn_splatter/dn_model.py:495-516
 *   - gsplat.rasterize_gaussians(...)       dn_splatter/dn_model.py:564-575
 * plus two helpers (dn_model.py:34-35).  The entry points below are the stages of
 * those two calls, cut where the reference's own
 * autograd graph is cut so that the Python binding in dn-splatter_amd/ can expose
 * exactly the tensors dn-splatter reads back (means2d incl. .grad/.absgrad, radii,
 * depths, conics, tiles_per_gauss — dn_model.py:517-524).
 *
 * Conventions.
 *   - Every pointer is a DEVICE pointer owned by the caller unless its name ends in
 *     _host.  Structs themselves live in host memory and are read during the call.
 *   - Kernels are enqueued on `stream`; no entry point allocates, frees or
 *     synchronises.  Workspaces are sized by the *_workspace_bytes queries.
 *   - Return value: 0 = ok, <0 = error (dnsplat_strerror).  No exceptions cross the ABI.
 *   - All floats are fp32, row-major, layouts as in the reference tensors:
 *     means[N,3], quats[N,4] (wxyz), scales[N,3], opacities[N], SH coefficients
 *     [N,K,3] (or split band-0 / rest, dn_model.py:466-468), viewmat[4,4] world->camera
 *     OpenCV, K[3,3].  Projection takes one camera per call (dn_model.py:421); binning and compositing also take
 *     batches of cameras (n_cameras).
 *   - The library keeps no global mutable state; calls are re-entrant across streams.
 *
 * Splat record.  Stage 1 packs what the compositing kernels gather per tile
 * intersection into one 64-byte record per Gaussian:
 *     [0]=x [1]=y [2..4]=conic(a,b,c) [5]=opacity [6..13]=up to 8 feature channels [14..15]=0
 * The gradient record written by dnsplat_raster_bwd mirrors it:
 *     [0..1]=v_xy [2..4]=v_conic [5]=v_opacity [6..13]=v_channels [14..15]=|v_xy| (absgrad, A11)
 */
#ifndef DNSPLAT_H
#define DNSPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever a struct gains a field, a field changes meaning or an entry point is added; the Python binding refuses a
 * library of another version.  Round 4: 10 = dnsplat_raster_args.det_partials / dnsplat_det_reduce (deterministic gradient
 * scatter), 11 = dnsplat_scale_reg, 12 = dnsplat_proj_grads.sh_factors (the colour-gradient slab written by the projection
 * backward), 13 = dnsplat_proj_out.tile_boxes carries width | height << 16 and dnsplat_bin_args.tile_boxes takes the counts
 * from it.  Round 6: 14 = dnsplat_scene.colors_are_logit (the sh_degree == 0 branch of get_outputs in the fused pass),
 * dnsplat_proj_out.skip_culled_records, dnsplat_proj_grads.sh_grad_scale / sh_zero_state (own-camera SH rows in the exchange step;
 * gradient rows of persistently culled Gaussians are not re-zeroed), dnsplat_sh_grads_add_factors, the packed (visible rows only)
 * colour-gradient slabs: dnsplat_visible_index, dnsplat_proj_grads.sh_packed, dnsplat_sh_grads_from_packed; 15 = dnsplat_ssim (the
 * SSIM term alone, for a loss stack that otherwise stays in PyTorch), dnsplat_proj_grads.zero_state_geometry (zero gradient rows of
 * culled Gaussians skipped per workgroup), dnsplat_edge_aware_logl1, dnsplat_tv_loss. */
#define DNSPLAT_ABI_VERSION 15
#define DNSPLAT_RECORD_FLOATS 16
#define DNSPLAT_MAX_CHANNELS 8

/* error codes */
#define DNSPLAT_OK 0
#define DNSPLAT_ERR_INVALID_ARG (-1)
#define DNSPLAT_ERR_WORKSPACE (-2)
#define DNSPLAT_ERR_LAUNCH (-3)
#define DNSPLAT_ERR_UNSUPPORTED (-4)

typedef void *dnsplat_stream_t; /* hipStream_t */

const char *dnsplat_strerror(int code);
int dnsplat_abi_version(void);

/* Appends the device's constant-rate wall clock (s_memrealtime ticks) to ring[(*cursor)++ % ring_size] from a one-thread kernel on
 * `stream`.  Works inside a frame captured into a HIP graph, where events cannot be recorded (ROCm 7.2): bench.py brackets the
 * dominant stage with two stamps per replay and calibrates the tick against HIP events. */
int dnsplat_stamp(uint64_t *ring, uint32_t *cursor, uint32_t ring_size, dnsplat_stream_t stream);

/* ------------------------------------------------------------------ stage 1
 * Fused per-Gaussian front end: activations (A0) + fully-fused projection (A1)
 * + tile count (A2a) + spherical harmonics (A5) + per-Gaussian normal (A7) +
 * record packing.  Replaces gsplat fully_fused_projection / spherical_harmonics
 * and the torch ops at dn_model.py:497-499, 543-560. */
typedef struct dnsplat_scene {
    int32_t N;
    const float *means;     /* [N,3] */
    const float *quats;     /* [N,4] wxyz, any norm (normalised inside, as gsplat does) */
    const float *scales;    /* [N,3] */
    const float *opacities; /* [N] */
    int32_t scales_are_log;       /* 1: apply exp()      (dn_model.py:498) */
    int32_t opacities_are_logit;  /* 1: apply sigmoid()  (dn_model.py:499) */
    /* colour source: SH if sh_degree >= 0, else direct colours */
    int32_t sh_degree;            /* active degree 0..3, or -1 */
    int32_t sh_K;                 /* number of SH bases held in storage (16 for degree-3 models); the
                                     K-1 higher bands live in shN, gradients of inactive bands are 0 */
    const float *sh0;             /* band 0, element g at sh0 + g*sh0_stride, 3 floats   */
    int32_t sh0_stride;
    const float *shN;             /* bands >= 1, element g at shN + g*shN_stride, 3*(K-1) floats */
    int32_t shN_stride;
    const float *colors;          /* direct colours [N, n_colors] when sh_degree < 0 */
    int32_t n_colors;             /* 0..8 */
    int32_t colors_are_logit;     /* 1 (ABI 14; direct colours only): apply sigmoid() — get_outputs with config.sh_degree == 0 feeds
                                     gsplat sigmoid(features_dc) and sh_degree = None (dn_model.py:486-493); the gradient written to
                                     v_colors is then w.r.t. the logits */
} dnsplat_scene;

typedef struct dnsplat_camera {
    const float *viewmat;      /* device [16] world->camera, row-major, OpenCV axes */
    const float *K;            /* device [9] */
    const float *normal_frame; /* device [12] or NULL: rows 0-8 = M with n_cam = M n_world
                                  (dn_model.py:560 `normals @ c2w[:3,:3]` => M = c2w[:3,:3]^T),
                                  9-11 = camera centre used for the facing flip (dn_model.py:550-556) */
    int32_t width, height, tile_size;
    float eps2d, near_plane, far_plane, radius_clip;
    int32_t antialiased;       /* rasterize_mode == "antialiased": opacity *= compensation */
    int32_t tight_tiles;       /* 0: tiles_per_gauss counts gsplat's 3-sigma tile box (A.3, what the drop-in calls return).
                                  1: the box of the part of the splat that can reach alpha >= 1/255 at a pixel centre, clipped
                                  to gsplat's box — shorter tile lists, same images and gradients; for callers that keep the
                                  lists to themselves (the fused get_outputs path).  dnsplat_bin_args.tight_tiles must match. */
} dnsplat_camera;

typedef struct dnsplat_proj_out {
    int32_t *radii;            /* [N] */
    float *means2d;            /* [N,2] */
    float *depths;             /* [N] */
    float *conics;             /* [N,3] */
    float *compensations;      /* [N] or NULL */
    int32_t *tiles_per_gauss;  /* [N] */
    float *splats;             /* [N,16] records, channels = colours | depth | normal */
    float *normals_world;      /* [N,3] or NULL: what dn_model.py:558 stores into gauss_params["normals"] */
    int32_t with_depth_channel;   /* append camera-space depth as a channel (RGB+D / RGB+ED) */
    int32_t with_normal_channels; /* append the 3 camera-frame normal channels (needs camera.normal_frame) */
    uint32_t *saturation_flag;    /* NULL, or a device word the caller zeroed: set to 1 when a visible Gaussian's opacity (after
                                     the antialiasing compensation) exceeds 0.999, i.e. when alpha = min(0.999, o x vis) can clamp
                                     at all in this frame (A.5).  dnsplat_raster_args.saturation_flag takes it. */
    int32_t *tiles_bin;           /* required iff camera.tight_tiles: [N] tile count over the TIGHT box (what dnsplat_bin_args.tiles_per_gauss
                                     must then be given); tiles_per_gauss itself always receives gsplat's count (A.3), which is what
                                     info["tiles_per_gauss"] / DNSplatterModel.num_tiles_hit (dn_model.py:524) report */
    int32_t *tile_boxes;          /* optional [N,2]: (id of the first tile, width | height << 16, in tiles; ABI 13) of the box the tile count
                                     was taken over — the tight one if camera.tight_tiles; width x height IS that count (0 | 0 for a culled
                                     Gaussian).  dnsplat_bin_args.tile_boxes takes it and saves the binning a gather of the records, the box
                                     arithmetic per Gaussian and the separate gather of the count */
    int32_t phase;                /* 0: everything in one launch.  SH colours only (scene.sh_degree >= 0): 1 = all outputs except the three
                                     colour channels of the records (left 0; no coefficient is read), 2 = those three channels for the
                                     Gaussians with radii > 0 (reads radii and the records' location only).  A caller may run 2 on
                                     another stream beside dnsplat_bin_*: binning reads nothing phase 2 writes. */
    int32_t skip_culled_records;  /* 1 (ABI 14): the 64-byte records of culled Gaussians (radii == 0) are NOT written (left as the caller
                                     allocated them): nothing downstream reads them — the tile lists hold visible Gaussians only — and
                                     they are 28 % of the record bytes on the benchmark scenes.  0: zero-filled, as before */
} dnsplat_proj_out;

int dnsplat_project_fwd(const dnsplat_scene *scene, const dnsplat_camera *cam,
                        const dnsplat_proj_out *out, dnsplat_stream_t stream);

/* Packs caller-provided 2-D splats into records; the front end of the legacy
 * gsplat.rasterize_gaussians call (dn_model.py:564-575) whose inputs are already
 * projected. colors [N,C], C <= 8. */
int dnsplat_pack_splats(int32_t N, const float *means2d, const float *conics, const float *opacities,
                        const float *colors, int32_t C, float *splats, dnsplat_stream_t stream);

/* ------------------------------------------------------------------ stage 2
 * Tile binning.  Replaces gsplat isect_tiles + radix sort + isect_offset_encode
 * (A2-A4) and, for the legacy call, map_gaussian_to_intersects + torch.sort +
 * get_tile_bin_edges (A8).  Produces the same ordering — ascending
 * (tile, depth bits), ties by Gaussian index — by (1) a stable 32-bit radix sort
 * of the Gaussians on depth, (2) emission of (tile, gaussian) pairs in that order
 * and (3) a stable radix sort on the tile id only. */
size_t dnsplat_bin_workspace_bytes(int32_t N, int64_t isect_capacity, int32_t n_tiles);

/* Byte offset, inside a workspace of that geometry, of a reserved 32-bit status word.  dnsplat_bin_emit_sort sets it to 0 and no
 * kernel of this build ever raises it: the tile sort has no inter-workgroup wait left that could time out (the decoupled look-back
 * of an earlier build, whose 20 ms bound this word reported, was removed — DESIGN.md 3.1).  Kept so that the workspace layout and
 * the symbol stay stable; tests read it after the large frames and expect 0. */
size_t dnsplat_bin_status_offset(int32_t N, int64_t isect_capacity);

typedef struct dnsplat_bin_args {
    int32_t N;                       /* entries = n_cameras x Gaussians; entry cam * (N / n_cameras) + g is Gaussian g seen by camera cam */
    int32_t n_cameras;               /* >= 1; images of one batch share width / height (gsplat rasterization with C cameras) */
    int32_t width, height, tile_size;
    const float *means2d;            /* [N,2] */
    const int32_t *radii;            /* [N] */
    const float *depths;             /* [N] */
    const int32_t *tiles_per_gauss;  /* [N] */
    int64_t isect_capacity;          /* entries flatten_ids can hold */
    int32_t *flatten_ids;            /* out [isect_capacity]: Gaussian index per sorted intersection */
    int32_t *tile_offsets;           /* out [n_cameras*n_tiles+1]: first sorted index per (camera, tile); last = n_isects */
    int64_t *n_isects;               /* out device scalar: true number of intersections (may exceed capacity) */
    int64_t *n_isects_host;          /* optional pinned host mirror, written by an async D2H copy; NULL to skip */
    void *workspace;
    size_t workspace_bytes;
    const float *splats;             /* [N,16] records of stage 1; read only when tight_tiles != 0 */
    int32_t tight_tiles;             /* as dnsplat_camera.tight_tiles of the projection that produced tiles_per_gauss */
    int32_t *tile_ends;              /* optional out [n_cameras*n_tiles]: one past the last sorted index of every non-empty tile (0 for an
                                        empty one).  dnsplat_raster_args.tile_ends takes it. */
    int32_t skip_offsets_fill;       /* with tile_ends: leave tile_offsets[t] of EMPTY tiles undefined (>= the tile's end) instead of
                                        giving them gsplat's value (the offset of the next non-empty tile) — one launch less */
    const int32_t *tile_boxes;       /* optional [N,2] = dnsplat_proj_out.tile_boxes of the projection that produced tiles_per_gauss (single
                                        camera or a batch; for camera c > 0 the kernel adds c * n_tiles to the first tile id).  With it the
                                        counts are taken from the boxes themselves (width x height, ABI 13): tiles_per_gauss must be the
                                        counts of exactly these boxes */
    int64_t *n_isects_max;           /* optional device scalar the caller zeroes once: running maximum of n_isects over the frames binned
                                        since.  For a host that never waits on a single frame's count (replayed HIP graphs). */
} dnsplat_bin_args;

/* 2a: depth sort + inclusive offsets + total.  After this (and a stream sync or
 * a look at n_isects_host) the caller knows how large flatten_ids must be. */
int dnsplat_bin_prepare(const dnsplat_bin_args *args, dnsplat_stream_t stream);
/* 2b: emit + tile sort + offsets.  If n_isects > isect_capacity nothing past the
 * capacity is written and tile_offsets are clamped: the caller must re-run with
 * a larger capacity (it detects this from n_isects). */
int dnsplat_bin_emit_sort(const dnsplat_bin_args *args, dnsplat_stream_t stream);
/* Reconstructs gsplat's 64-bit isect_ids (camera << (32 + tile bits) | tile << 32 | depth bits, tile bits =
 * floor(log2(n_tiles)) + 1) for the sorted list — only needed to populate the `isect_ids` entry of the info
 * dict.  n_tiles = tiles per camera. */
int dnsplat_bin_isect_ids(int32_t n_tiles, int32_t n_cameras, const int32_t *tile_offsets, const int32_t *flatten_ids,
                          const float *depths, int64_t *isect_ids, int64_t capacity, dnsplat_stream_t stream);

/* ------------------------------------------------------------------ stage 3/4
 * Per-tile front-to-back alpha compositing of D feature channels (A6 + A8 in one
 * pass) and its backward. */
/* Optional dn-splatter epilogue fused into the compositing kernels: the per-pixel
 * post-ops DNSplatterModel.get_outputs applies to the two gsplat results (dn_model.py:526-537, 577-578).
 * Only for the fused 7-channel layout rgb | expected depth | normal (D == 7, ed_channel == 3,
 * xy_split == 4, background == {0,0,0,0,1,1,1}).  Forward additionally writes
 *     rgb    = clamp(render_rgb + (1 - alpha) * background_rgb, 0, 1)          dn_model.py:526-528
 *     depth  = expected depth as composited (the alpha == 0 fill needs the image-wide maximum, which
 *              the forward reduces into *depth_max; dnsplat_dn_depth_normals applies it)  :533-537
 *     normal = (n / |n| + 1) / 2 with n the composited normal incl. its ones-background  :577-578
 * and backward takes the cotangents of those images (and of accumulation = alpha) instead of
 * v_render / v_alphas. */
typedef struct dnsplat_dn_post {
    const float *background_rgb; /* device [3] */
    float *rgb;                  /* out [H,W,3] */
    float *depth;                /* out [H,W] expected depth, unfilled */
    float *normal;               /* out [H,W,3] */
    float *depth_max;            /* device [n_cameras], caller zero-fills; receives max over each image of `depth` */
    const float *v_rgb;          /* backward: [H,W,3] */
    const float *v_depth;        /* backward: [H,W] cotangent of the FILLED depth image (masked by alpha > 0 inside) */
    const float *v_normal;       /* backward: [H,W,3] */
    const float *v_accumulation; /* backward: [H,W] or NULL */
} dnsplat_dn_post;

typedef struct dnsplat_raster_args {
    int32_t width, height, tile_size;   /* tile_size must be 16 */
    int32_t D;                          /* feature channels, 1..8 */
    const float *splats;                /* [N,16]; splats, flatten_ids and v_splats may be NULL iff every tile list is empty */
    const int32_t *flatten_ids;
    const int32_t *tile_offsets;        /* [n_tiles+1] */
    const float *background;            /* device [D] or NULL (gsplat v1: none; legacy normal pass: ones) */
    int32_t ed_channel;                 /* channel normalised by max(alpha,1e-10) ("ED"), or -1 */
    float *render;                      /* [H,W,D] */
    float *alphas;                      /* [H,W] */
    int32_t *last_ids;                  /* [H,W] where the backward starts its walk of the pixel's list: a sorted index >= that of the
                                           last splat applied such that no entry in between applies to the pixel (undefined where alpha == 0) */
    /* backward only */
    const float *v_render;              /* [H,W,D] */
    const float *v_alphas;              /* [H,W] or NULL */
    int32_t xy_split;                   /* channels >= xy_split do not feed v_xy/|v_xy|: the reference renders
                                           them with xys.detach() (dn_model.py:562). Use D for "all feed". */
    float *v_splats;                    /* [N,16] gradient records, accumulated into (caller zero-fills) */
    const dnsplat_dn_post *dn;          /* NULL, or the fused dn-splatter epilogue (then v_render / v_alphas are unused) */
    int32_t n_cameras;                  /* 0 or 1: one image.  C > 1: images [C,H,W,.] stacked, tile_offsets [C*n_tiles+1], splat /
                                           gradient records [C*N,16] (record cam*N + g), as produced by a dnsplat_bin_args batch */
    uint64_t *keep_masks;               /* NULL, or device [2 * keep_mask_stride] 64-bit words handed from the forward to the backward:
                                           for every 16x8 half tile and every batch of 64 consecutive list entries, which entries
                                           passed the forward's rectangle test (can reach alpha >= 1/255 somewhere in the half tile).
                                           Word of (list l = camera*n_tiles + tile, half h, batch b): [h * stride + (tile_offsets[l] >> 6) + l + b].
                                           The backward then skips its own test and never gathers the records of rejected entries. */
    int64_t keep_mask_stride;           /* >= (isect capacity >> 6) + n_cameras * n_tiles + 1 */
    uint64_t *pair_counters;            /* NULL, or device [8] (measurement builds of the fused pass, bench.py's VALU roofline), added to:
                                           forward  [0] list entries examined  [1] splats walked (kept by the rectangle test)
                                                    [2] live (pixel, splat) pairs evaluated  [3] pairs blended
                                           backward [4] (pixel, splat) slots issued (steps x 128)  [5] pairs replayed */
    const uint32_t *saturation_flag;    /* NULL, or dnsplat_proj_out.saturation_flag of the projection(s) behind `splats`: if the word
                                           is 0 no pair of this launch can clamp and dnsplat_raster_bwd (only; the forward ignores it) runs its step loop without
                                           the clamp handling (same results, ~3.6 % fewer cycles); read on the device, no host sync */
    const int32_t *tile_ends;           /* NULL: tile t's list is [tile_offsets[t], tile_offsets[t+1]).  Else dnsplat_bin_args.tile_ends:
                                           the list is [tile_offsets[t], tile_ends[t]) and empty where tile_ends[t] <= tile_offsets[t] */
    void *zero_fill;                    /* dnsplat_raster_fwd only: NULL, or a device buffer of zero_fill_bytes (multiple of 16) the launch clears
                                           on the way — meant for the v_splats buffer the backward of the same frame accumulates into */
    int64_t zero_fill_bytes;
    float *det_partials;                /* dnsplat_raster_bwd only: NULL (gradient records accumulated into v_splats with fp32 atomics, in arrival
                                           order), or a ZERO-FILLED device buffer [2, det_capacity, 16]: the deterministic mode.  The row a 16x8 half
                                           tile (half h) contributes to the splat at sorted list index i is then STORED at [h][i][:] and v_splats is
                                           not touched; dnsplat_det_reduce adds the rows up per gradient record in a fixed order */
    int64_t det_capacity;               /* >= the isect capacity of the lists (entries flatten_ids can hold) */
} dnsplat_raster_args;

int dnsplat_raster_fwd(const dnsplat_raster_args *args, dnsplat_stream_t stream);
int dnsplat_raster_bwd(const dnsplat_raster_args *args, dnsplat_stream_t stream);

/* Deterministic reduction of the rows dnsplat_raster_bwd left in det_partials (test / debug mode, DNSPLAT_DETERMINISTIC=1 in the
 * Python binding): the sorted list is stably re-sorted by record id (hand-written LSD radix, as the binning), and each record g
 * that has list entries receives
 *     v_splats[g][:] = sum over its entries in ascending list order of (partials[0][entry] + partials[1][entry]),
 * accumulated in float64 and rounded once.  Rows of records without entries are left as they are (the caller zero-fills).  Same
 * inputs give the same bits in every run, whatever order the compositing workgroups finished in.  Enqueued on `stream`, no sync. */
typedef struct dnsplat_det_args {
    int32_t n_records;          /* gradient records (cameras x Gaussians) */
    int64_t capacity;           /* dnsplat_raster_args.det_capacity */
    const int64_t *n_isects;    /* device scalar: number of valid list entries (dnsplat_bin_args.n_isects); clamped to capacity */
    const int32_t *flatten_ids; /* [capacity] the sorted list the compositing walked */
    const float *partials;      /* [2, capacity, 16] */
    float *v_splats;            /* [n_records, 16] */
    void *workspace;            /* dnsplat_det_workspace_bytes(capacity) */
    size_t workspace_bytes;
} dnsplat_det_args;
size_t dnsplat_det_workspace_bytes(int64_t capacity);
int dnsplat_det_reduce(const dnsplat_det_args *args, dnsplat_stream_t stream);

/* Depth fill + depth->normal stencil of get_outputs (dn_model.py:533-537 and 589-603 with
 * utils/normal_utils.py:9-48, utils/camera_utils.py:92-144, called with c2w = identity):
 *     depth_out      = alpha > 0 ? depth : *depth_max
 *     surface_normal = (1 + diag(1,-1,-1) * normalize(cross(right - left, top - bottom))) / 2 of the
 *                      back-projected (pixel centre + 0.5) depth_out; 0.5 on the one-pixel border.
 * No gradient flows through it in the reference (depth.detach(), dn_model.py:590). */
int dnsplat_dn_depth_normals(int32_t width, int32_t height, float fx, float fy, float cx, float cy,
                             const float *depth, const float *alphas, const float *depth_max,
                             float *depth_out /* [H,W] */, float *surface_normal /* [H,W,3] */,
                             dnsplat_stream_t stream);

/* nerfstudio get_viewmat + intrinsics + normal frame in one launch: from the camera-to-world matrix
 * c2w [3,4] (OpenGL axes, device) writes viewmat[16] (world->camera, OpenCV), K[9] and
 * normal_frame[12] (see dnsplat_camera).  Replaces ~20 tiny torch kernels per frame (dn_model.py:475-479).
 * zero_word (optional): n_zero_words (>= 1) device words set to 0 by the same launch — the frame's
 * dnsplat_proj_out.saturation_flag and dnsplat_dn_post.depth_max start here without fill launches of their own. */
int dnsplat_camera_prepare(const float *c2w, float fx, float fy, float cx, float cy,
                           float *viewmat, float *K, float *normal_frame, uint32_t *zero_word, int32_t n_zero_words,
                           dnsplat_stream_t stream);

/* Multi-view data parallelism, compact exchange of the SH gradients.  For one camera the gradient of Gaussian g's SH
 * coefficients is an outer product  v_coeff[g][k][c] = basis_k(dir_g) * v_colour[g][c]  (k < (degree+1)^2, c < 3), and the
 * direction dir_g = normalize(mean_g - camera position) is something every rank can work out for every camera: the means are
 * replicated and a camera position is 12 bytes.  So what travels per (camera, Gaussian) is the clamp-masked colour gradient
 * alone — 12 bytes instead of the 192 bytes of coefficient gradients: with one camera per GPU the ranks all-gather
 * [N,3] slabs (7 x 12 B received per Gaussian at 8 GPUs instead of ~2 x 7/8 x 192 B of an all-reduce) and every rank rebuilds
 *     v_coeff = scale * sum_views basis(normalize(mean - pos_view)) (x) v_colour_view
 * with dnsplat_sh_grads_from_factors.
 * factors: n_views slabs of 3 N + 4 floats: [N,3] colour gradients | camera position (3) | 0, as dnsplat_sh_factors writes one.
 * means: the [N,3] means of the scene.  Layouts of v_sh0 / v_shN as dnsplat_scene.sh0 / shN. */
int dnsplat_sh_grads_from_factors(int32_t N, int32_t n_views, const float *factors, const float *means, int32_t sh_degree,
                                  int32_t sh_K, float scale, float *v_sh0, int32_t v_sh0_stride, float *v_shN,
                                  int32_t v_shN_stride, dnsplat_stream_t stream);

/* One camera's slab (3 N + 4 floats, see above) from the forward's outputs (radii, viewmat of dnsplat_camera, splats =
 * dnsplat_proj_out.splats) and the gradient records of dnsplat_raster_bwd — without the rest of the projection backward, so
 * that the host can start the all-gather BEFORE dnsplat_project_bwd (called with sh_grads_skip = 1) and the exchange runs while
 * the geometry gradients are still being computed.  A colour that sits exactly on the clamp (c + 0.5 == 0) is masked. */
int dnsplat_sh_factors(int32_t N, const int32_t *radii, const float *viewmat, const float *splats, const float *v_splats,
                       float *factors, dnsplat_stream_t stream);

/* ABI 14.  As dnsplat_sh_grads_from_factors, but ADDS  scale * sum over the views v != skip_view  to rows that already hold the
 * (pre-scaled, dnsplat_proj_grads.sh_grad_scale) contribution of view skip_view — the rank's own camera, whose rows
 * dnsplat_project_bwd wrote itself as on a single GPU.  With n_views == 1 there is nothing to add and nothing is launched: the
 * exchange step costs a single rank no pass over the 192 B / Gaussian of coefficient gradients.  (At n_views >= 2 the read-modify-
 * write moves more bytes than rebuilding every row from the slabs — dp.ShFactorExchange picks per world size.) */
int dnsplat_sh_grads_add_factors(int32_t N, int32_t n_views, int32_t skip_view, const float *factors, const float *means,
                                 int32_t sh_degree, int32_t sh_K, float scale, float *v_sh0, int32_t v_sh0_stride, float *v_shN,
                                 int32_t v_shN_stride, dnsplat_stream_t stream);

/* ABI 14.  Slabs of visible rows only.  A camera sees ~72 % of the Gaussians of the benchmark scenes; the colour gradients of the
 * others are zero and need not travel.  One camera's PACKED slab (all ranks agree on `capacity` rows):
 *     header  [0] count of visible Gaussians (uint32; > capacity = overflow: rows beyond it were dropped, the caller must notice)
 *             [1..3] camera centre (float)        [4..7] reserved
 *     masks   ceil(N / 64) 64-bit words, bit = radii > 0
 *     offsets ceil(N / 64) uint32: visible Gaussians in front of the word's block (exclusive prefix sum of the popcounts)
 *     rows    capacity x 3 floats: the clamp-masked colour gradient of the k-th visible Gaussian
 * dnsplat_packed_slab_floats gives the size (in 4-byte words, a multiple of 4).  dnsplat_visible_index writes header, masks and
 * offsets from the forward's radii (two small launches; `scratch`: ceil(N / 64) uint32); dnsplat_project_bwd fills the rows
 * (dnsplat_proj_grads.sh_packed); dnsplat_sh_grads_from_packed is dnsplat_sh_grads_from_factors / _add_factors over n_views such slabs
 * laid out `slab_floats` words apart (skip_view < 0: rebuild every row; >= 0: add the other views to the rows in place). */
size_t dnsplat_packed_slab_floats(int32_t N, int32_t capacity);
int dnsplat_visible_index(int32_t N, int32_t capacity, const int32_t *radii, const float *viewmat, float *slab, uint32_t *scratch,
                          dnsplat_stream_t stream);
int dnsplat_sh_grads_from_packed(int32_t N, int32_t capacity, int32_t n_views, int32_t skip_view, const float *slabs,
                                 const float *means, int32_t sh_degree, int32_t sh_K, float scale, float *v_sh0, int32_t v_sh0_stride,
                                 float *v_shN, int32_t v_shN_stride, dnsplat_stream_t stream);

/* Densification statistics: the per-step accumulation nerfstudio's SplatfactoModel.after_train
 * performs on the renderer's outputs (called at dn_model.py:938-942, consumed by refinement_after dn_model.py:286-296):
 * for every Gaussian with radii > 0
 *     xys_grad_norm += |(gx, gy)|,  vis_counts += 1,  max_2Dsize = max(max_2Dsize, radii * inv_max_size),  inv_max_size = 1 / max(W, H).
 * xy_grads rows are grad_stride floats apart (2 for a [N,2] tensor, 16 to read the |v_xy| columns of the gradient
 * records in place: pass v_splats + 14). */
int dnsplat_densify_stats(int32_t N, const int32_t *radii, const float *xy_grads, int32_t grad_stride, float inv_max_size,
                          float *xys_grad_norm, float *vis_counts, float *max_2Dsize, dnsplat_stream_t stream);

/* Densification decisions of refinement_after (dn_model.py:271-386; cull rules of the inherited nerfstudio
 * SplatfactoModel.cull_gaussians), one flag byte per Gaussian.  do_densify = the `do_densification` branch is active;
 * screen_rules = step < stop_screen_size_at; cull_big = step > refine_every * reset_alpha_every.  Appended entries (split
 * children, duplicates) enter the cull with max_2Dsize = 0, children with scales log(exp(s) / 1.6). */
#define DNSPLAT_DENSIFY_SPLIT 1       /* split into n_split_samples children; the parent is then pruned */
#define DNSPLAT_DENSIFY_DUP 2         /* duplicated once */
#define DNSPLAT_DENSIFY_CULL 4        /* the original entry is removed */
#define DNSPLAT_DENSIFY_CULL_CHILD 8  /* its split children would be removed right away */
#define DNSPLAT_DENSIFY_CULL_DUP 16   /* its duplicate would be removed right away */
typedef struct dnsplat_densify_args {
    int32_t N;
    const float *scales;        /* [N,3] log-scales */
    const float *opacities;     /* [N] logits */
    const float *xys_grad_norm; /* [N] accumulated by dnsplat_densify_stats (all-reduced over the ranks when data parallel) */
    const float *vis_counts;    /* [N] */
    const float *max_2Dsize;    /* [N] or NULL */
    int32_t do_densify, screen_rules, cull_big;
    float max_image_side;       /* max(H, W) of the last rendered frame (dn_model.py:296) */
    float densify_grad_thresh, densify_size_thresh, split_screen_size;
    float cull_alpha_thresh, cull_scale_thresh, cull_screen_size;
    uint8_t *flags;             /* out [N] */
} dnsplat_densify_args;
int dnsplat_densify_classify(const dnsplat_densify_args *args, dnsplat_stream_t stream);

/* Means and log-scales of the split children (nerfstudio split_gaussians): child j belongs to parent parents[j % n_parents]
 * (samples are laid out sample-major, as `repeat(samps, 1)` does):
 *     mean = mean_p + R(q_p / |q_p|) (exp(scale_p) * noise_j),   scale = log(exp(scale_p) / 1.6). */
int dnsplat_densify_split(int32_t n_children, int32_t n_parents, const int32_t *parents, const float *noise /* [n_children,3] */,
                          const float *means, const float *scales, const float *quats, float *new_means /* [n_children,3] */,
                          float *new_scales /* [n_children,3] */, dnsplat_stream_t stream);

/* dn-splatter's per-pixel training loss and its gradient w.r.t. the rendered images in two launches:
 *   loss = (1 - l) mean|rgb - gt| + l (1 - SSIM(rgb, gt))                       nerfstudio splatfacto RGB term, l = ssim_lambda
 *        + depth_weight (EdgeAwareLogL1_x + EdgeAwareLogL1_y)(depth, gt_depth)  losses.py:187-224, mask gt_depth > tolerance,
 *                                                                               depth_weight = 1 + depth_lambda (regularization_strategy.py:184)
 *        + mean|normal - gt_normal| + TV(normal)                                regularization_strategy.py:188-193, losses.py:279-295
 * (dn_model.py:614-729; the per-Gaussian min-scale term is added by the caller).  Outputs the cotangents v_rgb,
 * v_depth, v_normal of a unit loss gradient and, in sums[8], the partial sums
 *   {sum SSIM, sum|rgb-gt|, sum EA_x, sum EA_y, sum|n-gt_n|, sum TV_h, sum TV_w, 0} from which the loss value follows. */
typedef struct dnsplat_dn_loss_args {
    int32_t width, height;
    const float *rgb;          /* [H,W,3] rendered */
    const float *depth;        /* [H,W]   rendered (filled) depth */
    const float *normal;       /* [H,W,3] rendered normal image in [0,1] */
    const float *gt_rgb;       /* [H,W,3] */
    const float *gt_depth;     /* [H,W] or NULL (no depth loss) */
    const float *gt_normal;    /* [H,W,3] or NULL (no normal losses) */
    const float *depth_counts; /* device [2]: number of pixels with gt_depth > tolerance in columns < W-1, in rows < H-1 */
    float ssim_lambda, depth_weight, depth_tolerance;
    float *maps;               /* scratch, 9 H W + 512 floats (512 partial-sum words in front of three [H,W,3] maps) */
    float *v_rgb, *v_depth, *v_normal;   /* out, shapes of rgb / depth / normal */
    float *sums;               /* out device [8] */
} dnsplat_dn_loss_args;

int dnsplat_dn_loss(const dnsplat_dn_loss_args *args, dnsplat_stream_t stream);

/* ABI 15.  The SSIM term ALONE, for a loss stack that otherwise stays in PyTorch (BASELINE.json: "losses stay in PyTorch-ROCm"):
 * what nerfstudio splatfacto's `self.ssim(gt, pred)` computes inside the RGB term dn-splatter inherits (dn_model.py:624-627, :663) —
 * pytorch_msssim.SSIM(data_range=1, size_average=True, channel=3): 11-tap Gaussian sigma 1.5, valid padding, mean over the
 * (W-10)(H-10) windows of the 3 channels.  MIOpen runs the sixteen depthwise conv2d calls of that module and its backward at
 * 250-500 us each on a 1600 x 1200 image (profiles/r06_c5_torch_loss_kernel_stats.txt).
 *   sums[0] = sum of the per-window SSIM values (mean = sums[0] / (3 (W-10)(H-10))),
 *   v_x     = d(mean SSIM)/dx, [H,W,3], or NULL (value only);   x, y: [H,W,3] fp32;   maps: scratch, 9 H W + 512 floats. */
int dnsplat_ssim(int32_t width, int32_t height, const float *x, const float *y, float *maps, float *v_x, float *sums /* device [8] */,
                 dnsplat_stream_t stream);

/* ABI 15.  Two more MODULES of the reference's loss stack one by one (the rest stays PyTorch), behind drop-in nn.Modules
 * (fused_loss.EdgeAwareLogL1 / TVLoss, install_losses(model)):
 * dnsplat_edge_aware_logl1 — losses.py:187-224, "scalar" implementation, as DNRegularization.get_depth_loss calls it
 *   (regularization_strategy.py:162-170): pred, gt [H,W] fp32, rgb [H,W,3], mask [H,W] bytes (torch.bool) or NULL.
 *   sums[0..3] = { sum lambda_x log(1+|d|), sum lambda_y log(1+|d|), pixels counted in x, pixels counted in y }  (loss = s0/s2 + s1/s3;
 *   the reference's boolean-mask gathers `loss_x[mask]` have a data-dependent size = a host synchronisation per call: none here);
 *   v_x, v_y [H,W] (both or neither): d s0 / d pred, d s1 / d pred — the caller divides by the counts.
 * dnsplat_tv_loss — losses.py:279-295 on an [H,W,C] image: sums[0..1] = { sum |p - right|, sum |p - lower| },
 *   v_pred = d/d pred of the loss s0 / (H (W-1) C) + s1 / ((H-1) W C) (or NULL).
 * scratch: 512 floats; sums: device [8]. */
int dnsplat_edge_aware_logl1(int32_t width, int32_t height, const float *pred, const float *gt, const float *rgb, const uint8_t *mask,
                             float *v_x, float *v_y, float *scratch, float *sums, dnsplat_stream_t stream);
int dnsplat_tv_loss(int32_t width, int32_t height, int32_t channels, const float *pred, float *v_pred, float *scratch, float *sums,
                    dnsplat_stream_t stream);

/* The per-Gaussian term of the same loss (regularization_strategy.py:195-199): mean_g min_k exp(scales[g][k]).  Adds
 * weight * sum_g min_k exp(s_gk) to *sum (device scalar, caller zeroes it) and WRITES the gradient rows
 * v_scales[g][k] = [k == argmin_k] * weight * exp(s_gk); weight = 1 / N gives the mean.  Replaces torch's exp / min / mean and their
 * five backward kernels over [N,3] tensors. */
int dnsplat_scale_reg(int32_t N, const float *scales_log, float weight, float *v_scales /* [N,3] */, float *sum, dnsplat_stream_t stream);

/* ------------------------------------------------------------------ stage 5
 * Fused per-Gaussian back end: gradient record -> parameter gradients.
 * Replaces gsplat fully_fused_projection_bwd (A10), spherical_harmonics bwd (A5),
 * the autograd of dn_model.py:543-560 (A7) and of the activations (A0). */
typedef struct dnsplat_proj_grads {
    const int32_t *radii;        /* [N] from the forward */
    const float *v_splats;       /* [N,16] gradient records */
    const float *v_means2d;      /* optional [N,2]: if non-NULL REPLACES record columns 0-1 (the binding
                                    routes means2d through autograd so callers may add their own terms) */
    const float *v_depths;       /* optional [N], added */
    const float *v_conics;       /* optional [N,3], added */
    const float *v_compensations;/* optional [N] */
    float *v_means;              /* [N,3] */
    float *v_quats;              /* [N,4] */
    float *v_scales;             /* [N,3] w.r.t. the tensor passed in (log-scales if scales_are_log) */
    float *v_opacities;          /* [N]   w.r.t. the tensor passed in */
    float *v_sh0; int32_t v_sh0_stride;   /* like scene.sh0 / shN; NULL to skip */
    float *v_shN; int32_t v_shN_stride;
    float *v_colors;             /* [N,n_colors] when sh_degree < 0; NULL to skip */
    int32_t sh_grads_skip;       /* non-zero: the SH coefficient gradients are not written — the caller took the colour gradients
                                    with dnsplat_sh_factors before this launch and rebuilds the rows from the gathered slabs
                                    (dnsplat_sh_grads_from_factors) */
    float *sh_factors;           /* optional [3 N + 4] (ABI 12; SH scenes only): this launch ALSO writes the slab dnsplat_sh_factors
                                    produces — per Gaussian the three colour gradients behind the clamp (zeros when culled), then the
                                    camera centre and a pad word — from the values it holds anyway, which saves that kernel's launch
                                    and its 128 B / Gaussian of reads.  For callers whose exchange starts after this launch (a
                                    captured step: graph.GraphedDpStep); dnsplat_sh_factors stays for those that start it before */
    float sh_grad_scale;         /* ABI 14.  0 or 1: as before.  Otherwise the SH coefficient gradient rows this launch writes (v_sh0 / v_shN)
                                    are multiplied by it — 1 / world size when the rank writes its OWN camera's rows itself and
                                    dnsplat_sh_grads_add_factors adds the other cameras' (the mean over cameras; geometry gradients are not
                                    scaled: their all-reduce averages) */
    uint64_t *sh_zero_state;     /* ABI 14.  NULL, or device [ceil(N / 64)] words owned by whoever owns the v_sh0 / v_shN buffers (split or
                                    [N,16,3] layout, K = 16): bit g % 64 of word g / 64 set = "the coefficient-gradient row of Gaussian g
                                    is zero in memory".  The launch skips the stores of rows that are culled now AND known zero, and
                                    leaves the word describing what memory holds afterwards (culled rows: zero).  Contract: nobody else
                                    writes non-zero values into those rows without clearing the words (dp.GradArena.invalidate_sh_state).
                                    Ignored with sh_grads_skip. */
    float *sh_packed;            /* ABI 14.  NULL, or this camera's PACKED slab, whose header / masks / offsets dnsplat_visible_index wrote for
                                    the same radii (see there): the launch fills its rows — the colour gradients of the VISIBLE Gaussians
                                    only, row offsets[g / 64] + popcount(mask bits below g % 64) — beside or instead of sh_factors.  Whole
                                    scenes only (the slices of dp.SlicedShExchange keep dense slabs). */
    int32_t zero_state_geometry; /* ABI 15.  1: the words of sh_zero_state describe ALL gradient rows of a Gaussian that this launch writes —
                                    v_means / v_quats / v_scales / v_opacities beside the coefficient rows (their owner keeps the same
                                    contract for them).  Rows are then skipped per WORKGROUP of 64 Gaussians, all or nothing: a workgroup
                                    whose rows are all culled now and all known zero writes nothing at all (236 B per Gaussian); any other
                                    workgroup stores its whole span as without the state.  Pays when the rows follow a space-filling
                                    curve (densify.spatial_order: a camera's culled Gaussians are then runs of rows). */
} dnsplat_proj_grads;

int dnsplat_project_bwd(const dnsplat_scene *scene, const dnsplat_camera *cam,
                        const dnsplat_proj_out *fwd, const dnsplat_proj_grads *grads,
                        dnsplat_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DNSPLAT_H */
