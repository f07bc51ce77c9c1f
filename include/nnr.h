/*
 * nnr.h -- C ABI of libnnr.so: the MI355X (gfx950) native volumetric-rendering hot path of NoPe-NeRF.
 *
 * The reference (ActiveVisionLab/nope-nerf) is pure Python on PyTorch and has no FFI of its own;
 * the drop-in boundary is the Python class surface `model.Renderer.nope_nerf`
 * (reference model/rendering.py:36-167) and the MLP it calls, `model.OfficialStaticNerf.forward`
 * (reference model/official_nerf.py:60-96).  This header is the C-ABI those two replace their bodies
 * with: plain device pointers and sizes, no torch types, no C++ types, no exceptions.  The ctypes
 * binding a maintainer adds on the reference side is shown in INTEGRATION.md and lives in
 * nope-nerf_amd/nnr/lib.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - every function is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant, and
 *     keeps no global state; all memory is owned and allocated by the caller;
 *   - return value 0 == NNR_OK, negative == error (nnr_strerror gives the text);
 *   - fp32 throughout; ray / sample indices are produced by the caller (torch.randperm / torch.rand
 *     stay in the host framework so indices and jitter are bit-identical to the reference's).
 */
#ifndef NNR_H
#define NNR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNR_ABI_VERSION 6

/* error codes */
#define NNR_OK 0
#define NNR_E_BADCFG (-1)      /* inconsistent sizes / null pointer */
#define NNR_E_UNSUPPORTED (-2) /* hidden width other than 128/256, encoding levels other than 10/4 ... */
#define NNR_E_ALIGN (-3)       /* a pointer is not 16-byte aligned */
#define NNR_E_HIP (-4)         /* a HIP runtime call failed (see nnr_last_hip_error) */

/* flags (nnr_cfg.flags) -- each mirrors a key of the reference's `rendering:` / `model:` YAML section */
#define NNR_F_DIST_ALPHA 1u /* rendering.dist_alpha   (model/rendering.py:121-128, official_nerf.py:82-83) */
#define NNR_F_WHITE_BG 2u   /* rendering.white_background (model/rendering.py:145-147) */
#define NNR_F_RELU_SIGMA 4u /* model.occ_activation != 'softplus' (model/official_nerf.py:77-80) */
#define NNR_F_TRAIN 8u      /* keep what the backward needs (activation stash, ReLU masks) */
#define NNR_F_BF16 16u      /* every nn.Linear as a bf16 x bf16 product with fp32 accumulation in the three MLP kernels (BASELINE
                             * configs[2]); biases, activation functions, compositing and every reduction stay fp32.  The
                             * packed-weight buffer has its own size and layout in this mode (nnr_packed_floats), and so has a
                             * TRAINING workspace (nnr_workspace_floats): the hidden-activation planes 11..18, 20, the encodings'
                             * copies 21, 22 and the gradient planes 31..38, 40 hold bf16 elements -- the values the MFMAs
                             * consumed -- in TILE-MAJOR order (1 KiB blocks [chunk of 32 samples][16 features], DESIGN.md
                             * section 3); nnr_ws_plane reports a pitch of half the feature count for them.  Planes 10 and 19 hold
                             * the chain-rule factors of the encodings in register order; all other planes are as in fp32. */
#define NNR_F_SPLIT3 32u    /* fp32 results from the bf16 matrix pipe: every operand of every nn.Linear product as the EXACT sum of three
                             * bf16 terms (8 + 8 + 8 significand bits), six of the nine term products as bf16 MFMAs with fp32
                             * accumulation, the three below 2^-24 of the product dropped (nnr_split.h).  As close to the exact result
                             * as the fp32 MFMA path (verified against fp64), 2.7 times fewer matrix-pipe cycles.  The planes are those
                             * of the fp32 mode; the packed-weight buffer has another size and layout (nnr_packed_floats) and the
                             * weight-gradient plan another cut (nnr_plan_bytes, nnr_workspace_floats: query with the same flags).
                             * Ignored with NNR_F_BF16. */
#define NNR_F_SPLIT2 64u    /* with NNR_F_SPLIT3: the three MLP kernels take every fp32 product as THREE fp16 MFMA terms of two-term operands
                             * (11 + 11 significand bits; operands scaled by powers of two, the residual term carried at 2^11: nnr_split2.h)
                             * instead of six bf16 terms -- half the matrix-pipe passes, about 2^-22 relative error per product, as close to an
                             * fp64 evaluation of the step as the other fp32 paths (tests/test_gpu_split3.py, all modes at the same bars).  One
                             * bound the other modes do not have: a hidden activation of 65520 or more rounds to inf in fp16; the kernels detect
                             * it and return NaN for that sample (rgb, sigma) rather than a finite wrong value.  The planes are NNR_F_SPLIT3's; the packed-weight buffer has its own contents and size, the weight-gradient
                             * plan its own cut, and the training workspace ends in 32 floats of plane maxima -- written by nnr_mlp_fwd
                             * (which zeroes them first) and nnr_mlp_dgrad, read by nnr_mlp_wgrad of the same step (its 4 x 4 tiles and the 128 x 64 tiles against
                             * the position encoding scale their fp16 terms per plane; the other narrow tiles stay on fp32 MFMAs).  Query all sizes with the same flags.
                             * Ignored without NNR_F_SPLIT3 or with NNR_F_BF16. */

/* Problem description.  POD, passed by pointer, read on the host only. */
typedef struct nnr_cfg {
    int32_t n_rays;    /* R: rays in this call                       (training.n_training_points) */
    int32_t n_samples; /* N: samples per ray                         (rendering.num_points)       */
    int32_t hidden;    /* D: MLP width, 128 or 256                   (model.hidden_dim)           */
    uint32_t flags;    /* NNR_F_*                                                                 */
} nnr_cfg;

/* The 12 nn.Linear layers of OfficialStaticNerf in state_dict order (model/official_nerf.py:20-37):
 * layers0.{0,2,4,6}, layers1.{0,2,4,6}, fc_density, fc_feature, rgb_layers.0, fc_rgb.
 * weight[i] is (out,in) row-major exactly as nn.Linear stores it; bias[i] is (out). */
#define NNR_N_LAYERS 12
typedef struct nnr_params {
    const float* weight[NNR_N_LAYERS];
    const float* bias[NNR_N_LAYERS];
} nnr_params;
typedef struct nnr_param_grads { /* OVERWRITTEN by the weight-gradient stage (ABI 3; accumulated into up to ABI 2): no zero-fill needed */
    float* weight[NNR_N_LAYERS];
    float* bias[NNR_N_LAYERS];
} nnr_param_grads;

int nnr_abi_version(void);
const char* nnr_strerror(int code);
int nnr_last_hip_error(void); /* hipError_t of the most recent NNR_E_HIP on this thread */

/* --- sizes (host-side, no GPU work) ------------------------------------------------------------- */
/* floats in the packed-weight buffer (MFMA-fragment order, forward + transposed copies + biases) */
size_t nnr_packed_floats(const nnr_cfg* cfg);
/* floats of scratch the forward(+backward, if NNR_F_TRAIN) needs; contents are opaque except via nnr_ws_plane */
size_t nnr_workspace_floats(const nnr_cfg* cfg);
/* The weight-gradient work plan: a static, balanced schedule of wave jobs (which tile of which dW, which sample range),
 * WgradJob[n_jobs] grouped by wave followed by int32 wave_first[n_waves + 1], int32 n_heads, int32 heads[n_heads] (the job that holds
 * split 0 of every tile: ABI 4).  Build on the host, upload once per cfg.
 * nnr_plan_counts reports n_jobs / n_waves (either pointer may be null). */
size_t nnr_plan_bytes(const nnr_cfg* cfg);
int nnr_plan_counts(const nnr_cfg* cfg, int32_t* n_jobs, int32_t* n_waves);
int nnr_plan_build(const nnr_cfg* cfg, void* plan_host);

/* --- weights ------------------------------------------------------------------------------------ */
/* Re-pack the nn.Linear tensors into fragment order.  Call after every optimiser step (one launch).
 * Replaces nothing in the reference: it is the layout change that lets the MLP read 1 KiB coalesced
 * fragments instead of (out,in) rows.  `packed` must be 16-byte aligned. */
int nnr_pack_weights(const nnr_cfg* cfg, const nnr_params* params_host, float* packed, void* stream);

/* --- forward ------------------------------------------------------------------------------------
 * Replaces the body of Renderer.nope_nerf from sampling to compositing (model/rendering.py:95-132,
 * 145-147) and OfficialStaticNerf.forward (model/official_nerf.py:60-96) it calls in 64 000-sample chunks.
 *   pts_o, pts_d : (R,3) sampling origin / direction of each ray: world-space (o, d_hat) for
 *                  sample_option 'uniform', the NDC-warped (o', d') for 'ndc'   (rendering.py:170-178,191-192)
 *   view_d       : (R,3) direction fed to the colour branch (= -d_hat, rendering.py:177-178,194-195)
 *   z_lo, z_hi   : (N) per-sample interval; z = z_lo + (z_hi - z_lo) * u  (rendering.py:184-190)
 *   jitter       : (R,N) u in [0,1) or NULL (z = z_lo)                    (torch.rand at rendering.py:189)
 * outputs
 *   rgb (R,3), dist (R) = sum w*z (rendering.py:131-132); opt_alpha (R,N) and opt_z (R,N) may be NULL.
 */
int nnr_render_fwd(const nnr_cfg* cfg, const float* pts_o, const float* pts_d, const float* view_d,
                   const float* z_lo, const float* z_hi, const float* jitter, const float* packed,
                   float* rgb, float* dist, float* opt_alpha, float* opt_z, float* workspace, void* stream);

/* --- backward -----------------------------------------------------------------------------------
 * Replaces autograd through the same region (loss.backward(), model/training.py:89).  Must follow an
 * nnr_render_fwd with NNR_F_TRAIN on the same workspace.  d_rgb (R,3), d_dist (R) are the upstream
 * gradients from the loss heads (model/losses.py:27-32,59-64).  Gradients w.r.t. the 24 parameter
 * tensors are written to `grads` (OVERWRITTEN: the buffers may be uninitialised); d_pts_o/d_pts_d/d_view (R,3) are overwritten.  `plan` is the
 * device copy of nnr_plan_build's table. */
int nnr_render_bwd(const nnr_cfg* cfg, const float* packed, const float* d_rgb, const float* d_dist,
                   const nnr_param_grads* grads_host, float* d_pts_o, float* d_pts_d, float* d_view,
                   const void* plan, float* workspace, void* stream);

/* --- introspection for parity tests ---------------------------------------------------------------
 * Offset (in floats) and row pitch of a workspace plane; returns <0 for an unknown plane.
 * Planes: 0 sample outputs (S,4: rgb, sigma_raw)   1 z (S)   2 d(sample outputs) (S,4)
 *         3 d(point) (S,4)  4 d(view) (S,4)   10 posenc (S,64)   11..18 hidden activations 1..8 (S,D)
 *         19 direction encoding (S,32)   20 colour hidden (S,D/2)   25 ReLU sign bits
 *         31..38 d(pre-activation) of hidden 1..8 (S,D)   40 d(colour hidden) (S,D/2)
 * (the feature vector of model/official_nerf.py:87 and its gradient are never formed: that layer is folded into the
 * colour-hidden layer, see nnr_layout.h) */
int64_t nnr_ws_plane(const nnr_cfg* cfg, int plane, int32_t* pitch_out);
/* How the plane's elements are ordered (ABI 4): 0 = row-major fp32 [sample][pitch]; 1 = tile-major bf16 (NNR_F_BF16 training, above);
 * 2 = tile-major fp32 -- NNR_F_SPLIT3 training, planes 10..20 (activations) and 31..38, 40 (gradients): 1 KiB blocks [chunk of 32 samples][octet j of features], a block
 * is [lane = 32 h + c][4 floats] = features 8 j + 4 h + {0..3} of sample 32 chunk + c (nnr_layout.h: tile32_index); offset and pitch
 * (= floats per sample) are those nnr_ws_plane reports.  <0 for an unknown plane. */
int nnr_ws_plane_layout(const nnr_cfg* cfg, int plane);

/* Individual stages, exported for profiling and bench.py's per-kernel roofline timing.  Same arguments
 * and workspace contract as the fused entry points above. */
int nnr_mlp_fwd(const nnr_cfg* cfg, const float* pts_o, const float* pts_d, const float* view_d,
                const float* z_lo, const float* z_hi, const float* jitter, const float* packed,
                float* workspace, void* stream);
int nnr_composite_fwd(const nnr_cfg* cfg, float* rgb, float* dist, float* opt_alpha, float* opt_z,
                      float* workspace, void* stream);
int nnr_composite_bwd(const nnr_cfg* cfg, const float* d_rgb, const float* d_dist, float* workspace, void* stream);
int nnr_mlp_dgrad(const nnr_cfg* cfg, const float* packed, float* workspace, void* stream);
int nnr_mlp_wgrad(const nnr_cfg* cfg, const float* packed, const nnr_param_grads* grads_host, const void* plan, float* workspace,
                  void* stream);
int nnr_ray_reduce(const nnr_cfg* cfg, float* d_pts_o, float* d_pts_d, float* d_view, float* workspace, void* stream);

/* --- camera front end / loss heads --------------------------------------------------------------------------------
 * One launch each for the O(1) / O(R) bookkeeping around the render call (the reference spends ~300 tiny ATen kernels
 * and four rocSOLVER LU inverses per step here).  All matrices are 4x4 row-major fp32 on the device. */

/* c2w = [[Exp(r_i), t_i],[0,0,0,1]] for camera `idx` of the (n_cams,3) tables -- LearnPose.forward
 * (model/poses.py:23-31) -> make_c2w / Exp (model/common.py:290-310).  The backward writes FULL (n_cams,3) gradient
 * tables (zero outside row idx), which is what autograd hands to the pose optimiser. */
int nnr_se3_exp_fwd(const float* r_all, const float* t_all, int32_t idx, float* c2w, void* stream);
int nnr_se3_exp_bwd(const float* r_all, int32_t idx, int32_t n_cams, const float* d_c2w, float* d_r_all, float* d_t_all,
                    void* stream);

/* batched 4x4 inverse and its backward dA = -Y^T dY Y^T -- torch.inverse at model/training.py:238 and the three
 * inverses of model/common.py:139-141,206-208 */
int nnr_inv4_fwd(const float* a, float* y, int32_t batch, void* stream);
int nnr_inv4_bwd(const float* y, const float* d_y, float* d_a, int32_t batch, void* stream);

/* Ray generation of Renderer.nope_nerf (model/rendering.py:54-87,194-195) for batch size 1: from pixels (R,2), per-ray
 * mono depth (R, may be NULL == 1), camera_mat K, world_mat W, scale_mat S:
 *   pts_o (R,3) camera centre, dir (R,3) ray direction (unit if `normalise`), view (R,3) = -dir (or ones if !use_dir),
 *   ray_norm (R) = |pixels_world - camera_world|, d_gt (R) = |points_world - camera_world| (divided by ray_norm if
 *   !normalise), mask (R, bytes) = finite(d_gt) && d_gt != 0.
 * Backward: upstream gradients (any may be NULL) -> d_depth (R, may be NULL), dK, dW, dS (16 each).
 * `scratch` = 12 floats. */
int nnr_ray_setup_fwd(const float* pixels, const float* depth, const float* K, const float* W, const float* S, int32_t n_rays,
                      int32_t normalise, int32_t use_dir, float* pts_o, float* dir, float* view, float* ray_norm, float* d_gt,
                      uint8_t* mask, void* stream);
int nnr_ray_setup_bwd(const float* pixels, const float* depth, const float* K, const float* W, const float* S, int32_t n_rays,
                      int32_t normalise, int32_t use_dir, const float* g_pts_o, const float* g_dir, const float* g_view,
                      const float* g_ray_norm, const float* g_d_gt, float* d_depth, float* dK, float* dW, float* dS,
                      float* scratch, void* stream);

/* pixels (R,2) = arange_pixels((h,w))[1][:, ray_idx] (model/common.py:13-40, model/training.py:260-261) without building
 * the (h*w,2) grid every step. */
int nnr_pixels_from_index(const int64_t* ray_idx, float* pixels, int32_t n_rays, int32_t h, int32_t w, void* stream);

/* Point-cloud loss.  nnr_pc_nearest: for every row of src (n_src,3) the index of the nearest row of dst
 * (n_dst,3) and the distance to it: Loss.comp_closest_pts_idx_with_split + the norm of comp_point_point_error
 * (model/losses.py:125-148) without the (3, S, D) difference tensor.  Same fp32 distance as torch.linalg.norm, first
 * index on ties.  scratch: n_src * 8 bytes, 8-byte aligned.
 * nnr_pc_error_bwd: gradient of mean_s dist[s] times the device scalar g_loss[0]: g_src (n_src,3) is overwritten,
 * g_dst (n_dst,3) is ACCUMULATED into (zero-fill first); either may be null.  Bit-reproducible: destination j gathers the terms
 * of the sources matched to it in source order (no float atomics). */
int nnr_pc_nearest(const float* src, const float* dst, int32_t n_src, int32_t n_dst, int64_t* idx, float* dist, void* scratch,
                   void* stream);
int nnr_pc_error_bwd(const float* src, const float* dst, const int64_t* idx, const float* dist, const float* g_loss, int32_t n_src,
                     int32_t n_dst, float* g_src, float* g_dst, void* stream);

/* Per-image losses between a frame ("1") and its neighbour ("2"), fused (SURVEY 8 f1 + f2): the inputs of reference
 * model/training.py:315-358 and the point-cloud / surface re-projection losses of model/losses.py:114-157 (with_ssim
 * through NNR_AUX_SSIM: the reference's SSIM module, losses.py:222-252, applied as the reference applies it, to the
 * (1, hr, wr, 3) colour tensors) and their backward.  d1_img/d2_img (hd,wd): the scaled+shifted depth maps; img1r/img2r (3,hr,wr): the images resized
 * (bilinear) to the sampling grid hr = hd/pc_ratio, wr = wd/pc_ratio; K, Kinv, rel: 4x4 row-major camera matrix, its
 * inverse and the relative transform Rt_rel_12; scale2: device scalar.  out[4] = {loss_pc, loss_rgb_s, n_valid, 0}.
 * The backward takes g_out[2] = dL/d{loss_pc, loss_rgb_s} (device) and ACCUMULATES into g_d1_img / g_d2_img (hd,wd;
 * zero-fill first; either may be null) and OVERWRITES g_rel_scale[16] = {dL/d rel rows 0..2 (12 floats), dL/d scale2, 0..};
 * with NNR_AUX_GRAD_K g_rel_scale has 40 floats: [16, 28) = dL/dK rows 0..2, [28, 40) = dL/dKinv rows 0..2 (K and Kinv as two
 * independent inputs, the way autograd sees camera_mat and its inverse in reference model/common.py:112-160, 436-457).
 * Every sum is taken in a fixed order (per-block partials added in block order; shared-destination scatters in 64-bit fixed
 * point), so losses and gradients are bit-reproducible from run to run.
 * Both calls use the same workspace (nnr_aux_workspace_floats) and the same inputs.
 * NNR_AUX_AFFINE (ABI 5): d1_img / d2_img are the RAW mono-depth maps and `aff` = device (scale1, shift1, scale2, shift2) -- the per-image
 * distortion of model/training.py:240-245, 294-296 ((depth + shift) * scale with NNR_AUX_SHIFT_FIRST) -- is applied to the sampled values
 * inside the kernels; the backward then returns dL/d aff at g_rel_scale[40, 44) (44 floats) and needs no g_d*_img (pass NULL).  Without the
 * flag `aff` must be NULL.
 * The nearest neighbours come from a search that uses what the clouds are -- depth maps lifted along the rays of one pixel grid (nnr_aux.hip:
 * the ray-window kernel for rough depths, the bounding-sphere tile kernel for smooth surfaces far apart, chosen per 8 x 8 tile of sources):
 * the same indices as the exhaustive nnr_pc_nearest, 55 us instead of 345 at 135 x 240 on same-pose clouds, 200 instead of 351 inside a
 * real training run's first epochs.  NNR_PC_SEARCH=brute in the environment selects the exhaustive search (rows / tiles: one of the two kernels on
 * everything).  The forward zeroes the backward's accumulators: ONE backward per forward. */
#define NNR_AUX_RGBS 1u         /* rgb_s_weight != 0 */
#define NNR_AUX_PC 2u           /* pc_weight != 0 */
#define NNR_AUX_SCALE_PCS 4u    /* training.scale_pcs */
#define NNR_AUX_DETACH_RGBS 8u  /* training.detach_rgbs_scale */
#define NNR_AUX_GRAD_K 32u      /* a learnable focal length: the backward also returns dL/dK and dL/dKinv (g_rel_scale has 40 floats) */
#define NNR_AUX_AFFINE 64u      /* the depth distortion is applied in the kernels (aff) */
#define NNR_AUX_SHIFT_FIRST 128u /* training.shift_first */
#define NNR_AUX_WEIGHTED 256u    /* out[3] = w_pc loss_pc + w_rgbs loss_rgb_s (the active terms; each product and the sum rounded to fp32, the torch
                                 * expression of model/losses.py:196-203); the backward's g_out is then ONE float, dL/d out[3] */
#define NNR_AUX_MATS_GRAD 512u   /* with NNR_AUX_AFFINE, without NNR_AUX_GRAD_K: rel, aff and scale2 are slices of nnr_step_rays_fwd's 56-float mats block
                                 * ([34, 50), [50, 54), [54]) and the backward writes g_rel_scale[56] in THAT layout (zeros elsewhere) */
#define NNR_AUX_SSIM 16u        /* training.with_ssim: 0.15 clamp|.| + 0.85 SSIM per re-projected colour (12 hr wr more workspace floats) */
typedef struct nnr_aux_cfg {
    int32_t hd, wd, hr, wr;
    float nearest_limit;
    uint32_t flags;
    /* Data parallelism: both losses are means over SOURCE points, so a rank evaluates the sums over the points
     * [shard_lo, shard_hi) of the hr*wr grid only (the nearest-neighbour search, O(S^2), shrinks by the world size; the O(S)
     * per-point passes still cover every point, because a rank's sources pull on arbitrary destinations) with the GLOBAL
     * normalisers (S, the number of valid re-projections): the SUM over ranks of the losses and of every gradient equals the
     * single-GPU value.  0, 0 = all points. */
    int32_t shard_lo, shard_hi;
    float w_pc, w_rgbs;      /* NNR_AUX_WEIGHTED: training.pc_weight / rgb_s_weight of the step */
} nnr_aux_cfg;
size_t nnr_aux_workspace_floats(const nnr_aux_cfg* cfg);
int nnr_aux_terms_fwd(const nnr_aux_cfg* cfg, const float* d1_img, const float* d2_img, const float* img1r, const float* img2r,
                      const float* K, const float* Kinv, const float* rel, const float* scale2, const float* aff, float* out,
                      float* workspace, void* stream);
int nnr_aux_terms_bwd(const nnr_aux_cfg* cfg, const float* d1_img, const float* d2_img, const float* img1r, const float* img2r,
                      const float* K, const float* Kinv, const float* rel, const float* scale2, const float* aff, const float* g_out,
                      float* g_d1_img, float* g_d2_img, float* g_rel_scale, float* workspace, void* stream);

/* out[0..r) = torch.randperm(n, device=cuda)[:r] (the pixel pick of model/training.py:257) from the n int64 keys torch's
 * randperm would have drawn (keys = empty(n, int64).random_(INT64_MIN, INT64_MAX)), the number of key bits it sorts by, and
 * the generator's (seed, philox offset) at the point where torch re-shuffles duplicate keys -- without sorting all n keys.
 * scratch: nnr_randperm_scratch_bytes(r) bytes = 8 + 20 * capacity (capacity 4096 up to r = 1401, 16384 up to r = 9943, 65536
 * up to r = 51463), 8-byte aligned, its first 8 + 4 * capacity bytes (count, status, ranks) ZERO on entry -- zero-fill the buffer once,
 * every call leaves them zeroed again (no memset launch per pick; calls that share a buffer must be ordered on one stream); if the candidate buffer under/overflowed (probability < 1e-50 by construction: the threshold
 * leaves >= 16 sigma below and >= 40 sigma above the expected count) scratch word [1] becomes 1 and the kernel TRAPS -- the
 * process aborts at its next synchronisation instead of training on a truncated pixel pick.  NNR_E_UNSUPPORTED where the
 * packing does not fit or r is beyond the largest buffer (bits + ceil(log2 n) > 64, scratch_bytes(r) == 0, n < 8r): callers fall
 * back to torch.randperm. */
size_t nnr_randperm_scratch_bytes(int32_t r);
int nnr_randperm_prefix(const int64_t* keys, int64_t n, int32_t bits, int32_t r, uint64_t seed, uint64_t offset, int64_t* out,
                        void* scratch, void* stream);

/* out[0..n) = torch.rand(total, device=cuda).flatten()[first : first + n] for the generator state (seed, philox offset) -- the jitter rows
 * of a data-parallel shard (model/rendering.py:157-159 draws the whole step's tensor) at O(n) instead of O(total).  threads = 256 * the
 * grid torch's uniform kernel would launch for `total` elements: min(multiProcessorCount * (maxThreadsPerMultiProcessor / 256),
 * ceil(total / 256)) blocks; the caller advances the generator by ((total - 1) / (4 threads) + 1) * 4, as the full draw would. */
int nnr_uniform_rows(uint64_t seed, uint64_t offset, uint64_t threads, uint64_t first, uint64_t n, float* out, void* stream);

/* The depth gather below with the per-image affine distortion of model/distortions.py:19-26 applied to the n_rays gathered values
 * instead of to the whole map (model/training.py:240-245 then model/network.py:22-24: the same numbers): out = raw * scale + shift,
 * or (raw + shift) * scale with shift_first.  scale, shift: one-element device tensors.  The backward writes g_scale_shift[0] =
 * d loss / d scale and [1] = d loss / d shift (the raw map is data and has no gradient). */
int nnr_depth_gather_affine_fwd(const float* depth_img, const int64_t* ray_idx, const float* scale, const float* shift, int32_t shift_first,
                                float* out, int32_t n_rays, int32_t h, int32_t w, int32_t hd, int32_t wd, void* stream);
int nnr_depth_gather_affine_bwd(const float* g_out, const float* depth_img, const int64_t* ray_idx, const float* scale, const float* shift,
                                int32_t shift_first, float* g_scale_shift, int32_t n_rays, int32_t h, int32_t w, int32_t hd, int32_t wd,
                                void* stream);

/* World rays -> NDC rays of a forward-facing scene: get_ndc_rays_fxfy (model/common.py:632-675) as Renderer.sample_ndc calls it
 * (model/rendering.py:168-180; near_plane = 1).  rays_o, rays_d, o_ndc, d_ndc: (n_rays, 3); camera_mat: the 4x4 K = diag(2f/w,
 * -2f/h, -1, 1) on the device (entries [0] and [5] are read).  The backward returns the gradients with respect to the world rays;
 * the intrinsics are constants here (with a learnable focal the caller keeps the torch expression). */
int nnr_ndc_rays_fwd(const float* rays_o, const float* rays_d, const float* camera_mat, float near_plane, float* o_ndc, float* d_ndc,
                     int32_t n_rays, void* stream);
int nnr_ndc_rays_bwd(const float* rays_o, const float* rays_d, const float* camera_mat, float near_plane, const float* g_o_ndc,
                     const float* g_d_ndc, float* g_rays_o, float* g_rays_d, int32_t n_rays, void* stream);

/* ONE launch for the Adam updates of a training step (reference model/training.py:90-96 steps up to four torch.optim.Adam per
 * iteration).  Plain Adam, weight_decay 0, no amsgrad, no maximize; the arithmetic is torch's fused implementation type by type
 * (double hyper-parameters, moments evaluated in double and rounded once, float bias corrections): bitwise the same parameters and
 * moments as torch.optim.Adam(fused=True).  The table is passed BY VALUE (host struct, <= NNR_ADAM_MAX_TENSORS tensors, all fp32,
 * contiguous).  step_in[i] / step_out[i]: one-element float counters; the kernel reads step_in, uses step_in + 1 and writes it to
 * step_out (distinct buffers: no block may see a counter another block has advanced).  block_first: prefix table in units of 1024
 * elements, block_first[i+1] - block_first[i] = ceil(numel[i] / 1024).
 * flavour (ABI 5) selects WHICH of torch's two Adam arithmetics is reproduced: NNR_ADAM_FUSED = the above; NNR_ADAM_SINGLE = torch's
 * single-tensor implementation (torch/optim/adam.py::_single_tensor_adam -- what the plain `optim.Adam(...)` objects of the reference's
 * train.py:58,99,117,140 run): float moments by one fma each (lerp_; mul_ + addcmul_), the bias corrections and the step size computed by
 * the CALLER as host doubles exactly as that function does -- lr[i] then carries -(lr / (1 - beta1^step)) (the table is a kernel argument:
 * 4 KiB at most), bc2_sqrt[i] = (1 - beta2^step)^0.5, step = the incremented counter --, denominator = sqrt(v) * float(1 / bc2_sqrt) + float(eps) (ATen divides by a host scalar through its
 * reciprocal), update = one fma.  Bitwise torch.optim.Adam(foreach=False, fused=False) (tests/test_gpu_optim.py). */
#define NNR_ADAM_FUSED 0
#define NNR_ADAM_SINGLE 1
#define NNR_ADAM_MAX_TENSORS 40
typedef struct nnr_adam_table {
    float* param[NNR_ADAM_MAX_TENSORS];
    const float* grad[NNR_ADAM_MAX_TENSORS];
    float* exp_avg[NNR_ADAM_MAX_TENSORS];
    float* exp_avg_sq[NNR_ADAM_MAX_TENSORS];
    const float* step_in[NNR_ADAM_MAX_TENSORS];
    float* step_out[NNR_ADAM_MAX_TENSORS];
    double lr[NNR_ADAM_MAX_TENSORS], beta1[NNR_ADAM_MAX_TENSORS], beta2[NNR_ADAM_MAX_TENSORS], eps[NNR_ADAM_MAX_TENSORS];
    int64_t numel[NNR_ADAM_MAX_TENSORS];
    int32_t block_first[NNR_ADAM_MAX_TENSORS + 1];
    int32_t n_tensors;
    int32_t flavour, reserved;
    double bc2_sqrt[NNR_ADAM_MAX_TENSORS];      /* NNR_ADAM_SINGLE only */
} nnr_adam_table;
int nnr_adam_step(const nnr_adam_table* table, void* stream);

/* Fused front end of a training step: everything between the learnable tables and the per-ray inputs of the render operator, ONE
 * launch each way.  Replaces, with the same arithmetic in the same order, nnr_se3_exp_fwd + nnr_inv4_fwd (world_mat = c2w^-1,
 * reference model/training.py:238) + Learn_Distortion.forward (model/distortions.py:19-26: scale floored at the constant 0.01, the
 * last camera's scale pinned to 1 with fix_scaleN) + nnr_pixels_from_index + nnr_depth_gather_affine_fwd + the colour-target gather
 * (training.py:258-259) + nnr_ray_setup_fwd, and on the way back nnr_ray_setup_bwd + nnr_depth_gather_affine_bwd + nnr_inv4_bwd +
 * nnr_se3_exp_bwd + the autograd of the distortion lookup.  Reductions run in the fixed order of those kernels (bit-reproducible).
 * r_all, t_all (n_cams,3); scales, shifts (n_cams); K, S 4x4 row-major; ray_idx (n_rays) int64; depth_img (hd,wd) RAW mono depth;
 * img (3,h,w) or null.  Outputs as nnr_ray_setup_fwd, plus rgb_gt (n_rays,3) [if img], pixels (n_rays,2) and mats[34] = c2w (16),
 * world_mat (16), effective scale, shift.  Backward: d_r, d_t (n_cams,3), d_scales, d_shifts (n_cams) are OVERWRITTEN (zeros outside
 * row cam).
 * cfg->ref >= 0 (ABI 5): the step also carries the frame PAIR of the per-image losses (model/training.py:280-313): mats then holds 56
 * floats -- [34, 50) rel = inverse(c2w_ref) inverse(world_mat) (the last camera: world_mat inverse(inverse(c2w_ref)), the roles swap),
 * [50, 54) the two clouds' depth distortions (scale1, shift1, scale2, shift2) in the order nnr_aux_terms_* takes them, [54] the second
 * cloud's scale -- what ~30 launches of se3_exp / inverse / matmul / indexing and their autograd made; the backward takes the upstream
 * gradient of mats (g_mats, 56 floats, or NULL) and chains it into the same tables (NNR_STEP_DETACH_REF = training.detach_ref_img: nothing
 * flows into the reference camera's rows). */
#define NNR_STEP_NORMALISE 1u       /* rendering.normalise_ray */
#define NNR_STEP_USE_DIR 2u         /* rendering.use_ray_dir */
#define NNR_STEP_SHIFT_FIRST 4u     /* training.shift_first: (depth + shift) * scale */
#define NNR_STEP_FIX_LAST_SCALE 8u  /* distortion.fix_scaleN */
#define NNR_STEP_DETACH_REF 16u     /* training.detach_ref_img (with ref >= 0) */
typedef struct nnr_step_cfg {
    int32_t n_rays, h, w, hd, wd; /* image size, mono-depth map size */
    int32_t cam, n_cams;
    uint32_t flags;
    int32_t ref;                  /* the reference camera of the per-image losses, -1 = none */
} nnr_step_cfg;
int nnr_step_rays_fwd(const nnr_step_cfg* cfg, const float* r_all, const float* t_all, const float* scales, const float* shifts,
                      const float* K, const float* S, const int64_t* ray_idx, const float* depth_img, const float* img, float* pts_o,
                      float* dir, float* view, float* ray_norm, float* d_gt, uint8_t* mask, float* rgb_gt, float* pixels, float* mats,
                      void* stream);
int nnr_step_rays_bwd(const nnr_step_cfg* cfg, const float* r_all, const float* t_all, const float* scales, const float* shifts,
                      const float* K, const float* S, const int64_t* ray_idx, const float* depth_img, const float* g_pts_o,
                      const float* g_dir, const float* g_view, const float* g_ray_norm, const float* g_d_gt, const float* g_mats,
                      float* d_r, float* d_t, float* d_scales, float* d_shifts, float* scratch, void* stream);
/* scratch: NNR_STEP_BWD_SCRATCH_FLOATS floats whose FIRST word is zero on entry -- zero-fill the buffer once, every call leaves the word zero
 * again (the ticket of the workgroups' fixed-order reduction); calls that share a buffer must be ordered on one stream. */
#define NNR_STEP_BWD_SCRATCH_FLOATS 528

/* depth = nearest-resize(depth_img (hd,wd) -> (h,w)).flatten()[ray_idx]  (model/network.py:22-24) without materialising
 * the resized image; backward scatter-adds into a zero-filled (hd,wd) gradient image. */
int nnr_depth_gather_fwd(const float* depth_img, const int64_t* ray_idx, float* out, int32_t n_rays, int32_t h, int32_t w,
                         int32_t hd, int32_t wd, void* stream);
int nnr_depth_gather_bwd(const float* g_out, const int64_t* ray_idx, float* g_img, int32_t n_rays, int32_t h, int32_t w,
                         int32_t hd, int32_t wd, void* stream);

/* Loss heads feeding the backward (model/losses.py:27-32,59-64,196-202): out[0] = w_rgb * L_rgb + w_depth * L_depth,
 * out[1] = L_rgb = sum|rgb - gt| (or squared if rgb_l2) / r_total, out[2] = L_depth = sum_valid |dist - d_gt| / m_total,
 * out[3] = mean squared rgb error, out[4] = number of valid depths in this call.  m_total < 0 means "this call's count";
 * data-parallel callers pass the global counts (m_total_dev, if not NULL, is a device scalar that overrides m_total so
 * that no host sync is needed to obtain it).  ndc applies depth_gt = 1 - 1/d_gt (rendering.py:157-158).  The
 * gradients of out[0] w.r.t. rgb, dist, d_gt are written to g_rgb (R,3), g_dist (R), g_d_gt (R). */
int nnr_render_loss(const float* rgb, const float* rgb_gt, const float* dist, const float* d_gt, const uint8_t* mask,
                    int32_t n_rays, float r_total, float m_total, float w_rgb, float w_depth, int32_t rgb_l2, int32_t ndc,
                    int32_t detach_gt, const float* m_total_dev, float* out5, float* g_rgb, float* g_dist, float* g_d_gt,
                    void* stream);

/* In-step kernel timing for benchmarks (no reference counterpart: the reference has no profiling hooks).  Between nnr_prof_begin and
 * nnr_prof_end every launch of a main MLP kernel -- training forward, input gradient, weight gradient, inference forward, in this
 * order in the result arrays -- is bracketed by two HIP events on its launch stream; nnr_prof_end waits for them and returns the
 * mean duration in milliseconds and the number of launches recorded per kind (at most max_launches each; later ones are not timed). */
int nnr_prof_begin(int32_t max_launches);
int nnr_prof_end(float* mean_ms4, int32_t* launches4);

#ifdef __cplusplus
}
#endif
#endif /* NNR_H */
