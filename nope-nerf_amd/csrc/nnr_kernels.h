// nnr_kernels.h -- kernel argument blocks and host-side launchers (internal to libnnr.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nnr.h"
#include "nnr_layout.h"

namespace nnr {

struct MlpFwdArgs {
    const float *pts_o, *pts_d, *view_d;  // (R,3)
    const float *z_lo, *z_hi;             // (N)
    const float* jitter;                  // (R,N) or null
    const float* packed;
    float* ws_out4;   // (S_pad,4)
    float* ws_z;      // (S_pad)
    float* ws_xe;     // (S_pad,64)
    float* ws_xh;     // 8 x (S_pad,D)
    float* ws_xf;     // (S_pad,32): direction encoding
    float* ws_xg;     // (S_pad,D/2)
    float *ws_xe16, *ws_xf16;   // bf16 training: tile-major bf16 copies of the encodings (nnr_layout.h)
    float *ws_pts, *ws_view;    // bf16 training: (S_pad,4) position / view direction of every sample for the input-gradient kernel (= P_DPTS / P_DVIEW)
    uint32_t* ws_mask;
    // NNR_F_SPLIT2 training: [17] non-negative floats, atomically maximised as integers -- the largest |value| this launch stashed in the planes
    // P_XH1..8 ([0..8)); the input-gradient kernel adds P_DH1..8 ([8..16)) and P_DG ([16]).  The weight-gradient kernel scales its fp16 terms by them.
    float* plane_max;
    int64_t S, S_pad;
    int N;
    // Inference only, ray mode only: composite in the kernel's epilogue (one HBM write of 16 bytes per ray instead of 20 bytes per
    // sample + a second kernel).  fuse_rgb != null selects it; ws_out4 / ws_z are then not written.
    float *fuse_rgb, *fuse_dist;   // (R,3), (R)
    uint32_t flags;                // nnr_cfg.flags (dist_alpha / white background / ReLU density)
    int chunks_per_ray;   // passes per ray in ray mode (R % 4 == 0 and N a multiple of the wave's samples -- 32, bf16 kernels 64: a
                          // wave walks one ray), 0 = flat decomposition
};

struct MlpDgradArgs {
    const float* packed;
    float* ws_dout4;        // (S_pad,4): d rgb_pre[3], d sigma_raw (rows >= S are zero-filled here)
    const float* ws_xe;     // posenc stash (for d gamma/dp)
    const float* ws_xf;     // direction-encoding stash
    const uint32_t* ws_mask;
    float* ws_dh;    // 8 x (S_pad,D)
    float* ws_dg;    // (S_pad,D/2)
    float* ws_dpts;  // (S_pad,4)
    float* ws_dview; // (S_pad,4)
    float* plane_max;   // NNR_F_SPLIT2: see MlpFwdArgs
    int64_t S, S_pad;
    int chunks_per_ray;   // as in MlpFwdArgs
};
constexpr int kPlaneMaxFloats = 32;      // the table's size in the workspace (17 used)

struct CompositeArgs {
    const float* ws_out4;  // (S_pad,4)
    const float* ws_z;
    float* ws_dout4;
    float *rgb, *dist, *opt_alpha, *opt_z;  // forward outputs
    const float *d_rgb, *d_dist;            // backward inputs
    int R, N;
    uint32_t flags;
};

struct RayReduceArgs {
    const float *ws_dpts, *ws_dview, *ws_z;
    float *d_pts_o, *d_pts_d, *d_view;  // (R,3)
    int R, N;
};

struct PackArgs {
    const float* w[13];   // 12 nn.Linear weights + [12] = the merged matrix W' inside `packed` (filled by the launcher)
    const float* b[13];
    float* packed;
};

struct WgradArgs {
    float* gw[13];             // [12] = dW' scratch (D/2 x D) in the workspace
    float* gb[13];             // [12] = db' scratch
    const float* packed;       // for the un-merge step: copies of Wf, Wg[:, :D], bf (nnr_layout.h merge area)
    int D, bf16;               // bf16: `packed` is the bf16-mode buffer (same merge area, different offsets)
    const WgradJob* jobs;
    const float* ws;           // workspace base
    float* slots;              // n_jobs partial slots of kSlotFloats (see nnr_layout.h)
    int64_t plane_off[48];     // offset (floats) of each plane id, -1 if absent
    int32_t plane_pitch[48];
    int32_t plane_tile[48];    // 1: the plane is tile-major fp32 (WsLayout::tiled, nnr_layout.h)
    const int32_t* wave_first; // wave w runs jobs [wave_first[w], wave_first[w+1])
    const int32_t* heads;      // job index of split 0 of every tile (the reduction kernel's workgroups)
    int n_jobs, n_waves, n_heads;
    const float* plane_max;    // NNR_F_SPLIT2 (bf16 == 3): the planes' largest magnitudes (MlpFwdArgs::plane_max), for the fp16-term tiles' scales
    int bias_rows[13];         // elements of gb[l]: the main kernel zeroes them (the reduction adds up to two shares per row)
};

struct WgradBArgs {            // bf16 training mode (nnr_wgrad_bf16.hip)
    float* gw[13];             // [12] = dW' scratch (D/2 x D) in the workspace
    float* gb[13];
    const float* packed;
    int D;
    const WgradJobB* jobs;     // n_jobs, grouped by workgroup
    const int32_t* block_first;   // workgroup b runs jobs [block_first[b], block_first[b+1])
    const WgradOutB* outs;     // n_outs destination rectangles
    const float* ws;           // workspace base (the jobs carry byte offsets into it)
    float* slots;              // 4 * n_jobs wave slots of kSlotBFloats
    int n_jobs, n_blocks, n_outs;
};

struct RaySetupArgs {
    const float* pixels;  // (R,2) in [-1,1]
    const float* depth;   // (R) or null (== 1)
    const float *K, *W, *S;  // 4x4 each: camera_mat, world_mat, scale_mat
    float *pts_o, *dir, *view;  // (R,3)
    float *ray_norm, *d_gt;      // (R)
    uint8_t* mask;               // (R)
    // backward
    const float *g_o, *g_dir, *g_view, *g_norm, *g_dgt;  // upstream gradients (any may be null)
    float* g_depth;              // (R) or null
    float* acc;                  // 12 floats: dL/dM[:3,:4], zeroed by the launcher
    float *gK, *gW, *gS;         // 16 each
    int R;
    int normalise, use_dir;
};

// fused front end of a training step (step_rays_fwd / step_rays_bwd, nnr_camera.hip)
struct StepRaysArgs {
    const float *r_all, *t_all;      // (n_cams,3) pose tables
    const float *scales, *shifts;    // (n_cams) depth-distortion tables
    const float *K, *S;              // 4x4 camera_mat, scale_mat
    const int64_t* ray_idx;          // (R) flat pixel indices
    const float* depth_img;          // (hd,wd) raw mono depth
    const float* img;                // (3,h,w) or null
    int R, h, w, hd, wd, cam, n_cams;
    int normalise, use_dir, shift_first, fix_last_scale;
    // forward outputs
    float *pts_o, *dir, *view, *ray_norm, *d_gt, *rgb_gt, *pixels, *mats;   // mats: c2w[16], world_mat[16], scale, shift
    uint8_t* mask;
    // backward
    const float *g_o, *g_dir, *g_view, *g_norm, *g_dgt;   // upstream gradients (any may be null)
    float *d_r, *d_t, *d_scales, *d_shifts;               // full tables, overwritten
    float* bwd_scratch;                                   // NNR_STEP_BWD_SCRATCH_FLOATS: [0] a ticket counter (zero on entry, left zero), [16 + 16 b ..): workgroup b's 14 sums
    // the frame pair of the per-image losses (ref >= 0): mats[34, 56) = rel (16), the pair's distortions (s1, t1, s2, t2), scale2, 0;
    // backward: g_mats = the upstream gradient of mats (may be null), detach_ref = training.detach_ref_img
    int ref, detach_ref;
    const float* g_mats;
};

struct LossArgs {
    const float *rgb, *rgb_gt, *dist, *d_gt;
    const uint8_t* mask;
    float* out;     // [loss, loss_rgb, loss_depth, l2_mean, n_valid]
    float *g_rgb, *g_dist, *g_dgt;
    int R;
    float r_total, m_total;   // normalisers (m_total < 0: use this call's valid count)
    const float* m_total_dev; // optional device scalar overriding m_total
    float w_rgb, w_depth;
    int rgb_l2, ndc, detach_gt;
};

// ---- in-step kernel timing (nnr_prof_begin / nnr_prof_end, nnr_api.cpp): HIP events on the launch stream around the main MLP kernels
enum ProfKind { PROF_FWD_TRAIN = 0, PROF_DGRAD = 1, PROF_WGRAD = 2, PROF_FWD_INFER = 3, PROF_KINDS = 4 };
void prof_before(int kind, hipStream_t st);   // no-ops unless profiling is on
void prof_after(int kind, hipStream_t st);

hipError_t launch_se3_exp_fwd(const float* r_all, const float* t_all, int idx, float* c2w, hipStream_t st);
hipError_t launch_se3_exp_bwd(const float* r_all, int idx, int n_cams, const float* d_c2w, float* d_r, float* d_t, hipStream_t st);
hipError_t launch_inv4(const float* a, float* y, int batch, hipStream_t st);
hipError_t launch_inv4_bwd(const float* y, const float* dy, float* da, int batch, hipStream_t st);
hipError_t launch_ray_setup_fwd(const RaySetupArgs& a, hipStream_t st);
hipError_t launch_ray_setup_bwd(const RaySetupArgs& a, hipStream_t st);
hipError_t launch_depth_gather_fwd(const float* img, const int64_t* idx, float* out, int R, int h, int w, int hd, int wd, hipStream_t st);
hipError_t launch_depth_gather_affine_fwd(const float* img, const int64_t* idx, const float* scale, const float* shift, int shift_first,
                                          float* out, int R, int h, int w, int hd, int wd, hipStream_t st);
hipError_t launch_depth_gather_affine_bwd(const float* g, const float* img, const int64_t* idx, const float* scale, const float* shift,
                                          int shift_first, float* g_ss, int R, int h, int w, int hd, int wd, hipStream_t st);
hipError_t launch_ndc_rays_fwd(const float* o, const float* d, const float* K, float near_, float* o_ndc, float* d_ndc, int R, hipStream_t st);
hipError_t launch_ndc_rays_bwd(const float* o, const float* d, const float* K, float near_, const float* g_o_ndc, const float* g_d_ndc,
                               float* g_o, float* g_d, int R, hipStream_t st);
hipError_t launch_adam_multi(const nnr_adam_table& t, hipStream_t st);
hipError_t launch_step_rays_fwd(const StepRaysArgs& a, hipStream_t st);
hipError_t launch_step_rays_bwd(const StepRaysArgs& a, hipStream_t st);
hipError_t launch_depth_gather_bwd(const float* g, const int64_t* idx, float* g_img, int R, int h, int w, int hd, int wd, hipStream_t st);
hipError_t launch_render_loss(const LossArgs& a, hipStream_t st);
hipError_t launch_pixels_from_index(const int64_t* idx, float* out, int R, int h, int w, hipStream_t st);
// per-image losses (nnr_aux.hip); flags = NNR_AUX_* of nnr.h
struct AuxArgs {
    const float *d1_img, *d2_img;   // (hd, wd) scaled / shifted depth maps of frame 1 and 2
    const float *img1r, *img2r;     // (3, hr, wr) images resized to the sampling grid
    const float *K, *Kinv, *rel;    // 4x4 row-major
    const float* scale2;            // device scalar (read only with NNR_AUX_SCALE_PCS)
    const float* aff;               // NNR_AUX_AFFINE: device (scale1, shift1, scale2, shift2) applied to the RAW maps d1_img / d2_img here; else null
    int shift_first;                // NNR_AUX_SHIFT_FIRST: (depth + shift) * scale
    float w_pc, w_rgbs;             // NNR_AUX_WEIGHTED: out[3] = w_pc loss_pc + w_rgbs loss_rgb_s; the backward's g_out is then the ONE gradient of out[3]
    int hd, wd, hr, wr, S;
    int s_lo, s_hi;                 // this rank's shard of the source points (data parallelism): [0, S) on one GPU
    float nl;
    uint32_t flags;
    // workspace
    float *X, *Y, *gxy;             // (S,3) (S,3) (S,2)
    float *rgb1, *rgb2, *drgb;      // NNR_AUX_SSIM only: (S,3) colours of both frames at every point, (S,2,3) d rgb2 / d (x, y)
    long long *gXq, *gYq;           // (S,3) each, adjacent: cloud gradients in 2^-44 fixed point (order-independent atomics)
    float *part_fwd, *part_bwd;     // per-block partial sums: [ceil(S/256)][4] and [ceil(S/256)][40]
    uint32_t* pflags;               // (S)
    unsigned long long* keys;       // (2S)
    int64_t *idx_xy, *idx_yx;
    float *dist_xy, *dist_yx;
    float* acc;                     // 8: rgb_s sum, valid count, sum dist xy, sum dist yx (written by the finishing kernel)
    float* out;                     // 4: loss_pc, loss_rgb_s, n_valid, 0
    // backward
    const float* g_out;             // 2: dL/d loss_pc, dL/d loss_rgb_s
    float *g_d1_img, *g_d2_img;     // (hd, wd), accumulated into; may be null
};
hipError_t launch_aux_fwd(const AuxArgs& a, hipStream_t st);
hipError_t launch_aux_bwd(const AuxArgs& a, float* g_rel_scale, hipStream_t st);   // (NNR_AUX_MATS_GRAD: 56 floats in the mats layout) g_rel_scale[16 | 40]: dL/d rel rows 0..2 (12), dL/d scale2 (1); NNR_AUX_GRAD_K: + dL/dK, dL/dKinv rows 0..2 at [16, 40)
hipError_t launch_pc_nearest_keys(const float* src, const float* dst, int S, int D, unsigned long long* keys, hipStream_t st);
hipError_t launch_pc_nearest(const float* src, const float* dst, int S, int D, int64_t* idx, float* dist, unsigned long long* keys,
                             hipStream_t st);
hipError_t launch_pc_error_bwd(const float* src, const float* dst, const int64_t* idx, const float* dist, const float* g_loss, int S,
                               int D, float* g_src, float* g_dst, hipStream_t st);
unsigned int randperm_capacity(int r);   // candidates the scratch buffer must hold for a pick of r; 0 = r not supported
hipError_t launch_randperm_prefix(const int64_t* keys, int64_t n, int bits, int r, unsigned long long seed, unsigned long long offset,
                                  int64_t* out, unsigned int* scratch, hipStream_t st);
hipError_t launch_uniform_rows(unsigned long long seed, unsigned long long offset, unsigned long long threads, unsigned long long first,
                               unsigned long long n, float* out, hipStream_t st);   // nnr_randperm.hip
hipError_t launch_pack(int D, const PackArgs& a, int mode, hipStream_t st);   // mode: Layout<D, MODE> (0 fp32, 1 bf16, 2 three bf16 terms, 3 two fp16 terms)
// fp32 products by Layout mode: 0 fp32 MFMAs, 2 six bf16 MFMA terms per product (nnr_split.h), 3 three fp16 MFMA terms (nnr_split2.h)
hipError_t launch_mlp_fwd(int D, const MlpFwdArgs& a, bool train, hipStream_t st, int mode = 0);
template <int D, bool TRAIN, int MODE = 0> hipError_t launch_mlp_fwd_variant(const MlpFwdArgs& a, hipStream_t st);   // one translation unit each
template <int D, int MODE = 0> hipError_t launch_mlp_dgrad_variant(const MlpDgradArgs& a, hipStream_t st);
hipError_t launch_mlp_dgrad(int D, const MlpDgradArgs& a, hipStream_t st, int mode = 0);
hipError_t launch_mlp_fwd_bf16(int D, const MlpFwdArgs& a, bool train, hipStream_t st);     // nnr_mlp_fwd_bf16.hip
hipError_t launch_mlp_dgrad_bf16(int D, const MlpDgradArgs& a, hipStream_t st);              // nnr_mlp_dgrad_bf16.hip
hipError_t launch_composite_fwd(const CompositeArgs& a, hipStream_t st);
hipError_t launch_composite_bwd(const CompositeArgs& a, hipStream_t st);
hipError_t launch_ray_reduce(const RayReduceArgs& a, hipStream_t st);
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t st);
hipError_t launch_wgrad_unmerge(const WgradArgs& a, hipStream_t st);   // uses gw, gb, packed, D, bf16 only
hipError_t launch_wgrad_bf16(const WgradBArgs& a, hipStream_t st);

}  // namespace nnr
