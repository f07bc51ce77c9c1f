// nnr_kernels.h -- kernel argument blocks and host-side launchers (internal to libnnr.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

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
    float* ws_xf;     // (S_pad,D+32)
    float* ws_xg;     // (S_pad,D/2)
    uint32_t* ws_mask;
    int64_t S, S_pad;
    int N;
};

struct MlpDgradArgs {
    const float* packed;
    float* ws_dout4;        // (S_pad,4): d rgb_pre[3], d sigma_raw (rows >= S are zero-filled here)
    const float* ws_xe;     // posenc stash (for d gamma/dp)
    const float* ws_xf;     // [feature|direnc] stash (direnc part used)
    const uint32_t* ws_mask;
    float* ws_dh;    // 8 x (S_pad,D)
    float* ws_df;    // (S_pad,D)
    float* ws_dg;    // (S_pad,D/2)
    float* ws_dpts;  // (S_pad,4)
    float* ws_dview; // (S_pad,4)
    int64_t S, S_pad;
};

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
    const float* w[12];
    const float* b[12];
    float* packed;
};

struct WgradArgs {
    float* gw[12];
    float* gb[12];
    const WgradJob* jobs;
    const float* ws;           // workspace base
    int64_t plane_off[48];     // offset (floats) of each plane id, -1 if absent
    int32_t plane_pitch[48];
    int n_jobs;
};

hipError_t launch_pack(int D, const PackArgs& a, hipStream_t st);
hipError_t launch_mlp_fwd(int D, const MlpFwdArgs& a, bool train, hipStream_t st);
hipError_t launch_mlp_dgrad(int D, const MlpDgradArgs& a, hipStream_t st);
hipError_t launch_composite_fwd(const CompositeArgs& a, hipStream_t st);
hipError_t launch_composite_bwd(const CompositeArgs& a, hipStream_t st);
hipError_t launch_ray_reduce(const RayReduceArgs& a, hipStream_t st);
hipError_t launch_wgrad(const WgradArgs& a, hipStream_t st);

}  // namespace nnr
