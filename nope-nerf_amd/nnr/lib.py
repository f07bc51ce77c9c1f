"""ctypes binding of libnnr.so (include/nnr.h).

This is the stub a maintainer of the reference would add: plain pointers and sizes,
no torch types cross the boundary.  There is deliberately NO fallback: if the shared library is missing or
a call fails, a RuntimeError is raised -- the product path never silently runs anything but the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NNR_LIB", os.path.join(_HERE, "libnnr.so"))   # NNR_LIB: profiling builds only (csrc/build.py --variant)

NNR_F_DIST_ALPHA = 1
NNR_F_WHITE_BG = 2
NNR_F_RELU_SIGMA = 4
NNR_F_TRAIN = 8
NNR_F_BF16 = 16
NNR_F_SPLIT3 = 32
NNR_F_SPLIT2 = 64

# How the fp32 mode multiplies in the forward and input-gradient kernels (include/nnr.h, NNR_F_SPLIT3 / NNR_F_SPLIT2):
#   "split2" (round 6, the default) = every operand as TWO fp16 terms (power-of-two scaled, the residual carried at 2^11), three fp16 MFMAs per
#            product with fp32 accumulation (csrc/nnr_split2.h) -- half the matrix-pipe passes of "split3", as close to an fp64 evaluation of the
#            step (tests/test_gpu_split2.py); the weight gradient keeps split3's six bf16 terms;
#   "split3" = every operand as three bf16 terms, six bf16 MFMAs per product (tests/test_gpu_split3.py): no range bound at all;
#   "mfma"   = v_mfma_f32_32x32x2_f32.
PRODUCT_KINDS = ("split2", "split3", "mfma")
_fp32_products = os.environ.get("NNR_FP32_PRODUCTS", "split2")
if _fp32_products not in PRODUCT_KINDS:      # a typo must not silently select the other arithmetic (packed layout, plan and kernels differ)
    raise ValueError("NNR_FP32_PRODUCTS=%r: expected one of %r" % (_fp32_products, PRODUCT_KINDS))


def set_fp32_products(kind: str) -> str:
    """Select "split2", "split3" or "mfma" for every fp32-mode call made from now on; returns the previous setting."""
    global _fp32_products
    if kind not in PRODUCT_KINDS:
        raise ValueError(kind)
    prev, _fp32_products = _fp32_products, kind
    return prev


def fp32_products() -> str:
    return _fp32_products
N_LAYERS = 12

#: state_dict order of the 12 nn.Linear layers (reference model/official_nerf.py:20-37)
LAYER_NAMES = ("layers0.0", "layers0.2", "layers0.4", "layers0.6", "layers1.0", "layers1.2", "layers1.4",
               "layers1.6", "fc_density", "fc_feature", "rgb_layers.0", "fc_rgb")

# = NNR_ABI_VERSION of include/nnr.h; bumped whenever a signature, a struct or a blob layout that crosses the C ABI changes
# (2: nnr_pc_error_bwd takes n_dst; round-2 layouts of nnr_aux_cfg and the bf16 plan blob.  3: nnr_step_rays_*; the weight-gradient
# stage overwrites nnr_param_grads instead of accumulating into it.  4: nnr_ws_plane_layout; the gradient planes of a three-term training workspace
# are tile-major fp32.  5: nnr_adam_table.flavour / bc2_sqrt -- torch's single-tensor Adam arithmetic beside the fused one; nnr_step_cfg.ref + g_mats: the frame pair of the
# per-image losses in the fused front end; nnr_aux_terms_*: `aff`, the depth distortion applied in the kernels)
ABI_VERSION = 6
EXPORTS = ("nnr_abi_version", "nnr_strerror", "nnr_last_hip_error", "nnr_packed_floats", "nnr_workspace_floats",
           "nnr_plan_bytes", "nnr_plan_counts", "nnr_plan_build", "nnr_pack_weights", "nnr_render_fwd", "nnr_render_bwd", "nnr_ws_plane",
           "nnr_ws_plane_layout",
           "nnr_mlp_fwd", "nnr_composite_fwd", "nnr_composite_bwd", "nnr_mlp_dgrad", "nnr_mlp_wgrad", "nnr_ray_reduce",
           "nnr_se3_exp_fwd", "nnr_se3_exp_bwd", "nnr_inv4_fwd", "nnr_inv4_bwd", "nnr_ray_setup_fwd", "nnr_ray_setup_bwd",
           "nnr_depth_gather_fwd", "nnr_depth_gather_bwd", "nnr_render_loss", "nnr_pixels_from_index", "nnr_pc_nearest",
           "nnr_pc_error_bwd", "nnr_aux_workspace_floats", "nnr_aux_terms_fwd", "nnr_aux_terms_bwd", "nnr_randperm_prefix",
           "nnr_randperm_scratch_bytes", "nnr_ndc_rays_fwd", "nnr_ndc_rays_bwd",
           "nnr_depth_gather_affine_fwd", "nnr_depth_gather_affine_bwd", "nnr_prof_begin", "nnr_prof_end",
           "nnr_step_rays_fwd", "nnr_step_rays_bwd", "nnr_adam_step", "nnr_uniform_rows")


class Cfg(C.Structure):
    _fields_ = [("n_rays", C.c_int32), ("n_samples", C.c_int32), ("hidden", C.c_int32), ("flags", C.c_uint32)]


class Params(C.Structure):
    _fields_ = [("weight", C.c_void_p * N_LAYERS), ("bias", C.c_void_p * N_LAYERS)]


class AuxCfg(C.Structure):
    _fields_ = [("hd", C.c_int32), ("wd", C.c_int32), ("hr", C.c_int32), ("wr", C.c_int32), ("nearest_limit", C.c_float),
                ("flags", C.c_uint32), ("shard_lo", C.c_int32), ("shard_hi", C.c_int32), ("w_pc", C.c_float), ("w_rgbs", C.c_float)]


AUX_RGBS, AUX_PC, AUX_SCALE_PCS, AUX_DETACH_RGBS, AUX_SSIM, AUX_GRAD_K, AUX_AFFINE, AUX_SHIFT_FIRST, AUX_WEIGHTED, AUX_MATS_GRAD = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512


class StepCfg(C.Structure):        # nnr_step_cfg: the fused front end of a training step
    _fields_ = [(n, C.c_int32) for n in ("n_rays", "h", "w", "hd", "wd", "cam", "n_cams")] + [("flags", C.c_uint32), ("ref", C.c_int32)]


STEP_NORMALISE, STEP_USE_DIR, STEP_SHIFT_FIRST, STEP_FIX_LAST_SCALE, STEP_DETACH_REF = 1, 2, 4, 8, 16
STEP_BWD_SCRATCH_FLOATS = 528      # NNR_STEP_BWD_SCRATCH_FLOATS


class WgradJobB(C.Structure):      # bf16 training mode: one workgroup job (nnr_layout.h)
    _fields_ = [("d_base", C.c_int64), ("x_base", C.c_int64)] + \
               [(n, C.c_int32) for n in ("d_stride", "x_stride", "d_groups", "x_groups", "unit", "MT", "NT", "WR", "WC", "c0", "c1", "bias",
                                         "split", "next_split")] + \
               [("x2_base", C.c_int64), ("x2_stride", C.c_int32), ("x2_groups", C.c_int32)]


class WgradOutB(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("unit", "layer", "d_row", "n_rows", "w_row", "x_col", "n_cols", "w_col", "ldw", "bias",
                                         "first_job", "MT", "NT", "WR", "WC", "reserved")]


class WgradJob(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("layer", "MI", "NI", "d_plane", "d_col0", "d_valid", "x_plane", "x_col0",
                                         "x_valid", "row0", "wcol0", "rows_real", "cols_real", "ldw", "k0", "k1", "bias",
                                         "split", "next_split", "reserved")]


_lib = None


def load():
    """dlopen libnnr.so (built by nope-nerf_amd/csrc/build.py).  Raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first (python nope-nerf_amd/csrc/build.py or "
            "__graft_entry__.build()).  There is no CPU fallback for the render path.")
    lib = C.CDLL(LIB_PATH)
    vp, cfgp, i64 = C.c_void_p, C.POINTER(Cfg), C.c_int64
    lib.nnr_abi_version.restype = C.c_int
    lib.nnr_strerror.restype = C.c_char_p
    lib.nnr_strerror.argtypes = [C.c_int]
    lib.nnr_last_hip_error.restype = C.c_int
    for n in ("nnr_packed_floats", "nnr_workspace_floats", "nnr_plan_bytes"):
        getattr(lib, n).restype = C.c_size_t
        getattr(lib, n).argtypes = [cfgp]
    lib.nnr_plan_build.argtypes = [cfgp, vp]
    lib.nnr_plan_counts.argtypes = [cfgp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.nnr_pack_weights.argtypes = [cfgp, C.POINTER(Params), vp, vp]
    lib.nnr_render_fwd.argtypes = [cfgp] + [vp] * 13
    lib.nnr_render_bwd.argtypes = [cfgp, vp, vp, vp, C.POINTER(Params), vp, vp, vp, vp, vp, vp]
    lib.nnr_ws_plane.restype = i64
    lib.nnr_ws_plane.argtypes = [cfgp, C.c_int, C.POINTER(C.c_int32)]
    lib.nnr_ws_plane_layout.argtypes = [cfgp, C.c_int]
    lib.nnr_mlp_fwd.argtypes = [cfgp] + [vp] * 9
    lib.nnr_composite_fwd.argtypes = [cfgp] + [vp] * 6
    lib.nnr_composite_bwd.argtypes = [cfgp] + [vp] * 4
    lib.nnr_mlp_dgrad.argtypes = [cfgp, vp, vp, vp]
    lib.nnr_mlp_wgrad.argtypes = [cfgp, vp, C.POINTER(Params), vp, vp, vp]
    lib.nnr_ray_reduce.argtypes = [cfgp] + [vp] * 5
    i32, f32 = C.c_int32, C.c_float
    lib.nnr_se3_exp_fwd.argtypes = [vp, vp, i32, vp, vp]
    lib.nnr_se3_exp_bwd.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    lib.nnr_inv4_fwd.argtypes = [vp, vp, i32, vp]
    lib.nnr_inv4_bwd.argtypes = [vp, vp, vp, i32, vp]
    lib.nnr_ray_setup_fwd.argtypes = [vp] * 5 + [i32] * 3 + [vp] * 7
    lib.nnr_ray_setup_bwd.argtypes = [vp] * 5 + [i32] * 3 + [vp] * 11
    lib.nnr_depth_gather_fwd.argtypes = [vp, vp, vp] + [i32] * 5 + [vp]
    lib.nnr_depth_gather_bwd.argtypes = [vp, vp, vp] + [i32] * 5 + [vp]
    lib.nnr_pixels_from_index.argtypes = [vp, vp, i32, i32, i32, vp]
    lib.nnr_pc_nearest.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp]
    lib.nnr_randperm_prefix.argtypes = [vp, i64, i32, i32, C.c_uint64, C.c_uint64, vp, vp, vp]
    lib.nnr_uniform_rows.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp, vp]
    lib.nnr_depth_gather_affine_fwd.argtypes = [vp, vp, vp, vp, i32, vp] + [i32] * 5 + [vp]
    lib.nnr_depth_gather_affine_bwd.argtypes = [vp, vp, vp, vp, vp, i32, vp] + [i32] * 5 + [vp]
    lib.nnr_ndc_rays_fwd.argtypes = [vp, vp, vp, C.c_float, vp, vp, i32, vp]
    lib.nnr_ndc_rays_bwd.argtypes = [vp, vp, vp, C.c_float, vp, vp, vp, vp, i32, vp]
    lib.nnr_randperm_scratch_bytes.restype = C.c_size_t
    lib.nnr_randperm_scratch_bytes.argtypes = [i32]
    auxp = C.POINTER(AuxCfg)
    lib.nnr_aux_workspace_floats.restype = C.c_size_t
    lib.nnr_aux_workspace_floats.argtypes = [auxp]
    lib.nnr_aux_terms_fwd.argtypes = [auxp] + [vp] * 12
    lib.nnr_aux_terms_bwd.argtypes = [auxp] + [vp] * 15
    lib.nnr_pc_error_bwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]
    lib.nnr_render_loss.argtypes = [vp] * 5 + [i32] + [f32] * 4 + [i32] * 3 + [vp] * 6
    lib.nnr_step_rays_fwd.argtypes = [C.POINTER(StepCfg)] + [vp] * 19
    lib.nnr_step_rays_bwd.argtypes = [C.POINTER(StepCfg)] + [vp] * 20
    lib.nnr_adam_step.argtypes = [vp, vp]
    lib.nnr_prof_begin.argtypes = [i32]
    lib.nnr_prof_end.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    for n in EXPORTS:
        if not hasattr(lib, n):
            raise RuntimeError(f"libnnr.so does not export {n}")
    if lib.nnr_abi_version() != ABI_VERSION:
        raise RuntimeError("libnnr.so ABI version %d, these bindings expect %d: rebuild with `python nope-nerf_amd/csrc/build.py`"
                           % (lib.nnr_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        lib = load()
        msg = lib.nnr_strerror(rc).decode()
        if rc == -4:
            msg += f" (hipError {lib.nnr_last_hip_error()})"
        raise RuntimeError(f"{what} failed: {msg}")


def make_cfg(n_rays: int, n_samples: int, hidden: int, *, dist_alpha=False, white_bg=False, relu_sigma=False,
             train=False, bf16=False) -> Cfg:
    flags = (NNR_F_DIST_ALPHA if dist_alpha else 0) | (NNR_F_WHITE_BG if white_bg else 0) | \
            (NNR_F_RELU_SIGMA if relu_sigma else 0) | (NNR_F_TRAIN if train else 0) | (NNR_F_BF16 if bf16 else 0)
    if not bf16 and _fp32_products in ("split3", "split2"):
        flags |= NNR_F_SPLIT3
        if _fp32_products == "split2":
            flags |= NNR_F_SPLIT2
    return Cfg(int(n_rays), int(n_samples), int(hidden), flags)


def ptr(t):
    """Device pointer of a torch tensor (or None -> NULL).  The tensor must be contiguous fp32/int32 storage."""
    if t is None:
        return None
    assert t.is_contiguous(), "nnr: tensors crossing the C ABI must be contiguous"
    return C.c_void_p(t.data_ptr())


def params_struct(weights, biases) -> Params:
    p = Params()
    for i in range(N_LAYERS):
        p.weight[i] = weights[i].data_ptr()
        p.bias[i] = biases[i].data_ptr()
    return p


def stream():
    """The current torch stream of the current device as the `void* stream` argument of the C ABI.  torch._C's raw accessor is
    a tenth of the cost of torch.cuda.current_stream().cuda_stream (a dozen lookups per training step)."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return C.c_void_p(raw(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
