"""Renderer -- host side of the volumetric-rendering hot path.

Same constructor, `forward` dispatch and output dictionary as the reference's model/rendering.py:11-34,159-166, but
the body of `nope_nerf` (reference :36-167) is: a few O(R) torch ops that turn (pixels, depth, K, W, S) into per-ray
sampling origins/directions inside the autograd graph (so gradients reach the learnable pose / focal / depth
distortion exactly as they do through the reference's matrix inverses), and ONE call into the fused HIP operator
`nnr.render_rays` for everything per-sample: stratified / NDC sampling, positional encoding, the 12-layer MLP,
alpha-compositing, and their backward.

`rendering.normal_loss: True` (reference :133-143; off by default, and no loss term of the reference consumes it) adds the
normal-consistency vector `out['normal']`: second-order autograd through the MLP trunk for the 2 M surface points, in stock torch
over the same nn.Linear parameters (OfficialStaticNerf.gradient) -- the fused kernels still render.

Not provided (outside the hot path): the phong / ray-marching visualiser
(`phong_renderer`, `ray_marching`, `secant`); they raise NotImplementedError.
"""
import torch
import torch.nn as nn

import nnr
from nnr import camera
from .common import get_ndc_rays_fxfy, pixel_to_world_matrix

epsilon = 1e-6   # the transmittance epsilon of the reference (rendering.py:9) -- applied inside the composite kernel


class RenderOutput(dict):
    """The renderer's output dictionary (reference rendering.py:159-166).  `depth_pred` / `depth_gt` -- the per-ray
    values compacted by the validity mask -- are materialised on first access: boolean-mask indexing costs a dozen small
    kernels and a device->host sync, and the trainer's fused loss works on the dense tensors + mask instead
    (keys `dist_dense`, `d_gt_dense`, `mask`)."""

    _LAZY = ('depth_pred', 'depth_gt', 'alpha', 'z_vals')

    def __missing__(self, key):
        if key not in self._LAZY:
            raise KeyError(key)
        if key in ('alpha', 'z_vals'):
            # forward-only renders composite inside the MLP kernel and write nothing per sample; whoever does want the per-sample
            # weights of an evaluation render gets them from a second, unfused pass
            self['alpha'], self['z_vals'] = dict.__getitem__(self, '_samples')()
            return dict.__getitem__(self, key)
        mask = dict.__getitem__(self, 'mask')
        if key == 'depth_pred':
            val = dict.__getitem__(self, 'dist_dense')[mask]
        else:
            val = dict.__getitem__(self, 'd_gt_dense')[mask]
            if dict.__getitem__(self, 'ndc'):
                val = 1 - 1 / val                                                   # rendering.py:157-158
        self[key] = val
        return val

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._LAZY

    def get(self, key, default=None):
        return self[key] if key in self else default


def _forward_only_samples(pts_o, pts_d, view, z_lo, z_hi, jitter, net, kw):
    """alpha / z_vals of an evaluation render, fetched late.  Under torch.no_grad() whatever the caller's mode is by then: read after the
    caller's no_grad block has ended, the operator would otherwise see grad-requiring weights and take the TRAINING path (full
    activation stash, a multi-GB workspace pinned by an autograd node) for two tensors nobody differentiates."""
    with torch.no_grad():
        return nnr.render_rays(pts_o, pts_d, view, z_lo, z_hi, jitter, net.weights(), net.biases(), **kw)[2:]


class Renderer(nn.Module):
    def __init__(self, model, cfg, device=None, **kwargs):
        super().__init__()
        self._device = device
        self.depth_range = cfg['depth_range']
        self.n_max_network_queries = cfg['n_max_network_queries']   # kept for API parity; the fused kernel needs no chunking
        self.white_background = cfg['white_background']
        self.cfg = cfg
        self.model = model.to(device)
        self._z_cache = {}
        self.jitter_window = None   # (first ray, rays in the whole step) when this process renders a shard
        self.normal_window = None   # (validity of ALL rays of the step, first ray) for the normal term of a shard

    def forward(self, pixels, depth, camera_mat, world_mat, scale_mat, rendering_technique, add_noise=True,
                eval_=False, it=1000000, rays=None):
        if rendering_technique == 'nope_nerf':
            return self.nope_nerf(pixels, depth, camera_mat, world_mat, scale_mat, it=it, add_noise=add_noise, eval_=eval_, rays=rays)
        if rendering_technique == 'phong_renderer':
            return self.phong_renderer(pixels, camera_mat, world_mat, scale_mat, it=it)
        raise ValueError('unknown rendering technique %r' % (rendering_technique,))

    # ------------------------------------------------------------------------------------------------ z tables
    def _z_tables(self, n_samples, near, far, stratified, device):
        """Per-sample interval [lo_j, hi_j]; z_j = lo_j + (hi_j - lo_j) * u_j (reference rendering.py:95-96,184-190).
        Built once per configuration on the CPU (bit-identical to the reference's CPU arithmetic), then cached on the
        device; the kernel applies the jitter."""
        key = (n_samples, float(near), float(far), bool(stratified), str(device))
        hit = self._z_cache.get(key)
        if hit is None:
            rng = torch.tensor([near, far])
            z = torch.linspace(0., 1., steps=n_samples)
            z = rng[0] * (1. - z) + rng[1] * z
            if stratified:
                mid = .5 * (z[1:] + z[:-1])
                lo, hi = torch.cat([z[:1], mid]), torch.cat([mid, z[-1:]])
            else:
                lo = hi = z
            hit = (lo.contiguous().to(device), hi.contiguous().to(device))
            self._z_cache[key] = hit
        return hit

    # ------------------------------------------------------------------------------------------------ hot path
    def nope_nerf(self, pixels, depth, camera_mat, world_mat, scale_mat, add_noise=False, it=100000, eval_=False, rays=None):
        cfg = self.cfg
        batch_size, n_rays, _ = pixels.shape
        if batch_size != 1:
            raise NotImplementedError("batch size 1 is baked into the reference (rendering.py:86-87); so it is here")
        n_samples = cfg['num_points'] - cfg['outside_steps']
        device = pixels.device

        # --- rays (reference :54-87).  inv(S) inv(W) inv(K) stays inside autograd: pose / focal gradients flow here ---
        if rays is not None:
            # the trainer's fused front end (nnr.camera.step_rays) has generated them together with everything before them
            origin, ray, view, ray_norm, d_gt, object_mask = rays
        elif pixels.is_cuda:
            # one HIP launch (nnr_ray_setup_fwd; backward: nnr_ray_setup_bwd) for the three inverses, the unprojection,
            # norms, d_gt and the validity mask
            origin, ray, view, ray_norm, d_gt, object_mask = camera.ray_setup(
                pixels, depth, camera_mat, world_mat, scale_mat, bool(cfg['normalise_ray']), bool(cfg['use_ray_dir']))
        else:
            m = pixel_to_world_matrix(camera_mat, world_mat, scale_mat)[0]             # (4,4)
            cam = m[:3, 3]                                                              # camera centre
            pix_h = torch.cat([pixels[0], torch.ones(n_rays, 1, device=device)], dim=-1)   # (R,3) = (x', y', 1)
            ray = pix_h @ m[:3, :3].t()                                                 # pixels_world - camera_world
            ray_norm = ray.norm(2, -1)
            d_gt = (ray * depth[0]).norm(2, -1)                                         # |points_world - camera_world|
            if cfg['normalise_ray']:
                ray = ray / ray_norm.unsqueeze(-1)
            else:
                d_gt = d_gt / ray_norm
            object_mask = torch.isfinite(d_gt) & (d_gt != 0)                            # :73-87
            view = -ray if cfg['use_ray_dir'] else torch.ones_like(ray)                 # :104-105,194-195
            origin = cam.unsqueeze(0).expand(n_rays, 3)

        # --- per-ray sampling frame ---
        jitter = None
        if cfg['sample_option'] == 'ndc':                                           # :168-180
            if pixels.is_cuda and not camera_mat.requires_grad:       # one launch each way instead of ~25 + ~50 torch ops
                pts_o, pts_d = camera.ndc_rays(origin, ray, camera_mat, 1.0)
            else:                                                    # CPU stand-in; a focal that is being learned
                focal = torch.cat([camera_mat[:, 0, 0], camera_mat[:, 1, 1]])
                pts_o, pts_d = get_ndc_rays_fxfy(focal, 1.0, rays_o=origin, rays_d=ray)
            z_lo, z_hi = self._z_tables(n_samples, 0., 1., False, device)
        elif cfg['sample_option'] == 'uniform':                                     # :182-197
            pts_o, pts_d = origin, ray
            z_lo, z_hi = self._z_tables(n_samples, self.depth_range[0], self.depth_range[1], bool(add_noise), device)
            if add_noise:
                if self.jitter_window is None:
                    jitter = torch.rand(batch_size, n_rays, n_samples, device=device)   # same draw as the reference (:189)
                else:   # data-parallel shard: this rank's rows of the whole step's jitter tensor (model/training.py) -- drawn alone,
                        # O(rays of this rank): nnr.sampling.rand_rows reproduces torch.rand's values and its generator side effects
                    lo, total = self.jitter_window
                    if batch_size == 1 and torch.device(device).type == 'cuda':
                        from nnr import sampling
                        jitter = sampling.rand_rows(total * n_samples, lo * n_samples, n_rays * n_samples, device).view(1, n_rays, n_samples)
                    else:
                        jitter = torch.rand(batch_size, total, n_samples, device=device)[:, lo:lo + n_rays].contiguous()
        else:
            raise ValueError('unknown sample_option %r' % (cfg['sample_option'],))

        net = self.model
        kw = dict(hidden=net.hidden_dim, dist_alpha=bool(cfg['dist_alpha']), white_bg=bool(self.white_background),
                  relu_sigma=(net.occ_activation != 'softplus'),
                  bf16=(str(cfg.get('mfma_dtype', 'fp32')).lower() == 'bf16'))   # rendering.mfma_dtype: fp32 (default) | bf16
        lazy_samples = not torch.is_grad_enabled()   # evaluation / visualisation: per-sample outputs only if somebody reads them
        rgb, dist_pred, alpha, z_val = nnr.render_rays(pts_o, pts_d, view, z_lo, z_hi, jitter, net.weights(), net.biases(),
                                                       samples=not lazy_samples, **kw)

        diff_norm = None
        if cfg['normal_loss'] and not eval_:
            diff_norm = self._normal_consistency(origin, ray, d_gt, object_mask, it)

        if eval_ and cfg['normalise_ray']:                                          # distance -> depth for evaluation (:150-154)
            dist_pred = dist_pred / ray_norm
            d_gt = d_gt / ray_norm
        return RenderOutput({
            'rgb': rgb.reshape(batch_size, -1, 3),
            'normal': diff_norm,
            **({'_samples': lambda: _forward_only_samples(pts_o, pts_d, view, z_lo, z_hi, jitter, net, kw)}
               if lazy_samples else {'z_vals': z_val, 'alpha': alpha}),
            # dense per-ray values + validity mask; 'depth_pred' / 'depth_gt' (masked) are derived lazily from these
            'dist_dense': dist_pred,
            'd_gt_dense': d_gt,
            'mask': object_mask,
            'ndc': cfg['sample_option'] == 'ndc',
        })

    def _normal_consistency(self, origin, ray, d_gt, mask, it):
        """|n(x) - n(x + eps)| at the surface points x the mono depth puts on the valid rays, n = normalised -d(sigma)/dx
        (reference rendering.py:76-93,133-141).  Stock autograd over the MLP's own parameters (OfficialStaticNerf.gradient); the
        perturbation draws torch.rand_like(surface_points) right after the jitter draw, as the reference does.  A data-parallel
        shard draws the whole step's perturbations and keeps the rows of its own valid rays, so every rank's generator stays where
        the single-process run leaves it."""
        surface = (origin + ray * d_gt.unsqueeze(-1))[mask]            # dists == d_i on the valid rays (:80-82,92)
        n = surface.shape[0]
        if self.normal_window is None:
            noise = torch.rand_like(surface)
        else:
            # data parallel: the rank draws the whole step's perturbations and keeps its rows, so that every rank's generator ends where
            # the single-process run's does (the reference draws rand_like(surface) over ALL valid rays).  The two int(...) below are
            # blocking device-to-host reads -- the only ones in a training step, taken only with rendering.normal_loss on AND more than
            # one rank; the surface[mask] boolean indexing above synchronises in the single-process path as well (it is the reference's
            # own expression).  Everything else in the step stays on the device (deferred NaN flag, device-side counts).
            valid_all, lo = self.normal_window
            first = int(valid_all[:lo].sum())
            noise = torch.rand(int(valid_all.sum()), 3, dtype=surface.dtype, device=surface.device)[first:first + n]
        neigh = surface + (noise - 0.5) * 0.01
        g = self.model.gradient(torch.cat([surface, neigh], dim=0), it)
        normals = g[:, 0, :] / (g[:, 0, :].norm(2, dim=1).unsqueeze(-1) + 10 ** (-5))
        return torch.norm(normals[:n] - normals[n:], dim=-1)

    # ------------------------------------------------------------------------------------------------ not on the hot path
    def phong_renderer(self, *args, **kwargs):
        raise NotImplementedError("phong_renderer (geometry visualisation, reference rendering.py:202-274) is outside the "
                                  "HIP hot path; set training.vis_geo: False")

    def ray_marching(self, *args, **kwargs):
        raise NotImplementedError("ray_marching / secant (reference rendering.py:277-418) serve only phong_renderer")
