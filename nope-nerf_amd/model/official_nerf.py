"""OfficialStaticNerf -- parameter container + standalone evaluation of the NeRF MLP.

Keeps the reference's constructor, sub-module names and state_dict keys (reference model/official_nerf.py:9-44:
layers0.{0,2,4,6}, layers1.{0,2,4,6}, fc_density, fc_feature, rgb_layers.0, fc_rgb) so that `train.py`, Adam,
`CheckpointIO` and `scheduling_mode == 'reset'` (train.py:342-344 walks nn.Linear children) keep working, but no
arithmetic happens in these nn.Linear modules: `model.Renderer` hands their tensors to the fused HIP kernels
(nnr.render_rays), and `forward` below evaluates the same kernels on free-standing points.
"""
import torch
import torch.nn as nn

import nnr
from nnr import ops as _ops

POS_LEVELS, DIR_LEVELS = 10, 4   # the reference hard-codes both in infer_occ/forward (official_nerf.py:61,87)


class OfficialStaticNerf(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        mcfg, rcfg = cfg['model'], cfg['rendering']
        width = mcfg['hidden_dim']
        n_pos = 3 * (2 * mcfg['pos_enc_levels'] + 1)
        n_dir = 3 * (2 * mcfg['dir_enc_levels'] + 1)
        if (mcfg['pos_enc_levels'], mcfg['dir_enc_levels']) != (POS_LEVELS, DIR_LEVELS):
            # the reference would build mismatching layer shapes and crash in forward; say so up front
            raise NotImplementedError("pos_enc_levels/dir_enc_levels other than 10/4 are not supported "
                                      "(the reference evaluates the encodings with 10/4 regardless)")
        self.white_bkgd = rcfg['white_background']
        self.dist_alpha = rcfg['dist_alpha']
        self.occ_activation = mcfg['occ_activation']
        self.hidden_dim = width

        def trunk(first_in):
            mods, fan_in = [], first_in
            for _ in range(4):
                mods += [nn.Linear(fan_in, width), nn.ReLU()]
                fan_in = width
            return nn.Sequential(*mods)

        self.layers0 = trunk(n_pos)
        self.layers1 = trunk(width + n_pos)          # skip connection: input is [h, posenc]
        self.fc_density = nn.Linear(width, 1)
        self.fc_feature = nn.Linear(width, width)
        self.rgb_layers = nn.Sequential(nn.Linear(width + n_dir, width // 2), nn.ReLU())
        self.fc_rgb = nn.Linear(width // 2, 3)
        self.sigmoid = nn.Sigmoid()
        with torch.no_grad():                       # bias overrides, official_nerf.py:39-44
            self.fc_density.bias.fill_(0.1)
            self.fc_rgb.bias.fill_(0.8 if self.white_bkgd else 0.02)

    # -- tensors in the order the C ABI expects (state_dict order of the 12 nn.Linear) --
    def linear_layers(self):
        return [self.get_submodule(n) for n in nnr.LAYER_NAMES]

    def weights(self):
        return [m.weight for m in self.linear_layers()]

    def biases(self):
        return [m.bias for m in self.linear_layers()]

    def forward(self, p, ray_d=None, only_occupancy=False, return_logits=False, return_addocc=False,
                noise=False, it=100000, **kwargs):
        """Evaluate the MLP on points p (S,3) with view directions ray_d (S,3) -- reference
        official_nerf.py:69-96.  Forward-only here (the differentiable route is Renderer -> nnr.render_rays)."""
        if torch.is_grad_enabled() and (p.requires_grad or any(w.requires_grad for w in self.weights())):
            raise NotImplementedError("OfficialStaticNerf.forward is forward-only on the HIP path; wrap the call in "
                                      "torch.no_grad() or go through model.Renderer for gradients")
        S = p.shape[0]
        view = ray_d if ray_d is not None else torch.ones_like(p)
        rgb, raw = _ops.mlp_points(p, view, self.weights(), self.biases(), hidden=self.hidden_dim)
        occ = torch.nn.functional.softplus(raw) if self.occ_activation == 'softplus' else raw.relu()
        if not self.dist_alpha:
            occ = 1 - torch.exp(-1.0 * occ)
        occ = occ.view(S, 1)
        if only_occupancy:
            return occ
        if ray_d is None:
            return None
        return (rgb, occ) if return_addocc else rgb

    def infer_occ(self, p):
        """Trunk activations and raw density of points p (..., 3) through the nn.Linear children in stock torch (reference
        official_nerf.py:60-67).  This is the UNSUPPORTED-configuration route of SURVEY section 8(b): the normal-consistency term needs
        d(sigma)/dp with create_graph=True, i.e. double backward through the trunk, which the fused kernels do not provide -- so
        the few surface points it is evaluated on (2 M <= 2 R per step, against R N samples in the fused render) go through
        autograd over the SAME parameters.  Works on any device."""
        enc = encode_position(p, levels=POS_LEVELS, inc_input=True)
        x = self.layers0(enc)
        x = self.layers1(torch.cat([x, enc], dim=-1))        # skip connection, input order [h, posenc] (official_nerf.py:63)
        return x, self.fc_density(x)

    def gradient(self, p, it):
        """-d(raw density)/dp as (S, 1, 3), differentiable (reference official_nerf.py:46-58)."""
        with torch.enable_grad():
            p.requires_grad_(True)
            _, y = self.infer_occ(p)
            grads = torch.autograd.grad(outputs=y, inputs=p, grad_outputs=torch.ones_like(y), create_graph=True, retain_graph=True,
                                        only_inputs=True, allow_unused=True)[0]
            return -grads.unsqueeze(1)


def encode_position(input, levels, inc_input):
    """gamma_L of the last axis, (..., C) -> (..., C (2L [+ 1])): [x,] sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x),
    every block C wide (reference model/official_nerf.py:99-119).  The render kernels compute this in registers; the function is
    kept for callers that import it."""
    freqs = 2.0 ** torch.arange(levels, dtype=input.dtype, device=input.device)
    scaled = input.unsqueeze(-2) * freqs.view(-1, 1)                                    # (..., L, C)
    waves = torch.stack([torch.sin(scaled), torch.cos(scaled)], dim=-2)                 # (..., L, 2, C)
    waves = waves.reshape(*input.shape[:-1], 2 * levels * input.shape[-1])
    return torch.cat([input, waves], dim=-1) if inc_input else waves
