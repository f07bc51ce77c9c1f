"""Training losses.  Same class, keys and weighting as reference model/losses.py:16-218; device-agnostic (the reference
hard-codes .cuda()).  The two heads that feed the fused backward are `get_rgb_full_loss` (L1|L2 *sum* over the batch
divided by the number of rays, losses.py:27-32) and `get_depth_loss` (L1 sum / number of valid rays, :59-64); the rest
are per-image auxiliary terms that stay in stock torch.
"""
import torch
from torch import nn
from torch.nn import functional as F


class Loss_Eval(nn.Module):
    """MSE on rgb, used by the test-time pose optimisation (reference losses.py:6-14)."""

    def forward(self, rgb_pred, rgb_gt):
        return {'loss': F.mse_loss(rgb_pred, rgb_gt)}


class SSIM(nn.Module):
    """3x3 average-pool SSIM dissimilarity map in [0,1] (reference losses.py:222-252)."""

    def __init__(self):
        super().__init__()
        self.pool = nn.AvgPool2d(3, 1)
        self.refl = nn.ReflectionPad2d(1)
        self.C1, self.C2 = 0.01 ** 2, 0.03 ** 2

    def forward(self, x, y):
        x, y = self.refl(x), self.refl(y)
        mu_x, mu_y = self.pool(x), self.pool(y)
        var_x = self.pool(x * x) - mu_x ** 2
        var_y = self.pool(y * y) - mu_y ** 2
        cov = self.pool(x * y) - mu_x * mu_y
        num = (2 * mu_x * mu_y + self.C1) * (2 * cov + self.C2)
        den = (mu_x ** 2 + mu_y ** 2 + self.C1) * (var_x + var_y + self.C2)
        return torch.clamp((1 - num / den) / 2, 0, 1)


compute_ssim_loss = SSIM()


_ZEROS = {}


def _zero(ref):
    """The constant 0 that inactive loss terms report, one tensor per device (a fresh torch.zeros(()) is a kernel launch, and
    a step reports five to eight of them).  Read-only by convention: callers log it or add it, never write into it."""
    z = _ZEROS.get(ref.device)
    if z is None:
        z = _ZEROS[ref.device] = torch.zeros((), dtype=torch.float32, device=ref.device)
    return z


class Loss(nn.Module):
    def __init__(self, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.depth_loss_type = cfg['depth_loss_type']

    # ---- heads on the render path ----
    def get_rgb_full_loss(self, rgb_values, rgb_gt, rgb_loss_type='l2'):
        diff = rgb_values - rgb_gt
        total = diff.abs().sum() if rgb_loss_type == 'l1' else (diff * diff).sum()
        return total / float(rgb_values.shape[1])

    def depth_loss_dpt(self, pred_depth, gt_depth, weight=None):
        """Scale/shift-invariant depth loss (median / mean-abs-deviation normalisation)."""
        def normalise(d):
            t = torch.median(d)
            return (d - t) / torch.mean(torch.abs(d - t))
        err = F.mse_loss(normalise(pred_depth), normalise(gt_depth), reduction='none')
        if weight is None:
            return err.mean()
        return (err * weight).sum() / (weight.sum() + 1e-8)

    def get_depth_loss(self, depth_pred, depth_gt):
        if self.depth_loss_type == 'l1':
            return (depth_pred - depth_gt).abs().sum() / float(depth_pred.shape[0])
        if self.depth_loss_type == 'invariant':
            return self.depth_loss_dpt(depth_pred, depth_gt)
        raise ValueError(self.depth_loss_type)

    # ---- per-image auxiliary terms ----
    def mean_on_mask(self, diff, valid_mask, shard=None):
        """Mean of diff over the valid points.  shard = (lo, hi): only the points [lo, hi) of the flattened (h, w) grid enter the
        sum, the count stays global -- a data-parallel rank's share; the shares of all ranks add up to the mean."""
        mask = valid_mask.expand_as(diff)
        n = mask.sum()
        if n > 0:
            if shard is not None:
                mine = torch.zeros(valid_mask.shape[0], valid_mask.shape[1] * valid_mask.shape[2], dtype=torch.bool, device=diff.device)
                mine[:, shard[0]:shard[1]] = True
                mask = mask & mine.view(valid_mask.shape).expand_as(diff)
            return diff[mask].sum() / n
        print('============invalid mask==========')
        return _zero(diff)

    def get_weight_dist_loss(self, t_list):
        step = (t_list - t_list.roll(shifts=1, dims=0))[1:].norm(dim=1)     # consecutive camera distances
        accel = (step - step.roll(shifts=1))[1:]
        return step.mean(), accel.pow(2.0).mean()

    def comp_closest_pts_idx_with_split(self, pts_src, pts_des):
        """Index of the nearest pts_des column for every pts_src column; (3,S),(3,D) -> (S,)."""
        out = []
        for chunk in torch.split(pts_src, 500000, dim=1):
            d = torch.linalg.norm(chunk[:, :, None] - pts_des[:, None, :], dim=0)
            out.append(torch.argmin(d, dim=1))
        return torch.cat(out)

    def comp_point_point_error(self, Xt, Yt, shard=None):
        """(3,S), (3,D) -> mean distance of every Xt column to its nearest Yt column (reference losses.py:143-148).
        shard = (lo, hi): the sum over the source columns [lo, hi) only, still divided by S (a data-parallel rank's share of the
        mean: the nearest-neighbour search, the O(S D) part, shrinks by the world size)."""
        S = Xt.shape[1]
        if shard is not None:
            Xt = Xt[:, shard[0]:shard[1]]
            if Xt.shape[1] == 0:
                return Xt.sum() * 0.0
        if Xt.is_cuda:   # one HIP launch instead of the (3, S, D) difference tensor; same argmin, same fp32 distances
            from nnr import pointcloud
            err = pointcloud.point_point_error(Xt.permute(1, 0), Yt.permute(1, 0))
            return err if shard is None else err * (Xt.shape[1] / float(S))
        idx = self.comp_closest_pts_idx_with_split(Xt, Yt)
        d = torch.linalg.norm(Xt - Yt[:, idx], dim=0)
        return d.mean() if shard is None else d.sum() / float(S)

    def get_pc_loss(self, Xt, Yt, shard=None):
        if self.cfg['match_method'] != 'dense':
            raise ValueError(self.cfg['match_method'])
        x, y = Xt[0].permute(1, 0), Yt[0].permute(1, 0)
        return self.comp_point_point_error(x, y, shard) + self.comp_point_point_error(y, x, shard)

    def get_depth_consistency_loss(self, d1_proj, d2, d2_proj=None, d1=None):
        loss = (d1_proj - d2).abs().sum() / float(d1_proj.shape[1])
        if d2_proj is not None:
            loss = 0.5 * loss + 0.5 * (d2_proj - d1).abs().sum() / float(d2_proj.shape[1])
        return loss

    def get_rgb_s_loss(self, rgb1, rgb2, valid_points, shard=None):
        diff = (rgb1 - rgb2).abs().clamp(0, 1)
        if self.cfg['with_ssim'] == True:  # noqa: E712  (YAML booleans)
            diff = 0.15 * diff + 0.85 * compute_ssim_loss.to(diff.device)(rgb1, rgb2)
        return self.mean_on_mask(diff, valid_points, shard)

    def aux_terms(self, ref, t_list=None, X=None, Y=None, rgb_pc1=None, rgb_pc1_proj=None, valid_points=None, d1_proj=None,
                  d2=None, d2_proj=None, d1=None, weights={}, fused_aux=None, point_shard=None, fused_aux_sum=None, **kwargs):
        """The per-image terms (point cloud, surface reprojection, trajectory smoothness, depth consistency) and their
        weighted sum; `ref` is any tensor on the target device.  `fused_aux` = (loss_pc, loss_rgb_s) already computed by the
        fused HIP path (nnr/aux.py) from the same inputs, `fused_aux_sum` their weighted sum when they are the only active terms.  `point_shard` = (lo, hi): under data parallelism loss_pc and
        loss_rgb_s are this rank's share (sums over its source points, global normalisers; model/training.py)."""
        z = _zero(ref)
        on = lambda k: weights[k] != 0.0
        if fused_aux is not None:
            l_pc, l_rgbs = fused_aux
        else:
            l_pc = self.get_pc_loss(X, Y, point_shard) if on('pc_weight') else z
            l_rgbs = self.get_rgb_s_loss(rgb_pc1, rgb_pc1_proj, valid_points, point_shard) if on('rgb_s_weight') else z
        parts = {
            'loss_pc': l_pc if on('pc_weight') else z,
            'loss_rgb_s': l_rgbs if on('rgb_s_weight') else z,
            'loss_depth_consistency': self.get_depth_consistency_loss(d1_proj, d2, d2_proj, d1)
            if on('depth_consistency_weight') else z,
        }
        if on('weight_dist_2nd_loss') or on('weight_dist_1st_loss'):
            parts['loss_dist_1st'], parts['loss_dist_2nd'] = self.get_weight_dist_loss(t_list)
        else:
            parts['loss_dist_1st'], parts['loss_dist_2nd'] = z, z
        if fused_aux_sum is not None:      # pc_weight * loss_pc + rgb_s_weight * loss_rgb_s from the fused path's finishing kernel: nothing else is active
            return fused_aux_sum, parts
        total = None   # only the active terms enter the sum: no arithmetic on constant zeros (each would be a kernel launch)
        for wk, pk in (('weight_dist_1st_loss', 'loss_dist_1st'), ('weight_dist_2nd_loss', 'loss_dist_2nd'),
                       ('pc_weight', 'loss_pc'), ('rgb_s_weight', 'loss_rgb_s'),
                       ('depth_consistency_weight', 'loss_depth_consistency')):
            if on(wk):
                term = weights[wk] * parts[pk]
                total = term if total is None else total + term
        return total, parts

    def forward(self, rgb_pred, rgb_gt, depth_pred=None, depth_gt=None, weights={}, rgb_loss_type='l2', **kwargs):
        z = _zero(rgb_gt)
        on = lambda k: weights[k] != 0.0
        rendering = on('rgb_weight') or on('depth_weight')
        aux, parts = self.aux_terms(rgb_gt, weights=weights, **kwargs)
        parts['loss_rgb'] = self.get_rgb_full_loss(rgb_pred, rgb_gt, rgb_loss_type) if on('rgb_weight') else z
        parts['loss_depth'] = self.get_depth_loss(depth_pred, depth_gt) if on('depth_weight') else z
        loss = weights['rgb_weight'] * parts['loss_rgb'] + weights['depth_weight'] * parts['loss_depth']
        if aux is not None:
            loss = loss + aux
        if torch.isnan(loss):
            raise FloatingPointError('NaN loss (the reference drops into breakpoint() here, losses.py:204-205)')
        out = {'loss': loss, 'l2_mean': F.mse_loss(rgb_pred, rgb_gt) if rendering else z}
        out.update(parts)
        return out
