"""Trainer -- the per-iteration training step; constructor, `train_step`, `render_visdata` and the returned dict keys
follow reference model/training.py:14-378 so that train.py drives it unchanged.

What is different underneath:
  * the render call (training.py:266-274) lands in the fused HIP operator through model.Renderer;
  * data-parallel training: when torch.distributed is initialised, every rank sees the same image, the same
    `ray_idx` permutation and the same jitter stream (same seed), renders only its contiguous slice of the rays, scales
    the per-ray losses by the *global* ray / valid-depth counts, and ONE flat all-reduce (RCCL over xGMI) of all
    gradients after backward reproduces the single-GPU gradient.  The point-cloud / re-projection
    losses (training.py:280-365) are means over source points: rank k sums over its shard of the sampling grid with the
    global normalisers (the O(S^2) nearest-neighbour search shrinks by the world size); the O(cameras) trajectory terms
    are replicated and weighted 1/world_size.
The per-image losses run as two fused HIP calls (nnr/aux.py); stock torch only on the CPU and for batches of images.
"""
import logging
import math
import os

import numpy as np
import torch
from PIL import Image
from torch.nn import functional as F

from model.common import arange_pixels, get_tensor_values, project_to_cam, transform_to_world
from model.losses import Loss, _zero
from nnr import camera, parallel, sampling

logger_py = logging.getLogger(__name__)

_WEIGHT_KEYS = ('rgb_weight', 'depth_weight', 'pc_weight', 'rgb_s_weight', 'depth_consistency_weight',
                'weight_dist_2nd_loss', 'weight_dist_1st_loss')
_CFG_KEYS = ('detach_gt_depth', 'pc_ratio', 'match_method', 'shift_first', 'detach_ref_img', 'scale_pcs',
             'detach_rgbs_scale', 'vis_reprojection_every', 'nearest_limit', 'annealing_epochs')


def _save_png(arr_u8, path):
    Image.fromarray(arr_u8).save(path)


def _use_fused_adam(opt):
    """the reference steps up to four torch.optim.Adam instances per iteration; torch's default
    ('foreach') implementation is ~7 launches per optimiser (21 of the ~90 small launches of a step, 0.18 ms).  The same
    class has a single-kernel implementation (fused=True, the same update rule and the same state_dict layout), which is
    selected here for plain Adam on GPU parameters.  Anything else (other optimisers, CPU parameters, amsgrad / capturable /
    differentiable variants) is left alone."""
    if type(opt) is not torch.optim.Adam:
        return False
    params = [p for g in opt.param_groups for p in g['params']]
    if not params or not all(p.is_cuda and p.dtype == torch.float32 for p in params):
        return False
    if any(g.get('amsgrad') or g.get('capturable') or g.get('differentiable') or g.get('maximize') for g in opt.param_groups):
        return False
    for g in opt.param_groups:
        g['fused'], g['foreach'] = True, False
    for p in params:   # state restored before this point keeps its step counter on the host: the fused kernel wants it beside p
        st = opt.state.get(p)
        if st and 'step' in st:
            st['step'] = torch.as_tensor(float(st['step']), dtype=torch.float32, device=p.device)
    return True


class _EarlyScalar(torch.Tensor):
    """A logged loss scalar whose VALUE is already on its way to the host.

    The reference's loop reads three scalars of every step with `.item()` (train.py:211-214) -- on a device tensor that is a blocking
    copy on the step's stream, i.e. it waits for the step's backward and optimizer kernels too, and the next step's host work (config
    annealing, a dozen launches of the front end) then runs with the GPU idle: 0.4 ms of a 3.5 ms step, 1.3 ms with the first-phase
    per-image losses (profiles/r04/b_scene_loop.json).  The values exist as soon as the FORWARD is done: train_step copies them into
    pinned host memory right behind the loss kernel (one stack launch + one asynchronous copy, enqueued before the backward) and returns
    these tensors -- device tensors in every respect (same data, `.detach().cpu()`, formatting, arithmetic), except that `.item()` waits
    for that copy's event only and answers from the host copy.  Same value, no drain of the launch queue."""
    __torch_function__ = torch._C._disabled_torch_function_impl

    def item(self):
        early = getattr(self, '_early', None)
        if early is None:
            return torch.Tensor.item(self)
        slot, generation, index, event = early
        if slot['generation'] != generation:      # a scalar of an older step whose host buffer has been reused since: the device copy is the value
            return torch.Tensor.item(self)
        event.synchronize()
        return slot['host'][index].item()

    # copies and pickles are PLAIN tensors with the same data: the host slot and its event stay with the step that made them
    # (copy.deepcopy / torch.save of a logged scalar must not try to copy a torch.cuda.Event -- ADVICE r04)
    def __deepcopy__(self, memo):
        return self.as_subclass(torch.Tensor).detach().clone()

    def __reduce_ex__(self, proto):
        return self.as_subclass(torch.Tensor).detach().clone().__reduce_ex__(proto)


class Trainer(object):
    def __init__(self, model, optimizer, cfg, device=None, optimizer_pose=None, pose_param_net=None,
                 optimizer_focal=None, focal_net=None, optimizer_distortion=None, distortion_net=None, **kwargs):
        self.model, self.optimizer, self.device = model, optimizer, device
        self.optimizer_pose, self.pose_param_net = optimizer_pose, pose_param_net
        self.optimizer_focal, self.focal_net = optimizer_focal, focal_net
        self.optimizer_distortion, self.distortion_net = optimizer_distortion, distortion_net
        self.n_training_points = cfg['n_training_points']
        self.rendering_technique = cfg['type']
        self.vis_geo = cfg['vis_geo']
        for k in _CFG_KEYS + _WEIGHT_KEYS:
            setattr(self, k, cfg[k])
        self.loss = Loss(cfg)
        self._warned_geo = False
        self._one = None
        self._nan_flag = None      # (pinned host flag, event) of the previous step's isnan(loss), read one step late
        self._nan_host = None
        self._early_host = None    # four pinned buffers for the step's logged scalars (_EarlyScalar), used in turn
        self._nan_with_early = False
        self.early_scalars = bool(cfg.get('early_scalars', True))   # training.early_scalars: False returns plain device tensors
        # the deferred NaN flag of the newest step is looked at before anything is written to disk (model/checkpoints.py)
        import weakref
        from model import checkpoints as _ck
        ref = weakref.ref(self)
        _ck.PRE_SAVE_HOOKS[:] = [h for h in _ck.PRE_SAVE_HOOKS if getattr(h, '_trainer', lambda: None)() is not None]
        hook = lambda: (ref() is not None and ref().flush_nan_check())
        hook._trainer = ref
        _ck.PRE_SAVE_HOOKS.append(hook)
        self.fuse_front_end = bool(cfg.get('fuse_front_end', True))   # training.fuse_front_end: False keeps the separate launches
        self.fuse_pair = bool(cfg.get('fuse_pair', True))             # training.fuse_pair: False keeps the first phase on the separate launches
        self._resize_cache, self._kinv_cache = {}, None
        # training.adam_arithmetic: WHICH of torch's two Adam arithmetics the step's updates carry out (nnr/optim.py).  "single" (default) =
        # torch's single-tensor implementation, i.e. what the plain optim.Adam objects of the reference's train.py:58,99,117,140 do (host
        # step counters, float moments); "fused" = torch's fused=True flavour (double-precision moments rounded once), the default of
        # rounds 2-4.  The two differ in last bits only -- and 800 steps of the convergence replay turn that into 0.15 dB (profiles/r05/).
        self.adam_arithmetic = str(cfg.get('adam_arithmetic', 'single'))
        if self.adam_arithmetic not in ('single', 'fused'):
            raise ValueError("training.adam_arithmetic: %r (expected 'single' or 'fused')" % self.adam_arithmetic)
        if cfg.get('fuse_optimizers', True) and self.adam_arithmetic == 'fused':   # training.fuse_optimizers: False keeps torch's default multi-kernel Adam
            for opt in (optimizer, optimizer_pose, optimizer_focal, optimizer_distortion):
                _use_fused_adam(opt)
        # training.one_launch_adam (default on): ALL of the step's Adam updates in one HIP launch (nnr.optim.MultiAdam: the same
        # optimizer objects, param_groups and state, bitwise the selected torch arithmetic) instead of 2-7 launches per optimiser
        self._multi_adam = None
        if cfg.get('fuse_optimizers', True) and cfg.get('one_launch_adam', True):
            from nnr.optim import MultiAdam
            self._multi_adam = MultiAdam([optimizer, optimizer_pose, optimizer_focal, optimizer_distortion], self.adam_arithmetic)

    # ------------------------------------------------------------------------------------------------ step
    def _groups(self):
        return [(self.pose_param_net, self.optimizer_pose), (self.focal_net, self.optimizer_focal),
                (self.distortion_net, self.optimizer_distortion)]

    def train_step(self, data, it=None, epoch=None, scheduling_start=None, render_path=None):
        if not self.model.training:        # Module.train() walks every sub-module: skip the walk when nothing changes
            self.model.train()
        self.optimizer.zero_grad()
        for net, opt in self._groups():
            if net:
                if not net.training:
                    net.train()
                opt.zero_grad()
        self._nan_with_early = bool(self.early_scalars and self.device is not None and torch.device(self.device).type == 'cuda'
                                    and not (parallel.world_size() > 1 or parallel.always_reduce()))
        try:
            loss_dict = self.compute_loss(data, it=it, epoch=epoch, scheduling_start=scheduling_start,
                                          out_render_path=render_path)
        finally:
            deferred, self._nan_with_early = self._nan_with_early, False
        loss = loss_dict['loss']
        early = self._copy_scalars_early(loss_dict) if (loss.is_cuda and self.early_scalars) else None
        if deferred and (early is None or self._nan_flag is None):      # (the early copy did not take the loss along: copy it now)
            self._check_nan(loss)
        if loss.is_cuda:      # the root gradient as a cached constant: loss.backward() would launch a ones_like fill every step
            if self._one is None or self._one.device != loss.device:
                self._one = torch.ones((), dtype=loss.dtype, device=loss.device)
                if self._one.dtype == torch.float32:
                    camera.register_unit_gradient(self._one)      # the loss head's backward then skips its `* 1`
            torch.autograd.backward(loss, grad_tensors=self._one if self._one.dtype == loss.dtype else None)
        else:
            loss.backward()
        if parallel.world_size() > 1 or parallel.always_reduce():
            nets = [self.model] + [n for n, _ in self._groups() if n]
            parallel.allreduce_gradients([p for n in nets for p in n.parameters()], loss_dict)
        if not (self._multi_adam is not None and loss.is_cuda and self._multi_adam.step()):
            self.optimizer.step()
            for net, opt in self._groups():
                if opt:
                    opt.step()
        if early is not None:
            keys, slot, event = early
            for k, i in keys:
                t = torch.Tensor._make_subclass(_EarlyScalar, loss_dict[k].detach())
                t._early = (slot, slot['generation'], i, event)
                loss_dict[k] = t
        return loss_dict

    def _copy_scalars_early(self, loss_dict):
        """Enqueue, BEFORE the backward, the copy of the step's 0-dim logged scalars to pinned host memory (see _EarlyScalar).  Not under
        data parallelism: there the logged scalars are summed over the ranks after the backward (parallel.allreduce_gradients).
        One asynchronous copy, no launch, when every scalar that is not the cached zero is a view of ONE small device buffer (the fused
        loss kernel's output: loss, loss_rgb, loss_depth, l2_mean -- the step without per-image losses); otherwise one stack launch + one
        copy.  The deferred NaN check reads the loss from the same host buffer (no second copy)."""
        if parallel.world_size() > 1 or parallel.always_reduce():
            return None
        keys = [k for k, v in loss_dict.items() if torch.is_tensor(v) and v.is_cuda and v.numel() == 1 and v.dtype == torch.float32
                and k not in ('scale', 'shift')]       # (the two distortion entries are (1,) views train.py only stores)
        if not keys:
            return None
        if self._early_host is None or self._early_host[0]['host'].numel() < max(16, len(keys)) + 1:
            self._early_host = [{'host': torch.zeros(max(16, len(keys)) + 17, dtype=torch.float32).pin_memory(), 'generation': 0} for _ in range(4)]
        slot = self._early_host[0]
        self._early_host.append(self._early_host.pop(0))      # the buffers of the last three steps stay valid for their readers
        slot['generation'] += 1
        host = slot['host']
        zero = _zero(loss_dict[keys[0]])
        live = [k for k in keys if loss_dict[k] is not zero]
        index = {}
        base = None
        if live:
            st = loss_dict[live[0]].untyped_storage()
            if all(loss_dict[k].untyped_storage().data_ptr() == st.data_ptr() for k in live) and st.nbytes() <= 64:
                base = torch.empty(0, dtype=torch.float32, device=loss_dict[live[0]].device).set_(st)      # the whole buffer, as it lies
        with torch.no_grad():
            if base is not None:
                n = base.numel()
                host[:n].copy_(base, non_blocking=True)
                for k in live:
                    index[k] = loss_dict[k].storage_offset()
                host[16] = 0.0                                 # (a host write: the slot the cached zeros answer from)
                for k in keys:
                    if k not in index:
                        index[k] = 16
            else:
                host[:len(keys)].copy_(torch.stack([loss_dict[k].detach().reshape(()) for k in keys]), non_blocking=True)
                index = {k: i for i, k in enumerate(keys)}
        event = torch.cuda.Event()
        event.record()
        if 'loss' in index:                                    # the deferred NaN check of this step looks at the same copy
            self._nan_flag = (host[index['loss']:index['loss'] + 1], event)
        return [(k, index[k]) for k in keys], slot, event

    # ------------------------------------------------------------------------------------------------ data
    @staticmethod
    def _host_index(idx):
        """The loaders hand the view index over as a one-element HOST tensor (default collate).  Used as it is -- `table[idx]` on a device
        table -- it is copied to the device by a blocking pageable copy that first waits for everything queued on the stream: with the
        per-image block on, four such lookups per step drained the launch queue behind the render kernels (the unmodified train.py ran at
        0.75 of the step's rate in its first phase, profiles/r04/e_*).  A Python int indexes without any copy."""
        if torch.is_tensor(idx) and not idx.is_cuda and idx.numel() == 1:
            return int(idx)
        return idx

    def process_data_dict(self, data):
        dev = self.device
        return (data.get('img').to(dev), data.get('img.dpt').to(dev).unsqueeze(1), data.get('img.camera_mat').to(dev),
                data.get('img.scale_mat').to(dev), self._host_index(data.get('img.idx')))

    def process_data_reference(self, data):
        dev = self.device
        return (data.get('img.ref_imgs').to(dev), data.get('img.ref_dpts').to(dev).unsqueeze(1), self._host_index(data.get('img.ref_idxs')))

    def anneal(self, start_weight, end_weight, anneal_start_epoch, anneal_epoches, current):
        if current <= anneal_start_epoch:
            return start_weight
        if current >= anneal_start_epoch + anneal_epoches:
            return end_weight
        return start_weight + (end_weight - start_weight) * (current - anneal_start_epoch) / anneal_epoches

    def _camera_from_focal(self, device):
        fxfy = self.focal_net(0)
        k = torch.zeros(4, 4, device=device)
        k[0, 0], k[1, 1], k[2, 2], k[3, 3] = fxfy[0], -fxfy[1], -1.0, 1.0
        return fxfy, k.unsqueeze(0)

    # ------------------------------------------------------------------------------------------------ loss
    def compute_loss(self, data, eval_mode=False, it=None, epoch=None, scheduling_start=None, out_render_path=None):
        weights = {k: self.anneal(getattr(self, k)[0], getattr(self, k)[1], scheduling_start, self.annealing_epochs, epoch)
                   for k in _WEIGHT_KEYS}
        rgb_loss_type = 'l1' if epoch < self.annealing_epochs + scheduling_start else 'l2'
        render_model = weights['rgb_weight'] != 0.0 or weights['depth_weight'] != 0.0
        use_ref_imgs = weights['pc_weight'] != 0.0 or weights['rgb_s_weight'] != 0.0
        world, rank = parallel.world_size(), parallel.rank()

        img, depth_input, camera_mat_gt, scale_mat, img_idx = self.process_data_dict(data)
        device = self.device
        batch_size, _, h, w = img.shape
        h_depth, w_depth = depth_input.shape[-2:]
        kwargs = {'t_list': self.pose_param_net.get_t(), 'weights': weights, 'rgb_loss_type': rgb_loss_type}

        num_cams = self.pose_param_net.num_cams
        n_points = self.n_training_points
        # The fused front end (nnr.camera.step_rays): pose -> world_mat, the frame's depth distortion, pixel coordinates, gathered
        # depths, colour targets and rays in ONE launch each way instead of ~9 forward and ~14 backward ones.  Same arithmetic in the
        # same order; taken whenever nothing needs the intermediate tensors (the per-image losses read the whole distorted map, a
        # learnable focal rebuilds K inside the graph, an initial pose multiplies onto c2w).
        fusable = (img.is_cuda and render_model and self.distortion_net is not None and not self.optimizer_focal
                   and getattr(self.pose_param_net, 'init_c2w', None) is None and self.rendering_technique == 'nope_nerf'
                   and getattr(self.pose_param_net, 'r', None) is not None and self.fuse_front_end)
        # Round 5: the first training phase takes it too.  The frame PAIR of the per-image losses rides along in the same two launches
        # (reference pose, inverses, relative transform, the reference frame's distortion: mats[34:55]) and the per-image block applies the
        # distortions to its sampled depths itself (nnr.aux, `aff`) -- no distorted full-size depth map, no se3_exp / inverse / matmul /
        # indexing launches and none of their autograd.  Not for the steps that dump the re-projection images (they want the intermediates).
        dump_reproj = (use_ref_imgs and weights['rgb_s_weight'] != 0.0 and (it % self.vis_reprojection_every) == 0 and out_render_path is not None)
        fused_pair = bool(fusable and use_ref_imgs and batch_size == 1 and not dump_reproj and self.fuse_pair)
        fused_front = bool(fusable and (not use_ref_imgs or fused_pair))
        scale_input = shift_input = None
        depth_affine = None
        rays = None
        pair = None
        if fused_front:
            if fused_pair:
                ref_img, depth_ref_raw, ref_idx = self.process_data_reference(data)
            ray_idx = sampling.randperm_prefix(h * w, n_points, device)   # == torch.randperm(h * w, device=device)[:n_points]
            n_total = ray_idx.shape[0]
            lo, hi = parallel.shard_bounds(n_total, rank, world)
            ray_loc = ray_idx[lo:hi]
            rcfg = self.model.renderer.cfg
            *rays, rgb_gt, p, mats = camera.step_rays(
                self.pose_param_net.r, self.pose_param_net.t, self.distortion_net.global_scales, self.distortion_net.global_shifts,
                depth_input, img, ray_loc, camera_mat_gt, scale_mat, cam=int(img_idx), h=h, w=w,
                fix_last_scale=bool(self.distortion_net.fix_scaleN), shift_first=bool(self.shift_first),
                normalise=bool(rcfg['normalise_ray']), use_dir=bool(rcfg['use_ray_dir']),
                ref=int(ref_idx) if fused_pair else -1, detach_ref=bool(self.detach_ref_img))
            if fused_pair:
                pair = (mats, ref_img, depth_ref_raw, int(ref_idx))
            rgb_gt = rgb_gt.unsqueeze(0)
            world_mat = mats[16:32].view(1, 4, 4)
            scale_input, shift_input = mats[32:33], mats[33:34]
            depth_affine = (scale_input, shift_input, bool(self.shift_first))
            camera_mat = camera_mat_gt
        else:
            world_mat = self._inverse(self.pose_param_net(img_idx)).unsqueeze(0)
            if self.distortion_net is not None:
                scale_input, shift_input = self.distortion_net(img_idx)
                if not use_ref_imgs:
                    # only the R picked pixels of the distorted map are ever used: distort those (model/network.py), not 518 400
                    depth_affine = (scale_input, shift_input, bool(self.shift_first))
                else:   # the per-image losses read the whole distorted map
                    depth_input = (depth_input + shift_input) * scale_input if self.shift_first \
                        else depth_input * scale_input + shift_input
            if self.optimizer_focal:
                fxfy, camera_mat = self._camera_from_focal(device)
            else:
                camera_mat = camera_mat_gt

            # pixel pick: the permutation is drawn exactly as the reference does (training.py:257), so indices are bit-identical
            ray_idx = sampling.randperm_prefix(h * w, n_points, device)   # == torch.randperm(h * w, device=device)[:n_points]
            n_total = ray_idx.shape[0]
            lo, hi = parallel.shard_bounds(n_total, rank, world)
            ray_loc = ray_idx[lo:hi]
            rgb_gt = img.view(batch_size, 3, h * w).permute(0, 2, 1)[:, ray_loc]
            if ray_loc.is_cuda:
                p = camera.pixels_from_index(ray_loc, h, w)                                # == arange_pixels()[1][:, idx], one launch
            else:
                xs = (ray_loc % w).float()
                ys = torch.div(ray_loc, w, rounding_mode='floor').float()
                p = torch.stack([2.0 * xs / (w - 1) - 1.0, 2.0 * ys / (h - 1) - 1.0], dim=-1).unsqueeze(0)

        rendered_rgb = rendered_depth = gt_depth = None
        if render_model:
            renderer = self.model.renderer
            renderer.jitter_window = (lo, n_total) if world > 1 else None
            renderer.normal_window = None
            if world > 1 and renderer.cfg.get('normal_loss'):
                # the normal term perturbs one surface point per VALID ray: the shard needs the validity of all rays to find its rows
                with torch.no_grad():
                    d_all = self._gather_depth_all(depth_input, ray_idx, (h, w), depth_affine)
                renderer.normal_window = (torch.isfinite(d_all) & (d_all != 0), lo)
            out = self.model(p, ray_loc, camera_mat, world_mat, scale_mat, self.rendering_technique, it=it,
                             eval_mode=eval_mode, depth_img=depth_input, img_size=(h, w), depth_affine=depth_affine,
                             **({'rays': tuple(rays)} if rays is not None else {}))
            renderer.jitter_window = renderer.normal_window = None
            rendered_rgb = out['rgb']
            if not (rendered_rgb.is_cuda and self.loss.depth_loss_type == 'l1'):
                rendered_depth, gt_depth = out['depth_pred'], out['depth_gt']     # masked views, only where the torch loss needs them
                if self.detach_gt_depth:
                    gt_depth = gt_depth.detach()

        if use_ref_imgs and pair is not None:
            self._pair_terms(kwargs, pair, img, depth_input, camera_mat, img_idx, num_cams, h_depth, w_depth, weights)
        elif use_ref_imgs:
            self._reference_terms(kwargs, data, img, depth_input, camera_mat, world_mat, scale_input, img_idx, num_cams,
                                  h_depth, w_depth, weights, it, out_render_path)

        loss_dict = self._total_loss(out if render_model else None, rendered_rgb, rgb_gt, rendered_depth, gt_depth, n_total,
                                     ray_idx, depth_input, (h, w), kwargs, world, depth_affine)
        if self.optimizer_focal:
            loss_dict['focalx'] = fxfy[0] / camera_mat_gt[0, 0, 0]
            loss_dict['focaly'] = fxfy[1] / camera_mat_gt[0, 1, 1]
        loss_dict['scale'] = scale_input
        loss_dict['shift'] = shift_input
        return loss_dict

    @staticmethod
    def _inverse(m):
        return camera.inverse4(m) if m.is_cuda else torch.inverse(m)

    def flush_nan_check(self):
        """Wait for the NaN flag of the newest training step and raise if it is set.  Called before every checkpoint save
        (CheckpointIO.save -> PRE_SAVE_HOOKS) and by training loops at their end: the per-step check is one step late by design."""
        if self._nan_flag is not None:
            host, ev = self._nan_flag
            self._nan_flag = None
            ev.synchronize()
            if math.isnan(float(host)):
                raise FloatingPointError('NaN loss in the last training step')

    def _check_nan(self, loss):
        """The reference stops on a NaN loss (losses.py:204-205).  Reading the flag synchronously (or even enqueueing its
        device->host copy late) would drain the launch queue every step, so the copy of step i is enqueued -- into pinned
        memory, with an event -- right behind its loss kernel and only looked at during step i+1."""
        if self._nan_flag is not None:
            host, ev = self._nan_flag
            self._nan_flag = None
            ev.synchronize()
            if math.isnan(float(host)):
                raise FloatingPointError('NaN loss in the previous training step')
        if getattr(self, '_nan_with_early', False) and loss.is_cuda:
            return      # train_step copies the step's logged scalars to the host in one go (_copy_scalars_early) and points the flag at it
        if loss.is_cuda:
            if self._nan_host is None:
                self._nan_host = [torch.empty((), dtype=torch.float32).pin_memory() for _ in range(2)]
            host = self._nan_host[0]
            self._nan_host.reverse()
            host.copy_(loss.detach(), non_blocking=True)      # the loss VALUE; isnan is evaluated on the host (no isnan launch)
            ev = torch.cuda.Event()
            ev.record()
            self._nan_flag = (host, ev)
        elif bool(torch.isnan(loss)):
            raise FloatingPointError('NaN loss')

    def _total_loss(self, out, rgb, rgb_gt, depth_pred, depth_gt, n_total, ray_idx, depth_input, img_size, kwargs, world,
                    depth_affine=None):
        """rgb + depth heads (fused HIP kernel on the GPU) + per-image terms.  Under data parallelism each rank holds a
        shard of the rays: per-ray terms are divided by the GLOBAL ray / valid-depth counts and per-image terms by
        world_size, so that the SUM over ranks equals the single-process loss."""
        w = kwargs['weights']
        m_total = -1.0
        if world > 1:
            h_img, w_img = img_size
            with torch.no_grad():   # global count of rays with a usable mono depth (finite, non-zero): no collective needed
                d_all = self._gather_depth_all(depth_input, ray_idx, (h_img, w_img), depth_affine)
                m_total = (torch.isfinite(d_all) & (d_all != 0)).sum().float()   # stays on the device: no sync
        fused = (out is not None and rgb.is_cuda and self.loss.depth_loss_type == 'l1' and 'dist_dense' in out)
        if not fused and world == 1:
            loss_dict = self.loss(rgb, rgb_gt, depth_pred, depth_gt, **kwargs)
            return loss_dict
        aux, parts = self.loss.aux_terms(rgb_gt, **kwargs)
        if world > 1:
            # loss_pc / loss_rgb_s arrive as this rank's SHARE (sums over its shard of the source points, _reference_terms); the
            # remaining per-image terms (trajectory smoothness, depth consistency) are O(cameras) and replicated: weight 1/W
            zero = _zero(rgb_gt)
            sharded = ('loss_pc', 'loss_rgb_s') if 'point_shard' in kwargs else ()
            parts = {k: (v if (v is zero or k in sharded) else v / world) for k, v in parts.items()}   # inactive terms stay the cached 0
            aux = None
            for wk, pk in (('weight_dist_1st_loss', 'loss_dist_1st'), ('weight_dist_2nd_loss', 'loss_dist_2nd'),
                           ('pc_weight', 'loss_pc'), ('rgb_s_weight', 'loss_rgb_s'),
                           ('depth_consistency_weight', 'loss_depth_consistency')):
                if w[wk] != 0.0:
                    term = w[wk] * parts[pk]
                    aux = term if aux is None else aux + term
        loss_dict = dict(parts)
        zero = _zero(rgb_gt)
        if out is None:
            lmain, lrgb, ldep, l2 = zero, zero, zero, zero
        elif fused:
            lmain, a4 = camera.render_loss(rgb, rgb_gt, out['dist_dense'], out['d_gt_dense'], out['mask'], r_total=n_total,
                                           m_total=m_total, w_rgb=w['rgb_weight'], w_depth=w['depth_weight'],
                                           rgb_l2=(kwargs['rgb_loss_type'] == 'l2'), ndc=out['ndc'],
                                           detach_gt=self.detach_gt_depth)
            lrgb, ldep, l2 = a4[0], a4[1], a4[2]
        else:
            invariant = None
            if self.loss.depth_loss_type != 'l1' and w['depth_weight'] != 0.0:
                # 'invariant' normalises by the median / mean deviation over ALL valid rays of the step (losses.py:42-46): gather the
                # dense per-ray values, keep this rank's slice live, evaluate the loss on the full vectors.  Its gradient with
                # respect to the local rays is then exact and the SUM over ranks is the single-process gradient; the reported value
                # is divided by the world size so that the summed log equals the loss (value / W, gradient x 1).
                pred_d, gt_d, valid = out['dist_dense'], out['d_gt_dense'], out['mask']
                if out['ndc']:
                    gt_d = 1 - 1 / gt_d
                if self.detach_gt_depth:
                    gt_d = gt_d.detach()
                pred_all, gt_all = parallel.gather_rays(pred_d, n_total), parallel.gather_rays(gt_d, n_total)
                valid_all = parallel.gather_rays(valid.to(pred_d.dtype), n_total) > 0.5
                full = self.loss.depth_loss_dpt(pred_all[valid_all], gt_all[valid_all])
                invariant = full.detach() / world + (full - full.detach())
            diff = rgb - rgb_gt
            lrgb = (diff.abs().sum() if kwargs['rgb_loss_type'] == 'l1' else (diff * diff).sum()) / float(n_total) \
                if w['rgb_weight'] != 0.0 else zero          # a term with weight 0 is reported as 0 (losses.py:164-171)
            if invariant is not None:
                ldep = invariant
            else:
                ldep = (depth_pred - depth_gt).abs().sum() / torch.clamp(torch.as_tensor(m_total, device=rgb.device), min=1.0) \
                    if w['depth_weight'] != 0.0 else zero
            l2 = (diff * diff).sum() / float(3 * n_total) if (w['rgb_weight'] != 0.0 or w['depth_weight'] != 0.0) else zero
            lmain = w['rgb_weight'] * lrgb + w['depth_weight'] * ldep
        loss_dict.update(loss_rgb=lrgb, loss_depth=ldep, l2_mean=l2, loss=lmain if aux is None else lmain + aux)
        self._check_nan(loss_dict['loss'])
        return loss_dict

    @staticmethod
    def _gather_depth_all(depth_input, ray_idx, img_size, depth_affine):
        """The (distorted) mono depth of EVERY ray of the step, flat (n_total,) -- what a data-parallel rank needs to know about
        the rays it does not render: their validity."""
        h_img, w_img = img_size
        if depth_input.is_cuda and depth_affine is not None:
            return camera.depth_gather_affine(depth_input, ray_idx, depth_affine[0], depth_affine[1], h_img, w_img, depth_affine[2]).reshape(-1)
        if depth_input.is_cuda:
            return camera.depth_gather(depth_input, ray_idx, h_img, w_img).reshape(-1)          # one launch
        from model.network import nearest_source_index
        hd, wd = depth_input.shape[-2:]
        ys = nearest_source_index(torch.div(ray_idx, w_img, rounding_mode='floor'), h_img, hd)
        xs = nearest_source_index(ray_idx % w_img, w_img, wd)
        d_all = depth_input[0, 0][ys, xs]
        if depth_affine is not None:
            d_all = (d_all + depth_affine[1]) * depth_affine[0] if depth_affine[2] else d_all * depth_affine[0] + depth_affine[1]
        return d_all.reshape(-1)

    def _pair_terms(self, kwargs, pair, img, depth_raw, camera_mat, img_idx, num_cams, h_depth, w_depth, weights):
        """The per-image block behind the fused front end (round 5): everything it needs beyond the frames themselves -- relative transform,
        the two distortions in cloud order, scale2 -- is in `mats` (nnr.camera.step_rays with ref >= 0); the depth maps stay RAW.
        Same losses as _reference_terms (reference training.py:280-365; pinned by the reference goldens of tests/test_aux_terms.py)."""
        from nnr import aux as nnr_aux
        mats, ref_img, depth_ref_raw, ref_idx = pair
        swap = not (img_idx < (num_cams - 1))               # the last camera takes the roles the other way round (training.py:301-311)
        raw1, raw2 = (depth_ref_raw, depth_raw) if swap else (depth_raw, depth_ref_raw)
        img1, img2 = (ref_img, img) if swap else (img, ref_img)
        res = (int(h_depth / self.pc_ratio), int(w_depth / self.pc_ratio))
        world = parallel.world_size()
        shard = parallel.shard_bounds(res[0] * res[1], parallel.rank(), world) if world > 1 else None
        if shard is not None:
            kwargs['point_shard'] = shard
        rgb_s = weights['rgb_s_weight'] != 0.0
        i1 = self._resized(img1, res) if rgb_s else None
        i2 = self._resized(img2, res) if rgb_s else None
        # the weighted sum of the two terms comes from the finishing kernel when nothing else enters the per-image sum (one GPU; no trajectory /
        # depth-consistency term): the two multiplies, the add and their backward are five launches
        only_pair = world == 1 and all(weights[k] == 0.0 for k in ('weight_dist_1st_loss', 'weight_dist_2nd_loss', 'depth_consistency_weight'))
        terms = nnr_aux.aux_terms(raw1, raw2, None, None, i1, i2, camera_mat, self._constant_inverse(camera_mat),
                                  res, self.nearest_limit, rgb_s=rgb_s, pc=weights['pc_weight'] != 0.0, scale_pcs=bool(self.scale_pcs),
                                  detach_rgbs_scale=self.detach_rgbs_scale, ssim=self.loss.cfg['with_ssim'] == True,  # noqa: E712 (YAML)
                                  shard=shard or (0, 0), shift_first=bool(self.shift_first), mats=mats,
                                  weights=(weights['pc_weight'], weights['rgb_s_weight']) if only_pair else None)
        kwargs.update(fused_aux=(terms[0], terms[1]), sample_resolution=res)
        if only_pair:
            kwargs['fused_aux_sum'] = terms[3]

    def _resized(self, frame, res):
        """F.interpolate(frame, res, mode='bilinear') -- for a frame that is a view of a RESIDENT scene tensor (dataloading.ResidentLoader: the
        storage lives for the whole run and is never rewritten) the result is computed once per (frame, resolution) and kept: two launches
        per step less.  Any other frame is resized every time."""
        base = getattr(frame, '_base', None)
        if base is None or not getattr(base, '_nnr_resident', False):
            return F.interpolate(frame, res, mode='bilinear')
        key = (frame.data_ptr(), tuple(frame.shape), tuple(res))
        hit = self._resize_cache.get(key)
        if hit is None:
            if len(self._resize_cache) >= 4096:
                self._resize_cache.clear()
            hit = self._resize_cache[key] = F.interpolate(frame, res, mode='bilinear').detach()
        return hit

    def _constant_inverse(self, m):
        """inverse of the camera matrix: for the matrix of a RESIDENT scene (dataloading.ResidentLoader hands the same tensor over every step,
        and nothing writes to it) computed once; any other tensor is inverted every time (one launch)."""
        if m.requires_grad or not getattr(m, '_nnr_resident', False):
            return self._inverse(m)
        key = (m.data_ptr(), m._version)
        if self._kinv_cache is None or self._kinv_cache[0] != key:
            self._kinv_cache = (key, self._inverse(m).detach())
        return self._kinv_cache[1]

    def _reference_terms(self, kwargs, data, img, depth_input, camera_mat, world_mat, scale_input, img_idx, num_cams,
                         h_depth, w_depth, weights, it, out_render_path):
        """Inputs of the point-cloud and surface-reprojection losses between this frame and its reference frame
        (reference training.py:280-365)."""
        device = self.device
        nl = self.nearest_limit
        ref_img, depth_ref, ref_idx = self.process_data_reference(data)
        c2w_ref = self.pose_param_net(ref_idx)
        scale_ref = shift_ref = None
        if self.distortion_net is not None:
            scale_ref, shift_ref = self.distortion_net(ref_idx)
            depth_ref = scale_ref * (depth_ref + shift_ref) if self.shift_first else scale_ref * depth_ref + shift_ref
        if self.detach_ref_img:
            c2w_ref, depth_ref = c2w_ref.detach(), depth_ref.detach()
            if scale_ref is not None:
                scale_ref, shift_ref = scale_ref.detach(), shift_ref.detach()
        ref_rt = self._inverse(c2w_ref).unsqueeze(0)
        if img_idx < (num_cams - 1):
            d1, d2, img1, img2 = depth_input, depth_ref, img, ref_img
            rel = ref_rt @ self._inverse(world_mat)
            scale2 = scale_ref
        else:
            d1, d2, img1, img2 = depth_ref, depth_input, ref_img, img
            rel = world_mat @ self._inverse(ref_rt)
            scale2 = scale_input
        r_rel, t_rel = rel[:, :3, :3], rel[:, :3, 3]

        res = (int(h_depth / self.pc_ratio), int(w_depth / self.pc_ratio))
        world = parallel.world_size()
        # data parallelism: both losses are means over source points -- rank k owns the points [lo, hi) of the sampling grid
        shard = parallel.shard_bounds(res[0] * res[1], parallel.rank(), world) if world > 1 else None
        if shard is not None:
            kwargs['point_shard'] = shard
        dump = (weights['rgb_s_weight'] != 0.0 and (it % self.vis_reprojection_every) == 0 and out_render_path is not None)
        if d1.is_cuda and img.shape[0] == 1 and not dump:
            # one fused forward / backward pair (nnr/aux.py) instead of ~290 small launches; same inputs, same losses
            from nnr import aux as nnr_aux
            rgb_s = weights['rgb_s_weight'] != 0.0
            i1 = F.interpolate(img1, res, mode='bilinear') if rgb_s else None
            i2 = F.interpolate(img2, res, mode='bilinear') if rgb_s else None
            l_pc, l_rgbs, _ = nnr_aux.aux_terms(d1, d2, rel, scale2, i1, i2, camera_mat, self._inverse(camera_mat), res, nl,
                                                rgb_s=rgb_s, pc=weights['pc_weight'] != 0.0, scale_pcs=bool(self.scale_pcs),
                                                detach_rgbs_scale=self.detach_rgbs_scale, ssim=self.loss.cfg['with_ssim'] == True,  # noqa: E712 (YAML)
                                                shard=shard or (0, 0))
            kwargs.update(fused_aux=(l_pc, l_rgbs), sample_resolution=res)
            return
        pixel_locations, p_pc = arange_pixels(resolution=res, device=device)
        d1 = F.interpolate(d1, res, mode='nearest')
        d2 = F.interpolate(d2, res, mode='nearest')
        d1[d1 < nl] = nl
        d2[d2 < nl] = nl
        pc1 = transform_to_world(p_pc, d1.view(1, -1, 1), camera_mat)
        pc2 = transform_to_world(p_pc, d2.view(1, -1, 1), camera_mat)

        if weights['rgb_s_weight'] != 0.0:
            img1 = F.interpolate(img1, res, mode='bilinear')
            img2 = F.interpolate(img2, res, mode='bilinear')
            rgb_pc1 = get_tensor_values(img1, p_pc, mode='bilinear', scale=False, detach=False, detach_p=False,
                                        align_corners=True)
            src = pc1.detach().clone() if self.detach_rgbs_scale else pc1
            pc1_rot = src @ r_rel.transpose(1, 2) + t_rel
            behind = (-pc1_rot[:, :, 2:] < nl).expand_as(pc1_rot)
            pc1_rot[behind] = nl
            p_reproj, valid = project_to_cam(pc1_rot, camera_mat, device)
            rgb_proj = get_tensor_values(img2, p_reproj, mode='bilinear', scale=False, detach=False, detach_p=False,
                                         align_corners=True)
            shape = (img.shape[0], res[0], res[1])
            kwargs['rgb_pc1'] = rgb_pc1.view(*shape, 3)
            kwargs['rgb_pc1_proj'] = rgb_proj.view(*shape, 3)
            kwargs['valid_points'] = valid.view(*shape, 1)
            if dump:
                for tag, t in (('img1', kwargs['rgb_pc1']), ('img2', kwargs['rgb_pc1_proj'])):
                    _save_png((t[0] * 255).detach().cpu().numpy().astype(np.uint8),
                              os.path.join(out_render_path, '%d_%04d_%s.png' % (it, img_idx, tag)))
        pc1 = pc1 @ r_rel.transpose(1, 2) + t_rel          # transform before scaling
        if self.scale_pcs:
            pc1, pc2 = pc1 / scale2, pc2 / scale2
        kwargs.update(X=pc1, Y=pc2, sample_resolution=res, p_2d=pixel_locations)

    # ------------------------------------------------------------------------------------------------ visual dumps
    def render_visdata(self, data, resolution, it, out_render_path):
        img, dpt, camera_mat, scale_mat, img_idx = self.process_data_dict(data)
        h, w = resolution
        world_mat = self._inverse(self.pose_param_net(img_idx)).unsqueeze(0)
        if self.optimizer_focal:
            _, camera_mat = self._camera_from_focal(self.device)
        p_idx = torch.arange(h * w, device=self.device)
        pixels = arange_pixels(resolution=(h, w), device=self.device)[1]
        with torch.no_grad():
            rgb, depth = [], []
            for pix_i, idx_i in zip(torch.split(pixels, 1024, dim=1), torch.split(p_idx, 1024, dim=0)):
                out = self.model(pix_i, idx_i, camera_mat, world_mat, scale_mat, self.rendering_technique, add_noise=False,
                                 eval_mode=True, it=it, depth_img=dpt, img_size=(h, w))
                rgb.append(out['rgb'])
                depth.append(out['depth_pred'])
            rgb = torch.cat(rgb, dim=1).view(h, w, 3).cpu().numpy()
            depth = torch.cat(depth, dim=0).view(h, w).cpu().numpy()
        img_out = (rgb * 255).astype(np.uint8)
        if not parallel.is_writer():      # data parallel: the frames are identical on every rank, rank 0 writes them
            return img_out
        depth_u8 = np.clip(255.0 / depth.max() * (depth - depth.min()), 0, 255).astype(np.uint8)
        _save_png(depth_u8, os.path.join(out_render_path, '%04d_depth.png' % img_idx))
        Image.fromarray(img_out).convert("RGB").save(os.path.join(out_render_path, '%04d_img.png' % img_idx))
        if self.vis_geo and not self._warned_geo:
            logger_py.warning("training.vis_geo: the phong geometry visualiser is outside the HIP hot path; skipping *_geo.png")
            self._warned_geo = True
        return img_out
