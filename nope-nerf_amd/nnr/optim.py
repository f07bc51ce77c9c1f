"""One HIP launch for all Adam updates of a training step (nnr_adam_step, csrc/nnr_optim.hip).

The reference steps up to four `torch.optim.Adam` instances per iteration (model/training.py:90-96).  `MultiAdam` drives the SAME
optimizer objects -- their `param_groups` (so LR schedulers keep working) and their `state` (so `state_dict()` / checkpoints keep
torch's layout: `step`, `exp_avg`, `exp_avg_sq` per parameter) -- but performs the update of all of them in one kernel whose
arithmetic is, operation by operation, one of torch's own two (tests/test_gpu_optim.py holds both bitwise):
  "single" (the default)  torch's single-tensor implementation, `Adam(foreach=False, fused=False).step()` -- what the plain
                          `optim.Adam(...)` objects of the reference's train.py:58,99,117,140 run; the step counters stay host tensors as there,
                          bias corrections and step size are computed on the host in python doubles exactly as torch/optim/adam.py does;
  "fused"                 `Adam(fused=True).step()`: double-precision moments rounded once, device step counters.
The two differ in the last bits of every update; over 800 steps of the convergence replay that was the difference between landing inside
and 0.15 dB below the reference's own run-to-run spread (profiles/r04/b_conv_envelope_hip.txt, profiles/r05/).  Anything it does not cover (other optimizer classes, weight decay, amsgrad, maximize, non-fp32 or CPU
parameters, more than NNR_ADAM_MAX_TENSORS tensors, non-contiguous gradients) makes `step()` return False and the caller steps the
optimizers itself."""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List, Sequence

import torch

from . import lib as L

MAX_TENSORS = 40


class AdamTable(C.Structure):       # nnr_adam_table (include/nnr.h)
    _fields_ = [("param", C.c_void_p * MAX_TENSORS), ("grad", C.c_void_p * MAX_TENSORS), ("exp_avg", C.c_void_p * MAX_TENSORS),
                ("exp_avg_sq", C.c_void_p * MAX_TENSORS), ("step_in", C.c_void_p * MAX_TENSORS), ("step_out", C.c_void_p * MAX_TENSORS),
                ("lr", C.c_double * MAX_TENSORS), ("beta1", C.c_double * MAX_TENSORS), ("beta2", C.c_double * MAX_TENSORS),
                ("eps", C.c_double * MAX_TENSORS), ("numel", C.c_int64 * MAX_TENSORS), ("block_first", C.c_int32 * (MAX_TENSORS + 1)),
                ("n_tensors", C.c_int32), ("flavour", C.c_int32), ("reserved", C.c_int32), ("bc2_sqrt", C.c_double * MAX_TENSORS)]


FLAVOURS = {"fused": 0, "single": 1}      # NNR_ADAM_FUSED / NNR_ADAM_SINGLE (include/nnr.h)


def _plain_adam(opt) -> bool:
    if type(opt) is not torch.optim.Adam:
        return False
    for g in opt.param_groups:
        if g.get('amsgrad') or g.get('maximize') or g.get('capturable') or g.get('differentiable') or g.get('weight_decay', 0) != 0:
            return False
        if torch.is_tensor(g['lr']):
            return False
        for p in g['params']:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                return False
    return True


class MultiAdam:
    def __init__(self, optimizers: Sequence[torch.optim.Optimizer], arithmetic: str = "single"):
        if arithmetic not in FLAVOURS:
            raise ValueError("adam arithmetic %r: expected 'single' or 'fused'" % (arithmetic,))
        self.arithmetic = arithmetic
        self.optimizers: List[torch.optim.Optimizer] = [o for o in optimizers if o is not None]
        self._table = AdamTable()
        self._steps = None        # two (MAX_TENSORS,) float arrays: the counters ping-pong between them
        self._flip = 0
        self._slot = {}           # id(parameter) -> its fixed index into the counter arrays
        self._free = []           # counter slots of parameters that no longer exist
        self._alive = {}          # id(parameter) -> weak reference: a recycled id must not inherit a dead parameter's counter slot

    def usable(self) -> bool:
        if not self.optimizers or not all(_plain_adam(o) for o in self.optimizers):
            return False
        for k in [k for k, r in self._alive.items() if r() is None]:      # parameters that were re-created: free their slots
            self._free.append(self._slot.pop(k))
            del self._alive[k]
        n = len({id(p) for o in self.optimizers for g in o.param_groups for p in g['params']} | set(self._slot))
        devs = {p.device for o in self.optimizers for g in o.param_groups for p in g['params']}
        return 0 < n <= MAX_TENSORS and len(devs) == 1

    def step(self) -> bool:
        """One launch for every parameter that has a gradient.  False (nothing done) when a precondition does not hold."""
        if not self.usable():
            return False
        t = self._table
        entries = []
        for opt in self.optimizers:
            for g in opt.param_groups:
                b1, b2 = g['betas']
                for p in g['params']:
                    if p.grad is None:
                        continue
                    gr = p.grad
                    if not (gr.is_cuda and gr.dtype == torch.float32 and gr.is_contiguous()) or gr.is_sparse:
                        return False
                    entries.append((opt, p, gr, float(g['lr']), float(b1), float(b2), float(g['eps'])))
        if not entries:
            return True
        dev = entries[0][1].device
        if self._steps is None or self._steps[0].device != dev:
            self._steps = [torch.zeros(MAX_TENSORS, dtype=torch.float32, device=dev) for _ in range(2)]
        src, dst = self._steps[self._flip], self._steps[1 - self._flip]
        blocks = 0
        rebind = []
        host_steps = []      # single-tensor arithmetic: the host-side step counters this launch advances
        if self.arithmetic == "single":
            # a state restored from a round-2..4 checkpoint carries fused = True in its param_groups (what _use_fused_adam stored there): should this
            # class ever hand the step back to torch (a non-contiguous gradient), torch's FUSED path would meet the host counters below -- the
            # groups say what the arithmetic is
            for opt in self.optimizers:
                for g in opt.param_groups:
                    if g.get('fused'):
                        g['fused'] = False
                    g['foreach'] = False
        for i, (opt, p, gr, lr, b1, b2, eps) in enumerate(entries):
            if id(p) not in self._slot:      # a parameter keeps its counter slot for life (<= MAX_TENSORS: usable())
                self._slot[id(p)] = self._free.pop() if self._free else len(self._slot)
                self._alive[id(p)] = weakref.ref(p)
            k = self._slot[id(p)]
            st = opt.state[p]
            single = self.arithmetic == "single"
            if len(st) == 0:        # torch's lazy state initialisation (Adam._init_group): a float32 counter, on the host for the
                                    # single-tensor implementation, beside the parameter for the fused one
                st['step'] = torch.zeros((), dtype=torch.float32, device="cpu" if single else dev)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            step = st['step']
            if single:
                # torch/optim/adam.py::_single_tensor_adam: step_t += 1; step = _get_value(step_t); bias_correction1 = 1 - beta1 ** step;
                # bias_correction2 = 1 - beta2 ** step; step_size = lr / bias_correction1; bias_correction2_sqrt = bias_correction2 ** 0.5
                if not (torch.is_tensor(step) and not step.is_cuda):      # restored from / handed over by the fused flavour: one host read
                    step = torch.as_tensor(float(step), dtype=torch.float32)
                    st['step'] = step
                sv = float(step) + 1.0                      # (the host counter itself advances only after the launch has succeeded)
                host_steps.append(step)
                lr = -(lr / (1 - b1 ** sv))
                t.bc2_sqrt[i] = (1 - b2 ** sv) ** 0.5
            else:
                if not (torch.is_tensor(step) and step.is_cuda and step.dtype == torch.float32):
                    step = torch.as_tensor(float(step), dtype=torch.float32, device=dev)
                if step.data_ptr() != src[k].data_ptr():      # a counter not in this step's source array yet (first step, restored
                    src[k].copy_(step.reshape(()))            # state, a parameter that sat out the previous step)
                rebind.append((st, dst[k]))
                t.bc2_sqrt[i] = 1.0
            t.param[i], t.grad[i] = p.data_ptr(), gr.data_ptr()
            t.exp_avg[i], t.exp_avg_sq[i] = st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr()
            t.step_in[i], t.step_out[i] = src[k].data_ptr(), dst[k].data_ptr()
            t.lr[i], t.beta1[i], t.beta2[i], t.eps[i] = lr, b1, b2, eps
            t.numel[i] = p.numel()
            t.block_first[i] = blocks
            blocks += (p.numel() + 1023) // 1024
        t.block_first[len(entries)] = blocks
        t.n_tensors = len(entries)
        t.flavour = FLAVOURS[self.arithmetic]
        L.check(L.load().nnr_adam_step(C.byref(t), L.stream()), "nnr_adam_step")
        for step in host_steps:
            step += 1
        for st, counter in rebind:          # only after a successful launch: torch's state keeps pointing at the CURRENT counter
            st['step'] = counter
        self._flip = 1 - self._flip
        for opt in self.optimizers:
            opt._opt_called = True          # what the LR schedulers' "step() before optimizer.step()" warning looks at
        from . import ops
        ops.invalidate_packed_weights()     # the parameters changed behind the optimizers' post-step hooks
        return True
