"""torch.randperm(n, device='cuda')[:r] without sorting all n keys -- the pixel pick of reference model/training.py:257.

torch draws n random int64 keys, sorts (key, index) and re-shuffles runs of equal keys; the first r indices of the result
are the r smallest keys in (key, index) order (nnr_randperm.hip).  This module draws the keys with the very call torch's
randperm makes, advances the generator by what randperm's duplicate pass consumes, and hands both to the kernel, so the
returned indices AND the generator state afterwards are the ones torch.randperm would leave.  That equivalence rests on
torch internals, so it is verified against torch.randperm itself on the first calls in every process (one host sync
each); on any mismatch -- e.g. a torch release that changes its algorithm -- the module permanently falls back to
torch.randperm."""
import math

import torch

from . import lib as L

_CHECKS = 3          # first calls (per process) verified against torch.randperm
_state = {"checked": 0, "enabled": True}
_scratch = {}


def _key_bits(n: int) -> int:
    """The number of key bits torch's randperm sorts by (ATen/native/cuda/Randperm.cu, note [Algorithm of randperm])."""
    log_threshold_12 = math.log(0.9) * 12
    nd = float(n)
    return min(64, int(math.ceil(math.log2(nd - (6 * nd * nd + 1) / log_threshold_12))))


def _capacity(r: int) -> int:
    """Candidate-buffer size of the kernel for a pick of r (mirror of randperm_capacity in nnr_randperm.hip; equality with
    nnr_randperm_scratch_bytes is asserted in tests/test_host_logic.py): the threshold is set for r + 20 sqrt(r) + 64 expected
    candidates and the buffer leaves 40 sigma above that.  0 = beyond the largest buffer."""
    expect = r + 20.0 * math.sqrt(r) + 64.0
    hi = expect + 40.0 * math.sqrt(expect)
    return next((c for c in (4096, 16384, 65536) if hi <= c), 0)


def _fast(n: int, r: int, device) -> torch.Tensor:
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    keys = torch.empty(n, dtype=torch.int64, device=device).random_(-2 ** 63, 2 ** 63 - 1)   # the call randperm makes (bits > 32)
    seed, offset = gen.initial_seed(), gen.get_offset()
    gen.set_offset(offset + (n + 3) // 4 * 4)        # philox_cuda_state(n) of randperm_handle_duplicate_keys
    out = torch.empty(r, dtype=torch.int64, device=device)
    # header, ranks, candidates, ordered.  Persistent per (device, capacity): zero-filled once -- the kernels leave the counters zeroed,
    # so no memset launch per pick (the calls of a device are ordered on its current stream)
    key = (device.index if device.index is not None else torch.cuda.current_device(), _capacity(r), torch.cuda.current_stream(device).cuda_stream)
    scratch = _scratch.get(key)
    if scratch is None:
        scratch = _scratch[key] = torch.zeros(2 + 5 * _capacity(r), dtype=torch.int32, device=device)
    L.check(L.load().nnr_randperm_prefix(L.ptr(keys), n, _key_bits(n), r, seed, offset, L.ptr(out), L.ptr(scratch),
                                         L.stream()), "nnr_randperm_prefix")
    return out


def supported(n: int, r: int) -> bool:
    bits = _key_bits(n)
    return bits > 32 and bits + max(1, (n - 1).bit_length()) <= 64 and _capacity(r) > 0 and n >= 8 * r


def randperm_prefix(n: int, r: int, device) -> torch.Tensor:
    """== torch.randperm(n, device=device)[:r], same generator side effects."""
    device = torch.device(device)
    if device.type != 'cuda' or not _state["enabled"] or not supported(n, r):
        return torch.randperm(n, device=device)[:r]
    if _state["checked"] < _CHECKS:
        _state["checked"] += 1
        before = torch.cuda.get_rng_state(device)
        ref = torch.randperm(n, device=device)[:r].clone()
        after = torch.cuda.get_rng_state(device)
        torch.cuda.set_rng_state(before, device)
        got = _fast(n, r, device)
        if not (torch.equal(got, ref) and torch.equal(torch.cuda.get_rng_state(device), after)):
            _state["enabled"] = False
            torch.cuda.set_rng_state(after, device)
            import warnings
            warnings.warn("nnr.sampling: the fast pixel pick does not reproduce this torch build's randperm; using torch.randperm")
            return ref
        return got
    return _fast(n, r, device)


# ---- rows of torch.rand(total) for a data-parallel shard ----------------------------------------------------------------------------
_rows_state = {"checked": 0, "enabled": True}
_rows_seen = set()      # (device index, total) whose draw has been compared with torch.rand
_max_blocks = {}


def _rand_launch_threads(total: int, device) -> int:
    """Threads of the launch torch's uniform kernel makes for `total` elements (ATen/native/cuda/DistributionTemplates.h:
    calc_execution_policy): 256 per block, min(SMs * (max threads per SM / 256), ceil(total / 256)) blocks."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    cap = _max_blocks.get(key)
    if cap is None:
        prop = torch.cuda.get_device_properties(device)
        cap = _max_blocks[key] = prop.multi_processor_count * (prop.max_threads_per_multi_processor // 256)
    return 256 * min(cap, (total + 255) // 256)


def _rows_fast(total: int, first: int, n: int, device) -> torch.Tensor:
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    seed, offset = gen.initial_seed(), gen.get_offset()
    threads = _rand_launch_threads(total, device)
    gen.set_offset(offset + ((total - 1) // (threads * 4) + 1) * 4)
    out = torch.empty(n, dtype=torch.float32, device=device)
    L.check(L.load().nnr_uniform_rows(seed, offset, threads, first, n, L.ptr(out), L.stream()), "nnr_uniform_rows")
    return out


def rand_rows(total: int, first: int, n: int, device) -> torch.Tensor:
    """== torch.rand(total, device=device)[first:first + n] with the same generator side effects, at the cost of n draws: the jitter rows
    of this rank's rays out of the whole step's tensor (model/rendering.py).  Verified against torch.rand on the first calls of a process and
    on the first call with every new `total`; on a mismatch (a torch build whose uniform kernel maps elements differently) it falls back to the full draw for good."""
    device = torch.device(device)
    if device.type != 'cuda' or not _rows_state["enabled"] or total <= 0:
        return torch.rand(total, device=device)[first:first + n].contiguous()
    # verified on the first _CHECKS calls AND on the first use of every distinct (device, total): another total may land in the other regime
    # of the launch (grid capped by the CU count or not), and a mapping that is wrong only there would silently desynchronise the jitter
    # streams of a data-parallel and a single-GPU run.  One extra full draw per new size.
    size_key = (device.index if device.index is not None else torch.cuda.current_device(), total)
    if _rows_state["checked"] < _CHECKS or size_key not in _rows_seen:
        _rows_state["checked"] += 1
        _rows_seen.add(size_key)
        before = torch.cuda.get_rng_state(device)
        ref = torch.rand(total, device=device)[first:first + n].clone()
        after = torch.cuda.get_rng_state(device)
        torch.cuda.set_rng_state(before, device)
        got = _rows_fast(total, first, n, device)
        if not (torch.equal(got, ref) and torch.equal(torch.cuda.get_rng_state(device), after)):
            _rows_state["enabled"] = False
            torch.cuda.set_rng_state(after, device)
            import warnings
            warnings.warn("nnr.sampling: the row-local jitter draw does not reproduce this torch build's rand; drawing the whole tensor")
            return ref
        return got
    return _rows_fast(total, first, n, device)
