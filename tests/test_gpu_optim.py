"""GPU (-m gpu): nnr.optim.MultiAdam -- one launch for the step's three Adam optimisers -- against torch.optim.Adam, BITWISE, in both of
torch's arithmetics: "single" = Adam(foreach=False, fused=False), the single-tensor implementation the reference's plain optim.Adam objects
run (train.py:58,99,117,140; the default of this repository since round 5), and "fused" = Adam(fused=True).

Training must not depend on which implementation stepped (the 800-step replay of the reference run, tests/test_conv_reference.py, sits
on top of this): parameters, first and second moments and step counters are compared with torch.equal after every block of steps, on
the shapes of the real step (the 24 tensors of the D = 256 network, pose tables, distortion tables), with gradients spanning twelve
orders of magnitude, exact zeros, a learning-rate change in mid-run, a parameter that sits out some steps, and a state_dict round trip
between the two implementations."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(dev, seed):
    import nerf_oracle as orc
    g = torch.Generator().manual_seed(seed)
    net = [v.clone().to(dev).requires_grad_(True) for v in orc.init_params(256, seed).values()]
    pose = [(0.01 * torch.randn(16, 3, generator=g)).to(dev).requires_grad_(True) for _ in range(2)]
    dist = [(1 + 0.05 * torch.randn(16, 1, generator=g)).to(dev).requires_grad_(True), (0.05 * torch.randn(16, 1, generator=g)).to(dev).requires_grad_(True)]
    return net, pose, dist


def _opts(groups, fused):
    lrs = (1e-3, 5e-4, 5e-4)
    kw = dict(fused=True) if fused else dict(fused=False, foreach=False)
    return [torch.optim.Adam(ps, lr=lr, **kw) for ps, lr in zip(groups, lrs)]


def _grads(groups, step, gen, skip_last):
    """The same pseudo-gradients for both copies: magnitudes 1e-9 .. 1e3, a tenth of the entries exactly zero."""
    out = []
    for gi, ps in enumerate(groups):
        for pi, p in enumerate(ps):
            mag = 10.0 ** (torch.rand(p.shape, generator=gen) * 12 - 9)
            g = torch.randn(p.shape, generator=gen) * mag
            g[torch.rand(p.shape, generator=gen) < 0.1] = 0.0
            out.append(None if (skip_last and gi == 2 and pi == 1 and step % 3 == 1) else g)
    return out


def _state(opts):
    out = []
    for o in opts:
        for grp in o.param_groups:
            for p in grp['params']:
                st = o.state.get(p, {})
                out.append((p.detach(), st.get('exp_avg'), st.get('exp_avg_sq'), st.get('step')))
    return out


def _assert_same(a, b, what):
    for i, (x, y) in enumerate(zip(_state(a), _state(b))):
        for j, (u, v) in enumerate(zip(x, y)):
            assert (u is None) == (v is None), (what, i, j)
            if u is not None:
                assert torch.equal(u.reshape(-1).float().cpu(), v.reshape(-1).float().cpu()), \
                    (what, i, ("param", "exp_avg", "exp_avg_sq", "step")[j], float((u.float().cpu().reshape(-1) - v.float().cpu().reshape(-1)).abs().max()))


@pytest.mark.parametrize("arithmetic", ["single", "fused"])
@pytest.mark.parametrize("skip_last", [False, True])
def test_one_launch_adam_is_bitwise_torch_adam(skip_last, arithmetic, capsys):
    from nnr.optim import MultiAdam
    dev = torch.device("cuda")
    ga, gb = _params(dev, 5), _params(dev, 5)
    fused = arithmetic == "fused"
    oa, ob = _opts(ga, fused), _opts(gb, fused)
    multi = MultiAdam(ob, arithmetic)
    assert multi.usable()
    gen = torch.Generator().manual_seed(11)
    n_steps = 240
    for step in range(n_steps):
        grads = _grads(ga, step, gen, skip_last)
        flat_a = [p for ps in ga for p in ps]
        flat_b = [p for ps in gb for p in ps]
        for p, q, g in zip(flat_a, flat_b, grads):
            p.grad = None if g is None else g.to(dev)
            q.grad = None if g is None else g.to(dev).clone()
        if step == 100:                      # what an LR scheduler does: rewrite param_groups on the host
            for o in oa + ob:
                for grp in o.param_groups:
                    grp['lr'] *= 0.37
        for o in oa:
            o.step()
        assert multi.step()
        if step in (0, 1, 2, 9, 99, 100, 101, n_steps - 1):
            _assert_same(oa, ob, "step %d" % step)
    # interchange: torch continues from MultiAdam's state and the other way round
    sd_a, sd_b = [copy.deepcopy(o.state_dict()) for o in oa], [copy.deepcopy(o.state_dict()) for o in ob]
    for o, sd in zip(oa, sd_b):
        o.load_state_dict(sd)
    for o, sd in zip(ob, sd_a):
        o.load_state_dict(sd)
    for step in range(n_steps, n_steps + 5):
        grads = _grads(ga, step, gen, False)
        for p, q, g in zip([p for ps in ga for p in ps], [p for ps in gb for p in ps], grads):
            p.grad, q.grad = g.to(dev), g.to(dev).clone()
        for o in oa:
            o.step()
        assert multi.step()
    _assert_same(oa, ob, "after the state_dict swap")
    with capsys.disabled():
        print("\nMultiAdam(%s) == torch.optim.Adam(%s) bitwise over %d steps (28 tensors, lr change at 100%s)"
              % (arithmetic, "fused=True" if fused else "foreach=False, fused=False", n_steps + 5, ", a parameter skipping steps" if skip_last else ""))


def test_the_two_arithmetics_differ_and_hand_over():
    """The flavours are not the same numbers (else the switch would be moot), and either continues from the other's state."""
    from nnr.optim import MultiAdam
    dev = torch.device("cuda")
    ga, gb = _params(dev, 7), _params(dev, 7)
    oa, ob = _opts(ga, False), _opts(gb, True)
    ma, mb = MultiAdam(oa, "single"), MultiAdam(ob, "fused")
    gen = torch.Generator().manual_seed(13)
    for step in range(20):
        grads = _grads(ga, step, gen, False)
        for p, q, g in zip([p for ps in ga for p in ps], [p for ps in gb for p in ps], grads):
            p.grad, q.grad = g.to(dev), g.to(dev).clone()
        assert ma.step() and mb.step()
    differ = sum(int((x[0] != y[0]).sum()) for x, y in zip(_state(oa), _state(ob)))
    assert differ > 0
    worst = max(float(((x[0] - y[0]).abs() / x[0].abs().clamp_min(1e-3)).max()) for x, y in zip(_state(oa), _state(ob)))
    assert worst < 1e-4, worst                      # last bits, not another optimiser
    # hand-over: the single flavour continues from the fused flavour's state (device counters become host counters) exactly like torch would
    sd = [copy.deepcopy(o.state_dict()) for o in ob]
    gc = _params(dev, 7)
    oc = _opts(gc, False)
    for p_c, p_b in zip([p for ps in gc for p in ps], [p for ps in gb for p in ps]):
        p_c.data.copy_(p_b.data)
    for o, s_ in zip(oc, sd):
        o.load_state_dict(copy.deepcopy(s_))      # (load_state_dict adopts tensors that already sit on the right device: two optimizers must not share them)
    mc = MultiAdam(oc, "single")
    od = _opts([[p.detach().clone().requires_grad_(True) for p in ps] for ps in gb], False)
    for o, s_ in zip(od, sd):
        o.load_state_dict(copy.deepcopy(s_))
    for o in oc + od:       # (a state_dict carries the param_groups' implementation switches too: back to the single-tensor implementation)
        for grp in o.param_groups:
            grp['fused'], grp['foreach'] = False, False
    grads = _grads(ga, 20, gen, False)
    for p, q, g in zip([p for o in oc for grp in o.param_groups for p in grp['params']], [p for o in od for grp in o.param_groups for p in grp['params']], grads):
        p.grad, q.grad = g.to(dev), g.to(dev).clone()
    assert mc.step()
    for o in od:
        o.step()
    _assert_same(od, oc, "hand-over from the fused flavour")


def test_multi_adam_declines_what_it_does_not_cover():
    from nnr.optim import MultiAdam
    dev = torch.device("cuda")
    p = torch.zeros(8, device=dev, requires_grad=True)
    assert not MultiAdam([torch.optim.Adam([p], lr=1e-3, weight_decay=0.1)]).usable()
    assert not MultiAdam([torch.optim.Adam([p], lr=1e-3, amsgrad=True)]).usable()
    assert not MultiAdam([torch.optim.SGD([p], lr=1e-3)]).usable()
    q = torch.zeros(8, requires_grad=True)
    assert not MultiAdam([torch.optim.Adam([q], lr=1e-3)]).usable()
    m = MultiAdam([torch.optim.Adam([p], lr=1e-3)])
    p.grad = torch.ones(16, device=dev)[::2]          # a non-contiguous gradient: declined, nothing touched
    assert m.usable() and not m.step() and float(p.abs().max()) == 0.0
