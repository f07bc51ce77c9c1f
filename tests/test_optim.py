"""the trainer switches plain torch.optim.Adam instances on GPU parameters to torch's single-kernel ('fused')
implementation.  CPU: what is left alone.  GPU: the update is the one the default implementation makes, and the state_dict
layout is unchanged."""
import pytest
import torch


def test_leaves_other_optimisers_and_cpu_parameters_alone():
    from model.training import _use_fused_adam
    w = torch.nn.Parameter(torch.zeros(4))
    assert _use_fused_adam(None) is False
    assert _use_fused_adam(torch.optim.SGD([w], lr=0.1)) is False
    adam = torch.optim.Adam([w], lr=0.1)
    assert _use_fused_adam(adam) is False and not adam.param_groups[0].get('fused')       # CPU parameters
    assert _use_fused_adam(torch.optim.AdamW([w], lr=0.1)) is False                         # only plain Adam


@pytest.mark.gpu
def test_fused_adam_matches_default_adam_and_keeps_the_state_layout():
    from model.training import _use_fused_adam
    g = torch.Generator().manual_seed(0)
    init = [torch.randn(37, 5, generator=g), torch.randn(11, generator=g)]
    grads = [[torch.randn_like(t) for t in init] for _ in range(4)]

    def run(fused, resume_at=None):
        ps = [torch.nn.Parameter(t.clone().cuda()) for t in init]
        opt = torch.optim.Adam(ps, lr=1e-2)
        if fused:
            assert _use_fused_adam(opt)
        for i, gs in enumerate(grads):
            if resume_at == i:           # state created by the default implementation, then switched (checkpoint resume order)
                assert _use_fused_adam(opt)
            for p, gr in zip(ps, gs):
                p.grad = gr.cuda()
            opt.step()
        return [p.detach().cpu() for p in ps], opt

    ref, opt_ref = run(False)
    got, opt_f = run(True)
    mid, _ = run(False, resume_at=2)
    for a, b, c in zip(ref, got, mid):
        torch.testing.assert_close(b, a, rtol=0, atol=1e-6)
        torch.testing.assert_close(c, a, rtol=0, atol=1e-6)
    sa, sb = opt_ref.state_dict()['state'][0], opt_f.state_dict()['state'][0]
    assert set(sa) == set(sb) == {'step', 'exp_avg', 'exp_avg_sq'} and float(sa['step']) == float(sb['step']) == 4.0


@pytest.mark.gpu
def test_packed_weights_follow_a_fused_optimizer_step():
    """torch's fused Adam updates parameters without bumping tensor._version; the packed copy the kernels read must still
    be refreshed (nnr/ops.py: global optimizer post-step hook).  A stale pack would freeze training silently."""
    import model as mdl
    from nnr import ops
    from model.training import _use_fused_adam
    dev = torch.device("cuda")
    torch.manual_seed(0)
    net = mdl.OfficialStaticNerf({'model': {'hidden_dim': 128, 'pos_enc_levels': 10, 'dir_enc_levels': 4,
                                            'occ_activation': 'softplus'}, 'rendering': {'white_background': False, 'dist_alpha': False}}).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)
    assert _use_fused_adam(opt)
    pts, view = torch.randn(256, 3, device=dev), torch.nn.functional.normalize(torch.randn(256, 3, device=dev), dim=-1)
    with torch.no_grad():
        before, _ = ops.mlp_points(pts, view, net.weights(), net.biases(), hidden=128)
        again, _ = ops.mlp_points(pts, view, net.weights(), net.biases(), hidden=128)      # cache hit: identical
    assert torch.equal(before, again)
    versions = [p._version for p in net.parameters()]
    for p in net.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    assert versions == [p._version for p in net.parameters()]                 # the premise: no version bump
    with torch.no_grad():
        after, _ = ops.mlp_points(pts, view, net.weights(), net.biases(), hidden=128)
    assert float((after - before).abs().max()) > 1e-4                         # the new weights are the ones evaluated
