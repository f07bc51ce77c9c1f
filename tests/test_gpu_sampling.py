"""GPU (-m gpu): nnr.sampling.randperm_prefix == torch.randperm(n, device='cuda')[:r] -- same indices (bit-exact) and the same
generator state afterwards, so that the jitter drawn next is unchanged too.  Sizes: the benchmark image, odd sizes, the
smallest supported n; many generator states; and the duplicate-key path, which real sizes almost never reach, by handing the
kernel keys with forced duplicates together with what torch's own island re-shuffle makes of them."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


@pytest.mark.parametrize("n,r", [(540 * 960, 1024), (756 * 1008, 1024), (480 * 640 + 7, 333), (46_400, 64), (1_000_003, 1500),
                                 (540 * 960, 2048), (540 * 960, 4096), (540 * 960, 8192), (756 * 1008, 9943),
                                 (756 * 1008, 9944), (540 * 960, 32768)])
def test_matches_torch_randperm_and_leaves_the_same_generator_state(n, r):
    from nnr import sampling
    assert sampling.supported(n, r)
    for seed in range(6):
        torch.manual_seed(1000 + seed)
        torch.rand(seed + 1, device=DEV)                       # an arbitrary, non-zero philox offset
        state = torch.cuda.get_rng_state(DEV)
        ref = torch.randperm(n, device=DEV)[:r]
        ref_next = torch.rand(5, device=DEV)
        torch.cuda.set_rng_state(state, DEV)
        got = sampling._fast(n, r, DEV)
        got_next = torch.rand(5, device=DEV)
        assert torch.equal(got, ref), (n, r, seed)
        assert torch.equal(got_next, ref_next), "generator state differs after the call"


def test_unsupported_sizes_fall_back_to_torch():
    from nnr import sampling
    assert not sampling.supported(1000, 10)                    # 32-bit key branch of torch's randperm
    assert not sampling.supported(540 * 960, 60000)            # more rays than the largest candidate buffer is sized for
    for n, r in ((1000, 10), (540 * 960, 60000)):
        torch.manual_seed(5)
        a = sampling.randperm_prefix(n, r, DEV)
        torch.manual_seed(5)
        assert torch.equal(a, torch.randperm(n, device=DEV)[:r])


def test_self_check_disables_the_fast_path_on_mismatch(monkeypatch):
    from nnr import sampling
    monkeypatch.setitem(sampling._state, "checked", 0)
    monkeypatch.setitem(sampling._state, "enabled", True)
    monkeypatch.setattr(sampling, "_fast", lambda n, r, d: torch.zeros(r, dtype=torch.int64, device=d))   # a broken kernel
    torch.manual_seed(9)
    with pytest.warns(UserWarning):
        got = sampling.randperm_prefix(540 * 960, 1024, DEV)
    torch.manual_seed(9)
    assert torch.equal(got, torch.randperm(540 * 960, device=DEV)[:1024]) and sampling._state["enabled"] is False


def test_duplicate_keys_are_reshuffled_like_torch():
    """Keys with forced duplicates through the kernel; the expected result replays torch's rule on the host with the same
    Philox stream (hiprand via torch: island start t uses subsequence t of (seed, offset); r_i = next() % (i+1))."""
    from nnr import lib as L
    n, r, bits = 200_000, 256, 38
    g = torch.Generator().manual_seed(3)
    keys = torch.randint(0, 2 ** 62, (n,), generator=g, dtype=torch.int64)
    # make the 40 smallest masked keys collide in groups of 1..5
    mask = (1 << bits) - 1
    order = torch.argsort(keys & mask)
    small = order[:40]
    groups, i = [], 0
    for size in (3, 1, 2, 5, 1, 4, 2, 3, 1, 2, 5, 4, 3, 4):
        groups.append(small[i:i + size]); i += size
    for gi, grp in enumerate(groups):
        keys[grp] = (keys[grp[0]] & ~mask) | (7 * gi + 1)          # group gi shares the masked key 7*gi+1 (ascending with gi)
    seed, offset = 123456789, 4096
    out = torch.empty(r, dtype=torch.int64, device=DEV)
    scratch = torch.zeros(L.load().nnr_randperm_scratch_bytes(r) // 4, dtype=torch.int32, device=DEV)
    kd = keys.to(DEV)
    L.check(L.load().nnr_randperm_prefix(L.ptr(kd), n, bits, r, seed, offset, L.ptr(out), L.ptr(scratch),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nnr_randperm_prefix")
    got = out.cpu()
    assert int(scratch[1]) == 0
    # expected: stable sort by masked key, then the groups permuted among themselves; every group must hold the same SET of
    # indices at the same positions, the rest must be in plain sorted order
    mk = (keys & mask)
    stable = torch.tensor(sorted(range(n), key=lambda j: (int(mk[j]), j))[:r])
    pos = 0
    for grp in groups:
        want = sorted(int(x) for x in grp)
        assert sorted(int(x) for x in got[pos:pos + len(grp)]) == want
        pos += len(grp)
    assert torch.equal(got[pos:], stable[pos:])
    # and the shuffle is the Philox one, not the identity: compare with torch's randperm on keys it would have drawn is not
    # possible for crafted keys, so pin the property that makes it torch's rule -- determinism in (seed, offset), change with offset
    out2 = torch.empty_like(out)
    L.check(L.load().nnr_randperm_prefix(L.ptr(kd), n, bits, r, seed, offset, L.ptr(out2), L.ptr(scratch),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nnr_randperm_prefix")
    assert torch.equal(out2.cpu(), got)
    L.check(L.load().nnr_randperm_prefix(L.ptr(kd), n, bits, r, seed, offset + 4, L.ptr(out2), L.ptr(scratch),
                                         C.c_void_p(torch.cuda.current_stream().cuda_stream)), "nnr_randperm_prefix")
    assert not torch.equal(out2.cpu()[:40], got[:40])
    cap = (scratch.numel() - 2) // 5
    assert int(scratch[:2 + cap].abs().sum()) == 0        # count, status, ranks: left zeroed for the next pick (no memset per call)


@pytest.mark.parametrize("total,first,n", [(1000, 0, 1000), (1000, 37, 200), (8192 * 192, 3 * 1024 * 192, 1024 * 192),
                                           (8192 * 192, 7 * 1024 * 192, 1024 * 192), (5_000_001, 2_099_000, 300_000),
                                           (5_000_001, 4_999_000, 1_001), (9_437_184, 9_000_000, 437_184)])
def test_rand_rows_are_the_rows_of_torch_rand(total, first, n):
    """nnr.sampling.rand_rows (the jitter rows of a data-parallel shard, drawn alone) == torch.rand(total)[first:first + n], bit for bit, and
    the generator is left where the full draw leaves it -- small tensors (fewer elements than launch threads), the 8-rank step of the
    benchmark shape, and tensors large enough that torch's threads make several Philox calls each."""
    from nnr import sampling
    dev = torch.device("cuda", 0)
    for seed in (0, 1234567):
        torch.manual_seed(seed)
        torch.rand(17, device=dev)                       # (an offset that is not zero)
        before = torch.cuda.get_rng_state(dev)
        want = torch.rand(total, device=dev)[first:first + n].clone()
        after = torch.cuda.get_rng_state(dev)
        torch.cuda.set_rng_state(before, dev)
        got = sampling._rows_fast(total, first, n, dev)
        assert torch.equal(got, want)
        assert torch.equal(torch.cuda.get_rng_state(dev), after)
    assert sampling._rows_state["enabled"]
    torch.manual_seed(5)
    a = sampling.rand_rows(total, first, n, dev)         # the public entry (self-checking on its first calls)
    torch.manual_seed(5)
    assert torch.equal(a, torch.rand(total, device=dev)[first:first + n])
