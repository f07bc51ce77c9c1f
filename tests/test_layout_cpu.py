"""CPU: the index algebra of the device layouts closes (numpy emulation of the MFMA operand mapping), the C-ABI
library loads and exports every symbol of include/nnr.h, sizes agree with the numpy layout, and the weight-gradient
plan tiles every parameter exactly once."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import layout_ref as lr

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rand_weights(D, seed=0):
    rng = np.random.default_rng(seed)
    P, Q = 63, 27
    shapes = [(D, P), (D, D), (D, D), (D, D), (D, D + P), (D, D), (D, D), (D, D), (1, D), (D, D), (D // 2, D + Q), (3, D // 2)]
    return [rng.standard_normal(s).astype(np.float32) for s in shapes], [rng.standard_normal(s[0]).astype(np.float32) for s in shapes]


def _layer(W_l, parts, idxs, x_regs):
    """Emulate the passes `idxs` of one layer on input registers; returns the dense (rows, 32) result."""
    outs = []
    for i in idxs:
        part = parts[i]
        pk = lr.pack_part(lr.part_matrix(W_l, part), part[2], part[3])
        acc = lr.gemm_part_emulated(pk, x_regs, part[2], part[3])
        outs.append(lr.from_regs(acc.reshape(-1, 64))[:part[4]])
    return np.concatenate(outs)


def test_three_term_parts_reproduce_the_fp32_products():
    """MODE 2 (NNR_F_SPLIT3): a part packed as three bf16 terms per weight, multiplied by the emulated six-term row of nnr_split.h
    against activations in the register layout, gives the plain matmul to fp32 rounding -- for every (KT, MT) shape the kernels use
    (forward and transposed parts, edge tiles with zero padding), i.e. the slot algebra (rows of 16 k, 3 MT fragments per row, 24-slot
    panels) closes; and the terms themselves add up to the weights exactly."""
    D = 128
    W, _ = rand_weights(D, seed=3)
    rng = np.random.default_rng(4)
    x = rng.standard_normal(8).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 6, 8).astype(np.float32)
    l, m, h = lr.split3(x)
    assert np.array_equal((h.astype(np.float64) + m + l).astype(np.float32), x) and np.all(np.abs(m) <= np.abs(h) * 2.0 ** -8)
    for parts, idx in ((lr.fwd_parts(D), (0, 2, 9, 19)), (lr.bwd_parts(D), (2, 3, 9, 18))):
        for i in idx:
            part = parts[i]
            layer, tr, KT, MT, m_real, k_real = part[:6]
            A = lr.part_matrix(W[layer] if layer != lr.MERGED else rng.standard_normal((D // 2, D)).astype(np.float32), part)
            X = np.zeros((32 * KT, 32), dtype=np.float32)
            X[:k_real] = rng.standard_normal((k_real, 32)).astype(np.float32)
            pk = lr.pack_part_split(A, KT, MT)
            assert pk.size == (-(-2 * KT // (8 // MT))) * 24 * 256
            acc = lr.gemm_part_split_emulated(pk, lr.to_regs(X), KT, MT)
            got = lr.from_regs(acc.reshape(-1, 64))
            want = A.astype(np.float64) @ X.astype(np.float64)
            scale = np.abs(A).astype(np.float64) @ np.abs(X).astype(np.float64) + 1e-30
            assert float((np.abs(got - want) / scale).max()) <= 3e-7, (part, float((np.abs(got - want) / scale).max()))


def test_two_term_fp16_parts_reproduce_the_fp32_products():
    """MODE 3 (NNR_F_SPLIT2, csrc/nnr_split2.h): a part packed as the two fp16 fragment classes of the power-of-two scaled weight (m, h; the
    third operand h 2^-11 made from h by an fp16 multiply), multiplied by the emulated three-MFMA row against activations whose residual term is carried at 2^11, gives the plain matmul to fp32
    rounding for every (KT, MT) shape the kernels use; the two terms of a value add up to it within 2^-22; tiny and large operands alike
    (weights are scaled into fp16's range, the activations' residual stays a normal fp16 number down to 2^-14)."""
    D = 128
    W, _ = rand_weights(D, seed=3)
    rng = np.random.default_rng(4)
    x = rng.standard_normal(64).astype(np.float32) * np.float32(10.0) ** rng.integers(-4, 4, 64).astype(np.float32)
    xh = x.astype(np.float16)
    xm = ((x - xh.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    back = xh.astype(np.float64) + xm.astype(np.float64) / 2048.0
    assert np.all(np.abs(back - x) <= np.maximum(np.abs(x) * 2.0 ** -21.9, 2.0 ** -36))
    for wscale in (1.0, 1e-3, 40.0):
        for parts, idx in ((lr.fwd_parts(D), (0, 2, 9, 19)), (lr.bwd_parts(D), (2, 3, 9, 18))):
            for i in idx:
                part = parts[i]
                layer, tr, KT, MT, m_real, k_real = part[:6]
                A = lr.part_matrix((W[layer] if layer != lr.MERGED else rng.standard_normal((D // 2, D)).astype(np.float32)) * np.float32(wscale), part)
                sc = lr.pow2_scale(np.abs(A).max())
                assert 2.0 ** 13 <= np.abs(A).max() * sc < 2.0 ** 14
                X = np.zeros((32 * KT, 32), dtype=np.float32)
                X[:k_real] = np.maximum(rng.standard_normal((k_real, 32)), 0).astype(np.float32) * np.float32(10.0) ** rng.integers(-3, 2, (k_real, 1)).astype(np.float32)
                pk = lr.pack_part_split2(A * sc, KT, MT)
                assert pk.size == (-(-2 * KT // (16 // MT))) * 32 * 256
                acc = lr.gemm_part_split2_emulated(pk, lr.to_regs(X), KT, MT)
                got = lr.from_regs(acc.reshape(-1, 64)).astype(np.float64) / float(sc)
                want = A.astype(np.float64) @ X.astype(np.float64)
                scale = np.abs(A).astype(np.float64) @ np.abs(X).astype(np.float64) + 1e-30
                assert float((np.abs(got - want) / scale).max()) <= 4e-7, (part, wscale, float((np.abs(got - want) / scale).max()))


def test_chained_layers_in_register_layout():
    """Two chained layers, each as its two half-output passes, computed with emulated MFMAs on packed fragments == plain
    matmuls; the output registers of one layer are directly the B operands of the next (no transpose)."""
    D = 128
    W, _ = rand_weights(D)
    rng = np.random.default_rng(1)
    e = np.zeros((64, 32), dtype=np.float32)
    e[:63] = rng.standard_normal((63, 32))
    parts = lr.fwd_parts(D)
    h1 = np.maximum(_layer(W[0], parts, (0, 1), lr.to_regs(e)), 0)
    np.testing.assert_allclose(h1, np.maximum(W[0] @ e[:63], 0), rtol=1e-5, atol=1e-5)
    h2 = _layer(W[1], parts, (2, 3), lr.to_regs(h1))
    np.testing.assert_allclose(h2, W[1] @ np.maximum(W[0] @ e[:63], 0), rtol=1e-4, atol=1e-4)


def test_transposed_parts_and_skip_layer():
    D = 128
    W, _ = rand_weights(D)
    rng = np.random.default_rng(2)
    d5 = rng.standard_normal((D, 32)).astype(np.float32)
    bp = lr.bwd_parts(D)
    idx = [i for i, p in enumerate(bp) if p[0] == 4]              # hidden 5 transposed: posenc rows, then the two h4 halves
    assert [bp[i][6] for i in idx] == [D, 0, D // 2]
    ref = W[4].T @ d5
    np.testing.assert_allclose(_layer(W[4], bp, idx[1:], lr.to_regs(d5)), ref[:D], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(_layer(W[4], bp, idx[:1], lr.to_regs(d5)), ref[D:], rtol=1e-4, atol=1e-4)
    fp = lr.fwd_parts(D)
    e_parts = [p for p in fp if p[0] == 4 and p[7] == D]          # skip layer, posenc columns D.. of layers1.0
    assert len(e_parts) == 2
    A = lr.part_matrix(W[4], e_parts[1])
    np.testing.assert_array_equal(A[:D // 2, :63], W[4][D // 2:, D:])
    assert np.all(A[:, 63] == 0)


def test_head_tables_are_register_ordered_rows():
    D = 256
    W, _ = rand_weights(D)
    t = lr.head_tables(W, D)
    h8 = np.random.default_rng(3).standard_normal((D, 32)).astype(np.float32)
    regs = lr.to_regs(h8)                                         # (128, 64): lane = 32*half + sample
    wsig = t[:2 * 128].reshape(2, 128)
    dot = np.zeros(64)
    for lane in range(64):
        dot[lane] = np.dot(wsig[lane >> 5], regs[:, lane])
    np.testing.assert_allclose(dot[:32] + dot[32:], (W[8] @ h8)[0], rtol=1e-4, atol=1e-4)
    assert t.size == 2 * 128 + 3 * 2 * 64


def test_library_exports_every_declared_symbol():
    from nnr import lib as L
    hdr = open(os.path.join(ROOT, "include", "nnr.h")).read()
    declared = set(re.findall(r"\b(nnr_[a-z_0-9]+)\s*\(", hdr)) - {"nnr_cfg", "nnr_params", "nnr_param_grads"}
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    # the three statements of the ABI version agree: header macro, library, bindings
    assert lib.nnr_abi_version() == L.ABI_VERSION == int(re.search(r"#define NNR_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.nnr_strerror(-2).decode().startswith("unsupported")


@pytest.mark.parametrize("D", [128, 256])
def test_sizes_and_error_codes(D):
    from nnr import lib as L
    lib = L.load()
    W, B = rand_weights(D)
    cfg = L.make_cfg(16, 64, D, train=True)
    cfg = L.Cfg(16, 64, D, cfg.flags & ~(L.NNR_F_SPLIT3 | L.NNR_F_SPLIT2))               # fp32-MFMA products
    assert lib.nnr_packed_floats(C.byref(cfg)) == lr.pack_all(W, B, D).size
    split_cfg = L.Cfg(16, 64, D, cfg.flags | L.NNR_F_SPLIT3)          # three-term products: only the packed weights differ
    assert lib.nnr_packed_floats(C.byref(split_cfg)) == lr.pack_all(W, B, D, mode=2).size
    # two-term fp16 products (the default): two fragment classes + the scale table; the training workspace carries the table of plane maxima
    split2_cfg = L.Cfg(16, 64, D, cfg.flags | L.NNR_F_SPLIT3 | L.NNR_F_SPLIT2)
    assert lib.nnr_packed_floats(C.byref(split2_cfg)) == lr.pack_all(W, B, D, mode=3).size
    nj3 = C.c_int32(0)
    assert lib.nnr_plan_counts(C.byref(split2_cfg), C.byref(nj3), None) == 0
    # ... and the weight-gradient plan (the 4 x 4 tiles are cheaper there, so the schedule cuts differently): same planes, other job count
    nj0, nj2 = C.c_int32(0), C.c_int32(0)
    assert lib.nnr_plan_counts(C.byref(cfg), C.byref(nj0), None) == 0 and lib.nnr_plan_counts(C.byref(split_cfg), C.byref(nj2), None) == 0
    slot = 128 * 128 + 256
    assert lib.nnr_workspace_floats(C.byref(split_cfg)) - nj2.value * slot == lib.nnr_workspace_floats(C.byref(cfg)) - nj0.value * slot
    assert lib.nnr_workspace_floats(C.byref(split2_cfg)) - nj3.value * slot == lib.nnr_workspace_floats(C.byref(cfg)) - nj0.value * slot + 32      # + kPlaneMaxFloats
    S_pad = 16 * 64
    x_width = 64 + 8 * D + 32 + D // 2      # posenc, h1..h8, direction encoding, colour hidden (no feature vector: merged)
    d_width = 8 * D + D // 2
    expect = S_pad * (4 + 1 + 4 + 4 + 4 + x_width + d_width + 9 * 2 * (D // 64))
    nj = C.c_int32(0)
    assert lib.nnr_plan_counts(C.byref(cfg), C.byref(nj), None) == 0
    n_jobs = nj.value
    merged = (D // 2) * D + D // 2                # dW', db' of the merged feature/colour matrix
    assert lib.nnr_workspace_floats(C.byref(cfg)) == expect + n_jobs * (128 * 128 + 256) + merged   # + one partial slot per job
    assert lib.nnr_workspace_floats(C.byref(L.make_cfg(16, 64, D))) == S_pad * 5
    assert lib.nnr_workspace_floats(C.byref(L.make_cfg(16, 64, 192))) == 0        # unsupported width
    assert lib.nnr_pack_weights(C.byref(L.make_cfg(16, 64, 192)), None, None, None) == -2
    assert lib.nnr_pack_weights(C.byref(cfg), None, None, None) == -1              # null pointers, no GPU touched


@pytest.mark.parametrize("D,R,N", [(128, 32, 64), (256, 1024, 192), (256, 5, 33)])
def test_wgrad_plan_covers_every_weight_once(D, R, N):
    """Every (layer,row,col) of the 12 weight tensors is produced by jobs whose sample ranges partition [0, S_pad), and
    bias rows are reduced exactly once per range."""
    from nnr import lib as L
    from nnr.ops import plan_jobs
    cfg = L.make_cfg(R, N, D, train=True)
    jobs = plan_jobs(cfg)
    S_pad = (R * N + 127) // 128 * 128
    # index 12 = the merged feature/colour matrix W' (D/2 x D); the feature layer (9) and the first D columns of the
    # colour-hidden layer (10) get their gradients from it in the un-merge step, so no job covers them
    shapes = [(D, 63), (D, D), (D, D), (D, D), (D, D + 63), (D, D), (D, D), (D, D), (1, D), (D, D), (D // 2, D + 27), (3, D // 2),
              (D // 2, D)]
    cover = [np.zeros(s, dtype=np.int64) for s in shapes]
    bias_cover = [np.zeros(s[0], dtype=np.int64) for s in shapes]
    m = np.arange(32)
    for j in jobs:
        assert j.k0 % 16 == 0 and j.k1 % 16 == 0 and 0 <= j.k0 < j.k1 <= S_pad
        assert j.ldw == shapes[j.layer][1] and j.rows_real == shapes[j.layer][0]
        rows = (j.row0 + j.MI * m[:, None] + np.arange(j.MI)[None, :]).reshape(-1)
        dvalid = (j.MI * m[:, None] + np.arange(j.MI)[None, :]).reshape(-1) < j.d_valid
        cols = (j.wcol0 + j.NI * m[:, None] + np.arange(j.NI)[None, :]).reshape(-1)
        xvalid = (j.NI * m[:, None] + np.arange(j.NI)[None, :]).reshape(-1) < j.x_valid
        rows = rows[dvalid & (rows < j.rows_real)]
        cols = cols[xvalid & (cols < j.cols_real)]
        cover[j.layer][np.ix_(rows, cols)] += j.k1 - j.k0
        if j.bias:   # 1: every sample pair; 2 / 3: the even / odd pairs (shared between the two tiles of a row block)
            bias_cover[j.layer][rows] += (j.k1 - j.k0) * (2 if j.bias == 1 else 1)
    for l in range(13):
        if l == 9:
            assert not cover[l].any() and not bias_cover[l].any()
        elif l == 10:
            assert not cover[l][:, :D].any() and np.all(cover[l][:, D:] == S_pad) and not bias_cover[l].any()
        else:
            assert np.all(cover[l] == S_pad), (l, np.unique(cover[l]))
            assert np.all(bias_cover[l] == 2 * S_pad), l
    allj, first = plan_jobs(cfg, with_waves=True)
    assert (len(first) - 1) % 4 == 0 and (len(first) - 1) // 4 <= 256 and first[0] == 0 and first[-1] == len(allj)
    assert all(a <= b for a, b in zip(first, first[1:]))
    for idx, j in enumerate(allj):           # the splits of a tile are chained in sample order, split 0 first
        if j.split == 0:
            k, n, cur = j.k0, 0, idx
            assert k == 0
            while cur >= 0:
                o = allj[cur]
                assert (o.layer, o.row0, o.wcol0, o.split, o.k0) == (j.layer, j.row0, j.wcol0, n, k)
                k, n, cur = o.k1, n + 1, o.next_split
            assert k == S_pad
    # balance: at the benchmark size no wave has more than 3 % above the mean work (at the plan weight 0.44 of round 4)
    if (D, R, N) == (256, 1024, 192):
        # (three-term mode, the default of make_cfg: the 4 x 4 tiles run on the bf16 matrix pipe and are weighed at 0.44 of an fp32 tile: kSplitWeight in nnr_api.cpp)
        # (two-term mode, the default since round 6: the workgroup jobs of hidden layers 2..8 take three fp16 terms, 0.34: split2_w in nnr_api.cpp)
        w44 = 0.44 if (cfg.flags & L.NNR_F_SPLIT3) else 1.0
        w34 = 0.34 if (cfg.flags & L.NNR_F_SPLIT2) else w44
        narrow = {8: 1.035, 4: 1.145}      # (measured cycles per MFMA of the narrow tiles relative to a 4 x 4 fp32 tile: the `weight` table of build_plan)
        # (the 128 x 64 tiles against the position encoding -- x_plane 10 -- as private two-term jobs: 0.625, enc2_w in nnr_api.cpp; marked reserved = 2)
        enc = lambda j: bool(cfg.flags & L.NNR_F_SPLIT2) and (j.MI, j.NI, j.x_plane) == (4, 2, 10)
        assert all((j.reserved == 2) == enc(j) for j in allj)
        cost = lambda j: ((w34 if 1 <= j.layer <= 7 else w44) if j.MI * j.NI == 16 else (0.625 if enc(j) else narrow.get(j.MI * j.NI, 1.25))) * j.MI * j.NI * (j.k1 - j.k0)
        work = [sum(cost(allj[i]) for i in range(first[w], first[w + 1])) for w in range(len(first) - 1)]
        # a workgroup is done when its slowest wave is (the narrow tiles run in bundles whose waves differ: nnr_api.cpp build_plan): no
        # workgroup more than 3 % above the mean
        load = [max(work[4 * b:4 * b + 4]) for b in range(len(work) // 4)]
        assert len(work) == 1024 and max(load) <= 1.03 * sum(load) / len(load)


# ---- bf16 training mode: tile-major planes, the workgroup-job plan, the DMA swizzle / transposing-read index algebra --------
def tile_major_index(s, f, G):
    """nnr_layout.h: bf16 index of (sample s, feature f) in a tile-major plane of G 16-feature groups."""
    return (((s // 32) * G + f // 16) * 64 + 32 * ((f % 8) // 4) + s % 32) * 8 + 4 * ((f % 16) // 8) + f % 4


@pytest.mark.parametrize("D,R,N", [(128, 32, 64), (256, 4096, 128), (256, 5, 33), (256, 1024, 192)])
def test_bf16_wgrad_plan_covers_every_weight_once(D, R, N):
    """bf16 mode: every (layer, row, col) of the weight tensors is the destination of exactly one output rectangle, the jobs of a
    unit partition the sample chunks, every job's tiles are owned by exactly one wave, and the staged groups exist in their planes."""
    from nnr import lib as L
    from nnr.ops import plan_bf16
    lib = L.load()
    cfg = L.make_cfg(R, N, D, train=True, bf16=True)
    jobs, first, outs = plan_bf16(cfg)
    S_pad = (R * N + 127) // 128 * 128
    chunks = S_pad // 32
    shapes = [(D, 63), (D, D), (D, D), (D, D), (D, D + 63), (D, D), (D, D), (D, D), (1, D), (D, D), (D // 2, D + 27), (3, D // 2),
              (D // 2, D)]
    cover = [np.zeros(s, dtype=np.int64) for s in shapes]
    bias_cover = [np.zeros(s[0], dtype=np.int64) for s in shapes]
    units = {}
    for j in jobs:
        units.setdefault(j.unit, []).append(j)
        assert (j.MT, j.NT) in {(4, 5), (5, 3), (4, 4), (2, 2), (5, 2), (3, 1), (1, 2), (1, 1)} and 1 <= j.WR * j.WC <= 4
        staged = j.d_groups + j.x_groups + j.x2_groups
        assert 0 <= j.c0 < j.c1 <= chunks and staged <= 36 and j.d_base % 1024 == 0 and j.x_base % 1024 == 0 and j.x2_base % 1024 == 0
        assert j.x2_groups == 0 or j.x_groups % 2 == 0       # the second activation plane continues the first one's tile grid
        # every tile row / column a wave reads lies inside the staged image (a tile = two blocks)
        assert 2 * j.MT * j.WR <= staged + 1 and 2 * j.NT * j.WC <= j.x_groups + j.x2_groups
    for u, js in units.items():                      # chunk ranges of a unit: a partition, chained in order
        js = sorted(js, key=lambda j: j.c0)
        assert js[0].c0 == 0 and js[-1].c1 == chunks and all(a.c1 == b.c0 for a, b in zip(js, js[1:]))
        assert [j.split for j in js] == list(range(len(js)))
    for o in outs:
        head = jobs[o.first_job]
        assert head.unit == o.unit and head.split == 0 and (o.MT, o.NT, o.WR, o.WC) == (head.MT, head.NT, head.WR, head.WC)
        n, cur = 0, o.first_job
        while cur >= 0:
            assert jobs[cur].unit == o.unit
            n, cur = n + 1, jobs[cur].next_split
        assert n == len(units[o.unit])
        # the rectangle lies inside the tiles the unit's waves own
        assert o.d_row + o.n_rows <= 32 * o.MT * o.WR and o.x_col + o.n_cols <= 32 * o.NT * o.WC
        assert o.ldw == shapes[o.layer][1]
        cover[o.layer][o.w_row:o.w_row + o.n_rows, o.w_col:o.w_col + o.n_cols] += 1
        if o.bias:
            assert head.bias == 1
            bias_cover[o.layer][o.w_row:o.w_row + o.n_rows] += 1
    for l in range(13):
        if l == 9:       # feature layer: from the merged matrix (un-merge step)
            assert not cover[l].any() and not bias_cover[l].any()
        elif l == 10:
            assert not cover[l][:, :D].any() and np.all(cover[l][:, D:] == 1) and not bias_cover[l].any()
        else:
            assert np.all(cover[l] == 1), l
            assert np.all(bias_cover[l] == 1), l
    n_blocks = len(first) - 1
    assert 1 <= n_blocks <= 256 and first[0] == 0 and first[-1] == len(jobs) and all(a <= b for a, b in zip(first, first[1:]))
    # balance at BASELINE configs[2]: the staged KiB per workgroup differ by at most one chunk of the widest unit
    if (R, N) == (4096, 128):
        kib = [sum((jobs[i].d_groups + jobs[i].x_groups + jobs[i].x2_groups) * (jobs[i].c1 - jobs[i].c0) for i in range(first[b], first[b + 1])) for b in range(n_blocks)]
        assert n_blocks == 256 and max(kib) - min(kib) <= 2 * 36
    # workspace: planes + four wave slots per job + the merged-matrix scratch
    nj = C.c_int32(0)
    assert lib.nnr_plan_counts(C.byref(cfg), C.byref(nj), None) == 0 and nj.value == len(jobs)
    pitch = C.c_int32(0)
    assert lib.nnr_ws_plane(C.byref(cfg), 40, C.byref(pitch)) >= 0 and pitch.value == 8 * (D // 32 + 1)     # P_DG: one extra group
    assert lib.nnr_ws_plane(C.byref(cfg), 21, C.byref(pitch)) >= 0 and pitch.value == 32                    # P_XE16
    assert lib.nnr_ws_plane(C.byref(cfg), 22, C.byref(pitch)) >= 0 and pitch.value == 16                    # P_XF16


def test_bf16_wgrad_lds_addressing_delivers_mfma_operands_without_bank_conflicts():
    """The index algebra of nnr_wgrad_bf16.hip, emulated: a tile-major block pair goes through the DMA's source-lane permutation
    into LDS; the lanes' transposing reads (ds_read_b64_tr_b16: lane i of a 16-lane group receives element i of each of the four
    8-byte rows the group's lanes 4 r + m address) must hand lane l feature 32 t + (l & 31) for the samples 16 ks + 8 (l >> 5) + 0..7
    -- the A / B operand of v_mfma_f32_32x32x16_bf16 -- and the 32 lanes serviced together must touch 64 distinct banks."""
    G = 4                                             # two tiles
    S = 32
    plane = np.zeros(S * G * 16, dtype=np.int64)      # element value encodes (sample, feature)
    for s in range(S):
        for f in range(16 * G):
            plane[tile_major_index(s, f, G)] = 1000 * s + f
    # DMA: block k of the image, LDS slot p (16-byte unit) <- global unit src(p, k & 1) of the same block
    lds = np.zeros(G * 512, dtype=np.int64)           # bf16 elements
    for k in range(G):
        for p in range(64):
            ph, pc = p >> 5, p & 31
            src = 32 * ph + (pc ^ (4 * ph + (8 if k & 1 else 0)))
            lds[k * 512 + p * 8: k * 512 + p * 8 + 8] = plane[k * 512 + src * 8: k * 512 + src * 8 + 8]
    for t in range(G // 2):
        for ks in range(2):
            got = np.zeros((64, 8), dtype=np.int64)
            for rd in range(2):
                addr = np.zeros(64, dtype=np.int64)   # byte address of every lane's 8-byte row
                for l in range(64):
                    grp, qq = l >> 4, l & 15
                    rr, mm = qq >> 2, qq & 3
                    par, khalf, hh, jj = grp & 1, grp >> 1, mm & 1, mm >> 1
                    lane_off = par * 1024 + hh * 512 + (rr + 8 * (khalf ^ par)) * 16 + jj * 8
                    addr[l] = t * 2048 + ks * 256 + lane_off + 64 * (hh if rd == 0 else 1 - hh)
                for half in range(2):                 # bank check per group of 32 lanes: 8 bytes each over 64 four-byte banks
                    banks = np.concatenate([[(a // 4) % 64, (a // 4 + 1) % 64] for a in addr[32 * half: 32 * half + 32]])
                    assert len(set(banks.tolist())) == 64
                for l in range(64):                   # the transposing read
                    base = l & ~15
                    i = l & 15
                    for r in range(4):
                        src_lane = base + 4 * r + (i >> 2)
                        got[l, 4 * rd + r] = lds[addr[src_lane] // 2 + (i & 3)]
            for l in range(64):
                for e in range(8):
                    assert got[l, e] == 1000 * (16 * ks + 8 * (l >> 5) + e) + 32 * t + (l & 31), (t, ks, l, e, got[l, e])


def _tile32_index(s, f, W):
    """nnr_layout.h: tile32_index, restated"""
    return (((s >> 5) * (W >> 3) + (f >> 3)) << 8) + ((((f >> 2) & 1) * 32 + (s & 31)) << 2) + (f & 3)


def test_tile_major_fp32_planes_layout_query_and_decode():
    """The stash planes (activations and gradients) of a three-term TRAINING workspace are tile-major fp32 (ABI 4: nnr_ws_plane_layout == 2), every other plane and
    every other mode row-major (0) or the bf16 tiles (1); ops.workspace_plane's decode is the inverse of tile32_index."""
    import torch
    from nnr import lib as L
    lib = L.load()
    prev = L.set_fp32_products("split3")
    try:
        train = L.make_cfg(8, 64, 256, train=True)
        infer = L.make_cfg(8, 64, 256, train=False)
    finally:
        L.set_fp32_products(prev)
    L.set_fp32_products("mfma")
    try:
        mfma = L.make_cfg(8, 64, 256, train=True)
    finally:
        L.set_fp32_products(prev)
    bf16 = L.make_cfg(8, 64, 256, train=True, bf16=True)
    for p in list(range(31, 39)) + [40]:
        assert lib.nnr_ws_plane_layout(C.byref(train), p) == 2
        assert lib.nnr_ws_plane_layout(C.byref(mfma), p) == 0
        assert lib.nnr_ws_plane_layout(C.byref(bf16), p) == 1
        assert lib.nnr_ws_plane_layout(C.byref(infer), p) == -1
    for p in [10] + list(range(11, 19)) + [19, 20]:      # the activation planes: position encoding, h1..h8, direction encoding, colour hidden
        assert lib.nnr_ws_plane_layout(C.byref(train), p) == 2, p
        assert lib.nnr_ws_plane_layout(C.byref(mfma), p) == 0, p
    for p in (0, 1, 2, 3, 4, 25):
        assert lib.nnr_ws_plane_layout(C.byref(train), p) == 0, p
    assert lib.nnr_ws_plane_layout(C.byref(train), 99) == -1
    # decode: fill a fake workspace so that element (s, f) of plane 33 holds 1000 s + f at tile32_index
    from nnr import ops
    pitch = C.c_int32(0)
    off = lib.nnr_ws_plane(C.byref(train), 33, C.byref(pitch))
    W, S = pitch.value, 8 * 64
    ws = torch.zeros(lib.nnr_workspace_floats(C.byref(train)))
    s_idx, f_idx = np.meshgrid(np.arange(S), np.arange(W), indexing="ij")
    flat = np.vectorize(_tile32_index)(s_idx, f_idx, W)
    assert len(np.unique(flat)) == S * W and flat.max() == S * W - 1          # a bijection onto the plane
    ws[off + torch.from_numpy(flat.reshape(-1))] = torch.from_numpy((1000.0 * s_idx + f_idx).reshape(-1)).float()
    got = ops.workspace_plane(train, ws, 33)
    assert torch.equal(got, torch.from_numpy(1000.0 * s_idx + f_idx).float())


def test_three_term_wgrad_tile_major_staging_delivers_the_row_major_operands():
    """The index algebra of wgrad_job_split<.., DTILE = true> (nnr_wgrad.hip), emulated.  A 16-sample step of a tile-major gradient plane
    goes through the eight DMA instructions (instruction j, lane i -> LDS slot 64 j + i; the source: feature quad 16 (j & 1) + (i & 15),
    sample 4 (j >> 1) + (i >> 4)); lane (h, m) then fetches pair P, component C at float 4 (256 h + 64 (m >> 4) + (m & 15)) + 512 (P >> 1) +
    128 (P & 1) + C and + 64.  It must receive samples k + 8 h + 2 P and + 1 of feature col0 + 4 m + C -- what the row-major image hands
    it -- every DMA lane must read a 16-byte unit inside the plane, four consecutive lanes-of-16 must form 64-byte runs, and the 16 lanes
    a read services together must address 16 different slots modulo 16 (the row-major image's bank pattern)."""
    W = 256
    for col0, k in ((0, 0), (128, 16), (128, 48), (0, 1008)):
        S = 1024 + 32
        plane = np.full(S * W, -1.0)
        s_idx, f_idx = np.meshgrid(np.arange(S), np.arange(W), indexing="ij")
        plane[np.vectorize(_tile32_index)(s_idx, f_idx, W)] = (1000.0 * s_idx + f_idx)
        lds = np.full(8 * 64 * 4, np.nan)
        dg = (col0 >> 3) * 256                                  # floats: first block of the tile's columns
        chunk_floats = 32 * W
        for j in range(8):
            starts = []
            for i in range(64):
                dlane = ((i & 15) >> 1) * 256 + (i & 1) * 128 + (i >> 4) * 4            # floats (the kernel: bytes)
                src = dg + (k >> 5) * chunk_floats + (k & 31) * 4 + (j & 1) * 2048 + (j >> 1) * 16 + dlane
                assert 0 <= src and src + 4 <= S * W and src % 4 == 0
                lds[(64 * j + i) * 4:(64 * j + i) * 4 + 4] = plane[src:src + 4]
                starts.append(src)
            for q in range(16):        # lanes q, q + 16, q + 32, q + 48: one 64-byte run
                run = [starts[q + 16 * t] for t in range(4)]
                assert run == [run[0] + 4 * t for t in range(4)], (j, q, run)
        assert not np.isnan(lds).any()
        for lane in range(64):
            h, m = lane >> 5, lane & 31
            base = 4 * (256 * h + 64 * (m >> 4) + (m & 15))
            for P in range(4):
                for Cc in range(4):
                    for second in range(2):
                        v = lds[base + 512 * (P >> 1) + 128 * (P & 1) + Cc + 64 * second]
                        assert v == 1000.0 * (k + 8 * h + 2 * P + second) + col0 + 4 * m + Cc, (lane, P, Cc, second, v)
        for h in range(2):
            for g16 in range(2):       # 16 consecutive lanes of one read
                slots = {((4 * (256 * h + 64 * (m >> 4) + (m & 15))) // 4) % 16 for m in range(16 * g16, 16 * g16 + 16)}
                assert len(slots) == 16


def test_narrow_wgrad_jobs_address_tile_major_planes():
    """wgrad_job's one address formula for both layouts (nnr_wgrad.hip): row of sample s = base + (s >> 5) A + (s & 31) B, columns c ..
    c + W - 1 of a tile-major plane at (c >> 3) 256 + ((c >> 2) & 1) 128 + (c & 3)."""
    W = 256
    for MI, col0 in ((4, 0), (4, 128), (2, 64), (1, 3), (1, 200)):
        for m in (0, 1, 7, 31):
            c = col0 + MI * m
            if c + MI > W:
                continue
            for s in (0, 1, 17, 31, 32, 95, 1000):
                base = (c >> 3) * 256 + ((c >> 2) & 1) * 128 + (c & 3)
                addr = base + (s >> 5) * 32 * W + (s & 31) * 4
                for e in range(MI):
                    assert addr + e == _tile32_index(s, c + e, W), (MI, col0, m, s, e)


def test_shared_split_stages_its_quarter_of_both_tile_major_operands():
    """wgrad_group_split<DTILE, XTILE> (nnr_wgrad.hip), emulated for wave (ta, tb) of a class-A workgroup: of a 16-sample step it stages the
    samples 8 h + 4 tb + {0..3} of gradient half ta (staged rows 0..3) and the samples 8 h + 4 ta + {0..3} of activation half tb (rows 4..7),
    each as four DMA instructions r (source: quad 16 (r & 1) + (i & 15), sample 4 (t + 2 (r >> 1)) + (i >> 4) with t = tb resp. ta), and
    lane (h, m) reads local pair pl, component C at float 4 (128 h + 64 (m >> 4) + (m & 15)) + 128 pl + C (+ 64; + 1024 for the activation
    rows).  Both operands use the same mapping with the roles of ta and tb exchanged; the planes have different widths (the merged layer)."""
    for Wd, Wx, dcol0, xcol0, k in ((256, 256, 0, 128, 32), (256, 256, 128, 0, 1008), (128, 256, 0, 128, 48)):
        S = 1024 + 32
        planes = {}
        for name, W in (("d", Wd), ("x", Wx)):
            pl = np.full(S * W, -1.0)
            s_idx, f_idx = np.meshgrid(np.arange(S), np.arange(W), indexing="ij")
            pl[np.vectorize(_tile32_index)(s_idx, f_idx, W)] = 1000.0 * s_idx + f_idx
            planes[name] = pl
        ta, tb = dcol0 >> 7, xcol0 >> 7
        lds = np.full(8 * 64 * 4, np.nan)
        for name, W, col0, t, row0 in (("d", Wd, dcol0, tb, 0), ("x", Wx, xcol0, ta, 4)):
            base = (col0 >> 3) * 256
            for r in range(4):
                for i in range(64):
                    tlane = ((i & 15) >> 1) * 256 + (i & 1) * 128 + (i >> 4) * 4          # floats (the kernel: bytes)
                    src = base + (k >> 5) * 32 * W + (k & 31) * 4 + (r & 1) * 2048 + (t + 2 * (r >> 1)) * 16 + tlane
                    assert 0 <= src and src + 4 <= S * W
                    slot = 64 * (row0 + r) + i
                    lds[4 * slot:4 * slot + 4] = planes[name][src:src + 4]
        assert not np.isnan(lds).any()
        for lane in range(64):
            h, m = lane >> 5, lane & 31
            lane_base = 4 * (128 * h + 64 * (m >> 4) + (m & 15))
            for name, col0, t, extra in (("d", dcol0, tb, 0), ("x", xcol0, ta, 1024)):
                for pl in range(2):
                    for Cc in range(4):
                        for second in range(2):
                            v = lds[lane_base + extra + 128 * pl + Cc + 64 * second]
                            want = 1000.0 * (k + 8 * h + 4 * t + 2 * pl + second) + col0 + 4 * m + Cc
                            assert v == want, (name, lane, pl, Cc, second, v, want)


def test_wgrad_plan_with_bundles_covers_every_weight_once():
    """NNR_WGRAD_BUNDLES (the narrow tiles scheduled per workgroup, nnr_api.cpp build_plan; off by default, read once per process): the same
    coverage / chaining / balance statements in a process that has it set."""
    import subprocess
    import sys
    env = dict(os.environ, NNR_WGRAD_BUNDLES="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "test_wgrad_plan_covers_every_weight_once"], env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-2000:]
