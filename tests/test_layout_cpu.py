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


def test_chained_layers_in_register_layout():
    """Two chained layers computed with emulated MFMAs on packed fragments == plain matmuls; the output registers of
    one layer are directly the B operands of the next (no transpose)."""
    D = 128
    W, _ = rand_weights(D)
    rng = np.random.default_rng(1)
    e = np.zeros((64, 32), dtype=np.float32)
    e[:63] = rng.standard_normal((63, 32))
    parts = lr.fwd_parts(D)
    p1 = lr.pack_part(lr.part_matrix(W[0], parts[0]), parts[0][2], parts[0][3])
    acc = lr.gemm_part_emulated(p1, lr.to_regs(e), parts[0][2], parts[0][3])
    h1 = np.maximum(acc.reshape(-1, 64), 0)                       # registers of hidden 1 (16*DT, 64)
    np.testing.assert_allclose(lr.from_regs(h1), np.maximum(W[0] @ e[:63], 0), rtol=1e-5, atol=1e-5)
    p2 = lr.pack_part(lr.part_matrix(W[1], parts[1]), parts[1][2], parts[1][3])
    acc2 = lr.gemm_part_emulated(p2, h1, parts[1][2], parts[1][3])
    np.testing.assert_allclose(lr.from_regs(acc2.reshape(-1, 64)), W[1] @ np.maximum(W[0] @ e[:63], 0), rtol=1e-4, atol=1e-4)


def test_transposed_parts_and_skip_layer():
    D = 128
    W, _ = rand_weights(D)
    rng = np.random.default_rng(2)
    d5 = rng.standard_normal((D, 32)).astype(np.float32)
    ref = W[4].T @ d5
    for idx, rows in ((9, slice(0, D)), (8, slice(D, D + 63))):  # hidden 5 transposed: rows [h4 (D)] and [posenc (63)]
        part = lr.bwd_parts(D)[idx]
        pk = lr.pack_part(lr.part_matrix(W[4], part), part[2], part[3])
        acc = lr.gemm_part_emulated(pk, lr.to_regs(d5), part[2], part[3])
        full = lr.from_regs(acc.reshape(-1, 64))
        n = rows.stop - rows.start
        np.testing.assert_allclose(full[:n], ref[rows], rtol=1e-4, atol=1e-4)
        assert np.all(full[n:] == 0)
    part = lr.fwd_parts(D)[5]                                      # skip layer, posenc part: columns D.. of layers1.0
    A = lr.part_matrix(W[4], part)
    np.testing.assert_array_equal(A[:, :63], W[4][:, D:])
    assert np.all(A[:, 63] == 0)


def test_library_exports_every_declared_symbol():
    from nnr import lib as L
    hdr = open(os.path.join(ROOT, "include", "nnr.h")).read()
    declared = set(re.findall(r"\b(nnr_[a-z_0-9]+)\s*\(", hdr)) - {"nnr_cfg", "nnr_params", "nnr_param_grads"}
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    lib = L.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nnr_abi_version() == 1
    assert lib.nnr_strerror(-2).decode().startswith("unsupported")


@pytest.mark.parametrize("D", [128, 256])
def test_sizes_and_error_codes(D):
    from nnr import lib as L
    lib = L.load()
    W, B = rand_weights(D)
    cfg = L.make_cfg(16, 64, D, train=True)
    assert lib.nnr_packed_floats(C.byref(cfg)) == lr.pack_all(W, B, D).size
    S_pad = 16 * 64
    x_width = 64 + 8 * D + (D + 32) + D // 2
    d_width = 8 * D + D + D // 2
    expect = S_pad * (4 + 1 + 4 + 4 + 4 + x_width + d_width + 9 * 2 * (D // 64))
    assert lib.nnr_workspace_floats(C.byref(cfg)) == expect
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
    jobs = [j for j in plan_jobs(cfg) if j.layer >= 0]
    S_pad = (R * N + 127) // 128 * 128
    shapes = [(D, 63), (D, D), (D, D), (D, D), (D, D + 63), (D, D), (D, D), (D, D), (1, D), (D, D), (D // 2, D + 27), (3, D // 2)]
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
        if j.bias:
            bias_cover[j.layer][rows] += j.k1 - j.k0
    for l in range(12):
        assert np.all(cover[l] == S_pad), (l, np.unique(cover[l]))
        assert np.all(bias_cover[l] == S_pad), l
    assert len(plan_jobs(cfg)) % 4 == 0
