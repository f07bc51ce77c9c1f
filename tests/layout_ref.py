"""numpy restatement of the device data layouts (nope-nerf_amd/csrc/nnr_layout.h): the packed A-fragment order and the
register layout of activations, plus an emulation of v_mfma_f32_32x32x2_f32's operand/result mapping as documented in
the CDNA4 guide (A: lane l holds A[l&31][l>>5]; B: lane l holds B[l>>5][l&31]; D register rho of lane l is
D[(rho&3)+8*(rho>>2)+4*(l>>5)][l&31]).  Used by CPU tests to prove the index algebra closes (pack -> chained layers
-> wgrad tiles reproduce plain matmuls) and by the GPU tests to check the pack kernel bit-exactly."""
import numpy as np

LAYERS = ("layers0.0", "layers0.2", "layers0.4", "layers0.6", "layers1.0", "layers1.2", "layers1.4", "layers1.6",
          "fc_density", "fc_feature", "rgb_layers.0", "fc_rgb")


MERGED = 12   # pseudo-layer: the feature layer folded into the colour-hidden layer (nnr_layout.h)


def merged(weights, biases, D):
    """W' = Wg[:, :D] Wf and b' = Wg[:, :D] bf + bg; float64 accumulation rounded once (the device uses an fp32 fma chain:
    equal to ~1e-7, compared with a tolerance)."""
    wg1 = weights[10][:, :D].astype(np.float64)
    return ((wg1 @ weights[9].astype(np.float64)).astype(np.float32),
            (wg1 @ biases[9].astype(np.float64) + biases[10].astype(np.float64)).astype(np.float32))


def fwd_parts(D):
    """(layer, transpose, KT, MT, m_real, k_real, moff, koff) in stream order -- mirrors Layout<D>::fwd."""
    DT, HT, Dh = D // 32, D // 64, D // 2
    parts = [(0, 0, 2, HT, Dh, 63, h * Dh, 0) for h in (0, 1)]
    for l in (1, 2, 3):
        parts += [(l, 0, DT, HT, Dh, D, h * Dh, 0) for h in (0, 1)]
    for h in (0, 1):
        parts += [(4, 0, DT, HT, Dh, D, h * Dh, 0), (4, 0, 2, HT, Dh, 63, h * Dh, D)]
    for l in (5, 6, 7):
        parts += [(l, 0, DT, HT, Dh, D, h * Dh, 0) for h in (0, 1)]
    parts += [(MERGED, 0, DT, HT, Dh, D, 0, 0), (10, 0, 1, HT, Dh, 27, 0, D)]   # W' = Wg[:, :D] Wf, then the direction columns
    return parts


def bwd_parts(D):
    DT, HT, Dh = D // 32, D // 64, D // 2
    parts = [(MERGED, 1, HT, HT, Dh, Dh, h * Dh, 0) for h in (0, 1)] + [(10, 1, HT, 1, 27, Dh, D, 0)]
    for l in (7, 6, 5):
        parts += [(l, 1, DT, HT, Dh, D, h * Dh, 0) for h in (0, 1)]
    parts += [(4, 1, DT, 2, 63, D, D, 0)] + [(4, 1, DT, HT, Dh, D, h * Dh, 0) for h in (0, 1)]
    for l in (3, 2, 1):
        parts += [(l, 1, DT, HT, Dh, D, h * Dh, 0) for h in (0, 1)]
    parts += [(0, 1, DT, 2, 63, D, 0, 0)]
    return parts


def bias_pads(D):
    return [D] * 8 + [32, D, (D // 2 + 31) // 32 * 32, 32]


def part_matrix(W, part):
    """Dense zero-padded A[32*MT][32*KT] of a part."""
    layer, tr, KT, MT, m_real, k_real, moff, koff = part
    A = np.zeros((32 * MT, 32 * KT), dtype=np.float32)
    if tr:
        A[:m_real, :k_real] = W[koff:koff + k_real, moff:moff + m_real].T
    else:
        A[:m_real, :k_real] = W[moff:moff + m_real, koff:koff + k_real]
    return A


PANEL_FRAGS = 32


def pack_part(A, KT, MT):
    """Panelised fragments of one part: [n_panels][32 slots][64 lanes][4]; k-group g lives in panel g//GP (GP = 32//MT),
    slot (g % GP)*MT + mt; lane l holds A[32*mt + (l&31)][8*g + 4*(l>>5) + i].  Unused slots are zero."""
    gp = PANEL_FRAGS // MT
    n_panels = (4 * KT + gp - 1) // gp
    out = np.zeros((n_panels, PANEL_FRAGS, 64, 4), dtype=np.float32)
    lane = np.arange(64)
    for g in range(4 * KT):
        for mt in range(MT):
            rows = 32 * mt + (lane & 31)
            for i in range(4):
                out[g // gp, (g % gp) * MT + mt, :, i] = A[rows, 8 * g + 4 * (lane >> 5) + i]
    return out.reshape(-1)


# ---- MODE 2 of nnr_layout.h: every weight as three bf16 terms (NNR_F_SPLIT3) ----------------------------------------------------------
SPLIT_PANEL_FRAGS = 24


def bf16_rn(x):
    """fp32 -> the nearest bf16 (ties to even), returned as fp32 (low 16 bits zero)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7fff + ((u >> 16) & 1)) & 0xffff0000).astype(np.uint32)
    return r.view(np.float32)


def split3(x):
    """(l, m, h): h = rn(x), m = rn(x - h), l = rn(x - h - m); the differences are exact in fp32 (csrc/nnr_split.h, nnr_pack.hip)."""
    x = np.asarray(x, dtype=np.float32)
    h = bf16_rn(x)
    r1 = (x - h).astype(np.float32)
    m = bf16_rn(r1)
    l = bf16_rn((r1 - m).astype(np.float32))
    return l, m, h


def pack_part_split(A, KT, MT):
    """Panelised fragments of one part in MODE 2, as the fp32 words the pack kernel writes: [n_panels][24 slots][64 lanes][4 words];
    fragment row b = 16 k-values lives in panel b // GP (GP = 8 // MT), term t (0 = l, 1 = m, 2 = h) and m-tile mt in slot
    ((b % GP) * 3 + t) * MT + mt; lane l holds 8 bf16: A[32 mt + (l & 31)][16 b + 8 (i >> 2) + 4 (l >> 5) + (i & 3)], i = 0..7, two per
    word (even i in the low half)."""
    gp = SPLIT_PANEL_FRAGS // (3 * MT)
    rows = 2 * KT
    n_panels = (rows + gp - 1) // gp
    out = np.zeros((n_panels, SPLIT_PANEL_FRAGS, 64, 4), dtype=np.uint32)
    lane = np.arange(64)
    terms = split3(A)
    for b in range(rows):
        for t in range(3):
            for mt in range(MT):
                r = 32 * mt + (lane & 31)
                for i in range(8):
                    v = terms[t][r, 16 * b + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)]
                    bits = (v.view(np.uint32) >> 16).astype(np.uint32)
                    out[b // gp, ((b % gp) * 3 + t) * MT + mt, :, i // 2] |= bits << (16 * (i & 1))
    return out.reshape(-1).view(np.float32)


# ---- MODE 3 of nnr_layout.h: every weight, scaled by its slot's power of two, as two fp16 terms in three fragment classes (NNR_F_SPLIT2) ------
def pow2_scale(mx):
    """scale_kernel (nnr_pack.hip): the power of two s with mx s in [2^13, 2^14); 1 for an all-zero (or non-finite) tensor."""
    mx = np.float32(mx)
    if not (mx > 0 and mx < 3.0e38):
        return np.float32(1.0)
    _, e = np.frexp(mx)                      # mx = f 2^e, f in [0.5, 1)
    return np.float32(2.0 ** int(np.clip(14 - int(e), -100, 100)))


def scale_slot(layer):
    return layer if layer < 8 else 8


def split2_classes(x):
    """(m, h) of an already scaled fp32 array: h = rn16(x), m = rn16(x - h) (exact difference) -- the fragment classes 0, 1 of
    csrc/nnr_split2.h, as float16 arrays (numpy's conversion rounds to nearest even, subnormals included)."""
    x = np.asarray(x, dtype=np.float32)
    h = x.astype(np.float16)
    m = (x - h.astype(np.float32)).astype(np.float32).astype(np.float16)
    return m, h


def pack_part_split2(A_scaled, KT, MT):
    """MODE 3: 32-slot panels, GP = 16 // MT rows of 16 k-values each, class c (0 = m, 1 = h) in slot ((b % GP) * 2 + c) * MT + mt; lane l holds
    8 fp16 of A_scaled[32 mt + (l & 31)][16 b + 8 (i >> 2) + 4 (l >> 5) + (i & 3)], two per word (even i in the low half)."""
    gp = PANEL_FRAGS // (2 * MT)
    rows = 2 * KT
    n_panels = (rows + gp - 1) // gp
    out = np.zeros((n_panels, PANEL_FRAGS, 64, 4), dtype=np.uint32)
    lane = np.arange(64)
    with np.errstate(over="ignore"):
        terms = split2_classes(A_scaled)
    for b in range(rows):
        for t in range(2):
            for mt in range(MT):
                r = 32 * mt + (lane & 31)
                for i in range(8):
                    v = np.ascontiguousarray(terms[t][r, 16 * b + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)])
                    out[b // gp, ((b % gp) * 2 + t) * MT + mt, :, i // 2] |= v.view(np.uint16).astype(np.uint32) << (16 * (i & 1))
    return out.reshape(-1).view(np.float32)


def gemm_part_split2_emulated(packed_part, in_regs, KT, MT):
    """What nnr_split2.h's gemm_part2 computes from a MODE 3 packed part and the fp32 input registers [16 KT][64 lanes] (unscaled activations):
    per row of 16 k-values the three products (w_m, x_h) (w_h 2^-11, x_m') (w_h, x_h) of v_mfma_f32_32x32x16_f16, x_h = rn16(x), x_m' = rn16((x - x_h)
    2^11), the middle weight operand made from the h fragment by an fp16 multiply, fp32 accumulation.  Returns [MT][16][64] accumulator registers
    (still carrying the weights' scale)."""
    gp = PANEL_FRAGS // (2 * MT)
    pan = packed_part.view(np.uint32).reshape(-1, PANEL_FRAGS, 64, 4)
    lane = np.arange(64)
    acc = np.zeros((MT, 16, 64), dtype=np.float32)

    def unpack(words):       # [64][4] uint32 -> [64][8] float64
        o = np.zeros((64, 8), dtype=np.float64)
        for i in range(8):
            o[:, i] = ((words[:, i // 2] >> (16 * (i & 1))) & 0xffff).astype(np.uint16).view(np.float16).astype(np.float64)
        return o
    for b in range(2 * KT):
        x = in_regs[8 * b:8 * b + 8].T.astype(np.float32)                      # [64][8]
        xh = x.astype(np.float16)
        xm = ((x - xh.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
        for wt, xs, down in ((0, xh, False), (1, xm, True), (1, xh, False)):
            for mt in range(MT):
                a = unpack(pan[b // gp, ((b % gp) * 2 + wt) * MT + mt])
                if down:      # v_pk_mul_f16 by 2^-11: exact unless the result is subnormal
                    a = (a.astype(np.float16) * np.float16(2.0 ** -11)).astype(np.float64)
                A = np.zeros((32, 16), dtype=np.float64)
                B = np.zeros((16, 32), dtype=np.float64)
                for i in range(8):
                    A[lane & 31, 8 * (lane >> 5) + i] = a[:, i]
                    B[8 * (lane >> 5) + i, lane & 31] = xs[:, i].astype(np.float64)
                Dm = A @ B
                for rho in range(16):
                    acc[mt, rho] = (acc[mt, rho] + Dm[(rho & 3) + 8 * (rho >> 2) + 4 * (lane >> 5), lane & 31]).astype(np.float32)
    return acc


def gemm_part_split_emulated(packed_part, in_regs, KT, MT):
    """What nnr_split.h's gemm_part computes from a MODE 2 packed part and the fp32 activation registers [16 KT][64 lanes]: per row of 16
    k-values the six term products (l,h) (m,m) (m,h) (h,l) (h,m) (h,h) of v_mfma_f32_32x32x16_bf16 -- lane l supplies A[l & 31][k = 8 (l >> 5)
    + i] and B[k][l & 31]; D register rho of lane l is D[(rho & 3) + 8 (rho >> 2) + 4 (l >> 5)][l & 31] -- accumulated in fp32.
    Returns [MT][16][64] accumulator registers."""
    gp = SPLIT_PANEL_FRAGS // (3 * MT)
    pan = packed_part.view(np.uint32).reshape(-1, SPLIT_PANEL_FRAGS, 64, 4)
    lane = np.arange(64)
    acc = np.zeros((MT, 16, 64), dtype=np.float32)

    def unpack(words):       # [64][4] uint32 -> [64][8] fp32
        o = np.zeros((64, 8), dtype=np.float32)
        for i in range(8):
            o[:, i] = (((words[:, i // 2] >> (16 * (i & 1))) & 0xffff) << 16).astype(np.uint32).view(np.float32)
        return o
    for b in range(2 * KT):
        xs = split3(in_regs[8 * b:8 * b + 8].T)      # each [64][8]: the lane's 8 k-values of the row, as three terms
        for wt, xt in ((0, 2), (1, 1), (1, 2), (2, 0), (2, 1), (2, 2)):
            for mt in range(MT):
                a = unpack(pan[b // gp, ((b % gp) * 3 + wt) * MT + mt])       # lane -> A[l & 31][8 (l >> 5) + i]
                A = np.zeros((32, 16), dtype=np.float64)
                B = np.zeros((16, 32), dtype=np.float64)
                for i in range(8):
                    A[lane & 31, 8 * (lane >> 5) + i] = a[:, i]
                    B[8 * (lane >> 5) + i, lane & 31] = xs[xt][:, i]
                Dm = A @ B
                for rho in range(16):
                    acc[mt, rho] = (acc[mt, rho] + Dm[(rho & 3) + 8 * (rho >> 2) + 4 * (lane >> 5), lane & 31]).astype(np.float32)
    return acc


def part_frags(packed_part, KT, MT):
    """Inverse view of pack_part: [4*KT][MT][64][4]."""
    gp = PANEL_FRAGS // MT
    pan = packed_part.reshape(-1, PANEL_FRAGS, 64, 4)
    out = np.zeros((4 * KT, MT, 64, 4), dtype=packed_part.dtype)
    for g in range(4 * KT):
        for mt in range(MT):
            out[g, mt] = pan[g // gp, (g % gp) * MT + mt]
    return out


def head_tables(weights, D):
    """Density and rgb rows in register order [half][r] (feature = frag_feature(r, half))."""
    DT, HT = D // 32, D // 64
    out = []
    for W, rows, nreg in ((weights[8], 1, 16 * DT), (weights[11], 3, 16 * HT)):
        for row in range(rows):
            for h in (0, 1):
                out.append(np.array([W[row, frag_feature(r, h)] for r in range(nreg)], dtype=np.float32))
    return np.concatenate(out)


def pack_all(weights, biases, D, with_exact_mask=False, mode=0):
    """weights/biases: lists of 12 numpy arrays in state_dict order -> the packed buffer nnr_pack_weights produces (mode 0: fp32
    fragments, mode 2: three bf16 terms per weight, NNR_F_SPLIT3; mode 3: two fp16 terms of the scaled weight in three classes, NNR_F_SPLIT2).
    with_exact_mask: also a bool array, False where the value derives from the merged matrix (compare with a tolerance)."""
    wm, bm = merged(weights, biases, D)
    w13 = list(weights) + [wm]
    chunks, exact = [], []
    def put(v, is_exact):
        chunks.append(v)
        exact.append(np.full(v.size, is_exact))
    scale = np.ones(16, dtype=np.float32)
    if mode == 3:
        for l in range(8):
            scale[l] = pow2_scale(np.abs(weights[l]).max())
        scale[8] = pow2_scale(max(np.abs(wm).max(), np.abs(weights[10]).max()))
    sc_of = lambda layer: scale[scale_slot(layer)] if mode == 3 else np.float32(1.0)
    for part in fwd_parts(D) + bwd_parts(D):
        A = part_matrix(w13[part[0]], part)
        if mode == 3:
            put(pack_part_split2(A * sc_of(part[0]), part[2], part[3]), part[0] != MERGED)
        else:
            put((pack_part_split if mode == 2 else pack_part)(A, part[2], part[3]), part[0] != MERGED)
    for l, (b, pad) in enumerate(zip(biases, bias_pads(D))):
        v = np.zeros(pad, dtype=np.float32)
        src = bm if l == 10 else b          # the colour-hidden slot holds the merged bias
        v[:src.size] = src * (sc_of(l) if (l < 8 or l == 10) else np.float32(1.0))      # mode 3: the accumulators start at s_w bias
        put(v, l != 10)
    put(head_tables(weights, D), True)
    if mode == 3:      # [16] scales by slot ([9]: max |w_sigma|, not a scale), [16] inverses
        tab = np.zeros(32, dtype=np.float32)
        tab[:9] = scale[:9]
        tab[9] = np.abs(weights[8]).max()
        tab[16:25] = np.float32(1.0) / scale[:9]
        put(tab, True)
    put(wm.reshape(-1), False)
    put(bm, False)
    put(weights[9].reshape(-1), True)       # copies for the un-merge step
    put(np.ascontiguousarray(weights[10][:, :D]).reshape(-1), True)
    put(biases[9], True)
    out = np.concatenate(chunks)
    return (out, np.concatenate(exact)) if with_exact_mask else out


# ---- register layout + MFMA emulation -------------------------------------------------------------------------------
def frag_feature(r, h):
    return 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * h


def to_regs(X):
    """X[features 32*T][32 samples] -> regs[16*T][64 lanes]."""
    T = X.shape[0] // 32
    regs = np.zeros((16 * T, 64), dtype=X.dtype)
    for r in range(16 * T):
        for l in range(64):
            regs[r, l] = X[frag_feature(r, l >> 5), l & 31]
    return regs


def from_regs(regs):
    T = regs.shape[0] // 16
    X = np.zeros((32 * T, 32), dtype=regs.dtype)
    for r in range(16 * T):
        for l in range(64):
            X[frag_feature(r, l >> 5), l & 31] = regs[r, l]
    return X


def mfma32(a, b, c):
    """a, b: (64,) per-lane operands; c: (16, 64) accumulator registers. Returns the updated accumulator."""
    A = np.zeros((32, 2), dtype=np.float64)
    B = np.zeros((2, 32), dtype=np.float64)
    lane = np.arange(64)
    A[lane & 31, lane >> 5] = a
    B[lane >> 5, lane & 31] = b
    Dm = A @ B
    out = c.copy()
    for rho in range(16):
        rows = (rho & 3) + 8 * (rho >> 2) + 4 * (lane >> 5)
        out[rho] += Dm[rows, lane & 31]
    return out


def gemm_part_emulated(packed_part, in_regs, KT, MT):
    """Mirror of nnr_device.h::gemm_part: returns acc[MT][16][64]."""
    frag = part_frags(packed_part, KT, MT)
    acc = np.zeros((MT, 16, 64), dtype=np.float64)
    for g in range(4 * KT):
        for i in range(4):
            for mt in range(MT):
                acc[mt] = mfma32(frag[g, mt, :, i], in_regs[4 * g + i], acc[mt])
    return acc
