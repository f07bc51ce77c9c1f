// nnr_pack.hip -- re-pack the 12 nn.Linear tensors of OfficialStaticNerf (model/official_nerf.py:20-37, (out,in)
// row-major) into MFMA A-fragment order (nnr_layout.h), once forward-oriented (A = W) and once transposed (A = W^T, for
// the input-gradient chain), grouped into the 32 KiB panels the MLP kernels DMA into LDS, plus zero-padded biases.
// One launch, ~5 MB written; runs after every optimiser step.
#include "nnr_device.h"
#include "nnr_kernels.h"

namespace nnr {

// W' = Wg[:, :D] Wf and b' = Wg[:, :D] bf + bg (nnr_layout.h), plus the copies the un-merge step of the weight-gradient pass
// reads.  Products accumulated in index order with fma (one chain per element: the value is defined by that order and pinned bit for
// bit by tests/layout_ref.py).  Round 4: a workgroup computes a 16 x 16 tile of W' from LDS copies of its 16 rows of Wg and 16 columns of
// Wf -- the same chains, but every operand is fetched from memory once per tile instead of once per element (one thread per element with
// both operands from L2 was 16 us of latency); the copies and b' run in the workgroups behind the tiles.
template <int D, int BF16>
__global__ __launch_bounds__(256) void merge_kernel(PackArgs a) {
    using L = Layout<D, BF16>;
    const float* Wf = a.w[9];
    const float* Wg = a.w[10];
    constexpr int ldg = D + kDirReal;
    constexpr int kTiles = (L::Dh / 16) * (D / 16);
    __shared__ float wg_s[16][D + 1];      // rows m0 .. m0 + 15 of Wg[:, :D] (+1: the 16 rows a wave reads sit in different banks)
    __shared__ float wf_s[D][16];          // columns k0 .. k0 + 15 of Wf
    if ((int)blockIdx.x < kTiles) {
        const int m0 = 16 * ((int)blockIdx.x / (D / 16)), k0 = 16 * ((int)blockIdx.x % (D / 16));
        for (int i = threadIdx.x; i < 16 * D; i += 256) {
            wg_s[i / D][i % D] = Wg[(m0 + i / D) * ldg + i % D];
            wf_s[i / 16][i % 16] = Wf[(i / 16) * D + k0 + i % 16];
        }
        __syncthreads();
        const int mi = threadIdx.x >> 4, ki = threadIdx.x & 15;
        float acc = 0.f;
#pragma unroll 16
        for (int j = 0; j < D; ++j) acc = fmaf(wg_s[mi][j], wf_s[j][ki], acc);
        const int gid = (m0 + mi) * D + k0 + ki;
        a.packed[L::merged_w_off + gid] = acc;
        a.packed[L::copy_wg_off + gid] = wg_s[mi][k0 + ki];
        return;
    }
    const int gid = ((int)blockIdx.x - kTiles) * 256 + threadIdx.x, n_rest = ((int)gridDim.x - kTiles) * 256;
    if (gid < L::Dh) {
        float acc = a.b[10][gid];
#pragma unroll 16
        for (int j = 0; j < D; ++j) acc = fmaf(Wg[gid * ldg + j], a.b[9][j], acc);
        a.packed[L::merged_b_off + gid] = acc;
    }
    if (gid < D) a.packed[L::copy_bf_off + gid] = a.b[9][gid];
    for (int i = gid; i < D * D; i += n_rest) a.packed[L::copy_wf_off + i] = Wf[i];
}

// MODE 3 (nnr_layout.h): the power-of-two scale of every scale slot -- the largest |w| of the slot's tensors times s lies in [2^13, 2^14) -- and its
// inverse.  One workgroup per slot; slots 0..7 = hidden 1..8, slot 8 = the merged colour matrix W' (formed by merge_kernel before this launch)
// together with all of param 10 (a superset of its direction columns).  An all-zero slot gets s = 1.
template <int D>
__global__ __launch_bounds__(1024) void scale_kernel(PackArgs a) {
    using L = Layout<D, 3>;
    const int slot = blockIdx.x;
    float m = 0.f;
    // (1024 lanes x 16-byte loads: the kernel runs in front of every forward; with 256 lanes and scalar loads the skip layer's 82 k weights
    // took 53 us, 2 % of the training step -- profiles/r06/o_bench_headline_only_kernel_stats.csv.  Every tensor's size is a multiple of 4
    // and the merged matrix sits at a multiple of 4 floats; a parameter the caller keeps at an odd offset of some flat buffer takes scalar loads.)
    static_assert(L::merged_w_off % 4 == 0 && (L::Dh * D) % 4 == 0 && (D * kPosReal) % 4 == 0 && (L::Dh * (D + kDirReal)) % 4 == 0, "16-byte loads");
    auto scan = [&](const float* w, int n) {
        if ((reinterpret_cast<uintptr_t>(w) & 15) != 0) {
            for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(w[i]));
            return;
        }
        const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
        for (int i = threadIdx.x; i < n / 4; i += 1024) {
            const f32x4 v = w4[i];
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    };
    if (slot < 8) {
        const int in = slot == 0 ? kPosReal : (slot == 4 ? D + kPosReal : D);
        scan(a.w[slot], D * in);
    } else if (slot == 9) {      // not a scale: the largest |w_sigma| (the density row enters the input-gradient chain as a rank-1 term, nnr_mlp_dgrad_f16.hip)
        scan(a.w[8], D);
    } else {
        scan(a.packed + L::merged_w_off, L::Dh * D);
        scan(a.w[10], L::Dh * (D + kDirReal));
    }
    __shared__ float red[1024];
    red[threadIdx.x] = m;
    __syncthreads();
    for (int d = 512; d >= 1; d >>= 1) {
        if ((int)threadIdx.x < d) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + d]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int e = 0;
        const float mx = red[0];
        float s = 1.f;
        if (mx > 0.f && mx < 3.0e38f) {
            (void)frexpf(mx, &e);            // mx = f 2^e, f in [0.5, 1)
            e = 14 - e;
            e = e > 100 ? 100 : (e < -100 ? -100 : e);
            s = ldexpf(1.f, e);
        }
        a.packed[L::scale_off + slot] = slot == 9 ? mx : s;
        a.packed[L::scale_off + 16 + slot] = slot == 9 ? 0.f : 1.f / s;
        if (slot == 9)      // the unused entries of the two tables: defined values (the buffer is the caller's, uninitialised)
            for (int i = 10; i < 16; ++i) a.packed[L::scale_off + i] = a.packed[L::scale_off + 16 + i] = 0.f;
    }
}

// MODE (= BF16 below): Layout<D, MODE> -- 0 fp32 fragments, 1 bf16 fragments, 2 three bf16 TERMS per weight (l, m, h fragments per row),
// 3 two fp16 terms of the SCALED weight in two fragment classes (m, h)
template <int D, int BF16>
__global__ __launch_bounds__(256) void pack_kernel(PackArgs a) {
    using L = Layout<D, BF16>;
    constexpr int kFrags = mode_panel_frags(BF16);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per packed float4
    const int64_t n_frag4 = L::bias_base / 4;
    if (gid < n_frag4) {
        const int panel = (int)(gid / (kFrags * 64));   // global panel index (forward stream, then backward stream)
        const int slot = (int)((gid / 64) % kFrags);
        const int lane = (int)(gid & 63);
        // locate the part that owns this panel
        PartDesc pd{};
        int base = 0;
        bool found = false;
#pragma unroll
        for (int p = 0; p < F_NPARTS; ++p) {
            const PartDesc d = L::fwd(p);
            const int np = mode_panels(d.KT, d.MT, BF16);
            if (!found && panel < base + np) { pd = d; found = true; }
            if (!found) base += np;
        }
#pragma unroll
        for (int p = 0; p < B_NPARTS; ++p) {
            const PartDesc d = L::bwd(p);
            const int np = mode_panels(d.KT, d.MT, BF16);
            if (!found && panel < base + np) { pd = d; found = true; }
            if (!found) base += np;
        }
        const int gp = mode_gp(pd.MT, BF16);
        constexpr int kTerms = BF16 == 2 ? 3 : (BF16 == 3 ? 2 : 1);           // fragments per row and m-tile
        const int g = (panel - base) * gp + slot / (kTerms * pd.MT);   // fragment row: k-group (fp32) or double k-group (bf16 modes)
        const int term = (slot / pd.MT) % kTerms;          // MODE 2: 0 = l, 1 = m, 2 = h (the order the kernels consume them in)
        const int mt = slot % pd.MT;
        const bool live = slot < gp * kTerms * pd.MT && g < mode_rows(pd.KT, BF16);
        const int m = 32 * mt + (lane & 31);
        const float* W = a.w[pd.layer];
        auto elem = [&](int k) -> float {
            if (live && m < pd.m_real && k < pd.k_real)
                return pd.transpose ? W[(int64_t)(pd.koff + k) * pd.ld + pd.moff + m] : W[(int64_t)(pd.moff + m) * pd.ld + pd.koff + k];
            return 0.f;
        };
        if constexpr (BF16 == 3) {   // the fragment of ONE class of the scaled weight ws = w s: 1 = h = rn16(ws), 0 = m = rn16(ws - h) (exact difference)
            typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
            const float sc = a.packed[L::scale_off + scale_slot(pd.layer)];
            f16x8 q;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float w = elem(16 * g + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3)) * sc;
                const _Float16 h = (_Float16)w;
                q[i] = term == 1 ? h : (_Float16)(w - (float)h);
            }
            reinterpret_cast<f16x8*>(a.packed)[gid] = q;
        } else if constexpr (BF16 == 2) {   // the fragment of ONE term: h = rn(w), m = rn(w - h), l = rn(w - h - m), differences exact in fp32
            bf16x8 q;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float w = elem(16 * g + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3));
                const __bf16 h = (__bf16)w;
                const float r1 = w - (float)h;
                const __bf16 m = (__bf16)r1;
                const __bf16 l = (__bf16)(r1 - (float)m);
                q[i] = term == 2 ? h : (term == 1 ? m : l);
            }
            reinterpret_cast<bf16x8*>(a.packed)[gid] = q;
        } else if constexpr (BF16 == 1) {   // 8 bf16: k = 16g + 4h + i, then 16g + 8 + 4h + i (nnr_layout.h)
            bf16x8 q;
#pragma unroll
            for (int i = 0; i < 8; ++i) q[i] = (__bf16)elem(16 * g + 8 * (i >> 2) + 4 * (lane >> 5) + (i & 3));
            reinterpret_cast<bf16x8*>(a.packed)[gid] = q;
        } else {
            f32x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = elem(8 * g + 4 * (lane >> 5) + i);
            reinterpret_cast<f32x4*>(a.packed)[gid] = v;
        }
    } else {
        const int64_t bi = (gid - n_frag4);  // one thread per float of the bias / head-table tail
        if (bi < L::bias_floats) {
            int64_t o = 0;
#pragma unroll
            for (int l = 0; l < 12; ++l) {
                const int pad = L::bias_pad(l);
                if (bi >= o && bi < o + pad) {
                    const int j = (int)(bi - o);
                    // the colour-hidden slot holds the merged bias b'; MODE 3: the biases of the MFMA layers times their weight scale
                    float sc = 1.f;
                    if constexpr (BF16 == 3) sc = (l < 8 || l == 10) ? a.packed[L::scale_off + scale_slot(l)] : 1.f;
                    a.packed[L::bias_base + bi] = j < L::bias_real(l) ? (l == 10 ? a.b[kMergedLayer][j] : a.b[l][j]) * sc : 0.f;
                }
                o += pad;
            }
        } else if (bi < L::bias_floats + L::head_floats) {
            // head tables in register order: value for (half h, register r) belongs to feature 32t + (rho&3) + 8(rho>>2) + 4h
            const int t = (int)(bi - L::bias_floats);
            const int nsig = 2 * 16 * L::DT, nrow = 2 * 16 * L::HT;
            int row, rem, nreg;
            const float* W;
            int ld;
            if (t < nsig) { row = 0; rem = t; nreg = 16 * L::DT; W = a.w[8]; ld = D; }
            else { row = (t - nsig) / nrow; rem = (t - nsig) % nrow; nreg = 16 * L::HT; W = a.w[11]; ld = D / 2; }
            const int h = rem / nreg, r = rem % nreg;
            const int f = 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * h;
            a.packed[L::head_base + t] = W[(int64_t)row * ld + f];
        }
    }
}

template <int D, int BF16>
static hipError_t launch(const PackArgs& a0, hipStream_t st) {
    using L = Layout<D, BF16>;
    PackArgs a = a0;
    a.w[kMergedLayer] = a.packed + L::merged_w_off;
    a.b[kMergedLayer] = a.packed + L::merged_b_off;
    hipLaunchKernelGGL((merge_kernel<D, BF16>), dim3((L::Dh / 16) * (D / 16) + 32), dim3(256), 0, st, a);      // tiles of W', then 32 workgroups of copies
    if constexpr (BF16 == 3) hipLaunchKernelGGL((scale_kernel<D>), dim3(kScaleSlots + 1), dim3(1024), 0, st, a);
    const int64_t threads = L::bias_base / 4 + L::table_floats;
    dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    hipLaunchKernelGGL((pack_kernel<D, BF16>), grid, block, 0, st, a);
    return hipGetLastError();
}

hipError_t launch_pack(int D, const PackArgs& a, int mode, hipStream_t st) {
    if (mode == 3) return D == 256 ? launch<256, 3>(a, st) : launch<128, 3>(a, st);
    if (mode == 2) return D == 256 ? launch<256, 2>(a, st) : launch<128, 2>(a, st);
    if (mode == 1) return D == 256 ? launch<256, 1>(a, st) : launch<128, 1>(a, st);
    return D == 256 ? launch<256, 0>(a, st) : launch<128, 0>(a, st);
}

}  // namespace nnr
