// nnr_mlp_fwd_ws.hip -- the three-term (NNR_F_SPLIT3) forward MLP with TWO SPECIALISED WAVES PER SIMD, gfx950 only, D = 256.
// Restates, per sample, what nnr_mlp_fwd.hip restates: model/rendering.py:184-195 (z, points, view direction) and
// model/official_nerf.py:60-96 (the MLP); inference with the compositing of model/rendering.py:119-132,145-147 in the epilogue.
//
// Why.  With one wave per SIMD (nnr_mlp_fwd.hip) everything that is not an MFMA -- the split of the next row's activations into their
// three bf16 terms (44 VALU instructions per row), the ReLU epilogue of the other output half, the weight DMA, the stash stores, and
// per 32-sample chunk the encodings (48 sin / cos per lane), the two heads and the compositing -- is issued by the wave that also has to
// keep the matrix pipe fed: 3.4 non-MFMA instructions per MFMA inside a layer, and ~10 % of a chunk with no MFMA in flight at all
// (profiles/r03/o_split3_variants.txt: 0.81 ms product, 0.69 without the side units, 0.70 without the DMA, 0.75 without the split;
// the six MFMA terms alone are 0.60 ms at the clock the chip holds).  Here a workgroup is 8 waves = 4 PAIRS, one pair per SIMD:
//   * the MFMA wave owns the accumulators (two sets of 64 AGPRs, ping-pong) and does nothing but: read weight fragments and the
//     ready-made bf16 term operands of its 32 samples from LDS, issue the row's 24 MFMAs, and hand finished accumulators back through
//     LDS, 1 KiB at a time, while the OTHER set accumulates -- about one non-MFMA instruction per MFMA, no vmcnt, no VALU arithmetic;
//   * the HELPER wave owns the activations (fp32, 128 AGPRs used as storage), and does everything else for the same 32 samples: weight
//     DMA for the whole workgroup (global_load_lds, a quarter of each panel per helper), bias + ReLU of what the MFMA wave hands back,
//     the three-term split of every row two rows ahead of its use, the encodings of the NEXT chunk while the current one runs, the
//     density / colour heads, sigmoid, compositing and the output stores.
// One s_barrier per row (768 matrix-pipe cycles) is the only synchronisation; every hand-over has a fixed lag in rows:
//     helper period j (between barriers E_{j-1} and E_j) writes the terms of row j + 2 into term slot j & 1;
//     MFMA row j reads the terms of row j + 1 (slot (j + 1) & 1) and the fragments of row j + 1, and writes the accumulator pieces
//       scheduled for row j into drain slots 4 (j & 1) .. + 3;  the helper takes those in in period j + 1;
//     the helper issues the DMA of panel q (rows 2 q, 2 q + 1) in period 2 q - 4 -- the ring buffer of panel q - 3 was last read in row
//       2 q - 6 -- and waits for it in period 2 q - 2, one barrier before the MFMA waves first read it.
// Both roles run the same sequence of segments (run_pass) with exactly one barrier per row, so the barrier counts agree by construction.
// LDS: 72 KiB weight ring + 24 KiB terms + 32 KiB drain slots + 16 KiB parked direction encodings + 12.3 KiB tables = 156.3 KiB.
//
// Arithmetic: the products and their order are those of nnr_split.h (same packed weights, same term order t0..t5 per row, rows in
// order); the one difference to nnr_mlp_fwd.hip is that the bias is ADDED to the finished accumulator by the helper instead of being the
// accumulator's initial value (the MFMA wave starts a group from the constant 0 at no cost) -- a rounding-order difference of 1e-7.
#include <utility>

#include "nnr_device.h"
#include "nnr_kernels.h"
#include "nnr_split.h"

namespace nnr {
namespace ws {

constexpr int kThreads = 512;
constexpr int kPairs = 4;
constexpr int kPanelBytes = kSplitPanelFrags * 1024;           // 24 KiB
constexpr int kRingBytes = kNBuf * kPanelBytes;                // 72 KiB
constexpr int kTermSlot = 3 * 1024;                            // classes l, m, h x 64 lanes x 16 B
constexpr int kTermPair = 2 * kTermSlot;
constexpr int kDrainPair = 8 * 1024;                           // 8 slots of 64 lanes x 16 B
constexpr int kParkPair = 4 * 1024;                            // gamma_4(v): 16 registers x 64 lanes
constexpr int kOffTerm = kRingBytes;
constexpr int kOffDrain = kOffTerm + kPairs * kTermPair;
constexpr int kOffPark = kOffDrain + kPairs * kDrainPair;
constexpr int kOffTab = kOffPark + kPairs * kParkPair;

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// f(integral_constant<int, 0>) .. f(integral_constant<int, N - 1>): loop indices that ARE constants (template arguments, asm immediates)
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(std::make_integer_sequence<int, N>{}, f);
}

// LDS accesses the compiler does not see (cf. frag_read, nnr_mlp_bf16.h): hipcc waits vmcnt(0) before an LDS read that may alias an
// LDS-DMA in flight, and lgkmcnt(0) before every use of its own LDS loads -- the counted waits are placed by hand.
template <int OFF>
__device__ __forceinline__ f32x4 lds_rd(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int OFF>
__device__ __forceinline__ void lds_wr(unsigned addr, f32x4 v) {
    asm volatile("ds_write_b128 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
// s_waitcnt lgkmcnt(0), ordered in front of every later use of v (a function: clang does not capture a local that a generic lambda
// names only in an asm operand)
template <class V>
__device__ __forceinline__ void landed(V& v) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
}
// keep a running sum where it is computed (left alone, hipcc sinks a whole dot product to its only use, hundreds of periods later, and
// keeps the operands in scratch memory until then)
__device__ __forceinline__ void keep(float& a, float& b) { asm volatile("" : "+v"(a), "+v"(b)); }
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    static_assert(N >= 0 && N <= 15, "lgkmcnt is a 4-bit field");
    asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(N) : "memory");
}
// Profiling builds only (results NOT valid): NNR_WS_NO_BARRIER, NNR_WS_NO_DMAWAIT, NNR_WS_NO_DMA, NNR_WS_NO_INTAKE, NNR_WS_NO_TERMS, NNR_WS_NO_DRAIN,
// NNR_WS_HELPER_IDLE (the helper only keeps the barriers), NNR_WS_MFMA_IDLE (the MFMA wave only keeps the barriers)
__device__ __forceinline__ void row_barrier() {
#ifndef NNR_WS_NO_BARRIER
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");
}

// an AGPR used as storage: only ever touched through these two, so the register allocator has no reason to give it a VGPR
__device__ __forceinline__ void aset(float& a, float v) { asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); }
__device__ __forceinline__ float aget(const float& a) {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
// a global load the compiler does not wait for (issued many periods before its use; the counted vmcnt waits of the DMA cover it)
__device__ __forceinline__ float aload(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The MFMA wave.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int D>
struct MfmaRole {
    static constexpr int MT = Layout<D, 2>::HT;
    static_assert(MT == 4, "the row schedule below is written for D = 256 (2 rows per 24 KiB panel)");
    f32x16 acc[2][MT];
    f32x4 fr[3][MT];
    u32x4 xs[3];
    unsigned panel_addr;          // LDS byte address (+ 16 lane) of the panel that holds the CURRENT row's fragments
    unsigned ring_lo, ring_hi;    // first panel, one past the last (+ 16 lane)
    unsigned term_addr, drain_addr;

    __device__ __forceinline__ void pin() {
        asm volatile("" : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]), "+a"(acc[1][0]), "+a"(acc[1][1]),
                     "+a"(acc[1][2]), "+a"(acc[1][3]));
    }

    // One row = 16 k-values = 24 MFMAs into set SET (ZERO: the row starts a group, the accumulators start from 0), in the order of
    // nnr_split.h: (weights term, activation term) = (l,h) (m,m) (m,h) (h,l) (h,m) (h,h).  PR = the row's place in its panel (0 / 1) =
    // the parity of the row number.  ND accumulator pieces W0 .. W0 + ND - 1 of the OTHER set (piece w = tile w / 4, registers 4 (w % 4) ..
    // + 3) go to the drain slots of this row.
    template <int SET, bool ZERO, int PR, int ND_, int W0>
    __device__ __forceinline__ void row() {
#ifdef NNR_WS_NO_DRAIN
        constexpr int ND = 0;
#else
        constexpr int ND = ND_;
#endif
#ifdef NNR_WS_MFMA_IDLE
        row_barrier();
        return;
#endif
        static_assert(ND >= 0 && ND <= 4 && W0 + ND <= 4 * MT, "drain pieces");
        // fragments of row j + 1: the other row of this panel, or the first row of the next panel (in the ring)
        unsigned nxt = panel_addr;
        if (PR == 1) {
            nxt = panel_addr + kPanelBytes;
            nxt = nxt == ring_hi ? ring_lo : nxt;
        }
        u32x4 xn[3];
        constexpr int kMid = 2 * MT + 3 + ND;      // reads / writes this row has issued behind the older fragment classes (<= 15)
        static_for<6 * MT>([&](auto J) __attribute__((always_inline)) {
            constexpr int j = J, t = j / MT, mt = j % MT;
            constexpr int wc = t == 0 ? 0 : (t < 3 ? 1 : 2), xc = t == 0 ? 2 : (t == 1 ? 1 : (t == 2 ? 2 : t - 3));
            __builtin_amdgcn_sched_barrier(0);
            // LDS operations return in order.  In front of the first MFMA of a fragment class: everything older than the operations issued
            // AFTER that class's refills may be outstanding -- t0 (l): the m and h refills of the row before; t1 (m): h refills of the row
            // before + this row's l refills, term reads and drain writes; t3 (h): this row's l, term reads, drains, m.
            if constexpr (mt == 0 && t == 0) asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(fr[0][0]), "+v"(fr[0][1]), "+v"(fr[0][2]), "+v"(fr[0][3]) : [n] "n"(2 * MT));
            if constexpr (mt == 0 && t == 1) asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(fr[1][0]), "+v"(fr[1][1]), "+v"(fr[1][2]), "+v"(fr[1][3]) : [n] "n"(kMid));
            if constexpr (mt == 0 && t == 3) asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(fr[2][0]), "+v"(fr[2][1]), "+v"(fr[2][2]), "+v"(fr[2][3]) : [n] "n"(kMid));
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[SET][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[wc][mt]), __builtin_bit_cast(bf16x8, xs[xc]),
                                                                   (ZERO && t == 0) ? zero : acc[SET][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (t == 0 || t == 2 || t == 5) {      // the fragment's last MFMA of this row: refill it in place for the next row
                constexpr int slot = (((PR ^ 1) * 3) + wc) * MT + mt;
                fr[wc][mt] = lds_rd<1024 * slot>(nxt);
            }
            if constexpr (t == 0 && mt == MT - 1) {          // behind the l refills: the next row's operands, this row's drain pieces
                static_for<3>([&](auto C) __attribute__((always_inline)) {
                    constexpr int c = C;
                    xn[c] = __builtin_bit_cast(u32x4, lds_rd<(PR ^ 1) * kTermSlot + 1024 * c>(term_addr));
                });
                static_for<ND>([&](auto I) __attribute__((always_inline)) {
                    constexpr int w = W0 + I, tl = w / 4, q = w % 4;
                    const f32x16& s = acc[SET ^ 1][tl];
                    lds_wr<(PR * 4 + I) * 1024>(drain_addr, f32x4{s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]});
                });
            }
        });
        pin();
        __builtin_amdgcn_sched_barrier(0);
        // the next row's operands have landed and this row's drain pieces are written: only the m and h refills are younger
        asm volatile("s_waitcnt lgkmcnt(%[n])" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]) : [n] "n"(2 * MT) : "memory");
        row_barrier();
        xs[0] = xn[0]; xs[1] = xn[1]; xs[2] = xn[2];
        panel_addr = nxt;
    }

    // A part of G rows into set SET starting a group (FIRST) or continuing one; rows [D0, D0 + NDR) each drain RATE pieces of the other set
    template <int G, int SET, bool FIRST, int D0, int NDR, int RATE>
    __device__ __forceinline__ void part() {
        static_for<G>([&](auto I) __attribute__((always_inline)) {
            constexpr int g = I;
            constexpr bool dr = g >= D0 && g < D0 + NDR;
            row<SET, FIRST && g == 0, g & 1, dr ? RATE : 0, dr ? RATE * (g - D0) : 0>();
        });
    }
    // row 0's fragments (panel 0, first row) and operands (term slot 0)
    __device__ __forceinline__ void prologue() {
        static_for<3>([&](auto C) __attribute__((always_inline)) {
            constexpr int c = C;
            static_for<MT>([&](auto T) __attribute__((always_inline)) {
                constexpr int t = T;
                fr[c][t] = lds_rd<1024 * (c * MT + t)>(panel_addr);
            });
            xs[c] = __builtin_bit_cast(u32x4, lds_rd<1024 * c>(term_addr));
        });
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]) : : "memory");
        static_for<3>([&](auto C) __attribute__((always_inline)) {
            constexpr int c = C;
            asm volatile("" : "+v"(fr[c][0]), "+v"(fr[c][1]), "+v"(fr[c][2]), "+v"(fr[c][3]));
        });
    }
    __device__ __forceinline__ void seg_l1() {
        part<4, 0, true, 0, 4, 4>();     // hidden 1, outputs [0, D/2): drains the colour layer of the chunk before (set 1)
        part<4, 1, true, 0, 4, 4>();     // outputs [D/2, D): drains set 0
    }
    __device__ __forceinline__ void seg_dense() {
        part<16, 0, true, 0, 8, 2>();    // pass A: the half B of the layer before leaves set 1 during its first 8 rows
        part<16, 1, true, 0, 8, 2>();    // pass B: half A of this layer leaves set 0
    }
    __device__ __forceinline__ void seg_l5() {
        part<16, 0, true, 0, 8, 2>();    // [h4 ; e] -> D, outputs [0, D/2): the hidden part ...
        part<4, 0, false, 0, 0, 0>();    // ... and the position-encoding part of the same accumulators
        part<16, 1, true, 0, 8, 2>();
        part<4, 1, false, 0, 0, 0>();
    }
    __device__ __forceinline__ void seg_colour() {
        part<16, 0, true, 0, 8, 2>();    // colour hidden (merged with the feature layer): h8 part ...
        part<2, 0, false, 0, 0, 0>();    // ... direction-encoding part
        // the colour layer's accumulators leave during the NEXT chunk's hidden 1, pass A -- which accumulates into set 0: park them in
        // set 1 (17 groups per chunk: without this the sets would swap roles every chunk and the chunk body would exist twice)
        static_for<MT>([&](auto T) __attribute__((always_inline)) { acc[1][T] = acc[0][T]; });
        pin();
    }
    __device__ __forceinline__ void tail() {      // after the last chunk: the colour layer's accumulators, 4 pieces per barrier
        static_for<4>([&](auto R) __attribute__((always_inline)) {
            constexpr int r = R;
            static_for<4>([&](auto I) __attribute__((always_inline)) {
                constexpr int w = 4 * r + I, tl = w / 4, q = w % 4;
                const f32x16& s = acc[1][tl];
                lds_wr<((r & 1) * 4 + I) * 1024>(drain_addr, f32x4{s[4 * q], s[4 * q + 1], s[4 * q + 2], s[4 * q + 3]});
            });
            wait_lgkm<0>();
            row_barrier();
        });
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
// The helper wave.
// ---------------------------------------------------------------------------------------------------------------------------------
enum IntakeKind { IN_NONE = 0, IN_A = 1, IN_B = 2, IN_COLOUR = 3 };
enum SrcKind { SRC_E = 0, SRC_H = 1, SRC_DIR = 2 };

template <int D>
struct HelperRole {
    using L = Layout<D, 2>;
    static constexpr int PW = kSplitPanelFrags / kPairs;      // 1 KiB pieces of a panel that one helper copies
    static_assert(PW == 6, "pieces are addressed as immediates -3 .. 2 KiB around the fourth");
    static constexpr int DT = L::DT, HT = L::HT, HR = 16 * HT;
    const MlpFwdArgs& a;
    const char* w_src;            // (wave-uniform) this helper's slice of panel 0 in the packed stream, + 3 KiB
    unsigned w_lds;               // (wave-uniform) the same place in ring buffer 0
    unsigned lane16;              // 16 lane
    int n_panels;
    bool more;                    // another chunk follows: the stream wraps
    int lane, half, col, pair;
    unsigned term_addr, drain_addr, park_addr, tab_addr;      // LDS byte addresses, lane part included (tab_addr: + 16 half)
    int dp;                       // chunk-relative index of the next panel to DMA
    float hs[16 * DT];            // the current layer's input, fragment layout, in AGPRs
    float e[32];                  // gamma_10(p) of the current chunk; rewritten with the next chunk's during the colour layer
    float ro[3], rd[3];
    float z_cur, zn_cur, z_epi, zn_epi, z_new, zn_new;      // z of this lane's sample and of its successor: current chunk, chunk whose epilogue is pending, next chunk
    float al[6];                  // async loads of the next chunk: z_lo, z_hi, jitter of the sample and of its successor
    int64_t chunk_epi;            // (wave-uniform) the chunk whose epilogue is pending
    float sg0, sg1, sigma_epi;
    float ra[3][2];
    float cT, cr, cg, cb, cz, cw;
    bool fuse;

    __device__ __forceinline__ HelperRole(const MlpFwdArgs& a_) : a(a_) {}

    // ---- sampling (model/rendering.py:184-195), unfused mul / add to round like the reference ----
    __device__ __forceinline__ void locate(int64_t chunk_id, int64_t& s, int64_t& sc, int& ray, int& j) const {
        s = chunk_id * kChunk + col;
        sc = s < a.S ? s : a.S - 1;
        ray = (int)(sc / a.N);
        j = (int)(sc - (int64_t)ray * a.N);
    }
    __device__ __forceinline__ void issue_sample_loads(int64_t chunk_id) {
        int64_t s, sc; int ray, j;
        locate(chunk_id, s, sc, ray, j);
        const int jn = j + 1 < a.N ? j + 1 : j;
        al[2] = 0.f; al[5] = 0.f;
        al[0] = aload(a.z_lo + j); al[1] = aload(a.z_hi + j);
        al[3] = aload(a.z_lo + jn); al[4] = aload(a.z_hi + jn);
        if (a.jitter) { al[2] = aload(a.jitter + sc); al[5] = aload(a.jitter + (j + 1 < a.N ? sc + 1 : sc)); }
    }
    __device__ __forceinline__ void finish_sample(int64_t chunk_id, float& z, float& zn) {
        asm volatile("" : "+v"(al[0]), "+v"(al[1]), "+v"(al[2]), "+v"(al[3]), "+v"(al[4]), "+v"(al[5]));      // (behind a counted vmcnt wait)
        z = al[0]; zn = al[3];
        if (a.jitter) {
            z = __fadd_rn(al[0], __fmul_rn(__fsub_rn(al[1], al[0]), al[2]));
            zn = __fadd_rn(al[3], __fmul_rn(__fsub_rn(al[4], al[3]), al[5]));
        }
        int64_t s, sc; int ray, j;
        locate(chunk_id, s, sc, ray, j);
        if (half == 0 && s < a.S && !fuse) a.ws_z[s] = z;
    }

    // ---- one period ----
    // terms of row j + 2 from SRC (row SROW of that vector), into term slot PR = j & 1
    template <int PR, int SRC, int SROW>
    __device__ __forceinline__ void make_terms() {
#if defined(NNR_WS_NO_TERMS) || defined(NNR_WS_HELPER_IDLE)
        return;
#endif
        float v[8];
        if constexpr (SRC == SRC_E) {
            static_for<8>([&](auto I) __attribute__((always_inline)) { v[I] = e[8 * SROW + I]; });
        } else if constexpr (SRC == SRC_H) {
            static_for<8>([&](auto I) __attribute__((always_inline)) { v[I] = aget(hs[8 * SROW + I]); });
        } else {
            const f32x4 p0 = lds_rd<(2 * SROW) * 1024>(park_addr), p1 = lds_rd<(2 * SROW + 1) * 1024>(park_addr);
            f32x4 q0 = p0, q1 = p1;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1));
            static_for<4>([&](auto I) __attribute__((always_inline)) { constexpr int i = I; v[i] = q0[i]; v[4 + i] = q1[i]; });
        }
        uint32_t th[4], tm[4], tl[4];
        static_for<4>([&](auto Q) __attribute__((always_inline)) { split_pair(v[2 * Q], v[2 * Q + 1], th[Q], tm[Q], tl[Q]); });
        lds_wr<PR * kTermSlot + 0>(term_addr, __builtin_bit_cast(f32x4, u32x4{tl[0], tl[1], tl[2], tl[3]}));
        lds_wr<PR * kTermSlot + 1024>(term_addr, __builtin_bit_cast(f32x4, u32x4{tm[0], tm[1], tm[2], tm[3]}));
        lds_wr<PR * kTermSlot + 2048>(term_addr, __builtin_bit_cast(f32x4, u32x4{th[0], th[1], th[2], th[3]}));
    }
    // the NI accumulator pieces IW0 .. of a group that the MFMA wave wrote in the row before this period (drain slots of parity PR ^ 1):
    // + bias, ReLU, into the activation registers (KIND A: registers [0, HR), B: [HR, 2 HR)) or into the colour head's dot products.
    // bias_addr: LDS byte address of the group's bias row (+ 16 half); SIGMA: the density head runs along (hidden 8 arriving)
    template <int PR, int KIND, int NI, int IW0, bool SIGMA>
    __device__ __forceinline__ void intake(unsigned bias_addr) {
#if defined(NNR_WS_NO_INTAKE) || defined(NNR_WS_HELPER_IDLE)
        return;
#endif
        if constexpr (KIND != IN_NONE && NI > 0) {
            static_for<NI>([&](auto I) __attribute__((always_inline)) {
                constexpr int w = IW0 + I, t = w / 4, q = w % 4;
                f32x4 v = lds_rd<((PR ^ 1) * 4 + I) * 1024>(drain_addr);
                f32x4 b = lds_rd<(32 * t + 8 * q) * 4>(bias_addr);
                landed(v); landed(b);
                float x[4];
                static_for<4>([&](auto K) __attribute__((always_inline)) { constexpr int k = K; x[k] = relu1(v[k] + b[k]); });
                if constexpr (KIND == IN_COLOUR) {
                    // rgb head: per-lane dot products over the lane's half of g, in the order of nnr_mlp_fwd.hip
                    static_for<3>([&](auto C) __attribute__((always_inline)) {
                        const f32x4 w4 = lds_rd<(L::wrgb_off - L::bias_base + (2 * C) * HR + 4 * w) * 4>(tab_addr + (unsigned)(half * HR * 4) - 16u * half);
                        f32x4 ww = w4;
                        landed(ww);
                        ra[C][0] = w == 0 ? ww[0] * x[0] : fmaf(ww[0], x[0], ra[C][0]);      // (the first piece starts the sums: nothing lives across a chunk)
                        ra[C][1] = w == 0 ? ww[1] * x[1] : fmaf(ww[1], x[1], ra[C][1]);
                        ra[C][0] = fmaf(ww[2], x[2], ra[C][0]);
                        ra[C][1] = fmaf(ww[3], x[3], ra[C][1]);
                        keep(ra[C][0], ra[C][1]);
                    });
                } else {
                    constexpr int r0 = (KIND == IN_B ? HR : 0) + 4 * w;
                    static_for<4>([&](auto K) __attribute__((always_inline)) { aset(hs[r0 + K], x[K]); });
                    if constexpr (SIGMA) {      // density head: sum_r w_sigma[r] h8[r], even / odd registers in two chains, ascending
                        const f32x4 w4 = lds_rd<(L::wsig_off - L::bias_base + r0) * 4>(tab_addr + (unsigned)(half * 16 * DT * 4) - 16u * half);
                        f32x4 ww = w4;
                        landed(ww);
                        sg0 = r0 == 0 ? ww[0] * x[0] : fmaf(ww[0], x[0], sg0);
                        sg1 = r0 == 0 ? ww[1] * x[1] : fmaf(ww[1], x[1], sg1);
                        sg0 = fmaf(ww[2], x[2], sg0);
                        sg1 = fmaf(ww[3], x[3], sg1);
                        keep(sg0, sg1);
                    }
                }
            });
        }
    }
    // This helper's six pieces of chunk-relative panel p (p >= n_panels: panel p - n_panels of the next chunk, if one follows): LDS-DMA with a
    // scalar base -- SGPR pair + 16 lane + immediate, M0 = the LDS destination, the immediate moves both -- so that no address lives in a
    // VGPR.  fwd_panels is a multiple of kNBuf: the ring position of a panel is p % kNBuf in every chunk.
    __device__ __forceinline__ bool issue_panel(int p) {
        if (!(p < n_panels || more)) return false;
        const int ps = p < n_panels ? p : p - n_panels;
        const char* g = w_src + (int64_t)ps * kPanelBytes;
        const unsigned l = w_lds + (unsigned)(p % kNBuf) * kPanelBytes;
        unsigned m0_saved;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0" : "=&s"(m0_saved) : "s"(l) : "memory");
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:-3072\n\tglobal_load_lds_dwordx4 %0, %1 offset:-2048\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:-1024\n\tglobal_load_lds_dwordx4 %0, %1\n\t"
                     "global_load_lds_dwordx4 %0, %1 offset:1024\n\tglobal_load_lds_dwordx4 %0, %1 offset:2048"
                     : : "v"(lane16), "s"(g) : "memory");
        asm volatile("s_mov_b32 m0, %0" : : "s"(m0_saved) : "memory");
        return true;
    }
    // weight DMA: every even period one panel -- panel (j + 4) / 2 of this chunk -- then wait for the panel issued one even period
    // earlier (the six pieces just issued may stay in flight)
    __device__ __forceinline__ void dma() {
#if defined(NNR_WS_NO_DMA) || defined(NNR_WS_HELPER_IDLE)
        ++dp;
        return;
#endif
        const bool issued = issue_panel(dp);
        ++dp;
#ifdef NNR_WS_NO_DMAWAIT
        return;
#endif
        if (issued) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(PW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    }
    template <int PR, int SRC, int SROW, int KIND, int NI, int IW0, bool SIGMA = false, class Extra>
    __device__ __forceinline__ void period(unsigned bias_addr, Extra&& extra) {
        intake<PR, KIND, NI, IW0, SIGMA>(bias_addr);
        make_terms<PR, SRC, SROW>();
        if constexpr (PR == 0) dma();
#ifndef NNR_WS_HELPER_IDLE
        extra();
#endif
        wait_lgkm<0>();
        row_barrier();
    }
    struct Nothing { __device__ __forceinline__ void operator()() const {} };

    __device__ __forceinline__ unsigned bias_lds(int layer, int hb) const {      // bias row of (layer, output half), + this lane's half
        return tab_addr + (unsigned)((L::bias_off(0) - L::bias_base) * 4) + (unsigned)(layer * D * 4) + (unsigned)(hb * L::Dh * 4);
    }

    // ---- epilogue of a chunk: colour head -> sigmoid; density; output (model/official_nerf.py:77-92) or compositing ----
    __device__ __forceinline__ void finalize(bool last_chunk) {
        float rgbv[3];
        static_for<3>([&](auto C) __attribute__((always_inline)) { rgbv[C] = sum_halves(ra[C][0] + ra[C][1]); });
        const f32x4 bq = lds_rd<(L::bias_off(11) - L::bias_base) * 4>(tab_addr - 16u * half);
        f32x4 b = bq;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b));
        f32x4 o;
        o[0] = sigmoid_ref(rgbv[0] + b[0]);
        o[1] = sigmoid_ref(rgbv[1] + b[1]);
        o[2] = sigmoid_ref(rgbv[2] + b[2]);
        o[3] = sigma_epi;
        int64_t s, sc; int ray_epi, j_epi;
        locate(chunk_epi, s, sc, ray_epi, j_epi);
        if (!fuse) {
            if (half == 0 && s < a.S) *reinterpret_cast<f32x4*>(a.ws_out4 + 4 * s) = o;
        } else {
            // model/rendering.py:119-132,145-147 for the 32 samples of the chunk, as nnr_mlp_fwd.hip does it
            const int jn = j_epi + 1;
            float unused;
            const float alpha = half == 0 ? sample_alpha(o[3], jn < a.N ? zn_epi - z_epi : 1e10f, jn == a.N, a.flags, unused) : 0.f;
            const float incl = wave_scan_mul(half == 0 ? (1.f - alpha) + kEpsT : 1.f, lane);
            float excl = __shfl_up(incl, 1, 64);
            if (lane == 0) excl = 1.f;
            const float w = alpha * cT * excl;
            cT *= __shfl(incl, 31, 64);
            cr += w * o[0]; cg += w * o[1]; cb += w * o[2]; cz += w * z_epi; cw += w;
            if (last_chunk) {
                const float sr = wave_sum(cr), sgn = wave_sum(cg), sb = wave_sum(cb), sz = wave_sum(cz), sw = wave_sum(cw);
                if (lane == 0) {
                    const float bg = (a.flags & kFlagWhiteBg) ? 1.f - sw : 0.f;
                    float* out = a.fuse_rgb + 3 * (int64_t)ray_epi;
                    out[0] = sr + bg; out[1] = sgn + bg; out[2] = sb + bg;
                    a.fuse_dist[ray_epi] = sz;
                }
            }
        }
    }

    // ---- the segments of a chunk (same rows, same barriers as MfmaRole's) ----
    // hidden 1 (8 rows).  Periods 1..4 take in the colour layer of the chunk BEFORE (have_prev), period 5 finishes that chunk.
    __device__ __forceinline__ void seg_l1(bool have_prev) {
        const unsigned bc = tab_addr + (unsigned)((L::bias_off(10) - L::bias_base) * 4);
        const unsigned b0 = bias_lds(0, 0);
        period<0, SRC_E, 2, IN_NONE, 0, 0>(0u, Nothing{});
        static_for<4>([&](auto I) __attribute__((always_inline)) {
            constexpr int j = 1 + I;
            auto body = [&](auto take) __attribute__((always_inline)) {
                constexpr bool tk = decltype(take)::value;
                if constexpr (j + 2 < 8) period<j & 1, SRC_E, (j + 2) & 3, tk ? IN_COLOUR : IN_NONE, 4, 4 * (j - 1)>(bc, Nothing{});
            };
            if (have_prev) body(std::true_type{}); else body(std::false_type{});
        });
        {
            auto fin = [&]() __attribute__((always_inline)) { if (have_prev) finalize(false); };
            period<1, SRC_E, 3, IN_A, 4, 0>(b0, fin);                 // period 5: hidden 1, half A, pieces written in row 4
        }
        period<0, SRC_H, 0, IN_A, 4, 4>(b0, Nothing{});               // period 6: terms of row 8 = hidden 2, row 0
        period<1, SRC_H, 1, IN_A, 4, 8>(b0, Nothing{});               // period 7
    }
    // one D -> D layer (32 rows); li = state_dict index of the layer.  AFTER_L1: period 0 still takes in hidden 1's last pieces.
    // SIGMA: the density head runs along the arrival of this layer's half A.  extra(G): further work of period G
    template <bool AFTER_L1, bool SIGMA, class Extra>
    __device__ __forceinline__ void seg_dense(int li, Extra&& extra) {
        const unsigned b_prev = bias_lds(li - 1, 1), b_this = bias_lds(li, 0);
        static_for<32>([&](auto G) __attribute__((always_inline)) {
            constexpr int g = G, pr = g & 1, srow = (g + 2) % 16;
            auto ex = [&]() __attribute__((always_inline)) { extra(G); };
            if constexpr (g == 0) {
                if constexpr (AFTER_L1) period<pr, SRC_H, srow, IN_A, 4, 12>(bias_lds(0, 0), ex);
                else period<pr, SRC_H, srow, IN_NONE, 0, 0>(0u, ex);
            } else if constexpr (g >= 1 && g <= 8) {
                period<pr, SRC_H, srow, IN_B, 2, 2 * (g - 1)>(b_prev, ex);
            } else if constexpr (g >= 17 && g <= 24) {
                period<pr, SRC_H, srow, IN_A, 2, 2 * (g - 17), SIGMA>(b_this, ex);
            } else {
                period<pr, SRC_H, srow, IN_NONE, 0, 0>(0u, ex);
            }
        });
    }
    struct NoExtra { template <class G> __device__ __forceinline__ void operator()(G) const {} };
    // hidden 8 (the last D -> D layer): the density head runs along as its output arrives, and the NEXT chunk is sampled (period 0: the
    // loads were issued during hidden 5) and encoded, one register of gamma_10 per period -- e is dead since hidden 5
    __device__ __forceinline__ void seg_l8(int64_t next_chunk) {
        auto enc = [&](auto G) __attribute__((always_inline)) {
            constexpr int g = decltype(G)::value;
            if (more) {
                if constexpr (g == 0) finish_sample(next_chunk, z_new, zn_new);
                // (the point is recomputed per register: three values that would otherwise live -- in scratch -- across the layer)
                const float px = __fadd_rn(ro[0], __fmul_rn(rd[0], z_new)), py = __fadd_rn(ro[1], __fmul_rn(rd[1], z_new)),
                            pz = __fadd_rn(ro[2], __fmul_rn(rd[2], z_new));
                e[g] = enc_register(g, half, kPosReal, px, py, pz);
            }
        };
        seg_dense<false, true>(7, enc);
    }
    // hidden 5: [h4 ; e] -> D (40 rows)
    __device__ __forceinline__ void seg_l5() {
        const unsigned b_prev = bias_lds(3, 1), b_this = bias_lds(4, 0);
        static_for<40>([&](auto G) __attribute__((always_inline)) {
            constexpr int g = G, pr = g & 1, r = g + 2;
            constexpr int src = (r < 16 || (r >= 20 && r < 36) || r >= 40) ? SRC_H : SRC_E;
            constexpr int srow = r < 16 ? r : (r < 20 ? r - 16 : (r < 36 ? r - 20 : (r < 40 ? r - 36 : r - 40)));
            if constexpr (g >= 1 && g <= 8) {
                period<pr, src, srow, IN_B, 2, 2 * (g - 1)>(b_prev, Nothing{});
            } else if constexpr (g >= 21 && g <= 28) {
                period<pr, src, srow, IN_A, 2, 2 * (g - 21)>(b_this, Nothing{});
            } else {
                period<pr, src, srow, IN_NONE, 0, 0>(0u, Nothing{});
            }
        });
    }
    // colour hidden (18 rows): hidden 8's half B arrives (density head); the last two periods write the next chunk's first two rows
    __device__ __forceinline__ void seg_colour() {
        const unsigned b_prev = bias_lds(7, 1);
        static_for<18>([&](auto G) __attribute__((always_inline)) {
            constexpr int g = G, pr = g & 1, r = g + 2;
            constexpr int src = r < 16 ? SRC_H : (r < 18 ? SRC_DIR : SRC_E);
            constexpr int srow = r < 16 ? r : (r < 18 ? r - 16 : r - 18);
            auto dens = [&]() __attribute__((always_inline)) {
                if constexpr (g == 9) {     // hidden 8 is complete (its half B arrived in periods 1..8): the density of this chunk
                    float sg = sg0 + sg1;
                    f32x4 b = lds_rd<(L::bias_off(8) - L::bias_base) * 4>(tab_addr - 16u * half);
                    landed(b);
                    sigma_epi = sum_halves(sg) + b[0];
                }
            };
            if constexpr (g >= 1 && g <= 8) period<pr, src, srow, IN_B, 2, 2 * (g - 1), true>(b_prev, dens);
            else period<pr, src, srow, IN_NONE, 0, 0>(0u, dens);
        });
    }
    // after the last chunk: its colour layer arrives in four barriers' worth of pieces, then its epilogue
    __device__ __forceinline__ void tail() {
        const unsigned bc = tab_addr + (unsigned)((L::bias_off(10) - L::bias_base) * 4);
        static_for<4>([&](auto R) __attribute__((always_inline)) {
            constexpr int r = R;
            if constexpr (r > 0) intake<(r & 1), IN_COLOUR, 4, 4 * (r - 1), false>(bc);      // pieces of tail row r - 1 sit in parity (r - 1) & 1
            wait_lgkm<0>();
            row_barrier();
        });
        intake<0, IN_COLOUR, 4, 12, false>(bc);      // tail row 3 wrote parity 1
        finalize(true);
    }
};

template <int D, bool TRAIN>
__global__ __launch_bounds__(kThreads) void mlp_fwd_ws_kernel(MlpFwdArgs a) {
    static_assert(D == 256 && !TRAIN, "inference, D = 256 (prototype)");
    using L = Layout<D, 2>;
    static_assert(L::fwd_panels % kNBuf == 0, "the ring position of a panel must not depend on the chunk");
    static_assert(L::fwd_panels * 2 == 8 + 6 * 32 + 40 + 18, "two rows per panel, 258 rows per chunk");
    __shared__ __attribute__((aligned(16))) char smem[kOffTab + ((L::table_floats + 3) / 4) * 16 + 32];
    float* const ltab = reinterpret_cast<float*>(smem + kOffTab);
    int* const cnt = reinterpret_cast<int*>(smem + kOffTab + ((L::table_floats + 3) / 4) * 16);
    for (int i = threadIdx.x; i < L::table_floats; i += kThreads) ltab[i] = a.packed[L::bias_base + i];
    if (threadIdx.x < 8) cnt[threadIdx.x] = 0;
    __syncthreads();
    // Roles by SIMD: the first wave to arrive on a SIMD is its MFMA wave, the second its helper (8 waves of 256 registers: two per
    // SIMD).  Should the hardware ever place them otherwise, fall back to the wave index.
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int simd = (int)(__builtin_amdgcn_s_getreg(4 | (4 << 6) | ((2 - 1) << 11)) & 3);   // HW_REG_HW_ID (id 4): SIMD_ID = bits [5:4]
    int rank = 0;
    if (lane == 0) rank = atomicAdd(&cnt[simd], 1);
    rank = __builtin_amdgcn_readfirstlane(rank);
    __syncthreads();
    const bool by_simd = cnt[0] == 2 && cnt[1] == 2 && cnt[2] == 2 && cnt[3] == 2;
    const int pair = __builtin_amdgcn_readfirstlane(by_simd ? simd : (wave & 3));
    const bool is_mfma = __builtin_amdgcn_readfirstlane(by_simd ? (rank == 0) : (wave < 4)) != 0;
    const int n_pass = a.chunks_per_ray > 0 ? a.chunks_per_ray : 1;
    const unsigned base = lds_byte_address(smem);

    if (is_mfma) {
        __builtin_amdgcn_s_setprio(3);
        MfmaRole<D> m;
        m.ring_lo = base + 16u * lane;
        m.ring_hi = m.ring_lo + kRingBytes;
        m.panel_addr = m.ring_lo;
        m.term_addr = base + kOffTerm + pair * kTermPair + 16u * lane;
        m.drain_addr = base + kOffDrain + pair * kDrainPair + 16u * lane;
        zero_acc(m.acc[0]);
        zero_acc(m.acc[1]);
        row_barrier();                                   // S: panels 0 and 1 have landed, the terms of rows 0 and 1 are written
        m.prologue();
#pragma unroll 1
        for (int pass = 0; pass < n_pass; ++pass) {
            m.seg_l1();
#pragma unroll 1
            for (int l = 0; l < 3; ++l) m.seg_dense();
            m.seg_l5();
#pragma unroll 1
            for (int l = 0; l < 3; ++l) m.seg_dense();
            m.seg_colour();
        }
        wait_lgkm<0>();
        m.tail();
    } else {
        using H = HelperRole<D>;
        H h(a);
        h.w_src = reinterpret_cast<const char*>(a.packed) + pair * (H::PW * 1024) + 3 * 1024;
        h.w_lds = base + pair * (H::PW * 1024) + 3 * 1024;
        h.lane16 = 16u * lane;
        h.n_panels = L::fwd_panels;
        h.more = n_pass > 1;
        h.lane = lane; h.half = lane >> 5; h.col = lane & 31; h.pair = pair;
        h.term_addr = base + kOffTerm + pair * kTermPair + 16u * lane;
        h.drain_addr = base + kOffDrain + pair * kDrainPair + 16u * lane;
        h.park_addr = base + kOffPark + pair * kParkPair + 16u * lane;
        h.tab_addr = base + kOffTab + 16u * (lane >> 5);
        h.fuse = a.fuse_rgb != nullptr;
        h.sigma_epi = 0.f;
        h.cT = 1.f; h.cr = h.cg = h.cb = h.cz = h.cw = 0.f;
        h.issue_panel(0);
        h.issue_panel(1);
        h.dp = 2;
        // Work decomposition as in nnr_mlp_fwd.hip: flat -- workgroup b takes the samples [128 b, 128 b + 128), pair p the 32 from 32 p
        // on; ray mode -- workgroup b the rays 4 b .. 4 b + 3, pair p walks the N / 32 chunks of ray 4 b + p.
        auto chunk_of = [&](int pass) -> int64_t {
            return a.chunks_per_ray > 0 ? ((int64_t)blockIdx.x * kPairs + pair) * n_pass + pass : (int64_t)blockIdx.x * kPairs + pair;
        };
        {   // first chunk: sampled and encoded before the first barrier
            int64_t s, sc; int ray, j;
            h.locate(chunk_of(0), s, sc, ray, j);
            const float* po = a.pts_o + 3 * (int64_t)ray; const float* pd = a.pts_d + 3 * (int64_t)ray; const float* pv = a.view_d + 3 * (int64_t)ray;
            static_for<3>([&](auto I) __attribute__((always_inline)) { h.ro[I] = po[I]; h.rd[I] = pd[I]; });
            const float vx = pv[0], vy = pv[1], vz = pv[2];
            const float zlo = a.z_lo[j], zhi = a.z_hi[j];
            float z = zlo, zn = 0.f;
            if (a.jitter) z = __fadd_rn(zlo, __fmul_rn(__fsub_rn(zhi, zlo), a.jitter[sc]));
            if (j + 1 < a.N) {
                const float lo1 = a.z_lo[j + 1], hi1 = a.z_hi[j + 1];
                zn = a.jitter ? __fadd_rn(lo1, __fmul_rn(__fsub_rn(hi1, lo1), a.jitter[sc + 1])) : lo1;
            }
            if (h.half == 0 && s < a.S && !h.fuse) a.ws_z[s] = z;
            h.z_cur = z; h.zn_cur = zn;
            const float px = __fadd_rn(h.ro[0], __fmul_rn(h.rd[0], z)), py = __fadd_rn(h.ro[1], __fmul_rn(h.rd[1], z)), pz = __fadd_rn(h.ro[2], __fmul_rn(h.rd[2], z));
            static_for<32>([&](auto R) __attribute__((always_inline)) { h.e[R] = enc_register(R, h.half, kPosReal, px, py, pz); });
            // gamma_4(v): constant along a ray (ray mode), one chunk per workgroup otherwise -- parked in LDS once
            static_for<4>([&](auto Q) __attribute__((always_inline)) {
                f32x4 dq;
                static_for<4>([&](auto I) __attribute__((always_inline)) { constexpr int i = I; dq[i] = enc_register(4 * Q + i, h.half, kDirReal, vx, vy, vz); });
                lds_wr<Q * 1024>(h.park_addr, dq);
            });
        }
        h.template make_terms<0, SRC_E, 0>();
        h.template make_terms<1, SRC_E, 1>();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        row_barrier();                                   // S
#pragma unroll 1
        for (int pass = 0; pass < n_pass; ++pass) {
            const bool more = pass + 1 < n_pass;
            h.more = more;
            const int64_t next_chunk = chunk_of(pass + 1);
            h.seg_l1(pass > 0);
            h.template seg_dense<true, false>(1, typename H::NoExtra{});
#pragma unroll 1
            for (int l = 1; l < 3; ++l) h.template seg_dense<false, false>(1 + l, typename H::NoExtra{});
            h.seg_l5();
#pragma unroll 1
            for (int l = 0; l < 2; ++l) {
                // hidden 7, period 24: the next chunk's z loads go out (eight periods before hidden 8's period 0 takes them)
                auto ld = [&](auto G) __attribute__((always_inline)) {
                    if constexpr (decltype(G)::value == 24) { if (more && l == 1) h.issue_sample_loads(next_chunk); }
                };
                h.template seg_dense<false, false>(5 + l, ld);
            }
            h.seg_l8(next_chunk);
            h.seg_colour();
            // this chunk's epilogue is pending (its colour layer arrives during the next chunk's first rows, or in the tail)
            h.z_epi = h.z_cur; h.zn_epi = h.zn_cur; h.chunk_epi = chunk_of(pass);
            if (more) { h.z_cur = h.z_new; h.zn_cur = h.zn_new; }
            h.dp -= L::fwd_panels;
        }
        h.tail();
    }
}

}  // namespace ws

// launched instead of mlp_fwd_kernel<256, false, 2> when NNR_FWD_WS is set (nnr_mlp_fwd.hip's dispatcher)
hipError_t launch_mlp_fwd_ws(const MlpFwdArgs& a, hipStream_t st) {
    dim3 grid((unsigned)(a.chunks_per_ray > 0 ? a.S_pad / kBlockSamples / a.chunks_per_ray : a.S_pad / kBlockSamples)), block(ws::kThreads);
    prof_before(PROF_FWD_INFER, st);
    hipLaunchKernelGGL((ws::mlp_fwd_ws_kernel<256, false>), grid, block, 0, st, a);
    prof_after(PROF_FWD_INFER, st);
    return hipGetLastError();
}

}  // namespace nnr
