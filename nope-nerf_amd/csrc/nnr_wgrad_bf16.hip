// nnr_wgrad_bf16.hip -- weight / bias gradients of the bf16 training mode (NNR_F_BF16):
//     dW_l[out][in] = sum_s Dlt_l[s][out] * X_l[s][in],   db_l[out] = sum_s Dlt_l[s][out]
// for the 12 nn.Linear layers of model/official_nerf.py:20-37 (autograd's 24 `mm` wgrad calls), bf16 products, fp32 accumulation.
//
// Roofline: HBM.  Both operands were left in HBM by the forward / input-gradient kernels as bf16 (9.1 KB per sample at D = 256:
// 2 x (8 D + D/2) x 2 B of activations and gradients + the two encodings + the output gradients) and every byte is needed exactly
// once; the products are 593 408 MACs per sample = 0.9 us of bf16 MFMA per 128 samples and CU against 4 us of streaming at 6 TB/s.
// So the kernel is a streaming kernel with an MFMA consumer:
//   * a WORKGROUP job (4 waves, one per SIMD, one workgroup per CU) owns ALL tiles of one layer's dW -- 256 x 256 as four 128 x 128
//     wave tiles of 4 x 4 MFMA tiles (v_mfma_f32_32x32x16_bf16) -- for a contiguous range of 32-sample chunks, so each operand byte
//     is fetched once per layer (round 1: one wave per tile, every operand half fetched by two waves, 1.26x the bytes over HBM);
//   * the operands are tile-major planes (nnr_layout.h): a chunk of 32 samples x 16 features is one contiguous 1 KiB block, and a
//     stage = the 16 + 16 blocks of one chunk of both operands = 32 KiB arrives by 32 LDS-DMA pieces (global_load_lds_dwordx4,
//     1 KiB per wave-instruction, 8 per wave) into a ring of four stages -- three chunks (96 KiB) in flight per CU, counted vmcnt,
//     one barrier per stage.  tools/ubench/hbm_stream.hip: this staging pattern alone streams 6.1 TB/s on the box;
//   * the two layers whose input is two planes (skip layer: hidden | position encoding; colour-hidden: feature | direction encoding)
//     are ONE job type with a second activation plane behind the first in the LDS image (36 / 27 blocks a stage, 4 x 5 / 5 x 3 tiles
//     per wave), so that their gradient plane is streamed once, not once per activation plane;
//   * the MFMA wants [feature][8 consecutive samples] per lane, the planes hold [sample][features]: the transposition is the LDS
//     read, ds_read_b64_tr_b16 (two per operand tile and k-step).  The DMA stores the 64 16-byte units of a block to LDS slots
//     permuted by an XOR on the sample index (applied to the per-lane SOURCE address, the LDS image of a DMA is lane-linear) such
//     that the 32 lanes a transposing read services together hit 64 distinct banks;
//   * d(bias): the A operand registers of a tile row are its gradient values, [feature = lane & 31][8 samples]; the tile-column-0
//     waves add them up with v_dot2c_f32_bf16 against (1, 1) -- 4 VALU instructions per tile row and k-step, one fp32 register
//     per tile row (64 more accumulator registers for an all-ones MFMA do not fit beside 256: the 4 x 4 variant spilled);
//   * each wave writes its accumulators once to its own slot; wgrad_b_reduce_kernel adds the slots of a unit's jobs in sample order
//     (fixed order: bit-reproducible) into dW / db.  The feature layer and the first D columns of the colour-hidden layer get their
//     gradients from the merged matrix W' (wgrad_unmerge_kernel in nnr_wgrad.hip), the density and rgb heads are ordinary tiles:
//     their gradient operand is the 16-wide group the input-gradient kernel appends to the colour-hidden gradient plane.
#include "nnr_device.h"
#include "nnr_kernels.h"

namespace nnr {

constexpr int kMaxStageBlocks = 36;         // blocks of 1 KiB per stage at most (the skip layer: 16 gradient + 16 + 4 activation groups)
constexpr int kRingBytes = 144 * 1024;   // the LDS ring: as many stages as fit (4 of the widest unit's, 14 of the rgb head's)
constexpr int kMaxInFlight = 56;            // DMA pieces a wave keeps outstanding at most (vmcnt is a 6-bit counter)
constexpr int kBlockBytes = 1024;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the count field is an immediate; 6 bits on gfx9)
__device__ __forceinline__ void wait_vmcnt(int n) {
#define NNR_W1(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define NNR_W8(a, b, c, d, e, f, g, h) NNR_W1(a) NNR_W1(b) NNR_W1(c) NNR_W1(d) NNR_W1(e) NNR_W1(f) NNR_W1(g) NNR_W1(h)
    switch (n) {
        NNR_W8(0, 1, 2, 3, 4, 5, 6, 7)
        NNR_W8(8, 9, 10, 11, 12, 13, 14, 15)
        NNR_W8(16, 17, 18, 19, 20, 21, 22, 23)
        NNR_W8(24, 25, 26, 27, 28, 29, 30, 31)
        NNR_W8(32, 33, 34, 35, 36, 37, 38, 39)
        NNR_W8(40, 41, 42, 43, 44, 45, 46, 47)
        NNR_W8(48, 49, 50, 51, 52, 53, 54, 55)
        default: asm volatile("s_waitcnt vmcnt(56)" ::: "memory"); break;
    }
#undef NNR_W8
#undef NNR_W1
}

// 8 bf16 (k = 8 consecutive samples of one feature) of MFMA operand tile `t` of the operand image at LDS byte address `img`:
// two transposing reads.  a0 / a1 = the lane's byte offsets for the samples 0-3 / 4-7 of its half of the k-step (see wgrad_b_job).
__device__ __forceinline__ bf16x8 read_operand(const char* lds, int img, int t, int ks, int a0, int a1) {
    const int base = img + t * (2 * kBlockBytes) + ks * 256;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + base + a0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(lds + base + a1));
    const s16x8 q = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, q);
}

// LDS-DMA pieces (StageFeed::issue): 64 lanes x 16 bytes from (wave-uniform base + per-lane 32-bit offset) to the LDS bytes [M0, M0 + 1 KiB).
// Inline assembly for two reasons: the address stays an SGPR pair + ONE VGPR (the builtin's flat per-lane pointers were hoisted out of
// the streaming loop as eight 64-bit VGPR pairs, spilled, and every reload carried an s_waitcnt vmcnt(0) that drained the DMA queue);
// and hipcc does not count asm memory operations, so no compiler-inserted wait ever covers them -- the counted waits below are the
// only ones.
constexpr int kMaxPieces = (kMaxStageBlocks + 3) / 4;    // DMA pieces per wave and stage

struct StageFeed {      // everything a wave needs to issue its pieces of a stage; all fields wave-uniform except src0 / src1
    // Piece q of a wave is block min(wave + 4 q, nblk - 1) of the stage image; which plane that block comes from is fixed for the
    // job, so the wave keeps one running source pointer per piece (chunk s, then += the plane's bytes per chunk): two scalar adds
    // per piece and stage.  (Deriving the source from the three plane descriptors per piece cost ~25 scalar instructions each and
    // one third of the kernel's rate.)
    const char* ptr[kMaxPieces];
    int stride[kMaxPieces];
    unsigned parity;    // bit q: block parity of piece q inside its operand image (selects the source permutation)
    int nblk, P, wave;
    int nst;            // ring depth for this job: narrow units (the heads: 9-10 KiB a stage) get a deeper ring, so that every
                        // workgroup keeps about the same number of bytes in flight -- with 3 stages of 9 KiB a CU streams at half
                        // the rate of one with 3 stages of 32 KiB, and the kernel ends with its slowest workgroup
    int stage_bytes;
    int head, tail;     // ring slot the next issue fills / the current stage occupies
    unsigned lds0;      // LDS byte address of the ring
    unsigned src0, src1;

    // Lane p of a piece fetches the 16-byte unit that belongs in LDS slot p of the block.  Slot of unit (h, c) [h = which quad pair,
    // c = sample in chunk] is 32 h + (c ^ (4 h + 8 par)), par = parity of the block inside its operand image: the samples the
    // transposing reads of one 32-lane group address then fall on 64 distinct banks (see the read offsets in wgrad_b_job).
    __device__ __forceinline__ StageFeed(const WgradJobB& jb, const char* ws, unsigned lds_addr, int wave_, int lane) {
        // the job's fields as opaque scalars: left as loads, the selects below became indexed loads from a scratch copy of the job
        const auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
        const auto sc64 = [&](int64_t v) { return (int64_t)(((uint64_t)(unsigned)sc((int)(v >> 32)) << 32) | (unsigned)sc((int)v)); };
        const int d_groups = sc(jb.d_groups), x_groups = sc(jb.x_groups), dx_groups = d_groups + x_groups;
        const int d_stride = sc(jb.d_stride), x_stride = sc(jb.x_stride), x2_stride = sc(jb.x2_stride), c0 = sc(jb.c0);
        const int64_t d_base = sc64(jb.d_base), x_base = sc64(jb.x_base), x2_base = sc64(jb.x2_base) - (int64_t)x_groups * kBlockBytes;
        nblk = dx_groups + sc(jb.x2_groups);
        wave = wave_;
        P = (nblk - wave + 3) >> 2;   // DMA pieces of THIS wave per stage: blocks wave, wave + 4, ... (vmcnt counts per wave)
        stage_bytes = nblk * kBlockBytes;
        const int fit = kRingBytes / stage_bytes, cap = kMaxInFlight / ((nblk + 3) >> 2) + 1;   // the same ring depth for all waves
        nst = fit < cap ? fit : cap;
        head = tail = 0;
        lds0 = lds_addr;
        parity = 0;
#pragma unroll
        for (int q = 0; q < kMaxPieces; ++q) {
            int i = wave + 4 * q;
            i = i < nblk ? i : nblk - 1;
            const bool is_d = i < d_groups, is_x = i < dx_groups;
            const int k = is_d ? i : i - d_groups;     // block index inside its operand image (the second activation plane
                                                       // continues the first: x_groups is even, the parity carries over)
            const int st = is_d ? d_stride : is_x ? x_stride : x2_stride;
            const int64_t base = is_d ? d_base : is_x ? x_base : x2_base;
            ptr[q] = ws + base + (int64_t)k * kBlockBytes + (int64_t)c0 * st;
            stride[q] = st;
            parity |= (unsigned)(k & 1) << q;
        }
        const int ph = lane >> 5, pc = lane & 31;
        src0 = (32 * ph + (pc ^ (4 * ph))) * 16;
        src1 = (32 * ph + (pc ^ (4 * ph + 8))) * 16;
    }
    // M0 (the LDS destination of a DMA) is saved, walked from block to block (+ 4 KiB per piece) and restored across the run of asm
    // statements: they are volatile, so nothing of the compiler's that could want M0 comes between them.
    __device__ __forceinline__ void issue() {          // the next chunk
        const unsigned stage = lds0 + head * stage_bytes + wave * kBlockBytes;
        head = head + 1 == nst ? 0 : head + 1;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0" : "=&s"(keep) : "s"(stage) : "memory");
#pragma unroll
        for (int q = 0; q < kMaxPieces; ++q) {
            if (q < P) {
                asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x1000\n\ts_nop 0"
                             :
                             : "v"((parity >> q & 1) ? src1 : src0), "s"(ptr[q])
                             : "memory", "scc");
                ptr[q] += stride[q];
            }
        }
        asm volatile("s_mov_b32 m0, %0" : : "s"(keep) : "memory");
    }
    // stage s has landed when at most the pieces of the (nst - 2) younger stages are outstanding; near the end of the range
    // fewer stages are in flight: drain.  Then the barrier: every wave's pieces of stage s are in LDS and everybody is done
    // reading stage s - 1, whose buffer the next issue overwrites.  Returns the LDS byte offset of stage s in the ring.
    __device__ __forceinline__ int enter(int s, int n) {
        wait_vmcnt(s + nst - 2 < n ? (nst - 2) * P : 0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + nst - 1 < n) issue();
        const int img = tail * stage_bytes;
        tail = tail + 1 == nst ? 0 : tail + 1;
        return img;
    }
    __device__ __forceinline__ void start(int n) {
        for (int s = 0; s < nst - 1 && s < n; ++s) issue();
    }
};

template <int MT, int NT>
__device__ __forceinline__ void pin_tiles(f32x16 (&acc)[MT][NT]) {   // keep the accumulators where they are: AGPRs (see pin_acc);
                                                                      // the 4 x 5 tiling has 320: the last 64 stay in VGPRs
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            if ((i * NT + j + 1) * 16 <= 256) asm volatile("" : "+a"(acc[i][j]));
            else asm volatile("" : "+v"(acc[i][j]));
        }
}

// The 4 x 5 tiling holds 320 accumulators: 256 in AGPRs, 64 in VGPRs.  With the builtin the register allocator treats all of them
// as one either-file class and shuffles whole tiles between the files inside the streaming loop (and spills some); an MFMA written
// as an asm statement fixes the file of its accumulator at every use.  hipcc's hazard recognizer does not look inside asm statements:
// the callers keep dependent MFMAs >= 16 MFMAs apart and put s_nops between the last MFMA and the first ordinary read.
__device__ __forceinline__ void mfma_pinned(bool agpr, f32x16& c, bf16x8 a, bf16x8 b) {   // `agpr` folds after unrolling
    if (agpr) asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// a wave without tiles in this job: it still moves its share of the data and keeps the barrier count
__device__ __forceinline__ void wgrad_b_idle(const WgradJobB& jb, const WgradBArgs& a, unsigned lds_addr, int wave, int lane) {
    StageFeed feed(jb, reinterpret_cast<const char*>(a.ws), lds_addr, wave, lane);
    const int n = jb.c1 - jb.c0;
    feed.start(n);
    for (int s = 0; s < n; ++s) feed.enter(s, n);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

template <int MT, int NT, bool BIAS>
__device__ __forceinline__ void wgrad_b_job(const WgradJobB& jb, const WgradBArgs& a, const char* lds, unsigned lds_addr, int wave, int lane,
                                            int ji) {
    StageFeed feed(jb, reinterpret_cast<const char*>(a.ws), lds_addr, wave, lane);
    const int n = jb.c1 - jb.c0;                         // stages = chunks
    const int d_groups = jb.d_groups;
    const int tr0 = MT * (wave / jb.WC), tc0 = NT * (wave % jb.WC);

    // ---- LDS read offsets of this lane.  A transposing read hands lane i of a 16-lane group the four 16-bit elements number i
    // of the four rows the group's lanes 4 r + m (r = row, m = which quarter) point at.  Rows = 4 consecutive samples, quarters =
    // the 4 feature quads of a 16-feature block in natural order (quad m = unit half m & 1, second 8 bytes if m >> 1): lane i then
    // holds feature i of the block for 4 samples.  16-lane group 0 / 1 = block 2 t / 2 t + 1 of the tile, groups 2 / 3 the same
    // blocks for the second half of the k-step's 16 samples (the MFMA's k = 8 (lane >> 5) + 0..7).
    const int grp = lane >> 4, qq = lane & 15, rr = qq >> 2, mm = qq & 3;
    const int par = grp & 1, khalf = grp >> 1, hh = mm & 1, jj = mm >> 1;
    const int lane_off = par * kBlockBytes + hh * 512 + (rr + 8 * (khalf ^ par)) * 16 + jj * 8;
    // samples 0-3 / 4-7 of the lane's half: slot bit 2 is XORed with hh.  The tile origins are folded in here once per job.
    const int a0 = lane_off + 64 * hh + tr0 * (2 * kBlockBytes), a1 = lane_off + 64 * (1 - hh) + tr0 * (2 * kBlockBytes);
    const int b0 = lane_off + 64 * hh + (d_groups + 2 * tc0) * kBlockBytes, b1 = lane_off + 64 * (1 - hh) + (d_groups + 2 * tc0) * kBlockBytes;

    f32x16 acc[MT][NT];
    float bsum[MT];      // this lane's share of d(bias): feature lane & 31 of tile row i, the samples of its k-step half
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        bsum[i] = 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    }

    feed.start(n);
    for (int s = 0; s < n; ++s) {
        const int img = feed.enter(s, n);
        pin_tiles<MT, NT>(acc);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 av[MT], bv[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) av[i] = read_operand(lds, img, i, ks, a0, a1);
#pragma unroll
            for (int j = 0; j < NT; ++j) bv[j] = read_operand(lds, img, j, ks, b0, b1);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    if constexpr (MT * NT * 16 <= 256) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                    else mfma_pinned((i * NT + j + 1) * 16 <= 256, acc[i][j], av[i], bv[j]);
                }
                if constexpr (BIAS) {
                    const u32x4 w = __builtin_bit_cast(u32x4, av[i]);
#pragma unroll
                    for (int q = 0; q < 4; ++q)   // bsum += lo(w) * 1 + hi(w) * 1.  Spelled out: __builtin_amdgcn_fdot2_f32_bf16 on the four
                                                  // words of an operand was compiled (ROCm 7.2) to four instructions on the FIRST word
                        asm("v_dot2c_f32_bf16 %0, 0x3f803f80, %1" : "+v"(bsum[i]) : "v"(w[q]));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of stage s have returned before it reports "done"
    }
    if constexpr (MT * NT * 16 > 256) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // asm MFMAs: their results are read below
    __builtin_amdgcn_s_barrier();              // the ring is free for the next job's prologue

    // ---- flush: tile (i, j) row-major [32][32]; register r of lane (half, col) is row (r & 3) + 8 (r >> 2) + 4 half
    float* slot = a.slots + ((int64_t)ji * 4 + wave) * kSlotBFloats;
    int lane_f = lane;
    asm volatile("" : "+v"(lane_f));   // opaque here: otherwise the 256 lane-constant store offsets are hoisted to kernel entry and
                                       // spilled around the streaming loop
    const int col = lane_f & 31, half = lane_f >> 5;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                slot[((i * NT + j) * 32 + row) * 32 + col] = acc[i][j][r];
            }
    if constexpr (BIAS) {   // [tile row][k-step half][feature]: the reduction adds the two halves
#pragma unroll
        for (int i = 0; i < MT; ++i) slot[kSlotBMaxTiles * kSlotBTile + (2 * i + half) * 32 + col] = bsum[i];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the flush stores are the compiler's; start the next job's counted waits from zero
}

__global__ __launch_bounds__(256, 1) void wgrad_b_kernel(WgradBArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const unsigned lds_addr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char*)lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j0 = a.block_first[blockIdx.x], j1 = a.block_first[blockIdx.x + 1];
    for (int ji = j0; ji < j1; ++ji) {
        const WgradJobB jb = a.jobs[ji];
        if (wave >= jb.WR * jb.WC) {   // no tile of this job
            wgrad_b_idle(jb, a, lds_addr, wave, lane);
            continue;
        }
        const bool bias = jb.bias && (wave % jb.WC) == 0;
        const int key = __builtin_amdgcn_readfirstlane(jb.MT * 8 + jb.NT + (bias ? 64 : 0));
#define NNR_WB_CASE(MT_, NT_)                                                                            \
    case MT_ * 8 + NT_: wgrad_b_job<MT_, NT_, false>(jb, a, lds, lds_addr, wave, lane, ji); break;       \
    case MT_ * 8 + NT_ + 64: wgrad_b_job<MT_, NT_, true>(jb, a, lds, lds_addr, wave, lane, ji); break;
        switch (key) {
            NNR_WB_CASE(4, 5)
            NNR_WB_CASE(5, 3)
            NNR_WB_CASE(4, 4)
            NNR_WB_CASE(2, 2)
            NNR_WB_CASE(5, 2)
            NNR_WB_CASE(3, 1)
            NNR_WB_CASE(1, 2)
            NNR_WB_CASE(1, 1)
            default: break;
        }
#undef NNR_WB_CASE
    }
}

// dW[output rectangle] += sum over the unit's jobs, in sample order, of their slots; d(bias) likewise.  grid = (64, n_outputs).
// The unit's jobs are first collected into LDS by their position in the chain (every thread inspects its share of the job table: no
// dependent walk of next_split), then each element is summed with independent loads -- the order of the additions stays the chain's.
constexpr int kMaxSplits = 512;
__global__ __launch_bounds__(256) void wgrad_b_reduce_kernel(WgradBArgs a) {
    __shared__ int list[kMaxSplits];
    __shared__ int n_list;
    const WgradOutB o = a.outs[blockIdx.y];
    if (threadIdx.x == 0) n_list = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < a.n_jobs; j += 256) {
        if (a.jobs[j].unit != o.unit) continue;
        const int s = a.jobs[j].split;
        if (s >= kMaxSplits) __builtin_trap();     // the plan builder bounds the splits of a unit (nnr_api.cpp)
        list[s] = j;
        atomicMax(&n_list, s + 1);
    }
    __syncthreads();
    const int n = n_list;
    const int64_t stride = 4 * (int64_t)kSlotBFloats;
    const int total = o.n_rows * o.n_cols;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += 64 * 256) {
        const int r = e / o.n_cols, c = e - r * o.n_cols;
        const int dr = o.d_row + r, xc = o.x_col + c;                  // position in the unit's product
        const int rt = dr >> 5, ct = xc >> 5;
        const int wave = (rt / o.MT) * o.WC + ct / o.NT;
        const int tile = (rt % o.MT) * o.NT + ct % o.NT;
        const float* p = a.slots + (int64_t)wave * kSlotBFloats + (tile * 32 + (dr & 31)) * 32 + (xc & 31);
        float sum = 0.f;
        int s = 0;
        for (; s + 8 <= n; s += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[list[s + u] * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; s < n; ++s) sum += p[list[s] * stride];
        a.gw[o.layer][(int64_t)(o.w_row + r) * o.ldw + o.w_col + c] = sum;     // every weight belongs to exactly one output rectangle: overwritten
    }
    if (o.bias && blockIdx.x == 0) {
        for (int r = threadIdx.x; r < o.n_rows; r += 256) {
            const int dr = o.d_row + r, rt = dr >> 5;
            const float* p = a.slots + (int64_t)((rt / o.MT) * o.WC) * kSlotBFloats + kSlotBMaxTiles * kSlotBTile + 2 * (rt % o.MT) * 32 + (dr & 31);
            float sum = 0.f;
            for (int s = 0; s < n; ++s) {
                const float* q = p + list[s] * stride;
                sum += q[0] + q[32];
            }
            a.gb[o.layer][o.w_row + r] = sum;       // one bias rectangle per layer (bf16_units)
        }
    }
}

hipError_t launch_wgrad_bf16(const WgradBArgs& a, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_b_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           kRingBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipError_t e;
    prof_before(PROF_WGRAD, st);
    hipLaunchKernelGGL(wgrad_b_kernel, dim3(a.n_blocks), dim3(256), kRingBytes, st, a);
    hipLaunchKernelGGL(wgrad_b_reduce_kernel, dim3(64, a.n_outs), dim3(256), 0, st, a);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    WgradArgs u{};
    for (int i = 0; i < 13; ++i) { u.gw[i] = a.gw[i]; u.gb[i] = a.gb[i]; }
    u.packed = a.packed;
    u.D = a.D;
    u.bf16 = 1;
    e = launch_wgrad_unmerge(u, st);
    prof_after(PROF_WGRAD, st);      // the bracket covers the whole stage: main kernel + slot reduction + un-merge
    return e;
}

}  // namespace nnr
