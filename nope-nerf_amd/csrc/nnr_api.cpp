// nnr_api.cpp -- the extern "C" surface of libnnr.so (include/nnr.h): argument checking, workspace carving, the
// weight-gradient plan, and kernel sequencing.  No global state; every call is asynchronous on the caller's stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <array>
#include <vector>

#include "../../include/nnr.h"
#include "nnr_device.h"
#include "nnr_kernels.h"
#include "nnr_layout.h"

using namespace nnr;

namespace {

thread_local int g_last_hip = 0;

int hip_fail(hipError_t e) {
    g_last_hip = (int)e;
    return NNR_E_HIP;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_cfg(const nnr_cfg* c) {
    if (!c || c->n_rays <= 0 || c->n_samples <= 0) return NNR_E_BADCFG;
    if (c->hidden != 128 && c->hidden != 256) return NNR_E_UNSUPPORTED;
    if ((c->flags & NNR_F_TRAIN) && c->n_samples > 1024) return NNR_E_UNSUPPORTED;
    return NNR_OK;
}

WsLayout ws_layout(const nnr_cfg* c) {
    WsLayout w;
    w.S = (int64_t)c->n_rays * c->n_samples;
    w.S_pad = (w.S + kBlockSamples - 1) / kBlockSamples * kBlockSamples;
    w.D = c->hidden;
    w.train = (c->flags & NNR_F_TRAIN) != 0;
    w.bf16 = w.train && (c->flags & NNR_F_BF16) != 0;
    w.tile32 = w.train && (c->flags & (NNR_F_BF16 | NNR_F_SPLIT3)) == NNR_F_SPLIT3;   // the three-term training mode's gradient planes (nnr_layout.h)
    return w;
}

int64_t plane(const WsLayout& w, int id) {
    int pitch;
    return w.plane(id, &pitch);
}

size_t packed_floats(int D, int mode) {
    if (mode == 3) return D == 256 ? (size_t)Layout<256, 3>::packed_floats : (size_t)Layout<128, 3>::packed_floats;
    if (mode == 2) return D == 256 ? (size_t)Layout<256, 2>::packed_floats : (size_t)Layout<128, 2>::packed_floats;
    if (mode == 1) return D == 256 ? (size_t)Layout<256, 1>::packed_floats : (size_t)Layout<128, 1>::packed_floats;
    return D == 256 ? (size_t)Layout<256>::packed_floats : (size_t)Layout<128>::packed_floats;
}
bool is_bf16(const nnr_cfg* c) { return (c->flags & NNR_F_BF16) != 0; }
bool is_split3(const nnr_cfg* c) { return (c->flags & (NNR_F_BF16 | NNR_F_SPLIT3)) == NNR_F_SPLIT3; }
bool is_split2(const nnr_cfg* c) { return is_split3(c) && (c->flags & NNR_F_SPLIT2) != 0; }      // forward / input gradient with two-term fp16 operands (nnr_split2.h)
int weight_mode(const nnr_cfg* c) { return is_bf16(c) ? 1 : (is_split2(c) ? 3 : (is_split3(c) ? 2 : 0)); }   // Layout<D, MODE>

// Ray mode of the two MLP kernels (nnr_mlp_fwd.hip): a wave walks the N / 32 chunks of ONE ray, a workgroup four rays.  Needs whole
// chunks per ray and whole workgroups; everything else runs the flat decomposition (same sample numbering, same planes).
// The bf16 kernels (nnr_mlp_fwd_bf16.hip) work on PAIRS of chunks: their unit is 64 samples.
int chunks_per_ray(const nnr_cfg* c) {
    static const bool off = std::getenv("NNR_FLAT_GRID") != nullptr;    // experiments: force the flat decomposition
    const int unit = is_bf16(c) ? kBf16Tiles * kChunk : kChunk;         // samples a wave takes per pass
    const int waves = is_bf16(c) ? kBf16Waves : kWavesPerBlock;         // = rays per workgroup in ray mode
    return (!off && c->n_samples % unit == 0 && c->n_rays % waves == 0) ? c->n_samples / unit : 0;
}

// ---- weight-gradient plan -------------------------------------------------------------------------------------------
struct Unit {  // a wave tile before the split over samples
    WgradJob j;
    int group;  // units of one group share operands: same-k jobs are placed in one workgroup
};

std::vector<Unit> wgrad_units(int D) {
    std::vector<Unit> u;
    const int nb = D / 128;  // 128-wide blocks per D
    int group = 0;
    auto add = [&](int layer, int MI, int NI, int dpl, int dcol, int dvalid, int xpl, int xcol, int xvalid, int row0, int wcol0,
                   int rows_real, int cols_real, int ldw, int bias) {
        Unit x{};
        x.j = WgradJob{layer, MI, NI, dpl, dcol, dvalid, xpl, xcol, xvalid, row0, wcol0, rows_real, cols_real, ldw, 0, 0, bias, 0, -1, 0};
        x.group = group;
        u.push_back(x);
    };
    // D x D layers: hidden 2,3,4,6,7,8 (params 1,2,3,5,6,7) and the h-part of hidden 5 (param 4), feature (param 9)
    auto dxd = [&](int layer, int dpl, int xpl, int ldw, int cols_real) {
        for (int a = 0; a < nb; ++a)
            for (int b = 0; b < nb; ++b)
                add(layer, 4, 4, dpl, 128 * a, D - 128 * a, xpl, 128 * b, D - 128 * b, 128 * a, 128 * b, D, cols_real, ldw,
                    nb == 2 ? 2 + b : 1);   // the two tiles of a row block share d(bias): even / odd sample pairs
        ++group;
    };
    // posenc-input parts: hidden 1 (param 0) and the e-part of hidden 5 (param 4, columns D..D+62)
    auto dxe = [&](int layer, int dpl, int wcol0, int ldw, int cols_real, int bias) {
        for (int a = 0; a < nb; ++a)
            add(layer, 4, 2, dpl, 128 * a, D - 128 * a, P_XE, 0, kPosPad, 128 * a, wcol0, D, cols_real, ldw, bias);
        ++group;
    };
    dxe(0, P_DH1 + 0, 0, kPosReal, kPosReal, 1);
    dxd(1, P_DH1 + 1, P_XH1 + 0, D, D);
    dxd(2, P_DH1 + 2, P_XH1 + 1, D, D);
    dxd(3, P_DH1 + 3, P_XH1 + 2, D, D);
    dxd(4, P_DH1 + 4, P_XH1 + 3, D + kPosReal, D + kPosReal);
    dxe(4, P_DH1 + 4, D, D + kPosReal, D + kPosReal, 0);
    dxd(5, P_DH1 + 5, P_XH1 + 4, D, D);
    dxd(6, P_DH1 + 6, P_XH1 + 5, D, D);
    dxd(7, P_DH1 + 7, P_XH1 + 6, D, D);
    // density head (param 8): 1 x D, gradient operand = column 3 of the per-sample output gradients
    for (int b = 0; b < nb; ++b) add(8, 1, 4, P_DOUT4, 3, 1, P_XH1 + 7, 128 * b, D - 128 * b, 0, 128 * b, 1, D, D, b == 0);
    ++group;
    // colour hidden: the merged matrix W' = Wg[:, :D] Wf (D/2 x D, pseudo-parameter kMergedLayer) against hidden 8, and the
    // direction-encoding columns of param 10 (D/2 x 27 at column D).  dWf, dWg[:, :D], dbf follow from dW', db' in the
    // un-merge step after the reduction (nnr_wgrad.hip).
    const int mi_g = D == 256 ? 4 : 2;
    for (int b = 0; b < nb; ++b)
        add(kMergedLayer, mi_g, 4, P_DG, 0, D / 2, P_XH1 + 7, 128 * b, D - 128 * b, 0, 128 * b, D / 2, D, D, b == 0);
    add(10, mi_g, 1, P_DG, 0, D / 2, P_XF, 0, kDirPad, 0, D, D / 2, D + kDirReal, D + kDirReal, 0);
    ++group;
    // rgb (param 11): 3 x D/2, gradient operand = columns 0..2 of the per-sample output gradients
    add(11, 1, D == 256 ? 4 : 2, P_DOUT4, 0, 3, P_XG, 0, D / 2, 0, 0, 3, D / 2, D / 2, 1);
    ++group;
    return u;
}

constexpr int kMaxBlocks = 256;   // one 4-wave workgroup per CU: the kernel needs the whole register file, and every
                                  // workgroup must be resident at once (a 257th would run as a second round)
constexpr int kGranule = 16;      // samples per loop iteration of the wgrad kernel (two stages of kU = 4 sample pairs)
constexpr int kMinGranulesPerBlock = 16;  // small problems use fewer workgroups: a job costs a 64 KB slot + its flush (~2 us),
                                          // a granule of a 4x4 tile ~3.4 us of MFMA, so 16 granules keep the flush under 4 %

struct Plan {
    std::vector<WgradJob> jobs;        // grouped by wave: wave w runs jobs [wave_first[w], wave_first[w+1])
    std::vector<int32_t> wave_first;   // n_waves + 1 entries, n_waves a multiple of 4
    std::vector<int32_t> heads;        // job index of split 0 of every tile: the reduction kernel launches 16 workgroups per HEAD, not per job
};

// Balanced static schedule.  Work is measured in cost-granules (MI*NI MFMAs-per-sample-pair x 16 samples).  The D x D
// layers (4 tiles of 4x4 that share their two operand column halves) are scheduled per WORKGROUP: the 8 layers form one
// tape of (layer, granule) positions that is cut into equal spans, a span crossing a layer boundary becoming two segments
// whose four tiles go to the four waves -- same sample range in one CU, so the operand re-reads hit L1/L2.  Everything else
// (posenc parts, density, colour, rgb: 14 % of the work) is scheduled per WAVE on a second tape weighted by tile cost.
// Each job flushes to its own slot; the splits of a tile are chained (next_split) for the reduction kernel.
Plan build_plan(const nnr_cfg* c) {
    const WsLayout w = ws_layout(c);
    const std::vector<Unit> units = wgrad_units(c->hidden);
    const int64_t granules = w.S_pad / kGranule;
    std::vector<std::vector<int>> groups;   // class A: groups of four 4x4 tiles
    std::vector<int> small;                  // class B: unit indices
    for (size_t i = 0; i < units.size();) {
        size_t e = i;
        while (e < units.size() && units[e].group == units[i].group) ++e;
        bool dxd = (e - i) == 4;
        for (size_t t = i; t < e; ++t) dxd = dxd && units[t].j.MI == 4 && units[t].j.NI == 4;
        if (dxd) groups.push_back({(int)i, (int)i + 1, (int)i + 2, (int)i + 3});
        else
            for (size_t t = i; t < e; ++t) small.push_back((int)t);
        i = e;
    }
    // Measured cycles per cost-granule relative to a 4x4 tile (tools/timeline.py, MI355X): narrow tiles issue the same
    // loads for fewer MFMAs.  Weights in 1/1000.
    // Three-term mode (nnr_wgrad.hip, wgrad_job_split): the 4 x 4 tiles run on the bf16 matrix pipe, the narrow ones still on fp32 MFMAs --
    // per MFMA-equivalent a 4 x 4 tile costs `split_w` / 1000 of what it costs in fp32 (measured: NNR_WGRAD_SPLIT_WEIGHT sweeps).
    static const int split_w = [] {
        const char* e = std::getenv("NNR_WGRAD_SPLIT_WEIGHT");
        return e ? std::max(50, std::atoi(e)) : 440;      // (round 4, shared split, row-major activations: 1.18 / 1.16 / 1.14 / 1.11 / 1.12 ms at 360 /
                                                         // 400 / 440 / 480 / 520; both operands tile-major: 1.12 / 1.11 / 1.10 / 1.11 at 380 / 420 / 440 / 460, 1.14 at 480 on
                                                         // another box where 440 gave 1.11 -- profiles/r04/r*_wgrad_weight_sweep_tile_x.txt)
    }();
    // two-term mode: the workgroup jobs take three fp16 MFMAs per product instead of six bf16 ones (wgrad_group_split2): cheaper again, relative to a narrow fp32 tile
    static const int split2_w = [] {
        const char* e = std::getenv("NNR_WGRAD_SPLIT2_WEIGHT");
        return e ? std::max(50, std::atoi(e)) : 340;      // (profiles/r06/h_wgrad_f16_plan_weight_sweep.txt, in sequence with the other kernels: 0.974 / 0.941 / 0.926 /
                                                         // 0.918 / 0.947 ms at 280 / 300 / 320 / 340 / 360)
    }();
    static const bool f16_off = std::getenv("NNR_WGRAD_BF16_TERMS") != nullptr;      // (= nnr_wgrad.hip's: the six-term workgroup jobs in the two-term mode)
    // two-term mode: the 128 x 64 tiles against the position encoding as private two-term jobs (wgrad_job_enc2) -- their weight relative to a narrow
    // fp32 tile; 0: leave them on fp32 MFMAs
    static const int enc2_w = [] {
        const char* e = std::getenv("NNR_WGRAD_ENC2_WEIGHT");
        return e ? std::max(0, std::atoi(e)) : 625;      // (profiles/r06/t2_wgrad_enc2_weight_sweep.txt, kernel in sequence: 0.930 ms without the jobs; 0.915 / 0.879 /
                                                          // 0.884 / 0.894 / 0.897 at 550 / 600 / 650 / 700 / 750 -- a cliff below the job's true cost, a gentle slope above)
    }();
    static const bool env_fp32 = std::getenv("NNR_WGRAD_FP32") != nullptr;      // (every knob of the plan is read ONCE per process, here: a plan built
                                                                                // under one setting never meets a launch that assumes another)
    const bool split = is_split3(c) && !env_fp32;
    const bool f16_groups = split && is_split2(c) && !f16_off && c->hidden == 256;      // (class-A groups exist at D = 256 only)
    const bool enc2 = f16_groups && enc2_w > 0;
    auto is_enc2 = [enc2](const WgradJob& j) { return enc2 && j.MI == 4 && j.NI == 2 && j.x_plane == P_XE; };
    auto weight = [split, f16_groups, is_enc2](const WgradJob& j) -> int64_t {
        const int mn = j.MI * j.NI;
        if (is_enc2(j)) return (int64_t)mn * enc2_w;      // (d(bias) rides in the split: no surcharge)
        // (the merged layer's two 4 x 4 tiles are class B: private six-term split in either mode; a class-A tile is recognised by its layer: hidden 2..8)
        const bool group_tile = mn == 16 && j.layer >= 1 && j.layer <= 7;
        int w = mn == 16 ? (split ? (f16_groups && group_tile ? split2_w : split_w) : 1000) : mn == 8 ? 1035 : mn == 4 ? 1145 : 1250;
        if (j.bias == 1) w += mn == 16 ? 20 : mn == 8 ? 42 : 20;
        return (int64_t)mn * w;
    };
    int64_t cost_a = 0, cost_b = 0;
    for (auto& g : groups) cost_a += 4 * weight(units[g[0]].j);   // bias halves: all four tiles of a group weigh the same
    for (int u : small) cost_b += weight(units[u].j);
    // Class B at D = 256 in BUNDLES (round 6, OFF unless NNR_WGRAD_BUNDLES is set -- a measured negative): the four waves of a workgroup take
    // narrow tiles that read the same planes over the SAME sample range at the same time, so a plane comes from HBM once and the other
    // readers find it in the CU's L1 / the XCD's L2.  On the per-wave tape below the tiles of one plane run on different workgroups -- other
    // XCDs, other L2s -- and the planes they share are fetched once per tile: 4.26 GB per launch against 3.4 GB of distinct planes.
    //   bundle 0: hidden-1 tiles a, b | skip layer's encoding tiles a, b             (share the position encoding)
    //   bundle 1: merged colour tiles a, b | density a + rgb | density b + direction   (share hidden 8, d colour-hidden, the 4-wide gradients)
    // A bundle costs its heaviest wave (7.0 / 7.0 / 9.2 / 9.2 in bundle 1: 12 % of those workgroups' time idle).  Measured, 1024 x 192
    // (profiles/r06/q_wgrad_bundles_ab.txt): fetched bytes 4.26 -> 4.04 GB, kernel 0.893 -> 0.915 ms.  The kernel is not waiting on those
    // bytes; the perfectly balanced tape wins.
    static const bool no_bundles = std::getenv("NNR_WGRAD_BUNDLES") == nullptr;
    std::vector<std::array<std::vector<int>, 4>> bundles;
    if (c->hidden == 256 && !no_bundles && !groups.empty()) {
        auto find = [&](int layer, int row0, int wcol0) {
            for (int u : small)
                if (units[u].j.layer == layer && units[u].j.row0 == row0 && units[u].j.wcol0 == wcol0) return u;
            return -1;
        };
        const int D = c->hidden;
        bundles.push_back({{{find(0, 0, 0)}, {find(0, 128, 0)}, {find(4, 0, D)}, {find(4, 128, D)}}});
        bundles.push_back({{{find(kMergedLayer, 0, 0)}, {find(kMergedLayer, 0, 128)}, {find(8, 0, 0), find(11, 0, 0)}, {find(8, 0, 128), find(10, 0, D)}}});
        size_t n = 0;
        bool ok = true;
        for (auto& b : bundles)
            for (auto& wv : b)
                for (int u : wv) { ok = ok && u >= 0; ++n; }
        if (!ok || n != small.size()) bundles.clear();      // (a unit list this table does not know: the per-wave tape)
    }
    std::vector<int64_t> bundle_cost;
    if (!bundles.empty()) {
        cost_b = 0;
        for (auto& b : bundles) {
            int64_t mx = 0;
            for (auto& wv : b) {
                int64_t s = 0;
                for (int u : wv) s += weight(units[u].j);
                mx = std::max(mx, s);
            }
            bundle_cost.push_back(mx);
            cost_b += 4 * mx;      // in wave-equivalents, like cost_a
        }
    }
    static const int max_blocks = [] {   // NNR_WGRAD_MAX_BLOCKS: tuning knob for experiments
        const char* e = std::getenv("NNR_WGRAD_MAX_BLOCKS");
        return e ? std::max(2, std::atoi(e)) : kMaxBlocks;
    }();
    // Workgroups: one per kMinGranulesPerBlock granules of a D x D-at-256 group's worth of work (four 4x4 tiles), counted over
    // BOTH classes -- at D = 128 every unit is class B (a D x D layer is a single tile there), and sizing the launch by the
    // class-A groups alone left that whole configuration on one workgroup.
    const int64_t group_cost = 4 * 16 * 1000;
    const int64_t group_equiv = std::max<int64_t>(1, (cost_a + cost_b + group_cost / 2) / group_cost);
    const int n_blocks = (int)std::max<int64_t>(2, std::min<int64_t>(max_blocks, granules * group_equiv / kMinGranulesPerBlock));
    int nb_b = (int)((cost_b * n_blocks + (cost_a + cost_b) / 2) / (cost_a + cost_b));
    nb_b = groups.empty() ? n_blocks : std::max(1, std::min(n_blocks - 1, nb_b));
    const int nb_a = n_blocks - nb_b;

    std::vector<std::vector<WgradJob>> per_wave((size_t)n_blocks * 4);
    auto emit = [&](int wave, int unit, int64_t g0, int64_t g1) {
        if (g1 <= g0) return;
        WgradJob j = units[unit].j;
        j.k0 = (int32_t)(g0 * kGranule);
        j.k1 = (int32_t)(g1 * kGranule);
        j.split = unit;   // temporarily: the tile id, replaced by the split index below
        per_wave[wave].push_back(j);
    };
    // class A
    const int64_t tape_a = granules * (int64_t)groups.size();
    for (int b = 0; b < nb_a; ++b) {
        const int64_t a0 = tape_a * b / nb_a, a1 = tape_a * (b + 1) / nb_a;
        for (int64_t g = a0 / granules; g <= (a1 - 1) / granules && a1 > a0; ++g) {
            const int64_t lo = std::max(a0, g * granules) - g * granules, hi = std::min(a1, (g + 1) * granules) - g * granules;
            for (int t = 0; t < 4; ++t) emit(4 * b + t, groups[g][t], lo, hi);
        }
    }
    // Three-term mode: the four tiles of a class-A segment sit in one workgroup over ONE sample range -- the kernel runs them as a workgroup
    // job in which every operand value is split once (wgrad_group_split, nnr_wgrad.hip: barriers inside, so all four waves must be there)
    static const bool coop = std::getenv("NNR_WGRAD_NO_COOP") == nullptr;
    if (split && coop)
        for (int wv = 0; wv < 4 * nb_a; ++wv)
            for (auto& j : per_wave[wv]) j.reserved = 1;
    // class B in bundles: bundle i occupies [off_i, off_i + cost_i * granules) of a tape that is cut per WORKGROUP
    if (!bundles.empty()) {
        int64_t tape = 0;
        for (int64_t cb : bundle_cost) tape += cb * granules;
        int64_t off = 0;
        for (size_t i = 0; i < bundles.size(); ++i) {
            const int64_t cb = bundle_cost[i], end = off + cb * granules;
            auto to_granule = [&](int64_t x) { return std::min(granules, std::max<int64_t>(0, (x - off + cb / 2) / cb)); };
            for (int b = 0; b < nb_b; ++b) {
                const int64_t c0 = tape * b / nb_b, c1 = tape * (b + 1) / nb_b;
                if (c1 <= off || c0 >= end) continue;
                const int64_t g0 = c0 <= off ? 0 : to_granule(c0), g1 = c1 >= end ? granules : to_granule(c1);
                for (int t = 0; t < 4; ++t)
                    for (int u : bundles[i][t]) emit(4 * (nb_a + b) + t, u, g0, g1);
            }
            off = end;
        }
    }
    // class B per wave (D = 128, or the bundles switched off): tile u occupies [off_u, off_u + cost_u * granules) of the tape; a cut inside a tile
    // is rounded to a granule
    const int nw_b = nb_b * 4;
    const int64_t tape_b = cost_b * granules;
    std::vector<int64_t> cuts((size_t)nw_b + 1);
    for (int v = 0; v <= nw_b; ++v) cuts[v] = tape_b * v / nw_b;
    int64_t off = 0;
    for (int u : bundles.empty() ? small : std::vector<int>{}) {
        const int64_t cu = weight(units[u].j), end = off + cu * granules;
        auto to_granule = [&](int64_t x) { return std::min(granules, std::max<int64_t>(0, (x - off + cu / 2) / cu)); };
        for (int v = 0; v < nw_b; ++v) {
            if (cuts[v + 1] <= off || cuts[v] >= end) continue;
            const int64_t g0 = cuts[v] <= off ? 0 : to_granule(cuts[v]);
            const int64_t g1 = cuts[v + 1] >= end ? granules : to_granule(cuts[v + 1]);
            emit(4 * nb_a + v, u, g0, g1);
        }
        off = end;
    }
    for (auto& v : per_wave)
        for (auto& j : v)
            if (is_enc2(j)) j.reserved = 2;
    // flatten by wave, then chain the splits of every tile in sample order
    Plan p;
    p.wave_first.push_back(0);
    for (auto& v : per_wave) {
        for (auto& j : v) p.jobs.push_back(j);
        p.wave_first.push_back((int32_t)p.jobs.size());
    }
    std::vector<std::vector<int>> by_tile(units.size());
    for (size_t i = 0; i < p.jobs.size(); ++i) by_tile[p.jobs[i].split].push_back((int)i);
    for (auto& v : by_tile) {
        std::sort(v.begin(), v.end(), [&](int x, int y) { return p.jobs[x].k0 < p.jobs[y].k0; });
        for (size_t s = 0; s < v.size(); ++s) {
            p.jobs[v[s]].split = (int32_t)s;
            p.jobs[v[s]].next_split = s + 1 < v.size() ? v[s + 1] : -1;
        }
        if (!v.empty()) p.heads.push_back(v[0]);
    }
    return p;
}

// blob: WgradJob[n_jobs], int32 wave_first[n_waves + 1], int32 n_heads, int32 heads[n_heads]
constexpr int32_t kPlanMagic = 0x4e4e5235;      // 'NNR5': the trailer of the fp32 / three-term plan blob (= nnr_wgrad.hip's)
size_t plan_bytes(const Plan& p) { return p.jobs.size() * sizeof(WgradJob) + (p.wave_first.size() + 1 + p.heads.size() + 4) * sizeof(int32_t); }

// ---- weight-gradient plan of the bf16 training mode (nnr_wgrad_bf16.hip) ------------------------------------------------------
// Units = the products dW = Dlt^T X of the 12 layers (the feature layer merged into the colour-hidden one, the density head riding
// on the merged unit's extra gradient group; at D = 256 the skip layer and the colour-hidden layer take their two input planes --
// hidden | encoding -- in one unit, at D = 128 as two units), each with its tiling over the four waves of a workgroup.  The kernel is bound by streaming the operands once, so a unit's cost per 32-sample chunk is the KiB it
// stages; the units form one tape of (unit, chunk) positions that is cut into equal spans, one per workgroup (a span that
// crosses a unit boundary becomes two jobs).  Outputs = where the rectangles of a unit's product go.
struct BUnit {
    int d_plane, d_g0, d_groups, x_plane, x_g0, x_groups, MT, NT, WR, WC, bias;
    int x2_plane = -1, x2_groups = 0;     // a second activation plane behind the first (x_groups even): one pass over the gradient
};
struct BPlan {
    std::vector<WgradJobB> jobs;
    std::vector<int32_t> block_first;   // n_blocks + 1
    std::vector<WgradOutB> outs;
};

void bf16_units(int D, std::vector<BUnit>& units, std::vector<WgradOutB>& outs) {
    const int G = D / 16, Gh = D / 32;            // groups of a D-wide / D/2-wide plane
    const bool big = D == 256;
    static const bool no_merge = std::getenv("NNR_WGRAD_NO_MERGE") != nullptr;   // profiling knob: the two-plane units as separate passes
    const bool merge = big && !no_merge;
    auto out = [&](int unit, int layer, int d_row, int n_rows, int w_row, int x_col, int n_cols, int w_col, int ldw, int bias) {
        const BUnit& u = units[unit];
        outs.push_back(WgradOutB{unit, layer, d_row, n_rows, w_row, x_col, n_cols, w_col, ldw, bias, -1, u.MT, u.NT, u.WR, u.WC, 0});
    };
    auto dxd = [&](int layer, int dpl, int xpl, int ldw) {          // D x D: 256 -> four waves of 4 x 4 tiles, 128 -> of 2 x 2
        units.push_back(BUnit{dpl, 0, G, xpl, 0, G, big ? 4 : 2, big ? 4 : 2, 2, 2, 1});
        out((int)units.size() - 1, layer, 0, D, 0, 0, D, 0, ldw, 1);
    };
    auto dxe = [&](int layer, int dpl, int w_col, int ldw, int bias) {   // D x 63 against the bf16 copy of the position encoding
        units.push_back(BUnit{dpl, 0, G, P_XE16, 0, kPosPad / 16, big ? 2 : 1, 2, 4, 1, bias});
        out((int)units.size() - 1, layer, 0, D, 0, 0, kPosReal, w_col, ldw, bias);
    };
    dxe(0, P_DH1 + 0, 0, kPosReal, 1);
    dxd(1, P_DH1 + 1, P_XH1 + 0, D);
    dxd(2, P_DH1 + 2, P_XH1 + 1, D);
    dxd(3, P_DH1 + 3, P_XH1 + 2, D);
    if (merge) {      // skip layer, input = hidden 4 | position encoding: 8 x 10 tiles as four waves of 4 x 5, the gradient read once
        units.push_back(BUnit{P_DH1 + 4, 0, G, P_XH1 + 3, 0, G, 4, 5, 2, 2, 1, P_XE16, kPosPad / 16});
        out((int)units.size() - 1, 4, 0, D, 0, 0, D, 0, D + kPosReal, 1);
        out((int)units.size() - 1, 4, 0, D, 0, D, kPosReal, D, D + kPosReal, 0);
    } else {
        dxd(4, P_DH1 + 4, P_XH1 + 3, D + kPosReal);
        dxe(4, P_DH1 + 4, D, D + kPosReal, 0);
    }
    dxd(5, P_DH1 + 5, P_XH1 + 4, D);
    dxd(6, P_DH1 + 6, P_XH1 + 5, D);
    dxd(7, P_DH1 + 7, P_XH1 + 6, D);
    // merged colour-hidden matrix W' (D/2 x D) and the density row: gradient operand = P_DG groups 0..Gh (the last group holds
    // d rgb_pre[0..2], d sigma_raw), activation operand = hidden 8; the direction-encoding columns of the colour-hidden layer are
    // the same gradient against the encoding
    if (merge) {      // 5 x 9 tiles: three waves of 5 x 3 (the fourth only moves data)
        units.push_back(BUnit{P_DG, 0, Gh + 1, P_XH1 + 7, 0, G, 5, 3, 1, 3, 1, P_XF16, kDirPad / 16});
        out((int)units.size() - 1, kMergedLayer, 0, D / 2, 0, 0, D, 0, D, 1);
        out((int)units.size() - 1, 8, D / 2 + 3, 1, 0, 0, D, 0, D, 1);
        out((int)units.size() - 1, 10, 0, D / 2, 0, D, kDirReal, D, D + kDirReal, 0);
    } else {
        units.push_back(BUnit{P_DG, 0, Gh + 1, P_XH1 + 7, 0, G, big ? 5 : 3, big ? 2 : 1, 1, 4, 1});
        out((int)units.size() - 1, kMergedLayer, 0, D / 2, 0, 0, D, 0, D, 1);
        out((int)units.size() - 1, 8, D / 2 + 3, 1, 0, 0, D, 0, D, 1);
        units.push_back(BUnit{P_DG, 0, Gh, P_XF16, 0, kDirPad / 16, 1, 1, big ? 4 : 2, 1, 0});
        out((int)units.size() - 1, 10, 0, D / 2, 0, 0, kDirReal, D, D + kDirReal, 0);
    }
    // rgb head: the 3 output-gradient rows against the colour-hidden activations
    units.push_back(BUnit{P_DG, Gh, 1, P_XG, 0, Gh, 1, 1, 1, big ? 4 : 2, 1});
    out((int)units.size() - 1, 11, 0, 3, 0, 0, D / 2, 0, D / 2, 1);
}

constexpr int64_t kMinTapePerBlock = 16 * 32;   // KiB: at least ~16 full stages per workgroup, or the ring never fills

BPlan build_plan_bf16(const nnr_cfg* c) {
    const WsLayout w = ws_layout(c);
    BPlan p;
    std::vector<BUnit> units;
    bf16_units(c->hidden, units, p.outs);
    const int64_t chunks = w.S_pad / 32;
    std::vector<int64_t> cost(units.size()), start(units.size() + 1, 0);
    for (size_t u = 0; u < units.size(); ++u) {
        cost[u] = units[u].d_groups + units[u].x_groups + units[u].x2_groups;
        start[u + 1] = start[u] + cost[u] * chunks;
    }
    const int64_t tape = start[units.size()];
    static const int max_blocks = [] {
        const char* e = std::getenv("NNR_WGRAD_MAX_BLOCKS");
        return e ? std::max(1, std::atoi(e)) : kMaxBlocks;
    }();
    const int n_blocks = (int)std::max<int64_t>(1, std::min<int64_t>(max_blocks, tape / kMinTapePerBlock));
    // cut u-th unit at chunk boundaries: position x on the tape inside unit u -> chunk round((x - start[u]) / cost[u])
    auto cut_chunk = [&](size_t u, int64_t x) {
        if (x <= start[u]) return (int64_t)0;
        if (x >= start[u + 1]) return chunks;
        return std::min(chunks, (x - start[u] + cost[u] / 2) / cost[u]);
    };
    p.block_first.push_back(0);
    for (int b = 0; b < n_blocks; ++b) {
        const int64_t lo = tape * b / n_blocks, hi = tape * (b + 1) / n_blocks;
        for (size_t u = 0; u < units.size(); ++u) {
            if (hi <= start[u] || lo >= start[u + 1]) continue;
            const int64_t c0 = cut_chunk(u, lo), c1 = cut_chunk(u, hi);
            if (c1 <= c0) continue;
            const BUnit& un = units[u];
            int dp = 0, xp = 0;
            int x2p = 0;
            const int64_t d_off = w.plane(un.d_plane, &dp), x_off = w.plane(un.x_plane, &xp);   // floats; pitch = floats per sample
            const int64_t x2_off = un.x2_groups ? w.plane(un.x2_plane, &x2p) : 0;
            p.jobs.push_back(WgradJobB{4 * d_off + 1024ll * un.d_g0, 4 * x_off + 1024ll * un.x_g0, 4 * 32 * dp, 4 * 32 * xp, un.d_groups,
                                       un.x_groups, (int32_t)u, un.MT, un.NT, un.WR, un.WC, (int32_t)c0, (int32_t)c1, un.bias, 0, -1,
                                       4 * x2_off, 4 * 32 * x2p, un.x2_groups});
        }
        p.block_first.push_back((int32_t)p.jobs.size());
    }
    // chain the jobs of every unit in sample order (they are generated in that order)
    std::vector<int> last(units.size(), -1), count(units.size(), 0), first(units.size(), -1);
    for (size_t j = 0; j < p.jobs.size(); ++j) {
        const int u = p.jobs[j].unit;
        p.jobs[j].split = count[u]++;
        if (last[u] >= 0) p.jobs[last[u]].next_split = (int32_t)j;
        else first[u] = (int)j;
        last[u] = (int)j;
    }
    for (auto& o : p.outs) o.first_job = first[o.unit];
    return p;
}

size_t plan_bytes(const BPlan& p) {
    return p.jobs.size() * sizeof(WgradJobB) + p.block_first.size() * sizeof(int32_t) + p.outs.size() * sizeof(WgradOutB);
}

}  // namespace

// ---- in-step kernel timing -------------------------------------------------------------------------------------------------
// bench.py switches this on for its timed steps: every launch of a main MLP kernel is bracketed by two HIP events recorded on the
// launch stream, so the durations are those of the kernels INSIDE the training step (between the step's other kernels, at the clocks
// and cache state of the step), not of a kernel run back to back with itself.  One process, one stream at a time: plain globals.
namespace {
struct ProfState {
    bool on = false;
    int cap = 0;
    int n[nnr::PROF_KINDS] = {0, 0, 0, 0};
    std::vector<hipEvent_t> ev[nnr::PROF_KINDS][2];
} g_prof;
}  // namespace
namespace nnr {
void prof_before(int kind, hipStream_t st) {
    if (g_prof.on && g_prof.n[kind] < g_prof.cap) (void)hipEventRecord(g_prof.ev[kind][0][g_prof.n[kind]], st);
}
void prof_after(int kind, hipStream_t st) {
    if (g_prof.on && g_prof.n[kind] < g_prof.cap) (void)hipEventRecord(g_prof.ev[kind][1][g_prof.n[kind]++], st);
}
}  // namespace nnr


static_assert(nnr::kFlagDistAlpha == NNR_F_DIST_ALPHA && nnr::kFlagWhiteBg == NNR_F_WHITE_BG && nnr::kFlagReluSigma == NNR_F_RELU_SIGMA,
              "the device-side copies of the rendering switches (nnr_device.h) must equal include/nnr.h");

extern "C" {

int nnr_abi_version(void) { return NNR_ABI_VERSION; }

const char* nnr_strerror(int code) {
    switch (code) {
        case NNR_OK: return "ok";
        case NNR_E_BADCFG: return "bad configuration or null pointer";
        case NNR_E_UNSUPPORTED: return "unsupported configuration (hidden must be 128 or 256; N <= 1024 when training)";
        case NNR_E_ALIGN: return "pointer not 16-byte aligned";
        case NNR_E_HIP: return "HIP runtime error";
        default: return "unknown error";
    }
}

int nnr_last_hip_error(void) { return g_last_hip; }

int nnr_prof_begin(int32_t max_launches) {
    if (g_prof.on || max_launches <= 0 || max_launches > 4096) return NNR_E_BADCFG;
    for (int k = 0; k < nnr::PROF_KINDS; ++k) {
        g_prof.n[k] = 0;
        for (int s = 0; s < 2; ++s) {
            g_prof.ev[k][s].resize(max_launches);
            for (auto& e : g_prof.ev[k][s])
                if (hipEventCreate(&e) != hipSuccess) return NNR_E_HIP;
        }
    }
    g_prof.cap = max_launches;
    g_prof.on = true;
    return NNR_OK;
}

int nnr_prof_end(float* mean_ms4, int32_t* launches4) {
    if (!g_prof.on || !mean_ms4 || !launches4) return NNR_E_BADCFG;
    g_prof.on = false;
    int rc = NNR_OK;
    for (int k = 0; k < nnr::PROF_KINDS; ++k) {
        double sum = 0.0;
        for (int i = 0; i < g_prof.n[k]; ++i) {
            float ms = 0.f;
            if (hipEventSynchronize(g_prof.ev[k][1][i]) != hipSuccess || hipEventElapsedTime(&ms, g_prof.ev[k][0][i], g_prof.ev[k][1][i]) != hipSuccess)
                rc = NNR_E_HIP;
            sum += ms;
        }
        launches4[k] = g_prof.n[k];
        mean_ms4[k] = g_prof.n[k] ? (float)(sum / g_prof.n[k]) : 0.f;
        for (int s = 0; s < 2; ++s) {
            for (auto& e : g_prof.ev[k][s]) (void)hipEventDestroy(e);
            g_prof.ev[k][s].clear();
        }
    }
    return rc;
}

size_t nnr_packed_floats(const nnr_cfg* cfg) {
    if (check_cfg(cfg) != NNR_OK) return 0;
    return packed_floats(cfg->hidden, weight_mode(cfg));
}

size_t nnr_workspace_floats(const nnr_cfg* cfg) {
    if (check_cfg(cfg) != NNR_OK) return 0;
    const WsLayout w = ws_layout(cfg);
    // training: the planes, one partial slot per weight-gradient job (bf16 mode: four wave slots per workgroup job), then dW'
    // (D/2 x D) and db' (D/2)
    const size_t merged = (size_t)(cfg->hidden / 2) * cfg->hidden + cfg->hidden / 2;
    if (!w.train) return (size_t)w.total();
    const size_t slots = w.bf16 ? build_plan_bf16(cfg).jobs.size() * 4 * (size_t)kSlotBFloats : build_plan(cfg).jobs.size() * (size_t)kSlotFloats;
    return (size_t)w.total() + slots + merged + (is_split2(cfg) ? kPlaneMaxFloats : 0);      // two-term mode: the planes' maxima behind everything else
}

int64_t nnr_ws_plane(const nnr_cfg* cfg, int pl, int32_t* pitch_out) {
    if (check_cfg(cfg) != NNR_OK) return -1;
    int pitch = 0;
    const int64_t o = ws_layout(cfg).plane(pl, &pitch);
    if (pitch_out) *pitch_out = pitch;
    return pl < 0 ? -1 : o;
}

int nnr_ws_plane_layout(const nnr_cfg* cfg, int pl) {
    if (check_cfg(cfg) != NNR_OK) return -1;
    const WsLayout w = ws_layout(cfg);
    int pitch = 0;
    if (pl < 0 || w.plane(pl, &pitch) < 0) return -1;
    if (w.tiled(pl)) return 2;
    if (w.bf16 && ((pl >= P_XH1 && pl < P_XH1 + 8) || pl == P_XG || pl == P_XE16 || pl == P_XF16 || (pl >= P_DH1 && pl < P_DH1 + 8) || pl == P_DG)) return 1;
    return 0;
}

size_t nnr_plan_bytes(const nnr_cfg* cfg) {
    if (check_cfg(cfg) != NNR_OK) return 0;
    if (is_bf16(cfg)) return plan_bytes(build_plan_bf16(cfg));
    return plan_bytes(build_plan(cfg));
}

int nnr_plan_counts(const nnr_cfg* cfg, int32_t* n_jobs, int32_t* n_waves) {
    int rc = check_cfg(cfg);
    if (rc != NNR_OK) return rc;
    if (is_bf16(cfg)) {   // bf16 mode: workgroup jobs, four waves per workgroup
        const BPlan p = build_plan_bf16(cfg);
        if (n_jobs) *n_jobs = (int32_t)p.jobs.size();
        if (n_waves) *n_waves = 4 * ((int32_t)p.block_first.size() - 1);
        return NNR_OK;
    }
    const Plan p = build_plan(cfg);
    if (n_jobs) *n_jobs = (int32_t)p.jobs.size();
    if (n_waves) *n_waves = (int32_t)p.wave_first.size() - 1;
    return NNR_OK;
}

int nnr_plan_build(const nnr_cfg* cfg, void* plan_host) {
    int rc = check_cfg(cfg);
    if (rc != NNR_OK) return rc;
    if (!plan_host) return NNR_E_BADCFG;
    if (is_bf16(cfg)) {   // layout: WgradJobB[n_jobs], int32 block_first[n_blocks + 1], WgradOutB[n_outs]
        const BPlan p = build_plan_bf16(cfg);
        char* out = static_cast<char*>(plan_host);
        std::memcpy(out, p.jobs.data(), p.jobs.size() * sizeof(WgradJobB));
        out += p.jobs.size() * sizeof(WgradJobB);
        std::memcpy(out, p.block_first.data(), p.block_first.size() * sizeof(int32_t));
        out += p.block_first.size() * sizeof(int32_t);
        std::memcpy(out, p.outs.data(), p.outs.size() * sizeof(WgradOutB));
        return NNR_OK;
    }
    const Plan p = build_plan(cfg);   // layout: WgradJob[n_jobs], then int32 wave_first[n_waves + 1]
    std::memcpy(plan_host, p.jobs.data(), p.jobs.size() * sizeof(WgradJob));
    char* tail = static_cast<char*>(plan_host) + p.jobs.size() * sizeof(WgradJob);
    std::memcpy(tail, p.wave_first.data(), p.wave_first.size() * sizeof(int32_t));
    tail += p.wave_first.size() * sizeof(int32_t);
    const int32_t n_heads = (int32_t)p.heads.size();
    std::memcpy(tail, &n_heads, sizeof(int32_t));
    std::memcpy(tail + sizeof(int32_t), p.heads.data(), p.heads.size() * sizeof(int32_t));
    // trailer: what the weight-gradient kernel checks before it trusts the blob (a blob of another ABI, shape or plan setting makes it trap
    // instead of indexing the job table with garbage)
    const int32_t trailer[4] = {kPlanMagic, (int32_t)p.jobs.size(), (int32_t)p.wave_first.size() - 1, n_heads};
    std::memcpy(tail + sizeof(int32_t) * (1 + p.heads.size()), trailer, sizeof(trailer));
    return NNR_OK;
}

int nnr_pack_weights(const nnr_cfg* cfg, const nnr_params* p, float* packed, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != NNR_OK) return rc;
    if (!p || !packed) return NNR_E_BADCFG;
    if (!aligned16(packed)) return NNR_E_ALIGN;
    PackArgs a;
    for (int i = 0; i < 12; ++i) {
        if (!p->weight[i] || !p->bias[i]) return NNR_E_BADCFG;
        a.w[i] = p->weight[i];
        a.b[i] = p->bias[i];
    }
    a.packed = packed;
    hipError_t e = launch_pack(cfg->hidden, a, weight_mode(cfg), (hipStream_t)stream);
    return e == hipSuccess ? NNR_OK : hip_fail(e);
}

// NNR_F_SPLIT2 training: where the table of plane maxima sits in the workspace (behind planes, slots and the merged-layer scratch: nnr_workspace_floats)
static float* plane_max_of(const nnr_cfg* cfg, float* ws) {
    const WsLayout w = ws_layout(cfg);
    const size_t merged = (size_t)(cfg->hidden / 2) * cfg->hidden + cfg->hidden / 2;
    return ws + (size_t)w.total() + build_plan(cfg).jobs.size() * (size_t)kSlotFloats + merged;
}

// the forward MLP launch; fuse_rgb / fuse_dist != null: inference with the compositing in the kernel's epilogue (ray mode only)
static int mlp_fwd_impl(const nnr_cfg* cfg, const float* pts_o, const float* pts_d, const float* view_d, const float* z_lo,
                        const float* z_hi, const float* jitter, const float* packed, float* ws, float* fuse_rgb, float* fuse_dist,
                        void* stream) {
    int rc = check_cfg(cfg);
    if (rc != NNR_OK) return rc;
    if (!pts_o || !pts_d || !view_d || !z_lo || !z_hi || !packed || !ws) return NNR_E_BADCFG;
    if (!aligned16(packed) || !aligned16(ws)) return NNR_E_ALIGN;
    const WsLayout w = ws_layout(cfg);
    MlpFwdArgs a{};
    a.pts_o = pts_o; a.pts_d = pts_d; a.view_d = view_d; a.z_lo = z_lo; a.z_hi = z_hi; a.jitter = jitter;
    a.packed = packed;
    a.ws_out4 = ws + plane(w, P_OUT4);
    a.ws_z = ws + plane(w, P_Z);
    if (w.train) {
        a.ws_xe = ws + plane(w, P_XE);
        a.ws_xh = ws + plane(w, P_XH1);
        a.ws_xf = ws + plane(w, P_XF);
        a.ws_xg = ws + plane(w, P_XG);
        if (w.bf16) {
            a.ws_xe16 = ws + plane(w, P_XE16);
            a.ws_xf16 = ws + plane(w, P_XF16);
            a.ws_pts = ws + plane(w, P_DPTS);      // the input-gradient kernel reads position / view direction here before it
            a.ws_view = ws + plane(w, P_DVIEW);    // writes their gradients to the same rows
        }
        a.ws_mask = reinterpret_cast<uint32_t*>(ws + plane(w, P_MASK));
        if (is_split2(cfg)) {      // the maxima start at zero in every training forward (the input-gradient kernel of the same step adds its planes)
            a.plane_max = plane_max_of(cfg, ws);
            hipError_t em = hipMemsetAsync(a.plane_max, 0, kPlaneMaxFloats * sizeof(float), (hipStream_t)stream);
            if (em != hipSuccess) return hip_fail(em);
        }
    }
    a.S = w.S; a.S_pad = w.S_pad; a.N = cfg->n_samples;
    a.chunks_per_ray = chunks_per_ray(cfg);
    a.fuse_rgb = fuse_rgb; a.fuse_dist = fuse_dist; a.flags = cfg->flags;
    hipError_t e = is_bf16(cfg) ? launch_mlp_fwd_bf16(cfg->hidden, a, w.train, (hipStream_t)stream)
                                : launch_mlp_fwd(cfg->hidden, a, w.train, (hipStream_t)stream, weight_mode(cfg));
    return e == hipSuccess ? NNR_OK : hip_fail(e);
}

int nnr_mlp_fwd(const nnr_cfg* cfg, const float* pts_o, const float* pts_d, const float* view_d, const float* z_lo,
                const float* z_hi, const float* jitter, const float* packed, float* ws, void* stream) {
    return mlp_fwd_impl(cfg, pts_o, pts_d, view_d, z_lo, z_hi, jitter, packed, ws, nullptr, nullptr, stream);
}

int nnr_composite_fwd(const nnr_cfg* cfg, float* rgb, float* dist, float* opt_alpha, float* opt_z, float* ws, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != NNR_OK) return rc;
    if (!rgb || !dist || !ws) return NNR_E_BADCFG;
    const WsLayout w = ws_layout(cfg);
    CompositeArgs a{};
    a.ws_out4 = ws + plane(w, P_OUT4);
    a.ws_z = ws + plane(w, P_Z);
    a.rgb = rgb; a.dist = dist; a.opt_alpha = opt_alpha; a.opt_z = opt_z;
    a.R = cfg->n_rays; a.N = cfg->n_samples; a.flags = cfg->flags;
    hipError_t e = launch_composite_fwd(a, (hipStream_t)stream);
    return e == hipSuccess ? NNR_OK : hip_fail(e);
}

int nnr_render_fwd(const nnr_cfg* cfg, const float* pts_o, const float* pts_d, const float* view_d, const float* z_lo,
                   const float* z_hi, const float* jitter, const float* packed, float* rgb, float* dist, float* opt_alpha,
                   float* opt_z, float* ws, void* stream) {
    // Inference without the per-sample outputs, whole chunks per ray: the forward kernel composites in its epilogue and writes 16
    // bytes per ray -- no per-sample (rgb, sigma, z) round trip through HBM, no second launch.
    static const bool no_fuse = std::getenv("NNR_NO_FUSED_COMPOSITE") != nullptr;     // experiments / A-B tests
    if (cfg && !(cfg->flags & NNR_F_TRAIN) && !opt_alpha && !opt_z && rgb && dist && !no_fuse && chunks_per_ray(cfg) > 0)
        return mlp_fwd_impl(cfg, pts_o, pts_d, view_d, z_lo, z_hi, jitter, packed, ws, rgb, dist, stream);
    int rc = nnr_mlp_fwd(cfg, pts_o, pts_d, view_d, z_lo, z_hi, jitter, packed, ws, stream);
    if (rc != NNR_OK) return rc;
    return nnr_composite_fwd(cfg, rgb, dist, opt_alpha, opt_z, ws, stream);
}

int nnr_composite_bwd(const nnr_cfg* cfg, const float* d_rgb, const float* d_dist, float* ws, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != NNR_OK) return rc;
    if (!(cfg->flags & NNR_F_TRAIN) || !d_rgb || !d_dist || !ws) return NNR_E_BADCFG;
    const WsLayout w = ws_layout(cfg);
    CompositeArgs a{};
    a.ws_out4 = ws + plane(w, P_OUT4);
    a.ws_z = ws + plane(w, P_Z);
    a.ws_dout4 = ws + plane(w, P_DOUT4);
    a.d_rgb = d_rgb; a.d_dist = d_dist;
    a.R = cfg->n_rays; a.N = cfg->n_samples; a.flags = cfg->flags;
    hipError_t e = launch_composite_bwd(a, (hipStream_t)stream);
    return e == hipSuccess ? NNR_OK : hip_fail(e);
}

int nnr_mlp_dgrad(const nnr_cfg* cfg, const float* packed, float* ws, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != NNR_OK) return rc;
    if (!(cfg->flags & NNR_F_TRAIN) || !packed || !ws) return NNR_E_BADCFG;
    if (!aligned16(packed) || !aligned16(ws)) return NNR_E_ALIGN;
    const WsLayout w = ws_layout(cfg);
    MlpDgradArgs a{};
    a.packed = packed;
    a.ws_dout4 = ws + plane(w, P_DOUT4);
    a.ws_xe = ws + plane(w, P_XE);
    a.ws_xf = ws + plane(w, P_XF);
    a.ws_mask = reinterpret_cast<const uint32_t*>(ws + plane(w, P_MASK));
    a.ws_dh = ws + plane(w, P_DH1);
    a.ws_dg = ws + plane(w, P_DG);
    a.ws_dpts = ws + plane(w, P_DPTS);
    a.ws_dview = ws + plane(w, P_DVIEW);
    a.plane_max = is_split2(cfg) ? plane_max_of(cfg, ws) : nullptr;
    a.S = w.S; a.S_pad = w.S_pad;
    a.chunks_per_ray = chunks_per_ray(cfg);
    hipError_t e = is_bf16(cfg) ? launch_mlp_dgrad_bf16(cfg->hidden, a, (hipStream_t)stream)
                                : launch_mlp_dgrad(cfg->hidden, a, (hipStream_t)stream, weight_mode(cfg));
    return e == hipSuccess ? NNR_OK : hip_fail(e);
}

int nnr_mlp_wgrad(const nnr_cfg* cfg, const float* packed, const nnr_param_grads* g, const void* plan, float* ws, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != NNR_OK) return rc;
    if (!(cfg->flags & NNR_F_TRAIN) || !packed || !g || !plan || !ws) return NNR_E_BADCFG;
    const WsLayout w = ws_layout(cfg);
    if (w.bf16) {
        WgradBArgs b{};
        for (int i = 0; i < 12; ++i) {
            if (!g->weight[i] || !g->bias[i]) return NNR_E_BADCFG;
            b.gw[i] = g->weight[i];
            b.gb[i] = g->bias[i];
        }
        const BPlan p = build_plan_bf16(cfg);   // host-only arithmetic, microseconds: the counts that locate the tables in `plan`
        b.n_jobs = (int)p.jobs.size();
        b.n_blocks = (int)p.block_first.size() - 1;
        b.n_outs = (int)p.outs.size();
        b.jobs = static_cast<const WgradJobB*>(plan);
        b.block_first = reinterpret_cast<const int32_t*>(b.jobs + b.n_jobs);
        b.outs = reinterpret_cast<const WgradOutB*>(b.block_first + b.n_blocks + 1);
        b.ws = ws;
        b.slots = ws + w.total();
        b.gw[kMergedLayer] = b.slots + (size_t)b.n_jobs * 4 * kSlotBFloats;
        b.gb[kMergedLayer] = b.gw[kMergedLayer] + (size_t)(cfg->hidden / 2) * cfg->hidden;
        b.packed = packed;
        b.D = cfg->hidden;
        hipError_t e = launch_wgrad_bf16(b, (hipStream_t)stream);
        return e == hipSuccess ? NNR_OK : hip_fail(e);
    }
    WgradArgs a{};
    for (int i = 0; i < 12; ++i) {
        if (!g->weight[i] || !g->bias[i]) return NNR_E_BADCFG;
        a.gw[i] = g->weight[i];
        a.gb[i] = g->bias[i];
    }
    a.jobs = static_cast<const WgradJob*>(plan);
    a.ws = ws;
    for (int p = 0; p < 48; ++p) {
        int pitch = 0;
        a.plane_off[p] = w.plane(p, &pitch);
        a.plane_pitch[p] = pitch;
        a.plane_tile[p] = w.tiled(p) ? 1 : 0;
    }
    {
        const Plan p = build_plan(cfg);   // host-only arithmetic, microseconds
        a.n_jobs = (int)p.jobs.size();
        a.n_waves = (int)p.wave_first.size() - 1;
        a.n_heads = (int)p.heads.size();
    }
    a.wave_first = reinterpret_cast<const int32_t*>(a.jobs + a.n_jobs);
    a.heads = a.wave_first + a.n_waves + 2;      // behind the wave table and the count
    a.slots = ws + w.total();
    a.gw[kMergedLayer] = a.slots + (size_t)a.n_jobs * kSlotFloats;
    a.gb[kMergedLayer] = a.gw[kMergedLayer] + (size_t)(cfg->hidden / 2) * cfg->hidden;
    a.packed = packed;
    a.D = cfg->hidden;
    a.plane_max = is_split2(cfg) ? plane_max_of(cfg, ws) : nullptr;
    a.bf16 = weight_mode(cfg);   // 0, 2 or 3 here (>= 2: the 4 x 4 tiles with six bf16 terms): locates the merge area of the packed buffer for the un-merge step
    {
        const int D = cfg->hidden;
        const int rows[13] = {D, D, D, D, D, D, D, D, 1, D, D / 2, 3, D / 2};   // outputs of the 12 nn.Linear + the merged colour-hidden matrix
        for (int l = 0; l < 13; ++l) a.bias_rows[l] = rows[l];
    }
    hipError_t e = launch_wgrad(a, (hipStream_t)stream);
    return e == hipSuccess ? NNR_OK : hip_fail(e);
}

int nnr_ray_reduce(const nnr_cfg* cfg, float* d_pts_o, float* d_pts_d, float* d_view, float* ws, void* stream) {
    int rc = check_cfg(cfg);
    if (rc != NNR_OK) return rc;
    if (!(cfg->flags & NNR_F_TRAIN) || !d_pts_o || !d_pts_d || !d_view || !ws) return NNR_E_BADCFG;
    const WsLayout w = ws_layout(cfg);
    RayReduceArgs a{};
    a.ws_dpts = ws + plane(w, P_DPTS);
    a.ws_dview = ws + plane(w, P_DVIEW);
    a.ws_z = ws + plane(w, P_Z);
    a.d_pts_o = d_pts_o; a.d_pts_d = d_pts_d; a.d_view = d_view;
    a.R = cfg->n_rays; a.N = cfg->n_samples;
    hipError_t e = launch_ray_reduce(a, (hipStream_t)stream);
    return e == hipSuccess ? NNR_OK : hip_fail(e);
}

int nnr_render_bwd(const nnr_cfg* cfg, const float* packed, const float* d_rgb, const float* d_dist,
                   const nnr_param_grads* grads, float* d_pts_o, float* d_pts_d, float* d_view, const void* plan, float* ws,
                   void* stream) {
    int rc = nnr_composite_bwd(cfg, d_rgb, d_dist, ws, stream);
    if (rc != NNR_OK) return rc;
    rc = nnr_mlp_dgrad(cfg, packed, ws, stream);
    if (rc != NNR_OK) return rc;
    rc = nnr_mlp_wgrad(cfg, packed, grads, plan, ws, stream);
    if (rc != NNR_OK) return rc;
    return nnr_ray_reduce(cfg, d_pts_o, d_pts_d, d_view, ws, stream);
}

#define NNR_LAUNCH(expr) do { hipError_t e_ = (expr); return e_ == hipSuccess ? NNR_OK : hip_fail(e_); } while (0)

int nnr_se3_exp_fwd(const float* r_all, const float* t_all, int32_t idx, float* c2w, void* stream) {
    if (!r_all || !t_all || !c2w || idx < 0) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_se3_exp_fwd(r_all, t_all, idx, c2w, (hipStream_t)stream));
}
int nnr_se3_exp_bwd(const float* r_all, int32_t idx, int32_t n_cams, const float* d_c2w, float* d_r_all, float* d_t_all,
                    void* stream) {
    if (!r_all || !d_c2w || !d_r_all || !d_t_all || idx < 0 || idx >= n_cams) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_se3_exp_bwd(r_all, idx, n_cams, d_c2w, d_r_all, d_t_all, (hipStream_t)stream));
}
int nnr_inv4_fwd(const float* a, float* y, int32_t batch, void* stream) {
    if (!a || !y || batch <= 0) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_inv4(a, y, batch, (hipStream_t)stream));
}
int nnr_inv4_bwd(const float* y, const float* d_y, float* d_a, int32_t batch, void* stream) {
    if (!y || !d_y || !d_a || batch <= 0) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_inv4_bwd(y, d_y, d_a, batch, (hipStream_t)stream));
}
int nnr_ray_setup_fwd(const float* pixels, const float* depth, const float* K, const float* W, const float* S, int32_t n_rays,
                      int32_t normalise, int32_t use_dir, float* pts_o, float* dir, float* view, float* ray_norm, float* d_gt,
                      uint8_t* mask, void* stream) {
    if (!pixels || !K || !W || !S || !pts_o || !dir || !view || !ray_norm || !d_gt || !mask || n_rays <= 0) return NNR_E_BADCFG;
    RaySetupArgs a{};
    a.pixels = pixels; a.depth = depth; a.K = K; a.W = W; a.S = S;
    a.pts_o = pts_o; a.dir = dir; a.view = view; a.ray_norm = ray_norm; a.d_gt = d_gt; a.mask = mask;
    a.R = n_rays; a.normalise = normalise; a.use_dir = use_dir;
    NNR_LAUNCH(launch_ray_setup_fwd(a, (hipStream_t)stream));
}
int nnr_ray_setup_bwd(const float* pixels, const float* depth, const float* K, const float* W, const float* S, int32_t n_rays,
                      int32_t normalise, int32_t use_dir, const float* g_pts_o, const float* g_dir, const float* g_view,
                      const float* g_ray_norm, const float* g_d_gt, float* d_depth, float* dK, float* dW, float* dS,
                      float* scratch, void* stream) {
    if (!pixels || !K || !W || !S || !dK || !dW || !dS || !scratch || n_rays <= 0) return NNR_E_BADCFG;
    RaySetupArgs a{};
    a.pixels = pixels; a.depth = depth; a.K = K; a.W = W; a.S = S;
    a.g_o = g_pts_o; a.g_dir = g_dir; a.g_view = g_view; a.g_norm = g_ray_norm; a.g_dgt = g_d_gt;
    a.g_depth = d_depth; a.acc = scratch; a.gK = dK; a.gW = dW; a.gS = dS;
    a.R = n_rays; a.normalise = normalise; a.use_dir = use_dir;
    NNR_LAUNCH(launch_ray_setup_bwd(a, (hipStream_t)stream));
}
int nnr_adam_step(const nnr_adam_table* t, void* stream) {
    if (!t || t->n_tensors < 0 || t->n_tensors > NNR_ADAM_MAX_TENSORS) return NNR_E_BADCFG;
    if (t->block_first[0] != 0 || (t->flavour != NNR_ADAM_FUSED && t->flavour != NNR_ADAM_SINGLE)) return NNR_E_BADCFG;
    for (int i = 0; i < t->n_tensors; ++i) {
        if (!t->param[i] || !t->grad[i] || !t->exp_avg[i] || !t->exp_avg_sq[i] || !t->step_in[i] || !t->step_out[i] || t->numel[i] <= 0 ||
            t->step_in[i] == t->step_out[i])
            return NNR_E_BADCFG;
        if (t->flavour == NNR_ADAM_SINGLE && !(t->bc2_sqrt[i] > 0.0)) return NNR_E_BADCFG;
        if (t->block_first[i + 1] - t->block_first[i] != (int32_t)((t->numel[i] + 1023) / 1024)) return NNR_E_BADCFG;
    }
    NNR_LAUNCH(launch_adam_multi(*t, (hipStream_t)stream));
}
static bool step_fill(const nnr_step_cfg* c, nnr::StepRaysArgs& a) {
    if (!c || c->n_rays <= 0 || c->h < 2 || c->w < 2 || c->hd <= 0 || c->wd <= 0 || c->n_cams <= 0 || c->cam < 0 || c->cam >= c->n_cams)
        return false;
    a.R = c->n_rays; a.h = c->h; a.w = c->w; a.hd = c->hd; a.wd = c->wd; a.cam = c->cam; a.n_cams = c->n_cams;
    a.normalise = (c->flags & NNR_STEP_NORMALISE) != 0; a.use_dir = (c->flags & NNR_STEP_USE_DIR) != 0;
    a.shift_first = (c->flags & NNR_STEP_SHIFT_FIRST) != 0; a.fix_last_scale = (c->flags & NNR_STEP_FIX_LAST_SCALE) != 0;
    if (c->ref >= c->n_cams || c->ref == c->cam) return false;
    a.ref = c->ref < 0 ? -1 : c->ref;
    a.detach_ref = (c->flags & NNR_STEP_DETACH_REF) != 0;
    a.g_mats = nullptr;
    return true;
}
int nnr_step_rays_fwd(const nnr_step_cfg* cfg, const float* r_all, const float* t_all, const float* scales, const float* shifts,
                      const float* K, const float* S, const int64_t* ray_idx, const float* depth_img, const float* img, float* pts_o,
                      float* dir, float* view, float* ray_norm, float* d_gt, uint8_t* mask, float* rgb_gt, float* pixels, float* mats,
                      void* stream) {
    nnr::StepRaysArgs a{};
    if (!step_fill(cfg, a)) return NNR_E_BADCFG;
    if (!r_all || !t_all || !scales || !shifts || !K || !S || !ray_idx || !depth_img || !pts_o || !dir || !view || !ray_norm || !d_gt ||
        !mask || !pixels || !mats || (img && !rgb_gt))
        return NNR_E_BADCFG;
    a.r_all = r_all; a.t_all = t_all; a.scales = scales; a.shifts = shifts; a.K = K; a.S = S; a.ray_idx = ray_idx;
    a.depth_img = depth_img; a.img = img; a.pts_o = pts_o; a.dir = dir; a.view = view; a.ray_norm = ray_norm; a.d_gt = d_gt;
    a.mask = mask; a.rgb_gt = rgb_gt; a.pixels = pixels; a.mats = mats;
    NNR_LAUNCH(launch_step_rays_fwd(a, (hipStream_t)stream));
}
int nnr_step_rays_bwd(const nnr_step_cfg* cfg, const float* r_all, const float* t_all, const float* scales, const float* shifts,
                      const float* K, const float* S, const int64_t* ray_idx, const float* depth_img, const float* g_pts_o,
                      const float* g_dir, const float* g_view, const float* g_ray_norm, const float* g_d_gt, const float* g_mats,
                      float* d_r, float* d_t, float* d_scales, float* d_shifts, float* scratch, void* stream) {
    nnr::StepRaysArgs a{};
    if (!step_fill(cfg, a)) return NNR_E_BADCFG;
    a.g_mats = g_mats;
    if (!r_all || !t_all || !scales || !shifts || !K || !S || !ray_idx || !depth_img || !d_r || !d_t || !d_scales || !d_shifts || !scratch)
        return NNR_E_BADCFG;
    a.bwd_scratch = scratch;
    a.r_all = r_all; a.t_all = t_all; a.scales = scales; a.shifts = shifts; a.K = K; a.S = S; a.ray_idx = ray_idx;
    a.depth_img = depth_img; a.g_o = g_pts_o; a.g_dir = g_dir; a.g_view = g_view; a.g_norm = g_ray_norm; a.g_dgt = g_d_gt;
    a.d_r = d_r; a.d_t = d_t; a.d_scales = d_scales; a.d_shifts = d_shifts;
    NNR_LAUNCH(launch_step_rays_bwd(a, (hipStream_t)stream));
}
int nnr_depth_gather_affine_fwd(const float* depth_img, const int64_t* ray_idx, const float* scale, const float* shift, int32_t shift_first,
                                float* out, int32_t n_rays, int32_t h, int32_t w, int32_t hd, int32_t wd, void* stream) {
    if (!depth_img || !ray_idx || !scale || !shift || !out || n_rays <= 0 || h <= 0 || w <= 0 || hd <= 0 || wd <= 0) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_depth_gather_affine_fwd(depth_img, ray_idx, scale, shift, shift_first, out, n_rays, h, w, hd, wd, (hipStream_t)stream));
}
int nnr_depth_gather_affine_bwd(const float* g_out, const float* depth_img, const int64_t* ray_idx, const float* scale, const float* shift,
                                int32_t shift_first, float* g_scale_shift, int32_t n_rays, int32_t h, int32_t w, int32_t hd, int32_t wd,
                                void* stream) {
    if (!g_out || !depth_img || !ray_idx || !scale || !shift || !g_scale_shift || n_rays <= 0 || h <= 0 || w <= 0 || hd <= 0 || wd <= 0)
        return NNR_E_BADCFG;
    NNR_LAUNCH(launch_depth_gather_affine_bwd(g_out, depth_img, ray_idx, scale, shift, shift_first, g_scale_shift, n_rays, h, w, hd, wd,
                                              (hipStream_t)stream));
}

int nnr_ndc_rays_fwd(const float* rays_o, const float* rays_d, const float* camera_mat, float near_plane, float* o_ndc, float* d_ndc,
                     int32_t n_rays, void* stream) {
    if (!rays_o || !rays_d || !camera_mat || !o_ndc || !d_ndc || n_rays <= 0) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_ndc_rays_fwd(rays_o, rays_d, camera_mat, near_plane, o_ndc, d_ndc, n_rays, (hipStream_t)stream));
}
int nnr_ndc_rays_bwd(const float* rays_o, const float* rays_d, const float* camera_mat, float near_plane, const float* g_o_ndc,
                     const float* g_d_ndc, float* g_rays_o, float* g_rays_d, int32_t n_rays, void* stream) {
    if (!rays_o || !rays_d || !camera_mat || !g_o_ndc || !g_d_ndc || !g_rays_o || !g_rays_d || n_rays <= 0) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_ndc_rays_bwd(rays_o, rays_d, camera_mat, near_plane, g_o_ndc, g_d_ndc, g_rays_o, g_rays_d, n_rays, (hipStream_t)stream));
}

int nnr_depth_gather_fwd(const float* depth_img, const int64_t* ray_idx, float* out, int32_t n_rays, int32_t h, int32_t w,
                         int32_t hd, int32_t wd, void* stream) {
    if (!depth_img || !ray_idx || !out || n_rays <= 0 || h <= 0 || w <= 0 || hd <= 0 || wd <= 0) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_depth_gather_fwd(depth_img, ray_idx, out, n_rays, h, w, hd, wd, (hipStream_t)stream));
}
int nnr_depth_gather_bwd(const float* g_out, const int64_t* ray_idx, float* g_img, int32_t n_rays, int32_t h, int32_t w,
                         int32_t hd, int32_t wd, void* stream) {
    if (!g_out || !ray_idx || !g_img || n_rays <= 0 || h <= 0 || w <= 0 || hd <= 0 || wd <= 0) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_depth_gather_bwd(g_out, ray_idx, g_img, n_rays, h, w, hd, wd, (hipStream_t)stream));
}
int nnr_render_loss(const float* rgb, const float* rgb_gt, const float* dist, const float* d_gt, const uint8_t* mask,
                    int32_t n_rays, float r_total, float m_total, float w_rgb, float w_depth, int32_t rgb_l2, int32_t ndc,
                    int32_t detach_gt, const float* m_total_dev, float* out5, float* g_rgb, float* g_dist, float* g_d_gt,
                    void* stream) {
    if (!rgb || !rgb_gt || !dist || !d_gt || !mask || !out5 || !g_rgb || !g_dist || !g_d_gt || n_rays <= 0 || r_total <= 0.f)
        return NNR_E_BADCFG;
    LossArgs a{};
    a.rgb = rgb; a.rgb_gt = rgb_gt; a.dist = dist; a.d_gt = d_gt; a.mask = mask; a.out = out5;
    a.g_rgb = g_rgb; a.g_dist = g_dist; a.g_dgt = g_d_gt; a.R = n_rays; a.r_total = r_total; a.m_total = m_total;
    a.w_rgb = w_rgb; a.w_depth = w_depth; a.rgb_l2 = rgb_l2; a.ndc = ndc; a.detach_gt = detach_gt;
    a.m_total_dev = m_total_dev;
    NNR_LAUNCH(launch_render_loss(a, (hipStream_t)stream));
}

int nnr_pixels_from_index(const int64_t* ray_idx, float* pixels, int32_t n_rays, int32_t h, int32_t w, void* stream) {
    if (!ray_idx || !pixels || n_rays <= 0 || h < 2 || w < 2) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_pixels_from_index(ray_idx, pixels, n_rays, h, w, (hipStream_t)stream));
}

int nnr_pc_nearest(const float* src, const float* dst, int32_t n_src, int32_t n_dst, int64_t* idx, float* dist, void* scratch,
                   void* stream) {
    if (!src || !dst || !idx || !dist || !scratch || n_src <= 0 || n_dst <= 0) return NNR_E_BADCFG;
    if (((uintptr_t)scratch & 7) != 0) return NNR_E_ALIGN;
    NNR_LAUNCH(launch_pc_nearest(src, dst, n_src, n_dst, idx, dist, static_cast<unsigned long long*>(scratch), (hipStream_t)stream));
}

int nnr_pc_error_bwd(const float* src, const float* dst, const int64_t* idx, const float* dist, const float* g_loss, int32_t n_src,
                     int32_t n_dst, float* g_src, float* g_dst, void* stream) {
    if (!src || !dst || !idx || !dist || !g_loss || n_src <= 0 || n_dst <= 0 || (!g_src && !g_dst)) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_pc_error_bwd(src, dst, idx, dist, g_loss, n_src, n_dst, g_src, g_dst, (hipStream_t)stream));
}

size_t nnr_randperm_scratch_bytes(int32_t r) {
    const unsigned int cap = r > 0 ? nnr::randperm_capacity(r) : 0;
    return cap ? 8 + 20 * (size_t)cap : 0;   // header, u32 ranks, u64 candidates, u64 candidates in order
}

int nnr_randperm_prefix(const int64_t* keys, int64_t n, int32_t bits, int32_t r, uint64_t seed, uint64_t offset, int64_t* out,
                        void* scratch, void* stream) {
    int idx_bits = 1;
    while ((1ll << idx_bits) < n) ++idx_bits;
    if (!keys || !out || !scratch || n <= 0 || r <= 0 || r > n || bits < 1 || bits > 64) return NNR_E_BADCFG;
    if (bits + idx_bits > 64 || nnr::randperm_capacity(r) == 0 || n < 8 * (int64_t)r) return NNR_E_UNSUPPORTED;
    if (((uintptr_t)scratch & 7) != 0) return NNR_E_ALIGN;
    NNR_LAUNCH(launch_randperm_prefix(keys, n, bits, r, seed, offset, out, static_cast<unsigned int*>(scratch), (hipStream_t)stream));
}

int nnr_uniform_rows(uint64_t seed, uint64_t offset, uint64_t threads, uint64_t first, uint64_t n, float* out, void* stream) {
    if (!out || threads == 0 || (threads & 255) != 0) return NNR_E_BADCFG;
    NNR_LAUNCH(launch_uniform_rows(seed, offset, threads, first, n, out, (hipStream_t)stream));
}

namespace {
// workspace of the per-image losses, in floats; 8-byte items first so that they stay aligned
size_t aux_fill(const nnr_aux_cfg* c, float* ws, nnr::AuxArgs& a) {   // returns the workspace size in floats, 0 = bad cfg
    if (!c || c->hd <= 0 || c->wd <= 0 || c->hr < 2 || c->wr < 2 || c->hr > c->hd || c->wr > c->wd) return 0;
    const int64_t S = (int64_t)c->hr * c->wr;
    if (c->shard_lo < 0 || c->shard_hi < c->shard_lo || c->shard_hi > S) return 0;
    a.hd = c->hd; a.wd = c->wd; a.hr = c->hr; a.wr = c->wr; a.S = (int)S;
    a.s_lo = c->shard_lo;
    a.s_hi = (c->shard_lo == 0 && c->shard_hi == 0) ? (int)S : c->shard_hi;   // 0, 0 = every point (one GPU)
    a.nl = c->nearest_limit;
    a.flags = c->flags;
    a.w_pc = c->w_pc; a.w_rgbs = c->w_rgbs;
    if ((c->flags & NNR_AUX_MATS_GRAD) && (!(c->flags & NNR_AUX_AFFINE) || (c->flags & NNR_AUX_GRAD_K))) return 0;
    float* p = ws;
    auto take = [&](int64_t n) { float* r = p; p += n; return r; };
    a.keys = reinterpret_cast<unsigned long long*>(take(4 * S));
    a.idx_xy = reinterpret_cast<int64_t*>(take(2 * S));
    a.idx_yx = reinterpret_cast<int64_t*>(take(2 * S));
    a.gXq = reinterpret_cast<long long*>(take(6 * S));
    a.gYq = reinterpret_cast<long long*>(take(6 * S));
    a.X = take(3 * S); a.Y = take(3 * S);
    a.gxy = take(2 * S);
    if (c->flags & NNR_AUX_SSIM) {
        a.rgb1 = take(3 * S); a.rgb2 = take(3 * S);
        a.drgb = take(6 * S);
    }
    a.dist_xy = take(S); a.dist_yx = take(S);
    a.pflags = reinterpret_cast<uint32_t*>(take(S));
    a.acc = take(8);
    const int64_t nb = (S + 255) / 256;
    a.part_fwd = take(4 * nb);
    a.part_bwd = take(44 * nb);
    return (size_t)(p - ws);
}
}  // namespace

size_t nnr_aux_workspace_floats(const nnr_aux_cfg* cfg) {
    nnr::AuxArgs a{};
    static float origin;   // only differences of pointers derived from it are used
    return aux_fill(cfg, &origin, a);
}

int nnr_aux_terms_fwd(const nnr_aux_cfg* cfg, const float* d1_img, const float* d2_img, const float* img1r, const float* img2r,
                      const float* K, const float* Kinv, const float* rel, const float* scale2, const float* aff, float* out, float* ws,
                      void* stream) {
    nnr::AuxArgs a{};
    if (!ws || !aux_fill(cfg, ws, a)) return NNR_E_BADCFG;
    if (!d1_img || !d2_img || !K || !Kinv || !rel || !out) return NNR_E_BADCFG;
    if (((cfg->flags & NNR_AUX_AFFINE) != 0) != (aff != nullptr)) return NNR_E_BADCFG;
    a.aff = aff; a.shift_first = (cfg->flags & NNR_AUX_SHIFT_FIRST) != 0;
    if ((cfg->flags & NNR_AUX_RGBS) && (!img1r || !img2r)) return NNR_E_BADCFG;
    if ((cfg->flags & NNR_AUX_SCALE_PCS) && !scale2) return NNR_E_BADCFG;
    if (((uintptr_t)ws & 7) != 0) return NNR_E_ALIGN;
    a.d1_img = d1_img; a.d2_img = d2_img; a.img1r = img1r; a.img2r = img2r; a.K = K; a.Kinv = Kinv; a.rel = rel; a.scale2 = scale2;
    a.out = out;
    NNR_LAUNCH(launch_aux_fwd(a, (hipStream_t)stream));
}

int nnr_aux_terms_bwd(const nnr_aux_cfg* cfg, const float* d1_img, const float* d2_img, const float* img1r, const float* img2r,
                      const float* K, const float* Kinv, const float* rel, const float* scale2, const float* aff, const float* g_out,
                      float* g_d1_img, float* g_d2_img, float* g_rel_scale, float* ws, void* stream) {
    nnr::AuxArgs a{};
    if (!ws || !aux_fill(cfg, ws, a)) return NNR_E_BADCFG;
    if (!d1_img || !d2_img || !K || !Kinv || !rel || !g_out || !g_rel_scale) return NNR_E_BADCFG;
    if (((cfg->flags & NNR_AUX_AFFINE) != 0) != (aff != nullptr)) return NNR_E_BADCFG;
    a.aff = aff; a.shift_first = (cfg->flags & NNR_AUX_SHIFT_FIRST) != 0;
    if ((cfg->flags & NNR_AUX_SCALE_PCS) && !scale2) return NNR_E_BADCFG;
    if (((uintptr_t)ws & 7) != 0) return NNR_E_ALIGN;
    a.d1_img = d1_img; a.d2_img = d2_img; a.img1r = img1r; a.img2r = img2r; a.K = K; a.Kinv = Kinv; a.rel = rel; a.scale2 = scale2;
    a.g_out = g_out; a.g_d1_img = g_d1_img; a.g_d2_img = g_d2_img;
    NNR_LAUNCH(launch_aux_bwd(a, g_rel_scale, (hipStream_t)stream));
}

}  // extern "C"
