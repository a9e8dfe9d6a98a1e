// C-ABI of the MI355X-native batched IPP environment-step engine (see include/ipp_engine.h).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC ipp_engine.hip -o libipp_hip.so
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "ipp_common.h"
#include "k_gain.h"
#include "k_gain_factor.h"
#include "k_step_factor.h"
#include "k_step_patch.h"
#include "k_step_split.h"
#include "k_gain_wave.h"
#include "k_misc.h"
#include "k_grf_dft.h"
#include "k_grf_hartley.h"
#include "k_grf_fft.h"
#include "k_score.h"
#include "k_plane.h"
#include "k_tree.h"
#include "k_tree_patch.h"
#include "k_mcts.h"
#include "k_prepare.h"

using namespace ipp;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
}  // namespace
namespace ipp {
int set_error(int code, const char* msg) {  // other translation units of the library (ipp_arena.hip) report through the same slot
    g_err = msg;
    return code;
}
}  // namespace ipp
namespace {
#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(-2, "%s failed: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

constexpr uint64_t kAlign = 256;
constexpr int kMaxChunks = 8;
inline uint64_t up(uint64_t x) { return (x + kAlign - 1) / kAlign * kAlign; }

#ifndef IPP_PATCH_BIGKP
#define IPP_PATCH_BIGKP 4  // rows per request group of the six-waves-per-SIMD instantiation of k_step_patch
#endif

struct ProfEvent { hipEvent_t a, b; int kind; };
constexpr int kProfKinds = 4;  // 0 streaming kernel, 1 dense downdate, 2 prologue kernel; 3: every launch of a step (busy time only)
struct ProfSlot {
    double total_ms = 0.0;
    int64_t launches = 0;
    double busy_ms = 0.0;       // union of the dispatches' [start, stop] intervals (launches of different streams overlap)
    int64_t busy_launches = 0;
};

struct Engine {
    ipp_config cfg;
    int device;
    View v;
    int n_bands;
    uint64_t used_bytes;
    size_t prep_lds;
    size_t gain_lds;
    int q_chunk;
    TreeView tv = {};      // node pool of ipp_tree_step (node_cap == 0: none)
    float* node_diag_scratch = nullptr;  // [Npad] assembled diagonal of a node state (ipp_tree_score_actions)
    bool scoring = false;  // arena holds the ipp_score_actions scratch
    ScoreView sv = {};
    bool grf_dft = false;  // even square grids up to 256: k_grf_dft instead of k_grf_conv + k_grf_norm
    int grf_tt = 0;        // > 0: even square grids up to 128: k_grf_hartley<grf_tt> (fp64 MFMA GEMMs)
    bool grf_fft = false;  // ... and n = 50 / 100: k_grf_fft (fast Hartley transforms) on the same amplitude table
    int grf_kc = 1;        // spectrum rows per LDS chunk
    int lut_cap;
    int lut_rows = 0;  // workgroup-per-item factor kernels: rows |drow| of the prior table kept in LDS
    bool profile = false;
    int step_chunks = 0;  // 0 = auto
    bool fused = false;   // k_step_factor instead of k_prepare + k_gain_factor
    bool tree_ok = false; // ipp_tree_step available (fused engines; MC = 25: the fused tree kernel beside two-launch env steps)
    size_t tree_fused_lds = 0;
    bool patch = false;   // k_step_patch on compact column patches (View::patch)
    int patch_waves = kPatchWavesDefault;  // waves per item of k_step_patch
    int split_min_items = 0;  // launches of at least this many items run the SPLIT step (k_step_split.h: prologue kernel + unit kernel); 0: never
    int split_waves = 3;      // waves per item of its prologue kernel
    int pcap_p = 0;           // records the prologue kernel of the split step stages in LDS (a multiple of 8)
    size_t lds_p = 0, lds_u = 0;
    int pcap_big = 0, big_min_items = 0;  // large launches: k_step_patch<2, 4, 6> with LDS for 12 workgroups per CU (0: never)
    size_t lds_big = 0;
    // three-wave engines (the default): launches of at least two_wave_min_items items run TWO waves per item (k_step_patch<2>, 12 items
    // per CU = 3072 slots; from big_min_items its four-rows-per-group form) -- 0: never
    int pcap2 = 0, two_wave_min_items = 0;
    size_t lds2 = 0;
    const double* reset_prior = nullptr;  // ipp_set_reset_prior: priors of the episodes started by ipp_step_autoreset
    bool rect_ok = false;      // rectangle tiles (k_gain_factor.h) possible: clipped windows, 128-cell tiles, even grid width
    bool rect_commit = false;  // ... used for committed steps too (else for predict-only calls only)
    bool rect_tree = false;    // tree steps on rectangle tiles (same width rule; either tile size)
    int tree_split_min = 0;   // ipp_tree_step: launches of at least this many items run k_tree_prepare + k_tree_gain (0: never)
    int tree_T = kStepThreads;  // workgroup size of k_tree_gain
    size_t tree_step_lds = 0;   // k_tree_step: the fused kernel's LDS + the column-pointer table
    size_t tree_gain_lds = 0;
    hipStream_t side = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipEvent_t ev_prep[8] = {};
    ProfSlot prof[kProfKinds];
    std::vector<ProfEvent> prof_pending;
    int last_n = 0;
    uint64_t last_needed_bytes = 0;  // of the last ipp_streamed_bytes(_detail) read
};

struct Layout {
    int N, Npad, T, n_tiles, win_tiles, MC, FC, QS, q_rows, VEC;
    bool patch;
    int patch_waves;
    PatchGeo pg;
    uint64_t off_mean, off_diag, off_gt, off_gtslot, off_prior, off_rank, off_span, off_cnt, off_icnt, off_cov, off_blk, off_hdr, off_linv, off_yv, off_q, off_wc,
        off_partial, off_dbg, off_grfh, off_grfcs, off_grfg, off_grfhp, off_grfamp, off_grfraw, off_grfraw2, off_sc_hdr, off_sc_ext, off_sc_mask, off_sc_G, off_sc_P, off_tr_cov, off_tr_diag, off_tr_meta, off_sc_ndiag, total, cov_slot_floats;
};

uint64_t q_item_floats(const Layout& L) {
    const uint64_t lq = ((uint64_t)L.MC * L.MC + L.MC + 3) & ~(uint64_t)3;
    // multiple of 16 floats: item blocks start on 64-byte lines (no scalar-cache line shared between two items)
    uint64_t q = lq + (uint64_t)(L.q_rows + 8) * L.QS;
    if (L.patch) q = std::max<uint64_t>(q, (uint64_t)L.q_rows * kPatchRec);  // column records that overflow the LDS staging
    return (q + 15) & ~(uint64_t)15;
}

// Compact column patches + k_step_patch / k_tree_patch (k_step_patch.h, k_tree_patch.h): windowed factor engines on grids wide enough for two-dimensional windows, MC = 9.  Any of the A/B
// switches of the band-tile kernels selects those kernels instead; IPP_PATCH=0 does so explicitly.
bool patch_layout(const ipp_config& c, int MC) {
    if (c.state_repr != IPP_FACTOR || c.window_rows <= 0 || MC != 9) return false;
    if (c.tile_threads != 0) return false;
    if (c.node_capacity > 0) {
        // tree nodes on patches (k_tree_patch.h): the records address column patches by 32-bit offsets in 8-byte units from
        // View::cov: root slots + node blocks must lie within 32 GB of it
        // (plan() checks the reach of the 32-bit record offsets on the finished layout)
    }
    if (c.x_dim % 2 != 0 || c.x_dim > 256 || c.y_dim > 256) return false;
    if (!(c.x_dim > 2 * c.window_rows + 13)) return false;
    if (c.rank_cap > kPatchMaxRank) return false;
    if (c.window_rows > 25) return false;  // (prior table of (R + 7)^2 floats and patches of (2 R + 6) x (2 R + 7) cells: small windows only)
    {   // the units map a cell index to (row, column) of a rectangle through a reciprocal: exact for every width and index this geometry has
        const PatchGeo g = patch_geometry(c.x_dim, c.y_dim, c.window_rows);
        if (!patch_units_exact(g.pw, g.ph)) return false;
    }
    for (const char* name : {"IPP_RECT_META", "IPP_RECT", "IPP_FUSED", "IPP_STEP_CHUNKS"})
        if (getenv(name)) return false;
    if (const char* p = getenv("IPP_PATCH")) return atoi(p) != 0;
    return true;
}
int patch_waves_wanted() {
    if (const char* w = getenv("IPP_PATCH_WAVES")) { const int n = atoi(w); if (n >= 1 && n <= 4) return n; }
    return kPatchWavesDefault;
}

// Windowed factor columns: the largest length scale a reset may install and the prior covariance dropped at the
// window edge for it (Matern 3/2 at R rows).
constexpr double kWindowBound = 1e-6;
double max_length_scale(const ipp_config& c) { return (c.fixed_prior ? 1.0 : 1.2) * c.length_scale; }
double window_bound(const ipp_config& c, int rows) {
    const double a = std::sqrt(3.0) * (double)rows * c.resolution / max_length_scale(c);
    return c.signal_variance * (1.0 + a) * std::exp(-a);
}
int min_window_rows(const ipp_config& c) {
    int r = 1;
    while (r < (1 << 20) && window_bound(c, r) > kWindowBound) ++r;
    return r;
}

int plan(const ipp_config& c, Layout& L, bool allow_patch = true) {
    if (c.x_dim <= 0 || c.y_dim <= 0) return fail(-1, "x_dim/y_dim must be positive");
    if (!(c.resolution > 0)) return fail(-1, "resolution must be positive");
    if (c.state_repr != IPP_DENSE && c.state_repr != IPP_FACTOR) return fail(-1, "state_repr must be IPP_DENSE or IPP_FACTOR");
    if (c.capacity <= 0 || c.max_batch <= 0) return fail(-1, "capacity and max_batch must be positive");
    if (c.state_repr == IPP_FACTOR && c.rank_cap <= 0) return fail(-1, "rank_cap must be positive for IPP_FACTOR");
    if (!(c.signal_variance > 0) || !(c.length_scale > 0)) return fail(-1, "signal_variance and length_scale must be positive");
    if (!(c.max_v > 0) || !(c.max_a > 0)) return fail(-1, "max_v and max_a must be positive");
    L.MC = (c.max_measurements <= 0 || c.max_measurements <= 9) ? 9 : 25;
    if (c.max_measurements > IPP_MAX_MEAS) return fail(-1, "max_measurements above %d is not compiled in", IPP_MAX_MEAS);
    L.FC = 4 * L.MC;
    L.QS = (L.MC + 3) & ~3;
    // cells per thread of the streaming kernels (register budget).  Windowed factor columns are stored and streamed on
    // whole tiles of 64 VEC cells: 128-cell tiles round a step's window up by half as much as 256-cell tiles do, and the
    // step moves 15 % fewer bytes (4096 envs of 50x50: 12.5 -> 13.8 M env-steps/s; every config gained 5-11 %)
    // Only while a window is a few tiles: at 200x200 (20 tiles of 256 cells per step) the rounding is 2.5 % of the bytes and
    // twice as many tiles cost more than that in per-tile work (configs[4] tree wave: 13.4 M steps/s with 256-cell tiles,
    // 12.1 M with 128).
    const bool windowed = c.state_repr == IPP_FACTOR && c.window_rows > 0;
    const long window_cells = std::min<long>(c.y_dim, 2L * c.window_rows + 5) * c.x_dim;
    L.VEC = (L.MC == 9 && !(windowed && window_cells < 16 * 256)) ? 4 : 2;
    L.patch = allow_patch && patch_layout(c, L.MC);
    L.patch_waves = patch_waves_wanted();
    if (L.patch) {
        L.VEC = 2;
        L.pg = patch_geometry(c.x_dim, c.y_dim, c.window_rows);
    }
    L.N = c.x_dim * c.y_dim;
    const int n4 = (L.N + L.VEC - 1) / L.VEC;
    if (c.tile_threads > 0) {
        if (c.tile_threads % 64 != 0 || c.tile_threads > kMaxTileThreads) return fail(-1, "tile_threads must be a multiple of 64 and <= %d", kMaxTileThreads);
        L.T = c.tile_threads;
    } else {
        // least padding first; among equals 128 threads (2 waves) measured best on MI355X (many small
        // workgroups per CU hide each other's prologue / epilogue), else the largest <= 320
        int best_t = 64;
        long best_pad = -1;
        for (int t = 64; t <= 320; t += 64) {
            const long tiles = (n4 + t - 1) / t, padded = tiles * t;
            const bool better = best_pad < 0 || padded < best_pad ||
                                (padded == best_pad && (t == 128 || (best_t != 128 && t > best_t)));
            if (better) { best_pad = padded; best_t = t; }
        }
        L.T = best_t;
    }
    L.n_tiles = (n4 + L.T - 1) / L.T;
    L.Npad = L.n_tiles * L.T * L.VEC;
    if (c.state_repr == IPP_FACTOR && c.window_rows > 0) {
        // the columns are cut where the prior covariance to the footprint has decayed: refuse windows that are too
        // narrow for this prior (length scale up to 1.2 x nominal under shuffle_prior_cov, mappings.py:238-240)
        const double bound = window_bound(c, c.window_rows);
        if (c.window_rows < std::max(c.x_dim, c.y_dim) && bound > kWindowBound)
            return fail(-1, "window_rows = %d drops prior covariances up to %.1e (> 1e-6) for length scales up to %.3g m at %.3g m cells: "
                            "use window_rows >= %d, or 0 for exact columns", c.window_rows, bound, max_length_scale(c), c.resolution,
                        min_window_rows(c));
        // windowed factor state: one workgroup per item, wave-granular tiles of 64 * VEC cells (k_gain_factor.h);
        // tile_threads is the workgroup size (waves share the item's Q block and prior table in LDS)
        L.T = (c.tile_threads > 0) ? c.tile_threads : 256;  // 256: fused workgroup kernel (k_step_factor.h), 64: one wave per item (k_gain_wave.h)
        // Large batches of short items on large grids (BASELINE configs[2]: 32768 envs of 100x100, ranks <= 144, ~10 tiles
        // of ~20 rows per item): the per-item latency chain of the fused kernel and its workgroup-granular dispatch leave
        // the stream at 32 % of peak; the prologue as its own kernel + a 128-thread gain kernel streams at 50 % and is 15 %
        // faster per step with the two pipelined over 4 chunks.  Tree steps need the fused layout.
        // With two-dimensional windows the stream of an item is short enough for that to hold at 50x50 too (32768 envs:
        // fused 22.4 M env-steps/s, split 24.7 M, split in 2 chunks 25.6 M; at 4096 envs the fused kernel stays ahead).
        // (Since the rectangle metadata the fused kernel is ahead up to 16384 envs of 50x50 -- 8192: 24.7 vs 23.4 M, 16384:
        // 26.0 vs 23.7 M env-steps/s -- and the split path from 32768: 28.3 vs 26.9 M; on 100x100 the split path stays 23 % ahead.)
        if (c.tile_threads <= 0 && c.node_capacity <= 0 && c.capacity >= ((int64_t)c.x_dim * c.y_dim >= 6000 ? 8192 : 24576)) L.T = 128;
        if (L.patch) L.T = 64 * L.patch_waves;  // one fused kernel for every batch size
        if (L.T > 512) return fail(-1, "tile_threads must be <= 512 for IPP_FACTOR");
        L.n_tiles = (n4 + 63) / 64;
        L.Npad = L.n_tiles * 64 * L.VEC;
    }
    L.win_tiles = L.n_tiles;
    if (c.state_repr == IPP_FACTOR && c.window_rows > 0) {
        // rows a step can touch: the footprint (ny <= m <= MC blocks of rf <= 2 rows) + window_rows on both sides
        const long rows = std::min<long>(c.y_dim, 2L * c.window_rows + 2L * L.MC);
        L.win_tiles = (int)std::min<long>(L.n_tiles, (rows * c.x_dim + 64 * L.VEC - 1) / (64 * L.VEC) + 1);
    }
    L.q_rows = (c.state_repr == IPP_FACTOR) ? c.rank_cap : L.FC;
    L.cov_slot_floats = (c.state_repr == IPP_FACTOR) ? (uint64_t)c.rank_cap * L.Npad : (uint64_t)L.N * L.Npad;
    if (L.patch) L.cov_slot_floats = (uint64_t)c.rank_cap * L.pg.pstride;
    uint64_t o = 0;
    const uint64_t cap = c.capacity, mb = c.max_batch, np = L.Npad;
    L.off_mean = o; o += up(cap * np * 4);
    L.off_diag = o; o += up(cap * np * 4);
    L.off_gt = o; o += up(2 * cap * np * 4);  // two ground-truth planes per env: the current one and the one staged for its next episode
    L.off_gtslot = o; o += up(2 * cap * 4);  // [cap] current plane of every env + [cap] "alternate plane staged" flags
    L.off_prior = o; o += up(cap * 2 * 8);
    L.off_rank = o; o += up(cap * 4);
    L.off_span = o; o += (c.state_repr == IPP_FACTOR) ? up(2 * cap * (uint64_t)c.rank_cap * 4) : 0;  // tile spans, then rectangles
    L.off_cnt = o; o += up((uint64_t)kCountSlots * 128);
    L.off_icnt = o; o += up((uint64_t)c.max_batch * 16);
    L.off_cov = o; o += up(cap * L.cov_slot_floats * 4);
    L.off_blk = o; o += L.patch ? up(mb * SplitBlk::floats(L.pg.plw, c.rank_cap) * 4) : 0;  // item blocks of the split step (k_step_split.h)
    L.off_hdr = o; o += up(mb * sizeof(ItemHdr));
    L.off_linv = o; o += up(mb * L.MC * L.MC * 4);
    L.off_yv = o; o += up(mb * L.MC * 4);
    L.off_q = o; o += up(mb * q_item_floats(L) * 4 + 4096);  // [L^-1|y] head + Q rows + pad rows, + DMA over-read slack
    L.off_wc = o; o += (c.state_repr == IPP_DENSE) ? up(mb * L.MC * np * 4) : 0;
    L.off_partial = o; o += up(mb * L.n_tiles * 8);
    L.off_dbg = o; o += up(mb * (2 * L.MC * L.MC + 2 * L.MC) * 8);
    L.off_grfh = o; o += up((uint64_t)L.N * 8);
    L.off_grfcs = o; o += up((uint64_t)c.x_dim * 16);                       // (cos, sin)(2 pi j / n)
    L.off_grfg = o; o += up((uint64_t)(c.y_dim / 2 + 1) * c.x_dim * 8);    // g_k[d], k = 0 .. n/2 (k_grf_dft.h)
    {
        const uint64_t npad = 16 * (uint64_t)((std::max(c.x_dim, c.y_dim) + 15) / 16);
        const uint64_t tab = (c.x_dim == c.y_dim && c.x_dim % 2 == 0 && c.x_dim <= 128) ? npad * npad * 8 : 8;
        L.off_grfhp = o; o += up(tab);
        L.off_grfamp = o; o += up(tab);
    }
    {   // un-normalised fields of the convolution path (k_grf_conv + k_grf_norm): even square grids up to 256 run the Hartley /
        // DFT kernels, which normalise in place (2 x 1.3 GB of the configs[2] arena)
        const bool conv = !(c.x_dim == c.y_dim && c.x_dim % 2 == 0 && c.x_dim >= 4 && c.x_dim <= 256);
        L.off_grfraw = o; o += conv ? up(mb * np * 4) : 0;
        L.off_grfraw2 = o; o += conv ? up(mb * np * 4) : 0;
    }
    L.off_sc_hdr = L.off_sc_ext = L.off_sc_mask = L.off_sc_G = L.off_sc_P = o;
    if (c.score_scratch) {  // ipp_score_actions (k_score.h)
        L.off_sc_hdr = o; o += up(mb * sizeof(ScoreHdr));
        L.off_sc_ext = o; o += up(64);
        L.off_sc_mask = o; o += up(np * 4);
        L.off_sc_G = o; o += up((uint64_t)kScoreSplit * L.N * kScoreBandCap * 8);
        L.off_sc_P = o; o += (c.state_repr == IPP_FACTOR) ? up((uint64_t)L.N * np * 4) : 0;
    }
    L.off_tr_cov = L.off_tr_diag = L.off_tr_meta = L.off_sc_ndiag = o;
    if (c.node_capacity > 0) {  // ipp_tree_step (k_tree.h)
        if (c.state_repr != IPP_FACTOR) return fail(-1, "node_capacity needs IPP_FACTOR");
        const uint64_t nc = c.node_capacity, wc = L.patch ? (uint64_t)L.pg.pstride : (uint64_t)L.win_tiles * 64 * L.VEC;  // a node lives on its step's tile span / patch
        L.off_tr_cov = o; o += up(nc * L.MC * wc * 4 + 4096);
        L.off_tr_diag = o; o += up(nc * wc * 4 + 4096);
        L.off_tr_meta = o; o += up(nc * kNodeMeta * 4);
        L.off_sc_ndiag = o; o += c.score_scratch ? up(np * 4) : 0;
        // tree nodes on patches (k_tree_patch.h): a record addresses its column patch by a 32-bit offset in 8-byte units from
        // View::cov - kTreePatchGuard, so root slots AND node blocks have to lie within 2^35 bytes of it -- decided on the finished
        // layout (the score scratch sits between the two regions: 6.5 GB at 200x200, 17 GB at 256x256); beyond that the band-tile
        // tree kernels take over
        if (L.patch && allow_patch) {
            const uint64_t reach = (L.off_tr_diag - L.off_cov) + (uint64_t)kTreePatchGuard + (1ull << 20);  // (+ the largest shift of a patch)
            if (reach >= (1ull << 35)) return plan(c, L, false);
        }
    }
    L.total = o;
    return 0;
}

size_t prep_lds_bytes(const Layout& L, const ipp_config& c) {
    const size_t MC = L.MC, FC = L.FC;
    const size_t small = (L.MC == 9) ? prep_small_bytes<9>() : prep_small_bytes<25>();
    size_t b = (small + 15) & ~(size_t)15;
    if (c.state_repr == IPP_FACTOR)
        b += MC * (size_t)((c.rank_cap + 3) & ~3) * sizeof(float);
    else
        b += FC * (FC + 1) * sizeof(float);
    return (b + 15) & ~(size_t)15;
}

// h = Re ifft2(amp) with the reference's amplitude table (simulations/ground_truths.py:7-29), on the host.
void grf_kernel_host(int H, int W, double c, std::vector<double>& h) {
    auto idx_list = [](int n) {
        std::vector<int> a;
        const int half = n / 2;
        for (int i = 0; i <= half; ++i) a.push_back(i);
        for (int i = half - 1; i >= 1; --i) a.push_back(-i);
        return a;  // one entry short for odd n, like the reference
    };
    const std::vector<int> ky = idx_list(H), kx = idx_list(W);
    std::vector<double> amp((size_t)H * W, 0.0);
    for (size_t i = 0; i < ky.size(); ++i)
        for (size_t j = 0; j < kx.size(); ++j) {
            if (ky[i] == 0 && kx[j] == 0) continue;
            const double k = std::sqrt((double)ky[i] * ky[i] + (double)kx[j] * kx[j]);
            amp[i * W + j] = std::sqrt(std::pow(k, -c));
        }
    // separable inverse DFT, real part only (amp is real; imaginary parts cancel for even amp)
    std::vector<double> are((size_t)H * W), aim((size_t)H * W);
    for (int u = 0; u < H; ++u)
        for (int x = 0; x < W; ++x) {
            double re = 0, im = 0;
            for (int q = 0; q < W; ++q) {
                const double ang = 2.0 * M_PI * (double)((long)q * x % W) / W;
                re += amp[(size_t)u * W + q] * std::cos(ang);
                im += amp[(size_t)u * W + q] * std::sin(ang);
            }
            are[(size_t)u * W + x] = re;
            aim[(size_t)u * W + x] = im;
        }
    h.assign((size_t)H * W, 0.0);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            double re = 0;
            for (int u = 0; u < H; ++u) {
                const double ang = 2.0 * M_PI * (double)((long)u * y % H) / H;
                re += are[(size_t)u * W + x] * std::cos(ang) - aim[(size_t)u * W + x] * std::sin(ang);
            }
            h[(size_t)y * W + x] = re / ((double)H * W);
        }
}

// Kernel launch; with profiling on (bench.py's roofline leg) the start / stop events are attached to the dispatch
// itself (hipExtLaunchKernelGGL), so their difference is the kernel's own duration as rocprofv3 reports it, not the
// interval between two event packets around it (which adds the dispatch latency, 20-25 us on this pool).
template <typename... P, typename... A>
void timed_launch(Engine* e, int kind, void (*kernel)(P...), dim3 grid, dim3 block, size_t lds, hipStream_t s, A... args) {
    if (!e->profile) {
        hipLaunchKernelGGL(kernel, grid, block, lds, s, args...);
        return;
    }
    hipEvent_t a = nullptr, b = nullptr;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, s, a, b, 0, static_cast<P>(args)...);
    e->prof_pending.push_back({a, b, kind});
}

void prof_drain(Engine* e) {
    // per kind: sum of the dispatches' durations + the union of their [start, stop] intervals (launches of different streams
    // overlap); slot 3: the union over EVERY launch (a split step is a prologue launch and a unit launch: both are the step)
    auto& pend = e->prof_pending;
    if (pend.empty()) return;
    std::vector<std::pair<float, float>> iv[kProfKinds];  // ms behind the first start of this batch
    for (auto& pr : pend) (void)hipEventSynchronize(pr.b);
    for (auto& pr : pend) {
        float ms = 0.f, t0 = 0.f;
        if (hipEventElapsedTime(&ms, pr.a, pr.b) == hipSuccess) {
            e->prof[pr.kind].total_ms += ms;
            e->prof[pr.kind].launches += 1;
            if (hipEventElapsedTime(&t0, pend.front().a, pr.a) == hipSuccess) { iv[pr.kind].emplace_back(t0, t0 + ms); iv[3].emplace_back(t0, t0 + ms); }
        }
    }
    if (const char* dump = getenv("IPP_PROFILE_DUMP")) {  // tools/region_timeline.py: every dispatch's [start, stop] of this batch
        if (FILE* f = fopen(dump, "a")) {
            fprintf(f, "# batch of %zu dispatches: kind start_us stop_us\n", pend.size());
            for (int k = 0; k < 3; ++k)
                for (auto& x : iv[k]) fprintf(f, "%d %.2f %.2f\n", k, 1e3 * x.first, 1e3 * x.second);
            fclose(f);
        }
    }
    for (auto& pr : pend) {
        (void)hipEventDestroy(pr.a);
        (void)hipEventDestroy(pr.b);
    }
    pend.clear();
    for (int k = 0; k < kProfKinds; ++k) {
        std::sort(iv[k].begin(), iv[k].end());
        float end = -1e30f;
        ProfSlot& p = e->prof[k];
        for (auto& x : iv[k]) {
            if (x.first > end) { p.busy_ms += x.second - x.first; end = x.second; }
            else if (x.second > end) { p.busy_ms += x.second - end; end = x.second; }
        }
        p.busy_launches += (int64_t)iv[k].size();
    }
}

size_t gain_lds_bytes(const View& v, int q_chunk, int lut_cap) {
    const size_t MC = v.meas_cap, QS = v.q_stride;
    size_t b = std::max((size_t)(q_chunk + 2 * kPipe) * QS, (size_t)((lut_cap + 3) & ~3)) * 4 + ((MC * MC + 3) & ~(size_t)3) * 4 + ((MC + 3) & ~(size_t)3) * 4 + 16 * 8;
    b += (size_t)std::max(q_chunk, v.q_rows) * 4;  // rowidx: streaming index -> row of the covariance slab
    return (b + 15) & ~(size_t)15;
}

// One chunk of items: prologue -> streaming gain (-> reward finalize) (-> dense downdate) on stream `s`.
// `v` carries scratch pointers already offset to the chunk's first item.
template <int MC, int VEC>
void launch_chunk(Engine* e, const View& v, const int32_t* env_ids, const int32_t* dst_ids, int n, const double* action,
                  const double* prev, const float* noise, unsigned flags, float* reward, int32_t* status, hipStream_t s,
                  hipEvent_t prep_done, const AutoReset& ar) {
    if (e->patch) {  // compact column patches: one fused kernel, one small workgroup per item (k_step_patch.h)
        if constexpr (MC == 9 && VEC == 2) {
            if (e->split_min_items > 0 && n >= e->split_min_items) {
                // split step (k_step_split.h): item-parallel prologue kernel, then one wave per (item, unit)
                View vp = v;
                vp.pcap = e->pcap_p;
                if (e->split_waves == 1)
                    timed_launch(e, 2, k_step_patch<1, kPatchKP, kSplitMinWP, true>, dim3(n), dim3(64), e->lds_p, s, vp, env_ids, n, action, prev, noise, flags, status, reward, ar);
                else if (e->split_waves == 2)
                    timed_launch(e, 2, k_step_patch<2, kPatchKP, kSplitMinWP, true>, dim3(n), dim3(128), e->lds_p, s, vp, env_ids, n, action, prev, noise, flags, status, reward, ar);
                else
                    timed_launch(e, 2, k_step_patch<3, kPatchKP, kSplitMinWP, true>, dim3(n), dim3(192), e->lds_p, s, vp, env_ids, n, action, prev, noise, flags, status, reward, ar);
                timed_launch(e, 0, k_step_units<>, dim3(split_grid(n, v.punits)), dim3(64), e->lds_u, s, v, n, flags, reward, ar);
            } else if (e->patch_waves == 1)
                timed_launch(e, 0, k_step_patch<1>, dim3(n), dim3(64), e->gain_lds, s, v, env_ids, n, action, prev, noise, flags, status, reward, ar);
            else if (e->patch_waves == 4)
                timed_launch(e, 0, k_step_patch<4>, dim3(n), dim3(256), e->gain_lds, s, v, env_ids, n, action, prev, noise, flags, status, reward, ar);
            else if (e->patch_waves == 3 && !(e->two_wave_min_items > 0 && n >= e->two_wave_min_items))
                if (v.rank_cap <= 192)       // (rounds of 192 threads over the columns' rectangles: k_step_patch.h, RJN)
                    timed_launch(e, 0, k_step_patch<3, kPatchKP, kPatchMinW, false, 1>, dim3(n), dim3(192), e->gain_lds, s, v, env_ids, n, action, prev, noise, flags, status, reward, ar);
                else if (v.rank_cap <= 384)
                    timed_launch(e, 0, k_step_patch<3, kPatchKP, kPatchMinW, false, 2>, dim3(n), dim3(192), e->gain_lds, s, v, env_ids, n, action, prev, noise, flags, status, reward, ar);
                else
                    timed_launch(e, 0, k_step_patch<3>, dim3(n), dim3(192), e->gain_lds, s, v, env_ids, n, action, prev, noise, flags, status, reward, ar);
            else if (e->big_min_items > 0 && n >= e->big_min_items) {
                View vb = v;  // (six waves per SIMD pay once a launch is many rounds of workgroups: k_step_patch.h)
                vb.pcap = e->pcap_big;
                timed_launch(e, 0, k_step_patch<2, IPP_PATCH_BIGKP, 6>, dim3(n), dim3(128), e->lds_big, s, vb, env_ids, n, action, prev, noise, flags, status, reward, ar);
            } else if (e->patch_waves == 3) {
                View v2 = v;  // a large launch of a three-wave engine: two waves per item, LDS share of 12 workgroups per CU
                v2.pcap = e->pcap2;
                timed_launch(e, 0, k_step_patch<2>, dim3(n), dim3(128), e->lds2, s, v2, env_ids, n, action, prev, noise, flags, status, reward, ar);
            } else
                timed_launch(e, 0, k_step_patch<2>, dim3(n), dim3(128), e->gain_lds, s, v, env_ids, n, action, prev, noise, flags, status, reward, ar);
        }
        if (prep_done) (void)hipEventRecord(prep_done, s);
        return;
    }
    if constexpr (MC == 9) {  // (MC = 25: the fused kernel would spill 431 VGPRs -- those engines run the prologue and the gain kernel as two launches)
    if (e->fused) {  // windowed factor state: prologue + gain in one kernel, one workgroup per item
        if (VEC == 2 && e->rect_ok && (e->rect_commit || (flags & IPP_PREDICT_ONLY)))
            timed_launch(e, 0, k_step_factor<MC, VEC, (MC == 9 && VEC == 2)>, dim3(n), dim3(kStepThreads), e->gain_lds, s, v, env_ids, n, action, prev, noise,
                     flags, e->lut_rows, status, reward, ar);
        else
            timed_launch(e, 0, k_step_factor<MC, VEC>, dim3(n), dim3(kStepThreads), e->gain_lds, s, v, env_ids, n, action, prev, noise,
                     flags, e->lut_rows, status, reward, ar);
        if (prep_done) (void)hipEventRecord(prep_done, s);
        return;
    }
    }
    if (v.mode == IPP_FACTOR)
        timed_launch(e, 2, k_prepare<MC, IPP_FACTOR>, dim3(n), dim3(kPrepThreads), e->prep_lds, s, v, env_ids, dst_ids, n, action, prev,
                     noise, flags, status, (float*)nullptr, (int*)nullptr, (int*)nullptr);
    else
        timed_launch(e, 2, k_prepare<MC, IPP_DENSE>, dim3(n), dim3(kPrepThreads), e->prep_lds, s, v, env_ids, dst_ids, n, action, prev,
                     noise, flags, status, (float*)nullptr, (int*)nullptr, (int*)nullptr);
    if (prep_done) (void)hipEventRecord(prep_done, s);
    {
        const int grid = grid_for(n, v.n_tiles);
        if (v.mode == IPP_FACTOR && v.window_rows > 0 && v.T == kWave)
            timed_launch(e, 0, k_gain_wave<MC, VEC>, dim3(n), dim3(kWave), e->gain_lds, s, v, v.q, n, flags, reward);
        else if (v.mode == IPP_FACTOR && v.window_rows > 0)
            if (MC == 9 && VEC == 2 && e->rect_ok && (e->rect_commit || (flags & IPP_PREDICT_ONLY)))
                timed_launch(e, 0, k_gain_factor<MC, VEC, (MC == 9 && VEC == 2)>, dim3(n), dim3(v.T), e->gain_lds, s, v, v.q, n, flags, e->lut_rows, reward);
            else
                timed_launch(e, 0, k_gain_factor<MC, VEC>, dim3(n), dim3(v.T), e->gain_lds, s, v, v.q, n, flags, e->lut_rows, reward);
        else if (v.mode == IPP_FACTOR)
            timed_launch(e, 0, k_gain<MC, VEC, IPP_FACTOR>, dim3(grid), dim3(v.T), e->gain_lds, s, v, n, flags, e->q_chunk, e->lut_cap, reward);
        else
            timed_launch(e, 0, k_gain<MC, VEC, IPP_DENSE>, dim3(grid), dim3(v.T), e->gain_lds, s, v, n, flags, e->q_chunk, e->lut_cap, reward);
    }
    if (v.n_tiles > 1 && !(v.mode == IPP_FACTOR && v.window_rows > 0)) hipLaunchKernelGGL(k_reward_finalize, dim3((n + 255) / 256), dim3(256), 0, s, v, n, reward);
    if (v.mode == IPP_DENSE && !(flags & IPP_PREDICT_ONLY)) {
        const int grid = grid_for(n, e->n_bands * v.n_tiles);
        timed_launch(e, 1, k_downdate<MC, VEC>, dim3(grid), dim3(v.T), 0, s, v, n, e->n_bands);
    }
}

// The prologue is latency-bound (dependent loads, a 9x9 fp64 factorisation per item) and the gain kernel is
// HBM-bound, so a large batch is cut into chunks that alternate between the caller's stream and an engine-owned
// side stream: chunk c's prologue waits for chunk c-1's prologue and therefore runs under chunk c-1's gain
// kernel.  Chunks touch disjoint items and (for committed steps) disjoint env slots.
template <int MC, int VEC>
int launch_step(Engine* e, const int32_t* env_ids, const int32_t* dst_ids, int n, const double* action,
                const double* prev, const float* noise, unsigned flags, float* reward, int32_t* status, hipStream_t s,
                const AutoReset& ar) {
    int chunks = e->step_chunks;
    // measured on MI355X: at 4096 items of 50x50 no gain from 2 chunks, slower from 4; at 32768 items of 100x100 on the
    // split path 4 chunks hide most of the prologue kernel under the previous chunk's gain kernel
    if (chunks <= 0) chunks = (!e->fused && e->v.mode == IPP_FACTOR && e->v.window_rows > 0 && e->v.T == 128 && n >= 16384) ? (e->v.N >= 6000 ? 4 : 2) : 1;
    if (e->profile) chunks = 1;  // kernels are timed alone (bench.py roofline leg)
    chunks = std::min(chunks, kMaxChunks);
    if (chunks <= 1 || !e->side) {
        launch_chunk<MC, VEC>(e, e->v, env_ids, dst_ids, n, action, prev, noise, flags, reward, status, s, nullptr, ar);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    HIP_TRY(hipEventRecord(e->ev_begin, s));
    HIP_TRY(hipStreamWaitEvent(e->side, e->ev_begin, 0));
    const int per = ((n + chunks - 1) / chunks + 7) / 8 * 8;
    int used = 0;
    const bool ordered = e->v.item_order && e->v.item_order_n == n && e->v.mode == IPP_FACTOR && e->v.window_rows > 0 && e->v.T != kWave;  // (k_gain_wave keeps its own item map)
    for (int c = 0, off = 0; off < n; ++c, off += per) {
        const int nc = std::min(per, n - off);
        hipStream_t st = (c % 2 == 0) ? s : e->side;
        View v = e->v;
        if (c > 0) HIP_TRY(hipStreamWaitEvent(st, e->ev_prep[c - 1], 0));
        if (ordered) {
            // a dispatch order is set for this launch size: chunk c takes positions [off, off + nc) of it -- any items, so the
            // per-item arrays keep their global indexing (heaviest chunk first, the last chunk ends the step on the lightest)
            v.item_order = e->v.item_order + off;
            v.item_order_n = nc;
            launch_chunk<MC, VEC>(e, v, env_ids, dst_ids, nc, action, prev, noise, flags, reward, status, st, e->ev_prep[c], ar);
            used = c + 1;
            continue;
        }
        v.item_order = nullptr;
        v.env_base = off;
        v.hdr += off;
        v.linv += (size_t)off * v.meas_cap * v.meas_cap;
        v.yv += (size_t)off * v.meas_cap;
        v.q += (size_t)off * v.q_item;
        if (v.wc) v.wc += (size_t)off * v.meas_cap * v.Npad;
        v.partial += (size_t)off * v.n_tiles;
        v.dbg += (size_t)off * (2 * v.meas_cap * v.meas_cap + 2 * v.meas_cap);
        AutoReset arc = ar;
        if (arc.src) arc.src += off;
        launch_chunk<MC, VEC>(e, v, env_ids ? env_ids + off : nullptr, dst_ids ? dst_ids + off : nullptr, nc,
                              action + 3 * (size_t)off, prev + 3 * (size_t)off, noise ? noise + (size_t)off * MC : nullptr,
                              flags, reward + off, status ? status + off : nullptr, st, e->ev_prep[c], arc);
        used = c + 1;
    }
    (void)used;
    HIP_TRY(hipEventRecord(e->ev_end, e->side));
    HIP_TRY(hipStreamWaitEvent(s, e->ev_end, 0));
    HIP_TRY(hipGetLastError());
    return 0;
}

// white [n][N] -> normalised field, either into the env slots (gt_out == nullptr) or into gt_out [n][N]
// Tables of k_grf_dft.h: cs[j] = (cos, sin)(2 pi j / n); g[k][d] = c_k / n * sum_l amp[k][l] cos(2 pi l d / n).
void grf_dft_tables_host(int n, double c, std::vector<double>& cs, std::vector<double>& g) {
    std::vector<int> kidx;
    for (int i = 0; i <= n / 2; ++i) kidx.push_back(i);
    for (int i = n / 2 - 1; i >= 1; --i) kidx.push_back(-i);  // ground_truths.py:7-11 (n entries for even n)
    cs.resize((size_t)2 * n);
    for (int j = 0; j < n; ++j) {
        cs[2 * j] = std::cos(2.0 * M_PI * j / n);
        cs[2 * j + 1] = std::sin(2.0 * M_PI * j / n);
    }
    const int n_k = n / 2 + 1;
    g.assign((size_t)n_k * n, 0.0);
    std::vector<double> amp(n);
    for (int k = 0; k < n_k; ++k) {
        for (int l = 0; l < n; ++l) {
            const double kk = std::sqrt((double)kidx[k] * kidx[k] + (double)kidx[l] * kidx[l]);
            amp[l] = (kidx[k] == 0 && kidx[l] == 0) ? 0.0 : std::sqrt(std::pow(kk, -c));
        }
        const double ck = (k == 0 || k == n / 2) ? 1.0 : 2.0;
        for (int d = 0; d < n; ++d) {
            double acc = 0.0;
            for (int l = 0; l < n; ++l) acc += amp[l] * cs[2 * (int)((long)l * d % n)];
            g[(size_t)k * n + d] = ck * acc / n;
        }
    }
}

// Tables of k_grf_hartley.h, zero padded to np x np: H[j][k] = cos + sin of 2 pi j k / n; amp as ground_truths.py:20-27.
// Returns false when the amplitude is not even in each index (then the Hartley form does not apply).
bool grf_hartley_tables_host(int n, int np, double c, std::vector<double>& hp, std::vector<double>& amp) {
    std::vector<int> kidx;
    for (int i = 0; i <= n / 2; ++i) kidx.push_back(i);
    for (int i = n / 2 - 1; i >= 1; --i) kidx.push_back(-i);  // ground_truths.py:7-11 (n entries for even n)
    hp.assign((size_t)np * np, 0.0);
    amp.assign((size_t)np * np, 0.0);
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k) {
            const double a = 2.0 * M_PI * (double)(((long)j * k) % n) / n;
            hp[(size_t)j * np + k] = std::cos(a) + std::sin(a);
            const double kk = std::sqrt((double)kidx[j] * kidx[j] + (double)kidx[k] * kidx[k]);
            amp[(size_t)j * np + k] = (kidx[j] == 0 && kidx[k] == 0) ? 0.0 : std::sqrt(std::pow(kk, -c));
        }
    for (int j = 0; j < n; ++j)
        for (int k = 0; k < n; ++k) {
            const int jm = (n - j) % n, km = (n - k) % n;
            if (amp[(size_t)j * np + k] != amp[(size_t)jm * np + k] || amp[(size_t)j * np + k] != amp[(size_t)j * np + km]) return false;
        }
    return true;
}

// 8 waves (one owned 16-row tile each) from TT = 5: half the registers per wave of the 4-wave form, so that waves of the
// streaming kernels still fit beside a field's workgroup on every SIMD
template <int TT>
void launch_grf_hartley(const View& v, int n, const float* white, const int32_t* env_ids, float* gt_out, hipStream_t s) {
    constexpr int NW = (TT > 4 && TT < 8) ? 8 : 4;  // (TT = 8 with one tile per wave spills 112 registers at the 168 it may use)
    // the grid sizes of BASELINE.json contract over ceil(n / 4) steps instead of the padded 4 TT (k_grf_hartley.h, KS)
    if (TT == 4 && v.W == 50)
        hipLaunchKernelGGL((k_grf_hartley<4, 4, 13>), dim3(n), dim3(256), grf_hartley_lds_bytes(4), s, v, env_ids, n, white,
                           (const double*)v.grf_hp, (const double*)v.grf_amp, gt_out);
    else if (TT == 7 && v.W == 100)
        hipLaunchKernelGGL((k_grf_hartley<7, 8, 25>), dim3(n), dim3(512), grf_hartley_lds_bytes(7), s, v, env_ids, n, white,
                           (const double*)v.grf_hp, (const double*)v.grf_amp, gt_out);
    else
        hipLaunchKernelGGL((k_grf_hartley<TT, NW>), dim3(n), dim3(64 * NW), grf_hartley_lds_bytes(TT), s, v, env_ids, n, white,
                           (const double*)v.grf_hp, (const double*)v.grf_amp, gt_out);
}

int launch_grf(Engine* e, int n, const float* white, float* raw, const int32_t* env_ids, float* gt_out, hipStream_t s,
               const GrfNoise* noise = nullptr) {
    const View& v = e->v;
    if (!white && !(noise && e->grf_tt > 0 && e->grf_fft)) return fail(-1, "in-kernel ground-truth noise needs the fast Hartley path (50x50 / 100x100 grids)");
    const GrfNoise gn = noise ? *noise : GrfNoise{nullptr, 0, 0, 0, 0, {0}, 0};
    if (v.W != v.H) return fail(-1, "device GRF needs a square grid (the reference transposes its dims, simulations/simulations.py:45-47)");
    if (e->grf_tt > 0 && e->grf_fft) {  // n = 50 / 100: fast Hartley transforms in LDS (k_grf_fft.h), same amplitude table
        const int np = 16 * e->grf_tt;
        const double* ampt = (const double*)v.grf_amp;
        const double2* twt = (const double2*)v.grf_cs;
        if (v.W == 100) hipLaunchKernelGGL((k_grf_fft<10, float>), dim3(n), dim3(512), grf_fft_lds_bytes(100, 4), s, v, env_ids, n, white, ampt, np, gt_out, gn, twt);
        else hipLaunchKernelGGL((k_grf_fft<5, float>), dim3(n), dim3(256), grf_fft_lds_bytes(50, 4), s, v, env_ids, n, white, ampt, np, gt_out, gn, twt);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (e->grf_tt > 0) {  // even n <= 128: four fp64 GEMMs on the matrix cores, normalisation fused (k_grf_hartley.h)
        switch (e->grf_tt) {
            case 1: launch_grf_hartley<1>(v, n, white, env_ids, gt_out, s); break;
            case 2: launch_grf_hartley<2>(v, n, white, env_ids, gt_out, s); break;
            case 3: launch_grf_hartley<3>(v, n, white, env_ids, gt_out, s); break;
            case 4: launch_grf_hartley<4>(v, n, white, env_ids, gt_out, s); break;
            case 5: launch_grf_hartley<5>(v, n, white, env_ids, gt_out, s); break;
            case 6: launch_grf_hartley<6>(v, n, white, env_ids, gt_out, s); break;
            case 7: launch_grf_hartley<7>(v, n, white, env_ids, gt_out, s); break;
            default: launch_grf_hartley<8>(v, n, white, env_ids, gt_out, s); break;
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (e->grf_dft) {  // even n <= 256: half-spectrum DFT, normalisation fused (k_grf_dft.h)
        const bool small = v.W <= 100;
        const int nt = small ? 256 : 1024;
        const int xb = (v.W % 4 == 0) ? 4 : 2;                 // columns per thread in the row-inverse stage
        const int rgroups = std::max(1, nt / (v.W / xb));      // row groups
        const int rows = (v.W + rgroups - 1) / rgroups;        // rows per thread
        const size_t lds = grf_dft_lds_bytes(v.W, e->grf_kc, small);
#define IPP_GRF_LAUNCH(OPT, XB, NT, WLDS) \
    hipLaunchKernelGGL((k_grf_dft<OPT, XB, NT, WLDS>), dim3(n), dim3(NT), lds, s, v, env_ids, n, white, v.grf_cs, v.grf_g, e->grf_kc, gt_out)
        if (small && xb == 4) IPP_GRF_LAUNCH(10, 4, 256, true);        // rows <= 10 for every n <= 100
        else if (small && rows <= 10) IPP_GRF_LAUNCH(10, 2, 256, true);
        else if (small) IPP_GRF_LAUNCH(20, 2, 256, true);              // n = 54 .. 98, n = 2 mod 4
        else if (xb == 4 && rows <= 10) IPP_GRF_LAUNCH(10, 4, 1024, false);   // n <= 200
        else if (xb == 4) IPP_GRF_LAUNCH(16, 4, 1024, false);                 // n <= 256
        else if (rows <= 20) IPP_GRF_LAUNCH(20, 2, 1024, false);              // n <= 202
        else IPP_GRF_LAUNCH(32, 2, 1024, false);                              // n <= 256
#undef IPP_GRF_LAUNCH
        HIP_TRY(hipGetLastError());
        return 0;
    }
    const size_t lds = (size_t)v.N * (sizeof(double) + sizeof(float));
    const bool use_lds = lds <= 120 * 1024;
    const int ot = (v.W <= 64) ? 5 : 8;
    const int tpr = (v.W + ot - 1) / ot;
    const int groups = v.H * tpr;
    if (use_lds) {  // one workgroup per env, tables in LDS
        if (ot == 5) hipLaunchKernelGGL((k_grf_conv<5, true>), dim3(1, n), dim3(256), lds, s, v, n, white, tpr, raw);
        else         hipLaunchKernelGGL((k_grf_conv<8, true>), dim3(1, n), dim3(256), lds, s, v, n, white, tpr, raw);
    } else {
        if (ot == 5) hipLaunchKernelGGL((k_grf_conv<5, false>), dim3((groups + 255) / 256, n), dim3(256), 0, s, v, n, white, tpr, raw);
        else         hipLaunchKernelGGL((k_grf_conv<8, false>), dim3((groups + 255) / 256, n), dim3(256), 0, s, v, n, white, tpr, raw);
    }
    hipLaunchKernelGGL(k_grf_norm, dim3(n), dim3(256), 0, s, v, env_ids, n, raw, gt_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

Engine* as_engine(void* p) { return reinterpret_cast<Engine*>(p); }

int check_env(Engine* e, int env) {
    if (env < 0 || env >= e->v.cap) return fail(-1, "env_id %d outside [0, %d)", env, e->v.cap);
    return 0;
}

}  // namespace

extern "C" {

int ipp_abi_version(void) { return IPP_ABI_VERSION; }
const char* ipp_last_error(void) { return g_err.c_str(); }

int ipp_min_window_rows(const ipp_config* cfg, int32_t* rows) {
    if (!cfg || !rows) return fail(-1, "null argument");
    if (!(cfg->resolution > 0) || !(cfg->length_scale > 0) || !(cfg->signal_variance > 0)) return fail(-1, "resolution, length_scale and signal_variance must be positive");
    *rows = min_window_rows(*cfg);
    return 0;
}

int ipp_engine_arena_bytes(const ipp_config* cfg, uint64_t* bytes) {
    if (!cfg || !bytes) return fail(-1, "null argument");
    Layout L;
    if (int rc = plan(*cfg, L)) return rc;
    *bytes = L.total;
    return 0;
}

int ipp_engine_create(const ipp_config* cfg, int device, void* arena, uint64_t arena_bytes, void** engine) {
    if (!cfg || !arena || !engine) return fail(-1, "null argument");
    Layout L;
    if (int rc = plan(*cfg, L)) return rc;
    if (arena_bytes < L.total) return fail(-1, "arena too small: %llu < %llu", (unsigned long long)arena_bytes, (unsigned long long)L.total);
    if ((reinterpret_cast<uintptr_t>(arena) & (kAlign - 1)) != 0) return fail(-1, "arena must be %llu-byte aligned", (unsigned long long)kAlign);
    HIP_TRY(hipSetDevice(device));
    Engine* e = new (std::nothrow) Engine();
    if (!e) return fail(-3, "out of host memory");
    e->cfg = *cfg;
    e->device = device;
    e->used_bytes = L.total;
    View& v = e->v;
    v.W = cfg->x_dim; v.H = cfg->y_dim; v.N = L.N; v.Npad = L.Npad; v.T = L.T; v.n_tiles = L.n_tiles; v.vec = L.VEC; v.env_base = 0;
    v.mode = cfg->state_repr; v.cap = cfg->capacity; v.rank_cap = cfg->rank_cap; v.max_batch = cfg->max_batch;
    v.meas_cap = L.MC; v.fp_cap = L.FC; v.q_stride = L.QS; v.q_rows = L.q_rows; v.q_item = q_item_floats(L);
    v.res = cfg->resolution; v.tanx = cfg->tan_half_fov_x; v.tany = cfg->tan_half_fov_y; v.rf_alt = cfg->rf_altitude;
    v.coeff_a = cfg->coeff_a; v.coeff_b = cfg->coeff_b; v.sv0 = cfg->signal_variance; v.ls0 = cfg->length_scale;
    v.ls_max = (cfg->state_repr == IPP_FACTOR && cfg->window_rows > 0 && cfg->window_rows < std::max(cfg->x_dim, cfg->y_dim)) ? max_length_scale(*cfg) : 0.0;
    v.vmax = cfg->max_v; v.amax = cfg->max_a; v.thr = cfg->value_threshold; v.kf = cfg->interval_factor;
    char* base = reinterpret_cast<char*>(arena);
    v.mean = reinterpret_cast<float*>(base + L.off_mean);
    v.diag = reinterpret_cast<float*>(base + L.off_diag);
    v.gt = reinterpret_cast<float*>(base + L.off_gt);
    v.gt_slot = reinterpret_cast<int*>(base + L.off_gtslot);
    v.prior = reinterpret_cast<double*>(base + L.off_prior);
    v.rank = reinterpret_cast<int*>(base + L.off_rank);
    v.colspan = reinterpret_cast<int*>(base + L.off_span);
    v.colrect = v.colspan + (size_t)cfg->capacity * cfg->rank_cap;
    v.dbg_capture = 0;
    v.counters = reinterpret_cast<unsigned long long*>(base + L.off_cnt);
    v.item_counts = reinterpret_cast<unsigned long long*>(base + L.off_icnt);
    v.window_rows = (cfg->state_repr == IPP_FACTOR) ? std::max(0, cfg->window_rows) : 0;
    v.tile_cells = (cfg->state_repr == IPP_FACTOR && cfg->window_rows > 0) ? 64 * L.VEC : L.T * L.VEC;
    v.tile_shift = -1;
    for (int sh = 0; sh < 20; ++sh) if ((1 << sh) == v.tile_cells) v.tile_shift = sh;
    // two-dimensional windows (k_gain_factor.h): every kernel that appends columns of a windowed state clips them (the
    // one-wave-per-item kernel of tile_threads = 64 and the pipelined kernel do not: off for those engines)
    v.clip_cols = (cfg->state_repr == IPP_FACTOR && cfg->window_rows > 0 && L.T != 64 && cfg->x_dim > 2 * cfg->window_rows + 13) ? 1 : 0;
    // rectangle tiles for committed steps where a rectangle (<= 2 R + 5 + 2 cells of alignment wide) is at most 0.4 grid rows
    // (the rectangle-tile kernels carry no sqrt / exp form of the prior term: the table P0(|drow|, |dcol|) must be complete,
    // i.e. not cut at 48 KiB -- same arithmetic as for lut_rows below)
    const int lut_tile_rows = (v.tile_cells + cfg->x_dim - 1) / cfg->x_dim + 1;
    const bool lut_complete = (size_t)std::min(cfg->y_dim, std::max(0, cfg->window_rows) + lut_tile_rows + 6) * cfg->x_dim <= 12288;
    e->rect_ok = v.clip_cols && L.MC == 9 && L.VEC == 2 && cfg->x_dim % 2 == 0 && lut_complete;
    e->rect_commit = e->rect_ok && 5 * (2 * cfg->window_rows + 7) <= 2 * cfg->x_dim;
    e->rect_tree = v.clip_cols && L.MC == 9 && cfg->x_dim % L.VEC == 0 && lut_complete && 5 * (2 * cfg->window_rows + 5 + 2 * (L.VEC - 1)) <= 2 * cfg->x_dim;
    if (const char* rt = getenv("IPP_RECT_TREE")) e->rect_tree = v.clip_cols && L.MC == 9 && cfg->x_dim % L.VEC == 0 && lut_complete && atoi(rt) != 0;  // A/B experiments
    if (const char* rc = getenv("IPP_RECT")) { e->rect_commit = e->rect_ok && atoi(rc) == 2; e->rect_ok = e->rect_ok && atoi(rc) != 0; }  // A/B: 0 off, 1 predict-only, 2 always
    // Rectangle metadata (View::rect_meta, ipp_common.h): steps on rectangle tiles store the new columns on the rectangle
    // only and record it per column; every reader of stored columns masks with it.  Without it the band cells outside the
    // rectangle received zeros from a store-only pass: 75 % of the gain kernel's HBM writes at 100x100, more at 200x200.
    v.rect_meta = (v.clip_cols && L.MC == 9 && cfg->x_dim % L.VEC == 0 && cfg->x_dim <= 256 && cfg->y_dim <= 256 && (e->rect_ok || e->rect_tree)) ? 1 : 0;
    if (const char* rm = getenv("IPP_RECT_META")) v.rect_meta = v.rect_meta && atoi(rm) != 0;  // A/B experiments
    // Only the rectangle-tile variants of the streaming kernels mask (k_gain_factor.h), so with the metadata on every
    // stream over columns that may carry a true rectangle runs on rectangle tiles: env steps where they exist (VEC = 2;
    // else the env columns are written on band tiles and carry full rectangles), tree steps always.
    if (v.rect_meta) { e->rect_commit = e->rect_ok; e->rect_tree = true; }
    v.patch = L.patch ? 1 : 0;
    v.pw = v.ph = v.pstride = v.plw = v.pcap = v.punits = 0;
    v.blk = nullptr; v.blk_stride = 0; v.blk_pos0 = 0;
    e->patch = L.patch;
    e->patch_waves = L.patch_waves;
    if (L.patch) {
        v.clip_cols = 1;
        v.rect_meta = 1;
        v.pw = L.pg.pw; v.ph = L.pg.ph; v.pstride = L.pg.pstride; v.plw = L.pg.plw; v.punits = L.pg.punits;
        // column records in LDS: what fits the share of a workgroup when kPatchWavesPerCu waves of the kernel are resident per CU
        // (the LDS of a workgroup is allocated in granules of 1280 bytes on gfx950: 16 KB would take 13 of the 128)
        int wgs = kPatchWavesPerCu / L.patch_waves;
        // (at most 13 granules: 170 records are more than the two register pages of the unit loop hold, and the 8 workgroups of the
        // default configuration then leave 30 KB of the CU's LDS to the ground-truth kernel that runs beside the steps)
        const size_t budget = std::min<size_t>((size_t)160 * 1024 / wgs / 1280 * 1280, (size_t)13 * 1280);
        const size_t fixed = PatchLds::bytes(0, v.plw * v.plw, L.patch_waves, v.punits, cfg->rank_cap);
        int pcap = fixed + 17 * kPatchRec * 4 <= budget ? (int)((budget - fixed) / (kPatchRec * 4)) - 1 : 16;
        if (const char* pc = getenv("IPP_PATCH_CAP")) pcap = std::max(8, atoi(pc));  // A/B experiments, overflow tests
        // (a staging capacity below the rank is a multiple of 8: the m x m algebra sums the records in groups of eight, so S does not
        // depend on where a record is staged -- solve_wave_fast)
        v.pcap = pcap >= cfg->rank_cap ? cfg->rank_cap : (pcap & ~7);
        v.blk = reinterpret_cast<float*>(base + L.off_blk);
        v.blk_stride = (int)SplitBlk::floats(v.plw, cfg->rank_cap);
        {   // split step: LDS of the prologue kernel for kSplitMinWP workgroups per SIMD, of the unit kernel as it comes
            if (const char* pwv = getenv("IPP_SPLIT_WAVES")) { const int w = atoi(pwv); if (w >= 1 && w <= 3) e->split_waves = w; }  // A/B
            const size_t budget = (size_t)160 * 1024 / (4 * kSplitMinWP / e->split_waves) / 1280 * 1280;
            const size_t fixed = PatchLds::bytes(0, 0, 1, v.punits, cfg->rank_cap);
            int pc = fixed + 8 * kPatchRec * 4 <= budget ? (int)((budget - fixed) / (kPatchRec * 4)) & ~7 : 8;
            e->pcap_p = pc >= cfg->rank_cap ? cfg->rank_cap : pc;
            e->lds_p = PatchLds::bytes(e->pcap_p, 0, 1, v.punits, cfg->rank_cap);
            e->lds_u = SplitLds::bytes(v.plw, cfg->rank_cap);
            // IPP_SPLIT=<n>: launches of at least n items take the split step (1: all of them, 0: never)
            e->split_min_items = 0;
            if (const char* sp = getenv("IPP_SPLIT")) e->split_min_items = std::max(0, atoi(sp));
        }
    }
    v.win_tiles = L.win_tiles;
    v.cov = reinterpret_cast<float*>(base + L.off_cov);
    v.cov_slot = L.cov_slot_floats;
    v.hdr = reinterpret_cast<ItemHdr*>(base + L.off_hdr);
    v.linv = reinterpret_cast<float*>(base + L.off_linv);
    v.yv = reinterpret_cast<float*>(base + L.off_yv);
    v.q = reinterpret_cast<float*>(base + L.off_q);
    v.wc = reinterpret_cast<float*>(base + L.off_wc);
    v.partial = reinterpret_cast<double*>(base + L.off_partial);
    v.dbg = reinterpret_cast<double*>(base + L.off_dbg);
    v.grf_h = reinterpret_cast<double*>(base + L.off_grfh);
    v.grf_cs = reinterpret_cast<double2*>(base + L.off_grfcs);
    v.grf_g = reinterpret_cast<double*>(base + L.off_grfg);
    v.grf_hp = reinterpret_cast<double*>(base + L.off_grfhp);
    v.grf_amp = reinterpret_cast<double*>(base + L.off_grfamp);
    v.grf_raw = reinterpret_cast<float*>(base + L.off_grfraw);
    v.grf_raw2 = reinterpret_cast<float*>(base + L.off_grfraw2);
    e->tv.node_cap = std::max(0, cfg->node_capacity);
    if (e->tv.node_cap > 0) {
        e->tv.node_cov = reinterpret_cast<float*>(base + L.off_tr_cov);
        e->tv.node_diag = reinterpret_cast<float*>(base + L.off_tr_diag);
        e->tv.node_meta = reinterpret_cast<int*>(base + L.off_tr_meta);
        e->tv.win_cells = L.patch ? L.pg.pstride : L.win_tiles * 64 * L.VEC;
        e->node_diag_scratch = cfg->score_scratch ? reinterpret_cast<float*>(base + L.off_sc_ndiag) : nullptr;
    }
    e->scoring = cfg->score_scratch != 0;
    if (e->scoring) {
        e->sv.hdr = reinterpret_cast<ScoreHdr*>(base + L.off_sc_hdr);
        e->sv.extent = reinterpret_cast<int*>(base + L.off_sc_ext);
        e->sv.mask = reinterpret_cast<float*>(base + L.off_sc_mask);
        e->sv.G = reinterpret_cast<double*>(base + L.off_sc_G);
        e->sv.P = (cfg->state_repr == IPP_FACTOR) ? reinterpret_cast<float*>(base + L.off_sc_P) : nullptr;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_score_band), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    }
    e->n_bands = (L.N + kBandRows - 1) / kBandRows;
    e->prep_lds = prep_lds_bytes(L, *cfg);
    e->q_chunk = 2 * kPipe;  // (k_gain reads Q through the scalar cache: the LDS area only holds the prior table)
    // prior table of the factor base term lives in LDS when the grid is small enough (<= 48 KiB)
    e->lut_cap = (v.mode == IPP_FACTOR && L.N <= 12288) ? L.N : 0;
    e->gain_lds = gain_lds_bytes(v, e->q_chunk, e->lut_cap);
    if (v.mode == IPP_FACTOR && v.window_rows > 0) {
        const size_t MCs = v.meas_cap, LQ = (MCs * MCs + MCs + 3) & ~(size_t)3;
        e->fused = ((v.T == kStepThreads) && v.meas_cap == 9) || e->patch;  // (MC = 25: two launches, see launch_chunk)
        if (const char* fu = getenv("IPP_FUSED")) e->fused = e->fused && atoi(fu) != 0;  // A/B experiments
        const int waves = v.T / 64;
        // prior table rows: a tile that holds new columns lies within window_rows of the footprint, so
        // |drow| <= window_rows + rows a tile spans + footprint height; farther rows (tall footprints) use sqrt / exp
        const int tile_rows = (v.tile_cells + v.W - 1) / v.W + 1;
        e->lut_rows = std::min(v.H, v.window_rows + tile_rows + 6);
        while (e->lut_rows > 0 && (size_t)e->lut_rows * v.W > 12288) --e->lut_rows;  // <= 48 KiB
        const int lutf = e->lut_rows * v.W;
        if (v.meas_cap == 9)
            e->gain_lds = e->fused ? std::max(GainLds<9>::bytes(v.rank_cap, step_work_floats<9>(v.rank_cap), lutf, step_small_floats<9>(), waves, v.win_tiles, v.win_tiles * kWave),
                                              GainLds<9>::bytes(v.rank_cap, step_work_floats<9>(v.rank_cap), lutf, step_small_floats<9>(), waves, v.win_tiles, 0, 0, v.vec))
                                   : GainLds<9>::bytes(v.rank_cap, 0, lutf, 0, waves, v.win_tiles, 0, 0, v.vec);
        else {
            // MC = 25: env steps are two launches (gain_lds = the gain kernel's), tree steps stay the fused tree kernel (its LDS below)
            e->gain_lds = GainLds<25>::bytes(v.rank_cap, 0, lutf, 0, waves, v.win_tiles, 0, 0, v.vec);
            e->tree_ok = v.T == kStepThreads;
            e->tree_fused_lds = std::max(GainLds<25>::bytes(v.rank_cap, step_work_floats<25>(v.rank_cap), lutf, step_small_floats<25>(), waves, v.win_tiles, v.win_tiles * kWave),
                                         GainLds<25>::bytes(v.rank_cap, step_work_floats<25>(v.rank_cap), lutf, step_small_floats<25>(), waves, v.win_tiles, 0, 0, v.vec));
        }
        if (e->patch) e->gain_lds = PatchLds::bytes(v.pcap, v.plw * v.plw, e->patch_waves, v.punits, v.rank_cap);
        if (e->patch && e->patch_waves == 3 && !getenv("IPP_PATCH_CAP")) {
            // Launch-size rule of the default engine (round 6).  Three waves per item shorten the chains of the heaviest items, which end
            // a launch of one or two rounds of workgroup slots (2048 items: 55 M env-steps/s against 37 M with two waves); a launch of
            // many rounds has no tail to shorten and is paid in items in flight: two waves per item are 12 items per CU instead of 8.
            // Same box, two groups of launches: 8192 envs of 50x50 (4096 items per launch) 64.9 M with three waves against 54.5 M with
            // two; 16384 envs 62.9 against 66.5 M; 32768 envs (configs[3] share) 61.8-64.3 against 66.1 M; configs[2] 66.1-67.0
            // against 70.1-71.6 M (profiles/r06_experiments.txt 3).  Same arithmetic per cell, same order: bit-identical results
            // (tests/test_hip_rect_meta.py).  IPP_PATCH_TWO_MIN=<items> (0: never) for A/B.
            const size_t budget = (size_t)160 * 1024 / 12 / 1280 * 1280;
            const size_t fixed = PatchLds::bytes(0, v.plw * v.plw, 2, v.punits, v.rank_cap);
            int p2 = fixed + 17 * kPatchRec * 4 <= budget ? (int)((budget - fixed) / (kPatchRec * 4)) - 1 : 16;
            p2 = p2 >= v.rank_cap ? v.rank_cap : (p2 & ~7);  // (below the rank a multiple of 8: solve_wave_fast sums the records in groups of eight)
            e->pcap2 = std::min(p2, (int)v.pcap);
            e->lds2 = PatchLds::bytes(e->pcap2, v.plw * v.plw, 2, v.punits, v.rank_cap);
            e->two_wave_min_items = 6144;
            if (const char* t = getenv("IPP_PATCH_TWO_MIN")) e->two_wave_min_items = std::max(0, atoi(t));
        }
        if (e->patch && (e->patch_waves == 2 || e->two_wave_min_items > 0) && !getenv("IPP_PATCH_CAP")) {
            // second configuration for large launches: LDS share of 12 workgroups per CU (10 granules of 1280 bytes)
            const size_t budget = (size_t)160 * 1024 / 12 / 1280 * 1280;
            const size_t fixed = PatchLds::bytes(0, v.plw * v.plw, 2, v.punits, v.rank_cap);
            const int pc = fixed + 33 * kPatchRec * 4 <= budget ? (int)((budget - fixed) / (kPatchRec * 4)) - 1 : 0;
            int min_items = 16384;
            if (const char* b = getenv("IPP_PATCH_BIG")) min_items = atoi(b);  // A/B: smallest launch that takes it (0: never)
            if (pc >= 32 && min_items > 0) {
                e->pcap_big = std::min(pc, (int)(e->patch_waves == 3 ? e->pcap2 : v.pcap));
                e->lds_big = PatchLds::bytes(e->pcap_big, v.plw * v.plw, 2, v.punits, v.rank_cap);
                e->big_min_items = min_items;
            }
        }
        if (v.T == kWave && !e->patch)
            e->gain_lds = (LQ + kTileLut) * 4 + (size_t)v.rank_cap * 4 + (8 * MCs) * 4 +
                          (size_t)(v.rank_cap + 8) * 2;
        e->gain_lds = (e->gain_lds + 15) & ~(size_t)15;
        if (e->gain_lds > 160 * 1024) {
            const size_t need = e->gain_lds;
            delete e;
            return fail(-1, "gain kernel needs %zu B of LDS (> 160 KiB): lower rank_cap", need);
        }
    }
    if (e->prep_lds > 160 * 1024) {
        delete e;
        return fail(-1, "prologue needs %zu B of LDS (> 160 KiB): lower rank_cap or max_measurements", e->prep_lds);
    }
    // opt in to > 64 KiB dynamic LDS for the prologue instantiations
    const int lds = (int)e->prep_lds;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prepare<9, IPP_FACTOR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prepare<9, IPP_DENSE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prepare<25, IPP_FACTOR>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_prepare<25, IPP_DENSE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    {  // patch kernels: opt in to more than the default 64 KB of dynamic LDS
        const int pl = 160 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<1>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<2>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<3>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<3, kPatchKP, kPatchMinW, false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<3, kPatchKP, kPatchMinW, false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<4>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<2, IPP_PATCH_BIGKP, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<1, kPatchKP, kSplitMinWP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<2, kPatchKP, kSplitMinWP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_patch<3, kPatchKP, kSplitMinWP, true>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_patch<2>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_patch<3>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_patch<3, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_patch<4>), hipFuncAttributeMaxDynamicSharedMemorySize, pl);
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grf_conv<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_grf_conv<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    const int glds = (int)e->gain_lds;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain<9, 4, IPP_FACTOR>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain<25, 2, IPP_FACTOR>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain_factor<9, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_factor<9, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain_factor<9, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_factor<9, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_step_factor<9, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain_factor<9, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    if (e->fused) e->tree_ok = true;
    e->tree_step_lds = ((e->tree_fused_lds ? e->tree_fused_lds : e->gain_lds) + (size_t)v.rank_cap * 8 + 15) & ~(size_t)15;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_step<9, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->tree_step_lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_step<9, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->tree_step_lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_step<9, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->tree_step_lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_step<9, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->tree_step_lds);
    if (e->fused && e->tv.node_cap > 0 && v.meas_cap == 9) {
        // Split tree steps: worth it once the launch fills the device several times over (below that the second launch
        // and the scratch round trip of L^-1 | Q cost more than the occupancy gains); IPP_TREE_SPLIT=<min items> / 0
        e->tree_gain_lds = (GainLds<9>::bytes(v.rank_cap, 0, e->lut_rows * v.W, 0, e->tree_T / kWave, v.win_tiles, 0, v.rank_cap, v.vec) + 15) & ~(size_t)15;
        e->tree_split_min = 2048;
        if (const char* ts = getenv("IPP_TREE_SPLIT")) e->tree_split_min = atoi(ts);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_prepare<9>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_gain<9, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->tree_gain_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_gain<9, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->tree_gain_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_gain<9, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->tree_gain_lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_gain<9, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->tree_gain_lds);
    }
    if (e->tv.node_cap > 0 && v.meas_cap == 25) {
        // MC = 25 tree steps: ALWAYS the two launches (prologue + gain kernel), like the env steps of these engines -- the fused
        // k_tree_step<25, 2> spilled 381 VGPRs (1412 bytes of scratch per lane) and is no longer instantiated
        e->tree_gain_lds = (GainLds<25>::bytes(v.rank_cap, 0, e->lut_rows * v.W, 0, e->tree_T / kWave, v.win_tiles, 0, v.rank_cap, v.vec) + 15) & ~(size_t)15;
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_prepare<25>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tree_gain<25, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)e->tree_gain_lds);
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain_factor<25, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain<9, 4, IPP_DENSE>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gain<25, 2, IPP_DENSE>), hipFuncAttributeMaxDynamicSharedMemorySize, glds);
    (void)hipGetLastError();
    // state starts zeroed (rank 0, padding 0); envs must still be ipp_reset before use
    HIP_TRY(hipMemset(base, 0, L.off_cov));
    hipLaunchKernelGGL(k_init_gt_slots, dim3((cfg->capacity + 255) / 256), dim3(256), 0, nullptr, v);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(base + L.off_hdr, 0, L.total - L.off_hdr));
    if (cfg->x_dim == cfg->y_dim) {
        std::vector<double> h;
        grf_kernel_host(cfg->y_dim, cfg->x_dim, cfg->cluster_radius, h);
        HIP_TRY(hipMemcpy(v.grf_h, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
        const int n = cfg->x_dim;
        e->grf_dft = (n % 2 == 0) && n >= 4 && n <= 256;  // odd n: the reference's amp is not even (ground_truths.py:8-11)
        if (e->grf_dft && n <= 128) {
            const int tt = (n + 15) / 16;
            std::vector<double> hp, amp;
            bool ok = grf_hartley_tables_host(n, 16 * tt, cfg->cluster_radius, hp, amp);
            if (ok) {
                HIP_TRY(hipMemcpy(v.grf_hp, hp.data(), hp.size() * sizeof(double), hipMemcpyHostToDevice));
                HIP_TRY(hipMemcpy(v.grf_amp, amp.data(), amp.size() * sizeof(double), hipMemcpyHostToDevice));
                e->grf_tt = tt;
                const int hl = (int)grf_hartley_lds_bytes(tt);
                const void* fn[] = {(const void*)&k_grf_hartley<1, 4>, (const void*)&k_grf_hartley<2, 4>, (const void*)&k_grf_hartley<3, 4>, (const void*)&k_grf_hartley<4, 4>,
                                    (const void*)&k_grf_hartley<5, 8>, (const void*)&k_grf_hartley<6, 8>, (const void*)&k_grf_hartley<7, 8>, (const void*)&k_grf_hartley<8, 4>};
                (void)hipFuncSetAttribute(fn[tt - 1], hipFuncAttributeMaxDynamicSharedMemorySize, hl);
                if (n == 50) (void)hipFuncSetAttribute((const void*)&k_grf_hartley<4, 4, 13>, hipFuncAttributeMaxDynamicSharedMemorySize, hl);
                if (n == 100) (void)hipFuncSetAttribute((const void*)&k_grf_hartley<7, 8, 25>, hipFuncAttributeMaxDynamicSharedMemorySize, hl);
                e->grf_fft = (n == 50 || n == 100);
                if (const char* gf = getenv("IPP_GRF_FFT")) e->grf_fft = e->grf_fft && atoi(gf) != 0;  // A/B: the GEMM form
            }
        }
        if (e->grf_dft) {
            std::vector<double> cs, g;
            grf_dft_tables_host(n, cfg->cluster_radius, cs, g);
            HIP_TRY(hipMemcpy(v.grf_cs, cs.data(), cs.size() * sizeof(double), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(v.grf_g, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice));
            const long budget = (n <= 64) ? 32768 : 65536;
            const long fixed = (n <= 100 ? (long)n * n * 4 : 0) + (long)n * 16 + 256;
            e->grf_kc = (int)std::max(1L, std::min((long)(n / 2 + 1), (budget - fixed) / ((long)n * 40)));
        }
    }
    if (const char* ch = getenv("IPP_STEP_CHUNKS")) e->step_chunks = atoi(ch);
    if (hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking) == hipSuccess) {
        (void)hipEventCreateWithFlags(&e->ev_begin, hipEventDisableTiming);
        (void)hipEventCreateWithFlags(&e->ev_end, hipEventDisableTiming);
        for (auto& ev : e->ev_prep) (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    } else {
        e->side = nullptr;
        (void)hipGetLastError();
    }
    *engine = e;
    return 0;
}

int ipp_engine_destroy(void* engine) {
#if IPP_MCTS_CLOCKS
    {
        unsigned long long c[16];
        if (hipMemcpyFromSymbol(c, HIP_SYMBOL(g_mclk), sizeof c) == hipSuccess && c[8]) {
            const double lv = (double)c[8], ds = (double)c[9];
            fprintf(stderr, "[k_mcts_select, us per level (lane 0, sections serialised)] rows + min/max %.2f  PUCT + argmax %.2f  edge fields + cost %.2f  "
                            "hash lookup %.2f  child record %.2f  bookkeeping %.2f | per descent: start %.2f  leaf + fence %.2f | levels %.0f descents %.0f (%.2f levels each)\n",
                    c[1] / lv / 100, c[2] / lv / 100, c[3] / lv / 100, c[4] / lv / 100, c[5] / lv / 100, c[6] / lv / 100, c[0] / ds / 100, c[7] / ds / 100, lv, ds, lv / ds);
            memset(c, 0, sizeof c);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mclk), c, sizeof c);
        }
    }
#endif
    Engine* e = as_engine(engine);
    if (!e) return 0;
    prof_drain(e);
    if (e->side) {
        (void)hipStreamSynchronize(e->side);
        (void)hipStreamDestroy(e->side);
        (void)hipEventDestroy(e->ev_begin);
        (void)hipEventDestroy(e->ev_end);
        for (auto& ev : e->ev_prep) (void)hipEventDestroy(ev);
    }
    delete e;
    return 0;
}

int ipp_engine_info(void* engine, ipp_info* out) {
    Engine* e = as_engine(engine);
    if (!e || !out) return fail(-1, "null argument");
    out->abi_version = IPP_ABI_VERSION;
    out->n_cells = e->v.N;
    out->n_pad = e->v.Npad;
    out->tile_threads = e->v.T;
    out->n_tiles = e->v.n_tiles;
    out->meas_cap = e->v.meas_cap;
    out->fp_cap = e->v.fp_cap;
    out->window_rows = e->v.window_rows;
    out->arena_bytes = e->used_bytes;
    out->cov_slot_bytes = e->v.cov_slot * 4;
    out->step_lds_bytes = e->gain_lds;
    out->fused_step = e->fused ? 1 : 0;
    out->patch_layout = e->patch ? 1 : 0;
    out->patch_waves = e->patch ? e->patch_waves : 0;
    out->patch_big_min_items = e->patch ? e->big_min_items : 0;
    out->patch_two_wave_min_items = e->patch ? e->two_wave_min_items : 0;
    out->patch_split_min_items = e->patch ? e->split_min_items : 0;
    return 0;
}

int ipp_reset_episode(void* engine, const int32_t* env_ids, int32_t n, const double* prior_scale, const float* gt,
                      const float* white_noise, double* prev_action, const double* init_action, void* stream) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    if (n < 0 || n > e->v.max_batch) return fail(-1, "n = %d outside [0, max_batch = %d]", n, e->v.max_batch);
    if (n == 0) return 0;
    if (!env_ids && n > e->v.cap) return fail(-1, "n exceeds capacity");
    if (prev_action && !init_action) return fail(-1, "prev_action needs init_action");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const View& v = e->v;
    HIP_TRY(hipSetDevice(e->device));
    InitAction ia = {{0.0, 0.0, 0.0}};
    if (init_action) for (int k = 0; k < 3; ++k) ia.p[k] = init_action[k];
    hipLaunchKernelGGL(k_reset_small, dim3((v.Npad + 255) / 256, n), dim3(256), 0, s, v, env_ids, n, prior_scale, gt, prev_action, ia);
    if (!gt && white_noise) {
        if (v.W != v.H) return fail(-1, "device GRF needs a square grid (the reference transposes its dims, simulations/simulations.py:45-47)");
        if (int rc = launch_grf(e, n, white_noise, v.grf_raw, env_ids, nullptr, s)) return rc;
    }
    if (v.mode == IPP_DENSE) {
        const int n_ctiles = (v.Npad + 1023) / 1024;
        const int grid = grid_for(n, e->n_bands * n_ctiles);
        hipLaunchKernelGGL(k_reset_dense, dim3(grid), dim3(256), 0, s, v, env_ids, n, e->n_bands, n_ctiles);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_reset(void* engine, const int32_t* env_ids, int32_t n, const double* prior_scale, const float* gt,
              const float* white_noise, void* stream) {
    return ipp_reset_episode(engine, env_ids, n, prior_scale, gt, white_noise, nullptr, nullptr, stream);
}

static int step_impl(void* engine, const int32_t* env_ids, const int32_t* dst_ids, int32_t n, const double* action,
                     const double* prev_action, const float* meas_noise, uint32_t flags, float* reward, int32_t* status,
                     void* stream, const AutoReset& ar) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    if (!action || !prev_action || !reward) return fail(-1, "action, prev_action and reward are required");
    if (n < 0 || n > e->v.max_batch) return fail(-1, "n = %d outside [0, max_batch = %d]", n, e->v.max_batch);
    if (n == 0) return 0;
    if (!env_ids && n > e->v.cap) return fail(-1, "n exceeds capacity");
    if (flags & ~(IPP_COV_ONLY | IPP_PREDICT_ONLY | IPP_ADAPTIVE | IPP_USE_FLIGHT_TIME | IPP_GIVEN_OBSERVATION | IPP_UPDATE_PREV)) return fail(-1, "unknown flag bits 0x%x", flags);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    e->last_n = n;
    if (e->v.mode == IPP_FACTOR && dst_ids && !(flags & IPP_PREDICT_ONLY)) {
        // factor state: an out-of-place step is a slot copy followed by an in-place step on the copy
        // (the dense downdate kernel writes P_dst = P_src - Wc Wc^T directly instead)
        if (!env_ids) return fail(-1, "dst_ids needs explicit env_ids");
        if (int rc = ipp_fork(engine, env_ids, dst_ids, n, stream)) return rc;
        env_ids = dst_ids;
        dst_ids = nullptr;
    }
    // the fused kernel resets the flagged envs itself; every other path gets a separate launch behind the step
    const bool in_kernel = e->fused;
    const AutoReset none = {nullptr, nullptr, nullptr, nullptr, {0.0, 0.0, 0.0}};
    int rc;
    if (e->v.meas_cap == 9)
        rc = (e->v.vec == 2) ? launch_step<9, 2>(e, env_ids, dst_ids, n, action, prev_action, meas_noise, flags, reward, status, s, in_kernel ? ar : none)
                             : launch_step<9, 4>(e, env_ids, dst_ids, n, action, prev_action, meas_noise, flags, reward, status, s, in_kernel ? ar : none);
    else
        rc = launch_step<25, 2>(e, env_ids, dst_ids, n, action, prev_action, meas_noise, flags, reward, status, s, in_kernel ? ar : none);
    if (rc == 0 && ar.src && !in_kernel) {
        hipLaunchKernelGGL(k_reset_flagged, dim3((e->v.Npad + 255) / 256, n), dim3(256), 0, s, e->v, env_ids, n, ar);
        HIP_TRY(hipGetLastError());
    }
    return rc;
}

int ipp_step(void* engine, const int32_t* env_ids, const int32_t* dst_ids, int32_t n, const double* action,
             const double* prev_action, const float* meas_noise, uint32_t flags, float* reward, int32_t* status,
             void* stream) {
    const AutoReset none = {nullptr, nullptr, nullptr, nullptr, {0.0, 0.0, 0.0}};
    return step_impl(engine, env_ids, dst_ids, n, action, prev_action, meas_noise, flags, reward, status, stream, none);
}

int ipp_step_autoreset(void* engine, const int32_t* env_ids, int32_t n, const double* action, double* prev_action,
                       const float* meas_noise, uint32_t flags, float* reward, int32_t* status, const int32_t* reset_src,
                       const float* reset_gt, const double* init_action, void* stream) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    if (e->v.mode != IPP_FACTOR) return fail(-1, "ipp_step_autoreset: factor state only (dense engines: ipp_step + ipp_reset_episode)");
    if (flags & IPP_PREDICT_ONLY) return fail(-1, "ipp_step_autoreset: not with IPP_PREDICT_ONLY");
    if (reset_src && !init_action) return fail(-1, "reset_src needs init_action");  // (reset_gt == NULL: the ground truths were staged into the alternate planes)
    AutoReset ar = {reset_src, reset_gt, reset_src ? e->reset_prior : nullptr, reset_src ? prev_action : nullptr, {0.0, 0.0, 0.0}};
    if (init_action) for (int k = 0; k < 3; ++k) ar.init[k] = init_action[k];
    return step_impl(engine, env_ids, nullptr, n, action, prev_action, meas_noise, flags, reward, status, stream, ar);
}

int ipp_step_parts(void* engine, int32_t n, const double* action, double* prev_action, const float* meas_noise,
                   uint32_t flags, float* reward, int32_t* status, const int32_t* reset_src, const float* reset_gt,
                   const double* init_action, int32_t n_parts, const int32_t* part_begin, void* const* streams) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    if (!action || !prev_action || !reward || !part_begin || !streams) return fail(-1, "null argument");
    if (e->v.mode != IPP_FACTOR || !(e->fused || e->patch) || e->v.meas_cap != 9 || e->v.vec != 2)
        return fail(-1, "ipp_step_parts: engines whose step is one fused kernel only (ipp_info.fused_step)");
    // (predict-only parts: reward / status of every item, no state write -- consecutive calls do not depend on each other at all)
    if ((flags & IPP_PREDICT_ONLY) && (reset_src || (flags & IPP_UPDATE_PREV))) return fail(-1, "ipp_step_parts: IPP_PREDICT_ONLY with resets or IPP_UPDATE_PREV");
    if (flags & ~(IPP_COV_ONLY | IPP_PREDICT_ONLY | IPP_ADAPTIVE | IPP_USE_FLIGHT_TIME | IPP_GIVEN_OBSERVATION | IPP_UPDATE_PREV)) return fail(-1, "unknown flag bits 0x%x", flags);
    if (n <= 0 || n > e->v.max_batch || n > e->v.cap) return fail(-1, "n = %d outside [1, min(max_batch, capacity)]", n);
    if (!e->v.item_order || e->v.item_order_n != n) return fail(-1, "ipp_step_parts: needs the dispatch order of all n items (ipp_set_item_order)");
    if (n_parts < 1 || n_parts > kMaxChunks) return fail(-1, "n_parts = %d outside [1, %d]", n_parts, kMaxChunks);
    if (part_begin[0] != 0 || part_begin[n_parts] != n) return fail(-1, "part_begin has to run from 0 to n");
    for (int p = 0; p < n_parts; ++p)
        if (part_begin[p + 1] <= part_begin[p]) return fail(-1, "part %d is empty", p);
    if (reset_src && !init_action) return fail(-1, "reset_src needs init_action");
    AutoReset ar = {reset_src, reset_gt, reset_src ? e->reset_prior : nullptr, reset_src ? prev_action : nullptr, {0.0, 0.0, 0.0}};
    if (init_action) for (int k = 0; k < 3; ++k) ar.init[k] = init_action[k];
    HIP_TRY(hipSetDevice(e->device));
    e->last_n = n;
    for (int p = 0; p < n_parts; ++p) {
        View v = e->v;  // (the per-item arrays keep their batch indexing: a part is a range of positions of the order)
        v.blk_pos0 = part_begin[p];  // (the split step's item blocks are indexed by the dispatch position)
        v.item_order = e->v.item_order + part_begin[p];
        v.item_order_n = part_begin[p + 1] - part_begin[p];
        launch_chunk<9, 2>(e, v, nullptr, nullptr, v.item_order_n, action, prev_action, meas_noise, flags, reward, status,
                           reinterpret_cast<hipStream_t>(streams[p]), nullptr, ar);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_generate_grf(void* engine, int32_t n, const float* white_noise, float* gt_out, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !white_noise || !gt_out) return fail(-1, "null argument");
    if (n < 0 || n > e->v.max_batch) return fail(-1, "n = %d outside [0, max_batch = %d]", n, e->v.max_batch);
    if (n == 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    return launch_grf(e, n, white_noise, e->v.grf_raw2, nullptr, gt_out, reinterpret_cast<hipStream_t>(stream));
}

int ipp_generate_grf_rows(void* engine, int32_t n, const int32_t* row_ids, int64_t row_offset, uint64_t seed, uint64_t subsequence,
                          float* gt_out, void* stream) {
    return ipp_generate_grf_groups(engine, n, 0, nullptr, row_ids, row_offset, seed, subsequence, gt_out, stream);
}

int ipp_generate_grf_groups(void* engine, int32_t n, int32_t group_rows, const int64_t* group_subsequence, const int32_t* row_ids,
                            int64_t row_offset, uint64_t seed, uint64_t subsequence, float* gt_out, void* stream) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    if (!gt_out && !row_ids) return fail(-1, "ipp_generate_grf_groups: gt_out == NULL writes the alternate plane of env row_ids[i]: row_ids needed");
    if (n < 0 || n > e->v.max_batch) return fail(-1, "n = %d outside [0, max_batch = %d]", n, e->v.max_batch);
    if (n == 0) return 0;
    if (group_rows < 0 || (group_rows > 0 && (!group_subsequence || (n + group_rows - 1) / group_rows > 16)))
        return fail(-1, "ipp_generate_grf_groups: at most 16 groups of group_rows fields, with their subsequence offsets");
    if (!(e->grf_tt > 0 && e->grf_fft)) return fail(-3, "ipp_generate_grf_rows: this grid has no generator that draws its own noise (ipp_fill_normal_rows + ipp_generate_grf)");
    HIP_TRY(hipSetDevice(e->device));
    GrfNoise gn = {row_ids, (long long)row_offset, seed, subsequence, group_rows, {0}, gt_out ? 0 : 1};
    for (int g = 0; group_rows > 0 && g < (n + group_rows - 1) / group_rows; ++g) gn.group_subseq[g] = (long long)group_subsequence[g];
    return launch_grf(e, n, nullptr, e->v.grf_raw2, nullptr, gt_out, reinterpret_cast<hipStream_t>(stream), &gn);
}

int ipp_observe(void* engine, const int32_t* env_ids, int32_t n, const double* action, const float* meas_noise,
                float* z_out, int32_t* m_out, int32_t* shape_out, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !action || !z_out || !m_out) return fail(-1, "null argument");
    if (n < 0 || n > e->v.max_batch) return fail(-1, "n = %d outside [0, max_batch = %d]", n, e->v.max_batch);
    if (n == 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    const View& v = e->v;
    // the prologue kernel in observation-only mode (prev_action is unused there: pass action)
    if (v.meas_cap == 9) {
        if (v.mode == IPP_FACTOR)
            hipLaunchKernelGGL((k_prepare<9, IPP_FACTOR>), dim3(n), dim3(kPrepThreads), e->prep_lds, s, v, env_ids, (const int*)nullptr, n, action, action, meas_noise, 0u, (int*)nullptr, z_out, m_out, shape_out);
        else
            hipLaunchKernelGGL((k_prepare<9, IPP_DENSE>), dim3(n), dim3(kPrepThreads), e->prep_lds, s, v, env_ids, (const int*)nullptr, n, action, action, meas_noise, 0u, (int*)nullptr, z_out, m_out, shape_out);
    } else {
        if (v.mode == IPP_FACTOR)
            hipLaunchKernelGGL((k_prepare<25, IPP_FACTOR>), dim3(n), dim3(kPrepThreads), e->prep_lds, s, v, env_ids, (const int*)nullptr, n, action, action, meas_noise, 0u, (int*)nullptr, z_out, m_out, shape_out);
        else
            hipLaunchKernelGGL((k_prepare<25, IPP_DENSE>), dim3(n), dim3(kPrepThreads), e->prep_lds, s, v, env_ids, (const int*)nullptr, n, action, action, meas_noise, 0u, (int*)nullptr, z_out, m_out, shape_out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_set_uav(void* engine, double max_v, double max_a) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    if (!(max_v > 0) || !(max_a > 0)) return fail(-1, "max_v and max_a must be positive");
    e->v.vmax = max_v;
    e->v.amax = max_a;
    return 0;
}

int ipp_set_item_order(void* engine, const int32_t* order, int32_t n) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    if (order && (n <= 0 || n > e->v.max_batch)) return fail(-1, "n = %d outside [1, max_batch = %d]", n, e->v.max_batch);
    e->v.item_order = order;
    e->v.item_order_n = order ? n : 0;
    return 0;
}

int ipp_set_reset_prior(void* engine, const double* prior) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    e->reset_prior = prior;
    return 0;
}

int ipp_set_adaptive(void* engine, double value_threshold, double interval_factor) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    e->v.thr = value_threshold;
    e->v.kf = interval_factor;
    return 0;
}

static int score_actions_impl(void* engine, int32_t env_id, const ScorePath& path, const float* node_diag, const double* actions,
                              int32_t n, const double* prev_action, uint32_t flags, float* reward, int32_t* status, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !actions || !prev_action || !reward) return fail(-1, "null argument");
    if (!e->scoring) return fail(-1, "ipp_score_actions needs ipp_config.score_scratch = 1");
    if (int rc = check_env(e, env_id)) return rc;
    if (n < 0 || n > e->v.max_batch) return fail(-1, "n = %d outside [0, max_batch = %d]", n, e->v.max_batch);
    if (n == 0) return 0;
    if (flags & ~(IPP_COV_ONLY | IPP_PREDICT_ONLY | IPP_ADAPTIVE | IPP_USE_FLIGHT_TIME)) return fail(-1, "unsupported flag bits 0x%x", flags);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    const View& v = e->v;
    int band_kc = 64;  // columns k per LDS tile: 2 row blocks of W cells in fp64
    while (band_kc > 8 && (size_t)2 * v.W * (band_kc + 1) * sizeof(double) > 64 * 1024) band_kc /= 2;
    const size_t band_lds = (size_t)2 * v.W * (band_kc + 1) * sizeof(double);
    if (band_lds > 150 * 1024) return fail(-1, "grid rows of %d cells exceed the band kernel's LDS tiles", v.W);
    ScoreView sv = e->sv;
    if (v.mode == IPP_DENSE) sv.P = v.cov + (size_t)env_id * v.cov_slot;
    PrevAction pa = {{prev_action[0], prev_action[1], prev_action[2]}};
    HIP_TRY(hipMemsetAsync(sv.extent, 0, 8, s));
    const int hdr_blocks = std::max((n + 255) / 256, std::min(64, (v.Npad + 255) / 256));
    if (v.meas_cap == 9) hipLaunchKernelGGL((k_score_hdr<9>), dim3(hdr_blocks), dim3(256), 0, s, v, sv, env_id, actions, n, pa, flags, reward, status, node_diag);
    else                 hipLaunchKernelGGL((k_score_hdr<25>), dim3(hdr_blocks), dim3(256), 0, s, v, sv, env_id, actions, n, pa, flags, reward, status, node_diag);
    if (v.mode == IPP_FACTOR && path.depth > 0)
        hipLaunchKernelGGL(k_score_densify<true>, dim3(v.Npad / 64, (v.N + 63) / 64), dim3(256), 0, s, v, sv, env_id, path, e->tv.node_cov, e->tv.node_meta, e->tv.win_cells);
    else if (v.mode == IPP_FACTOR)
        hipLaunchKernelGGL(k_score_densify<false>, dim3(v.Npad / 64, (v.N + 63) / 64), dim3(256), 0, s, v, sv, env_id, path, nullptr, nullptr, 0);
    hipLaunchKernelGGL(k_score_band, dim3(v.H, kScoreDCap + 1, kScoreSplit), dim3(256), band_lds, s, v, sv, band_kc);
    if (v.meas_cap == 9) hipLaunchKernelGGL((k_score_eval<9>), dim3((n + 3) / 4), dim3(256), 0, s, v, sv, n, reward);
    else                 hipLaunchKernelGGL((k_score_eval<25>), dim3((n + 3) / 4), dim3(256), 0, s, v, sv, n, reward);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int tree_step_impl(void* engine, const int32_t* root_ids, const int32_t* path_ids, const int32_t* new_ids, int32_t n,
                          const double* action, const double* prev_action, uint32_t flags, float* reward, int32_t* status,
                          void* stream, const int32_t* n_dev, const TreeEdgeOut* edge_out = nullptr) {
    Engine* e = as_engine(engine);
    if (!e || !root_ids || !path_ids || !action || !prev_action || !reward) return fail(-1, "null argument");
    if (e->tv.node_cap <= 0) return fail(-1, "ipp_tree_step needs ipp_config.node_capacity > 0");
    if (!e->tree_ok) return fail(-1, "ipp_tree_step needs IPP_FACTOR with window_rows > 0 and the default tile_threads");
    if (n < 0 || n > e->v.max_batch) return fail(-1, "n = %d outside [0, max_batch = %d]", n, e->v.max_batch);
    if (n == 0) return 0;
    if (flags & ~(IPP_COV_ONLY | IPP_PREDICT_ONLY | IPP_ADAPTIVE | IPP_USE_FLIGHT_TIME)) return fail(-1, "unsupported flag bits 0x%x", flags);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    e->last_n = n;
    const View& v = e->v;
    if (e->patch) {  // tree nodes as patches: one fused kernel for every launch size (k_tree_patch.h)
        // (LDS sized for the waves this kernel really has: the step kernel may run 1 wave per item -- IPP_PATCH_WAVES --, the tree kernel 2 to 4)
        const int nw = e->patch_waves >= 2 ? e->patch_waves : 2;
        const size_t tlds = PatchLds::bytes(v.pcap, v.plw * v.plw, nw, v.punits, v.rank_cap);
        const TreeEdgeOut eo = edge_out ? *edge_out : TreeEdgeOut{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
        if (nw == 3 && v.rank_cap <= 192)
            timed_launch(e, 0, k_tree_patch<3, 1>, dim3(n), dim3(192), tlds, s, v, e->tv, root_ids, path_ids, new_ids, n, action, prev_action, flags, status, reward, n_dev, eo);
        else if (nw == 3)
            timed_launch(e, 0, k_tree_patch<3>, dim3(n), dim3(192), tlds, s, v, e->tv, root_ids, path_ids, new_ids, n, action, prev_action, flags, status, reward, n_dev, eo);
        else if (nw == 4)
            timed_launch(e, 0, k_tree_patch<4>, dim3(n), dim3(256), tlds, s, v, e->tv, root_ids, path_ids, new_ids, n, action, prev_action, flags, status, reward, n_dev, eo);
        else
            timed_launch(e, 0, k_tree_patch<2>, dim3(n), dim3(128), tlds, s, v, e->tv, root_ids, path_ids, new_ids, n, action, prev_action, flags, status, reward, n_dev, eo);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (n_dev) return fail(-1, "device-side item counts need the patch layout (ipp_info.patch_layout)");
    if (v.meas_cap == 9 && e->tree_split_min > 0 && n >= e->tree_split_min) {
        timed_launch(e, 2, k_tree_prepare<9>, dim3(n), dim3(kPrepThreads), e->prep_lds, s, v, e->tv, root_ids, path_ids, new_ids, n, action,
                     prev_action, flags, status);
#define IPP_TREE_GAIN(V, R) timed_launch(e, 0, k_tree_gain<9, V, R>, dim3(n), dim3(e->tree_T), e->tree_gain_lds, s, v, e->tv, (const float*)v.q, \
                                         root_ids, path_ids, new_ids, n, flags, e->lut_rows, reward)
        if (v.vec == 2) { if (e->rect_tree) IPP_TREE_GAIN(2, true); else IPP_TREE_GAIN(2, false); }
        else            { if (e->rect_tree) IPP_TREE_GAIN(4, true); else IPP_TREE_GAIN(4, false); }
#undef IPP_TREE_GAIN
    } else if (v.meas_cap == 9) {
#define IPP_TREE_STEP(V, R) timed_launch(e, 0, k_tree_step<9, V, R>, dim3(n), dim3(kStepThreads), e->tree_step_lds, s, v, e->tv, root_ids, path_ids, \
                                         new_ids, n, action, prev_action, flags, e->lut_rows, status, reward)
        if (v.vec == 2) { if (e->rect_tree) IPP_TREE_STEP(2, true); else IPP_TREE_STEP(2, false); }
        else            { if (e->rect_tree) IPP_TREE_STEP(4, true); else IPP_TREE_STEP(4, false); }
#undef IPP_TREE_STEP
    }
    else {
        if (v.vec != 2) return fail(-1, "tree steps with max_measurements = 25 need two cells per lane");
        timed_launch(e, 2, k_tree_prepare<25>, dim3(n), dim3(kPrepThreads), e->prep_lds, s, v, e->tv, root_ids, path_ids, new_ids, n, action,
                     prev_action, flags, status);
        timed_launch(e, 0, k_tree_gain<25, 2>, dim3(n), dim3(e->tree_T), e->tree_gain_lds, s, v, e->tv, (const float*)v.q, root_ids, path_ids,
                     new_ids, n, flags, e->lut_rows, reward);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_tree_step(void* engine, const int32_t* root_ids, const int32_t* path_ids, const int32_t* new_ids, int32_t n,
                  const double* action, const double* prev_action, uint32_t flags, float* reward, int32_t* status,
                  void* stream) {
    return tree_step_impl(engine, root_ids, path_ids, new_ids, n, action, prev_action, flags, reward, status, stream, nullptr);
}

// ---- device-side tree search (k_mcts.h)
static int mcts_check(const ipp_mcts_tables* t) {
    if (!t) return fail(-1, "null tables");
    if (t->roots <= 0 || t->kmax <= 0 || t->nodes_per_root <= 1 || t->dev_per_root <= 0 || t->wave <= 0 || t->max_depth <= 0)
        return fail(-1, "ipp_mcts_tables: roots, kmax, nodes_per_root, dev_per_root, wave and max_depth must be positive");
    if (t->table_size < 2 * t->nodes_per_root || (t->table_size & (t->table_size - 1)))
        return fail(-1, "ipp_mcts_tables.table_size = %d must be a power of two >= 2 nodes_per_root", t->table_size);
    if (t->max_depth > 8 || t->horizon + 1 > kMctsPath) return fail(-1, "ipp_mcts_tables: horizon %d exceeds the path length %d", t->horizon, kMctsPath - 1);
    if (!t->actions || !t->cell_action || !t->off_x || !t->off_y || !t->zkey || !t->uniform_ps || !t->t_idx || !t->t_ps || !t->t_nsa ||
        !t->t_qsa || !t->t_num || !t->t_child || !t->n_k || !t->n_ns || !t->n_flags || !t->n_hash || !t->n_value || !t->n_devpath ||
        !t->root_count || !t->dev_count || !t->h_keys || !t->h_vals || !t->p_node || !t->p_k || !t->p_cost || !t->p_len || !t->leaf ||
        !t->pend_node || !t->pend_depth || !t->pend_sim || !t->pend_prev || !t->pend_budget || !t->pend_count || !t->rq_root ||
        !t->rq_parent || !t->rq_k || !t->rq_child || !t->rq_newdev || !t->rq_cost || !t->rq_prev || !t->rq_action || !t->rq_count ||
        !t->ts_paths || !t->ts_reward || !t->ts_status || !t->err)
        return fail(-1, "ipp_mcts_tables: null buffer");
    if (t->root_base < 0 || t->dev_base < 0 || t->scratch_base < 0) return fail(-1, "ipp_mcts_tables: root_base, dev_base and scratch_base must be >= 0");
    return 0;
}

int ipp_mcts_select(const ipp_mcts_tables* t, const int32_t* root_env, const double* prev0, const double* budget0, int32_t depth,
                    int32_t sim0, int32_t wave, uint64_t seed, void* stream) {
    if (int rc = mcts_check(t)) return rc;
    if (!root_env || !prev0 || !budget0) return fail(-1, "null argument");
    if (wave <= 0 || wave > t->wave) return fail(-1, "wave = %d outside [1, %d]", wave, t->wave);
    if (depth < 0 || t->horizon + 1 - depth > t->max_depth) return fail(-1, "depth %d: the descents need %d steps, max_depth = %d", depth, t->horizon + 1 - depth, t->max_depth);
    HIP_TRY(hipSetDevice(t->device));
    const int blocks = (t->roots * kWave + 255) / 256;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int ne = (t->kmax + 63) / 64;  // edge rows in registers up to 256 valid actions per node
    if (ne == 1)      hipLaunchKernelGGL(k_mcts_select<1>, dim3(blocks), dim3(256), 0, s, *t, root_env, prev0, budget0, (int)depth, (int)sim0, (int)wave, seed);
    else if (ne == 2) hipLaunchKernelGGL(k_mcts_select<2>, dim3(blocks), dim3(256), 0, s, *t, root_env, prev0, budget0, (int)depth, (int)sim0, (int)wave, seed);
    else if (ne == 3) hipLaunchKernelGGL(k_mcts_select<3>, dim3(blocks), dim3(256), 0, s, *t, root_env, prev0, budget0, (int)depth, (int)sim0, (int)wave, seed);
    else if (ne == 4) hipLaunchKernelGGL(k_mcts_select<4>, dim3(blocks), dim3(256), 0, s, *t, root_env, prev0, budget0, (int)depth, (int)sim0, (int)wave, seed);
    else              hipLaunchKernelGGL(k_mcts_select<0>, dim3(blocks), dim3(256), 0, s, *t, root_env, prev0, budget0, (int)depth, (int)sim0, (int)wave, seed);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_mcts_steps(void* engine, const ipp_mcts_tables* t, int32_t first, int32_t n, uint32_t flags, void* stream) {
    if (int rc = mcts_check(t)) return rc;
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    const int64_t cap = (int64_t)t->max_depth * t->roots * t->wave;
    // n < 0: the request count stays on the device (t->rq_count[0]); the launch is sized for roots x wave items (patch engines:
    // k_tree_patch reads the count) -- a search wave needs no read-back between select and the steps
    const bool dev_count = n < 0;
    if (dev_count && !e->patch) return fail(-1, "n < 0 (device-side count) needs the patch layout");
    if (dev_count && first != 0) return fail(-1, "n < 0 (device-side count) goes with first = 0");
    if (first < 0 || (!dev_count && (int64_t)first + n > cap)) return fail(-1, "requests [%d, %d) outside the list of %lld", first, first + n, (long long)cap);
    if (n == 0) return 0;
    const int n_grid = dev_count ? (int)std::min<int64_t>(e->v.max_batch, (int64_t)t->roots * t->wave) : n;
    if ((int64_t)t->dev_base + (int64_t)t->roots * t->dev_per_root > e->tv.node_cap)
        return fail(-1, "dev_base + roots x dev_per_root = %lld device nodes exceed ipp_config.node_capacity = %d",
                    (long long)t->dev_base + (long long)t->roots * t->dev_per_root, e->tv.node_cap);
    if (t->scratch_base > 0 && !e->patch) return fail(-1, "ipp_mcts_tables.scratch_base needs the patch layout");
    if ((int64_t)t->scratch_base + n_grid > e->v.max_batch)
        return fail(-1, "scratch_base + items = %lld exceed ipp_config.max_batch = %d", (long long)t->scratch_base + n_grid, e->v.max_batch);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    const size_t o = (size_t)first;
    // (the path arguments ts_paths were written by ipp_mcts_select; the patch kernel writes the edge numerators itself)
    const TreeEdgeOut eo{t->rq_parent + o, t->rq_k + o, t->rq_cost + o, t->t_num, t->err, t->kmax, (int)t->scratch_base};
    if (int rc = tree_step_impl(engine, t->rq_root + o, t->ts_paths + kMctsPath * o, t->rq_newdev + o, n_grid, t->rq_action + 3 * o, t->rq_prev + 3 * o,
                                flags, t->ts_reward + o, t->ts_status + o, stream, dev_count ? t->rq_count : nullptr, e->patch ? &eo : nullptr))
        return rc;
    if (!e->patch) hipLaunchKernelGGL(k_mcts_apply, dim3((n_grid + 255) / 256), dim3(256), 0, s, *t, (int)first, (int)n);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_mcts_expand(const ipp_mcts_tables* t, const double* prior, const double* value, double value_const, int32_t sets_only,
                    double alpha, double eps, uint64_t seed, void* stream) {
    if (int rc = mcts_check(t)) return rc;
    if (!(alpha > 0.0) || eps < 0.0 || eps > 1.0) return fail(-1, "Dirichlet alpha must be > 0 and eps in [0, 1]");
    HIP_TRY(hipSetDevice(t->device));
    const int waves = t->roots * t->wave;
    hipLaunchKernelGGL(k_mcts_expand, dim3((waves * kWave + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *t, prior, value,
                       value_const, (int)sets_only, alpha, eps, seed);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_mcts_backup(const ipp_mcts_tables* t, int32_t wave, void* stream) {
    if (int rc = mcts_check(t)) return rc;
    if (wave <= 0 || wave > t->wave) return fail(-1, "wave = %d outside [1, %d]", wave, t->wave);
    HIP_TRY(hipSetDevice(t->device));
    if (t->wave * t->max_depth <= kWave)  // one wave per root, one lane per recorded step
        hipLaunchKernelGGL(k_mcts_backup, dim3(t->roots), dim3(kWave), 0, reinterpret_cast<hipStream_t>(stream), *t, (int)wave);
    else
        hipLaunchKernelGGL(k_mcts_backup_serial, dim3((std::max(t->roots, t->max_depth) + 63) / 64), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), *t, (int)wave);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_mcts_policy(const ipp_mcts_tables* t, const double* tie_uniform, double temperature, int32_t deploy_time, double* policy,
                    int32_t* valid_idx, int32_t* ok, void* stream) {
    if (int rc = mcts_check(t)) return rc;
    if (!policy || !ok) return fail(-1, "null argument");
    if (!(temperature > 0)) return fail(-1, "temperature = %g: the read-out on the device is the one for temperature > 0 (mcts.py:133-143)", temperature);
    HIP_TRY(hipSetDevice(t->device));
    hipLaunchKernelGGL(k_mcts_policy, dim3((t->roots * kWave + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), *t, tie_uniform,
                       1.0 / temperature, (int)(deploy_time != 0), policy, valid_idx, ok);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_tree_read_diag(void* engine, int32_t node_id, float* out, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !out) return fail(-1, "null argument");
    if (node_id < 0 || node_id >= e->tv.node_cap) return fail(-1, "node_id %d outside [0, %d)", node_id, e->tv.node_cap);
    HIP_TRY(hipSetDevice(e->device));
    if (e->patch)
        hipLaunchKernelGGL(k_tree_read_diag_patch, dim3((e->v.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), e->v, e->tv, node_id, out);
    else
        hipLaunchKernelGGL(k_tree_read_diag, dim3((e->v.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), e->v, e->tv,
                           node_id, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_state_plane(void* engine, int32_t env_id, const float* mean_for_mask, uint32_t flags, float* out, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !out) return fail(-1, "null argument");
    if (int rc = check_env(e, env_id)) return rc;
    if (flags & ~IPP_ADAPTIVE) return fail(-1, "unsupported flag bits 0x%x", flags);
    if (!e->scoring) return fail(-1, "ipp_state_plane needs ipp_config.score_scratch = 1");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    const View& v = e->v;
    ScoreView sv = e->sv;
    if (v.mode == IPP_DENSE) sv.P = v.cov + (size_t)env_id * v.cov_slot;
    hipLaunchKernelGGL(k_plane_mask, dim3((v.Npad + 255) / 256), dim3(256), 0, s, v, env_id, mean_for_mask, flags, sv.mask, sv.extent);
    if (v.mode == IPP_FACTOR)
        hipLaunchKernelGGL(k_score_densify<false>, dim3(v.Npad / 64, (v.N + 63) / 64), dim3(256), 0, s, v, sv, env_id, ScorePath{}, nullptr, nullptr, 0);
    const int blocks = std::min(v.N, 1024);
    hipLaunchKernelGGL(k_plane_minmax, dim3(blocks), dim3(256), 0, s, v, sv.P, sv.mask, sv.extent);
    hipLaunchKernelGGL(k_plane_write, dim3(blocks), dim3(256), 0, s, v, sv.P, sv.mask, sv.extent, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_score_actions(void* engine, int32_t env_id, const double* actions, int32_t n, const double* prev_action,
                      uint32_t flags, float* reward, int32_t* status, void* stream) {
    return score_actions_impl(engine, env_id, ScorePath{}, nullptr, actions, n, prev_action, flags, reward, status, stream);
}

int ipp_tree_score_actions(void* engine, int32_t root_id, const int32_t* path_ids, const double* actions, int32_t n,
                           const double* prev_action, uint32_t flags, float* reward, int32_t* status, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !path_ids) return fail(-1, "null argument");
    if (e->tv.node_cap <= 0 || e->v.mode != IPP_FACTOR) return fail(-1, "ipp_tree_score_actions needs IPP_FACTOR and node_capacity > 0");
    ScorePath path{};
    for (int j = 0; j < kTreeDepth; ++j) {
        const int id = path_ids[j];
        if (id < 0) continue;
        if (id >= e->tv.node_cap) return fail(-1, "node id %d outside [0, %d)", id, e->tv.node_cap);
        path.ids[path.depth++] = id;
    }
    const float* node_diag = nullptr;
    if (path.depth) {  // the node's diagonal lives on its span only: assemble the whole one along the parent chain
        if (!e->node_diag_scratch) return fail(-1, "ipp_tree_score_actions needs ipp_config.score_scratch = 1");
        HIP_TRY(hipSetDevice(e->device));
        if (e->patch)
            hipLaunchKernelGGL(k_tree_read_diag_patch, dim3((e->v.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), e->v, e->tv,
                               path.ids[path.depth - 1], e->node_diag_scratch);
        else
            hipLaunchKernelGGL(k_tree_read_diag, dim3((e->v.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), e->v, e->tv,
                               path.ids[path.depth - 1], e->node_diag_scratch);
        node_diag = e->node_diag_scratch;
    }
    return score_actions_impl(engine, root_id, path, node_diag, actions, n, prev_action, flags, reward, status, stream);
}

int ipp_fork(void* engine, const int32_t* src_ids, const int32_t* dst_ids, int32_t n, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !src_ids || !dst_ids) return fail(-1, "null argument");
    if (n <= 0) return 0;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    const uint64_t n4 = e->v.cov_slot / 4;
    int chunks = (int)std::min<uint64_t>(512, (n4 + 1023) / 1024);
    if (chunks < 1) chunks = 1;
    hipLaunchKernelGGL(k_fork, dim3(chunks, n), dim3(256), 0, s, e->v, src_ids, dst_ids, n);
    hipLaunchKernelGGL(k_fork_rank, dim3((n + 255) / 256), dim3(256), 0, s, e->v, src_ids, dst_ids, n);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int read_row(Engine* e, const float* slab, int env, float* out, void* stream) {
    if (!e || !out) return fail(-1, "null argument");
    if (int rc = check_env(e, env)) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(out, slab + (size_t)env * e->v.Npad, (size_t)e->v.N * 4, hipMemcpyDeviceToDevice,
                           reinterpret_cast<hipStream_t>(stream)));
    return 0;
}
int ipp_read_mean(void* engine, int32_t env_id, float* out, void* stream) { Engine* e = as_engine(engine); return read_row(e, e ? e->v.mean : nullptr, env_id, out, stream); }
int ipp_read_diag(void* engine, int32_t env_id, float* out, void* stream) { Engine* e = as_engine(engine); return read_row(e, e ? e->v.diag : nullptr, env_id, out, stream); }
static int copy_gt(Engine* e, int env, float* out, const float* in, void* stream) {
    if (!e || (!out && !in)) return fail(-1, "null argument");
    if (int rc = check_env(e, env)) return rc;
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_copy_gt, dim3((e->v.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), e->v, env, out, in);
    HIP_TRY(hipGetLastError());
    return 0;
}
int ipp_read_gt(void* engine, int32_t env_id, float* out, void* stream) { return copy_gt(as_engine(engine), env_id, out, nullptr, stream); }

int ipp_read_cov_dense(void* engine, int32_t env_id, float* out, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !out) return fail(-1, "null argument");
    if (int rc = check_env(e, env_id)) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    const View& v = e->v;
    if (v.mode == IPP_DENSE) {
        HIP_TRY(hipMemcpy2DAsync(out, (size_t)v.N * 4, v.cov + (size_t)env_id * v.cov_slot, (size_t)v.Npad * 4,
                                 (size_t)v.N * 4, v.N, hipMemcpyDeviceToDevice, s));
    } else {
        if (v.patch) hipLaunchKernelGGL(k_read_cov_patch, dim3((v.N + 255) / 256, v.N), dim3(256), 0, s, v, env_id, out);
        else hipLaunchKernelGGL(k_read_cov_factor, dim3((v.N + 255) / 256, v.N), dim3(256), 0, s, v, env_id, out);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

int ipp_read_rank(void* engine, int32_t env_id, int32_t* rank, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !rank) return fail(-1, "null argument");
    if (int rc = check_env(e, env_id)) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(rank, e->v.rank + env_id, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

int ipp_read_ranks(void* engine, int32_t* out, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !out) return fail(-1, "null argument");
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(out, e->v.rank, (size_t)e->v.cap * 4, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
    return 0;
}

static int write_row(Engine* e, float* slab, int env, const float* in, void* stream) {
    if (!e || !in) return fail(-1, "null argument");
    if (int rc = check_env(e, env)) return rc;
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpyAsync(slab + (size_t)env * e->v.Npad, in, (size_t)e->v.N * 4, hipMemcpyDeviceToDevice,
                           reinterpret_cast<hipStream_t>(stream)));
    return 0;
}
int ipp_write_mean(void* engine, int32_t env_id, const float* mean, void* stream) { Engine* e = as_engine(engine); return write_row(e, e ? e->v.mean : nullptr, env_id, mean, stream); }
int ipp_write_gt(void* engine, int32_t env_id, const float* gt, void* stream) { return copy_gt(as_engine(engine), env_id, nullptr, gt, stream); }

int ipp_write_cov_dense(void* engine, int32_t env_id, const float* P, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !P) return fail(-1, "null argument");
    if (int rc = check_env(e, env_id)) return rc;
    const View& v = e->v;
    if (v.mode != IPP_DENSE) return fail(-1, "ipp_write_cov_dense needs an IPP_DENSE engine (a factor state cannot hold an arbitrary P)");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    HIP_TRY(hipMemcpy2DAsync(v.cov + (size_t)env_id * v.cov_slot, (size_t)v.Npad * 4, P, (size_t)v.N * 4, (size_t)v.N * 4,
                             v.N, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(k_dense_fixup, dim3((v.N + 255) / 256), dim3(256), 0, s, v, env_id);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_metrics(void* engine, const int32_t* env_ids, int32_t n, float* out, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !out) return fail(-1, "null argument");
    if (n <= 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_metrics, dim3(n), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), e->v, env_ids, n, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_fill_normal(void* engine, float* out, uint64_t count, uint64_t seed, uint64_t subsequence, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !out) return fail(-1, "null argument");
    if (count == 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    const uint64_t quads = (count + 3) / 4;
    hipLaunchKernelGGL(k_fill_normal, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), out, count, seed, subsequence);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_fill_normal_rows(void* engine, float* out, int32_t planes, int32_t rows, int32_t row_len, const int32_t* row_ids,
                         int64_t row_offset, uint64_t seed, uint64_t subsequence, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !out) return fail(-1, "null argument");
    if (planes < 0 || rows < 0 || row_len <= 0 || planes > 65535) return fail(-1, "bad shape %d x %d x %d", planes, rows, row_len);
    if (planes == 0 || rows == 0) return 0;
    HIP_TRY(hipSetDevice(e->device));
    const long long quads = (long long)rows * ((row_len + 3) / 4);
    hipLaunchKernelGGL(k_fill_normal_rows, dim3((unsigned)((quads + 255) / 256), (unsigned)planes), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), out, planes, rows, row_len, row_ids, (long long)row_offset, seed, subsequence);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_debug_step_item(void* engine, int32_t idx, ipp_step_item* out, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !out) return fail(-1, "null argument");
    if (idx < 0 || idx >= e->last_n) return fail(-1, "item %d outside the last step's batch of %d", idx, e->last_n);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    const int MC = e->v.meas_cap;
    ItemHdr h;
    std::vector<double> d((size_t)2 * MC * MC + 2 * MC);
    HIP_TRY(hipMemcpyAsync(&h, e->v.hdr + idx, sizeof h, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(d.data(), e->v.dbg + (size_t)idx * d.size(), d.size() * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    memset(out, 0, sizeof *out);
    out->env = h.env; out->dst = h.dst; out->rank_before = h.rank; out->status = h.status;
    out->xl = h.xl; out->xr = h.xr; out->yu = h.yu; out->yd = h.yd;
    out->rf = h.rf; out->m = h.m; out->f = h.f;
    out->cost = h.cost_d;
    out->noise_var = h.nv_d;
    for (int i = 0; i < h.m; ++i)
        for (int j = 0; j < h.m; ++j) {
            out->S[i * h.m + j] = d[(size_t)i * MC + j];
            out->Linv[i * h.m + j] = d[(size_t)MC * MC + i * MC + j];
        }
    for (int i = 0; i < h.m; ++i) {
        out->z[i] = d[(size_t)2 * MC * MC + i];
        out->y[i] = d[(size_t)2 * MC * MC + MC + i];
    }
    return 0;
}

int ipp_streamed_bytes(void* engine, uint64_t* bytes, int32_t reset, void* stream) {
    return ipp_streamed_bytes_detail(engine, bytes, nullptr, reset, stream);
}

int ipp_streamed_bytes_detail(void* engine, uint64_t* bytes, uint64_t* mask_reread_bytes, int32_t reset, void* stream) {
    Engine* e = as_engine(engine);
    if (!e || !bytes) return fail(-1, "null argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipSetDevice(e->device));
    static_assert(kCountSlots * 16 <= 1024, "counter slots");
    unsigned long long cnt[kCountSlots * 16] = {};
    HIP_TRY(hipMemcpyAsync(cnt, e->v.counters, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    uint64_t units = 0, extra = 0;
    for (int k = 0; k < kCountSlots; ++k) { units += cnt[16 * k]; extra += cnt[16 * k + 8]; }
    {  // patch kernels: per-item totals
        std::vector<unsigned long long> ic((size_t)2 * e->v.max_batch);
        HIP_TRY(hipMemcpyAsync(ic.data(), e->v.item_counts, ic.size() * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        uint64_t needed = 0;
        for (size_t i = 0; i < ic.size(); i += 2) { units += ic[i]; needed += ic[i + 1]; }
        e->last_needed_bytes = needed * 4;
    }
    *bytes = units * 4;
    if (mask_reread_bytes) *mask_reread_bytes = extra * 4;
#if IPP_PHASE_TIMING
    {
        unsigned long long c[8];
        HIP_TRY(hipMemcpy(c, e->v.counters, 64, hipMemcpyDeviceToHost));
        fprintf(stderr, "[phase timing, 10 ns ticks summed over workgroups] c1 hdr %llu c2 obs %llu c3 gather %llu c4 %llu c5 %llu c6 %llu c7 %llu\n", c[1], c[2], c[3], c[4], c[5], c[6], c[7]);
    }
#endif
#if IPP_TIMELINE
    if (const char* path = getenv("IPP_TIMELINE_FILE")) {
        const size_t n = (size_t)8 * std::min((int)e->v.max_batch, kTimelineItems);
        std::vector<unsigned long long> tl(n);
        HIP_TRY(hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_timeline), n * 8));
        if (FILE* f = fopen(path, "wb")) { fwrite(tl.data(), 8, n, f); fclose(f); }
        std::vector<unsigned long long> ut((size_t)kUnitTraceItems * 2 * 8 * 4);
        HIP_TRY(hipMemcpyFromSymbol(ut.data(), HIP_SYMBOL(g_unit_trace), ut.size() * 8));
        const std::string upath = std::string(path) + ".units";
        if (FILE* f = fopen(upath.c_str(), "wb")) { fwrite(ut.data(), 8, ut.size(), f); fclose(f); }
        std::fill(ut.begin(), ut.end(), 0ull);
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_unit_trace), ut.data(), ut.size() * 8));
    }
    {
        unsigned long long wp[16];
        HIP_TRY(hipMemcpyFromSymbol(wp, HIP_SYMBOL(g_wphase), sizeof wp));
        if (wp[10]) {
            const double u = (double)wp[10], w = (double)std::max<unsigned long long>(wp[11], 1);
            fprintf(stderr, "[k_step_patch wave phases, shader clocks] per unit: setup+compaction %.0f  prior term %.0f  stream %.0f  solve wait %.0f  "
                            "epilogue %.0f  stores %.0f | per wave: prologue %.0f  algebra/observation %.0f  results %.0f | units %.0f waves %.0f groups/unit %.2f\n",
                    wp[0] / u, wp[1] / u, wp[2] / u, wp[3] / u, wp[4] / u, wp[5] / u, wp[6] / w, wp[7] / w, wp[8] / w, u, w, wp[9] / u);
        }
        memset(wp, 0, sizeof wp);
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_wphase), wp, sizeof wp));
    }
#endif
    if (reset) {
        HIP_TRY(hipMemsetAsync(e->v.counters, 0, (size_t)kCountSlots * 128, s));
        HIP_TRY(hipMemsetAsync(e->v.item_counts, 0, (size_t)e->v.max_batch * 16, s));
    }
    return 0;
}

int ipp_probe_stream_pair(void* engine, void* stream_a, void* stream_b, int32_t launches, double* ms) {
    Engine* e = as_engine(engine);
    if (!e || !ms || launches < 1 || launches > 256) return fail(-1, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    hipStream_t a = reinterpret_cast<hipStream_t>(stream_a), b = reinterpret_cast<hipStream_t>(stream_b);
    hipEvent_t e0, ea, eb;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&ea)); HIP_TRY(hipEventCreate(&eb));
    HIP_TRY(hipStreamSynchronize(a)); HIP_TRY(hipStreamSynchronize(b));
    // warm both queues, then a chain of dependent launches on each, issued alternately like the parts of a batched step
    hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, a, 100ull, (int*)nullptr);
    hipLaunchKernelGGL(k_spin, dim3(64), dim3(64), 0, b, 100ull, (int*)nullptr);
    HIP_TRY(hipStreamSynchronize(a)); HIP_TRY(hipStreamSynchronize(b));
    HIP_TRY(hipEventRecord(e0, a));
    HIP_TRY(hipStreamWaitEvent(b, e0, 0));
    for (int i = 0; i < launches; ++i) {
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, a, 3000ull, (int*)nullptr);  // 30 us each
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(64), 0, b, 3000ull, (int*)nullptr);
    }
    HIP_TRY(hipEventRecord(ea, a)); HIP_TRY(hipEventRecord(eb, b));
    HIP_TRY(hipEventSynchronize(ea)); HIP_TRY(hipEventSynchronize(eb));
    float ta = 0.f, tb = 0.f;
    HIP_TRY(hipEventElapsedTime(&ta, e0, ea)); HIP_TRY(hipEventElapsedTime(&tb, e0, eb));
    *ms = (double)std::max(ta, tb);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    HIP_TRY(hipGetLastError());
    return 0;
}

int ipp_debug_capture(void* engine, int32_t enable) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
#if IPP_EXIT_POINTS || defined(IPP_ISSUE_TEST)
    e->v.dbg_capture = enable;  // (instruction-count build: enable = 1 + exit point, k_step_patch.h)
#else
    e->v.dbg_capture = enable != 0;
#endif
    return 0;
}

int ipp_streamed_bytes_needed(void* engine, uint64_t* bytes) {
    Engine* e = as_engine(engine);
    if (!e || !bytes) return fail(-1, "null argument");
    *bytes = e->last_needed_bytes;
    return 0;
}

int ipp_profile_enable(void* engine, int32_t enable) {
    Engine* e = as_engine(engine);
    if (!e) return fail(-1, "null engine");
    e->profile = enable != 0;
    return 0;
}

int ipp_profile_read(void* engine, int32_t kind, double* avg_ms, int64_t* launches, int32_t reset) {
    Engine* e = as_engine(engine);
    if (!e || kind < 0 || kind > 2) return fail(-1, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    ProfSlot& p = e->prof[kind];
    prof_drain(e);
    if (avg_ms) *avg_ms = p.launches ? p.total_ms / (double)p.launches : 0.0;
    if (launches) *launches = p.launches;
    if (reset) { p.total_ms = 0.0; p.launches = 0; }
    return 0;
}

int ipp_profile_read_busy(void* engine, int32_t kind, double* busy_ms, int64_t* launches, int32_t reset) {
    Engine* e = as_engine(engine);
    if (!e || kind < 0 || kind > 3) return fail(-1, "bad argument");
    HIP_TRY(hipSetDevice(e->device));
    ProfSlot& p = e->prof[kind];
    prof_drain(e);
    if (busy_ms) *busy_ms = p.busy_ms;
    if (launches) *launches = p.busy_launches;
    if (reset) { p.busy_ms = 0.0; p.busy_launches = 0; }
    return 0;
}

}  // extern "C"
