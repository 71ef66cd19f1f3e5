// ka_plan.hpp — host-side planning of a launch: no HIP call in here.
//
// A call of the batch entry point goes through four steps, each a function of its own (ka_engine.hip strings them together):
//   1. plan_forms      shapes, which lattices run tiled (KA_MODE_AUTO's cost model), the tile width, which are walked
//                      back chunk-parallel (KA_BACKTRACE_AUTO's cost model), descriptor order
//   2. carve_workspace byte offsets of everything the launch keeps in the engine's device workspace
//   3. (ka_engine.hip) descriptors + tile tasks filled in pinned memory, copied, kernels enqueued
//   4. (ka_engine.hip) ka_batch_finish: statuses, and the redo of wide lattices the scores-only forms declined
// ka_engine_workspace_bytes runs 1 + 2 only; the CPU tests reach the cost models through the ka_debug_* probes.
#pragma once
#include "ka_types.hpp"

#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

namespace ka {
namespace plan {

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

struct Shape {
    int64_t T, S, L, W;
    int32_t labx_len;
    bool fast;
    // tiled form: tiles 0 .. n_act-1 of P positions each are alive in frames [t_in, t_end)
    bool tileable = false;
    bool tiled = false;          // this call runs the lattice in the tiled form
    std::vector<int32_t> t_in, t_end;
    int32_t n_final = 0;         // tiles alive in the last frame
    uint32_t ck_mask = 1023;     // checkpoint row: position p at float index p & ck_mask
    size_t ck_pitch = 4096;      // bytes per checkpoint row
    size_t halo_bytes = 0;       // halo slots of all tile boundaries
    bool par_bt = false;         // this call walks the lattice's chunks in parallel (ka_parallel_bt.hpp)
};

// chunk-parallel backtrace: chunk maps (a byte per ring slot and chunk), super-chunk maps (two bytes), entry positions
inline int64_t chunks_of_T(int64_t T) { return (T - 1) / kCkFrames + 1; }
inline int64_t supers_of_T(int64_t T) { return (chunks_of_T(T) + kSuperChunks - 1) / kSuperChunks; }
inline size_t par_bt_bytes(const Shape &sh)
{
    const size_t R = (sh.tiled ? sh.ck_pitch : 4096) / 4;
    return align_up((size_t)chunks_of_T(sh.T) * R) + align_up((size_t)supers_of_T(sh.T) * R * 2) +
           align_up((size_t)(chunks_of_T(sh.T) + supers_of_T(sh.T)) * 4);
}

// Which frames each tile of P positions (256, or 128: ka_tiled_stream.hpp) is alive in, from the band of align.py:64-65:
//   lo(t) = max(0, floor(L t / T) - B/2),  hi(t) = min(lo(t) + B, L)
//   t_in(b)  = first t with hi(t) > P b        = 0 if P b < B, else ceil((P b - B + B/2 + 1) T / L)
//   t_end(b) = first t with lo(t) >= P (b+1)   = ceil((P (b+1) + B/2) T / L), at most T
inline void plan_tiles(Shape &sh, int32_t V, int32_t beam, int32_t max_move, int64_t P = kTpTile)
{
    sh.tileable = false;
    if (V > 64 || max_move > 4 || beam < 1 || sh.T >= (int64_t(1) << 26)) return;
    const int64_t T = sh.T, L = sh.L, B = beam, h = B / 2;
    const int64_t n_tiles = ceil_div(L, P);
    sh.t_in.clear();
    sh.t_end.clear();
    {   // (tiles a band of B positions keeps alive at once, plus those it passes through: no reallocation while they are listed)
        const size_t guess = (size_t)std::min<int64_t>(n_tiles, 16 + (B + L) / P);
        sh.t_in.reserve(guess);
        sh.t_end.reserve(guess);
    }
    sh.n_final = 0;
    sh.halo_bytes = 0;
    // ceil((x - B + h + 1) T / L) and ceil((z + h) T / L) for x = b P, z = (b + 1) P, b = 0, 1, ...: both numerators grow by P T per
    // tile, so the quotients are stepped (one division per lattice, not two per tile: a corpus launch lists 10^5 tiles)
    const int64_t step_q = (P * T) / L, step_r = (P * T) % L;
    int64_t qi = 0, ri = 0, qe, re;                       // floor((numerator + L - 1) / L) and its remainder
    bool started = false;                                 // (the t_in formula applies from the first tile with x >= B)
    {
        const int64_t e0 = (P + h) * T + L - 1;
        qe = e0 / L;
        re = e0 % L;
    }
    for (int64_t b = 0; b < n_tiles; ++b) {
        const int64_t x = b * P;
        int64_t ti = 0;
        if (x >= B) {
            if (!started) {
                const int64_t a0 = (x - B + h + 1) * T + L - 1;
                qi = a0 / L;
                ri = a0 % L;
                started = true;
            }
            ti = qi;
            qi += step_q;
            ri += step_r;
            if (ri >= L) { ri -= L; ++qi; }
        }
        const int64_t te_full = qe;
        qe += step_q;
        re += step_r;
        if (re >= L) { re -= L; ++qe; }
        if (ti >= T) break;
        const int64_t te = std::min<int64_t>(T, te_full);
        if (te <= ti) return;   // the band jumps over a whole tile in one frame (L/T > P): not worth a pipeline
        sh.t_in.push_back((int32_t)ti);
        sh.t_end.push_back((int32_t)te);
        if (te == T) ++sh.n_final;
    }
    if (sh.t_in.empty()) return;
    for (size_t b = 0; b < sh.t_in.size(); ++b)   // boundary above tile b: slots t_in(b) .. t_end(b+1) (top tile: its own t_end; nobody reads it)
        sh.halo_bytes += align_up((size_t)(sh.t_end[b + 1 < sh.t_in.size() ? b + 1 : b] - sh.t_in[b] + 1) * 16);
    // checkpoint row: every tile the band can touch at once spans < W + 512 positions; a power-of-two ring of that
    // size, or simply the whole label axis when that is not larger
    size_t ring = 1024;
    while (ring < (size_t)sh.W + 512) ring *= 2;
    const size_t whole = (size_t)ceil_div(L, kTpTile) * kTpTile;   // (whatever P: the readers' windows may reach up to the next multiple of 256)
    if (whole <= ring) {
        sh.ck_mask = 0xffffffffu;
        sh.ck_pitch = whole * 4;
    } else {
        sh.ck_mask = (uint32_t)ring - 1;
        sh.ck_pitch = ring * 4;
    }
    sh.tileable = true;
}

// How many tiles plan_tiles would list and whether the shape runs tiled at all, in O(1): the decisions of plan_forms (which
// lattices run tiled, which width) need only the counts; listing every tile of every candidate at both widths was most of the
// host's planning time (0.6 ms for a corpus launch).
//   tile b exists iff t_in(b) < T;  for b P >= B:  ceil((b P - B + h + 1) T / L) < T  <=>  b P - B + h + 1 <= floor((T - 1) L / T)
//   every tile lives at least one frame when the band plus a tile cannot be crossed in one: (P + B - 1) T >= L (else: list and see)
struct TileCount {
    bool tileable;
    int64_t n_tiles;
};
inline TileCount count_tiles(const Shape &sh, int32_t V, int32_t beam, int32_t max_move, int64_t P)
{
    if (V > 64 || max_move > 4 || beam < 1 || sh.T >= (int64_t(1) << 26)) return {false, 0};
    const int64_t T = sh.T, L = sh.L, B = beam, h = B / 2;
    if ((P + B - 1) * T < L) {
        Shape q = sh;
        plan_tiles(q, V, beam, max_move, P);
        return {q.tileable, (int64_t)q.t_in.size()};
    }
    const int64_t all = ceil_div(L, P);
    const int64_t reach = ((T - 1) * L) / T + B - h - 1;       // largest b P whose tile still begins before frame T
    const int64_t n = std::min<int64_t>(all, reach / P + 1);
    return {n > 0, n};
}

// Tile width of a launch's tiled lattices: 128 positions (two cells per lane and three wavefronts per tile, ka_tiled_stream.hpp:
// a frame of half the instructions, twice the tiles and twice the hand-offs, 46-52 KB of LDS per tile) while the tiles alive
// at once are no more than 3.2 per workgroup slot of the device, else 256.  Measured on prefixes of the corpus stand-in, all
// tiled, V = 39: three workgroups per CU (tools/sweep_width.py, profiles/r03_sweep_width.jsonl): against 256 positions the
// forward kernel takes 0.66 x the time for one chapter, 0.70 x for 64 (~580 tiles alive), 0.78 x for 128, 0.94 x for 200 (~1800),
// 1.17 x for 320 (~2900).  Tiles that never die (a band as wide as the label axis) must all hold a slot at once: the whole
// 500 000 x 100 001 lattice, 782 tiles of 128 positions on 512 slots, took 89 ms instead of 56.
// `counts`: the 128-position tile count of every tiled lattice; `W`: their band widths.  forced: 0 = by the rule, 128, 256.
inline bool narrow_tiles_pay(const std::vector<TileCount> &counts, const std::vector<int64_t> &W, int32_t n_simd, int32_t forced)
{
    if (forced == kTpTile || counts.empty()) return false;
    int64_t alive_now = 0, permanent = 0;     // tiles alive at once: of banded lattices (they come and go), of those that are all band
    for (size_t j = 0; j < counts.size(); ++j) {
        if (!counts[j].tileable) return false;        // (L/T above 128: the band jumps over a whole tile in one frame)
        const int64_t n_tiles = counts[j].n_tiles, in_band = (W[j] + 2 * kTnTile - 1) / kTnTile;
        if (n_tiles <= in_band) permanent += n_tiles;
        else alive_now += in_band;
    }
    if (forced == kTnTile) return true;
    const int64_t slots = (int64_t)(n_simd / 4) * 3;   // 46-52 KB of LDS per workgroup: three per CU
    // (round 4, ka_tiled_stream.hpp against 256 positions, profiles/r04_sweep_width.jsonl: 0.49 x for one chapter, 0.56 x for 64,
    //  0.73 x for 160, 0.93 x for 250 (~2250 tiles alive), 1.04 x for 320 (~2900): 3.2 tiles per slot; round 3's kernels: 2.6)
    return permanent <= slots && 5 * alive_now <= 16 * (slots - permanent);
}

inline bool shape_of(int64_t T, int64_t S, int32_t V, int32_t beam, int32_t max_move, Shape &sh)
{
    if (T < 1 || S < 0 || V < 1 || beam < 0 || max_move < 1 || max_move > 255) return false;
    if (T >= (int64_t(1) << 31) - 64 || S >= (int64_t(1) << 29)) return false;
    sh.T = T;
    sh.S = S;
    sh.L = 2 * S + 1;
    sh.W = std::max<int64_t>(1, std::min<int64_t>(beam, sh.L));
    sh.labx_len = (int32_t)align_up((size_t)S + 1024, 8);
    sh.fast = V <= 64 && max_move <= 4 && std::min<int64_t>(beam, sh.L) <= kFastMaxBand;
    return true;
}

// bytes of the back-pointer / checkpoint region of a lattice
inline size_t bp_region_bytes(const Shape &sh)
{
    size_t b = 0;
    if (sh.fast) b = (((size_t)sh.T + 3) / 4) * 1024;                       // exact forms: 256 B per frame (checkpoints: 128)
    else if (!sh.tiled) b = (size_t)sh.T * (size_t)sh.W;                     // generic: a byte per band cell
    if (sh.tiled) b = std::max(b, (size_t)((sh.T - 1) / kCkFrames) * sh.ck_pitch);
    return align_up(b);
}
// device bytes a lattice needs besides the caller's buffers
inline size_t lattice_ws_bytes(const Shape &sh)
{
    size_t b = align_up((size_t)sh.labx_len * 4) + bp_region_bytes(sh);
    if (!sh.fast && !sh.tiled) b += align_up((size_t)sh.L * 2 * sizeof(float) + (size_t)sh.L * 2);
    if (sh.tiled) b += sh.halo_bytes;
    if (sh.par_bt) b += par_bt_bytes(sh);
    return b;
}

// ---- KA_MODE_AUTO / KA_BACKTRACE_AUTO: which lattices of a launch run in which form --------------------------------------
// A launch is a set of lattices of very different length (the chapters of a book or of a corpus span 20k .. 160k frames).
// Each form has a CHAIN cost - the frames of a lattice are serial, so the longest lattice in a form bounds it - and a
// THROUGHPUT cost - the chip's SIMDs are shared by everything in the launch.  Costs in microseconds per frame, measured on
// MI355X (profiles/r03_sweep_auto_*.jsonl, tools/sweep_auto.py; cfg2-like stepping of the band):
//   one wavefront per lattice, forward   chain 0.224 (11.2 ms / 50000 frames alone on its SIMD), vector-ALU time 0.097 per
//                                        frame and SIMD (46 instructions x 4 cycles at 1.9 GHz: 45.8 ms for 8192 lattices)
//   tiled, forward                       chain 0.085 (two wavefronts per tile: 4.0 ms for cfg2 + the lag of the tile chain),
//                                        a tile holds one of the chip's 1024 workgroup slots for 0.12 per frame it lives
//                                        (corpus: 462 chapters, all tiled, 12.0 ms), vector-ALU time 0.023 per tile and frame
//   serial backtrace                     chain 0.173 (8.7 ms / 50000), throughput 0.0675 per frame and SIMD (27 ms for 8192)
//   chunk-parallel backtrace             0.00055 per frame of every lattice in it (it recomputes the whole band) + 0.12 ms
// The two forward kernels run side by side on two streams, and so do the two backtraces; with lattices sorted longest first
// the longest k go tiled and the longest m are walked back chunk-parallel, k and m minimising
//   forward(k)   = chain (+) throughput, chain = max(chain_tiled(T_0) x tmult, chain_wave(T_k) x wmult),
//                  throughput = max(slots(k) x tmult, alu(k)),  a (+) b = max(a, b) + min(a, b) / 2
//   backtrace(m) = max(chain_serial(T_m), parallel(m) + throughput_serial(m))
// by a scan over the sorted lengths.  The two forms are not independent: they share the SIMDs.  A tile's wavefronts run at
// the latency of their own instruction stream (55 % of it vector ALU), so w one-wavefront lattices on the same SIMD stretch
// its frames by tmult = 1 + 0.55 w (w averaged over the tile chain's duration: short lattices are gone early), and a resident
// tile stretches a one-wavefront lattice by wmult = 1 + 0.4 (sweep: 300
// chapters of 80k-160k frames, the longest 225 tiled: 29 ms against 19 ms all tiled; 2000 chapters of 20k-100k: 28 ms with
// one wavefront each, 43 ms with the longest 250 tiled).  Mixed launches pay when a few long lattices come with many short
// ones (40 of 100k-160k + 1500 of 20k-40k: 41 ms one wavefront each, 31 ms all tiled, 21.5 ms with the longest ~100 tiled).
// (Rounds 1-2 used a lattice count: <= 288 lattices tiled, sum(T) < 256 max(T) parallel, calibrated on equal lengths - 300
// long chapters and 300 short ones got the same form.)
struct AutoCosts {
    double wave_chain = 0.224, wave_alu = 0.097;
    double tile_chain = 0.095, tile_slot = 0.12, tile_alu = 0.023;
    double tile_stretch = 0.55, wave_stretch = 0.40;
    double serial_chain = 0.19, serial_thr = 0.0675;      // (0.173 - a lone walk - until round 4: beside the chunk maps of the same launch a walk takes 6.1 us per chunk, not 5.5)
    double par_frame = 0.00055, par_fixed = 120.0;      // (0.00066 until round 4's 18-cell map wavefronts: three map wavefronts per chunk became one)
    double fork = 15.0;      // a second stream and its two event waits
};
constexpr AutoCosts kAuto;

// lattices sorted longest first; alive[i] = tiles of lattice i that run at the same time.  Returns how many of the longest to tile.
inline int32_t auto_split_forward(const std::vector<int64_t> &T, const std::vector<int32_t> &alive, int32_t n_simd)
{
    const int32_t n = (int32_t)T.size();
    if (n == 0) return 0;
    const double simds = (double)n_simd;
    std::vector<double> tile_slot(n + 1, 0.0), tile_alu(n + 1, 0.0), wave_alu(n + 1, 0.0), tiles(n + 1, 0.0);
    for (int32_t i = 0; i < n; ++i) {
        tile_slot[i + 1] = tile_slot[i] + (double)T[i] * alive[i] * kAuto.tile_slot / simds;
        tile_alu[i + 1] = tile_alu[i] + (double)T[i] * alive[i] * kAuto.tile_alu / simds;
        tiles[i + 1] = tiles[i] + alive[i];
    }
    for (int32_t i = n - 1; i >= 0; --i) wave_alu[i] = wave_alu[i + 1] + (double)T[i] * kAuto.wave_alu / simds;
    std::vector<double> est(n + 1, 0.0);
    for (int32_t k = 0; k <= n; ++k) {
        // one-wavefront lattices that share the SIMDs with the tiles, averaged over the tile chain's duration (short ones are
        // gone long before the longest tiled lattice ends)
        double w = (double)(n - k) / simds;
        if (k > 0) w = std::min(w, wave_alu[k] * (kAuto.wave_chain / kAuto.wave_alu) / (kAuto.tile_chain * (double)T[0]));
        const double tmult = 1.0 + kAuto.tile_stretch * w;
        const double wmult = 1.0 + kAuto.wave_stretch * std::min(1.0, tiles[k] / simds);
        double chain = 0.0;
        if (k > 0) chain = kAuto.tile_chain * (double)T[0] * tmult;
        if (k < n) chain = std::max(chain, kAuto.wave_chain * (double)T[k] * wmult);
        const double thr = std::max(tile_slot[k] * tmult, tile_alu[k] + wave_alu[k]);
        est[k] = std::max(chain, thr) + 0.5 * std::min(chain, thr);
        if (k > 0 && k < n) est[k] += kAuto.fork;
    }
    const int32_t best_k = (int32_t)(std::min_element(est.begin(), est.end()) - est.begin());
    // one kernel form is preferred when it is within 3 % of the best mix (the model is no better than that)
    if (est[n] <= est[best_k] * 1.03) return n;
    if (est[0] <= est[best_k] * 1.03) return 0;
    return best_k;
}
// ... and how many of the longest to walk back chunk-parallel
inline int32_t auto_split_backtrace(const std::vector<int64_t> &T, int32_t n_simd)
{
    const int32_t n = (int32_t)T.size();
    if (n == 0) return 0;
    std::vector<double> par(n + 1, 0.0), ser(n + 1, 0.0);
    for (int32_t i = 0; i < n; ++i) par[i + 1] = par[i] + (double)T[i] * kAuto.par_frame;
    for (int32_t i = n - 1; i >= 0; --i) ser[i] = ser[i + 1] + (double)T[i] * kAuto.serial_thr / (double)n_simd;
    std::vector<double> est(n + 1, 0.0);
    for (int32_t m = 0; m <= n; ++m) {
        est[m] = (m > 0 ? kAuto.par_fixed + par[m] : 0.0) + ser[m];
        if (m < n) est[m] = std::max(est[m], kAuto.serial_chain * (double)T[m]);
        if (m > 0 && m < n) est[m] += kAuto.fork;
    }
    const int32_t best_m = (int32_t)(std::min_element(est.begin(), est.end()) - est.begin());
    if (est[n] <= est[best_m] * 1.05) return n;     // (one form when it is within the model's accuracy of the best mix)
    if (est[0] <= est[best_m] * 1.05) return 0;
    return best_m;
}

// ---- the plan of one launch ------------------------------------------------------------------------------------------
// what the engine's setters contribute (mode and backtrace are the KA_MODE_* / KA_BACKTRACE_* codes of the public header)
struct Knobs {
    int32_t mode = 0, backtrace = 0, n_simd = 1024;
    int32_t tile_width = 0;                  // 0 = by the rule, 128, 256
    int32_t split_tiled = -1, split_par = -1;   // ka_debug_set_split
    bool force_generic = false;              // ka_batch_finish's redo: everything through the generic kernels
};
constexpr int32_t kModeAuto = 0, kModeWave = 1, kModeWaveExact = 3, kModeTiled = 4;
constexpr int32_t kBacktraceAuto = 0, kBacktraceSerial = 1, kBacktraceParallel = 2;

struct Carve { size_t labx, bp, col, halo, map0, map1, entry, lp, lab, path, labo, sco; };

struct LaunchPlan {
    int32_t n = 0, V = 0, beam = 0, max_move = 0;
    bool host_buffers = false;     // KA_MEM_HOST: inputs and outputs are staged in the workspace too
    std::vector<Shape> sh;         // by the caller's index
    // descriptor k describes lattice order[k]: [0, n_tiled) tiled, [n_tiled, n_tiled + n_fast) one wavefront each (longest
    // first: short tail), the rest generic
    std::vector<int32_t> order;
    int32_t n_tiled = 0, n_fast = 0;
    bool narrow = false;           // the tiled lattices run in 128-position tiles
    int64_t alive_tiles = 0;       // tiles of the launch that are alive at the same time (upper estimate: band width / tile width + 2 per lattice)
    bool checkpointed_waves = true;   // the one-wavefront lattices end in backtrace_rc (not in the exact form's stored back-pointers)
    // workspace offsets (bytes)
    size_t n_tasks = 0;
    size_t off_desc = 0, off_meta = 0, off_zero = 0, zero_bytes = 0, off_prog = 0, off_aux = 0, off_ticket = 0, off_cu_rank = 0, off_tasks = 0, off_stats = 0;
    size_t off_halo = 0, ninf_bytes = 0, halo_bytes = 0, total_bytes = 0;   // halo region: the -inf slots, then every tiled lattice's boundaries
    std::vector<Carve> cv;         // by the caller's index

    int32_t n_ring() const { return n_tiled + n_fast; }   // lattices whose checkpointed results backtrace_rc walks / the exact kernels may redo
    size_t pinned_bytes() const { return align_up((size_t)n * sizeof(Lattice)) + align_up((size_t)n * 16) + align_up(n_tasks * sizeof(TileTask)); }
};

// Step 1.  Returns the caller's index of the first lattice with an unsupported shape, or -1.
inline int32_t plan_forms(LaunchPlan &p, int32_t n, const int64_t *T, const int64_t *S, int32_t V, int32_t beam_size, int32_t max_move, bool host_buffers,
                          const Knobs &kn)
{
    p = LaunchPlan();
    p.n = n;
    p.V = V;
    p.beam = beam_size;
    p.max_move = max_move;
    p.host_buffers = host_buffers;
    p.sh.resize(n);
    std::vector<Shape> &sh = p.sh;
    for (int32_t i = 0; i < n; ++i)
        if (!shape_of(T[i], S[i], V, beam_size, max_move, sh[i])) return i;
    // ---- which lattices run in the tiled form ----
    //   KA_MODE_TILED: every lattice that can;  KA_MODE_AUTO: bands too wide for the one-wavefront ring always, and of the
    //   others the longest k, k from the cost model above (split_tiled >= 0: k given, for the calibration sweeps)
    p.checkpointed_waves = kn.mode != kModeWaveExact;
    if (!kn.force_generic && (kn.mode == kModeTiled || kn.mode == kModeAuto)) {
        std::vector<int32_t> cand;      // fast-shaped lattices that could run tiled, longest first
        std::vector<TileCount> wide(n, TileCount{false, 0});
        for (int32_t i = 0; i < n; ++i) {
            if (kn.mode == kModeAuto && sh[i].fast && sh[i].T >= (int64_t(1) << 26)) continue;   // (runs in the exact form)
            wide[i] = count_tiles(sh[i], V, beam_size, max_move, kTpTile);
            if (!wide[i].tileable) continue;
            if (kn.mode == kModeTiled || !sh[i].fast) sh[i].tiled = true;
            else cand.push_back(i);
        }
        if (!cand.empty()) {
            std::stable_sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return sh[a].T > sh[b].T; });
            std::vector<int64_t> Ts(cand.size());
            std::vector<int32_t> alive(cand.size());
            for (size_t j = 0; j < cand.size(); ++j) {
                const Shape &q = sh[cand[j]];
                Ts[j] = q.T;
                alive[j] = (int32_t)std::min<int64_t>(wide[cand[j]].n_tiles, (q.W + 2 * kTpTile - 1) / kTpTile);
            }
            const int32_t k = kn.split_tiled >= 0 ? std::min<int32_t>(kn.split_tiled, (int32_t)cand.size()) : auto_split_forward(Ts, alive, kn.n_simd);
            for (int32_t j = 0; j < k; ++j) sh[cand[j]].tiled = true;
        }
        for (int32_t i = 0; i < n; ++i) p.n_tiled += sh[i].tiled ? 1 : 0;
    }
    // ---- tile width (narrow_tiles_pay above), then the tiles themselves, listed once, at that width ----
    if (p.n_tiled > 0) {
        if (kn.tile_width != kTpTile) {
            std::vector<TileCount> counts;
            std::vector<int64_t> widths;
            for (int32_t i = 0; i < n; ++i)
                if (sh[i].tiled) {
                    counts.push_back(count_tiles(sh[i], V, beam_size, max_move, kTnTile));
                    widths.push_back(sh[i].W);
                }
            p.narrow = narrow_tiles_pay(counts, widths, kn.n_simd, kn.tile_width);
        }
        const int64_t P = p.narrow ? kTnTile : kTpTile;
        for (int32_t i = 0; i < n; ++i)
            if (sh[i].tiled) {
                plan_tiles(sh[i], V, beam_size, max_move, P);
                p.alive_tiles += std::min<int64_t>((int64_t)sh[i].t_in.size(), (sh[i].W + 2 * P - 1) / P);
            }
    }
    // ---- chunk-parallel backtrace (ka_parallel_bt.hpp) for the longest of the checkpointed results: it recomputes the
    // whole band of every chunk, ~8x the serial form's work, but all chunks at once; the others are walked back serially,
    // one wavefront each, at the same time on the engine's second stream
    {
        std::vector<int32_t> ring;      // lattices whose result backtrace_rc walks, longest first
        for (int32_t i = 0; i < n; ++i)
            if (sh[i].tiled || (sh[i].fast && p.checkpointed_waves && sh[i].T < (int64_t(1) << 26))) ring.push_back(i);
        std::stable_sort(ring.begin(), ring.end(), [&](int32_t a, int32_t b) { return sh[a].T > sh[b].T; });
        int32_t m = 0;
        if (kn.backtrace == kBacktraceParallel) m = (int32_t)ring.size();
        else if (kn.backtrace == kBacktraceAuto) {
            std::vector<int64_t> Ts(ring.size());
            for (size_t j = 0; j < ring.size(); ++j) Ts[j] = sh[ring[j]].T;
            m = kn.split_par >= 0 ? std::min<int32_t>(kn.split_par, (int32_t)ring.size()) : auto_split_backtrace(Ts, kn.n_simd);
        }
        // grid limits of the chunk-parallel kernels
        int64_t chunks = 0;
        bool fits = (int64_t)n <= 65535;
        for (int32_t j = 0; j < m && fits; ++j) {
            const Shape &q = sh[ring[j]];
            chunks += chunks_of_T(q.T);
            fits = (q.W + 7 + kCmOutWide - 1) / kCmOutWide <= 65535 && supers_of_T(q.T) <= 65535 && chunks < (int64_t(1) << 31);
        }
        if (!fits) m = 0;
        for (int32_t j = 0; j < m; ++j) sh[ring[j]].par_bt = true;
    }
    // ---- descriptor order ----
    auto klass = [&](int32_t i) { return sh[i].tiled ? 0 : (sh[i].fast ? 1 : 2); };
    p.order.resize(n);
    std::iota(p.order.begin(), p.order.end(), 0);
    std::stable_sort(p.order.begin(), p.order.end(), [&](int32_t a, int32_t b) {
        if (klass(a) != klass(b)) return klass(a) < klass(b);
        return sh[a].T > sh[b].T;
    });
    for (int32_t i = 0; i < n; ++i) p.n_fast += klass(i) == 1 ? 1 : 0;
    return -1;
}

// Step 2.  Layout of the engine's device workspace for this launch.
inline void carve_workspace(LaunchPlan &p)
{
    const int32_t n = p.n;
    const std::vector<Shape> &sh = p.sh;
    size_t off = 0;
    p.off_desc = off;
    off += align_up((size_t)n * sizeof(Lattice));
    p.off_meta = off;
    off += align_up((size_t)n * 16);
    // tiled form, per launch: [progress words | per-lattice terminal records | ticket] (zeroed every launch), the tile
    // tasks, and the halo region, which starts with the -inf slots that stand in for "the tile below tile 0"
    int64_t ninf_slots = 0;
    p.n_tasks = 0;
    for (int32_t i = 0; i < n; ++i)
        if (sh[i].tiled) {
            p.n_tasks += sh[i].t_in.size();
            ninf_slots = std::max<int64_t>(ninf_slots, sh[i].t_end[0]);
        }
    p.off_zero = off;
    // ... [progress words | terminal records | ticket (16 bytes) | workgroups per CU (kCuSlots words: ka_tiled_stream.hpp)]
    p.zero_bytes = p.n_tiled ? align_up((1 + p.n_tasks) * 4 + (size_t)n * sizeof(TileAux) + 16, 16) + (size_t)kCuSlots * 4 : 0;
    p.off_prog = p.off_zero;
    p.off_aux = p.off_zero + align_up((1 + p.n_tasks) * 4, 16);
    p.off_ticket = p.off_aux + (size_t)n * sizeof(TileAux);
    p.off_cu_rank = p.off_ticket + 16;
    off += align_up(p.zero_bytes);
    p.off_tasks = off;
    off += align_up(p.n_tasks * sizeof(TileTask));
    p.off_stats = off;
    off += align_up(p.n_tasks * sizeof(TpStats));
    p.off_halo = off;
    p.ninf_bytes = p.n_tiled ? align_up((size_t)(ninf_slots + 2 * kTpBlock) * 16) : 0;
    off += p.ninf_bytes;
    p.cv.assign(n, Carve());
    // ... then the halo slots of every tiled lattice, in one piece (one fill with the sentinel per launch: ka_tiled_stream.hpp)
    p.halo_bytes = 0;
    for (int32_t i = 0; i < n; ++i) {
        p.cv[i].halo = off;
        if (sh[i].tiled) {
            off += sh[i].halo_bytes;
            p.halo_bytes += sh[i].halo_bytes;
        }
    }
    for (int32_t i = 0; i < n; ++i) {
        Carve &c = p.cv[i];
        c.labx = off;
        off += align_up((size_t)sh[i].labx_len * 4);
        c.bp = off;
        off += bp_region_bytes(sh[i]);
        c.col = off;
        if (!sh[i].fast && !sh[i].tiled) off += align_up((size_t)sh[i].L * 2 * sizeof(float) + (size_t)sh[i].L * 2);
        c.map0 = c.map1 = c.entry = off;
        if (sh[i].par_bt) {
            const size_t R = (sh[i].tiled ? sh[i].ck_pitch : 4096) / 4;
            off += align_up((size_t)chunks_of_T(sh[i].T) * R);
            c.map1 = off;
            off += align_up((size_t)supers_of_T(sh[i].T) * R * 2);
            c.entry = off;
            off += align_up((size_t)(chunks_of_T(sh[i].T) + supers_of_T(sh[i].T)) * 4);
        }
        if (p.host_buffers) {
            c.lp = off;
            off += align_up((size_t)sh[i].T * (size_t)p.V * 4);
            c.lab = off;
            off += align_up((size_t)std::max<int64_t>(sh[i].S, 1) * 4);
            c.path = off;
            off += align_up((size_t)sh[i].T * 4);
            c.labo = off;
            off += align_up((size_t)sh[i].T * 4);
            c.sco = off;
            off += align_up((size_t)sh[i].T * 4);
        }
    }
    p.total_bytes = off;
}

// Step 3a.  The tile tasks of the launch in ticket order: by first frame, so that a tile's producer (the tile below it in the same
// lattice, whose first frame is not later) holds an earlier ticket and the tiles that are needed first get the first workgroup
// slots.  Exactness of the order across lattices does not matter - a ticket that comes a little early only spins a little
// longer - so the tasks are BUCKETED by first frame (256 frames a bucket, counting sort, stable: within a lattice tile b-1 stays
// in front of tile b) instead of sorted: the host builds ~10 000 tasks per book while the GPU waits for them (round 3 sorted
// them with a comparator: 0.5 ms for the Kokoro stand-in's 6000 tasks, 1.9 ms for the corpus).
// `tasks` has room for p.n_tasks entries.
inline void fill_tile_tasks(const LaunchPlan &p, TileTask *tasks)
{
    constexpr int kBucketShift = 8;
    int32_t max_t_in = 0;
    for (int32_t k = 0; k < p.n_tiled; ++k) max_t_in = std::max(max_t_in, p.sh[p.order[k]].t_in.back());
    std::vector<uint32_t> start((size_t)(max_t_in >> kBucketShift) + 2, 0u);      // start[b + 1] = tasks in buckets <= b, then running cursors
    for (int32_t k = 0; k < p.n_tiled; ++k)
        for (int32_t ti : p.sh[p.order[k]].t_in) ++start[(size_t)(ti >> kBucketShift) + 1];
    for (size_t b = 1; b < start.size(); ++b) start[b] += start[b - 1];
    size_t word = 1;      // progress word of the lattice's tile 0 (word 0 = "nothing below"; the 256-position kernel's hand-off)
    for (int32_t k = 0; k < p.n_tiled; ++k) {
        const int32_t i = p.order[k];
        const Shape &q = p.sh[i];
        const size_t nt = q.t_in.size();
        size_t below = 0, o = p.cv[i].halo - p.off_halo;      // halo region offset of the boundary below / above the current tile
        for (size_t b = 0; b < nt; ++b) {
            TileTask &tk = tasks[start[(size_t)(q.t_in[b] >> kBucketShift)]++];
            tk.lat = k;
            tk.tile = (int32_t)b;
            tk.t_in = q.t_in[b];
            tk.t_end = q.t_end[b];
            // slot j of a boundary lies at its base + (j - t_in(lower tile)) * 16; the reader addresses from ITS t_in
            tk.halo_in = b == 0 ? 0 : (int64_t)(below + (size_t)(q.t_in[b] - q.t_in[b - 1]) * 16);
            tk.halo_out = (int64_t)o;
            tk.fill_end = b + 1 < nt ? q.t_end[b + 1] - 1 : 0;
            tk.prog_in = b == 0 ? 0 : (int32_t)(word + b - 1);
            tk.prog_out = (int32_t)(word + b);
            tk.below_end = b == 0 ? INT32_MAX : q.t_end[b - 1];
            below = o;
            o += align_up((size_t)(q.t_end[b + 1 < nt ? b + 1 : b] - q.t_in[b] + 1) * 16);
        }
        word += nt;
    }
}

// Upper bound of the workspace of a call over all modes (ka_workspace_bytes): every lattice priced in its most expensive form.
inline size_t workspace_upper_bound(int32_t n, const int64_t *T, const int64_t *S, int32_t V, int32_t beam_size, int32_t max_move)
{
    size_t total = align_up((size_t)n * sizeof(Lattice)) + align_up((size_t)n * 16);
    size_t tasks = 0;
    int64_t ninf_slots = 0;
    for (int32_t i = 0; i < n; ++i) {
        Shape sh;
        if (!shape_of(T[i], S[i], V, beam_size, max_move, sh)) return 0;
        plan_tiles(sh, V, beam_size, max_move);       // whichever form the call ends up in: the larger of the two
        sh.par_bt = sh.fast;
        size_t plain = lattice_ws_bytes(sh);
        if (sh.tileable) {
            sh.tiled = true;
            sh.par_bt = true;
            plain = std::max(plain, lattice_ws_bytes(sh));
            size_t wide_tasks = sh.t_in.size();
            ninf_slots = std::max<int64_t>(ninf_slots, sh.t_end[0]);
            plan_tiles(sh, V, beam_size, max_move, kTnTile);   // (the 128-position tiles: twice the boundaries)
            if (sh.tileable) {
                plain = std::max(plain, lattice_ws_bytes(sh));
                wide_tasks = std::max(wide_tasks, sh.t_in.size());
            }
            tasks += wide_tasks;
        }
        total += plain;
    }
    if (tasks)
        total += align_up(align_up((1 + tasks) * 4 + (size_t)n * sizeof(TileAux) + 16, 16) + (size_t)kCuSlots * 4) + align_up(tasks * sizeof(TileTask)) +
                 align_up(tasks * sizeof(TpStats)) + align_up((size_t)(ninf_slots + 2 * kTpBlock) * 16);
    return total;
}

}  // namespace plan
}  // namespace ka
