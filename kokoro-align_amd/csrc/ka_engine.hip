// ka_engine.hip — host side of the C ABI declared in include/kokoro_align_amd.h.
//
// Host code only: the kernels live in the other translation units and are reached through ka_launch.hpp; the planning of a
// launch (forms, tile plans, cost models, workspace layout) is ka_plan.hpp.  Here: the engine object, the enqueue of a
// planned launch (descriptors, copies, kernel order, the second stream of mixed launches), ka_batch_finish and the thin
// C entry points.  No torch, no oracle, no CPU fallback: if HIP fails the call fails.
#include "../../include/kokoro_align_amd.h"
#include "ka_launch.hpp"
#include "ka_plan.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

static_assert(ka::plan::kModeAuto == KA_MODE_AUTO && ka::plan::kModeWave == KA_MODE_WAVE && ka::plan::kModeWaveExact == KA_MODE_WAVE_EXACT &&
                  ka::plan::kModeTiled == KA_MODE_TILED && ka::plan::kBacktraceAuto == KA_BACKTRACE_AUTO &&
                  ka::plan::kBacktraceSerial == KA_BACKTRACE_SERIAL && ka::plan::kBacktraceParallel == KA_BACKTRACE_PARALLEL,
              "ka_plan.hpp's mode codes are the public header's");

namespace {

using ka::plan::align_up;
using ka::plan::LaunchPlan;
using ka::plan::Shape;

thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
    g_err = msg;
    return code;
}

#define KA_HIP(expr)                                                                            \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(KA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));         \
    } while (0)

// The engine works on ITS device and leaves the caller's current device (which PyTorch shares, per thread) as it
// found it, on every exit path.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t enter(int dev)
    {
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess || prev == dev) return e;
        e = hipSetDevice(dev);
        switched = e == hipSuccess;
        return e;
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
};

}  // namespace

struct ka_engine {
    int device = 0;
    char *ws = nullptr;
    size_t ws_bytes = 0;
    char *pin = nullptr;
    size_t pin_bytes = 0;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    bool profiling = false;
    bool have_times = false;
    // last enqueued batch
    int32_t n_last = 0;
    hipStream_t stream_last = nullptr;
    int32_t *h_meta = nullptr;  // pinned, 4 ints per lattice
    bool pending = false;
    int32_t mode = KA_MODE_AUTO;
    int32_t backtrace = KA_BACKTRACE_AUTO;
    int32_t n_simd = 1024;                 // SIMDs of the device
    // what ka_batch_finish needs to hand the wide tiled lattices that the scores-only form declined (non-finite log-probs) to
    // the generic kernels: the caller's buffers (valid until finish returns, by the contract of the split form)
    struct Redo { const float *lp; const int32_t *labels; int32_t *path, *lab_out; float *sc_out; int64_t T, S, ld; int32_t idx; };
    std::vector<Redo> redo;
    int32_t last_V = 0, last_beam = 0, last_max_move = 0, last_mem = KA_MEM_DEVICE;
    int32_t verify = 0;                    // ka_engine_set_verify: self-checks of the tiled form's hand-off
    int32_t rc_gather = -1;                // ka_debug_set_rc_gather: -1 the library's rule, 0 / 1 the serial backtrace's output form
    int32_t tile_width = 0;                // ka_debug_set_tile_width: 0 = the engine chooses, 128 or 256
    int32_t tile_lds = 0;                  // ka_debug_set_tile_lds: LDS bytes a tile workgroup requests (0: the library's choice)
    int32_t split_tiled = -1, split_par = -1;   // ka_debug_set_split: how many of the longest lattices run tiled / are walked back chunk-parallel (-1: cost model)
    hipStream_t aux = nullptr;             // second stream: the other kernel form of a mixed launch runs beside the first
    hipEvent_t sync[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t dbg_entry = 0, dbg_entry_n = 0, dbg_map0 = 0, dbg_map0_bytes = 0;   // last batch, descriptor 0: chunk entries and chunk maps
    size_t dbg_tasks = 0, dbg_stats = 0, dbg_n_tasks = 0;   // last batch: workspace offsets of the tile tasks and their timing records
    // Workspace bytes [clean_lo, clean_hi) hold the halo sentinel already: refilled BEHIND the last tile kernel, on the side stream,
    // while that launch's backtrace ran (refill_done marks the end of it).  A launch whose halo slots lie inside the range skips
    // its own fill - 0.15-0.2 ms for a book, in front of the first tile - and only waits for the event.
    size_t clean_lo = 0, clean_hi = 0;
    hipEvent_t refill_done = nullptr, refill_go = nullptr;
    hipStream_t fill = nullptr;            // the stream of that refill (not `aux`: a mixed launch's second backtrace runs there)
};

namespace {

int ensure_ws(ka_engine *e, size_t bytes)
{
    if (bytes <= e->ws_bytes) return KA_OK;
    KA_HIP(hipDeviceSynchronize());
    if (e->ws) KA_HIP(hipFree(e->ws));
    e->ws = nullptr;
    e->ws_bytes = 0;
    e->clean_lo = e->clean_hi = 0;
    const size_t want = align_up(bytes + bytes / 16, 1 << 20);
    hipError_t er = hipMalloc((void **)&e->ws, want);
    if (er != hipSuccess) {
        (void)hipGetLastError();
        return fail(KA_ERR_NOMEM, "hipMalloc of " + std::to_string(want) + " workspace bytes failed: " + hipGetErrorString(er));
    }
    e->ws_bytes = want;
    return KA_OK;
}

int ensure_pin(ka_engine *e, size_t bytes)
{
    if (bytes <= e->pin_bytes) return KA_OK;
    KA_HIP(hipDeviceSynchronize());
    if (e->pin) KA_HIP(hipHostFree(e->pin));
    e->pin = nullptr;
    e->pin_bytes = 0;
    const size_t want = align_up(bytes * 2, 4096);
    KA_HIP(hipHostMalloc((void **)&e->pin, want, hipHostMallocDefault));
    e->pin_bytes = want;
    return KA_OK;
}

ka::plan::Knobs knobs_of(const ka_engine *e, bool force_generic)
{
    ka::plan::Knobs kn;
    kn.mode = e->mode;
    kn.backtrace = e->backtrace;
    kn.n_simd = e->n_simd;
    kn.tile_width = e->tile_width;
    kn.split_tiled = e->split_tiled;
    kn.split_par = e->split_par;
    kn.force_generic = force_generic;
    return kn;
}

// the caller's arrays of one batch call
struct BatchArgs {
    const float *const *log_probs;
    const int64_t *T;
    const int64_t *ld;
    const int32_t *const *labels;
    const int64_t *S;
    int32_t *const *best_path;
    int32_t *const *best_labels;
    float *const *best_scores;
};

// device addresses of the launch's shared structures
struct DevicePtrs {
    ka::Lattice *lats;
    int32_t *meta;
};

// ---- step 3b: descriptors (pinned memory), in descriptor order ----
void fill_descriptors(ka_engine *e, const LaunchPlan &p, const BatchArgs &a, ka::Lattice *h_lats)
{
    using ka::plan::chunks_of_T;
    using ka::plan::supers_of_T;
    int64_t chunk_cursor = 0;
    for (int32_t k = 0; k < p.n; ++k) {
        const int32_t i = p.order[k];
        const Shape &sh = p.sh[i];
        const ka::plan::Carve &cv = p.cv[i];
        ka::Lattice &d = h_lats[k];
        std::memset(&d, 0, sizeof(d));
        // chunks of the launch's chunk-parallel lattices are numbered consecutively; a lattice that is walked back serially
        // carries the running total and owns none (lattice_of_chunk picks the LAST descriptor whose chunk0 <= chunk)
        d.chunk0 = chunk_cursor;
        d.par = sh.par_bt ? 1 : 0;
        if (sh.par_bt) chunk_cursor += chunks_of_T(sh.T);
        if (p.host_buffers) {
            d.lp = reinterpret_cast<const float *>(e->ws + cv.lp);
            d.labels = reinterpret_cast<const int32_t *>(e->ws + cv.lab);
            d.path = reinterpret_cast<int32_t *>(e->ws + cv.path);
            d.lab_out = reinterpret_cast<int32_t *>(e->ws + cv.labo);
            d.sc_out = reinterpret_cast<float *>(e->ws + cv.sco);
            d.ld = p.V;
        } else {
            d.lp = a.log_probs[i];
            d.labels = a.labels[i];
            d.path = a.best_path[i];
            d.lab_out = a.best_labels[i];
            d.sc_out = a.best_scores[i];
            d.ld = a.ld[i];
        }
        d.labx = reinterpret_cast<int32_t *>(e->ws + cv.labx);
        d.bp = e->ws + cv.bp;
        d.col = reinterpret_cast<float *>(e->ws + cv.col);
        d.T = (int32_t)sh.T;
        d.S = (int32_t)sh.S;
        d.L = (int32_t)sh.L;
        d.V = p.V;
        d.beam = p.beam;
        d.max_move = p.max_move;
        d.labx_len = sh.labx_len;
        d.W = (int32_t)sh.W;
        d.idx = i;
        d.n_final = sh.tiled ? sh.n_final : 0;
        d.ck_mask = sh.tiled ? sh.ck_mask : 1023u;
        d.ck_pitch = sh.tiled ? (int32_t)sh.ck_pitch : 4096;
        d.map0 = reinterpret_cast<uint8_t *>(e->ws + cv.map0);
        d.map1 = reinterpret_cast<uint16_t *>(e->ws + cv.map1);
        d.entry = reinterpret_cast<int32_t *>(e->ws + cv.entry);
        if (k == 0) {
            e->dbg_entry = cv.entry;
            e->dbg_entry_n = sh.par_bt ? (size_t)(chunks_of_T(sh.T) + supers_of_T(sh.T)) : 0;
            e->dbg_map0 = cv.map0;
            e->dbg_map0_bytes = sh.par_bt ? (size_t)chunks_of_T(sh.T) * ((sh.tiled ? sh.ck_pitch : 4096) / 4) : 0;
        }
        if (sh.tiled && !sh.fast) e->redo.push_back({a.log_probs[i], a.labels[i], a.best_path[i], a.best_labels[i], a.best_scores[i], a.T[i], a.S[i], a.ld[i], i});
    }
}

// A mixed launch runs its two kernel forms side by side: the second one on the engine's own stream, forked from the
// caller's stream and joined to it again (events; nothing here blocks the host).
hipError_t fork_aux(ka_engine *e, hipStream_t stream, int k)
{
    if (!e->aux) {
        hipError_t er = hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking);
        if (er != hipSuccess) return er;
    }
    hipError_t er = hipEventRecord(e->sync[k], stream);
    return er != hipSuccess ? er : hipStreamWaitEvent(e->aux, e->sync[k], 0);
}
hipError_t join_aux(ka_engine *e, hipStream_t stream, int k)
{
    hipError_t er = hipEventRecord(e->sync[k], e->aux);
    return er != hipSuccess ? er : hipStreamWaitEvent(stream, e->sync[k], 0);
}

// The halo slots of the launch start as the NaN sentinel: always for the 128-position tiles' self-vouching packets
// (ka_tiled_stream.hpp), and under ka_engine_set_verify(1) for the 256-position form (a tile that consumes a slot nobody wrote
// reports KA_ERR_INTERNAL).
bool wants_halo_sentinel(const ka_engine *e, const LaunchPlan &p) { return p.halo_bytes && ((e->verify & 1) || p.narrow); }
int fill_halo_sentinel(ka_engine *e, const LaunchPlan &p, hipStream_t stream)
{
    if (!wants_halo_sentinel(e, p)) return KA_OK;
    const size_t lo = p.off_halo + p.ninf_bytes, hi = lo + p.halo_bytes;
    if (lo >= e->clean_lo && hi <= e->clean_hi) return KA_OK;      // (the refill behind the last launch's tiles: enqueue_impl has waited for it)
    KA_HIP(hipMemsetD32Async((hipDeviceptr_t)(e->ws + lo), (int)ka::kTpSentinel, p.halo_bytes / 4, stream));
    return KA_OK;
}
// ... and behind the launch's forward pass the slots are made the sentinel again, beside the backtrace (which does not touch
// them), for the next launch with the same or a smaller halo region.
int refill_halo_sentinel(ka_engine *e, const LaunchPlan &p, hipStream_t stream)
{
    e->clean_lo = e->clean_hi = 0;
    if (!wants_halo_sentinel(e, p)) return KA_OK;
    if (!e->refill_done) KA_HIP(hipEventCreateWithFlags(&e->refill_done, hipEventDisableTiming));
    if (!e->refill_go) KA_HIP(hipEventCreateWithFlags(&e->refill_go, hipEventDisableTiming));
    if (!e->fill) KA_HIP(hipStreamCreateWithFlags(&e->fill, hipStreamNonBlocking));
    KA_HIP(hipEventRecord(e->refill_go, stream));
    KA_HIP(hipStreamWaitEvent(e->fill, e->refill_go, 0));
    const size_t lo = p.off_halo + p.ninf_bytes;
    KA_HIP(hipMemsetD32Async((hipDeviceptr_t)(e->ws + lo), (int)ka::kTpSentinel, p.halo_bytes / 4, e->fill));
    KA_HIP(hipEventRecord(e->refill_done, e->fill));
    e->clean_lo = lo;
    e->clean_hi = lo + p.halo_bytes;
    return KA_OK;
}

// ---- the tile pipeline of the launch's tiled lattices (descriptors [0, n_tiled)) ----
int enqueue_tiles(ka_engine *e, const LaunchPlan &p, const BatchArgs &a, const DevicePtrs &dv, hipStream_t stream)
{
    // 40 KB of LDS per tile workgroup = four workgroups per CU; when the launch has no more tiles than two per CU, 80 KB
    // keeps them at two per CU, i.e. (two wavefronts each) one wavefront per SIMD: two tiles whose wavefronts share a SIMD
    // run at 95-106 ns per frame instead of 55-62, and a chain runs at the pace of its slowest tile (cfg5's whole lattice,
    // 391 tiles alive for all 500 000 frames: profiles/r03_tile_stats_cfg5_full.txt)
    const unsigned lds = e->tile_lds ? (unsigned)e->tile_lds : ((int64_t)p.n_tasks <= (int64_t)e->n_simd / 2 ? 2u * ka::kTpLdsRequest : ka::kTpLdsRequest);
    ka::TileLaunch tl;
    tl.lats = dv.lats;
    tl.tasks = reinterpret_cast<const ka::TileTask *>(e->ws + p.off_tasks);
    tl.n_tasks = (int)p.n_tasks;
    tl.meta = dv.meta;
    tl.halo = e->ws + p.off_halo;
    tl.prog = reinterpret_cast<uint32_t *>(e->ws + p.off_prog);
    tl.aux = reinterpret_cast<ka::TileAux *>(e->ws + p.off_aux);
    tl.ticket = reinterpret_cast<uint32_t *>(e->ws + p.off_ticket);
    tl.verify = e->verify;
    tl.stats = reinterpret_cast<ka::TpStats *>(e->ws + p.off_stats);
    tl.cu_rank = reinterpret_cast<uint32_t *>(e->ws + p.off_cu_rank);
    tl.max_move = p.max_move;
    // staging mode: when every tiled lattice's rows are contiguous (row stride = V, V = 64 or 39, 16-byte aligned) a block
    // is copied as it lies in memory (1 KB per LDS-DMA instruction); otherwise row by row
    tl.pitch = ((p.V == 64 || p.V == 39) && p.max_move == 4) ? 4 * p.V : 0;
    for (int32_t k = 0; k < p.n_tiled && tl.pitch; ++k) {
        const int32_t i = p.order[k];
        const bool contiguous = p.host_buffers || (a.ld[i] == p.V && ((uintptr_t)a.log_probs[i] & 15) == 0);
        if (!contiguous) tl.pitch = 0;
    }
    if (p.narrow) {
        // Three of these workgroups fit a CU, and three that are alive together slow each other down: a tile puts ~28 cycles of
        // traffic per frame on the CU's LDS pipe (pairs written and read, emission gathers, packets, staging) and runs a frame in
        // 86, so the pipe saturates (the Kokoro stand-in's frames took 101 cycles at three per CU, 95 at two, 86 alone:
        // tools/tile_stats_book.py).  While the launch's tiles that are alive at once fit two per CU - with some slack: a tile
        // that waits a little for a slot costs less than sharing the pipe - ask for the LDS that keeps them at two.
        const int64_t n_cu = e->n_simd / 4;
        tl.lds = (unsigned)e->tile_lds;      // (0: what the kernel needs, 46-52 KB; the launch function takes the larger)
        if (!tl.lds && p.alive_tiles <= 11 * n_cu / 4) tl.lds = 64 * 1024;
        ka::launch_forward_tiled128(tl, stream);
    } else {
        // ... and when the tiles alive at once outnumber four per CU, no more than the kernel uses: with V = 39 and contiguous rows
        // that is 27 KB, FIVE workgroups per CU - a slot-bound launch (the corpus) wants slots more than it wants fast tiles
        tl.lds = (!e->tile_lds && p.alive_tiles > (int64_t)e->n_simd) ? 0u : lds;
        ka::launch_forward_tiled256(tl, stream);
    }
    return KA_OK;
}

// ---- forward pass: [0, n_tiled) tiled, [n_tiled, n_ring) one wavefront each, the rest generic.  Returns (through
// `wave_form`) which form the one-wavefront lattices ended up in. ----
int enqueue_forward(ka_engine *e, const LaunchPlan &p, const BatchArgs &a, const DevicePtrs &dv, hipStream_t stream, ka::WaveForm *wave_form)
{
    const int32_t n_tiled = p.n_tiled, n_fast = p.n_fast, n_ring = p.n_ring();
    const bool two_forward = n_tiled > 0 && n_fast > 0;
    if (two_forward) KA_HIP(fork_aux(e, stream, 0));      // (before the tile kernel is enqueued: the second stream must not wait for it)
    ka::WaveForm form = n_tiled > 0 ? ka::kWaveCheckpointed : ka::kWaveExact;
    if (n_tiled > 0) {
        const int rc = enqueue_tiles(e, p, a, dv, stream);
        if (rc != KA_OK) return rc;
    }
    if (n_fast > 0) {
        form = p.checkpointed_waves ? ka::kWaveCheckpointed : ka::kWaveExact;
        // backtrace_rc_kernel keeps 34*T in 32 bits (descriptors are sorted longest first)
        if (form == ka::kWaveCheckpointed && p.sh[p.order[n_tiled]].T >= (int64_t(1) << 26)) form = ka::kWaveExact;
        ka::launch_forward_wave(p.max_move, dv.lats + n_tiled, n_fast, dv.meta, two_forward ? e->aux : stream, form);
    }
    // tiled lattices that the scores-only form declined (non-finite log-probs) and that fit the one-wavefront ring are
    // redone by the exact kernels (kFlagExact; wider ones get kFlagDeclined and are handed to the generic kernels by
    // ka_batch_finish)
    if (n_tiled > 0) ka::launch_forward_flagged(p.max_move, dv.lats, n_tiled, dv.meta, stream);
    if (two_forward) KA_HIP(join_aux(e, stream, 1));
    if (p.n > n_ring) ka::launch_forward_generic(dv.lats + n_ring, p.n - n_ring, dv.meta, stream);
    *wave_form = form;
    return KA_OK;
}

// ---- backtrace: checkpointed results (tiled + checkpointed one-wavefront form) by recomputation, stored back-pointers by
// the walk.  Returns (through `rc_hi`) the end of the descriptor range [0, rc_hi) whose outputs backtrace_rc writes itself. ----
int enqueue_backtrace(ka_engine *e, const LaunchPlan &p, const DevicePtrs &dv, hipStream_t stream, ka::WaveForm wave_form, int32_t *rc_hi_out)
{
    using ka::plan::chunks_of_T;
    using ka::plan::supers_of_T;
    const int32_t n_ring = p.n_ring();
    const int32_t rc_hi = p.n_tiled + (wave_form == ka::kWaveCheckpointed ? p.n_fast : 0);
    if (rc_hi > 0) {
        // the chunk-parallel lattices of the range (Lattice::par; their chunks are numbered consecutively) and the others
        int64_t total_chunks = 0, max_seg = 1, max_sup = 1, max_w = 1;
        int32_t n_par = 0;
        for (int32_t k = 0; k < rc_hi; ++k) {
            const Shape &q = p.sh[p.order[k]];
            if (!q.par_bt) continue;
            ++n_par;
            total_chunks += chunks_of_T(q.T);
            max_sup = std::max<int64_t>(max_sup, supers_of_T(q.T));
            max_w = std::max<int64_t>(max_w, q.W);
        }
        // segments of the band a map wavefront delivers: 408 positions (8 cells per lane) when every band fits one such
        // wavefront, else 1048 (18 cells: the reference's band of 1000 in one wavefront instead of three)
        max_seg = (max_w + 7 + ka::cm_out_for(max_w) - 1) / ka::cm_out_for(max_w);
        const bool two_backtraces = n_par > 0 && n_par < rc_hi;
        if (n_par < rc_hi) {      // one wavefront per lattice, chunk after chunk (skips the chunk-parallel ones)
            if (two_backtraces) KA_HIP(fork_aux(e, stream, 2));
            // (the gather form is opt-in: 27.2 -> 26.1 ms for 8192 lattices alone on the GPU, nothing with four launches in flight,
            //  and 47 GB more HBM traffic per step by the counters - DESIGN.md section 8)
            ka::launch_backtrace_rc_serial(p.max_move, dv.lats, rc_hi, dv.meta, two_backtraces ? e->aux : stream, e->rc_gather == 1);
        }
        if (n_par > 0) {
            ka::launch_chunk_entries(p.max_move, dv.lats, rc_hi, dv.meta, stream, (unsigned)total_chunks, (unsigned)max_seg, (unsigned)max_sup, (unsigned)max_w);
            ka::launch_backtrace_rc_chunks(p.max_move, dv.lats, rc_hi, dv.meta, stream, (unsigned)total_chunks);
        }
        if (two_backtraces) KA_HIP(join_aux(e, stream, 3));
        ka::launch_backtrace_w16(dv.lats, rc_hi, dv.meta, stream, 1);      // only what the exact kernels redid
    }
    if (n_ring > rc_hi) ka::launch_backtrace_w16(dv.lats + rc_hi, n_ring - rc_hi, dv.meta, stream, 0);
    if (p.n > n_ring) ka::launch_backtrace_generic(dv.lats + n_ring, p.n - n_ring, dv.meta, stream);
    *rc_hi_out = rc_hi;
    return KA_OK;
}

// ---- labels and scores of the lattices whose backtrace wrote the path only ----
void enqueue_output_gathers(const LaunchPlan &p, const DevicePtrs &dv, hipStream_t stream, int32_t rc_hi)
{
    int64_t t_max = 1;
    for (int32_t i = 0; i < p.n; ++i) t_max = std::max<int64_t>(t_max, p.sh[i].T);
    const unsigned gx = (unsigned)((t_max + 1023) / 1024);
    for (int32_t y0 = 0; y0 < rc_hi; y0 += 65535)     // only what the exact kernels redid
        ka::launch_gather_outputs(dv.lats + y0, 1, (unsigned)std::min<int32_t>(65535, rc_hi - y0), dv.meta, stream, 1);
    for (int32_t y0 = rc_hi; y0 < p.n; y0 += 65535)   // grid.y limit
        ka::launch_gather_outputs(dv.lats + y0, gx, (unsigned)std::min<int32_t>(65535, p.n - y0), dv.meta, stream, 0);
}

// Plans and enqueues one batch on `stream`.  plan_only_bytes: ka_engine_workspace_bytes - sizes only, nothing is launched
// and the engine is left untouched.
int enqueue_impl(ka_engine *e, int32_t n, const BatchArgs &a, int32_t V, int32_t beam_size, int32_t max_move, int32_t mem, hipStream_t stream,
                 bool force_generic, size_t *plan_only_bytes)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    const bool plan_only = plan_only_bytes != nullptr;
    if (!plan_only && e->pending) return fail(KA_ERR_BAD_ARGS, "a batch is already enqueued: call ka_batch_finish first");
    if (n < 0 || (n > 0 && (!a.T || !a.S || (!plan_only && (!a.log_probs || !a.ld || !a.labels || !a.best_path || !a.best_labels || !a.best_scores)))))
        return fail(KA_ERR_BAD_ARGS, "batch: NULL array argument");
    if (mem != KA_MEM_HOST && mem != KA_MEM_DEVICE) return fail(KA_ERR_BAD_ARGS, "mem must be KA_MEM_HOST or KA_MEM_DEVICE");
    DeviceGuard guard;
    if (!plan_only) {
        KA_HIP(guard.enter(e->device));
        e->n_last = n;
        e->stream_last = stream;
        e->have_times = false;
        e->redo.clear();
        e->last_V = V;
        e->last_beam = beam_size;
        e->last_max_move = max_move;
        e->last_mem = mem;
    }
    if (n == 0) {
        if (plan_only) *plan_only_bytes = 0;
        else e->pending = true;
        return KA_OK;
    }

    // ---- steps 1 and 2 (ka_plan.hpp) ----
    LaunchPlan p;
    const int32_t bad = ka::plan::plan_forms(p, n, a.T, a.S, V, beam_size, max_move, mem == KA_MEM_HOST, knobs_of(e, force_generic));
    if (bad >= 0) return fail(KA_ERR_BAD_ARGS, "lattice " + std::to_string(bad) + ": unsupported T/S/V/ld/beam_size/max_move");
    ka::plan::carve_workspace(p);
    if (plan_only) {
        *plan_only_bytes = p.total_bytes;
        return KA_OK;
    }
    for (int32_t i = 0; i < n; ++i) {
        if (a.ld[i] < V) return fail(KA_ERR_BAD_ARGS, "lattice " + std::to_string(i) + ": unsupported T/S/V/ld/beam_size/max_move");
        if (!a.log_probs[i] || !a.best_path[i] || !a.best_labels[i] || !a.best_scores[i] || (a.S[i] > 0 && !a.labels[i]))
            return fail(KA_ERR_BAD_ARGS, "lattice " + std::to_string(i) + ": NULL buffer");
    }
    int rc = ensure_ws(e, p.total_bytes);
    if (rc != KA_OK) return rc;
    rc = ensure_pin(e, p.pinned_bytes());
    if (rc != KA_OK) return rc;
    // The refill behind the LAST launch's tiles runs on a stream of its own and nothing has waited for it yet: this launch
    // lays its regions out from the start of the same workspace, and its first copies and fills must not race a fill that is
    // still writing sentinels there (descriptors overwritten by a late refill were a memory fault with four engines running
    // launches of different shapes: tools/stress_streams_tiled.py).
    if (e->refill_done) KA_HIP(hipStreamWaitEvent(stream, e->refill_done, 0));
    // (... and a launch without the sentinel protocol - other kernel forms, the generic redo of ka_batch_finish - writes over
    //  what that refill left clean)
    if (!wants_halo_sentinel(e, p)) e->clean_lo = e->clean_hi = 0;
    e->dbg_tasks = p.off_tasks;
    e->dbg_stats = p.off_stats;
    e->dbg_n_tasks = p.n_tasks;

    // ---- step 3: descriptors and tile tasks in pinned memory, copies in, kernels, copies out ----
    ka::Lattice *h_lats = reinterpret_cast<ka::Lattice *>(e->pin);
    e->h_meta = reinterpret_cast<int32_t *>(e->pin + align_up((size_t)n * sizeof(ka::Lattice)));
    ka::TileTask *h_tasks = reinterpret_cast<ka::TileTask *>(e->pin + align_up((size_t)n * sizeof(ka::Lattice)) + align_up((size_t)n * 16));
    fill_descriptors(e, p, a, h_lats);
    if (mem == KA_MEM_HOST)
        for (int32_t i = 0; i < n; ++i) {
            KA_HIP(hipMemcpy2DAsync(e->ws + p.cv[i].lp, (size_t)V * 4, a.log_probs[i], (size_t)a.ld[i] * 4, (size_t)V * 4, (size_t)p.sh[i].T,
                                    hipMemcpyHostToDevice, stream));
            if (p.sh[i].S > 0) KA_HIP(hipMemcpyAsync(e->ws + p.cv[i].lab, a.labels[i], (size_t)p.sh[i].S * 4, hipMemcpyHostToDevice, stream));
        }
    DevicePtrs dv;
    dv.lats = reinterpret_cast<ka::Lattice *>(e->ws + p.off_desc);
    dv.meta = reinterpret_cast<int32_t *>(e->ws + p.off_meta);
    KA_HIP(hipMemcpyAsync(dv.lats, h_lats, (size_t)n * sizeof(ka::Lattice), hipMemcpyHostToDevice, stream));
    KA_HIP(hipMemsetAsync(dv.meta, 0, (size_t)n * 16, stream));
    // the device starts on what needs no tile tasks - the fills of the tile pipeline's regions, the label preparation - while
    // the host lists the tasks (a book has ~10 000)
    if (p.n_tiled) {
        KA_HIP(hipMemsetAsync(e->ws + p.off_zero, 0, p.zero_bytes, stream));
        KA_HIP(hipMemsetD32Async((hipDeviceptr_t)(e->ws + p.off_prog), (int)ka::kTpProgDone, 1, stream));
        KA_HIP(hipMemsetD32Async((hipDeviceptr_t)(e->ws + p.off_halo), (int)0xff800000u, p.ninf_bytes / 4, stream));   // -inf packets
        const int rcf = fill_halo_sentinel(e, p, stream);
        if (rcf != KA_OK) return rcf;
    }
    if (p.n_tiled) {
        // (the copy goes in FRONT of the label preparation, not between it and the tile kernel: with a copy right before them the
        //  whole 500 000 x 100 001 lattice's 391 permanent tiles were placed differently and ran 69 ms instead of 52)
        ka::plan::fill_tile_tasks(p, h_tasks);
        KA_HIP(hipMemcpyAsync(e->ws + p.off_tasks, h_tasks, p.n_tasks * sizeof(ka::TileTask), hipMemcpyHostToDevice, stream));
    }
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[0], stream));
    ka::launch_prep_labels(dv.lats, n, dv.meta, stream);
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[1], stream));
    ka::WaveForm wave_form = ka::kWaveExact;
    rc = enqueue_forward(e, p, a, dv, stream, &wave_form);
    if (rc != KA_OK) return rc;
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[2], stream));
    if (p.n_tiled) {
        rc = refill_halo_sentinel(e, p, stream);
        if (rc != KA_OK) return rc;
    }
    int32_t rc_hi = 0;
    rc = enqueue_backtrace(e, p, dv, stream, wave_form, &rc_hi);
    if (rc != KA_OK) return rc;
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[3], stream));
    enqueue_output_gathers(p, dv, stream, rc_hi);
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[4], stream));
    KA_HIP(hipGetLastError());

    KA_HIP(hipMemcpyAsync(e->h_meta, dv.meta, (size_t)n * 16, hipMemcpyDeviceToHost, stream));
    if (mem == KA_MEM_HOST)
        for (int32_t i = 0; i < n; ++i) {
            const size_t b = (size_t)p.sh[i].T * 4;
            KA_HIP(hipMemcpyAsync(a.best_path[i], e->ws + p.cv[i].path, b, hipMemcpyDeviceToHost, stream));
            KA_HIP(hipMemcpyAsync(a.best_labels[i], e->ws + p.cv[i].labo, b, hipMemcpyDeviceToHost, stream));
            KA_HIP(hipMemcpyAsync(a.best_scores[i], e->ws + p.cv[i].sco, b, hipMemcpyDeviceToHost, stream));
        }
    e->pending = true;
    return KA_OK;
}

// ka_batch_finish, second part: lattices in the tiled form whose band is wider than the exact kernels' ring and whose
// log-probs are not all finite (flag set by the forward kernel) have no result yet - the scores-only forms are valid only
// while "live" and "score > -inf" coincide.  The reference answers such input (align.py:67-85 tracks the live set
// explicitly), so they are handed to the generic kernels now, into the caller's buffers.  `meta` is the finished batch's
// host copy; a redo that cannot run marks ITS lattices with the error and leaves the others' results standing.
void redo_declined(ka_engine *e, const std::vector<ka_engine::Redo> &again, std::vector<int32_t> &meta)
{
    const int32_t m = (int32_t)again.size();
    std::vector<const float *> lp(m);
    std::vector<const int32_t *> lab(m);
    std::vector<int32_t *> path(m), lab_out(m);
    std::vector<float *> sc(m);
    std::vector<int64_t> T(m), S(m), ld(m);
    for (int32_t j = 0; j < m; ++j) {
        lp[j] = again[j].lp; lab[j] = again[j].labels; path[j] = again[j].path; lab_out[j] = again[j].lab_out; sc[j] = again[j].sc_out;
        T[j] = again[j].T; S[j] = again[j].S; ld[j] = again[j].ld;
    }
    const BatchArgs a{lp.data(), T.data(), ld.data(), lab.data(), S.data(), path.data(), lab_out.data(), sc.data()};
    // the nested enqueue must not disturb what belongs to the batch itself: its size, its event times, its profiling flag
    const int32_t n_all = e->n_last;
    const bool prof = e->profiling, have_times = e->have_times;
    const hipStream_t stream = e->stream_last;
    e->profiling = false;
    int rc = enqueue_impl(e, m, a, e->last_V, e->last_beam, e->last_max_move, e->last_mem, stream, /*force_generic=*/true, nullptr);
    if (rc == KA_OK && hipStreamSynchronize(stream) != hipSuccess) rc = fail(KA_ERR_HIP, "hipStreamSynchronize after the redo of wide lattices with non-finite log-probs failed");
    e->profiling = prof;
    e->have_times = have_times;
    e->pending = false;
    e->n_last = n_all;
    e->redo.clear();
    for (int32_t j = 0; j < m; ++j) {
        int32_t *dst = meta.data() + 4 * (size_t)again[j].idx;
        if (rc == KA_OK) std::memcpy(dst, e->h_meta + 4 * (size_t)j, 16);
        else dst[0] = rc;       // (KA_ERR_NOMEM for the byte-per-cell workspace of a very wide lattice, most likely)
    }
}

}  // namespace

extern "C" {

int32_t ka_version(void) { return KA_VERSION; }

const char *ka_last_error(void) { return g_err.c_str(); }

int ka_engine_create(int32_t device, ka_engine **out)
{
    if (!out) return fail(KA_ERR_BAD_ARGS, "ka_engine_create: out is NULL");
    int ndev = 0;
    KA_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev)
        return fail(KA_ERR_BAD_ARGS, "ka_engine_create: device " + std::to_string(device) + " of " + std::to_string(ndev));
    DeviceGuard guard;
    KA_HIP(guard.enter(device));
    ka_engine *e = new ka_engine();
    e->device = device;
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) e->n_simd = 4 * prop.multiProcessorCount;
    }
    for (int i = 0; i < 5; ++i) {
        hipError_t er = hipEventCreate(&e->ev[i]);
        if (er != hipSuccess) {
            delete e;
            return fail(KA_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(er));
        }
    }
    for (int i = 0; i < 4; ++i) {
        hipError_t er = hipEventCreateWithFlags(&e->sync[i], hipEventDisableTiming);
        if (er != hipSuccess) {
            ka_engine_destroy(e);
            return fail(KA_ERR_HIP, std::string("hipEventCreateWithFlags: ") + hipGetErrorString(er));
        }
    }
    *out = e;
    return KA_OK;
}

void ka_engine_destroy(ka_engine *e)
{
    if (!e) return;
    DeviceGuard guard;
    (void)guard.enter(e->device);
    (void)hipDeviceSynchronize();
    if (e->ws) (void)hipFree(e->ws);
    if (e->pin) (void)hipHostFree(e->pin);
    for (int i = 0; i < 5; ++i)
        if (e->ev[i]) (void)hipEventDestroy(e->ev[i]);
    for (int i = 0; i < 4; ++i)
        if (e->sync[i]) (void)hipEventDestroy(e->sync[i]);
    if (e->aux) (void)hipStreamDestroy(e->aux);
    if (e->refill_done) (void)hipEventDestroy(e->refill_done);
    if (e->refill_go) (void)hipEventDestroy(e->refill_go);
    if (e->fill) (void)hipStreamDestroy(e->fill);
    delete e;
}

int ka_stream_create(int32_t device, void **stream)
{
    if (!stream) return fail(KA_ERR_BAD_ARGS, "ka_stream_create: stream is NULL");
    DeviceGuard guard;
    KA_HIP(guard.enter(device));
    hipStream_t s = nullptr;
    KA_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return KA_OK;
}

int ka_stream_destroy(int32_t device, void *stream)
{
    if (!stream) return KA_OK;
    DeviceGuard guard;
    KA_HIP(guard.enter(device));
    KA_HIP(hipStreamSynchronize((hipStream_t)stream));
    KA_HIP(hipStreamDestroy((hipStream_t)stream));
    return KA_OK;
}

int ka_engine_reserve(ka_engine *e, size_t workspace_bytes)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    DeviceGuard guard;
    KA_HIP(guard.enter(e->device));
    return ensure_ws(e, workspace_bytes);
}

size_t ka_workspace_bytes(int32_t n, const int64_t *T, const int64_t *S, int32_t V, int32_t beam_size, int32_t max_move)
{
    if (n < 0 || !T || !S) return 0;
    return ka::plan::workspace_upper_bound(n, T, S, V, beam_size, max_move);
}

// Must be called from the engine's own host thread (like every other call on an engine); it reads the engine's settings and
// changes nothing.
size_t ka_engine_workspace_bytes(ka_engine *e, int32_t n, const int64_t *T, const int64_t *S, int32_t V, int32_t beam_size, int32_t max_move,
                                 int32_t mem)
{
    if (!e) return 0;
    size_t bytes = 0;
    const BatchArgs a{nullptr, T, nullptr, nullptr, S, nullptr, nullptr, nullptr};
    const int rc = enqueue_impl(e, n, a, V, beam_size, max_move, mem, nullptr, false, &bytes);
    return rc == KA_OK ? bytes : 0;
}

int ka_engine_set_mode(ka_engine *e, int32_t mode)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (mode != KA_MODE_AUTO && mode != KA_MODE_WAVE && mode != KA_MODE_WAVE_EXACT && mode != KA_MODE_TILED)
        return fail(KA_ERR_BAD_ARGS, "ka_engine_set_mode: unknown mode");
    e->mode = mode;
    return KA_OK;
}

int ka_engine_set_backtrace(ka_engine *e, int32_t how)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (how != KA_BACKTRACE_AUTO && how != KA_BACKTRACE_SERIAL && how != KA_BACKTRACE_PARALLEL)
        return fail(KA_ERR_BAD_ARGS, "ka_engine_set_backtrace: unknown value");
    e->backtrace = how;
    return KA_OK;
}

int ka_engine_set_verify(ka_engine *e, int32_t flags)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (flags < 0 || flags > 7) return fail(KA_ERR_BAD_ARGS, "ka_engine_set_verify: flags are a combination of 1, 2 and 4");
    e->verify = flags;
    return KA_OK;
}

int ka_debug_set_rc_gather(ka_engine *e, int32_t how)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    e->rc_gather = how < 0 ? -1 : (how ? 1 : 0);
    return KA_OK;
}

int ka_debug_set_tile_lds(ka_engine *e, int32_t bytes)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (bytes < 0 || bytes > 160 * 1024) return fail(KA_ERR_BAD_ARGS, "ka_debug_set_tile_lds: 0 .. 160 KB");
    e->tile_lds = bytes;
    return KA_OK;
}

int ka_debug_set_split(ka_engine *e, int32_t n_tiled, int32_t n_parallel)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    e->split_tiled = n_tiled < 0 ? -1 : n_tiled;
    e->split_par = n_parallel < 0 ? -1 : n_parallel;
    return KA_OK;
}

int ka_debug_set_tile_width(ka_engine *e, int32_t positions)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (positions != 0 && positions != ka::kTnTile && positions != ka::kTpTile) return fail(KA_ERR_BAD_ARGS, "ka_debug_set_tile_width: 0, 128 or 256");
    e->tile_width = positions;
    return KA_OK;
}

int ka_engine_set_profiling(ka_engine *e, int32_t on)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    e->profiling = on != 0;
    e->have_times = false;
    return KA_OK;
}

int ka_engine_last_kernel_ms(ka_engine *e, float ms[4])
{
    if (!e || !ms) return fail(KA_ERR_BAD_ARGS, "engine or ms is NULL");
    if (!e->have_times) return fail(KA_ERR_BAD_ARGS, "no profiled batch has been finished");
    for (int i = 0; i < 4; ++i) KA_HIP(hipEventElapsedTime(&ms[i], e->ev[i], e->ev[i + 1]));
    return KA_OK;
}

int ka_ctc_best_path_batch_enqueue_f32(ka_engine *e, int32_t n, const float *const *log_probs, const int64_t *T,
                                       int32_t V, const int64_t *ld, const int32_t *const *labels, const int64_t *S,
                                       int32_t beam_size, int32_t max_move, int32_t *const *best_path,
                                       int32_t *const *best_labels, float *const *best_scores, void *stream)
{
    const BatchArgs a{log_probs, T, ld, labels, S, best_path, best_labels, best_scores};
    return enqueue_impl(e, n, a, V, beam_size, max_move, KA_MEM_DEVICE, (hipStream_t)stream, false, nullptr);
}

int ka_batch_finish(ka_engine *e, float *total_score, int32_t *status)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (!e->pending) return fail(KA_ERR_BAD_ARGS, "no batch enqueued");
    e->pending = false;
    DeviceGuard guard;
    KA_HIP(guard.enter(e->device));
    KA_HIP(hipStreamSynchronize(e->stream_last));
    if (e->profiling && e->n_last > 0) e->have_times = true;
    const int32_t n_all = e->n_last;
    std::vector<int32_t> meta(e->h_meta, e->h_meta + 4 * (size_t)n_all);
    // wide tiled lattices the scores-only form declined: KA_MODE_AUTO redoes them through the generic kernels, an explicit
    // KA_MODE_TILED reports KA_ERR_NONFINITE
    std::vector<ka_engine::Redo> again;
    for (const ka_engine::Redo &r : e->redo) {
        int32_t *m = meta.data() + 4 * (size_t)r.idx;
        if (m[0] != KA_OK || !(m[2] & ka::kFlagDeclined)) continue;
        if (e->mode == KA_MODE_TILED) m[0] = KA_ERR_NONFINITE;
        else again.push_back(r);
    }
    std::string redo_error;
    if (!again.empty()) {
        redo_declined(e, again, meta);
        redo_error = g_err;
    }
    int first_bad = KA_OK;
    for (int32_t i = 0; i < n_all; ++i) {
        const int32_t *m = meta.data() + 4 * (size_t)i;
        if (status) status[i] = m[0];
        if (total_score) std::memcpy(&total_score[i], &m[3], 4);
        if (m[0] != KA_OK && first_bad == KA_OK) {
            first_bad = m[0];
            g_err = "lattice " + std::to_string(i) + (m[0] == KA_ERR_EMPTY_BEAM ? ": no live state in the last frame (empty beam)"
                                                      : m[0] == KA_ERR_BAD_LABEL ? ": label outside [0, V)"
                                                      : m[0] == KA_ERR_INTERNAL  ? ": internal error in the tile hand-off"
                                                      : m[0] == KA_ERR_NAN       ? ": a log-prob is NaN"
                                                      : m[0] == KA_ERR_NONFINITE ? ": log-probs with infinities in a band wider than 1009 positions (KA_MODE_TILED cannot answer it: use KA_MODE_AUTO)"
                                                      : !redo_error.empty()      ? ": the redo through the generic kernels failed: " + redo_error
                                                                                 : ": failed");
        }
    }
    return first_bad;
}

int ka_ctc_best_path_batch_f32(ka_engine *e, int32_t n, const float *const *log_probs, const int64_t *T, int32_t V,
                               const int64_t *ld, const int32_t *const *labels, const int64_t *S, int32_t beam_size,
                               int32_t max_move, int32_t *const *best_path, int32_t *const *best_labels,
                               float *const *best_scores, float *total_score, int32_t *status, int32_t mem,
                               void *stream)
{
    const BatchArgs a{log_probs, T, ld, labels, S, best_path, best_labels, best_scores};
    int rc = enqueue_impl(e, n, a, V, beam_size, max_move, mem, (hipStream_t)stream, false, nullptr);
    if (rc != KA_OK) return rc;
    return ka_batch_finish(e, total_score, status);
}

int ka_ctc_best_path_f32(ka_engine *e, const float *log_probs, int64_t T, int32_t V, int64_t ld,
                         const int32_t *labels, int64_t S, int32_t beam_size, int32_t max_move, int32_t *best_path,
                         int32_t *best_labels, float *best_scores, float *total_score, int32_t mem, void *stream)
{
    int32_t status = 0;
    float total = 0.0f;
    int rc = ka_ctc_best_path_batch_f32(e, 1, &log_probs, &T, V, &ld, &labels, &S, beam_size, max_move, &best_path,
                                        &best_labels, &best_scores, &total, &status, mem, stream);
    if (total_score) *total_score = total;
    return rc;
}

int ka_debug_chunk_entries(ka_engine *e, int32_t *out, int32_t max_entries, uint8_t *map0_out, int64_t map0_max)
{
    if (!e || !out || max_entries < 0) return fail(KA_ERR_BAD_ARGS, "ka_debug_chunk_entries: bad arguments");
    DeviceGuard guard;
    KA_HIP(guard.enter(e->device));
    const size_t n = std::min<size_t>(e->dbg_entry_n, (size_t)max_entries);
    if (n) KA_HIP(hipMemcpy(out, e->ws + e->dbg_entry, n * 4, hipMemcpyDeviceToHost));
    if (map0_out && map0_max > 0 && e->dbg_map0_bytes)
        KA_HIP(hipMemcpy(map0_out, e->ws + e->dbg_map0, std::min<size_t>(e->dbg_map0_bytes, (size_t)map0_max), hipMemcpyDeviceToHost));
    return (int)n;
}

int ka_debug_plan_tiles_width(int64_t T, int64_t S, int32_t V, int32_t beam_size, int32_t max_move, int32_t positions, int32_t *t_in, int32_t *t_end,
                              int32_t max_tiles, int64_t *checkpoint_pitch)
{
    Shape sh;
    if (max_tiles < 0 || (max_tiles > 0 && (!t_in || !t_end)) || (positions != ka::kTnTile && positions != ka::kTpTile) ||
        !ka::plan::shape_of(T, S, V, beam_size, max_move, sh))
        return fail(KA_ERR_BAD_ARGS, "ka_debug_plan_tiles_width: bad arguments");
    ka::plan::plan_tiles(sh, V, beam_size, max_move, positions);
    {   // the O(1) count the planner decides by must agree with the listing (the CPU tests of the tile plan come through here)
        const ka::plan::TileCount tc = ka::plan::count_tiles(sh, V, beam_size, max_move, positions);
        if (tc.tileable != sh.tileable || (sh.tileable && tc.n_tiles != (int64_t)sh.t_in.size()))
            return fail(KA_ERR_INTERNAL, "ka_debug_plan_tiles_width: count_tiles disagrees with plan_tiles");
    }
    if (!sh.tileable) return 0;
    for (size_t b = 0; b < sh.t_in.size() && b < (size_t)max_tiles; ++b) {
        t_in[b] = sh.t_in[b];
        t_end[b] = sh.t_end[b];
    }
    if (checkpoint_pitch) *checkpoint_pitch = (int64_t)sh.ck_pitch;
    return (int)sh.t_in.size();
}

int ka_debug_plan_tiles(int64_t T, int64_t S, int32_t V, int32_t beam_size, int32_t max_move, int32_t *t_in, int32_t *t_end,
                        int32_t max_tiles, int64_t *checkpoint_pitch)
{
    return ka_debug_plan_tiles_width(T, S, V, beam_size, max_move, ka::kTpTile, t_in, t_end, max_tiles, checkpoint_pitch);
}

int ka_debug_tile_width_choice(const int64_t *T, const int64_t *S, int32_t n, int32_t V, int32_t beam_size, int32_t max_move, int32_t n_simd)
{
    if (n < 0 || (n > 0 && (!T || !S)) || n_simd < 4) return fail(KA_ERR_BAD_ARGS, "ka_debug_tile_width_choice: bad arguments");
    std::vector<ka::plan::TileCount> counts(n);
    std::vector<int64_t> widths(n);
    for (int32_t i = 0; i < n; ++i) {
        Shape sh;
        if (!ka::plan::shape_of(T[i], S[i], V, beam_size, max_move, sh)) return fail(KA_ERR_BAD_ARGS, "ka_debug_tile_width_choice: bad shape");
        if (!ka::plan::count_tiles(sh, V, beam_size, max_move, ka::kTpTile).tileable) return 0;
        counts[i] = ka::plan::count_tiles(sh, V, beam_size, max_move, ka::kTnTile);
        widths[i] = sh.W;
    }
    return ka::plan::narrow_tiles_pay(counts, widths, n_simd, 0) ? ka::kTnTile : ka::kTpTile;
}

int ka_debug_auto_split(const int64_t *T, int32_t n, int32_t tiles_alive, int32_t n_simd, int32_t *n_tiled, int32_t *n_parallel)
{
    if (n < 0 || (n > 0 && !T) || tiles_alive < 1 || n_simd < 1 || !n_tiled || !n_parallel)
        return fail(KA_ERR_BAD_ARGS, "ka_debug_auto_split: bad arguments");
    std::vector<int64_t> Ts(T, T + n);
    std::sort(Ts.begin(), Ts.end(), [](int64_t a, int64_t b) { return a > b; });
    std::vector<int32_t> alive((size_t)n, tiles_alive);
    *n_tiled = ka::plan::auto_split_forward(Ts, alive, n_simd);
    *n_parallel = ka::plan::auto_split_backtrace(Ts, n_simd);
    return KA_OK;
}

int ka_debug_tile_stats(ka_engine *e, uint64_t *out, int32_t max_tasks)
{
    if (!e || !out || max_tasks < 0) return fail(KA_ERR_BAD_ARGS, "ka_debug_tile_stats: bad arguments");
    DeviceGuard guard;
    KA_HIP(guard.enter(e->device));
    const size_t n = std::min<size_t>(e->dbg_n_tasks, (size_t)max_tasks);
    std::vector<ka::TileTask> tk(n);
    std::vector<ka::TpStats> st(n);
    if (n) {
        KA_HIP(hipMemcpy(tk.data(), e->ws + e->dbg_tasks, n * sizeof(ka::TileTask), hipMemcpyDeviceToHost));
        KA_HIP(hipMemcpy(st.data(), e->ws + e->dbg_stats, n * sizeof(ka::TpStats), hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < n; ++i) {
        uint64_t *o = out + 8 * i;
        o[0] = (uint64_t)tk[i].lat; o[1] = (uint64_t)tk[i].tile; o[2] = (uint64_t)tk[i].t_in; o[3] = (uint64_t)tk[i].t_end;
        o[4] = st[i].wait_ticks; o[5] = st[i].total_ticks; o[6] = st[i].spins; o[7] = st[i].start_tick;
        o[2] |= (st[i].phase[2] >> 32) << 32;      // (two-wavefront tiles: HW_ID of the compute wavefront in the high half of t_in)
        if ((e->verify & 4) && (i == 0 || i == 10 || i == 20)) std::fprintf(stderr, "[ka_debug_tile_stats] ticket %zu cycles per phase: wait %llu, check+sum %llu, progress %llu, requests %llu, publish %llu\n", i,
                                 (unsigned long long)(uint32_t)st[i].phase[0], (unsigned long long)(st[i].phase[0] >> 32), (unsigned long long)(uint32_t)st[i].phase[1],
                                 (unsigned long long)(st[i].phase[1] >> 32), (unsigned long long)(uint32_t)st[i].phase[2]);
        if ((e->verify & 4) && (i == 0 || i == 10 || i == 20) && st[i].extra[0]) std::fprintf(stderr, "[ka_debug_tile_stats] ticket %zu compute wavefront: %llu cycles in frame blocks, %llu at barriers; look-up wavefront busy %llu\n", i,
                                 (unsigned long long)(st[i].extra[0] >> 32), (unsigned long long)(uint32_t)st[i].extra[0], (unsigned long long)st[i].extra[1]);
    }
    return (int)n;
}

int ka_log_softmax_f32(const float *logits, float *log_probs, int64_t T, int32_t V, int64_t ld_in, int64_t ld_out,
                       void *stream)
{
    if (!logits || !log_probs || T < 0 || V < 1 || ld_in < V || ld_out < V) return fail(KA_ERR_BAD_ARGS, "ka_log_softmax_f32: bad arguments");
    if (T == 0) return KA_OK;
    if ((T + 3) / 4 > 0x7fffffff) return fail(KA_ERR_BAD_ARGS, "ka_log_softmax_f32: T too large");
    ka::launch_log_softmax(logits, log_probs, T, V, ld_in, ld_out, (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_lstm_step_f32(const float *gin, int64_t ldg, const float *rec, int64_t rec_dir_stride, float *c, float *h,
                     int64_t state_dir_stride, float *out, int64_t ldo, const int32_t *rows, int64_t rows_dir_stride,
                     int32_t n, int32_t H, void *stream)
{
    if (!gin || !rec || !c || !h || !out || !rows || n < 0 || H < 1 || ldg < 8 * (int64_t)H || ldo < 2 * (int64_t)H)
        return fail(KA_ERR_BAD_ARGS, "ka_lstm_step_f32: bad arguments");
    if (n == 0) return KA_OK;
    ka::launch_lstm_step(gin, ldg, rec, rec_dir_stride, c, h, state_dir_stride, out, ldo, rows, rows_dir_stride, n, H, (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_lstm_layer_f32(const float *gin, int64_t ldg, const float *w_hh, float *out, int64_t ldo, const int32_t *seq_off,
                      const int32_t *seq_len, int32_t nseq, int32_t H, void *stream)
{
    if (!gin || !w_hh || !out || !seq_off || !seq_len || nseq < 0 || ldg < 8 * (int64_t)H || ldo < 2 * (int64_t)H)
        return fail(KA_ERR_BAD_ARGS, "ka_lstm_layer_f32: bad arguments");
    if (H != ka::kLstmH) return fail(KA_ERR_BAD_ARGS, "ka_lstm_layer_f32: the persistent kernel is built for hidden size 128");
    if (nseq == 0) return KA_OK;
    ka::launch_lstm_layer(false, gin, ldg, w_hh, out, ldo, seq_off, seq_len, nseq, nullptr, nullptr, (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_lstm_layer0_f32(const float *x, int64_t ldx, int32_t n_in, const float *w_ih, const float *bias, const float *w_hh, float *out, int64_t ldo,
                       const int32_t *seq_off, const int32_t *seq_len, int32_t nseq, int32_t H, void *stream)
{
    if (!x || !w_ih || !bias || !w_hh || !out || !seq_off || !seq_len || nseq < 0 || ldx < n_in || ldo < 2 * (int64_t)H)
        return fail(KA_ERR_BAD_ARGS, "ka_lstm_layer0_f32: bad arguments");
    if (H != ka::kLstmH || n_in != ka::kLstmIn)
        return fail(KA_ERR_BAD_ARGS, "ka_lstm_layer0_f32: built for hidden size 128 and 40 input features");
    if (nseq == 0) return KA_OK;
    ka::launch_lstm_layer(true, x, ldx, w_hh, out, ldo, seq_off, seq_len, nseq, w_ih, bias, (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_window_energy_f32(const float *x, int64_t n_windows, int32_t window, float *out, void *stream)
{
    if (!x || !out || n_windows < 0) return fail(KA_ERR_BAD_ARGS, "ka_window_energy_f32: bad arguments");
    if (window != 256) return fail(KA_ERR_BAD_ARGS, "ka_window_energy_f32: the summation order is NumPy's for windows of 256 samples only");
    if (n_windows == 0) return KA_OK;
    if ((n_windows + 15) / 16 > 0x7fffffff) return fail(KA_ERR_BAD_ARGS, "ka_window_energy_f32: too many windows");
    ka::launch_window_energy(x, n_windows, out, (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_stft_frames_f32(const float *y, const int64_t *seg_start, const int64_t *seg_len, const int64_t *frame_off, int32_t nseg,
                       int64_t max_frames, int32_t n_fft, int32_t hop, const float *window, float *frames, int64_t ld, void *stream)
{
    if (!y || !seg_start || !seg_len || !frame_off || !window || !frames || nseg < 0 || n_fft < 2 || hop < 1 || ld < n_fft || max_frames < 0)
        return fail(KA_ERR_BAD_ARGS, "ka_stft_frames_f32: bad arguments");
    if (nseg == 0 || max_frames == 0) return KA_OK;
    if (nseg > 65535) return fail(KA_ERR_BAD_ARGS, "ka_stft_frames_f32: more than 65535 segments in one call");
    ka::launch_stft_frames(y, seg_start, seg_len, frame_off, (unsigned)std::min<int64_t>(max_frames, 4096), (unsigned)nseg, n_fft, hop, window, frames, ld,
                           (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_power_f32(const float *reim, int64_t ld_in, float *power, int64_t ld_out, int64_t n, int32_t nf, void *stream)
{
    if (!reim || !power || n < 0 || nf < 1 || ld_in < 2 * (int64_t)nf || ld_out < nf) return fail(KA_ERR_BAD_ARGS, "ka_power_f32: bad arguments");
    if (n == 0) return KA_OK;
    ka::launch_power(reim, ld_in, power, ld_out, n, nf, (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_power_to_db_f32(float *x, int64_t ld, int32_t cols, const int64_t *frame_off, int32_t nseg, int64_t max_frames, float top_db,
                       float *segmax, void *stream)
{
    if (!x || !frame_off || !segmax || nseg < 0 || cols < 1 || ld < cols || max_frames < 0) return fail(KA_ERR_BAD_ARGS, "ka_power_to_db_f32: bad arguments");
    if (nseg == 0 || max_frames == 0) return KA_OK;
    if (nseg > 65535) return fail(KA_ERR_BAD_ARGS, "ka_power_to_db_f32: more than 65535 segments in one call");
    ka::launch_power_to_db(x, ld, cols, frame_off, (unsigned)std::min<int64_t>((max_frames * cols + 255) / 256, 256), (unsigned)nseg, top_db, segmax,
                           (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_hash_logprobs_batch_f32(float *dev_log_probs, int32_t n, int64_t T, int32_t V, int64_t ld, int64_t lattice_stride,
                               uint64_t seed0, void *stream)
{
    if (!dev_log_probs || n < 0 || T < 0 || V < 1 || ld < V || (n > 1 && lattice_stride < T * ld))
        return fail(KA_ERR_BAD_ARGS, "ka_hash_logprobs_batch_f32: bad arguments");
    if (T == 0 || n == 0) return KA_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((T * V + 255) / 256, 512);
    for (int32_t y0 = 0; y0 < n; y0 += 65535)
        ka::launch_hash_logprobs(dev_log_probs + (size_t)y0 * (size_t)lattice_stride, blocks, (unsigned)std::min<int32_t>(65535, n - y0), T, V, ld,
                                 seed0 + (uint64_t)y0, lattice_stride, (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_hash_labels_batch_i32(int32_t *dev_labels, int32_t n, int64_t S, int32_t V, int64_t lattice_stride, uint64_t seed0,
                             void *stream)
{
    if (!dev_labels || n < 0 || S < 0 || V < 2 || (n > 1 && lattice_stride < S))
        return fail(KA_ERR_BAD_ARGS, "ka_hash_labels_batch_i32: bad arguments");
    if (S == 0 || n == 0) return KA_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((S + 255) / 256, 64);
    for (int32_t y0 = 0; y0 < n; y0 += 65535)
        ka::launch_hash_labels(dev_labels + (size_t)y0 * (size_t)lattice_stride, blocks, (unsigned)std::min<int32_t>(65535, n - y0), S, V, seed0 + (uint64_t)y0,
                               lattice_stride, (hipStream_t)stream);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_hash_logprobs_f32(float *dev_log_probs, int64_t T, int32_t V, int64_t ld, uint64_t seed, void *stream)
{
    return ka_hash_logprobs_batch_f32(dev_log_probs, 1, T, V, ld, T * ld, seed, stream);
}

int ka_hash_labels_i32(int32_t *dev_labels, int64_t S, int32_t V, uint64_t seed, void *stream)
{
    return ka_hash_labels_batch_i32(dev_labels, 1, S, V, S, seed, stream);
}

}  // extern "C"
