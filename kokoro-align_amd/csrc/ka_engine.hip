// ka_engine.hip — host side of the C ABI declared in include/kokoro_align_amd.h.
//
// Builds one descriptor per lattice, carves the device workspace (padded labels,
// back-pointer storage), launches prep -> forward DP -> backtrace on the caller's stream and
// reports per-lattice status.  No torch, no oracle, no CPU fallback: if HIP fails the call
// fails.
#include "../../include/kokoro_align_amd.h"
#include "ka_kernels.hpp"
#include "ka_tiled.hpp"
#include "ka_tiled2.hpp"
#include "ka_tiled_narrow.hpp"
#include "ka_parallel_bt.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

namespace {

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

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

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

// KA_MODE_AUTO never picks the 4-wavefront form any more: since the checkpointed forward kernel lost a third of its
// instructions a lone wavefront does a cfg2 lattice in 11.2 + 8.2 ms against 18.7 + 1.4 ms for four wavefronts with
// stored back-pointers, and it stays ahead at every batch size (64: 19.9 vs 20.4 ms, 512: 21.4 vs 23.1 ms).

struct Shape {
    int64_t T, S, L, W;
    int32_t labx_len;
    bool fast;
    // tiled form (ka_tiled.hpp): tiles 0 .. n_act-1 of 256 positions each are alive in frames [t_in, t_end)
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
inline int64_t chunks_of_T(int64_t T) { return (T - 1) / ka::kCkFrames + 1; }
inline int64_t supers_of_T(int64_t T) { return (chunks_of_T(T) + ka::kSuperChunks - 1) / ka::kSuperChunks; }
inline size_t par_bt_bytes(const Shape &sh)
{
    const size_t R = (sh.tiled ? sh.ck_pitch : 4096) / 4;
    return align_up((size_t)chunks_of_T(sh.T) * R) + align_up((size_t)supers_of_T(sh.T) * R * 2) +
           align_up((size_t)(chunks_of_T(sh.T) + supers_of_T(sh.T)) * 4);
}


inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Which frames each tile of P positions (256, or 128: ka_tiled_narrow.hpp) is alive in, from the band of align.py:64-65:
//   lo(t) = max(0, floor(L t / T) - B/2),  hi(t) = min(lo(t) + B, L)
//   t_in(b)  = first t with hi(t) > P b        = 0 if P b < B, else ceil((P b - B + B/2 + 1) T / L)
//   t_end(b) = first t with lo(t) >= P (b+1)   = ceil((P (b+1) + B/2) T / L), at most T
void plan_tiles(Shape &sh, int32_t V, int32_t beam, int32_t max_move, int64_t P = ka::kTpTile)
{
    sh.tileable = false;
    if (V > 64 || max_move > 4 || beam < 1 || sh.T >= (int64_t(1) << 26)) return;
    const int64_t T = sh.T, L = sh.L, B = beam, h = B / 2;
    const int64_t n_tiles = ceil_div(L, P);
    sh.t_in.clear();
    sh.t_end.clear();
    sh.n_final = 0;
    sh.halo_bytes = 0;
    for (int64_t b = 0; b < n_tiles; ++b) {
        const int64_t x = b * P, z = (b + 1) * P;
        const int64_t ti = x < B ? 0 : ceil_div((x - B + h + 1) * T, L);
        if (ti >= T) break;
        const int64_t te = std::min<int64_t>(T, ceil_div((z + h) * T, L));
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
    const size_t whole = (size_t)ceil_div(L, ka::kTpTile) * ka::kTpTile;   // (whatever P: the readers' windows may reach up to the next multiple of 256)
    if (whole <= ring) {
        sh.ck_mask = 0xffffffffu;
        sh.ck_pitch = whole * 4;
    } else {
        sh.ck_mask = (uint32_t)ring - 1;
        sh.ck_pitch = ring * 4;
    }
    sh.tileable = true;
}

// Tile width of a launch's tiled lattices: 128 positions (two cells per lane and three wavefronts per tile, ka_tiled_narrow.hpp:
// a frame of half the instructions, twice the tiles and twice the hand-offs, 46-52 KB of LDS per tile) while the tiles alive
// at once are no more than 2.6 per workgroup slot of the device, else 256.  Measured on prefixes of the corpus stand-in, all
// tiled, V = 39: three workgroups per CU (tools/sweep_width.py, profiles/r03_sweep_width.jsonl): against 256 positions the
// forward kernel takes 0.66 x the time for one chapter, 0.70 x for 64 (~580 tiles alive), 0.78 x for 128, 0.94 x for 200 (~1800),
// 1.17 x for 320 (~2900).  Tiles that never die (a band as wide as the label axis) must all hold a slot at once: the whole
// 500 000 x 100 001 lattice, 782 tiles of 128 positions on 512 slots, took 89 ms instead of 56.
// `plans`: the 128-position plan of every tiled lattice (tileable or not).  forced: 0 = by the rule, 128, 256.
bool narrow_tiles_pay(const std::vector<Shape> &plans, int32_t V, int32_t max_move, int32_t n_simd, int32_t forced)
{
    if (forced == ka::kTpTile || plans.empty()) return false;
    int64_t alive_now = 0, permanent = 0;     // tiles alive at once: of banded lattices (they come and go), of those that are all band
    for (const Shape &p : plans) {
        if (!p.tileable) return false;        // (L/T above 128: the band jumps over a whole tile in one frame)
        const int64_t n_tiles = (int64_t)p.t_in.size(), in_band = (p.W + 2 * ka::kTnTile - 1) / ka::kTnTile;
        if (n_tiles <= in_band) permanent += n_tiles;
        else alive_now += in_band;
    }
    if (forced == ka::kTnTile) return true;
    (void)V; (void)max_move;
    const int64_t slots = (int64_t)(n_simd / 4) * 3;   // 46-52 KB of LDS per workgroup: three per CU
    return permanent <= slots && 5 * alive_now <= 13 * (slots - permanent);
}

bool shape_of(int64_t T, int64_t S, int32_t V, int32_t beam, int32_t max_move, Shape &sh)
{
    if (T < 1 || S < 0 || V < 1 || beam < 0 || max_move < 1 || max_move > 255) return false;
    if (T >= (int64_t(1) << 31) - 64 || S >= (int64_t(1) << 29)) return false;
    sh.T = T;
    sh.S = S;
    sh.L = 2 * S + 1;
    sh.W = std::max<int64_t>(1, std::min<int64_t>(beam, sh.L));
    sh.labx_len = (int32_t)align_up((size_t)S + 1024, 8);
    sh.fast = V <= 64 && max_move <= 4 && std::min<int64_t>(beam, sh.L) <= ka::kFastMaxBand;
    return true;
}

// bytes of the back-pointer / checkpoint region of a lattice
size_t bp_region_bytes(const Shape &sh)
{
    size_t b = 0;
    if (sh.fast) b = (((size_t)sh.T + 3) / 4) * 1024;                       // exact forms: 256 B per frame (checkpoints: 128)
    else if (!sh.tiled) b = (size_t)sh.T * (size_t)sh.W;                     // generic: a byte per band cell
    if (sh.tiled) b = std::max(b, (size_t)((sh.T - 1) / ka::kCkFrames) * sh.ck_pitch);
    return align_up(b);
}
// device bytes a lattice needs besides the caller's buffers
size_t lattice_ws_bytes(const Shape &sh)
{
    size_t b = align_up((size_t)sh.labx_len * 4) + bp_region_bytes(sh);
    if (!sh.fast && !sh.tiled) b += align_up((size_t)sh.L * 2 * sizeof(float) + (size_t)sh.L * 2);
    if (sh.tiled) b += sh.halo_bytes;
    if (sh.par_bt) b += par_bt_bytes(sh);
    return b;
}

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
    int32_t n_simd = 1024;                 // SIMDs of the device = persistent workers of the tiled form
    std::vector<int32_t> wide_tiled;       // last batch: lattices in the tiled form that the exact kernels cannot redo
    // ... and what ka_batch_finish needs to hand those of them that the tiled form declined (non-finite log-probs) to the
    // generic kernels: the caller's buffers (valid until finish returns, by the contract of the split form)
    struct Redo { const float *lp; const int32_t *labels; int32_t *path, *lab_out; float *sc_out; int64_t T, S, ld; int32_t idx; };
    std::vector<Redo> redo;
    int32_t last_V = 0, last_beam = 0, last_max_move = 0, last_mem = KA_MEM_DEVICE;
    int32_t verify = 0;                    // ka_engine_set_verify: self-checks of the tiled form's hand-off
    int32_t tile_waves = 2;                // ka_engine_set_tile_waves: wavefronts per tile of the tiled form
    int32_t rc_gather = -1;                // ka_debug_set_rc_gather: -1 the library's rule, 0 / 1 the serial backtrace's output form
    int32_t tile_gather = -1;              // ka_debug_set_tile_gather: the 128-position tiles' feeder looks up the emissions (1), does not (0), -1 = the engine chooses
    int32_t tile_width = 0;                // ka_debug_set_tile_width: 0 = the engine chooses, 128 or 256
    int32_t tile_lds = 0;                  // ka_debug_set_split's third knob: LDS bytes a tile workgroup requests (0: kTpLdsRequest)
    int32_t split_tiled = -1, split_par = -1;   // ka_debug_set_split: how many of the longest lattices run tiled / are walked back chunk-parallel (-1: cost model)
    hipStream_t aux = nullptr;             // second stream: the other kernel form of a mixed launch runs beside the first
    hipEvent_t sync[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t dbg_entry = 0, dbg_entry_n = 0, dbg_map0 = 0, dbg_map0_bytes = 0;   // last batch, descriptor 0: chunk entries and chunk maps
    size_t dbg_tasks = 0, dbg_stats = 0, dbg_n_tasks = 0;   // last batch: workspace offsets of the tile tasks and their timing records
};

namespace {

int ensure_ws(ka_engine *e, size_t bytes)
{
    if (bytes <= e->ws_bytes) return KA_OK;
    KA_HIP(hipDeviceSynchronize());
    if (e->ws) KA_HIP(hipFree(e->ws));
    e->ws = nullptr;
    e->ws_bytes = 0;
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

enum Form { kFormWorkgroup, kFormWaveExact, kFormWaveCheckpointed };

template <int M>
void launch_forward(const ka::Lattice *d_lats, int n, int32_t *d_meta, hipStream_t s, Form form)
{
    // two launches over the same lattices: a lattice is taken by the kernel that matches its
    // "transcript contains label 0" flag, the other one's waves exit at once
    if (form == kFormWorkgroup) {
        hipLaunchKernelGGL((ka::forward_wg4_kernel<M, false>), dim3(n), dim3(256), 0, s, d_lats, d_meta);
        hipLaunchKernelGGL((ka::forward_wg4_kernel<M, true>), dim3(n), dim3(256), 0, s, d_lats, d_meta);
        return;
    }
    if (form == kFormWaveCheckpointed) {
        hipLaunchKernelGGL((ka::forward_ck_kernel<M, false>), dim3(n), dim3(64), 0, s, d_lats, d_meta);
        hipLaunchKernelGGL((ka::forward_ck_kernel<M, true>), dim3(n), dim3(64), 0, s, d_lats, d_meta);
    }
    // exact kernels: everything (kFormWaveExact) or only what the checkpointed kernels declined
    const int only_flagged = form == kFormWaveCheckpointed ? 1 : 0;
    hipLaunchKernelGGL((ka::forward_w16_kernel<M, false>), dim3(n), dim3(64), 0, s, d_lats, d_meta, only_flagged);
    hipLaunchKernelGGL((ka::forward_w16_kernel<M, true>), dim3(n), dim3(64), 0, s, d_lats, d_meta, only_flagged);
}

// chunk maps -> super-chunk maps -> entry position of every chunk -> every chunk walked at once (ka_parallel_bt.hpp)
template <int M>
void launch_parallel_bt(const ka::Lattice *d_lats, int n, int32_t *d_meta, hipStream_t s, unsigned total_chunks, unsigned max_seg, unsigned max_sup,
                        unsigned max_w)
{
    hipLaunchKernelGGL((ka::chunk_map_kernel<M, false>), dim3(total_chunks, max_seg), dim3(64), 0, s, d_lats, d_meta, n);
    hipLaunchKernelGGL((ka::chunk_map_kernel<M, true>), dim3(total_chunks, max_seg), dim3(64), 0, s, d_lats, d_meta, n);
    hipLaunchKernelGGL(ka::compose_maps_kernel, dim3(std::min(64u, (max_w + 255u) / 256u), max_sup, (unsigned)n), dim3(256), 0, s, d_lats, d_meta);
    hipLaunchKernelGGL(ka::chain_entries_kernel, dim3((unsigned)n), dim3(256), 0, s, d_lats, d_meta);
    hipLaunchKernelGGL((ka::backtrace_rc_kernel<M, false, true>), dim3(total_chunks), dim3(64), 0, s, d_lats, d_meta, n);
    hipLaunchKernelGGL((ka::backtrace_rc_kernel<M, true, true>), dim3(total_chunks), dim3(64), 0, s, d_lats, d_meta, n);
}

// exact kernels over lattices another forward kernel has flagged kFlagExact
template <int M>
void launch_forward_flagged(const ka::Lattice *d_lats, int n, int32_t *d_meta, hipStream_t s)
{
    hipLaunchKernelGGL((ka::forward_w16_kernel<M, false>), dim3(n), dim3(64), 0, s, d_lats, d_meta, 1);
    hipLaunchKernelGGL((ka::forward_w16_kernel<M, true>), dim3(n), dim3(64), 0, s, d_lats, d_meta, 1);
}

// gather: the serial walk of at least a wavefront per SIMD fetches labels and scores after the walk (backtrace_rc_kernel<.., GO>)
template <int M>
void launch_backtrace_rc(const ka::Lattice *d_lats, int n, int32_t *d_meta, hipStream_t s, bool gather)
{
    if (gather) {
        hipLaunchKernelGGL((ka::backtrace_rc_kernel<M, false, false, true>), dim3(n), dim3(64), 0, s, d_lats, d_meta, n);
        hipLaunchKernelGGL((ka::backtrace_rc_kernel<M, true, false, true>), dim3(n), dim3(64), 0, s, d_lats, d_meta, n);
    } else {
        hipLaunchKernelGGL((ka::backtrace_rc_kernel<M, false, false, false>), dim3(n), dim3(64), 0, s, d_lats, d_meta, n);
        hipLaunchKernelGGL((ka::backtrace_rc_kernel<M, true, false, false>), dim3(n), dim3(64), 0, s, d_lats, d_meta, n);
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

size_t ka_workspace_bytes(int32_t n, const int64_t *T, const int64_t *S, int32_t V, int32_t beam_size,
                          int32_t max_move)
{
    if (n < 0 || !T || !S) return 0;
    size_t total = align_up((size_t)n * sizeof(ka::Lattice)) + align_up((size_t)n * 16);
    size_t tasks = 0;
    int64_t ninf_slots = 0;
    for (int32_t i = 0; i < n; ++i) {
        Shape sh;
        if (!shape_of(T[i], S[i], V, beam_size, max_move, sh)) return 0;
        size_t plain = lattice_ws_bytes(sh);
        plan_tiles(sh, V, beam_size, max_move);       // whichever form the call ends up in: the larger of the two
        sh.par_bt = sh.fast;
        plain = lattice_ws_bytes(sh);
        if (sh.tileable) {
            sh.tiled = true;
            sh.par_bt = true;
            plain = std::max(plain, lattice_ws_bytes(sh));
            size_t wide_tasks = sh.t_in.size();
            ninf_slots = std::max<int64_t>(ninf_slots, sh.t_end[0]);
            plan_tiles(sh, V, beam_size, max_move, ka::kTnTile);   // (the 128-position tiles: twice the boundaries)
            if (sh.tileable) {
                plain = std::max(plain, lattice_ws_bytes(sh));
                wide_tasks = std::max(wide_tasks, sh.t_in.size());
            }
            tasks += wide_tasks;
        }
        total += plain;
    }
    if (tasks)
        total += align_up(align_up((1 + tasks) * 4 + (size_t)n * sizeof(ka::TileAux) + 16, 16)) + align_up(tasks * sizeof(ka::TileTask)) +
                 align_up(tasks * sizeof(ka::TpStats)) + align_up((size_t)(ninf_slots + 2 * ka::kTpBlock) * 16);
    return total;
}

static int enqueue_impl(ka_engine *e, int32_t n, const float *const *log_probs, const int64_t *T, int32_t V, const int64_t *ld,
                        const int32_t *const *labels, const int64_t *S, int32_t beam_size, int32_t max_move, int32_t *const *best_path,
                        int32_t *const *best_labels, float *const *best_scores, int32_t mem, hipStream_t stream, bool force_generic,
                        size_t *plan_only_bytes);

size_t ka_engine_workspace_bytes(ka_engine *e, int32_t n, const int64_t *T, const int64_t *S, int32_t V, int32_t beam_size, int32_t max_move,
                                 int32_t mem)
{
    if (!e) return 0;
    size_t bytes = 0;
    const bool pending = e->pending;
    e->pending = false;
    const int rc = enqueue_impl(e, n, nullptr, T, V, nullptr, nullptr, S, beam_size, max_move, nullptr, nullptr, nullptr, mem, nullptr, false, &bytes);
    e->pending = pending;
    return rc == KA_OK ? bytes : 0;
}

int ka_engine_set_mode(ka_engine *e, int32_t mode)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (mode != KA_MODE_AUTO && mode != KA_MODE_WAVE && mode != KA_MODE_WORKGROUP && mode != KA_MODE_WAVE_EXACT && mode != KA_MODE_TILED)
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
    if (bytes != 0 && (bytes < (int32_t)ka::kTpLdsRequest || bytes > 160 * 1024)) return fail(KA_ERR_BAD_ARGS, "ka_debug_set_tile_lds: 0 or 40 KB .. 160 KB");
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

int ka_debug_set_tile_gather(ka_engine *e, int32_t how)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (how < -1 || how > 1) return fail(KA_ERR_BAD_ARGS, "ka_debug_set_tile_gather: -1, 0 or 1");
    e->tile_gather = how;
    return KA_OK;
}

int ka_engine_set_tile_waves(ka_engine *e, int32_t waves)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (waves != 1 && waves != 2) return fail(KA_ERR_BAD_ARGS, "ka_engine_set_tile_waves: 1 or 2");
    e->tile_waves = waves;
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
//   chunk-parallel backtrace             0.00066 per frame of every lattice in it (it recomputes the whole band) + 0.12 ms
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
    double serial_chain = 0.173, serial_thr = 0.0675;
    double par_frame = 0.00066, par_fixed = 120.0;
    double fork = 15.0;      // a second stream and its two event waits
};
constexpr AutoCosts kAuto;

// lattices sorted longest first; alive[i] = tiles of lattice i that run at the same time.  Returns how many of the longest to tile.
static int32_t auto_split_forward(const std::vector<int64_t> &T, const std::vector<int32_t> &alive, int32_t n_simd)
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
static int32_t auto_split_backtrace(const std::vector<int64_t> &T, int32_t n_simd)
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

static int enqueue_impl(ka_engine *e, int32_t n, const float *const *log_probs, const int64_t *T, int32_t V,
                        const int64_t *ld, const int32_t *const *labels, const int64_t *S, int32_t beam_size,
                        int32_t max_move, int32_t *const *best_path, int32_t *const *best_labels,
                        float *const *best_scores, int32_t mem, hipStream_t stream, bool force_generic, size_t *plan_only_bytes)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (e->pending) return fail(KA_ERR_BAD_ARGS, "a batch is already enqueued: call ka_batch_finish first");
    const bool plan_only = plan_only_bytes != nullptr;     // ka_engine_workspace_bytes: sizes only, nothing is launched
    if (n < 0 || (n > 0 && (!T || !S || (!plan_only && (!log_probs || !ld || !labels || !best_path || !best_labels || !best_scores)))))
        return fail(KA_ERR_BAD_ARGS, "batch: NULL array argument");
    if (mem != KA_MEM_HOST && mem != KA_MEM_DEVICE) return fail(KA_ERR_BAD_ARGS, "mem must be KA_MEM_HOST or KA_MEM_DEVICE");
    DeviceGuard guard;
    if (!plan_only) {
        KA_HIP(guard.enter(e->device));
        e->n_last = n;
        e->stream_last = stream;
        e->have_times = false;
        e->wide_tiled.clear();
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

    std::vector<Shape> sh(n);
    for (int32_t i = 0; i < n; ++i) {
        if (!shape_of(T[i], S[i], V, beam_size, max_move, sh[i]) || (!plan_only && ld[i] < V))
            return fail(KA_ERR_BAD_ARGS, "lattice " + std::to_string(i) + ": unsupported T/S/V/ld/beam_size/max_move");
        if (!plan_only && (!log_probs[i] || !best_path[i] || !best_labels[i] || !best_scores[i] || (S[i] > 0 && !labels[i])))
            return fail(KA_ERR_BAD_ARGS, "lattice " + std::to_string(i) + ": NULL buffer");
    }
    // ---- which lattices run in the tiled form (ka_tiled.hpp, ka_tiled2.hpp) ----
    //   KA_MODE_TILED: every lattice that can;  KA_MODE_AUTO: bands too wide for the one-wavefront ring always, and of the
    //   others the longest k, k from the cost model above (e->split_tiled >= 0: k given, for the calibration sweeps)
    const bool checkpointed_waves = e->mode != KA_MODE_WORKGROUP && e->mode != KA_MODE_WAVE_EXACT;   // the one-wavefront lattices end in backtrace_rc
    int32_t n_tiled = 0;
    if (!force_generic && (e->mode == KA_MODE_TILED || e->mode == KA_MODE_AUTO)) {
        std::vector<int32_t> cand;      // fast-shaped lattices that could run tiled, longest first
        for (int32_t i = 0; i < n; ++i) {
            if (e->mode == KA_MODE_AUTO && sh[i].fast && sh[i].T >= (int64_t(1) << 26)) continue;   // (runs in the exact form)
            plan_tiles(sh[i], V, beam_size, max_move);
            if (!sh[i].tileable) continue;
            if (e->mode == KA_MODE_TILED || !sh[i].fast) sh[i].tiled = true;
            else cand.push_back(i);
        }
        if (!cand.empty()) {
            std::stable_sort(cand.begin(), cand.end(), [&](int32_t a, int32_t b) { return sh[a].T > sh[b].T; });
            std::vector<int64_t> Ts(cand.size());
            std::vector<int32_t> alive(cand.size());
            for (size_t j = 0; j < cand.size(); ++j) {
                const Shape &p = sh[cand[j]];
                Ts[j] = p.T;
                alive[j] = (int32_t)std::min<int64_t>((int64_t)p.t_in.size(), (p.W + 2 * ka::kTpTile - 1) / ka::kTpTile);
            }
            int32_t k = e->split_tiled >= 0 ? std::min<int32_t>(e->split_tiled, (int32_t)cand.size()) : auto_split_forward(Ts, alive, e->n_simd);
            for (int32_t j = 0; j < k; ++j) sh[cand[j]].tiled = true;
        }
        for (int32_t i = 0; i < n; ++i) n_tiled += sh[i].tiled ? 1 : 0;
    }
    // ---- tile width (narrow_tiles_pay above) ----
    bool narrow = false;
    if (n_tiled > 0 && e->tile_waves == 2 && e->tile_width != ka::kTpTile) {
        std::vector<Shape> alt;
        for (int32_t i = 0; i < n; ++i) {
            if (!sh[i].tiled) continue;
            Shape p = sh[i];
            plan_tiles(p, V, beam_size, max_move, ka::kTnTile);
            alt.push_back(std::move(p));
        }
        if (narrow_tiles_pay(alt, V, max_move, e->n_simd, e->tile_width)) {
            narrow = true;
            size_t j = 0;
            for (int32_t i = 0; i < n; ++i)
                if (sh[i].tiled) {
                    Shape &p = alt[j++];
                    sh[i].t_in = std::move(p.t_in);
                    sh[i].t_end = std::move(p.t_end);
                    sh[i].n_final = p.n_final;
                    sh[i].halo_bytes = p.halo_bytes;
                    sh[i].ck_mask = p.ck_mask;
                    sh[i].ck_pitch = p.ck_pitch;
                }
        }
    }

    // ---- chunk-parallel backtrace (ka_parallel_bt.hpp) for the longest of the checkpointed results: it recomputes the
    // whole band of every chunk, ~8x the serial form's work, but all chunks at once; the others are walked back serially,
    // one wavefront each, at the same time on the engine's second stream
    {
        std::vector<int32_t> ring;      // lattices whose result backtrace_rc walks, longest first
        for (int32_t i = 0; i < n; ++i)
            if (sh[i].tiled || (sh[i].fast && checkpointed_waves && sh[i].T < (int64_t(1) << 26))) ring.push_back(i);
        std::stable_sort(ring.begin(), ring.end(), [&](int32_t a, int32_t b) { return sh[a].T > sh[b].T; });
        int32_t m = 0;
        if (e->backtrace == KA_BACKTRACE_PARALLEL) m = (int32_t)ring.size();
        else if (e->backtrace == KA_BACKTRACE_AUTO) {
            std::vector<int64_t> Ts(ring.size());
            for (size_t j = 0; j < ring.size(); ++j) Ts[j] = sh[ring[j]].T;
            m = e->split_par >= 0 ? std::min<int32_t>(e->split_par, (int32_t)ring.size()) : auto_split_backtrace(Ts, e->n_simd);
        }
        // grid limits of the chunk-parallel kernels
        int64_t chunks = 0;
        bool fits = (int64_t)n <= 65535;
        for (int32_t j = 0; j < m && fits; ++j) {
            const Shape &p = sh[ring[j]];
            chunks += chunks_of_T(p.T);
            fits = (p.W + 7 + ka::kCmOut - 1) / ka::kCmOut <= 65535 && supers_of_T(p.T) <= 65535 && chunks < (int64_t(1) << 31);
        }
        if (!fits) m = 0;
        for (int32_t j = 0; j < m; ++j) sh[ring[j]].par_bt = true;
    }

    // ---- carve the workspace ----
    size_t off = 0;
    const size_t off_desc = off;
    off += align_up((size_t)n * sizeof(ka::Lattice));
    const size_t off_meta = off;
    off += align_up((size_t)n * 16);
    // tiled form, per launch: [progress words | per-lattice terminal records | ticket] (zeroed every launch), the tile
    // tasks, and the halo region, which starts with the -inf slots that stand in for "the tile below tile 0"
    size_t n_tasks = 0;
    int64_t ninf_slots = 0;
    for (int32_t i = 0; i < n; ++i)
        if (sh[i].tiled) {
            n_tasks += sh[i].t_in.size();
            ninf_slots = std::max<int64_t>(ninf_slots, sh[i].t_end[0]);
        }
    const size_t off_zero = off;
    const size_t zero_bytes = n_tiled ? align_up((1 + n_tasks) * 4 + (size_t)n * sizeof(ka::TileAux) + 16, 16) : 0;
    const size_t off_prog = off_zero, off_aux = off_zero + align_up((1 + n_tasks) * 4, 16);
    const size_t off_ticket = off_aux + (size_t)n * sizeof(ka::TileAux);
    off += align_up(zero_bytes);
    const size_t off_tasks = off;
    off += align_up(n_tasks * sizeof(ka::TileTask));
    const size_t off_stats = off;
    off += align_up(n_tasks * sizeof(ka::TpStats));
    e->dbg_tasks = off_tasks;
    e->dbg_stats = off_stats;
    e->dbg_n_tasks = n_tasks;
    const size_t off_halo = off;
    const size_t ninf_bytes = n_tiled ? align_up((size_t)(ninf_slots + 2 * ka::kTpBlock) * 16) : 0;
    off += ninf_bytes;
    struct Carve { size_t labx, bp, col, halo, map0, map1, entry, lp, lab, path, labo, sco; };
    std::vector<Carve> cv(n);
    for (int32_t i = 0; i < n; ++i) {
        cv[i].labx = off;
        off += align_up((size_t)sh[i].labx_len * 4);
        cv[i].bp = off;
        off += bp_region_bytes(sh[i]);
        cv[i].col = off;
        if (!sh[i].fast && !sh[i].tiled) off += align_up((size_t)sh[i].L * 2 * sizeof(float) + (size_t)sh[i].L * 2);
        cv[i].halo = off;
        if (sh[i].tiled) off += sh[i].halo_bytes;
        cv[i].map0 = cv[i].map1 = cv[i].entry = off;
        if (sh[i].par_bt) {
            const size_t R = (sh[i].tiled ? sh[i].ck_pitch : 4096) / 4;
            off += align_up((size_t)chunks_of_T(sh[i].T) * R);
            cv[i].map1 = off;
            off += align_up((size_t)supers_of_T(sh[i].T) * R * 2);
            cv[i].entry = off;
            off += align_up((size_t)(chunks_of_T(sh[i].T) + supers_of_T(sh[i].T)) * 4);
        }
        if (mem == KA_MEM_HOST) {
            cv[i].lp = off;
            off += align_up((size_t)sh[i].T * (size_t)V * 4);
            cv[i].lab = off;
            off += align_up((size_t)std::max<int64_t>(sh[i].S, 1) * 4);
            cv[i].path = off;
            off += align_up((size_t)sh[i].T * 4);
            cv[i].labo = off;
            off += align_up((size_t)sh[i].T * 4);
            cv[i].sco = off;
            off += align_up((size_t)sh[i].T * 4);
        }
    }
    if (plan_only) {
        *plan_only_bytes = off;
        return KA_OK;
    }
    int rc = ensure_ws(e, off);
    if (rc != KA_OK) return rc;
    rc = ensure_pin(e, align_up((size_t)n * sizeof(ka::Lattice)) + align_up((size_t)n * 16) + align_up(n_tasks * sizeof(ka::TileTask)));
    if (rc != KA_OK) return rc;

    // ---- descriptors: tiled lattices first, then the one-wavefront ones (longest first: short tail), then generic ones ----
    auto klass = [&](int32_t i) { return sh[i].tiled ? 0 : (sh[i].fast ? 1 : 2); };
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        if (klass(a) != klass(b)) return klass(a) < klass(b);
        return sh[a].T > sh[b].T;
    });
    int32_t n_fast = 0;
    for (int32_t i = 0; i < n; ++i) n_fast += klass(i) == 1 ? 1 : 0;
    ka::Lattice *h_lats = reinterpret_cast<ka::Lattice *>(e->pin);
    e->h_meta = reinterpret_cast<int32_t *>(e->pin + align_up((size_t)n * sizeof(ka::Lattice)));
    ka::TileTask *h_tasks = reinterpret_cast<ka::TileTask *>(e->pin + align_up((size_t)n * sizeof(ka::Lattice)) + align_up((size_t)n * 16));
    int64_t chunk_cursor = 0;
    for (int32_t k = 0; k < n; ++k) {
        const int32_t i = order[k];
        ka::Lattice &d = h_lats[k];
        std::memset(&d, 0, sizeof(d));
        // chunks of the launch's chunk-parallel lattices are numbered consecutively; a lattice that is walked back serially
        // carries the running total and owns none (lattice_of_chunk picks the LAST descriptor whose chunk0 <= chunk)
        d.chunk0 = chunk_cursor;
        d.par = sh[i].par_bt ? 1 : 0;
        if (sh[i].par_bt) chunk_cursor += chunks_of_T(sh[i].T);
        if (mem == KA_MEM_HOST) {
            d.lp = reinterpret_cast<const float *>(e->ws + cv[i].lp);
            d.labels = reinterpret_cast<const int32_t *>(e->ws + cv[i].lab);
            d.path = reinterpret_cast<int32_t *>(e->ws + cv[i].path);
            d.lab_out = reinterpret_cast<int32_t *>(e->ws + cv[i].labo);
            d.sc_out = reinterpret_cast<float *>(e->ws + cv[i].sco);
            d.ld = V;
        } else {
            d.lp = log_probs[i];
            d.labels = labels[i];
            d.path = best_path[i];
            d.lab_out = best_labels[i];
            d.sc_out = best_scores[i];
            d.ld = ld[i];
        }
        d.labx = reinterpret_cast<int32_t *>(e->ws + cv[i].labx);
        d.bp = e->ws + cv[i].bp;
        d.col = reinterpret_cast<float *>(e->ws + cv[i].col);
        d.T = (int32_t)sh[i].T;
        d.S = (int32_t)sh[i].S;
        d.L = (int32_t)sh[i].L;
        d.V = V;
        d.beam = beam_size;
        d.max_move = max_move;
        d.labx_len = sh[i].labx_len;
        d.W = (int32_t)sh[i].W;
        d.idx = i;
        d.n_final = sh[i].tiled ? sh[i].n_final : 0;
        d.ck_mask = sh[i].tiled ? sh[i].ck_mask : 1023u;
        d.ck_pitch = sh[i].tiled ? (int32_t)sh[i].ck_pitch : 4096;
        d.map0 = reinterpret_cast<uint8_t *>(e->ws + cv[i].map0);
        d.map1 = reinterpret_cast<uint16_t *>(e->ws + cv[i].map1);
        d.entry = reinterpret_cast<int32_t *>(e->ws + cv[i].entry);
        if (k == 0) {
            e->dbg_entry = cv[i].entry;
            e->dbg_entry_n = sh[i].par_bt ? (size_t)(chunks_of_T(sh[i].T) + supers_of_T(sh[i].T)) : 0;
            e->dbg_map0 = cv[i].map0;
            e->dbg_map0_bytes = sh[i].par_bt ? (size_t)chunks_of_T(sh[i].T) * ((sh[i].tiled ? sh[i].ck_pitch : 4096) / 4) : 0;
        }
        if (sh[i].tiled && !sh[i].fast) {
            e->wide_tiled.push_back(i);
            e->redo.push_back({log_probs[i], labels[i], best_path[i], best_labels[i], best_scores[i], T[i], S[i], ld[i], i});
        }
    }
    // ---- tile tasks, sorted by first frame (then tile, then lattice): a tile's producer holds an earlier ticket ----
    if (n_tiled) {
        struct Key { int32_t t_in, tile, k; };
        std::vector<Key> keys;
        keys.reserve(n_tasks);
        std::vector<size_t> first_word(n, 0);   // progress word of tile 0 of descriptor k (word 0 = "nothing below")
        size_t w = 1;
        for (int32_t k = 0; k < n_tiled; ++k) {
            const Shape &p = sh[order[k]];
            first_word[k] = w;
            w += p.t_in.size();
            for (size_t b = 0; b < p.t_in.size(); ++b) keys.push_back({p.t_in[b], (int32_t)b, k});
        }
        std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
            if (a.t_in != b.t_in) return a.t_in < b.t_in;
            if (a.tile != b.tile) return a.tile < b.tile;
            return a.k < b.k;
        });
        std::vector<std::vector<size_t>> bound(n_tiled);   // halo region offset of the boundary above tile b
        for (int32_t k = 0; k < n_tiled; ++k) {
            const int32_t i = order[k];
            const Shape &p = sh[i];
            size_t o = cv[i].halo - off_halo;
            bound[k].resize(p.t_in.size());
            for (size_t b = 0; b < p.t_in.size(); ++b) {
                bound[k][b] = o;
                o += align_up((size_t)(p.t_end[b + 1 < p.t_in.size() ? b + 1 : b] - p.t_in[b] + 1) * 16);
            }
        }
        for (size_t j = 0; j < keys.size(); ++j) {
            const Key &key = keys[j];
            const Shape &p = sh[order[key.k]];
            const size_t b = (size_t)key.tile;
            ka::TileTask &tk = h_tasks[j];
            std::memset(&tk, 0, sizeof(tk));
            tk.lat = key.k;
            tk.tile = key.tile;
            tk.t_in = p.t_in[b];
            tk.t_end = p.t_end[b];
            // slot j of a boundary lies at its base + (j - t_in(lower tile)) * 16; the reader addresses from ITS t_in
            tk.halo_in = b == 0 ? 0 : (int64_t)(bound[key.k][b - 1] + (size_t)(p.t_in[b] - p.t_in[b - 1]) * 16);
            const bool has_above = b + 1 < p.t_in.size();
            tk.halo_out = (int64_t)bound[key.k][b];
            tk.fill_end = has_above ? p.t_end[b + 1] - 1 : 0;
            tk.below_end = b == 0 ? INT32_MAX : p.t_end[b - 1];
            tk.prog_in = b == 0 ? 0 : (int32_t)(first_word[key.k] + b - 1);
            tk.prog_out = (int32_t)(first_word[key.k] + b);
        }
    }

    // ---- copy in (host mode) ----
    if (mem == KA_MEM_HOST) {
        for (int32_t i = 0; i < n; ++i) {
            KA_HIP(hipMemcpy2DAsync(e->ws + cv[i].lp, (size_t)V * 4, log_probs[i], (size_t)ld[i] * 4, (size_t)V * 4,
                                    (size_t)sh[i].T, hipMemcpyHostToDevice, stream));
            if (sh[i].S > 0)
                KA_HIP(hipMemcpyAsync(e->ws + cv[i].lab, labels[i], (size_t)sh[i].S * 4, hipMemcpyHostToDevice, stream));
        }
    }
    ka::Lattice *d_lats = reinterpret_cast<ka::Lattice *>(e->ws + off_desc);
    int32_t *d_meta = reinterpret_cast<int32_t *>(e->ws + off_meta);
    KA_HIP(hipMemcpyAsync(d_lats, h_lats, (size_t)n * sizeof(ka::Lattice), hipMemcpyHostToDevice, stream));
    KA_HIP(hipMemsetAsync(d_meta, 0, (size_t)n * 16, stream));
    if (n_tiled) {
        KA_HIP(hipMemcpyAsync(e->ws + off_tasks, h_tasks, n_tasks * sizeof(ka::TileTask), hipMemcpyHostToDevice, stream));
        KA_HIP(hipMemsetAsync(e->ws + off_zero, 0, zero_bytes, stream));
        KA_HIP(hipMemsetD32Async((hipDeviceptr_t)(e->ws + off_prog), (int)ka::kTpProgDone, 1, stream));
        KA_HIP(hipMemsetD32Async((hipDeviceptr_t)(e->ws + off_halo), (int)0xff800000u, ninf_bytes / 4, stream));   // -inf packets
    }

    // ---- kernels ----
    // descriptors [0, n_tiled) tiled, [n_tiled, n_tiled + n_fast) one wavefront each, the rest generic
    const int32_t n_ring = n_tiled + n_fast;   // lattices whose checkpointed results backtrace_rc walks / exact kernels may redo
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[0], stream));
    hipLaunchKernelGGL(ka::prep_labels_kernel, dim3(n), dim3(256), 0, stream, d_lats, d_meta);
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[1], stream));
    // A mixed launch runs its two kernel forms side by side: the second one on the engine's own stream, forked from the
    // caller's stream behind the label preparation and joined to it again (events; nothing here blocks the host).
    auto fork = [&](int k) -> hipError_t {
        if (!e->aux) {
            hipError_t er = hipStreamCreateWithFlags(&e->aux, hipStreamNonBlocking);
            if (er != hipSuccess) return er;
        }
        hipError_t er = hipEventRecord(e->sync[k], stream);
        return er != hipSuccess ? er : hipStreamWaitEvent(e->aux, e->sync[k], 0);
    };
    auto join = [&](int k) -> hipError_t {
        hipError_t er = hipEventRecord(e->sync[k], e->aux);
        return er != hipSuccess ? er : hipStreamWaitEvent(stream, e->sync[k], 0);
    };
    const bool two_forward = n_tiled > 0 && n_fast > 0;
    if (two_forward) KA_HIP(fork(0));      // (before the tile kernel is enqueued: the second stream must not wait for it)
    Form form = kFormWaveExact;
    if (n_tiled > 0) {
        // 40 KB of LDS per tile workgroup = four workgroups per CU; when the launch has no more tiles than two per CU, 80 KB
        // keeps them at two per CU, i.e. (two wavefronts each) one wavefront per SIMD: two tiles whose wavefronts share a SIMD
        // run at 95-106 ns per frame instead of 55-62, and a chain runs at the pace of its slowest tile (cfg5's whole lattice,
        // 391 tiles alive for all 500 000 frames: profiles/r03_tile_stats_cfg5_full.txt)
        const unsigned grid = (unsigned)n_tasks;
        const unsigned lds = e->tile_lds ? (unsigned)e->tile_lds : ((int64_t)n_tasks <= (int64_t)e->n_simd / 2 && e->tile_waves == 2 ? 2u * ka::kTpLdsRequest : ka::kTpLdsRequest);
        // ka_engine_set_verify(1) (tests): every halo slot starts as a NaN pattern and a tile that consumes one reports KA_ERR_INTERNAL
        const int verify = e->verify;
        const bool gather = narrow && e->tile_gather != 0;
        if (verify & 1) {
            size_t lo_b = ~size_t(0), hi_b = 0;
            for (int32_t i = 0; i < n; ++i)
                if (sh[i].tiled && sh[i].halo_bytes) {
                    lo_b = std::min(lo_b, cv[i].halo);
                    hi_b = std::max(hi_b, cv[i].halo + sh[i].halo_bytes);
                }
            for (int32_t i = 0; i < n && hi_b > lo_b; ++i)
                if (sh[i].tiled && sh[i].halo_bytes)
                    KA_HIP(hipMemsetD32Async((hipDeviceptr_t)(e->ws + cv[i].halo), (int)ka::kTpSentinel, sh[i].halo_bytes / 4, stream));
        }
        const ka::TileTask *d_tasks = reinterpret_cast<const ka::TileTask *>(e->ws + off_tasks);
        char *d_halo = e->ws + off_halo;
        uint32_t *d_prog = reinterpret_cast<uint32_t *>(e->ws + off_prog);
        ka::TileAux *d_aux = reinterpret_cast<ka::TileAux *>(e->ws + off_aux);
        uint32_t *d_ticket = reinterpret_cast<uint32_t *>(e->ws + off_ticket);
        ka::TpStats *d_stats = reinterpret_cast<ka::TpStats *>(e->ws + off_stats);
        // staging mode: when every tiled lattice's rows are contiguous (row stride = V, V = 64 or 39, 16-byte aligned) a block
        // is copied as it lies in memory (1 KB per LDS-DMA instruction); otherwise row by row
        int pitch = ((V == 64 || V == 39) && max_move == 4) ? 4 * V : 0;   // 0: row by row
        for (int32_t k = 0; k < n_tiled && pitch; ++k) {
            const int32_t i = order[k];
            const bool contiguous = mem == KA_MEM_HOST || (ld[i] == V && ((uintptr_t)log_probs[i] & 15) == 0);
            if (!contiguous) pitch = 0;
        }
        // two wavefronts per tile (ka_tiled2.hpp: one computes, one feeds) unless the engine was told otherwise
#define KA_TP_LAUNCH(MM, PP, CC)                                                                                                                      \
    do {                                                                                                                                              \
        if (narrow && gather) {                                                                                                                       \
            const unsigned need = (unsigned)ka::TnLds<PP, CC, true>::kTotal;                                                                          \
            hipLaunchKernelGGL((ka::forward_tn_kernel<MM, PP, CC, true>), dim3(grid), dim3(192), std::max(need, (unsigned)e->tile_lds), stream, d_lats, \
                               d_tasks, (int)n_tasks, d_meta, d_halo, d_prog, d_aux, d_ticket, verify, d_stats);                                      \
        } else if (narrow)                                                                                                                            \
            hipLaunchKernelGGL((ka::forward_tn_kernel<MM, PP, CC, false>), dim3(grid), dim3(128), lds, stream, d_lats, d_tasks, (int)n_tasks, d_meta,  \
                               d_halo, d_prog, d_aux, d_ticket, verify, d_stats);                                                                     \
        else if (e->tile_waves == 2)                                                                                                                       \
            hipLaunchKernelGGL((ka::forward_tp2_kernel<MM, PP, CC>), dim3(grid), dim3(128), lds, stream, d_lats, d_tasks, (int)n_tasks, d_meta, d_halo, \
                               d_prog, d_aux, d_ticket, verify, d_stats);                                                                             \
        else                                                                                                                                          \
            hipLaunchKernelGGL((ka::forward_tp_kernel<MM, PP, CC>), dim3(grid), dim3(64), lds, stream, d_lats, d_tasks, (int)n_tasks, d_meta, d_halo,  \
                               d_prog, d_aux, d_ticket, verify, d_stats);                                                                             \
    } while (0)
        if (pitch == 256 && V == 64) KA_TP_LAUNCH(4, 256, true);
        else if (pitch == 156) KA_TP_LAUNCH(4, 156, true);
        else switch (max_move) {
        case 1: KA_TP_LAUNCH(1, 256, false); break;
        case 2: KA_TP_LAUNCH(2, 256, false); break;
        case 3: KA_TP_LAUNCH(3, 256, false); break;
        default: KA_TP_LAUNCH(4, 256, false); break;
        }
#undef KA_TP_LAUNCH
        form = kFormWaveCheckpointed;
    }
    if (n_fast > 0) {
        const bool wg = e->mode == KA_MODE_WORKGROUP;
        form = wg ? kFormWorkgroup : (e->mode == KA_MODE_WAVE_EXACT ? kFormWaveExact : kFormWaveCheckpointed);
        // backtrace_rc_kernel keeps 34*T in 32 bits (descriptors are sorted longest first)
        if (form == kFormWaveCheckpointed && sh[order[n_tiled]].T >= (int64_t(1) << 26)) form = kFormWaveExact;
        hipStream_t sw = two_forward ? e->aux : stream;
        switch (max_move) {
        case 1: launch_forward<1>(d_lats + n_tiled, n_fast, d_meta, sw, form); break;
        case 2: launch_forward<2>(d_lats + n_tiled, n_fast, d_meta, sw, form); break;
        case 3: launch_forward<3>(d_lats + n_tiled, n_fast, d_meta, sw, form); break;
        default: launch_forward<4>(d_lats + n_tiled, n_fast, d_meta, sw, form); break;
        }
    }
    if (n_tiled > 0) {
        // tiled lattices that the scores-only form declined (non-finite log-probs) and that fit the one-wavefront ring
        // are redone by the exact kernels (kFlagExact; wider ones get kFlagDeclined and are handed to the generic kernels
        // by ka_batch_finish)
        switch (max_move) {
        case 1: launch_forward_flagged<1>(d_lats, n_tiled, d_meta, stream); break;
        case 2: launch_forward_flagged<2>(d_lats, n_tiled, d_meta, stream); break;
        case 3: launch_forward_flagged<3>(d_lats, n_tiled, d_meta, stream); break;
        default: launch_forward_flagged<4>(d_lats, n_tiled, d_meta, stream); break;
        }
    }
    if (two_forward) KA_HIP(join(1));
    if (n > n_ring)
        hipLaunchKernelGGL(ka::forward_generic_kernel, dim3(n - n_ring), dim3(256), 0, stream, d_lats + n_ring, d_meta);
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[2], stream));
    // backtrace: checkpointed results (tiled + checkpointed one-wavefront form) by recomputation, stored back-pointers by
    // the walk; [rc_lo, rc_hi) = descriptors whose outputs backtrace_rc writes itself
    const int32_t rc_lo = 0, rc_hi = n_tiled + (form == kFormWaveCheckpointed ? n_fast : 0);
    if (rc_hi > rc_lo) {
        // the chunk-parallel lattices of the range (Lattice::par; their chunks are numbered consecutively) and the others
        int64_t total_chunks = 0, max_seg = 1, max_sup = 1, max_w = 1;
        int32_t n_par = 0;
        for (int32_t k = rc_lo; k < rc_hi; ++k) {
            const Shape &p = sh[order[k]];
            if (!p.par_bt) continue;
            ++n_par;
            total_chunks += chunks_of_T(p.T);
            max_seg = std::max<int64_t>(max_seg, (p.W + 7 + ka::kCmOut - 1) / ka::kCmOut);
            max_sup = std::max<int64_t>(max_sup, supers_of_T(p.T));
            max_w = std::max<int64_t>(max_w, p.W);
        }
        const int nl = rc_hi - rc_lo;
        const bool two_backtraces = n_par > 0 && n_par < nl;
        if (n_par < nl) {      // one wavefront per lattice, chunk after chunk (skips the chunk-parallel ones)
            if (two_backtraces) KA_HIP(fork(2));
            hipStream_t ss = two_backtraces ? e->aux : stream;
            // (the gather form is opt-in: 27.2 -> 26.1 ms for 8192 lattices alone on the GPU, nothing with four launches in flight,
            //  and 47 GB more HBM traffic per step by the counters - DESIGN.md section 8)
            const bool gather = e->rc_gather == 1;
            switch (max_move) {
            case 1: launch_backtrace_rc<1>(d_lats + rc_lo, nl, d_meta, ss, gather); break;
            case 2: launch_backtrace_rc<2>(d_lats + rc_lo, nl, d_meta, ss, gather); break;
            case 3: launch_backtrace_rc<3>(d_lats + rc_lo, nl, d_meta, ss, gather); break;
            default: launch_backtrace_rc<4>(d_lats + rc_lo, nl, d_meta, ss, gather); break;
            }
        }
        if (n_par > 0) {
            const unsigned gc = (unsigned)total_chunks, gs = (unsigned)max_seg;
            switch (max_move) {
            case 1: launch_parallel_bt<1>(d_lats + rc_lo, nl, d_meta, stream, gc, gs, (unsigned)max_sup, (unsigned)max_w); break;
            case 2: launch_parallel_bt<2>(d_lats + rc_lo, nl, d_meta, stream, gc, gs, (unsigned)max_sup, (unsigned)max_w); break;
            case 3: launch_parallel_bt<3>(d_lats + rc_lo, nl, d_meta, stream, gc, gs, (unsigned)max_sup, (unsigned)max_w); break;
            default: launch_parallel_bt<4>(d_lats + rc_lo, nl, d_meta, stream, gc, gs, (unsigned)max_sup, (unsigned)max_w); break;
            }
        }
        if (two_backtraces) KA_HIP(join(3));
        hipLaunchKernelGGL(ka::backtrace_w16_kernel, dim3(rc_hi - rc_lo), dim3(64), 0, stream, d_lats + rc_lo, d_meta, 1);
    }
    if (n_ring > rc_hi)
        hipLaunchKernelGGL(ka::backtrace_w16_kernel, dim3(n_ring - rc_hi), dim3(64), 0, stream, d_lats + rc_hi, d_meta, 0);
    if (n > n_ring)
        hipLaunchKernelGGL(ka::backtrace_generic_kernel, dim3(n - n_ring), dim3(64), 0, stream, d_lats + n_ring, d_meta);
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[3], stream));
    {
        int64_t t_max = 1;
        for (int32_t i = 0; i < n; ++i) t_max = std::max<int64_t>(t_max, sh[i].T);
        const unsigned gx = (unsigned)((t_max + 1023) / 1024);
        for (int32_t y0 = 0; y0 < rc_hi; y0 += 65535) {     // only what the exact kernels redid
            const unsigned gy = (unsigned)std::min<int32_t>(65535, rc_hi - y0);
            hipLaunchKernelGGL(ka::gather_outputs_kernel, dim3(1, gy), dim3(256), 0, stream, d_lats + y0, d_meta, 1);
        }
        for (int32_t y0 = rc_hi; y0 < n; y0 += 65535) {   // grid.y limit
            const unsigned gy = (unsigned)std::min<int32_t>(65535, n - y0);
            hipLaunchKernelGGL(ka::gather_outputs_kernel, dim3(gx, gy), dim3(256), 0, stream, d_lats + y0, d_meta, 0);
        }
    }
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[4], stream));
    KA_HIP(hipGetLastError());

    // ---- copy out ----
    KA_HIP(hipMemcpyAsync(e->h_meta, d_meta, (size_t)n * 16, hipMemcpyDeviceToHost, stream));
    if (mem == KA_MEM_HOST) {
        for (int32_t i = 0; i < n; ++i) {
            const size_t b = (size_t)sh[i].T * 4;
            KA_HIP(hipMemcpyAsync(best_path[i], e->ws + cv[i].path, b, hipMemcpyDeviceToHost, stream));
            KA_HIP(hipMemcpyAsync(best_labels[i], e->ws + cv[i].labo, b, hipMemcpyDeviceToHost, stream));
            KA_HIP(hipMemcpyAsync(best_scores[i], e->ws + cv[i].sco, b, hipMemcpyDeviceToHost, stream));
        }
    }
    e->pending = true;
    return KA_OK;
}

int ka_ctc_best_path_batch_enqueue_f32(ka_engine *e, int32_t n, const float *const *log_probs, const int64_t *T,
                                       int32_t V, const int64_t *ld, const int32_t *const *labels, const int64_t *S,
                                       int32_t beam_size, int32_t max_move, int32_t *const *best_path,
                                       int32_t *const *best_labels, float *const *best_scores, void *stream)
{
    return enqueue_impl(e, n, log_probs, T, V, ld, labels, S, beam_size, max_move, best_path, best_labels,
                        best_scores, KA_MEM_DEVICE, (hipStream_t)stream, false, nullptr);
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
    // A lattice in the tiled form whose band is wider than the exact kernels' ring and whose log-probs are not all
    // finite (flag set by the forward kernel) has no result yet: the scores-only forms are valid only while "live" and
    // "score > -inf" coincide.  The reference answers such input (align.py:67-85 tracks the live set explicitly), so in
    // KA_MODE_AUTO those lattices are handed to the generic kernels now, into the caller's buffers; an explicit
    // KA_MODE_TILED reports KA_ERR_NONFINITE.
    const int32_t n_all = e->n_last;
    std::vector<int32_t> meta(e->h_meta, e->h_meta + 4 * (size_t)n_all);
    std::vector<ka_engine::Redo> again;
    for (const ka_engine::Redo &r : e->redo) {
        int32_t *m = meta.data() + 4 * (size_t)r.idx;
        if (m[0] != KA_OK || !(m[2] & ka::kFlagDeclined)) continue;
        if (e->mode == KA_MODE_TILED) m[0] = KA_ERR_NONFINITE;
        else again.push_back(r);
    }
    if (!again.empty()) {
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
        const bool prof = e->profiling;
        e->profiling = false;      // the events keep the times of the batch itself
        const hipStream_t stream = e->stream_last;
        int rc = enqueue_impl(e, m, lp.data(), T.data(), e->last_V, ld.data(), lab.data(), S.data(), e->last_beam, e->last_max_move,
                              path.data(), lab_out.data(), sc.data(), e->last_mem, stream, /*force_generic=*/true, nullptr);
        e->profiling = prof;
        e->pending = false;
        if (rc != KA_OK) return rc;
        KA_HIP(hipStreamSynchronize(stream));
        for (int32_t j = 0; j < m; ++j) std::memcpy(meta.data() + 4 * (size_t)again[j].idx, e->h_meta + 4 * (size_t)j, 16);
        e->n_last = n_all;
        e->redo.clear();
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
    int rc = enqueue_impl(e, n, log_probs, T, V, ld, labels, S, beam_size, max_move, best_path, best_labels,
                          best_scores, mem, (hipStream_t)stream, false, nullptr);
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

int ka_debug_plan_tiles(int64_t T, int64_t S, int32_t V, int32_t beam_size, int32_t max_move, int32_t *t_in, int32_t *t_end,
                        int32_t max_tiles, int64_t *checkpoint_pitch)
{
    Shape sh;
    if (max_tiles < 0 || (max_tiles > 0 && (!t_in || !t_end)) || !shape_of(T, S, V, beam_size, max_move, sh))
        return fail(KA_ERR_BAD_ARGS, "ka_debug_plan_tiles: bad arguments");
    plan_tiles(sh, V, beam_size, max_move);
    if (!sh.tileable) return 0;
    for (size_t b = 0; b < sh.t_in.size() && b < (size_t)max_tiles; ++b) {
        t_in[b] = sh.t_in[b];
        t_end[b] = sh.t_end[b];
    }
    if (checkpoint_pitch) *checkpoint_pitch = (int64_t)sh.ck_pitch;
    return (int)sh.t_in.size();
}

int ka_debug_plan_tiles_width(int64_t T, int64_t S, int32_t V, int32_t beam_size, int32_t max_move, int32_t positions, int32_t *t_in, int32_t *t_end,
                              int32_t max_tiles, int64_t *checkpoint_pitch)
{
    Shape sh;
    if (max_tiles < 0 || (max_tiles > 0 && (!t_in || !t_end)) || (positions != ka::kTnTile && positions != ka::kTpTile) ||
        !shape_of(T, S, V, beam_size, max_move, sh))
        return fail(KA_ERR_BAD_ARGS, "ka_debug_plan_tiles_width: bad arguments");
    plan_tiles(sh, V, beam_size, max_move, positions);
    if (!sh.tileable) return 0;
    for (size_t b = 0; b < sh.t_in.size() && b < (size_t)max_tiles; ++b) {
        t_in[b] = sh.t_in[b];
        t_end[b] = sh.t_end[b];
    }
    if (checkpoint_pitch) *checkpoint_pitch = (int64_t)sh.ck_pitch;
    return (int)sh.t_in.size();
}

int ka_debug_tile_width_choice(const int64_t *T, const int64_t *S, int32_t n, int32_t V, int32_t beam_size, int32_t max_move, int32_t n_simd)
{
    if (n < 0 || (n > 0 && (!T || !S)) || n_simd < 4) return fail(KA_ERR_BAD_ARGS, "ka_debug_tile_width_choice: bad arguments");
    std::vector<Shape> plans(n);
    for (int32_t i = 0; i < n; ++i) {
        if (!shape_of(T[i], S[i], V, beam_size, max_move, plans[i])) return fail(KA_ERR_BAD_ARGS, "ka_debug_tile_width_choice: bad shape");
        plan_tiles(plans[i], V, beam_size, max_move, ka::kTpTile);
        if (!plans[i].tileable) return 0;
        plan_tiles(plans[i], V, beam_size, max_move, ka::kTnTile);
    }
    return narrow_tiles_pay(plans, V, max_move, n_simd, 0) ? ka::kTnTile : ka::kTpTile;
}

int ka_debug_auto_split(const int64_t *T, int32_t n, int32_t tiles_alive, int32_t n_simd, int32_t *n_tiled, int32_t *n_parallel)
{
    if (n < 0 || (n > 0 && !T) || tiles_alive < 1 || n_simd < 1 || !n_tiled || !n_parallel)
        return fail(KA_ERR_BAD_ARGS, "ka_debug_auto_split: bad arguments");
    std::vector<int64_t> Ts(T, T + n);
    std::sort(Ts.begin(), Ts.end(), [](int64_t a, int64_t b) { return a > b; });
    std::vector<int32_t> alive((size_t)n, tiles_alive);
    *n_tiled = auto_split_forward(Ts, alive, n_simd);
    *n_parallel = auto_split_backtrace(Ts, n_simd);
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
    }
    return (int)n;
}

int ka_log_softmax_f32(const float *logits, float *log_probs, int64_t T, int32_t V, int64_t ld_in, int64_t ld_out,
                       void *stream)
{
    if (!logits || !log_probs || T < 0 || V < 1 || ld_in < V || ld_out < V) return fail(KA_ERR_BAD_ARGS, "ka_log_softmax_f32: bad arguments");
    if (T == 0) return KA_OK;
    const int64_t blocks = (T + 3) / 4;
    if (blocks > 0x7fffffff) return fail(KA_ERR_BAD_ARGS, "ka_log_softmax_f32: T too large");
    hipLaunchKernelGGL(ka::log_softmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, log_probs,
                       T, V, ld_in, ld_out);
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
    const int64_t blocks = ((int64_t)n * H + 255) / 256;
    hipLaunchKernelGGL(ka::lstm_step_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, (hipStream_t)stream, gin, ldg, rec,
                       rec_dir_stride, c, h, state_dir_stride, out, ldo, rows, rows_dir_stride, n, H);
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
    hipLaunchKernelGGL(ka::lstm_layer_kernel<false>, dim3(2u * (unsigned)((nseq + ka::kLstmTile - 1) / ka::kLstmTile)), dim3(256), 0, (hipStream_t)stream, gin, ldg, w_hh,
                       out, ldo, seq_off, seq_len, nseq, (const float *)nullptr, (const float *)nullptr);
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
    hipLaunchKernelGGL(ka::lstm_layer_kernel<true>, dim3(2u * (unsigned)((nseq + ka::kLstmTile - 1) / ka::kLstmTile)), dim3(256), 0, (hipStream_t)stream, x, ldx, w_hh,
                       out, ldo, seq_off, seq_len, nseq, w_ih, bias);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_window_energy_f32(const float *x, int64_t n_windows, int32_t window, float *out, void *stream)
{
    if (!x || !out || n_windows < 0) return fail(KA_ERR_BAD_ARGS, "ka_window_energy_f32: bad arguments");
    if (window != 256) return fail(KA_ERR_BAD_ARGS, "ka_window_energy_f32: the summation order is NumPy's for windows of 256 samples only");
    if (n_windows == 0) return KA_OK;
    const int64_t blocks = (n_windows + 15) / 16;
    if (blocks > 0x7fffffff) return fail(KA_ERR_BAD_ARGS, "ka_window_energy_f32: too many windows");
    hipLaunchKernelGGL(ka::window_energy_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n_windows, out);
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
    const unsigned gx = (unsigned)std::min<int64_t>(max_frames, 4096);
    hipLaunchKernelGGL(ka::stft_frames_kernel, dim3(gx, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, y, seg_start, seg_len,
                       frame_off, n_fft, hop, window, frames, ld);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_power_f32(const float *reim, int64_t ld_in, float *power, int64_t ld_out, int64_t n, int32_t nf, void *stream)
{
    if (!reim || !power || n < 0 || nf < 1 || ld_in < 2 * (int64_t)nf || ld_out < nf) return fail(KA_ERR_BAD_ARGS, "ka_power_f32: bad arguments");
    if (n == 0) return KA_OK;
    const unsigned blocks = (unsigned)std::min<int64_t>((n * nf + 255) / 256, 65536);
    hipLaunchKernelGGL(ka::power_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reim, ld_in, power, ld_out, n, nf);
    KA_HIP(hipGetLastError());
    return KA_OK;
}

int ka_power_to_db_f32(float *x, int64_t ld, int32_t cols, const int64_t *frame_off, int32_t nseg, int64_t max_frames, float top_db,
                       float *segmax, void *stream)
{
    if (!x || !frame_off || !segmax || nseg < 0 || cols < 1 || ld < cols || max_frames < 0) return fail(KA_ERR_BAD_ARGS, "ka_power_to_db_f32: bad arguments");
    if (nseg == 0 || max_frames == 0) return KA_OK;
    if (nseg > 65535) return fail(KA_ERR_BAD_ARGS, "ka_power_to_db_f32: more than 65535 segments in one call");
    const unsigned gx = (unsigned)std::min<int64_t>((max_frames * cols + 255) / 256, 256);
    hipLaunchKernelGGL(ka::power_to_db_kernel, dim3(gx, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, x, ld, cols, frame_off, segmax);
    hipLaunchKernelGGL(ka::db_floor_kernel, dim3(gx, (unsigned)nseg), dim3(256), 0, (hipStream_t)stream, x, ld, cols, frame_off, segmax, top_db);
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
    for (int32_t y0 = 0; y0 < n; y0 += 65535) {
        const unsigned gy = (unsigned)std::min<int32_t>(65535, n - y0);
        hipLaunchKernelGGL(ka::hash_logprobs_kernel, dim3(blocks, gy), dim3(256), 0, (hipStream_t)stream,
                           dev_log_probs + (size_t)y0 * (size_t)lattice_stride, T, V, ld, seed0 + (uint64_t)y0, lattice_stride);
    }
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
    for (int32_t y0 = 0; y0 < n; y0 += 65535) {
        const unsigned gy = (unsigned)std::min<int32_t>(65535, n - y0);
        hipLaunchKernelGGL(ka::hash_labels_kernel, dim3(blocks, gy), dim3(256), 0, (hipStream_t)stream,
                           dev_labels + (size_t)y0 * (size_t)lattice_stride, S, V, seed0 + (uint64_t)y0, lattice_stride);
    }
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
