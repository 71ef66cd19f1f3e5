// ka_engine.hip — host side of the C ABI declared in include/kokoro_align_amd.h.
//
// Builds one descriptor per lattice, carves the device workspace (padded labels,
// back-pointer storage), launches prep -> forward DP -> backtrace on the caller's stream and
// reports per-lattice status.  No torch, no oracle, no CPU fallback: if HIP fails the call
// fails.
#include "../../include/kokoro_align_amd.h"
#include "ka_kernels.hpp"

#include <algorithm>
#include <cstdio>
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
constexpr int32_t kAutoWorkgroupMaxLattices = 0;

struct Shape {
    int64_t T, S, L, W;
    int32_t labx_len;
    bool fast;
};

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

// device bytes a lattice needs besides the caller's buffers
size_t lattice_ws_bytes(const Shape &sh)
{
    size_t b = align_up((size_t)sh.labx_len * 4);
    if (sh.fast) {
        b += align_up((((size_t)sh.T + 3) / 4) * 1024);
    } else {
        b += align_up((size_t)sh.T * (size_t)sh.W);
        b += align_up((size_t)sh.L * 2 * sizeof(float) + (size_t)sh.L * 2);
    }
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

template <int M>
void launch_backtrace_rc(const ka::Lattice *d_lats, int n, int32_t *d_meta, hipStream_t s)
{
    hipLaunchKernelGGL((ka::backtrace_rc_kernel<M, false>), dim3(n), dim3(64), 0, s, d_lats, d_meta);
    hipLaunchKernelGGL((ka::backtrace_rc_kernel<M, true>), dim3(n), dim3(64), 0, s, d_lats, d_meta);
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
    for (int i = 0; i < 5; ++i) {
        hipError_t er = hipEventCreate(&e->ev[i]);
        if (er != hipSuccess) {
            delete e;
            return fail(KA_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(er));
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
    delete e;
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
    for (int32_t i = 0; i < n; ++i) {
        Shape sh;
        if (!shape_of(T[i], S[i], V, beam_size, max_move, sh)) return 0;
        total += lattice_ws_bytes(sh);
    }
    return total;
}

int ka_engine_set_mode(ka_engine *e, int32_t mode)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (mode != KA_MODE_AUTO && mode != KA_MODE_WAVE && mode != KA_MODE_WORKGROUP && mode != KA_MODE_WAVE_EXACT)
        return fail(KA_ERR_BAD_ARGS, "ka_engine_set_mode: unknown mode");
    e->mode = mode;
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

static int enqueue_impl(ka_engine *e, int32_t n, const float *const *log_probs, const int64_t *T, int32_t V,
                        const int64_t *ld, const int32_t *const *labels, const int64_t *S, int32_t beam_size,
                        int32_t max_move, int32_t *const *best_path, int32_t *const *best_labels,
                        float *const *best_scores, int32_t mem, hipStream_t stream)
{
    if (!e) return fail(KA_ERR_BAD_ARGS, "engine is NULL");
    if (e->pending) return fail(KA_ERR_BAD_ARGS, "a batch is already enqueued: call ka_batch_finish first");
    if (n < 0 || (n > 0 && (!log_probs || !T || !ld || !labels || !S || !best_path || !best_labels || !best_scores)))
        return fail(KA_ERR_BAD_ARGS, "batch: NULL array argument");
    if (mem != KA_MEM_HOST && mem != KA_MEM_DEVICE) return fail(KA_ERR_BAD_ARGS, "mem must be KA_MEM_HOST or KA_MEM_DEVICE");
    DeviceGuard guard;
    KA_HIP(guard.enter(e->device));
    e->n_last = n;
    e->stream_last = stream;
    e->have_times = false;
    if (n == 0) {
        e->pending = true;
        return KA_OK;
    }

    std::vector<Shape> sh(n);
    for (int32_t i = 0; i < n; ++i) {
        if (!shape_of(T[i], S[i], V, beam_size, max_move, sh[i]) || ld[i] < V)
            return fail(KA_ERR_BAD_ARGS, "lattice " + std::to_string(i) + ": unsupported T/S/V/ld/beam_size/max_move");
        if (!log_probs[i] || !best_path[i] || !best_labels[i] || !best_scores[i] || (S[i] > 0 && !labels[i]))
            return fail(KA_ERR_BAD_ARGS, "lattice " + std::to_string(i) + ": NULL buffer");
    }

    // ---- carve the workspace ----
    size_t off = 0;
    const size_t off_desc = off;
    off += align_up((size_t)n * sizeof(ka::Lattice));
    const size_t off_meta = off;
    off += align_up((size_t)n * 16);
    struct Carve { size_t labx, bp, col, lp, lab, path, labo, sco; };
    std::vector<Carve> cv(n);
    for (int32_t i = 0; i < n; ++i) {
        cv[i].labx = off;
        off += align_up((size_t)sh[i].labx_len * 4);
        cv[i].bp = off;
        off += sh[i].fast ? align_up((((size_t)sh[i].T + 3) / 4) * 1024) : align_up((size_t)sh[i].T * (size_t)sh[i].W);
        cv[i].col = off;
        if (!sh[i].fast) off += align_up((size_t)sh[i].L * 2 * sizeof(float) + (size_t)sh[i].L * 2);
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
    int rc = ensure_ws(e, off);
    if (rc != KA_OK) return rc;
    rc = ensure_pin(e, align_up((size_t)n * sizeof(ka::Lattice)) + align_up((size_t)n * 16));
    if (rc != KA_OK) return rc;

    // ---- descriptors: w16 lattices first (longest first: short tail), then generic ones ----
    std::vector<int32_t> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        if (sh[a].fast != sh[b].fast) return sh[a].fast;
        return sh[a].T > sh[b].T;
    });
    int32_t n_fast = 0;
    for (int32_t i = 0; i < n; ++i) n_fast += sh[i].fast ? 1 : 0;
    ka::Lattice *h_lats = reinterpret_cast<ka::Lattice *>(e->pin);
    e->h_meta = reinterpret_cast<int32_t *>(e->pin + align_up((size_t)n * sizeof(ka::Lattice)));
    for (int32_t k = 0; k < n; ++k) {
        const int32_t i = order[k];
        ka::Lattice &d = h_lats[k];
        std::memset(&d, 0, sizeof(d));
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

    // ---- kernels ----
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[0], stream));
    hipLaunchKernelGGL(ka::prep_labels_kernel, dim3(n), dim3(256), 0, stream, d_lats, d_meta);
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[1], stream));
    Form form = kFormWaveExact;
    if (n_fast > 0) {
        // few lattices: 4 wavefronts per lattice (per-frame latency); many: 1 wavefront per lattice (throughput)
        const bool wg = e->mode == KA_MODE_WORKGROUP || (e->mode == KA_MODE_AUTO && n_fast <= kAutoWorkgroupMaxLattices);
        form = wg ? kFormWorkgroup : (e->mode == KA_MODE_WAVE_EXACT ? kFormWaveExact : kFormWaveCheckpointed);
        // backtrace_rc_kernel keeps 34*T in 32 bits (descriptors are sorted longest first)
        if (form == kFormWaveCheckpointed && sh[order[0]].T >= (int64_t(1) << 26)) form = kFormWaveExact;
        switch (max_move) {
        case 1: launch_forward<1>(d_lats, n_fast, d_meta, stream, form); break;
        case 2: launch_forward<2>(d_lats, n_fast, d_meta, stream, form); break;
        case 3: launch_forward<3>(d_lats, n_fast, d_meta, stream, form); break;
        default: launch_forward<4>(d_lats, n_fast, d_meta, stream, form); break;
        }
    }
    if (n > n_fast)
        hipLaunchKernelGGL(ka::forward_generic_kernel, dim3(n - n_fast), dim3(256), 0, stream, d_lats + n_fast, d_meta);
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[2], stream));
    const int only_flagged = form == kFormWaveCheckpointed ? 1 : 0;
    if (n_fast > 0) {
        if (form == kFormWaveCheckpointed) {
            switch (max_move) {
            case 1: launch_backtrace_rc<1>(d_lats, n_fast, d_meta, stream); break;
            case 2: launch_backtrace_rc<2>(d_lats, n_fast, d_meta, stream); break;
            case 3: launch_backtrace_rc<3>(d_lats, n_fast, d_meta, stream); break;
            default: launch_backtrace_rc<4>(d_lats, n_fast, d_meta, stream); break;
            }
        }
        hipLaunchKernelGGL(ka::backtrace_w16_kernel, dim3(n_fast), dim3(64), 0, stream, d_lats, d_meta, only_flagged);
    }
    if (n > n_fast)
        hipLaunchKernelGGL(ka::backtrace_generic_kernel, dim3(n - n_fast), dim3(64), 0, stream, d_lats + n_fast, d_meta);
    if (e->profiling) KA_HIP(hipEventRecord(e->ev[3], stream));
    {
        // descriptors are sorted fast-first: [0, n_fast) may already have their outputs (checkpointed form)
        int64_t t_max = 1;
        for (int32_t i = 0; i < n; ++i) t_max = std::max<int64_t>(t_max, sh[i].T);
        const unsigned gx = (unsigned)((t_max + 1023) / 1024);
        const int32_t n_own = only_flagged ? n_fast : 0;   // lattices whose outputs the backtrace kernel wrote itself
        for (int32_t y0 = 0; y0 < n_own; y0 += 65535) {
            const unsigned gy = (unsigned)std::min<int32_t>(65535, n_own - y0);
            hipLaunchKernelGGL(ka::gather_outputs_kernel, dim3(1, gy), dim3(256), 0, stream, d_lats + y0, d_meta, 1);
        }
        for (int32_t y0 = n_own; y0 < n; y0 += 65535) {   // grid.y limit
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
                        best_scores, KA_MEM_DEVICE, (hipStream_t)stream);
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
    int first_bad = KA_OK;
    for (int32_t i = 0; i < e->n_last; ++i) {
        const int32_t *m = e->h_meta + 4 * (size_t)i;
        if (status) status[i] = m[0];
        if (total_score) std::memcpy(&total_score[i], &m[3], 4);
        if (m[0] != KA_OK && first_bad == KA_OK) {
            first_bad = m[0];
            g_err = "lattice " + std::to_string(i) + (m[0] == KA_ERR_EMPTY_BEAM ? ": no live state in the last frame (empty beam)"
                                                      : m[0] == KA_ERR_BAD_LABEL ? ": label outside [0, V)"
                                                      : m[0] == KA_ERR_NAN       ? ": a log-prob is NaN"
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
                          best_scores, mem, (hipStream_t)stream);
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
    hipLaunchKernelGGL(ka::lstm_layer_kernel, dim3(2u * (unsigned)((nseq + ka::kLstmTile - 1) / ka::kLstmTile)), dim3(256), 0, (hipStream_t)stream, gin, ldg, w_hh,
                       out, ldo, seq_off, seq_len, nseq);
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
