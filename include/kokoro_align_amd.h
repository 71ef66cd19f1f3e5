/*
 * kokoro_align_amd.h — C ABI of the MI355X (gfx950) CTC forced-alignment hot path.
 *
 * Drop-in boundary for the alignment step of kaiidams/Kokoro-Align.  The reference has no
 * FFI layer: its boundary is three Python functions in kokoro_align/align.py.  Each entry
 * point below names the reference interface it replaces (file:line); the Python mirror
 * that keeps the reference's signatures lives in kokoro-align_amd/align.py and binds these
 * symbols with ctypes.
 *
 * Plain pointers and sizes only; no torch types.  Device pointers may come from any
 * allocator (hipMalloc, a torch tensor's data_ptr(), ...).  The library never retains a
 * caller pointer after a call returns.
 *
 * Thread-safety: one ka_engine per host thread / stream; distinct engines are independent.
 */
#ifndef KOKORO_ALIGN_AMD_H
#define KOKORO_ALIGN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KA_VERSION 100 /* 0.1.0 */

/* status codes (per call and per lattice) */
#define KA_OK 0
#define KA_ERR_EMPTY_BEAM (-1) /* no live state in the last frame: reference raises ValueError, align.py:101 */
#define KA_ERR_BAD_ARGS (-2)
#define KA_ERR_HIP (-3)       /* HIP runtime error, text in ka_last_error() */
#define KA_ERR_NOMEM (-4)
#define KA_ERR_BAD_LABEL (-5) /* label outside [0,V): reference raises IndexError at align.py:77 */
#define KA_ERR_NONFINITE (-7) /* explicit KA_MODE_TILED only: an infinity among the log-probs of a lattice whose band is wider
                                 than 1009 positions.  KA_MODE_AUTO answers such lattices (the exact kernels for bands up to
                                 1009, the generic kernels above): -inf is legal input, as in the reference */
#define KA_ERR_INTERNAL (-8)  /* tiled form: a tile's hand-off timed out (an internal error, never an input condition) */
#define KA_ERR_NAN (-6)       /* a log-prob is NaN: the reference's np.argmax treats NaN as the maximum (align.py:83); that
                                 is not reproduced - the lattice is rejected (fast path, V <= 64, band <= 1009 or tiled form) */

/* where the caller's buffers live */
#define KA_MEM_HOST 0
#define KA_MEM_DEVICE 1

typedef struct ka_engine ka_engine;

int32_t ka_version(void);
/* message of the last error on the calling thread ("" if none) */
const char *ka_last_error(void);

/* An engine owns the device workspace (back-pointer storage, padded labels, descriptors),
 * pinned staging and timing events for ONE device.  device = HIP ordinal. */
int ka_engine_create(int32_t device, ka_engine **out);
void ka_engine_destroy(ka_engine *e);
/* A HIP stream (hipStream_t, non-blocking) for launches that should run beside others: one engine + one such stream per
 * host thread lets the forward pass of one launch overlap the backtrace of another (kokoro_align_amd.streams).  Streams
 * created one after the other land on different hardware queues while the runtime has any to spare (4 per device). */
int ka_stream_create(int32_t device, void **stream);
int ka_stream_destroy(int32_t device, void *stream);
/* pre-size the workspace so that later calls do not allocate (optional) */
int ka_engine_reserve(ka_engine *e, size_t workspace_bytes);
/* device-workspace bytes one batch call needs at most, whatever the engine's mode (every lattice priced in its most
 * expensive form: tiled with chunk-parallel backtrace); ka_engine_workspace_bytes gives the exact figure for an engine */
size_t ka_workspace_bytes(int32_t n, const int64_t *T, const int64_t *S, int32_t V,
                          int32_t beam_size, int32_t max_move);

/* ... exactly what THIS engine (its mode, backtrace and device) will carve for such a call: the same planning code as the
 * call itself, nothing is launched.  ka_workspace_bytes above is an upper bound over all modes. */
size_t ka_engine_workspace_bytes(ka_engine *e, int32_t n, const int64_t *T, const int64_t *S, int32_t V, int32_t beam_size,
                                 int32_t max_move, int32_t mem);

/*
 * ctc_best_path(log_probs, labels, beam_size=1000, max_move=4) -> (best_path, best_labels, best_scores)
 * replaces kokoro_align/align.py:43-109 (DP align.py:62-93, backtrace :21-40,:99-102, gathers :105-107).
 *
 *   log_probs  [T, V] float32, row stride ld (elements)
 *   labels     [S] int32, values in [0, V)   (un-expanded transcript; blanks are inserted here)
 *   best_path  [T] int32  positions in the blank-expanded label sequence (0 .. 2S)
 *   best_labels[T] int32, best_scores [T] float32 (per-frame emission of the chosen label)
 *   total_score  optional (may be NULL): cumulative float32 score of the terminal state
 *   mem        KA_MEM_HOST: all pointers are host memory (the call copies in and out)
 *              KA_MEM_DEVICE: all pointers except total_score are device memory
 *   stream     hipStream_t to launch on (NULL = default stream).  The call returns after the
 *              outputs are valid (it synchronises the stream).
 * Returns KA_OK, KA_ERR_EMPTY_BEAM (-> ValueError), KA_ERR_BAD_LABEL, KA_ERR_BAD_ARGS, KA_ERR_HIP.
 */
int ka_ctc_best_path_f32(ka_engine *e, const float *log_probs, int64_t T, int32_t V, int64_t ld,
                         const int32_t *labels, int64_t S, int32_t beam_size, int32_t max_move,
                         int32_t *best_path, int32_t *best_labels, float *best_scores,
                         float *total_score, int32_t mem, void *stream);

/*
 * The same over n independent lattices (one per audio file; the reference loops files
 * sequentially, run_example.py:248-254).  Arrays of n pointers / sizes are HOST arrays;
 * the pointed-to buffers live where `mem` says.  status[n] and total_score[n] are host
 * arrays (either may be NULL).  Returns KA_OK if every lattice succeeded, otherwise the
 * status of the first lattice that failed; the other lattices' outputs are still valid.
 */
int ka_ctc_best_path_batch_f32(ka_engine *e, int32_t n, const float *const *log_probs,
                               const int64_t *T, int32_t V, const int64_t *ld,
                               const int32_t *const *labels, const int64_t *S,
                               int32_t beam_size, int32_t max_move, int32_t *const *best_path,
                               int32_t *const *best_labels, float *const *best_scores,
                               float *total_score, int32_t *status, int32_t mem, void *stream);

/* Split form for device-resident batches: enqueue launches everything on `stream` without a
 * host sync; finish synchronises and fetches per-lattice status / total scores. */
int ka_ctc_best_path_batch_enqueue_f32(ka_engine *e, int32_t n, const float *const *log_probs,
                                       const int64_t *T, int32_t V, const int64_t *ld,
                                       const int32_t *const *labels, const int64_t *S,
                                       int32_t beam_size, int32_t max_move,
                                       int32_t *const *best_path, int32_t *const *best_labels,
                                       float *const *best_scores, void *stream);
int ka_batch_finish(ka_engine *e, float *total_score, int32_t *status);

/* Kernel form of the fast path.
 *   KA_MODE_WAVE        one wavefront per lattice, checkpointed (throughput; fills the chip from ~4096
 *                       lattices): the forward kernel keeps scores only and stores the score ring every 32
 *                       frames; the backtrace kernel recomputes the back-pointers of the ~100 cells around the
 *                       path from those checkpoints and writes all three outputs.  Lattices whose log-probs are
 *                       not all finite are redone by the exact kernels in the same call; a call with a lattice of
 *                       2^26 frames or more runs entirely in the exact form.
 *   KA_MODE_WAVE_EXACT  one wavefront per lattice, every back-pointer stored (2 bits per band cell).
 *   KA_MODE_TILED       a workgroup of two or three wavefronts per 256- or 128-position TILE of the label axis, the tiles of a lattice run as a pipeline
 *                       (scores only, checkpoints as WAVE): a lone lattice or a book's few dozen chapters, and ANY band
 *                       width (beam_size >= 2L is the reference's unbanded DP).  Non-finite log-probs: bands up to 1009 are
 *                       redone by the exact kernels, wider ones return KA_ERR_NONFINITE in this explicit mode.
 *   KA_MODE_AUTO        (default) per launch: lattices whose band is wider than 1009 positions always run TILED (and are
 *                       handed to the generic kernels by ka_batch_finish if their log-probs turn out to hold infinities);
 *                       the others run TILED while the launch is too small to fill the chip with one wavefront per lattice
 *                       (the rule is in ka_engine.hip, next to its measurements), else WAVE.
 * Results are identical in every form. */
#define KA_MODE_AUTO 0
#define KA_MODE_WAVE 1
/* (2 was KA_MODE_WORKGROUP, four wavefronts per lattice: superseded by the tiled form in round 2, removed in round 4) */
#define KA_MODE_WAVE_EXACT 3
#define KA_MODE_TILED 4
int ka_engine_set_mode(ka_engine *e, int32_t mode);

/* How the checkpointed forms (KA_MODE_WAVE, KA_MODE_TILED) walk the best path back.
 *   KA_BACKTRACE_SERIAL    one wavefront per lattice, chunk after chunk of 32 frames (the position a chunk is entered
 *                          at comes out of the chunk above it): least work, right for thousands of lattices.
 *   KA_BACKTRACE_PARALLEL  every chunk of every lattice at once: per chunk a map "position at its last frame -> rise
 *                          over the chunk" is recomputed for the whole band, 32 maps are composed into a super-chunk map,
 *                          the end position is run down the super-chunk maps and then, in parallel, down the chunk maps,
 *                          which gives every chunk its entry position.  ~8x the work, ~60x shorter for a lone lattice.
 *   KA_BACKTRACE_AUTO      (default) PARALLEL up to a few hundred lattices per call, else SERIAL.
 * Results are identical. */
#define KA_BACKTRACE_AUTO 0
#define KA_BACKTRACE_SERIAL 1
#define KA_BACKTRACE_PARALLEL 2
int ka_engine_set_backtrace(ka_engine *e, int32_t how);

/* Per-kernel timing of the LAST enqueued batch, measured with HIP events recorded on the
 * launch stream: ms[0] label prep, ms[1] forward DP, ms[2] backtrace walk, ms[3] output
 * gathers (best_labels / best_scores).  Enable before the call; costs one event per kernel. */
/* Calibration of KA_MODE_AUTO / KA_BACKTRACE_AUTO (tools/sweep_auto.py): with the lattices of a launch sorted longest
 * first, run the longest n_tiled in the tiled form and walk the longest n_parallel back chunk-parallel instead of asking
 * the cost model (ka_engine.hip); -1 = the cost model.  Results are identical whatever the split. */
int ka_debug_set_split(ka_engine *e, int32_t n_tiled, int32_t n_parallel);
/* Host-side probe of that cost model (no GPU needed): for a launch of n lattices of T[i] frames whose band keeps
 * `tiles_alive` tiles running at once (5 for the reference's beam of 1000) on a device of n_simd SIMDs, how many of the
 * longest it runs tiled and how many it walks back chunk-parallel. */
int ka_debug_auto_split(const int64_t *T, int32_t n, int32_t tiles_alive, int32_t n_simd, int32_t *n_tiled, int32_t *n_parallel);
/* LDS bytes a tile workgroup of the tiled form requests (0 = the library's choice; 40 KB lets four workgroups share a CU, 80 KB
 * two; a request below what the kernel uses - 27-52 KB by tile width, V and row layout - is raised to that): an occupancy
 * experiment knob, results are identical. */
int ka_debug_set_tile_lds(ka_engine *e, int32_t bytes);
/* Positions per tile of the tiled form: 256 (four cells per lane, two wavefronts per tile: ka_tiled2.hpp), 128 (two cells per lane,
 * three wavefronts per tile, self-vouching halo packets: ka_tiled_stream.hpp - a shorter frame, twice the tiles), or 0 = the
 * library's choice (128 while the tiles alive at once are no more than 3.2 per workgroup slot of the device).  Results are identical. */
int ka_debug_set_tile_width(ka_engine *e, int32_t positions);
/* Host-side probe of the library's choice (no GPU needed): the tile width - 128 or 256 - a launch of these n lattices, ALL run in
 * the tiled form, gets on a device of n_simd SIMDs; 0 if one of them is not run in the tiled form at all, a negative status for
 * bad arguments. */
int ka_debug_tile_width_choice(const int64_t *T, const int64_t *S, int32_t n, int32_t V, int32_t beam_size, int32_t max_move, int32_t n_simd);
/* The serial backtrace's output form: 1 = labels and scores gathered from memory after the walk (fewer vector instructions:
 * launches that fill the chip; by the counters 17 % more HBM traffic per step), 0 or -1 (default) = collected by the walk in
 * registers.  Results are identical. */
int ka_debug_set_rc_gather(ka_engine *e, int32_t how);
/* Self-checks of the tiled form's hand-off, a combination of:
 *   1  the halo region is filled with a NaN sentinel before the launch and every packet a tile consumes is checked
 *      against it: a packet read before it was written gives the lattice KA_ERR_INTERNAL (tests)
 *   2  every publish waits for all of the tile's outstanding memory operations (rules out the counted waits)
 *   4  per-tile phase stamps for ka_debug_tile_stats
 * 0 (default) = none.  Applies to the engine's later launches. */
int ka_engine_set_verify(ka_engine *e, int32_t flags);
/* Diagnostics of the tiled form (after ka_engine_set_verify(e, 4)): per tile of the last batch, 8 values
 * {descriptor, tile, t_in, t_end, ticks spent waiting for the tile below, ticks alive, waits, start tick}, 100 MHz
 * ticks, in ticket order.  Returns the number of tiles written (at most max_tasks). */
int ka_debug_tile_stats(ka_engine *e, uint64_t *out, int32_t max_tasks);
/* Diagnostics of the chunk-parallel backtrace: for the first lattice of the last batch, the best-path position at the
 * last frame of every chunk followed by that of every super-chunk (returns how many values), and optionally its
 * chunk maps (one row of ring-size bytes per chunk). */
int ka_debug_chunk_entries(ka_engine *e, int32_t *out, int32_t max_entries, uint8_t *map0_out, int64_t map0_max);
/* Host-side probe of the tiled form's plan (no GPU needed): for a lattice of T frames, S phonemes, V classes the
 * 256-position tiles the band of align.py:64-65 ever touches and the frames [t_in, t_end) each of them is alive in.
 * Returns the number of tiles (t_in / t_end are filled up to max_tiles), 0 if the shape is not run in the tiled form,
 * a negative status for bad arguments.  checkpoint_pitch (may be NULL): bytes per checkpoint row. */
int ka_debug_plan_tiles(int64_t T, int64_t S, int32_t V, int32_t beam_size, int32_t max_move, int32_t *t_in, int32_t *t_end,
                        int32_t max_tiles, int64_t *checkpoint_pitch);
/* The same for tiles of `positions` = 128 or 256 positions (ka_debug_set_tile_width). */
int ka_debug_plan_tiles_width(int64_t T, int64_t S, int32_t V, int32_t beam_size, int32_t max_move, int32_t positions, int32_t *t_in,
                              int32_t *t_end, int32_t max_tiles, int64_t *checkpoint_pitch);
int ka_engine_set_profiling(ka_engine *e, int32_t on);
int ka_engine_last_kernel_ms(ka_engine *e, float ms[4]);

/*
 * Mean-subtracted log-softmax of kokoro_align/align.py:116-117 on device:
 *   x = logits - mean(logits, -1);  log_probs = x - log(sum(exp(x), -1))
 * (float32, not max-subtracted, like the reference).  In-place allowed.
 */
int ka_log_softmax_f32(const float *logits, float *log_probs, int64_t T, int32_t V,
                       int64_t ld_in, int64_t ld_out, void *stream);

/*
 * One time step of one layer of the log-prob producer's bidirectional LSTM (AudioToChar, kokoro_align/train.py:54-65;
 * called from predict, train.py:201-231), both directions at once, for the n sequences that are still running
 * (sequences sorted by length, longest first: the running ones are a prefix).  Fused element-wise part:
 *   gates = gin[rows[dir][s], dir*4H : (dir+1)*4H] + rec[dir][s]          (PyTorch order i, f, g, o)
 *   c[dir][s] = sigmoid(f)*c[dir][s] + sigmoid(i)*tanh(g);   h[dir][s] = sigmoid(o)*tanh(c[dir][s])
 *   out[rows[dir][s], dir*H : (dir+1)*H] = h[dir][s]
 * gin [frames, ldg >= 8H] = x @ W_ih^T + b_ih + b_hh of both directions (one library GEMM per layer),
 * rec [2][n][4H] = h @ W_hh^T (one batched library GEMM per step; *_dir_stride = elements between directions),
 * rows [2][..] int32 = frame row of sequence s at this step (forward: offset+t, backward: offset+len-1-t).
 * All pointers are device pointers.
 */
int ka_lstm_step_f32(const float *gin, int64_t ldg, const float *rec, int64_t rec_dir_stride, float *c, float *h,
                     int64_t state_dir_stride, float *out, int64_t ldo, const int32_t *rows, int64_t rows_dir_stride,
                     int32_t n, int32_t H, void *stream);

/*
 * One whole layer of the same LSTM (hidden size 128), both directions, every time step, in ONE persistent launch:
 * a workgroup owns 16 sequences of one direction, the recurrent product runs on the float32 MFMA with its slice
 * of W_hh register-resident, h in LDS.  Sequences must be sorted by length, longest first; they may lie anywhere
 * in gin / out (frames need not be reordered); gin must hold at least one row.  sigmoid/tanh use the hardware
 * exp2 and reciprocal (about 1 ulp each).
 *   gin [frames, ldg >= 8H] as above; w_hh [2][4H][H] = weight_hh of the forward and the backward direction
 *   (PyTorch layout, gate order i, f, g, o); out [frames, ldo >= 2H] = layer output (forward | backward);
 *   seq_off / seq_len [nseq] int32 = first frame row and length of every sequence.  Device pointers.
 */
int ka_lstm_layer_f32(const float *gin, int64_t ldg, const float *w_hh, float *out, int64_t ldo, const int32_t *seq_off,
                      const int32_t *seq_len, int32_t nseq, int32_t H, void *stream);
/* The FIRST layer with its input projection inside the step (n_in = 40 MFCC coefficients, train.py:54-65 `n_mfcc`): x [frames,
 * ldx >= 40] instead of gin; w_ih [2][4H][40] = weight_ih of the two directions, bias [2][4H] = bias_ih + bias_hh.  The
 * [frames, 8H] projection is never materialised (11 GB for an 8.8-hour book).  Otherwise as ka_lstm_layer_f32. */
int ka_lstm_layer0_f32(const float *x, int64_t ldx, int32_t n_in, const float *w_ih, const float *bias, const float *w_hh, float *out, int64_t ldo,
                       const int32_t *seq_off, const int32_t *seq_len, int32_t nseq, int32_t H, void *stream);

/*
 * Audio front end (kokoro_align/preprocess.py:51-131).  Device pointers throughout.
 *
 * ka_window_energy_f32: out[w] = mean(x[256w : 256w+256]**2), the level get_split_points thresholds
 *   (preprocess.py:53-54), float32 summed in NumPy's order for a contiguous row of 256 so that the split points
 *   are the reference's bit for bit (the remaining steps of get_split_points run on the host on these values).
 * ka_stft_frames_f32: windowed frames of torchaudio's Spectrogram as called by split_audio (preprocess.py:110-127:
 *   center=True, reflect padding, frame f of a segment starts at f*hop - n_fft/2) for all segments of a recording:
 *   frames[frame_off[s] + f][k] = window[k] * y[seg_start[s] + reflect(f*hop - n_fft/2 + k)], f < 1 + seg_len[s]/hop.
 *   Every segment must be longer than n_fft/2 samples (as the reference's transform requires).
 * ka_power_f32: power[r][c] = re^2 + im^2 of a transform stored [n][2*nf] = (real | imaginary) - the output of one
 *   library GEMM of the frames with the [n_fft][2*nf] cosine / sine basis.
 * ka_power_to_db_f32: AmplitudeToDB("power", top_db) per segment, in place: x = 10*log10(max(x, 1e-10)), then
 *   x = max(x, max over the segment - top_db).  frame_off has nseg+1 entries; segmax [nseg] must hold -inf on entry
 *   and returns the segment maxima.
 */
int ka_window_energy_f32(const float *x, int64_t n_windows, int32_t window, float *out, void *stream);
int ka_stft_frames_f32(const float *y, const int64_t *seg_start, const int64_t *seg_len, const int64_t *frame_off, int32_t nseg,
                       int64_t max_frames, int32_t n_fft, int32_t hop, const float *window, float *frames, int64_t ld, void *stream);
int ka_power_f32(const float *reim, int64_t ld_in, float *power, int64_t ld_out, int64_t n, int32_t nf, void *stream);
int ka_power_to_db_f32(float *x, int64_t ld, int32_t cols, const int64_t *frame_off, int32_t nseg, int64_t max_frames, float top_db,
                       float *segmax, void *stream);

/* Bit-reproducible synthetic inputs generated in HBM (same definition as the CPU oracle's
 * hash generator):  lp[t,c] = -8*u24(mix(seed, t*V+c)),
 * labels[k] = 1 + mix(seed^salt, k) % (V-1). */
int ka_hash_logprobs_f32(float *dev_log_probs, int64_t T, int32_t V, int64_t ld, uint64_t seed,
                         void *stream);
int ka_hash_labels_i32(int32_t *dev_labels, int64_t S, int32_t V, uint64_t seed, void *stream);
/* n lattices in one launch: lattice i at base + i*lattice_stride (elements) with seed0 + i */
int ka_hash_logprobs_batch_f32(float *dev_log_probs, int32_t n, int64_t T, int32_t V, int64_t ld,
                               int64_t lattice_stride, uint64_t seed0, void *stream);
int ka_hash_labels_batch_i32(int32_t *dev_labels, int32_t n, int64_t S, int32_t V,
                             int64_t lattice_stride, uint64_t seed0, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KOKORO_ALIGN_AMD_H */
