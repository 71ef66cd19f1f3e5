// ka_launch.hpp — the host-callable launch functions of every device translation unit.
//
// The library is built from seven translation units so that the device code compiles in parallel (a single unit took three
// minutes): ka_engine.hip is host code only and reaches the kernels through these functions; each kernel family lives in
// the .hip file named below and nowhere else.  All functions only enqueue; errors surface through hipGetLastError().
#pragma once
#include "ka_types.hpp"

namespace ka {

// which forward kernels of the one-wavefront-per-lattice family run over a range of descriptors
enum WaveForm { kWaveExact = 0, kWaveCheckpointed = 1 };

// ---- ka_wave_fwd.hip ----
void launch_prep_labels(const Lattice *lats, int n, int32_t *meta, hipStream_t s);
// checkpointed: forward_ck over all, then the exact kernels over what it flagged; exact: forward_w16 over all
void launch_forward_wave(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s, WaveForm form);
// the exact kernels over lattices another forward kernel has flagged kFlagExact
void launch_forward_flagged(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s);
void launch_forward_generic(const Lattice *lats, int n, int32_t *meta, hipStream_t s);

// ---- ka_wave_bt.hip ----
// serial: one wavefront per lattice (skips Lattice::par ones); gather: labels and scores fetched after the walk
void launch_backtrace_rc_serial(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s, bool gather);
// one wavefront per chunk of the launch's chunk-parallel lattices (Lattice::entry holds where the path enters each)
void launch_backtrace_rc_chunks(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s, unsigned total_chunks);
void launch_backtrace_w16(const Lattice *lats, int n, const int32_t *meta, hipStream_t s, int only_flagged);
void launch_gather_outputs(const Lattice *lats, unsigned grid_x, unsigned grid_y, const int32_t *meta, hipStream_t s, int only_flagged);
void launch_backtrace_generic(const Lattice *lats, int n, const int32_t *meta, hipStream_t s);

// ---- ka_pbt.hip: chunk maps -> super-chunk maps -> entry position of every chunk (ka_parallel_bt.hpp) ----
void launch_chunk_entries(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s, unsigned total_chunks, unsigned max_seg,
                          unsigned max_sup, unsigned max_w);

// ---- ka_tiled256.hip / ka_tiled128.hip: the tile pipelines ----
struct TileLaunch {
    const Lattice *lats;
    const TileTask *tasks;
    int n_tasks;
    int32_t *meta;
    char *halo;
    uint32_t *prog;
    TileAux *aux;
    uint32_t *ticket;
    int verify;
    TpStats *stats;
    uint32_t *cu_rank;  // [kCuSlots] workgroups per CU, zeroed per launch (128-position tiles: ka_tiled_stream.hpp)
    unsigned lds;       // LDS bytes a workgroup requests (at least what the kernel uses)
    int max_move;
    int pitch;          // 0: rows staged one by one; 256 (V = 64) or 156 (V = 39): contiguous rows, copied as they lie
};
void launch_forward_tiled256(const TileLaunch &a, hipStream_t s);               // two wavefronts per 256-position tile (ka_tiled2.hpp)
void launch_forward_tiled128(const TileLaunch &a, hipStream_t s);               // three wavefronts per 128-position tile (ka_tiled_stream.hpp); `prog` unused

// ---- ka_misc.hip: log-softmax, hash generators, the log-prob producer's LSTM, the audio front end ----
void launch_log_softmax(const float *in, float *out, int64_t T, int V, int64_t ld_in, int64_t ld_out, hipStream_t s);
void launch_hash_logprobs(float *lp, unsigned blocks, unsigned n, int64_t T, int V, int64_t ld, uint64_t seed0, int64_t lattice_stride, hipStream_t s);
void launch_hash_labels(int32_t *labels, unsigned blocks, unsigned n, int64_t S, int V, uint64_t seed0, int64_t lattice_stride, hipStream_t s);
void launch_lstm_step(const float *gin, int64_t ldg, const float *rec, int64_t rec_dir_stride, float *c, float *h, int64_t state_dir_stride,
                      float *out, int64_t ldo, const int32_t *rows, int64_t rows_dir_stride, int n, int H, hipStream_t s);
// x_in: layer 0 with its input projection inside (gin is x [frames, ldg >= 40]; w_ih, bias given)
void launch_lstm_layer(bool x_in, const float *gin, int64_t ldg, const float *w_hh, float *out, int64_t ldo, const int32_t *seq_off,
                       const int32_t *seq_len, int nseq, const float *w_ih, const float *bias, hipStream_t s);
void launch_window_energy(const float *x, int64_t n_windows, float *out, hipStream_t s);
void launch_stft_frames(const float *y, const int64_t *seg_start, const int64_t *seg_len, const int64_t *frame_off, unsigned grid_x, unsigned nseg,
                        int n_fft, int hop, const float *window, float *frames, int64_t ld, hipStream_t s);
void launch_power(const float *reim, int64_t ld_in, float *power, int64_t ld_out, int64_t n, int nf, hipStream_t s);
void launch_power_to_db(float *x, int64_t ld, int cols, const int64_t *frame_off, unsigned grid_x, unsigned nseg, float top_db, float *segmax, hipStream_t s);

}  // namespace ka
