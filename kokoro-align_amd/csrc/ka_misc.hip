// ka_misc.hip — translation unit of everything beside the DP: log-softmax, hash generators (ka_misc.hpp), the log-prob
// producer's LSTM (ka_lstm.hpp), the audio front end (ka_frontend.hpp).
#include "ka_launch.hpp"
#include "ka_misc.hpp"
#include "ka_lstm.hpp"
#include "ka_frontend.hpp"

namespace ka {

void launch_log_softmax(const float *in, float *out, int64_t T, int V, int64_t ld_in, int64_t ld_out, hipStream_t s)
{
    hipLaunchKernelGGL(log_softmax_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, s, in, out, T, V, ld_in, ld_out);
}

void launch_hash_logprobs(float *lp, unsigned blocks, unsigned n, int64_t T, int V, int64_t ld, uint64_t seed0, int64_t lattice_stride, hipStream_t s)
{
    hipLaunchKernelGGL(hash_logprobs_kernel, dim3(blocks, n), dim3(256), 0, s, lp, T, V, ld, seed0, lattice_stride);
}

void launch_hash_labels(int32_t *labels, unsigned blocks, unsigned n, int64_t S, int V, uint64_t seed0, int64_t lattice_stride, hipStream_t s)
{
    hipLaunchKernelGGL(hash_labels_kernel, dim3(blocks, n), dim3(256), 0, s, labels, S, V, seed0, lattice_stride);
}

void launch_lstm_step(const float *gin, int64_t ldg, const float *rec, int64_t rec_dir_stride, float *c, float *h, int64_t state_dir_stride,
                      float *out, int64_t ldo, const int32_t *rows, int64_t rows_dir_stride, int n, int H, hipStream_t s)
{
    const int64_t blocks = ((int64_t)n * H + 255) / 256;
    hipLaunchKernelGGL(lstm_step_kernel, dim3((unsigned)blocks, 2), dim3(256), 0, s, gin, ldg, rec, rec_dir_stride, c, h, state_dir_stride, out, ldo, rows,
                       rows_dir_stride, n, H);
}

void launch_lstm_layer(bool x_in, const float *gin, int64_t ldg, const float *w_hh, float *out, int64_t ldo, const int32_t *seq_off,
                       const int32_t *seq_len, int nseq, const float *w_ih, const float *bias, hipStream_t s)
{
    const dim3 grid(2u * (unsigned)((nseq + kLstmTile - 1) / kLstmTile));
    if (x_in)
        hipLaunchKernelGGL(lstm_layer_kernel<true>, grid, dim3(256), 0, s, gin, ldg, w_hh, out, ldo, seq_off, seq_len, nseq, w_ih, bias);
    else
        hipLaunchKernelGGL(lstm_layer_kernel<false>, grid, dim3(256), 0, s, gin, ldg, w_hh, out, ldo, seq_off, seq_len, nseq, (const float *)nullptr,
                           (const float *)nullptr);
}

void launch_window_energy(const float *x, int64_t n_windows, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(window_energy_kernel, dim3((unsigned)((n_windows + 15) / 16)), dim3(256), 0, s, x, n_windows, out);
}

void launch_stft_frames(const float *y, const int64_t *seg_start, const int64_t *seg_len, const int64_t *frame_off, unsigned grid_x, unsigned nseg,
                        int n_fft, int hop, const float *window, float *frames, int64_t ld, hipStream_t s)
{
    hipLaunchKernelGGL(stft_frames_kernel, dim3(grid_x, nseg), dim3(256), 0, s, y, seg_start, seg_len, frame_off, n_fft, hop, window, frames, ld);
}

void launch_power(const float *reim, int64_t ld_in, float *power, int64_t ld_out, int64_t n, int nf, hipStream_t s)
{
    const unsigned blocks = (unsigned)((n * nf + 255) / 256 < 65536 ? (n * nf + 255) / 256 : 65536);
    hipLaunchKernelGGL(power_kernel, dim3(blocks), dim3(256), 0, s, reim, ld_in, power, ld_out, n, nf);
}

void launch_power_to_db(float *x, int64_t ld, int cols, const int64_t *frame_off, unsigned grid_x, unsigned nseg, float top_db, float *segmax, hipStream_t s)
{
    hipLaunchKernelGGL(power_to_db_kernel, dim3(grid_x, nseg), dim3(256), 0, s, x, ld, cols, frame_off, segmax);
    hipLaunchKernelGGL(db_floor_kernel, dim3(grid_x, nseg), dim3(256), 0, s, x, ld, cols, frame_off, segmax, top_db);
}

}  // namespace ka
