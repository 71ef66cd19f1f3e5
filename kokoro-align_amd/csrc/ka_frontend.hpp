// ka_frontend.hpp — audio front end kernels (kokoro_align/preprocess.py:51-131).  Included by ka_misc.hip only.
#pragma once
#include "ka_types.hpp"

namespace ka {

// ---------------------------------------------------------------------------------------
// Audio front end (kokoro_align/preprocess.py:51-131)
// ---------------------------------------------------------------------------------------
// Mean square of every 256-sample window (preprocess.py:53-54: np.mean(x.reshape(-1, 256)**2, axis=1), float32).
// The split decision compares these levels with a threshold, so the sum is taken in EXACTLY NumPy's order for a
// contiguous float32 row of 256: two halves of 128; in a half, 8 running sums over elements j, j+8, j+16, ...,
// combined as ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)); then half0 + half1; then / 256.  16 lanes per window: lane
// (h, j) owns running sum j of half h.  grid: 16 windows per 256-thread block.
__global__ __launch_bounds__(256) void window_energy_kernel(const float *__restrict__ x, int64_t n_windows, float *__restrict__ out)
{
    const int64_t w = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int sub = threadIdx.x & 15, h = sub >> 3, j = sub & 7;
    float r = 0.0f;
    if (w < n_windows) {
        const float *p = x + (size_t)w * 256 + h * 128 + j;
        const float v0 = p[0];
        r = v0 * v0;
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const float v = p[8 * k];
            r += v * v;
        }
    }
    // ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)) within each group of 8 lanes, then the two halves
    r = r + __shfl_xor(r, 1);          // lanes 2i, 2i+1 both hold r[2i] + r[2i+1] (float add is commutative)
    r = r + __shfl_xor(r, 2);
    r = r + __shfl_xor(r, 4);
    r = r + __shfl_xor(r, 8);
    if (w < n_windows && sub == 0) out[w] = r / 256.0f;
}

// Windowed frames of the short-time transform, all segments of a recording in one launch:
//   frames[f][k] = window[k] * y_seg[reflect(f_local*hop - n_fft/2 + k)]        (center=True, pad_mode="reflect")
// seg_start / seg_len = first sample and length of every segment, frame_off = first frame row of every segment
// (a segment of len samples has 1 + len/hop frames).  grid: x strides the frames of a segment, y = segment.
__global__ __launch_bounds__(256) void stft_frames_kernel(const float *__restrict__ y, const int64_t *__restrict__ seg_start,
                                                          const int64_t *__restrict__ seg_len, const int64_t *__restrict__ frame_off,
                                                          int n_fft, int hop, const float *__restrict__ window,
                                                          float *__restrict__ frames, int64_t ld)
{
    const int s = blockIdx.y;
    const int64_t len = seg_len[s], start = seg_start[s], f0 = frame_off[s];
    const int64_t nfr = 1 + len / hop;
    const int half = n_fft / 2;
    for (int64_t f = blockIdx.x; f < nfr; f += gridDim.x) {
        float *row = frames + (size_t)(f0 + f) * (size_t)ld;
        for (int k = threadIdx.x; k < n_fft; k += blockDim.x) {
            int64_t i = f * hop - half + k;
            i = i < 0 ? -i : i;
            i = i >= len ? 2 * (len - 1) - i : i;
            row[k] = window[k] * y[start + i];
        }
    }
}

// |X|^2 of a transform stored as [n][2*nf] = (real parts | imaginary parts)
__global__ __launch_bounds__(256) void power_kernel(const float *__restrict__ reim, int64_t ld_in, float *__restrict__ power,
                                                    int64_t ld_out, int64_t n, int nf)
{
    const int64_t total = n * (int64_t)nf;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / nf;
        const int c = (int)(i - r * nf);
        const float re = reim[(size_t)r * ld_in + c], im = reim[(size_t)r * ld_in + nf + c];
        power[(size_t)r * ld_out + c] = re * re + im * im;
    }
}

// 10*log10(max(x, 1e-10)) in place, and the maximum of every segment (AmplitudeToDB("power"), first half).
// segmax must hold -inf on entry.  grid: x strides the rows of a segment, y = segment.
__device__ __forceinline__ void atomic_max_float(float *addr, float v)
{
    // order-preserving integer view: non-negative floats compare as ints, negative ones reversed as unsigned
    if (v >= 0.0f) atomicMax(reinterpret_cast<int *>(addr), __builtin_bit_cast(int, v));
    else atomicMin(reinterpret_cast<unsigned int *>(addr), __builtin_bit_cast(unsigned int, v));
}
__global__ __launch_bounds__(256) void power_to_db_kernel(float *__restrict__ x, int64_t ld, int cols, const int64_t *__restrict__ frame_off,
                                                          float *__restrict__ segmax)
{
    const int s = blockIdx.y;
    const int64_t r0 = frame_off[s], r1 = frame_off[s + 1];
    const int64_t total = (r1 - r0) * cols;
    float m = -__builtin_inff();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        float *p = x + (size_t)(r0 + r) * (size_t)ld + (i - r * cols);
        const float v = 10.0f * log10f(fmaxf(*p, 1e-10f));
        *p = v;
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0 && m > -__builtin_inff()) atomic_max_float(&segmax[s], m);
}
// x = max(x, segmax - top_db) (AmplitudeToDB, second half: top_db is relative to the maximum of one call = one segment)
__global__ __launch_bounds__(256) void db_floor_kernel(float *__restrict__ x, int64_t ld, int cols, const int64_t *__restrict__ frame_off,
                                                       const float *__restrict__ segmax, float top_db)
{
    const int s = blockIdx.y;
    const int64_t r0 = frame_off[s], r1 = frame_off[s + 1];
    const int64_t total = (r1 - r0) * cols;
    const float lo = segmax[s] - top_db;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / cols;
        float *p = x + (size_t)(r0 + r) * (size_t)ld + (i - r * cols);
        *p = fmaxf(*p, lo);
    }
}


}  // namespace ka
