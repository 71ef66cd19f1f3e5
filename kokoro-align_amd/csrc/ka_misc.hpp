// ka_misc.hpp — log-softmax (align.py:116-117) and the hash generators of the synthetic inputs.  Included by ka_misc.hip only.
#pragma once
#include "ka_types.hpp"

namespace ka {

// ---------------------------------------------------------------------------------------
// mean-subtracted log-softmax (align.py:116-117), one wavefront per row
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float x)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;
}
__global__ __launch_bounds__(256) void log_softmax_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          int64_t T, int V, int64_t ld_in, int64_t ld_out)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const float *x = in + (size_t)row * (size_t)ld_in;
    float *y = out + (size_t)row * (size_t)ld_out;
    float s = 0.0f;
    for (int c = lane; c < V; c += 64) s += x[c];
    const float mean = wave_sum(s) / (float)V;
    float z = 0.0f;
    for (int c = lane; c < V; c += 64) z += expf(x[c] - mean);
    const float lz = logf(wave_sum(z));
    for (int c = lane; c < V; c += 64) y[c] = (x[c] - mean) - lz;
}

// ---------------------------------------------------------------------------------------
// hash generator of synthetic inputs (definition: include/kokoro_align_amd.h)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t mix64(uint64_t seed, uint64_t idx)
{
    uint64_t z = (seed * 0x9E3779B97F4A7C15ull + idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// blockIdx.y = lattice: lattice i lives at base + i*stride elements and uses seed + i
__global__ __launch_bounds__(256) void hash_logprobs_kernel(float *lp0, int64_t T, int V, int64_t ld, uint64_t seed0,
                                                            int64_t lattice_stride)
{
    float *lp = lp0 + (size_t)blockIdx.y * (size_t)lattice_stride;
    const uint64_t seed = seed0 + blockIdx.y;
    const int64_t n = T * (int64_t)V;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / V;
        const int c = (int)(i - t * V);
        const uint64_t h = mix64(seed, (uint64_t)i);
        lp[(size_t)t * (size_t)ld + c] = -8.0f * ((float)(h >> 40) * (1.0f / 16777216.0f));
    }
}
__global__ __launch_bounds__(256) void hash_labels_kernel(int32_t *labels0, int64_t S, int V, uint64_t seed0,
                                                          int64_t lattice_stride)
{
    int32_t *labels = labels0 + (size_t)blockIdx.y * (size_t)lattice_stride;
    const uint64_t seed = seed0 + blockIdx.y;
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < S; k += (int64_t)gridDim.x * blockDim.x)
        labels[k] = (int32_t)(1 + mix64(seed ^ 0x4C4142454C53ull, (uint64_t)k) % (uint64_t)(V - 1));
}

}  // namespace ka
