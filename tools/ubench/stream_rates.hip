// Micro-benchmark: HBM read rate of "one stream per wavefront" (the forward kernel's access pattern: every wave
// walks its own lattice's log-prob rows, 256 B per frame) with 256-byte and with 1-KB requests.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_rates stream_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// KIND 0: one dword per lane per row (256 B per wave-instruction), 4 rows in flight
// KIND 1: one dwordx4 per lane per 4 rows (1 KB per wave-instruction), 1-2 blocks in flight
template <int KIND>
__global__ __launch_bounds__(64) void k(const float *base, size_t stream_floats, int rows, float *out)
{
    const float *p = base + (size_t)blockIdx.x * stream_floats;
    const int lane = threadIdx.x;
    float acc = 0;
    if constexpr (KIND == 0) {
        float r0 = p[lane], r1 = p[64 + lane], r2 = p[128 + lane], r3 = p[192 + lane];
        for (int t = 4; t + 4 <= rows; t += 4) {
            const float n0 = p[(size_t)t * 64 + lane], n1 = p[(size_t)(t + 1) * 64 + lane], n2 = p[(size_t)(t + 2) * 64 + lane], n3 = p[(size_t)(t + 3) * 64 + lane];
            acc += r0 + r1 + r2 + r3;
            r0 = n0; r1 = n1; r2 = n2; r3 = n3;
        }
        acc += r0 + r1 + r2 + r3;
    } else {
        const float4 *q = reinterpret_cast<const float4 *>(p);
        float4 r = q[lane];
        for (int t = 4; t + 4 <= rows; t += 4) {
            const float4 n = q[(size_t)(t / 4) * 64 + lane];
            acc += r.x + r.y + r.z + r.w;
            r = n;
        }
        acc += r.x + r.y + r.z + r.w;
    }
    out[blockIdx.x * 64 + lane] = acc;
}

int main()
{
    const int streams = 8192, rows = 50000;
    const size_t stream_floats = (size_t)rows * 64;
    float *d, *o;
    if (hipMalloc(&d, streams * stream_floats * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&o, streams * 64 * 4);
    hipMemset(d, 0, streams * stream_floats * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        for (int kind = 0; kind < 2; ++kind) {
            hipEventRecord(a);
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(streams), dim3(64), 0, 0, d, stream_floats, rows, o);
            else hipLaunchKernelGGL(k<1>, dim3(streams), dim3(64), 0, 0, d, stream_floats, rows, o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            printf("%s: %.2f ms  %.0f GB/s\n", kind == 0 ? "256 B per request (dword per lane)" : "1 KB per request (dwordx4 per lane)", ms, streams * stream_floats * 4.0 / ms / 1e6);
        }
    }
    return 0;
}
