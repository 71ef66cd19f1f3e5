// Micro-benchmark: throughput of the LDS-pipe instructions the emission gather could be made of, 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_rates lds_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, int seed)
{
    __shared__ float4 tab[64 * 8];   // 8 KB per wave-workgroup is too much for 32 waves per CU: index modulo below
    __shared__ float small[64 * 4];
    const int lane = threadIdx.x;
    uint32_t h = (lane * 2654435761u + seed) >> 7;
    int a0 = (h & 63) * 4, a1 = ((h >> 6) & 63) * 4, a2 = ((h >> 12) & 63) * 4, a3 = ((h >> 18) & 63) * 4;
    for (int i = lane; i < 256; i += 64) small[i] = i;
    __syncthreads();
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
    float r0, r1, r2, r3, r4, r5, r6, r7;
    double d0, d1, d2, d3;
    float4 q0, q1;
    float acc = 0;
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {          // 8 ds_bpermute_b32, then wait
            asm volatile("ds_bpermute_b32 %0, %8, %12\n ds_bpermute_b32 %1, %9, %12\n ds_bpermute_b32 %2, %10, %12\n ds_bpermute_b32 %3, %11, %12\n"
                         "ds_bpermute_b32 %4, %8, %13\n ds_bpermute_b32 %5, %9, %13\n ds_bpermute_b32 %6, %10, %13\n ds_bpermute_b32 %7, %11, %13\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(v0), "v"(v1));
            acc += r0 + r7;
        } else if constexpr (KIND == 1) {   // 8 ds_read_b32 (random addresses inside 256 B)
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %9\n ds_read_b32 %2, %10\n ds_read_b32 %3, %11\n"
                         "ds_read_b32 %4, %8 offset:256\n ds_read_b32 %5, %9 offset:256\n ds_read_b32 %6, %10 offset:256\n ds_read_b32 %7, %11 offset:256\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            acc += r0 + r7;
        } else if constexpr (KIND == 2) {   // 4 ds_read_b64 (8 B per label: two frames)
            asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %5\n ds_read_b64 %2, %6\n ds_read_b64 %3, %7\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(a0 * 2), "v"(a1 * 2), "v"(a2 * 2), "v"(a3 * 2));
            acc += (float)d0 + (float)d3;
        } else if constexpr (KIND == 3) {   // 2 ds_read_b128 (16 B per label: four frames)
            asm volatile("ds_read_b128 %0, %2\n ds_read_b128 %1, %3\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1) : "v"(a0 * 4), "v"(a1 * 4));
            acc += q0.x + q1.w;
        }
        v0 += acc * 1e-30f;
    }
    out[blockIdx.x * 64 + lane] = acc + tab[lane & 7].x * 0.0f;
}

template <int KIND>
void run(const char *name, int n_per_iter, float *d_out)
{
    const int iters = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wps : {2, 8}) {
        const int blocks = 1024 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 10, 1);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        // cycles of one CU (4 SIMDs) per wave-instruction, and per "frame's worth" of gathers (8 label cells x 1 frame)
        const double cyc_cu = ms * 1e-3 * 2.4e9 / ((double)iters * n_per_iter * wps * 4);
        printf("%-40s waves/SIMD=%d  %.3f ms  %.2f cyc per wave-instr per CU (@2.4GHz)\n", name, wps, ms, cyc_cu);
    }
}

int main()
{
    float *d_out; hipMalloc(&d_out, 8192 * 64 * 4);
    run<0>("ds_bpermute_b32", 8, d_out);
    run<1>("ds_read_b32 (random in 256 B)", 8, d_out);
    run<2>("ds_read_b64 (random in 512 B)", 4, d_out);
    run<3>("ds_read_b128 (random in 1 KB)", 2, d_out);
    return 0;
}
