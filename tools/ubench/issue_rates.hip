// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the instruction kinds the
// forward DP kernel is made of, at 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f;
    uint32_t w0 = threadIdx.x, w1 = w0 + 1, w2 = w0 + 2, w3 = w0 + 3;
    uint64_t m0 = 0x0123456789abcdefull, m1 = ~m0, m2, m3;
    float e = seed * 0.5f;
    for (int i = 0; i < iters; ++i) {
        if constexpr (KIND == 0) {  // independent v_add_f32
            asm volatile(REP8("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));
        } else if constexpr (KIND == 1) {  // v_max_f32
            asm volatile(REP8("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));
        } else if constexpr (KIND == 2) {  // v_cmp_gt_f32 -> SGPR pair (VOP3)
            asm volatile(REP8("v_cmp_gt_f32 %8, %0, %1\n v_cmp_gt_f32 %9, %1, %2\n v_cmp_gt_f32 %8, %2, %3\n v_cmp_gt_f32 %9, %3, %4\n v_cmp_gt_f32 %8, %4, %5\n v_cmp_gt_f32 %9, %5, %6\n v_cmp_gt_f32 %8, %6, %7\n v_cmp_gt_f32 %9, %7, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=s"(m2), "=s"(m3));
        } else if constexpr (KIND == 3) {  // v_cmp_gt_f32 -> VCC (VOPC e32)
            asm volatile(REP8("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 vcc, %1, %2\n v_cmp_gt_f32 vcc, %2, %3\n v_cmp_gt_f32 vcc, %3, %4\n v_cmp_gt_f32 vcc, %4, %5\n v_cmp_gt_f32 vcc, %5, %6\n v_cmp_gt_f32 vcc, %6, %7\n v_cmp_gt_f32 vcc, %7, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if constexpr (KIND == 4) {  // v_addc_co_u32 with SGPR carry-in/out, 4 independent chains
            asm volatile(REP8("v_addc_co_u32 %0, %4, %0, %0, %6\n v_addc_co_u32 %1, %5, %1, %1, %7\n v_addc_co_u32 %2, %4, %2, %2, %6\n v_addc_co_u32 %3, %5, %3, %3, %7\n v_addc_co_u32 %0, %4, %0, %0, %6\n v_addc_co_u32 %1, %5, %1, %1, %7\n v_addc_co_u32 %2, %4, %2, %2, %6\n v_addc_co_u32 %3, %5, %3, %3, %7\n")
                         : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "=&s"(m2), "=&s"(m3) : "s"(m0), "s"(m1));
        } else if constexpr (KIND == 5) {  // v_cndmask_b32 with SGPR mask
            asm volatile(REP8("v_cndmask_b32 %0, %8, %0, %9\n v_cndmask_b32 %1, %8, %1, %10\n v_cndmask_b32 %2, %8, %2, %9\n v_cndmask_b32 %3, %8, %3, %10\n v_cndmask_b32 %4, %8, %4, %9\n v_cndmask_b32 %5, %8, %5, %10\n v_cndmask_b32 %6, %8, %6, %9\n v_cndmask_b32 %7, %8, %7, %10\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e), "s"(m0), "s"(m1));
        } else if constexpr (KIND == 6) {  // s_and_b64 / s_xor_b64 only
            asm volatile(REP8("s_and_b64 %0, %0, %2\n s_xor_b64 %1, %1, %3\n s_or_b64 %0, %0, %3\n s_xor_b64 %1, %1, %2\n s_and_b64 %0, %0, %2\n s_xor_b64 %1, %1, %3\n s_or_b64 %0, %0, %3\n s_xor_b64 %1, %1, %2\n")
                         : "+s"(m0), "+s"(m1) : "s"(0x5555555555555555ull), "s"(0x3333333333333333ull) : "scc");
        } else if constexpr (KIND == 7) {  // the cell pattern: 4 add, 3 cmp(sgpr), 3 max, 3 salu, 2 addc, 1 cndmask  (x4 cells, x2)
            asm volatile(REP8(
                "v_add_f32 %0, %4, %8\n v_add_f32 %1, %5, %8\n v_add_f32 %2, %6, %8\n v_add_f32 %3, %7, %8\n"
                "v_cmp_gt_f32 vcc, %1, %0\n v_max_f32 %0, %0, %1\n v_cmp_gt_f32 %10, %3, %2\n v_max_f32 %2, %2, %3\n"
                "v_cmp_gt_f32 %11, %2, %0\n v_max_f32 %0, %0, %2\n s_xor_b64 %10, %10, vcc\n s_and_b64 %10, %10, %11\n"
                "v_addc_co_u32 %9, %11, %9, %9, %11\n s_xor_b64 %10, %10, vcc\n v_cndmask_b32 %4, %8, %0, %12\n v_addc_co_u32 %9, %10, %9, %9, %10\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(e), "+v"(w0), "=&s"(m2), "=&s"(m3) : "s"(m0) : "vcc", "scc");
        } else if constexpr (KIND == 8) {  // v_add_f32 with DPP wave_ror
            asm volatile(REP8("v_add_f32_dpp %0, %1, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %3, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %5, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %7, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %0, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %2, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %4, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %6, %8 wave_ror:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));
        } else if constexpr (KIND == 9) {  // v_pk_add_f32
            asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n")
                         : "+v"(*(double *)&a0), "+v"(*(double *)&a2), "+v"(*(double *)&a4), "+v"(*(double *)&a6) : "v"(*(double *)&m0));
        } else if constexpr (KIND == 10) {  // v_max3_f32
            asm volatile(REP8("v_max3_f32 %0, %0, %8, %1\n v_max3_f32 %1, %1, %8, %2\n v_max3_f32 %2, %2, %8, %3\n v_max3_f32 %3, %3, %8, %4\n v_max3_f32 %4, %4, %8, %5\n v_max3_f32 %5, %5, %8, %6\n v_max3_f32 %6, %6, %8, %7\n v_max3_f32 %7, %7, %8, %0\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e));
        } else if constexpr (KIND == 11) {  // v_cmp_eq_f32 -> sgpr + v_cndmask using it (dependent pair)
            asm volatile(REP8("v_cmp_gt_f32 %8, %0, %1\n v_cndmask_b32 %0, %0, %1, %8\n v_cmp_gt_f32 %9, %2, %3\n v_cndmask_b32 %2, %2, %3, %9\n v_cmp_gt_f32 %8, %4, %5\n v_cndmask_b32 %4, %4, %5, %8\n v_cmp_gt_f32 %9, %6, %7\n v_cndmask_b32 %6, %6, %7, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&s"(m2), "=&s"(m3));
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + e + (float)(w0 + w1 + w2 + w3) + (float)(m0 ^ m1);
}


#define OP8(NAME, TXT) \
template <> __global__ __launch_bounds__(64) void kk<NAME>(float *out, int iters, float seed) { \
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, a4 = a0 * 1.4f, a5 = a0 * 1.5f, a6 = a0 * 1.6f, a7 = a0 * 1.7f; \
    float e = seed * 0.5f; \
    for (int i = 0; i < iters; ++i) { \
        asm volatile(REP8(TXT) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(e)); \
    } \
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + e; \
}
template <int NAME> __global__ void kk(float *out, int iters, float seed);
#define T2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define T3(op, tail) op " %0, %0, %8" tail "\n" op " %1, %1, %8" tail "\n" op " %2, %2, %8" tail "\n" op " %3, %3, %8" tail "\n" op " %4, %4, %8" tail "\n" op " %5, %5, %8" tail "\n" op " %6, %6, %8" tail "\n" op " %7, %7, %8" tail "\n"
OP8(100, T2("v_sub_f32"))
OP8(101, T2("v_mul_f32"))
OP8(102, T3("v_fma_f32", ", %8"))
OP8(103, T2("v_min_f32"))
OP8(104, T2("v_max_f32_e64"))
OP8(105, T2("v_add_u32"))
OP8(106, T2("v_sub_u32"))
OP8(107, T2("v_min_u32"))
OP8(108, T3("v_alignbit_b32", ", 31"))
OP8(109, T3("v_lshl_or_b32", ", 1"))
OP8(110, T3("v_and_or_b32", ", %8"))
OP8(111, T3("v_bfi_b32", ", %8"))
OP8(112, T3("v_med3_f32", ", %8"))
OP8(113, T2("v_xor_b32"))
OP8(114, T2("v_lshlrev_b32"))
OP8(115, T3("v_add3_u32", ", %8"))
OP8(116, T3("v_lshl_add_u32", ", 1"))
OP8(117, T2("v_and_b32"))
OP8(118, T3("v_mad_u32_u24", ", %8"))
OP8(119, T2("v_ashrrev_i32"))
OP8(120, T3("v_bitop3_b32", ", %8 bitop3:0xc8"))
OP8(121, T2("v_max_i32"))
OP8(122, T2("v_add_f32_e64"))
OP8(123, T3("v_max3_f32", ", %8"))
OP8(124, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")

OP8(130, T2("v_or_b32"))
// exec-masked bit set: s_mov exec, mask ; v_or_b32 ; ... ; restore
template <> __global__ __launch_bounds__(64) void kk<131>(float *out, int iters, float seed) {
    uint32_t w0 = threadIdx.x, w1 = 3;
    uint64_t m0 = 0x0123456789abcdefull * (uint64_t)(iters | 1), m1 = ~m0;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8(
            "s_mov_b64 exec, %2\n v_or_b32 %0, 2, %0\n s_mov_b64 exec, %3\n v_or_b32 %0, 1, %0\n s_mov_b64 exec, -1\n v_add_u32 %0, %0, %0\n v_add_u32 %0, %0, %0\n"
            "s_mov_b64 exec, %3\n v_or_b32 %1, 2, %1\n s_mov_b64 exec, %2\n v_or_b32 %1, 1, %1\n s_mov_b64 exec, -1\n v_add_u32 %1, %1, %1\n v_add_u32 %1, %1, %1\n")
            : "+v"(w0), "+v"(w1) : "s"(m0), "s"(m1));
    }
    out[blockIdx.x * 64 + threadIdx.x] = (float)(w0 + w1);
}
// cell pattern A (current): 4 add, 2 max(max3+max), 3 cmp_eq, 3 salu, 2 addc, 1 cndmask
template <> __global__ __launch_bounds__(64) void kk<132>(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, c0, c1, c2, c3, m, e = 0.5f; uint32_t w = 1; uint64_t mk = 0xffff0000ffff0000ull, s0, s1, s2;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8(
            "v_add_f32 %4, %0, %9\n v_add_f32 %5, %1, %9\n v_add_f32 %6, %2, %9\n v_add_f32 %7, %3, %9\n"
            "v_max3_f32 %8, %4, %5, %6\n v_max_f32 %8, %8, %7\n"
            "v_cmp_eq_f32 %11, %4, %8\n v_cmp_eq_f32 %12, %5, %8\n v_cmp_eq_f32 %13, %6, %8\n"
            "s_or_b64 %11, %11, %12\n s_andn2_b64 %13, %13, %12\n s_or_b64 %13, %13, %11\n"
            "v_addc_co_u32 %10, %12, %10, %10, %11\n v_addc_co_u32 %10, %12, %10, %10, %13\n"
            "v_cndmask_b32 %0, %9, %8, %14\n")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(m), "+v"(e), "+v"(w), "=&s"(s0), "=&s"(s1), "=&s"(s2) : "s"(mk) : "vcc", "scc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + e + (float)w;
}
// cell pattern B: same but bits set with exec-masked v_or (word pre-shifted by 2 with two fast adds)
template <> __global__ __launch_bounds__(64) void kk<133>(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 * 1.1f, a2 = a0 * 1.2f, a3 = a0 * 1.3f, c0, c1, c2, c3, m, e = 0.5f; uint32_t w = 1; uint64_t mk = 0xffff0000ffff0000ull, s0, s1, s2;
    for (int i = 0; i < iters; ++i) {
        asm volatile(REP8(
            "v_add_f32 %4, %0, %9\n v_add_f32 %5, %1, %9\n v_add_f32 %6, %2, %9\n v_add_f32 %7, %3, %9\n"
            "v_max3_f32 %8, %4, %5, %6\n v_max_f32 %8, %8, %7\n"
            "v_cmp_eq_f32 %11, %4, %8\n v_cmp_eq_f32 %12, %5, %8\n v_cmp_eq_f32 %13, %6, %8\n"
            "s_or_b64 %11, %11, %12\n s_andn2_b64 %13, %13, %12\n s_or_b64 %13, %13, %11\n"
            "s_mov_b64 exec, %11\n v_or_b32 %10, 2, %10\n s_mov_b64 exec, %13\n v_or_b32 %10, 1, %10\n s_mov_b64 exec, -1\n"
            "v_cndmask_b32 %0, %9, %8, %14\n")
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3), "=&v"(m), "+v"(e), "+v"(w), "=&s"(s0), "=&s"(s1), "=&s"(s2) : "s"(mk) : "vcc", "scc");
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + e + (float)w;
}

template <int NAME>
void run2(const char *name, float *d_out)
{
    const int iters = 4000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int wps : {2, 8}) {
        const int blocks = 1024 * wps;
        hipLaunchKernelGGL(kk<NAME>, dim3(blocks), dim3(64), 0, 0, d_out, 10, 1.0f);
        hipEventRecord(a);
        hipLaunchKernelGGL(kk<NAME>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f);
        hipEventRecord(b); hipEventSynchronize(b);
        hipError_t er = hipGetLastError();
        if (er != hipSuccess) printf("%s: launch error %s\n", name, hipGetErrorString(er));
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        printf("%-44s waves/SIMD=%d  %.3f ms  %.2f cyc per wave-instr per SIMD (@2.4GHz)\n", name, wps, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
    }
}

template <int KIND>
void run(const char *name, int n_per_iter, float *d_out)
{
    const int iters = 2000;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 1024 * wps;
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, 10, 1.0f);
        hipEventRecord(a);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d_out, iters, 1.0f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        const double cyc = ms * 1e-3 * 2.4e9;               // at 2.4 GHz
        const double per = cyc / ((double)iters * n_per_iter * wps);
        printf("%-44s waves/SIMD=%d  %.3f ms  %.2f cyc per wave-instr per SIMD (@2.4GHz)\n", name, wps, ms, per);
    }
}

int main()
{
    float *d_out; hipMalloc(&d_out, 8192 * 64 * 4);
    run<0>("v_add_f32 (indep)", 64, d_out);
    run<1>("v_max_f32", 64, d_out);
    run<10>("v_max3_f32", 64, d_out);
    run<2>("v_cmp_gt_f32 -> SGPR pair (e64)", 64, d_out);
    run<3>("v_cmp_gt_f32 -> vcc (e32)", 64, d_out);
    run<4>("v_addc_co_u32 sgpr carry", 64, d_out);
    run<5>("v_cndmask_b32 sgpr mask", 64, d_out);
    run<11>("v_cmp->sgpr + dependent v_cndmask", 64, d_out);
    run<8>("v_add_f32_dpp wave_ror:1", 64, d_out);
    run<9>("v_pk_add_f32", 64, d_out);
    run<6>("s_and/xor/or_b64", 64, d_out);
    run<7>("cell pattern (12 VALU + 3 SALU)", 8 * 16, d_out);
    run2<100>("v_sub_f32", d_out); run2<101>("v_mul_f32", d_out); run2<102>("v_fma_f32", d_out); run2<103>("v_min_f32", d_out);
    run2<104>("v_max_f32_e64", d_out); run2<122>("v_add_f32_e64", d_out); run2<123>("v_max3_f32 (3 vgpr)", d_out); run2<112>("v_med3_f32", d_out);
    run2<105>("v_add_u32", d_out); run2<106>("v_sub_u32", d_out); run2<107>("v_min_u32", d_out); run2<121>("v_max_i32", d_out);
    run2<108>("v_alignbit_b32 (shift 31)", d_out); run2<109>("v_lshl_or_b32", d_out); run2<110>("v_and_or_b32", d_out); run2<111>("v_bfi_b32", d_out);
    run2<113>("v_xor_b32", d_out); run2<117>("v_and_b32", d_out); run2<114>("v_lshlrev_b32", d_out); run2<119>("v_ashrrev_i32", d_out);
    run2<115>("v_add3_u32", d_out); run2<116>("v_lshl_add_u32", d_out); run2<118>("v_mad_u32_u24", d_out); run2<120>("v_bitop3_b32", d_out);
    run2<124>("v_mov_b32", d_out);
    run2<130>("v_or_b32", d_out); run2<131>("exec-masked v_or pairs (7 instr/unit, x64 => per unit)", d_out);
    run2<132>("cell A: addc packing (15 instr per cell, reported per 1/64 iter)", d_out); run2<133>("cell B: exec-masked v_or packing (18 instr per cell)", d_out);
    return 0;
}
