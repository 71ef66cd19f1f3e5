// Probe: facts the sign-of-difference compare relies on (gfx950, the library's compile flags).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
__global__ void k(const float *a, const float *b, uint32_t *out, int n)
{
    int i = threadIdx.x;
    if (i < n) {
        float d;
        asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a[i]), "v"(b[i]));
        out[i] = __builtin_bit_cast(uint32_t, d);
        float m = __builtin_fmaxf(a[i], b[i]);
        out[n + i] = __builtin_bit_cast(uint32_t, m);
    }
}
int main()
{
    const float inf = INFINITY;
    float ha[] = {-inf, -inf, 1.0f, -inf, 1.0e-38f, 1.17549435e-38f, -0.0f, 0.0f, -5.0f, 3.0e-39f};
    float hb[] = {-inf, 1.0f, -inf, -5.0f, 0.9999e-38f, 1.17549421e-38f, 0.0f, -0.0f, -5.0f, 2.9e-39f};
    const int n = 10;
    float *a, *b; uint32_t *o, ho[2 * n];
    hipMalloc(&a, 64); hipMalloc(&b, 64); hipMalloc(&o, 256);
    hipMemcpy(a, ha, n * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, o, n);
    hipMemcpy(ho, o, 2 * n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("a=%-14g b=%-14g a-b bits=0x%08x sign=%u   max bits=0x%08x\n", ha[i], hb[i], ho[i], ho[i] >> 31, ho[n + i]);
    return 0;
}
