// Where do the single-wavefront workgroups of a grid land?  Each workgroup records its XCC / SE / CU / SIMD and then
// stays resident for ~100 us, so that the whole grid is on the chip together.  Prints, per grid size, the number of
// distinct SIMDs used and the largest number of waves that share one.
//   hipcc --offload-arch=gfx950 -O2 -o census census.hip && ./census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(64) void census(uint32_t *out, long long spin)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
int main()
{
    for (int G : {256, 512, 1024, 1536, 2048, 4096}) {
        uint32_t *d;
        hipMalloc(&d, G * 8);
        hipLaunchKernelGGL(census, dim3(G), dim3(64), 0, 0, d, 10000LL /* 100 MHz ticks = 100 us */);
        std::vector<uint32_t> h(2 * G);
        hipMemcpy(h.data(), d, G * 8, hipMemcpyDeviceToHost);
        std::map<uint32_t, int> simd, cu;
        for (int i = 0; i < G; ++i) {
            const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
            const uint32_t simd_id = (hw >> 4) & 3, cu_id = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            const uint32_t cukey = (xcc << 12) | (se << 8) | (sh << 4) | cu_id;
            cu[cukey]++;
            simd[(cukey << 2) | simd_id]++;
        }
        int mx = 0, cmx = 0;
        for (auto &kv : simd) mx = kv.second > mx ? kv.second : mx;
        for (auto &kv : cu) cmx = kv.second > cmx ? kv.second : cmx;
        printf("grid %5d: %4zu CUs (max %d waves on one), %4zu SIMDs used, max %d waves on one SIMD; first 8 blocks:", G, cu.size(), cmx, simd.size(), mx);
        for (int i = 0; i < 8 && i < G; ++i) printf(" [x%u se%u cu%u s%u]", h[2 * i + 1] & 0xf, (h[2 * i] >> 13) & 7, (h[2 * i] >> 8) & 0xf, (h[2 * i] >> 4) & 3);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
