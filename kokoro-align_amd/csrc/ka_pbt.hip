// ka_pbt.hip — translation unit of the chunk-parallel backtrace's map kernels (ka_parallel_bt.hpp).
#include "ka_launch.hpp"
#include "ka_parallel_bt.hpp"

#include <algorithm>

namespace ka {

template <int M>
static void chunk_entries(const Lattice *lats, int n, int32_t *meta, hipStream_t s, unsigned total_chunks, unsigned max_seg, unsigned max_sup, unsigned max_w)
{
    // (max_seg was counted by the host with the same rule: cm_out_for(max_w))
    if (cm_out_for(max_w) == kCmOut)
        hipLaunchKernelGGL((chunk_map_kernel<M, kCmCells>), dim3(total_chunks, max_seg), dim3(64), 0, s, lats, meta, n);
    else
        hipLaunchKernelGGL((chunk_map_kernel<M, kCmCellsWide>), dim3(total_chunks, max_seg), dim3(64), 0, s, lats, meta, n);
    hipLaunchKernelGGL(compose_maps_kernel, dim3(std::min(64u, (max_w + 255u) / 256u), max_sup, (unsigned)n), dim3(256), 0, s, lats, meta);
    hipLaunchKernelGGL(chain_entries_kernel, dim3((unsigned)n), dim3(256), 0, s, lats, meta);
}

void launch_chunk_entries(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s, unsigned total_chunks, unsigned max_seg,
                          unsigned max_sup, unsigned max_w)
{
    switch (max_move) {
    case 1: chunk_entries<1>(lats, n, meta, s, total_chunks, max_seg, max_sup, max_w); break;
    case 2: chunk_entries<2>(lats, n, meta, s, total_chunks, max_seg, max_sup, max_w); break;
    case 3: chunk_entries<3>(lats, n, meta, s, total_chunks, max_seg, max_sup, max_w); break;
    default: chunk_entries<4>(lats, n, meta, s, total_chunks, max_seg, max_sup, max_w); break;
    }
}

}  // namespace ka
