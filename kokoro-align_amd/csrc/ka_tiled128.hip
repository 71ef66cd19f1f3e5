// ka_tiled128.hip — translation unit of the 128-position tile pipeline (ka_tiled_stream.hpp): a workgroup of three wavefronts
// per tile - compute, emission look-up, feeder.
#include "ka_launch.hpp"
#include "ka_tiled_stream.hpp"

#include <algorithm>

namespace ka {

template <int M, int PITCH, bool CONTIG>
static void tiled128(const TileLaunch &a, hipStream_t s)
{
    const unsigned need = (unsigned)TsLds<PITCH, CONTIG>::kTotal;
    hipLaunchKernelGGL((forward_ts_kernel<M, PITCH, CONTIG>), dim3((unsigned)a.n_tasks), dim3(192), std::max(need, a.lds), s, a.lats, a.tasks, a.n_tasks, a.meta, a.halo,
                       a.aux, a.ticket, a.verify, a.stats, a.cu_rank);
}

void launch_forward_tiled128(const TileLaunch &a, hipStream_t s)
{
    if (a.pitch == 256) return tiled128<4, 256, true>(a, s);
    if (a.pitch == 156) return tiled128<4, 156, true>(a, s);
    switch (a.max_move) {
    case 1: tiled128<1, 256, false>(a, s); break;
    case 2: tiled128<2, 256, false>(a, s); break;
    case 3: tiled128<3, 256, false>(a, s); break;
    default: tiled128<4, 256, false>(a, s); break;
    }
}

}  // namespace ka
