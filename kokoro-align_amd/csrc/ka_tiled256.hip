// ka_tiled256.hip — translation unit of the 256-position tile pipeline, two wavefronts per tile (ka_tiled2.hpp).
#include "ka_launch.hpp"
#include "ka_tiled2.hpp"

namespace ka {

template <int M, int PITCH, bool CONTIG>
static void tiled256(const TileLaunch &a, hipStream_t s)
{
    // (a.lds: what the engine wants a workgroup to hold - 0 = no more than the kernel uses)
    const unsigned lds = a.lds > (unsigned)Tp2Lds<PITCH, CONTIG>::kTotal ? a.lds : (unsigned)Tp2Lds<PITCH, CONTIG>::kTotal;
    hipLaunchKernelGGL((forward_tp2_kernel<M, PITCH, CONTIG>), dim3((unsigned)a.n_tasks), dim3(128), lds, s, a.lats, a.tasks, a.n_tasks, a.meta, a.halo, a.prog,
                       a.aux, a.ticket, a.verify, a.stats);
}

void launch_forward_tiled256(const TileLaunch &a, hipStream_t s)
{
    if (a.pitch == 256) return tiled256<4, 256, true>(a, s);
    if (a.pitch == 156) return tiled256<4, 156, true>(a, s);
    switch (a.max_move) {
    case 1: tiled256<1, 256, false>(a, s); break;
    case 2: tiled256<2, 256, false>(a, s); break;
    case 3: tiled256<3, 256, false>(a, s); break;
    default: tiled256<4, 256, false>(a, s); break;
    }
}

}  // namespace ka
