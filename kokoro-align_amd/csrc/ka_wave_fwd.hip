// ka_wave_fwd.hip — translation unit of the forward kernels with one wavefront per lattice (ka_wave_forward.hpp).
#include "ka_launch.hpp"
#include "ka_wave_forward.hpp"

namespace ka {

void launch_prep_labels(const Lattice *lats, int n, int32_t *meta, hipStream_t s)
{
    hipLaunchKernelGGL(prep_labels_kernel, dim3(n), dim3(256), 0, s, lats, meta);
}

// two launches over the same lattices: a lattice is taken by the kernel that matches its "transcript contains label 0"
// flag, the other one's waves exit at once
template <int M>
static void forward_wave(const Lattice *lats, int n, int32_t *meta, hipStream_t s, WaveForm form)
{
    if (form == kWaveCheckpointed) {
        hipLaunchKernelGGL((forward_ck_kernel<M, false>), dim3(n), dim3(64), 0, s, lats, meta);
        hipLaunchKernelGGL((forward_ck_kernel<M, true>), dim3(n), dim3(64), 0, s, lats, meta);
    }
    // exact kernels: everything (kWaveExact) or only what the checkpointed kernels declined
    const int only_flagged = form == kWaveCheckpointed ? 1 : 0;
    hipLaunchKernelGGL((forward_w16_kernel<M, false>), dim3(n), dim3(64), 0, s, lats, meta, only_flagged);
    hipLaunchKernelGGL((forward_w16_kernel<M, true>), dim3(n), dim3(64), 0, s, lats, meta, only_flagged);
}

void launch_forward_wave(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s, WaveForm form)
{
    switch (max_move) {
    case 1: forward_wave<1>(lats, n, meta, s, form); break;
    case 2: forward_wave<2>(lats, n, meta, s, form); break;
    case 3: forward_wave<3>(lats, n, meta, s, form); break;
    default: forward_wave<4>(lats, n, meta, s, form); break;
    }
}

template <int M>
static void forward_flagged(const Lattice *lats, int n, int32_t *meta, hipStream_t s)
{
    hipLaunchKernelGGL((forward_w16_kernel<M, false>), dim3(n), dim3(64), 0, s, lats, meta, 1);
    hipLaunchKernelGGL((forward_w16_kernel<M, true>), dim3(n), dim3(64), 0, s, lats, meta, 1);
}

void launch_forward_flagged(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s)
{
    switch (max_move) {
    case 1: forward_flagged<1>(lats, n, meta, s); break;
    case 2: forward_flagged<2>(lats, n, meta, s); break;
    case 3: forward_flagged<3>(lats, n, meta, s); break;
    default: forward_flagged<4>(lats, n, meta, s); break;
    }
}

void launch_forward_generic(const Lattice *lats, int n, int32_t *meta, hipStream_t s)
{
    hipLaunchKernelGGL(forward_generic_kernel, dim3(n), dim3(256), 0, s, lats, meta);
}

}  // namespace ka
