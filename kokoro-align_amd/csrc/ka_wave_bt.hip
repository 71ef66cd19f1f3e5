// ka_wave_bt.hip — translation unit of the backtrace kernels (ka_wave_backtrace.hpp).
#include "ka_launch.hpp"
#include "ka_wave_backtrace.hpp"

namespace ka {

template <int M>
static void rc_serial(const Lattice *lats, int n, int32_t *meta, hipStream_t s, bool gather)
{
    if (gather) {
        hipLaunchKernelGGL((backtrace_rc_kernel<M, false, false, true>), dim3(n), dim3(64), 0, s, lats, meta, n);
        hipLaunchKernelGGL((backtrace_rc_kernel<M, true, false, true>), dim3(n), dim3(64), 0, s, lats, meta, n);
    } else {
        hipLaunchKernelGGL((backtrace_rc_kernel<M, false, false, false>), dim3(n), dim3(64), 0, s, lats, meta, n);
        hipLaunchKernelGGL((backtrace_rc_kernel<M, true, false, false>), dim3(n), dim3(64), 0, s, lats, meta, n);
    }
}

void launch_backtrace_rc_serial(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s, bool gather)
{
    switch (max_move) {
    case 1: rc_serial<1>(lats, n, meta, s, gather); break;
    case 2: rc_serial<2>(lats, n, meta, s, gather); break;
    case 3: rc_serial<3>(lats, n, meta, s, gather); break;
    default: rc_serial<4>(lats, n, meta, s, gather); break;
    }
}

template <int M>
static void rc_chunks(const Lattice *lats, int n, int32_t *meta, hipStream_t s, unsigned total_chunks)
{
    hipLaunchKernelGGL((backtrace_rc_chunks_kernel<M>), dim3(total_chunks), dim3(64), 0, s, lats, meta, n);
}

void launch_backtrace_rc_chunks(int max_move, const Lattice *lats, int n, int32_t *meta, hipStream_t s, unsigned total_chunks)
{
    switch (max_move) {
    case 1: rc_chunks<1>(lats, n, meta, s, total_chunks); break;
    case 2: rc_chunks<2>(lats, n, meta, s, total_chunks); break;
    case 3: rc_chunks<3>(lats, n, meta, s, total_chunks); break;
    default: rc_chunks<4>(lats, n, meta, s, total_chunks); break;
    }
}

void launch_backtrace_w16(const Lattice *lats, int n, const int32_t *meta, hipStream_t s, int only_flagged)
{
    hipLaunchKernelGGL(backtrace_w16_kernel, dim3(n), dim3(64), 0, s, lats, meta, only_flagged);
}

void launch_gather_outputs(const Lattice *lats, unsigned grid_x, unsigned grid_y, const int32_t *meta, hipStream_t s, int only_flagged)
{
    hipLaunchKernelGGL(gather_outputs_kernel, dim3(grid_x, grid_y), dim3(256), 0, s, lats, meta, only_flagged);
}

void launch_backtrace_generic(const Lattice *lats, int n, const int32_t *meta, hipStream_t s)
{
    hipLaunchKernelGGL(backtrace_generic_kernel, dim3(n), dim3(64), 0, s, lats, meta);
}

}  // namespace ka
