// islands.hip — the island kernel's translation unit (compiled with -fno-slp-vectorize, see island_view.h) and its launcher.
#include "island_kernel.h"

namespace phx {

void launch_solve_islands(hipStream_t stream, int groups, bool big_shape, bool half_state, bool trace, const SolverView& v, const IslandView& iv,
                          const BodyView& bodies, phx_contact_joint* joints, const phx_contact_point* cps, int ci, int pi)
{
    const dim3 grid(groups);
    if (trace && big_shape)      hipLaunchKernelGGL((k_solve_islands<ISL_T_BIG, ISL_B_BIG, false, true>), grid, dim3(ISL_T_BIG), 0, stream, v, iv, bodies, joints, cps, ci, pi);
    else if (trace)              hipLaunchKernelGGL((k_solve_islands<ISL_T, ISL_B, false, true>), grid, dim3(ISL_T), 0, stream, v, iv, bodies, joints, cps, ci, pi);
    else if (big_shape && half_state) hipLaunchKernelGGL((k_solve_islands<ISL_T_BIG, ISL_B_BIG, true>), grid, dim3(ISL_T_BIG), 0, stream, v, iv, bodies, joints, cps, ci, pi);
    else if (big_shape)          hipLaunchKernelGGL((k_solve_islands<ISL_T_BIG, ISL_B_BIG, false>), grid, dim3(ISL_T_BIG), 0, stream, v, iv, bodies, joints, cps, ci, pi);
    else if (half_state)         hipLaunchKernelGGL((k_solve_islands<ISL_T, ISL_B, true>), grid, dim3(ISL_T), 0, stream, v, iv, bodies, joints, cps, ci, pi);
    else                         hipLaunchKernelGGL((k_solve_islands<ISL_T, ISL_B, false>), grid, dim3(ISL_T), 0, stream, v, iv, bodies, joints, cps, ci, pi);
}

} // namespace phx
