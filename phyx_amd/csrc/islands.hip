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

// workgroups of the island kernel one CU holds at once (register, LDS and wave limits of the exact instantiation a verified
// launch would use): the workgroups of an ISL_VERIFY launch wait for each other, so all of them must be resident
int island_blocks_per_cu(bool big_shape, bool half_state)
{
    static int cached[2][2] = {{-1, -1}, {-1, -1}};
    int& c = cached[big_shape ? 1 : 0][half_state ? 1 : 0];
    if (c >= 0) return c;
    int n = 0;
    hipError_t e;
    if (big_shape && half_state) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_solve_islands<ISL_T_BIG, ISL_B_BIG, true>, ISL_T_BIG, 0);
    else if (big_shape)          e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_solve_islands<ISL_T_BIG, ISL_B_BIG, false>, ISL_T_BIG, 0);
    else if (half_state)         e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_solve_islands<ISL_T, ISL_B, true>, ISL_T, 0);
    else                         e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_solve_islands<ISL_T, ISL_B, false>, ISL_T, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    c = n;
    return c;
}

} // namespace phx
