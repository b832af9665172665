// island_view.h — what the host code (solver.hip) and the island kernel's translation unit (islands.hip) share: the kernel's
// shapes, its argument block and its launcher.  The kernel lives in a translation unit of its own because it is compiled with
// -fno-slp-vectorize (build.py): the SLP vectoriser pairs the joint update's multiplies and adds into v_pk_mul_f32 /
// v_pk_add_f32 and pays for every pair with moves into adjacent registers — bit-identical results, 83.0 us with it, 78.3 us
// without — while the HBM path's kernels are a few per cent faster WITH it.
#pragma once

#include "common.h"

namespace phx {

struct SolverView;

constexpr int ISL_T = 256, ISL_B = 768;        // lanes = unit capacity of a group (joints: twice that); body capacity (dynamic + touched static)
constexpr int ISL_T_BIG = 512, ISL_B_BIG = 1024;

struct IslandView {
    const int4* desc;                 // per group {slot_begin, slot_count, body_begin, body_count}
    const int* ncol;                  // per group: classes
    const int* units;                 // per group: units
    // per group g, unit u (class-major): two 16-byte words at [2 * (g * T + u)] = {leader joint, follower joint or -1, leader's contact
    // point, follower's contact point}, {local body1 | local body2 << 16, class, -, -}: everything a lane needs to start its joint
    // and contact-point loads after ONE round trip (round 2's chain was descriptor -> unit -> order -> joint -> contact point)
    const int4* unit_recs;
    const int* bodies;                // global body ids, group-local order, group g's table at [g * NB]
    int* executed;                    // per slot (group % ISL_STAT_SLOTS): [2 * slot] max impulse sweeps run by a group, [2 * slot + 1] displacement
    unsigned long long* visits;       // per slot: sum over groups of impulse sweeps * joints
    const int* ngroups_dev;           // null, or the group count where the launch grid is only an upper bound of it (speculative binning, solver.hip)
    int first, stride;                // workgroup w solves group first + w * stride (island sharding across ranks; 0, 1 = all)
    int stamp_begin, stamp_end;       // this launch is the first / last kernel of the solve: it leaves the solve's time stamps (solver_kernels.h)
    unsigned long long* wave_trace;   // null, or 8 words per wave of every group: cycles {working with <= 32 lanes, at the barrier after work, idle steps}, counts, cycles working with > 32 lanes, count
    unsigned long long* trace;        // null, or 8 words per group: shader-clock stamps of the kernel's phases (phx_solver_set_trace)
};

// launches k_solve_islands<shape, body-state, trace> over `groups` workgroups (islands.hip)
void launch_solve_islands(hipStream_t stream, int groups, bool big_shape, bool half_state, bool trace, const SolverView& v, const IslandView& iv,
                          phx_rigid_body* bodies, phx_contact_joint* joints, const phx_contact_point* cps, int ci, int pi);

} // namespace phx
