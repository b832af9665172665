// island_view.h — what the host code (solver.hip) and the island kernel's translation unit (islands.hip) share: the kernel's
// shapes, its argument block and its launcher.  The kernel lives in a translation unit of its own because it is compiled with
// -fno-slp-vectorize (build.py): the SLP vectoriser pairs the joint update's multiplies and adds into v_pk_mul_f32 /
// v_pk_add_f32 and pays for every pair with moves into adjacent registers — bit-identical results, 83.0 us with it, 78.3 us
// without — while the HBM path's kernels are a few per cent faster WITH it.
#pragma once

#include "common.h"
#include "body_view.h"

namespace phx {

struct SolverView;

// How a launch of the island kernel decides whether it may write to the caller's arrays:
//   ISL_GATED     commit iff *v.fingerprint == v.expected_fingerprint — the word was settled by a kernel in FRONT of this launch
//                 (the topology hash of a schedule with an HBM group / of a sharded solve, or the schedule build itself);
//   ISL_VERIFY    the launch checks the cached schedule against the caller's arrays ITSELF: every workgroup compares the
//                 {contact point, body1, body2} of the joints it has just loaded, and the static-ness of its bodies, with what the
//                 schedule builder recorded for its units, and ARRIVES — with its verdict — on one of ISL_SHARDS counters (a thousand
//                 same-address device atomics serialise: ~12 ns each, and the polls at the end queue behind them); the last arriver of
//                 a shard forwards the shard's verdict to the solve's control word, which therefore reads ISL_SHARDS-many arrivals
//                 (+ ISL_BAD per shard that saw a difference) once EVERY workgroup has compared.  A workgroup commits only when it
//                 reads exactly that.  Verification costs nothing this way: the joints and the bodies are loaded by the set-up anyway
//                 (the separate hash pass it replaces was 13 % of a cfg-2 solve);
//   ISL_COMPLETE  solves the groups a verified launch left uncommitted (its bounded wait for the other workgroups ran out —
//                 it never does on an otherwise idle GPU): no check, unconditional commit.
enum { ISL_GATED = 0, ISL_VERIFY = 1, ISL_COMPLETE = 2 };
constexpr unsigned ISL_BAD = 1u << 20;              // (a verified launch has at most 1024 workgroups: 1024 * ISL_BAD < 2^32)
constexpr unsigned ISL_ARRIVE_MASK = ISL_BAD - 1u;
constexpr int ISL_SHARDS = 16, ISL_SHARD_STRIDE = 16;      // arrival counters of a verified launch: workgroup b arrives on shard b % 16; one 128-byte line each
constexpr unsigned long long ISL_TIMEOUT = 1ull << 40;     // control-word bit: some workgroup gave up waiting and left its group uncommitted
constexpr int ISL_WAIT_POLLS = 20000;               // bounded wait for the other workgroups' arrival (~1 us per poll); PHX_ISL_WAIT_POLLS overrides (tests)

// IslandView::units
__host__ __device__ inline int island_units_word(int units, int classes, int nstatic) { return units | (classes << 12) | (nstatic << 20); }
__host__ __device__ inline int island_word_units(int w) { return w & 0xFFF; }
__host__ __device__ inline int island_word_classes(int w) { return (w >> 12) & 0xFF; }
__host__ __device__ inline int island_word_static(int w) { return (int)((unsigned)w >> 20); }

constexpr int ISL_T = 256, ISL_B = 768;        // lanes = unit capacity of a group (joints: twice that); body capacity (dynamic + touched static)
constexpr int ISL_T_BIG = 512, ISL_B_BIG = 1024;

struct IslandView {
    const int4* desc;                 // per group {slot_begin, slot_count, body_begin, body_count}
    const int* ncol;                  // per group: classes
    const int* units;                 // per group: island_units_word(units, classes, static bodies of its table — they sit first in the table)
    // per group g and LANE l (schedule.h LANES): two 16-byte words at [2 * (g * T + l)] = {leader joint or -1 (nobody's lane), follower
    // joint or -1, leader's contact point, follower's contact point}, {local body1 | local body2 << 16, class, leader slot, follower
    // slot}: everything a lane needs to start its joint and contact-point loads after ONE round trip (round 2's chain was descriptor ->
    // unit -> order -> joint -> contact point)
    const int4* unit_recs;
    const int* bodies;                // global body ids, group-local order, group g's table at [g * NB]
    int* executed;                    // per slot (group % ISL_STAT_SLOTS): [2 * slot] max impulse sweeps run by a group, [2 * slot + 1] displacement
    unsigned long long* visits;       // per slot: sum over groups of impulse sweeps * joints
    const int* ngroups_dev;           // null, or the group count where the launch grid is only an upper bound of it (speculative binning, solver.hip)
    const int* group_list;            // null: workgroup w solves group w; else group_list[w] (island sharding across ranks: the groups this rank owns)
    int stamp_begin, stamp_end;       // this launch is the first / last kernel of the solve: it leaves the solve's time stamps (solver_kernels.h)
    unsigned long long* wave_trace;   // null, or 8 words per wave of every group: cycles {working with <= 32 lanes, at the barrier after work, idle steps}, counts, cycles working with > 32 lanes, count
    unsigned long long* trace;        // null, or 8 words per group: shader-clock stamps of the kernel's phases (phx_solver_set_trace)
    int mode;                         // ISL_GATED / ISL_VERIFY / ISL_COMPLETE
    unsigned nexpect;                 // ISL_VERIFY: workgroups of this launch (all of them must arrive)
    unsigned long long* ctl;          // the solve's control word (= v.fingerprint): complete shards | ISL_BAD * bad shards | ISL_TIMEOUT
    unsigned long long* shards;       // ISL_VERIFY: the arrival counters of this solve's control set (ISL_SHARDS words, ISL_SHARD_STRIDE apart)
    unsigned* done; unsigned epoch;   // ISL_VERIFY / ISL_COMPLETE: done[group] = epoch once the group's results are committed
    int wait_polls;                   // ISL_VERIFY: polls of the control word before a workgroup gives up
    // this launch is the first kernel of its solve: it clears the control set of the NEXT solve (two sets alternate; the hash
    // kernel does the same when it runs first) — null otherwise
    unsigned long long* next_ctl; int* next_executed; unsigned long long* next_visits; unsigned long long* next_shards;
};

// launches k_solve_islands<shape, body-state, trace> over `groups` workgroups (islands.hip)
void launch_solve_islands(hipStream_t stream, int groups, bool big_shape, bool half_state, bool trace, const SolverView& v, const IslandView& iv,
                          const BodyView& bodies, phx_contact_joint* joints, const phx_contact_point* cps, int ci, int pi);

// resident workgroups per CU of that instantiation (hipOccupancyMaxActiveBlocksPerMultiprocessor, cached; 0 if the query failed)
int island_blocks_per_cu(bool big_shape, bool half_state);

} // namespace phx
