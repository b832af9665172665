/*
 * phyx_amd.h — C ABI of the MI355X-native contact solver + sweep-and-prune broadphase.
 *
 * The reference (zeux/phyx) has no plugin/FFI layer; its de-facto boundary is what World::Update
 * calls (ref: src/World.cpp:19-37): Collider::UpdateBroadphase / UpdatePairs (src/Collider.h:28-29),
 * Solver::SolveJoints (src/Solver.h:54) and the Configuration struct (src/Configuration.h:3-23),
 * all operating on the public POD arrays RigidBody (src/RigidBody.h:12-57, 128 B), ContactPoint
 * (src/Manifold.h:12-43, 32 B), Manifold (src/Manifold.h:45-67, 16 B) and ContactJoint
 * (src/Joints.h:6-23, 20 B).  Every entry point below names the reference interface it replaces
 * and takes those exact layouts, so buffers can cross the boundary unchanged.
 *
 * Conventions: plain pointers and sizes only; every function returns PHX_OK (0) or a negative
 * phx_status (the reference's void/assert convention is the one deliberate deviation);
 * phx_last_error() gives the message of the last failure on the calling thread.  A handle is not
 * re-entrant (like the reference's member scratch, ref: src/Solver.h:108-128): one thread per
 * handle at a time.  Pointers are borrowed for the duration of the call only.
 *
 * There is no CPU fallback: every compute entry point fails with PHX_ERR_NO_DEVICE when no
 * gfx950 device is usable.
 */
#ifndef PHYX_AMD_H
#define PHYX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHX_ABI_VERSION 2

typedef enum {
    PHX_OK = 0,
    PHX_ERR_INVALID = -1,     /* bad argument / size / enum value                         */
    PHX_ERR_NO_DEVICE = -2,   /* no usable HIP device                                      */
    PHX_ERR_HIP = -3,         /* a HIP runtime call failed (see phx_last_error)            */
    PHX_ERR_CAPACITY = -4,    /* caller-provided output buffer too small                   */
    PHX_ERR_STATE = -5        /* call order violated (e.g. query before first solve)       */
} phx_status;

/* ref: src/Configuration.h:5-18 — same numeric values */
enum { PHX_SOLVE_SCALAR = 0, PHX_SOLVE_SSE2 = 1, PHX_SOLVE_AVX2 = 2 };
enum { PHX_ISLAND_SINGLE = 0, PHX_ISLAND_MULTIPLE = 1, PHX_ISLAND_SINGLE_SLOPPY = 2, PHX_ISLAND_MULTIPLE_SLOPPY = 3 };

/* ref: src/Configuration.h:20-23.  On this backend the wavefront is the SIMD unit, so solve_mode
 * does not select a code path: every mode runs per-joint (scalar, N=1) skip semantics in the
 * device's own colour order.  island_mode picks the schedule:
 *   Single                      one coupled system (one group), solved class by class out of HBM;
 *   Multiple, Single Sloppy,    the island-aware schedule: connected components (GatherIslands semantics) binned
 *   Multiple Sloppy             into workgroup-sized groups solved out of LDS, each with its own early exit and its
 *                               own copy of the static bodies' tags; components too big for a workgroup go to one
 *                               trailing group solved out of HBM.
 * The reference's Sloppy modes promise no order (racy 512-joint batches, ref: src/Solver.cpp:138-139); here
 * they run that deterministic island-aware schedule — a legal outcome of those modes.  islandCount /
 * islandMaxSize are published only by the two Multiple modes (1 / joint count otherwise), like the reference. */
typedef struct {
    int32_t solve_mode;
    int32_t island_mode;
    int32_t contact_iterations;
    int32_t penetration_iterations;
} phx_config;

/* byte-exact POD records (static_asserted in the implementation) */
typedef struct { float x, y; } phx_vec2;
typedef struct {                                  /* ref: src/RigidBody.h:12-57 */
    uint32_t index;
    phx_vec2 geom_size, geom_xvector, geom_yvector, geom_pos, aabb_min, aabb_max;
    phx_vec2 velocity, acceleration, displacing_velocity;
    float    angular_velocity, angular_acceleration, displacing_angular_velocity;
    float    inv_mass, inv_inertia;
    phx_vec2 xvector, yvector, pos;
    int32_t  last_iteration, last_displacement_iteration;
} phx_rigid_body;                                 /* 128 B */
typedef struct {                                  /* ref: src/Manifold.h:12-43 */
    phx_vec2 delta1, delta2, normal;
    uint8_t  is_merged, is_newly_created, pad_[2];
    int32_t  solver_index;
} phx_contact_point;                              /* 32 B */
typedef struct { int32_t body1, body2, point_count, point_index; } phx_manifold;   /* ref: src/Manifold.h:45-67, 16 B */
typedef struct {                                  /* ref: src/Joints.h:6-23 */
    int32_t contact_point_index, body1, body2;
    float   normal_accumulated_impulse, friction_accumulated_impulse;
} phx_contact_joint;                              /* 20 B */
typedef struct { float minx, maxx, centery, extenty; uint32_t index; } phx_broadphase_entry; /* ref: src/Collider.h:45-50, 20 B */
typedef struct { uint32_t value, index; } phx_sort_entry;                                     /* ref: src/Collider.h:52-56, 8 B  */

/* ---------------------------------------------------------------------------------------------- */
/* library                                                                                         */
int          phx_abi_version(void);
/* The arithmetic of the sweeps (PreStepJoints, SolveJointsImpulses, SolveJointsDisplacement; ref: src/Solver.cpp:697-1018), fixed
 * when the library is built.  The reference writes `dV -= projector * velocity` / `velocity += compMass * dImpulse` and ships
 * -ffast-math -mfma (ref: Makefile:11, 17-24), which leaves fusing those pairs to its compiler; this backend states it:
 *   PHX_ARITH_FUSED   every such multiply-add pair is one fused multiply-add (fmaf), in the reference's source order — the default;
 *   PHX_ARITH_SOURCE  a rounded product and a rounded sum, as the source spells it (`PHX_ARITH=source python -m phyx_amd.build`).
 * RefreshJoints and everything outside the sweeps is source order in both.  oracle/ has both forms; parity is bit-exact against
 * the matching one, and the two forms stay within SURVEY.md section 8(c)'s tolerances of each other (tests/test_arith_modes.py). */
enum { PHX_ARITH_SOURCE = 0, PHX_ARITH_FUSED = 1 };
int          phx_arith_mode(void);
const char*  phx_last_error(void);
int          phx_device_count(void);              /* >=0, or a negative phx_status */
/* name / CU count / LDS per workgroup of `device`; name_cap bytes incl. NUL */
int          phx_device_info(int device, char* name, int name_cap, int* compute_units, int* lds_bytes, int64_t* hbm_bytes);

/* ---------------------------------------------------------------------------------------------- */
/* Solver — replaces Solver::SolveJoints (ref: src/Solver.h:54, src/Solver.cpp:17-119)             */
typedef struct phx_solver phx_solver;

int  phx_solver_create(phx_solver** out, int device);
void phx_solver_destroy(phx_solver* s);

/* Drop-in for Solver::SolveJoints: HOST arrays in the reference's layouts.  Reads per body
 * invMass/invInertia/coords.pos and the four velocity fields, per contact point delta1/delta2/
 * normal, per joint indices + accumulated impulses; writes back bodies[i].velocity /
 * angularVelocity / displacingVelocity / displacingAngularVelocity (ref: Solver.cpp:488-492) and
 * the joints' two accumulated impulses (ref: Solver.cpp:543-544).  Uploads, solves on the device,
 * downloads — the PCIe cost is part of this call. */
int phx_solver_solve(phx_solver* s, phx_rigid_body* bodies, int32_t body_count,
                     const phx_contact_point* contact_points, int32_t contact_point_count,
                     phx_contact_joint* joints, int32_t joint_count, const phx_config* config);

/* Same computation on DEVICE-resident records (same layouts, HBM pointers), asynchronous on the
 * solver's stream; nothing crosses PCIe except the solve's control word.  The records are converted to the
 * resident arrays below and back around the solve (two extra passes over the bodies): the drop-in for a caller
 * that keeps the reference's RigidBody array in HBM. */
int phx_solver_solve_device(phx_solver* s, void* d_bodies, int32_t body_count,
                            const void* d_contact_points, int32_t contact_point_count,
                            void* d_joints, int32_t joint_count, const phx_config* config);
int phx_solver_synchronize(phx_solver* s);

/* The RESIDENT form of body state (north star: "bodies ... laid out SoA in HBM with coalesced loads").  The reference stages
 * exactly these fields into solver-side arrays on every call (PrepareBodies / FinishBodies, ref: src/Solver.cpp:456-494:
 * SolveBody {velocity, angularVelocity, lastIteration}, SolveBodyParams {invMass, invInertia, coords}); here the staged form is
 * what a resident pipeline KEEPS — this library's World does, and bench.py times this entry point — three device arrays of one
 * 16-byte granule per body:
 *   vel[b]  = {velocity.x, velocity.y, angularVelocity, 0}                        read + written in place
 *   dvel[b] = {displacingVelocity.x, .y, displacingAngularVelocity, 0}            read + written in place
 *   mpos[b] = {invMass, invInertia, pos.x, pos.y}                                 read only
 * phx_bodies_to_view / phx_view_to_bodies convert between a device array of records and a view (queued on `stream`, a
 * hipStream_t, e.g. phx_solver_stream; the second writes the four velocity fields only, like FinishBodies). */
typedef struct { void* vel; void* dvel; void* mpos; } phx_body_view;
int phx_solver_solve_resident(phx_solver* s, const phx_body_view* bodies, int32_t body_count,
                              const void* d_contact_points, int32_t contact_point_count,
                              void* d_joints, int32_t joint_count, const phx_config* config);
int phx_bodies_to_view(int device, const void* d_bodies, int32_t body_count, const phx_body_view* out, void* stream);
int phx_view_to_bodies(int device, const phx_body_view* in, int32_t body_count, void* d_bodies, void* stream);

/* Precision ablation (BASELINE config 5): 32 (default) keeps the solver-side body state {velocity, angular velocity}
 * in fp32 like the reference's SolveBody (ref: src/Solver.h:95-101); 16 stores it as IEEE half between joint updates
 * (arithmetic stays fp32, every store rounds to nearest-even) in the groups solved out of LDS.  Not a drop-in mode:
 * results differ from the reference's by the rounding; tests compare against an oracle that rounds the same way. */
int phx_solver_set_body_state_bits(phx_solver* s, int32_t bits);

/* Island sharding across ranks: the schedule's groups (phx_solver_get_groups) are body-disjoint, so rank `shard` of
 * `shard_count` sweeps only the groups it owns (phx_exchange_layout's deal: longest processing time first by joint count;
 * the trailing HBM group counts as group lds_count) and leaves every other body and joint untouched.  All ranks must be given the same joints; the union of
 * their results is the unsharded result, bit for bit.  Default 0 / 1 = everything. */
int phx_solver_set_shard(phx_solver* s, int32_t shard, int32_t shard_count);

/* 1 (default): a solve whose joint topology fingerprint matches the cached schedule reuses it (checked on the device,
 * see DESIGN.md §4.1).  0: every solve rebuilds its schedule, like the reference rebuilds PrepareIndices / GatherIslands on
 * every call (ref: src/Solver.cpp:77, 135) — the cost a world whose contact graph changes every step pays. */
int phx_solver_set_schedule_reuse(phx_solver* s, int32_t on);

/* Diagnostics: with tracing on, every workgroup of the island kernel stamps the 100 MHz wall clock at its phase boundaries.
 * phx_solver_get_island_trace copies 8 words per LDS group of the last solve: [0] start, [1] records loaded, [2] refreshed,
 * [3] pre-stepped, [4] swept, [5] written back, [6] XCC id, [7] colours << 32 | impulse sweeps executed.  *groups receives
 * the group count (out may be NULL to query it).  on: 0 = off, 1 = the phase stamps only (a few stores per workgroup: the kernel
 * keeps its speed), any other value = also the per-wave cycle counts of every class step (phx_solver_get_wave_trace; ~15 % slower). */
int phx_solver_set_trace(phx_solver* s, int32_t on);
int phx_solver_get_island_trace(phx_solver* s, uint64_t* out, int32_t cap_groups, int32_t* groups);
/* per wave of every LDS group (groups x *waves_per_group x 8 words): shader cycles spent in colour steps in which the wave
 * worked {[0] in the joint update with at most 32 lanes active, [1] at the barrier behind it}, [2] cycles of the steps it only
 * waited in, [3] working steps with at most 32 lanes << 32 | idle steps, [4] cycles in the joint update with more than 32
 * lanes active, [5] such steps */
int phx_solver_get_wave_trace(phx_solver* s, uint64_t* out, int32_t cap_words, int32_t* waves_per_group);

/* Post-solve exchange of an island-sharded solve (BASELINE config 3; counterpart of the reference merging every island's
 * bodies back after its parallel island loop, ref: src/Solver.cpp:86-91, 482-494, 527-547).  Every rank holds a replica
 * of the inputs and solves only its own groups; afterwards
 *   phx_solver_exchange_pack     queues kernels that pack this rank's results (6 floats per body, 2 per joint of its
 *                                groups, behind a 32-byte header) into the send buffer and returns the segment size —
 *                                the same on every rank, because the layout is a pure function of the schedule;
 *   the CALLER all-gathers       segment_bytes from every rank's send buffer into the recv buffer (rank r at byte offset
 *                                r * segment_bytes) on phx_solver_stream() — RCCL over xGMI on a GPU node;
 *   phx_solver_exchange_unpack   queues kernels that scatter the other ranks' results into this rank's arrays and check
 *                                every peer's header (step serial, status word, topology fingerprint).
 * After the unpack all replicas are bit-identical to the unsharded solve.  Buffers are caller-owned device memory
 * (16-byte aligned; recv holds shard_count segments of segment_capacity_bytes, a multiple of 256).  status_word != 0 in
 * pack tells the peers that this rank failed earlier in the step.  phx_solver_exchange_status synchronises and returns
 * the OR of PHX_XCH_* bits seen by the unpacks so far (0 = every exchange was consistent). */
enum { PHX_XCH_PEER_ERROR = 1, PHX_XCH_SERIAL_MISMATCH = 2, PHX_XCH_TOPOLOGY_MISMATCH = 4, PHX_XCH_BAD_SEGMENT = 8 };
int    phx_solver_set_exchange_buffers(phx_solver* s, void* d_send, void* d_recv, size_t segment_capacity_bytes);
int    phx_solver_exchange_pack(phx_solver* s, const void* d_bodies, const void* d_joints, int32_t status_word, size_t* segment_bytes);
int    phx_solver_exchange_unpack(phx_solver* s, void* d_bodies, void* d_joints);
int    phx_solver_exchange_status(phx_solver* s, int32_t* status);
size_t phx_solver_exchange_segment_bytes(phx_solver* s);      /* of the last pack */
/* Host-only: who solves what, and the segment layout.  The groups (body table of group_bodies[g] entries, group_slots[g] joints)
 * are dealt to the ranks longest-processing-time first: by decreasing joint count (ties: group number), each to the rank with
 * the fewest joints so far (ties: lowest rank) — round-robin on uniform columns, balanced on anything else; group_owner[g] = that
 * rank.  group_offset_words[g] = 32-bit word offset of the group's block inside its owner's segment (a rank's groups in ascending
 * group order), rank_words[r] = words rank r actually fills (all three optional), *segment_words = the common padded length. */
int    phx_exchange_layout(const int32_t* group_bodies, const int32_t* group_slots, int32_t group_count, int32_t shard_count,
                           int32_t* group_owner, int64_t* group_offset_words, int64_t* rank_words, int64_t* segment_words);

/* ---------------------------------------------------------------------------------------------- */
/* Native transport of the island-sharded solve: an RCCL communicator, one process per GPU (xGMI between them).  The        */
/* reference's islands are solved by threads of one process and merged in its one address space (ref: src/Solver.cpp:86-91, */
/* 482-494, 527-547); across GPUs that merge is one all-gather per step on the solver's stream, which is also the per-step   */
/* barrier.  RCCL is resolved at run time (librccl.so.1; PHX_RCCL_LIB overrides): no link dependency, PHX_ERR_NO_DEVICE when */
/* it cannot be loaded.  Rendezvous is the caller's: rank 0 calls phx_comm_unique_id and hands the PHX_COMM_ID_BYTES bytes   */
/* to every rank by any out-of-band channel (a file, a socket, MPI, torch.distributed's store), then every rank creates.    */
#define PHX_COMM_ID_BYTES 128
typedef struct phx_comm phx_comm;
int  phx_comm_unique_id(void* out_id);                                  /* ncclGetUniqueId */
/* ncclCommInitRank (collective).  A rank whose peers do not arrive within PHX_COMM_TIMEOUT_S seconds (environment, default 120) gives */
/* up with PHX_ERR_STATE instead of hanging; the same bound holds wherever this library blocks the host on a collective.            */
int  phx_comm_create(phx_comm** out, const void* unique_id, int32_t rank, int32_t nranks, int device);
void phx_comm_destroy(phx_comm* c);
int  phx_comm_rccl_version(void);                                       /* ncclGetVersion of the RCCL in use, 0 if none could be loaded */
int  phx_comm_rank(phx_comm* c);
int  phx_comm_size(phx_comm* c);
/* bytes_per_rank from every rank's d_send into d_recv (rank r at r * bytes_per_rank), queued on `stream` (a hipStream_t) */
int  phx_comm_all_gather(phx_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream);
int  phx_comm_barrier(phx_comm* c, void* stream);                       /* 4-byte all-reduce + stream wait: the pure barrier */
int  phx_comm_barrier_async(phx_comm* c, void* stream);                 /* the same all-reduce, only queued: work queued on `stream` behind it waits for every rank */
int  phx_comm_async_error(phx_comm* c, int32_t* error);                 /* ncclCommGetAsyncError: 0 = healthy */
/* attach to a solver: phx_solver_bench then runs pack -> all-gather -> unpack natively in every step (exchange buffers must be set) */
int  phx_solver_set_comm(phx_solver* s, phx_comm* c);

/* results of the last solve (valid after a synchronizing call) — counterparts of
 * Solver::islandCount / islandMaxSize (ref: src/Solver.h:105-106) plus executed sweep counts */
typedef struct {
    int32_t island_count, island_max_size;
    int32_t colour_count;
    int32_t impulse_iterations;        /* sweeps executed before the no-productive-joint exit (ref: Solver.cpp:189) */
    int32_t displacement_iterations;   /* ref: Solver.cpp:210 */
    int32_t lds_islands;               /* islands solved by the one-workgroup-per-island kernel */
    int32_t recoloured;                /* 1 if the joint topology changed and the schedule was rebuilt; 2 if that rebuild ran without a host
                                          round trip (bins made on the device with last build's bin count as the launch grid) */
    int32_t graph_replay;              /* 1 if the launch sequence was replayed from cached hipGraphs */
    double  device_ms;                 /* HIP-event time of the device work of the last solve */
    int64_t joint_visits;              /* joints swept by the impulse loop, skipped ones included; per-island early exits honoured */
} phx_solve_stats;
int phx_solver_get_stats(phx_solver* s, phx_solve_stats* out);

/* The schedule the device used: order[k] = index of the joint occupying slot k; colour (class) c owns slots
 * [colour_offsets[c], colour_offsets[c+1]).  Sweeping the slots front to back with the reference's
 * scalar loop reproduces the device result bit for bit (tests/ feeds this to the oracle).
 * A class is a set of UNITS that share no dynamic body; a unit is the one or two joints of a body pair (the
 * two contact points of a manifold).  Its slots are laid out as: the leaders that have a follower (joint
 * order), the single leaders (joint order), then the followers in their leaders' order — a lane of the
 * device sweeps a leader and then its follower on one read and one write of the two bodies. */
int phx_solver_get_schedule(phx_solver* s, int32_t* order, int32_t order_cap,
                            int32_t* colour_offsets, int32_t offsets_cap, int32_t* colour_count);
/* Groups of the schedule: group g owns slots [group_offsets[g], group_offsets[g+1]) and is an independent
 * Gauss-Seidel problem (body-disjoint from every other group, static bodies aside) with its own early exit and
 * its own copy of the static bodies' lastIteration tags — the counterpart of the reference's islands
 * (ref: Solver.cpp:86-91).  The first *lds_group_count groups ran one-workgroup-per-group out of LDS, the rest
 * (at most one) class by class out of HBM.  Single island mode always reports one group. */
int phx_solver_get_groups(phx_solver* s, int32_t* group_offsets, int32_t offsets_cap, int32_t* group_count, int32_t* lds_group_count);

/* Partitioned components: a connected component of more than 1024 joints — a settled pile — has its units
 * split into INTERIOR ones (both bodies dynamic and in the same block of 512 consecutive body indices: level 0; or, failing that,
 * in the same block of that grid shifted by 256: level 1) and the rest; the interior units occupy the first `interior_classes`
 * classes of the schedule's last group (level 0 before level 1) and are swept by ONE launch per level and sweep (a workgroup per
 * block, `parts` of them over both levels; 0 when the schedule has no such component or PHX_NO_PARTS=1).  `sweep_launches`:
 * kernel launches of the last solve's sweeps. */
int phx_solver_get_partition(phx_solver* s, int32_t* interior_classes, int32_t* parts, int32_t* sweep_launches);
/* The lanes the island kernel gave the units of the last solve's LDS groups (execution detail, for tests and diagnostics: the classes'
 * lane ranges are placed on wave boundaries where the lanes allow it, see phx_schedule_groups): per unit the slot of its leader in
 * phx_solver_get_schedule's order and its lane inside its group's workgroup; *count = units (pass NULL arrays to ask for it). */
int phx_solver_get_lanes(phx_solver* s, int32_t* leader_slot, int32_t* lane, int32_t cap, int32_t* count);

/* RefreshJoints output for joint `joint_index` of the last solve (ref: Solver.cpp:592-695), expanded
 * to the reference's 30-float ContactJointPacked<1> order: normal limiter 13, 0, dstVelocity,
 * dstDisplacingVelocity, accumulatedDisplacingImpulse(after solve), friction limiter 13. */
int phx_solver_get_refreshed(phx_solver* s, int32_t joint_index, float out30[30]);

/* Host-only schedule builders (no device needed) — exposed so the ordering logic can be checked on
 * its own.  phx_schedule_colours is this backend's counterpart of Solver::PrepareIndices
 * (ref: src/Solver.cpp:217-273).  Joints are first paired into units: two joints whose priority ids
 * differ in the lowest bit only (contact points 2m and 2m + 1 of manifold m) and whose bodies are the
 * same form a unit, led by the even id; every other joint is a unit of its own.  The units are
 * partitioned into classes that share no dynamic body (is_static[b] != 0 exempts body b): they take
 * their class first-fit in order of DECREASING phx_schedule_priority(priority_ids[j], j, min(body1[j],
 * body2[j])) of their leader j — units whose lower body index is even first, inside a parity a fixed
 * pseudo-random order: the device reaches the same classes in a few parallel rounds (two on a stacked
 * column), whereas joint-index order needs one round per box of a stacked column — with
 * two candidates per connected component (smallest free class / two-ended) of which the component keeps
 * the one that needs fewer classes.  The layout of a class is described at phx_solver_get_schedule.
 * priority_ids may be NULL (the joint index is used); the solver passes each joint's
 * contactPointIndex, which survives compaction of the joint list (island sharding).
 * phx_schedule_islands follows Solver::GatherIslands (ref: src/Solver.cpp:285-454):
 * joint_island[j] = coalesced island of joint j (-1 if both bodies are static), island_size[i] =
 * joints in island i; returns the island count. */
int phx_schedule_colours(const int32_t* body1, const int32_t* body2, int32_t joint_count,
                         const uint8_t* is_static, int32_t body_count, const int32_t* priority_ids,
                         int32_t* order, int32_t* colour_offsets, int32_t offsets_cap, int32_t* colour_count);
int phx_schedule_islands(const int32_t* body1, const int32_t* body2, int32_t joint_count,
                         const uint8_t* is_static, int32_t body_count,
                         int32_t* joint_island, int32_t* island_size, int32_t island_cap);
/* Host-only: the island-mode schedule (ref: Solver.cpp:285-454 GatherIslands + :217-273 PrepareIndices) as the workgroup-sized groups
 * this backend solves out of LDS: connected components binned into groups of at most `lanes` units / 2 * lanes joints / body_cap
 * bodies (whatever does not fit goes to one trailing group); the classes of a group are coloured like phx_schedule_colours colours a
 * component.  order / colour_offsets as there; group_offsets (slots) and group_first_colour have groups + 1 entries, *lds_groups of
 * the groups are LDS groups; per unit of the LDS groups (class-major, group by group; the return value is their number): the slot of
 * its leader and its lane in the island kernel — the classes' lane ranges are placed on wave boundaries where the lanes allow it
 * (a class costs one pass of every wave it has a lane in); lanes are execution detail, no result depends on them. */
int phx_schedule_groups(const int32_t* body1, const int32_t* body2, int32_t joint_count, const uint8_t* is_static, int32_t body_count,
                        const int32_t* priority_ids, int32_t lanes, int32_t body_cap, int32_t* order, int32_t* colour_offsets,
                        int32_t offsets_cap, int32_t* colour_count, int32_t* group_offsets, int32_t* group_first_colour,
                        int32_t groups_cap, int32_t* lds_groups, int32_t* unit_lane, int32_t* unit_leader_slot);
uint64_t phx_schedule_priority(uint32_t priority_id, uint32_t joint_index, uint32_t lower_body);

/* ---------------------------------------------------------------------------------------------- */
/* Broadphase — replaces Collider::UpdateBroadphase + UpdatePairs                                   */
/* (ref: src/Collider.h:28-29, src/Collider.cpp:251-366, src/base/RadixSort.h:19-95)               */
typedef struct phx_broadphase phx_broadphase;

int  phx_broadphase_create(phx_broadphase** out, int device);
void phx_broadphase_destroy(phx_broadphase* b);
/* forget every persistent pair (ref: main.cpp:88 manifoldMap.clear()) */
int  phx_broadphase_clear(phx_broadphase* b);

/* UpdateBroadphase + UpdatePairs on HOST bodies (reads geom.aabb only).  new_pairs receives the
 * pairs that were not yet in the persistent pair set, as (index_i, index_j) in the reference's
 * serial emission order (ref: Collider.cpp:296-318), and inserts them into the set.
 * *new_pair_count always receives the full count (PHX_ERR_CAPACITY if it exceeds the cap). */
int phx_broadphase_update(phx_broadphase* b, const phx_rigid_body* bodies, int32_t body_count,
                          uint32_t* new_pairs, int32_t new_pairs_cap, int32_t* new_pair_count);
int phx_broadphase_update_device(phx_broadphase* b, const void* d_bodies, int32_t body_count);
/* after an update: broadphaseSort[1] and broadphase[] of the reference (ref: Collider.h:64-65) */
int phx_broadphase_get_sorted(phx_broadphase* b, phx_sort_entry* sorted, phx_broadphase_entry* entries, int32_t cap);
/* pairs emitted by the last update (device variant leaves them on the device until asked) */
int phx_broadphase_get_new_pairs(phx_broadphase* b, uint32_t* new_pairs, int32_t cap, int32_t* count);
/* remove pairs from the persistent set (ref: Collider.cpp:391 manifoldMap.erase) */
int phx_broadphase_erase_pairs(phx_broadphase* b, const uint32_t* pairs, int32_t pair_count);
typedef struct {
    int64_t candidate_tests;      /* y-overlap tests executed by the sweep (20 B each) */
    int64_t overlapping_pairs;    /* candidates that passed the y test                               */
    int32_t new_pairs;
    int32_t set_size;             /* persistent pairs after the update                               */
    double  device_ms;
} phx_broadphase_stats;
int phx_broadphase_get_stats(phx_broadphase* b, phx_broadphase_stats* out);

/* ---------------------------------------------------------------------------------------------- */
/* World — replaces World (ref: src/World.h:9-36, src/World.cpp:11-37)                              */
typedef struct phx_world phx_world;

int  phx_world_create(phx_world** out, int device);
void phx_world_destroy(phx_world* w);
/* World::AddBody (ref: World.cpp:11-17; RigidBody ctor RigidBody.h:15-36, density 1e-5); returns index */
int  phx_world_add_body(phx_world* w, float px, float py, float angle, float half_x, float half_y);
/* main.cpp:91-93 groundBody->invMass = invInertia = 0 */
int  phx_world_set_body_static(phx_world* w, int32_t body);
/* body->invMass = ..., body->invInertia = ... on a public RigidBody (the demo scenes pin shelves with invMass = 0 only, so they
 * still rotate: main.cpp:176-177, 194-195; a body is static — exempt from islands — only when both are 0, ref: Solver.cpp:304) */
int  phx_world_set_body_inverse_mass(phx_world* w, int32_t body, float inv_mass, float inv_inertia);
int  phx_world_set_gravity(phx_world* w, float gravity);            /* ref: World.h:35 */
/* Multi-GPU island sharding: every rank steps a replica of the same world and solves only the schedule groups g with
 * g % shard_count == shard (phx_solver_set_shard; the default 0/1 solves everything).  A sharded world (shard_count > 1)
 * steps in two halves around the caller's all-gather (see phx_solver_exchange_pack; the buffers are set on
 * phx_world_solver(w) with phx_solver_set_exchange_buffers):
 *   phx_world_step_begin   World::Update up to and including SolveJoints of this rank's groups, then the pack;
 *   all-gather             of *segment_bytes per rank, send buffer -> recv buffer, on phx_world_stream(w);
 *   phx_world_step_end     scatter of the other ranks' results, then IntegratePosition.
 * phx_world_update / phx_world_finish_step refuse to run on a sharded world (PHX_ERR_STATE): without the exchange they
 * would integrate the other ranks' bodies with unsolved velocities. */
int  phx_world_set_shard(phx_world* w, int32_t shard, int32_t shard_count);
int  phx_world_step_begin(phx_world* w, float dt, const phx_config* config, size_t* segment_bytes);
int  phx_world_step_end(phx_world* w, float dt);
void* phx_world_stream(phx_world* w);       /* the hipStream_t all of the world's work is queued on */
/* The same step with the native transport: phx_world_set_comm makes this world rank phx_comm_rank(c) of phx_comm_size(c)
 * (the world owns and grows its exchange buffers), and phx_world_step_sharded is World::Update of the sharded world in ONE
 * call — step_begin, ncclAllGather of the segments on the world's stream, step_end — with no host wait around the collective.
 * Every 16th step (and on phx_world_check_exchange) the peers' headers and ncclCommGetAsyncError are checked: PHX_ERR_STATE
 * means a peer failed, is at another step or solved another topology (phx_solver_exchange_status has the bits). */
int  phx_world_set_comm(phx_world* w, phx_comm* c);
int  phx_world_step_sharded(phx_world* w, float dt, const phx_config* config);
int  phx_world_check_exchange(phx_world* w);
int  phx_world_update(phx_world* w, float dt, const phx_config* config);   /* ref: World.cpp:19-37 */
/* phx_world_update returns once the step is QUEUED on the world's stream (the host waits only where it needs a count
 * to size a launch); every getter synchronises before it reads.  This waits for the device explicitly. */
int  phx_world_synchronize(phx_world* w);
/* World::Update split at the solver boundary: pre_solve = everything before Solver::SolveJoints
 * (ref: World.cpp:25-32), after which bodies / contact points / joints are exactly the solver's inputs;
 * finish_step = SolveJoints + IntegratePosition (ref: World.cpp:34-36).  pre_solve + finish_step == update. */
int  phx_world_pre_solve(phx_world* w, float dt);
int  phx_world_finish_step(phx_world* w, float dt, const phx_config* config);
int  phx_world_counts(phx_world* w, int32_t* bodies, int32_t* manifolds, int32_t* contact_points, int32_t* joints);
int  phx_world_get_bodies(phx_world* w, phx_rigid_body* out, int32_t cap);
int  phx_world_get_manifolds(phx_world* w, phx_manifold* out, int32_t cap);
int  phx_world_get_contact_points(phx_world* w, phx_contact_point* out, int32_t cap);
int  phx_world_get_joints(phx_world* w, phx_contact_joint* out, int32_t cap);
/* Restore a world from what the four getters above returned (checkpoint / resume; the hand-over of bodies between the ranks of an
 * ownership-sharded world): bodies, the contact cache — manifolds with their two contact-point slots each, ref: Collider.h:57-58 —
 * and the joints with their warm-start impulses (ref: World.h:33).  The broadphase's pair set is rebuilt from the manifolds'
 * body pairs.  A world restored from a saved state steps exactly like the one it was saved from.  Checked: contact_point_count ==
 * 2 * manifold_count, manifold i owns slots 2i and 2i + 1, every joint's bodies are its manifold's and its contact point's
 * solver_index points back at it; PHX_ERR_INVALID otherwise.  Any number of bodies (the world's previous content is dropped). */
int  phx_world_set_state(phx_world* w, const phx_rigid_body* bodies, int32_t body_count, const phx_manifold* manifolds, int32_t manifold_count,
                         const phx_contact_point* contact_points, int32_t contact_point_count, const phx_contact_joint* joints, int32_t joint_count);
int  phx_world_get_solve_stats(phx_world* w, phx_solve_stats* out);
int  phx_world_get_broadphase_stats(phx_world* w, phx_broadphase_stats* out);
/* handles owned by the world (for stage-level queries after an update) */
phx_solver*     phx_world_solver(phx_world* w);
phx_broadphase* phx_world_broadphase(phx_world* w);
/* per-phase wall/device milliseconds of the last update, in World::Update order:
 * 0 IntegrateVelocity 1 UpdateBroadphase 2 UpdatePairs 3 UpdateManifolds 4 PackManifolds
 * 5 RefreshContactJoints 6 SolveJoints 7 IntegratePosition */
int  phx_world_get_phase_ms(phx_world* w, double out8[8]);
/* [min x, max x] over the AABBs of the dynamic bodies, reduced on the device.  An ownership-sharded run (one world per GPU, each
 * holding the islands of one x-slab: phyx_amd/dist.py SlabWorld) checks it against its slab: as long as no body
 * leaves its slab no island can span two ranks and the ranks need nothing from each other but the per-step barrier. */
int  phx_world_x_extent(phx_world* w, float out2[2]);
/* RE-SLAB of an ownership-sharded run (csrc/reslab.hip): the collective hand-over of bodies between the ranks' worlds
 * when a body has reached its slab's boundary.  Every rank calls phx_world_reslab at the same step with the scene indices of its world's
 * bodies (static bodies of the scene live on every rank).  Phase 1: the ranks all-gather {scene index, x-interval} of their dynamic bodies
 * (two bodies that share a manifold cover each other's interval), cut the x axis anew in the gaps no interval covers (phx_reslab_cuts:
 * blocks of overlapping intervals are never split, the cuts balance the body counts) and see whether anybody changes owner; if nobody
 * does, only `bounds` changes (*moved = 0) and the world keeps its allocations, its cached schedule and its broadphase state.  Phase 2,
 * only otherwise: the ranks all-gather their worlds' states (what phx_world_set_state restores, warm-start impulses included) and every
 * rank restores the part of the union world that lives in its new slab (*moved = 1; global_index / *body_count describe the new world).
 * The hand-over is lossless: with unchanged cuts every rank gets its world back byte for byte.
 * Transport: `comm` — the library's RCCL communicator, collectives on device buffers — or, where none can exist (several ranks on one
 * GPU, another process group), the two host callbacks; transport == NULL or size 1: one rank, nothing is exchanged. */
typedef struct {
    int32_t rank, size;
    phx_comm* comm;                                                                        /* may be NULL: the callbacks are used */
    int (*all_gather)(void* user, const void* send, void* recv, size_t bytes_per_rank);  /* host buffers; rank r's share at r * bytes_per_rank; 0 = ok */
    int (*all_reduce_max)(void* user, int64_t* value);                                    /* in place; 0 = ok */
    void* user;
} phx_slab_transport;
int  phx_world_reslab(phx_world* w, const phx_slab_transport* transport, int64_t* global_index, int32_t capacity, int32_t* body_count, int32_t scene_size,
                      double margin, double bounds[2], int32_t* moved);
/* the phases on their own (no device needed for the second and third): this world's dynamic bodies' scene indices and widened x-intervals;
 * every rank's intervals -> sorted by scene index, with the new owner of every body and the ranks' bounds (2 per rank); the cuts themselves */
int  phx_world_reslab_intervals(phx_world* w, const int64_t* global_index, int32_t body_count, int64_t* gi, double* lo, double* hi, int32_t cap, int32_t* count);
int  phx_reslab_plan(int64_t* gi, double* lo, double* hi, int32_t n, int32_t nranks, double margin, int32_t* owner, double* bounds);
int  phx_reslab_cuts(const double* lo, const double* hi, int32_t n, int32_t nranks, double margin, int32_t* owner, double* bounds);
/* diagnostics: [0] steps whose PackManifolds count was settled together with the joint counts (the bet that no manifold dies),
 * [1] those of them that lost the bet (pack run late, joint match repeated), [2] solves repeated because the cached schedule was
 * stale or a group was left uncommitted, [3] third contact points dropped (ref: Collider.cpp:241-242 would overflow) */
int  phx_world_debug_counters(phx_world* w, int64_t out4[4]);
/* diagnostics of the schedule rebuild: [0] rebuilds whose connected components, joint counts and bins came from the MANIFOLDS (made on a
 * side stream while the joint list was refreshed), [1] rebuilds that took them from the joints.  The schedule is the same pure function of
 * the joints either way; PHX_NO_PRELABEL=1 forces [0] to stay 0. */
int  phx_world_build_counts(phx_world* w, int64_t out2[2]);
/* per-phase host timers cost one stream synchronisation per phase; off by default (get_phase_ms then returns the last
 * values measured while it was on) */
int  phx_world_set_phase_timing(phx_world* w, int32_t on);

/* ---------------------------------------------------------------------------------------------- */
/* measurement helpers used by bench.py: K solves queued back to back, HIP events on the handle's own stream */
typedef struct {
    double  total_ms;             /* events around the whole timed region                      */
    double  impulse_kernel_ms;    /* sum of the HIP-event brackets around the sweep launches of every 4th step (every step if steps < 8) */
    int64_t impulse_launches;     /* sweep kernels launched by all the steps                    */
    int64_t joint_visits;         /* joints swept by them (skipped joints count as visited)     */
    int64_t impulse_iterations;   /* sweeps executed                                            */
    int64_t bracketed_launches;   /* sweep kernels inside the brackets: impulse_kernel_ms / this = average launch */
} phx_bench_result;
/* runs `steps` solves of identical device-resident input and reports event timings.  Every step needs the input afresh
 * (a solve overwrites velocities and impulses): phx_solver_bench_stage, called BEFORE the caller starts its clock, makes
 * `steps` private copies of (bodies, joints) in HBM and the next bench call on the same arrays solves copy k in step k; without
 * staged copies (or for warm-up steps) a working copy is restored from the caller's arrays in front of every step, inside the
 * timed region (two copy dispatches per step). */
int phx_solver_bench_stage(phx_solver* s, const void* d_bodies, int32_t body_count, const void* d_joints, int32_t joint_count, int32_t steps);
int phx_solver_bench(phx_solver* s, const void* d_bodies, int32_t body_count, const void* d_contact_points,
                     int32_t contact_point_count, const void* d_joints, int32_t joint_count,
                     const phx_config* config, int32_t warmup, int32_t steps, phx_bench_result* out);
/* 64-bit position-sensitive checksum of what the LAST step of the last phx_solver_bench call left in its copy of the input
 * (velocities, displacing velocities, accumulated impulses): identical input + identical schedule => identical checksum, which is
 * how bench.py checks, outside its clock, that the timed solves computed what the warm-up solve computed. */
int phx_solver_bench_checksum(phx_solver* s, uint64_t* out);
/* Same, with a host callback around every step so that a multi-GPU caller can keep its ranks in lock step without the host
 * ever waiting inside the timed region.  hook(user, step, phase) is called
 *   phase 0  right after step `step` has been queued on the handle's stream (phx_solver_stream): start the per-step
 *            exchange here, ordered behind the step on that stream (e.g. an asynchronous RCCL all-reduce);
 *   phase 1  when the rank-local preparation of step `step` (input restore, topology fingerprint) is queued and its sweeps
 *            are not: make the stream wait for the exchange started after step - 1 here, so the exchange overlaps the
 *            preparation.  Called once more with step = `steps` after the last step, to drain the last exchange.
 *   phase 2  only when exchange buffers are set (phx_solver_set_exchange_buffers): step `step` is solved and its results
 *            are packed; run the all-gather of phx_solver_exchange_segment_bytes() on the stream now — the unpack is
 *            queued right after the hook returns.
 * Warm-up steps are numbered -warmup .. -1.  A nonzero return aborts the run with PHX_ERR_STATE. */
typedef int (*phx_step_hook)(void* user, int32_t step, int32_t phase);
int phx_solver_bench_hooked(phx_solver* s, const void* d_bodies, int32_t body_count, const void* d_contact_points,
                            int32_t contact_point_count, const void* d_joints, int32_t joint_count,
                            const phx_config* config, int32_t warmup, int32_t steps, phx_step_hook hook, void* user,
                            phx_bench_result* out);
/* the hipStream_t every launch of this handle goes to (as void*), for callers that order their own work against it */
void* phx_solver_stream(phx_solver* s);

/* raw device memory helpers so callers without a HIP binding (ctypes) can stage resident inputs */
int phx_device_malloc(int device, size_t bytes, void** out);
int phx_device_free(int device, void* p);
int phx_memcpy_h2d(int device, void* dst, const void* src, size_t bytes);
int phx_memcpy_d2h(int device, void* dst, const void* src, size_t bytes);
int phx_memcpy_d2d(int device, void* dst, const void* src, size_t bytes);
/* the same, ordered on `stream` (a hipStream_t such as phx_solver_stream / phx_world_stream): the host-touching copies
 * return when the data has arrived, the device-to-device copy returns when it is queued */
int phx_memcpy_d2h_on(int device, void* dst, const void* src, size_t bytes, void* stream);
int phx_memcpy_h2d_on(int device, void* dst, const void* src, size_t bytes, void* stream);
int phx_memcpy_d2d_on(int device, void* dst, const void* src, size_t bytes, void* stream);
/* diagnostics: with PHX_WAIT_CLOCK=1 in the environment, the time this process has spent waiting for device->host readbacks */
int phx_debug_wait_clock(long long* ns, long long* calls);

#ifdef __cplusplus
}
#endif
#endif /* PHYX_AMD_H */
