/*
 * phx_oracle.h — CPU restatement of zeux/phyx's simulation step.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libphyx_amd.so, HIP)
 * never links or calls anything in oracle/.
 *
 * Parity pinning status
 *   PINNED   (against the real reference, compiled header-only from /root/reference by
 *             oracle/Makefile into oracle/_ref/, vectors in tests/golden/):
 *             radixFloat, radixSort3, std::hash<pair<u32,u32>>, RigidBody construction
 *             (mass/inertia/frame/AABB), Geom::RecomputeAABB, Geom::GetSupportPointSet,
 *             Vector2::Rotate, the ContactPoint constructor and ContactPoint::Equals, the plane form
 *             of ProjectPointToLine, AABB2::Intersects, the scalar / SSE2 / AVX2 flipsign, max and abs wrappers the
 *             solve loops use, DenseHashSet insert/contains (set
 *             semantics on tombstone-free sequences).
 *   UNPINNED ("parity unpinned"): everything that lives in the reference's .cpp files —
 *             Solver.cpp, Collider.cpp, World.cpp.  Those translation units include
 *             "microprofile.h", an un-vendored submodule (/root/reference/.gitmodules:1-3,
 *             src/microprofile/ is empty), so they cannot be built here without writing a
 *             stand-in header, which the build rules forbid.  The reference ships no tests,
 *             fixtures or golden vectors either.  For those functions this
 *             file is a line-cited restatement only.
 *
 * Arithmetic: strict IEEE-754 binary32, one rounding per operation, evaluated in the order the
 * reference source writes it (built with -ffp-contract=off, no fast-math).  The reference's own
 * Makefile uses -ffast-math -mfma, whose contraction choices are compiler-dependent; the strict
 * form is the canonical one both this oracle and the HIP kernels implement, which is what makes
 * bit-exact GPU-vs-oracle comparison possible.
 *
 * POD layouts below are byte-identical to the reference's (RigidBody.h:12-57, Manifold.h:12-67,
 * Joints.h:6-23) so that buffers can be handed to either side unchanged.
 */
#ifndef PHX_ORACLE_H
#define PHX_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y; } phxo_vec2;

/* RigidBody.h:12-57 — 128 bytes */
typedef struct {
    uint32_t  index;                                     /* @0   */
    phxo_vec2 geom_size;                                 /* @4   */
    phxo_vec2 geom_xv, geom_yv, geom_pos;                /* @12  Geom::coords */
    phxo_vec2 aabb_min, aabb_max;                        /* @36  Geom::aabb   */
    phxo_vec2 velocity, acceleration;                    /* @52, @60 */
    phxo_vec2 displacing_velocity;                       /* @68  */
    float     angular_velocity, angular_acceleration;    /* @76, @80 */
    float     displacing_angular_velocity;               /* @84  */
    float     inv_mass, inv_inertia;                     /* @88, @92 */
    phxo_vec2 xv, yv, pos;                               /* @96  RigidBody::coords */
    int32_t   last_iteration, last_displacement_iteration; /* @120, @124 */
} phxo_body;

/* Manifold.h:12-43 — 32 bytes */
typedef struct {
    phxo_vec2 delta1, delta2, normal;
    uint8_t   is_merged, is_newly_created, pad_[2];
    int32_t   solver_index;
} phxo_contact_point;

/* Manifold.h:45-67 — 16 bytes */
typedef struct { int32_t body1, body2, point_count, point_index; } phxo_manifold;

/* Joints.h:6-23 — 20 bytes */
typedef struct {
    int32_t contact_point_index, body1, body2;
    float   normal_acc, friction_acc;
} phxo_contact_joint;

/* Collider.h:45-56 */
typedef struct { float minx, maxx, centery, extenty; uint32_t index; } phxo_bp_entry;
typedef struct { uint32_t value, index; } phxo_sort_entry;

/* Configuration.h:5-18 (same numeric values) */
enum { PHXO_SOLVE_SCALAR = 0, PHXO_SOLVE_SSE2 = 1, PHXO_SOLVE_AVX2 = 2 };
enum { PHXO_ISLAND_SINGLE = 0, PHXO_ISLAND_MULTIPLE = 1, PHXO_ISLAND_SINGLE_SLOPPY = 2, PHXO_ISLAND_MULTIPLE_SLOPPY = 3 };

/* How a static body's lastIteration tag is observed inside one iteration. */
enum {
    PHXO_STAG_SEQUENTIAL = 0, /* reference: a later joint sees the tag an earlier joint just wrote */
    PHXO_STAG_COLOUR_SYNC = 1 /* writes become visible at the next colour boundary (the HIP path)  */
};

/* ---- leaf functions (pinned against oracle/_ref) ---- */
uint32_t phxo_radix_float(float v);                                        /* RadixSort.h:19-26 */
void     phxo_radix_sort3(phxo_sort_entry* e0, phxo_sort_entry* e1, size_t n); /* :28-95, result in e1 */
uint32_t phxo_pair_hash(uint32_t lb, uint32_t rb);                         /* Collider.h:10-18 */
void     phxo_body_init(phxo_body* b, float px, float py, float angle, float sx, float sy, float density); /* RigidBody.h:15-41, Coords2.h:10-17 */
void     phxo_recompute_aabb(phxo_body* b);                                /* Geom.h:79-85 */
void     phxo_rotate_vec(phxo_vec2* v, float angle);                       /* Vector2.h:48-56 */
int      phxo_support_points(const phxo_body* b, float ax, float ay, phxo_vec2 out[2]); /* Geom.h:66-77 */
int      phxo_contact_equals(const phxo_contact_point* a, const phxo_contact_point* o, float tol); /* Manifold.h:28-36 */
void     phxo_contact_point_make(phxo_contact_point* out, float p1x, float p1y, float p2x, float p2y, float nx, float ny,
                                 const phxo_body* b1, const phxo_body* b2);  /* Manifold.h:18-26 */
void     phxo_project_point_to_line(float px, float py, float qx, float qy, float nx, float ny, float dx, float dy, float out[2]); /* Vector2.h ProjectPointToLine */
int      phxo_aabb_intersects(const phxo_body* a, const phxo_body* b);      /* AABB2.h:18-24 */
float    phxo_flipsign(float x, float y, int simd);   /* simd = 0: SIMD_Scalar.h:265-268; 1: SIMD_SSE2.h / SIMD_AVX2.h:272-275 (sign-bit xor) */
/* the sweeps' arithmetic form (phx_oracle.c: mul_add / mul_sub): 0 = PHXO_ARITH_SOURCE (default), 1 = PHXO_ARITH_FUSED.  Process-wide. */
enum { PHXO_ARITH_SOURCE = 0, PHXO_ARITH_FUSED = 1 };
void     phxo_set_arith(int fused);
int      phxo_get_arith(void);
float    phxo_max(float l, float r);                  /* SIMD_Scalar.h:275-278 = the SSE2 / AVX2 max on non-NaN inputs */

/* ---- broadphase stages on raw arrays (Collider.cpp:251-366) ---- */
void   phxo_broadphase_build(const phxo_body* bodies, size_t n, phxo_sort_entry* keys_unsorted /*n, may be NULL*/,
                             phxo_sort_entry* sorted /*n*/, phxo_bp_entry* entries /*n*/);
/* all overlapping (sorted-order) pairs, no pair-set filtering; returns count, writes min(count,cap) */
size_t phxo_sweep_candidates(const phxo_bp_entry* e, size_t n, uint32_t* pairs /*2*cap*/, size_t cap, uint64_t* tests);

/* ---- solver on raw arrays ---- */
typedef struct {
    int32_t island_count, island_max_size;
    int32_t group_offset;          /* PrepareIndices result for the (last) island          */
    int32_t impulse_iterations;    /* sweeps actually executed before the early exit (max over islands) */
    int32_t displacement_iterations;
    int64_t joint_visits;          /* joints swept (skipped ones included), impulse loop     */
    int64_t joints_computed;       /* joints whose skip test passed, impulse loop            */
    int64_t stag_events;           /* times a static-body tag written earlier in the SAME colour/iteration
                                      changed a skip decision (only counted in SEQUENTIAL mode with colours given) */
} phxo_solve_stats;

/* Solver::SolveJoints (Solver.cpp:17-119) with the reference's own ordering:
 * solve_mode picks the pack width N (1/4/8), island_mode the split.  joint_order_out (nj ints,
 * may be NULL) receives joint_index after PrepareIndices (aligned island slots hold -1). */
void phxo_solver_solve(phxo_body* bodies, int nb, const phxo_contact_point* cps,
                       phxo_contact_joint* joints, int nj,
                       int solve_mode, int island_mode, int contact_iters, int penetration_iters,
                       int32_t* joint_order_out, int order_cap, phxo_solve_stats* stats);

/* The same arithmetic with scalar (N=1) semantics swept in a caller-given order: order[k] is the
 * joint solved k-th in every sweep.  colour_offsets (ncolours+1 entries into order[], or NULL)
 * only matters for PHXO_STAG_COLOUR_SYNC.  This is what the HIP path is compared against. */
void phxo_solver_solve_ordered(phxo_body* bodies, int nb, const phxo_contact_point* cps,
                               phxo_contact_joint* joints, int nj,
                               const int32_t* order, const int32_t* colour_offsets, int ncolours,
                               int contact_iters, int penetration_iters, int stag_mode,
                               phxo_solve_stats* stats);

/* Grouped form of the above (what the HIP path does in the island-aware modes): group g = slots
 * [group_offsets[g], group_offsets[g+1]) is solved as an independent island with its own early exit and
 * its own copy of the static bodies' lastIteration tags.  See the comment at the definition. */
void phxo_solver_solve_grouped(phxo_body* bodies, int nb, const phxo_contact_point* cps,
                               phxo_contact_joint* joints, int nj,
                               const int32_t* order, const int32_t* colour_offsets, int ncolours,
                               const int32_t* group_offsets, int ngroups,
                               int contact_iters, int penetration_iters, int stag_mode,
                               phxo_solve_stats* stats);

float phxo_round_f16(float x);   /* float -> IEEE binary16 (nearest even) -> float */

/* Ablation, not reference behaviour: the first `fp16_groups` groups keep body velocities in IEEE binary16 between
 * joint updates (round to nearest even on every store, fp32 arithmetic) — the model of phx_solver_set_body_state_bits(16). */
void phxo_solver_solve_grouped_fp16(phxo_body* bodies, int nb, const phxo_contact_point* cps,
                                    phxo_contact_joint* joints, int nj,
                                    const int32_t* order, const int32_t* colour_offsets, int ncolours,
                                    const int32_t* group_offsets, int ngroups,
                                    int contact_iters, int penetration_iters, int stag_mode, int fp16_groups,
                                    phxo_solve_stats* stats);

/* RefreshJoints only (Solver.cpp:592-695): 29 floats per joint in the field order of
 * ContactJointPacked<1> minus indices: normal limiter 13, normalLimiter_compInvMass(unused, 0),
 * dstVelocity, dstDisplacingVelocity, accumulatedDisplacingImpulse, friction limiter 13. */
void phxo_refresh_joint(const phxo_body* bodies, const phxo_contact_point* cps,
                        const phxo_contact_joint* j, float out[30]);

/* Solver::GatherIslands (Solver.cpp:285-454) */
int phxo_gather_islands(const phxo_body* bodies, int nb, const phxo_contact_joint* joints, int nj, int group_size,
                        int32_t* joint_index /*aligned count*/, int cap,
                        int32_t* island_offset, int32_t* island_size /* nb entries each */,
                        int32_t* island_count, int32_t* island_max);
/* Solver::PrepareIndices (Solver.cpp:217-273) on joint_index[begin,end) */
int phxo_prepare_indices(const phxo_contact_joint* joints, int nb, int32_t* joint_index, int begin, int end, int group_size);

/* ---- whole world (World.cpp:19-37) ---- */
typedef struct phxo_world phxo_world;
phxo_world* phxo_world_create(void);
void        phxo_world_destroy(phxo_world* w);
int         phxo_world_add_body(phxo_world* w, float px, float py, float angle, float sx, float sy); /* World.cpp:11-17 */
void        phxo_world_set_gravity(phxo_world* w, float g);
void        phxo_world_update(phxo_world* w, float dt, int solve_mode, int island_mode, int contact_iters, int penetration_iters);
/* run everything of World::Update that precedes Solver::SolveJoints (World.cpp:25-32) */
void        phxo_world_pre_solve(phxo_world* w, float dt);
/* run Solver::SolveJoints + IntegratePosition (World.cpp:34-36) */
void        phxo_world_solve_and_integrate(phxo_world* w, float dt, int solve_mode, int island_mode, int contact_iters, int penetration_iters);
void        phxo_world_integrate_position(phxo_world* w, float dt);

phxo_body*           phxo_world_bodies(phxo_world* w, int* n);
phxo_manifold*       phxo_world_manifolds(phxo_world* w, int* n);
phxo_contact_point*  phxo_world_contact_points(phxo_world* w, int* n);
phxo_contact_joint*  phxo_world_joints(phxo_world* w, int* n);
const phxo_sort_entry* phxo_world_sorted(phxo_world* w, int* n);     /* broadphaseSort[1] of the last step */
const phxo_bp_entry*   phxo_world_bp_entries(phxo_world* w, int* n);
const uint32_t*        phxo_world_new_pairs(phxo_world* w, int* npairs); /* pairs created in the last step, emission order */
const phxo_solve_stats* phxo_world_stats(phxo_world* w);
uint64_t               phxo_world_sweep_tests(phxo_world* w);         /* candidate y-tests in the last UpdatePairs */
int                    phxo_world_point_overflows(phxo_world* w);     /* times quirk C.4 (>2 merged points) was clamped */

/* multi-threaded timing harness for bench.py's cpu_baseline leg: Single-Sloppy-style 512-joint
 * batches over `threads` pthreads (races on shared bodies exactly like the reference's sloppy
 * modes; results are NOT used for parity). Returns seconds spent in the impulse loop. */
double phxo_time_impulse_loop(phxo_body* bodies, int nb, const phxo_contact_point* cps,
                              phxo_contact_joint* joints, int nj, int iters, int threads, int64_t* joint_visits);

#ifdef __cplusplus
}
#endif
#endif
