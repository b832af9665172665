// world_kernels.h — the stages of World::Update that sit around the two hot halves, as HIP kernels, so that the
// whole step runs on HBM-resident arrays.
//
//   IntegrateVelocity / IntegratePosition        ref: src/World.cpp:39-70
//   manifold creation for new pairs              ref: src/Collider.cpp:313-316
//   UpdateManifolds (SAT + contact generation)   ref: src/Collider.cpp:368-377 -> narrowphase.h
//   PackManifolds                                ref: src/Collider.cpp:379-416
//   RefreshContactJoints                         ref: src/World.cpp:72-149
//
// The last two are written sequentially in the reference (swap-remove while scanning forward), and the ORDER they
// leave behind is semantics: joint order is solve order.  The parallel form reproduces that order exactly.  Scanning
// i upward and replacing every dead a[i] by the current last element (re-examining i) ends with: survivors count
// n' = n - dead; every dead position below n' (a "hole"), taken in ascending order, receives the live elements that
// sat at positions >= n' (the "movers"), taken in DESCENDING order; everything else stays where it was.
#pragma once

#include "common.h"
#include "narrowphase.h"

namespace phx {

// ---- 128-byte records <-> the World's resident arrays (body_view.h): upload after construction, download for the getters -------
static __global__ void __launch_bounds__(256) k_bodies_to_world(const phx_rigid_body* __restrict__ bodies, int n, WorldBodies w)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const phx_rigid_body& b = bodies[i];
        w.s.vel[i] = make_float4(b.velocity.x, b.velocity.y, b.angular_velocity, 0.f);
        w.s.dvel[i] = make_float4(b.displacing_velocity.x, b.displacing_velocity.y, b.displacing_angular_velocity, 0.f);
        w.s.mpos[i] = make_float4(b.inv_mass, b.inv_inertia, b.pos.x, b.pos.y);
        w.frame[i] = make_float4(b.xvector.x, b.xvector.y, b.yvector.x, b.yvector.y);
        w.aabb[i] = make_float4(b.aabb_min.x, b.aabb_min.y, b.aabb_max.x, b.aabb_max.y);
        w.size[i] = make_float2(b.geom_size.x, b.geom_size.y);
    }
}

// everything a step changes goes back into the records (inverse masses, size, index, the vestigial fields and the accelerations —
// zero after every IntegrateVelocity, ref: World.cpp:49-52 — are what the upload left there)
static __global__ void __launch_bounds__(256) k_world_to_bodies(WorldBodies w, int n, phx_rigid_body* __restrict__ bodies)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 v = w.s.vel[i], d = w.s.dvel[i], m = w.s.mpos[i], f = w.frame[i], a = w.aabb[i];
        phx_rigid_body& b = bodies[i];
        b.velocity.x = v.x; b.velocity.y = v.y; b.angular_velocity = v.z;
        b.displacing_velocity.x = d.x; b.displacing_velocity.y = d.y; b.displacing_angular_velocity = d.z;
        b.pos.x = m.z; b.pos.y = m.w;
        b.xvector.x = f.x; b.xvector.y = f.y; b.yvector.x = f.z; b.yvector.y = f.w;
        b.geom_xvector = b.xvector; b.geom_yvector = b.yvector; b.geom_pos = b.pos;      // UpdateGeom (ref: RigidBody.h:38-42)
        b.aabb_min.x = a.x; b.aabb_min.y = a.y; b.aabb_max.x = a.z; b.aabb_max.y = a.w;
    }
}

// IntegrateVelocity (ref: World.cpp:39-55).  The reference zeroes both accelerations at the end of every IntegrateVelocity
// (World.cpp:49-52) and nothing on the path sets them, so the resident world carries no acceleration arrays: they are zero
// whenever this runs — except in the FIRST step after an upload of records that came with accelerations (phx_world_set_state /
// set_bodies of a foreign or handed-over state): `accel` = {acceleration.x, .y, angularAcceleration} per body for that one step,
// null otherwise.  The statements keep the reference's form (x + 0 * dt is not x for x = -0).
// (first kernel of a step: it also clears the step's four counters)
static __global__ void __launch_bounds__(256) k_integrate_velocity(float4* __restrict__ vel, const float4* __restrict__ mpos, int n, float gravity, float dt,
                                                                   unsigned* __restrict__ counters, const float4* __restrict__ accel)
{
    if (blockIdx.x == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 v = vel[i];
        float ax = 0.f, ay = 0.f, aa = 0.f;
        if (accel) { const float4 a = accel[i]; ax = a.x; ay = a.y; aa = a.z; }
        if (mpos[i].x > 0.0f) ay += gravity;
        v.x += ax * dt; v.y += ay * dt;
        v.z += aa * dt;
        vel[i] = v;
    }
}

// the records' accelerations once IntegrateVelocity has consumed them (ref: World.cpp:50, 53)
static __global__ void __launch_bounds__(256) k_clear_accelerations(phx_rigid_body* __restrict__ bodies, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        bodies[i].acceleration.x = 0.f; bodies[i].acceleration.y = 0.f; bodies[i].angular_acceleration = 0.f;
    }
}

// Vector2::Rotate (ref: Vector2.h:48-56); the reference's unqualified cos/sin resolve to the double overloads
__device__ __forceinline__ void rotate_vec(float& vx, float& vy, float c, float s)
{
    const V2 x = v2(vx, vy), y = perp(x);
    const V2 delta = (x * c + y * s) - x;
    vx = vx + delta.x; vy = vy + delta.y;
}

// IntegratePosition (ref: World.cpp:57-70) on the resident arrays: reads 72 bytes per body, writes 64.
// `gate` (may be null): the control word of the solve queued in front; if it differs from `expected` that solve
// committed nothing (stale or spoiled schedule, solver.hip) and neither does this — the host repeats both.
// `ride`: the settle's mailbox post, taken along by the first workgroup (common.h post_mail_block) — before the gate: the settle reads it.
static __global__ void __launch_bounds__(256) k_integrate_position(WorldBodies w, int n, float dt,
                                                                   const unsigned long long* __restrict__ gate, unsigned long long expected, MailRide ride)
{
    if (ride.args.count && blockIdx.x == 0) post_mail_block(ride.args, ride.host_words, ride.host_seq);
    if (gate && *gate != expected) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 v = w.s.vel[i], d = w.s.dvel[i];
        float4 m = w.s.mpos[i], f = w.frame[i];
        const float2 sz = w.size[i];
        m.z += d.x + v.x * dt;
        m.w += d.y + v.y * dt;
        const float ang = -(d.z + v.z * dt);
        const float c = (float)cos((double)ang), s = (float)sin((double)ang);
        rotate_vec(f.x, f.y, c, s);
        rotate_vec(f.z, f.w, c, s);
        float4 box;
        geom_aabb(v2(m.z, m.w), v2(f.x, f.y), v2(f.z, f.w), v2(sz.x, sz.y), box.x, box.y, box.z, box.w);
        w.s.mpos[i] = m;
        w.frame[i] = f;
        w.s.dvel[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        w.aabb[i] = box;
    }
}

// x-extent of the dynamic bodies' AABBs (ownership-sharded worlds, dist.py: a rank's bodies must stay inside its slab): the floats
// are kept as order-preserving unsigned keys so that atomicMin / atomicMax work; out[0] = min key, out[1] = max key
__device__ __forceinline__ unsigned float_key(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
static __global__ void __launch_bounds__(256) k_x_extent(WorldBodies w, int n, unsigned* __restrict__ out)
{
    unsigned lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 m = w.s.mpos[i];
        if (m.x == 0.f && m.y == 0.f) continue;                            // static bodies belong to every slab
        const float4 a = w.aabb[i];
        lo = min(lo, float_key(a.x)); hi = max(hi, float_key(a.z));
    }
    for (int off = 32; off > 0; off >>= 1) { lo = min(lo, (unsigned)__shfl_down(lo, off)); hi = max(hi, (unsigned)__shfl_down(hi, off)); }
    if ((threadIdx.x & 63) == 0) { if (lo != 0xFFFFFFFFu) atomicMin(&out[0], lo); if (hi) atomicMax(&out[1], hi); }
}

__device__ __forceinline__ NpBody np_load(const WorldBodies& w, int i)
{
    const float4 m = w.s.mpos[i], f = w.frame[i];
    const float2 sz = w.size[i];
    NpBody b;
    b.pos = v2(m.z, m.w); b.xv = v2(f.x, f.y); b.yv = v2(f.z, f.w); b.size = v2(sz.x, sz.y);
    return b;
}

// ref: Collider.cpp:368-377; also flags the manifolds PackManifolds will drop (ref: Collider.cpp:387).
// Manifolds [nm_old, nm) are the pairs UpdatePairs has just found (ref: Collider.cpp:313-316 — Manifold(index_i, index_j,
// manifolds.size * kMaxContactPoints), contact slots blank): they are created here, in the lane that updates them, instead of
// by an append kernel of their own in front of this one.
static __global__ void __launch_bounds__(256) k_update_manifolds(phx_manifold* __restrict__ manifolds, int nm, WorldBodies bodies,
                                                                 phx_contact_point* __restrict__ cps, unsigned* __restrict__ dead, int* __restrict__ dropped,
                                                                 int nm_old, const uint2* __restrict__ new_pairs, int first, unsigned* __restrict__ dead_count,
                                                                 MailRide ride)
{
    // (`ride`: the broadphase's new-pair count goes to the host with this launch's first workgroup — common.h post_mail_block)
    if (ride.args.count && blockIdx.x == 0) post_mail_block(ride.args, ride.host_words, ride.host_seq);
    // (`dead_count`: the step's dead-manifold counter, world.hip — dead manifolds are rare, and a count taken here lets PackManifolds'
    //  scan over all manifolds run only in the steps that have one)
    // (`first`: manifolds [first, nm) — the old manifolds are updated while the host waits for the new-pair count, the new ones
    //  in a launch of their own once it is known, world.hip)
    for (int i = first + blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += gridDim.x * blockDim.x) {
        phx_manifold m;
        if (i >= nm_old) {
            const uint2 pr = new_pairs[i - nm_old];
            m.body1 = (int)pr.x; m.body2 = (int)pr.y; m.point_count = 0; m.point_index = 2 * i;
            phx_contact_point blank;
            blank.delta1.x = blank.delta1.y = blank.delta2.x = blank.delta2.y = blank.normal.x = blank.normal.y = 0.f;
            blank.is_merged = 0; blank.is_newly_created = 0; blank.pad_[0] = 0; blank.pad_[1] = 0; blank.solver_index = -1;
            cps[2 * i] = blank; cps[2 * i + 1] = blank;
        } else m = manifolds[i];
        // (two bodies = 2 x 40 bytes of the resident arrays; the 128-byte records cost 2 x 128 for the same fields)
        const NpBody b1 = np_load(bodies, m.body1), b2 = np_load(bodies, m.body2);
        if (update_manifold(m, b1, b2, cps + m.point_index)) atomicAdd(dropped, 1);
        manifolds[i] = m;
        const bool gone = m.point_count == 0 && !aabb_intersects(bodies.aabb[m.body1], bodies.aabb[m.body2]);
        dead[i] = gone ? 1u : 0u;
        if (gone) atomicAdd(dead_count, 1u);
    }
}

// ---- hole-filling compaction ---------------------------------------------------------------------------
// dead_before = exclusive scan of the dead flags; D = total dead; n' = n - D.
// mover_pos[r] = position of the r-th live element counted from the end (only those at positions >= n').
// (`n_flagged`: dead_before[] covers positions [0, n_flagged); positions beyond it were appended after the flags were
//  taken — new joints — and are alive, with every dead element before them)
__device__ __forceinline__ void compact_movers(int block, int blocks, const unsigned* __restrict__ dead_before, const unsigned* __restrict__ dead_total,
                                               int n, int n_flagged, int* __restrict__ mover_pos)
{
    const int D = (int)*dead_total, live = n - D;
    for (int p = live + block * (int)blockDim.x + (int)threadIdx.x; p < n; p += blocks * (int)blockDim.x) {
        const int here = p < n_flagged ? (int)dead_before[p] : D;
        const int next = (p + 1 < n_flagged) ? (int)dead_before[p + 1] : D;
        if (next != here) continue;                                      // p itself is dead
        const int dead_after = D - here;
        mover_pos[(n - 1 - p) - dead_after] = p;
    }
}

static __global__ void __launch_bounds__(256) k_compact_movers(const unsigned* __restrict__ dead_before, const unsigned* __restrict__ dead_total,
                                                               int n, int n_flagged, int* __restrict__ mover_pos)
{
    compact_movers((int)blockIdx.x, (int)gridDim.x, dead_before, dead_total, n, n_flagged, mover_pos);
}

// ref: Collider.cpp:385-410.  Every dead manifold's pair is listed for removal from the pair set; holes take movers.
static __global__ void __launch_bounds__(256) k_pack_manifolds(phx_manifold* __restrict__ manifolds, phx_contact_point* __restrict__ cps, int nm,
                                                               const unsigned* __restrict__ dead_before, const unsigned* __restrict__ dead_total,
                                                               const int* __restrict__ mover_pos, uint2* __restrict__ erased)
{
    const int D = (int)*dead_total, live = nm - D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += gridDim.x * blockDim.x) {
        const int h = (int)dead_before[i];
        const int next = (i + 1 < nm) ? (int)dead_before[i + 1] : D;
        if (next == h) continue;                                          // live: stays
        const phx_manifold gone = manifolds[i];
        erased[h] = make_uint2((unsigned)gone.body1, (unsigned)gone.body2);
        if (i >= live) continue;                                          // dead in the tail: just dropped
        const phx_manifold me = manifolds[mover_pos[h]];
        for (int k = 0; k < me.point_count; ++k) cps[2 * i + k] = cps[me.point_index + k];
        phx_manifold m = me;
        m.point_index = 2 * i;
        manifolds[i] = m;
    }
}

// ---- RefreshContactJoints (ref: World.cpp:72-149) -----------------------------------------------------------
// The reference resets every joint's contact point, lets the live points re-attach theirs and deletes what stayed reset.
// Here a joint is alive iff its `seen` stamp carries this step's epoch (no reset pass), and the dead-joint flags are the LOADER
// of their scan (device_scan.h) instead of a kernel of their own.
// Match, pass 1: matched points re-attach their joint and stamp it; count the points that need a new joint.  (A kernel of its
// own: as the loader of its scan it was slower — 25 us against 9 + 5 at 2e5 manifolds, 112 us at 1e6: a scan workgroup is 1024
// lanes of four items each, too few lanes in flight for this chain of dependent gathers.)
static __global__ void __launch_bounds__(256) k_joints_match(const phx_manifold* __restrict__ manifolds, int nm, const phx_contact_point* __restrict__ cps,
                                                             phx_contact_joint* __restrict__ joints, unsigned* __restrict__ seen, unsigned epoch,
                                                             unsigned* __restrict__ new_count)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += gridDim.x * blockDim.x) {
        const phx_manifold m = manifolds[i];
        unsigned fresh = 0;
        for (int k = 0; k < m.point_count; ++k) {
            const int si = cps[m.point_index + k].solver_index;
            if (si < 0) ++fresh;
            else { joints[si].contact_point_index = m.point_index + k; seen[si] = epoch; }
        }
        new_count[i] = fresh;
    }
}

// loader of the 'dead joints before joint i' scan (queued behind the match)
struct JointDeadLoad {
    static constexpr bool in_place = false;
    const unsigned* seen; unsigned epoch;
    __device__ unsigned operator()(int i) const { return seen[i] != epoch ? 1u : 0u; }
    __device__ bool load4(int base, uint4& out) const      // (base is a multiple of four: device_scan.h; a dead joint is the rare case)
    {
        if (reinterpret_cast<uintptr_t>(seen) & 15u) return false;
        const uint4 s = *reinterpret_cast<const uint4*>(seen + base);
        if (s.x == epoch && s.y == epoch && s.z == epoch && s.w == epoch) { out = make_uint4(0u, 0u, 0u, 0u); return true; }
        out = make_uint4((*this)(base), (*this)(base + 1), (*this)(base + 2), (*this)(base + 3));
        return true;
    }
};

// Match, pass 2: new joints appended in manifold order, then point order (ref: World.cpp:108-114)
// (the movers of the clean-up behind it do not depend on the new joints: when there are dead joints too, the last `mover_blocks`
//  workgroups of the same launch compute them — one dispatch instead of two)
static __global__ void __launch_bounds__(256) k_joints_create(const phx_manifold* __restrict__ manifolds, int nm, phx_contact_point* __restrict__ cps,
                                                              phx_contact_joint* __restrict__ joints, int nj_old, const unsigned* __restrict__ new_before,
                                                              int mover_blocks, const unsigned* __restrict__ dead_before, const unsigned* __restrict__ dead_total,
                                                              int total, int* __restrict__ mover_pos)
{
    const int create_blocks = (int)gridDim.x - mover_blocks;
    if ((int)blockIdx.x >= create_blocks) {
        compact_movers((int)blockIdx.x - create_blocks, mover_blocks, dead_before, dead_total, total, nj_old, mover_pos);
        return;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += create_blocks * blockDim.x) {
        const phx_manifold m = manifolds[i];
        int at = nj_old + (int)new_before[i];
        for (int k = 0; k < m.point_count; ++k) {
            phx_contact_point& cp = cps[m.point_index + k];
            if (cp.solver_index >= 0) continue;
            cp.solver_index = at;
            phx_contact_joint j;
            j.contact_point_index = m.point_index + k; j.body1 = m.body1; j.body2 = m.body2;
            j.normal_accumulated_impulse = 0.f; j.friction_accumulated_impulse = 0.f;
            joints[at++] = j;
        }
    }
}

// Cleanup (ref: World.cpp:125-143): holes take movers
// (a joint that moves takes its contact point's back-pointer along, ref: World.cpp:139 — every other live joint's contact point
//  points at it already: the match re-attached it, or k_joints_create has just made it.  Rounds 1-3 rewrote all nj back-pointers
//  in a kernel of their own behind this one: 8 us at cfg 2, 24 us at cfg 4, for a few hundred moved joints.)
static __global__ void __launch_bounds__(256) k_joints_fill(phx_contact_joint* __restrict__ joints, int nj, int n_flagged, const unsigned* __restrict__ dead_before,
                                                            const unsigned* __restrict__ dead_total, const int* __restrict__ mover_pos,
                                                            phx_contact_point* __restrict__ cps)
{
    const int D = (int)*dead_total, live = nj - D;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < live; i += gridDim.x * blockDim.x) {
        if (i >= n_flagged) continue;                                     // appended after the flags were taken: alive
        const int h = (int)dead_before[i];
        const int next = (i + 1 < n_flagged) ? (int)dead_before[i + 1] : D;
        if (next == h) continue;
        const phx_contact_joint moved = joints[mover_pos[h]];
        joints[i] = moved;
        cps[moved.contact_point_index].solver_index = i;
    }
}

} // namespace phx
