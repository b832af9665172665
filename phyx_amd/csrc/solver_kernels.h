// solver_kernels.h — device data layout + HIP kernels of the sequential-impulse solver (gfx950).
//
// What it replaces (ref = /root/reference/src): Solver::PrepareBodies/FinishBodies (Solver.cpp:456-494),
// PrepareJoints/FinishJoints (:496-547), RefreshJoints (:592-695), PreStepJoints (:697-758),
// SolveJointsImpulses (:760-914), SolveJointsDisplacement (:916-1018).
//
// Layout in HBM:
//   bodies   resident structure of arrays (body_view.h): vel, dvel, mpos = {invMass, invInertia, pos.x, pos.y}
//            (ref SolveBodyParams minus the unused frame), read in place (sb_par aliases mpos);
//            sb_imp[b]  = {vx, vy, w, lastIteration}   float4   (ref SolveBody, Solver.h:95-101), the HBM path's working copy
//            sb_disp[b] = same for the displacing velocities
//            one body = one 16-B gather/scatter granule per array.
//   joints   stored in SCHEDULE order (slot s), colour c owns the contiguous slot range crange[c]:
//            q0[s] = {n.x, n.y, angN1, angN2}          normal projector + its angular projectors
//            q1[s] = {angF1, angF2, invMassF, dstVelocity}
//            q2[s] = {invMassN, invMass1, invInertia1, invMass2}
//            q3[s] = {invInertia2 (bits), body1, body2, static slot or -1}
//            acc[s] = {accumulated normal, accumulated friction}            (read+write each sweep)
//            dd[s]  = {dstDisplacingVelocity, accumulatedDisplacingImpulse}
//            A lane owns one slot, so every array is read as one coalesced 16-B (or 8-B) load per lane.
//   The reference keeps 31 words per joint (ContactLimiterPacked, Solver.h:7-45); projector2 = -projector1,
//   the friction projector is the rotated normal and compMass* = projector * invMass are single IEEE
//   multiplies, so they are recomputed in registers from 16 stored words — bit-identical, 44 % fewer bytes.
//
// Arithmetic contract: every expression below is written in the reference's operation order and the
// file is compiled with -ffp-contract=off, so results equal the strict-IEEE oracle bit for bit.  The sweeps
// (PreStep, impulses, displacement) come in two stated forms, chosen when the library is built (mul_add below).
#pragma once

#include "solver.h"
#include "island_view.h"

#include <hip/hip_fp16.h>

namespace phx {

// ---- static-body tags ----------------------------------------------------------------------------
// A static body (invMass == invInertia == 0, ref: Solver.cpp:304) never changes velocity, so it does
// not serialise the joints that touch it; only its lastIteration tag is shared (ref: Solver.cpp:900-910
// writes it, :793-797 reads it).  The tag lives in two words per static body, selected by sweep parity:
//   word = (iter + 1) << 16 | (0xFFFF - colour)   written with atomicMax by productive joints.
// A joint of (iter, colour) sees the body as productive iff some joint on it was productive in sweep
// iter-1, or in sweep iter in an EARLIER colour — deterministic regardless of scheduling.
__device__ __forceinline__ unsigned static_word(int iter, int colour) { return ((unsigned)(iter + 1) << 16) | (0xFFFFu - (unsigned)colour); }

__device__ __forceinline__ bool static_productive(const unsigned* sw, int nstatic, int slot, int iter, int colour)
{
    if (iter == 0) return true;                                    // tags start at -1 > 0 - 2 (ref: Solver.cpp:474)
    unsigned prev = __hip_atomic_load(&sw[((iter - 1) & 1) * nstatic + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((prev >> 16) == (unsigned)iter) return true;               // productive in sweep iter-1
    unsigned cur = __hip_atomic_load(&sw[(iter & 1) * nstatic + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (cur >> 16) == (unsigned)(iter + 1) && (0xFFFFu - (cur & 0xFFFFu)) < (unsigned)colour;
}

__device__ __forceinline__ float max_ref(float l, float r) { return l > r ? l : r; }      // ref: base/SIMD_Scalar.h:275-278

// ---- the sweeps' arithmetic contract ---------------------------------------------------------------------------------
// The reference writes `dV -= projector * velocity` and `velocity += compMass * dImpulse` (ref: Solver.cpp:833-858 and its
// siblings) and ships -ffast-math -mfma (ref: Makefile:11, 17-24): whether such a pair is one fused multiply-add or a rounded
// product and a rounded sum is the reference compiler's choice, not the source's.  This backend states its choice:
//   PHX_ARITH_FMA = 1 (default)  every such pair is ONE fmaf, in the reference's source order — half the instructions of the class
//                                step, which is what bounds the island kernel;
//   PHX_ARITH_FMA = 0            a rounded product and a rounded sum (rounds 1-4; `PHX_ARITH=source python -m phyx_amd.build`).
// compMass = projector * invMass stays one rounded product of its own in both (the reference stores it, ref: Solver.cpp:573-580).
// The oracle has both forms and the tests ask the library which one it was built with (phx_arith_mode): every
// parity test is bit-exact against the matching form, and the two forms are held within SURVEY.md §8(c)'s tolerances of each other.
#ifndef PHX_ARITH_FMA
#define PHX_ARITH_FMA 1
#endif
__device__ __forceinline__ float mul_add(float a, float b, float acc) { return PHX_ARITH_FMA ? __builtin_fmaf(a, b, acc) : acc + a * b; }      // acc + a * b
__device__ __forceinline__ float mul_sub(float a, float b, float acc) { return PHX_ARITH_FMA ? __builtin_fmaf(-a, b, acc) : acc - a * b; }     // acc - a * b

__device__ __forceinline__ int clamp_index(int i, int n) { return i < 0 ? 0 : (i >= n ? (n > 0 ? n - 1 : 0) : i); }

// Device time of a solve (phx_solve_stats.device_ms) without HIP events — an event record is a barrier packet of its own that
// idles the queue for ~5 us: the first kernel of the solve leaves the constant 100 MHz clock in stamps[0] (min), every workgroup
// of its last kernel in stamps[1] (max).
__device__ __forceinline__ void solve_stamp_begin(unsigned long long* stamps) { if (blockIdx.x == 0 && threadIdx.x == 0) atomicMin(&stamps[0], (unsigned long long)wall_clock64()); }
__device__ __forceinline__ void solve_stamp_end(unsigned long long* stamps) { if (threadIdx.x == 0) atomicMax(&stamps[1], (unsigned long long)wall_clock64()); }
// the same from a wide streaming kernel whose workgroups all end at once: every 32nd workgroup and the last one stamp (all of them firing at
// the one word were ~25 of k_finish_bodies' 30 us in a 200k-body world: same-address atomics take their turns)
__device__ __forceinline__ void solve_stamp_end_sparse(unsigned long long* stamps)
{
    if (threadIdx.x == 0 && ((blockIdx.x & 31u) == 31u || blockIdx.x == gridDim.x - 1)) atomicMax(&stamps[1], (unsigned long long)wall_clock64());
}

// ---- PrepareBodies (ref: Solver.cpp:456-480) -----------------------------------------------------
// `list` = the bodies the HBM group touches (islands solved in LDS read the records directly)
// (the solver works on private copies of the velocities — sb_imp / sb_disp, with the lastIteration tag in the fourth lane — because
//  its results are committed only behind the topology gate; {invMass, invInertia, pos} is read from the resident array in place)
static __global__ void __launch_bounds__(256) k_unpack_bodies(BodyView bodies, const int* __restrict__ list, int count,
                                                       float4* __restrict__ sb_imp, float4* __restrict__ sb_disp,
                                                       unsigned long long* __restrict__ stamps)
{
    solve_stamp_begin(stamps);
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const int i = list[k];
        float4 a = bodies.vel[i], d = bodies.dvel[i];
        a.w = __int_as_float(-1); d.w = __int_as_float(-1);
        sb_imp[i] = a;
        sb_disp[i] = d;
    }
}

// ---- topology fingerprint --------------------------------------------------------------------------
// Order-sensitive 64-bit fingerprint of the joint list's body pairs and of which bodies are static: the
// colouring is a pure function of exactly that, so an unchanged fingerprint lets the schedule be reused.
__device__ __forceinline__ unsigned long long mix64(unsigned long long z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// the island kernel's per-group statistics go to ISL_STAT_SLOTS slots: thousands of same-address atomics would serialise at the L2
constexpr int ISL_STAT_SLOTS = 64;

// the per-solve control words: the HBM path's per-sweep 'productive' flags and static-tag words (cleared for THIS solve), and the
// control set of the NEXT solve — its control word, the island kernel's counters, the solve's time stamps; two sets alternate and the
// first kernel of every solve (this one, or the island kernel when no hash pass runs: island_view.h) clears the other set
struct ControlWords {
    int* flags; int nflags;
    unsigned* sw; int nsw;
    unsigned long long* next_ctl;
    int* next_executed;
    unsigned long long* next_visits;
    unsigned long long* next_shards;      // the island kernel's arrival counters (island_view.h)
};

constexpr int HASH_T = 1024;          // few, fat workgroups: the final same-address atomics serialise (~10 ns each)
constexpr int HASH_BLOCKS = 128;

static __global__ void __launch_bounds__(HASH_T) k_topology_hash(const phx_contact_joint* __restrict__ joints, int nj,
                                                       const float4* __restrict__ mpos, int nb, int ncp, unsigned long long* out,
                                                       ControlWords cw)
{
    // First kernel of the solves that run it (schedules with an HBM group, sharded solves, rebuilds the long way): it also clears
    // the solve's HBM-path words and the NEXT solve's control set (two sets alternate): one dispatch where there used to be five memsets.
    {
        const int i = blockIdx.x * blockDim.x + threadIdx.x, n = gridDim.x * blockDim.x;
        if (i == 0) *cw.next_ctl = 0ull;
        if (i < ISL_STAT_SLOTS) { cw.next_visits[i] = 0ull; cw.next_executed[2 * i] = 0; cw.next_executed[2 * i + 1] = 0; }
        if (i == 0) { cw.next_visits[ISL_STAT_SLOTS] = ~0ull; cw.next_visits[ISL_STAT_SLOTS + 1] = 0ull; }      // the solve's time stamps (solve_stamp)
        if (i < ISL_SHARDS) cw.next_shards[i * ISL_SHARD_STRIDE] = 0ull;
        for (int k = i; k < cw.nflags; k += n) cw.flags[k] = 0;
        for (int k = i; k < cw.nsw; k += n) cw.sw[k] = 0u;
    }
    __shared__ unsigned long long part[HASH_T / 64];
    unsigned long long h = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nj; i += gridDim.x * blockDim.x) {
        const phx_contact_joint j = joints[i];
        unsigned long long k = ((unsigned long long)(unsigned)j.body1 << 32) | (unsigned)j.body2;
        h += mix64(k + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1));
        h += mix64(0xA24BAED4963EE407ull * (unsigned long long)(unsigned)j.contact_point_index + (unsigned long long)i);   // the colouring priority id
        // an out-of-range index can never belong to the topology a schedule was built (and validated) for
        if ((unsigned)j.body1 >= (unsigned)nb || (unsigned)j.body2 >= (unsigned)nb || (unsigned)j.contact_point_index >= (unsigned)ncp)
            h += 0xBADBADBADBADBAD1ull + (unsigned long long)i;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const float4 p = mpos[i];
        if (p.x == 0.f && p.y == 0.f) h += mix64(0xD1B54A32D192ED03ull * (unsigned long long)(i + 1));
    }
    for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) {                                        // one device-scope atomic per workgroup
        unsigned long long t = 0;
        for (int w = 0; w < HASH_T / 64; ++w) t += part[w];
        if (t) atomicAdd(out, t);
    }
}

// ---- PrepareJoints + RefreshJoints (ref: Solver.cpp:496-521, 549-695) ----------------------------
struct Limiter { float a1, a2, cim; };

// RefreshLimiter (ref: Solver.cpp:549-590) for projector (n1x,n1y) on body 1 and its negation on body 2
__device__ __forceinline__ Limiter refresh_limiter(float n1x, float n1y, float w1x, float w1y, float w2x, float w2y,
                                                   float im1, float ii1, float im2, float ii2)
{
    const float n2x = -n1x, n2y = -n1y;
    Limiter L;
    L.a1 = n1x * w1y - n1y * w1x;
    L.a2 = n2x * w2y - n2y * w2x;
    const float c1x = n1x * im1, c1y = n1y * im1, c1a = L.a1 * ii1;
    const float c2x = n2x * im2, c2y = n2y * im2, c2a = L.a2 * ii2;
    const float m1 = n1x * c1x + n1y * c1y + L.a1 * c1a;
    const float m2 = n2x * c2x + n2y * c2y + L.a2 * c2a;
    const float m = m1 + m2;
    L.cim = fabsf(m) > 0.f ? 1.0f / m : 0.f;
    return L;
}

static __global__ void __launch_bounds__(256) k_pack_refresh(SolverView v, int begin, int end, const phx_contact_joint* __restrict__ joints,
                                                      const phx_contact_point* __restrict__ cps, const int* __restrict__ static_slot)
{
    for (int s = begin + blockIdx.x * blockDim.x + threadIdx.x; s < end; s += gridDim.x * blockDim.x) {
        phx_contact_joint j = joints[v.order[s]];
        // indices come from the caller's arrays; clamp so that a bad or stale record cannot fault (the host
        // validates them whenever it rebuilds the schedule and the fingerprint kernel poisons itself on a bad one)
        j.body1 = clamp_index(j.body1, v.nb); j.body2 = clamp_index(j.body2, v.nb);
        const phx_contact_point& cp = cps[clamp_index(j.contact_point_index, v.ncp)];
        const float d1x = cp.delta1.x, d1y = cp.delta1.y, d2x = cp.delta2.x, d2y = cp.delta2.y;
        const float nx = cp.normal.x, ny = cp.normal.y;
        const float4 p1 = v.sb_par[j.body1], p2 = v.sb_par[j.body2];       // {im, ii, pos.x, pos.y}

        const float pt1x = d1x + p1.z, pt1y = d1y + p1.w;
        const float pt2x = d2x + p2.z, pt2y = d2y + p2.w;
        const float w1x = d1x, w1y = d1y;
        const float w2x = pt1x - p2.z, w2y = pt1y - p2.w;                  // ref: Solver.cpp:649-650 (body-1's point, sic)

        const Limiter N = refresh_limiter(nx, ny, w1x, w1y, w2x, w2y, p1.x, p1.y, p2.x, p2.y);
        const Limiter F = refresh_limiter(-ny, nx, w1x, w1y, w2x, w2y, p1.x, p1.y, p2.x, p2.y);

        // ref: Solver.cpp:658-680.  bounce == 0 makes dv = -0 * (relV . n); max(dv - 1, 0) is then +0 for
        // every finite or non-finite relV, so the velocity gathers of :619-625 are dead and dropped.
        const float depth = (pt2x - pt1x) * nx + (pt2y - pt1y) * ny;
        const float dst = 0.f;
        const float n_dst = depth < 1.f ? dst - 0.1f : dst;
        const float n_dst_disp = 0.1f * max_ref(0.f, depth - 2.0f * 1.f);

        const int s1 = static_slot[j.body1], s2 = static_slot[j.body2];
        v.q0[s] = make_float4(nx, ny, N.a1, N.a2);
        v.q1[s] = make_float4(F.a1, F.a2, F.cim, n_dst);
        v.q2[s] = make_float4(N.cim, p1.x, p1.y, p2.x);
        v.qn[s] = N.cim;                        // (again, 4 bytes apart: all a unit's FOLLOWER needs of q2 / q3 — the rest is its leader's)
        v.q3[s] = make_int4(__float_as_int(p2.y), j.body1, j.body2, s1 >= 0 ? s1 : s2);
        v.acc[s] = make_float2(j.normal_accumulated_impulse, j.friction_accumulated_impulse);
        v.dd[s] = make_float2(n_dst_disp, 0.f);
    }
}

// ---- PreStepJoints (ref: Solver.cpp:697-758), one class: a lane applies its unit's leader, then its follower ----------
// (a class = `leaders` leader slots followed by `followers` follower slots; follower i belongs to leader i, schedule.h)
__device__ __forceinline__ void prestep_one(const SolverView& v, int s, float4& B1, float4& B2, float im1, float ii1, float im2, float ii2, bool st1, bool st2)
{
    const float4 a = v.q0[s], b = v.q1[s];
    const float2 acc = v.acc[s];
    const float nx = a.x, ny = a.y, tx = -ny, ty = nx;
    if (!st1) {
        B1.x = mul_add(nx * im1, acc.x, B1.x); B1.y = mul_add(ny * im1, acc.x, B1.y); B1.z = mul_add(a.z * ii1, acc.x, B1.z);
        B1.x = mul_add(tx * im1, acc.y, B1.x); B1.y = mul_add(ty * im1, acc.y, B1.y); B1.z = mul_add(b.x * ii1, acc.y, B1.z);
    }
    if (!st2) {
        B2.x = mul_add((-nx) * im2, acc.x, B2.x); B2.y = mul_add((-ny) * im2, acc.x, B2.y); B2.z = mul_add(a.w * ii2, acc.x, B2.z);
        B2.x = mul_add((-tx) * im2, acc.y, B2.x); B2.y = mul_add((-ty) * im2, acc.y, B2.y); B2.z = mul_add(b.y * ii2, acc.y, B2.z);
    }
}

static __global__ void __launch_bounds__(256) k_prestep(SolverView v, int begin, int leaders, int followers)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < leaders; i += gridDim.x * blockDim.x) {
        const float4 c = v.q2[begin + i];                  // the unit's bodies and their masses: the leader's record serves both joints
        const int4 k = v.q3[begin + i];
        const float im1 = c.y, ii1 = c.z, im2 = c.w, ii2 = __int_as_float(k.x);
        const bool st1 = (im1 == 0.f && ii1 == 0.f), st2 = (im2 == 0.f && ii2 == 0.f);
        float4 B1 = make_float4(0.f, 0.f, 0.f, 0.f), B2 = B1;
        if (!st1) B1 = v.sb_imp[k.y];
        if (!st2) B2 = v.sb_imp[k.z];
        prestep_one(v, begin + i, B1, B2, im1, ii1, im2, ii2, st1, st2);
        if (i < followers) prestep_one(v, begin + leaders + i, B1, B2, im1, ii1, im2, ii2, st1, st2);
        if (!st1) v.sb_imp[k.y] = B1;
        if (!st2) v.sb_imp[k.z] = B2;
    }
}

// ---- SolveJointsImpulses + SolveJointsDisplacement (ref: Solver.cpp:760-1018), one colour, one sweep
// Launch shape: one wavefront per workgroup (64 lanes, one unit per lane).  A class of the 200k-box scene has
// only 2e4-1e5 joints, i.e. a few waves per CU: the kernel is bound by the dependent-load chain
// (slot -> body index -> body state), not by bandwidth, so (1) single-wave workgroups spread the joints over all
// 256 CUs and (2) every load that does not depend on the body index is issued up front, before the skip test —
// the reference loads the joint constants only after the test (ref: Solver.cpp:798-831), which on this machine
// would add a third dependent round trip to save bytes that are not the bottleneck.
constexpr int SOLVE_BLOCK = 64;

// atomicMax(&words[slot], word) for every lane with `want`, issued once per distinct slot in the wave
__device__ __forceinline__ void wave_tag_update(unsigned* words, bool want, int slot, unsigned word)
{
    unsigned long long todo = __ballot(want);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int key = __shfl(slot, leader);
        const unsigned long long same = __ballot(want && slot == key);
        if (lane == leader) atomicMax(&words[key], word);
        todo &= ~same;
    }
}

// the constants and accumulators of one slot, loaded up front (nothing here depends on the body gathers)
struct HbmJoint { float4 a, f, c; int4 k; float2 acc, d; };

__device__ __forceinline__ HbmJoint hbm_load(const SolverView& v, int s, bool imp_on, bool disp_on, bool follower)
{
    HbmJoint q;
    if (follower) {                                        // of q2 / q3 a follower needs its own 1 / (normal mass) only: 4 bytes instead of 32
        q.k = make_int4(0, 0, 0, 0);
        q.c = make_float4(v.qn[s], 0.f, 0.f, 0.f);
    } else { q.k = v.q3[s]; q.c = v.q2[s]; }
    q.a = v.q0[s];
    q.f = make_float4(0.f, 0.f, 0.f, 0.f); q.acc = make_float2(0.f, 0.f); q.d = make_float2(0.f, 0.f);
    if (imp_on) { q.f = v.q1[s]; q.acc = v.acc[s]; }
    if (disp_on) q.d = v.dd[s];
    return q;
}

// one joint of a unit on the body state the lane holds in registers (ref: Solver.cpp:790-896 impulses, :960-1005 displacement)
__device__ __forceinline__ void solve_one(const SolverView& v, int s, HbmJoint& q, int colour, int iter, bool imp_on, bool disp_on,
                                          float4& B1, float4& B2, float4& D1, float4& D2, float im1, float ii1, float im2, float ii2, bool st1, bool st2, int ss,
                                          bool sp_imp, bool sp_disp,
                                          bool& any_imp, bool& any_disp, bool& tag_imp, bool& tag_disp, bool& dirty_imp, bool& dirty_disp)
{
    // sp_imp / sp_disp: 'the unit's static body was productive' (static_productive) — read once per unit: a class cannot
    // change what the test returns for that class (tags raised in it carry the class itself, which is not 'earlier')
    const float4 a = q.a, f = q.f, c = q.c;
    const float nx = a.x, ny = a.y, tx = -ny, ty = nx;
    if (imp_on) {
        // ref: Solver.cpp:790-798
        const bool p1 = st1 ? sp_imp : (__float_as_int(B1.w) > iter - 2);
        const bool p2 = st2 ? sp_imp : (__float_as_int(B2.w) > iter - 2);
        if (p1 || p2) {
            float2 acc = q.acc;
            // normal limiter (ref: :833-858)
            float dv = f.w;
            dv = mul_sub(nx, B1.x, dv); dv = mul_sub(ny, B1.y, dv); dv = mul_sub(a.z, B1.z, dv);
            dv = mul_sub(-nx, B2.x, dv); dv = mul_sub(-ny, B2.y, dv); dv = mul_sub(a.w, B2.z, dv);
            float dn = dv * c.x;
            dn = max_ref(dn, -acc.x);
            B1.x = mul_add(nx * im1, dn, B1.x); B1.y = mul_add(ny * im1, dn, B1.y); B1.z = mul_add(a.z * ii1, dn, B1.z);
            B2.x = mul_add((-nx) * im2, dn, B2.x); B2.y = mul_add((-ny) * im2, dn, B2.y); B2.z = mul_add(a.w * ii2, dn, B2.z);
            acc.x += dn;
            // friction limiter (ref: :860-889)
            float fv = 0.f;
            fv = mul_sub(tx, B1.x, fv); fv = mul_sub(ty, B1.y, fv); fv = mul_sub(f.x, B1.z, fv);
            fv = mul_sub(-tx, B2.x, fv); fv = mul_sub(-ty, B2.y, fv); fv = mul_sub(f.y, B2.z, fv);
            float df = fv * f.z;
            const float force = acc.y + df;
            const float limit = acc.x * 0.3f;
            const float signed_limit = force < 0.f ? -limit : limit;          // scalar flipsign, ref: SIMD_Scalar.h:265-268
            const float adjusted = signed_limit - acc.y;
            if (fabsf(force) > limit) df = adjusted;
            acc.y += df;
            B1.x = mul_add(tx * im1, df, B1.x); B1.y = mul_add(ty * im1, df, B1.y); B1.z = mul_add(f.x * ii1, df, B1.z);
            B2.x = mul_add((-tx) * im2, df, B2.x); B2.y = mul_add((-ty) * im2, df, B2.y); B2.z = mul_add(f.y * ii2, df, B2.z);
            v.acc[s] = acc;
            const bool productive = max_ref(fabsf(dn), fabsf(df)) > 1e-4f;      // ref: :894-896
            if (productive) {
                B1.w = __int_as_float(iter); B2.w = __int_as_float(iter);
                any_imp = true;
                if ((st1 || st2) && ss >= 0) tag_imp = true;
            }
            dirty_imp = true;
        }
    }
    if (disp_on) {
        const bool p1 = st1 ? sp_disp : (__float_as_int(D1.w) > iter - 2);
        const bool p2 = st2 ? sp_disp : (__float_as_int(D2.w) > iter - 2);
        if (p1 || p2) {
            float2 d = q.d;
            float dv = d.x;                                                      // ref: :973-981
            dv = mul_sub(nx, D1.x, dv); dv = mul_sub(ny, D1.y, dv); dv = mul_sub(a.z, D1.z, dv);
            dv = mul_sub(-nx, D2.x, dv); dv = mul_sub(-ny, D2.y, dv); dv = mul_sub(a.w, D2.z, dv);
            float di = dv * c.x;
            di = max_ref(di, -d.y);
            D1.x = mul_add(nx * im1, di, D1.x); D1.y = mul_add(ny * im1, di, D1.y); D1.z = mul_add(a.z * ii1, di, D1.z);
            D2.x = mul_add((-nx) * im2, di, D2.x); D2.y = mul_add((-ny) * im2, di, D2.y); D2.z = mul_add(a.w * ii2, di, D2.z);
            d.y += di;
            v.dd[s] = d;
            const bool productive = fabsf(di) > 1e-4f;                           // ref: :999
            if (productive) {
                D1.w = __int_as_float(iter); D2.w = __int_as_float(iter);
                any_disp = true;
                if ((st1 || st2) && ss >= 0) tag_disp = true;
            }
            dirty_disp = true;
        }
    }
}

// one class: `leaders` leader slots from `begin`, then `followers` follower slots; lane i sweeps leader i, then follower i
// on the same two bodies (schedule.h) — one gather and one scatter of the body state per unit
template <bool DO_IMP, bool DO_DISP>
static __global__ void __launch_bounds__(SOLVE_BLOCK) k_solve_colour(SolverView v, int begin, int leaders, int followers, int colour, int iter)
{
    // A sweep after an unproductive sweep skips every joint (all tags <= iter-2), which is why the reference may
    // stop there (ref: Solver.cpp:189, 210).  The impulse half needs no flag for that — each joint's own skip
    // test does it — so its loads are issued unconditionally; the displacement half (dead after the first sweep
    // on a resting scene) is gated by its flag, whose scalar load overlaps the vector loads below.
    const bool imp_on = DO_IMP;
    const bool disp_on = DO_DISP && (iter == 0 || v.disp_active[iter - 1] != 0);

    bool any_imp = false, any_disp = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < leaders; i += gridDim.x * blockDim.x) {
        const bool has2 = i < followers;
        const int s0 = begin + i, s1 = begin + leaders + i;
        HbmJoint q0 = hbm_load(v, s0, imp_on, disp_on, false), q1{};
        if (has2) q1 = hbm_load(v, s1, imp_on, disp_on, true);
        const int b1 = q0.k.y, b2 = q0.k.z, ss = q0.k.w;
        float4 B1 = make_float4(0.f, 0.f, 0.f, 0.f), B2 = B1, D1 = B1, D2 = B1;
        if (imp_on) { B1 = v.sb_imp[b1]; B2 = v.sb_imp[b2]; }
        if (disp_on) { D1 = v.sb_disp[b1]; D2 = v.sb_disp[b2]; }
        const float im1 = q0.c.y, ii1 = q0.c.z, im2 = q0.c.w, ii2 = __int_as_float(q0.k.x);
        const bool st1 = (im1 == 0.f && ii1 == 0.f), st2 = (im2 == 0.f && ii2 == 0.f);
        const float4 S1 = B1, S2 = B2, T1 = D1, T2 = D2;
        bool tag_imp = false, tag_disp = false, dirty_imp = false, dirty_disp = false;
        const bool sp_imp = imp_on && (st1 || st2) && static_productive(v.sw_imp, v.nstatic, ss, iter, colour);
        const bool sp_disp = disp_on && (st1 || st2) && static_productive(v.sw_disp, v.nstatic, ss, iter, colour);
        solve_one(v, s0, q0, colour, iter, imp_on, disp_on, B1, B2, D1, D2, im1, ii1, im2, ii2, st1, st2, ss, sp_imp, sp_disp, any_imp, any_disp, tag_imp, tag_disp, dirty_imp, dirty_disp);
        if (has2) {                                    // a static body's record is never stored: the follower must see it untouched
            if (st1) { B1 = S1; D1 = T1; }
            if (st2) { B2 = S2; D2 = T2; }
            solve_one(v, s1, q1, colour, iter, imp_on, disp_on, B1, B2, D1, D2, im1, ii1, im2, ii2, st1, st2, ss, sp_imp, sp_disp, any_imp, any_disp, tag_imp, tag_disp, dirty_imp, dirty_disp);
        }
        if (dirty_imp) { if (!st1) v.sb_imp[b1] = B1; if (!st2) v.sb_imp[b2] = B2; }
        if (dirty_disp) { if (!st1) v.sb_disp[b1] = D1; if (!st2) v.sb_disp[b2] = D2; }
        // static-tag updates of this wave, one atomic per distinct static body: every joint on the ground raises the
        // same word to the same value, and same-address atomics serialise at the L2 (thousands per colour otherwise)
        if (DO_IMP) wave_tag_update(v.sw_imp + (iter & 1) * v.nstatic, tag_imp, ss, static_word(iter, colour));
        if (DO_DISP) wave_tag_update(v.sw_disp + (iter & 1) * v.nstatic, tag_disp, ss, static_word(iter, colour));
    }
    // any(productive) of the sweep (ref: Solver.cpp:913, 1017): one store per wave that saw one
    if (DO_IMP && __any(any_imp) && (threadIdx.x & 63) == 0) v.imp_active[iter] = 1;
    if (DO_DISP && __any(any_disp) && (threadIdx.x & 63) == 0) v.disp_active[iter] = 1;
}

// ---- the TAIL of the HBM group's classes in one launch, one workgroup (round 6) ---------------------------------------------------------
// The last classes of an island too big for a workgroup are tiny — the units of a pile's loose ends and of bodies whose neighbours lie far
// away in the body order: ten classes of 3 .. 1600 joints each once the settled 200k-box pile loosens again (steps 80+), one launch of
// k_solve_colour per class and sweep, 4 us + a kernel boundary for a microsecond of work: 60 % of that world's solve.  Here ONE workgroup
// sweeps the trailing classes of at most TAIL_T units one after the other with a workgroup barrier where the launches had a kernel
// boundary (one workgroup = one CU = one L1: its own stores are visible to its later loads behind __syncthreads(); the static bodies' tags
// are atomics at the coherent level and read past the L1 by static_productive) — and, which is what round 3's form of this lacked (it was no
// faster than the launches: a class cost its two dependent memory round trips wherever it ran), lane t requests the constants of its
// unit of class c + 1 BEFORE it sweeps its unit of class c, under no branch (k_solve_parts_ahead says why), so a class step is the body
// gather, the arithmetic and the stores' acknowledgement, not an HBM round trip.  Same arithmetic (solve_one), same slot order: bit-identical
// to the launches.
constexpr int TAIL_T = 1024, TAIL_CLASSES_MAX = 64;

struct TailUnit { HbmJoint q0, q1; int s0, s1; bool have; float4 B1, B2, D1, D2; unsigned tag[4]; };

// static_productive() on words already loaded ({sweep iter - 1's, sweep iter's} of the body's slot)
__device__ __forceinline__ bool static_productive_words(unsigned prev, unsigned cur, int iter, int colour)
{
    if (iter == 0) return true;
    if ((prev >> 16) == (unsigned)iter) return true;
    return (cur >> 16) == (unsigned)(iter + 1) && (0xFFFFu - (cur & 0xFFFFu)) < (unsigned)colour;
}

template <bool DO_IMP, bool DO_DISP>
__device__ __forceinline__ void tail_body(const SolverView& v, const int4* s_tab, int c_first, int nclass, int iter)
{
    const int tid = threadIdx.x;
    bool any_imp = false, any_disp = false;
    // Every load of a class step is made by every lane under no branch (a lane without a unit takes the class's first unit's, a unit
    // without a static body the first static slot's words), and in this order: the bodies and tags of the class at hand, THEN the
    // constants of the next class — the arithmetic waits for the first lot with the second still in flight (the counter is in order).
    auto request = [&](int k, TailUnit& r) {
        const int4 tab = s_tab[min(k, nclass - 1)];             // {first slot, leaders, followers, -}: lane t's unit = leader t (+ follower t)
        r.have = k < nclass && tid < tab.y;
        const int u = r.have ? tid : 0;
        const bool has2 = u < tab.z;
        const int s0 = tab.x + u, s1 = tab.x + tab.y + u;
        r.q0 = hbm_load(v, s0, DO_IMP, DO_DISP, false);
        r.q1 = hbm_load(v, has2 ? s1 : s0, DO_IMP, DO_DISP, true);
        r.s0 = s0; r.s1 = has2 ? s1 : -1;
    };
    auto gather = [&](TailUnit& r) {                            // (r's constants have arrived: requested a class ago, behind a barrier since)
        const int b1 = clamp_index(r.q0.k.y, v.nb), b2 = clamp_index(r.q0.k.z, v.nb), slot = max(r.q0.k.w, 0);
        r.B1 = r.B2 = r.D1 = r.D2 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (DO_IMP) { r.B1 = v.sb_imp[b1]; r.B2 = v.sb_imp[b2]; }
        if (DO_DISP) { r.D1 = v.sb_disp[b1]; r.D2 = v.sb_disp[b2]; }
        r.tag[0] = r.tag[1] = r.tag[2] = r.tag[3] = 0u;
        if (DO_IMP) {
            r.tag[0] = __hip_atomic_load(&v.sw_imp[((iter + 1) & 1) * v.nstatic + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r.tag[1] = __hip_atomic_load(&v.sw_imp[(iter & 1) * v.nstatic + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (DO_DISP) {
            r.tag[2] = __hip_atomic_load(&v.sw_disp[((iter + 1) & 1) * v.nstatic + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            r.tag[3] = __hip_atomic_load(&v.sw_disp[(iter & 1) * v.nstatic + slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto sweep = [&](TailUnit& r, int colour) {                // (k_solve_colour's loop body on requested constants and gathered bodies)
        if (!r.have) return;
        const int b1 = r.q0.k.y, b2 = r.q0.k.z, ss = r.q0.k.w;
        float4 B1 = r.B1, B2 = r.B2, D1 = r.D1, D2 = r.D2;
        const float im1 = r.q0.c.y, ii1 = r.q0.c.z, im2 = r.q0.c.w, ii2 = __int_as_float(r.q0.k.x);
        const bool st1 = (im1 == 0.f && ii1 == 0.f), st2 = (im2 == 0.f && ii2 == 0.f);
        const float4 S1 = B1, S2 = B2, T1 = D1, T2 = D2;
        bool tag_imp = false, tag_disp = false, dirty_imp = false, dirty_disp = false;
        const bool sp_imp = DO_IMP && (st1 || st2) && static_productive_words(r.tag[0], r.tag[1], iter, colour);
        const bool sp_disp = DO_DISP && (st1 || st2) && static_productive_words(r.tag[2], r.tag[3], iter, colour);
        solve_one(v, r.s0, r.q0, colour, iter, DO_IMP, DO_DISP, B1, B2, D1, D2, im1, ii1, im2, ii2, st1, st2, ss, sp_imp, sp_disp, any_imp, any_disp, tag_imp, tag_disp, dirty_imp, dirty_disp);
        if (r.s1 >= 0) {                               // a static body's record is never stored: the follower must see it untouched
            if (st1) { B1 = S1; D1 = T1; }
            if (st2) { B2 = S2; D2 = T2; }
            solve_one(v, r.s1, r.q1, colour, iter, DO_IMP, DO_DISP, B1, B2, D1, D2, im1, ii1, im2, ii2, st1, st2, ss, sp_imp, sp_disp, any_imp, any_disp, tag_imp, tag_disp, dirty_imp, dirty_disp);
        }
        if (dirty_imp) { if (!st1) v.sb_imp[b1] = B1; if (!st2) v.sb_imp[b2] = B2; }
        if (dirty_disp) { if (!st1) v.sb_disp[b1] = D1; if (!st2) v.sb_disp[b2] = D2; }
        if (DO_IMP) wave_tag_update(v.sw_imp + (iter & 1) * v.nstatic, tag_imp, ss, static_word(iter, colour));
        if (DO_DISP) wave_tag_update(v.sw_disp + (iter & 1) * v.nstatic, tag_disp, ss, static_word(iter, colour));
    };
    TailUnit a{}, b{};
    request(0, a);
    for (int k = 0; k < nclass; k += 2) {
        gather(a);
        request(k + 1, b);
        sweep(a, c_first + k);
        __syncthreads();                                       // (the class's stores and tag atomics are through: the next class may read them)
        if (k + 1 >= nclass) break;
        gather(b);
        request(k + 2, a);
        sweep(b, c_first + k + 1);
        __syncthreads();
    }
    if (DO_IMP && __any(any_imp) && (threadIdx.x & 63) == 0) v.imp_active[iter] = 1;
    if (DO_DISP && __any(any_disp) && (threadIdx.x & 63) == 0) v.disp_active[iter] = 1;
}

// class_tab: the HBM group's class table (solver.h PartsView); classes [c_first, c_first + nclass), each of at most TAIL_T units
template <bool DO_IMP, bool DO_DISP>
static __global__ void __launch_bounds__(TAIL_T) k_solve_tail(SolverView v, const int4* __restrict__ class_tab, int c_first, int nclass, int iter)
{
    __shared__ int4 s_tab[TAIL_CLASSES_MAX];
    const bool disp_on = DO_DISP && (iter == 0 || v.disp_active[iter - 1] != 0);
    if (!DO_IMP && !disp_on) return;
    if ((int)threadIdx.x < nclass) s_tab[threadIdx.x] = class_tab[c_first + threadIdx.x];
    __syncthreads();
    if (DO_DISP && disp_on) tail_body<DO_IMP, true>(v, s_tab, c_first, nclass, iter);
    else if (DO_IMP)        tail_body<true, false>(v, s_tab, c_first, nclass, iter);
}

// ---- the interior classes of partitioned components: ONE launch per sweep (schedule.h) ------------------------------------
// A merged island (a settled pile: 1e5-1e6 joints in one connected component) is swept class by class out of HBM, one launch
// per class and sweep.  Its interior units — both bodies in one PART of PART_BODIES consecutive indices — occupy the leading
// classes [0, KI) of the HBM group, and parts share no body: here a workgroup takes one part, holds the part's body
// velocities in LDS, and sweeps the part's units class by class with a barrier where the HBM path has a kernel boundary.
// Joint constants and accumulators stay where the HBM path keeps them (q0..q3, acc, dd in schedule order); the arithmetic is
// solve_one()'s, the slot order is the schedule's: results are bit-identical to KI launches of k_solve_colour.
// Interior units touch no static body, so the static tags play no part.
constexpr int PARTS_T = 256;
// (Measured, settled 200k-box world, 392 parts of ~1050 units in 12 classes: 31 us per sweep with the slots of a class in joint
//  order — five scattered 16-byte gathers per joint, 4x the bytes — and 20 us with the interior classes laid out part by part
//  (schedule.h).  Requesting a part's constants ahead of the class steps was built four ways — all of a part's units in registers
//  (512 lanes x 3, 256 x 5, 256 x 4 with a slimmed record: 42-57 us, spills; beyond 128 VGPRs the parts need two rounds), and
//  chunks of two units per lane (no spills, 188 VGPRs): none is faster than this form, whose ~60 VGPRs let every part be
//  resident at once; the step is bound by the part with the longest chain of classes.  Also built: the workgroup's eight waves as
//  four pairs that take the classes in turn, each requesting the constants of its next class the moment it has swept one, with
//  bare `s_waitcnt lgkmcnt(0); s_barrier` between classes so that those loads stay in flight — bit-exact, and 23 us per launch
//  against 15.)

// OWN_ONE (the level-1 launch: a part there has a few dozen units): lane t owns the part's t-th unit and requests its constants
// together with the part's bodies — one memory round trip for the whole launch instead of one per class; units beyond the lanes
// (a part with more than PARTS_T of them) are requested in their class step as in the plain form.
template <bool DO_IMP, bool DO_DISP, bool OWN_ONE>
static __global__ void __launch_bounds__(PARTS_T) k_solve_parts(SolverView v, PartsView pv, int iter)
{
    __shared__ float4 s_imp[DO_IMP ? PART_BODIES : 1];
    __shared__ float4 s_disp[DO_DISP ? PART_BODIES : 1];
    // (round 6: the level's rows of the class tables staged in LDS with the part's bodies — read from memory in the class loop they were a
    //  round trip per class, and twice that in the level-1 launch, whose lanes first walk the classes to find their unit: 8.4 us for a few
    //  dozen units per part)
    __shared__ int4 s_tab[PARTS_CLASS_STRIDE], s_rg[PARTS_CLASS_STRIDE];
    const int part = pv.first_part + (int)blockIdx.x, tid = threadIdx.x;
    const bool imp_on = DO_IMP;
    const bool disp_on = DO_DISP && (iter == 0 || v.disp_active[iter - 1] != 0);
    const int nclass = min(pv.c1 - pv.c0, PARTS_CLASS_STRIDE);
    const int base = part_first_body(part, v.nb);           // (level 1: shifted by half a part; its first part starts below body 0)
    const int units_before = pv.part_begin[part], units_after = pv.part_begin[part + 1];
    if (tid < nclass) { s_tab[tid] = pv.class_tab[pv.c0 + tid]; s_rg[tid] = pv.ranges[(size_t)part * PARTS_CLASS_STRIDE + pv.c0 + tid]; }
    for (int i = tid; i < PART_BODIES; i += PARTS_T) {
        const int g = base + i;
        if (g < 0 || g >= v.nb) continue;
        if (DO_IMP) s_imp[i] = v.sb_imp[g];
        if (DO_DISP) { if (disp_on) s_disp[i] = v.sb_disp[g]; }
    }
    if (units_before == units_after) return;                // nothing of a partitioned component in this part
    if (!imp_on && !disp_on) return;
    __syncthreads();
    int own_c = -1, own_s0 = 0, own_s1 = -1;
    HbmJoint own_q0{}, own_q1{};
    if (OWN_ONE) {
        int before = 0;
        for (int k = 0; k < nclass && own_c < 0; ++k) {
            const int4 tab = s_tab[k], rg = s_rg[k];
            const int n2 = rg.y - rg.x, n = n2 + rg.w - rg.z, u = tid - before;
            if (u < n) {
                own_c = pv.c0 + k;
                own_s0 = u < n2 ? rg.x + u : rg.z + (u - n2);
                own_q0 = hbm_load(v, own_s0, imp_on, disp_on, false);
                if (u < n2) { own_s1 = tab.x + tab.y + (own_s0 - tab.x); own_q1 = hbm_load(v, own_s1, imp_on, disp_on, true); }
            }
            before += n;
        }
    }
    bool any_imp = false, any_disp = false;
    auto sweep = [&](int s0, int s1, HbmJoint& q0, HbmJoint& q1, int c) {
        const int b1 = (q0.k.y - base) & (PART_BODIES - 1), b2 = (q0.k.z - base) & (PART_BODIES - 1);      // (masked: a stale schedule may meet other joints, solver.h)
        float4 B1 = make_float4(0.f, 0.f, 0.f, 0.f), B2 = B1, D1 = B1, D2 = B1;
        if (DO_IMP) { B1 = s_imp[b1]; B2 = s_imp[b2]; }
        if (DO_DISP) { if (disp_on) { D1 = s_disp[b1]; D2 = s_disp[b2]; } }
        const float im1 = q0.c.y, ii1 = q0.c.z, im2 = q0.c.w, ii2 = __int_as_float(q0.k.x);
        bool tag_imp = false, tag_disp = false, dirty_imp = false, dirty_disp = false;
        solve_one(v, s0, q0, c, iter, imp_on, disp_on, B1, B2, D1, D2, im1, ii1, im2, ii2, false, false, -1, false, false, any_imp, any_disp, tag_imp, tag_disp, dirty_imp, dirty_disp);
        if (s1 >= 0)
            solve_one(v, s1, q1, c, iter, imp_on, disp_on, B1, B2, D1, D2, im1, ii1, im2, ii2, false, false, -1, false, false, any_imp, any_disp, tag_imp, tag_disp, dirty_imp, dirty_disp);
        if (DO_IMP) { if (dirty_imp) { s_imp[b1] = B1; s_imp[b2] = B2; } }
        if (DO_DISP) { if (dirty_disp) { s_disp[b1] = D1; s_disp[b2] = D2; } }
    };
    int before = 0;                                         // units of the level's earlier classes in this part
    for (int k = 0; k < nclass; ++k) {
        const int c = pv.c0 + k;
        const int4 tab = s_tab[k], rg = s_rg[k];
        const int n2 = rg.y - rg.x, n = n2 + rg.w - rg.z;      // the part's units of this class: n2 with a follower, then the single ones
        if (OWN_ONE) { if (own_c == c) sweep(own_s0, own_s1, own_q0, own_q1, c); }
        for (int u = OWN_ONE ? max(PARTS_T - before, 0) + tid : tid; u < n; u += PARTS_T) {      // (OWN_ONE: the units no lane owns)
            const bool has2 = u < n2;
            const int s0 = has2 ? rg.x + u : rg.z + (u - n2), i = s0 - tab.x;
            const int s1 = has2 ? tab.x + tab.y + i : -1;
            HbmJoint q0 = hbm_load(v, s0, imp_on, disp_on, false), q1{};
            if (has2) q1 = hbm_load(v, s1, imp_on, disp_on, true);
            sweep(s0, s1, q0, q1, c);
        }
        before += n;
        __syncthreads();
    }
    for (int i = tid; i < PART_BODIES; i += PARTS_T) {
        const int g = base + i;
        if (g < 0 || g >= v.nb) continue;
        if (DO_IMP) v.sb_imp[g] = s_imp[i];
        if (DO_DISP) { if (disp_on) v.sb_disp[g] = s_disp[i]; }
    }
    if (DO_IMP && __any(any_imp) && (threadIdx.x & 63) == 0) v.imp_active[iter] = 1;
    if (DO_DISP && __any(any_disp) && (threadIdx.x & 63) == 0) v.disp_active[iter] = 1;
}


// ---- the same sweep of a level's parts with the NEXT class's constants requested a class ahead (round 6, level 0) ------------------------
// k_solve_parts' class step was an HBM round trip: the step's loads are requested when the step begins (12 classes x ~1.6 us were the
// launch).  Here lane t locates its unit of class c + 1 and requests its constants BEFORE it sweeps its unit of class c; the barrier
// between two steps waits for the LDS traffic only (parts_lds_barrier), so the requests stay in flight across it.  What made earlier
// attempts at this slower: the class tables read by vector loads in the loop (a `vmcnt(0)` per class for two table
// rows: here they are staged in LDS once), register copies of the requested values at the loop's back edge (a wait for them right
// behind the barrier: here two explicit register sets take turns) and a loop over units beyond the lanes in the class step (the
// compiler drains the counter in front of it: an interior class of a part is body-disjoint inside the part's PART_BODIES bodies, so
// it has at most PART_BODIES / 2 = PARTS_T units and there is nothing beyond the lanes).
__device__ __forceinline__ void parts_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
static_assert(PART_BODIES / 2 <= PARTS_T, "a lane per unit of a class");

struct PartUnit { HbmJoint q0, q1; int s0, s1; bool have; };

// the body with BOTH halves decided at compile time: the displacement half's run-time gate (dead after the first sweep of a resting pile)
// is taken once, by the kernel below — a request under a branch, even a uniform one, would be waited for where the branches join
template <bool DO_IMP, bool DO_DISP>
__device__ __forceinline__ void parts_ahead_body(const SolverView& v, const PartsView& pv, int iter, float4* s_imp, float4* s_disp, const int4* s_tab, const int4* s_rg,
                                                 int base, int nclass)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < PART_BODIES; i += PARTS_T) {
        const int g = base + i;
        if (g < 0 || g >= v.nb) continue;
        if (DO_IMP) s_imp[i] = v.sb_imp[g];
        if (DO_DISP) s_disp[i] = v.sb_disp[g];
    }
    __syncthreads();
    bool any_imp = false, any_disp = false;
    // lane t's unit of the level's k-th class: located (n2 units with a follower, then the single ones) and requested.  EVERY lane requests
    // something in every step, under no branch — a lane without a unit the class's first, a unit without a follower its leader's row, a
    // step beyond the last class the last class again: requests under a branch make the compiler wait, where the branches join, as if
    // they had not been made (`vmcnt(0)` in front of the sweep: the request a class ahead was waited for in the same step).
    auto request = [&](int k, PartUnit& r) {
        const int kk = min(k, nclass - 1);
        const int4 tab = s_tab[kk], rg = s_rg[kk];
        const int n2 = rg.y - rg.x, n = n2 + rg.w - rg.z;
        r.have = k < nclass && tid < n;
        const int u = r.have ? tid : 0;
        const bool has2 = u < n2;
        const int s0 = has2 ? rg.x + u : rg.z + (u - n2);      // (a class with no unit in this part: slot 0 — some joint's row, read and dropped)
        const int s1 = tab.x + tab.y + (s0 - tab.x);
        r.q0 = hbm_load(v, s0, DO_IMP, DO_DISP, false);
        r.q1 = hbm_load(v, has2 ? s1 : s0, DO_IMP, DO_DISP, true);
        r.s0 = s0; r.s1 = has2 ? s1 : -1;
    };
    auto sweep = [&](PartUnit& r, int c) {
        if (!r.have) return;
        const int b1 = (r.q0.k.y - base) & (PART_BODIES - 1), b2 = (r.q0.k.z - base) & (PART_BODIES - 1);      // (masked: a stale schedule may meet other joints, solver.h)
        float4 B1 = make_float4(0.f, 0.f, 0.f, 0.f), B2 = B1, D1 = B1, D2 = B1;
        if (DO_IMP) { B1 = s_imp[b1]; B2 = s_imp[b2]; }
        if (DO_DISP) { D1 = s_disp[b1]; D2 = s_disp[b2]; }
        const float im1 = r.q0.c.y, ii1 = r.q0.c.z, im2 = r.q0.c.w, ii2 = __int_as_float(r.q0.k.x);
        bool tag_imp = false, tag_disp = false, dirty_imp = false, dirty_disp = false;
        solve_one(v, r.s0, r.q0, c, iter, DO_IMP, DO_DISP, B1, B2, D1, D2, im1, ii1, im2, ii2, false, false, -1, false, false, any_imp, any_disp, tag_imp, tag_disp, dirty_imp, dirty_disp);
        if (r.s1 >= 0)
            solve_one(v, r.s1, r.q1, c, iter, DO_IMP, DO_DISP, B1, B2, D1, D2, im1, ii1, im2, ii2, false, false, -1, false, false, any_imp, any_disp, tag_imp, tag_disp, dirty_imp, dirty_disp);
        if (DO_IMP) { if (dirty_imp) { s_imp[b1] = B1; s_imp[b2] = B2; } }
        if (DO_DISP) { if (dirty_disp) { s_disp[b1] = D1; s_disp[b2] = D2; } }
    };
    // (ONE class ahead, two register sets taking turns.  Two ahead — three sets, 112 registers — measured slower: 17.2 against 16.3 us per
    //  launch in the settled 200k world, the plain form 19.8.)
    PartUnit a{}, b{};
    request(0, a);
    for (int k = 0; k < nclass; k += 2) {
        request(k + 1, b);
        sweep(a, pv.c0 + k);
        parts_lds_barrier();
        if (k + 1 >= nclass) break;
        request(k + 2, a);
        sweep(b, pv.c0 + k + 1);
        parts_lds_barrier();
    }
    for (int i = tid; i < PART_BODIES; i += PARTS_T) {
        const int g = base + i;
        if (g < 0 || g >= v.nb) continue;
        if (DO_IMP) v.sb_imp[g] = s_imp[i];
        if (DO_DISP) v.sb_disp[g] = s_disp[i];
    }
    if (DO_IMP && __any(any_imp) && (threadIdx.x & 63) == 0) v.imp_active[iter] = 1;
    if (DO_DISP && __any(any_disp) && (threadIdx.x & 63) == 0) v.disp_active[iter] = 1;
}

template <bool DO_IMP, bool DO_DISP>
static __global__ void __launch_bounds__(PARTS_T) k_solve_parts_ahead(SolverView v, PartsView pv, int iter)
{
    __shared__ float4 s_imp[DO_IMP ? PART_BODIES : 1];
    __shared__ float4 s_disp[DO_DISP ? PART_BODIES : 1];
    __shared__ int4 s_tab[PARTS_CLASS_STRIDE], s_rg[PARTS_CLASS_STRIDE];
    const int part = pv.first_part + (int)blockIdx.x, tid = threadIdx.x;
    if (pv.part_begin[part] == pv.part_begin[part + 1]) return;      // nothing of a partitioned component in this part
    const bool disp_on = DO_DISP && (iter == 0 || v.disp_active[iter - 1] != 0);
    if (!DO_IMP && !disp_on) return;
    const int base = part_first_body(part, v.nb);
    const int nclass = min(pv.c1 - pv.c0, PARTS_CLASS_STRIDE);
    if (nclass <= 0) return;
    if (tid < nclass) { s_tab[tid] = pv.class_tab[pv.c0 + tid]; s_rg[tid] = pv.ranges[(size_t)part * PARTS_CLASS_STRIDE + pv.c0 + tid]; }
    // (the tables are published by the body's first barrier, behind its load of the part's bodies)
    if (DO_DISP && disp_on) parts_ahead_body<DO_IMP, true>(v, pv, iter, s_imp, s_disp, s_tab, s_rg, base, nclass);
    else if (DO_IMP)        parts_ahead_body<true, false>(v, pv, iter, s_imp, s_disp, s_tab, s_rg, base, nclass);
}

// PreStepJoints of the interior classes, the same way (k_prestep's arithmetic and order)
static __global__ void __launch_bounds__(PARTS_T) k_prestep_parts(SolverView v, PartsView pv)
{
    __shared__ float4 s_imp[PART_BODIES];
    __shared__ int4 s_tab[PARTS_CLASS_STRIDE], s_rg[PARTS_CLASS_STRIDE];      // (the level's rows of the class tables: k_solve_parts)
    const int part = pv.first_part + (int)blockIdx.x;
    const int nclass = min(pv.c1 - pv.c0, PARTS_CLASS_STRIDE);
    const int base = part_first_body(part, v.nb);
    const int units_before = pv.part_begin[part], units_after = pv.part_begin[part + 1];
    if ((int)threadIdx.x < nclass) { s_tab[threadIdx.x] = pv.class_tab[pv.c0 + threadIdx.x]; s_rg[threadIdx.x] = pv.ranges[(size_t)part * PARTS_CLASS_STRIDE + pv.c0 + threadIdx.x]; }
    for (int i = threadIdx.x; i < PART_BODIES; i += PARTS_T) { const int g = base + i; if (g >= 0 && g < v.nb) s_imp[i] = v.sb_imp[g]; }
    if (units_before == units_after) return;
    __syncthreads();
    for (int k = 0; k < nclass; ++k) {
        const int4 tab = s_tab[k], rg = s_rg[k];
        const int n2 = rg.y - rg.x, n = n2 + rg.w - rg.z;
        for (int u = (int)threadIdx.x; u < n; u += PARTS_T) {
            const int s0 = u < n2 ? rg.x + u : rg.z + (u - n2), i = s0 - tab.x;
            const float4 m = v.q2[s0];
            const int4 k3 = v.q3[s0];
            const float im1 = m.y, ii1 = m.z, im2 = m.w, ii2 = __int_as_float(k3.x);
            const int l1 = (k3.y - base) & (PART_BODIES - 1), l2 = (k3.z - base) & (PART_BODIES - 1);
            float4 B1 = s_imp[l1], B2 = s_imp[l2];
            prestep_one(v, s0, B1, B2, im1, ii1, im2, ii2, false, false);
            if (i < tab.z) prestep_one(v, tab.x + tab.y + i, B1, B2, im1, ii1, im2, ii2, false, false);
            s_imp[l1] = B1; s_imp[l2] = B2;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < PART_BODIES; i += PARTS_T) { const int g = base + i; if (g >= 0 && g < v.nb) v.sb_imp[g] = s_imp[i]; }
}

// ---- FinishJoints + FinishBodies (ref: Solver.cpp:482-494, 527-547) --------------------------------
// The two finish kernels (and the island kernel's epilogue) are the only places that write to the caller's
// arrays.  They commit only if the topology fingerprint computed for THIS call equals the one the schedule was
// built for; otherwise the solve ran on a stale schedule, nothing is written, and the host rebuilds and repeats.
static __global__ void __launch_bounds__(256) k_finish_joints(SolverView v, int begin, int end, phx_contact_joint* __restrict__ joints)
{
    if (*v.fingerprint != v.expected_fingerprint) return;
    for (int s = begin + blockIdx.x * blockDim.x + threadIdx.x; s < end; s += gridDim.x * blockDim.x) {
        const float2 acc = v.acc[s];
        phx_contact_joint& j = joints[v.order[s]];
        j.normal_accumulated_impulse = acc.x;
        j.friction_accumulated_impulse = acc.y;
    }
}

static __global__ void __launch_bounds__(256) k_finish_bodies(SolverView v, const int* __restrict__ list, int count, BodyView bodies)
{
    if (*v.fingerprint != v.expected_fingerprint) { solve_stamp_end_sparse(v.stamps); return; }
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < count; k += gridDim.x * blockDim.x) {
        const int i = list[k];
        float4 a = v.sb_imp[i], d = v.sb_disp[i];
        a.w = 0.f; d.w = 0.f;
        bodies.vel[i] = a;
        bodies.dvel[i] = d;
    }
    solve_stamp_end_sparse(v.stamps);
}

} // namespace phx
