// island_kernel.h — the island kernel (included by islands.hip only; see island_view.h for why it has a translation unit of
// its own).
#pragma once

#include "solver_kernels.h"

#include <type_traits>

namespace phx {

// ---- island kernel: one workgroup solves one GROUP of the schedule entirely out of LDS -----------------
// (Refresh, PreStep and every impulse / displacement sweep of ref: Solver.cpp:130-215 SolveJointIsland.)
// A lane owns one UNIT (schedule.h: the one or two joints of a body pair) for the whole solve: the refreshed constants and
// the accumulators of both joints live in registers, the group's body velocities live in LDS, classes are separated by
// workgroup barriers instead of kernel launches, and HBM is touched once on the way in and once on the way out.  A step
// sweeps the unit's leader and then its follower on one read and one write of the two bodies: half the barriers and LDS
// round trips per joint of the one-joint-per-lane kernel of round 2.  The early exit of ref: Solver.cpp:189 / :210 is per
// group, exactly like the reference's per-island loop.
// Two shapes: 256 lanes / 512 joints / 768 bodies (4 workgroups per CU — the 200-box columns of cfg 2 are ~205 units each)
// and 512 lanes / 1024 joints / 1024 bodies (2 per CU — the 500-box columns of cfg 5 are ~510 units each); both leave a
// lane 128 VGPRs.
// phase stamps of the island kernel (tools/island_trace.py; the constant 100 MHz clock all XCDs share): 0 start, 1 records loaded, 2 refreshed, 3 pre-stepped, 4 swept,
// 5 written back; word 6 = XCC id | s_memtime ticks of the whole workgroup << 4, word 7 = classes << 32 | impulse sweeps executed
#define PHX_ISL_STAMP(k) do { if (TRACE && threadIdx.x == 0) iv.trace[(size_t)group * 8 + (k)] = wall_clock64(); } while (0)

// ---- body-state storage of the island kernel: fp32 (default) or the fp16 ablation ------------------------------
template <bool HALF> struct BodyStore { using type = float4; };
template <> struct BodyStore<true> { using type = uint2; };

__device__ __forceinline__ float4 body_load(const float4* p, int i) { return p[i]; }
__device__ __forceinline__ void body_store(float4* p, int i, float4 v) { p[i] = v; }
__device__ __forceinline__ unsigned f2h_bits(float f) { return (unsigned)__half_as_ushort(__float2half_rn(f)); }
__device__ __forceinline__ float h2f_bits(unsigned h) { return __half2float(__ushort_as_half((unsigned short)h)); }
__device__ __forceinline__ float4 body_load(const uint2* p, int i)
{
    const uint2 r = p[i];
    return make_float4(h2f_bits(r.x & 0xFFFFu), h2f_bits(r.x >> 16), h2f_bits(r.y & 0xFFFFu), __int_as_float((int)(short)(r.y >> 16)));
}
__device__ __forceinline__ void body_store(uint2* p, int i, float4 v)
{
    p[i] = make_uint2(f2h_bits(v.x) | (f2h_bits(v.y) << 16), f2h_bits(v.z) | ((unsigned)(unsigned short)(short)__float_as_int(v.w) << 16));
}

// fp16 ablation: what a store + load of the body record would leave in the registers (identity for fp32 state)
template <bool HALF> __device__ __forceinline__ float4 body_round(float4 v)
{
    if (!HALF) return v;
    return make_float4(h2f_bits(f2h_bits(v.x)), h2f_bits(f2h_bits(v.y)), h2f_bits(f2h_bits(v.z)), __int_as_float((int)(short)__float_as_int(v.w)));
}

// the refreshed constants and accumulators of one joint, in registers for the whole solve
struct IslJoint { float nx, ny, aN1, aN2, aF1, aF2, cimN, cimF, dstV, dstD, accN, accF, accD; };

// RefreshJoints (ref: Solver.cpp:642-693) — same expressions as k_pack_refresh
__device__ __forceinline__ void isl_refresh(IslJoint& q, float d1x, float d1y, float d2x, float d2y, const float4& p1, const float4& p2)
{
    const float pt1x = d1x + p1.z, pt1y = d1y + p1.w;
    const float pt2x = d2x + p2.z, pt2y = d2y + p2.w;
    const float w2x = pt1x - p2.z, w2y = pt1y - p2.w;
    const Limiter N = refresh_limiter(q.nx, q.ny, d1x, d1y, w2x, w2y, p1.x, p1.y, p2.x, p2.y);
    const Limiter F = refresh_limiter(-q.ny, q.nx, d1x, d1y, w2x, w2y, p1.x, p1.y, p2.x, p2.y);
    const float depth = (pt2x - pt1x) * q.nx + (pt2y - pt1y) * q.ny;
    const float dst = 0.f;
    q.dstV = depth < 1.f ? dst - 0.1f : dst;
    q.dstD = 0.1f * max_ref(0.f, depth - 2.0f * 1.f);
    q.aN1 = N.a1; q.aN2 = N.a2; q.cimN = N.cim; q.aF1 = F.a1; q.aF2 = F.a2; q.cimF = F.cim;
}

// PreStepJoints (ref: Solver.cpp:736-750) of one joint on the bodies held in registers
__device__ __forceinline__ void isl_prestep(const IslJoint& q, float4& B1, float4& B2, float im1, float ii1, float im2, float ii2)
{
    const float tx = -q.ny, ty = q.nx;
    B1.x = mul_add(q.nx * im1, q.accN, B1.x); B1.y = mul_add(q.ny * im1, q.accN, B1.y); B1.z = mul_add(q.aN1 * ii1, q.accN, B1.z);
    B1.x = mul_add(tx * im1, q.accF, B1.x); B1.y = mul_add(ty * im1, q.accF, B1.y); B1.z = mul_add(q.aF1 * ii1, q.accF, B1.z);
    B2.x = mul_add((-q.nx) * im2, q.accN, B2.x); B2.y = mul_add((-q.ny) * im2, q.accN, B2.y); B2.z = mul_add(q.aN2 * ii2, q.accN, B2.z);
    B2.x = mul_add((-tx) * im2, q.accF, B2.x); B2.y = mul_add((-ty) * im2, q.accF, B2.y); B2.z = mul_add(q.aF2 * ii2, q.accF, B2.z);
}

// the arithmetic of one impulse visit (ref: Solver.cpp:800-896) on the two bodies held in registers; returns whether the joint moved
// (kProductiveImpulse, ref: Solver.cpp:8, 894-896).  The skip test and the tags are the caller's.
__device__ __forceinline__ bool isl_impulse_eval(IslJoint& q, float4& B1, float4& B2, float im1, float ii1, float im2, float ii2)
{
    // (Measured and removed: the x / y halves of every body-wide step as v_pk_mul_f32 / v_pk_add_f32 on the register pairs a
    //  ds_read_b128 leaves — 24 VALU instructions fewer per unit of ~145, no extra moves, bit-exact — is 3 % SLOWER: the step is a
    //  dependent chain, a packed fp32 operation occupies the pipe twice as long as a plain one, and nothing waits to fill the slots
    //  it frees.  DESIGN.md §4.2.)
    const float nx = q.nx, ny = q.ny, tx = -ny, ty = nx;
    float dv = q.dstV;
    dv = mul_sub(nx, B1.x, dv); dv = mul_sub(ny, B1.y, dv); dv = mul_sub(q.aN1, B1.z, dv);
    dv = mul_sub(-nx, B2.x, dv); dv = mul_sub(-ny, B2.y, dv); dv = mul_sub(q.aN2, B2.z, dv);
    float dn = dv * q.cimN;
    dn = max_ref(dn, -q.accN);
    B1.x = mul_add(nx * im1, dn, B1.x); B1.y = mul_add(ny * im1, dn, B1.y); B1.z = mul_add(q.aN1 * ii1, dn, B1.z);
    B2.x = mul_add((-nx) * im2, dn, B2.x); B2.y = mul_add((-ny) * im2, dn, B2.y); B2.z = mul_add(q.aN2 * ii2, dn, B2.z);
    q.accN += dn;
    float fv = 0.f;
    fv = mul_sub(tx, B1.x, fv); fv = mul_sub(ty, B1.y, fv); fv = mul_sub(q.aF1, B1.z, fv);
    fv = mul_sub(-tx, B2.x, fv); fv = mul_sub(-ty, B2.y, fv); fv = mul_sub(q.aF2, B2.z, fv);
    float df = fv * q.cimF;
    const float force = q.accF + df;
    const float limit = q.accN * 0.3f;
    const float signed_limit = force < 0.f ? -limit : limit;
    const float adjusted = signed_limit - q.accF;
    if (fabsf(force) > limit) df = adjusted;
    q.accF += df;
    B1.x = mul_add(tx * im1, df, B1.x); B1.y = mul_add(ty * im1, df, B1.y); B1.z = mul_add(q.aF1 * ii1, df, B1.z);
    B2.x = mul_add((-tx) * im2, df, B2.x); B2.y = mul_add((-ty) * im2, df, B2.y); B2.z = mul_add(q.aF2 * ii2, df, B2.z);
    return max_ref(fabsf(dn), fabsf(df)) > 1e-4f;
}

// the arithmetic of one displacement visit (ref: Solver.cpp:960-1005) on the two bodies' displacing velocities held in registers;
// returns whether the joint moved.  The skip test and the tags are the caller's.
__device__ __forceinline__ bool isl_displace_eval(IslJoint& q, float4& D1, float4& D2, float im1, float ii1, float im2, float ii2)
{
    const float nx = q.nx, ny = q.ny;
    float dv = q.dstD;
    dv = mul_sub(nx, D1.x, dv); dv = mul_sub(ny, D1.y, dv); dv = mul_sub(q.aN1, D1.z, dv);
    dv = mul_sub(-nx, D2.x, dv); dv = mul_sub(-ny, D2.y, dv); dv = mul_sub(q.aN2, D2.z, dv);
    float di = dv * q.cimN;
    di = max_ref(di, -q.accD);
    D1.x = mul_add(nx * im1, di, D1.x); D1.y = mul_add(ny * im1, di, D1.y); D1.z = mul_add(q.aN1 * ii1, di, D1.z);
    D2.x = mul_add((-nx) * im2, di, D2.x); D2.y = mul_add((-ny) * im2, di, D2.y); D2.z = mul_add(q.aN2 * ii2, di, D2.z);
    q.accD += di;
    return fabsf(di) > 1e-4f;
}

template <int T, int NB, bool HALF, bool TRACE = false>
__global__ void __launch_bounds__(T, 4) k_solve_islands(SolverView v, IslandView iv, BodyView bv,
                                                            phx_contact_joint* __restrict__ joints,
                                                            const phx_contact_point* __restrict__ cps, int ci, int pi)
{
    // body velocities in LDS: float4 {vx, vy, w, tag}, or — fp16 body-state ablation (BASELINE config 5) — four 16-bit
    // words {half vx, half vy, half w, int16 tag}; arithmetic is fp32 either way, HALF rounds on every store
    using BodyT = typename BodyStore<HALF>::type;
    __shared__ BodyT imp[NB];
    __shared__ BodyT disp[NB];
    // static-tag words [imp|disp][parity][body]; during set-up the same 12 KB hold {invMass, invInertia, pos} per body
    __shared__ __attribute__((aligned(16))) unsigned sw_raw[4 * NB];
    __shared__ unsigned char is_st[NB];
    __shared__ int flag_imp[3], flag_disp[3];     // 'some joint was productive in sweep it': slot it % 3
    __shared__ int s_commit;                      // ISL_VERIFY: every workgroup of the launch arrived and none found a difference
    unsigned (*swi)[NB] = reinterpret_cast<unsigned (*)[NB]>(sw_raw);
    unsigned (*swd)[NB] = reinterpret_cast<unsigned (*)[NB]>(sw_raw + 2 * NB);
    float4* par = reinterpret_cast<float4*>(sw_raw);

    const int group = iv.group_list ? iv.group_list[blockIdx.x] : (int)blockIdx.x;
    if (iv.next_ctl && blockIdx.x == 0) {                 // first kernel of its solve: the NEXT solve's control set (two sets alternate)
        if (threadIdx.x < ISL_STAT_SLOTS) { iv.next_visits[threadIdx.x] = 0ull; iv.next_executed[2 * threadIdx.x] = 0; iv.next_executed[2 * threadIdx.x + 1] = 0; }
        if (threadIdx.x == 0) { *iv.next_ctl = 0ull; iv.next_visits[ISL_STAT_SLOTS] = ~0ull; iv.next_visits[ISL_STAT_SLOTS + 1] = 0ull; }
        if (threadIdx.x < ISL_SHARDS) iv.next_shards[threadIdx.x * ISL_SHARD_STRIDE] = 0ull;
    }
    if (iv.stamp_begin) solve_stamp_begin(v.stamps);      // (no HBM group in front of this launch: it is the solve's first kernel)
    if (iv.ngroups_dev && group >= *iv.ngroups_dev) return;  // (workgroup-uniform, in front of every barrier)
    if (iv.mode == ISL_COMPLETE && iv.done[group] == iv.epoch) return;      // (workgroup-uniform) committed by the launch this one completes
    PHX_ISL_STAMP(0);
    const unsigned long long cycles0 = TRACE ? __builtin_readcyclecounter() : 0ull;
    // TRACE: per wave, shader cycles spent in class steps {working: in the unit update, then at the barrier; idle: whole step}
    unsigned long long tw_work = 0, tw_bar = 0, tw_idle = 0, tw_work_big = 0; unsigned tw_nwork = 0, tw_nidle = 0, tw_nbig = 0;
    const int tid = threadIdx.x;

    // Set-up is two dependent HBM round trips: level 1 = the group's descriptor, the lane's unit record and its body ids (all at
    // addresses that depend on the group number only), level 2 = the body records, the joints and the contact points.
    constexpr int BI = (NB + T - 1) / T;                   // body records per lane
    int body_id[BI];
#pragma unroll
    for (int k = 0; k < BI; ++k) body_id[k] = tid + k * T < NB ? iv.bodies[(size_t)group * NB + tid + k * T] : -1;      // (entries past the group's count: unused words of its table)
    const int4 ua = iv.unit_recs[2 * ((size_t)group * T + tid)], ub = iv.unit_recs[2 * ((size_t)group * T + tid) + 1];
    const int4 d = iv.desc[group];
    const int units_word = iv.units[group];
    // (schedule.h LANES: the classes' lane ranges sit on wave boundaries where the lanes allow it — a lane has a unit or it has not)
    const int ncol = island_word_classes(units_word), nstatic = island_word_static(units_word);
    const bool live = ua.x >= 0;
#pragma unroll
    for (int k = 0; k < BI; ++k) if (tid + k * T >= d.w) body_id[k] = -1;
    const bool has2 = live && ua.y >= 0;
    const int jid0 = live ? ua.x : 0, jid1 = has2 ? ua.y : 0;
    const unsigned loc = live ? (unsigned)ub.x : 0u;
    const int col = live ? ub.y : -1;
    if (tid < 3) { flag_imp[tid] = 0; flag_disp[tid] = 0; }

    float4 rec_imp[BI], rec_disp[BI], rec_par[BI];
#pragma unroll
    for (int k = 0; k < BI; ++k) {                         // level 2: the resident arrays ARE PrepareBodies' staged form
        if (body_id[k] < 0) continue;                      //          (ref: Solver.cpp:456-480; body_view.h): three coalesced 16-byte loads
        rec_imp[k] = bv.vel[body_id[k]]; rec_imp[k].w = __int_as_float(-1);
        rec_disp[k] = bv.dvel[body_id[k]]; rec_disp[k].w = __int_as_float(-1);
        rec_par[k] = bv.mpos[body_id[k]];
    }
    IslJoint q0{}, q1{};
    float4 da0 = make_float4(0.f, 0.f, 0.f, 0.f), da1 = da0;     // delta1, delta2 of the two contact points
    int l1 = 0, l2 = 0;
    const bool verify = iv.mode == ISL_VERIFY;             // (launch-uniform)
    bool differs = false;                                  // ISL_VERIFY: the schedule was built for other joints / other static bodies
    phx_contact_joint jf{};
    if (has2) jf = joints[jid1];
    if (live) {                                            // PrepareJoints (ref: Solver.cpp:509-521)
        const phx_contact_joint j = joints[jid0];
        // (the contact point index is part of the topology the schedule was built — and is gated — for: ua.z == j.contact_point_index)
        const float4* cp4 = reinterpret_cast<const float4*>(&cps[clamp_index(ua.z, v.ncp)]);   // 32-byte records
        da0 = cp4[0];
        const float2 nn = *reinterpret_cast<const float2*>(cp4 + 1);
        q0.nx = nn.x; q0.ny = nn.y;
        l1 = (int)(loc & 0xFFFFu); l2 = (int)(loc >> 16);
        q0.accN = j.normal_accumulated_impulse; q0.accF = j.friction_accumulated_impulse;
        if (verify) {      // the unit record against the joint it points at (the body ids: two more words of the group's table, L2-warm)
            const int g1 = iv.bodies[(size_t)group * NB + l1], g2 = iv.bodies[(size_t)group * NB + l2];
            differs = j.contact_point_index != ua.z || j.body1 != g1 || j.body2 != g2;
            if (has2) differs |= jf.contact_point_index != ua.w || jf.body1 != g1 || jf.body2 != g2;
        }
    }
    if (has2) {
        const float4* cp4 = reinterpret_cast<const float4*>(&cps[clamp_index(ua.w, v.ncp)]);
        da1 = cp4[0];
        const float2 nn = *reinterpret_cast<const float2*>(cp4 + 1);
        q1.nx = nn.x; q1.ny = nn.y;
        q1.accN = jf.normal_accumulated_impulse; q1.accF = jf.friction_accumulated_impulse;
    }
#pragma unroll
    for (int k = 0; k < BI; ++k) {
        if (body_id[k] < 0) continue;
        const int i = tid + k * T;
        body_store(imp, i, rec_imp[k]);
        body_store(disp, i, rec_disp[k]);
        par[i] = rec_par[k];
        const bool st = rec_par[k].x == 0.f && rec_par[k].y == 0.f;
        is_st[i] = st ? 1 : 0;
        differs |= st != (i < nstatic);                    // (the builders list a group's static bodies first)
    }
    unsigned long long arrived_before = 0ull;              // (lane 0) what the shard counter read when this workgroup arrived
    unsigned my_arrival = 1u;
    if (verify) {
        // ARRIVE: this workgroup has compared everything it owns.  One device-scope atomic carries the arrival and the verdict, so
        // whoever sees all arrivals also sees every verdict (island_view.h).  The returned value is looked at behind PreStep:
        // nothing waits for this round trip.
        const int any = __syncthreads_or(differs ? 1 : 0);
        my_arrival = any ? 1u + ISL_BAD : 1u;
        if (tid == 0) arrived_before = atomicAdd(&iv.shards[((int)blockIdx.x % ISL_SHARDS) * ISL_SHARD_STRIDE], (unsigned long long)my_arrival);
    } else __syncthreads();
    PHX_ISL_STAMP(1);
    float im1 = 0.f, ii1 = 0.f, im2 = 0.f, ii2 = 0.f;
    if (live) {
        const float4 p1 = par[l1], p2 = par[l2];           // {im, ii, pos.x, pos.y} of the two bodies
        isl_refresh(q0, da0.x, da0.y, da0.z, da0.w, p1, p2);
        if (has2) isl_refresh(q1, da1.x, da1.y, da1.z, da1.w, p1, p2);
        im1 = p1.x; ii1 = p1.y; im2 = p2.x; ii2 = p2.y;
    }
    // THE DISPLACEMENT HALF OF A GROUP THAT HAS NOTHING TO PUSH APART IS A NO-OP, bit for bit: if every displacing velocity of the group
    // is +0, no joint is deeper than the allowed penetration (dstD = +0, ref: Solver.cpp:672-680) and everything a visit multiplies is
    // finite, then a visit (ref: Solver.cpp:960-1005) computes dv = +0 - (+-0) ... = +0, di = max(+-0, -accD = -0) = -0, adds (finite x -0)
    // to +0 velocities (= +0) and -0 to accD = +0 (= +0), and is not productive — in either arithmetic form.  So the reference's first
    // displacement sweep finds nothing productive and is its last (ref: Solver.cpp:210); the group skips it and reports the one sweep
    // (a resting stack's every solve: 1.5 us of cfg 2's launch).
    auto finite_bits = [](float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; };
    auto joint_quiet = [&](const IslJoint& q) {
        return __float_as_uint(q.dstD) == 0u && finite_bits(q.nx) && finite_bits(q.ny) && finite_bits(q.aN1) && finite_bits(q.aN2) && finite_bits(q.cimN)
               && finite_bits(im1) && finite_bits(ii1) && finite_bits(im2) && finite_bits(ii2);
    };
    bool stirs = live && !(joint_quiet(q0) && (!has2 || joint_quiet(q1)));
#pragma unroll
    for (int k = 0; k < BI; ++k)
        if (body_id[k] >= 0) stirs |= (__float_as_uint(rec_disp[k].x) | __float_as_uint(rec_disp[k].y) | __float_as_uint(rec_disp[k].z)) != 0u;
    const bool disp_quiet = __syncthreads_or(stirs ? 1 : 0) == 0;      // (the barrier that was here anyway)
    for (int i = tid; i < 4 * NB; i += T) sw_raw[i] = 0;     // the parameter table is dead: now the tag words
    const bool st1 = (im1 == 0.f && ii1 == 0.f), st2 = (im2 == 0.f && ii2 == 0.f);
    const bool wave_static = __any(live && (st1 || st2));      // (wave-uniform, fixed for the solve)
    int sm1 = st1 ? -1 : 0, sm2 = st2 ? -1 : 0;                // (as masks: the hot form selects with them instead of branching)
    asm volatile("" : "+v"(sm1), "+v"(sm2));
    __syncthreads();
    PHX_ISL_STAMP(2);

    // PreStepJoints (ref: Solver.cpp:736-750), class by class: leader, then follower
    for (int c = 0; c < ncol; ++c) {
        if (col == c) {
            float4 B1 = body_load(imp, l1), B2 = body_load(imp, l2);
            isl_prestep(q0, B1, B2, im1, ii1, im2, ii2);
            if (has2) {
                if (HALF) { B1 = body_round<HALF>(B1); B2 = body_round<HALF>(B2); }      // (the ablation rounds on every joint's store)
                isl_prestep(q1, B1, B2, im1, ii1, im2, ii2);
            }
            if (!st1) body_store(imp, l1, B1);
            if (!st2) body_store(imp, l2, B2);
        }
        __syncthreads();
    }

    if (verify && tid == 0) {
        // the last arriver of a shard forwards the shard's verdict to the solve's control word
        const unsigned now = (unsigned)arrived_before + my_arrival;
        const unsigned shard = blockIdx.x % ISL_SHARDS, want = (iv.nexpect - shard + ISL_SHARDS - 1) / ISL_SHARDS;
        if ((now & ISL_ARRIVE_MASK) == want) atomicAdd(iv.ctl, now >= ISL_BAD ? (unsigned long long)(1u + ISL_BAD) : 1ull);
    }
    PHX_ISL_STAMP(3);
    int done_imp = 0, done_disp = (pi > 0 && disp_quiet) ? 1 : 0;      // (a quiet group's one displacement sweep: see above)
    bool imp_alive = ci > 0, disp_alive = pi > 0 && !disp_quiet;
    const int iters = ci > pi ? ci : pi;
    unsigned early_ctl = 0u;                               // (lane 0) the control word as read a few sweeps in: by then every workgroup has long arrived
    int it = 0, c = 0, slot = 0;                           // the sweep, the class and the flag slot the step forms below work on
    // ISL_VERIFY: look at the control word a few sweeps in, off the critical path (the load returns while the sweeps run); the commit
    // decision at the end then needs no memory round trip of its own unless some workgroup really is that late
    auto peek_ctl = [&]() { if (verify && it == 3 && tid == 0 && iv.wait_polls > 0) early_ctl = (unsigned)__hip_atomic_load(iv.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    // three flag slots in rotation: the one cleared for the next sweep was read last at the end of sweep it - 2, and every class
    // step of sweep it - 1 has put a barrier in between — so a sweep needs no barrier of its own at its end
    auto next_slot = [&]() { slot = it % 3; if (tid == 0) { const int next = slot == 2 ? 0 : slot + 1; flag_imp[next] = 0; flag_disp[next] = 0; } };
    // TRACE level 2 (phx_solver_set_trace: per-wave cycle counts of every class step — ~15 % slower)
    const bool wt = TRACE && iv.wave_trace;
    unsigned long long ts0 = 0ull, ts1 = 0ull; bool working = false;
    auto step_begin = [&]() { if (wt) { ts0 = __builtin_readcyclecounter(); working = __any(col == c); } };
    auto step_work_done = [&]() { if (wt) ts1 = __builtin_readcyclecounter(); };
    auto step_end = [&]() {
        if (!wt) return;
        const unsigned long long ts2 = __builtin_readcyclecounter();
        if (working) {
            if (__popcll(__ballot(col == c)) > 32) { tw_work_big += ts1 - ts0; ++tw_nbig; } else { tw_work += ts1 - ts0; ++tw_nwork; }
            tw_bar += ts2 - ts1;
        } else { tw_idle += ts2 - ts0; ++tw_nidle; }
    };
    // A CLASS STEP of one sweep half (IMP: the impulses on the velocities, ref: Solver.cpp:790-896; else the displacement on the displacing
    // velocities, ref: Solver.cpp:960-1005) for the lane's unit: both body records in one LDS round trip, ONE skip test per unit — the
    // follower's test equals its leader's: a skipped leader changes no tag, and an evaluated one either leaves the tags as they were or
    // raises them to `it` — and one tag update; straight-line but for the follower's mask: a class step is one wave's instruction
    // stream, and every taken branch in it is ~20 cycles.  `ws` = some unit of the wave touches a static body (wave-uniform, fixed for
    // the solve): only then the static tags are looked up and raised (they are class-synchronous: solver_kernels.h), the static records
    // restored between the joints (a static body's record is never stored: the follower must see it untouched) and left unstored.
    // (Rounds 2-4 had a general form beside this one — a skip test and a tag update per joint, two dependent LDS round trips — which
    //  every first sweep still took: same results, ~2.5 x the cycles.)
    auto half_step = [&](auto IMPC, const bool ws) {
        constexpr bool IMP = decltype(IMPC)::value;
        BodyT* const rec = IMP ? imp : disp;
        unsigned (*const sw)[NB] = IMP ? swi : swd;
        float4 B1 = body_load(rec, l1), B2 = body_load(rec, l2);
        // (`ws`: the static tags' words travel with the body records — every lane of the wave reads them, a dynamic body's
        //  are zero — instead of two more dependent LDS round trips for the one lane that needs them)
        unsigned pw1 = 0u, cw1 = 0u, pw2 = 0u, cw2 = 0u;
        if (ws) { pw1 = sw[(it - 1) & 1][l1]; cw1 = sw[it & 1][l1]; pw2 = sw[(it - 1) & 1][l2]; cw2 = sw[it & 1][l2]; }
        // (everything in ONE LDS round trip: left alone, the compiler reads the two tags, tests, and only then — under the
        //  branch — the six velocity words: two dependent round trips on the critical path of every class step)
        if (!HALF) asm volatile("" : "+v"(B1.x), "+v"(B1.y), "+v"(B1.z), "+v"(B1.w), "+v"(B2.x), "+v"(B2.y), "+v"(B2.z), "+v"(B2.w));
        if (ws) asm volatile("" : "+v"(pw1), "+v"(cw1), "+v"(pw2), "+v"(cw2));
        bool active = max(__float_as_int(B1.w), __float_as_int(B2.w)) > it - 2;
        if (ws) {
            // solver_kernels.h static_productive on the words already here — every lane evaluates both bodies' tests and selects (no
            // short-circuit: as `st1 && sp(..)` this was four exec-mask regions in the one wave whose step everybody waits for)
            const unsigned itu = (unsigned)it, clu = (unsigned)c;
            auto sp = [&](unsigned pw, unsigned cw) {
                return (int)(it == 0) | (int)((pw >> 16) == itu) | ((int)((cw >> 16) == itu + 1u) & (int)((0xFFFFu - (cw & 0xFFFFu)) < clu));
            };
            const int a1 = (sp(pw1, cw1) & sm1) | ((int)(__float_as_int(B1.w) > it - 2) & ~sm1);
            const int a2 = (sp(pw2, cw2) & sm2) | ((int)(__float_as_int(B2.w) > it - 2) & ~sm2);
            active = ((a1 | a2) & 1) != 0;
        }
        if (active) {
            const float4 S1 = B1, S2 = B2;
            bool prod = IMP ? isl_impulse_eval(q0, B1, B2, im1, ii1, im2, ii2) : isl_displace_eval(q0, B1, B2, im1, ii1, im2, ii2);
            if (has2) {
                if (HALF) { B1 = body_round<HALF>(B1); B2 = body_round<HALF>(B2); }      // (the ablation rounds on every joint's store)
                if (ws) {
                    B1.x = sm1 ? S1.x : B1.x; B1.y = sm1 ? S1.y : B1.y; B1.z = sm1 ? S1.z : B1.z; B1.w = sm1 ? S1.w : B1.w;
                    B2.x = sm2 ? S2.x : B2.x; B2.y = sm2 ? S2.y : B2.y; B2.z = sm2 ? S2.z : B2.z; B2.w = sm2 ? S2.w : B2.w;
                }
                prod |= IMP ? isl_impulse_eval(q1, B1, B2, im1, ii1, im2, ii2) : isl_displace_eval(q1, B1, B2, im1, ii1, im2, ii2);
            }
            B1.w = prod ? __int_as_float(it) : B1.w; B2.w = prod ? __int_as_float(it) : B2.w;
            if (prod) {
                if (IMP) flag_imp[slot] = 1; else flag_disp[slot] = 1;
                if (ws) {
                    if (st1) atomicMax(&sw[it & 1][l1], static_word(it, c));
                    if (st2) atomicMax(&sw[it & 1][l2], static_word(it, c));
                }
            }
            if (!ws || !st1) body_store(rec, l1, B1);
            if (!ws || !st2) body_store(rec, l2, B2);
        }
    };
    const std::true_type IMPULSES{}; const std::false_type DISPLACEMENT{};
    // The sweeps, in two loops: both halves while the displacement half still runs (the displacement sweeps of a resting scene end
    // after the first: nothing is deeper than the allowed penetration, ref: Solver.cpp:672-680, 210), then the impulses alone — once
    // over, the displacement sweeps stay over (disp_alive only falls, `it` only grows).  (One loop for both cost the impulse-only step
    // a dozen register copies: the displacement accumulators' values flowed through its exits.)
    bool hot_from_here = false;
    for (; it < iters; ++it) {
        const bool imp_on = imp_alive && it < ci, disp_on = disp_alive && it < pi;
        if (!imp_on && !disp_on) break;
        if (imp_on && !disp_on) { hot_from_here = true; break; }
        peek_ctl();
        next_slot();
        for (c = 0; c < ncol; ++c) {
            step_begin();
            if (col == c) {
                if (wave_static) {
                    if (imp_on) half_step(IMPULSES, true);
                    if (disp_on) half_step(DISPLACEMENT, true);
                } else {
                    if (imp_on) half_step(IMPULSES, false);
                    if (disp_on) half_step(DISPLACEMENT, false);
                }
            }
            step_work_done();
            __syncthreads();
            step_end();
        }
        if (imp_on) { done_imp = it + 1; imp_alive = flag_imp[slot] != 0; }        // (behind the last class step's barrier)
        if (disp_on) { done_disp = it + 1; disp_alive = flag_disp[slot] != 0; }
    }
    if (hot_from_here)
        for (; it < ci && imp_alive; ++it) {
            peek_ctl();
            next_slot();
            for (c = 0; c < ncol; ++c) {
                step_begin();
                if (col == c) { if (wave_static) half_step(IMPULSES, true); else half_step(IMPULSES, false); }
                step_work_done();
                __syncthreads();
                step_end();
            }
            done_imp = it + 1; imp_alive = flag_imp[slot] != 0;
        }

    PHX_ISL_STAMP(4);
    // results go straight back into the caller's records (commit-gated like k_finish_*); the refreshed constants
    // never leave the registers
    if (verify) {
        // every workgroup of the launch is resident at once (the host launches ISL_VERIFY only then), so by now — tens of
        // microseconds after its own arrival — all of them have arrived and this wait is one load; it is BOUNDED all the same: if
        // it runs out (a GPU shared with somebody else's kernels) the group stays uncommitted and the host completes it (ISL_COMPLETE)
        if (tid == 0) {
            const unsigned shards = iv.nexpect < (unsigned)ISL_SHARDS ? iv.nexpect : (unsigned)ISL_SHARDS;
            unsigned lo = early_ctl;
            int polls = 0;
            const bool settled = (lo & ISL_ARRIVE_MASK) >= shards || lo >= ISL_BAD;      // (complete, or spoiled: both final)
            for (; !settled && polls < iv.wait_polls; ++polls) {
                lo = (unsigned)__hip_atomic_load(iv.ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((lo & ISL_ARRIVE_MASK) >= shards || lo >= ISL_BAD) break;
                __builtin_amdgcn_s_sleep(32);
            }
            s_commit = lo == shards ? 1 : 0;
            if (!settled && polls == iv.wait_polls) atomicOr(iv.ctl, ISL_TIMEOUT);      // (nobody may take this solve for complete)
        }
        __syncthreads();
        if (!s_commit) { if (iv.stamp_end) solve_stamp_end(v.stamps); return; }
    } else if (iv.mode == ISL_GATED && *v.fingerprint != v.expected_fingerprint) { if (iv.stamp_end) solve_stamp_end(v.stamps); return; }
    if (live) {                                            // FinishJoints (ref: Solver.cpp:543-544)
        phx_contact_joint& out = joints[jid0];
        __builtin_nontemporal_store(q0.accN, &out.normal_accumulated_impulse);
        __builtin_nontemporal_store(q0.accF, &out.friction_accumulated_impulse);
    }
    if (has2) {
        phx_contact_joint& out = joints[jid1];
        __builtin_nontemporal_store(q1.accN, &out.normal_accumulated_impulse);
        __builtin_nontemporal_store(q1.accF, &out.friction_accumulated_impulse);
    }
#pragma unroll
    for (int k = 0; k < BI; ++k) {                         // FinishBodies (ref: Solver.cpp:488-492), dynamic bodies only
        const int i = tid + k * T;
        if (body_id[k] < 0 || is_st[i]) continue;
        const float4 a = body_load(imp, i), e = body_load(disp, i);
        store_nt(&bv.vel[body_id[k]], a.x, a.y, a.z, 0.f);
        store_nt(&bv.dvel[body_id[k]], e.x, e.y, e.z, 0.f);
    }
    if (iv.stamp_end) solve_stamp_end(v.stamps);           // (no HBM group behind this launch: it is the solve's last kernel)
    if (tid == 0) {
        if (iv.mode != ISL_GATED) iv.done[group] = iv.epoch;      // committed
        const int slot = group % ISL_STAT_SLOTS;
        atomicMax(&iv.executed[2 * slot], done_imp);
        atomicMax(&iv.executed[2 * slot + 1], done_disp);
        atomicAdd(&iv.visits[slot], (unsigned long long)done_imp * (unsigned long long)d.y);
    }
    if (TRACE && iv.wave_trace && (tid & 63) == 0) {
        unsigned long long* w = iv.wave_trace + ((size_t)group * (T / 64) + (tid >> 6)) * 8;
        w[0] = tw_work; w[1] = tw_bar; w[2] = tw_idle; w[3] = ((unsigned long long)tw_nwork << 32) | tw_nidle; w[4] = tw_work_big; w[5] = tw_nbig;
        w[6] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));        // HW_REG_HW_ID: wave, SIMD, pipe, CU, SH, SE ... (tools/simd_map.py)
        w[7] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF;  // XCC id
    }
    if (TRACE) {
        __builtin_amdgcn_s_waitcnt(0);         // the stores above have left the wave
        PHX_ISL_STAMP(5);
        if (tid == 0) {
            iv.trace[(size_t)group * 8 + 6] = (unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF)    // HW_REG_XCC_ID[3:0]
                                              | ((__builtin_readcyclecounter() - cycles0) << 4);                                  // + s_memtime ticks start -> end
            iv.trace[(size_t)group * 8 + 7] = ((unsigned long long)ncol << 32) | (unsigned)done_imp;
        }
    }
}

} // namespace phx
