// exchange.h — post-solve exchange of an island-sharded solve (BASELINE config 3).
//
// Every rank holds a replica of the world, builds the same schedule from the same joints and sweeps only the
// groups it OWNS (DeviceSolver::set_shard): the groups are dealt to the ranks longest-processing-time first by their joint
// count (exchange_partition below) — a pure function of the schedule, so every rank computes the same deal.  The reference merges every island's bodies back
// into the one body array after its parallel island loop (ref: Solver.cpp:86-91 parallelFor over islands, then
// FinishBodies :114, 482-494 and FinishJoints :527-547); across GPUs the counterpart is ONE all-gather per step:
// each rank packs what its groups produced — per body the six solved floats {velocity.xy, angularVelocity,
// displacingVelocity.xy, displacingAngularVelocity}, per joint the two accumulated impulses — into a contiguous
// SEGMENT, the segments are all-gathered (RCCL over xGMI by the caller, on the solver's stream), and every rank
// scatters the other ranks' segments into its own replica.  After that the replicas are bit-identical again and
// IntegratePosition (ref: World.cpp:57-70) may run on all bodies.
//
// Segment of rank r (32-bit words; the layout is a pure function of the schedule, so every rank computes all of it):
//   [0..7]   header {magic, serial, status, shard, fingerprint lo, fingerprint hi, segment words, 0}
//   then, for each of its groups g (ascending; the trailing HBM group counts as group lds_groups):
//            6 words per body of the group's body table (static bodies included: never unpacked),
//            2 words per slot of the group, the block padded to a multiple of 4 words.
// All segments are padded to the same length (the longest, rounded up to 64 words) so that one equal-count
// all-gather moves them; `segment_words` is that common length.
#pragma once

#include "common.h"
#include "body_view.h"

namespace phx {

constexpr int XCH_HEADER_WORDS = 8;
constexpr unsigned XCH_MAGIC = 0x45584850u;        // "PHXE"
// status bits accumulated by the unpack check (phx_solver_exchange_status)
constexpr int XCH_ERR_PEER = 1;          // a peer posted a non-zero status word (it failed before the exchange)
constexpr int XCH_ERR_SERIAL = 2;        // a peer is at a different step
constexpr int XCH_ERR_TOPOLOGY = 4;      // a peer solved a different joint topology: the replicas have diverged
constexpr int XCH_ERR_MAGIC = 8;         // the segment was never written (collective did not run / wrong buffer)

inline long long xch_group_words(int bodies, int slots) { return (6ll * bodies + 2ll * slots + 3) & ~3ll; }

// Which rank solves which group: longest processing time first — the groups taken by decreasing joint (slot) count, ties by
// group number, each given to the rank with the least joints so far (ties: the lowest rank).  A group is one latency-bound
// workgroup (or, the trailing HBM group, a sequence of launches) whose time grows with its joints; on uniform columns this
// deals them round-robin.  Pure host function of (slot counts, shard count): every rank computes the same owners.
inline void exchange_partition(const int* group_slots, int ngroups, int shard_count, int* owner)
{
    std::vector<int> by_size((size_t)ngroups);
    for (int g = 0; g < ngroups; ++g) by_size[g] = g;
    std::stable_sort(by_size.begin(), by_size.end(), [&](int a, int b) { return group_slots[a] > group_slots[b]; });
    std::vector<long long> load((size_t)shard_count, 0ll);
    for (int g : by_size) {
        int best = 0;
        for (int r = 1; r < shard_count; ++r) if (load[r] < load[best]) best = r;
        owner[g] = best;
        load[best] += group_slots[g];
    }
}

// The common segment length is padded to 64 KB: in a running world the joint list changes a little in every step, and so would an
// exactly fitted length — and with it the byte count of the all-gather, which the ranks must agree on before they enter it (a host
// wait, world.hip step_sharded).  Padded, the length changes rarely; the pad is <= 6 % of a cfg-2 segment at eight ranks.
constexpr int XCH_SEGMENT_GRANULE_WORDS = 16384;
// Pure host function (unit-tested on CPU through phx_exchange_layout): word offset of every group inside its owner's
// segment (header included; a rank's groups in ascending group order) and the common padded segment length.
inline long long exchange_layout(const int* group_bodies, const int* group_slots, int ngroups, int shard_count, const int* owner, long long* group_offset_words,
                                 long long* rank_words /* shard_count entries, may be null */)
{
    std::vector<long long> used((size_t)shard_count, (long long)XCH_HEADER_WORDS);
    for (int g = 0; g < ngroups; ++g) {
        const int r = owner[g];
        if (group_offset_words) group_offset_words[g] = used[r];
        used[r] += xch_group_words(group_bodies[g], group_slots[g]);
    }
    long long longest = XCH_HEADER_WORDS;
    for (int r = 0; r < shard_count; ++r) { if (rank_words) rank_words[r] = used[r]; longest = std::max(longest, used[r]); }
    return (longest + XCH_SEGMENT_GRANULE_WORDS - 1) & ~(long long)(XCH_SEGMENT_GRANULE_WORDS - 1);
}

struct ExchangeView {
    const int4* desc;          // per LDS group {slot_begin, slot_count, body_begin, body_count}
    const int* group_bodies;   // body tables of the LDS groups
    const int* order;          // slot -> joint
    const long long* xoff;     // per group (HBM group = index lds_groups): word offset inside the owner's segment
    const int* owner;          // per group (HBM group = index lds_groups): the rank that solves it (exchange_partition)
    const int* mine;           // the LDS groups this rank owns, ascending
    int lds_groups, shard, shard_count;
    // the HBM group, if any
    const int* hbm_bodies; int hbm_body_count, hbm_begin, hbm_end;
    long long segment_words;   // common padded segment length = stride between ranks in the gathered buffer
};

// One workgroup per OWNED LDS group (group = mine[blockIdx.x]): solved fields -> send segment.
__global__ void __launch_bounds__(256) k_exchange_pack(ExchangeView x, BodyView bodies, const phx_contact_joint* __restrict__ joints,
                                                       unsigned* __restrict__ send, unsigned serial, unsigned status, unsigned long long fingerprint, int mine_count)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        send[0] = XCH_MAGIC; send[1] = serial; send[2] = status; send[3] = (unsigned)x.shard;
        send[4] = (unsigned)fingerprint; send[5] = (unsigned)(fingerprint >> 32); send[6] = (unsigned)x.segment_words; send[7] = 0u;
    }
    if ((int)blockIdx.x >= mine_count) return;
    const int g = x.mine[blockIdx.x];
    const int4 d = x.desc[g];
    float* out = reinterpret_cast<float*>(send + x.xoff[g]);
    for (int i = threadIdx.x; i < d.w; i += blockDim.x) {
        const int id = x.group_bodies[d.z + i];
        const float4 a = bodies.vel[id], e = bodies.dvel[id];
        float* o = out + 6 * i;
        o[0] = a.x; o[1] = a.y; o[2] = a.z;
        o[3] = e.x; o[4] = e.y; o[5] = e.z;
    }
    float* jo = out + 6 * (size_t)d.w;
    for (int s = threadIdx.x; s < d.y; s += blockDim.x) {
        const phx_contact_joint& j = joints[x.order[d.x + s]];
        jo[2 * s] = j.normal_accumulated_impulse; jo[2 * s + 1] = j.friction_accumulated_impulse;
    }
}

// the HBM group of the owning rank (grid-stride)
__global__ void __launch_bounds__(256) k_exchange_pack_hbm(ExchangeView x, BodyView bodies, const phx_contact_joint* __restrict__ joints,
                                                           unsigned* __restrict__ send)
{
    float* out = reinterpret_cast<float*>(send + x.xoff[x.lds_groups]);
    const int n = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = t; i < x.hbm_body_count; i += n) {
        const int id = x.hbm_bodies[i];
        const float4 a = bodies.vel[id], e = bodies.dvel[id];
        float* o = out + 6 * (size_t)i;
        o[0] = a.x; o[1] = a.y; o[2] = a.z;
        o[3] = e.x; o[4] = e.y; o[5] = e.z;
    }
    float* jo = out + 6 * (size_t)x.hbm_body_count;
    for (int s = t; s < x.hbm_end - x.hbm_begin; s += n) {
        const phx_contact_joint& j = joints[x.order[x.hbm_begin + s]];
        jo[2 * (size_t)s] = j.normal_accumulated_impulse; jo[2 * (size_t)s + 1] = j.friction_accumulated_impulse;
    }
}

// what the header of rank r's segment says about it (0 = consistent with this rank's step)
__device__ __forceinline__ int xch_check_header(const unsigned* __restrict__ h, unsigned serial, unsigned long long fingerprint)
{
    if (h[0] != XCH_MAGIC) return XCH_ERR_MAGIC;
    int e = 0;
    if (h[2] != 0u) e |= XCH_ERR_PEER;
    if (h[1] != serial) e |= XCH_ERR_SERIAL;
    if (h[4] != (unsigned)fingerprint || h[5] != (unsigned)(fingerprint >> 32)) e |= XCH_ERR_TOPOLOGY;
    return e;
}

// One workgroup per LDS group; groups of this rank return at once.  Static bodies are never written (their owner never
// wrote them either, ref: Solver.cpp:562-567 — zero inverse mass leaves the velocity alone).
// Every workgroup checks ITS owner's header before it scatters anything: a segment whose header is inconsistent (never written,
// another step, another topology, a peer that failed) is not unpacked at all — the replica keeps its own unsolved values for those
// groups instead of garbage — and the error word says so.  Workgroup 0 also checks every peer's header (any shard count).
__global__ void __launch_bounds__(256) k_exchange_unpack(ExchangeView x, BodyView bodies, phx_contact_joint* __restrict__ joints,
                                                         const unsigned* __restrict__ recv, unsigned serial, unsigned long long fingerprint, int* __restrict__ error)
{
    if (blockIdx.x == 0) {
        int e = 0;
        for (int r = (int)threadIdx.x; r < x.shard_count; r += (int)blockDim.x) e |= xch_check_header(recv + (size_t)r * x.segment_words, serial, fingerprint);
        if (e) atomicOr(error, e);
    }
    const int g = (int)blockIdx.x;
    if (g >= x.lds_groups) return;
    const int owner = x.owner[g];
    if (owner == x.shard) return;
    if (xch_check_header(recv + (size_t)owner * x.segment_words, serial, fingerprint)) return;      // (workgroup-uniform)
    const int4 d = x.desc[g];
    const float* in = reinterpret_cast<const float*>(recv + (size_t)owner * x.segment_words + x.xoff[g]);
    for (int i = threadIdx.x; i < d.w; i += blockDim.x) {
        const int id = x.group_bodies[d.z + i];
        const float4 p = bodies.mpos[id];
        if (p.x == 0.f && p.y == 0.f) continue;
        const float* o = in + 6 * i;
        bodies.vel[id] = make_float4(o[0], o[1], o[2], 0.f);
        bodies.dvel[id] = make_float4(o[3], o[4], o[5], 0.f);
    }
    const float* ji = in + 6 * (size_t)d.w;
    for (int s = threadIdx.x; s < d.y; s += blockDim.x) {
        phx_contact_joint& j = joints[x.order[d.x + s]];
        j.normal_accumulated_impulse = ji[2 * s]; j.friction_accumulated_impulse = ji[2 * s + 1];
    }
}

__global__ void __launch_bounds__(256) k_exchange_unpack_hbm(ExchangeView x, BodyView bodies, phx_contact_joint* __restrict__ joints,
                                                             const unsigned* __restrict__ recv, unsigned serial, unsigned long long fingerprint)
{
    const int owner = x.owner[x.lds_groups];
    if (xch_check_header(recv + (size_t)owner * x.segment_words, serial, fingerprint)) return;      // (see k_exchange_unpack)
    const float* in = reinterpret_cast<const float*>(recv + (size_t)owner * x.segment_words + x.xoff[x.lds_groups]);
    const int n = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = t; i < x.hbm_body_count; i += n) {
        const int id = x.hbm_bodies[i];
        const float4 p = bodies.mpos[id];
        if (p.x == 0.f && p.y == 0.f) continue;
        const float* o = in + 6 * (size_t)i;
        bodies.vel[id] = make_float4(o[0], o[1], o[2], 0.f);
        bodies.dvel[id] = make_float4(o[3], o[4], o[5], 0.f);
    }
    const float* ji = in + 6 * (size_t)x.hbm_body_count;
    for (int s = t; s < x.hbm_end - x.hbm_begin; s += n) {
        phx_contact_joint& j = joints[x.order[x.hbm_begin + s]];
        j.normal_accumulated_impulse = ji[2 * (size_t)s]; j.friction_accumulated_impulse = ji[2 * (size_t)s + 1];
    }
}

} // namespace phx
