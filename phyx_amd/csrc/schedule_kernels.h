// schedule_kernels.h — building the island-aware schedule on the device.
//
// The schedule is a pure function of the joints' body pairs and of which bodies are static (schedule.h).  The
// host builder (schedule.hip) is the specification; this file produces the SAME schedule — same groups, same
// colours, same slot order (tests compare the two) — without pulling the joint list over PCIe:
//   connected components   one linking pass over the joint list (atomicMin chains) + one flattening pass
//                          (the device form of the union-find of ref: Solver.cpp:275-323)
//   numbering              components numbered by their smallest body index (= body order, ref: Solver.cpp:344-356)
//   binning                greedy over consecutive components — ncomp integers, done on the host
//   joint order            stable radix sort of the joints by bin (device_radix.h)
//   per bin                one workgroup: local body table (static first), first-fit colouring in priority order
//                          (Jones-Plassmann rounds in LDS), stable placement by colour -> slot arrays
#pragma once

#include "common.h"
#include "schedule.h"
#include "device_scan.h"
#include "device_radix.h"

namespace phx {

// ---- connected components over dynamic bodies --------------------------------------------------------------
// (`clear`: a word to zero on the way — saves a memset dispatch)
// The same launch fills the contact point -> first joint table of the unit pairing below (schedule.h): first[id] = the smallest
// joint index carrying contact point id, kept as `tag << 32 | joint` under atomicMin.  The tag COUNTS DOWN from build to build,
// so this build's entries undercut whatever older builds left and the table never has to be reset (a reset was a pass over ncp
// words that had to finish before the first atomicMin: a launch of its own); an entry whose tag is not this build's reads as
// 'nobody'.  The host clears the table when it is new or the tag runs out.
static __global__ void __launch_bounds__(256) k_cc_init(const float4* __restrict__ mpos, int nb, int* __restrict__ parent,
                                                        unsigned char* __restrict__ is_static, int* __restrict__ clear,
                                                        const phx_contact_joint* __restrict__ joints, int nj, int ncp, unsigned long long* __restrict__ first, unsigned tag)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *clear = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const float4 p = mpos[i];                                       // resident {invMass, invInertia, pos} (body_view.h)
        const bool st = p.x == 0.f && p.y == 0.f;                       // ref: Solver.cpp:304
        is_static[i] = st ? 1 : 0;
        parent[i] = st ? -1 : i;
    }
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x) {
        const unsigned id = (unsigned)joints[j].contact_point_index;
        if (id < (unsigned)ncp) atomicMin(&first[id], ((unsigned long long)tag << 32) | (unsigned)j);
    }
}

// (`first` / `partner`: the joints are paired into units on the way, schedule.h; the table is complete, k_cc_init filled it)
__device__ __forceinline__ int partner_of(const phx_contact_joint* __restrict__ joints, int j, const phx_contact_joint& me, int ncp,
                                          const unsigned long long* __restrict__ first, unsigned tag)
{
    const unsigned id = (unsigned)me.contact_point_index;
    if (id >= (unsigned)ncp || first[id] != (((unsigned long long)tag << 32) | (unsigned)j) || (id ^ 1u) >= (unsigned)ncp) return -1;
    const unsigned long long mate = first[id ^ 1u];
    if ((unsigned)(mate >> 32) != tag) return -1;                       // nobody carries that contact point in this build
    const int other = (int)(unsigned)mate;
    const phx_contact_joint o = joints[other];
    return (o.body1 == me.body1 && o.body2 == me.body2) ? other : -1;
}

// Connected components in ONE pass over the joint list: link(hi, lo) = atomicMin(&parent[hi], lo).  If that returns hi, hi was a
// root and hangs under lo now; if it returns some p < hi, hi hung under p already — it now hangs under min(p, lo) and the thread
// goes on to link max(p, lo) under min(p, lo), a strictly smaller pair, so it ends.  parent[x] <= x throughout (no cycles), every
// joint's bodies end in one tree, and a tree's root is its smallest body whatever order the joints ran in — the labels the serial
// union of ref: Solver.cpp:275-323 has after its numbering by smallest body.  Everything a thread learns about other threads'
// work it learns from the values its atomics return (the coherent level), never from a cached load: on eight XCDs a find() that
// walks parent[] by loads either reads stale lines or pays an agent-scope load per hop (measured: 100 us at cfg 2).  Rounds 1-3
// alternated min-label hooking of the two current ROOTS with full compression until nothing hooked any more: two pairs of
// launches for stacks, five for a merged world, plus the 'did it converge' readback.
static __global__ void __launch_bounds__(256) k_cc_link(const phx_contact_joint* __restrict__ joints, int nj, int nb, int* parent,
                                                        const unsigned char* __restrict__ is_static, const unsigned long long* __restrict__ first, unsigned tag, int ncp,
                                                        int* __restrict__ partner, int hops)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x) {
        const phx_contact_joint me = joints[j];
        const int mate = partner_of(joints, j, me, ncp, first, tag);
        partner[j] = mate;
        if (mate >= 0 && (me.contact_point_index & 1)) continue;           // the follower of a unit: its leader links the same two bodies
        const unsigned u = (unsigned)me.body1, v = (unsigned)me.body2;
        if (u >= (unsigned)nb || v >= (unsigned)nb || u == v) continue;     // reported by the fingerprint / validation path
        if (is_static[u] || is_static[v]) continue;                        // a static body joins nothing (ref: Solver.cpp:304)
        int hi = (int)(u > v ? u : v), lo = (int)(u > v ? v : u);
        // `hops` hops up by plain loads first: whatever a load returns — however stale — was stored in parent[] at some point and so
        // ends up in the same tree as its body, which is all a link needs of its two ends.  Linking nearer the roots keeps the trees
        // shallow: in a world merged into one island the flattening pass falls from 41 to 16 us for 11 us more here; on separate
        // stacks the loads only cost (6.6 -> 9.2 us at cfg 2), so the host asks for them when the last schedule had an HBM group.
        // (link + flatten in the settled 200k world by hops: 0: 60 + 40 us, 1: 56 + 34, 2: 57 + 35, 3: 68 + 15, 5: 110 + 13, 8: 113 + 12.)
        // (round 6: the flattening pass is two launches now, ~20 us whatever the trees look like — the host asks for ONE hop where it asked for
        //  three: settled step 1.47 - 1.54 -> 1.43 - 1.48 ms, the loosened world's 1.96 - 2.04 -> 1.91 - 1.96; none: in between.)
        for (int hop = 0; hop < hops; ++hop) {
            const int ph = parent[hi], pl = parent[lo];
            if (ph == hi && pl == lo) break;
            hi = ph; lo = pl;
        }
        if (hi == lo) continue;
        if (hi < lo) { const int t = hi; hi = lo; lo = t; }
        while (true) {
            const int old = atomicMin(parent + hi, lo);
            if (old == hi || old == lo) break;
            hi = old > lo ? old : lo; lo = old > lo ? lo : old;
        }
    }
}

// The components WITHOUT the joints (round 5, the World's step): a manifold that has a contact point has a joint on its two bodies, so the
// manifolds say which bodies hang together as soon as UpdateManifolds is through — while RefreshContactJoints still matches, creates and
// compacts the joints.  DeviceSolver::prelabel_components queues these two kernels, the flattening pass and the roots' scan on the side
// stream; the rebuild then takes the label-keeping path (k_cc_init_lite), whose k_joint_components still spoils the build if some joint's
// bodies carry different labels.  Same edges, same smallest-body roots, same labels as k_cc_init + k_cc_link make from the joints.
static __global__ void __launch_bounds__(256) k_cc_init_bodies(const float4* __restrict__ mpos, int nb, int* __restrict__ parent, unsigned char* __restrict__ is_static, int* __restrict__ clear)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *clear = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const float4 p = mpos[i];
        const bool st = p.x == 0.f && p.y == 0.f;                       // ref: Solver.cpp:304
        is_static[i] = st ? 1 : 0;
        parent[i] = st ? -1 : i;
    }
}

// (manifolds that PackManifolds is moving while this runs are read at their old place, their new one or both: the same link twice; a
//  dead manifold has no contact point wherever it is read)
static __global__ void __launch_bounds__(256) k_cc_link_manifolds(const phx_manifold* __restrict__ manifolds, int nm, int nb, int* parent, const unsigned char* __restrict__ is_static)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nm; i += gridDim.x * blockDim.x) {
        const phx_manifold m = manifolds[i];
        if (m.point_count <= 0) continue;
        const unsigned u = (unsigned)m.body1, v = (unsigned)m.body2;
        if (u >= (unsigned)nb || v >= (unsigned)nb || u == v) continue;
        if (is_static[u] || is_static[v]) continue;                        // a static body joins nothing (ref: Solver.cpp:304)
        int hi = (int)(u > v ? u : v), lo = (int)(u > v ? v : u);
        while (true) {                                                     // (k_cc_link's linking step)
            const int old = atomicMin(parent + hi, lo);
            if (old == hi || old == lo) break;
            hi = old > lo ? old : lo; lo = old > lo ? lo : old;
        }
    }
}

// full path compression: afterwards parent[b] is the representative (the component's smallest body)
// (`clear`: the 'labels disagree' flag k_joint_components may raise, zeroed on the way)
// (Walks that cross read entries their owners are overwriting — old parent or root, an ancestor either way, and mostly the
//  root: that race is what keeps a 200-box chain from costing 200 loads per box.  Jumping pointers in LDS first, a workgroup
//  per 1024 consecutive bodies, measured no faster: 14 against 11 us at cfg 2, 26 against 29 at cfg 4.)
static __global__ void __launch_bounds__(256) k_cc_compress(int* parent, int nb, int* __restrict__ clear)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *clear = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        int p = parent[i];
        if (p < 0) continue;
        while (true) { const int q = parent[p]; if (q == p) break; p = q; }
        parent[i] = p;
    }
}

// The flattening pass in TWO launches (round 6): first every window of CCW_BODIES consecutive bodies jumps its pointers in LDS until each
// body points at its root or at an ancestor BELOW the window (parent[x] <= x: chains only run downwards; a stacked column is a chain as
// deep as it is tall), then k_cc_compress walks what is left — a hop per window the chain spans instead of a hop per body.  The kernel
// boundary is what makes the windows' work visible to each other (eight XCDs, L2s not coherent: inside ONE launch a walk that leaves its
// window meets the other windows' entries as they were — the single-launch form of this measured no faster than the plain walk).
constexpr int CCW_T = 256, CCW_BODIES = 256;      // (1024-body windows, four bodies a lane: 10.9 us at cfg 2 where 256-body windows take less — more workgroups in flight)
static __global__ void __launch_bounds__(CCW_T) k_cc_compress_window(int* parent, int nb)
{
    __shared__ int p[CCW_BODIES];
    __shared__ int changed;
    const int base = (int)blockIdx.x * CCW_BODIES, tid = threadIdx.x;
    for (int j = tid; j < CCW_BODIES; j += CCW_T) p[j] = base + j < nb ? parent[base + j] : -1;
    __syncthreads();
    static_assert(CCW_BODIES <= 1024, "ten doublings");
    for (int round = 0; round < 10; ++round) {               // 2^10 >= the window: a chain through every body of it is flat after that many doublings
        if (tid == 0) changed = 0;
        __syncthreads();
        bool any = false;
        for (int j = tid; j < CCW_BODIES; j += CCW_T) {
            const int q = p[j];
            if (q >= base && q != base + j) {                // my parent is in the window and is not me: take its parent (an ancestor of mine)
                const int r = p[q - base];
                if (r != q) { p[j] = r; any = true; }        // (r == q: q is a root — I am flat)
            }
        }
        if (any) changed = 1;
        __syncthreads();
        if (!changed) break;
        __syncthreads();                                     // (everybody has read the flag before the next round clears it)
    }
    for (int j = tid; j < CCW_BODIES; j += CCW_T) if (base + j < nb) parent[base + j] = p[j];
}

// loader of the 'roots before body i' scan (device_scan.h): 1 for every component root; also zeroes the per-component joint
// and unit counters that k_joint_components fills next (nb + 1 words)
struct RootFlagLoad {
    static constexpr bool in_place = false;
    const int* parent; int nb; unsigned* comp_size; unsigned* comp_units;
    __device__ unsigned operator()(int i) const
    {
        comp_size[i] = 0u; comp_units[i] = 0u;
        return (i < nb && parent[i] == i) ? 1u : 0u;
    }
    __device__ bool load4(int base, uint4& out) const      // (base is a multiple of four: device_scan.h)
    {
        if (base + 3 >= nb || ((reinterpret_cast<uintptr_t>(parent) | reinterpret_cast<uintptr_t>(comp_size) | reinterpret_cast<uintptr_t>(comp_units)) & 15u)) return false;
        *reinterpret_cast<uint4*>(comp_size + base) = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(comp_units + base) = make_uint4(0u, 0u, 0u, 0u);
        const int4 p = *reinterpret_cast<const int4*>(parent + base);
        out = make_uint4(p.x == base ? 1u : 0u, p.y == base + 1 ? 1u : 0u, p.z == base + 2 ? 1u : 0u, p.w == base + 3 ? 1u : 0u);
        return true;
    }
};

// joint -> component number (-1 if both bodies are static), and joints and units (schedule.h) per component.
// The counts are accumulated in a per-workgroup LDS hash table (every lane inserts its own item: LDS atomics on distinct
// slots run in parallel, on one slot they cost a few cycles each) and flushed once per workgroup.  Counting straight into
// memory was fine while every column was its own island, but once a settling scene has merged into one island every
// wave fired at the SAME counter: 1e4 same-address device atomics were 115 us of this 13 us kernel.
constexpr int JC_T = 1024, JC_TABLE = 2048;
// (round 6: the table remembers which slots it handed out — `used`, in the order they were claimed — so that the flush visits those and
//  resets them on the way, instead of every pass over 1024 items clearing and scanning all 2048 slots: at 1M boxes the clearing and
//  scanning WAS the kernel — 55 us for a count whose loads take 15.  `nused` has one counter per pass parity: the flush of pass k
//  resets the one pass k + 1 counts with, between its two barriers, when pass k - 1's readers are all through.)
struct CompCountTable { int key[JC_TABLE]; unsigned cnt[JC_TABLE], units[JC_TABLE]; int used[JC_T]; int nused[2]; };
__device__ __forceinline__ void comp_count_clear(CompCountTable& t)
{
    for (int i = threadIdx.x; i < JC_TABLE; i += JC_T) { t.key[i] = -1; t.cnt[i] = 0; t.units[i] = 0; }
    if (threadIdx.x < 2) t.nused[threadIdx.x] = 0;
    __syncthreads();
}
// every lane adds (1 + extra_joint joints, is_unit units) to component `mine` (< 0: nothing)
// one table insert per distinct component of the wave (in a merged world every lane carries the SAME component: a
// thousand same-address LDS atomics per workgroup made this the slowest kernel of the schedule build)
// (... and a world that has been running for a while keeps its joints in no particular order: a wave's 64 joints then belong
//  to dozens of components and the leader loop — one serial round of LDS atomics per distinct component — was most of this
//  kernel's 18 us at cfg 2.  Three rounds take care of waves with a few components, merged worlds included; whoever is left
//  inserts for himself, all at once: different components, different slots.)
__device__ __forceinline__ void comp_count_add(CompCountTable& t, int pass, int mine, bool extra_joint, bool is_unit)
{
    auto insert = [&](int comp, unsigned jn, unsigned un) {
        unsigned h = ((unsigned)comp * 2654435761u) >> 21;                             // 11 bits
        for (;; h = (h + 1) & (JC_TABLE - 1)) {                                        // <= JC_T distinct keys in a table of 2 * JC_T
            const int seen = atomicCAS(&t.key[h], -1, comp);
            if (seen == -1) t.used[atomicAdd(&t.nused[pass & 1], 1)] = (int)h;         // (claimed: at most one claim per distinct key, <= JC_T of them)
            if (seen == -1 || seen == comp) { atomicAdd(&t.cnt[h], jn); if (un) atomicAdd(&t.units[h], un); break; }
        }
    };
    unsigned long long todo = __ballot(mine >= 0);
    for (int round = 0; round < 3 && todo; ++round) {
        const int leader = __builtin_ctzll(todo);
        const int comp = __shfl(mine, leader);
        const unsigned long long same = __ballot(mine == comp);
        const unsigned sj = (unsigned)__popcll(same) + (unsigned)__popcll(__ballot(mine == comp && extra_joint));
        const unsigned su = (unsigned)__popcll(__ballot(mine == comp && is_unit));
        if ((int)(threadIdx.x & 63) == leader) insert(comp, sj, su);
        todo &= ~same;
    }
    if ((todo >> (threadIdx.x & 63)) & 1ull) insert(mine, extra_joint ? 2u : 1u, is_unit ? 1u : 0u);
}
__device__ __forceinline__ void comp_count_flush(CompCountTable& t, int pass, unsigned* __restrict__ comp_size, unsigned* __restrict__ comp_units)
{
    __syncthreads();
    const int n = t.nused[pass & 1];
    for (int i = threadIdx.x; i < n; i += JC_T) {
        const int h = t.used[i];
        atomicAdd(&comp_size[t.key[h]], t.cnt[h]); if (t.units[h]) atomicAdd(&comp_units[t.key[h]], t.units[h]);
        t.key[h] = -1; t.cnt[h] = 0; t.units[h] = 0;
    }
    if (threadIdx.x == 0) t.nused[(pass & 1) ^ 1] = 0;
    __syncthreads();
}

static __global__ void __launch_bounds__(JC_T) k_joint_components(const phx_contact_joint* __restrict__ joints, int nj, int nb, const int* __restrict__ parent,
                                                                  const unsigned* __restrict__ root_number, const int* __restrict__ partner,
                                                                  int* __restrict__ joint_comp, unsigned* __restrict__ comp_size, unsigned* __restrict__ comp_units,
                                                                  int* __restrict__ unconverged)
{
    // (`unconverged`, may be null: raised if some joint's two dynamic bodies still carry different labels — a caller that skipped
    //  the hook round which only confirms convergence, solver.hip's speculative build, finds out here instead)
    __shared__ CompCountTable table;
    comp_count_clear(table);
    int pass = 0;
    for (int j0 = blockIdx.x * blockDim.x; j0 < nj; j0 += gridDim.x * blockDim.x, ++pass) {       // uniform trip count per workgroup
        const int j = j0 + (int)threadIdx.x;
        int mine = -1;
        unsigned lead_one = 0;
        if (j < nj) {
            const phx_contact_joint me = joints[j];
            const unsigned u = (unsigned)me.body1, v = (unsigned)me.body2;
            const int mate = partner[j];
            const bool leads = !(mate >= 0 && (me.contact_point_index & 1));      // not the follower of a unit
            int comp = -1;
            if (u < (unsigned)nb && v < (unsigned)nb) {
                const int pu = parent[u], pv = parent[v];
                const int r = pu >= 0 ? pu : pv;
                if (r >= 0) comp = (int)root_number[r];
                if (unconverged && pu >= 0 && pv >= 0 && pu != pv) *unconverged = 1;
            }
            joint_comp[j] = comp;
            mine = comp; lead_one = leads ? 1u : 0u;
        }
        comp_count_add(table, pass, mine, false, lead_one != 0);
        comp_count_flush(table, pass, comp_size, comp_units);
    }
}

// The same counts WITHOUT the joints (round 6, the World's step): after RefreshContactJoints every contact point of a live manifold has
// exactly one joint (ref: World.cpp:92-118 creates the missing ones, :125-143 deletes the orphans), so a manifold with n contact points
// is n joints and one unit of its bodies' component — known as soon as UpdateManifolds is through.  The side stream counts here and
// bins (k_bin_components) while the joint list is still being matched, extended and compacted; the rebuild proper is then two
// launches (k_joint_scatter, k_build_bin).  `flags` |= 1: a manifold with contact points between two static bodies, or a body out of
// range — such joints belong to no component (the HBM group's business): the build is spoiled and the caller rebuilds the long way.
static __global__ void __launch_bounds__(JC_T) k_manifold_components(const phx_manifold* __restrict__ manifolds, int nm, int nb, const int* __restrict__ parent,
                                                                     const unsigned* __restrict__ root_number, unsigned* __restrict__ comp_size,
                                                                     unsigned* __restrict__ comp_units, int* __restrict__ flags)
{
    __shared__ CompCountTable table;
    comp_count_clear(table);
    int pass = 0;
    for (int i0 = blockIdx.x * blockDim.x; i0 < nm; i0 += gridDim.x * blockDim.x, ++pass) {
        const int i = i0 + (int)threadIdx.x;
        int mine = -1;
        unsigned points = 0;
        if (i < nm) {
            const phx_manifold m = manifolds[i];
            if (m.point_count > 0) {
                const unsigned u = (unsigned)m.body1, v = (unsigned)m.body2;
                int r = -1;
                if (u < (unsigned)nb && v < (unsigned)nb && m.point_count <= 2) { const int pu = parent[u], pv = parent[v]; r = pu >= 0 ? pu : pv; }
                if (r >= 0) { mine = (int)root_number[r]; points = (unsigned)m.point_count; }
                else atomicOr(flags, 1);
            }
        }
        comp_count_add(table, pass, mine, points == 2u, true);
        comp_count_flush(table, pass, comp_size, comp_units);
    }
}

// flags of the joints that belong to the HBM group (the builder's long way): their component fits no workgroup, or they join two static
// bodies; n + 1 words (the last one 0) for the scan behind it
static __global__ void __launch_bounds__(256) k_rest_flags(const int* __restrict__ joint_comp, const int* __restrict__ bin_of_comp, int nj, int nbins, int ncomp_cap,
                                                           unsigned* __restrict__ flags)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j <= nj; j += gridDim.x * blockDim.x) {
        unsigned f = 0u;
        if (j < nj) { const int c = joint_comp[j]; f = (c < 0 || c >= ncomp_cap || bin_of_comp[c] >= nbins) ? 1u : 0u; }
        flags[j] = f;
    }
}

// ---- binning on the device -------------------------------------------------------------------------------------------
// The binning rule (schedule.h BIN_CHUNK; host: schedule.hip::build_island_schedule and DeviceSolver::build_device) restated for
// ONE workgroup, so that a rebuild needs no host round trip between the connected components and the bins: the host launches
// everything behind it with last build's bin count as the grid, and reads what came out when it settles the solve (solver.hip,
// "speculative binning").  Consecutive components are packed greedily, and a bin never spans a multiple of BIN_CHUNK component
// numbers: greedy packing is a chain (a bin ends where the next component would overflow it), and the chunk boundaries cut it
// into independent pieces of 64 components — a wave packs a chunk, a lane per component, a scan over the chunks numbers the bins
// and places their slots.  (Round 3 followed the unbroken chain by pointer doubling over windows of 8192 components:
// ~65 barrier-separated rounds per window, 16 us for the 1000 columns of cfg 2 and 67 us for the 10000 of cfg 4; the chunked
// rule costs one partly filled bin per 64 components and takes a few microseconds.)
// GatherIslands' published numbers (ref: Solver.cpp:400, 414, 449) are statistics: the host computes them from the component
// sizes when it settles the solve.  Whatever the host would have decided differently poisons the solve's fingerprint word
// (`fail` bits below), which makes every kernel of the solve commit nothing; the host then rebuilds the slow way.
constexpr int BINC_T = BINC_MAX / BIN_CHUNK;      // lanes = chunks (BINC_MAX, BINC_JOINT_BITS: schedule.h)
static_assert(BINC_T == 1024 && BIN_CHUNK == 64, "one lane per chunk of 64 components, one workgroup");
constexpr int BINC_FAIL_CC = 1, BINC_FAIL_COUNT = 2, BINC_FAIL_FIT = 4, BINC_FAIL_SHAPE = 8, BINC_FAIL_REST = 16, BINC_FAIL_GRID = 32;
constexpr unsigned long long BINC_POISON = 0x9E3779B97F4A7C15ull;

struct BinCompView {
    const unsigned* comp_size;        // joints per component (body order)
    const unsigned* comp_units;       // units per component
    const int* cc_small;              // [0] 'the last hook round still hooked something', [1] component count
    int nj;
    int cap_units;                    // lanes of the workgroup shape the launches behind this kernel use
    int small_units;                  // lanes of the small shape (the host picks the roomier one only if some component needs it)
    int max_bins;                     // grid of the launches behind this kernel
    int* bin_of; int* rank_of;        // out: per component (BINC_MAX each)
    int* goff;                        // out: first slot of every bin, max_bins + 1 words
    int* result;                      // out: [0] bins (0 if spoiled), [1] slots in bins, [4] fail bits, [5] components, [6] bins found, [7] = 0 (k_joint_scatter's spoil bits)
    unsigned* cursor;                 // out: max_bins + 1 zeros — the bins' fill counts of k_joint_scatter
    unsigned long long* scratch;      // launches of more than one workgroup: [0, BINC_T) head masks, [BINC_T, 2 BINC_T) bins << 32 | slots per chunk,
                                      // [2 BINC_T] arrival counter, [2 BINC_T + 1] fail bits | needs-big << 8 — both left zero for the next launch
    unsigned long long* fingerprint;  // the solve's topology fingerprint word: saved to `hash_out`, then replaced by `gate`
                                      //   (null — the bins are made on the side stream, from the manifolds' counts: k_joint_scatter arms the gate; nj < 0 then: it checks the total too)
    unsigned long long* hash_out;     //   (the host does not know the hash yet: the solve's kernels compare the word with a constant
    unsigned long long gate;          //    it does know) — or by a spoiled `gate` if the build cannot be used
};

// (a wave's own LDS operations are ordered; the compiler only has to be told that other lanes' stores count)
#define PHX_WAVE_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

// One workgroup packs 16 chunks at a time (a wave each, ~2000 cycles a chunk); the 157 chunks of the 1M-box scene were ten such
// rounds, 31 us on one CU.  A launch of several workgroups deals the chunks out, every workgroup leaves its chunks' results in
// `scratch`, and the LAST one to arrive (a counter) does what is left — the scan over the chunks and the tables — alone: that part
// is a few hundred cycles a chunk.  What the workgroups tell each other goes through agent-scope stores and loads (different XCDs).
static __global__ void __launch_bounds__(BINC_T) k_bin_components(BinCompView v)
{
    // a WAVE packs a chunk, a lane per component: prefix sums of joints and units by shuffles, then the chain of bin heads — the next
    // head is the first lane whose prefix, counted from the current head, overflows the shape (a ballot), one step per bin — which
    // leaves the chunk's heads as a 64-bit mask; everything else (a component's bin, its rank in it, the bin's first slot) is
    // bit counting on that mask.  The masks wait in LDS for the scan over the chunks.
    __shared__ unsigned long long head_mask[BINC_T];
    __shared__ unsigned chunk_bins[BINC_T], chunk_slots[BINC_T];      // per chunk, then exclusive over the chunks
    __shared__ unsigned long long wave_sum[BINC_T / 64];
    __shared__ unsigned pre_s[BINC_T / 64][64], pre_u[BINC_T / 64][64];      // a wave's chunk: inclusive prefixes of joints / units
    __shared__ unsigned char jump[BINC_T / 64][64], reach[BINC_T / 64][64];
    __shared__ int s_fail, s_needs_big, s_last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int groups = (int)gridDim.x, waves_all = groups * (BINC_T / 64);
    // (a handful of workgroups on the step's critical path, beside a wide kernel of the other stream — the joint match — that fills the
    //  chip: their waves ask for the instruction arbiter's top priority; alone 22 us at 1M boxes, beside the match 79 without it)
    __builtin_amdgcn_s_setprio(3);
    const int n_all = v.cc_small[1];
    const int n_total = n_all < BINC_MAX ? n_all : BINC_MAX;
    const int nchunks = (n_total + BIN_CHUNK - 1) / BIN_CHUNK;
    if (tid == 0) { s_fail = (v.cc_small[0] ? BINC_FAIL_CC : 0) | (n_all > BINC_MAX ? BINC_FAIL_COUNT : 0); s_needs_big = 0; }
    chunk_bins[tid] = 0u; chunk_slots[tid] = 0u; head_mask[tid] = 0ull;
    __syncthreads();
    const unsigned cap_s = 2u * (unsigned)v.cap_units, cap_u = (unsigned)v.cap_units;
    const unsigned long long below = (1ull << lane) - 1ull;
    bool misfit = false, wants_big = false;
    for (int ch = (int)blockIdx.x * (BINC_T / 64) + wave; ch < nchunks; ch += waves_all) {      // (wave-uniform)
        const int c = ch * BIN_CHUNK + lane;
        const unsigned n = c < n_total ? v.comp_size[c] : 0u, u = n ? v.comp_units[c] : 0u;
        if (n && (n > cap_s || u > cap_u)) misfit = true;
        if (n && (n > 2u * (unsigned)v.small_units || u > (unsigned)v.small_units)) wants_big = true;
        unsigned ps = n, pu = u;
        for (int off = 1; off < 64; off <<= 1) { const unsigned a = __shfl_up(ps, off), b = __shfl_up(pu, off); if (lane >= off) { ps += a; pu += b; } }
        const unsigned long long nonempty = __ballot(n != 0u);
        // Where a bin opened at lane l ends — the first lane whose prefix, counted from l, overflows the shape — is a binary search per
        // lane over the chunk's 64 prefixes (the conditions are monotone), and the heads are the lanes reachable from the first
        // non-empty one along those links: six doubling rounds.  ~2000 cycles a chunk however many bins it holds (walking the chain
        // head by head, one ballot and two broadcasts a step, was 300 cycles per BIN: 19 us for the 1000 one-column bins of cfg 2).
        unsigned* wps = &pre_s[wave][0]; unsigned* wpu = &pre_u[wave][0]; unsigned char* wj = &jump[wave][0]; unsigned char* wr = &reach[wave][0];
        wps[lane] = ps; wpu[lane] = pu;
        PHX_WAVE_FENCE();
        const unsigned lim_s = ps - n + cap_s, lim_u = pu - u + cap_u;      // a bin opened here holds lanes whose prefix stays <= these
        int lo = 0;                                                          // first lane with ps > lim_s or pu > lim_u (64: none)
        for (int step = 32; step > 0; step >>= 1) { const int probe = lo + step - 1; if (wps[probe] <= lim_s && wpu[probe] <= lim_u) lo += step; }
        if (lo == 63 && wps[63] <= lim_s && wpu[63] <= lim_u) lo = 64;
        int nxt = lo > lane ? lo : lane + 1;                                 // (a misfit overflows alone: keep the chain moving)
        wj[lane] = (unsigned char)nxt; wr[lane] = 0;
        PHX_WAVE_FENCE();
        if (nonempty && lane == __builtin_ctzll(nonempty)) wr[lane] = 1;
        PHX_WAVE_FENCE();
        for (int round = 0; round < 6; ++round) {
            const int j = wj[lane];
            const bool mark = wr[lane] && j < 64;
            const int jj = j < 64 ? wj[j] : 64;
            PHX_WAVE_FENCE();
            if (mark) wr[j] = 1;
            wj[lane] = (unsigned char)jj;
            PHX_WAVE_FENCE();
        }
        const unsigned long long heads = __ballot(wr[lane] != 0);
        PHX_WAVE_FENCE();
        if (lane == 0) { head_mask[ch] = heads; chunk_bins[ch] = (unsigned)__popcll(heads); }
        if (lane == 63) chunk_slots[ch] = ps;
    }
    if (misfit) atomicOr(&s_fail, BINC_FAIL_FIT);
    if (wants_big) s_needs_big = 1;
    __syncthreads();
    if (groups > 1) {
        unsigned long long* masks = v.scratch, * counts = v.scratch + BINC_T, * arrived = v.scratch + 2 * BINC_T, * flags = arrived + 1;
        for (int ch = (int)blockIdx.x * (BINC_T / 64) + wave; ch < nchunks; ch += waves_all)
            if (lane == 0) {
                __hip_atomic_store(&masks[ch], head_mask[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&counts[ch], ((unsigned long long)chunk_bins[ch] << 32) | chunk_slots[ch], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        if (tid == 0 && (s_fail || s_needs_big)) atomicOr(flags, (unsigned long long)(unsigned)s_fail | (s_needs_big ? 256ull : 0ull));
        __threadfence();
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(arrived, 1ull) == (unsigned long long)(groups - 1) ? 1 : 0;
        __syncthreads();
        if (!s_last) return;
        __threadfence();
        if (tid < nchunks) {
            head_mask[tid] = __hip_atomic_load(&masks[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long w = __hip_atomic_load(&counts[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            chunk_bins[tid] = (unsigned)(w >> 32); chunk_slots[tid] = (unsigned)w;
        }
        if (tid == 0) {
            const unsigned long long f = __hip_atomic_load(flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_fail |= (int)(f & 255ull); if (f & 256ull) s_needs_big = 1;
            __hip_atomic_store(flags, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(arrived, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    // exclusive scan over the chunks (one per lane of the workgroup): bins << 32 | slots
    const unsigned long long mine = ((unsigned long long)chunk_bins[tid] << 32) | chunk_slots[tid];
    unsigned long long incl = mine;
    for (int off = 1; off < 64; off <<= 1) { const unsigned long long y = __shfl_up(incl, off); if (lane >= off) incl += y; }
    if (lane == 63) wave_sum[wave] = incl;
    __syncthreads();
    unsigned long long before = incl - mine, total = 0;
    for (int w = 0; w < BINC_T / 64; ++w) { const unsigned long long t = wave_sum[w]; if (w < wave) before += t; total += t; }
    chunk_bins[tid] = (unsigned)(before >> 32); chunk_slots[tid] = (unsigned)before;
    __syncthreads();
    for (int ch = wave; ch < nchunks; ch += BINC_T / 64) {
        const int c = ch * BIN_CHUNK + lane;
        const unsigned n = c < n_total ? v.comp_size[c] : 0u;
        unsigned ps = n;
        for (int off = 1; off < 64; off <<= 1) { const unsigned a = __shfl_up(ps, off); if (lane >= off) ps += a; }
        const unsigned long long nonempty = __ballot(n != 0u), heads = head_mask[ch];
        const unsigned long long upto = below | (1ull << lane);
        const int local_bin = __popcll(heads & upto) - 1;          // (-1: an empty component in front of the chunk's first head)
        if (c < n_total) {
            const int bin = (int)chunk_bins[ch] + (local_bin < 0 ? 0 : local_bin);
            int rank = 0;
            if (n && local_bin >= 0) {
                const int h = 63 - __builtin_clzll(heads & upto);     // my bin's head
                rank = __popcll(nonempty & upto & ~((1ull << h) - 1ull)) - 1;
            }
            v.bin_of[c] = bin; v.rank_of[c] = rank;
            if ((heads >> lane) & 1ull) { if (bin <= v.max_bins) v.goff[bin] = (int)(chunk_slots[ch] + ps - n); }
        }
    }
    __syncthreads();
    if (tid == 0) {
        // the host takes the roomier shape iff some component needs it; joints outside every component (both bodies static) and
        // components that fit no shape go to the HBM group, which this path does not build
        const int nbins = (int)(total >> 32), slots_all = (int)(unsigned)total;
        if ((s_needs_big != 0) != (v.cap_units > v.small_units)) s_fail |= BINC_FAIL_SHAPE;
        if (v.nj >= 0 && slots_all != v.nj) s_fail |= BINC_FAIL_REST;
        if (nbins <= v.max_bins) v.goff[nbins] = slots_all;
        if (nbins > v.max_bins) s_fail |= BINC_FAIL_GRID;
        v.result[0] = s_fail ? 0 : nbins;                      // (a spoiled build's tables may be incomplete: nobody runs on them)
        v.result[1] = slots_all; v.result[4] = s_fail; v.result[5] = n_all; v.result[6] = nbins; v.result[7] = 0;
        if (v.fingerprint) {
            *v.hash_out = *v.fingerprint;
            *v.fingerprint = s_fail ? v.gate + BINC_POISON : v.gate;
        }
    }
    for (int i = tid; i <= v.max_bins; i += BINC_T) v.cursor[i] = 0u;
}

// ---- the units, bin by bin -----------------------------------------------------------------------------------------------
// Rounds 2-5 grouped the JOINTS by bin with a stable radix sort of (bin, joint) — keys + histogram, scan, scatter: 29 us in three
// launches at cfg 2 — and built a bin with a lane per joint.  A bin's builder does not need the joints of OTHER bins in any order; it
// needs its own UNITS (schedule.h: the one or two joints of a body pair), and it can put its few hundred in joint order itself
// (k_build_bin: a bitonic network over the leaders' joint indices).  So the units are DEALT to their bins through a fill counter per
// bin (the bins' first slots are known: k_bin_components) — whatever order the atomics return, nothing depends on it once the
// builder has sorted:
//   * from the joints (k_joint_scatter): every leader leaves a RECORD of everything k_build_bin wants of the unit — which the dealing
//     lane holds in registers anyway — instead of an index the builder would have to chase through three levels of memory;
//   * in the World, from the MANIFOLDS, on the side stream (k_manifold_slots): a manifold with contact points IS a unit, its joints
//     are its contact points' (ContactPoint::solverIndex, ref: World.cpp:100, 139), and k_build_bin reads them through the manifold —
//     the rebuild proper is then ONE launch.
// Whatever cannot be dealt (a body or contact point out of range, labels that disagree, a component outside the tables, a bin that is
// full) raises a bit of a flag word; k_build_bin then spoils the solve's control word and the caller rebuilds the long way.
struct BinRecord { int4 a; int2 b; };      // a = {leader joint, follower joint or -1, body1, body2}, b = {leader's contact point, rank of the component in its bin | body1 static << 30 | body2 static << 31}
constexpr int SCAT_FAIL_RANGE = 1, SCAT_FAIL_LABELS = 2, SCAT_FAIL_COMP = 4, SCAT_FAIL_FULL = 8, SCAT_FAIL_UNIT = 16, SCAT_FAIL_TOTAL = 32;

// one fill-counter atomic per distinct bin of the wave (three rounds; whoever is left goes alone — k_joint_components' scheme);
// returns the lane's position in its bin (bin < 0: no position)
__device__ __forceinline__ unsigned deal_position(int bin, unsigned* __restrict__ cursor)
{
    const int lane = threadIdx.x & 63;
    unsigned pos = 0;
    unsigned long long todo = __ballot(bin >= 0);
    for (int round = 0; round < 3 && todo; ++round) {
        const int leader = __builtin_ctzll(todo);
        const int lb = __shfl(bin, leader);
        const unsigned long long same = __ballot(bin == lb);
        unsigned base = 0;
        if (lane == leader) base = atomicAdd(&cursor[lb], (unsigned)__popcll(same));
        base = __shfl(base, leader);
        if (bin == lb) pos = base + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) pos = atomicAdd(&cursor[bin], 1u);
    return pos;
}

struct ScatterView {
    const phx_contact_joint* joints; int nj, nb;
    const int* parent;                // body -> root body of its component (-1: static)
    const int* joint_comp;            // joint -> component number (k_joint_components)
    const int* partner;               // joint -> the other joint of its unit, or -1 (k_cc_link)
    const int* bin_of; const int* rank_of; const int* goff;      // the bins' tables (k_bin_components, or the host's)
    const int* result;                // (may be null: `nbins` counts) k_bin_components' results: [4] fail bits, [6] bins
    int nbins, max_bins, ncomp_cap;
    unsigned* cursor;                 // per bin: units dealt so far (zero on entry)
    int4* rec_a; int2* rec_b;         // out: the units' records, bin by bin (a bin's records from its first slot on)
    int* rejected;                    // the 'a bin was rejected' flag k_build_bin may raise: cleared here
    int* spoil;                       // |= SCAT_FAIL_*
};

static __global__ void __launch_bounds__(256) k_joint_scatter(ScatterView v)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *v.rejected = 0;
    if (v.result && v.result[4]) return;                      // (uniform: the tables may be incomplete — nobody runs on this build)
    const int nbins = v.result ? v.result[6] : v.nbins;
    const int lane = threadIdx.x & 63;
    int bad = 0;
    for (int j0 = blockIdx.x * blockDim.x; j0 < v.nj; j0 += gridDim.x * blockDim.x) {      // (wave-uniform trip count)
        const int j = j0 + (int)threadIdx.x;
        int bin = -1, mate = -1, rank = 0;
        phx_contact_joint me{};
        bool st1 = false, st2 = false;
        if (j < v.nj) {
            me = v.joints[j];
            mate = v.partner[j];
            const int comp = v.joint_comp[j];
            const unsigned a = (unsigned)me.body1, b = (unsigned)me.body2;
            const bool follower = mate >= 0 && (me.contact_point_index & 1);
            if (a >= (unsigned)v.nb || b >= (unsigned)v.nb || a == b) bad |= SCAT_FAIL_RANGE;
            else if (!follower && comp >= 0) {                // (a joint between static bodies belongs to no bin: the HBM group's)
                st1 = v.parent[a] < 0; st2 = v.parent[b] < 0;
                if (comp >= v.ncomp_cap) bad |= SCAT_FAIL_COMP;
                else {
                    bin = v.bin_of[comp]; rank = v.rank_of[comp];
                    if (bin > v.max_bins) { bad |= SCAT_FAIL_COMP; bin = -1; }
                    else if ((unsigned)bin >= (unsigned)nbins) bin = -1;      // (a component too big for a workgroup: the HBM group's)
                }
            }
        }
        const unsigned pos = deal_position(bin, v.cursor);
        if (bin >= 0) {
            const int first = v.goff[bin], room = v.goff[bin + 1] - first;
            if (pos >= (unsigned)room || (unsigned)(first + (int)pos) >= (unsigned)v.nj) bad |= SCAT_FAIL_FULL;
            else {
                const int at = first + (int)pos;
                v.rec_a[at] = make_int4(j, mate, me.body1, me.body2);
                v.rec_b[at] = make_int2(me.contact_point_index, rank | (st1 ? 1 << 30 : 0) | (st2 ? (int)(1u << 31) : 0));
            }
        }
    }
    if (__any(bad != 0)) {
        int all = bad;
        for (int off = 32; off > 0; off >>= 1) all |= __shfl_xor(all, off);
        if (lane == 0) atomicOr(v.spoil, all);
    }
}

// The World's units, dealt on the side stream: manifold m with contact points is the unit of contact points 2m (and 2m + 1); its slot
// holds m and what the builder wants of its component and bodies.  (`flags` |= 1: cannot be dealt — k_build_bin spoils the build.)
struct ManifoldSlotsView {
    const phx_manifold* manifolds; int nm, nb;
    const int* parent; const unsigned* root_number;
    const int* bin_of; const int* rank_of; const int* goff; const int* result; int max_bins;
    unsigned* cursor;
    int2* unit_m;                     // out, per slot: {manifold, rank of the component in its bin | body1 static << 30 | body2 static << 31}
    int* flags;
};

static __global__ void __launch_bounds__(256) k_manifold_slots(ManifoldSlotsView v)
{
    if (v.result[4]) return;
    const int nbins = v.result[6];
    bool bad = false;
    for (int i0 = blockIdx.x * blockDim.x; i0 < v.nm; i0 += gridDim.x * blockDim.x) {
        const int i = i0 + (int)threadIdx.x;
        int bin = -1, info = 0;
        if (i < v.nm) {
            const phx_manifold m = v.manifolds[i];
            if (m.point_count > 0) {
                const unsigned a = (unsigned)m.body1, b = (unsigned)m.body2;
                if (a >= (unsigned)v.nb || b >= (unsigned)v.nb || a == b || m.point_count > 2) bad = true;
                else {
                    const int pa = v.parent[a], pb = v.parent[b];
                    const int r = pa >= 0 ? pa : pb;
                    if (r < 0 || (pa >= 0 && pb >= 0 && pa != pb)) bad = true;
                    else {
                        const unsigned comp = v.root_number[r];
                        if (comp >= (unsigned)BINC_MAX) bad = true;
                        else {
                            bin = v.bin_of[comp];
                            info = v.rank_of[comp] | (pa < 0 ? 1 << 30 : 0) | (pb < 0 ? (int)(1u << 31) : 0);
                            if ((unsigned)bin >= (unsigned)nbins || bin > v.max_bins) { bad = true; bin = -1; }
                        }
                    }
                }
            }
        }
        const unsigned pos = deal_position(bin, v.cursor);
        if (bin >= 0) {
            const int first = v.goff[bin], room = v.goff[bin + 1] - first;
            if (pos >= (unsigned)room) bad = true;
            else v.unit_m[first + (int)pos] = make_int2(i, info);
        }
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(v.flags, 1);
}

// ---- one workgroup builds one bin ------------------------------------------------------------------------------
struct BinBuildView {
    // the bin's units, in any order: records (k_joint_scatter) ...
    const int4* rec_a; const int2* rec_b;
    // ... or manifolds (k_manifold_slots; rec_a null): the unit's joints are its contact points' solver_index
    const int2* unit_m; const phx_manifold* manifolds; const phx_contact_point* cps; const phx_contact_joint* joints; int nj;
    const int* group_offsets;         // slots of bin g = [group_offsets[g], group_offsets[g+1]) — its units' records / manifolds from the first slot on
    const unsigned* cursor;           // units dealt to each bin
    const int* spoil;                 // (may be null) the dealers' fail bits: nobody builds on a spoiled deal
    const int* side_flags;            // (manifolds) the side stream's flags
    const int* result;                // (manifolds) k_bin_components' results: [1] the joints the manifolds add up to, [4] its fail bits
    unsigned long long gate;          // (manifolds) what arms the solve's control word: added once, by this launch (k_bin_components could not, from the side stream)
    int nb, max_static;
    int* order;                       // out: slot -> joint
    unsigned* slot_local;             // out: local body1 | local body2 << 16
    unsigned char* slot_colour;       // out: class
    int4* desc;                       // out: {slot_begin, slot_count, body_begin, body_count}
    int* ncol;                        // out: classes
    int* units;                       // out: units | static bodies of the bin's table << 16
    int4* unit_recs;                  // out: two words per LANE at [2 * (g * T + lane)] (island_view.h; schedule.h LANES): {leader joint or -1: nobody's lane, follower joint or -1,
                                      //      leader's contact point, follower's}, {local body1 | local body2 << 16, class, leader slot, follower slot or -1}
    int* bodies;                      // out: body table of bin g at [g * NB, g * NB + body_count)
    const int* nbins_dev;             // (may be null) the bin count, if the launch grid is only an upper bound of it (speculative binning, solver.hip)
    int* rejected;                    // out: set to 1 if any bin exceeds the caps (caller falls back to the host builder)
    unsigned long long* poison;       // the solve's topology fingerprint word: a rejected bin spoils it, so that every kernel that would commit
                                      // results on this schedule refuses to (the host checks `rejected` only after the solve is queued)
};

// stable rank of the lanes with `want` among the lanes of the whole workgroup that share their key (< 64), in lane order:
// returns the rank inside the wave and leaves in wave_count[w * 64 + key] the number of such lanes in wave w
template <int LANES>
__device__ __forceinline__ int bin_wave_rank(bool want, int key, unsigned short* wave_count)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rank = 0;
    unsigned long long todo = __ballot(want);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int k = __shfl(key, leader);
        const unsigned long long same = __ballot(want && key == k);
        if (want && key == k) rank = __popcll(same & ((1ull << lane) - 1ull));
        if (lane == leader) wave_count[wave * 64 + k] = (unsigned short)__popcll(same);
        todo &= ~same;
    }
    return rank;
}

// T = unit capacity = lanes of the island kernel = lanes of this builder: a lane per UNIT (round 5 ran a lane per joint: twice the
// waves, and with four bins on a CU the builder is bound by the instructions its waves issue).
template <int T, int NB>
static __global__ void __launch_bounds__(T) k_build_bin(BinBuildView v)
{
    constexpr int LANES = T;
    constexpr int HT = 8 * LANES;                        // open-addressing table, <= 2 LANES distinct bodies
    __shared__ __align__(16) int pool[2 * HT];            // the units' sort and staging, then the hash table
    int* ht_key = pool;                                  // later reused as the per-body priority table of the colouring
    static_assert((size_t)NB * 8 <= (size_t)HT * 4, "priority table must fit the hash table");
    int* ht_val = pool + HT;                             // first occurrence position, later the local index
    static_assert(6 * LANES <= 2 * HT, "record staging must fit the hash table");
    __shared__ unsigned long long used[NB];              // candidate A (smallest free colour): colours taken per local body
    __shared__ unsigned long long used_b[NB];            // candidate B (two-ended, schedule.h)
    __shared__ int degree[NB];                           // units of the bin on each local body
    __shared__ unsigned long long seen_a[T], seen_b[T];  // per component of the bin (at most one per unit): classes in use under either candidate
    __shared__ unsigned char bad_b[T];
    __shared__ unsigned scan_lds[LANES / 64];
    __shared__ unsigned with_n[64], single_n[64];        // per class: leaders that have a follower / single leaders
    __shared__ unsigned class_begin[64], unit_begin[64]; // per class: first slot (relative), first unit
    __shared__ unsigned cls_span[64];                    // per class: its first lane in the island kernel | its units << 16 (schedule.h LANES)
    __shared__ unsigned short wave_with[(LANES / 64) * 64], wave_single[(LANES / 64) * 64];
    __shared__ int n_static, n_bodies, n_col, bad;

    const int g = blockIdx.x, tid = threadIdx.x;
    const bool from_manifolds = v.rec_a == nullptr;
    auto spoil_build = [&]() { if (tid == 0) { *v.rejected = 1; atomicAdd(v.poison, 0x9E3779B97F4A7C15ull); } };
    if (from_manifolds && g == 0 && tid == 0) {
        // the gate of the solve queued behind this build: ADDED to the control word (zero: no hash pass runs in front of a World rebuild),
        // so that it commutes with whatever other workgroups add to spoil it
        atomicAdd(v.poison, v.gate);
        if (v.result[4] || *v.side_flags || v.result[1] != v.nj) { *v.rejected = 1; atomicAdd(v.poison, 0x9E3779B97F4A7C15ull); }
    }
    if (v.spoil && *v.spoil) { if (g == 0) spoil_build(); return; }
    if (from_manifolds && (v.result[4] || *v.side_flags || v.result[1] != v.nj)) return;      // (uniform; workgroup 0 has spoiled the word)
    if (v.nbins_dev && g >= *v.nbins_dev) return;
    const int begin = v.group_offsets[g], count = v.group_offsets[g + 1] - begin;      // the bin's joints
    const int nu = (int)v.cursor[g];                                                  // ... and units
    if (count < 0 || count > 2 * LANES || nu < 0 || nu > LANES || nu > count) { spoil_build(); return; }      // the deal does not match the bins: nobody runs on this build
    for (int i = tid; i < NB; i += LANES) { used[i] = 0ull; used_b[i] = 0ull; degree[i] = 0; }
    seen_a[tid] = 0ull; seen_b[tid] = 0ull; bad_b[tid] = 0;
    for (int i = tid; i < (LANES / 64) * 64; i += LANES) { wave_with[i] = 0; wave_single[i] = 0; }
    if (tid == 0) { bad = 0; n_col = 0; }

    // the bin's units, put in the order of their leaders' joint indices: a unit per lane, a bitonic network over (leader joint, lane) —
    // strides below 64 by shuffles inside the wave, the few above through LDS — then every lane fetches the unit of the lane that held the
    // tid-th smallest leader
    int j = 0, b[2] = {0, 0}, hs[2] = {0, 0}, mate = -1, cpi = 0, comp = 0;
    bool stat[2] = {false, false}, mismatch = false;
    {
        int4 ra = make_int4(0, -1, 0, 0); int2 rb = make_int2(0, 0);
        if (tid < nu) {
            if (!from_manifolds) { ra = v.rec_a[begin + tid]; rb = v.rec_b[begin + tid]; }
            else {
                const int2 um = v.unit_m[begin + tid];
                const phx_manifold m = v.manifolds[um.x];
                const int j0 = v.cps[2 * um.x].solver_index, j1 = m.point_count == 2 ? v.cps[2 * um.x + 1].solver_index : -1;
                ra = make_int4(j0, j1, m.body1, m.body2); rb = make_int2(2 * um.x, um.y);
                // (the joints must name their contact points back — checked once the colouring is under way: the loads' latency hides there)
                if ((unsigned)j0 >= (unsigned)v.nj || (j1 >= 0 && (unsigned)j1 >= (unsigned)v.nj)) { mismatch = true; ra.x = 0; ra.y = -1; }
            }
        }
        int src = tid;
        unsigned* skey = reinterpret_cast<unsigned*>(pool); int* ssrc = pool + LANES;
        __syncthreads();
        constexpr int LANE_BITS = LANES > 256 ? 9 : 8;
        static_assert(LANES <= (1 << LANE_BITS), "lane index must fit the packed sort key");
        if (v.nj < (1 << (31 - LANE_BITS))) {
            // joint indices below 2^(31 - LANE_BITS): (joint, lane) is ONE 32-bit key — a shuffle, a min and a max per step of the network
            unsigned kv = tid < nu ? ((unsigned)ra.x << LANE_BITS) | (unsigned)tid : 0x80000000u | (unsigned)tid;      // (the lanes beyond the bin's units sort behind them)
#pragma unroll
            for (int k = 2; k <= LANES; k <<= 1) {
#pragma unroll
                for (int st = k >> 1; st > 0; st >>= 1) {
                    unsigned other;
                    if (st >= 64) {
                        skey[tid] = kv;
                        __syncthreads();
                        other = skey[tid ^ st];
                        __syncthreads();
                    } else other = (unsigned)__shfl_xor((int)kv, st);
                    const bool take_min = ((tid & st) == 0) == ((tid & k) == 0);
                    const unsigned lo = kv < other ? kv : other, hi = kv < other ? other : kv;
                    kv = take_min ? lo : hi;
                }
            }
            src = (int)(kv & (unsigned)((1 << LANE_BITS) - 1));
        } else {
            unsigned key = tid < nu ? (unsigned)ra.x : 0x80000000u + (unsigned)tid;
#pragma unroll
            for (int k = 2; k <= LANES; k <<= 1) {
#pragma unroll
                for (int st = k >> 1; st > 0; st >>= 1) {
                    unsigned okey; int osrc;
                    if (st >= 64) {
                        skey[tid] = key; ssrc[tid] = src;
                        __syncthreads();
                        okey = skey[tid ^ st]; osrc = ssrc[tid ^ st];
                        __syncthreads();
                    } else { okey = (unsigned)__shfl_xor((int)key, st); osrc = __shfl_xor(src, st); }
                    const bool take_min = ((tid & st) == 0) == ((tid & k) == 0);
                    if (take_min ? okey < key : okey > key) { key = okey; src = osrc; }
                }
            }
        }
        // (staging: six words per lane, in the words the hash table takes next)
        int* stage = pool;
        stage[tid] = ra.x; stage[LANES + tid] = ra.y; stage[2 * LANES + tid] = ra.z; stage[3 * LANES + tid] = ra.w; stage[4 * LANES + tid] = rb.x; stage[5 * LANES + tid] = rb.y;
        __syncthreads();
        j = stage[src]; mate = stage[LANES + src]; b[0] = stage[2 * LANES + src]; b[1] = stage[3 * LANES + src]; cpi = stage[4 * LANES + src];
        const int info = stage[5 * LANES + src];
        comp = info & 0x3FFFFFFF; stat[0] = (info >> 30) & 1; stat[1] = (info >> 31) & 1;
        __syncthreads();
    }
    const bool live = tid < nu;
    // (manifolds) the unit's joints as the joint list has them: asked for here, looked at after the colouring
    phx_contact_joint q0{}, q1{};
    if (from_manifolds && live && !mismatch) { q0 = v.joints[j]; if (mate >= 0) q1 = v.joints[mate]; }
    for (int i = tid; i < HT / 4; i += LANES) {          // (16 bytes per store)
        reinterpret_cast<int4*>(ht_key)[i] = make_int4(-1, -1, -1, -1);
        reinterpret_cast<int4*>(ht_val)[i] = make_int4(0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff);
    }
    __syncthreads();
    if (live) {
        for (int s = 0; s < 2; ++s) {                   // insert, keep the earliest occurrence position 2*tid+s
            unsigned p = ((unsigned)b[s] * 2654435761u) & (HT - 1);
            for (;;) {
                const int k = atomicCAS(&ht_key[p], -1, b[s]);
                if (k == -1 || k == b[s]) break;
                p = (p + 1) & (HT - 1);
            }
            hs[s] = (int)p;
            atomicMin(&ht_val[p], 2 * tid + s);
        }
    }
    __syncthreads();
    // table leaders = first occurrences; local index = rank among static ones, or n_static + rank among dynamic ones
    bool lead[2] = {false, false};
    unsigned mine = 0;                                   // static first occurrences << 16 | dynamic ones
    if (live)
        for (int s = 0; s < 2; ++s) {
            lead[s] = ht_val[hs[s]] == 2 * tid + s;
            if (lead[s]) mine += stat[s] ? 0x10000u : 1u;
        }
    // block exclusive scan of `mine`; the joints are counted on the side
    unsigned x = mine;
    const int lane = tid & 63, wave = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) scan_lds[wave] = x;
    const int joints_here = nu + __syncthreads_count(live && mate >= 0);
    unsigned before = x - mine, total = 0;
    for (int w = 0; w < LANES / 64; ++w) { const unsigned t = scan_lds[w]; if (w < wave) before += t; total += t; }
    if (tid == 0) { n_static = (int)(total >> 16); n_bodies = (int)(total >> 16) + (int)(total & 0xFFFFu); }
    __syncthreads();                                     // every lane has read ht_val as "first position"
    const bool fits = n_bodies <= NB && n_static <= v.max_static && joints_here == count;      // (the units' joints must be the bin's)
    if (live && fits) {
        unsigned sb = before >> 16, db = before & 0xFFFFu;
        for (int s = 0; s < 2; ++s)
            if (lead[s]) {
                const int local = stat[s] ? (int)sb++ : n_static + (int)db++;
                ht_val[hs[s]] = local;
                v.bodies[(size_t)g * NB + local] = b[s];
            }
    }
    __syncthreads();
    const bool unit = live && fits;                      // this lane has a unit
    int loc[2] = {0, 0};
    if (unit) { loc[0] = ht_val[hs[0]]; loc[1] = ht_val[hs[1]]; }
    if (unit) { atomicAdd(&degree[loc[0]], 1); atomicAdd(&degree[loc[1]], 1); }
    __syncthreads();
    // First-fit colouring of the units in priority order by Jones-Plassmann rounds (schedule.h): every round, an uncoloured
    // unit that holds the highest priority on both its dynamic bodies takes the smallest colour free on them.  One winner
    // per body per round, so the mask updates do not race.  A few rounds (two on a plain column: schedule.h colour_priority) of
    // TWO barriers each: the keys carry the round in their top bits, so a round's maxima outrank whatever earlier rounds left in the
    // table and nothing has to be cleared between rounds (round 5 cleared the winners' entries behind a third barrier).
    // The key inside the bin: round | parity of the lower body | the hash's 31 bits | the lane (lanes are in joint order: the joint
    // index's tie-break) — the order of colour_priority among the bin's units.
    unsigned long long* best = reinterpret_cast<unsigned long long*>(ht_key);        // the hash table is dead by now (NB * 8 <= HT * 4)
    for (int i = tid; i < NB; i += LANES) best[i] = 0ull;
    const bool dyn0 = loc[0] >= n_static, dyn1 = loc[1] >= n_static;                  // static bodies sit first in the table
    const unsigned long long key0 = ((unsigned long long)(~(unsigned)(b[0] < b[1] ? b[0] : b[1]) & 1u) << 57) | ((unsigned long long)colour_hash31((unsigned)cpi) << 26) | ((unsigned long long)tid << 16) | 1ull;
    bool pending = unit;
    int mycol = 0, mycol_b = 0;
    const bool from_top = ((b[0] < b[1] ? b[0] : b[1]) & 1) != 0;
    const int d0 = dyn0 ? degree[loc[0]] : 0, d1 = dyn1 ? degree[loc[1]] : 0;      // (complete: behind the barrier above)
    __syncthreads();
    for (int round = 0;; ++round) {
        const unsigned tag = (unsigned)(round % 62) + 1u;                                 // 6 bits; the table is wiped when the tags wrap (a bin of hundreds of rounds: never a stack)
        if (round > 0 && tag == 1u) {
            for (int i = tid; i < NB; i += LANES) best[i] = 0ull;
            __syncthreads();
        }
        const unsigned long long key = ((unsigned long long)tag << 58) | key0;
        if (pending) { if (dyn0) atomicMax(&best[loc[0]], key); if (dyn1) atomicMax(&best[loc[1]], key); }
        __syncthreads();
        const bool win = pending && (!dyn0 || best[loc[0]] == key) && (!dyn1 || best[loc[1]] == key);
        if (win) {
            unsigned long long m = 0, mb = 0;
            if (dyn0) { m |= used[loc[0]]; mb |= used_b[loc[0]]; }
            if (dyn1) { m |= used[loc[1]]; mb |= used_b[loc[1]]; }
            if (!~m) bad = 1;
            else {
                mycol = __builtin_ctzll(~m);
                if (dyn0) used[loc[0]] |= 1ull << mycol;
                if (dyn1) used[loc[1]] |= 1ull << mycol;
                atomicOr(&seen_a[comp], 1ull << mycol);
            }
            // candidate B: the same winner, the two-ended choice
            const int cb = colour_pick_two_ended(mb, d0 > d1 ? d0 : d1, from_top);
            if (cb < 0) bad_b[comp] = 1;
            else {
                mycol_b = cb;
                if (dyn0) used_b[loc[0]] |= 1ull << cb;
                if (dyn1) used_b[loc[1]] |= 1ull << cb;
                atomicOr(&seen_b[comp], 1ull << cb);
            }
            pending = false;
        }
        if (!__syncthreads_or(pending ? 1 : 0)) break;
    }
    // (manifolds) the joints' side of the pairing: a joint that does not name its contact point, or sits on other bodies, spoils the build
    if (from_manifolds && live && !mismatch)
        mismatch = q0.contact_point_index != cpi || q0.body1 != b[0] || q0.body2 != b[1] || (mate >= 0 && (q1.contact_point_index != (cpi | 1) || q1.body1 != b[0] || q1.body2 != b[1]));
    if (mismatch) bad = 1;
    // every component keeps the candidate that gives it fewer colours (A on a tie), renumbered densely in increasing order
    {
        const bool use_b = !bad_b[comp] && __popcll(seen_b[comp]) < __popcll(seen_a[comp]);
        const unsigned long long seen = use_b ? seen_b[comp] : seen_a[comp];
        const int c = use_b ? mycol_b : mycol;
        mycol = __popcll(seen & ((1ull << c) - 1ull));
    }
    // placement, class by class: leaders that have a follower (joint order), single leaders (joint order), then the followers
    // in their leaders' order
    const bool placed = unit && !bad;
    const bool paired = placed && mate >= 0;
    const int rank_with = bin_wave_rank<LANES>(paired, mycol, wave_with);
    const int rank_single = bin_wave_rank<LANES>(placed && !paired, mycol, wave_single);
    __syncthreads();
    if (tid < 64) {                                      // lane c: per wave -> exclusive over waves; class sizes and starts
        unsigned run_w = 0, run_s = 0;
        for (int w = 0; w < LANES / 64; ++w) {
            const unsigned tw = wave_with[w * 64 + tid], ts = wave_single[w * 64 + tid];
            wave_with[w * 64 + tid] = (unsigned short)run_w; wave_single[w * 64 + tid] = (unsigned short)run_s;
            run_w += tw; run_s += ts;
        }
        with_n[tid] = run_w; single_n[tid] = run_s;
        unsigned xs = 2 * run_w + run_s, xu = run_w + run_s;
        const unsigned slots_c = xs, units_c = xu;
        for (int off = 1; off < 64; off <<= 1) { const unsigned ys = __shfl_up(xs, off), yu = __shfl_up(xu, off); if (lane >= off) { xs += ys; xu += yu; } }
        class_begin[tid] = xs - slots_c; unit_begin[tid] = xu - units_c;
        const unsigned long long nonempty = __ballot(units_c != 0);
        const int ncol_here = nonempty ? 64 - __builtin_clzll(nonempty) : 0;
        if (tid == 0) n_col = ncol_here;
        // the classes' lane ranges (schedule.h layout_classes, restated on the wave's scalar unit: lane c holds class c's count, the
        // running state is wave-uniform)
        int remaining = __builtin_amdgcn_readlane((int)xu, 63), cursor = 0, gap_at = 0, gap_n = 0, my_begin = 0;
        for (int c = 0; c < ncol_here; ++c) {
            const int n = __builtin_amdgcn_readlane((int)units_c, c);
            int at;
            if (n <= gap_n) { at = gap_at; gap_at += n; gap_n -= n; }
            else {
                at = cursor;
                const int aligned = (at + 63) & ~63;
                if ((at & 63) + n > ((n + 63) & ~63) && aligned + remaining <= T) { gap_at = cursor; gap_n = aligned - cursor; at = aligned; }
                cursor = at + n;
            }
            if (lane == c) my_begin = at;
            remaining -= n;
        }
        cls_span[tid] = (unsigned)my_begin | (units_c << 16);
    }
    __syncthreads();
    if (!fits || bad) { spoil_build(); return; }
    if (placed) {
        const int c = mycol;
        const int r = paired ? (int)wave_with[wave * 64 + c] + rank_with : (int)wave_single[wave * 64 + c] + rank_single;
        const int in_class = paired ? r : (int)with_n[c] + r;          // position among the class's leaders
        const int slot = begin + (int)class_begin[c] + in_class;
        const unsigned local = (unsigned)loc[0] | ((unsigned)loc[1] << 16);
        v.order[slot] = j; v.slot_local[slot] = local; v.slot_colour[slot] = (unsigned char)c;
        int fslot = -1;
        if (paired) {
            fslot = begin + (int)class_begin[c] + (int)with_n[c] + (int)single_n[c] + r;
            v.order[fslot] = mate; v.slot_local[fslot] = local; v.slot_colour[fslot] = (unsigned char)c;
        }
        // (a follower's contact point is its leader's + 1: the two ids of a unit differ in the lowest bit and the leader carries the even one)
        const size_t at = 2 * ((size_t)g * T + (cls_span[c] & 0xFFFFu) + in_class);      // (its lane: schedule.h LANES)
        v.unit_recs[at] = make_int4(j, paired ? mate : -1, cpi, cpi ^ 1);
        v.unit_recs[at + 1] = make_int4((int)local, c, slot, fslot);
    }
    {                                                    // the lanes no class covers are nobody's (gaps of the layout, the tail)
        bool taken = false;
        for (int c = 0; c < n_col; ++c) { const unsigned sp = cls_span[c]; taken |= (unsigned)(tid - (int)(sp & 0xFFFFu)) < (sp >> 16); }
        if (!taken) v.unit_recs[2 * ((size_t)g * T + tid)] = make_int4(-1, -1, 0, 0);
    }
    if (tid == 0) { v.desc[g] = make_int4(begin, count, g * NB, n_bodies); v.ncol[g] = n_col; v.units[g] = island_units_word(nu, n_col, n_static); }      // (static bodies sit first in the table: the island kernel checks exactly that)
}

// ---- the HBM group (islands too big for a workgroup, or everything in Single mode): the same colouring in HBM ----------
// First fit in priority order (schedule.h) is a dependency graph: an entry may take its colour once every higher-priority
// entry on its two dynamic bodies has taken one.  Round 2's builder ran Jones-Plassmann rounds over ALL uncoloured entries
// (each recomputing 'am I the highest priority left on my bodies' through per-body tables): ~9 random accesses per survivor
// per round, 40 rounds and 1.9 ms for a merged 7e5-joint island.  Here the graph is made explicit once and then WALKED:
//   prepare   per entry: bodies + priority cached in 16 bytes, dynamic bodies' entry counts
//   lists     per dynamic body: its entries (counting sort by body); one lane per list slot finds the entry just above it
//             in priority, which gives every entry its successor on that body and the number of predecessors it waits
//             for (0, 1 or 2)
//   rounds    a round colours the current FRONTIER (entries whose predecessors are all coloured) and releases their
//             successors into the next frontier; every entry is visited once, a round costs its frontier, and two frontier
//             entries never share a dynamic body (both would have to be the first uncoloured entry of its list), so the
//             bodies' colour masks need no atomics.
// The colours are the ones the host builder computes (tests/test_solver_gpu.py compares the schedules).
constexpr unsigned JP_NONE = 0xFFFFFFFFu;
constexpr int JP_MAX_COLOURS = 64;
constexpr int JP_LIST_MAX = 4096;            // longer lists (one body in thousands of joints): host builder
constexpr unsigned JP_STATIC_BIT = 0x80000000u;
constexpr unsigned char JP_INTERIOR = 4, JP_LEVEL1 = 8;      // an interior unit of a partitioned component (schedule.h), and its level
constexpr int JP_FRONT_T = 256;       // lanes per workgroup of a round (1024 was measured slower: a 5e4-entry frontier then covers 50 CUs)
constexpr int JP_SUBLISTS = 8;        // a frontier is kept as 8 lists with a counter each: same-address atomics with a return value cost ~35 ns
                                      // apiece, serialised — one counter made them half of a round

struct JpView {
    const unsigned* ids;              // the group's joints, ascending joint index
    int count;
    const phx_contact_joint* joints;
    const unsigned char* is_static;
    int nb;
    uint4* ent;                       // per entry: {body1 | static bit, body2 | static bit, priority lo, priority hi}
    unsigned* offset;                 // per body + 1: entries of the group on it (dynamic bodies only), then their exclusive scan
    unsigned* cursor;                 // per body: fill position while the lists are built
    uint4* adj;                       // per (body, position): {priority lo, hi, entry, which of its bodies} — unordered
    unsigned* ent_comp;               // per entry: connected component (ncomp: both bodies static)
    uint2* succ;                      // per entry: the next entry on body1 / body2 (JP_NONE: last, or the body is static)
    unsigned* pred;                   // per entry: predecessors not coloured yet | predecessors << 16
    unsigned long long* used;         // per body: colours taken (candidate A: smallest free colour)
    unsigned long long* used_b;       // per body: colours taken under candidate B (two-ended, schedule.h)
    unsigned* colour_b;               // per entry: candidate B's colour
    const int* joint_comp;            // joint -> connected component (-1: both bodies static)
    const int* partner;               // joint -> the other joint of its unit, or -1
    unsigned char* kind;              // per entry: 0 leads a unit of two, 1 a unit of one, 2 follower (takes no part in the colouring);
                                      // | JP_INTERIOR: an interior unit of a partitioned component (schedule.h) — it is coloured inside
                                      // its part (k_colour_parts) and reports through seen_b / colour_b, which candidate B never touches
                                      // in a component that big
    int ncomp;
    unsigned long long* seen_a;       // per component (entry ncomp = the static-static joints): colours in use under A / B
    unsigned long long* seen_b;
    unsigned long long* seen_c;       // per component: classes in use among its level-1 interior units (level 0: seen_b)
    unsigned char* bad_b;             // per component: B ran out of its 64 colours
    const unsigned* comp_size;        // per component: joints (B is attempted only up to COLOUR_B_MAX_JOINTS)
    unsigned* colour;                 // per entry: JP_NONE until coloured
    unsigned* touched;                // per body: 1 if the group touches it (nb + 1 words, scanned afterwards)
    int* counts;                      // per round and sublist: size of the frontier it colours
    int* flags;                       // bit 0: body index out of range, bit 1: more than JP_MAX_COLOURS colours, bit 2: a list longer than JP_LIST_MAX;
                                      // flags[1] = KI0, flags[2] = KI1: the group's interior classes per level (k_jp_interior_classes)
    unsigned* hist;                   // per sort key 2 * class + kind: leaders (filled by the choice)
};

// every per-body table and the small words in ONE launch (a memset is a dispatch of its own, and there were a dozen)
static __global__ void __launch_bounds__(256) k_jp_clear(JpView v, int rounds_max)
{
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
    for (int i = i0; i <= v.nb; i += stride) {
        v.touched[i] = 0u; v.offset[i] = 0u;
        if (i < v.nb) { v.cursor[i] = 0u; v.used[i] = 0ull; v.used_b[i] = 0ull; }
    }
    for (int i = i0; i <= v.ncomp; i += stride) { v.seen_a[i] = 0ull; v.seen_b[i] = 0ull; v.seen_c[i] = 0ull; v.bad_b[i] = 0; }
    for (int i = i0; i < (rounds_max + 1) * JP_SUBLISTS; i += stride) v.counts[i] = 0;
    if (i0 < 2 * JP_MAX_COLOURS) v.hist[i0] = 0u;
    if (i0 == 0) { v.flags[0] = 0; v.flags[1] = 0; v.flags[2] = 0; }
}

static __global__ void __launch_bounds__(256) k_jp_prepare(JpView v)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < v.count; k += gridDim.x * blockDim.x) {
        const unsigned j = v.ids[k];
        const phx_contact_joint jt = v.joints[j];
        unsigned a = (unsigned)jt.body1, b = (unsigned)jt.body2;
        const unsigned long long key = colour_priority((unsigned)jt.contact_point_index, j, a < b ? a : b);
        v.pred[k] = 0u; v.colour_b[k] = 0u; v.colour[k] = JP_NONE;
        { const int jc = v.joint_comp[j]; v.ent_comp[k] = jc < 0 ? (unsigned)v.ncomp : (unsigned)jc; }
        v.succ[k] = make_uint2(JP_NONE, JP_NONE);
        const int mate = v.partner[j];
        unsigned char kind = mate < 0 ? 1 : ((jt.contact_point_index & 1) ? 2 : 0);
        if (kind != 2 && a < (unsigned)v.nb && b < (unsigned)v.nb) {
            const int jc = v.joint_comp[j];
            if (jc >= 0 && v.comp_size[jc] > (unsigned)COLOUR_B_MAX_JOINTS) {
                const int part = unit_part(a, b, v.is_static[a] != 0, v.is_static[b] != 0, v.nb);
                if (part >= 0) kind |= part < parts_per_level(v.nb) ? JP_INTERIOR : (unsigned char)(JP_INTERIOR | JP_LEVEL1);
            }
        }
        v.kind[k] = kind;
        if (a >= (unsigned)v.nb || b >= (unsigned)v.nb) {                  // reported; the entry is parked on nothing
            atomicOr(v.flags, 1);
            v.ent[k] = make_uint4(JP_STATIC_BIT, JP_STATIC_BIT, (unsigned)key, (unsigned)(key >> 32));
            continue;
        }
        v.touched[a] = 1u; v.touched[b] = 1u;
        if ((kind & 3) == 2) {                                             // a follower: its leader colours the unit
            v.ent[k] = make_uint4(JP_STATIC_BIT, JP_STATIC_BIT, (unsigned)key, (unsigned)(key >> 32));
            continue;
        }
        if (kind & JP_INTERIOR) {                                          // coloured inside its part (k_colour_parts): no list, no frontier
            v.ent[k] = make_uint4(a, b, (unsigned)key, (unsigned)(key >> 32));
            continue;
        }
        if (v.is_static[a]) a |= JP_STATIC_BIT; else atomicAdd(&v.offset[a], 1u);
        if (v.is_static[b]) b |= JP_STATIC_BIT; else atomicAdd(&v.offset[b], 1u);
        v.ent[k] = make_uint4(a, b, (unsigned)key, (unsigned)(key >> 32));
    }
}

// the lists, unordered: a slot carries the entry's priority and which of its bodies this is, so that ordering a list reads
// nothing but the list
static __global__ void __launch_bounds__(256) k_jp_fill(JpView v)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < v.count; k += gridDim.x * blockDim.x) {
        const uint4 e = v.ent[k];
        if (v.kind[k] & JP_INTERIOR) continue;
        if (!(e.x & JP_STATIC_BIT)) v.adj[v.offset[e.x] + atomicAdd(&v.cursor[e.x], 1u)] = make_uint4(e.z, e.w, (unsigned)k, e.x);
        if (!(e.y & JP_STATIC_BIT)) v.adj[v.offset[e.y] + atomicAdd(&v.cursor[e.y], 1u)] = make_uint4(e.z, e.w, (unsigned)k, e.y | JP_STATIC_BIT);
    }
}

__device__ __forceinline__ unsigned long long jp_key(const uint4& slot) { return ((unsigned long long)slot.y << 32) | slot.x; }

// One lane per list slot.  Keys are unique, so counting the keys above mine is a sort: the smallest of them belongs to my
// predecessor on this body, which learns here that I am its successor; unless I am the head I wait for one more entry.
// (word 3 of a slot: the body, top bit = 'the entry's second body')
static __global__ void __launch_bounds__(256) k_jp_lists(JpView v)
{
    const int slots = (int)v.offset[v.nb];              // dynamic sides only: at most two per entry
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < slots; t += gridDim.x * blockDim.x) {
        const uint4 me = v.adj[t];
        const unsigned body = me.w & ~JP_STATIC_BIT;
        const unsigned o = v.offset[body];
        const int d = (int)(v.offset[body + 1] - o);
        if (d > JP_LIST_MAX) { if ((unsigned)t == o) atomicOr(v.flags, 4); continue; }
        const unsigned long long key = jp_key(me);
        unsigned long long above = ~0ull;
        unsigned pred_entry = JP_NONE, pred_word = 0;
        for (int j = 0; j < d; ++j) {
            const uint4 other = v.adj[o + j];
            const unsigned long long ok = jp_key(other);
            if (ok > key && ok < above) { above = ok; pred_entry = other.z; pred_word = other.w; }
        }
        if (pred_entry == JP_NONE) continue;             // head of the list
        atomicAdd(&v.pred[me.z], 0x10001u);              // low half: predecessors still uncoloured; high half: how many there were
        if (pred_word & JP_STATIC_BIT) v.succ[pred_entry].y = me.z; else v.succ[pred_entry].x = me.z;
    }
}

// round 0's frontier: flag the entries that wait for nobody, scan, compact
static __global__ void __launch_bounds__(256) k_jp_seed_flags(JpView v, unsigned* __restrict__ flags)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k <= v.count; k += gridDim.x * blockDim.x) flags[k] = (k < v.count && v.pred[k] == 0u && (v.kind[k] & 3) != 2 && !(v.kind[k] & JP_INTERIOR)) ? 1u : 0u;
}

static __global__ void __launch_bounds__(256) k_jp_seed(JpView v, const unsigned* __restrict__ scan, unsigned* __restrict__ list_out)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < v.count; k += gridDim.x * blockDim.x)
        if (scan[k + 1] != scan[k]) list_out[(size_t)(scan[k] % JP_SUBLISTS) * v.count + scan[k] / JP_SUBLISTS] = (unsigned)k;
    if (blockIdx.x == 0 && threadIdx.x < JP_SUBLISTS) v.counts[threadIdx.x] = ((int)scan[v.count] + JP_SUBLISTS - 1 - (int)threadIdx.x) / JP_SUBLISTS;
}

// One round: colour the frontier, release the successors.  (Round 0's frontier — every entry that waits for nobody — is
// compacted by k_jp_seed_flags + a scan + k_jp_seed.)  A lane takes JP_ITEMS entries per trip (their dependent loads overlap) and the next frontier is appended with ONE
// atomic per workgroup and trip (thousands of same-address atomics per round serialise: measured 126 us for seeding 7e5
// entries with one per wave).
constexpr int JP_ITEMS = 1;           // (4 entries per lane was measured slower: a round is latency bound and wants the lanes)


// one frontier entry takes its classes under both candidates and releases its successors (s0, s1: the ones it was the last predecessor of)
__device__ __forceinline__ void jp_colour_entry(const JpView& v, unsigned k, int& comp, unsigned long long& got_a, unsigned long long& got_b, unsigned& s0, unsigned& s1)
{
    const uint4 e = v.ent[k];
    const bool da = !(e.x & JP_STATIC_BIT), db = !(e.y & JP_STATIC_BIT);
    const unsigned a = e.x & ~JP_STATIC_BIT, b = e.y & ~JP_STATIC_BIT;
    comp = (int)v.ent_comp[k];
    unsigned long long m = 0;
    int c = 0;
    if (da) m |= v.used[a];
    if (db) m |= v.used[b];
    if (!~m) atomicOr(v.flags, 2);
    else {
        c = __builtin_ctzll(~m);
        if (da) v.used[a] |= 1ull << c;
        if (db) v.used[b] |= 1ull << c;
        got_a = 1ull << c;
    }
    if (comp < v.ncomp && v.comp_size[comp] <= (unsigned)COLOUR_B_MAX_JOINTS) {      // candidate B: the same turn, the two-ended choice
        unsigned long long mb = 0;
        if (da) mb |= v.used_b[a];
        if (db) mb |= v.used_b[b];
        const int d0 = da ? (int)(v.offset[a + 1] - v.offset[a]) : 0, d1 = db ? (int)(v.offset[b + 1] - v.offset[b]) : 0;
        const int cb = colour_pick_two_ended(mb, d0 > d1 ? d0 : d1, ((a < b ? a : b) & 1u) != 0);
        if (cb < 0) v.bad_b[comp] = 1;
        else {
            if (da) v.used_b[a] |= 1ull << cb;
            if (db) v.used_b[b] |= 1ull << cb;
            got_b = 1ull << cb;
            v.colour_b[k] = (unsigned)cb;
        }
    }
    v.colour[k] = (unsigned)c;
    const uint2 s = v.succ[k];
    if (s.x != JP_NONE && (atomicSub(&v.pred[s.x], 1u) & 0xFFFFu) == 1u) s0 = s.x;
    if (s.y != JP_NONE && (atomicSub(&v.pred[s.y], 1u) & 0xFFFFu) == 1u) s1 = s.y;
}

// 'colours in use' of the components, wave-aggregated: in a merged island every entry of a round belongs to ONE component, and
// thousands of same-address atomics serialise
__device__ __forceinline__ void jp_note_colours(const JpView& v, int comp, unsigned long long got_a, unsigned long long got_b)
{
    const int lane = threadIdx.x & 63;
    for (unsigned long long todo = __ballot((got_a | got_b) != 0); todo;) {
        const int leader = __builtin_ctzll(todo);
        const int lc = __shfl(comp, leader);
        const bool mine = (got_a | got_b) != 0 && comp == lc;
        unsigned long long ra = mine ? got_a : 0ull, rb = mine ? got_b : 0ull;
        for (int off = 32; off > 0; off >>= 1) { ra |= __shfl_xor(ra, off); rb |= __shfl_xor(rb, off); }
        if (lane == leader) {
            if (ra & ~v.seen_a[lc]) atomicOr(&v.seen_a[lc], ra);
            if (rb & ~v.seen_b[lc]) atomicOr(&v.seen_b[lc], rb);
        }
        todo &= ~__ballot(mine);
    }
}

static __global__ void __launch_bounds__(JP_FRONT_T) k_jp_front(JpView v, int round, const unsigned* __restrict__ list_in, unsigned* __restrict__ list_out)
{
    __shared__ int wave_n[JP_FRONT_T / 64];
    __shared__ int block_base;
    const int sub = blockIdx.x % JP_SUBLISTS, sub_block = blockIdx.x / JP_SUBLISTS, sub_blocks = gridDim.x / JP_SUBLISTS;      // (the grid is a multiple of JP_SUBLISTS)
    const int n = v.counts[round * JP_SUBLISTS + sub];
    list_in += (size_t)sub * v.count; list_out += (size_t)sub * v.count;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int base = sub_block * blockDim.x * JP_ITEMS; base < n; base += sub_blocks * blockDim.x * JP_ITEMS) {      // workgroup-uniform trip count
        unsigned rel[2 * JP_ITEMS];                          // released successors
        int released = 0;
#pragma unroll
        for (int it = 0; it < JP_ITEMS; ++it) {
            const int i = base + it * (int)blockDim.x + (int)threadIdx.x;
            unsigned s0 = JP_NONE, s1 = JP_NONE;
            int comp = 0;
            unsigned long long got_a = 0, got_b = 0;         // the colour bits this entry took under candidates A / B
            if (i < n) jp_colour_entry(v, list_in[i], comp, got_a, got_b, s0, s1);
            rel[2 * it] = s0; rel[2 * it + 1] = s1;
            released += (s0 != JP_NONE ? 1 : 0) + (s1 != JP_NONE ? 1 : 0);
            jp_note_colours(v, comp, got_a, got_b);
        }
        // the released successors -> next frontier: exclusive position of this lane's first one inside the workgroup
        int incl = released;
        for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off); if (lane >= off) incl += y; }
        if (lane == 63) wave_n[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            int total = 0;
            for (int w = 0; w < JP_FRONT_T / 64; ++w) total += wave_n[w];
            block_base = total ? atomicAdd(v.counts + (round + 1) * JP_SUBLISTS + sub, total) : 0;
        }
        __syncthreads();
        int at = block_base + incl - released;
        for (int w = 0; w < wave; ++w) at += wave_n[w];
#pragma unroll
        for (int q = 0; q < 2 * JP_ITEMS; ++q) if (rel[q] != JP_NONE) list_out[at++] = rel[q];
        __syncthreads();                                                     // wave_n / block_base are reused by the next trip
    }
}

// The WHOLE walk in one launch, by one workgroup (round 6): for the few thousand units that are not interior to a part of a partitioned
// island (its boundary units) or the joints of a small Single-mode world, a round is a microsecond of work behind a launch and the walk is
// eight to twenty rounds deep — plus the host's look at the frontier sizes in between.  One workgroup walks round after round with a
// barrier in between (its own stores are visible to its later loads: one CU, one L1; the predecessor counts are atomics at the coherent
// level) until the frontier is empty, and leaves {rounds, entries walked, 'ran into rounds_max'} for the build's last readback.  Frontier r
// is read from all JP_SUBLISTS sublists (round 0's come from k_jp_seed) and frontier r + 1 is written to sublist 0 alone — one workgroup
// needs no atomics to append.  The host takes this path when the previous build walked few entries (solver_build.hip); the classes are
// k_jp_front's: both call jp_colour_entry, and a frontier's entries share no dynamic body whatever order they are taken in.
constexpr int JP_WALK_T = 1024;
constexpr long long JP_WALK_ONE_MAX = 32768;      // entries the previous build walked, at most, for the host to take this path (a trip of the one workgroup is ~1 us)
static __global__ void __launch_bounds__(JP_WALK_T) k_jp_walk_one(JpView v, int rounds_max, unsigned* __restrict__ list0, unsigned* __restrict__ list1, int* __restrict__ result)
{
    __shared__ int pre[JP_SUBLISTS + 1];
    __shared__ int wave_n[JP_WALK_T / 64];
    __shared__ int out_n;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int walked = 0, round = 0;
    if (tid == 0) out_n = 0;
    for (; round < rounds_max; ++round) {
        const unsigned* list_in = (round & 1) ? list1 : list0;
        unsigned* list_out = ((round & 1) ? list0 : list1);                  // (sublist 0)
        if (tid == 0) {                                                      // (round 0: k_jp_seed's eight counts; later rounds: what this workgroup appended to sublist 0)
            int at = 0;
            for (int q = 0; q < JP_SUBLISTS; ++q) { pre[q] = at; at += round == 0 ? v.counts[q] : (q == 0 ? out_n : 0); }
            pre[JP_SUBLISTS] = at; out_n = 0;
        }
        __syncthreads();
        const int n = pre[JP_SUBLISTS];
        if (n == 0) break;                                                   // (workgroup-uniform)
        walked += n;
        for (int base = 0; base < n; base += JP_WALK_T) {
            const int i = base + tid;
            unsigned s0 = JP_NONE, s1 = JP_NONE;
            int comp = 0;
            unsigned long long got_a = 0, got_b = 0;
            if (i < n) {
                int q = 0;
                while (i >= pre[q + 1]) ++q;
                jp_colour_entry(v, list_in[(size_t)q * v.count + (i - pre[q])], comp, got_a, got_b, s0, s1);
            }
            const int released = (s0 != JP_NONE ? 1 : 0) + (s1 != JP_NONE ? 1 : 0);
            jp_note_colours(v, comp, got_a, got_b);
            int incl = released;
            for (int off = 1; off < 64; off <<= 1) { const int y = __shfl_up(incl, off); if (lane >= off) incl += y; }
            if (lane == 63) wave_n[wave] = incl;
            __syncthreads();
            int at = out_n + incl - released, total = 0;
            for (int w = 0; w < JP_WALK_T / 64; ++w) { if (w < wave) at += wave_n[w]; total += wave_n[w]; }
            if (s0 != JP_NONE) list_out[at++] = s0;
            if (s1 != JP_NONE) list_out[at++] = s1;
            __syncthreads();                                                 // (everybody has read out_n and wave_n)
            if (tid == 0) out_n += total;
        }
        __syncthreads();
        if (tid == 0) v.counts[(round + 1) * JP_SUBLISTS] = out_n;          // (the other sublists of round + 1 are empty: k_jp_clear)
        __syncthreads();
    }
    if (tid == 0) { result[0] = round; result[1] = walked; result[2] = round >= rounds_max ? 1 : 0; }
}

// ---- the interior units of partitioned components, coloured part by part in LDS ----------------------------------------
// An interior unit conflicts only with interior units of its own part (schedule.h), so first fit in priority order is a problem
// of ~1e3 units on 512 bodies: one workgroup per part, Jones-Plassmann rounds on LDS — a unit takes the smallest class free on
// its two bodies once it holds the highest priority among the uncoloured units on both — with barriers where the global walk
// (k_jp_front) has kernel launches.  The same classes as one sequential pass in decreasing priority.
constexpr int CP_T = 256, CP_MAXU = 3072;      // a denser part (> 6 units per body) sends the build to the host builder; (512 lanes: the same time, 1024: 73 us against 55, 128: 87)

// keys for the sort by part: an interior entry's part, every other entry behind them all
static __global__ void __launch_bounds__(256) k_part_sort_keys(JpView v, unsigned behind, unsigned* __restrict__ keys, unsigned* __restrict__ vals)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < v.count; k += gridDim.x * blockDim.x) {
        keys[k] = (v.kind[k] & JP_INTERIOR) ? (unsigned)unit_part(v.ent[k].x, v.ent[k].y, false, false, v.nb) : behind;
        vals[k] = (unsigned)k;
    }
}

// out[t] = the first position of `sorted_keys` (n of them) whose key is >= t, for t = 0 .. count
static __global__ void __launch_bounds__(256) k_lower_bounds(const unsigned* __restrict__ sorted_keys, int n, int count, int* __restrict__ out)
{
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t <= count; t += gridDim.x * blockDim.x) {
        int lo = 0, hi = n;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (sorted_keys[mid] < (unsigned)t) lo = mid + 1; else hi = mid; }
        out[t] = lo;
    }
}

static __global__ void __launch_bounds__(CP_T) k_colour_parts(JpView v, const unsigned* __restrict__ sorted_entries, const int* __restrict__ part_begin)
{
    __shared__ unsigned long long s_prio[CP_MAXU], s_used[PART_BODIES], s_max[2][PART_BODIES];      // (two tables of maxima in turn: one is cleared while the other decides)
    __shared__ unsigned s_bodies[CP_MAXU];
    __shared__ unsigned char s_col[CP_MAXU];
    const int part = blockIdx.x, tid = threadIdx.x;
    const int pb = part_begin[part], n = part_begin[part + 1] - pb;
    if (n <= 0) return;
    if (n > CP_MAXU) { if (tid == 0) atomicOr(v.flags, 4); return; }
    const int base = part_first_body(part, v.nb);
    unsigned long long* seen = part < parts_per_level(v.nb) ? v.seen_b : v.seen_c;
    for (int i = tid; i < n; i += CP_T) {
        const uint4 e = v.ent[sorted_entries[pb + i]];
        s_prio[i] = ((unsigned long long)e.w << 32) | e.z;
        s_bodies[i] = (unsigned)(((int)e.x - base) & (PART_BODIES - 1)) | ((unsigned)(((int)e.y - base) & (PART_BODIES - 1)) << 16);
        s_col[i] = 0xFF;
    }
    for (int b = tid; b < PART_BODIES; b += CP_T) { s_used[b] = 0ull; s_max[0][b] = 0ull; s_max[1][b] = 0ull; }
    __syncthreads();
    for (int round = 0;; ++round) {
        unsigned long long* mx = s_max[round & 1];
        for (int i = tid; i < n; i += CP_T)
            if (s_col[i] == 0xFF) {
                const unsigned bb = s_bodies[i];
                atomicMax(&mx[bb & 0xFFFFu], s_prio[i]);
                atomicMax(&mx[bb >> 16], s_prio[i]);
            }
        __syncthreads();
        int left = 0;
        for (int i = tid; i < n; i += CP_T)
            if (s_col[i] == 0xFF) {
                const unsigned bb = s_bodies[i], b1 = bb & 0xFFFFu, b2 = bb >> 16;
                const unsigned long long p = s_prio[i];
                if (mx[b1] == p && mx[b2] == p) {                  // nobody else on these two bodies takes a class in this round
                    const unsigned long long m = s_used[b1] | s_used[b2];
                    int c = 0;
                    if (!~m) atomicOr(v.flags, 2); else c = __builtin_ctzll(~m);
                    s_used[b1] |= 1ull << c; s_used[b2] |= 1ull << c;
                    s_col[i] = (unsigned char)c;
                } else left = 1;
            }
        for (int b = tid; b < PART_BODIES; b += CP_T) s_max[(round + 1) & 1][b] = 0ull;      // (last read before the previous round's closing barrier)
        if (!__syncthreads_or(left)) break;
    }
    for (int i = tid; i < n; i += CP_T) {
        const unsigned k = sorted_entries[pb + i], c = s_col[i];
        v.colour_b[k] = c; v.colour[k] = c;
        const unsigned comp = v.ent_comp[k];
        const unsigned long long bit = 1ull << c;
        if (!(__hip_atomic_load(&seen[comp], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(&seen[comp], bit);
    }
}

// KI0, KI1 = the largest interior class counts, per level, among the group's partitioned components (schedule.h)
static __global__ void __launch_bounds__(256) k_jp_interior_classes(JpView v)
{
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < v.ncomp; c += gridDim.x * blockDim.x)
        if (v.comp_size[c] > (unsigned)COLOUR_B_MAX_JOINTS) {
            if (v.seen_b[c]) atomicMax(v.flags + 1, __popcll(v.seen_b[c]));
            if (v.seen_c[c]) atomicMax(v.flags + 2, __popcll(v.seen_c[c]));
        }
}

// every component keeps the candidate that gives it fewer colours (A on a tie), renumbered densely in increasing order; the
// interior classes of the partitioned components come first, everything else is numbered from KI
static __global__ void __launch_bounds__(256) k_jp_choose(JpView v)
{
    __shared__ unsigned h[2 * JP_MAX_COLOURS];
    if (threadIdx.x < 2 * JP_MAX_COLOURS) h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned ki0 = (unsigned)v.flags[1], ki = ki0 + (unsigned)v.flags[2];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < v.count; k += gridDim.x * blockDim.x) {
        const unsigned char kind = v.kind[k] & 3;
        if (kind == 2) { v.colour[k] = 255u; continue; }                         // followers sort behind every leader; their leaders place them
        const int comp = (int)v.ent_comp[k];
        const bool interior = (v.kind[k] & JP_INTERIOR) != 0, level1 = (v.kind[k] & JP_LEVEL1) != 0;
        const unsigned long long sa = v.seen_a[comp], sb = level1 ? v.seen_c[comp] : v.seen_b[comp];
        const bool use_b = !interior && comp < v.ncomp && v.comp_size[comp] <= (unsigned)COLOUR_B_MAX_JOINTS && !v.bad_b[comp] && __popcll(sb) < __popcll(sa);
        const unsigned c = (use_b || interior) ? v.colour_b[k] : v.colour[k];
        unsigned cls = (unsigned)__popcll(((use_b || interior) ? sb : sa) & ((1ull << c) - 1ull));
        cls += interior ? (level1 ? ki0 : 0u) : ki;
        if (cls >= (unsigned)JP_MAX_COLOURS) { atomicOr(v.flags, 2); cls = JP_MAX_COLOURS - 1; }
        const unsigned key = 2 * cls + kind;            // sort key: class, then 'leads a unit of two' before 'single'
        v.colour[k] = key;
        atomicAdd(&h[key & (2 * JP_MAX_COLOURS - 1)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 2 * JP_MAX_COLOURS && h[threadIdx.x]) atomicAdd(&v.hist[threadIdx.x], h[threadIdx.x]);
}

// the sort's input in the order `perm` (the entries sorted by part, everything but interior units behind them in entry order):
// a stable sort by (class, kind) of THAT sequence leaves an interior class laid out part by part (schedule.h) and every other class
// in joint order, in one 8-bit pass
static __global__ void __launch_bounds__(256) k_jp_sort_input(JpView v, const unsigned* __restrict__ perm, unsigned* __restrict__ keys, unsigned* __restrict__ vals)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < v.count; i += gridDim.x * blockDim.x) {
        const unsigned k = perm ? perm[i] : (unsigned)i;
        keys[i] = v.colour[k];
        vals[i] = v.ids[k];
    }
}

// the leaders, sorted by (class, kind) — interior classes part by part, otherwise stable in joint order — take their slots and give
// their followers theirs: class c = [leaders with a follower][single leaders][followers, in their leaders' order]  (hist: leaders
// per sort key).  Interior classes also leave their parts' slot ranges for k_solve_parts: ranges[part * 64 + c] = {first, end of
// the leaders with a follower, first, end of the single leaders} (the table is zero on entry).
static __global__ void __launch_bounds__(256) k_jp_place(JpView v, const unsigned* __restrict__ sorted_keys, const unsigned* __restrict__ sorted_joints,
                                                        int* __restrict__ order_out, int slot_base, int* __restrict__ ranges)
{
    __shared__ unsigned slot_begin[JP_MAX_COLOURS], lead_begin[JP_MAX_COLOURS], lead_n[JP_MAX_COLOURS];
    if (threadIdx.x < 64) {
        const unsigned w = v.hist[2 * threadIdx.x], s = v.hist[2 * threadIdx.x + 1];
        unsigned xs = 2 * w + s, xl = w + s;
        const unsigned slots_c = xs, lead_c = xl;
        for (int off = 1; off < 64; off <<= 1) { const unsigned ys = __shfl_up(xs, off), yl = __shfl_up(xl, off); if ((int)threadIdx.x >= off) { xs += ys; xl += yl; } }
        slot_begin[threadIdx.x] = xs - slots_c; lead_begin[threadIdx.x] = xl - lead_c; lead_n[threadIdx.x] = lead_c;
    }
    __syncthreads();
    const int leaders = (int)(lead_begin[JP_MAX_COLOURS - 1] + lead_n[JP_MAX_COLOURS - 1]);
    const unsigned ki = (unsigned)(v.flags[1] + v.flags[2]);
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < leaders; p += gridDim.x * blockDim.x) {
        const unsigned key = sorted_keys[p], c = key >> 1;
        const int j = (int)sorted_joints[p];
        const unsigned r = (unsigned)p - lead_begin[c];
        const int slot = (int)(slot_begin[c] + r);
        order_out[slot] = j;
        if (!(key & 1u)) order_out[slot_begin[c] + lead_n[c] + r] = v.partner[j];
        if (ranges && c < ki) {                        // an interior unit: first / last of its (part, class, kind) run?
            auto part_of = [&](int joint) { const phx_contact_joint q = v.joints[joint]; return unit_part((unsigned)q.body1, (unsigned)q.body2, false, false, v.nb); };
            const int part = max(part_of(j), 0);           // (an interior class holds interior units only)
            int* row = ranges + ((size_t)part * JP_MAX_COLOURS + c) * 4 + 2 * (key & 1u);
            const bool first = p == 0 || sorted_keys[p - 1] != key || part_of((int)sorted_joints[p - 1]) != part;
            const bool last = p + 1 >= leaders || sorted_keys[p + 1] != key || part_of((int)sorted_joints[p + 1]) != part;
            if (first) row[0] = slot_base + slot;
            if (last) row[1] = slot_base + slot + 1;
        }
    }
}

// bodies flagged in `scan` (already exclusive-scanned, nb + 1 entries) -> ascending list
static __global__ void __launch_bounds__(256) k_compact_flagged(const unsigned* __restrict__ scan, int nb, int* __restrict__ list)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x)
        if (scan[i + 1] != scan[i]) list[scan[i]] = i;
}

static __global__ void __launch_bounds__(256) k_static_flags(const unsigned char* __restrict__ is_static, int nb, unsigned* __restrict__ flags)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= nb; i += gridDim.x * blockDim.x) flags[i] = (i < nb && is_static[i]) ? 1u : 0u;
}

// static body -> its slot in the static-tag tables (rank among static bodies), dynamic body -> -1
static __global__ void __launch_bounds__(256) k_static_slots(const unsigned char* __restrict__ is_static, const unsigned* __restrict__ scan, int nb, int* __restrict__ slot)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) slot[i] = is_static[i] ? (int)scan[i] : -1;
}

static __global__ void __launch_bounds__(256) k_iota(unsigned* __restrict__ out, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = (unsigned)i;
}

} // namespace phx
