// schedule.h — host-side construction of the solve schedule (which joint is swept when, and where).
//
// A schedule is the device's answer to Solver::PrepareIndices + Solver::GatherIslands
// (ref: src/Solver.cpp:217-273, 285-454): it fixes ONE sequential sweep order of the joints and marks the
// places where that order may be executed in parallel without changing its result:
//   * a COLOUR is a run of consecutive slots whose joints share no dynamic body (PrepareIndices' idea of an
//     N-independent group, with N = the whole class instead of 4 or 8) — its joints run in parallel lanes;
//   * a GROUP is a run of consecutive colours whose joints share no dynamic body with any other group
//     (GatherIslands' islands, coalesced into workgroup-sized bins) — groups run in parallel workgroups,
//     each keeping its bodies in LDS for the whole solve.  The last group may be an "HBM group": whatever does
//     not fit a workgroup (one huge island) is solved class by class out of HBM.
// Static bodies (invMass == invInertia == 0, ref: Solver.cpp:304) never conflict; each group keeps a private
// copy of their lastIteration tag.
#pragma once

#include <algorithm>
#include <cstdint>
#include <vector>

namespace phx {

// Colouring priority of joint j whose priority id is `id` and whose LOWER body index is `lower` (the colouring rule is stated
// below).  The solver uses the joint's contactPointIndex as the id: unlike the joint's position it survives compacting the joint
// list (island sharding solves a subset of the joints and must reach the same colours).  The joint index only breaks ties between
// equal ids, so keys are unique; they are never zero (zero means "nobody" in the builders' tables).
// PARITY-MAJOR (round 6): every unit whose lower body index is EVEN ranks above every unit whose lower body index is odd; inside a
// parity the order is the fixed pseudo-random one of the hash.  First fit in priority order is a dependency graph whose depth is the
// number of parallel rounds the device builders need (a unit takes its class once every higher-priority unit on its dynamic bodies
// has one): under the hash alone a stacked column — a path — is ~7 rounds deep; with the parity in front the even units of a path
// conflict with nobody of their parity and the odd ones only wait for them: two rounds, plus one per duplicate manifold in the way.
// On irregular piles the depth is that of two hash orders in sequence.  Like every order it is one more legal Gauss-Seidel order
// and a pure function of the joints.
__host__ __device__ inline unsigned colour_hash31(unsigned id)
{
    unsigned x = id * 0x9E3779B1u;
    x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13;
    return x >> 1;
}
__host__ __device__ inline unsigned long long colour_priority(unsigned id, unsigned j, unsigned lower)
{
    return ((((unsigned long long)(~lower & 1u)) << 63) | ((unsigned long long)colour_hash31(id) << 32) | j) + 1ull;
}

// UNITS.  The joints of one body pair (the two contact points of a manifold) conflict with each other and with nobody
// else more than either does alone, so they are scheduled as one unit: two joints whose priority ids differ in the lowest bit
// only (contact points 2m and 2m + 1), whose bodies are the same and which are each the smallest-index joint carrying their
// id form a unit LED by the even id; every other joint is a unit of its own.  Units — not joints — are coloured (a unit's
// priority is its leader's), so a 'colour' is a CLASS of units that share no dynamic body, and its slots are: the leaders
// that have a follower (joint order), the single leaders (joint order), the followers in their leaders' order.  Swept
// front to back that is: every leader, then every follower — which a lane reproduces by sweeping its leader and then its
// follower on one read and one write of the two bodies.  Half the barriers (LDS groups) / kernel launches (HBM group) per
// joint; the static bodies' tags are class-synchronous (solver_kernels.h).
// Two first-fit candidates, same priority order, same dependency rounds (a unit's turn depends on the priorities only,
// not on the colours):
//   A  the smallest class free on the unit's dynamic bodies;
//   B  'two-ended': a unit whose lower body index is odd takes the LARGEST free class below K = the larger unit count of its
//      dynamic bodies (the smallest free class >= K if there is none), every other unit the smallest free class.  On layered
//      structures (a stack: consecutive body indices alternate parity along a column) the two halves of a body's units are
//      drawn from opposite ends and never collide, which reaches the optimum of max-degree classes where A needs up to 1.5x as
//      many; on irregular piles B is a little worse than A.  B is defined for at most 64 classes.
// B is only attempted for components of at most COLOUR_B_MAX_JOINTS joints — the ones that fit a workgroup: the layered
// structures it helps are small, and a larger component is partitioned (below).
// Every CONNECTED COMPONENT keeps the candidate that gives IT fewer classes (A on a tie) and renumbers its classes densely in
// increasing order.  The choice is per component, so an island's classes — hence its results — do not depend on which other
// islands share its group, on the workgroup shape or on the island mode.  A class is one barrier-separated step (LDS groups)
// or one kernel launch (HBM group) of every sweep, so the largest class count sets the solve time.
constexpr int COLOUR_B_MAX_JOINTS = 1024;

// LANES of an LDS group (execution only: no result depends on them).  The island kernel gives every unit of a group a lane for the
// whole solve and sweeps class by class, so a class costs one pass of every WAVE it has a lane in — with four groups on a CU the
// sweeps are bound by the waves' instruction issue, not by the chain of steps.  The classes' lane ranges are therefore placed so
// that they straddle as few waves as the workgroup's lanes allow: in class order, a class that would straddle one wave more than
// its size needs starts on the next wave boundary if the remaining classes still fit behind it, and a small class goes into the gap
// such a move left.  (A stacked column of cfg 2: classes of 100, 96, 5 and 4 units = 2 + 2 + 1 + 1 wave passes per sweep instead
// of the 2 + 3 + 1 + 1 of back-to-back ranges.)  A unit's lane = its class's first lane + its position among the class's leaders.
// (`total` = the sum of units[]; at most one gap is kept: a later one replaces it)
__host__ __device__ inline void layout_classes(const unsigned short* units, int ncol, int total, int T, unsigned short* begin)
{
    int remaining = total, cursor = 0, gap_at = 0, gap_n = 0;
    for (int c = 0; c < ncol; ++c) {
        const int n = units[c];
        if (n <= gap_n) { begin[c] = (unsigned short)gap_at; gap_at += n; gap_n -= n; }      // a small class into the gap an aligned one left (inside one wave)
        else {
            int at = cursor;
            const int aligned = (at + 63) & ~63;
            if ((at & 63) + n > ((n + 63) & ~63) && aligned + remaining <= T) { gap_at = cursor; gap_n = aligned - cursor; at = aligned; }      // straddles one wave more than it has to, and there is room
            begin[c] = (unsigned short)at;                                 // (cursor + remaining <= T holds throughout: the units fit the lanes)
            cursor = at + n;
        }
        remaining -= n;
    }
}

// BINNING.  Groups run in parallel workgroups and share nothing but their private copies of the static bodies' tags (a group's components
// share ITS copy: solver_kernels.h), so WHICH components share a group changes no result beyond that skip rule — only how
// well the workgroups are filled.  Consecutive components (body order) are packed greedily into a bin until the next one would
// overflow the workgroup shape's joints or units, and a bin never spans a multiple of BIN_CHUNK component numbers: that cuts the
// chain 'a bin ends where the next component would overflow it' into independent pieces, which is what lets the device make the
// bins in a few microseconds (schedule_kernels.h k_bin_components) at the price of one partly filled bin per 64 components.
constexpr int BIN_CHUNK = 64;
constexpr int BINC_MAX = 65536;          // speculative binning (schedule_kernels.h k_bin_components): components at most (tables, indices)
constexpr int BINC_JOINT_BITS = 30;      // ... and solves of < 2^30 joints (the lanes' scan packs bins << 32 | slots)

// PARTITIONED COMPONENTS.  A component of more than COLOUR_B_MAX_JOINTS joints (a settled pile: one island of 1e5-1e6 joints)
// is swept class by class out of HBM, one launch per class and sweep — a solve is classes x sweeps dependent launches.  Most of
// such an island is local: cut the bodies into PARTS of PART_BODIES consecutive indices, twice — level 0 at multiples of
// PART_BODIES, level 1 shifted by half a part.  A unit of a partitioned component whose bodies are both dynamic is INTERIOR AT
// LEVEL 0 if they lie in one level-0 part, otherwise INTERIOR AT LEVEL 1 if they lie in one level-1 part (the units that straddle
// a level-0 boundary); every other unit of the group is a REST unit.  The three kinds are coloured independently of each other
// (first fit in the same priority order, candidate A only, a unit conflicting with the units OF ITS KIND on its dynamic bodies),
// and in the group that holds the component they come in that order:
//   classes [0, KI0)        the level-0 interior units of the group's partitioned components — class c of every one of them;
//                           KI0 = the largest such class count among them (0 if there are none: then nothing here changes anything);
//   classes [KI0, KI)       the level-1 interior units, likewise (KI = KI0 + KI1);
//   classes [KI, ..)        everything else of the group: the rest units of the partitioned components and the units of its other
//                           components, each component's classes renumbered densely from KI.
// Inside an interior class the slots are laid out part by part.  Interior units of different parts of one level share no body,
// so the classes of a level need no synchronisation ACROSS parts: one launch per level and sweep sweeps them, a workgroup per
// part with the part's bodies in LDS and a barrier per class (k_solve_parts); only the rest classes remain launches of their
// own.  Like every colouring it is one more legal Gauss-Seidel order, a pure function of the component (its joints' bodies and
// ids), hence the same in every island mode and on every rank.
constexpr int PART_BODIES = 512;
static_assert((PART_BODIES & (PART_BODIES - 1)) == 0, "local body indices are masked with PART_BODIES - 1");
// parts are numbered over both levels: level 0 = [0, P), level 1 = [P, 2P + 1), P = ceil(bodies / PART_BODIES)
__host__ __device__ inline int parts_per_level(int nb) { return (nb + PART_BODIES - 1) / PART_BODIES; }
__host__ __device__ inline int parts_total(int nb) { return 2 * parts_per_level(nb) + 1; }
__host__ __device__ inline int part_first_body(int part, int nb)      // (negative for the first part of level 1)
{
    const int P = parts_per_level(nb);
    return part < P ? part * PART_BODIES : (part - P) * PART_BODIES - PART_BODIES / 2;
}
// the part a unit on bodies (a, b) is interior to, or -1 (a rest unit)
__host__ __device__ inline int unit_part(unsigned a, unsigned b, bool a_static, bool b_static, int nb)
{
    if (a_static || b_static) return -1;
    if (a / (unsigned)PART_BODIES == b / (unsigned)PART_BODIES) return (int)(a / (unsigned)PART_BODIES);
    const unsigned h = (unsigned)PART_BODIES / 2u;
    if ((a + h) / (unsigned)PART_BODIES == (b + h) / (unsigned)PART_BODIES) return parts_per_level(nb) + (int)((a + h) / (unsigned)PART_BODIES);
    return -1;
}

__host__ __device__ inline int colour_pick_two_ended(unsigned long long used_mask, int k_limit, bool from_top)
{
    const unsigned long long free_mask = ~used_mask;
    if (!from_top) return free_mask ? __builtin_ctzll(free_mask) : -1;
    const unsigned long long below = k_limit >= 64 ? ~0ull : ((1ull << (k_limit > 0 ? k_limit : 0)) - 1ull);
    if (free_mask & below) return 63 - __builtin_clzll(free_mask & below);
    return (free_mask & ~below) ? __builtin_ctzll(free_mask & ~below) : -1;
}

struct Schedule {
    std::vector<int> order;               // slot -> joint
    std::vector<int> colour_offsets;      // ncolours + 1, over all groups
    std::vector<int> group_offsets;       // ngroups + 1 (slots)
    std::vector<int> group_first_colour;  // ngroups + 1 (indices into colour_offsets)
    int lds_groups = 0;                   // leading groups solved out of LDS; the rest (0 or 1 group) out of HBM
    int lds_lanes = 0;                    // joint capacity the LDS groups were binned for (selects the kernel shape)
    // per LDS group g: its bodies = group_bodies[group_body_offsets[g] .. group_body_offsets[g+1])
    std::vector<int> group_body_offsets;
    std::vector<int> group_bodies;
    std::vector<uint32_t> slot_local;     // per slot of an LDS group: local body1 | local body2 << 16
    std::vector<uint8_t> slot_colour;     // per slot of an LDS group: colour index inside the group
    std::vector<int> hbm_bodies;          // bodies touched by the HBM group, ascending (the only ones it stages / writes back)
    int hbm_body_count = 0;               // = hbm_bodies.size() when the list is on the host; a device-built list lives in HBM only
    std::vector<int> hbm_colour_offsets;  // slots of the HBM group's classes (absolute), empty if there is no HBM group
    std::vector<int> hbm_class_leaders;   // per class of the HBM group: its leaders (the slots behind them are the followers)
    int hbm_interior_classes = 0;         // KI: the HBM group's leading classes that hold interior units of partitioned components only
    int hbm_interior_classes0 = 0;        // KI0: the level-0 ones among them (the first KI0)
    // the interior units by part (host-built schedules; the device builder leaves its tables in HBM): per part (parts_total) and
    // interior class (64 classes per part) the slot ranges {first, end} of its leaders with a follower and {first, end} of its
    // single leaders, and the number of interior units before each part (parts_total + 1)
    std::vector<int> part_ranges, part_begin;
    // units of the LDS groups, in class order: slots of a unit's leader and follower (-1: none); group g's units are
    // [group_unit_offsets[g], group_unit_offsets[g + 1])
    std::vector<int> group_unit_offsets, unit_leader, unit_follower;
    std::vector<int> unit_lane;           // ... and the unit's lane in the island kernel (LANES above)
    // A schedule built on the device keeps the LDS groups' order / colours in HBM only; `lds_on_host` says whether
    // order[], colour_offsets[], group_first_colour[] above already cover the LDS groups (DeviceSolver::materialise).
    bool lds_on_host = true;
    int lds_colours = 0;                  // sum of the LDS groups' colour counts
    int island_count = 1, island_max_size = 0;   // GatherIslands' published numbers (ref: Solver.h:105-106)
    unsigned long long fingerprint = 0;
    bool valid = false, islands = false;
    int ngroups() const { return (int)group_offsets.size() - 1; }
    int ncolours() const { return lds_on_host ? (int)colour_offsets.size() - 1 : lds_colours + std::max(0, (int)hbm_colour_offsets.size() - 1); }
    bool has_hbm_group() const { return hbm_colour_offsets.size() > 1; }
    int hbm_begin() const { return has_hbm_group() ? hbm_colour_offsets.front() : 0; }
    int hbm_end() const { return has_hbm_group() ? hbm_colour_offsets.back() : 0; }
};

struct LdsCaps { int max_joints = 512, max_units = 256, max_bodies = 768, max_colours = 64, max_static = 1 << 30; };
// (an LDS group's local body table lists its static bodies first, so a static body's local index is also its
//  slot in the group's small static-tag table)

// Colouring rule (every group, host and device builders alike; stated in full above): UNITS take their class FIRST-FIT IN ORDER
// OF DECREASING colour_priority(id, joint index, lower body index) of their leader — parity of the lower body first, then a fixed
// pseudo-random order.  Sequentially that is one pass over the sorted units; in parallel it is a dependency graph a few rounds deep
// (a unit takes its class once every higher-priority unit on its dynamic bodies has one), where joint-index order would need one
// round per body of a stacked column.  The layout of a class: the leaders that have a follower, the single leaders, the followers in their leaders' order.

// One HBM group holding every joint.  `prio_id` (optional, per joint): priority ids; the joint index itself if null.
void build_colour_schedule(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb, Schedule& out,
                           const int* prio_id = nullptr);

// Island-aware schedule: connected components binned into LDS groups where they fit `caps`, the rest in one
// trailing HBM group.
// `big` (optional) is a roomier shape used for ALL groups when some component fits it but not `caps`.
void build_island_schedule(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb,
                           const LdsCaps& caps, Schedule& out, const LdsCaps* big = nullptr, const int* prio_id = nullptr);

// partner[j] = the other joint of j's unit or -1 (the unit rule above)
void find_partners(const int* body1, const int* body2, int nj, const int* prio_id, std::vector<int>& partner);

// Solver::GatherIslands semantics (ref: Solver.cpp:285-454): per-joint coalesced island id (-1 for
// static-static joints) and per-island joint counts.
void gather_islands(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb,
                    std::vector<int>& joint_island, std::vector<int>& island_size);

} // namespace phx
