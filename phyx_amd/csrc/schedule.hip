// schedule.hip — host code only (kept in a .hip so the whole library goes through one compiler).
#include "schedule.h"

#include <algorithm>
#include <atomic>
#include <thread>

namespace phx {

// run fn(begin, end) over [0, count) in chunks on the host's cores (the caller takes a share too)
template <typename F>
static void parallel_chunks(int count, int min_per_thread, F&& fn)
{
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    const int threads = std::max(1, std::min({hw, 32, count / std::max(min_per_thread, 1)}));
    if (threads <= 1) { if (count > 0) fn(0, count); return; }
    std::atomic<int> next{0};
    const int chunk = std::max(1, count / (threads * 4));
    auto work = [&] { for (;;) { const int b = next.fetch_add(chunk); if (b >= count) return; fn(b, std::min(count, b + chunk)); } };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
}

// positions of `joints` in colouring order: decreasing colour_priority (schedule.h)
static void priority_order(const std::vector<int>& joints, const int* prio_id, const int* body1, const int* body2, std::vector<int>& perm)
{
    std::vector<std::pair<unsigned long long, int>> keyed(joints.size());
    for (size_t k = 0; k < joints.size(); ++k)
        keyed[k] = {colour_priority((unsigned)(prio_id ? prio_id[joints[k]] : joints[k]), (unsigned)joints[k], (unsigned)std::min(body1[joints[k]], body2[joints[k]])), (int)k};
    std::sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    perm.resize(joints.size());
    for (size_t k = 0; k < joints.size(); ++k) perm[k] = keyed[k].second;
}

// Units (schedule.h): partner[j] = the other joint of j's unit, or -1.  Two joints form a unit when their priority ids differ
// in the lowest bit only (the two contact points of one manifold: ids 2m and 2m + 1), their bodies are the same, and each is
// the smallest-index joint carrying its id (ids are unique in a World; the rule keeps foreign input deterministic).
void find_partners(const int* body1, const int* body2, int nj, const int* prio_id, std::vector<int>& partner)
{
    partner.assign((size_t)std::max(nj, 0), -1);
    auto id_of = [&](int j) { return prio_id ? prio_id[j] : j; };
    int max_id = -1;
    for (int j = 0; j < nj; ++j) max_id = std::max(max_id, id_of(j));
    if (max_id < 0) return;
    std::vector<int> first((size_t)max_id + 2, -1);       // id -> smallest joint index carrying it
    for (int j = nj - 1; j >= 0; --j) if (id_of(j) >= 0) first[id_of(j)] = j;
    for (int j = 0; j < nj; ++j) {
        const int id = id_of(j);
        if (id < 0 || first[id] != j) continue;
        const int other = first[id ^ 1];
        if (other < 0 || body1[other] != body1[j] || body2[other] != body2[j]) continue;
        partner[j] = other;
    }
}

static inline bool is_follower(const std::vector<int>& partner, const int* prio_id, int j) { return partner[j] >= 0 && ((prio_id ? prio_id[j] : j) & 1) != 0; }

// first-fit colouring of `joints` (indices into body1/body2), taken in priority order; returns colour per entry.
// Colouring in decreasing-priority order is what Jones-Plassmann rounds compute in parallel (a joint takes its colour
// once it holds the highest priority among the uncoloured joints on both its dynamic bodies) — that is how the device
// builder reaches the same colours in ~log n rounds instead of one serial pass (schedule_kernels.h).
// `used` is caller-owned scratch (nb * words 64-bit masks, all zero on entry and on exit) so that colouring a
// thousand small bins does not allocate or clear a world-sized array a thousand times.
struct ColourScratch {
    std::vector<unsigned long long> used;
    int words = 1;
    void ensure(int nb) { if (used.size() < (size_t)nb * words) used.assign((size_t)nb * words, 0ull); }
};

// `joints`: the LEADERS of the units to colour (schedule.h); `comp` (per entry): connected component of the joint (any labels;
// joints between two static bodies may carry -1).
static int colour_joints(const std::vector<int>& joints, const int* body1, const int* body2, const unsigned char* is_static,
                         int nb, std::vector<int>& colour, ColourScratch& sc, const int* prio_id, const std::vector<int>& comp,
                         const std::vector<int>& partner, int* interior_classes = nullptr, std::vector<int>* part_of = nullptr)
{
    // interior_classes (optional): {KI, KI0} of the group (schedule.h); part_of (optional): per entry the part it is interior to, or -1
    colour.assign(joints.size(), 0);
    if (interior_classes) { interior_classes[0] = 0; interior_classes[1] = 0; }
    std::vector<int> perm;
    priority_order(joints, prio_id, body1, body2, perm);
    // components, densely numbered; the big ones are PARTITIONED: their interior units form a kind of their own (schedule.h)
    std::vector<unsigned char> comp_bad;                       // per dense component (also set for components too big for B)
    std::vector<int> dense(joints.size());
    std::vector<unsigned char> interior(joints.size(), 0);      // 0: a rest unit; 1 + level: interior at that level
    std::vector<int> upart(joints.size(), -1);
    bool any_interior = false;
    const int P = parts_per_level(nb);
    {
        std::vector<std::pair<int, int>> keyed(joints.size());
        for (size_t k = 0; k < joints.size(); ++k) keyed[k] = {comp[k], (int)k};
        std::sort(keyed.begin(), keyed.end());
        int count = 0;
        for (size_t i = 0; i < keyed.size(); ++i) {
            if (i == 0 || keyed[i].first != keyed[i - 1].first || keyed[i].first < 0) ++count;     // every static-static joint is its own class
            dense[keyed[i].second] = count - 1;
        }
        comp_bad.assign(count, 0);
        std::vector<int> size(count, 0);
        for (size_t k = 0; k < joints.size(); ++k) size[dense[k]] += partner[joints[k]] >= 0 ? 2 : 1;      // joints of the component, not units
        for (int d = 0; d < count; ++d) if (size[d] > COLOUR_B_MAX_JOINTS) comp_bad[d] = 1;      // B is not attempted there (schedule.h)
        for (size_t k = 0; k < joints.size(); ++k) {
            const int a = body1[joints[k]], b = body2[joints[k]];
            if (comp[k] >= 0 && size[dense[k]] > COLOUR_B_MAX_JOINTS) {
                upart[k] = unit_part((unsigned)a, (unsigned)b, is_static[a] != 0, is_static[b] != 0, nb);
                if (upart[k] >= 0) { interior[k] = upart[k] < P ? 1 : 2; any_interior = true; }
            }
        }
        for (size_t k = 0; k < joints.size(); ++k) if (comp[k] < 0) comp_bad[dense[k]] = 1;      // static-static joints: colour 0 either way
    }
    // candidate A: smallest free colour (masks widen beyond 64 colours on demand).  Interior units keep masks of their own, per
    // level: rows nb .. 3 nb of the scratch
    std::vector<int> col_a(joints.size(), 0);
    const int rows = any_interior ? 3 * nb : nb;
    for (;;) {
        sc.ensure(rows);
        const int words = sc.words;
        unsigned long long* used = sc.used.data();
        bool overflow = false;
        size_t done = 0;
        for (size_t i = 0; i < joints.size(); ++i) {
            const size_t k = (size_t)perm[i];
            const int shift = interior[k] * nb;
            const int a = body1[joints[k]] + shift, b = body2[joints[k]] + shift;
            const bool da = !is_static[a - shift], db = !is_static[b - shift];
            int c = -1;
            for (int w = 0; w < words; ++w) {
                unsigned long long m = 0;
                if (da) m |= used[(size_t)a * words + w];
                if (db) m |= used[(size_t)b * words + w];
                if (~m) { c = w * 64 + __builtin_ctzll(~m); break; }
            }
            if (c < 0) { overflow = true; break; }
            col_a[k] = c;
            if (da) used[(size_t)a * words + c / 64] |= 1ull << (c % 64);
            if (db) used[(size_t)b * words + c / 64] |= 1ull << (c % 64);
            done = i + 1;
        }
        for (size_t i = 0; i < done; ++i) {                    // leave the scratch clean for the next caller
            const int shift = interior[perm[i]] * nb;
            for (int body : {body1[joints[perm[i]]] + shift, body2[joints[perm[i]]] + shift})
                for (int w = 0; w < words; ++w) used[(size_t)body * words + w] = 0ull;
        }
        if (!overflow) break;
        sc.words *= 2;                                         // > 64 * words colours needed: widen the masks and redo
        sc.used.clear();
    }
    // candidate B: two-ended (schedule.h), one 64-bit mask per body; a component it cannot colour within 64 colours keeps A
    std::vector<int> col_b(joints.size(), 0);
    {
        sc.ensure(nb);
        const int words = sc.words;
        unsigned long long* used = sc.used.data();             // word 0 of every body's mask row
        std::vector<int> degree_of;                            // per touched body, via the same rows: count in a side map
        std::vector<int> deg(nb, 0);
        for (int j : joints) { deg[body1[j]]++; deg[body2[j]]++; }
        for (size_t i = 0; i < joints.size(); ++i) {
            const size_t k = (size_t)perm[i];
            const int a = body1[joints[k]], b = body2[joints[k]];
            const bool da = !is_static[a], db = !is_static[b];
            unsigned long long m = 0;
            if (da) m |= used[(size_t)a * words];
            if (db) m |= used[(size_t)b * words];
            const int klim = std::max(da ? deg[a] : 0, db ? deg[b] : 0);
            if (comp_bad[dense[k]]) continue;
            const int c = colour_pick_two_ended(m, klim, (std::min(a, b) & 1) != 0);
            if (c < 0) { comp_bad[dense[k]] = 1; continue; }
            col_b[k] = c;
            if (da) used[(size_t)a * words] |= 1ull << c;
            if (db) used[(size_t)b * words] |= 1ull << c;
        }
        for (int j : joints) { used[(size_t)body1[j] * words] = 0ull; used[(size_t)body2[j] * words] = 0ull; }
    }
    // per component: colours in use under either candidate, the choice, the dense renumbering
    const size_t ncomp = comp_bad.size();
    std::vector<int> max_a(ncomp, -1);
    std::vector<unsigned long long> seen_a(ncomp, 0ull), seen_b(ncomp, 0ull);
    int ki0 = 0, ki1 = 0;                                      // interior classes of the group per level (first fit leaves no gaps inside a component and kind)
    for (size_t k = 0; k < joints.size(); ++k) {
        if (interior[k] == 1) { ki0 = std::max(ki0, col_a[k] + 1); continue; }
        if (interior[k] == 2) { ki1 = std::max(ki1, col_a[k] + 1); continue; }
        max_a[dense[k]] = std::max(max_a[dense[k]], col_a[k]);
        if (col_a[k] < 64) seen_a[dense[k]] |= 1ull << col_a[k];
        seen_b[dense[k]] |= 1ull << col_b[k];
    }
    const int ki = ki0 + ki1;
    int ncolours = ki;
    for (size_t k = 0; k < joints.size(); ++k) {
        if (interior[k]) { colour[k] = (interior[k] == 2 ? ki0 : 0) + col_a[k]; continue; }
        const int d = dense[k];
        const int count_a = max_a[d] + 1;                      // candidate A leaves no gaps inside a component
        const bool use_b = !comp_bad[d] && __builtin_popcountll(seen_b[d]) < count_a;
        colour[k] = ki + (use_b ? __builtin_popcountll(seen_b[d] & ((1ull << col_b[k]) - 1ull)) : col_a[k]);
        ncolours = std::max(ncolours, colour[k] + 1);
    }
    if (interior_classes) { interior_classes[0] = ki; interior_classes[1] = ki0; }
    if (part_of) *part_of = upart;
    return ncolours;
}

// the interior classes of the HBM group by part (schedule.h): an interior class is laid out part by part, so a part's units of a
// class are two runs of slots — its leaders with a follower, its single leaders
static void build_part_tables(Schedule& out, const int* body1, const int* body2, const unsigned char* is_static, int nb, const std::vector<int>& partner)
{
    out.part_ranges.clear(); out.part_begin.clear();
    const int ki = out.hbm_interior_classes;
    if (ki <= 0 || ki > 64) return;
    const int parts = parts_total(nb);
    out.part_ranges.assign((size_t)parts * 64 * 4, 0);
    out.part_begin.assign((size_t)parts + 1, 0);
    for (int c = 0; c < ki; ++c) {
        const int cb = out.hbm_colour_offsets[c], lead = out.hbm_class_leaders[c];
        for (int s = cb; s < cb + lead; ++s) {
            const int j = out.order[s], kind = partner[j] >= 0 ? 0 : 1;
            const int part = unit_part((unsigned)body1[j], (unsigned)body2[j], is_static[body1[j]] != 0, is_static[body2[j]] != 0, nb);
            int* row = &out.part_ranges[((size_t)part * 64 + c) * 4 + 2 * kind];
            if (row[1] == 0) row[0] = s;
            row[1] = s + 1;
            out.part_begin[part + 1]++;
        }
    }
    for (int p = 0; p < parts; ++p) out.part_begin[p + 1] += out.part_begin[p];
}

// append one group made of the units led by `leaders`, coloured by `colour`: class by class, the leaders that have a follower
// (joint order), the single leaders (joint order), then the followers in their leaders' order (schedule.h)
// (interior classes — `interior_classes` leading ones, schedule.h — are laid out part by part: leaders ordered by (part, joint),
//  so that the workgroup that sweeps a part reads its units' constants from consecutive slots)
static void append_group(Schedule& out, const std::vector<int>& leaders_in, const std::vector<int>& colour_in, int ncolours, const std::vector<int>& partner,
                         std::vector<int>* class_leaders, int interior_classes = 0, const std::vector<int>* unit_parts = nullptr)
{
    std::vector<int> leaders_sorted, colour_sorted;
    if (interior_classes > 0) {
        std::vector<int> idx(leaders_in.size());
        for (size_t k = 0; k < idx.size(); ++k) idx[k] = (int)k;
        auto part_of = [&](int k) { return colour_in[k] < interior_classes ? (*unit_parts)[k] : 0; };
        std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return part_of(x) < part_of(y); });
        leaders_sorted.resize(idx.size()); colour_sorted.resize(idx.size());
        for (size_t k = 0; k < idx.size(); ++k) { leaders_sorted[k] = leaders_in[idx[k]]; colour_sorted[k] = colour_in[idx[k]]; }
    }
    const std::vector<int>& leaders = interior_classes > 0 ? leaders_sorted : leaders_in;
    const std::vector<int>& colour = interior_classes > 0 ? colour_sorted : colour_in;
    const int base = (int)out.order.size();
    std::vector<int> with(ncolours, 0), single(ncolours, 0);
    for (size_t k = 0; k < leaders.size(); ++k) (partner[leaders[k]] >= 0 ? with : single)[colour[k]]++;
    std::vector<int> begin(ncolours + 1, base);
    for (int c = 0; c < ncolours; ++c) begin[c + 1] = begin[c] + 2 * with[c] + single[c];
    out.order.resize(begin[ncolours]);
    std::vector<int> cur_with(ncolours, 0), cur_single(ncolours, 0);
    for (size_t k = 0; k < leaders.size(); ++k) {
        const int c = colour[k], j = leaders[k];
        if (partner[j] >= 0) {
            const int i = cur_with[c]++;
            out.order[begin[c] + i] = j;
            out.order[begin[c] + with[c] + single[c] + i] = partner[j];
        } else out.order[begin[c] + with[c] + cur_single[c]++] = j;
    }
    for (int c = 0; c < ncolours; ++c) {
        out.colour_offsets.push_back(begin[c + 1]);
        if (class_leaders) class_leaders->push_back(with[c] + single[c]);
    }
    out.group_offsets.push_back(begin[ncolours]);
    out.group_first_colour.push_back((int)out.colour_offsets.size() - 1);
}

static void reset(Schedule& out)
{
    out = Schedule{};
    out.colour_offsets.assign(1, 0);
    out.group_offsets.assign(1, 0);
    out.group_first_colour.assign(1, 0);
    out.group_body_offsets.assign(1, 0);
}

// bodies touched by `joints`, ascending
static void touched_bodies(const std::vector<int>& joints, const int* body1, const int* body2, int nb, std::vector<int>& out)
{
    std::vector<unsigned char> seen(nb, 0);
    for (int j : joints) { seen[body1[j]] = 1; seen[body2[j]] = 1; }
    out.clear();
    for (int b = 0; b < nb; ++b) if (seen[b]) out.push_back(b);
}

static int components(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb, std::vector<int>& root, std::vector<int>& number);

void build_colour_schedule(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb, Schedule& out, const int* prio_id)
{
    reset(out);
    std::vector<int> partner;
    find_partners(body1, body2, nj, prio_id, partner);
    std::vector<int> all(nj), leaders, colour;
    for (int j = 0; j < nj; ++j) { all[j] = j; if (!is_follower(partner, prio_id, j)) leaders.push_back(j); }
    ColourScratch scratch;
    std::vector<int> root, number, comp(leaders.size());
    components(body1, body2, nj, is_static, nb, root, number);
    for (size_t k = 0; k < leaders.size(); ++k) {
        const int j = leaders[k];
        comp[k] = (is_static[body1[j]] && is_static[body2[j]]) ? -1 : number[root[is_static[body1[j]] ? body2[j] : body1[j]]];
    }
    int ki[2] = {0, 0};
    std::vector<int> unit_parts;
    const int ncol = colour_joints(leaders, body1, body2, is_static, nb, colour, scratch, prio_id, comp, partner, ki, &unit_parts);
    out.hbm_interior_classes = ki[0]; out.hbm_interior_classes0 = ki[1];
    if (nj) {
        append_group(out, leaders, colour, ncol, partner, &out.hbm_class_leaders, out.hbm_interior_classes, &unit_parts);
        out.hbm_colour_offsets.assign(out.colour_offsets.begin(), out.colour_offsets.end());
        build_part_tables(out, body1, body2, is_static, nb, partner);
    }
    out.lds_groups = 0;
    out.islands = false;
    touched_bodies(all, body1, body2, nb, out.hbm_bodies);
    out.hbm_body_count = (int)out.hbm_bodies.size();
}

static int uf_find(std::vector<int>& t, int i)
{
    int r = i;
    while (r != t[r]) r = t[r];
    while (t[i] != r) { int n = t[i]; t[i] = r; i = n; }
    return r;
}

// connected components over dynamic bodies, numbered in body order (ref: Solver.cpp:302-356)
static int components(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb,
                      std::vector<int>& root, std::vector<int>& number)
{
    root.resize(nb);
    for (int i = 0; i < nb; ++i) root[i] = is_static[i] ? -1 : i;
    for (int j = 0; j < nj; ++j) {
        const int a = body1[j], b = body2[j];
        if (is_static[a] || is_static[b]) continue;
        const int ra = uf_find(root, a), rb = uf_find(root, b);
        root[ra] = rb;
    }
    number.assign(nb, -1);
    int count = 0;
    for (int i = 0; i < nb; ++i) {
        if (root[i] < 0) continue;
        const int r = uf_find(root, i);
        root[i] = r;                                   // fully compressed: root[i] is the representative from here on
        if (number[r] < 0) number[r] = count++;
    }
    return count;
}

void gather_islands(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb,
                    std::vector<int>& joint_island, std::vector<int>& island_size)
{
    std::vector<int> root, number;
    const int count = components(body1, body2, nj, is_static, nb, root, number);
    std::vector<int> raw_size(count, 0);
    auto island_of = [&](int j) {
        const int a = body1[j], b = body2[j];
        if (is_static[a] && is_static[b]) return -1;
        return number[root[is_static[a] ? b : a]];
    };
    for (int j = 0; j < nj; ++j) { const int i = island_of(j); if (i >= 0) raw_size[i]++; }
    // coalesce consecutive islands until >= kIslandMinSize joints (ref: Solver.cpp:382-413)
    std::vector<int> merged(count, 0);
    island_size.clear();
    int run = 0;
    for (int i = 0; i < count; ++i) {
        run += raw_size[i];
        merged[i] = (int)island_size.size();
        if (run >= 256 || (run > 0 && i == count - 1)) { island_size.push_back(run); run = 0; }
    }
    joint_island.resize(nj);
    for (int j = 0; j < nj; ++j) { const int i = island_of(j); joint_island[j] = i < 0 ? -1 : merged[i]; }
}

// ---- island-aware builder ---------------------------------------------------------------------------------
// serial: components + greedy binning of consecutive components; parallel over bins: local body table + colouring
// + slot arrays (bins are independent); serial: concatenation.
namespace {

struct BinOut {
    std::vector<int> order;               // joints, class-major: leaders with a follower, single leaders, followers (schedule.h)
    std::vector<int> colour_sizes;        // joints per class
    std::vector<uint32_t> slot_local;
    std::vector<uint8_t> slot_colour;
    std::vector<int> unit_leader, unit_follower;   // per unit, in class order: its slots relative to the bin's first slot (follower: -1 if none)
    std::vector<int> unit_lane;                    // per unit: its lane in the island kernel (schedule.h LANES)
    std::vector<int> bodies;              // local body table, static first
    bool rejected = false;                // does not fit the caps: its joints go to the HBM group
};

// open-addressing map body id -> local index for one bin (a bin touches at most a few hundred bodies)
struct LocalMap {
    std::vector<int> key, val;
    unsigned mask = 0;
    void reset(unsigned want)
    {
        unsigned cap = 64;
        while (cap < 2 * want) cap <<= 1;
        if (key.size() != cap) { key.assign(cap, -1); val.assign(cap, 0); } else std::fill(key.begin(), key.end(), -1);
        mask = cap - 1;
    }
    int* find_or_insert(int k, bool& fresh)
    {
        unsigned p = ((unsigned)k * 2654435761u) & mask;
        while (key[p] != -1 && key[p] != k) p = (p + 1) & mask;
        fresh = key[p] == -1;
        key[p] = k;
        return &val[p];
    }
};

void build_bin(const std::vector<int>& joints, const int* body1, const int* body2, const unsigned char* is_static,
               const LdsCaps& caps, LocalMap& map, BinOut& out, const int* prio_id, const int* comp_of, const std::vector<int>& partner)
{
    out = BinOut{};
    map.reset((unsigned)joints.size() * 2u + 8u);
    // local body table: static bodies first (their local index doubles as the slot in the group's static-tag table), each kind in
    // the order the UNITS — in their leaders' joint order — first touch them (a follower shares its leader's bodies)
    for (int pass = 0; pass < 2; ++pass)
        for (int j : joints)
            for (int b : {body1[j], body2[j]}) {
                if (is_follower(partner, prio_id, j)) continue;
                if ((is_static[b] != 0) != (pass == 0)) continue;
                bool fresh;
                int* slot = map.find_or_insert(b, fresh);
                if (fresh) { *slot = (int)out.bodies.size(); out.bodies.push_back(b); }
            }
    int nstatic = 0;
    for (int b : out.bodies) nstatic += is_static[b] ? 1 : 0;
    if ((int)out.bodies.size() > caps.max_bodies || (int)out.bodies.size() > 65535 || nstatic > caps.max_static) { out.rejected = true; return; }
    // the units of the bin, represented by their leaders (joint order)
    std::vector<int> leaders;
    for (int j : joints) if (!is_follower(partner, prio_id, j)) leaders.push_back(j);
    if ((int)leaders.size() > caps.max_units) { out.rejected = true; return; }
    // two first-fit candidates in priority order on per-local-body masks (schedule.h): A = smallest free colour,
    // B = two-ended; keep the one with fewer colours
    if (caps.max_colours > 64) { out.rejected = true; return; }          // one 64-bit mask per body (the device builder's limit)
    std::vector<unsigned long long> used_a(out.bodies.size(), 0ull), used_b(out.bodies.size(), 0ull);
    std::vector<int> degree(out.bodies.size(), 0);
    std::vector<int> col_a(leaders.size()), col_b(leaders.size());
    std::vector<uint32_t> local(leaders.size());
    std::vector<int> perm;
    priority_order(leaders, prio_id, body1, body2, perm);
    for (size_t k = 0; k < leaders.size(); ++k) {
        bool fresh;
        const int a = *map.find_or_insert(body1[leaders[k]], fresh), b = *map.find_or_insert(body2[leaders[k]], fresh);
        local[k] = (uint32_t)a | ((uint32_t)b << 16);
        degree[a]++; degree[b]++;
    }
    // per component of the bin (components are consecutive numbers): colours in use under either candidate
    int comp_lo = 0x7fffffff, comp_hi = -1;
    for (int j : joints) { comp_lo = std::min(comp_lo, comp_of[j]); comp_hi = std::max(comp_hi, comp_of[j]); }
    const size_t span = joints.empty() ? 0 : (size_t)(comp_hi - comp_lo + 1);
    std::vector<unsigned long long> seen_a(span, 0ull), seen_b(span, 0ull);
    std::vector<unsigned char> bad_b(span, 0);
    for (size_t i = 0; i < leaders.size(); ++i) {
        const size_t k = (size_t)perm[i];
        const int a = (int)(local[k] & 0xFFFFu), b = (int)(local[k] >> 16);
        const int ga = body1[leaders[k]], gb = body2[leaders[k]];
        const bool da = !is_static[ga], db = !is_static[gb];
        const size_t comp = (size_t)(comp_of[leaders[k]] - comp_lo);
        unsigned long long ma = 0, mb = 0;
        if (da) { ma |= used_a[a]; mb |= used_b[a]; }
        if (db) { ma |= used_a[b]; mb |= used_b[b]; }
        const int ca = colour_pick_two_ended(ma, 0, false);
        if (ca < 0 || ca >= caps.max_colours) { out.rejected = true; return; }
        if (da) used_a[a] |= 1ull << ca;
        if (db) used_a[b] |= 1ull << ca;
        col_a[k] = ca; seen_a[comp] |= 1ull << ca;
        const int klim = std::max(da ? degree[a] : 0, db ? degree[b] : 0);
        const int cb = colour_pick_two_ended(mb, klim, (std::min(ga, gb) & 1) != 0);
        if (cb < 0 || cb >= caps.max_colours) { bad_b[comp] = 1; col_b[k] = 0; }
        else { if (da) used_b[a] |= 1ull << cb; if (db) used_b[b] |= 1ull << cb; col_b[k] = cb; seen_b[comp] |= 1ull << cb; }
    }
    int ncol = 0;
    std::vector<int> colour(leaders.size());
    for (size_t k = 0; k < leaders.size(); ++k) {                      // the component's choice, dense renumbering (increasing)
        const size_t comp = (size_t)(comp_of[leaders[k]] - comp_lo);
        const bool use_b = !bad_b[comp] && __builtin_popcountll(seen_b[comp]) < __builtin_popcountll(seen_a[comp]);
        const unsigned long long seen = use_b ? seen_b[comp] : seen_a[comp];
        const int c = use_b ? col_b[k] : col_a[k];
        colour[k] = __builtin_popcountll(seen & ((1ull << c) - 1ull));
        ncol = std::max(ncol, colour[k] + 1);
    }
    // class by class: leaders with a follower, single leaders, followers in their leaders' order (all in joint order)
    std::vector<int> with(ncol, 0), single(ncol, 0);
    for (size_t k = 0; k < leaders.size(); ++k) (partner[leaders[k]] >= 0 ? with : single)[colour[k]]++;
    std::vector<int> begin(ncol + 1, 0), unit_begin(ncol + 1, 0);
    for (int c = 0; c < ncol; ++c) { begin[c + 1] = begin[c] + 2 * with[c] + single[c]; unit_begin[c + 1] = unit_begin[c] + with[c] + single[c]; }
    out.colour_sizes.resize(ncol);
    for (int c = 0; c < ncol; ++c) out.colour_sizes[c] = begin[c + 1] - begin[c];
    out.order.resize(joints.size()); out.slot_local.resize(joints.size()); out.slot_colour.resize(joints.size());
    out.unit_leader.resize(leaders.size()); out.unit_follower.resize(leaders.size()); out.unit_lane.resize(leaders.size());
    std::vector<int> cur_with(ncol, 0), cur_single(ncol, 0);
    unsigned short units_c[64], lane_begin[64];
    for (int c = 0; c < ncol; ++c) units_c[c] = (unsigned short)(with[c] + single[c]);
    layout_classes(units_c, ncol, (int)leaders.size(), caps.max_units, lane_begin);
    for (size_t k = 0; k < leaders.size(); ++k) {
        const int c = colour[k], j = leaders[k];
        int at, unit;
        if (partner[j] >= 0) {
            const int i = cur_with[c]++;
            at = begin[c] + i; unit = unit_begin[c] + i;
            const int fat = begin[c] + with[c] + single[c] + i;
            out.order[fat] = partner[j]; out.slot_local[fat] = local[k]; out.slot_colour[fat] = (uint8_t)c;
            out.unit_follower[unit] = fat;
        } else {
            const int i = cur_single[c]++;
            at = begin[c] + with[c] + i; unit = unit_begin[c] + with[c] + i;
            out.unit_follower[unit] = -1;
        }
        out.order[at] = j; out.slot_local[at] = local[k]; out.slot_colour[at] = (uint8_t)c;
        out.unit_leader[unit] = at;
        out.unit_lane[unit] = lane_begin[c] + (unit - unit_begin[c]);
    }
}

} // namespace

void build_island_schedule(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb,
                           const LdsCaps& small_caps, Schedule& out, const LdsCaps* big, const int* prio_id)
{
    reset(out);
    out.islands = true;
    std::vector<int> partner;
    find_partners(body1, body2, nj, prio_id, partner);
    std::vector<int> root, number;
    const int ncomp = components(body1, body2, nj, is_static, nb, root, number);
    // joints per component, in joint order (CSR); units per component
    std::vector<int> comp_of(nj), comp_count(ncomp + 1, 0), comp_units(std::max(ncomp, 1), 0);
    for (int j = 0; j < nj; ++j) {
        const int a = body1[j], b = body2[j];
        comp_of[j] = (is_static[a] && is_static[b]) ? -1 : number[root[is_static[a] ? b : a]];
        if (comp_of[j] >= 0) { comp_count[comp_of[j] + 1]++; if (!is_follower(partner, prio_id, j)) comp_units[comp_of[j]]++; }
    }
    for (int c = 0; c < ncomp; ++c) comp_count[c + 1] += comp_count[c];
    std::vector<int> comp_joints(comp_count[ncomp]);
    {
        std::vector<int> cur(comp_count.begin(), comp_count.end() - 1);
        for (int j = 0; j < nj; ++j) if (comp_of[j] >= 0) comp_joints[cur[comp_of[j]]++] = j;
    }
    // GatherIslands' published numbers (ref: Solver.cpp:382-413 coalescing rule), from the same components
    {
        int run = 0, count = 0, mx = 0;
        for (int c = 0; c < ncomp; ++c) {
            run += comp_count[c + 1] - comp_count[c];
            if (run >= 256 || (run > 0 && c == ncomp - 1)) { ++count; mx = std::max(mx, run); run = 0; }
        }
        out.island_count = count; out.island_max_size = mx;
    }
    // pick the workgroup shape: the roomier one only if some component needs it and fits it
    auto fits = [&](int c, const LdsCaps& k) { return comp_count[c + 1] - comp_count[c] <= k.max_joints && comp_units[c] <= k.max_units; };
    LdsCaps caps = small_caps;
    if (big) {
        bool need = false;
        for (int c = 0; c < ncomp; ++c) if (comp_count[c + 1] > comp_count[c] && !fits(c, small_caps) && fits(c, *big)) need = true;
        if (need) caps = *big;
    }
    out.lds_lanes = caps.max_units;
    // greedy binning of consecutive components (serial, one pass); oversized components go to the HBM group whole
    std::vector<int> rest;
    std::vector<std::pair<int, int>> bins;       // [first component, last component)
    {
        int begin = -1, size = 0, units = 0;
        auto flush = [&](int end) { if (begin >= 0 && size > 0) bins.emplace_back(begin, end); begin = -1; size = 0; units = 0; };
        for (int c = 0; c < ncomp; ++c) {
            if (c % BIN_CHUNK == 0) flush(c);              // (schedule.h BINNING: a bin never spans a chunk boundary)
            const int n = comp_count[c + 1] - comp_count[c];
            if (n == 0) continue;
            if (!fits(c, caps)) {
                flush(c);
                rest.insert(rest.end(), comp_joints.begin() + comp_count[c], comp_joints.begin() + comp_count[c + 1]);
                continue;
            }
            if (size + n > caps.max_joints || units + comp_units[c] > caps.max_units) flush(c);
            if (begin < 0) begin = c;
            size += n; units += comp_units[c];
        }
        flush(ncomp);
    }
    // bins are independent: build them on the host's cores
    std::vector<BinOut> built(bins.size());
    parallel_chunks((int)bins.size(), 8, [&](int b0, int b1) {
        LocalMap map;
        std::vector<int> joints;
        for (int b = b0; b < b1; ++b) {
            joints.assign(comp_joints.begin() + comp_count[bins[b].first], comp_joints.begin() + comp_count[bins[b].second]);
            if (bins[b].second - bins[b].first > 1) std::sort(joints.begin(), joints.end());      // joint-index order inside the bin
            build_bin(joints, body1, body2, is_static, caps, map, built[b], prio_id, comp_of.data(), partner);
            if (built[b].rejected) built[b].order = joints;
        }
    });
    // concatenate
    size_t total_slots = 0, total_bodies = 0;
    for (const BinOut& b : built) if (!b.rejected) { total_slots += b.order.size(); total_bodies += b.bodies.size(); }
    out.order.reserve(nj); out.slot_local.reserve(total_slots); out.slot_colour.reserve(total_slots); out.group_bodies.reserve(total_bodies);
    out.group_unit_offsets.assign(1, 0);
    for (const BinOut& b : built) {
        if (b.rejected) { rest.insert(rest.end(), b.order.begin(), b.order.end()); continue; }
        const int base = (int)out.order.size();
        out.order.insert(out.order.end(), b.order.begin(), b.order.end());
        out.slot_local.insert(out.slot_local.end(), b.slot_local.begin(), b.slot_local.end());
        out.slot_colour.insert(out.slot_colour.end(), b.slot_colour.begin(), b.slot_colour.end());
        for (size_t u = 0; u < b.unit_leader.size(); ++u) {
            out.unit_leader.push_back(base + b.unit_leader[u]);
            out.unit_follower.push_back(b.unit_follower[u] < 0 ? -1 : base + b.unit_follower[u]);
        }
        out.unit_lane.insert(out.unit_lane.end(), b.unit_lane.begin(), b.unit_lane.end());
        out.group_unit_offsets.push_back((int)out.unit_leader.size());
        int at = base;
        for (int n : b.colour_sizes) { at += n; out.colour_offsets.push_back(at); }
        out.group_offsets.push_back(at);
        out.group_first_colour.push_back((int)out.colour_offsets.size() - 1);
        out.group_bodies.insert(out.group_bodies.end(), b.bodies.begin(), b.bodies.end());
        out.group_body_offsets.push_back((int)out.group_bodies.size());
        out.lds_groups++;
    }
    out.lds_colours = (int)out.colour_offsets.size() - 1;
    for (int j = 0; j < nj; ++j) if (comp_of[j] < 0) rest.push_back(j);
    if (!rest.empty()) {
        std::sort(rest.begin(), rest.end());
        std::vector<int> leaders, colour, rest_comp;
        for (int j : rest) if (!is_follower(partner, prio_id, j)) { leaders.push_back(j); rest_comp.push_back(comp_of[j]); }
        ColourScratch scratch;
        int ki[2] = {0, 0};
        std::vector<int> unit_parts;
        const int ncol = colour_joints(leaders, body1, body2, is_static, nb, colour, scratch, prio_id, rest_comp, partner, ki, &unit_parts);
        out.hbm_interior_classes = ki[0]; out.hbm_interior_classes0 = ki[1];
        const size_t first = out.colour_offsets.size() - 1;
        append_group(out, leaders, colour, ncol, partner, &out.hbm_class_leaders, out.hbm_interior_classes, &unit_parts);
        out.hbm_colour_offsets.assign(out.colour_offsets.begin() + first, out.colour_offsets.end());
        build_part_tables(out, body1, body2, is_static, nb, partner);
        touched_bodies(rest, body1, body2, nb, out.hbm_bodies);
        out.hbm_body_count = (int)out.hbm_bodies.size();
    }
}

} // namespace phx
