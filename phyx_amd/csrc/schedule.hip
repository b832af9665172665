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
static void priority_order(const std::vector<int>& joints, const int* prio_id, std::vector<int>& perm)
{
    std::vector<std::pair<unsigned long long, int>> keyed(joints.size());
    for (size_t k = 0; k < joints.size(); ++k)
        keyed[k] = {colour_priority((unsigned)(prio_id ? prio_id[joints[k]] : joints[k]), (unsigned)joints[k]), (int)k};
    std::sort(keyed.begin(), keyed.end(), [](const auto& a, const auto& b) { return a.first > b.first; });
    perm.resize(joints.size());
    for (size_t k = 0; k < joints.size(); ++k) perm[k] = keyed[k].second;
}

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

static int colour_joints(const std::vector<int>& joints, const int* body1, const int* body2, const unsigned char* is_static,
                         int nb, std::vector<int>& colour, ColourScratch& sc, const int* prio_id)
{
    colour.assign(joints.size(), 0);
    std::vector<int> perm;
    priority_order(joints, prio_id, perm);
    for (;;) {
        sc.ensure(nb);
        const int words = sc.words;
        unsigned long long* used = sc.used.data();
        bool overflow = false;
        int ncolours = 0;
        size_t done = 0;
        for (size_t i = 0; i < joints.size(); ++i) {
            const size_t k = (size_t)perm[i];
            const int a = body1[joints[k]], b = body2[joints[k]];
            const bool da = !is_static[a], db = !is_static[b];
            int c = -1;
            for (int w = 0; w < words; ++w) {
                unsigned long long m = 0;
                if (da) m |= used[(size_t)a * words + w];
                if (db) m |= used[(size_t)b * words + w];
                if (~m) { c = w * 64 + __builtin_ctzll(~m); break; }
            }
            if (c < 0) { overflow = true; break; }
            colour[k] = c;
            if (da) used[(size_t)a * words + c / 64] |= 1ull << (c % 64);
            if (db) used[(size_t)b * words + c / 64] |= 1ull << (c % 64);
            ncolours = std::max(ncolours, c + 1);
            done = i + 1;
        }
        for (size_t i = 0; i < done; ++i)                      // leave the scratch clean for the next caller
            for (int body : {body1[joints[perm[i]]], body2[joints[perm[i]]]})
                for (int w = 0; w < words; ++w) used[(size_t)body * words + w] = 0ull;
        if (!overflow) return ncolours;
        sc.words *= 2;                                         // > 64 * words colours needed: widen the masks and redo
        sc.used.clear();
    }
}

// append one group made of `joints` coloured by `colour` (stable counting sort by colour)
static void append_group(Schedule& out, const std::vector<int>& joints, const std::vector<int>& colour, int ncolours)
{
    const int base = (int)out.order.size();
    std::vector<int> count(ncolours + 1, 0);
    for (int c : colour) count[c + 1]++;
    for (int c = 0; c < ncolours; ++c) count[c + 1] += count[c];
    out.order.resize(base + joints.size());
    std::vector<int> cursor(count.begin(), count.end() - 1);
    for (size_t k = 0; k < joints.size(); ++k) out.order[base + cursor[colour[k]]++] = joints[k];
    for (int c = 0; c < ncolours; ++c) out.colour_offsets.push_back(base + count[c + 1]);
    out.group_offsets.push_back(base + (int)joints.size());
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

void build_colour_schedule(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb, Schedule& out, const int* prio_id)
{
    reset(out);
    std::vector<int> all(nj), colour;
    for (int j = 0; j < nj; ++j) all[j] = j;
    ColourScratch scratch;
    const int ncol = colour_joints(all, body1, body2, is_static, nb, colour, scratch, prio_id);
    if (nj) {
        append_group(out, all, colour, ncol);
        out.hbm_colour_offsets.assign(out.colour_offsets.begin(), out.colour_offsets.end());
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
    std::vector<int> order;               // joints, colour-major
    std::vector<int> colour_sizes;        // joints per colour
    std::vector<uint32_t> slot_local;
    std::vector<uint8_t> slot_colour;
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
               const LdsCaps& caps, LocalMap& map, BinOut& out, const int* prio_id)
{
    out = BinOut{};
    map.reset((unsigned)joints.size() * 2u + 8u);
    // local body table: static bodies first (their local index doubles as the slot in the group's static-tag table)
    std::vector<int> dynamic;
    for (int pass = 0; pass < 2; ++pass)
        for (int j : joints)
            for (int b : {body1[j], body2[j]}) {
                if ((is_static[b] != 0) != (pass == 0)) continue;
                bool fresh;
                int* slot = map.find_or_insert(b, fresh);
                if (fresh) { *slot = (int)out.bodies.size(); out.bodies.push_back(b); }
            }
    int nstatic = 0;
    for (int b : out.bodies) nstatic += is_static[b] ? 1 : 0;
    if ((int)out.bodies.size() > caps.max_bodies || (int)out.bodies.size() > 65535 || nstatic > caps.max_static) { out.rejected = true; return; }
    // first-fit colouring in priority order on per-local-body masks
    const int words = (caps.max_colours + 63) / 64;
    std::vector<unsigned long long> used(out.bodies.size() * (size_t)words, 0ull);
    std::vector<int> colour(joints.size());
    std::vector<uint32_t> local(joints.size());
    std::vector<int> perm;
    priority_order(joints, prio_id, perm);
    int ncol = 0;
    for (size_t i = 0; i < joints.size(); ++i) {
        const size_t k = (size_t)perm[i];
        bool fresh;
        const int a = *map.find_or_insert(body1[joints[k]], fresh), b = *map.find_or_insert(body2[joints[k]], fresh);
        const bool da = !is_static[body1[joints[k]]], db = !is_static[body2[joints[k]]];
        int c = -1;
        for (int w = 0; w < words && c < 0; ++w) {
            unsigned long long m = 0;
            if (da) m |= used[(size_t)a * words + w];
            if (db) m |= used[(size_t)b * words + w];
            if (~m) c = w * 64 + __builtin_ctzll(~m);
        }
        if (c < 0 || c >= caps.max_colours) { out.rejected = true; return; }
        if (da) used[(size_t)a * words + c / 64] |= 1ull << (c % 64);
        if (db) used[(size_t)b * words + c / 64] |= 1ull << (c % 64);
        colour[k] = c;
        local[k] = (uint32_t)a | ((uint32_t)b << 16);
        ncol = std::max(ncol, c + 1);
    }
    // stable counting sort by colour
    out.colour_sizes.assign(ncol, 0);
    for (int c : colour) out.colour_sizes[c]++;
    std::vector<int> cursor(ncol, 0);
    for (int c = 1; c < ncol; ++c) cursor[c] = cursor[c - 1] + out.colour_sizes[c - 1];
    out.order.resize(joints.size()); out.slot_local.resize(joints.size()); out.slot_colour.resize(joints.size());
    for (size_t k = 0; k < joints.size(); ++k) {
        const int at = cursor[colour[k]]++;
        out.order[at] = joints[k]; out.slot_local[at] = local[k]; out.slot_colour[at] = (uint8_t)colour[k];
    }
}

} // namespace

void build_island_schedule(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb,
                           const LdsCaps& small_caps, Schedule& out, const LdsCaps* big, const int* prio_id)
{
    reset(out);
    out.islands = true;
    std::vector<int> root, number;
    const int ncomp = components(body1, body2, nj, is_static, nb, root, number);
    // joints per component, in joint order (CSR)
    std::vector<int> comp_of(nj), comp_count(ncomp + 1, 0);
    for (int j = 0; j < nj; ++j) {
        const int a = body1[j], b = body2[j];
        comp_of[j] = (is_static[a] && is_static[b]) ? -1 : number[root[is_static[a] ? b : a]];
        if (comp_of[j] >= 0) comp_count[comp_of[j] + 1]++;
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
    LdsCaps caps = small_caps;
    if (big) {
        int need = 0;
        for (int c = 0; c < ncomp; ++c) { const int n = comp_count[c + 1] - comp_count[c]; if (n > small_caps.max_joints && n <= big->max_joints) need = std::max(need, n); }
        if (need) caps = *big;
    }
    out.lds_lanes = caps.max_joints;
    // greedy binning of consecutive components (serial, one pass); oversized components go to the HBM group whole
    std::vector<int> rest;
    std::vector<std::pair<int, int>> bins;       // [first component, last component)
    {
        int begin = -1, size = 0;
        auto flush = [&](int end) { if (begin >= 0 && size > 0) bins.emplace_back(begin, end); begin = -1; size = 0; };
        for (int c = 0; c < ncomp; ++c) {
            const int n = comp_count[c + 1] - comp_count[c];
            if (n == 0) continue;
            if (n > caps.max_joints) {
                flush(c);
                rest.insert(rest.end(), comp_joints.begin() + comp_count[c], comp_joints.begin() + comp_count[c + 1]);
                continue;
            }
            if (size + n > caps.max_joints) flush(c);
            if (begin < 0) begin = c;
            size += n;
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
            build_bin(joints, body1, body2, is_static, caps, map, built[b], prio_id);
            if (built[b].rejected) built[b].order = joints;
        }
    });
    // concatenate
    size_t total_slots = 0, total_bodies = 0;
    for (const BinOut& b : built) if (!b.rejected) { total_slots += b.order.size(); total_bodies += b.bodies.size(); }
    out.order.reserve(nj); out.slot_local.reserve(total_slots); out.slot_colour.reserve(total_slots); out.group_bodies.reserve(total_bodies);
    for (const BinOut& b : built) {
        if (b.rejected) { rest.insert(rest.end(), b.order.begin(), b.order.end()); continue; }
        const int base = (int)out.order.size();
        out.order.insert(out.order.end(), b.order.begin(), b.order.end());
        out.slot_local.insert(out.slot_local.end(), b.slot_local.begin(), b.slot_local.end());
        out.slot_colour.insert(out.slot_colour.end(), b.slot_colour.begin(), b.slot_colour.end());
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
        std::vector<int> colour;
        ColourScratch scratch;
        const int ncol = colour_joints(rest, body1, body2, is_static, nb, colour, scratch, prio_id);
        const size_t first = out.colour_offsets.size() - 1;
        append_group(out, rest, colour, ncol);
        out.hbm_colour_offsets.assign(out.colour_offsets.begin() + first, out.colour_offsets.end());
        touched_bodies(rest, body1, body2, nb, out.hbm_bodies);
        out.hbm_body_count = (int)out.hbm_bodies.size();
    }
}

} // namespace phx
