// schedule.hip — host code only (kept in a .hip so the whole library goes through one compiler).
#include "schedule.h"

#include <algorithm>

namespace phx {

// first-fit colouring of `joints` (indices into body1/body2) in the given order; returns colour per entry.
// `used` is caller-owned scratch (nb * words 64-bit masks, all zero on entry and on exit) so that colouring a
// thousand small bins does not allocate or clear a world-sized array a thousand times.
struct ColourScratch {
    std::vector<unsigned long long> used;
    int words = 1;
    void ensure(int nb) { if (used.size() < (size_t)nb * words) used.assign((size_t)nb * words, 0ull); }
};

static int colour_joints(const std::vector<int>& joints, const int* body1, const int* body2, const unsigned char* is_static,
                         int nb, std::vector<int>& colour, ColourScratch& sc)
{
    colour.assign(joints.size(), 0);
    for (;;) {
        sc.ensure(nb);
        const int words = sc.words;
        unsigned long long* used = sc.used.data();
        bool overflow = false;
        int ncolours = 0;
        size_t done = 0;
        for (size_t k = 0; k < joints.size(); ++k) {
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
            done = k + 1;
        }
        for (size_t k = 0; k < done; ++k)                      // leave the scratch clean for the next caller
            for (int body : {body1[joints[k]], body2[joints[k]]})
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

void build_colour_schedule(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb, Schedule& out)
{
    reset(out);
    std::vector<int> all(nj), colour;
    for (int j = 0; j < nj; ++j) all[j] = j;
    ColourScratch scratch;
    const int ncol = colour_joints(all, body1, body2, is_static, nb, colour, scratch);
    if (nj) append_group(out, all, colour, ncol);
    out.lds_groups = 0;
    out.islands = false;
    std::vector<unsigned char> seen(nb, 0);
    for (int j = 0; j < nj; ++j)
        for (int b : {body1[j], body2[j]})
            if (!seen[b]) { seen[b] = 1; out.hbm_bodies.push_back(b); }
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
        return number[uf_find(root, is_static[a] ? b : a)];
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

void build_island_schedule(const int* body1, const int* body2, int nj, const unsigned char* is_static, int nb,
                           const LdsCaps& caps, Schedule& out)
{
    reset(out);
    out.islands = true;
    std::vector<int> root, number;
    const int ncomp = components(body1, body2, nj, is_static, nb, root, number);
    // joints per component, in joint order (CSR)
    std::vector<int> comp_of(nj), comp_count(ncomp + 1, 0);
    for (int j = 0; j < nj; ++j) {
        const int a = body1[j], b = body2[j];
        comp_of[j] = (is_static[a] && is_static[b]) ? -1 : number[uf_find(root, is_static[a] ? b : a)];
        if (comp_of[j] >= 0) comp_count[comp_of[j] + 1]++;
    }
    for (int c = 0; c < ncomp; ++c) comp_count[c + 1] += comp_count[c];
    std::vector<int> comp_joints(comp_count[ncomp]);
    {
        std::vector<int> cur(comp_count.begin(), comp_count.end() - 1);
        for (int j = 0; j < nj; ++j) if (comp_of[j] >= 0) comp_joints[cur[comp_of[j]]++] = j;
    }
    // bin consecutive components while they fit one workgroup
    std::vector<int> rest;                       // joints left to the HBM group
    std::vector<int> stamp(nb, -1), local(nb, 0);
    std::vector<int> bin, colour;
    ColourScratch scratch;
    int bin_id = 0;
    auto flush = [&]() {
        if (bin.empty()) return;
        std::sort(bin.begin(), bin.end());       // joint-index order inside the bin
        // local body table
        std::vector<int> bodies, dynamic;
        for (int j : bin)
            for (int b : {body1[j], body2[j]})
                if (stamp[b] != bin_id) { stamp[b] = bin_id; (is_static[b] ? bodies : dynamic).push_back(b); }
        const int nstatic = (int)bodies.size();
        bodies.insert(bodies.end(), dynamic.begin(), dynamic.end());      // static bodies first
        for (size_t i = 0; i < bodies.size(); ++i) local[bodies[i]] = (int)i;
        const int ncol = colour_joints(bin, body1, body2, is_static, nb, colour, scratch);
        if ((int)bodies.size() > caps.max_bodies || ncol > caps.max_colours || (int)bodies.size() > 65535 || nstatic > caps.max_static) {
            rest.insert(rest.end(), bin.begin(), bin.end());
        } else {
            const int base = (int)out.order.size();
            append_group(out, bin, colour, ncol);
            out.slot_local.resize(out.order.size());
            out.slot_colour.resize(out.order.size());
            const int first_colour = out.group_first_colour[out.group_first_colour.size() - 2];
            for (int c = 0; c < ncol; ++c)
                for (int s = out.colour_offsets[first_colour + c]; s < out.colour_offsets[first_colour + c + 1]; ++s) {
                    const int j = out.order[s];
                    out.slot_local[s] = (uint32_t)local[body1[j]] | ((uint32_t)local[body2[j]] << 16);
                    out.slot_colour[s] = (uint8_t)c;
                }
            out.group_bodies.insert(out.group_bodies.end(), bodies.begin(), bodies.end());
            out.group_body_offsets.push_back((int)out.group_bodies.size());
            out.lds_groups++;
            (void)base;
        }
        bin.clear();
        ++bin_id;
    };
    for (int c = 0; c < ncomp; ++c) {
        const int n = comp_count[c + 1] - comp_count[c];
        if (n == 0) continue;
        if (n > caps.max_joints) {               // a component too big for a workgroup goes to HBM whole
            flush();
            rest.insert(rest.end(), comp_joints.begin() + comp_count[c], comp_joints.begin() + comp_count[c + 1]);
            continue;
        }
        if ((int)bin.size() + n > caps.max_joints) flush();
        bin.insert(bin.end(), comp_joints.begin() + comp_count[c], comp_joints.begin() + comp_count[c + 1]);
    }
    flush();
    for (int j = 0; j < nj; ++j) if (comp_of[j] < 0) rest.push_back(j);
    if (!rest.empty()) {
        std::sort(rest.begin(), rest.end());
        const int ncol = colour_joints(rest, body1, body2, is_static, nb, colour, scratch);
        append_group(out, rest, colour, ncol);
        std::vector<unsigned char> seen(nb, 0);
        for (int j : rest)
            for (int b : {body1[j], body2[j]})
                if (!seen[b]) { seen[b] = 1; out.hbm_bodies.push_back(b); }
    }
}

} // namespace phx
