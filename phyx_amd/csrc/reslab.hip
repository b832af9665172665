// reslab.hip — the hand-over of an ownership-sharded world's bodies between ranks (slab mode), behind the C ABI.
//
// A rank of a slab-mode run steps a complete World on its x-slab of whole islands (phyx_amd/dist.py SlabWorld states the scheme).  When
// a body reaches its slab's boundary every rank re-slabs, in two phases:
//   1. the ranks all-gather {scene index, x-interval} of their dynamic bodies (24 bytes per body; two bodies that share a manifold cover
//      each other's interval), cut the x axis anew in the gaps no interval covers (slab_cuts) and see whether anybody changes owner;
//   2. only if so they all-gather their worlds' STATES (bodies, manifolds + contact points, joints with their warm-start impulses: what
//      phx_world_set_state restores) and every rank restores the part of the union world that lives in its new slab.
// Round 5 had this in Python over numpy-staged collectives; a C consumer of include/phyx_amd.h could not re-slab.  Here the planning is
// host C++ and the collectives run on DEVICE buffers through the library's own RCCL communicator (comm.hip) — or, where no RCCL
// communicator can exist (several ranks on one GPU in the tests, gloo), through two caller-supplied host callbacks.
// Host code only (a .hip so that the whole library goes through one compiler).
#include "reslab.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

namespace phx {

// Island-safe slabs for dynamic bodies whose AABBs span [lo[i], hi[i]] on the x axis.  Bodies whose (margin-widened) intervals overlap
// form a block that is never split; the nranks - 1 cuts sit in the middle of the gaps between blocks, chosen so that the slabs hold
// nearly equal numbers of bodies.  owner[i] = the rank of body i; bounds[2 r .. 2 r + 1] = rank r's open x-interval ((inf, inf): the
// rank holds nothing).
void slab_cuts(const double* lo, const double* hi, int n, int nranks, double margin, int* owner, double* bounds)
{
    const double inf = std::numeric_limits<double>::infinity();
    for (int i = 0; i < n; ++i) owner[i] = 0;
    if (n == 0) { for (int r = 0; r < nranks; ++r) { bounds[2 * r] = -inf; bounds[2 * r + 1] = inf; } return; }
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lo[a] < lo[b]; });
    std::vector<int> starts, ends;                       // first sorted position of every block / one past its last
    {
        double reach = 0.0;
        for (int k = 0; k < n; ++k) {
            const double slo = lo[order[k]] - margin, shi = hi[order[k]] + margin;
            if (k == 0 || slo > reach) starts.push_back(k);
            reach = k == 0 ? shi : std::max(reach, shi);
        }
        ends.assign(starts.begin() + 1, starts.end());
        ends.push_back(n);
    }
    const int nblocks = (int)starts.size();
    std::vector<int> cut_at(1, 0);                       // indices into `starts`: the block where each slab begins
    for (int r = 1; r < nranks; ++r) {
        const double ideal = (double)n * (double)r / (double)nranks;
        int k = 0;
        double best = std::fabs((double)ends[0] - ideal);
        for (int q = 1; q < nblocks; ++q) { const double d = std::fabs((double)ends[q] - ideal); if (d < best) { best = d; k = q; } }      // the block end nearest to the ideal count (the first of equals)
        cut_at.push_back(std::min(std::max(k + 1, cut_at.back()), nblocks));
    }
    cut_at.push_back(nblocks);
    std::vector<double> true_hi(n);
    for (int k = 0; k < n; ++k) true_hi[k] = k == 0 ? hi[order[0]] : std::max(true_hi[k - 1], hi[order[k]]);
    for (int r = 0; r < nranks; ++r) {
        const int b0 = cut_at[r], b1 = cut_at[r + 1];
        if (b0 >= b1) { bounds[2 * r] = inf; bounds[2 * r + 1] = inf; continue; }      // no body: an empty interval (nothing lives there, nothing can violate it)
        const int first = starts[b0], last = ends[b1 - 1];
        for (int k = first; k < last; ++k) owner[order[k]] = r;
        bounds[2 * r] = first == 0 ? -inf : 0.5 * (true_hi[first - 1] + lo[order[first]]);
        bounds[2 * r + 1] = last == n ? inf : 0.5 * (true_hi[last - 1] + lo[order[last]]);
    }
}

static inline bool body_static(const phx_rigid_body& b) { return b.inv_mass == 0.f && b.inv_inertia == 0.f; }

// x-intervals of the bodies, each widened to cover the bodies it shares a manifold with (two passes, like dist.py: a manifold never
// spans two ranks, so every rank can widen its own)
static void widened_intervals(const phx_rigid_body* bodies, int nb, const phx_manifold* manifolds, int nm, std::vector<double>& lo, std::vector<double>& hi)
{
    lo.resize(nb); hi.resize(nb);
    for (int i = 0; i < nb; ++i) { lo[i] = (double)bodies[i].aabb_min.x; hi[i] = (double)bodies[i].aabb_max.x; }
    std::vector<double> l, h;
    for (int pass = 0; pass < 2; ++pass) {
        l.assign(nm, 0.0); h.assign(nm, 0.0);
        for (int k = 0; k < nm; ++k) { const int a = manifolds[k].body1, b = manifolds[k].body2; l[k] = std::min(lo[a], lo[b]); h[k] = std::max(hi[a], hi[b]); }      // (from the intervals as the pass found them)
        for (int k = 0; k < nm; ++k) {
            const int a = manifolds[k].body1, b = manifolds[k].body2;
            if (body_static(bodies[a]) || body_static(bodies[b])) continue;
            lo[a] = std::min(lo[a], l[k]); lo[b] = std::min(lo[b], l[k]); hi[a] = std::max(hi[a], h[k]); hi[b] = std::max(hi[b], h[k]);
        }
    }
}

int reslab_intervals(const SlabState& st, std::vector<long long>& gi, std::vector<double>& lo, std::vector<double>& hi)
{
    const int nb = (int)st.bodies.size(), nm = (int)st.manifolds.size();
    for (int k = 0; k < nm; ++k)
        if ((unsigned)st.manifolds[k].body1 >= (unsigned)nb || (unsigned)st.manifolds[k].body2 >= (unsigned)nb) { set_error("re-slab: a manifold names a body out of range"); return PHX_ERR_INVALID; }
    std::vector<double> wl, wh;
    widened_intervals(st.bodies.data(), nb, st.manifolds.data(), nm, wl, wh);
    gi.clear(); lo.clear(); hi.clear();
    for (int i = 0; i < nb; ++i)
        if (!body_static(st.bodies[i])) { gi.push_back(st.global_index[i]); lo.push_back(wl[i]); hi.push_back(wh[i]); }
    return PHX_OK;
}

// every rank's intervals (any order) -> the new owner of every dynamic body, aligned with the ascending scene indices left in `gi`
void reslab_plan(std::vector<long long>& gi, std::vector<double>& lo, std::vector<double>& hi, int nranks, double margin, std::vector<int>& owner, std::vector<double>& bounds)
{
    const int n = (int)gi.size();
    std::vector<int> order(n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return gi[a] < gi[b]; });
    std::vector<long long> g2(n); std::vector<double> l2(n), h2(n);
    for (int k = 0; k < n; ++k) { g2[k] = gi[order[k]]; l2[k] = lo[order[k]]; h2[k] = hi[order[k]]; }
    gi.swap(g2); lo.swap(l2); hi.swap(h2);
    owner.assign(std::max(n, 1), 0);
    bounds.assign(2 * (size_t)nranks, 0.0);
    slab_cuts(lo.data(), hi.data(), n, nranks, margin, owner.data(), bounds.data());
    owner.resize(n);
}

// ---- the byte strings the ranks exchange: int64 count + int64 byte sizes + the arrays (8-byte aligned) ----------------------
static void blob_put(std::vector<unsigned char>& out, std::initializer_list<std::pair<const void*, size_t>> parts)
{
    std::vector<long long> head(1 + parts.size());
    head[0] = (long long)parts.size();
    size_t k = 1, total = head.size() * 8;
    for (const auto& p : parts) { head[k++] = (long long)p.second; total += (p.second + 7) & ~size_t(7); }
    out.assign(total, 0);
    std::memcpy(out.data(), head.data(), head.size() * 8);
    size_t at = head.size() * 8;
    for (const auto& p : parts) { if (p.second) std::memcpy(out.data() + at, p.first, p.second); at += (p.second + 7) & ~size_t(7); }
}

struct BlobPart { const unsigned char* p; size_t bytes; };
static bool blob_get(const std::vector<unsigned char>& blob, int want, std::vector<BlobPart>& parts)
{
    if (blob.size() < 8) return false;
    long long k = 0;
    std::memcpy(&k, blob.data(), 8);
    if (k != want || blob.size() < 8 + 8 * (size_t)k) return false;
    size_t at = 8 + 8 * (size_t)k;
    parts.clear();
    for (int q = 0; q < want; ++q) {
        long long sz = 0;
        std::memcpy(&sz, blob.data() + 8 + 8 * (size_t)q, 8);
        if (sz < 0 || at + (size_t)sz > blob.size()) return false;
        parts.push_back(BlobPart{blob.data() + at, (size_t)sz});
        at += ((size_t)sz + 7) & ~size_t(7);
    }
    return true;
}

// all-gather of byte strings of different lengths: padded to the longest (agreed by an all-reduce max), prefixed by their length
int SlabTransport::all_gather_var(const std::vector<unsigned char>& mine, std::vector<std::vector<unsigned char>>& all)
{
    all.clear();
    if (size <= 1) { all.push_back(mine); return PHX_OK; }
    long long longest = (long long)mine.size();
    PHX_TRY(reduce_max(&longest));
    const size_t seg = 8 + (((size_t)longest + 7) & ~size_t(7));
    if (seg >= (size_t)1 << 31) { set_error("re-slab: a rank's share of %zu bytes is more than one collective carries", seg); return PHX_ERR_CAPACITY; }
    std::vector<unsigned char> send(seg, 0), recv(seg * (size_t)size, 0);
    const long long len = (long long)mine.size();
    std::memcpy(send.data(), &len, 8);
    if (len) std::memcpy(send.data() + 8, mine.data(), (size_t)len);
    if (comm) {
        // device buffers through the library's RCCL communicator
        PHX_TRY(use_device(comm->device()));
        PHX_TRY(d_send.reserve(seg)); PHX_TRY(d_recv.reserve(seg * (size_t)size));
        PHX_HIP(hipMemcpyAsync(d_send.p, send.data(), seg, hipMemcpyHostToDevice, stream));
        PHX_TRY(comm->all_gather(d_send.p, d_recv.p, seg, stream));
        PHX_TRY(comm->wait_stream(stream, "re-slab: all-gather"));
        PHX_HIP(hipMemcpy(recv.data(), d_recv.p, recv.size(), hipMemcpyDeviceToHost));
    } else {
        if (!gather_fn) { set_error("re-slab: %d ranks but neither a communicator nor an all-gather callback", size); return PHX_ERR_INVALID; }
        if (gather_fn(user, send.data(), recv.data(), seg) != 0) { set_error("re-slab: the caller's all-gather failed"); return PHX_ERR_STATE; }
    }
    for (int r = 0; r < size; ++r) {
        long long n = 0;
        std::memcpy(&n, recv.data() + (size_t)r * seg, 8);
        if (n < 0 || (size_t)n + 8 > seg) { set_error("re-slab: rank %d's share arrived damaged", r); return PHX_ERR_STATE; }
        all.emplace_back(recv.data() + (size_t)r * seg + 8, recv.data() + (size_t)r * seg + 8 + (size_t)n);
    }
    return PHX_OK;
}

int SlabTransport::reduce_max(long long* value)
{
    if (size <= 1) return PHX_OK;
    if (comm) {
        if (*value < 0 || *value >= ((long long)1 << 31)) { set_error("re-slab: value out of the collective's range"); return PHX_ERR_CAPACITY; }
        PHX_TRY(use_device(comm->device()));
        PHX_TRY(d_word.reserve(4));
        const int v = (int)*value;
        PHX_HIP(hipMemcpyAsync(d_word.p, &v, sizeof v, hipMemcpyHostToDevice, stream));
        PHX_TRY(comm->all_reduce_max_int(d_word.p, stream));
        PHX_TRY(comm->wait_stream(stream, "re-slab: all-reduce"));
        int out = 0;
        PHX_HIP(hipMemcpy(&out, d_word.p, sizeof out, hipMemcpyDeviceToHost));
        *value = out;
        return PHX_OK;
    }
    if (!max_fn) { set_error("re-slab: %d ranks but neither a communicator nor an all-reduce callback", size); return PHX_ERR_INVALID; }
    long long v = *value;
    if (max_fn(user, &v) != 0) { set_error("re-slab: the caller's all-reduce failed"); return PHX_ERR_STATE; }
    *value = v;
    return PHX_OK;
}

// this rank's share of the hand-over: its world's state with body ids in the full scene's numbering
static void pack_state(const SlabState& st, std::vector<unsigned char>& out)
{
    std::vector<phx_manifold> m(st.manifolds);
    std::vector<phx_contact_joint> j(st.joints);
    for (auto& x : m) { x.body1 = (int)st.global_index[x.body1]; x.body2 = (int)st.global_index[x.body2]; }
    for (auto& x : j) { x.body1 = (int)st.global_index[x.body1]; x.body2 = (int)st.global_index[x.body2]; }
    blob_put(out, {{st.global_index.data(), st.global_index.size() * sizeof(long long)}, {st.bodies.data(), st.bodies.size() * sizeof(phx_rigid_body)},
                   {m.data(), m.size() * sizeof(phx_manifold)}, {st.cps.data(), st.cps.size() * sizeof(phx_contact_point)}, {j.data(), j.size() * sizeof(phx_contact_joint)}});
}

// Every rank's state -> this rank's new slab: the union world is assembled (bodies in scene order; manifolds, contact points and
// joints rank after rank), cut anew (slab_cuts on the dynamic bodies' widened intervals) and the part that lives in rank `me`'s new slab
// comes back in `out` with local body ids.
static int apply_states(const std::vector<std::vector<unsigned char>>& blobs, int scene_size, int me, int nranks, double margin, SlabState& out, double bounds_out[2])
{
    std::vector<phx_rigid_body> full((size_t)scene_size);
    std::vector<unsigned char> seen((size_t)scene_size, 0);
    std::vector<phx_manifold> M; std::vector<phx_contact_point> Cp; std::vector<phx_contact_joint> J;
    std::vector<BlobPart> parts;
    for (const auto& blob : blobs) {
        if (!blob_get(blob, 5, parts) || parts[0].bytes % 8 || parts[1].bytes != parts[0].bytes / 8 * sizeof(phx_rigid_body) || parts[2].bytes % sizeof(phx_manifold) ||
            parts[3].bytes != parts[2].bytes / sizeof(phx_manifold) * 2 * sizeof(phx_contact_point) || parts[4].bytes % sizeof(phx_contact_joint)) {
            set_error("re-slab: a rank's state arrived damaged"); return PHX_ERR_STATE;
        }
        const size_t nb = parts[0].bytes / 8, nm = parts[2].bytes / sizeof(phx_manifold), nj = parts[4].bytes / sizeof(phx_contact_joint);
        const int m_off = (int)M.size();
        for (size_t i = 0; i < nb; ++i) {
            long long g = 0;
            std::memcpy(&g, parts[0].p + 8 * i, 8);
            if (g < 0 || g >= scene_size) { set_error("re-slab: a body's scene index is out of range"); return PHX_ERR_INVALID; }
            std::memcpy(&full[(size_t)g], parts[1].p + i * sizeof(phx_rigid_body), sizeof(phx_rigid_body));
            seen[(size_t)g] = 1;
        }
        M.resize(M.size() + nm); Cp.resize(Cp.size() + 2 * nm); J.resize(J.size() + nj);
        if (nm) { std::memcpy(&M[(size_t)m_off], parts[2].p, parts[2].bytes); std::memcpy(&Cp[2 * (size_t)m_off], parts[3].p, parts[3].bytes); }
        if (nj) std::memcpy(&J[J.size() - nj], parts[4].p, parts[4].bytes);
        for (size_t k = 0; k < nm; ++k) M[(size_t)m_off + k].point_index = 2 * (m_off + (int)k);
        for (size_t k = J.size() - nj; k < J.size(); ++k) J[k].contact_point_index += 2 * m_off;
    }
    for (int g = 0; g < scene_size; ++g) if (!seen[(size_t)g]) { set_error("re-slab: body %d of the scene is on no rank", g); return PHX_ERR_STATE; }
    for (const auto& m : M) if ((unsigned)m.body1 >= (unsigned)scene_size || (unsigned)m.body2 >= (unsigned)scene_size) { set_error("re-slab: a manifold names a body out of range"); return PHX_ERR_INVALID; }
    for (const auto& j : J) if ((unsigned)j.contact_point_index >= 2 * M.size()) { set_error("re-slab: a joint names a contact point out of range"); return PHX_ERR_INVALID; }
    std::vector<double> lo, hi;
    widened_intervals(full.data(), scene_size, M.data(), (int)M.size(), lo, hi);
    std::vector<int> dyn;
    for (int g = 0; g < scene_size; ++g) if (!body_static(full[(size_t)g])) dyn.push_back(g);
    std::vector<double> dlo(dyn.size()), dhi(dyn.size());
    for (size_t k = 0; k < dyn.size(); ++k) { dlo[k] = lo[(size_t)dyn[k]]; dhi[k] = hi[(size_t)dyn[k]]; }
    std::vector<int> owner_dyn(std::max<size_t>(dyn.size(), 1), 0);
    std::vector<double> bounds(2 * (size_t)nranks);
    slab_cuts(dlo.data(), dhi.data(), (int)dyn.size(), nranks, margin, owner_dyn.data(), bounds.data());
    std::vector<int> owner((size_t)scene_size, -1);
    for (size_t k = 0; k < dyn.size(); ++k) owner[(size_t)dyn[k]] = owner_dyn[k];
    std::vector<int> local_of((size_t)scene_size, -1);
    out = SlabState{};
    for (int g = 0; g < scene_size; ++g)
        if (owner[(size_t)g] < 0 || owner[(size_t)g] == me) {              // static bodies live on every rank
            local_of[(size_t)g] = (int)out.bodies.size();
            out.global_index.push_back(g);
            out.bodies.push_back(full[(size_t)g]);
            out.bodies.back().index = (uint32_t)(out.bodies.size() - 1);
        }
    std::vector<int> new_m_of(M.size(), -1);
    for (size_t k = 0; k < M.size(); ++k) {
        const phx_manifold& m = M[k];
        const bool s1 = owner[(size_t)m.body1] < 0, s2 = owner[(size_t)m.body2] < 0;
        if (!s1 && !s2 && owner[(size_t)m.body1] != owner[(size_t)m.body2]) { set_error("re-slab: a manifold spans two slabs"); return PHX_ERR_STATE; }
        const int m_owner = s1 ? owner[(size_t)m.body2] : owner[(size_t)m.body1];
        if (m_owner != me) continue;
        new_m_of[k] = (int)out.manifolds.size();
        phx_manifold q = m;
        q.body1 = local_of[(size_t)m.body1]; q.body2 = local_of[(size_t)m.body2]; q.point_index = 2 * (int)out.manifolds.size();
        out.manifolds.push_back(q);
        out.cps.push_back(Cp[2 * k]); out.cps.push_back(Cp[2 * k + 1]);
    }
    for (const auto& j : J) {
        const int nm_ = new_m_of[(size_t)j.contact_point_index / 2];
        if (nm_ < 0) continue;
        phx_contact_joint q = j;
        q.contact_point_index = 2 * nm_ + j.contact_point_index % 2;
        q.body1 = local_of[(size_t)j.body1]; q.body2 = local_of[(size_t)j.body2];
        out.cps[(size_t)q.contact_point_index].solver_index = (int)out.joints.size();      // (slots no joint points at keep their bytes: the step never reads them)
        out.joints.push_back(q);
    }
    bounds_out[0] = bounds[2 * (size_t)me]; bounds_out[1] = bounds[2 * (size_t)me + 1];
    return PHX_OK;
}

// The whole re-slab of one rank (collective: every rank calls it at the same step).  `st` is this rank's world on entry and — if
// anybody moved — its new world on return (`moved` says which); `bounds` is its slab either way.
int reslab(SlabTransport& tp, SlabState& st, int scene_size, double margin, double bounds[2], int* moved)
{
    *moved = 0;
    std::vector<long long> gi; std::vector<double> lo, hi;
    PHX_TRY(reslab_intervals(st, gi, lo, hi));
    std::vector<unsigned char> mine;
    blob_put(mine, {{gi.data(), gi.size() * 8}, {lo.data(), lo.size() * 8}, {hi.data(), hi.size() * 8}});
    std::vector<std::vector<unsigned char>> all;
    PHX_TRY(tp.all_gather_var(mine, all));
    std::vector<long long> agi; std::vector<double> alo, ahi;
    std::vector<BlobPart> parts;
    for (const auto& blob : all) {
        if (!blob_get(blob, 3, parts) || parts[0].bytes != parts[1].bytes || parts[0].bytes != parts[2].bytes || parts[0].bytes % 8) { set_error("re-slab: a rank's intervals arrived damaged"); return PHX_ERR_STATE; }
        const size_t n = parts[0].bytes / 8, at = agi.size();
        agi.resize(at + n); alo.resize(at + n); ahi.resize(at + n);
        if (n) { std::memcpy(&agi[at], parts[0].p, 8 * n); std::memcpy(&alo[at], parts[1].p, 8 * n); std::memcpy(&ahi[at], parts[2].p, 8 * n); }
    }
    std::vector<int> owner; std::vector<double> all_bounds;
    reslab_plan(agi, alo, ahi, tp.size, margin, owner, all_bounds);
    std::vector<long long> now(gi), then;
    std::sort(now.begin(), now.end());
    for (size_t k = 0; k < agi.size(); ++k) if (owner[k] == tp.rank) then.push_back(agi[k]);      // (ascending already)
    long long any = now == then ? 0 : 1;
    PHX_TRY(tp.reduce_max(&any));
    if (!any) { bounds[0] = all_bounds[2 * (size_t)tp.rank]; bounds[1] = all_bounds[2 * (size_t)tp.rank + 1]; return PHX_OK; }
    std::vector<unsigned char> state;
    pack_state(st, state);
    PHX_TRY(tp.all_gather_var(state, all));
    SlabState fresh;
    PHX_TRY(apply_states(all, scene_size, tp.rank, tp.size, margin, fresh, bounds));
    st = std::move(fresh);
    *moved = 1;
    return PHX_OK;
}

} // namespace phx

extern "C" int phx_reslab_cuts(const double* lo, const double* hi, int32_t n, int32_t nranks, double margin, int32_t* owner, double* bounds)
{
    PHX_REQUIRE(n >= 0 && nranks >= 1 && bounds && (n == 0 || (lo && hi && owner)), "bad arguments");
    std::vector<int> tmp((size_t)std::max(n, 1));
    phx::slab_cuts(lo, hi, n, nranks, margin, tmp.data(), bounds);
    for (int i = 0; i < n; ++i) owner[i] = tmp[(size_t)i];
    return PHX_OK;
}
