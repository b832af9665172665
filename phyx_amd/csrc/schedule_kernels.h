// schedule_kernels.h — building the island-aware schedule on the device.
//
// The schedule is a pure function of the joints' body pairs and of which bodies are static (schedule.h).  The
// host builder (schedule.hip) is the specification; this file produces the SAME schedule — same groups, same
// colours, same slot order (tests compare the two) — without pulling the joint list over PCIe:
//   connected components   min-label hooking + pointer jumping over the joint list
//                          (the device form of the union-find of ref: Solver.cpp:275-323)
//   numbering              components numbered by their smallest body index (= body order, ref: Solver.cpp:344-356)
//   binning                greedy over consecutive components — ncomp integers, done on the host
//   joint order            stable radix sort of the joints by bin (device_radix.h)
//   per bin                one workgroup: local body table (static first), first-fit colouring in joint order,
//                          stable counting sort by colour -> slot arrays
#pragma once

#include "common.h"

namespace phx {

// ---- connected components over dynamic bodies --------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_cc_init(const phx_rigid_body* __restrict__ bodies, int nb, int* __restrict__ parent,
                                                        unsigned char* __restrict__ is_static)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const bool st = bodies[i].inv_mass == 0.f && bodies[i].inv_inertia == 0.f;      // ref: Solver.cpp:304
        is_static[i] = st ? 1 : 0;
        parent[i] = st ? -1 : i;
    }
}

// every joint between two dynamic bodies hooks the larger of the two current labels under the smaller
static __global__ void __launch_bounds__(256) k_cc_hook(const phx_contact_joint* __restrict__ joints, int nj, int nb, int* parent, int* __restrict__ changed)
{
    bool any = false;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x) {
        const unsigned u = (unsigned)joints[j].body1, v = (unsigned)joints[j].body2;
        if (u >= (unsigned)nb || v >= (unsigned)nb) continue;          // reported by the fingerprint / validation path
        const int pu = parent[u], pv = parent[v];
        if (pu < 0 || pv < 0 || pu == pv) continue;
        atomicMin(&parent[pu > pv ? pu : pv], pu > pv ? pv : pu);
        any = true;
    }
    if (__any(any) && (threadIdx.x & 63) == 0) *changed = 1;
}

// full path compression: afterwards parent[b] is the representative (smallest label reached so far)
static __global__ void __launch_bounds__(256) k_cc_compress(int* parent, int nb)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        int p = parent[i];
        if (p < 0) continue;
        while (true) { const int q = parent[p]; if (q == p) break; p = q; }
        parent[i] = p;
    }
}

static __global__ void __launch_bounds__(256) k_cc_root_flags(const int* __restrict__ parent, int nb, unsigned* __restrict__ flags)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) flags[i] = parent[i] == i ? 1u : 0u;
}

// joint -> component number (-1 if both bodies are static), and joints per component
static __global__ void __launch_bounds__(256) k_joint_components(const phx_contact_joint* __restrict__ joints, int nj, int nb, const int* __restrict__ parent,
                                                                 const unsigned* __restrict__ root_number, int* __restrict__ joint_comp,
                                                                 unsigned* __restrict__ comp_size)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x) {
        const unsigned u = (unsigned)joints[j].body1, v = (unsigned)joints[j].body2;
        int comp = -1;
        if (u < (unsigned)nb && v < (unsigned)nb) {
            const int pu = parent[u], pv = parent[v];
            const int r = pu >= 0 ? pu : pv;
            if (r >= 0) comp = (int)root_number[r];
        }
        joint_comp[j] = comp;
        // neighbouring joints mostly share a component: one atomic per distinct component per wave
        unsigned long long todo = __ballot(comp >= 0);
        const int lane = threadIdx.x & 63;
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int key = __shfl(comp, leader);
            const unsigned long long same = __ballot(comp == key) & todo;
            if (lane == leader) atomicAdd(&comp_size[key], (unsigned)__popcll(same));
            todo &= ~same;
        }
    }
}

static __global__ void __launch_bounds__(256) k_joint_bin_keys(const int* __restrict__ joint_comp, const int* __restrict__ bin_of_comp, int nj, int rest_key,
                                                               unsigned* __restrict__ keys, unsigned* __restrict__ vals)
{
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < nj; j += gridDim.x * blockDim.x) {
        const int c = joint_comp[j];
        keys[j] = (unsigned)(c < 0 ? rest_key : bin_of_comp[c]);
        vals[j] = (unsigned)j;
    }
}

// ---- one workgroup builds one bin ------------------------------------------------------------------------------
struct BinBuildView {
    const unsigned* sorted_joints;    // joint indices grouped by bin, joint order inside a bin
    const int* group_offsets;         // slots of bin g = [group_offsets[g], group_offsets[g+1])
    const phx_contact_joint* joints;
    const unsigned char* is_static;
    int nb, max_static;
    int* order;                       // out: slot -> joint
    unsigned* slot_local;             // out: local body1 | local body2 << 16
    unsigned char* slot_colour;       // out
    int4* desc;                       // out: {slot_begin, slot_count, body_begin, body_count}
    int* ncol;                        // out
    int* bodies;                      // out: body table of bin g at [g * NB, g * NB + body_count)
    int* rejected;                    // out: set to 1 if any bin exceeds the caps (caller falls back to the host builder)
};

template <int T, int NB>
static __global__ void __launch_bounds__(T) k_build_bin(BinBuildView v)
{
    constexpr int HT = 4 * T;                           // open-addressing table, <= 2T distinct bodies
    __shared__ int ht_key[HT];
    __shared__ int ht_val[HT];                          // first occurrence position, later the local index
    __shared__ int jb[2][T];
    __shared__ unsigned long long used[NB];
    __shared__ unsigned short col[T], pos[T];
    __shared__ unsigned scan_lds[T / 64];
    __shared__ unsigned hist[64];                         // per-colour counts, then cursors (LDS: a private array would live in scratch)
    __shared__ int n_static, n_bodies, n_col, bad;

    const int g = blockIdx.x, tid = threadIdx.x;
    const int begin = v.group_offsets[g], count = v.group_offsets[g + 1] - begin;
    for (int i = tid; i < HT; i += T) { ht_key[i] = -1; ht_val[i] = 0x7fffffff; }
    for (int i = tid; i < NB; i += T) used[i] = 0ull;
    if (tid < 64) hist[tid] = 0;
    if (tid == 0) { bad = 0; n_col = 0; }
    __syncthreads();

    const bool live = tid < count;
    int j = 0, b[2] = {0, 0}, hs[2] = {0, 0};
    if (live) {
        j = (int)v.sorted_joints[begin + tid];
        b[0] = v.joints[j].body1; b[1] = v.joints[j].body2;
        jb[0][tid] = b[0]; jb[1][tid] = b[1];
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
    // leaders = first occurrences; local index = rank among static leaders, or n_static + rank among dynamic leaders
    bool lead[2] = {false, false}, stat[2] = {false, false};
    unsigned mine = 0;                                   // static leaders << 16 | dynamic leaders
    if (live)
        for (int s = 0; s < 2; ++s) {
            lead[s] = ht_val[hs[s]] == 2 * tid + s;
            stat[s] = v.is_static[b[s]] != 0;
            if (lead[s]) mine += stat[s] ? 0x10000u : 1u;
        }
    // block exclusive scan of `mine`
    unsigned x = mine;
    const int lane = tid & 63, wave = tid >> 6;
    for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) scan_lds[wave] = x;
    __syncthreads();
    unsigned before = x - mine, total = 0;
    for (int w = 0; w < T / 64; ++w) { const unsigned t = scan_lds[w]; if (w < wave) before += t; total += t; }
    if (tid == 0) { n_static = (int)(total >> 16); n_bodies = (int)(total >> 16) + (int)(total & 0xFFFFu); }
    __syncthreads();                                     // every lane has read ht_val as "first position"
    const bool fits = n_bodies <= NB && n_static <= v.max_static;
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
    int loc[2] = {0, 0};
    if (live && fits) { loc[0] = ht_val[hs[0]]; loc[1] = ht_val[hs[1]]; jb[0][tid] = loc[0]; jb[1][tid] = loc[1]; }
    __syncthreads();
    // greedy first-fit colouring in joint order (one lane; the masks live in LDS), then stable positions by colour
    if (tid == 0 && fits) {
        int ncol = 0;
        for (int k = 0; k < count; ++k) {
            const int a = jb[0][k], c2 = jb[1][k];
            const bool da = a >= n_static, db = c2 >= n_static;          // static bodies sit first in the table
            unsigned long long m = 0;
            if (da) m |= used[a];
            if (db) m |= used[c2];
            if (!~m) { bad = 1; break; }
            const int c = __builtin_ctzll(~m);
            if (da) used[a] |= 1ull << c;
            if (db) used[c2] |= 1ull << c;
            col[k] = (unsigned short)c;
            hist[c]++;
            if (c + 1 > ncol) ncol = c + 1;
        }
        if (!bad) {
            unsigned run = 0;
            for (int c = 0; c < ncol; ++c) { const unsigned n = hist[c]; hist[c] = run; run += n; }
            for (int k = 0; k < count; ++k) pos[k] = (unsigned short)hist[col[k]]++;
            n_col = ncol;
        }
    }
    __syncthreads();
    if (!fits || bad) { if (tid == 0) *v.rejected = 1; return; }
    if (live) {
        const int slot = begin + pos[tid];
        v.order[slot] = j;
        v.slot_local[slot] = (unsigned)loc[0] | ((unsigned)loc[1] << 16);
        v.slot_colour[slot] = (unsigned char)col[tid];
    }
    if (tid == 0) { v.desc[g] = make_int4(begin, count, g * NB, n_bodies); v.ncol[g] = n_col; }
}

} // namespace phx
