// broadphase.hip — single-axis sweep-and-prune broadphase on gfx950.
//
// Replaces (ref = /root/reference/src): Collider::UpdateBroadphase (Collider.cpp:251-284) = key build with
// radixFloat (base/RadixSort.h:19-26) + radixSort3 (base/RadixSort.h:28-95) + gather of sorted
// BroadphaseEntry records; Collider::UpdatePairs* (Collider.cpp:286-366) = forward sweep with the y-overlap
// test and the persistent pair set `manifoldMap` (Collider.h:58, base/DenseHash.h).
//
// Device layout:
//   keys/idx      u32 key = radixFloat(aabb.min.x), u32 idx — two ping-pong pairs of SoA arrays
//   entries       float4 {minx, maxx, centery, extenty} per sorted position + u32 body index (SoA)
//   pair set      open-addressing table of u64 (index_i << 32 | index_j), linear probing, in HBM
//   new pairs     uint2 list in the reference's serial emission order (row i ascending, then j)
//
// The sort is a stable LSD radix sort on the full 32-bit key.  The reference splits the key 11/11/10; a
// stable sort's output permutation does not depend on the digit split, so 4 passes of 8 bits (256-bin
// LDS histograms, one wave-private counter row per wave) give the identical sequence.
#include "handles.h"
#include "device_scan.h"
#include "device_radix.h"
#include "splitter_sort.h"

#include <algorithm>

namespace phx {

static inline int grid_for(int n, int per_block = 256, int cap = 4096) { return std::max(1, std::min(div_up(n, per_block), cap)); }

// ---- radixFloat (ref: base/RadixSort.h:19-26) -------------------------------------------------------
__device__ __forceinline__ unsigned radix_float(float v)
{
    const int f = __float_as_int(v);
    const unsigned mask = (unsigned)(f >> 31) | 0x80000000u;
    return (unsigned)f ^ mask;
}

// ref: Collider.cpp:259-265
// (first kernel of an update: it also clears the update's counters and the hub-chunk counts — two dispatches fewer)
__global__ void __launch_bounds__(256) k_extract_aabb(const phx_rigid_body* __restrict__ bodies, int n, float4* __restrict__ aabb)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        aabb[i] = make_float4(bodies[i].aabb_min.x, bodies[i].aabb_min.y, bodies[i].aabb_max.x, bodies[i].aabb_max.y);
}

template <bool INTEGRATE>
__global__ void __launch_bounds__(256) k_build_keys(const float4* __restrict__ aabb, float4* __restrict__ vel, const float4* __restrict__ mpos, int n,
                                                    unsigned* __restrict__ keys, unsigned* __restrict__ idx,
                                                    unsigned long long* __restrict__ small, int nsmall, unsigned* __restrict__ chunk_count, int nchunks,
                                                    unsigned long long* __restrict__ stamps, float gravity, float dt, unsigned* __restrict__ counters, const float4* __restrict__ accel)
{
    if (INTEGRATE && blockIdx.x == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0u;      // (the World's step counters)
    // the update's device time without HIP events (an event record is a barrier packet of its own: ~5 us of idle queue): this, its
    // first kernel, leaves the 100 MHz clock in stamps[0]; the count pass's mailbox post and the insert kernel raise stamps[1]
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = (unsigned long long)wall_clock64(); stamps[1] = 0ull; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nsmall; i += gridDim.x * blockDim.x) small[i] = 0ull;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += gridDim.x * blockDim.x) chunk_count[i] = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        keys[i] = radix_float(aabb[i].x);
        idx[i] = (unsigned)i;
        if (INTEGRATE) {                                       // IntegrateVelocity (ref: World.cpp:39-55), same statements as k_integrate_velocity
            float4 v = vel[i];
            float ax = 0.f, ay = 0.f, aa = 0.f;                // (the resident world carries no accelerations: world_kernels.h)
            if (accel) { const float4 a = accel[i]; ax = a.x; ay = a.y; aa = a.z; }      // (... but for the first step after an upload that came with some)
            if (mpos[i].x > 0.0f) ay += gravity;
            v.x += ax * dt; v.y += ay * dt;
            v.z += aa * dt;
            vel[i] = v;
        }
    }
}

// ref: Collider.cpp:269-283
// (`splitters`: the records at positions SS_STRIDE, 2 SS_STRIDE, ... of the sorted sequence — what the next update deals its bodies
//  into buckets by, splitter_sort.h)
__global__ void __launch_bounds__(256) k_gather_entries(const float4* __restrict__ aabb, const unsigned* __restrict__ keys, const unsigned* __restrict__ idx, int n,
                                                        float4* __restrict__ entries, unsigned long long* __restrict__ splitters, int stride)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 b = aabb[idx[i]];
        const float minx = b.x, miny = b.y, maxx = b.z, maxy = b.w;
        entries[i] = make_float4(minx, maxx, (miny + maxy) * 0.5f, (maxy - miny) * 0.5f);
        if (i && i % stride == 0) splitters[i / stride - 1] = ((unsigned long long)keys[i] << 32) | idx[i];
    }
}

// ---- persistent pair set ------------------------------------------------------------------------------
constexpr unsigned long long PS_EMPTY = 0xFFFFFFFFFFFFFFFFull;
constexpr unsigned long long PS_TOMB = 0xFFFFFFFFFFFFFFFEull;

__device__ __forceinline__ unsigned ps_hash(unsigned long long k)
{
    // same mixing idea as the reference's pair hash (ref: Collider.h:10-18), finished with a multiply
    const unsigned lb = (unsigned)(k >> 32), rb = (unsigned)k;
    return (lb ^ (rb + 0x9e3779b9u + (lb << 6) + (lb >> 2))) * 0x9E3779B1u;
}

__device__ __forceinline__ bool ps_contains(const unsigned long long* __restrict__ table, unsigned mask, unsigned long long k)
{
    unsigned p = ps_hash(k) & mask;
    for (;;) {
        const unsigned long long s = table[p];
        if (s == k) return true;
        if (s == PS_EMPTY) return false;
        p = (p + 1) & mask;
    }
}

// keys are distinct and known to be absent
__global__ void __launch_bounds__(256) k_ps_insert(unsigned long long* table, unsigned mask, const uint2* __restrict__ pairs, int n, unsigned long long* __restrict__ end_stamp)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long k = ((unsigned long long)pairs[i].x << 32) | pairs[i].y;
        unsigned p = ps_hash(k) & mask;
        for (;;) {
            const unsigned long long s = table[p];
            if (s == PS_EMPTY || s == PS_TOMB) {
                if (atomicCAS(&table[p], s, k) == s) break;
                continue;      // lost the slot to another inserter: look at it again
            }
            p = (p + 1) & mask;
        }
    }
    if (end_stamp && threadIdx.x == 0) atomicMax(end_stamp, (unsigned long long)wall_clock64());      // (the update's last kernel)
}

__global__ void __launch_bounds__(256) k_ps_erase(unsigned long long* table, unsigned mask, const uint2* __restrict__ pairs, int n, int* erased)
{
    int mine = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const unsigned long long k = ((unsigned long long)pairs[i].x << 32) | pairs[i].y;
        unsigned p = ps_hash(k) & mask;
        for (;;) {
            const unsigned long long s = table[p];
            if (s == k) { table[p] = PS_TOMB; ++mine; break; }
            if (s == PS_EMPTY) break;
            p = (p + 1) & mask;
        }
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);      // one atomic per wave
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(erased, mine);
}

__global__ void __launch_bounds__(256) k_ps_rehash(const unsigned long long* __restrict__ old_table, unsigned old_cap,
                                                   unsigned long long* table, unsigned mask)
{
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < old_cap; i += gridDim.x * blockDim.x) {
        const unsigned long long k = old_table[i];
        if (k == PS_EMPTY || k == PS_TOMB) continue;
        unsigned p = ps_hash(k) & mask;
        for (;;) {
            if (table[p] == PS_EMPTY && atomicCAS(&table[p], PS_EMPTY, k) == PS_EMPTY) break;
            p = (p + 1) & mask;
        }
    }
}

// ---- sweep (ref: Collider.cpp:296-318 serial order, :347-366 per-row body) ----------------------------
// Rows with a short scan range are swept one row per lane: lane l of a wave owns sorted row i0+l and reads
// entries[i0+l+1+t] in step t, so a wave's loads are contiguous.  Rows whose range exceeds HUB_LEN (the
// ground box spans every column) are deferred to a workgroup-per-row kernel.
constexpr int STAT_SLOTS = 64;     // same-address atomics serialise (~10 ns each): 31k of them cost 0.3 ms of a 0.4 ms kernel
constexpr int ROW_CACHE = 4;        // new pairs remembered per row by the count pass (rows with more are rescanned by the emit pass)
constexpr int HUB_LEN = 2048;      // rows scanning more candidates than this are cut into chunks
constexpr int HUB_CHUNK = 512;     // candidates per chunk = one 256-lane workgroup x 2 tiles (2048: the ground row of the 200k-box scene was 98 workgroups
                                   // walking eight dependent tiles each, 10 us per pass; 391 workgroups of two are done in a third of that)

struct SweepView {
    const float4* entries;
    const unsigned* idx;
    int n;
    const unsigned long long* table;
    unsigned mask;
    unsigned* row_count;       // new pairs per row
    unsigned* row_cache;       // count pass: the first ROW_CACHE new partners of every row (index_j)
    int* cache_overflow;       // set by the count pass if some row found more than ROW_CACHE new pairs
    int4* chunks;              // hub chunks {row, j_begin, j_end, index of the row's first chunk}
    unsigned* chunk_count;     // new pairs per chunk, later: the chunk's base inside its row
    int* n_chunks;
    int chunk_cap;
    unsigned long long* counters;   // statistics, spread over STAT_SLOTS slots to keep the atomics apart: [2 * slot] candidate tests, [2 * slot + 1] overlapping pairs
};

constexpr int SWEEP_CAND = 16;       // y-overlapping candidates a row can hold before it must look them up in the pair set
constexpr int SWEEP_LOOK = 8;        // lookups a row keeps in flight
constexpr int SWEEP_GROUP = 8;       // candidates the wave reads from its staged block per step of its walk

// first position j > i with minx[j] > maxx (entries sorted by minx)
__device__ __forceinline__ int scan_end(const float4* __restrict__ entries, int n, int i, float maxx)
{
    int lo = i + 1, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (entries[mid].x > maxx) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// One sorted row per lane, and the wave walks the CANDIDATES together: lane l owns row i0 + l, the wave steps through positions
// c = i0 + 1, i0 + 2, ... and every lane whose row has begun (c > its row) and not ended (minx[c] <= its maxx) tests the SAME
// candidate — whose record is therefore one scalar load for the wave (eight per fetch) instead of a 16-byte gather per lane.
// (Round 3 let every lane fetch its own window, eight loads in flight per lane: 20 M tests were 320 MB through the L1s and a
// row's scan a chain of dependent gathers — 44 us at cfg 2, 95 us at cfg 4.)  A lane sees its candidates in the same increasing
// order as before, so counts, caches and the emission order are unchanged.
// (Round 6 built the transposed walk — the lanes hold a block of 64 CANDIDATES in registers, the wave tests them against one row after
//  the other, row data by v_readlane, ballots give tests / overlaps / the row's end, the lane that holds an overlapping candidate looks
//  the pair up — byte-equal lists and counts, and SLOWER: 50.1 against 38.7 us at cfg 2, 103.9 against 80.6 at cfg 4.  Both walks do
//  one 64-wide step per (row block, candidate) resp. (row, candidate block) and the triangle of a tie column half fills either; the
//  transposed step is ~27 mostly scalar instructions, this one ~15 vector ones.  tools/probe/sweep_tiles.hip.txt keeps it.)
#define PHX_SWEEP_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

template <bool EMIT>
__global__ void __launch_bounds__(256) k_sweep_rows(SweepView v, const unsigned* __restrict__ row_offset, uint2* __restrict__ out, unsigned total)
{
    // EMIT = false: the count pass (also remembers each row's first ROW_CACHE new partners).
    // EMIT = true : rescans ONLY the rows that found more than ROW_CACHE new pairs; all other rows are emitted from the
    //               cache by k_emit_cached without touching the entries or the pair set again.
    __shared__ unsigned cand[SWEEP_CAND][256];          // per lane: positions of the candidates that overlap in y, not looked up yet
    __shared__ float4 stage[4][2][64];                  // per wave: the block of 64 candidates under test, and the next one
    unsigned long long tests = 0, overlaps = 0;
    const int lane = threadIdx.x & 63;
    for (int base = blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < v.n; base += gridDim.x * blockDim.x) {      // (wave-uniform)
        const int i = base + lane;
        const bool in_range = i < v.n;
        const float4 a = in_range ? v.entries[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        // a hub row (more than HUB_LEN candidates) is recognised by one probe: entries are sorted by minx, so the row is a
        // hub iff the candidate HUB_LEN places ahead still starts at or before this row's maxx.  Only hub rows pay the
        // binary search for their end; every other row finds it by scanning (ref: Collider.cpp:300-303 breaks the same way).
        const bool hub = in_range && i + 1 + HUB_LEN < v.n && !(v.entries[i + 1 + HUB_LEN].x > a.y);
        if (!EMIT) {
            // hand a hub row to the chunk kernels: its chunks sit contiguously and in j order in the list.  The row's lane finds the
            // end and takes the list positions; the WAVE writes the descriptors (the ground row of the 1M-box scene is 1954 of them).
            int end = 0, nc = 0, first = 0;
            if (hub) {
                end = scan_end(v.entries, v.n, i, a.y);
                const int len = end - i - 1;
                nc = (len + HUB_CHUNK - 1) / HUB_CHUNK;
                first = atomicAdd(v.n_chunks, nc);
                v.row_count[i] = 0;
                tests += (unsigned long long)len;
            }
            for (unsigned long long hubs = __ballot(hub); hubs; hubs &= hubs - 1ull) {
                const int l = __builtin_ctzll(hubs);
                const int hi = __shfl(i, l), hend = __shfl(end, l), hnc = __shfl(nc, l), hfirst = __shfl(first, l);
                for (int k = lane; k < hnc && hfirst + k < v.chunk_cap; k += 64)
                    v.chunks[hfirst + k] = make_int4(hi, hi + 1 + k * HUB_CHUNK, min(hend, hi + 1 + (k + 1) * HUB_CHUNK), hfirst);
            }
        }
        bool scanning = in_range && !hub;
        const unsigned dst = EMIT && scanning ? row_offset[i] : 0u;
        if (EMIT && scanning) {
            const unsigned next = i + 1 < v.n ? row_offset[i + 1] : total;
            if (next - dst <= (unsigned)ROW_CACHE) scanning = false;      // emitted from the cache
        }
        const unsigned ia = scanning ? v.idx[i] : 0u;
        unsigned found = 0;
        // The candidates that overlap in y are only COLLECTED by the scan (their positions, in j order, in LDS); the pair-set
        // lookups — two dependent memory round trips each — are made afterwards, all in flight together.  Done inside the scan
        // they were its whole cost: some lane of the wave hits an overlap in almost every step, and the wave waits for it.
        int ncand = 0;
        auto flush = [&]() {
            for (int h = 0; h < ncand; h += SWEEP_LOOK) {          // SWEEP_LOOK lookups in flight at a time
                unsigned ib[SWEEP_LOOK];
                unsigned long long first[SWEEP_LOOK];
#pragma unroll
                for (int k = 0; k < SWEEP_LOOK; ++k) ib[k] = h + k < ncand ? v.idx[cand[h + k][threadIdx.x]] : 0u;
#pragma unroll
                for (int k = 0; k < SWEEP_LOOK; ++k) first[k] = h + k < ncand ? v.table[ps_hash(((unsigned long long)ia << 32) | ib[k]) & v.mask] : 0ull;
#pragma unroll
                for (int k = 0; k < SWEEP_LOOK; ++k) {
                    if (h + k >= ncand) break;
                    const unsigned long long key = ((unsigned long long)ia << 32) | ib[k];
                    bool present = first[k] == key;
                    if (!present && first[k] != PS_EMPTY) present = ps_contains(v.table, v.mask, key);       // a collision at the home slot: walk on
                    if (!present) {
                        if (EMIT) out[dst + found] = make_uint2(ia, ib[k]);
                        else if (found < (unsigned)ROW_CACHE) v.row_cache[(size_t)i * ROW_CACHE + found] = ib[k];
                        ++found;
                    }
                }
            }
            ncand = 0;
        };
        // the wave's walk, 64 candidates per fetch: lane l fetches record c + l (one coalesced load for the wave), the block waits in
        // LDS, and the wave reads it record by record at a uniform address (a broadcast) — the NEXT block's fetch is already in
        // flight while this one is tested, so the walk never waits for memory: a fetch per 8 candidates (scalar loads, or eight
        // gathers per lane before that) was a ~1 us round trip each, 30 of them in a row per wave.
        int len = 0;
        bool ended = !scanning;
        int ended_w = scanning ? 0 : -1;
        const int c0 = __builtin_amdgcn_readfirstlane(base) + 1;
        float4* my_stage = &stage[threadIdx.x >> 6][0][0];
        float4 nxt = v.entries[min(c0 + lane, v.n - 1)];
        int buf = 0;
        for (int c = c0; c < v.n && __ballot(!ended); c += 64) {
            my_stage[buf * 64 + lane] = nxt;
            if (c + 64 < v.n) nxt = v.entries[min(c + 64 + lane, v.n - 1)];
            PHX_SWEEP_WAVE_SYNC();
            const float4* blockp = my_stage + buf * 64;
            const int rel = i - c;                                   // candidate g of this block lies behind my row iff g > rel
            for (int g = 0; g < 64 && c + g < v.n && __ballot(!ended); g += SWEEP_GROUP) {
                // (the list holds SWEEP_CAND = 2 x SWEEP_GROUP positions and is emptied HERE, once per group, when the next eight might
                //  not fit: emptied where it fills up — inside the unrolled tests — the lookups' code stood eight times in the hot loop)
                if (ncand > SWEEP_CAND - SWEEP_GROUP) flush();
                float4 b[SWEEP_GROUP];
#pragma unroll
                for (int k = 0; k < SWEEP_GROUP; ++k) b[k] = blockp[g + k];
                // The tests are PREDICATED, not branched: a CU's waves share one scalar unit, and a branch is three or four scalar
                // instructions (compare-to-mask, s_and_saveexec, s_cbranch, the exec restore) — twelve waves walking 260 candidates
                // with three nested branches each were bound by it (~20 us of the kernel's 45).  Lane state lives in VGPRs as
                // all-ones / zero words; the only branch left is 'some lane of the wave overlaps this candidate in y'.
#pragma unroll
                for (int k = 0; k < SWEEP_GROUP; ++k) {
                    const int j = c + g + k;
                    const int started = (rel - (g + k)) >> 31;                          // -1 iff this candidate lies behind my row (j > i)
                    const int in_list = (j - v.n) >> 31;                                // -1 iff j < n
                    const int beyond = b[k].x > a.y ? -1 : 0;                           // ref: Collider.cpp:300-303: the row ends here
                    const int live = started & in_list & ~ended_w;
                    ended_w |= live & beyond;
                    const int tested = live & ~beyond;
                    len -= tested;
                    const int ov = (fabsf(b[k].z - a.z) <= a.w + b[k].w ? -1 : 0) & tested;
                    if (__any(ov)) {
                        if (ov) {
                            if (!EMIT) ++overlaps;
                            cand[ncand++][threadIdx.x] = (unsigned)j;
                        }
                    }
                }
                ended = ended_w != 0;
            }
            buf ^= 1;
        }
        if (scanning) {
            flush();
            if (!EMIT) { v.row_count[i] = found; tests += (unsigned long long)len; if (found > (unsigned)ROW_CACHE) *v.cache_overflow = 1; }
        }
    }
    if (!EMIT) {
        for (int off = 32; off > 0; off >>= 1) { tests += __shfl_down(tests, off); overlaps += __shfl_down(overlaps, off); }
        __shared__ unsigned long long part[2][4];
        if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = tests; part[1][threadIdx.x >> 6] = overlaps; }
        __syncthreads();
        if (threadIdx.x < 2) {                             // one atomic per workgroup per counter, 64 slots
            const unsigned long long t = part[threadIdx.x][0] + part[threadIdx.x][1] + part[threadIdx.x][2] + part[threadIdx.x][3];
            if (t) atomicAdd(&v.counters[2 * (blockIdx.x % STAT_SLOTS) + threadIdx.x], t);
        }
    }
}
// emit pass for every row with at most ROW_CACHE new pairs: straight from what the count pass remembered
__device__ __forceinline__ void emit_cached_rows(const SweepView& v, const unsigned* __restrict__ row_offset, uint2* __restrict__ out, unsigned total, int block, int blocks)
{
    for (int i = block * (int)blockDim.x + (int)threadIdx.x; i < v.n; i += blocks * (int)blockDim.x) {
        const unsigned dst = row_offset[i];
        const unsigned count = (i + 1 < v.n ? row_offset[i + 1] : total) - dst;
        if (count == 0 || count > (unsigned)ROW_CACHE) continue;     // nothing new, or a rescanned row
        // a hub row's pairs are emitted by its chunks and its row_cache entries were never written: skip it explicitly (the
        // same one-probe test as the count pass) instead of relying on k_sweep_chunks<true> overwriting the slots afterwards
        if (i + 1 + HUB_LEN < v.n && !(v.entries[i + 1 + HUB_LEN].x > v.entries[i].y)) continue;
        const unsigned ia = v.idx[i];
        for (unsigned k = 0; k < count; ++k) out[dst + k] = make_uint2(ia, v.row_cache[(size_t)i * ROW_CACHE + k]);
    }
}

// one workgroup per hub chunk; tiles of 256 candidates in j order, a running base keeps the emission order
template <bool EMIT>
__device__ __forceinline__ void sweep_chunks(const SweepView& v, const unsigned* __restrict__ row_offset, uint2* __restrict__ out, int block, int blocks)
{
    __shared__ unsigned wave_cnt[4];
    __shared__ unsigned running;
    const int total = min(*v.n_chunks, v.chunk_cap);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned long long overlaps_all = 0;
    for (int c = block; c < total; c += blocks) {
        const int4 ch = v.chunks[c];
        const float4 a = v.entries[ch.x];
        const unsigned ia = v.idx[ch.x];
        unsigned long long overlaps = 0;
        if constexpr (!EMIT) {
            // the count pass only wants the chunk's number of new pairs: every lane counts its own candidates over all tiles (their
            // loads and lookups in flight together) and the workgroup adds up once — no per-tile ordering, one barrier pair per chunk
            unsigned mine = 0;
            for (int j = ch.y + (int)threadIdx.x; j < ch.z; j += 256) {
                const float4 b = v.entries[j];
                if (fabsf(b.z - a.z) <= a.w + b.w) {
                    ++overlaps;
                    if (!ps_contains(v.table, v.mask, ((unsigned long long)ia << 32) | v.idx[j])) ++mine;
                }
            }
            for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
            if (lane == 0) wave_cnt[wave] = mine;
            __syncthreads();
            if (threadIdx.x == 0) v.chunk_count[c] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            overlaps_all += overlaps;
            __syncthreads();
            continue;
        }
        if (threadIdx.x == 0) running = row_offset[ch.x] + v.chunk_count[c];
        __syncthreads();
        for (int j0 = ch.y; j0 < ch.z; j0 += 256) {
            const int j = j0 + threadIdx.x;
            bool hit = false;
            unsigned ib = 0;
            if (j < ch.z) {
                const float4 b = v.entries[j];
                if (fabsf(b.z - a.z) <= a.w + b.w) {
                    ib = v.idx[j];
                    ++overlaps;
                    hit = !ps_contains(v.table, v.mask, ((unsigned long long)ia << 32) | ib);
                }
            }
            const unsigned long long bal = __ballot(hit);
            if (lane == 0) wave_cnt[wave] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned before = 0;
            for (int w = 0; w < wave; ++w) before += wave_cnt[w];
            const unsigned base = running;
            if (EMIT && hit) out[base + before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull))] = make_uint2(ia, ib);
            __syncthreads();
            if (threadIdx.x == 0) running = base + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
            __syncthreads();
        }
        __syncthreads();
    }
    if (!EMIT) {                                           // one statistics atomic per wave for the whole launch, not per chunk
        for (int off = 32; off > 0; off >>= 1) overlaps_all += __shfl_down(overlaps_all, off);
        if (lane == 0 && overlaps_all) atomicAdd(&v.counters[2 * ((block * 4 + wave) % STAT_SLOTS) + 1], overlaps_all);
    }
}

// the count pass over the hub rows' chunks
__global__ void __launch_bounds__(256) k_sweep_chunks_count(SweepView v) { sweep_chunks<false>(v, nullptr, nullptr, (int)blockIdx.x, (int)gridDim.x); }

// The emit pass, ONE launch: the first `row_blocks` workgroups emit the ordinary rows' new pairs from what the count pass remembered,
// the others the hub rows' chunks — different rows, different places in the list (two launches of ~5 us each at the dispatch floor).
__global__ void __launch_bounds__(256) k_emit_pairs(SweepView v, const unsigned* __restrict__ row_offset, uint2* __restrict__ out, unsigned total, int row_blocks)
{
    if ((int)blockIdx.x < row_blocks) emit_cached_rows(v, row_offset, out, total, (int)blockIdx.x, row_blocks);
    else sweep_chunks<true>(v, row_offset, out, (int)blockIdx.x - row_blocks, (int)gridDim.x - row_blocks);
}

// per hub row: chunk counts -> chunk bases inside the row, row_count[row] = sum.  A row's chunks are contiguous in the
// list, so differences of the exclusive scan of the counts give both.  The list is short (the ground row of the 200k-box
// scene is 391 chunks, of the 1M-box scene 1954), so ONE workgroup scans it tile by tile with a running carry and then takes the differences — one
// dispatch where a copy, a second copy, a scan and a difference kernel used to be four.
__global__ void __launch_bounds__(1024) k_chunk_bases(SweepView v, unsigned* __restrict__ scanned)
{
    __shared__ unsigned lds[16];
    __shared__ unsigned tot;
    const int total = min(*v.n_chunks, v.chunk_cap);
    unsigned carry = 0;
    for (int tile = 0; tile < total; tile += 1024) {
        const int c = tile + (int)threadIdx.x;
        const unsigned mine = c < total ? v.chunk_count[c] : 0u;
        const unsigned ex = block_exclusive_scan_1024(mine, lds, threadIdx.x == 0 ? &tot : nullptr);
        if (c < total) scanned[c] = carry + ex;
        __syncthreads();
        carry += tot;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    for (int c = threadIdx.x; c < total; c += 1024) {
        const int4 ch = v.chunks[c];
        const unsigned count = v.chunk_count[c];
        const bool last = (c + 1 == total) || v.chunks[c + 1].x != ch.x;
        if (last) v.row_count[ch.x] = scanned[c] + count - scanned[ch.w];
    }
    __syncthreads();                                                        // every count has been read: now overwrite them with the bases
    for (int c = threadIdx.x; c < total; c += 1024) v.chunk_count[c] = scanned[c] - scanned[v.chunks[c].w];
}

__global__ void __launch_bounds__(256) k_entries_to_aos(const float4* __restrict__ entries, const unsigned* __restrict__ keys,
                                                        const unsigned* __restrict__ idx, int n,
                                                        phx_broadphase_entry* __restrict__ out_entries, phx_sort_entry* __restrict__ out_sorted)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 e = entries[i];
        out_entries[i].minx = e.x; out_entries[i].maxx = e.y; out_entries[i].centery = e.z; out_entries[i].extenty = e.w;
        out_entries[i].index = idx[i];
        out_sorted[i].value = keys[i]; out_sorted[i].index = idx[i];
    }
}

// ---------------------------------------------------------------------------------------------------------

DeviceBroadphase::~DeviceBroadphase()
{
    if (hipSetDevice(device_) != hipSuccess) return;
    if (stream_) (void)hipStreamSynchronize(stream_);
    // (device buffers are DevBuf members: freed with the object)
    if (stream_) (void)hipStreamDestroy(stream_);
}

int DeviceBroadphase::init()
{
    PHX_TRY(use_device(device_));
    PHX_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    PHX_TRY(small_.reserve(16 + 2 * STAT_SLOTS));
    PHX_TRY(erase_count_.reserve(1));
    PHX_HIP(hipMemsetAsync(erase_count_.p, 0, sizeof(int), stream_));
    return clear();
}

int DeviceBroadphase::exclusive_scan(unsigned* data, int count, unsigned* total_out)
{
    return device_exclusive_scan(data, count, total_out, scan_tiles_, stream_);
}

int DeviceBroadphase::resize_table(unsigned want_cap)
{
    unsigned cap = 1024;
    while (cap < want_cap) cap <<= 1;
    DevBuf<unsigned long long> fresh;
    PHX_TRY(fresh.reserve(cap));
    PHX_HIP(hipMemsetAsync(fresh.p, 0xFF, (size_t)cap * sizeof(unsigned long long), stream_));
    if (table_.p && table_cap_) {
        hipLaunchKernelGGL(k_ps_rehash, dim3(grid_for((int)table_cap_)), dim3(256), 0, stream_, table_.p, table_cap_, fresh.p, cap - 1);
        PHX_HIP(hipGetLastError());
    }
    PHX_HIP(hipStreamSynchronize(stream_));
    table_ = std::move(fresh);
    table_cap_ = cap;
    tombstones_ = 0;
    return PHX_OK;
}

int DeviceBroadphase::clear()
{
    PHX_TRY(use_device(device_));
    table_.release();
    table_cap_ = 0;
    set_size_ = 0;
    tombstones_ = 0;
    if (erase_unchecked_) { erase_unchecked_ = 0; PHX_HIP(hipMemsetAsync(erase_count_.p, 0, sizeof(int), stream_)); }
    return resize_table(1024);
}

int DeviceBroadphase::update_device(const phx_rigid_body* d_bodies, int n)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(n >= 0 && (n == 0 || d_bodies), "bad body array");
    PHX_TRY(st_aabb_.reserve(std::max(n, 1)));
    if (n) hipLaunchKernelGGL(k_extract_aabb, dim3(grid_for(n)), dim3(256), 0, stream_, d_bodies, n, st_aabb_.p);
    PHX_HIP(hipGetLastError());
    return update_resident(st_aabb_.p, n);
}

int DeviceBroadphase::update_resident(const float4* d_bodies, int n, const StepPrologue* prologue, const std::function<int()>* while_waiting, const MailCarrier* carrier)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(n >= 0 && (n == 0 || d_bodies), "bad body array");
    n_ = n;
    last_new_ = 0;
    stats_ = phx_broadphase_stats{};
    for (int k = 0; k < 2; ++k) { PHX_TRY(keys_[k].reserve(std::max(n, 1))); PHX_TRY(idx_[k].reserve(std::max(n, 1))); }
    PHX_TRY(hist_.reserve(radix_hist_words(n)));
    int chunk_cap = std::max<int>((int)chunks_.cap, div_up(std::max(n, 1), HUB_CHUNK) * 8 + 64);   // grows on demand below
    PHX_TRY(chunks_.reserve(chunk_cap));
    PHX_TRY(chunk_count_.reserve(chunk_cap));
    PHX_TRY(entries_.reserve(std::max(n, 1)));
    PHX_TRY(row_count_.reserve(std::max(n, 1) + 1));
    PHX_TRY(row_cache_.reserve((size_t)std::max(n, 1) * ROW_CACHE));
    // keep the table at most half full counting tombstones, before anything reads it
    if ((unsigned long long)(set_size_ + tombstones_) * 2 > table_cap_) PHX_TRY(resize_table((unsigned)std::max<long long>(4 * set_size_, 1024)));

    PHX_TRY(stamps_.reserve(2));
    if (n == 0) {
        if (prologue) PHX_HIP(hipMemsetAsync(prologue->counters, 0, 8 * sizeof(unsigned), stream_));
        PHX_HIP(hipMemsetAsync(stamps_.p, 0, 2 * sizeof(unsigned long long), stream_));
        PHX_HIP(hipStreamSynchronize(stream_));
        stats_.candidate_tests = 0; stats_.overlapping_pairs = 0; stats_.new_pairs = 0; last_new_ = 0;
        stats_.set_size = (int)set_size_; have_update_ = true; ms_pending_ = true;
        return PHX_OK;
    }

    // The sort.  With last update's splitters on record (same body count): the two-level sort of splitter_sort.h — three launches;
    // otherwise (first update, another body count, buckets that got out of balance) the stable LSD radix sort — eleven — whose
    // gather leaves the splitters for the next update.  Both produce the reference's sorted sequence (ref: base/RadixSort.h:28-95).
    const int buckets = ss_buckets(n);
    PHX_TRY(splitters_.reserve(SS_MAX_BUCKETS)); PHX_TRY(ss_stats_.reserve(2));
    static const bool no_split = getenv("PHX_NO_SPLIT_SORT") != nullptr;      // A/B measurements, tests
    const bool split = !no_split && buckets <= SS_MAX_BUCKETS && (buckets == 1 || (splitters_n_ == n && !split_unbalanced_));
    int src = 0;
    if (split) {
        if (!ss_count_.p) { PHX_TRY(ss_count_.reserve(2 * SS_MAX_BUCKETS)); PHX_HIP(hipMemsetAsync(ss_count_.p, 0, ss_count_.cap * sizeof(unsigned), stream_)); }
        PHX_TRY(ss_base_.reserve(SS_MAX_BUCKETS + 1)); PHX_TRY(bucket_of_.reserve(n)); PHX_TRY(bucketed_.reserve(n));
        SplitSortView sv{};
        sv.aabb = d_bodies; sv.n = n; sv.buckets = buckets; sv.stride = ss_stride(n); sv.splitters = splitters_.p; sv.keys = keys_[0].p; sv.bucket_of = bucket_of_.p;
        sv.count = ss_count_.p; sv.cursor = ss_count_.p + SS_MAX_BUCKETS; sv.base = ss_base_.p; sv.bucketed = bucketed_.p;
        sv.keys_out = keys_[1].p; sv.idx_out = idx_[1].p; sv.entries = entries_.p; sv.next_splitters = splitters_.p; sv.max_bucket = ss_stats_.p;
        const int items = ss_tile_items(buckets);
        const dim3 tiles(div_up(n, SS_TILE_T * items));
#define PHX_KEYS_BUCKETS(INTEGRATE, ITEMS, ...) hipLaunchKernelGGL((k_keys_buckets<INTEGRATE, ITEMS>), tiles, dim3(SS_TILE_T), 0, stream_, sv, __VA_ARGS__)
        if (prologue) {
            if (items == SS_ITEMS_LARGE) PHX_KEYS_BUCKETS(true, SS_ITEMS_LARGE, prologue->vel, prologue->mpos, small_.p, 16 + 2 * STAT_SLOTS, chunk_count_.p, chunk_cap, stamps_.p, prologue->gravity, prologue->dt, prologue->counters, prologue->accel);
            else PHX_KEYS_BUCKETS(true, SS_ITEMS_SMALL, prologue->vel, prologue->mpos, small_.p, 16 + 2 * STAT_SLOTS, chunk_count_.p, chunk_cap, stamps_.p, prologue->gravity, prologue->dt, prologue->counters, prologue->accel);
        } else {
            if (items == SS_ITEMS_LARGE) PHX_KEYS_BUCKETS(false, SS_ITEMS_LARGE, (float4*)nullptr, (const float4*)nullptr, small_.p, 16 + 2 * STAT_SLOTS, chunk_count_.p, chunk_cap, stamps_.p, 0.f, 0.f, (unsigned*)nullptr, (const float4*)nullptr);
            else PHX_KEYS_BUCKETS(false, SS_ITEMS_SMALL, (float4*)nullptr, (const float4*)nullptr, small_.p, 16 + 2 * STAT_SLOTS, chunk_count_.p, chunk_cap, stamps_.p, 0.f, 0.f, (unsigned*)nullptr, (const float4*)nullptr);
        }
#undef PHX_KEYS_BUCKETS
        if (items == SS_ITEMS_LARGE) hipLaunchKernelGGL((k_bucket_scatter<SS_ITEMS_LARGE>), tiles, dim3(SS_TILE_T), 0, stream_, sv);
        else hipLaunchKernelGGL((k_bucket_scatter<SS_ITEMS_SMALL>), tiles, dim3(SS_TILE_T), 0, stream_, sv);
        // (the small LDS shape while the last update's largest bucket left room: all buckets resident at once)
        if (ss_last_max_ <= (unsigned)SS_LDS_SMALL_LIMIT) hipLaunchKernelGGL((k_bucket_sort<SS_LDS_SMALL>), dim3(buckets), dim3(SS_SORT_T), 0, stream_, sv);
        else hipLaunchKernelGGL((k_bucket_sort<SS_LDS_RECORDS>), dim3(buckets), dim3(SS_SORT_T), 0, stream_, sv);
        src = 1;
    } else {
        if (prologue) hipLaunchKernelGGL((k_build_keys<true>), dim3(grid_for(n)), dim3(256), 0, stream_, d_bodies, prologue->vel, prologue->mpos, n, keys_[0].p, idx_[0].p, small_.p, 16 + 2 * STAT_SLOTS,
                                         chunk_count_.p, chunk_cap, stamps_.p, prologue->gravity, prologue->dt, prologue->counters, prologue->accel);
        else hipLaunchKernelGGL((k_build_keys<false>), dim3(grid_for(n)), dim3(256), 0, stream_, d_bodies, (float4*)nullptr, (const float4*)nullptr, n, keys_[0].p, idx_[0].p, small_.p, 16 + 2 * STAT_SLOTS,
                                chunk_count_.p, chunk_cap, stamps_.p, 0.f, 0.f, (unsigned*)nullptr, (const float4*)nullptr);
        PHX_TRY(device_radix_sort_pairs(keys_[0].p, idx_[0].p, keys_[1].p, idx_[1].p, n, 32, hist_.p, scan_tiles_, stream_, &src));
        hipLaunchKernelGGL(k_gather_entries, dim3(grid_for(n)), dim3(256), 0, stream_, d_bodies, (const unsigned*)keys_[src].p, (const unsigned*)idx_[src].p, n, entries_.p, splitters_.p, ss_stride(n));
        PHX_HIP(hipMemsetAsync(ss_stats_.p, 0, sizeof(unsigned), stream_));
        split_unbalanced_ = false;
    }
    sorted_ = src;
    splitters_n_ = n;
    split_sorted_ = split;

    // sweep: count -> scan -> emit
    SweepView v{};
    v.entries = entries_.p; v.idx = idx_[src].p; v.n = n; v.table = table_.p; v.mask = table_cap_ - 1;
    v.row_count = row_count_.p; v.row_cache = row_cache_.p; v.cache_overflow = reinterpret_cast<int*>(small_.p + 4);
    v.n_chunks = reinterpret_cast<int*>(small_.p + 2);
    v.counters = small_.p + 16;
    unsigned long long host_small[16 + 2 * STAT_SLOTS] = {0};      // [2] chunks needed, [3] new pairs, [16..] statistics slots
    int chunk_grid = 1;
    for (int attempt = 0;; ++attempt) {
        v.chunks = chunks_.p; v.chunk_count = chunk_count_.p; v.chunk_cap = chunk_cap;
        chunk_grid = std::min(chunk_cap, 2048);
        hipLaunchKernelGGL((k_sweep_rows<false>), dim3(grid_for(n)), dim3(256), 0, stream_, v, (const unsigned*)nullptr, (uint2*)nullptr, 0u);
        hipLaunchKernelGGL(k_sweep_chunks_count, dim3(chunk_grid), dim3(256), 0, stream_, v);
        // chunk counts -> per-row bases and the hub rows' totals (one small workgroup)
        PHX_TRY(chunk_scan_.reserve(chunk_cap + 1));
        hipLaunchKernelGGL(k_chunk_bases, dim3(1), dim3(1024), 0, stream_, v, chunk_scan_.p);
        PHX_TRY(exclusive_scan(row_count_.p, n, reinterpret_cast<unsigned*>(small_.p + 3)));
        PHX_HIP(hipGetLastError());
        int erased = 0;
        PHX_TRY(queue_erase_check(&erased));
        PHX_TRY(rb_.add(host_small, small_.p, sizeof host_small, stream_));
        unsigned max_bucket = 0;
        if (split) PHX_TRY(rb_.add(&max_bucket, ss_stats_.p, sizeof max_bucket, stream_));
        PHX_TRY(rb_.wait(stream_, stamps_.p + 1, attempt == 0 ? while_waiting : nullptr, attempt == 0 ? carrier : nullptr));
        PHX_TRY(settle_erase_check(erased));
        ss_last_max_ = split ? max_bucket : 0u;
        if (split && max_bucket > (unsigned)(4 * ss_stride(n))) split_unbalanced_ = true;      // (stale splitters: the next update sorts the long way and takes fresh ones)
        const int needed = (int)(host_small[2] & 0xFFFFFFFFull);
        if (needed <= chunk_cap) break;
        // pathological overlap (many rows each spanning thousands of candidates): the chunk list was too short.
        // Its exact length is now known; grow it and redo the count pass.
        if (attempt) { set_error("broadphase: hub chunk list overflow twice (%d chunks)", needed); return PHX_ERR_CAPACITY; }
        chunk_cap = needed + 64;
        PHX_TRY(chunks_.reserve(chunk_cap));
        PHX_TRY(chunk_count_.reserve(chunk_cap));
        PHX_HIP(hipMemsetAsync(small_.p, 0, (16 + 2 * STAT_SLOTS) * sizeof(unsigned long long), stream_));
        PHX_HIP(hipMemsetAsync(chunk_count_.p, 0, (size_t)chunk_cap * sizeof(unsigned), stream_));
    }
    const unsigned total = (unsigned)host_small[3];
    stats_.candidate_tests = 0; stats_.overlapping_pairs = 0;
    for (int k = 0; k < STAT_SLOTS; ++k) { stats_.candidate_tests += (long long)host_small[16 + 2 * k]; stats_.overlapping_pairs += (long long)host_small[17 + 2 * k]; }
    stats_.new_pairs = (int)total;
    last_new_ = (int)total;
    if (total) {
        PHX_TRY(new_pairs_.reserve(total));
        if ((unsigned long long)(set_size_ + tombstones_ + (long long)total) * 2 > table_cap_)
            PHX_TRY(resize_table((unsigned)std::min<long long>(4ll * (set_size_ + (long long)total), 1ll << 30)));
        v.table = table_.p; v.mask = table_cap_ - 1;
        {
            const int row_blocks = grid_for(n), chunk_blocks = (host_small[2] & 0xFFFFFFFFull) ? std::min((int)(host_small[2] & 0xFFFFFFFFull), chunk_grid) : 0;
            hipLaunchKernelGGL(k_emit_pairs, dim3(row_blocks + chunk_blocks), dim3(256), 0, stream_, v, (const unsigned*)row_count_.p, new_pairs_.p, total, row_blocks);
        }
        if (host_small[4] & 0xFFFFFFFFull)      // some row found more than ROW_CACHE new pairs: those rows are rescanned
            hipLaunchKernelGGL((k_sweep_rows<true>), dim3(grid_for(n)), dim3(256), 0, stream_, v, (const unsigned*)row_count_.p, new_pairs_.p, total);
        // ref: Collider.cpp:313 / :341 — the emitted pairs join the persistent set
        hipLaunchKernelGGL(k_ps_insert, dim3(grid_for((int)total)), dim3(256), 0, stream_, table_.p, table_cap_ - 1, (const uint2*)new_pairs_.p, (int)total, stamps_.p + 1);
        PHX_HIP(hipGetLastError());
        set_size_ += total;
    }
    // (not synchronised here: what follows on this stream — insertions' consumers, the next phase of a World — is ordered
    //  behind it; get_stats / get_new_pairs / get_sorted synchronise before they read)
    stats_.set_size = (int)set_size_;
    have_update_ = true;
    ms_pending_ = true;
    return PHX_OK;
}

int DeviceBroadphase::update_host(const phx_rigid_body* bodies, int n, uint32_t* new_pairs, int cap, int* count)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(n >= 0 && (n == 0 || bodies), "bad body array");
    PHX_TRY(st_bodies_.reserve(std::max(n, 1)));
    if (n) PHX_HIP(hipMemcpyAsync(st_bodies_.p, bodies, (size_t)n * sizeof(phx_rigid_body), hipMemcpyHostToDevice, stream_));
    PHX_TRY(update_device(st_bodies_.p, n));
    PHX_HIP(hipStreamSynchronize(stream_));
    return get_new_pairs(new_pairs, cap, count);
}

int DeviceBroadphase::get_new_pairs(uint32_t* out, int cap, int* count)
{
    if (!have_update_) { set_error("no broadphase update has run yet"); return PHX_ERR_STATE; }
    if (count) *count = last_new_;
    if (!out) return PHX_OK;
    if (cap < last_new_) { set_error("new-pair buffer too small: need %d", last_new_); return PHX_ERR_CAPACITY; }
    PHX_TRY(use_device(device_));
    PHX_HIP(hipStreamSynchronize(stream_));
    if (last_new_) PHX_HIP(hipMemcpy(out, new_pairs_.p, (size_t)last_new_ * sizeof(uint2), hipMemcpyDeviceToHost));
    return PHX_OK;
}

int DeviceBroadphase::get_sorted(phx_sort_entry* sorted, phx_broadphase_entry* entries, int cap)
{
    if (!have_update_) { set_error("no broadphase update has run yet"); return PHX_ERR_STATE; }
    if (cap < n_) { set_error("sorted buffers too small: need %d", n_); return PHX_ERR_CAPACITY; }
    PHX_TRY(use_device(device_));
    if (!n_) return PHX_OK;
    DevBuf<phx_broadphase_entry> de;
    DevBuf<phx_sort_entry> ds;
    PHX_TRY(de.reserve(n_));
    PHX_TRY(ds.reserve(n_));
    hipLaunchKernelGGL(k_entries_to_aos, dim3(grid_for(n_)), dim3(256), 0, stream_, (const float4*)entries_.p, (const unsigned*)keys_[sorted_].p,
                       (const unsigned*)idx_[sorted_].p, n_, de.p, ds.p);
    PHX_HIP(hipGetLastError());
    if (entries) PHX_HIP(hipMemcpyAsync(entries, de.p, (size_t)n_ * sizeof(phx_broadphase_entry), hipMemcpyDeviceToHost, stream_));
    if (sorted) PHX_HIP(hipMemcpyAsync(sorted, ds.p, (size_t)n_ * sizeof(phx_sort_entry), hipMemcpyDeviceToHost, stream_));
    PHX_HIP(hipStreamSynchronize(stream_));
    de.release();
    ds.release();
    return PHX_OK;
}

int DeviceBroadphase::erase_pairs_device(const uint2* d_pairs, int count)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(count >= 0 && (count == 0 || d_pairs), "bad pair list");
    if (!count) return PHX_OK;
    hipLaunchKernelGGL(k_ps_erase, dim3(grid_for(count)), dim3(256), 0, stream_, table_.p, table_cap_ - 1, d_pairs, count, erase_count_.p);
    PHX_HIP(hipGetLastError());
    // Booked as if every pair was found (a World only erases pairs it inserted); the device counter says how many really
    // were, and the difference is settled with the next readback this handle makes anyway (reconcile_erases) — an erase
    // costs no host round trip of its own.  set_size_ + tombstones_, which sizes the table, is right either way.
    set_size_ -= count;
    tombstones_ += count;
    erase_unchecked_ += count;
    return PHX_OK;
}

// the pair set of a restored world (World::set_state): one pair per live manifold, as the broadphase would hold it (ref: Collider.h:58
// manifoldMap is keyed by the manifold's ordered body pair)
int DeviceBroadphase::reset_pairs(const uint2* pairs, int count)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(count >= 0 && (count == 0 || pairs), "bad pair list");
    PHX_TRY(clear());
    have_update_ = false;
    if (!count) return PHX_OK;
    PHX_TRY(resize_table(4u * (unsigned)count + 1024u));
    PHX_TRY(scratch_pairs_.reserve((size_t)count));
    PHX_HIP(hipMemcpyAsync(scratch_pairs_.p, pairs, (size_t)count * sizeof(uint2), hipMemcpyHostToDevice, stream_));
    hipLaunchKernelGGL(k_ps_insert, dim3(grid_for(count)), dim3(256), 0, stream_, table_.p, table_cap_ - 1, (const uint2*)scratch_pairs_.p, count, (unsigned long long*)nullptr);
    PHX_HIP(hipGetLastError());
    PHX_HIP(hipStreamSynchronize(stream_));          // (`pairs` is the caller's)
    set_size_ = count;
    stats_.set_size = count;
    return PHX_OK;
}

// queue the read of the device's erase counter with a batch the caller is about to wait for ...
int DeviceBroadphase::queue_erase_check(int* erased)
{
    *erased = 0;
    if (!erase_unchecked_) return PHX_OK;
    return rb_.add(erased, erase_count_.p, sizeof(int), stream_);
}

// ... and settle the books afterwards
int DeviceBroadphase::settle_erase_check(int erased)
{
    if (!erase_unchecked_) return PHX_OK;
    const long long missing = erase_unchecked_ - erased;             // requested pairs that were not in the set
    set_size_ += missing;
    tombstones_ = std::max<long long>(0, tombstones_ - missing);
    erase_unchecked_ = 0;
    PHX_HIP(hipMemsetAsync(erase_count_.p, 0, sizeof(int), stream_));
    return PHX_OK;
}

int DeviceBroadphase::erase_pairs(const uint32_t* pairs, int count)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(count >= 0 && (count == 0 || pairs), "bad pair list");
    if (!count) return PHX_OK;
    PHX_TRY(scratch_pairs_.reserve(count));
    PHX_HIP(hipMemcpyAsync(scratch_pairs_.p, pairs, (size_t)count * sizeof(uint2), hipMemcpyHostToDevice, stream_));
    return erase_pairs_device(scratch_pairs_.p, count);
}

int DeviceBroadphase::get_stats(phx_broadphase_stats* out)
{
    PHX_REQUIRE(out, "null out");
    if (!have_update_) { set_error("no broadphase update has run yet"); return PHX_ERR_STATE; }
    PHX_TRY(use_device(device_));
    if (ms_pending_) {
        unsigned long long stamps[2] = {0, 0};
        PHX_TRY(rb_.add(stamps, stamps_.p, sizeof stamps, stream_));
        PHX_TRY(rb_.wait(stream_));
        stats_.device_ms = stamps[1] > stamps[0] ? (double)(stamps[1] - stamps[0]) * 1e-5 : 0.0;      // 100 MHz ticks -> ms
        ms_pending_ = false;
    }
    if (erase_unchecked_) {
        int erased = 0;
        PHX_TRY(queue_erase_check(&erased));
        PHX_TRY(rb_.wait(stream_));
        PHX_TRY(settle_erase_check(erased));
    }
    *out = stats_;
    out->set_size = (int)set_size_;
    return PHX_OK;
}

} // namespace phx

// ---- C ABI ------------------------------------------------------------------------------------------------

extern "C" {

int phx_broadphase_create(phx_broadphase** out, int device)
{
    PHX_REQUIRE(out, "null out");
    *out = nullptr;
    PHX_TRY(phx::use_device(device));
    phx_broadphase* b = new (std::nothrow) phx_broadphase(device);
    PHX_REQUIRE(b, "out of host memory");
    int st = b->impl.init();
    if (st != PHX_OK) { delete b; return st; }
    *out = b;
    return PHX_OK;
}

void phx_broadphase_destroy(phx_broadphase* b) { delete b; }

int phx_broadphase_clear(phx_broadphase* b) { PHX_REQUIRE(b, "null handle"); return b->impl.clear(); }

int phx_broadphase_update(phx_broadphase* b, const phx_rigid_body* bodies, int32_t n, uint32_t* new_pairs, int32_t cap, int32_t* count)
{
    PHX_REQUIRE(b, "null handle");
    return b->impl.update_host(bodies, n, new_pairs, cap, count);
}

int phx_broadphase_update_device(phx_broadphase* b, const void* d_bodies, int32_t n)
{
    PHX_REQUIRE(b, "null handle");
    return b->impl.update_device(static_cast<const phx_rigid_body*>(d_bodies), n);
}

int phx_broadphase_get_sorted(phx_broadphase* b, phx_sort_entry* sorted, phx_broadphase_entry* entries, int32_t cap)
{
    PHX_REQUIRE(b, "null handle");
    return b->impl.get_sorted(sorted, entries, cap);
}

int phx_broadphase_get_new_pairs(phx_broadphase* b, uint32_t* new_pairs, int32_t cap, int32_t* count)
{
    PHX_REQUIRE(b, "null handle");
    return b->impl.get_new_pairs(new_pairs, cap, count);
}

int phx_broadphase_erase_pairs(phx_broadphase* b, const uint32_t* pairs, int32_t count)
{
    PHX_REQUIRE(b, "null handle");
    return b->impl.erase_pairs(pairs, count);
}

int phx_broadphase_get_stats(phx_broadphase* b, phx_broadphase_stats* out)
{
    PHX_REQUIRE(b, "null handle");
    return b->impl.get_stats(out);
}

} // extern "C"
