// splitter_sort.h — the broadphase's sort as a two-level sort on the 64-bit composite (key << 32 | body index).
//
// What it replaces: radixSort3 (ref: base/RadixSort.h:28-95) of {radixFloat(aabb.min.x), index} records — a STABLE sort by key of
// records whose index field is their initial position, i.e. exactly the plain sort of the composites, which are unique.  So
// no pass has to be stable and any two-level scheme is exact: the bodies are dealt into buckets by SPLITTERS (bucket b holds the
// composites between splitter b - 1 and splitter b), every bucket is sorted on its own in LDS, and the buckets are laid out one
// after the other.  The splitters are every SS_STRIDE-th record of the PREVIOUS update's sorted sequence: a body moves a few
// positions from one step to the next, so the buckets stay at ~SS_STRIDE records (they only balance the work — the result is the
// same sorted sequence for any splitters; a bucket that outgrows LDS is sorted in HBM by its workgroup, slowly and correctly, and
// the host then goes back to the LSD sort for an update to take fresh splitters).
// Three launches instead of eleven (keys, 3 x (histogram, scan, scatter), gather):
//   k_keys_buckets    key + IntegrateVelocity (what k_build_keys does) + the bucket of every body (binary search over the splitters in
//                     LDS) + the buckets' sizes (LDS histogram per tile, one global atomic per touched bucket and tile)
//   k_bucket_scatter  bucket bases (every tile scans the <= SS_MAX_BUCKETS sizes itself), a slot range per (tile, bucket) by one
//                     global atomic, the composites written into their buckets (any order inside a bucket)
//   k_bucket_sort     one workgroup per bucket: bitonic network in LDS, then the sorted records, the BroadphaseEntry gather
//                     (ref: Collider.cpp:269-283) and the next update's splitters
#pragma once

#include "common.h"

namespace phx {

constexpr int SS_STRIDE = 256;                 // records per bucket the splitters aim at (the bucket sort is by rank: quadratic in that)
constexpr int SS_MAX_BUCKETS = 4096;           // (LDS: the splitters, 8 bytes each, + a counter each)
// bodies per lane of the two tile kernels: every tile starts by loading the splitters (8 bytes per bucket) and clearing a counter per
// bucket, whatever it then does — at 1e6 bodies that is 31 KB + 16 KB of set-up in front of TWO bodies per lane (round 4: 31 + 28 us
// for 70 + 14 MB of traffic).  Small tiles where the table is small (2e5 bodies must still be a few hundred workgroups), eight
// bodies per lane where it is large.
constexpr int SS_TILE_T = 256, SS_ITEMS_SMALL = 2, SS_ITEMS_LARGE = 8, SS_LARGE_BUCKETS = 1024;
static inline int ss_tile_items(int buckets) { return buckets > SS_LARGE_BUCKETS ? SS_ITEMS_LARGE : SS_ITEMS_SMALL; }
constexpr int SS_SORT_T = 256;
// a bucket of at most that many records is sorted in LDS: 1024 (8 KB: every bucket's workgroup resident at once) while the last
// update's largest bucket stayed below SS_LDS_SMALL_LIMIT, else 4096 (32 KB: five workgroups per CU); larger buckets: in HBM
constexpr int SS_LDS_RECORDS = 4096, SS_LDS_SMALL = 1024, SS_LDS_SMALL_LIMIT = 768;

// the stride of a body count: SS_STRIDE while that makes at most SS_MAX_BUCKETS buckets, wider (in steps of 64) beyond
static inline int ss_stride(int n) { return std::max(SS_STRIDE, div_up(div_up(std::max(n, 1), SS_MAX_BUCKETS), 64) * 64); }
static inline int ss_buckets(int n) { return std::max(1, div_up(n, ss_stride(n))); }

struct SplitSortView {
    const float4* aabb; int n, buckets, stride;      // stride = ss_stride(n)
    const unsigned long long* splitters;       // buckets - 1 composites, ascending (last update's records at positions SS_STRIDE, 2 SS_STRIDE, ...)
    unsigned* keys;                            // scratch: key per body
    unsigned short* bucket_of;                 // scratch: bucket per body
    unsigned* count;                           // per bucket: records (zero on entry; k_bucket_sort leaves it zero again)
    unsigned* cursor;                          // per bucket: records placed so far (likewise)
    unsigned* base;                            // buckets + 1: first sorted position of every bucket
    unsigned long long* bucketed;              // n composites, bucket by bucket
    unsigned* keys_out; unsigned* idx_out;     // the sorted records (ref: Collider.h broadphaseSort)
    float4* entries;                           // {minx, maxx, centery, extenty} in sorted order
    unsigned long long* next_splitters;
    unsigned* max_bucket;                      // statistics: the largest bucket of this update (atomicMax)
};

__device__ __forceinline__ unsigned ss_radix_float(float v)      // radixFloat (ref: base/RadixSort.h:19-26)
{
    const int f = __float_as_int(v);
    const unsigned mask = (unsigned)(f >> 31) | 0x80000000u;
    return (unsigned)f ^ mask;
}

// bucket of composite c = number of splitters <= c
// (fixed trip count, no divergent loop: `steps` = bits of nspl; the lanes' searches run in lockstep and the items of a lane interleave)
__device__ __forceinline__ int ss_bucket(const unsigned long long* spl, int nspl, int steps, unsigned long long c)
{
    int lo = 0;                                   // invariant: every splitter before lo is <= c
    for (int s = steps - 1; s >= 0; --s) {
        const int probe = lo + (1 << s);          // is splitter probe - 1 <= c ?
        if (probe <= nspl && spl[probe - 1] <= c) lo = probe;
    }
    return lo;
}

// INTEGRATE: IntegrateVelocity (ref: World.cpp:39-55) rides along exactly as in k_build_keys (broadphase.hip)
template <bool INTEGRATE, int ITEMS>
__global__ void __launch_bounds__(SS_TILE_T) k_keys_buckets(SplitSortView v, float4* __restrict__ vel, const float4* __restrict__ mpos,
                                                            unsigned long long* __restrict__ small, int nsmall, unsigned* __restrict__ chunk_count, int nchunks,
                                                            unsigned long long* __restrict__ stamps, float gravity, float dt, unsigned* __restrict__ counters, const float4* __restrict__ accel)
{
    __shared__ unsigned long long spl[SS_MAX_BUCKETS];
    __shared__ unsigned hist[SS_MAX_BUCKETS];
    if (INTEGRATE && blockIdx.x == 0 && threadIdx.x < 8) counters[threadIdx.x] = 0u;
    if (blockIdx.x == 0 && threadIdx.x == 0) { stamps[0] = (unsigned long long)wall_clock64(); stamps[1] = 0ull; *v.max_bucket = 0u; }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nsmall; i += gridDim.x * blockDim.x) small[i] = 0ull;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nchunks; i += gridDim.x * blockDim.x) chunk_count[i] = 0u;
    const int tile0 = blockIdx.x * (SS_TILE_T * ITEMS);
    // every load of the lane's ITEMS bodies is issued before anything is done with one of them — and before the splitters are fetched: body after body, a lane of the
    // eight-body shape made eight dependent trips to memory (26 us at 1e6 bodies for 70 MB)
    float minx[ITEMS]; float4 w[ITEMS]; float im[ITEMS]; float4 acc[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int i = tile0 + k * SS_TILE_T + threadIdx.x;
        minx[k] = 0.f; w[k] = make_float4(0.f, 0.f, 0.f, 0.f); im[k] = 0.f; acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < v.n) {
            minx[k] = v.aabb[i].x;
            if (INTEGRATE) { w[k] = vel[i]; im[k] = mpos[i].x; if (accel) acc[k] = accel[i]; }
        }
    }
    const int nspl = v.buckets - 1;
    for (int i = threadIdx.x; i < nspl; i += SS_TILE_T) spl[i] = v.splitters[i];
    for (int i = threadIdx.x; i < v.buckets; i += SS_TILE_T) hist[i] = 0u;
    __syncthreads();
    int steps = 0;
    while ((1 << steps) <= nspl) ++steps;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int i = tile0 + k * SS_TILE_T + threadIdx.x;
        if (i >= v.n) continue;
        const unsigned key = ss_radix_float(minx[k]);
        const int b = ss_bucket(spl, nspl, steps, ((unsigned long long)key << 32) | (unsigned)i);
        v.keys[i] = key;
        v.bucket_of[i] = (unsigned short)b;
        atomicAdd(&hist[b], 1u);
        if (INTEGRATE) {
            float4 u = w[k];
            float ax = acc[k].x, ay = acc[k].y, aa = acc[k].z;
            if (im[k] > 0.0f) ay += gravity;
            u.x += ax * dt; u.y += ay * dt;
            u.z += aa * dt;
            vel[i] = u;
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < v.buckets; b += SS_TILE_T) if (hist[b]) atomicAdd(&v.count[b], hist[b]);
}

template <int ITEMS>
__global__ void __launch_bounds__(SS_TILE_T) k_bucket_scatter(SplitSortView v)
{
    __shared__ unsigned base[SS_MAX_BUCKETS];      // first position of every bucket, then + this tile's range inside it
    __shared__ unsigned local[SS_MAX_BUCKETS];     // records of this tile per bucket
    __shared__ unsigned wave_sum[SS_TILE_T / 64];
    // exclusive scan of the bucket sizes: SS_MAX_BUCKETS / SS_TILE_T consecutive buckets per lane
    constexpr int PER = SS_MAX_BUCKETS / SS_TILE_T;
    unsigned mine[PER], sum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) { const int b = threadIdx.x * PER + k; mine[k] = b < v.buckets ? v.count[b] : 0u; sum += mine[k]; }
    unsigned x = sum;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) wave_sum[wave] = x;
    __syncthreads();
    unsigned before = x - sum;
    for (int w = 0; w < wave; ++w) before += wave_sum[w];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int b = threadIdx.x * PER + k;
        if (b < v.buckets) { base[b] = before; local[b] = 0u; if (blockIdx.x == 0) v.base[b] = before; }
        before += mine[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == SS_TILE_T - 1) v.base[v.buckets] = before;
    __syncthreads();
    const int tile0 = blockIdx.x * (SS_TILE_T * ITEMS);
    unsigned key[ITEMS], off[ITEMS]; int bk[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int i = tile0 + k * SS_TILE_T + threadIdx.x;
        bk[k] = -1;
        if (i >= v.n) continue;
        key[k] = v.keys[i]; bk[k] = (int)v.bucket_of[i];
        off[k] = atomicAdd(&local[bk[k]], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < v.buckets; b += SS_TILE_T) if (local[b]) base[b] += atomicAdd(&v.cursor[b], local[b]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
        const int i = tile0 + k * SS_TILE_T + threadIdx.x;
        if (bk[k] >= 0) v.bucketed[base[bk[k]] + off[k]] = ((unsigned long long)key[k] << 32) | (unsigned)i;
    }
}

// the ascending bitonic network on n records behind accessor A (records past n count as +infinity: a pair that would touch one is
// skipped, which is what comparing with +infinity does); P = n rounded up to a power of two.  Pair q of a step touches records
// inside [2 q0, 2 q0 + 128) for the 64 pairs q0 .. q0 + 63 a wave takes in one trip whenever the step spans at most 128 records
// (flip of k <= 128, disperse of j <= 64): those steps need no workgroup barrier between them — a wave's own accesses are ordered —
// only the wide ones do: 9 barriers instead of 55 for 1024 records.
template <typename A>
__device__ __forceinline__ void ss_bitonic(A a, int n, int P)
{
    bool wide_before = true;                     // the records were written by other waves (the load)
    for (int lk = 1; (1 << lk) <= P; ++lk) {
        const int k = 1 << lk;
        {
            const bool wide = k > 128;
            if (wide || wide_before) a.sync(); else a.wave_sync();
            for (int q = threadIdx.x; q < P / 2; q += SS_SORT_T) {      // flip: i with its mirror image inside the block of k
                const int i = ((q >> (lk - 1)) << lk) + (q & ((k >> 1) - 1)), p = i ^ (k - 1);
                if (p < n) { const unsigned long long x = a.get(i), y = a.get(p); if (x > y) { a.set(i, y); a.set(p, x); } }
            }
            wide_before = wide;
        }
        for (int lj = lk - 2; lj >= 0; --lj) {
            const int j = 1 << lj;
            const bool wide = j > 64;
            if (wide || wide_before) a.sync(); else a.wave_sync();
            for (int q = threadIdx.x; q < P / 2; q += SS_SORT_T) {
                const int i = ((q >> lj) << (lj + 1)) + (q & (j - 1)), p = i + j;
                if (p < n) { const unsigned long long x = a.get(i), y = a.get(p); if (x > y) { a.set(i, y); a.set(p, x); } }
            }
            wide_before = wide;
        }
    }
    a.sync();
}

struct SsLds { unsigned long long* d; __device__ unsigned long long get(int i) const { return d[i]; } __device__ void set(int i, unsigned long long x) const { d[i] = x; }
               __device__ void sync() const { __syncthreads(); }
               __device__ void wave_sync() const { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } };
// (one workgroup owns the bucket: a barrier — which orders the workgroup's global accesses — is all the network needs)
struct SsHbm { unsigned long long* d; __device__ unsigned long long get(int i) const { return __hip_atomic_load(d + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
               __device__ void set(int i, unsigned long long x) const { __hip_atomic_store(d + i, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
               __device__ void sync() const { __threadfence(); __syncthreads(); }
               __device__ void wave_sync() const { sync(); } };

template <int LDS_RECORDS>
__global__ void __launch_bounds__(SS_SORT_T) k_bucket_sort(SplitSortView v)
{
    __shared__ __align__(16) unsigned long long rec[LDS_RECORDS];
    const int b = blockIdx.x;
    const unsigned first = v.base[b], m = v.base[b + 1] - first;
    // (the largest bucket, for the host: unbalanced splitters, and which LDS shape the next update's launch takes; buckets of the
    //  usual size stay away from the counter)
    if (threadIdx.x == 0) { v.count[b] = 0u; v.cursor[b] = 0u; if (m > (unsigned)(2 * v.stride)) atomicMax(v.max_bucket, m); }
    if (m == 0) return;
    int P = 2;
    while ((unsigned)P < m) P <<= 1;
    const bool in_lds = m <= (unsigned)LDS_RECORDS;
    unsigned long long* src = v.bucketed + first;
    if (in_lds) {
        // by RANK: the composites are distinct, so a record's sorted position is the number of records of the bucket below it — m
        // broadcast reads of LDS, two records each, independent of each other.  (A bitonic network in LDS is ~50 DEPENDENT steps of
        // ~0.25 us for ~1e3 records: 18 us per launch where this takes 8.)
        for (int i = threadIdx.x; i < (int)((m + 1u) & ~1u); i += SS_SORT_T) rec[i] = i < (int)m ? src[i] : ~0ull;
        __syncthreads();
    } else {
        ss_bitonic(SsHbm{src}, (int)m, P);
    }
    const ulonglong2* two = reinterpret_cast<const ulonglong2*>(rec);
    for (int i = threadIdx.x; i < (int)m; i += SS_SORT_T) {
        const unsigned long long c = in_lds ? rec[i] : __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned at = (unsigned)i;
        if (in_lds) {
            at = 0u;
            for (int q = 0; q < (int)((m + 1u) >> 1); ++q) { const ulonglong2 e = two[q]; at += (e.x < c) + (e.y < c); }
        }
        const unsigned p = first + at, body = (unsigned)c;
        v.keys_out[p] = (unsigned)(c >> 32);
        v.idx_out[p] = body;
        const float4 bb = v.aabb[body];                    // ref: Collider.cpp:269-283
        const float minx = bb.x, miny = bb.y, maxx = bb.z, maxy = bb.w;
        v.entries[p] = make_float4(minx, maxx, (miny + maxy) * 0.5f, (maxy - miny) * 0.5f);
        if (p && p % (unsigned)v.stride == 0u) v.next_splitters[p / (unsigned)v.stride - 1u] = c;
    }
}

} // namespace phx
