// device_radix.h — stable LSD radix sort of (u32 key, u32 value) pairs on the device, 8 bits per pass.
// Used by the broadphase (radixSort3 replacement, ref: base/RadixSort.h:28-95) and by the schedule builder.
#pragma once

#include "common.h"
#include "device_scan.h"

#include <algorithm>

namespace phx {

// ---- one radix pass = histogram -> scan -> scatter --------------------------------------------------
constexpr int RS_THREADS = 256;                 // 4 waves
constexpr int RS_ITEMS = 8;                     // keys per lane
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 2048 keys per workgroup
constexpr int RS_BINS = 256;

// per-workgroup digit histogram -> hist[digit * nblocks + block]
static __global__ void __launch_bounds__(RS_THREADS) k_radix_hist(const unsigned* __restrict__ keys, int n, int shift, int nblocks,
                                                           unsigned* __restrict__ hist)
{
    __shared__ unsigned h[RS_BINS];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int e = base + i * RS_THREADS + threadIdx.x;
        if (e < n) atomicAdd(&h[(keys[e] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// stable scatter of one 8-bit digit.  Element order inside the tile is (wave, item, lane) = index order,
// so ranks are assigned in that order: per item a wave-wide match on the digit gives each lane the number
// of equal digits in lower lanes; a wave-private LDS counter row carries the count across items; an
// exclusive scan over the 4 waves' rows orders the waves.
static __global__ void __launch_bounds__(RS_THREADS) k_radix_scatter(const unsigned* __restrict__ keys_in, const unsigned* __restrict__ idx_in,
                                                              unsigned* __restrict__ keys_out, unsigned* __restrict__ idx_out,
                                                              int n, int shift, int nblocks, const unsigned* __restrict__ hist)
{
    __shared__ unsigned cnt[4][RS_BINS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4 * RS_BINS; i += RS_THREADS) (&cnt[0][0])[i] = 0;
    __syncthreads();

    const int base = blockIdx.x * RS_TILE + wave * (64 * RS_ITEMS);
    unsigned key[RS_ITEMS], val[RS_ITEMS], rank[RS_ITEMS];
    volatile unsigned* my = cnt[wave];
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int e = base + i * 64 + lane;
        const bool live = e < n;
        key[i] = live ? keys_in[e] : 0xFFFFFFFFu;
        val[i] = live ? idx_in[e] : 0u;
        const unsigned d = (key[i] >> shift) & 255u;
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned lower = (unsigned)__popcll(peers & ((1ull << lane) - 1ull));
        const unsigned prior = live ? my[d] : 0u;
        __builtin_amdgcn_wave_barrier();
        if (live && lower == 0) my[d] = prior + (unsigned)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        rank[i] = prior + lower;
    }
    __syncthreads();
    // exclusive scan over waves per digit, plus the workgroup's global base for that digit
    {
        const int d = threadIdx.x;
        unsigned run = hist[d * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; ++w) { const unsigned c = cnt[w][d]; cnt[w][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int e = base + i * 64 + lane;
        if (e < n) {
            const unsigned d = (key[i] >> shift) & 255u;
            const unsigned dst = cnt[wave][d] + rank[i];
            keys_out[dst] = key[i];
            idx_out[dst] = val[i];
        }
    }
}


// Sorts n pairs by the low `bits` bits of the key (rounded up to whole 8-bit passes), ping-ponging between
// (k0,v0) and (k1,v1); *result_buffer = 0 or 1 tells where the sorted sequence ended up.
// hist scratch: 256 * ceil(n / RS_TILE) words; scan scratch: ceil(that / SCAN_TILE) words.
static inline int device_radix_sort_pairs(unsigned* k0, unsigned* v0, unsigned* k1, unsigned* v1, int n, int bits,
                                          unsigned* hist, unsigned* scan_scratch, hipStream_t stream, int* result_buffer)
{
    unsigned* kb[2] = {k0, k1};
    unsigned* vb[2] = {v0, v1};
    int src = 0;
    const int nblocks = std::max(1, div_up(n, RS_TILE));
    const int passes = std::max(1, div_up(bits, 8));
    if (n > 0)
        for (int pass = 0; pass < passes; ++pass) {
            const int shift = pass * 8;
            hipLaunchKernelGGL(k_radix_hist, dim3(nblocks), dim3(RS_THREADS), 0, stream, (const unsigned*)kb[src], n, shift, nblocks, hist);
            PHX_TRY(device_exclusive_scan(hist, RS_BINS * nblocks, nullptr, scan_scratch, stream));
            hipLaunchKernelGGL(k_radix_scatter, dim3(nblocks), dim3(RS_THREADS), 0, stream, (const unsigned*)kb[src], (const unsigned*)vb[src],
                               kb[src ^ 1], vb[src ^ 1], n, shift, nblocks, (const unsigned*)hist);
            src ^= 1;
        }
    PHX_HIP(hipGetLastError());
    *result_buffer = src;
    return PHX_OK;
}

} // namespace phx
