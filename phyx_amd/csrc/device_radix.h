// device_radix.h — stable LSD radix sort of (u32 key, u32 value) pairs on the device, 8 or 11 bits per pass (the reference's
// radixSort3 also splits 11 / 11 / 10, ref: base/RadixSort.h:40-42): a pass is three dependent launches of a few microseconds, so
// fewer, wider passes win at the sizes of this path (32-bit keys: 3 passes instead of 4; up to 2048 bins: 1 instead of 2).
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
constexpr int RS_BINS = 256;                    // 8-bit passes
constexpr int RS_WIDE_BITS = 11, RS_WIDE_BINS = 1 << RS_WIDE_BITS;

// per-workgroup digit histogram -> hist[digit * nblocks + block]
template <int BITS>
static __global__ void __launch_bounds__(RS_THREADS) k_radix_hist(const unsigned* __restrict__ keys, int n, int shift, int nblocks,
                                                           unsigned* __restrict__ hist)
{
    constexpr int BINS = 1 << BITS;
    __shared__ unsigned h[BINS];
    for (int d = threadIdx.x; d < BINS; d += RS_THREADS) h[d] = 0;
    __syncthreads();
    const int base = blockIdx.x * RS_TILE;
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int e = base + i * RS_THREADS + threadIdx.x;
        if (e < n) atomicAdd(&h[(keys[e] >> shift) & (unsigned)(BINS - 1)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < BINS; d += RS_THREADS) hist[d * nblocks + blockIdx.x] = h[d];
}

// stable scatter of one BITS-wide digit.  Element order inside the tile is (wave, item, lane) = index order,
// so ranks are assigned in that order: per item a wave-wide match on the digit gives each lane the number
// of equal digits in lower lanes; a wave-private LDS counter row carries the count across items; an
// exclusive scan over the 4 waves' rows orders the waves.
template <int BITS>
static __global__ void __launch_bounds__(RS_THREADS) k_radix_scatter(const unsigned* __restrict__ keys_in, const unsigned* __restrict__ idx_in,
                                                              unsigned* __restrict__ keys_out, unsigned* __restrict__ idx_out,
                                                              int n, int shift, int nblocks, const unsigned* __restrict__ hist)
{
    constexpr int BINS = 1 << BITS;
    __shared__ unsigned cnt[4][BINS];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4 * BINS; i += RS_THREADS) (&cnt[0][0])[i] = 0;
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
        const unsigned d = (key[i] >> shift) & (unsigned)(BINS - 1);
        unsigned long long peers = __ballot(live);
#pragma unroll
        for (int b = 0; b < BITS; ++b) {
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
    for (int d = threadIdx.x; d < BINS; d += RS_THREADS) {
        unsigned run = hist[d * nblocks + blockIdx.x];
#pragma unroll
        for (int w = 0; w < 4; ++w) { const unsigned c = cnt[w][d]; cnt[w][d] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RS_ITEMS; ++i) {
        const int e = base + i * 64 + lane;
        if (e < n) {
            const unsigned d = (key[i] >> shift) & (unsigned)(BINS - 1);
            const unsigned dst = cnt[wave][d] + rank[i];
            keys_out[dst] = key[i];
            idx_out[dst] = val[i];
        }
    }
}


// Sorts n pairs by the low `bits` bits of the key, ping-ponging between (k0,v0) and (k1,v1); *result_buffer = 0 or 1 tells
// where the sorted sequence ended up.  Digits are 11 bits wide when that needs fewer passes than 8-bit digits (a stable LSD
// sort's result does not depend on the digit split), 8 bits otherwise.
// hist scratch: radix_hist_words(n) words; scan scratch: ceil(that / SCAN_TILE) words.
static inline size_t radix_hist_words(int n) { return (size_t)RS_WIDE_BINS * (size_t)std::max(1, div_up(n, RS_TILE)); }

// digit width the sort will use for keys of `bits` bits (a caller that takes the first pass's histogram itself needs to know)
static inline int radix_digit_bits(int bits) { return std::max(1, div_up(bits, RS_WIDE_BITS)) < std::max(1, div_up(bits, 8)) ? RS_WIDE_BITS : 8; }

// (`first_hist_done`: hist already holds the first pass's per-workgroup digit counts in k_radix_hist's layout — the caller's key
//  kernel counted them while it wrote the keys)
static inline int device_radix_sort_pairs(unsigned* k0, unsigned* v0, unsigned* k1, unsigned* v1, int n, int bits,
                                          unsigned* hist, ScanScratch& scan_scratch, hipStream_t stream, int* result_buffer, bool first_hist_done = false)
{
    unsigned* kb[2] = {k0, k1};
    unsigned* vb[2] = {v0, v1};
    int src = 0;
    const int nblocks = std::max(1, div_up(n, RS_TILE));
    const int passes8 = std::max(1, div_up(bits, 8)), passes11 = std::max(1, div_up(bits, RS_WIDE_BITS));
    const bool wide = passes11 < passes8;
    const int passes = wide ? passes11 : passes8, width = wide ? RS_WIDE_BITS : 8;
    if (n > 0)
        for (int pass = 0; pass < passes; ++pass) {
            const int shift = pass * width;
            if (pass == 0 && first_hist_done) {}
            else if (wide) hipLaunchKernelGGL((k_radix_hist<RS_WIDE_BITS>), dim3(nblocks), dim3(RS_THREADS), 0, stream, (const unsigned*)kb[src], n, shift, nblocks, hist);
            else           hipLaunchKernelGGL((k_radix_hist<8>), dim3(nblocks), dim3(RS_THREADS), 0, stream, (const unsigned*)kb[src], n, shift, nblocks, hist);
            PHX_TRY(device_exclusive_scan(hist, (wide ? RS_WIDE_BINS : RS_BINS) * nblocks, nullptr, scan_scratch, stream));
            if (wide) hipLaunchKernelGGL((k_radix_scatter<RS_WIDE_BITS>), dim3(nblocks), dim3(RS_THREADS), 0, stream, (const unsigned*)kb[src], (const unsigned*)vb[src],
                                         kb[src ^ 1], vb[src ^ 1], n, shift, nblocks, (const unsigned*)hist);
            else      hipLaunchKernelGGL((k_radix_scatter<8>), dim3(nblocks), dim3(RS_THREADS), 0, stream, (const unsigned*)kb[src], (const unsigned*)vb[src],
                                         kb[src ^ 1], vb[src ^ 1], n, shift, nblocks, (const unsigned*)hist);
            src ^= 1;
        }
    PHX_HIP(hipGetLastError());
    *result_buffer = src;
    return PHX_OK;
}

} // namespace phx
