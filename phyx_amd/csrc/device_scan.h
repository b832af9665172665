// device_scan.h — exclusive prefix sum over u32 words on the device (shared by the broadphase and the world step).
#pragma once

#include "common.h"

namespace phx {

// ---- exclusive prefix sum over `count` words, in place ---------------------------------------------------
//   <= 64k words   k_scan_single: one workgroup, one launch
//   <= 4M words    k_scan_tiles (each 1024-lane workgroup scans a 4096-word tile in LDS and records the tile total)
//                  + k_scan_add_totals (each workgroup sums the totals before its tile and adds them): two launches
//   beyond         k_scan_tiles, k_scan_totals (one workgroup scans the tile totals), k_scan_add
constexpr int SCAN_TILE = 4096;

__device__ __forceinline__ unsigned block_exclusive_scan_1024(unsigned v, unsigned* lds, unsigned* total)
{
    // wave-level inclusive scan, then a 16-entry scan of the wave totals
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned x = v;
    for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
    if (lane == 63) lds[wave] = x;
    __syncthreads();
    if (wave == 0) {
        unsigned t = lane < 16 ? lds[lane] : 0u;
        for (int off = 1; off < 16; off <<= 1) { const unsigned y = __shfl_up(t, off); if (lane >= off) t += y; }
        if (lane < 16) lds[lane] = t;            // inclusive totals of waves 0..lane
    }
    __syncthreads();
    const unsigned before = wave ? lds[wave - 1] : 0u;
    if (total) *total = lds[15];
    return before + x - v;
}

static __global__ void __launch_bounds__(1024) k_scan_tiles(unsigned* __restrict__ data, int count, unsigned* __restrict__ tile_total)
{
    __shared__ unsigned lds[16];
    __shared__ unsigned tot;
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
    unsigned v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = (base + k < count) ? data[base + k] : 0u;
    const unsigned mine = v[0] + v[1] + v[2] + v[3];
    unsigned run = block_exclusive_scan_1024(mine, lds, threadIdx.x == 0 ? &tot : nullptr);
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (base + k < count) data[base + k] = run; run += v[k]; }
    __syncthreads();
    if (threadIdx.x == 0) tile_total[blockIdx.x] = tot;
}

static __global__ void __launch_bounds__(1024) k_scan_totals(unsigned* __restrict__ tile_total, int tiles, unsigned* __restrict__ grand_total)
{
    __shared__ unsigned lds[16];
    __shared__ unsigned tot;
    unsigned carry = 0;
    for (int b = 0; b < tiles; b += 1024) {                     // tiles <= 1024 in practice (4M words)
        const int i = b + threadIdx.x;
        const unsigned v = i < tiles ? tile_total[i] : 0u;
        const unsigned ex = block_exclusive_scan_1024(v, lds, threadIdx.x == 0 ? &tot : nullptr);
        if (i < tiles) tile_total[i] = carry + ex;
        __syncthreads();
        carry += tot;
        __syncthreads();
    }
    if (grand_total && threadIdx.x == 0) *grand_total = carry;
}

static __global__ void __launch_bounds__(1024) k_scan_add(unsigned* __restrict__ data, int count, const unsigned* __restrict__ tile_base)
{
    const unsigned add = tile_base[blockIdx.x];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < count) data[base + k] += add;
}

// second launch of the two-launch form (<= 1024 tiles): every workgroup sums the totals of the tiles before its own
// (at most 1024 words, one per lane) instead of waiting for a separate scan-of-totals launch
static __global__ void __launch_bounds__(1024) k_scan_add_totals(unsigned* __restrict__ data, int count, const unsigned* __restrict__ tile_total, int tiles,
                                                                 unsigned* __restrict__ grand_total)
{
    __shared__ unsigned part[16];
    unsigned x = ((int)threadIdx.x < (int)blockIdx.x) ? tile_total[threadIdx.x] : 0u;
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = x;
    __syncthreads();
    unsigned add = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) add += part[w];
    const int base = blockIdx.x * SCAN_TILE + threadIdx.x * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) if (base + k < count) data[base + k] += add;
    if (grand_total && (int)blockIdx.x == tiles - 1 && threadIdx.x == 0) *grand_total = add + tile_total[tiles - 1];
}

// Short inputs (radix histograms, per-bin tables: a few thousand words) are launch-latency bound, not bandwidth bound:
// one workgroup walks them tile by tile with a running carry — one launch instead of three.
constexpr int SCAN_SINGLE_MAX = 8 * SCAN_TILE;      // 32 words per lane

static __global__ void __launch_bounds__(1024) k_scan_single(unsigned* __restrict__ data, int count, unsigned* __restrict__ grand_total)
{
    // word i belongs to tile i / 4096, lane (i % 4096) / 4: every lane reads one 16-byte vector per tile (coalesced), all
    // tiles' loads in flight together; the tiles' block scans run side by side on one pair of barriers
    constexpr int TILES_MAX = SCAN_SINGLE_MAX / SCAN_TILE;
    __shared__ unsigned wave_total[TILES_MAX][16];
    const int tiles = (count + SCAN_TILE - 1) / SCAN_TILE;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint4 v[TILES_MAX];
    unsigned incl[TILES_MAX];
#pragma unroll
    for (int t = 0; t < TILES_MAX; ++t) {
        const int base = t * SCAN_TILE + threadIdx.x * 4;
        v[t] = make_uint4(0u, 0u, 0u, 0u);
        if (t < tiles) {
            if (base + 3 < count) v[t] = *reinterpret_cast<const uint4*>(data + base);
            else {
                if (base < count) v[t].x = data[base];
                if (base + 1 < count) v[t].y = data[base + 1];
                if (base + 2 < count) v[t].z = data[base + 2];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TILES_MAX; ++t) {
        unsigned x = v[t].x + v[t].y + v[t].z + v[t].w;
        for (int off = 1; off < 64; off <<= 1) { const unsigned y = __shfl_up(x, off); if (lane >= off) x += y; }
        incl[t] = x;
        if (lane == 63) wave_total[t][wave] = x;
    }
    __syncthreads();
    unsigned carry = 0;                                    // everything before the current tile
#pragma unroll
    for (int t = 0; t < TILES_MAX; ++t) {
        if (t >= tiles) break;
        unsigned before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const unsigned x = wave_total[t][w]; if (w < wave) before += x; total += x; }
        unsigned run = carry + before + incl[t] - (v[t].x + v[t].y + v[t].z + v[t].w);
        const int base = t * SCAN_TILE + threadIdx.x * 4;
        const uint4 o = make_uint4(run, run + v[t].x, run + v[t].x + v[t].y, run + v[t].x + v[t].y + v[t].z);
        if (base + 3 < count) *reinterpret_cast<uint4*>(data + base) = o;
        else {
            if (base < count) data[base] = o.x;
            if (base + 1 < count) data[base + 1] = o.y;
            if (base + 2 < count) data[base + 2] = o.z;
        }
        carry += total;
    }
    if (grand_total && threadIdx.x == 0) *grand_total = carry;
}

// scratch must hold div_up(count, SCAN_TILE) words.  total_out (device pointer, may be null) receives the sum.
static inline int device_exclusive_scan(unsigned* data, int count, unsigned* total_out, unsigned* scratch, hipStream_t stream)
{
    if (count <= 0) { if (total_out) PHX_HIP(hipMemsetAsync(total_out, 0, sizeof(unsigned), stream)); return PHX_OK; }
    if (count <= SCAN_SINGLE_MAX && (reinterpret_cast<uintptr_t>(data) & 15u) == 0) {
        hipLaunchKernelGGL(k_scan_single, dim3(1), dim3(1024), 0, stream, data, count, total_out);
        PHX_HIP(hipGetLastError());
        return PHX_OK;
    }
    const int tiles = div_up(count, SCAN_TILE);
    hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(1024), 0, stream, data, count, scratch);
    if (tiles <= 1024) {
        hipLaunchKernelGGL(k_scan_add_totals, dim3(tiles), dim3(1024), 0, stream, data, count, (const unsigned*)scratch, tiles, total_out);
        PHX_HIP(hipGetLastError());
        return PHX_OK;
    }
    hipLaunchKernelGGL(k_scan_totals, dim3(1), dim3(1024), 0, stream, scratch, tiles, total_out);
    if (tiles > 1) hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(1024), 0, stream, data, count, (const unsigned*)scratch);
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

} // namespace phx
