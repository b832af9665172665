// device_scan.h — exclusive prefix sum over u32 words on the device (shared by the broadphase and the world step).
#pragma once

#include "common.h"

#include <type_traits>
#include <utility>

namespace phx {

// ---- exclusive prefix sum over `count` words, in place ---------------------------------------------------
//   <= 32k words   k_scan_single: one workgroup, one launch
//   beyond         k_scan_lookback: ONE launch — every 1024-lane workgroup scans a 4096-word tile in LDS, publishes the tile's
//                  total, and finds the sum of the tiles before it by looking back over what its predecessors published
//                  (Merrill & Garland's decoupled look-back; a launch fewer than scan-tiles + add-totals, ~4 us each here)
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

// ---- where the scanned words come from ------------------------------------------------------------------------
// The scan kernels take a LOADER: `ScanInPlace` reads the words from the array they are scanned into; any other loader
// computes word i on the fly (`unsigned operator()(int i) const`, called exactly once per i) — the flag / count kernel that
// would otherwise run in front of the scan as a dispatch of its own (3-5 us each at the dispatch floor).
struct ScanInPlace { static constexpr bool in_place = true; __device__ unsigned operator()(int) const { return 0u; } };

template <typename Load, typename = void> struct scan_has_load4 : std::false_type {};
template <typename Load> struct scan_has_load4<Load, std::void_t<decltype(std::declval<const Load&>().load4(0, std::declval<uint4&>()))>> : std::true_type {};

template <typename Load>
__device__ __forceinline__ uint4 scan_load4(const Load& load, const unsigned* __restrict__ data, int base, int count)
{
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if constexpr (Load::in_place) {
        if ((reinterpret_cast<uintptr_t>(data) & 15u) == 0 && base + 3 < count) v = *reinterpret_cast<const uint4*>(data + base);
        else {
            if (base < count) v.x = data[base];
            if (base + 1 < count) v.y = data[base + 1];
            if (base + 2 < count) v.z = data[base + 2];
            if (base + 3 < count) v.w = data[base + 3];
        }
    } else {
        // (a loader may offer `bool load4(int base, uint4& out) const`: words base .. base + 3 in one go — 16-byte loads of what it
        //  reads instead of four 4-byte ones — returning false where it cannot: a ragged end, an unaligned array)
        if constexpr (scan_has_load4<Load>::value) { if (base + 3 < count && load.load4(base, v)) return v; }
        if (base < count) v.x = load(base);
        if (base + 1 < count) v.y = load(base + 1);
        if (base + 2 < count) v.z = load(base + 2);
        if (base + 3 < count) v.w = load(base + 3);
    }
    return v;
}

__device__ __forceinline__ void scan_store4(unsigned* __restrict__ data, int base, int count, uint4 o)
{
    if ((reinterpret_cast<uintptr_t>(data) & 15u) == 0 && base + 3 < count) *reinterpret_cast<uint4*>(data + base) = o;
    else {
        if (base < count) data[base] = o.x;
        if (base + 1 < count) data[base + 1] = o.y;
        if (base + 2 < count) data[base + 2] = o.z;
        if (base + 3 < count) data[base + 3] = o.w;
    }
}

// ---- single-pass scan with decoupled look-back ------------------------------------------------------------
// state[0] = ticket counter (tiles are taken in ticket order, so a tile's predecessors are always running or done: the
// look-back cannot wait for a workgroup that was never scheduled); state[1 + t] = status of tile t, ONE 64-bit word
// {epoch:30, flag:2, value:32} written with a single store, so value and flag can never be seen apart.  The epoch is the
// host's call counter: words of earlier calls read as 'not published yet', which is why nothing has to be cleared between
// calls (a memset is a dispatch of its own).  Agent-scope accesses: the tiles run on different XCDs (non-coherent L2s).
constexpr unsigned SCAN_AGGREGATE = 1u, SCAN_PREFIX = 2u;
constexpr int SCAN_SPIN_LIMIT = 1 << 22;                 // ~seconds; then trap: a HIP error the host reports instead of a hung device

template <typename Load>
static __global__ void __launch_bounds__(1024) k_scan_lookback(Load load, unsigned* __restrict__ data, int count, unsigned long long* __restrict__ state, unsigned epoch,
                                                               unsigned* __restrict__ grand_total)
{
    __shared__ unsigned lds[16];
    __shared__ unsigned s_tile, s_prefix;
    const int tiles = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_tile = (unsigned)atomicAdd(&state[0], 1ull);
    __syncthreads();
    const int tile = (int)s_tile;
    const int base = tile * SCAN_TILE + threadIdx.x * 4;
    const uint4 v = scan_load4(load, data, base, count);
    const unsigned run = block_exclusive_scan_1024(v.x + v.y + v.z + v.w, lds, nullptr);
    const unsigned total = lds[15];                        // (stable: nothing writes lds[] below)
    unsigned long long* status = state + 1;
    if (threadIdx.x == 0) {
        const unsigned flag = tile == 0 ? SCAN_PREFIX : SCAN_AGGREGATE;
        __hip_atomic_store(&status[tile], ((unsigned long long)((epoch << 2) | flag) << 32) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tile == tiles - 1) __hip_atomic_store(&state[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every ticket is out: ready for the next call
        if (tile == 0) s_prefix = 0u;
    }
    if (wave == 0 && tile > 0) {
        unsigned exclusive = 0;
        int idx = tile - 1 - lane;                         // the predecessor this lane inspects in the current window of 64
        for (int spins = 0;;) {
            unsigned long long w = 0;
            bool ready = true, prefix = true;              // lanes before tile 0: nothing there, which is a prefix of 0
            if (idx >= 0) {
                w = __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned tag = (unsigned)(w >> 32);
                ready = (tag >> 2) == epoch && (tag & 3u) != 0u;
                prefix = ready && (tag & 3u) == SCAN_PREFIX;
            }
            const unsigned long long pm = __ballot(prefix), nr = __ballot(!ready);
            const int nearest = pm ? __builtin_ctzll(pm) : 63;            // lanes 0..nearest are needed
            const unsigned long long need = nearest >= 63 ? ~0ull : ((2ull << nearest) - 1ull);
            if (nr & need) {
                if (++spins > SCAN_SPIN_LIMIT) __builtin_trap();
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            unsigned x = (lane <= nearest && idx >= 0) ? (unsigned)w : 0u;
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
            exclusive += __shfl(x, 0);
            if (pm) break;
            idx -= 64;
        }
        if (lane == 0) {
            __hip_atomic_store(&status[tile], ((unsigned long long)((epoch << 2) | SCAN_PREFIX) << 32) | (exclusive + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_prefix = exclusive;
        }
    }
    __syncthreads();
    const unsigned r = s_prefix + run;
    const uint4 o = make_uint4(r, r + v.x, r + v.x + v.y, r + v.x + v.y + v.z);
    scan_store4(data, base, count, o);
    if (grand_total && tile == tiles - 1 && threadIdx.x == 0) *grand_total = s_prefix + total;
}

// Short inputs (radix histograms, per-bin tables: a few thousand words) are launch-latency bound, not bandwidth bound:
// one workgroup walks them tile by tile with a running carry — one launch instead of three.
constexpr int SCAN_SINGLE_MAX = 8 * SCAN_TILE;      // 32 words per lane

template <typename Load>
static __global__ void __launch_bounds__(1024) k_scan_single(Load load, unsigned* __restrict__ data, int count, unsigned* __restrict__ grand_total)
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
        v[t] = t < tiles ? scan_load4(load, data, base, count) : make_uint4(0u, 0u, 0u, 0u);
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
        scan_store4(data, base, count, o);
        carry += total;
    }
    if (grand_total && threadIdx.x == 0) *grand_total = carry;
}

// total_out (device pointer, may be null) receives the sum.  `load` computes the words (see the loaders above); the exclusive
// prefix sums land in data[0 .. count).
template <typename Load>
static inline int device_exclusive_scan_of(Load load, unsigned* data, int count, unsigned* total_out, ScanScratch& scratch, hipStream_t stream)
{
    if (count <= 0) { if (total_out) PHX_HIP(hipMemsetAsync(total_out, 0, sizeof(unsigned), stream)); return PHX_OK; }
    if (count <= SCAN_SINGLE_MAX) {
        hipLaunchKernelGGL((k_scan_single<Load>), dim3(1), dim3(1024), 0, stream, load, data, count, total_out);
        PHX_HIP(hipGetLastError());
        return PHX_OK;
    }
    const int tiles = div_up(count, SCAN_TILE);
    PHX_TRY(scratch.prepare(tiles, stream));
    hipLaunchKernelGGL((k_scan_lookback<Load>), dim3(tiles), dim3(1024), 0, stream, load, data, count, scratch.state.p, scratch.epoch, total_out);
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

static inline int device_exclusive_scan(unsigned* data, int count, unsigned* total_out, ScanScratch& scratch, hipStream_t stream)
{
    return device_exclusive_scan_of(ScanInPlace{}, data, count, total_out, scratch, stream);
}

} // namespace phx
