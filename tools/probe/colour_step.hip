// colour_step.hip — what one colour step of the island kernel costs on MI355X, piece by piece.
// 512-thread workgroups; in step s only wave (s % 8) works: 2 x ds_read_b128, a dependent fp32 chain of CHAIN mul+add pairs,
// 2 x ds_write_b128; then every wave meets at a workgroup barrier.  Variants switch the pieces off.  Prints cycles per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <bool LDS, int CHAIN, bool BARRIER, bool ALLWAVES>
__global__ void __launch_bounds__(512, 8) k_steps(float4* out, unsigned long long* cycles, int steps, float a, float b)
{
    __shared__ float4 body[1024];
    const int tid = threadIdx.x, wave = tid >> 6;
    body[tid] = make_float4(tid * 0.001f, 1.f, 2.f, 3.f);
    body[tid + 512] = make_float4(tid * 0.002f, 1.f, 2.f, 3.f);
    __syncthreads();
    const int l1 = (tid * 7) & 1023, l2 = (tid * 13 + 5) & 1023;
    float acc = a;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        if (ALLWAVES || wave == (s & 7)) {
            float4 B1 = make_float4(acc, 1.f, 2.f, 3.f), B2 = B1;
            if (LDS) { B1 = body[l1]; B2 = body[l2]; }
            float x = B1.x + B2.y + acc;
#pragma unroll
            for (int k = 0; k < CHAIN; ++k) { x = x * a; x = x + b; }        // -ffp-contract=off: separate dependent ops
            acc = x;
            if (LDS) { B1.x = x; B2.y = x; body[l1] = B1; body[l2] = B2; }
        }
        if (BARRIER) __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + tid] = make_float4(acc, 0, 0, 0);
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <bool LDS, int CHAIN, bool BARRIER, bool ALLWAVES>
static void run(const char* name, int blocks)
{
    const int steps = 2000;
    float4* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 512 * sizeof(float4));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_steps<LDS, CHAIN, BARRIER, ALLWAVES>), dim3(blocks), dim3(512), 0, 0, out, cyc, steps, 1.0001f, 0.0001f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0; unsigned long long mx = 0;
    for (auto v : h) { sum += (double)v; mx = v > mx ? v : mx; }
    printf("%-58s blocks %4d: %7.1f cycles/step (mean), %7.1f (slowest block)\n", name, blocks, sum / blocks / steps, (double)mx / steps);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int blocks : {256, 1024}) {
        run<true, 26, true, false>("LDS + 26-pair chain + barrier, 1 of 8 waves works", blocks);
        run<true, 0, true, false>("LDS + barrier (no chain)", blocks);
        run<false, 26, true, false>("26-pair chain + barrier (no LDS)", blocks);
        run<false, 0, true, false>("barrier only", blocks);
        run<true, 26, false, false>("LDS + chain, no barrier (each wave every 8th step)", blocks);
        run<true, 26, true, true>("LDS + chain + barrier, ALL 8 waves work every step", blocks);
        run<true, 60, true, false>("LDS + 60-pair chain + barrier", blocks);
    }
    return 0;
}
