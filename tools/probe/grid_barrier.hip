// Probe: what does a device-wide barrier cost on MI355X, per step, against a dependent kernel launch?
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier grid_barrier.hip && ./grid_barrier
// Every spin is bounded: a lost barrier ends the kernel with an error flag instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_barrier_loop(unsigned* counter, float* data, int n, int steps, int* error)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    for (int s = 0; s < steps; ++s) {
        for (int i = tid; i < n; i += nthreads) data[i] = data[(i * 7 + s) % n] * 0.5f + 1.0f;      // cross-workgroup traffic
        __threadfence();
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)(s + 1) * gridDim.x;
            long spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want)
                if (++spins > 20000000) { *error = 1; break; }
        }
        __syncthreads();
        if (*error) return;
    }
}

__global__ void k_one_step(float* data, int n, int s)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthreads = gridDim.x * blockDim.x;
    for (int i = tid; i < n; i += nthreads) data[i] = data[(i * 7 + s) % n] * 0.5f + 1.0f;
}

int main()
{
    const int n = 1 << 18, steps = 2000;
    unsigned* counter; float* data; int* error;
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&data, n * 4)); CHECK(hipMalloc(&error, 4));
    CHECK(hipMemset(data, 0, n * 4));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (int blocks : {32, 64, 128, 256, 512}) {
        for (int threads : {256, 1024}) {
            CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(error, 0, 4));
            int nn = n, ss = steps;
            void* args[] = {&counter, &data, &nn, &ss, &error};
            CHECK(hipEventRecord(a, st));
            hipError_t e = hipLaunchCooperativeKernel((const void*)k_barrier_loop, dim3(blocks), dim3(threads), args, 0, st);
            if (e != hipSuccess) { printf("blocks %d threads %d: cooperative launch refused (%s)\n", blocks, threads, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
            CHECK(hipEventRecord(b, st)); CHECK(hipStreamSynchronize(st));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            int herr; CHECK(hipMemcpy(&herr, error, 4, hipMemcpyDeviceToHost));
            printf("grid barrier: %4d blocks x %4d threads: %.2f us per step%s\n", blocks, threads, 1e3 * ms / steps, herr ? "  (BARRIER LOST)" : "");
        }
    }
    CHECK(hipEventRecord(a, st));
    for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(k_one_step, dim3(256), dim3(256), 0, st, data, n, s);
    CHECK(hipEventRecord(b, st)); CHECK(hipStreamSynchronize(st));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    printf("dependent launches: 256 blocks x 256 threads: %.2f us per step\n", 1e3 * ms / steps);
    return 0;
}
