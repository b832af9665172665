// runtime.hip — error capture, device selection and the raw device-memory helpers of the C ABI.
#include "common.h"

#include <cstdlib>
#include <dlfcn.h>

namespace phx {

namespace {
struct RoctxApi {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi()
    {
        const char* off = getenv("PHX_NO_ROCTX");
        if (off && off[0] == '1') return;
        for (const char* lib : {"librocprofiler-sdk-roctx.so", "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "/opt/rocm/lib/libroctx64.so"}) {
            void* h = dlopen(lib, RTLD_LAZY | RTLD_LOCAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr; pop = nullptr;
        }
    }
};
const RoctxApi& roctx() { static const RoctxApi api; return api; }
} // namespace

void roctx_push(const char* name) { if (roctx().push) roctx().push(name); }
void roctx_pop() { if (roctx().pop) roctx().pop(); }

static thread_local char g_error[1024] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

const char* last_error() { return g_error; }

int use_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        set_error("no usable HIP device (hipGetDeviceCount: %s, count %d) — libphyx_amd has no CPU fallback",
                  e == hipSuccess ? "ok" : hipGetErrorString(e), count);
        return PHX_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        set_error("device %d out of range (have %d)", device, count);
        return PHX_ERR_INVALID;
    }
    PHX_HIP(hipSetDevice(device));
    return PHX_OK;
}

} // namespace phx

extern "C" {

int phx_abi_version(void) { return PHX_ABI_VERSION; }
int phx_arith_mode(void) { return PHX_ARITH_FMA ? PHX_ARITH_FUSED : PHX_ARITH_SOURCE; }
const char* phx_last_error(void) { return phx::last_error(); }

int phx_device_count(void)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess) {
        phx::set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
        return PHX_ERR_NO_DEVICE;
    }
    return count;
}

int phx_device_info(int device, char* name, int name_cap, int* compute_units, int* lds_bytes, int64_t* hbm_bytes)
{
    PHX_TRY(phx::use_device(device));
    hipDeviceProp_t p;
    PHX_HIP(hipGetDeviceProperties(&p, device));
    if (name && name_cap > 0) snprintf(name, (size_t)name_cap, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return PHX_OK;
}

int phx_device_malloc(int device, size_t bytes, void** out)
{
    PHX_REQUIRE(out, "null out");
    PHX_TRY(phx::use_device(device));
    PHX_HIP(hipMalloc(out, bytes ? bytes : 1));
    return PHX_OK;
}

int phx_device_free(int device, void* p)
{
    PHX_TRY(phx::use_device(device));
    if (p) PHX_HIP(hipFree(p));
    return PHX_OK;
}

int phx_memcpy_h2d(int device, void* dst, const void* src, size_t bytes)
{
    PHX_TRY(phx::use_device(device));
    if (bytes) PHX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return PHX_OK;
}

int phx_memcpy_d2h(int device, void* dst, const void* src, size_t bytes)
{
    PHX_TRY(phx::use_device(device));
    if (bytes) PHX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return PHX_OK;
}

// copies ordered on a caller-named stream (the host returns when the copy is done): for callers that stage device
// buffers through the host between two pieces of work queued on that stream (gloo transport of the sharded exchange)
int phx_memcpy_d2h_on(int device, void* dst, const void* src, size_t bytes, void* stream)
{
    PHX_TRY(phx::use_device(device));
    if (bytes) PHX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, static_cast<hipStream_t>(stream)));
    PHX_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PHX_OK;
}

int phx_memcpy_h2d_on(int device, void* dst, const void* src, size_t bytes, void* stream)
{
    PHX_TRY(phx::use_device(device));
    if (bytes) PHX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, static_cast<hipStream_t>(stream)));
    PHX_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return PHX_OK;
}

int phx_memcpy_d2d_on(int device, void* dst, const void* src, size_t bytes, void* stream)
{
    PHX_TRY(phx::use_device(device));
    if (bytes) PHX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return PHX_OK;
}

// diagnostics (PHX_WAIT_CLOCK=1): nanoseconds this process has spent waiting for mailbox posts, and the number of waits
int phx_debug_wait_clock(long long* ns, long long* calls)
{
    if (ns) *ns = phx::wait_clock_ns();
    if (calls) *calls = phx::wait_clock_calls();
    return PHX_OK;
}

int phx_memcpy_d2d(int device, void* dst, const void* src, size_t bytes)
{
    PHX_TRY(phx::use_device(device));
    if (bytes) PHX_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToDevice));
    return PHX_OK;
}

} // extern "C"
