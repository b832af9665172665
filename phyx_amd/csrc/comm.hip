// comm.hip — the native RCCL transport of the island-sharded solve (BASELINE config 3).
//
// The reference merges its islands' results inside one address space (ref: src/Solver.cpp:86-91 parallelFor over islands, then
// FinishBodies :482-494 / FinishJoints :527-547); across GPUs the counterpart is ONE collective per step on the solver's stream:
// the all-gather of the ranks' packed results (exchange.h), which is also the per-step barrier of the north star.  This file lets a
// C / C++ caller of include/phyx_amd.h run that step without any Python: phx_comm_* wraps an RCCL communicator (one process per
// GPU, xGMI between them), phx_world_set_comm attaches it to a World, and phx_world_step_sharded is then one call per step —
// step_begin, ncclAllGather on the world's stream, step_end — with ncclCommGetAsyncError folded into the exchange status.
//
// RCCL is resolved at run time (dlopen librccl.so.1, like roctx in runtime.hip): the library has no link dependency on it, a box
// without it simply cannot create a communicator, and a process that already carries an RCCL (PyTorch's) shares that copy.
#include "handles.h"
#include "comm.h"

#include <dlfcn.h>

#include <chrono>
#include <future>
#include <mutex>
#include <string>
#include <thread>

namespace phx {

namespace {

// the slice of rccl.h this transport uses (ref: /opt/rocm/include/rccl/rccl.h — enum values are part of the NCCL ABI)
typedef struct ncclComm* ncclComm_t;
struct ncclUniqueId { char internal[PHX_COMM_ID_BYTES]; };
enum { ncclSuccess = 0, ncclInProgress = 7 };
enum { ncclUint8 = 1, ncclInt32 = 2 };
enum { ncclSum = 0, ncclMax = 2 };

struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(ncclUniqueId*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*CommAbort)(ncclComm_t) = nullptr;
    int (*CommGetAsyncError)(ncclComm_t, int*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int*) = nullptr;
    std::string why;                  // why the library is not usable (dlopen's message, or the symbol that is missing), captured where it happened
    bool ok = false;
};

Rccl& rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* env = getenv("PHX_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
            const char* e = dlerror();                 // (dlerror() clears the state it returns: read it once, here)
            r.why = e ? e : "dlopen failed";
        }
        if (!r.lib) return;
        auto sym = [&](const char* name) { return dlsym(r.lib, name); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
        r.CommGetAsyncError = reinterpret_cast<decltype(r.CommGetAsyncError)>(sym("ncclCommGetAsyncError"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.CommGetAsyncError && r.AllGather && r.AllReduce;
        r.why = r.ok ? "" : "librccl was loaded but lacks one of ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclCommGetAsyncError / ncclAllGather / ncclAllReduce";
    });
    return r;
}

int rccl_fail(const char* what, int code)
{
    Rccl& r = rccl();
    set_error("%s: RCCL error %d (%s)", what, code, r.GetErrorString ? r.GetErrorString(code) : "?");
    return PHX_ERR_HIP;
}

#define PHX_RCCL(call, what) do { const int rc_ = (call); if (rc_ != ncclSuccess) return rccl_fail(what, rc_); } while (0)

// how long a rank waits for its peers — in ncclCommInitRank, and wherever the host blocks on a collective — before it gives up
// with an error instead of hanging its launcher: PHX_COMM_TIMEOUT_S seconds (default 120)
double comm_timeout_s()
{
    static const double t = [] { const char* e = getenv("PHX_COMM_TIMEOUT_S"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 120.0; }();
    return t;
}

} // namespace

struct Comm::Impl { ncclComm_t comm = nullptr; };

Comm::~Comm()
{
    if (impl_ && impl_->comm && rccl().ok) { (void)hipSetDevice(device_); (void)rccl().CommDestroy(impl_->comm); }
    delete impl_;
    if (flag_) (void)hipFree(flag_);
}

int Comm::unique_id(void* out)
{
    Rccl& r = rccl();
    if (!r.ok) { set_error("RCCL is not available (%s)", r.why.c_str()); return PHX_ERR_NO_DEVICE; }
    ncclUniqueId id;
    PHX_RCCL(r.GetUniqueId(&id), "ncclGetUniqueId");
    std::memcpy(out, id.internal, PHX_COMM_ID_BYTES);
    return PHX_OK;
}

int Comm::init(const void* unique_id, int rank, int nranks)
{
    PHX_REQUIRE(unique_id && nranks >= 1 && rank >= 0 && rank < nranks, "bad communicator arguments");
    Rccl& r = rccl();
    if (!r.ok) { set_error("RCCL is not available (%s)", r.why.c_str()); return PHX_ERR_NO_DEVICE; }
    PHX_TRY(use_device(device_));
    rank_ = rank; nranks_ = nranks;
    impl_ = new (std::nothrow) Impl;
    PHX_REQUIRE(impl_, "out of host memory");
    ncclUniqueId id;
    std::memcpy(id.internal, unique_id, PHX_COMM_ID_BYTES);
    // ncclCommInitRank blocks until all `nranks` ranks have called it: a peer that never arrives (it crashed, it was given
    // another id) would hang this rank — and its launcher — for good.  It runs on a helper thread that is waited for with a
    // bound; a rank that gives up leaves the thread behind (it cannot be cancelled) and reports.
    struct Result { ncclComm_t comm = nullptr; int rc = ncclSuccess; };
    // (`abandoned`: the caller gave up.  Should the peers arrive after all, the helper owns the communicator it gets and aborts it
    //  at once — nobody else ever will, and a live communicator nobody drives would hold the device and its peers' bootstrap.)
    struct Pending { std::promise<Result> done; std::mutex m; bool abandoned = false, finished = false; };
    auto pend = std::make_shared<Pending>();
    std::future<Result> fut = pend->done.get_future();
    const int device = device_;
    std::thread([pend, device, nranks, id, rank] {
        Result res;
        (void)hipSetDevice(device);
        res.rc = rccl().CommInitRank(&res.comm, nranks, id, rank);
        bool orphan;
        { std::lock_guard<std::mutex> g(pend->m); pend->finished = true; orphan = pend->abandoned; }
        if (orphan) { if (res.rc == ncclSuccess && res.comm && rccl().CommAbort) (void)rccl().CommAbort(res.comm); return; }
        pend->done.set_value(res);
    }).detach();
    if (fut.wait_for(std::chrono::duration<double>(comm_timeout_s())) != std::future_status::ready) {
        bool late;
        { std::lock_guard<std::mutex> g(pend->m); late = pend->finished; if (!late) pend->abandoned = true; }
        if (!late) {
            set_error("ncclCommInitRank: rank %d of %d still waits for its peers after %.0f s (PHX_COMM_TIMEOUT_S) — a rank is missing or holds another communicator id", rank, nranks, comm_timeout_s());
            return PHX_ERR_STATE;
        }      // (else it finished while the bound ran out: its result is on the way)
    }
    const Result res = fut.get();
    if (res.rc != ncclSuccess) return rccl_fail("ncclCommInitRank", res.rc);
    impl_->comm = res.comm;
    PHX_HIP(hipMalloc(reinterpret_cast<void**>(&flag_), 8 * sizeof(int)));
    PHX_HIP(hipMemset(flag_, 0, 8 * sizeof(int)));
    return PHX_OK;
}

// bounded hipStreamSynchronize: a collective some peer never joins must not hang the host for good
int Comm::wait_stream(hipStream_t stream, const char* what)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spin = 0;; ++spin) {
        const hipError_t q = hipStreamQuery(stream);
        if (q == hipSuccess) return PHX_OK;
        if (q != hipErrorNotReady) { set_error("%s: %s", what, hipGetErrorString(q)); return PHX_ERR_HIP; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > comm_timeout_s()) {
            if (impl_ && impl_->comm && rccl().CommAbort) { (void)rccl().CommAbort(impl_->comm); impl_->comm = nullptr; }      // (later calls fail at once)
            set_error("%s: rank %d of %d still waits after %.0f s (PHX_COMM_TIMEOUT_S) — a peer never entered the collective; the communicator was aborted", what, rank_, nranks_, comm_timeout_s());
            return PHX_ERR_STATE;
        }
        if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
}

// what every rank must know before a collective whose size depends on the step: did a peer fail, and do all ranks mean the same
// byte count?  One all-reduce (max) of {status, bytes, -bytes} on `stream`.  Posting it costs the host nothing (the words travel as
// kernel arguments; round 4 staged them through a host copy and waited for the result in EVERY step, which put a host round trip
// back into a step that was stream-ordered end to end); reading it is the bounded wait.
static __global__ void k_agree_words(int* w, int a, int b, int c) { w[0] = a; w[1] = b; w[2] = c; w[3] = 0; }

int Comm::agree_post(int status, long long bytes, hipStream_t stream)
{
    PHX_REQUIRE(impl_ && impl_->comm, "communicator not initialised");
    PHX_TRY(use_device(device_));
    // (an out-of-range size is this rank's failure, reported INSIDE the collective: returning before it would leave the peers
    //  waiting out their whole time bound)
    const bool bad = bytes < 0 || bytes >= (1ll << 31);
    const bool failed = status != 0 || bad;
    hipLaunchKernelGGL(k_agree_words, dim3(1), dim3(1), 0, stream, flag_ + 4, failed ? (status ? status : 1) : 0, failed ? 0 : (int)bytes, failed ? -0x7FFFFFFF : -(int)bytes);
    PHX_HIP(hipGetLastError());
    PHX_RCCL(rccl().AllReduce(flag_ + 4, flag_ + 4, 4, ncclInt32, ncclMax, impl_->comm, stream), "ncclAllReduce");
    if (bad) { set_error("segment size out of range"); return PHX_ERR_INVALID; }
    return PHX_OK;
}

int Comm::agree_read(int* worst_status, long long* min_bytes, long long* max_bytes, hipStream_t stream)
{
    PHX_REQUIRE(impl_ && impl_->comm, "communicator not initialised");
    PHX_TRY(use_device(device_));
    PHX_TRY(wait_stream(stream, "agreement before the all-gather"));
    int h[4] = {0, 0, 0, 0};
    PHX_HIP(hipMemcpy(h, flag_ + 4, sizeof h, hipMemcpyDeviceToHost));
    if (worst_status) *worst_status = h[0];
    if (max_bytes) *max_bytes = h[1];
    if (min_bytes) *min_bytes = -(long long)h[2];      // (2^31 - 1 if every rank failed)
    return PHX_OK;
}

int Comm::agree(int status, long long bytes, int* worst_status, long long* min_bytes, long long* max_bytes, hipStream_t stream)
{
    const int posted = agree_post(status, bytes, stream);
    if (posted != PHX_OK && posted != PHX_ERR_INVALID) return posted;      // (INVALID: the collective was entered, with a failure status)
    PHX_TRY(agree_read(worst_status, min_bytes, max_bytes, stream));
    return posted;
}

double Comm::timeout_s() { return comm_timeout_s(); }

int Comm::version()
{
    int v = 0;
    if (rccl().ok && rccl().GetVersion) (void)rccl().GetVersion(&v);
    return v;
}

int Comm::all_gather(const void* d_send, void* d_recv, size_t bytes_per_rank, hipStream_t stream)
{
    PHX_REQUIRE(impl_ && impl_->comm, "communicator not initialised");
    PHX_TRY(use_device(device_));
    PHX_RCCL(rccl().AllGather(d_send, d_recv, bytes_per_rank, ncclUint8, impl_->comm, stream), "ncclAllGather");
    return PHX_OK;
}

// the pure-barrier variant of the north star: a 4-byte all-reduce (max) of a status word on `stream`; *d_word is in place
int Comm::all_reduce_max_int(int* d_word, hipStream_t stream)
{
    PHX_REQUIRE(impl_ && impl_->comm, "communicator not initialised");
    PHX_TRY(use_device(device_));
    PHX_RCCL(rccl().AllReduce(d_word, d_word, 1, ncclInt32, ncclMax, impl_->comm, stream), "ncclAllReduce");
    return PHX_OK;
}

int Comm::barrier_async(hipStream_t stream)
{
    PHX_HIP(hipMemsetAsync(flag_ + 1, 0, sizeof(int), stream));
    return all_reduce_max_int(flag_ + 1, stream);
}

int Comm::barrier(hipStream_t stream)
{
    PHX_HIP(hipMemsetAsync(flag_, 0, sizeof(int), stream));
    PHX_TRY(all_reduce_max_int(flag_, stream));
    return wait_stream(stream, "barrier");
}

// ncclCommGetAsyncError: 0 = healthy or still in progress, else the RCCL error code
int Comm::async_error(int* out)
{
    PHX_REQUIRE(impl_ && impl_->comm && out, "communicator not initialised");
    int e = ncclSuccess;
    PHX_RCCL(rccl().CommGetAsyncError(impl_->comm, &e), "ncclCommGetAsyncError");
    *out = (e == ncclSuccess || e == ncclInProgress) ? 0 : e;
    return PHX_OK;
}

} // namespace phx

// ---- C ABI ------------------------------------------------------------------------------------------------
extern "C" {

int phx_comm_unique_id(void* out_id)
{
    PHX_REQUIRE(out_id, "null out");
    return phx::Comm::unique_id(out_id);
}

int phx_comm_create(phx_comm** out, const void* unique_id, int32_t rank, int32_t nranks, int device)
{
    PHX_REQUIRE(out, "null out");
    *out = nullptr;
    PHX_TRY(phx::use_device(device));
    phx_comm* c = new (std::nothrow) phx_comm(device);
    PHX_REQUIRE(c, "out of host memory");
    const int st = c->impl.init(unique_id, rank, nranks);
    if (st != PHX_OK) { delete c; return st; }
    *out = c;
    return PHX_OK;
}

void phx_comm_destroy(phx_comm* c) { delete c; }

int phx_comm_rccl_version(void) { return phx::Comm::version(); }
int phx_comm_rank(phx_comm* c) { return c ? c->impl.rank() : -1; }
int phx_comm_size(phx_comm* c) { return c ? c->impl.size() : 0; }

int phx_comm_all_gather(phx_comm* c, const void* d_send, void* d_recv, size_t bytes_per_rank, void* stream)
{
    PHX_REQUIRE(c, "null handle");
    return c->impl.all_gather(d_send, d_recv, bytes_per_rank, static_cast<hipStream_t>(stream));
}

int phx_comm_barrier(phx_comm* c, void* stream)
{
    PHX_REQUIRE(c, "null handle");
    return c->impl.barrier(static_cast<hipStream_t>(stream));
}

int phx_comm_barrier_async(phx_comm* c, void* stream)
{
    PHX_REQUIRE(c, "null handle");
    return c->impl.barrier_async(static_cast<hipStream_t>(stream));
}

int phx_comm_async_error(phx_comm* c, int32_t* error)
{
    PHX_REQUIRE(c && error, "null handle / out");
    int e = 0;
    PHX_TRY(c->impl.async_error(&e));
    *error = e;
    return PHX_OK;
}

} // extern "C"
