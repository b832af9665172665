// common.h — shared plumbing of libphyx_amd: status codes, HIP error capture, POD checks.
#pragma once
#include <chrono>

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/phyx_amd.h"

// the sweeps' arithmetic contract (solver_kernels.h mul_add / mul_sub; phx_arith_mode): 1 = fused multiply-adds (default), 0 = source order
#ifndef PHX_ARITH_FMA
#define PHX_ARITH_FMA 1
#endif

static_assert(sizeof(phx_rigid_body) == 128, "RigidBody layout (ref: src/RigidBody.h:12-57)");
static_assert(sizeof(phx_contact_point) == 32, "ContactPoint layout (ref: src/Manifold.h:12-43)");
static_assert(sizeof(phx_manifold) == 16, "Manifold layout (ref: src/Manifold.h:45-67)");
static_assert(sizeof(phx_contact_joint) == 20, "ContactJoint layout (ref: src/Joints.h:6-23)");
static_assert(sizeof(phx_broadphase_entry) == 20, "BroadphaseEntry layout (ref: src/Collider.h:45-50)");
static_assert(sizeof(phx_sort_entry) == 8, "BroadphaseSortEntry layout (ref: src/Collider.h:52-56)");
static_assert(offsetof(phx_rigid_body, velocity) == 52 && offsetof(phx_rigid_body, inv_mass) == 88 &&
              offsetof(phx_rigid_body, xvector) == 96 && offsetof(phx_rigid_body, pos) == 112 &&
              offsetof(phx_rigid_body, displacing_velocity) == 68 && offsetof(phx_rigid_body, angular_velocity) == 76 &&
              offsetof(phx_rigid_body, displacing_angular_velocity) == 84, "RigidBody offsets");

namespace phx {

void set_error(const char* fmt, ...);
const char* last_error();

// Evaluates a HIP call; on failure records file:line + hipGetErrorString and returns PHX_ERR_HIP
// from the enclosing function.
#define PHX_HIP(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            ::phx::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return PHX_ERR_HIP;                                                                    \
        }                                                                                          \
    } while (0)

#define PHX_TRY(expr)            \
    do {                         \
        int st_ = (expr);        \
        if (st_ != PHX_OK) return st_; \
    } while (0)

#define PHX_REQUIRE(cond, msg)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::phx::set_error("%s", msg);       \
            return PHX_ERR_INVALID;            \
        }                                      \
    } while (0)

// Selects `device` after checking that a usable gfx950-class device exists.
int use_device(int device);

// Growable device buffer; never shrinks (the reference's scratch also only grows, ref: AlignedArray.h).
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    // owns its allocation: freed with the object (the owners' destructors select the device first; no hand-kept release lists)
    DevBuf() = default;
    ~DevBuf() { release(); }
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap) { o.p = nullptr; o.cap = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept { if (this != &o) { release(); p = o.p; cap = o.cap; o.p = nullptr; o.cap = 0; } return *this; }
    int reserve(size_t n)
    {
        if (n <= cap) return PHX_OK;
        size_t want = 2 * n + 64;          // hipMalloc + hipFree cost ~0.2 ms each and synchronise the device, and a growing world
                                           // regrows a score of arrays in the same step (a 3 ms hiccup): double — HBM is 288 GB
        T* np = nullptr;
        PHX_HIP(hipMalloc(reinterpret_cast<void**>(&np), want * sizeof(T)));
        if (p) (void)hipFree(p);
        p = np;
        cap = want;
        return PHX_OK;
    }
    // like reserve() but keeps the first `keep` elements
    int reserve_keep(size_t n, size_t keep, hipStream_t stream)
    {
        if (n <= cap) return PHX_OK;
        size_t want = 2 * n + 64;
        T* np = nullptr;
        PHX_HIP(hipMalloc(reinterpret_cast<void**>(&np), want * sizeof(T)));
        if (p && keep) {
            PHX_HIP(hipMemcpyAsync(np, p, keep * sizeof(T), hipMemcpyDeviceToDevice, stream));
            PHX_HIP(hipStreamSynchronize(stream));
        }
        if (p) (void)hipFree(p);
        p = np;
        cap = want;
        return PHX_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Growable pinned host staging for small uploads: a hipMemcpyAsync out of pageable memory is staged synchronously by the
// runtime, one copy dispatch per call; the caller packs its tables here and uploads them with ONE asynchronous copy.  The
// contents must stay untouched until that copy has run (callers reuse it only behind a wait on the same stream).
template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    size_t cap = 0;
    PinnedBuf() = default;
    ~PinnedBuf() { release(); }
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    int reserve(size_t n)
    {
        if (n <= cap) return PHX_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        const size_t want = n + n / 2 + 64;
        PHX_HIP(hipHostMalloc(reinterpret_cast<void**>(&p), want * sizeof(T), hipHostMallocCoherent | hipHostMallocMapped));
        cap = want;
        return PHX_OK;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

// small host -> device upload by a kernel that reads the (host-coherent, device-visible) pinned staging directly: a DMA of a
// few KB waits its turn in the copy engine's queue before anything moves; a dispatch reads them over PCIe in a few microseconds
__attribute__((unused)) static __global__ void __launch_bounds__(256) k_upload_words(unsigned* __restrict__ dst, const unsigned* __restrict__ src_host, int words)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x) dst[i] = src_host[i];
}

// scratch of device_exclusive_scan's single-pass form (device_scan.h): ticket word + one status word per tile + the call counter
struct ScanScratch {
    DevBuf<unsigned long long> state;
    unsigned epoch = 0;
    int prepare(int tiles, hipStream_t stream)
    {
        if ((size_t)tiles + 1 > state.cap || epoch >= (1u << 30) - 2) {
            PHX_TRY(state.reserve((size_t)tiles + 1));
            PHX_HIP(hipMemsetAsync(state.p, 0, state.cap * sizeof(unsigned long long), stream));
            epoch = 0;
        }
        ++epoch;
        return PHX_OK;
    }
    void release() { state.release(); epoch = 0; }
};

inline int div_up(int a, int b) { return (a + b - 1) / b; }

// roctx ranges named after the reference's MICROPROFILE scopes (ref: World.cpp:21, Solver.cpp:133-198, Collider.cpp:253-381),
// so that rocprofv3 --marker-trace groups this library's kernels by the reference's phases.  The marker library is resolved
// at run time (dlopen in runtime.hip: librocprofiler-sdk-roctx.so, then libroctx64.so) — no link dependency, and a no-op when
// it is absent.  A range costs two calls on the host; nothing is synchronised.
void roctx_push(const char* name);
void roctx_pop();
struct RoctxRange {
    explicit RoctxRange(const char* name) { roctx_push(name); }
    ~RoctxRange() { roctx_pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};

// Small device->host readbacks (counts, flags, fingerprints).  A batch of values is posted by ONE tiny kernel that
// copies them into host-coherent pinned memory and then stores the batch's sequence number behind a system-scope release;
// the host polls that word.  Compared with one DMA per value + hipStreamSynchronize (round 1: ~25-35 us of idle GPU per
// round trip, plus a ~4 us copy dispatch per value) this is one dispatch per batch and a wake-up bounded by the PCIe write.
// The post kernel runs behind everything queued on the stream, so seeing its sequence number also means that all earlier
// work of the stream has finished.  NOTE: sources are read when wait() runs, not when add() is called — nothing queued
// between the two may overwrite them.  Batches with a large item (> MAIL_WORDS in total) fall back to DMAs + a stream wait.
struct MailItem { const unsigned* src; unsigned off, words; };
struct MailArgs { MailItem it[16]; int count; unsigned seq; unsigned long long* stamp; };      // stamp (may be null): receives max(itself, the 100 MHz clock)

// the post itself, by ONE whole workgroup (k_post_mail, or the first workgroup of a CARRIER kernel: a kernel the caller was going to
// queue behind the post anyway takes the batch along in its arguments and posts it before its own work — a launch fewer per round trip;
// a post is a 4-5 us dispatch that moves a few words)
__device__ __forceinline__ void post_mail_block(const MailArgs& a, unsigned* __restrict__ host_words, unsigned* __restrict__ host_seq)
{
    for (int i = 0; i < a.count; ++i) {
        const unsigned* src = a.it[i].src;
        unsigned* dst = host_words + a.it[i].off;
        const unsigned words = a.it[i].words;
        const unsigned quads = (reinterpret_cast<uintptr_t>(src) & 15u) == 0 ? words / 4 : 0;
        for (unsigned q = threadIdx.x; q < quads; q += blockDim.x) reinterpret_cast<uint4*>(dst)[q] = reinterpret_cast<const uint4*>(src)[q];
        for (unsigned w = 4 * quads + threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
    }
    if (a.stamp && threadIdx.x == 0) atomicMax(a.stamp, (unsigned long long)wall_clock64());
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// what a carrier kernel takes: the batch (count == 0: nothing to post) and where it goes
struct MailRide { MailArgs args; unsigned* host_words; unsigned* host_seq; };
// Readback::wait's carrier: queues the kernel that carries `ride` (null: this batch goes by DMA, queue the kernel without a post)
using MailCarrier = std::function<int(const MailRide* ride)>;

static __global__ void __launch_bounds__(1024) k_post_mail(MailArgs a, unsigned* __restrict__ host_words, unsigned* __restrict__ host_seq)
{
    for (int i = 0; i < a.count; ++i) {
        const unsigned* src = a.it[i].src;
        unsigned* dst = host_words + a.it[i].off;             // (16-byte aligned: Readback::add)
        const unsigned words = a.it[i].words;
        // 16 bytes per store where the source allows it: a table of a few thousand words went over the link one dword per lane
        const unsigned quads = (reinterpret_cast<uintptr_t>(src) & 15u) == 0 ? words / 4 : 0;
        for (unsigned q = threadIdx.x; q < quads; q += blockDim.x) reinterpret_cast<uint4*>(dst)[q] = reinterpret_cast<const uint4*>(src)[q];
        for (unsigned w = 4 * quads + threadIdx.x; w < words; w += blockDim.x) dst[w] = src[w];
    }
    if (a.stamp && threadIdx.x == 0) atomicMax(a.stamp, (unsigned long long)wall_clock64());      // 'the stream got this far at ...' (device timing without events)
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

inline long long& wait_clock_ns() { static long long v = 0; return v; }
inline long long& wait_clock_calls() { static long long v = 0; return v; }

class Readback {
public:
    Readback() = default;
    Readback(const Readback&) = delete;
    Readback& operator=(const Readback&) = delete;
    ~Readback() { if (pin_) (void)hipHostFree(pin_); }
    // queue a copy of `bytes` from device memory `src` on `stream`; the value is delivered to `dst` by wait()
    int add(void* dst, const void* src, size_t bytes, hipStream_t stream)
    {
        if (!bytes) return PHX_OK;
        if (!pin_) {
            cap_ = std::max<size_t>(want_, 1u << 20);
            PHX_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin_), cap_ + 64, hipHostMallocCoherent | hipHostMallocMapped));
            *seq_word() = 0;
        }
        const size_t off = (used_ + 15) & ~size_t(15);
        if (off + bytes > cap_ || count_ == MAX_ITEMS) {   // does not fit while copies are pending: plain copy now, bigger buffer next time
            if (off + bytes > cap_) want_ = std::max(want_, 2 * (off + bytes));
            PHX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
            dma_ = true;
            return PHX_OK;
        }
        items_[count_++] = Item{dst, src, off, bytes};
        used_ = off + bytes;
        if ((bytes & 3u) || (reinterpret_cast<uintptr_t>(src) & 3u)) odd_ = true;
        return PHX_OK;
    }
    // A handle whose stream carries collectives (a sharded World, world.hip set_comm) bounds every host wait: a peer that died or stopped
    // stepping leaves the stream blocked behind an all-reduce / all-gather for good, and the waits below would spin with it
    // (ADVICE r5).  0 = unbounded (no collective can be queued on this handle's stream).
    void set_timeout(double seconds) { timeout_s_ = seconds; }
    // `stamp` (device pointer, may be null): the post kernel also leaves the clock there (atomicMax) — see k_post_mail
    // `while_waiting` (may be null): called once the batch is on its way and before the host starts to wait — whatever it queues
    // on the stream runs while the post crosses the link and the host digests it, instead of the GPU idling through the round trip
    // `carrier` (may be null; then `while_waiting` is not looked at): the kernel the caller queues behind the post takes the post along
    int wait(hipStream_t stream, unsigned long long* stamp = nullptr, const std::function<int()>* while_waiting = nullptr, const MailCarrier* carrier = nullptr)
    {
        // the batch is over however this function leaves: a failed wait must not keep destinations on a dead caller's stack
        struct Reset { Readback& r; ~Reset() { r.count_ = 0; r.used_ = 0; r.odd_ = false; r.dma_ = false;
                                               if (r.want_ > r.cap_ && r.pin_) { (void)hipHostFree(r.pin_); r.pin_ = nullptr; } } } reset{*this};      // (reallocated by the next add())
        const bool mail = count_ > 0 && !odd_ && !dma_ && used_ <= MAIL_WORDS * 4 && !no_mail();
        if (mail) {
            MailArgs a;
            a.count = count_; a.seq = ++seq_; a.stamp = stamp;
            for (int i = 0; i < count_; ++i) a.it[i] = MailItem{static_cast<const unsigned*>(items_[i].src), (unsigned)(items_[i].off / 4), (unsigned)(items_[i].bytes / 4)};
            if (carrier && used_ <= 16384 && !no_carrier()) {      // (batches of a few KB: the carrier's first workgroup is whatever size its kernel has)
                MailRide ride{a, reinterpret_cast<unsigned*>(pin_), seq_word()};
                PHX_TRY((*carrier)(&ride));
            } else {
                hipLaunchKernelGGL(k_post_mail, dim3(1), dim3(used_ > 4096 ? 1024 : (used_ > 1024 ? 256 : 64)), 0, stream, a, reinterpret_cast<unsigned*>(pin_), seq_word());
                PHX_HIP(hipGetLastError());
                if (carrier) PHX_TRY((*carrier)(nullptr));
                else if (while_waiting) PHX_TRY((*while_waiting)());
            }
            PHX_TRY(poll(stream));
        } else {
            for (int i = 0; i < count_; ++i) PHX_HIP(hipMemcpyAsync(pin_ + items_[i].off, items_[i].src, items_[i].bytes, hipMemcpyDeviceToHost, stream));
            if (carrier) PHX_TRY((*carrier)(nullptr));
            else if (while_waiting) PHX_TRY((*while_waiting)());
            if (timeout_s_ > 0) {
                const auto t0 = std::chrono::steady_clock::now();
                for (unsigned spins = 0;; ++spins) {
                    const hipError_t q = hipStreamQuery(stream);
                    if (q == hipSuccess) break;
                    if (q != hipErrorNotReady) { set_error("readback: %s", hipGetErrorString(q)); return PHX_ERR_HIP; }
                    if ((spins & 0xFFu) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s_) return timed_out();
                    __builtin_ia32_pause();
                }
            } else PHX_HIP(hipStreamSynchronize(stream));
        }
        for (int i = 0; i < count_; ++i) std::memcpy(items_[i].dst, pin_ + items_[i].off, items_[i].bytes);
        return PHX_OK;
    }
private:
    static constexpr int MAX_ITEMS = 16;
    static constexpr size_t MAIL_WORDS = 65536;          // 256 KB per batch through the post kernel (the settle of a 1M-box step reads ~100 KB of tables: six DMAs + a stream wait were 30 us)
    struct Item { void* dst; const void* src; size_t off, bytes; };
    unsigned* seq_word() const { return reinterpret_cast<unsigned*>(pin_ + cap_); }
    static bool no_mail() { static const bool off = std::getenv("PHX_NO_MAILBOX") != nullptr; return off; }
    static bool no_carrier() { static const bool off = std::getenv("PHX_NO_MAIL_CARRIER") != nullptr; return off; }
    int poll(hipStream_t stream)
    {
        volatile unsigned* word = seq_word();
        // PHX_WAIT_CLOCK=1 (diagnostics): how long the host spent waiting for mailbox posts, per process (printed by tools/world_quick.py)
        static const bool clocked = getenv("PHX_WAIT_CLOCK") != nullptr;
        const auto t0 = clocked ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        struct Stop { bool on; std::chrono::steady_clock::time_point t0; ~Stop() { if (on) { wait_clock_ns() += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); ++wait_clock_calls(); } } } stop{clocked, t0};
        const auto t_begin = timeout_s_ > 0 ? std::chrono::steady_clock::now() : std::chrono::steady_clock::time_point();
        for (unsigned long long spins = 1;; ++spins) {
            if (__atomic_load_n(const_cast<unsigned*>(word), __ATOMIC_ACQUIRE) == seq_) return PHX_OK;
            __builtin_ia32_pause();
            if ((spins & 0xFFFFu) == 0) {                // every ~65k polls: did the stream fail, or finish without posting?
                const hipError_t q = hipStreamQuery(stream);
                if (q == hipErrorNotReady) {
                    if (timeout_s_ > 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > timeout_s_) return timed_out();
                    continue;
                }
                if (q != hipSuccess) { set_error("readback: %s", hipGetErrorString(q)); return PHX_ERR_HIP; }
                if (__atomic_load_n(const_cast<unsigned*>(word), __ATOMIC_ACQUIRE) == seq_) return PHX_OK;
                set_error("readback: the stream drained without posting batch %u", seq_);
                return PHX_ERR_HIP;
            }
        }
    }
    int timed_out() const
    {
        set_error("readback: the stream is still busy after %.0f s (PHX_COMM_TIMEOUT_S) — it carries collectives, and a peer never entered one of them", timeout_s_);
        return PHX_ERR_STATE;
    }
    double timeout_s_ = 0.0;
    char* pin_ = nullptr;
    size_t cap_ = 0, used_ = 0, want_ = 0;
    int count_ = 0;
    unsigned seq_ = 0;
    bool odd_ = false, dma_ = false;
    Item items_[MAX_ITEMS];
};

} // namespace phx
