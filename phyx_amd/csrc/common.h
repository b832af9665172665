// common.h — shared plumbing of libphyx_amd: status codes, HIP error capture, POD checks.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/phyx_amd.h"

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
    int reserve(size_t n)
    {
        if (n <= cap) return PHX_OK;
        size_t want = n + n / 2 + 64;      // hipMalloc + hipFree cost ~0.2 ms each and synchronise the device: grow generously
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
        size_t want = n + n / 2 + 64;
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

// Small device->host readbacks (counts, flags, fingerprints) through one pinned staging buffer: asynchronous DMAs into
// pinned memory and ONE stream synchronisation per batch, instead of one blocking staged copy per value into pageable
// memory (measured: ~25 us of idle GPU per pageable readback, ~16 of them per world step).
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
            PHX_HIP(hipHostMalloc(reinterpret_cast<void**>(&pin_), cap_, hipHostMallocDefault));
        }
        const size_t off = (used_ + 15) & ~size_t(15);
        if (off + bytes > cap_ || count_ == MAX_ITEMS) {   // does not fit while DMAs are pending: plain copy now, bigger buffer next time
            if (off + bytes > cap_) want_ = std::max(want_, 2 * (off + bytes));
            PHX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
            return PHX_OK;
        }
        PHX_HIP(hipMemcpyAsync(pin_ + off, src, bytes, hipMemcpyDeviceToHost, stream));
        items_[count_++] = Item{dst, off, bytes};
        used_ = off + bytes;
        return PHX_OK;
    }
    int wait(hipStream_t stream)
    {
        PHX_HIP(hipStreamSynchronize(stream));
        for (int i = 0; i < count_; ++i) std::memcpy(items_[i].dst, pin_ + items_[i].off, items_[i].bytes);
        count_ = 0; used_ = 0;
        if (want_ > cap_) { (void)hipHostFree(pin_); pin_ = nullptr; }      // reallocated by the next add()
        return PHX_OK;
    }
private:
    static constexpr int MAX_ITEMS = 16;
    struct Item { void* dst; size_t off, bytes; };
    char* pin_ = nullptr;
    size_t cap_ = 0, used_ = 0, want_ = 0;
    int count_ = 0;
    Item items_[MAX_ITEMS];
};

} // namespace phx
