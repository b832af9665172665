// broadphase.h — host side of the device broadphase (kernels + layout notes: broadphase.hip).
#pragma once

#include "common.h"

namespace phx {

class DeviceBroadphase {
public:
    explicit DeviceBroadphase(int device) : device_(device) {}
    ~DeviceBroadphase();
    int init();
    int clear();
    // `prologue` (World only): IntegrateVelocity (ref: World.cpp:39-55) rides on the update's first kernel — the key of a body is
    // its AABB's min x, which the velocity step does not touch — and the step's four counters are cleared on the way: a dispatch fewer
    struct StepPrologue { float gravity, dt; unsigned* counters; float4* vel; const float4* mpos; const float4* accel; };      // (resident arrays, body_view.h)
    // the resident form: one float4 {min.x, min.y, max.x, max.y} per body (what the World keeps, body_view.h)
    // `while_waiting` (may be null): queued-work hook of the update's one host round trip (Readback::wait) — called at most once;
    // the caller checks whether it ran
    // `carrier` (may be null): the kernel `while_waiting` would queue takes the round trip's post along (Readback::wait)
    int update_resident(const float4* d_aabb, int n, const StepPrologue* prologue = nullptr, const std::function<int()>* while_waiting = nullptr,
                        const MailCarrier* carrier = nullptr);
    // the C-ABI edge: 128-byte records (their AABBs are extracted into a scratch array first)
    int update_device(const phx_rigid_body* d_bodies, int n);
    int update_host(const phx_rigid_body* bodies, int n, uint32_t* new_pairs, int cap, int* count);
    int get_new_pairs(uint32_t* out, int cap, int* count);
    int get_sorted(phx_sort_entry* sorted, phx_broadphase_entry* entries, int cap);
    int erase_pairs(const uint32_t* pairs, int count);
    int erase_pairs_device(const uint2* d_pairs, int count);      // pairs already in HBM
    int reset_pairs(const uint2* pairs, int count);               // the pair set becomes exactly these (host) pairs: a world restored from a saved state
    const uint2* new_pairs_device() const { return new_pairs_.p; }   // pairs emitted by the last update, in HBM
    int get_stats(phx_broadphase_stats* out);
    int new_pair_count() const { return last_new_; }
    hipStream_t stream() const { return stream_; }

private:
    int resize_table(unsigned want_cap);
    int exclusive_scan(unsigned* data, int count, unsigned* total_out);
    int queue_erase_check(int* erased);
    int settle_erase_check(int erased);

    int device_;
    hipStream_t stream_ = nullptr;
    DevBuf<unsigned long long> stamps_;      // [0] clock at the update's first kernel, [1] at its last (device_ms without HIP events)
    DevBuf<unsigned> keys_[2], idx_[2], hist_, row_count_, row_cache_;
    DevBuf<float4> entries_;
    DevBuf<unsigned long long> table_, small_;
    DevBuf<int4> chunks_;
    DevBuf<unsigned> chunk_count_, chunk_scan_;
    ScanScratch scan_tiles_;
    DevBuf<uint2> new_pairs_, scratch_pairs_;
    DevBuf<phx_rigid_body> st_bodies_;
    DevBuf<float4> st_aabb_;
    // the two-level sort (splitter_sort.h): last update's splitters, bucket sizes + cursors, bucket bases, bucket per body, bucketed composites
    DevBuf<unsigned long long> splitters_, bucketed_;
    DevBuf<unsigned> ss_count_, ss_base_, ss_stats_;
    DevBuf<unsigned short> bucket_of_;
    int splitters_n_ = -1;                    // body count the splitters on record were taken for
    bool split_unbalanced_ = false, split_sorted_ = false;
    unsigned ss_last_max_ = 0u;          // the largest bucket of the last split-sorted update (0: none above twice the stride): picks k_bucket_sort's LDS shape
    DevBuf<int> erase_count_;                 // pairs really tombstoned since the last settle_erase_check()
    unsigned table_cap_ = 0;
    long long set_size_ = 0, tombstones_ = 0, erase_unchecked_ = 0;
    int n_ = 0, sorted_ = 0, last_new_ = 0;
    bool have_update_ = false, ms_pending_ = false;
    phx_broadphase_stats stats_{};
    Readback rb_;
public:
    void set_wait_timeout(double seconds) { rb_.set_timeout(seconds); }      // (a sharded World: the stream carries collectives, common.h Readback)
};

} // namespace phx
