// exchange.hip — DeviceSolver's side of the island-sharded exchange (layout + kernels: exchange.h).
#include "handles.h"
#include "exchange.h"

namespace phx {

int DeviceSolver::set_exchange_buffers(void* d_send, void* d_recv, size_t segment_capacity_bytes)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE((d_send && d_recv) || segment_capacity_bytes == 0, "null exchange buffers");
    PHX_REQUIRE((reinterpret_cast<uintptr_t>(d_send) & 15u) == 0 && (reinterpret_cast<uintptr_t>(d_recv) & 15u) == 0 && segment_capacity_bytes % 256 == 0,
                "exchange buffers must be 16-byte aligned, the segment capacity a multiple of 256 bytes");
    if (!xch_send_ && d_send) { PHX_TRY(synchronize()); sched_.valid = false; }      // the next schedule build also fetches the groups' body counts
    xch_send_ = static_cast<unsigned*>(d_send); xch_recv_ = static_cast<unsigned*>(d_recv); xch_cap_words_ = (long long)(segment_capacity_bytes / 4);
    if (!xch_err_.p) {
        PHX_TRY(xch_err_.reserve(1));
        PHX_HIP(hipMemsetAsync(xch_err_.p, 0, sizeof(int), stream_));
    }
    return PHX_OK;
}

// Which groups this rank solves (exchange_partition: longest processing time first by joint count).  A pure function of
// (schedule, shard count): recomputed when either changed.  An unsharded solver owns everything and needs no tables.
int DeviceSolver::ensure_partition()
{
    if (partition_version_ == schedule_version_ && partition_shards_ == shard_count_ && partition_shard_ == shard_) return PHX_OK;
    const bool live = sched_.valid && nj_ > 0;
    const int lg = live ? sched_.lds_groups : 0;
    const bool hbm = live && sched_.has_hbm_group();
    const int ng = lg + (hbm ? 1 : 0);
    owner_host_.assign((size_t)lg + 1, 0);             // entry lds_groups = the HBM group (present or not)
    mine_count_ = lg;
    if (shard_count_ > 1) {
        if ((int)sched_.group_offsets.size() < lg + 1) { set_error("island sharding: the schedule's group sizes are not on the host"); return PHX_ERR_STATE; }
        std::vector<int> gs(std::max(ng, 1));
        for (int g = 0; g < lg; ++g) gs[g] = sched_.group_offsets[g + 1] - sched_.group_offsets[g];
        if (hbm) gs[lg] = sched_.hbm_end() - sched_.hbm_begin();
        exchange_partition(gs.data(), ng, shard_count_, owner_host_.data());
        std::vector<int> mine;
        for (int g = 0; g < lg; ++g) if (owner_host_[g] == shard_) mine.push_back(g);
        mine_count_ = (int)mine.size();
        PHX_TRY(grp_owner_.reserve((size_t)lg + 1)); PHX_TRY(grp_mine_.reserve(std::max<size_t>(mine.size(), 1)));
        PHX_HIP(hipMemcpyAsync(grp_owner_.p, owner_host_.data(), ((size_t)lg + 1) * sizeof(int), hipMemcpyHostToDevice, stream_));
        if (!mine.empty()) PHX_HIP(hipMemcpyAsync(grp_mine_.p, mine.data(), mine.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
        PHX_HIP(hipStreamSynchronize(stream_));          // (the host vectors go out of scope)
    } else if (xch_send_) {                              // (one rank with an exchange: the kernels still index the tables)
        PHX_TRY(grp_owner_.reserve((size_t)lg + 1)); PHX_TRY(grp_mine_.reserve(std::max<size_t>((size_t)lg, 1)));
        std::vector<int> all((size_t)std::max(lg, 1));
        for (int g = 0; g < lg; ++g) all[g] = g;
        PHX_HIP(hipMemcpyAsync(grp_owner_.p, owner_host_.data(), ((size_t)lg + 1) * sizeof(int), hipMemcpyHostToDevice, stream_));
        if (lg) PHX_HIP(hipMemcpyAsync(grp_mine_.p, all.data(), (size_t)lg * sizeof(int), hipMemcpyHostToDevice, stream_));
        PHX_HIP(hipStreamSynchronize(stream_));
    }
    partition_version_ = schedule_version_; partition_shards_ = shard_count_; partition_shard_ = shard_;
    return PHX_OK;
}

// the layout is a pure function of (schedule, shard count): recomputed on the host when either changed, one small upload
int DeviceSolver::ensure_exchange_layout()
{
    if (xch_layout_version_ == schedule_version_ && xch_layout_shards_ == shard_count_) return PHX_OK;
    const bool live = sched_.valid && nj_ > 0;
    const int lg = live ? sched_.lds_groups : 0;
    const bool hbm = live && sched_.has_hbm_group();
    const int ng = lg + (hbm ? 1 : 0);
    if ((int)grp_body_count_.size() < lg) { set_error("exchange: the schedule carries no body counts"); return PHX_ERR_STATE; }
    PHX_TRY(ensure_partition());
    std::vector<int> gb(std::max(ng, 1)), gs(std::max(ng, 1));
    for (int g = 0; g < lg; ++g) { gb[g] = grp_body_count_[g]; gs[g] = sched_.group_offsets[g + 1] - sched_.group_offsets[g]; }
    if (hbm) { gb[lg] = sched_.hbm_body_count; gs[lg] = sched_.hbm_end() - sched_.hbm_begin(); }
    xch_off_host_.assign((size_t)lg + 1, 0ll);       // entry lds_groups = the HBM group (present or not)
    xch_seg_words_ = exchange_layout(gb.data(), gs.data(), ng, shard_count_, owner_host_.data(), xch_off_host_.data(), nullptr);
    PHX_TRY(xch_off_.reserve((size_t)lg + 1));
    PHX_HIP(hipMemcpyAsync(xch_off_.p, xch_off_host_.data(), ((size_t)lg + 1) * sizeof(long long), hipMemcpyHostToDevice, stream_));
    xch_layout_version_ = schedule_version_;
    xch_layout_shards_ = shard_count_;
    return PHX_OK;
}

static ExchangeView exchange_view(const Schedule& sc, bool valid, const int4* desc, const int* group_bodies, const int* order, const long long* xoff,
                                  const int* hbm_bodies, int shard, int shard_count, long long seg_words, const int* owner, const int* mine)
{
    ExchangeView x{};
    x.desc = desc; x.group_bodies = group_bodies; x.order = order; x.xoff = xoff; x.owner = owner; x.mine = mine;
    x.lds_groups = valid ? sc.lds_groups : 0; x.shard = shard; x.shard_count = shard_count;
    x.hbm_bodies = hbm_bodies;
    const bool hbm = valid && sc.has_hbm_group();
    x.hbm_body_count = hbm ? sc.hbm_body_count : 0; x.hbm_begin = hbm ? sc.hbm_begin() : 0; x.hbm_end = hbm ? sc.hbm_end() : 0;
    x.segment_words = seg_words;
    return x;
}

// the C-ABI edge: records in, records out (converted around the resident kernels, body_view.h)
int DeviceSolver::exchange_pack(const void* d_bodies, const void* d_joints, size_t* segment_bytes, int status_word)
{
    PHX_TRY(use_device(device_));
    if (!d_bodies || !d_joints) return exchange_pack_resident(nullptr, nullptr, segment_bytes, status_word);
    PHX_TRY(synchronize());                            // (the edge arrays may still belong to an unverified solve)
    Arrays a;
    PHX_TRY(edge_view(d_bodies, nb_, &a));
    return exchange_pack_resident(&a.view, d_joints, segment_bytes, status_word);
}

int DeviceSolver::exchange_unpack(void* d_bodies, void* d_joints)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(d_bodies && d_joints, "null arrays");
    PHX_TRY(synchronize());
    Arrays a;
    PHX_TRY(edge_view(d_bodies, nb_, &a));
    PHX_TRY(exchange_unpack_resident(a.view, d_joints));
    if (nb_) hipLaunchKernelGGL(k_view_to_bodies, dim3(std::max(1, std::min(div_up(nb_, 256), 2048))), dim3(256), 0, stream_, a.view, nb_, a.aos, (const unsigned long long*)nullptr, 0ull);
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

int DeviceSolver::exchange_pack_resident(const BodyView* d_bodies, const void* d_joints, size_t* segment_bytes, int status_word)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(xch_send_ && xch_recv_, "exchange buffers not set (phx_solver_set_exchange_buffers)");
    PHX_TRY(ensure_exchange_layout());
    if (xch_seg_words_ > xch_cap_words_) {
        set_error("exchange: segment of %lld bytes exceeds the buffer capacity of %lld bytes", 4 * xch_seg_words_, 4 * xch_cap_words_);
        return PHX_ERR_CAPACITY;
    }
    // null arrays = header only: a rank that failed earlier in the step still posts its status word
    const ExchangeView x = exchange_view(sched_, sched_.valid && nj_ > 0 && d_bodies && d_joints, isl_.desc.p, isl_.bodies.p, hbm_.order.p, xch_off_.p, hbm_.hbm_body_list.p, shard_,
                                         shard_count_, xch_seg_words_, grp_owner_.p, grp_mine_.p);
    ++xch_serial_;
    const int mine = x.lds_groups ? mine_count_ : 0;
    const BodyView bodies = d_bodies ? *d_bodies : BodyView{nullptr, nullptr, nullptr};
    hipLaunchKernelGGL(k_exchange_pack, dim3(std::max(mine, 1)), dim3(256), 0, stream_, x, bodies,
                       static_cast<const phx_contact_joint*>(d_joints), xch_send_, xch_serial_, (unsigned)status_word, raw_fingerprint_, mine);
    if (x.hbm_end > x.hbm_begin && owns_hbm_group()) {
        const int n = std::max(x.hbm_body_count, x.hbm_end - x.hbm_begin);
        hipLaunchKernelGGL(k_exchange_pack_hbm, dim3(std::max(1, std::min(div_up(n, 256), 2048))), dim3(256), 0, stream_, x, bodies,
                           static_cast<const phx_contact_joint*>(d_joints), xch_send_);
    }
    PHX_HIP(hipGetLastError());
    if (segment_bytes) *segment_bytes = (size_t)xch_seg_words_ * 4;
    return PHX_OK;
}

int DeviceSolver::exchange_unpack_resident(const BodyView& d_bodies, void* d_joints)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(xch_send_ && xch_recv_, "exchange buffers not set (phx_solver_set_exchange_buffers)");
    if (xch_layout_version_ != schedule_version_ || xch_layout_shards_ != shard_count_) { set_error("exchange_unpack without a matching exchange_pack"); return PHX_ERR_STATE; }
    const ExchangeView x = exchange_view(sched_, sched_.valid && nj_ > 0, isl_.desc.p, isl_.bodies.p, hbm_.order.p, xch_off_.p, hbm_.hbm_body_list.p, shard_, shard_count_, xch_seg_words_,
                                         grp_owner_.p, grp_mine_.p);
    hipLaunchKernelGGL(k_exchange_unpack, dim3(std::max(x.lds_groups, 1)), dim3(256), 0, stream_, x, d_bodies,
                       static_cast<phx_contact_joint*>(d_joints), (const unsigned*)xch_recv_, xch_serial_, raw_fingerprint_, xch_err_.p);
    if (x.hbm_end > x.hbm_begin && !owns_hbm_group()) {
        const int n = std::max(x.hbm_body_count, x.hbm_end - x.hbm_begin);
        hipLaunchKernelGGL(k_exchange_unpack_hbm, dim3(std::max(1, std::min(div_up(n, 256), 2048))), dim3(256), 0, stream_, x, d_bodies,
                           static_cast<phx_contact_joint*>(d_joints), (const unsigned*)xch_recv_, xch_serial_, raw_fingerprint_);
    }
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

int DeviceSolver::exchange_all_gather()
{
    PHX_REQUIRE(comm_ && xch_send_ && xch_recv_, "no communicator / exchange buffers");
    return comm_->all_gather(xch_send_, xch_recv_, (size_t)xch_seg_words_ * 4, stream_);
}

int DeviceSolver::exchange_status(int* out)
{
    PHX_REQUIRE(out, "null out");
    *out = 0;
    if (!xch_err_.p) return PHX_OK;
    PHX_TRY(use_device(device_));
    PHX_TRY(rb_.add(out, xch_err_.p, sizeof(int), stream_));
    return rb_.wait(stream_);
}

} // namespace phx

extern "C" {

int phx_exchange_layout(const int32_t* group_bodies, const int32_t* group_slots, int32_t group_count, int32_t shard_count,
                        int32_t* group_owner, int64_t* group_offset_words, int64_t* rank_words, int64_t* segment_words)
{
    PHX_REQUIRE(group_count >= 0 && shard_count >= 1 && (group_count == 0 || (group_bodies && group_slots)) && segment_words, "bad arguments");
    for (int g = 0; g < group_count; ++g) PHX_REQUIRE(group_bodies[g] >= 0 && group_slots[g] >= 0, "negative count");
    static_assert(sizeof(long long) == sizeof(int64_t), "layout words");
    std::vector<int> owner((size_t)std::max(group_count, 1), 0);
    phx::exchange_partition(group_slots, group_count, shard_count, owner.data());
    if (group_owner) std::copy(owner.begin(), owner.begin() + group_count, group_owner);
    *segment_words = phx::exchange_layout(group_bodies, group_slots, group_count, shard_count, owner.data(), reinterpret_cast<long long*>(group_offset_words),
                                          reinterpret_cast<long long*>(rank_words));
    return PHX_OK;
}

int phx_solver_set_exchange_buffers(phx_solver* s, void* d_send, void* d_recv, size_t segment_capacity_bytes)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.set_exchange_buffers(d_send, d_recv, segment_capacity_bytes);
}

int phx_solver_set_comm(phx_solver* s, phx_comm* c)
{
    PHX_REQUIRE(s, "null handle");
    s->impl.set_comm(c ? &c->impl : nullptr);
    return PHX_OK;
}

int phx_solver_exchange_pack(phx_solver* s, const void* d_bodies, const void* d_joints, int32_t status_word, size_t* segment_bytes)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.exchange_pack(d_bodies, d_joints, segment_bytes, status_word);
}

int phx_solver_exchange_unpack(phx_solver* s, void* d_bodies, void* d_joints)
{
    PHX_REQUIRE(s, "null handle");
    return s->impl.exchange_unpack(d_bodies, d_joints);
}

int phx_solver_exchange_status(phx_solver* s, int32_t* status)
{
    PHX_REQUIRE(s, "null handle");
    int v = 0;
    PHX_TRY(s->impl.exchange_status(&v));
    if (status) *status = v;
    return PHX_OK;
}

size_t phx_solver_exchange_segment_bytes(phx_solver* s) { return s ? s->impl.exchange_segment_bytes() : 0; }

} // extern "C"
