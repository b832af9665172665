// solver.hip — DeviceSolver: schedule construction on the host, residency and launch sequence.
#include "solver.h"
#include "solver_kernels.h"
#include "schedule_kernels.h"
#include "device_radix.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <numeric>

namespace phx {

// ---------------------------------------------------------------------------------------------------
// small kernels private to this file

__global__ void __launch_bounds__(256) k_extract_topology(const phx_contact_joint* __restrict__ joints, int nj,
                                                          const float4* __restrict__ mpos, int nb,
                                                          int2* __restrict__ pairs, int* __restrict__ prio_id, unsigned char* __restrict__ is_static)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nj; i += gridDim.x * blockDim.x) {
        pairs[i] = make_int2(joints[i].body1, joints[i].body2);
        prio_id[i] = joints[i].contact_point_index;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x)
        is_static[i] = (mpos[i].x == 0.f && mpos[i].y == 0.f) ? 1 : 0;
}

static inline int grid_for(int n) { return std::max(1, std::min(div_up(n, 256), 2048)); }

constexpr int STATS_SET = 2 * ISL_STAT_SLOTS, VISITS_SET = ISL_STAT_SLOTS + 2, SHARDS_SET = ISL_SHARDS * ISL_SHARD_STRIDE;      // words of one control set in isl_.stats / isl_.visits / isl_.shards

// ---------------------------------------------------------------------------------------------------

DeviceSolver::~DeviceSolver()
{
    if (hipSetDevice(device_) != hipSuccess) return;
    if (stream_) (void)hipStreamSynchronize(stream_);
    drop_graphs();
    for (hipEvent_t e : bench_events_) (void)hipEventDestroy(e);
    // (every device buffer is a DevBuf member: freed with the object, after this body — the device is selected above)
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    if (side_stream_) { (void)hipStreamSynchronize(side_stream_); (void)hipStreamDestroy(side_stream_); }
    if (ev_begin_) (void)hipEventDestroy(ev_begin_);
    if (ev_end_) (void)hipEventDestroy(ev_end_);
    if (ev_sweep_begin_) (void)hipEventDestroy(ev_sweep_begin_);
    if (ev_sweep_end_) (void)hipEventDestroy(ev_sweep_end_);
    if (stream_ && owns_stream_) (void)hipStreamDestroy(stream_);
}

int DeviceSolver::adopt_stream(hipStream_t s)
{
    PHX_TRY(use_device(device_));
    PHX_TRY(synchronize());
    drop_graphs();
    if (stream_ && owns_stream_) PHX_HIP(hipStreamDestroy(stream_));
    stream_ = s;
    owns_stream_ = false;
    return PHX_OK;
}

// the PHX_* knobs, read once per handle (measurement and debugging only: README.md)
DeviceSolver::Options DeviceSolver::Options::from_env()
{
    auto on = [](const char* name) { const char* v = getenv(name); return v && v[0] == '1'; };
    Options o;
    o.no_side_stream = on("PHX_NO_SIDE_STREAM");          // everything on the one stream (A/B measurements)
    o.no_parts = on("PHX_NO_PARTS");                      // the interior classes of partitioned components one launch each (A/B, tests)
    o.no_fused_verify = on("PHX_NO_FUSED_VERIFY");        // the topology hash pass in front of every solve on a cached schedule
    const char* wp = getenv("PHX_ISL_WAIT_POLLS");        // tests: 0 makes every workgroup of a verified launch give up, so that ISL_COMPLETE runs
    o.isl_wait_polls = wp ? std::max(0, atoi(wp)) : ISL_WAIT_POLLS;
    o.use_graphs = on("PHX_GRAPHS");                      // replay the launch sequence from hipGraphs (measured slower: off by default)
    const char* sb = getenv("PHX_SCHEDULE_BUILDER");      // "host" forces the host builder
    o.gpu_builder = !(sb && sb[0] == 'h');
    o.speculate = !on("PHX_NO_SPECULATION");
    o.no_islands = on("PHX_NO_ISLANDS");                  // ignore island modes, always the HBM colour path
    o.no_spec_bins = on("PHX_NO_SPEC_BINS") || o.use_graphs;      // every rebuild reads the component sizes back and bins them on the host
    o.trace_schedule = getenv("PHX_TRACE_SCHEDULE") != nullptr;  // print the schedule builders' laps to stderr
    return o;
}

int DeviceSolver::init()
{
    PHX_TRY(use_device(device_));
    PHX_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    PHX_HIP(hipEventCreate(&ev_begin_));
    PHX_HIP(hipEventCreate(&ev_end_));
    PHX_HIP(hipEventCreate(&ev_sweep_begin_));
    PHX_HIP(hipEventCreate(&ev_sweep_end_));
    // the LDS islands of a schedule that also has an HBM group run beside its sweeps (enqueue_sweeps)
    PHX_HIP(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking));
    PHX_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    PHX_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    // two control sets alternate between consecutive solves; the first kernel of a solve clears the other one (solver_kernels.h)
    PHX_TRY(hash_.reserve(2));
    PHX_HIP(hipMemsetAsync(hash_.p, 0, 2 * sizeof(unsigned long long), stream_));
    PHX_TRY(isl_.stats.reserve(2 * STATS_SET));
    PHX_HIP(hipMemsetAsync(isl_.stats.p, 0, 2 * STATS_SET * sizeof(int), stream_));
    PHX_TRY(isl_.visits.reserve(2 * VISITS_SET));               // per set: the slots + the solve's two time stamps
    {
        unsigned long long init[2 * VISITS_SET] = {0};
        init[ISL_STAT_SLOTS] = ~0ull; init[VISITS_SET + ISL_STAT_SLOTS] = ~0ull;
        PHX_HIP(hipMemcpyAsync(isl_.visits.p, init, sizeof init, hipMemcpyHostToDevice, stream_));
        PHX_HIP(hipStreamSynchronize(stream_));
    }
    PHX_TRY(isl_.shards.reserve(2 * SHARDS_SET));
    PHX_HIP(hipMemsetAsync(isl_.shards.p, 0, 2 * SHARDS_SET * sizeof(unsigned long long), stream_));
    PHX_HIP(hipDeviceGetAttribute(&cu_count_, hipDeviceAttributeMultiprocessorCount, device_));
    opt_ = Options::from_env();
    defer_build_check_ = opt_.speculate;                     // (PHX_NO_SPECULATION=1 also waits for the device build's 'every bin fits' flag)
    return PHX_OK;
}

SolverView DeviceSolver::view() const
{
    SolverView v{};
    v.nb = nb_; v.nj = nj_; v.ncp = ncp_; v.nstatic = std::max(nstatic_, 1); v.ncolours = sched_.ncolours();
    v.fingerprint = hash_.p + hash_slot_; v.expected_fingerprint = gate_expected_;
    v.sb_imp = hbm_.sb_imp.p; v.sb_disp = hbm_.sb_disp.p; v.sb_par = cur_.view.mpos;
    v.q0 = hbm_.q0.p; v.q1 = hbm_.q1.p; v.q2 = hbm_.q2.p; v.q3 = hbm_.q3.p; v.acc = hbm_.acc.p; v.dd = hbm_.dd.p; v.qn = hbm_.qn.p;
    v.order = hbm_.order.p;
    v.sw_imp = hbm_.sw.p; v.sw_disp = hbm_.sw.p + 2 * (size_t)v.nstatic;
    v.imp_active = hbm_.flags.p; v.disp_active = hbm_.flags.p + max_iters_;
    v.stamps = isl_.visits.p + (size_t)hash_slot_ * VISITS_SET + ISL_STAT_SLOTS;
    return v;
}

// The solve being queued takes the other control set (control word, island counters, time stamps).  Its first kernel clears
// the set after it: the hash pass if one runs, else the island launch (enqueue_sweeps).
void DeviceSolver::begin_set(bool hash_runs)
{
    hash_slot_ ^= 1;
    island_clears_next_ = !hash_runs;
}

int DeviceSolver::launch_fingerprint(const float4* d_mpos, int nb, const phx_contact_joint* d_joints, int nj, int ncp)
{
    // the hash pass is the first kernel of the solves that run it: it also clears the solve's HBM-path words and the control set of
    // the next solve (the two sets alternate)
    begin_set(true);
    ControlWords cw{};
    cw.flags = hbm_.flags.p; cw.nflags = hbm_.flags.p ? 2 * max_iters_ : 0;
    cw.sw = hbm_.sw.p; cw.nsw = hbm_.sw.p ? (int)std::min<size_t>(hbm_.sw.cap, 1u << 30) : 0;      // (the whole table: a rebuilt schedule may use more of it)
    sw_cleared_ = hbm_.sw.p; sw_cleared_words_ = (size_t)cw.nsw;
    int next = hash_slot_ ^ 1;
    cw.next_ctl = hash_.p + next;
    if (opt_.use_graphs) {        // captured graphs have the control set's addresses baked in: always set 0 — its word cleared by a memset, its
        hash_slot_ = 0;       // counters by this kernel (nothing else touches them while it runs)
        PHX_HIP(hipMemsetAsync(hash_.p, 0, sizeof(unsigned long long), stream_));
        next = 0; cw.next_ctl = hash_.p + 1;
    }
    cw.next_executed = isl_.stats.p + (size_t)next * STATS_SET; cw.next_visits = isl_.visits.p + (size_t)next * VISITS_SET;
    cw.next_shards = isl_.shards.p + (size_t)next * SHARDS_SET;
    hipLaunchKernelGGL(k_topology_hash, dim3(std::max(1, std::min(div_up(std::max(nj, nb), HASH_T), HASH_BLOCKS))), dim3(HASH_T), 0, stream_, d_joints, nj, d_mpos, nb, ncp,
                       hash_.p + hash_slot_, cw);
    isl_mode_ = ISL_GATED;
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

// May a launch of `groups` island workgroups check the cached schedule itself (ISL_VERIFY, island_view.h)?  Only if all of them
// are resident at once — they wait for each other before they commit — and nobody else needs the topology hash (a sharded solve
// puts it into its exchange header; graphs bake the launch arguments in).
bool DeviceSolver::verify_eligible(int groups, bool big_shape) const
{
    if (opt_.no_fused_verify || !opt_.speculate || opt_.use_graphs || shard_count_ != 1 || xch_send_ || groups <= 0) return false;
    // (LDS — 37 / 50 KB — and the 128-register budget admit 4 / 2 workgroups per CU; asked of the runtime for the instantiation
    //  at hand rather than assumed, and never more than that: the occupancy query has been seen one block high)
    const int per_cu = std::min(island_blocks_per_cu(big_shape, half_state_), big_shape ? 2 : 4);
    return per_cu > 0 && groups <= per_cu * cu_count_ && groups < (int)ISL_BAD / 2;
}

// would a rebuild take the path without a host round trip (build_bins_speculative)?
bool DeviceSolver::spec_build_applies(bool want_islands, int nj) const
{
    return spec_bins_ok_ && !opt_.no_spec_bins && want_islands && defer_build_check_ && shard_count_ == 1 && !xch_send_ && nj > 0 && nj < (1 << BINC_JOINT_BITS) &&
           !opt_.trace_schedule;
}

// Arms the gate of a solve on the cached schedule.  ISL_VERIFY where the island launch can check the schedule itself — no kernel in
// front of it at all; else the hash pass, if the schedule's hash is on record.
bool DeviceSolver::arm_cached_solve(const float4* d_mpos, int nb, const phx_contact_joint* d_joints, int nj, int ncp, int* status)
{
    *status = PHX_OK;
    const int mine = sched_.lds_groups;                  // (ISL_VERIFY is for unsharded solves: verify_eligible)
    if (nj > 0 && !sched_.has_hbm_group() && verify_eligible(mine, sched_.lds_lanes > ISL_T)) {
        begin_set(false);
        isl_mode_ = ISL_VERIFY;
        isl_nexpect_ = (unsigned)mine;
        gate_expected_ = (unsigned long long)std::min(mine, ISL_SHARDS);      // every shard of workgroups arrived, none saw a difference, nobody gave up
        if (++solve_epoch_ == 0) solve_epoch_ = 1;
        return true;
    }
    if (!have_hash_) return false;
    *status = launch_fingerprint(d_mpos, nb, d_joints, nj, ncp);
    gate_expected_ = raw_fingerprint_;
    return true;
}

int DeviceSolver::ensure_schedule(const float4* d_bodies, int nb, const phx_contact_joint* d_joints, int nj, int ncp, const phx_config& cfg, bool force_rebuild,
                                  bool known_changed)
{
    // 1. fingerprint of the joint topology (8 bytes over PCIe).  A caller that KNOWS the topology changed (the World, when
    //    joints were created or destroyed this step) does not wait for it: the value rides along with the builder's first
    //    readback — and a rebuild that needs no host round trip (build_bins_speculative) whose solves the island kernel can
    //    check itself (ISL_VERIFY) skips the hash pass altogether: nobody would ever compare it with anything.
    unsigned long long fp = 0;
    bool have_fp = false;
    // Single = one coupled system swept class by class out of HBM; every other island mode lets the schedule
    // exploit body-disjoint islands (groups solved out of LDS)
    const bool want_islands = cfg.island_mode != PHX_ISLAND_SINGLE && !opt_.no_islands;
    const bool device_builder = opt_.gpu_builder && !force_host_builder_;
    const bool no_hash = known_changed && device_builder && spec_build_applies(want_islands, nj) && verify_eligible(spec_bins_guess_, spec_lanes_ > ISL_T);
    if (no_hash) begin_set(false);
    else PHX_TRY(launch_fingerprint(d_bodies, nb, d_joints, nj, ncp));
    isl_mode_ = ISL_GATED;
    have_hash_ = false;
    stats_.recoloured = 0;
    ncp_ = ncp;
    force_host_builder_ = false;
    build_unverified_ = false;
    if (!(known_changed && device_builder)) {
        PHX_TRY(rb_.add(&fp, hash_.p + hash_slot_, sizeof fp, stream_));
        PHX_TRY(rb_.wait(stream_));
        have_fp = true;
        const unsigned long long mixed = fp ^ ((unsigned long long)(unsigned)nj << 32) ^ (unsigned)nb;
        if (!force_rebuild && !known_changed && sched_.valid && sched_.fingerprint == mixed && nb == nb_ && nj == nj_ && sched_.islands == want_islands) {
            raw_fingerprint_ = fp; gate_expected_ = fp; have_hash_ = true;
            return PHX_OK;
        }
    }

    // 2. topology changed.  Schedules are built on the device (only component sizes cross PCIe); the host builder below is
    //    the specification and the fallback (bins that exceed the caps, more than 64 colours, ...).
    if (device_builder) {
        bool fallback = false;
        nb_ = nb; nj_ = nj;
        fp_wanted_ = have_fp ? nullptr : &fp;
        const int st = build_schedule_device(d_bodies, nb, d_joints, nj, want_islands, &fallback);
        if (st != PHX_OK) { fp_wanted_ = nullptr; return st; }
        if (spec_bins_pending_) fp_wanted_ = nullptr;   // (the hash comes back with everything else when the solve is settled: collect_stats)
        if (fp_wanted_) {                              // the builder had nothing to read back (no joints) or bailed out early
            fp_wanted_ = nullptr;
            PHX_TRY(rb_.add(&fp, hash_.p + hash_slot_, sizeof fp, stream_));
            PHX_TRY(rb_.wait(stream_));
        }
        have_fp = true;
        if (!fallback) {
            sched_.fingerprint = fp ^ ((unsigned long long)(unsigned)nj << 32) ^ (unsigned)nb;
            raw_fingerprint_ = fp;
            have_hash_ = !no_hash && !spec_bins_pending_;      // (a speculative build's hash comes back when the solve is settled)
            spec_hash_ran_ = !no_hash;
            if (!spec_bins_pending_) gate_expected_ = fp;
            sched_.valid = true;
            ++schedule_version_;
            drop_graphs();
            stats_.recoloured = spec_bins_pending_ ? 2 : 1;
            return PHX_OK;
        }
        sched_.valid = false;
        build_unverified_ = false;                     // (the host builder's schedules need no verification)
    }
    const unsigned long long raw = fp;
    fp ^= ((unsigned long long)(unsigned)nj << 32) ^ (unsigned)nb;
    const bool trace = opt_.trace_schedule;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!trace) return; auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[schedule] %-18s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t0).count()); t0 = n; };
    DevBuf<int2> d_pairs;
    DevBuf<int> d_prio;
    DevBuf<unsigned char> d_static;
    PHX_TRY(d_pairs.reserve(std::max(nj, 1)));
    PHX_TRY(d_prio.reserve(std::max(nj, 1)));
    PHX_TRY(d_static.reserve(std::max(nb, 1)));
    hipLaunchKernelGGL(k_extract_topology, dim3(grid_for(std::max(nj, nb))), dim3(256), 0, stream_, d_joints, nj, d_bodies, nb, d_pairs.p, d_prio.p, d_static.p);
    PHX_HIP(hipGetLastError());
    std::vector<int2> pairs(std::max(nj, 1));
    std::vector<int> prio_id(std::max(nj, 1));
    std::vector<unsigned char> is_static(std::max(nb, 1));
    PHX_HIP(hipMemcpyAsync(pairs.data(), d_pairs.p, (size_t)nj * sizeof(int2), hipMemcpyDeviceToHost, stream_));
    PHX_HIP(hipMemcpyAsync(prio_id.data(), d_prio.p, (size_t)nj * sizeof(int), hipMemcpyDeviceToHost, stream_));
    PHX_HIP(hipMemcpyAsync(is_static.data(), d_static.p, (size_t)nb, hipMemcpyDeviceToHost, stream_));
    PHX_HIP(hipStreamSynchronize(stream_));
    d_pairs.release();
    d_prio.release();
    d_static.release();
    lap("download");

    std::vector<int> b1(nj), b2(nj);
    for (int j = 0; j < nj; ++j) {
        b1[j] = pairs[j].x; b2[j] = pairs[j].y;
        if ((unsigned)b1[j] >= (unsigned)nb || (unsigned)b2[j] >= (unsigned)nb) { sched_.valid = false; set_error("joint %d references body out of range", j); return PHX_ERR_INVALID; }
    }
    if (want_islands) {
        LdsCaps caps;
        caps.max_units = ISL_T; caps.max_joints = 2 * ISL_T; caps.max_bodies = ISL_B; caps.max_colours = 64;
        LdsCaps big;
        big.max_units = ISL_T_BIG; big.max_joints = 2 * ISL_T_BIG; big.max_bodies = ISL_B_BIG; big.max_colours = 64;
        build_island_schedule(b1.data(), b2.data(), nj, is_static.data(), nb, caps, sched_, &big, prio_id.data());
    } else {
        build_colour_schedule(b1.data(), b2.data(), nj, is_static.data(), nb, sched_, prio_id.data());
    }
    lap("build");
    if (sched_.ncolours() > 65000) { set_error("more than 65000 colours"); return PHX_ERR_INVALID; }
    if (!want_islands) {       // the island-aware builder publishes GatherIslands' numbers itself
        std::vector<int> joint_island, island_size;
        gather_islands(b1.data(), b2.data(), nj, is_static.data(), nb, joint_island, island_size);
        sched_.island_count = (int)island_size.size();
        sched_.island_max_size = island_size.empty() ? 0 : *std::max_element(island_size.begin(), island_size.end());
    }
    lap("gather_islands");
    h_static_slot_.assign(nb, -1);
    nstatic_ = 0;
    for (int i = 0; i < nb; ++i) if (is_static[i]) h_static_slot_[i] = nstatic_++;

    nb_ = nb; nj_ = nj;
    PHX_TRY(hbm_.order.reserve(std::max(nj, 1)));
    PHX_TRY(hbm_.static_slot.reserve(std::max(nb, 1)));
    PHX_TRY(hbm_.sw.reserve(4 * (size_t)std::max(nstatic_, 1)));
    // new table for this solve — already cleared by this solve's fingerprint kernel unless it has just been (re)allocated
    if (hbm_.sw.p != sw_cleared_ || 4 * (size_t)std::max(nstatic_, 1) > sw_cleared_words_)
        PHX_HIP(hipMemsetAsync(hbm_.sw.p, 0, 4 * (size_t)std::max(nstatic_, 1) * sizeof(unsigned), stream_));
    PHX_TRY(hbm_.sb_imp.reserve(nb)); PHX_TRY(hbm_.sb_disp.reserve(nb));
    PHX_TRY(hbm_.q0.reserve(nj)); PHX_TRY(hbm_.q1.reserve(nj)); PHX_TRY(hbm_.q2.reserve(nj)); PHX_TRY(hbm_.q3.reserve(nj)); PHX_TRY(hbm_.qn.reserve(nj));
    PHX_TRY(hbm_.acc.reserve(nj)); PHX_TRY(hbm_.dd.reserve(nj));
    if (nj) PHX_HIP(hipMemcpyAsync(hbm_.order.p, sched_.order.data(), (size_t)nj * sizeof(int), hipMemcpyHostToDevice, stream_));
    if (nb) PHX_HIP(hipMemcpyAsync(hbm_.static_slot.p, h_static_slot_.data(), (size_t)nb * sizeof(int), hipMemcpyHostToDevice, stream_));
    const int ng = sched_.lds_groups;
    std::vector<int4> desc(std::max(ng, 1));
    std::vector<int> ncol(std::max(ng, 1));
    grp_body_count_.assign(ng, 0);
    for (int g = 0; g < ng; ++g) grp_body_count_[g] = sched_.group_body_offsets[g + 1] - sched_.group_body_offsets[g];
    if (ng) {
        // the layout the device builder leaves (k_build_bin): group g's body table at g * (body capacity of the shape), its units at
        // g * lanes — the island kernel addresses both by the group number alone
        const int lanes = sched_.lds_lanes, cap_bodies = lanes > ISL_T ? ISL_B_BIG : ISL_B;
        std::vector<int> bodies_strided((size_t)ng * cap_bodies, 0);
        for (int g = 0; g < ng; ++g) {
            desc[g] = make_int4(sched_.group_offsets[g], sched_.group_offsets[g + 1] - sched_.group_offsets[g], g * cap_bodies, grp_body_count_[g]);
            ncol[g] = sched_.group_first_colour[g + 1] - sched_.group_first_colour[g];
            std::copy(sched_.group_bodies.begin() + sched_.group_body_offsets[g], sched_.group_bodies.begin() + sched_.group_body_offsets[g + 1], bodies_strided.begin() + (size_t)g * cap_bodies);
        }
        const size_t lds_slots = (size_t)sched_.group_offsets[ng];
        PHX_TRY(isl_.desc.reserve(ng)); PHX_TRY(isl_.ncol.reserve(ng));
        PHX_TRY(isl_.bodies.reserve(bodies_strided.size())); PHX_TRY(isl_.slot_local.reserve(lds_slots)); PHX_TRY(isl_.slot_colour.reserve(lds_slots));
        PHX_HIP(hipMemcpyAsync(isl_.desc.p, desc.data(), (size_t)ng * sizeof(int4), hipMemcpyHostToDevice, stream_));
        PHX_HIP(hipMemcpyAsync(isl_.ncol.p, ncol.data(), (size_t)ng * sizeof(int), hipMemcpyHostToDevice, stream_));
        PHX_HIP(hipMemcpyAsync(isl_.bodies.p, bodies_strided.data(), bodies_strided.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
        PHX_HIP(hipMemcpyAsync(isl_.slot_local.p, sched_.slot_local.data(), lds_slots * sizeof(unsigned), hipMemcpyHostToDevice, stream_));
        PHX_HIP(hipMemcpyAsync(isl_.slot_colour.p, sched_.slot_colour.data(), lds_slots, hipMemcpyHostToDevice, stream_));
        // the units, class-major, at a fixed stride of one workgroup's lanes per group
        std::vector<int> units(ng);
        std::vector<int4> unit_recs(2 * (size_t)ng * lanes, make_int4(0, -1, 0, 0));
        for (int g = 0; g < ng; ++g) {
            const int nunits = sched_.group_unit_offsets[g + 1] - sched_.group_unit_offsets[g];
            int nstatic_g = 0;                                   // (the group's static bodies sit first in its table)
            for (int k = sched_.group_body_offsets[g]; k < sched_.group_body_offsets[g + 1] && is_static[sched_.group_bodies[k]]; ++k) ++nstatic_g;
            units[g] = nunits | (nstatic_g << 16);
            for (int u = 0; u < nunits; ++u) {
                const int at = sched_.group_unit_offsets[g] + u;
                const int ls = sched_.unit_leader[at], fs = sched_.unit_follower[at];
                const int lj = sched_.order[ls], fj = fs >= 0 ? sched_.order[fs] : -1;
                unit_recs[2 * ((size_t)g * lanes + u)] = make_int4(lj, fj, prio_id[lj], fj >= 0 ? prio_id[fj] : 0);
                unit_recs[2 * ((size_t)g * lanes + u) + 1] = make_int4((int)sched_.slot_local[ls], (int)sched_.slot_colour[ls], ls, fs);
            }
        }
        PHX_TRY(isl_.units.reserve(ng)); PHX_TRY(isl_.unit_recs.reserve(unit_recs.size()));
        PHX_HIP(hipMemcpyAsync(isl_.units.p, units.data(), (size_t)ng * sizeof(int), hipMemcpyHostToDevice, stream_));
        PHX_HIP(hipMemcpyAsync(isl_.unit_recs.p, unit_recs.data(), unit_recs.size() * sizeof(int4), hipMemcpyHostToDevice, stream_));
    }
    PHX_TRY(hbm_.hbm_body_list.reserve(std::max<size_t>(sched_.hbm_bodies.size(), 1)));
    if (!sched_.hbm_bodies.empty())
        PHX_HIP(hipMemcpyAsync(hbm_.hbm_body_list.p, sched_.hbm_bodies.data(), sched_.hbm_bodies.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    PHX_TRY(upload_part_tables());
    PHX_HIP(hipStreamSynchronize(stream_));
    lap("upload");
    sched_.fingerprint = fp;
    raw_fingerprint_ = raw; gate_expected_ = raw; have_hash_ = true;
    sched_.valid = true;
    ++schedule_version_;
    drop_graphs();
    stats_.recoloured = 1;
    return PHX_OK;
}

// ---------------------------------------------------------------------------------------------------
// device schedule builder (kernels: schedule_kernels.h)

constexpr int JP_BATCH = 8;          // colouring rounds queued between two looks at the frontier sizes
constexpr int JP_ROUNDS_MAX = 512;

int DeviceSolver::build_schedule_device(const float4* d_bodies, int nb, const phx_contact_joint* d_joints, int nj, bool want_islands, bool* fallback)
{
    *fallback = false;
    RoctxRange range("GatherIslands + PrepareIndices (schedule build)");          // ref: Solver.cpp:77, 135, 217, 285
    // the topology fingerprint (already queued on the stream) rides along with the first readback of the build
    auto with_fingerprint = [&]() -> int { if (fp_wanted_) { PHX_TRY(rb_.add(fp_wanted_, hash_.p + hash_slot_, sizeof *fp_wanted_, stream_)); fp_wanted_ = nullptr; } return PHX_OK; };
    const bool trace = opt_.trace_schedule;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!trace) return; (void)hipStreamSynchronize(stream_); auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[schedule/gpu] %-18s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t0).count()); t0 = n; };
    const int nbs = std::max(nb, 1), njs = std::max(nj, 1);
    PHX_TRY(bld_.cc_parent.reserve(nbs)); PHX_TRY(bld_.cc_static.reserve(nbs)); PHX_TRY(bld_.cc_flags.reserve(nbs + 1)); PHX_TRY(bld_.comp_size.reserve(nbs + 1));
    PHX_TRY(bld_.joint_comp.reserve(njs)); PHX_TRY(bld_.sb_small.reserve(8));
    for (int k = 0; k < 2; ++k) { PHX_TRY(bld_.sort_keys[k].reserve(njs)); PHX_TRY(bld_.sort_vals[k].reserve(njs)); }
    PHX_TRY(bld_.sort_hist.reserve(radix_hist_words(nj)));
    PHX_TRY(hbm_.order.reserve(njs));

    // units (schedule.h): contact point -> first joint carrying it (reset by k_cc_init); the partners are found by the first hook
    PHX_TRY(bld_.partner.reserve(njs)); PHX_TRY(bld_.comp_units.reserve(nbs + 1));
    {
        const size_t had = bld_.partner_first.cap;
        PHX_TRY(bld_.partner_first.reserve(std::max(ncp_, 1)));
        if (bld_.partner_first.cap != had || bld_.partner_tag <= 1) {          // a new table, or the tags ran out: every entry reads 'nobody' again
            PHX_HIP(hipMemsetAsync(bld_.partner_first.p, 0xFF, bld_.partner_first.cap * sizeof(unsigned long long), stream_));
            bld_.partner_tag = 0xFFFFFFFEu;
        } else --bld_.partner_tag;
    }
    hipLaunchKernelGGL(k_cc_init, dim3(grid_for(std::max(nb, nj))), dim3(256), 0, stream_, d_bodies, nb, bld_.cc_parent.p, bld_.cc_static.p, bld_.sb_small.p,
                       d_joints, nj, ncp_, bld_.partner_first.p, bld_.partner_tag);
    Schedule sc;
    sc.colour_offsets.assign(1, 0); sc.group_offsets.assign(1, 0); sc.group_first_colour.assign(1, 0); sc.group_body_offsets.assign(1, 0);
    sc.islands = want_islands; sc.lds_on_host = false;
    int nbins = 0, lds_slots = 0, where = 0, ncomp_total = 0;
    bool any_partitioned = false;        // some component has more than COLOUR_B_MAX_JOINTS joints (schedule.h)
    spec_bins_pending_ = false;
    if (spec_build_applies(want_islands, nj)) {
        PHX_TRY(build_bins_speculative(d_bodies, nb, d_joints, nj, sc));
        sched_ = std::move(sc);
        return PHX_OK;
    }
    {
    // (Single mode needs the components too: the colouring candidate is chosen per component, schedule.h — it then sends
    //  every component to the HBM group)
    // 1. connected components: one linking pass + one flattening pass (schedule_kernels.h), and
    // 2. the components numbered in body order with their joints counted; count and sizes come back in one round trip
    unsigned ncomp_u = 0;
    std::vector<unsigned> comp_size, comp_units;
    int guess = 0;
    {
        hipLaunchKernelGGL(k_cc_link, dim3(grid_for(nj)), dim3(256), 0, stream_, d_joints, nj, nb, bld_.cc_parent.p, (const unsigned char*)bld_.cc_static.p,
                           (const unsigned long long*)bld_.partner_first.p, bld_.partner_tag, ncp_, bld_.partner.p);
        hipLaunchKernelGGL(k_cc_compress, dim3(grid_for(nb)), dim3(256), 0, stream_, bld_.cc_parent.p, nb, bld_.sb_small.p);
        PHX_TRY(device_exclusive_scan_of(RootFlagLoad{(const int*)bld_.cc_parent.p, nb, bld_.comp_size.p, bld_.comp_units.p}, bld_.cc_flags.p, nb + 1,
                                         reinterpret_cast<unsigned*>(bld_.sb_small.p + 1), bld_.sort_scan, stream_));
        hipLaunchKernelGGL(k_joint_components, dim3(std::max(1, std::min(div_up(nj, JC_T), 1024))), dim3(JC_T), 0, stream_, d_joints, nj, nb, (const int*)bld_.cc_parent.p,
                           (const unsigned*)bld_.cc_flags.p, (const int*)bld_.partner.p, bld_.joint_comp.p, bld_.comp_size.p, bld_.comp_units.p, bld_.sb_small.p);
        // fetch as many sizes as the previous build needed (+25 %); the rest, if any, in a second trip
        guess = std::min(nb, std::max(1024, ncomp_guess_ + ncomp_guess_ / 4));
        comp_size.assign(std::max(guess, 1), 0u); comp_units.assign(std::max(guess, 1), 0u);
        PHX_TRY(with_fingerprint());
        int pair[2] = {0, 0};                              // {labels disagree, component count}: adjacent words, one copy
        PHX_TRY(rb_.add(pair, bld_.sb_small.p, sizeof pair, stream_));
        PHX_TRY(rb_.add(comp_size.data(), bld_.comp_size.p, (size_t)guess * sizeof(unsigned), stream_));
        PHX_TRY(rb_.add(comp_units.data(), bld_.comp_units.p, (size_t)guess * sizeof(unsigned), stream_));
        PHX_TRY(rb_.wait(stream_));
        if (pair[0]) { set_error("connected components: a joint's bodies carry different labels"); return PHX_ERR_STATE; }
        ncomp_u = (unsigned)pair[1];
    }
    lap("components+count");
    const int ncomp = (int)ncomp_u;
    ncomp_total = ncomp;
    if (ncomp > guess) {
        comp_size.resize(ncomp); comp_units.resize(ncomp);
        PHX_TRY(rb_.add(comp_size.data() + guess, bld_.comp_size.p + guess, (size_t)(ncomp - guess) * sizeof(unsigned), stream_));
        PHX_TRY(rb_.add(comp_units.data() + guess, bld_.comp_units.p + guess, (size_t)(ncomp - guess) * sizeof(unsigned), stream_));
        PHX_TRY(rb_.wait(stream_));
    }
    comp_size.resize(std::max(ncomp, 1)); comp_units.resize(std::max(ncomp, 1));
    ncomp_guess_ = ncomp;
    for (int c = 0; c < ncomp && !any_partitioned; ++c) any_partitioned = comp_size[c] > (unsigned)COLOUR_B_MAX_JOINTS;      // (schedule.h: such a component is partitioned)

    // 3. host: GatherIslands' published numbers, workgroup shape, greedy binning of consecutive components
    //    (identical to schedule.hip::build_island_schedule — ncomp integers of work)
    {
        int run = 0, count = 0, mx = 0;
        for (int c = 0; c < ncomp; ++c) {
            run += (int)comp_size[c];
            if (run >= 256 || (run > 0 && c == ncomp - 1)) { ++count; mx = std::max(mx, run); run = 0; }
        }
        sc.island_count = count; sc.island_max_size = mx;
    }
    // the workgroup shape: units = lanes of the island kernel, joints = twice that; the roomier shape only if some component
    // needs it and fits it (identical to schedule.hip::build_island_schedule)
    int cap_units = ISL_T, cap_bodies = ISL_B;
    auto fits = [&](int c, int units) { return (int)comp_size[c] <= 2 * units && (int)comp_units[c] <= units; };
    for (int c = 0; c < ncomp; ++c) if (comp_size[c] && !fits(c, ISL_T) && fits(c, ISL_T_BIG)) { cap_units = ISL_T_BIG; cap_bodies = ISL_B_BIG; break; }
    sc.lds_lanes = cap_units;
    std::vector<int> bin_of(std::max(ncomp, 1), -1), rank_of(std::max(ncomp, 1), 0);     // rank of a component inside its bin (schedule.h: the colouring candidate is chosen per component)
    {
        int size = 0, units = 0, rank = 0;
        bool open = false;
        for (int c = 0; c < ncomp; ++c) {
            if (c % BIN_CHUNK == 0) { open = false; size = 0; units = 0; }      // (schedule.h BINNING: a bin never spans a chunk boundary)
            const int n = (int)comp_size[c], u = (int)comp_units[c];
            if (n == 0) continue;
            if (!want_islands || !fits(c, cap_units)) { open = false; size = 0; units = 0; continue; }     // -> HBM group (Single mode: every component)
            if (!open || size + n > 2 * cap_units || units + u > cap_units) { sc.group_offsets.push_back(sc.group_offsets.back()); ++nbins; open = true; size = 0; units = 0; rank = 0; }
            bin_of[c] = nbins - 1;
            rank_of[c] = rank++;
            size += n; units += u;
            sc.group_offsets.back() += n;
        }
    }
    lds_slots = sc.group_offsets.back();
    for (int c = 0; c < ncomp; ++c) if (bin_of[c] < 0) bin_of[c] = nbins;
    sc.lds_groups = nbins;
    // one upload: component -> bin, component -> rank inside its bin, bin -> first slot
    const size_t nc1 = (size_t)std::max(ncomp, 1), table_words = 2 * nc1 + (size_t)nbins + 2;
    PHX_TRY(bld_.bin_tables.reserve(table_words)); PHX_TRY(bld_.bin_tables_host.reserve(table_words));
    std::copy(bin_of.begin(), bin_of.end(), bld_.bin_tables_host.p);
    std::copy(rank_of.begin(), rank_of.end(), bld_.bin_tables_host.p + nc1);
    std::copy(sc.group_offsets.begin(), sc.group_offsets.begin() + nbins + 1, bld_.bin_tables_host.p + 2 * nc1);
    if (table_words <= 65536)
        hipLaunchKernelGGL(k_upload_words, dim3(std::max(1, std::min(div_up((int)table_words, 256), 64))), dim3(256), 0, stream_,
                           reinterpret_cast<unsigned*>(bld_.bin_tables.p), reinterpret_cast<const unsigned*>(bld_.bin_tables_host.p), (int)table_words);
    else PHX_HIP(hipMemcpyAsync(bld_.bin_tables.p, bld_.bin_tables_host.p, table_words * sizeof(int), hipMemcpyHostToDevice, stream_));
    const int* bin_of_comp = bld_.bin_tables.p; const int* rank_of_comp = bld_.bin_tables.p + nc1; const int* grp_goff = bld_.bin_tables.p + 2 * nc1;
    lap("bin");

    // 4. joints grouped by bin, joint order inside a bin (stable sort), HBM-group joints last
    if (nbins) {
        hipLaunchKernelGGL(k_joint_bin_keys, dim3(grid_for(nj)), dim3(256), 0, stream_, (const int*)bld_.joint_comp.p, bin_of_comp, nj, nbins,
                           bld_.sort_keys[0].p, bld_.sort_vals[0].p, bld_.sb_small.p + 2, std::max(ncomp, 1));
        int bits = 1;
        while ((1 << bits) <= nbins) ++bits;
        PHX_TRY(device_radix_sort_pairs(bld_.sort_keys[0].p, bld_.sort_vals[0].p, bld_.sort_keys[1].p, bld_.sort_vals[1].p, nj, bits, bld_.sort_hist.p, bld_.sort_scan, stream_, &where));
    } else {                                    // no bins (Single mode, or nothing fits a workgroup): the HBM group is every joint, in joint order
        hipLaunchKernelGGL(k_iota, dim3(grid_for(nj)), dim3(256), 0, stream_, bld_.sort_vals[0].p, nj);
        PHX_HIP(hipMemsetAsync(bld_.sb_small.p + 2, 0, sizeof(int), stream_));
    }
    lap("sort");

    // 5. one workgroup per bin: body table, colouring, slot arrays
    PHX_TRY(isl_.desc.reserve(std::max(nbins, 1))); PHX_TRY(isl_.ncol.reserve(std::max(nbins, 1)));
    PHX_TRY(isl_.bodies.reserve((size_t)std::max(nbins, 1) * cap_bodies));
    PHX_TRY(isl_.slot_local.reserve(std::max(lds_slots, 1))); PHX_TRY(isl_.slot_colour.reserve(std::max(lds_slots, 1)));
    if (nbins) {
        BinBuildView bv{};
        PHX_TRY(isl_.units.reserve(nbins)); PHX_TRY(isl_.unit_recs.reserve(2 * (size_t)nbins * cap_units));
        bv.sorted_joints = bld_.sort_vals[where].p; bv.group_offsets = grp_goff; bv.joints = d_joints; bv.partner = bld_.partner.p; bv.is_static = bld_.cc_static.p;
        bv.joint_comp = bld_.joint_comp.p; bv.comp_rank = rank_of_comp;
        bv.nb = nb; bv.max_static = 1 << 30;
        bv.order = hbm_.order.p; bv.slot_local = isl_.slot_local.p; bv.slot_colour = isl_.slot_colour.p; bv.desc = isl_.desc.p; bv.ncol = isl_.ncol.p;
        bv.units = isl_.units.p; bv.unit_recs = isl_.unit_recs.p;
        bv.bodies = isl_.bodies.p; bv.rejected = bld_.sb_small.p + 2; bv.poison = hash_.p + hash_slot_;
        if (cap_units > ISL_T) hipLaunchKernelGGL((k_build_bin<ISL_T_BIG, ISL_B_BIG>), dim3(nbins), dim3(2 * ISL_T_BIG), 0, stream_, bv);
        else hipLaunchKernelGGL((k_build_bin<ISL_T, ISL_B>), dim3(nbins), dim3(2 * ISL_T), 0, stream_, bv);
    }
    PHX_HIP(hipGetLastError());
    // Did every bin fit?  Normally NOT waited for here: a rejected bin spoils the solve's fingerprint word on the device, the
    // solve is queued behind the build, commits nothing if that happened, and synchronize() finds out (with the classes per group,
    // a statistic) in the round trip it makes anyway — the host's wait then overlaps the island kernel instead of idling the GPU.
    // A sharded solve needs the groups' body counts for its exchange layout now.
    grp_body_count_.clear();
    sc.lds_colours = 0;
    if (nbins && (shard_count_ > 1 || xch_send_ || !defer_build_check_)) {
        int rejected = 0;
        std::vector<int> ncol(nbins, 0);
        std::vector<int4> desc(nbins);
        PHX_TRY(rb_.add(&rejected, bld_.sb_small.p + 2, sizeof rejected, stream_));
        PHX_TRY(rb_.add(ncol.data(), isl_.ncol.p, (size_t)nbins * sizeof(int), stream_));
        PHX_TRY(rb_.add(desc.data(), isl_.desc.p, (size_t)nbins * sizeof(int4), stream_));
        PHX_TRY(rb_.wait(stream_));
        lap("bins");
        if (rejected) { *fallback = true; return PHX_OK; }      // some bin exceeds the LDS caps: let the host builder sort it out
        for (int g = 0; g < nbins; ++g) sc.lds_colours += ncol[g];
        for (const int4& d : desc) grp_body_count_.push_back(d.w);
    } else if (nbins) {
        build_unverified_ = true;
        unverified_bins_ = nbins;
    }
    }
    const int rest = nj - lds_slots;

    // 6. the HBM group (components too big for a workgroup, static-static joints; every joint in Single mode): the same
    //    first-fit-by-priority colouring — a walk of the dependency graph, one launch per frontier — then a stable sort by class
    nstatic_ = 0;
    sc.hbm_body_count = 0;
    if (rest > 0) {
        const unsigned* ids = bld_.sort_vals[where].p + lds_slots;
        PHX_TRY(bld_.jp_used.reserve(nbs)); PHX_TRY(bld_.jp_used_b.reserve(nbs)); PHX_TRY(bld_.jp_touched.reserve(nbs + 1)); PHX_TRY(bld_.jp_degree.reserve(nbs + 1));
        PHX_TRY(bld_.jp_offset.reserve(nbs + 1)); PHX_TRY(bld_.jp_cursor.reserve(nbs));
        PHX_TRY(bld_.jp_small.reserve(2 * JP_MAX_COLOURS + 8)); PHX_TRY(bld_.jp_kind.reserve(njs)); PHX_TRY(bld_.jp_counts.reserve((size_t)(JP_ROUNDS_MAX + 2) * JP_SUBLISTS));
        PHX_TRY(bld_.jp_seen.reserve(3 * ((size_t)ncomp_total + 1))); PHX_TRY(bld_.jp_bad_b.reserve((size_t)ncomp_total + 1));
        // (sized by the joint count, not by the group's: while a world settles the HBM group grows every step, and regrowing a score
        //  of arrays — hipMalloc + hipFree each — cost 3 ms whenever it crossed a capacity)
        PHX_TRY(bld_.jp_ent.reserve(njs)); PHX_TRY(bld_.jp_succ.reserve(njs)); PHX_TRY(bld_.jp_pred.reserve(njs)); PHX_TRY(bld_.jp_colour_b.reserve(njs));
        for (int k = 0; k < 2; ++k) { PHX_TRY(bld_.jp_keys[k].reserve(njs)); PHX_TRY(bld_.jp_vals[k].reserve(njs)); PHX_TRY(bld_.jp_list[k].reserve((size_t)njs * JP_SUBLISTS)); }
        PHX_TRY(bld_.jp_adj.reserve(2 * (size_t)njs)); PHX_TRY(bld_.jp_ent_comp.reserve(njs));
        JpView jv{};
        jv.ids = ids; jv.count = rest; jv.joints = d_joints; jv.is_static = bld_.cc_static.p; jv.nb = nb;
        jv.ent = bld_.jp_ent.p; jv.offset = bld_.jp_offset.p; jv.cursor = bld_.jp_cursor.p; jv.adj = bld_.jp_adj.p; jv.ent_comp = bld_.jp_ent_comp.p;
        jv.succ = bld_.jp_succ.p; jv.pred = bld_.jp_pred.p;
        jv.used = bld_.jp_used.p; jv.used_b = bld_.jp_used_b.p; jv.colour = bld_.jp_keys[0].p; jv.colour_b = bld_.jp_colour_b.p; jv.touched = bld_.jp_touched.p;
        jv.joint_comp = bld_.joint_comp.p; jv.partner = bld_.partner.p; jv.kind = bld_.jp_kind.p; jv.ncomp = ncomp_total; jv.comp_size = bld_.comp_size.p;
        jv.seen_a = bld_.jp_seen.p; jv.seen_b = bld_.jp_seen.p + ncomp_total + 1; jv.seen_c = bld_.jp_seen.p + 2 * ((size_t)ncomp_total + 1); jv.bad_b = bld_.jp_bad_b.p;
        jv.counts = bld_.jp_counts.p; jv.flags = bld_.jp_small.p; jv.hist = reinterpret_cast<unsigned*>(bld_.jp_small.p + 4);
        const unsigned* perm = nullptr;                     // the entries sorted by part (partitioned components only)
        const int parts = parts_total(nb);                    // over both levels (schedule.h)
        // the dependency graph of the colouring (schedule_kernels.h): entry cache + degrees, lists per dynamic body ordered by
        // priority, successor links and predecessor counts
        hipLaunchKernelGGL(k_jp_clear, dim3(grid_for(std::max(nb + 1, ncomp_total + 1))), dim3(256), 0, stream_, jv, JP_ROUNDS_MAX + 1);
        hipLaunchKernelGGL(k_jp_prepare, dim3(grid_for(rest)), dim3(256), 0, stream_, jv);
        // the interior units of partitioned components take their classes inside their parts (k_colour_parts): entries sorted by
        // part (everything else behind them), the parts' ranges, one workgroup per part — out of the global walk below altogether
        if (any_partitioned) {
            for (int k = 0; k < 2; ++k) { PHX_TRY(parts_.keys[k].reserve(njs)); PHX_TRY(parts_.vals[k].reserve(njs)); }
            PHX_TRY(parts_.begin.reserve((size_t)parts + 2));
            hipLaunchKernelGGL(k_part_sort_keys, dim3(grid_for(rest)), dim3(256), 0, stream_, jv, (unsigned)parts, parts_.keys[0].p, parts_.vals[0].p);
            int bits = 1;
            while ((1 << bits) <= parts) ++bits;                 // keys 0 .. parts
            int wherep = 0;
            PHX_TRY(device_radix_sort_pairs(parts_.keys[0].p, parts_.vals[0].p, parts_.keys[1].p, parts_.vals[1].p, rest, bits, bld_.sort_hist.p, bld_.sort_scan, stream_, &wherep));
            hipLaunchKernelGGL(k_lower_bounds, dim3(grid_for(parts + 1)), dim3(256), 0, stream_, (const unsigned*)parts_.keys[wherep].p, rest, parts, parts_.begin.p);
            hipLaunchKernelGGL(k_colour_parts, dim3(parts), dim3(CP_T), 0, stream_, jv, (const unsigned*)parts_.vals[wherep].p, (const int*)parts_.begin.p);
            perm = parts_.vals[wherep].p;
            // the parts' slot ranges per interior class, left by k_jp_place below
            PHX_TRY(parts_.ranges.reserve((size_t)parts * JP_MAX_COLOURS));
            PHX_HIP(hipMemsetAsync(parts_.ranges.p, 0, (size_t)parts * JP_MAX_COLOURS * sizeof(int4), stream_));
        }
        PHX_TRY(device_exclusive_scan(bld_.jp_offset.p, nb + 1, nullptr, bld_.sort_scan, stream_));
        hipLaunchKernelGGL(k_jp_fill, dim3(grid_for(rest)), dim3(256), 0, stream_, jv);
        hipLaunchKernelGGL(k_jp_lists, dim3(std::max(1, std::min(div_up(2 * rest, 256), 8192))), dim3(256), 0, stream_, jv);
        {   // round 0's frontier: flags, scan, compaction
            PHX_TRY(bld_.jp_seed.reserve((size_t)njs + 1));
            hipLaunchKernelGGL(k_jp_seed_flags, dim3(grid_for(rest + 1)), dim3(256), 0, stream_, jv, bld_.jp_seed.p);
            PHX_TRY(device_exclusive_scan(bld_.jp_seed.p, rest + 1, nullptr, bld_.sort_scan, stream_));
            hipLaunchKernelGGL(k_jp_seed, dim3(grid_for(rest)), dim3(256), 0, stream_, jv, (const unsigned*)bld_.jp_seed.p, bld_.jp_list[0].p);
        }
        // the rounds: as many as the previous build needed (+2) before the first look at the frontier, then in small batches
        int round = 0;
        for (bool done = false; !done;) {
            const int batch = round == 0 ? std::min(std::max(jp_rounds_guess_ + 2, JP_BATCH), JP_ROUNDS_MAX) : JP_BATCH;
            if (round + batch > JP_ROUNDS_MAX) { *fallback = true; return PHX_OK; }           // pathological dependency chain: host builder
            for (int k = 0; k < batch; ++k, ++round)
                hipLaunchKernelGGL(k_jp_front, dim3(JP_SUBLISTS * std::max(1, std::min(div_up(rest, JP_FRONT_T * JP_ITEMS * JP_SUBLISTS), 64))), dim3(JP_FRONT_T), 0, stream_, jv, round, (const unsigned*)bld_.jp_list[round & 1].p, bld_.jp_list[(round + 1) & 1].p);
            int flags = 0;
            std::vector<int> sizes(((size_t)batch + 1) * JP_SUBLISTS, 0);                      // the frontiers of this batch's rounds and of the next one
            PHX_TRY(with_fingerprint());
            PHX_TRY(rb_.add(sizes.data(), bld_.jp_counts.p + (size_t)(round - batch) * JP_SUBLISTS, sizes.size() * sizeof(int), stream_));
            PHX_TRY(rb_.add(&flags, bld_.jp_small.p, sizeof(int), stream_));
            PHX_TRY(rb_.wait(stream_));
            if (flags & 1) { set_error("a joint references a body out of range"); return PHX_ERR_INVALID; }
            if (flags & 6) { *fallback = true; return PHX_OK; }                               // > 64 colours or a body in thousands of joints: host builder
            for (int k = 0; k <= batch && !done; ++k) {
                int n = 0;
                for (int q = 0; q < JP_SUBLISTS; ++q) n += sizes[(size_t)k * JP_SUBLISTS + q];
                if (n == 0) { done = true; jp_rounds_guess_ = round - batch + k; }
            }
        }
        if (trace) fprintf(stderr, "[schedule/gpu] HBM group: %d joints, %d rounds (%d launched)\n", rest, jp_rounds_guess_, round);
        lap("rest/colour");
        hipLaunchKernelGGL(k_jp_interior_classes, dim3(grid_for(std::max(ncomp_total, 1))), dim3(256), 0, stream_, jv);
        hipLaunchKernelGGL(k_jp_choose, dim3(grid_for(rest)), dim3(256), 0, stream_, jv);          // sort keys (class, kind) + their histogram
        // bodies touched, static slots: two small scans; one readback with the histogram
        unsigned* hist = jv.hist;
        PHX_TRY(device_exclusive_scan(bld_.jp_touched.p, nb + 1, nullptr, bld_.sort_scan, stream_));
        PHX_TRY(hbm_.hbm_body_list.reserve(nbs));
        hipLaunchKernelGGL(k_compact_flagged, dim3(grid_for(nb)), dim3(256), 0, stream_, (const unsigned*)bld_.jp_touched.p, nb, hbm_.hbm_body_list.p);
        // leaders sorted by (class, kind), stable in joint order (followers behind them all); then every leader places itself
        // and its follower
        // (the sort's input is gathered in part order where there are parts: one 8-bit pass then leaves the interior classes laid out
        //  part by part; jv.colour IS bld_.jp_keys[0], so the gather goes to the other pair)
        int where2 = 0;
        hipLaunchKernelGGL(k_jp_sort_input, dim3(grid_for(rest)), dim3(256), 0, stream_, jv, perm, bld_.jp_keys[1].p, bld_.jp_vals[1].p);
        PHX_TRY(device_radix_sort_pairs(bld_.jp_keys[1].p, bld_.jp_vals[1].p, bld_.jp_keys[0].p, bld_.jp_vals[0].p, rest, 8, bld_.sort_hist.p, bld_.sort_scan, stream_, &where2));
        hipLaunchKernelGGL(k_jp_place, dim3(grid_for(rest)), dim3(256), 0, stream_, jv, (const unsigned*)bld_.jp_keys[where2 ^ 1].p, (const unsigned*)bld_.jp_vals[where2 ^ 1].p,
                           hbm_.order.p + lds_slots, lds_slots, perm ? reinterpret_cast<int*>(parts_.ranges.p) : (int*)nullptr);
        unsigned h_hist[2 * JP_MAX_COLOURS], h_touched = 0;
        PHX_TRY(rb_.add(h_hist, hist, sizeof h_hist, stream_));
        PHX_TRY(rb_.add(&h_touched, bld_.jp_touched.p + nb, sizeof h_touched, stream_));
        // static slots (only the HBM path indexes the global static-tag tables)
        // (a table of its own: the readback batch reads its sources at wait(), so bld_.jp_touched must stay as it is until then)
        PHX_TRY(hbm_.static_slot.reserve(nbs));
        unsigned* sflags = bld_.jp_degree.p;                       // per body + 1; the colouring is done with it
        hipLaunchKernelGGL(k_static_flags, dim3(grid_for(nb + 1)), dim3(256), 0, stream_, (const unsigned char*)bld_.cc_static.p, nb, sflags);
        PHX_TRY(device_exclusive_scan(sflags, nb + 1, nullptr, bld_.sort_scan, stream_));
        hipLaunchKernelGGL(k_static_slots, dim3(grid_for(nb)), dim3(256), 0, stream_, (const unsigned char*)bld_.cc_static.p, (const unsigned*)sflags, nb, hbm_.static_slot.p);
        unsigned h_nstatic = 0;
        int h_flags[3] = {0, 0, 0};                            // [0] bit 1: the interior + other classes exceed the device builder's 64; [1] KI0; [2] KI1
        PHX_TRY(rb_.add(&h_nstatic, sflags + nb, sizeof h_nstatic, stream_));
        PHX_TRY(rb_.add(h_flags, bld_.jp_small.p, sizeof h_flags, stream_));
        PHX_TRY(rb_.wait(stream_));
        if (h_flags[0] & 2) { *fallback = true; return PHX_OK; }
        sc.hbm_interior_classes = h_flags[1] + h_flags[2]; sc.hbm_interior_classes0 = h_flags[1];
        nstatic_ = (int)h_nstatic;
        sc.hbm_body_count = (int)h_touched;
        sc.hbm_colour_offsets.assign(1, lds_slots);
        sc.hbm_class_leaders.clear();
        for (int c = 0; c < JP_MAX_COLOURS; ++c) {
            const int with = (int)h_hist[2 * c], single = (int)h_hist[2 * c + 1];
            if (!(with + single)) continue;
            sc.hbm_colour_offsets.push_back(sc.hbm_colour_offsets.back() + 2 * with + single);
            sc.hbm_class_leaders.push_back(with + single);
        }
        if (sc.hbm_colour_offsets.back() != nj) { set_error("HBM group colouring lost joints"); return PHX_ERR_STATE; }
        sc.group_offsets.push_back(nj);
        // k_solve_parts' tables: the classes' slot layout; the parts' ranges were left by k_jp_place, their unit counts by the sort by part
        parts_.count = 0;
        if (sc.hbm_interior_classes > 0) {
            const int ki = sc.hbm_interior_classes;
            if (!perm || ki >= (int)sc.hbm_class_leaders.size() + 1 || ki > JP_MAX_COLOURS) { set_error("interior classes out of range"); return PHX_ERR_STATE; }
            int interior_leaders = 0;
            PHX_TRY(upload_class_tab(sc, &interior_leaders));
            parts_.count = parts;
        }
        PHX_TRY(hbm_.sb_imp.reserve(nbs)); PHX_TRY(hbm_.sb_disp.reserve(nbs));
        PHX_TRY(hbm_.q0.reserve(njs)); PHX_TRY(hbm_.q1.reserve(njs)); PHX_TRY(hbm_.q2.reserve(njs)); PHX_TRY(hbm_.q3.reserve(njs)); PHX_TRY(hbm_.qn.reserve(njs));
        PHX_TRY(hbm_.acc.reserve(njs)); PHX_TRY(hbm_.dd.reserve(njs));
    }
    PHX_TRY(hbm_.sw.reserve(4 * (size_t)std::max(nstatic_, 1)));
    // new table for this solve — already cleared by this solve's fingerprint kernel unless it has just been (re)allocated
    if (hbm_.sw.p != sw_cleared_ || 4 * (size_t)std::max(nstatic_, 1) > sw_cleared_words_)
        PHX_HIP(hipMemsetAsync(hbm_.sw.p, 0, 4 * (size_t)std::max(nstatic_, 1) * sizeof(unsigned), stream_));
    lap("rest");
    // the next rebuild may skip the host altogether (build_bins_speculative) if this one was nothing but bins of one shape
    spec_bins_ok_ = want_islands && rest == 0 && nbins > 0 && ncomp_total <= BINC_MAX;
    spec_bins_guess_ = nbins; spec_lanes_ = sc.lds_lanes;
    sched_ = std::move(sc);
    return PHX_OK;
}

// Speculative binning: the rebuild of a world that was nothing but LDS-sized islands last time (every stack scene) runs without a
// single host round trip.  The connected components are followed by k_bin_components (schedule_kernels.h), which makes the
// bins the host loop above would make; the sort, k_build_bin and the island kernel are launched with last build's bin count
// (+ slack) as their grid and take the real count from the device.  Whatever does not hold any more — a component that fits
// no workgroup, joints between static bodies, more bins than the grid, the other workgroup shape, unconverged components —
// spoils the solve's fingerprint word like a rejected bin does: the solve commits nothing, synchronize() rebuilds the
// long way and repeats it.  The topology hash the host has not seen is replaced on the device by a constant it knows
// (`gate_expected_`), which is what the solve's kernels compare the word with; the hash itself comes back with the results.
int DeviceSolver::build_bins_speculative(const float4* d_bodies, int nb, const phx_contact_joint* d_joints, int nj, Schedule& sc)
{
    hipLaunchKernelGGL(k_cc_link, dim3(grid_for(nj)), dim3(256), 0, stream_, d_joints, nj, nb, bld_.cc_parent.p, (const unsigned char*)bld_.cc_static.p,
                       (const unsigned long long*)bld_.partner_first.p, bld_.partner_tag, ncp_, bld_.partner.p);
    hipLaunchKernelGGL(k_cc_compress, dim3(grid_for(nb)), dim3(256), 0, stream_, bld_.cc_parent.p, nb, bld_.sb_small.p);
    PHX_TRY(device_exclusive_scan_of(RootFlagLoad{(const int*)bld_.cc_parent.p, nb, bld_.comp_size.p, bld_.comp_units.p}, bld_.cc_flags.p, nb + 1,
                                     reinterpret_cast<unsigned*>(bld_.sb_small.p + 1), bld_.sort_scan, stream_));
    hipLaunchKernelGGL(k_joint_components, dim3(std::max(1, std::min(div_up(nj, JC_T), 1024))), dim3(JC_T), 0, stream_, d_joints, nj, nb, (const int*)bld_.cc_parent.p,
                       (const unsigned*)bld_.cc_flags.p, (const int*)bld_.partner.p, bld_.joint_comp.p, bld_.comp_size.p, bld_.comp_units.p, bld_.sb_small.p);
    // (workgroups beyond the real bin count leave at once, and a settling world doubles its bins within a few steps — columns
    //  break in two: a roomy grid costs nothing, a grid too small costs a repeated solve)
    // (up to 2047 bins the joints are grouped by ONE 11-bit radix pass — three launches instead of six — so a grid just above
    //  that is capped there while it still leaves a quarter of slack)
    int grid = 2 * spec_bins_guess_ + 64;
    if (grid > 2047 && spec_bins_guess_ + spec_bins_guess_ / 4 + 16 <= 2047) grid = 2047;
    const int cap_units = spec_lanes_, cap_bodies = cap_units > ISL_T ? ISL_B_BIG : ISL_B;
    PHX_TRY(bld_.bin_tables.reserve(2 * (size_t)BINC_MAX + (size_t)grid + 2));
    PHX_TRY(bld_.bin_result.reserve(16));
    BinCompView cv{};
    cv.comp_size = bld_.comp_size.p; cv.comp_units = bld_.comp_units.p; cv.cc_small = bld_.sb_small.p; cv.nj = nj;
    cv.cap_units = cap_units; cv.small_units = ISL_T; cv.max_bins = grid;
    cv.bin_of = bld_.bin_tables.p; cv.rank_of = bld_.bin_tables.p + BINC_MAX; cv.goff = bld_.bin_tables.p + 2 * BINC_MAX;
    cv.result = bld_.bin_result.p;
    cv.fingerprint = hash_.p + hash_slot_; cv.hash_out = reinterpret_cast<unsigned long long*>(bld_.bin_result.p + 8);
    gate_expected_ = 0x5EED000000000000ull | (++gate_serial_ & 0xFFFFFFFFFFFFull);
    cv.gate = gate_expected_;
    hipLaunchKernelGGL(k_bin_components, dim3(1), dim3(BINC_T), 0, stream_, cv);
    hipLaunchKernelGGL(k_joint_bin_keys, dim3(grid_for(nj)), dim3(256), 0, stream_, (const int*)bld_.joint_comp.p, (const int*)cv.bin_of, nj, grid,
                       bld_.sort_keys[0].p, bld_.sort_vals[0].p, bld_.sb_small.p + 2, BINC_MAX);
    int bits = 1, where = 0;
    while ((1 << bits) <= grid) ++bits;
    PHX_TRY(device_radix_sort_pairs(bld_.sort_keys[0].p, bld_.sort_vals[0].p, bld_.sort_keys[1].p, bld_.sort_vals[1].p, nj, bits, bld_.sort_hist.p, bld_.sort_scan, stream_, &where));
    PHX_TRY(isl_.desc.reserve(grid)); PHX_TRY(isl_.ncol.reserve(grid));
    PHX_TRY(isl_.bodies.reserve((size_t)grid * cap_bodies));
    PHX_TRY(isl_.slot_local.reserve(nj)); PHX_TRY(isl_.slot_colour.reserve(nj));
    PHX_TRY(isl_.units.reserve(grid)); PHX_TRY(isl_.unit_recs.reserve(2 * (size_t)grid * cap_units));
    BinBuildView bv{};
    bv.sorted_joints = bld_.sort_vals[where].p; bv.group_offsets = cv.goff; bv.joints = d_joints; bv.partner = bld_.partner.p; bv.is_static = bld_.cc_static.p;
    bv.joint_comp = bld_.joint_comp.p; bv.comp_rank = cv.rank_of;
    bv.nb = nb; bv.max_static = 1 << 30;
    bv.order = hbm_.order.p; bv.slot_local = isl_.slot_local.p; bv.slot_colour = isl_.slot_colour.p; bv.desc = isl_.desc.p; bv.ncol = isl_.ncol.p;
    bv.units = isl_.units.p; bv.unit_recs = isl_.unit_recs.p;
    bv.bodies = isl_.bodies.p; bv.rejected = bld_.sb_small.p + 2; bv.poison = hash_.p + hash_slot_;
    bv.nbins_dev = bld_.bin_result.p;
    if (cap_units > ISL_T) hipLaunchKernelGGL((k_build_bin<ISL_T_BIG, ISL_B_BIG>), dim3(grid), dim3(2 * ISL_T_BIG), 0, stream_, bv);
    else hipLaunchKernelGGL((k_build_bin<ISL_T, ISL_B>), dim3(grid), dim3(2 * ISL_T), 0, stream_, bv);
    PHX_HIP(hipGetLastError());
    // provisional: the launch grid stands in for the group count until the solve is settled (collect_stats)
    sc.lds_groups = grid; sc.lds_lanes = cap_units;
    sc.group_offsets.assign((size_t)grid + 1, nj); sc.group_offsets[0] = 0;
    sc.island_count = sched_.island_count; sc.island_max_size = sched_.island_max_size;
    sc.lds_colours = sched_.lds_colours; sc.hbm_body_count = 0;
    grp_body_count_.clear();
    nstatic_ = 0;
    PHX_TRY(hbm_.sw.reserve(4));      // (the HBM path's static-tag table: a schedule of nothing but LDS groups never reads it — no clearing dispatch)
    build_unverified_ = true;
    unverified_bins_ = grid;
    spec_bins_pending_ = true;
    return PHX_OK;
}

// The LDS groups of a device-built schedule live in HBM; the query API (and the parity tests that replay the
// schedule through the oracle) need them on the host.
int DeviceSolver::materialise_schedule()
{
    if (sched_.lds_on_host) return PHX_OK;
    const int lg = sched_.lds_groups;
    const int lds_slots = lg ? sched_.group_offsets[lg] : 0;
    std::vector<int> order(std::max(nj_, 1)), ncol(std::max(lg, 1));
    std::vector<unsigned char> colour(std::max(lds_slots, 1));
    PHX_TRY(use_device(device_));
    if (nj_) PHX_HIP(hipMemcpy(order.data(), hbm_.order.p, (size_t)nj_ * sizeof(int), hipMemcpyDeviceToHost));
    if (lds_slots) {
        PHX_HIP(hipMemcpy(colour.data(), isl_.slot_colour.p, (size_t)lds_slots, hipMemcpyDeviceToHost));
        PHX_HIP(hipMemcpy(ncol.data(), isl_.ncol.p, (size_t)lg * sizeof(int), hipMemcpyDeviceToHost));
    }
    sched_.order.assign(order.begin(), order.begin() + nj_);
    sched_.colour_offsets.assign(1, 0);
    sched_.group_first_colour.assign(1, 0);
    for (int g = 0; g < lg; ++g) {
        std::vector<int> count(ncol[g], 0);
        for (int s = sched_.group_offsets[g]; s < sched_.group_offsets[g + 1]; ++s) count[colour[s]]++;
        int at = sched_.group_offsets[g];
        for (int c = 0; c < ncol[g]; ++c) { at += count[c]; sched_.colour_offsets.push_back(at); }
        sched_.group_first_colour.push_back((int)sched_.colour_offsets.size() - 1);
    }
    if (sched_.has_hbm_group()) {
        for (size_t c = 1; c < sched_.hbm_colour_offsets.size(); ++c) sched_.colour_offsets.push_back(sched_.hbm_colour_offsets[c]);
        sched_.group_first_colour.push_back((int)sched_.colour_offsets.size() - 1);
    }
    sched_.lds_on_host = true;
    return PHX_OK;
}

// The launch sequence of one SolveJoints, in three capturable segments (no sync, no allocation inside):
//   pre    PrepareBodies, PrepareJoints+RefreshJoints, PreStepJoints class by class
//   sweeps `iters` x colours fused impulse+displacement launches
//   post   FinishJoints, FinishBodies
// class_tab[c] = {first slot, leaders, followers, leaders of the classes before c} of the HBM group's classes (k_solve_parts)
int DeviceSolver::upload_class_tab(const Schedule& sc, int* interior_leaders)
{
    std::vector<int4>& tab = parts_.class_tab_host;          // (a member: the copy below is asynchronous)
    tab.assign(sc.hbm_class_leaders.size(), make_int4(0, 0, 0, 0));
    int before = 0;
    *interior_leaders = 0;
    for (size_t c = 0; c < tab.size(); ++c) {
        const int cb = sc.hbm_colour_offsets[c], lead = sc.hbm_class_leaders[c];
        tab[c] = make_int4(cb, lead, sc.hbm_colour_offsets[c + 1] - cb - lead, before);
        before += lead;
        if ((int)c < sc.hbm_interior_classes) *interior_leaders = before;
    }
    PHX_TRY(parts_.class_tab.reserve(std::max<size_t>(tab.size(), 64)));
    PHX_HIP(hipMemcpyAsync(parts_.class_tab.p, tab.data(), tab.size() * sizeof(int4), hipMemcpyHostToDevice, stream_));
    return PHX_OK;
}

// host-built schedules: the interior units by part as the builder left them (schedule.hip build_part_tables)
int DeviceSolver::upload_part_tables()
{
    parts_.count = 0;
    const int ki = sched_.hbm_interior_classes;
    if (ki <= 0 || sched_.part_begin.empty()) return PHX_OK;      // (more than 64 interior classes: no tables, one launch per class)
    int interior_leaders = 0;
    PHX_TRY(upload_class_tab(sched_, &interior_leaders));
    if (interior_leaders != sched_.part_begin.back()) { set_error("part tables do not match the interior classes"); return PHX_ERR_STATE; }
    const size_t parts = sched_.part_begin.size() - 1;
    PHX_TRY(parts_.ranges.reserve(parts * PARTS_CLASS_STRIDE)); PHX_TRY(parts_.begin.reserve(parts + 2));
    PHX_HIP(hipMemcpyAsync(parts_.ranges.p, sched_.part_ranges.data(), sched_.part_ranges.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    PHX_HIP(hipMemcpyAsync(parts_.begin.p, sched_.part_begin.data(), sched_.part_begin.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    PHX_HIP(hipStreamSynchronize(stream_));
    parts_.count = (int)parts;
    return PHX_OK;
}

int DeviceSolver::enqueue_pre(const BodyView& d_bodies, int nb, const phx_contact_point* d_cps, phx_contact_joint* d_joints, int nj)
{
    const SolverView v = view();
    // (the control words — productive flags, static tags, island counters — were cleared by launch_fingerprint's
    //  fingerprint kernel, which every solve runs first)
    // the HBM group (if any): PrepareBodies for the bodies it touches, PrepareJoints + RefreshJoints over its slots,
    // PreStep class by class.  Groups solved in LDS read and write the caller's records directly.
    const int hbm_bodies = sched_.hbm_body_count;
    if (nj && owns_hbm_group()) {
        hipLaunchKernelGGL(k_unpack_bodies, dim3(grid_for(hbm_bodies)), dim3(256), 0, stream_, d_bodies, (const int*)hbm_.hbm_body_list.p,
                           hbm_bodies, hbm_.sb_imp.p, hbm_.sb_disp.p, v.stamps);
        const int hb = sched_.hbm_begin(), he = sched_.hbm_end();
        hipLaunchKernelGGL(k_pack_refresh, dim3(grid_for(he - hb)), dim3(256), 0, stream_, v, hb, he, d_joints, d_cps, hbm_.static_slot.p);
        size_t c0 = 0;
        if (parts_in_use()) {          // the interior classes of partitioned components: one launch per level, a workgroup per part
            for (int level = 0; level < part_levels(); ++level) {
                const PartsView pv = parts_view(level, nb);
                hipLaunchKernelGGL(k_prestep_parts, dim3(pv.parts), dim3(PARTS_T), 0, stream_, v, pv);
            }
            c0 = (size_t)sched_.hbm_interior_classes;
        }
        for (size_t c = c0; c + 1 < sched_.hbm_colour_offsets.size(); ++c) {
            const int cb = sched_.hbm_colour_offsets[c], ce = sched_.hbm_colour_offsets[c + 1], lead = sched_.hbm_class_leaders[c];
            hipLaunchKernelGGL(k_prestep, dim3(grid_for(lead)), dim3(256), 0, stream_, v, cb, lead, ce - cb - lead);
        }
    }
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

int DeviceSolver::enqueue_sweeps(const BodyView& d_bodies, const phx_contact_point* d_cps, phx_contact_joint* d_joints, int nj, int ci, int pi, int mode_override)
{
    const SolverView v = view();
    const int iters = std::max(ci, pi);
    sweep_launches_ = 0;
    if (!nj) return PHX_OK;
    const int lg = sched_.lds_groups;
    if (shard_count_ > 1) PHX_TRY(ensure_partition());       // which groups this rank solves (exchange.h: longest processing time first)
    const int mine = shard_count_ > 1 ? mine_count_ : lg;
    bool forked = false;
    if (mine) {   // every LDS group: Refresh + PreStep + all sweeps in one launch, one workgroup per group
        IslandView iv{};
        iv.group_list = shard_count_ > 1 ? grp_mine_.p : nullptr;
        iv.ngroups_dev = spec_bins_pending_ ? bld_.bin_result.p : nullptr;
        iv.stamp_begin = iv.stamp_end = owns_hbm_group() ? 0 : 1;      // (with an HBM group, its first and last kernels leave the stamps)
        iv.desc = isl_.desc.p; iv.ncol = isl_.ncol.p; iv.units = isl_.units.p; iv.unit_recs = isl_.unit_recs.p; iv.bodies = isl_.bodies.p;
        iv.executed = isl_.stats.p + (size_t)hash_slot_ * STATS_SET; iv.visits = isl_.visits.p + (size_t)hash_slot_ * VISITS_SET;
        iv.trace = nullptr; iv.wave_trace = nullptr;
        // how the launch is gated (island_view.h).  A launch whose grid is only an upper bound of the group count (speculative
        // binning) is gated by the build it follows.
        iv.mode = mode_override >= 0 ? mode_override : isl_mode_;
        iv.nexpect = isl_nexpect_; iv.ctl = hash_.p + hash_slot_; iv.epoch = solve_epoch_;
        iv.shards = isl_.shards.p + (size_t)hash_slot_ * SHARDS_SET;
        iv.wait_polls = opt_.isl_wait_polls;
        if (iv.mode != ISL_GATED) {
            if (isl_.done.cap < (size_t)lg) {      // (a new table: no group carries any epoch)
                if (isl_.done.reserve((size_t)std::max(lg, 1)) != PHX_OK) return PHX_ERR_HIP;
                PHX_HIP(hipMemsetAsync(isl_.done.p, 0, isl_.done.cap * sizeof(unsigned), stream_));
            }
            iv.done = isl_.done.p;
        }
        if (island_clears_next_) {                 // no hash pass in front: this launch is the solve's first kernel
            const int next = hash_slot_ ^ 1;
            iv.next_ctl = hash_.p + next; iv.next_executed = isl_.stats.p + (size_t)next * STATS_SET; iv.next_visits = isl_.visits.p + (size_t)next * VISITS_SET;
            iv.next_shards = isl_.shards.p + (size_t)next * SHARDS_SET;
            island_clears_next_ = false;
        }
        if (trace_islands_) {
            // 8 words per group, then 8 words per wave (16 waves at most) of every group
            if (isl_.trace.reserve((size_t)std::max(lg, 1) * (8 + 128)) != PHX_OK) return PHX_ERR_HIP;
            PHX_HIP(hipMemsetAsync(isl_.trace.p, 0, (size_t)lg * (8 + 128) * sizeof(unsigned long long), stream_));
            iv.trace = isl_.trace.p;
            iv.wave_trace = isl_.trace.p + (size_t)lg * 8;
        }
        const bool big = sched_.lds_lanes > ISL_T;
        // A schedule with LDS islands AND an HBM group (a world that is merging, or settled around a few loose stacks): the island
        // launch is one group's chain of class steps — ~90 us whatever the group count — and touches nothing the HBM group's
        // classes x sweeps launches touch, so it runs beside them on a second stream: fork here, join behind the sweeps.
        forked = owns_hbm_group() && !opt_.use_graphs && !opt_.no_side_stream && !trace_islands_ && side_stream_;
        if (forked) {
            PHX_HIP(hipEventRecord(ev_fork_, stream_));
            PHX_HIP(hipStreamWaitEvent(side_stream_, ev_fork_, 0));
        }
        launch_solve_islands(forked ? side_stream_ : stream_, mine, big, half_state_, iv.trace != nullptr, v, iv, d_bodies, d_joints, d_cps, ci, pi);
        if (forked) PHX_HIP(hipEventRecord(ev_join_, side_stream_));
        ++sweep_launches_;
    }
    if (owns_hbm_group()) {
        const int ncol = (int)sched_.hbm_colour_offsets.size() - 1;
        for (int it = 0; it < iters; ++it) {
            const bool imp = it < ci, disp = it < pi;
            int c0 = 0;
            if (parts_in_use()) {      // classes [0, KI) of this sweep: one launch per level (solver_kernels.h k_solve_parts)
                for (int level = 0; level < part_levels(); ++level) {
                    const PartsView pv = parts_view(level, v.nb);
                    const dim3 g(pv.parts), b(PARTS_T);
                    if (level == 0) {
                        if (imp && disp) hipLaunchKernelGGL((k_solve_parts<true, true, false>), g, b, 0, stream_, v, pv, it);
                        else if (imp)    hipLaunchKernelGGL((k_solve_parts<true, false, false>), g, b, 0, stream_, v, pv, it);
                        else             hipLaunchKernelGGL((k_solve_parts<false, true, false>), g, b, 0, stream_, v, pv, it);
                    } else {           // a level-1 part has a few dozen units: a lane owns one, everything requested up front
                        if (imp && disp) hipLaunchKernelGGL((k_solve_parts<true, true, true>), g, b, 0, stream_, v, pv, it);
                        else if (imp)    hipLaunchKernelGGL((k_solve_parts<true, false, true>), g, b, 0, stream_, v, pv, it);
                        else             hipLaunchKernelGGL((k_solve_parts<false, true, true>), g, b, 0, stream_, v, pv, it);
                    }
                    ++sweep_launches_;
                }
                c0 = sched_.hbm_interior_classes;
            }
            for (int c = c0; c < ncol; ++c) {
                const int cb = sched_.hbm_colour_offsets[c], ce = sched_.hbm_colour_offsets[c + 1], lead = sched_.hbm_class_leaders[c], foll = ce - cb - lead;
                const dim3 g(std::max(1, std::min(div_up(lead, SOLVE_BLOCK), 8192))), b(SOLVE_BLOCK);
                if (imp && disp) hipLaunchKernelGGL((k_solve_colour<true, true>), g, b, 0, stream_, v, cb, lead, foll, c, it);
                else if (imp)    hipLaunchKernelGGL((k_solve_colour<true, false>), g, b, 0, stream_, v, cb, lead, foll, c, it);
                else             hipLaunchKernelGGL((k_solve_colour<false, true>), g, b, 0, stream_, v, cb, lead, foll, c, it);
                ++sweep_launches_;
            }
        }
    }
    if (forked) PHX_HIP(hipStreamWaitEvent(stream_, ev_join_, 0));
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

int DeviceSolver::enqueue_post(const BodyView& d_bodies, int nb, phx_contact_joint* d_joints, int nj)
{
    const SolverView v = view();
    if (nj && owns_hbm_group()) {      // only the HBM group has results parked in the solver arrays
        const int hb = sched_.hbm_begin(), he = sched_.hbm_end();
        const int hbm_bodies = sched_.hbm_body_count;
        hipLaunchKernelGGL(k_finish_joints, dim3(grid_for(he - hb)), dim3(256), 0, stream_, v, hb, he, d_joints);
        hipLaunchKernelGGL(k_finish_bodies, dim3(grid_for(hbm_bodies)), dim3(256), 0, stream_, v, (const int*)hbm_.hbm_body_list.p, hbm_bodies, d_bodies);
    }
    (void)nb;
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

void DeviceSolver::drop_graphs()
{
    for (hipGraphExec_t& g : graph_) { if (g) (void)hipGraphExecDestroy(g); g = nullptr; }
    graph_key_ = GraphKey{};
}

// Capture the three segments into hipGraphs.  A solve of the 200k-box scene is ~230 launches of 1-4 us
// kernels; launched eagerly the host (~4 us per launch) is the bottleneck, replayed from a graph it is not.
int DeviceSolver::capture_graphs(const GraphKey& key, const BodyView& d_bodies, const phx_contact_point* d_cps, phx_contact_joint* d_joints)
{
    drop_graphs();
    for (int seg = 0; seg < 3; ++seg) {
        hipGraph_t graph = nullptr;
        PHX_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
        int st = seg == 0 ? enqueue_pre(d_bodies, key.nb, d_cps, d_joints, key.nj)
               : seg == 1 ? enqueue_sweeps(d_bodies, d_cps, d_joints, key.nj, key.ci, key.pi)
                          : enqueue_post(d_bodies, key.nb, d_joints, key.nj);
        hipError_t e = hipStreamEndCapture(stream_, &graph);
        if (st != PHX_OK) { if (graph) (void)hipGraphDestroy(graph); drop_graphs(); return st; }
        if (e != hipSuccess) { set_error("hipStreamEndCapture: %s", hipGetErrorString(e)); drop_graphs(); return PHX_ERR_HIP; }
        if (graph) {
            e = hipGraphInstantiate(&graph_[seg], graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            if (e != hipSuccess) { set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); drop_graphs(); return PHX_ERR_HIP; }
        }
    }
    graph_key_ = key;
    graph_sweep_launches_ = sweep_launches_;
    return PHX_OK;
}

int DeviceSolver::enqueue(const Arrays& arrays, int nb, const phx_contact_point* d_cps, phx_contact_joint* d_joints, int nj, const phx_config& cfg)
{
    cur_ = arrays;
    const BodyView& d_bodies = arrays.view;
    if (shard_count_ > 1) PHX_TRY(ensure_partition());       // which groups this rank solves (before anything asks owns_hbm_group())
    const int ci = cfg.contact_iterations, pi = cfg.penetration_iterations;
    const int iters = std::max(ci, pi);
    if (iters + 1 > max_iters_ || !hbm_.flags.p) {
        max_iters_ = std::max(iters + 1, 64);
        PHX_TRY(hbm_.flags.reserve(2 * (size_t)max_iters_));
        PHX_HIP(hipMemsetAsync(hbm_.flags.p, 0, 2 * (size_t)max_iters_ * sizeof(int), stream_));
        drop_graphs();
    }
    GraphKey key;
    key.bodies = d_bodies.vel; key.cps = d_cps; key.joints = d_joints; key.nb = nb; key.nj = nj; key.ncp = ncp_; key.ci = ci; key.pi = pi;
    key.schedule_version = schedule_version_; key.valid = true;
    // graphs pay off from the second solve of an unchanged (schedule, buffers, iteration counts) tuple on
    const bool have = graph_key_.valid && graph_key_ == key;
    if (!have && opt_.use_graphs && last_key_.valid && last_key_ == key) PHX_TRY(capture_graphs(key, d_bodies, d_cps, d_joints));
    last_key_ = key;
    const bool replay = graph_key_.valid && graph_key_ == key;

    // No HIP events around a solve: an event record is a barrier packet of its own and idles the queue for ~5 us.  The device time
    // (phx_solve_stats.device_ms) comes from clock stamps the solve's first and last kernels leave (solve_stamp_begin / _end); only
    // bench() brackets the sweeps with events — the live launch time its roofline is computed from — and device_ms reports those.
    {
        RoctxRange r("PrepareBodies + PrepareJoints + RefreshJoints + PreStepJoints (HBM group)");      // ref: Solver.cpp:70, 135, 146, 157
        if (replay) { if (graph_[0]) PHX_HIP(hipGraphLaunch(graph_[0], stream_)); }
        else PHX_TRY(enqueue_pre(d_bodies, nb, d_cps, d_joints, nj));
    }
    if (time_sweeps_) PHX_HIP(hipEventRecord(ev_sweep_begin_, stream_));
    {
        RoctxRange r("SolveJointIsland: Impulse + Displacement");                                        // ref: Solver.cpp:133, 171, 193
        if (replay) { if (graph_[1]) PHX_HIP(hipGraphLaunch(graph_[1], stream_)); sweep_launches_ = graph_sweep_launches_; }
        else PHX_TRY(enqueue_sweeps(d_bodies, d_cps, d_joints, nj, ci, pi));
    }
    if (time_sweeps_) PHX_HIP(hipEventRecord(ev_sweep_end_, stream_));
    {
        RoctxRange r("FinishJoints + FinishBodies (HBM group)");                                         // ref: Solver.cpp:213, 114
        if (replay) { if (graph_[2]) PHX_HIP(hipGraphLaunch(graph_[2], stream_)); }
        else PHX_TRY(enqueue_post(d_bodies, nb, d_joints, nj));
        // a solve that came through the C-ABI edge: FinishBodies into the caller's records, behind the same gate
        if (arrays.aos && nb) {
            hipLaunchKernelGGL(k_view_to_bodies, dim3(grid_for(nb)), dim3(256), 0, stream_, d_bodies, nb, arrays.aos, (const unsigned long long*)(hash_.p + hash_slot_), gate_expected_);
            PHX_HIP(hipGetLastError());
        }
    }
    timed_sweeps_ = time_sweeps_;
    last_ci_ = ci; last_pi_ = pi; last_island_mode_ = cfg.island_mode;
    stats_pending_ = true;
    have_solve_ = true;
    stats_.graph_replay = replay ? 1 : 0;
    return PHX_OK;
}

// the C-ABI edge: PrepareBodies (ref: Solver.cpp:456-480) of the caller's 128-byte records into this handle's resident arrays
int DeviceSolver::edge_view(const void* d_bodies_aos, int nb, Arrays* out)
{
    PHX_TRY(edge_vel_.reserve(std::max(nb, 1))); PHX_TRY(edge_dvel_.reserve(std::max(nb, 1))); PHX_TRY(edge_mpos_.reserve(std::max(nb, 1)));
    out->view = BodyView{edge_vel_.p, edge_dvel_.p, edge_mpos_.p};
    out->aos = static_cast<phx_rigid_body*>(const_cast<void*>(d_bodies_aos));
    if (nb) hipLaunchKernelGGL(k_bodies_to_view, dim3(grid_for(nb)), dim3(256), 0, stream_, static_cast<const phx_rigid_body*>(d_bodies_aos), nb, out->view);
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

int DeviceSolver::solve_device(void* d_bodies, int nb, const void* d_cps, int ncp, void* d_joints, int nj, const phx_config& cfg, bool topology_changed)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(nb >= 0, "negative count");
    PHX_REQUIRE(nb == 0 || d_bodies, "null bodies");
    PHX_REQUIRE((reinterpret_cast<uintptr_t>(d_bodies) & 15u) == 0, "device arrays must be 16-byte aligned");
    Arrays a;
    a.aos = static_cast<phx_rigid_body*>(d_bodies);
    a.view = BodyView{edge_vel_.p, edge_dvel_.p, edge_mpos_.p};      // (identity of the pending solve: refreshed by edge_view below)
    // an unverified solve on OTHER arrays is still in flight: settle it before the edge arrays are overwritten
    if (pending_.active && pending_.arrays.aos != a.aos) PHX_TRY(synchronize());
    if (build_unverified_) PHX_TRY(synchronize());
    if ((size_t)nb > edge_vel_.cap && pending_.active) PHX_TRY(synchronize());      // (the edge arrays are about to be reallocated)
    PHX_TRY(edge_view(d_bodies, nb, &a));
    return solve_common(a, nb, d_cps, ncp, d_joints, nj, cfg, topology_changed);
}

int DeviceSolver::solve_resident(const BodyView& bodies, int nb, const void* d_cps, int ncp, void* d_joints, int nj, const phx_config& cfg, bool topology_changed)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(nb >= 0, "negative count");
    PHX_REQUIRE(nb == 0 || (bodies.vel && bodies.dvel && bodies.mpos), "null bodies");
    Arrays a;
    a.view = bodies;
    return solve_common(a, nb, d_cps, ncp, d_joints, nj, cfg, topology_changed);
}

bool DeviceSolver::same_as_pending(const Arrays& a, int nb, const void* cps, int ncp, const void* joints, int nj, const phx_config& cfg) const
{
    return pending_.arrays.aos == a.aos && pending_.arrays.view.vel == a.view.vel && pending_.arrays.view.dvel == a.view.dvel && pending_.arrays.view.mpos == a.view.mpos &&
           pending_.cps == cps && pending_.joints == joints && pending_.nb == nb && pending_.nj == nj && pending_.ncp == ncp && std::memcmp(&pending_.cfg, &cfg, sizeof cfg) == 0;
}

void DeviceSolver::register_pending(const Arrays& a, int nb, const void* cps, int ncp, void* joints, int nj, const phx_config& cfg, bool repeat)
{
    pending_.count = repeat && pending_.active ? pending_.count + 1 : 1;
    pending_.active = true; pending_.arrays = a; pending_.cps = cps; pending_.joints = joints;
    pending_.nb = nb; pending_.ncp = ncp; pending_.nj = nj; pending_.cfg = cfg;
    pending_.mode = isl_mode_; pending_.nexpect = isl_nexpect_;
}

int DeviceSolver::solve_common(const Arrays& a, int nb, const void* d_cps, int ncp, void* d_joints, int nj, const phx_config& cfg, bool topology_changed)
{
    PHX_REQUIRE(nb >= 0 && nj >= 0 && ncp >= 0, "negative count");
    PHX_REQUIRE(cfg.contact_iterations >= 0 && cfg.penetration_iterations >= 0 && cfg.contact_iterations < 60000 && cfg.penetration_iterations < 60000, "iteration count out of range");
    PHX_REQUIRE(!half_state_ || (cfg.contact_iterations < 32000 && cfg.penetration_iterations < 32000), "fp16 body state keeps the iteration tag in 16 bits");
    PHX_REQUIRE(cfg.solve_mode >= PHX_SOLVE_SCALAR && cfg.solve_mode <= PHX_SOLVE_AVX2, "unknown solve mode");
    PHX_REQUIRE(cfg.island_mode >= PHX_ISLAND_SINGLE && cfg.island_mode <= PHX_ISLAND_MULTIPLE_SLOPPY, "unknown island mode");
    PHX_REQUIRE(nj == 0 || (d_joints && d_cps), "null joints / contact points");
    PHX_REQUIRE((reinterpret_cast<uintptr_t>(a.view.vel) & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.view.dvel) & 15u) == 0 && (reinterpret_cast<uintptr_t>(a.view.mpos) & 15u) == 0 &&
                (reinterpret_cast<uintptr_t>(d_cps) & 15u) == 0, "device arrays must be 16-byte aligned");
    // an unverified solve on OTHER arrays is still in flight: settle it first (a repeat on the same arrays simply
    // supersedes it — each solve is gated for itself)
    if (pending_.active && !same_as_pending(a, nb, d_cps, ncp, d_joints, nj, cfg)) PHX_TRY(synchronize());
    // ... and so is an unverified solve whose launch checks the schedule ITSELF (ISL_VERIFY): a workgroup that gives up its bounded wait
    // marks the solve's control word, which the NEXT solve's first kernel clears (two control sets alternate) — chained, a timed-out
    // solve would never be completed (complete_partial) and its late groups would silently get one solve fewer.  Hash-gated repeats
    // share one verdict and commit all or nothing: they may chain.
    // (bench()'s timed loop queues its steps back to back by design and settles them at the end: a measurement, not a caller)
    if (pending_.active && pending_.mode == ISL_VERIFY && !in_bench_loop_) PHX_TRY(synchronize());
    // a device-built schedule is verified (did every bin fit?) before anything else runs on it: only the solve that was queued
    // with the build is covered by the spoiled control word
    if (build_unverified_) PHX_TRY(synchronize());
    cur_ = a;
    const float4* mpos = a.view.mpos;
    const phx_contact_joint* joints = static_cast<const phx_contact_joint*>(d_joints);
    const bool want_islands = cfg.island_mode != PHX_ISLAND_SINGLE && !opt_.no_islands;
    if (!reuse_schedule_) topology_changed = true;       // live-topology measurements: rebuild like the reference does every call (ref: Solver.cpp:77, 135)
    bool armed = false;
    if (!topology_changed && opt_.speculate && sched_.valid && nb == nb_ && nj == nj_ && ncp == ncp_ && sched_.islands == want_islands) {
        // Same sizes as the schedule in hand: run on it without a host round trip.  Every kernel that writes to the caller's
        // arrays commits only behind the solve's gate — the island launch's own check of the schedule against the arrays
        // (ISL_VERIFY, island_view.h), or the topology hash pass queued in front (ISL_GATED) — and synchronize() reads the
        // control word back and, if the schedule was stale, rebuilds it and repeats the solve.
        int st = PHX_OK;
        armed = arm_cached_solve(mpos, nb, joints, nj, ncp, &st);
        PHX_TRY(st);
    }
    if (armed) {
        // a repeat on the same arrays while the previous one is still unverified: both ran on the same cached schedule and are
        // gated by the same topology, so they are verified together — and replayed together if the schedule was stale
        // (bench() on staged copies of the input the schedule was verified for: the device gates all the same, bench() checks the
        //  last control word itself, and nothing is registered for a replay — every step has arrays of its own)
        if (!bench_trusted_) register_pending(a, nb, d_cps, ncp, d_joints, nj, cfg, true);
        stats_.recoloured = 0;
    } else {
        PHX_TRY(ensure_schedule(mpos, nb, joints, nj, ncp, cfg, false, topology_changed));
        // like a speculative solve: verified (and replayed on a host-built schedule if a bin was rejected) by synchronize()
        if (build_unverified_) register_pending(a, nb, d_cps, ncp, d_joints, nj, cfg, false);
    }
    const bool split = cfg.island_mode == PHX_ISLAND_MULTIPLE || cfg.island_mode == PHX_ISLAND_MULTIPLE_SLOPPY;
    stats_.island_count = split ? sched_.island_count : 1;
    stats_.island_max_size = split ? sched_.island_max_size : nj;
    stats_.colour_count = sched_.ncolours();
    stats_.lds_islands = sched_.lds_groups;
    if (step_hook_ && step_hook_(step_hook_user_, step_hook_step_, 1)) { set_error("bench: step hook failed"); return PHX_ERR_STATE; }
    return enqueue(a, nb, static_cast<const phx_contact_point*>(d_cps), static_cast<phx_contact_joint*>(d_joints), nj, cfg);
}

int DeviceSolver::solve_host(phx_rigid_body* bodies, int nb, const phx_contact_point* cps, int ncp, phx_contact_joint* joints, int nj, const phx_config& cfg)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(nb >= 0 && nj >= 0 && ncp >= 0, "negative count");
    PHX_REQUIRE(nb == 0 || bodies, "null bodies");
    PHX_REQUIRE(nj == 0 || (joints && cps), "null joints / contact points");
    for (int j = 0; j < nj; ++j)
        if ((unsigned)joints[j].contact_point_index >= (unsigned)ncp) { set_error("joint %d references contact point out of range", j); return PHX_ERR_INVALID; }
    PHX_TRY(st_bodies_.reserve(std::max(nb, 1)));
    PHX_TRY(st_cps_.reserve(std::max(ncp, 1)));
    PHX_TRY(st_joints_.reserve(std::max(nj, 1)));
    if (nb) PHX_HIP(hipMemcpyAsync(st_bodies_.p, bodies, (size_t)nb * sizeof(phx_rigid_body), hipMemcpyHostToDevice, stream_));
    if (ncp) PHX_HIP(hipMemcpyAsync(st_cps_.p, cps, (size_t)ncp * sizeof(phx_contact_point), hipMemcpyHostToDevice, stream_));
    if (nj) PHX_HIP(hipMemcpyAsync(st_joints_.p, joints, (size_t)nj * sizeof(phx_contact_joint), hipMemcpyHostToDevice, stream_));
    PHX_TRY(solve_device(st_bodies_.p, nb, st_cps_.p, ncp, st_joints_.p, nj, cfg));
    PHX_TRY(synchronize());            // settles the speculative run (rebuild + repeat if the topology changed) before anything is read back
    if (nb) PHX_HIP(hipMemcpyAsync(bodies, st_bodies_.p, (size_t)nb * sizeof(phx_rigid_body), hipMemcpyDeviceToHost, stream_));
    if (nj) PHX_HIP(hipMemcpyAsync(joints, st_joints_.p, (size_t)nj * sizeof(phx_contact_joint), hipMemcpyDeviceToHost, stream_));
    PHX_HIP(hipStreamSynchronize(stream_));
    return PHX_OK;
}

// `extra`/`extra_src`: one more 8-byte value to fetch in the same round trip (the fingerprint of a speculative solve)
int DeviceSolver::collect_stats(unsigned long long* extra, const unsigned long long* extra_src)
{
    // an unverified device build (build_schedule_device): the classes per group ride along; whether a bin was rejected shows in
    // the fingerprint the caller compares
    std::vector<int> ncol, goff;
    std::vector<unsigned> comp_size;
    int spec[16] = {0};
    auto with_build = [&]() -> int {
        if (!build_unverified_) return PHX_OK;
        ncol.assign((size_t)unverified_bins_, 0);
        if (spec_bins_pending_) {                      // speculative binning: what the build's round trip would have brought
            goff.assign((size_t)unverified_bins_ + 1, 0);
            comp_size.assign((size_t)std::min(std::min(BINC_MAX, nb_), std::max(1024, ncomp_guess_ + ncomp_guess_ / 4)), 0u);
            PHX_TRY(rb_.add(spec, bld_.bin_result.p, sizeof spec, stream_));
            PHX_TRY(rb_.add(goff.data(), bld_.bin_tables.p + 2 * BINC_MAX, goff.size() * sizeof(int), stream_));
            if (!comp_size.empty()) PHX_TRY(rb_.add(comp_size.data(), bld_.comp_size.p, comp_size.size() * sizeof(unsigned), stream_));
        }
        return rb_.add(ncol.data(), isl_.ncol.p, ncol.size() * sizeof(int), stream_);
    };
    auto rest_of_sizes = [&]() -> int {                // more components than last time (+ 25 %): fetch the rest
        if (!build_unverified_ || !spec_bins_pending_ || spec[4] != 0 || (size_t)spec[5] <= comp_size.size()) return PHX_OK;
        const size_t have = comp_size.size();
        comp_size.resize((size_t)spec[5], 0u);
        PHX_TRY(rb_.add(comp_size.data() + have, bld_.comp_size.p + have, (comp_size.size() - have) * sizeof(unsigned), stream_));
        return rb_.wait(stream_);
    };
    auto settle_build = [&]() {
        if (!build_unverified_) return;
        build_unverified_ = false;
        build_was_unverified_ = true;                  // (read by synchronize() if the fingerprint does not match)
        if (spec_bins_pending_) {
            spec_bins_pending_ = false;
            spec_bins_failed_ = spec[4] != 0;
            if ((opt_.trace_schedule || getenv("PHX_TRACE_SPEC")) && spec_bins_failed_) fprintf(stderr, "[schedule/gpu] speculative binning spoiled: bits %d (%d components, %d bins, grid %d)\n", spec[4], spec[5], spec[6], unverified_bins_);
            if (spec_bins_failed_) { spec_bins_ok_ = false; return; }      // (the fingerprint word is spoiled: synchronize() rebuilds)
            const int nbins = std::min(spec[0], unverified_bins_);
            unsigned long long hash = 0;
            std::memcpy(&hash, spec + 8, sizeof hash);
            raw_fingerprint_ = hash; have_hash_ = spec_hash_ran_;
            sched_.fingerprint = hash ^ ((unsigned long long)(unsigned)nj_ << 32) ^ (unsigned)nb_;
            sched_.lds_groups = nbins;
            sched_.group_offsets.assign(goff.begin(), goff.begin() + nbins + 1);
            {   // GatherIslands' published numbers from the component sizes, as the builder's long way computes them
                const int ncomp = spec[5];
                int run = 0, count = 0, mx = 0;
                for (int c = 0; c < ncomp; ++c) {
                    run += (int)comp_size[c];
                    if (run >= 256 || (run > 0 && c == ncomp - 1)) { ++count; mx = std::max(mx, run); run = 0; }
                }
                sched_.island_count = count; sched_.island_max_size = mx;
            }
            ncol.resize((size_t)nbins);
            spec_bins_guess_ = nbins; ncomp_guess_ = spec[5];
            stats_.lds_islands = nbins;
            const bool split = last_island_mode_ == PHX_ISLAND_MULTIPLE || last_island_mode_ == PHX_ISLAND_MULTIPLE_SLOPPY;
            stats_.island_count = split ? sched_.island_count : 1;
            stats_.island_max_size = split ? sched_.island_max_size : nj_;
        }
        sched_.lds_colours = 0;
        for (int n : ncol) sched_.lds_colours += n;
        stats_.colour_count = sched_.ncolours();
    };
    if (!stats_pending_) {
        if (extra) { PHX_TRY(rb_.add(extra, extra_src, sizeof *extra, stream_)); }
        PHX_TRY(with_build());
        PHX_TRY(rb_.wait(stream_));
        PHX_TRY(rest_of_sizes());
        settle_build();
        return PHX_OK;
    }
    std::vector<int> flags(2 * (size_t)max_iters_);
    int isl_slots[2 * ISL_STAT_SLOTS] = {0};
    unsigned long long visit_slots[ISL_STAT_SLOTS] = {0};
    if (extra) PHX_TRY(rb_.add(extra, extra_src, sizeof *extra, stream_));
    PHX_TRY(rb_.add(flags.data(), hbm_.flags.p, flags.size() * sizeof(int), stream_));
    PHX_TRY(rb_.add(isl_slots, isl_.stats.p + (size_t)hash_slot_ * STATS_SET, sizeof isl_slots, stream_));
    unsigned long long stamps[2] = {0, 0};
    PHX_TRY(rb_.add(visit_slots, isl_.visits.p + (size_t)hash_slot_ * VISITS_SET, sizeof visit_slots, stream_));
    PHX_TRY(rb_.add(stamps, isl_.visits.p + (size_t)hash_slot_ * VISITS_SET + ISL_STAT_SLOTS, sizeof stamps, stream_));
    PHX_TRY(with_build());
    PHX_TRY(rb_.wait(stream_));
    PHX_TRY(rest_of_sizes());
    settle_build();
    int isl[2] = {0, 0};
    unsigned long long isl_visits = 0;
    for (int k = 0; k < ISL_STAT_SLOTS; ++k) { isl[0] = std::max(isl[0], isl_slots[2 * k]); isl[1] = std::max(isl[1], isl_slots[2 * k + 1]); isl_visits += visit_slots[k]; }
    auto executed = [&](const int* active, int limit) {
        int n = 0;
        for (int k = 0; k < limit; ++k) { ++n; if (!active[k]) break; }     // ref: Solver.cpp:175-190
        return n;
    };
    const int hbm_joints = sched_.hbm_end() - sched_.hbm_begin();
    const int h_imp = hbm_joints ? executed(flags.data(), last_ci_) : 0;
    const int h_disp = hbm_joints ? executed(flags.data() + max_iters_, last_pi_) : 0;
    // like the reference's per-island loops, report the longest-running island (ref: Solver.cpp:175-190 per island)
    stats_.impulse_iterations = nj_ ? std::max(h_imp, isl[0]) : std::min(last_ci_, 1);
    stats_.displacement_iterations = nj_ ? std::max(h_disp, isl[1]) : std::min(last_pi_, 1);
    stats_.joint_visits = (long long)isl_visits + (long long)h_imp * hbm_joints;
    float ms = 0.f;
    if (timed_sweeps_) {
        PHX_HIP(hipEventSynchronize(ev_sweep_end_));       // (already reached: the mailbox post ran behind it)
        PHX_HIP(hipEventElapsedTime(&ms, ev_sweep_begin_, ev_sweep_end_));
    } else if (stamps[1] > stamps[0]) ms = (float)((double)(stamps[1] - stamps[0]) * 1e-5);      // 100 MHz ticks -> ms
    stats_.device_ms = ms;
    stats_pending_ = false;
    return PHX_OK;
}

// ISL_COMPLETE: the groups a verified launch left uncommitted (the bounded wait of some workgroup ran out before all had
// arrived; the schedule itself was found correct by every workgroup) are solved by a second launch that skips the committed ones.
int DeviceSolver::complete_partial()
{
    const Pending p = pending_;
    const int ci = p.cfg.contact_iterations, pi = p.cfg.penetration_iterations;
    cur_ = p.arrays;
    island_clears_next_ = false;
    PHX_TRY(enqueue_sweeps(p.arrays.view, static_cast<const phx_contact_point*>(p.cps), static_cast<phx_contact_joint*>(p.joints), p.nj, ci, pi, ISL_COMPLETE));
    if (p.arrays.aos && p.nb) hipLaunchKernelGGL(k_view_to_bodies, dim3(grid_for(p.nb)), dim3(256), 0, stream_, p.arrays.view, p.nb, p.arrays.aos, (const unsigned long long*)nullptr, 0ull);
    PHX_HIP(hipGetLastError());
    unsigned long long word = 0;
    PHX_TRY(rb_.add(&word, hash_.p + hash_slot_, sizeof word, stream_));
    PHX_TRY(rb_.wait(stream_));
    (void)word;       // (ISL_COMPLETE commits unconditionally: behind it every group carries this solve's epoch)
    return PHX_OK;
}

int DeviceSolver::synchronize()
{
    PHX_TRY(use_device(device_));
    if (pending_.active) {
        // one round trip: the speculative solve's control word and its counters together
        unsigned long long fp = 0;
        PHX_TRY(collect_stats(&fp, hash_.p + hash_slot_));
        const Pending p = pending_;
        pending_.active = false;
        const bool spoiled_build = build_was_unverified_ && fp != gate_expected_;      // a bin did not fit: the device build spoiled the control word
        build_was_unverified_ = false;
        if (spoiled_build && !spec_bins_failed_) force_host_builder_ = true;      // (a spoiled speculative binning only needs the builder's long way)
        spec_bins_failed_ = false;
        if (fp != gate_expected_ && p.mode == ISL_VERIFY && (fp & ~ISL_TIMEOUT) == gate_expected_) {
            // every workgroup arrived and found the schedule correct, but some gave up waiting for the others: finish their groups
            // (and stop checking the schedule inside the launch on this handle: whatever delayed them — a GPU shared with somebody
            //  else's kernels — makes every such solve a ~20 ms cliff; the hash-gated form has no wait between workgroups)
            opt_.no_fused_verify = true;
            stats_pending_ = true;
            ++replays_;                                // (callers that queued work behind the solve's gate repeat it)
            pending_ = p;
            const int st = complete_partial();
            pending_.active = false;
            PHX_TRY(st);
        } else if (fp != gate_expected_) {
            stats_pending_ = true;                     // those counters belong to a solve that committed nothing
            ++replays_;
            // the joint topology changed under the cached schedule (or the build it ran on had a bin that did not fit): nothing was
            // committed; rebuild — verified on the spot this time — and solve again
            const bool keep_defer = defer_build_check_;
            defer_build_check_ = false;
            cur_ = p.arrays;
            const int rebuilt = ensure_schedule(p.arrays.view.mpos, p.nb, static_cast<const phx_contact_joint*>(p.joints), p.nj, p.ncp, p.cfg, true);
            defer_build_check_ = keep_defer;
            PHX_TRY(rebuilt);
            stats_.colour_count = sched_.ncolours();
            stats_.lds_islands = sched_.lds_groups;
            const bool split = p.cfg.island_mode == PHX_ISLAND_MULTIPLE || p.cfg.island_mode == PHX_ISLAND_MULTIPLE_SLOPPY;
            stats_.island_count = split ? sched_.island_count : 1;
            stats_.island_max_size = split ? sched_.island_max_size : p.nj;
            // none of the queued solves committed anything: repeat as many as were asked for (e.g. solver-only sub-stepping)
            for (int k = 0; k < std::max(p.count, 1); ++k) {
                if (k) PHX_TRY(launch_fingerprint(p.arrays.view.mpos, p.nb, static_cast<const phx_contact_joint*>(p.joints), p.nj, p.ncp));
                PHX_TRY(enqueue(p.arrays, p.nb, static_cast<const phx_contact_point*>(p.cps), static_cast<phx_contact_joint*>(p.joints), p.nj, p.cfg));
            }
            PHX_HIP(hipStreamSynchronize(stream_));
        }
    }
    return collect_stats();
}

int DeviceSolver::get_stats(phx_solve_stats* out)
{
    PHX_REQUIRE(out, "null out");
    if (!have_solve_) { set_error("no solve has run yet"); return PHX_ERR_STATE; }
    PHX_TRY(synchronize());
    *out = stats_;
    return PHX_OK;
}

int DeviceSolver::get_schedule(int* order, int order_cap, int* offsets, int offsets_cap, int* ncolours)
{
    if (!sched_.valid) { set_error("no solve has run yet"); return PHX_ERR_STATE; }
    PHX_TRY(synchronize());                            // (an unverified device build is settled first)
    PHX_TRY(materialise_schedule());
    const int ncol = sched_.ncolours();
    if (ncolours) *ncolours = ncol;
    if ((order && order_cap < nj_) || (offsets && offsets_cap < ncol + 1)) { set_error("schedule buffers too small"); return PHX_ERR_CAPACITY; }
    if (order) std::copy(sched_.order.begin(), sched_.order.end(), order);
    if (offsets) std::copy(sched_.colour_offsets.begin(), sched_.colour_offsets.end(), offsets);
    return PHX_OK;
}

// phase stamps of the island kernel's workgroups (diagnostics: tools/island_trace.py)
int DeviceSolver::get_island_trace(unsigned long long* out, int cap_groups, int* groups)
{
    PHX_TRY(synchronize());
    const int lg = sched_.valid ? sched_.lds_groups : 0;
    if (groups) *groups = lg;
    if (!out) return PHX_OK;
    if (!trace_islands_ || !isl_.trace.p) { set_error("island trace is off (phx_solver_set_trace)"); return PHX_ERR_STATE; }
    if (cap_groups < lg) { set_error("island trace buffer too small"); return PHX_ERR_CAPACITY; }
    if (lg) PHX_HIP(hipMemcpy(out, isl_.trace.p, (size_t)lg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PHX_OK;
}

int DeviceSolver::get_wave_trace(unsigned long long* out, int cap_words, int* waves_per_group)
{
    PHX_TRY(synchronize());
    const int lg = sched_.valid ? sched_.lds_groups : 0;
    const int wpg = sched_.lds_lanes > ISL_T ? ISL_T_BIG / 64 : ISL_T / 64;
    if (waves_per_group) *waves_per_group = wpg;
    if (!out) return PHX_OK;
    if (!trace_islands_ || !isl_.trace.p) { set_error("island trace is off (phx_solver_set_trace)"); return PHX_ERR_STATE; }
    if (cap_words < lg * wpg * 8) { set_error("wave trace buffer too small"); return PHX_ERR_CAPACITY; }
    if (lg) PHX_HIP(hipMemcpy(out, isl_.trace.p + (size_t)lg * 8, (size_t)lg * wpg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return PHX_OK;
}

int DeviceSolver::set_body_state_bits(int bits)
{
    PHX_REQUIRE(bits == 16 || bits == 32, "body state precision must be 16 or 32 bits");
    if ((bits == 16) != half_state_) { half_state_ = bits == 16; drop_graphs(); }
    return PHX_OK;
}

// Island sharding: every rank builds the same schedule from the same joints and sweeps only the groups
// g with g % count == shard.  Groups are body-disjoint, so the other ranks' bodies and joints are simply left untouched;
// stitching the ranks' results together reproduces the unsharded solve bit for bit.
int DeviceSolver::set_shard(int shard, int count)
{
    PHX_REQUIRE(count >= 1 && shard >= 0 && shard < count, "bad shard");
    if (shard != shard_ || count != shard_count_) {
        PHX_TRY(synchronize());
        // a sharded solve's schedule also carries the groups' body counts for the exchange layout: rebuild on a new count
        if (count != shard_count_) sched_.valid = false;
        shard_ = shard; shard_count_ = count; drop_graphs();
    }
    return PHX_OK;
}

int DeviceSolver::get_groups(int* offsets, int cap, int* count, int* lds_count)
{
    if (!sched_.valid) { set_error("no solve has run yet"); return PHX_ERR_STATE; }
    PHX_TRY(synchronize());
    PHX_TRY(materialise_schedule());
    if (count) *count = sched_.ngroups();
    if (lds_count) *lds_count = sched_.lds_groups;
    if (offsets) {
        if (cap < (int)sched_.group_offsets.size()) { set_error("group_offsets too small"); return PHX_ERR_CAPACITY; }
        std::copy(sched_.group_offsets.begin(), sched_.group_offsets.end(), offsets);
    }
    return PHX_OK;
}

int DeviceSolver::get_refreshed(int joint, float out[30])
{
    if (!have_solve_) { set_error("no solve has run yet"); return PHX_ERR_STATE; }
    PHX_REQUIRE(joint >= 0 && joint < nj_ && out, "joint index out of range");
    PHX_TRY(synchronize());
    PHX_TRY(materialise_schedule());
    const int slot = (int)(std::find(sched_.order.begin(), sched_.order.end(), joint) - sched_.order.begin());
    if (sched_.lds_groups && slot < sched_.group_offsets[sched_.lds_groups]) {
        set_error("joint %d was solved by the island kernel, whose refreshed constants live only in registers; query it under island mode Single", joint);
        return PHX_ERR_STATE;
    }
    float4 a, f, c; int4 k; float2 acc, d;
    PHX_HIP(hipMemcpy(&a, hbm_.q0.p + slot, sizeof a, hipMemcpyDeviceToHost));
    PHX_HIP(hipMemcpy(&f, hbm_.q1.p + slot, sizeof f, hipMemcpyDeviceToHost));
    PHX_HIP(hipMemcpy(&c, hbm_.q2.p + slot, sizeof c, hipMemcpyDeviceToHost));
    PHX_HIP(hipMemcpy(&k, hbm_.q3.p + slot, sizeof k, hipMemcpyDeviceToHost));
    PHX_HIP(hipMemcpy(&acc, hbm_.acc.p + slot, sizeof acc, hipMemcpyDeviceToHost));
    PHX_HIP(hipMemcpy(&d, hbm_.dd.p + slot, sizeof d, hipMemcpyDeviceToHost));
    float ii2; std::memcpy(&ii2, &k.x, 4);
    const float im1 = c.y, ii1 = c.z, im2 = c.w;
    const float nx = a.x, ny = a.y, tx = -ny, ty = nx;
    // expand to ContactLimiterPacked order (ref: Solver.h:7-24): projectors, angular, compMass, compInvMass
    float* o = out;
    *o++ = nx; *o++ = ny; *o++ = -nx; *o++ = -ny; *o++ = a.z; *o++ = a.w;
    *o++ = nx * im1; *o++ = ny * im1; *o++ = (-nx) * im2; *o++ = (-ny) * im2; *o++ = a.z * ii1; *o++ = a.w * ii2; *o++ = c.x;
    *o++ = 0.f; *o++ = f.w; *o++ = d.x; *o++ = d.y;
    *o++ = tx; *o++ = ty; *o++ = -tx; *o++ = -ty; *o++ = f.x; *o++ = f.y;
    *o++ = tx * im1; *o++ = ty * im1; *o++ = (-tx) * im2; *o++ = (-ty) * im2; *o++ = f.x * ii1; *o++ = f.y * ii2; *o++ = f.z;
    (void)acc;
    return PHX_OK;
}

// One private copy of the solver's in/out arrays per timed step — the resident velocities (body_view.h) and the joints — made
// outside the timed region: bench() then solves copy k in step k instead of restoring one working copy in front of every step (copy
// dispatches that are the bench's own scaffolding, not SolveJoints).  The records the caller hands over are converted to the
// resident layout here, before the clock starts: the timed solves run on HBM-resident inputs in the layout the World keeps
// them in.  Consumed by the next bench() call on the same arrays.
int DeviceSolver::bench_stage(const void* d_bodies, int nb, const void* d_joints, int nj, int steps)
{
    PHX_REQUIRE(nb >= 0 && nj >= 0 && steps >= 0 && steps <= 4096, "bad bench_stage arguments");
    PHX_TRY(use_device(device_));
    PHX_TRY(synchronize());
    staged_steps_ = 0;
    const size_t bytes = (size_t)steps * ((size_t)nb * 2 * sizeof(float4) + (size_t)nj * sizeof(phx_contact_joint));
    if (!steps || bytes > (8ull << 30)) return PHX_OK;      // too big to stage: bench() restores in front of every step
    PHX_TRY(stage_vel_.reserve(std::max<size_t>((size_t)steps * nb, 1)));
    PHX_TRY(stage_dvel_.reserve(std::max<size_t>((size_t)steps * nb, 1)));
    PHX_TRY(stage_mpos_.reserve(std::max<size_t>(nb, 1)));
    PHX_TRY(stage_joints_.reserve(std::max<size_t>((size_t)steps * nj, 1)));
    for (int k = 0; k < steps; ++k) {
        if (nb) hipLaunchKernelGGL(k_bodies_to_view, dim3(grid_for(nb)), dim3(256), 0, stream_, static_cast<const phx_rigid_body*>(d_bodies), nb,
                                   BodyView{stage_vel_.p + (size_t)k * nb, stage_dvel_.p + (size_t)k * nb, stage_mpos_.p});
        if (nj) PHX_HIP(hipMemcpyAsync(stage_joints_.p + (size_t)k * nj, d_joints, (size_t)nj * sizeof(phx_contact_joint), hipMemcpyDeviceToDevice, stream_));
    }
    PHX_HIP(hipGetLastError());
    PHX_HIP(hipStreamSynchronize(stream_));
    staged_src_bodies_ = d_bodies; staged_src_joints_ = d_joints; staged_nb_ = nb; staged_nj_ = nj; staged_steps_ = steps;
    return PHX_OK;
}

int DeviceSolver::bench(const void* d_bodies, int nb, const void* d_cps, int ncp, const void* d_joints, int nj,
                        const phx_config& cfg, int warmup, int steps, phx_bench_result* out, phx_step_hook hook, void* user)
{
    PHX_REQUIRE(out && warmup >= 0 && steps >= 0 && steps <= 4096, "bad bench arguments");
    PHX_TRY(use_device(device_));
    PHX_TRY(snap_vel_.reserve(std::max(nb, 1))); PHX_TRY(snap_dvel_.reserve(std::max(nb, 1))); PHX_TRY(snap_mpos_.reserve(std::max(nb, 1)));
    PHX_TRY(snap_joints_.reserve(std::max(nj, 1)));
    std::memset(out, 0, sizeof *out);
    while ((int)bench_events_.size() < 2 * steps + 2) { hipEvent_t e; PHX_HIP(hipEventCreate(&e)); bench_events_.push_back(e); }
    // every step solves the SAME input: its own staged copy (bench_stage, made before the clock started), or — warm-up steps, or
    // nothing staged — a working copy restored from the caller's (untouched) arrays in front of the step
    const bool staged = staged_steps_ >= steps && steps > 0 && staged_src_bodies_ == d_bodies && staged_src_joints_ == d_joints && staged_nb_ == nb && staged_nj_ == nj;
    staged_steps_ = 0;                                       // (consumed: the solves overwrite the copies)
    auto one_step = [&](int k) -> int {
        BodyView b{snap_vel_.p, snap_dvel_.p, snap_mpos_.p}; phx_contact_joint* j = snap_joints_.p;
        if (staged && k >= 0) { b = BodyView{stage_vel_.p + (size_t)k * nb, stage_dvel_.p + (size_t)k * nb, stage_mpos_.p}; j = stage_joints_.p + (size_t)k * nj; }
        else {
            if (nb) hipLaunchKernelGGL(k_bodies_to_view, dim3(grid_for(nb)), dim3(256), 0, stream_, static_cast<const phx_rigid_body*>(d_bodies), nb, b);
            if (nj) PHX_HIP(hipMemcpyAsync(j, d_joints, (size_t)nj * sizeof(phx_contact_joint), hipMemcpyDeviceToDevice, stream_));
        }
        PHX_TRY(solve_resident(b, nb, d_cps, ncp, j, nj, cfg));
        if (xch_send_) {       // island-sharded solve: pack, the caller's all-gather (hook phase 2), unpack — all on the stream
            PHX_TRY(exchange_pack_resident(&b, j, nullptr));
            if (comm_) PHX_TRY(exchange_all_gather());      // native transport: RCCL on this stream, no callback
            else if (hook && hook(user, step_hook_step_, 2)) { set_error("bench: step hook failed"); return PHX_ERR_STATE; }
            PHX_TRY(exchange_unpack_resident(b, j));
        }
        return PHX_OK;
    };
    // the hook's phase 1 fires inside solve_device, between the step's fingerprint and its sweeps
    struct HookScope {
        DeviceSolver& s;
        HookScope(DeviceSolver& s_, phx_step_hook h, void* u) : s(s_) { s.step_hook_ = h; s.step_hook_user_ = u; }
        ~HookScope() { s.step_hook_ = nullptr; s.step_hook_user_ = nullptr; }
    } scope(*this, hook, user);
    for (int i = 0; i < warmup; ++i) {
        step_hook_step_ = i - warmup;
        PHX_TRY(one_step(-1));
        if (hook && hook(user, i - warmup, 0)) { set_error("bench: step hook failed"); return PHX_ERR_STATE; }
        PHX_TRY(synchronize());
    }
    if (!steps) { if (hook && warmup && hook(user, 0, 1)) { set_error("bench: step hook failed"); return PHX_ERR_STATE; } return PHX_OK; }
    // timed steps are queued back to back; the device never waits for the host between them
    // (the handle's own sweep events are put back — and the bracketing returned to whole solves — however this function leaves)
    struct SweepEvents {
        DeviceSolver& s; hipEvent_t b, e;
        explicit SweepEvents(DeviceSolver& s_) : s(s_), b(s_.ev_sweep_begin_), e(s_.ev_sweep_end_) { s.time_sweeps_ = true; }
        ~SweepEvents() { s.ev_sweep_begin_ = b; s.ev_sweep_end_ = e; s.time_sweeps_ = false; s.timed_sweeps_ = false; }
    } sweep_events(*this);
    int st = PHX_OK;
    struct InLoop { DeviceSolver& s; explicit InLoop(DeviceSolver& s_) : s(s_) { s.in_bench_loop_ = true; } ~InLoop() { s.in_bench_loop_ = false; } } in_loop(*this);
    struct Trusted { DeviceSolver& s; Trusted(DeviceSolver& s_, bool on) : s(s_) { s.bench_trusted_ = on; } ~Trusted() { s.bench_trusted_ = false; } } trusted(*this, staged && reuse_schedule_ && opt_.speculate);
    PHX_HIP(hipEventRecord(bench_events_[2 * steps], stream_));
    // HIP events bracket the sweep launches of every 4th step only: an event record is a barrier packet of its own (~3 us of idle
    // queue), and bracketing every step's sweeps cost 6.4 us per step — 7 % of the value being measured (tools/exp_events.py)
    // (PHX_BENCH_BRACKET_STRIDE=n, measurement of the measurement: 1 = every step, 0 = no events at all — tools/exp_events.py)
    const char* bs_env = getenv("PHX_BENCH_BRACKET_STRIDE");
    const int bracket_stride = bs_env ? (atoi(bs_env) > 0 ? atoi(bs_env) : steps + 1) : (steps >= 8 ? 4 : 1);
    const bool bracket_any = !(bs_env && atoi(bs_env) <= 0);
    for (int i = 0; i < steps && st == PHX_OK; ++i) {
        time_sweeps_ = bracket_any && i % bracket_stride == 0;
        if (time_sweeps_) { ev_sweep_begin_ = bench_events_[2 * i]; ev_sweep_end_ = bench_events_[2 * i + 1]; }      // (else: still the last bracketed step's pair — a
                                                                                                                  //  solve settled later reads the pair it recorded)
        step_hook_step_ = i;
        st = one_step(i);
        if (st == PHX_OK && hook && hook(user, i, 0)) { set_error("bench: step hook failed"); st = PHX_ERR_STATE; }
    }
    if (st == PHX_OK && hook && hook(user, steps, 1)) { set_error("bench: step hook failed"); st = PHX_ERR_STATE; }      // drain the last exchange
    PHX_TRY(st);
    PHX_HIP(hipEventRecord(bench_events_[2 * steps + 1], stream_));
    const unsigned replays = replays_;
    if (bench_trusted_ && !pending_.active) {       // staged copies: the last step's fingerprint comes back with its counters
        unsigned long long fp = 0;
        PHX_TRY(collect_stats(&fp, hash_.p + hash_slot_));
        if (fp != gate_expected_) { set_error("bench: a staged copy of the input did not match the schedule's topology"); return PHX_ERR_STATE; }
    } else PHX_TRY(synchronize());
    if (replays_ != replays && reuse_schedule_) { set_error("bench: topology changed during the timed region"); return PHX_ERR_STATE; }
    float ms = 0.f;
    PHX_HIP(hipEventSynchronize(bench_events_[2 * steps + 1]));      // (reached long ago — the mailbox post ran behind it — but the runtime may not have looked yet)
    PHX_HIP(hipEventElapsedTime(&ms, bench_events_[2 * steps], bench_events_[2 * steps + 1]));
    out->total_ms = ms;
    for (int i = 0; bracket_any && i < steps; i += bracket_stride) {
        PHX_HIP(hipEventElapsedTime(&ms, bench_events_[2 * i], bench_events_[2 * i + 1]));
        out->impulse_kernel_ms += ms;
    }
    // identical input every step => identical counters every step
    out->impulse_launches = (long long)sweep_launches_ * steps;
    out->bracketed_launches = bracket_any ? (long long)sweep_launches_ * ((steps + bracket_stride - 1) / bracket_stride) : 0;
    out->impulse_iterations = (long long)stats_.impulse_iterations * steps;
    out->joint_visits = stats_.joint_visits * steps;
    return PHX_OK;
}

} // namespace phx
