// solver.hip — DeviceSolver: residency, the launch sequence of a solve and its settling (schedule construction: solver_build.hip).
#include "solver.h"
#include "solver_kernels.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <numeric>

namespace phx {

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
    if (ev_pre_fork_) (void)hipEventDestroy(ev_pre_fork_);
    if (ev_pre_join_) (void)hipEventDestroy(ev_pre_join_);
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
    o.no_jp_walk_one = on("PHX_NO_JP_WALK_ONE");
    o.no_tail = on("PHX_NO_TAIL");                        // the trailing tiny classes of the HBM group one launch each (A/B, tests)
    o.no_parts = on("PHX_NO_PARTS");                      // the interior classes of partitioned components one launch each (A/B, tests)
    o.no_fused_verify = on("PHX_NO_FUSED_VERIFY");        // the topology hash pass in front of every solve on a cached schedule
    const char* wp = getenv("PHX_ISL_WAIT_POLLS");        // tests: 0 makes every workgroup of a verified launch give up, so that ISL_COMPLETE runs
    o.isl_wait_polls = wp ? std::max(0, atoi(wp)) : ISL_WAIT_POLLS;
    o.use_graphs = on("PHX_GRAPHS");                      // replay the launch sequence from hipGraphs (measured slower: off by default)
    const char* sb = getenv("PHX_SCHEDULE_BUILDER");      // "host" forces the host builder
    o.gpu_builder = !(sb && sb[0] == 'h');
    o.speculate = !on("PHX_NO_SPECULATION");
    o.no_islands = on("PHX_NO_ISLANDS");                  // ignore island modes, always the HBM colour path
    o.no_prelabel = on("PHX_NO_PRELABEL");
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
    PHX_HIP(hipEventCreateWithFlags(&ev_pre_fork_, hipEventDisableTiming));
    PHX_HIP(hipEventCreateWithFlags(&ev_pre_join_, hipEventDisableTiming));
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

// The launch sequence of one SolveJoints, in three capturable segments (no sync, no allocation inside):
//   pre    PrepareBodies, PrepareJoints+RefreshJoints, PreStepJoints class by class
//   sweeps `iters` x colours fused impulse+displacement launches
//   post   FinishJoints, FinishBodies
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

// First class of the HBM group's trailing run of classes of at most TAIL_T units each (at most TAIL_CLASSES_MAX of them), not before class
// `from`; the class count if the run is shorter than two classes (one class is as well off in its own launch) or the class table is not
// on the device.
int DeviceSolver::tail_first_class(int from) const
{
    const int ncol = (int)sched_.hbm_class_leaders.size();
    if (opt_.no_tail || !class_tab_ok_) return ncol;
    int t = ncol;
    while (t > from && sched_.hbm_class_leaders[t - 1] <= TAIL_T && ncol - (t - 1) <= TAIL_CLASSES_MAX) --t;
    return ncol - t >= 2 ? t : ncol;
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
            iv.wave_trace = trace_waves_ ? isl_.trace.p + (size_t)lg * 8 : nullptr;
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
                    if (level == 0) {      // a part's ~1000 units, class by class, the next class's constants requested a class ahead (solver_kernels.h)
                        if (imp && disp) hipLaunchKernelGGL((k_solve_parts_ahead<true, true>), g, b, 0, stream_, v, pv, it);
                        else if (imp)    hipLaunchKernelGGL((k_solve_parts_ahead<true, false>), g, b, 0, stream_, v, pv, it);
                        else             hipLaunchKernelGGL((k_solve_parts_ahead<false, true>), g, b, 0, stream_, v, pv, it);
                    } else {           // a level-1 part has a few dozen units: a lane owns one, everything requested up front
                        if (imp && disp) hipLaunchKernelGGL((k_solve_parts<true, true, true>), g, b, 0, stream_, v, pv, it);
                        else if (imp)    hipLaunchKernelGGL((k_solve_parts<true, false, true>), g, b, 0, stream_, v, pv, it);
                        else             hipLaunchKernelGGL((k_solve_parts<false, true, true>), g, b, 0, stream_, v, pv, it);
                    }
                    ++sweep_launches_;
                }
                c0 = sched_.hbm_interior_classes;
            }
            const int tail = tail_first_class(c0);             // the trailing run of tiny classes: one launch, one workgroup (k_solve_tail)
            for (int c = c0; c < tail; ++c) {
                const int cb = sched_.hbm_colour_offsets[c], ce = sched_.hbm_colour_offsets[c + 1], lead = sched_.hbm_class_leaders[c], foll = ce - cb - lead;
                const dim3 g(std::max(1, std::min(div_up(lead, SOLVE_BLOCK), 8192))), b(SOLVE_BLOCK);
                if (imp && disp) hipLaunchKernelGGL((k_solve_colour<true, true>), g, b, 0, stream_, v, cb, lead, foll, c, it);
                else if (imp)    hipLaunchKernelGGL((k_solve_colour<true, false>), g, b, 0, stream_, v, cb, lead, foll, c, it);
                else             hipLaunchKernelGGL((k_solve_colour<false, true>), g, b, 0, stream_, v, cb, lead, foll, c, it);
                ++sweep_launches_;
            }
            if (tail < ncol) {
                const int4* tab = parts_.class_tab.p;
                if (imp && disp) hipLaunchKernelGGL((k_solve_tail<true, true>), dim3(1), dim3(TAIL_T), 0, stream_, v, tab, tail, ncol - tail, it);
                else if (imp)    hipLaunchKernelGGL((k_solve_tail<true, false>), dim3(1), dim3(TAIL_T), 0, stream_, v, tab, tail, ncol - tail, it);
                else             hipLaunchKernelGGL((k_solve_tail<false, true>), dim3(1), dim3(TAIL_T), 0, stream_, v, tab, tail, ncol - tail, it);
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
        build_cps_ = static_cast<const phx_contact_point*>(d_cps);
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
int DeviceSolver::collect_stats(unsigned long long* extra, const unsigned long long* extra_src, const std::function<int()>* while_waiting, const MailCarrier* carrier)
{
    // an unverified device build (build_schedule_device): the classes per group ride along; whether a bin was rejected shows in
    // the fingerprint the caller compares
    // (a SPECULATIVE build brings only its 64 bytes of results here; its tables — first slots and classes of every bin, component
    //  sizes: 16 KB at cfg 2, 100 KB at 1M boxes — are what the query API and the statistics want, not the next step: they are
    //  fetched when somebody asks, fetch_build_tables())
    std::vector<int> ncol;
    int spec[16] = {0};
    auto with_build = [&]() -> int {
        if (!build_unverified_) return PHX_OK;
        if (spec_bins_pending_) return rb_.add(spec, bld_.bin_result.p, sizeof spec, stream_);
        ncol.assign((size_t)unverified_bins_, 0);
        return rb_.add(ncol.data(), isl_.ncol.p, ncol.size() * sizeof(int), stream_);
    };
    auto rest_of_sizes = [&]() -> int { return PHX_OK; };
    auto settle_build = [&]() {
        if (!build_unverified_) return;
        build_unverified_ = false;
        build_was_unverified_ = true;                  // (read by synchronize() if the fingerprint does not match)
        if (spec_bins_pending_) {
            spec_bins_pending_ = false;
            spec_bins_failed_ = spec[4] != 0 || spec[7] != 0;      // (k_bin_components' / k_joint_scatter's fail bits)
            if ((opt_.trace_schedule || getenv("PHX_TRACE_SPEC")) && spec_bins_failed_) fprintf(stderr, "[schedule/gpu] speculative binning spoiled: bits %d, deal bits %d (%d components, %d bins, grid %d)\n", spec[4], spec[7], spec[5], spec[6], unverified_bins_);
            if (spec_bins_failed_) { spec_bins_ok_ = false; return; }      // (the fingerprint word is spoiled: synchronize() rebuilds — and recomputes the components)
            const int nbins = std::min(spec[0], unverified_bins_);
            unsigned long long hash = 0;
            std::memcpy(&hash, spec + 8, sizeof hash);
            raw_fingerprint_ = hash; have_hash_ = spec_hash_ran_;
            sched_.fingerprint = hash ^ ((unsigned long long)(unsigned)nj_ << 32) ^ (unsigned)nb_;
            sched_.lds_groups = nbins;
            spec_bins_guess_ = nbins; ncomp_guess_ = spec[5];
            stats_.lds_islands = nbins;
            tables_pending_ = true; tables_bins_ = nbins; tables_comps_ = std::min(spec[5], std::min(BINC_MAX, nb_)); tables_set_ = bld_.cur;
            return;
        }
        sched_.lds_colours = 0;
        for (int n : ncol) sched_.lds_colours += n;
        stats_.colour_count = sched_.ncolours();
    };
    if (!stats_pending_) {
        if (extra) { PHX_TRY(rb_.add(extra, extra_src, sizeof *extra, stream_)); }
        PHX_TRY(with_build());
        PHX_TRY(rb_.wait(stream_, nullptr, while_waiting, carrier));
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
    PHX_TRY(rb_.wait(stream_, nullptr, while_waiting, carrier));
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

int DeviceSolver::synchronize(const std::function<int()>* while_waiting, const MailCarrier* carrier)
{
    PHX_TRY(use_device(device_));
    if (pending_.active) {
        // one round trip: the speculative solve's control word and its counters together
        unsigned long long fp = 0;
        PHX_TRY(collect_stats(&fp, hash_.p + hash_slot_, while_waiting, carrier));
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
            build_cps_ = static_cast<const phx_contact_point*>(p.cps);
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

// The tables of the last speculative build, on demand (collect_stats): what a build with a host round trip would have left in
// sched_ and stats_.  Callers have settled the solve (synchronize()), so nothing queued since can have overwritten them.
int DeviceSolver::fetch_build_tables()
{
    if (!tables_pending_) return PHX_OK;
    PHX_TRY(use_device(device_));
    const int nbins = tables_bins_, ncomp = tables_comps_;
    std::vector<int> goff((size_t)nbins + 1, 0), ncol((size_t)std::max(nbins, 1), 0);
    std::vector<unsigned> comp_size((size_t)std::max(ncomp, 1), 0u);
    PHX_TRY(rb_.add(goff.data(), bld_.bin_tables_s[tables_set_].p + 2 * BINC_MAX, goff.size() * sizeof(int), stream_));
    if (nbins) PHX_TRY(rb_.add(ncol.data(), isl_.ncol.p, (size_t)nbins * sizeof(int), stream_));
    if (ncomp) PHX_TRY(rb_.add(comp_size.data(), bld_.comp_size_s[tables_set_].p, (size_t)ncomp * sizeof(unsigned), stream_));
    PHX_TRY(rb_.wait(stream_));
    tables_pending_ = false;
    sched_.group_offsets.assign(goff.begin(), goff.end());
    {   // GatherIslands' published numbers from the component sizes, as the builder's long way computes them
        int run = 0, count = 0, mx = 0;
        for (int c = 0; c < ncomp; ++c) {
            run += (int)comp_size[c];
            if (run >= 256 || (run > 0 && c == ncomp - 1)) { ++count; mx = std::max(mx, run); run = 0; }
        }
        sched_.island_count = count; sched_.island_max_size = mx;
    }
    sched_.lds_colours = 0;
    for (int g = 0; g < nbins; ++g) sched_.lds_colours += ncol[(size_t)g];
    const bool split = last_island_mode_ == PHX_ISLAND_MULTIPLE || last_island_mode_ == PHX_ISLAND_MULTIPLE_SLOPPY;
    stats_.island_count = split ? sched_.island_count : 1;
    stats_.island_max_size = split ? sched_.island_max_size : nj_;
    stats_.colour_count = sched_.ncolours();
    return PHX_OK;
}

int DeviceSolver::get_stats(phx_solve_stats* out)
{
    PHX_REQUIRE(out, "null out");
    if (!have_solve_) { set_error("no solve has run yet"); return PHX_ERR_STATE; }
    PHX_TRY(synchronize());
    PHX_TRY(fetch_build_tables());
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

int DeviceSolver::get_lanes(int* leader_slot, int* lane, int cap, int* count)
{
    if (!sched_.valid) { set_error("no solve has run yet"); return PHX_ERR_STATE; }
    PHX_TRY(synchronize());
    PHX_TRY(materialise_schedule());
    const int lg = sched_.lds_groups, lanes = sched_.lds_lanes;
    std::vector<int4> recs(2 * (size_t)std::max(lg, 1) * std::max(lanes, 1));
    PHX_TRY(use_device(device_));
    if (lg) PHX_HIP(hipMemcpy(recs.data(), isl_.unit_recs.p, 2 * (size_t)lg * lanes * sizeof(int4), hipMemcpyDeviceToHost));
    int n = 0;
    for (int g = 0; g < lg; ++g)
        for (int l = 0; l < lanes; ++l) {
            const int4 a = recs[2 * ((size_t)g * lanes + l)], b = recs[2 * ((size_t)g * lanes + l) + 1];
            if (a.x < 0) continue;                                      // nobody's lane
            if (leader_slot && lane) {
                if (n >= cap) { set_error("lane arrays too small"); return PHX_ERR_CAPACITY; }
                leader_slot[n] = b.z; lane[n] = l;
            }
            ++n;
        }
    if (count) *count = n;
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
        bench_last_b_ = b; bench_last_j_ = j; bench_last_nb_ = nb; bench_last_nj_ = nj;
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

// position-sensitive sum of the result words (order of the threads does not matter: a sum of mixed (index, bits) terms)
static __global__ void __launch_bounds__(256) k_result_checksum(BodyView b, int nb, const phx_contact_joint* __restrict__ joints, int nj, unsigned long long* out)
{
    unsigned long long h = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
        const float4 v = b.vel[i], d = b.dvel[i];
        const unsigned w[6] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z)};
        for (int k = 0; k < 6; ++k) h += mix64(((unsigned long long)(6u * (unsigned)i + k + 1u) << 32) | w[k]);
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nj; i += gridDim.x * blockDim.x) {
        const phx_contact_joint j = joints[i];
        h += mix64(0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1) + __float_as_uint(j.normal_accumulated_impulse));
        h += mix64(0xC2B2AE3D27D4EB4Full * (unsigned long long)(i + 1) + __float_as_uint(j.friction_accumulated_impulse));
    }
    for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off);
    if ((threadIdx.x & 63) == 0 && h) atomicAdd(out, h);
}

int DeviceSolver::bench_checksum(unsigned long long* out)
{
    PHX_REQUIRE(out, "null output");
    PHX_TRY(use_device(device_));
    PHX_TRY(synchronize());
    if (!bench_last_j_ && !bench_last_b_.vel) { set_error("bench_checksum: no bench step has run"); return PHX_ERR_STATE; }
    unsigned long long* d = nullptr;
    PHX_HIP(hipMalloc(reinterpret_cast<void**>(&d), sizeof *d));
    hipError_t e = hipMemsetAsync(d, 0, sizeof *d, stream_);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_result_checksum, dim3(512), dim3(256), 0, stream_, bench_last_b_, bench_last_nb_, bench_last_j_, bench_last_nj_, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, sizeof *out, hipMemcpyDeviceToHost, stream_);
    if (e == hipSuccess) e = hipStreamSynchronize(stream_);
    (void)hipFree(d);
    if (e != hipSuccess) { set_error("bench_checksum: %s", hipGetErrorString(e)); return PHX_ERR_HIP; }
    return PHX_OK;
}

} // namespace phx
