// solver_build.hip — DeviceSolver's schedule side: deciding whether the cached schedule still holds, and building a new one
// (GatherIslands + PrepareIndices of ref: Solver.cpp:77, 135, 217, 285) with the host builder (schedule.hip, the specification)
// or on the device (schedule_kernels.h).  The launch sequence of a solve and its settling live in solver.hip.
#include "solver.h"
#include "island_view.h"
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

int DeviceSolver::ensure_schedule(const float4* d_bodies, int nb, const phx_contact_joint* d_joints, int nj, int ncp, const phx_config& cfg, bool force_rebuild,
                                  bool known_changed)
{
    // 1. fingerprint of the joint topology (8 bytes over PCIe).  A caller that KNOWS the topology changed (the World, when
    //    joints were created or destroyed this step) does not wait for it: the value rides along with the builder's first
    //    readback — and a rebuild that needs no host round trip (build_bins_speculative) skips the hash pass altogether: its
    //    solve is gated by the constant k_bin_components leaves, and a later solve on unchanged joints either checks the schedule
    //    inside the island launch (ISL_VERIFY) or, where that does not apply (more groups than are resident at once: the 1M-box
    //    scene), finds no hash on record and rebuilds ONCE with one — instead of every changing step paying the pass (18 us there).
    unsigned long long fp = 0;
    bool have_fp = false;
    // Single = one coupled system swept class by class out of HBM; every other island mode lets the schedule
    // exploit body-disjoint islands (groups solved out of LDS)
    const bool want_islands = cfg.island_mode != PHX_ISLAND_SINGLE && !opt_.no_islands;
    const bool device_builder = opt_.gpu_builder && !force_host_builder_;
    if (prelabel_pending_ && !(known_changed && device_builder)) PHX_TRY(cancel_prelabel());      // (nobody will pick the side stream's bins up: they must not survive into a later step)
    const bool no_hash = known_changed && device_builder && spec_build_applies(want_islands, nj);
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

    tables_pending_ = false;      // (a rebuild: whatever the last speculative build left unfetched on the device is about to be overwritten)
    class_tab_ok_ = false;        // (... and the HBM group's class table belongs to the schedule that is being replaced)
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
        if (opt_.trace_schedule || getenv("PHX_TRACE_SPEC")) fprintf(stderr, "[schedule] the device builder handed %d joints to the host builder\n", nj);
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
        std::vector<int4> unit_recs(2 * (size_t)ng * lanes, make_int4(-1, -1, 0, 0));      // (leader -1: nobody's lane)
        for (int g = 0; g < ng; ++g) {
            const int nunits = sched_.group_unit_offsets[g + 1] - sched_.group_unit_offsets[g];
            int nstatic_g = 0;                                   // (the group's static bodies sit first in its table)
            for (int k = sched_.group_body_offsets[g]; k < sched_.group_body_offsets[g + 1] && is_static[sched_.group_bodies[k]]; ++k) ++nstatic_g;
            units[g] = island_units_word(nunits, ncol[g], nstatic_g);
            for (int u = 0; u < nunits; ++u) {
                const int at = sched_.group_unit_offsets[g] + u;
                const int ls = sched_.unit_leader[at], fs = sched_.unit_follower[at];
                const int lj = sched_.order[ls], fj = fs >= 0 ? sched_.order[fs] : -1;
                const size_t lane = (size_t)g * lanes + sched_.unit_lane[at];      // (schedule.h LANES)
                unit_recs[2 * lane] = make_int4(lj, fj, prio_id[lj], fj >= 0 ? prio_id[fj] : 0);
                unit_recs[2 * lane + 1] = make_int4((int)sched_.slot_local[ls], (int)sched_.slot_colour[ls], ls, fs);
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

// the launch grid of a speculative build's per-bin kernels: last build's bin count with slack
// (workgroups beyond the real bin count leave at once, and a settling world doubles its bins within a few steps — columns
//  break in two: a roomy grid costs nothing, a grid too small costs a repeated solve)
int DeviceSolver::spec_grid() const { return 2 * spec_bins_guess_ + 64; }

int DeviceSolver::prelabel_mark()
{
    prelabel_marked_ = false;
    if (prelabel_pending_) PHX_TRY(cancel_prelabel());          // (a mark without a rebuild in between: the old side work is void)
    if (opt_.no_prelabel || !opt_.gpu_builder || !side_stream_) return PHX_OK;
    // only where the next rebuild would use them: the path without a host round trip (spec_build_applies), on an island-mode schedule
    if (!sched_.valid || !sched_.islands || sched_.has_hbm_group() || !spec_build_applies(true, 1)) return PHX_OK;
    PHX_TRY(use_device(device_));
    PHX_HIP(hipEventRecord(ev_pre_fork_, stream_));          // behind everything that wrote the manifolds
    prelabel_marked_ = true;
    return PHX_OK;
}

int DeviceSolver::prelabel_components(const float4* d_mpos, int nb, const phx_manifold* d_manifolds, int nm)
{
    if (!prelabel_marked_) return PHX_OK;
    prelabel_marked_ = false;
    if (nb <= 0 || nm <= 0 || !d_mpos || !d_manifolds) return PHX_OK;
    PHX_TRY(use_device(device_));
    const int nbs = std::max(nb, 1);
    const int set = bld_.cur ^ 1;                            // (the set the last build did not use: solver.h)
    const int grid = spec_grid();
    PHX_TRY(bld_.cc_parent.reserve(nbs)); PHX_TRY(bld_.cc_static.reserve(nbs)); PHX_TRY(bld_.cc_flags.reserve(nbs + 1)); PHX_TRY(bld_.comp_size_s[set].reserve(nbs + 1));
    PHX_TRY(bld_.comp_units_s[set].reserve(nbs + 1)); PHX_TRY(bld_.sb_small.reserve(8)); PHX_TRY(bld_.side_flags.reserve(4));
    PHX_TRY(bld_.bin_tables_s[set].reserve(2 * (size_t)BINC_MAX + (size_t)grid + 2));
    PHX_TRY(bld_.bin_result.reserve(16)); PHX_TRY(bld_.bin_cursor.reserve((size_t)grid + 2));
    if (!bld_.bin_scratch.p) {
        PHX_TRY(bld_.bin_scratch.reserve(2 * (size_t)BINC_T + 2));
        PHX_HIP(hipMemsetAsync(bld_.bin_scratch.p, 0, bld_.bin_scratch.cap * sizeof(unsigned long long), stream_));
        PHX_HIP(hipEventRecord(ev_pre_fork_, stream_));      // (the side stream must see the cleared scratch)
    }
    RoctxRange range("GatherIslands (components, counts and bins from the manifolds, side stream)");
    PHX_HIP(hipStreamWaitEvent(side_stream_, ev_pre_fork_, 0));
    hipLaunchKernelGGL(k_cc_init_bodies, dim3(grid_for(nb)), dim3(256), 0, side_stream_, d_mpos, nb, bld_.cc_parent.p, bld_.cc_static.p, bld_.side_flags.p);
    hipLaunchKernelGGL(k_cc_link_manifolds, dim3(grid_for(nm)), dim3(256), 0, side_stream_, d_manifolds, nm, nb, bld_.cc_parent.p, (const unsigned char*)bld_.cc_static.p);
    hipLaunchKernelGGL(k_cc_compress_window, dim3(std::max(1, div_up(nb, CCW_BODIES))), dim3(CCW_T), 0, side_stream_, bld_.cc_parent.p, nb);
    hipLaunchKernelGGL(k_cc_compress, dim3(grid_for(nb)), dim3(256), 0, side_stream_, bld_.cc_parent.p, nb, bld_.sb_small.p);
    PHX_TRY(device_exclusive_scan_of(RootFlagLoad{(const int*)bld_.cc_parent.p, nb, bld_.comp_size_s[set].p, bld_.comp_units_s[set].p}, bld_.cc_flags.p, nb + 1,
                                     reinterpret_cast<unsigned*>(bld_.sb_small.p + 1), bld_.prelabel_scan, side_stream_));
    hipLaunchKernelGGL(k_manifold_components, dim3(std::max(1, std::min(div_up(nm, JC_T), 1024))), dim3(JC_T), 0, side_stream_, d_manifolds, nm, nb, (const int*)bld_.cc_parent.p,
                       (const unsigned*)bld_.cc_flags.p, bld_.comp_size_s[set].p, bld_.comp_units_s[set].p, bld_.side_flags.p);
    // the bins: k_bin_components, as the rebuild would launch it — but the solve's gate is armed later, by k_joint_scatter (the control word belongs to the main stream)
    BinCompView cv{};
    cv.comp_size = bld_.comp_size_s[set].p; cv.comp_units = bld_.comp_units_s[set].p; cv.cc_small = bld_.sb_small.p; cv.nj = -1;
    cv.cap_units = spec_lanes_; cv.small_units = ISL_T; cv.max_bins = grid;
    cv.bin_of = bld_.bin_tables_s[set].p; cv.rank_of = bld_.bin_tables_s[set].p + BINC_MAX; cv.goff = bld_.bin_tables_s[set].p + 2 * BINC_MAX;
    cv.result = bld_.bin_result.p; cv.cursor = bld_.bin_cursor.p;
    cv.fingerprint = nullptr; cv.hash_out = nullptr; cv.gate = 0;
    cv.scratch = bld_.bin_scratch.p;
    const int bin_chunks = div_up(ncomp_guess_ + ncomp_guess_ / 4, BIN_CHUNK);
    const int bin_groups = bin_chunks <= 3 * (BINC_T / 64) ? 1 : std::min(16, div_up(bin_chunks, BINC_T / 64));
    hipLaunchKernelGGL(k_bin_components, dim3(bin_groups), dim3(BINC_T), 0, side_stream_, cv);
    // ... and the units dealt to them: a manifold with contact points is a unit (k_build_bin reads its joints through the contact points)
    PHX_TRY(bld_.unit_m.reserve((size_t)2 * nm + 2));
    ManifoldSlotsView mv{};
    mv.manifolds = d_manifolds; mv.nm = nm; mv.nb = nb; mv.parent = bld_.cc_parent.p; mv.root_number = bld_.cc_flags.p;
    mv.bin_of = cv.bin_of; mv.rank_of = cv.rank_of; mv.goff = cv.goff; mv.result = bld_.bin_result.p; mv.max_bins = grid;
    mv.cursor = bld_.bin_cursor.p; mv.unit_m = bld_.unit_m.p; mv.flags = bld_.side_flags.p;
    hipLaunchKernelGGL(k_manifold_slots, dim3(std::max(1, std::min(div_up(nm, 256), 2048))), dim3(256), 0, side_stream_, mv);
    PHX_HIP(hipGetLastError());
    PHX_HIP(hipEventRecord(ev_pre_join_, side_stream_));
    prelabel_pending_ = true; prelabel_nb_ = nb;
    pre_manifolds_ = d_manifolds; pre_nm_ = nm; pre_grid_ = grid; pre_lanes_ = spec_lanes_;
    return PHX_OK;
}

int DeviceSolver::cancel_prelabel()
{
    prelabel_marked_ = false;
    if (!prelabel_pending_) return PHX_OK;
    prelabel_pending_ = false;
    PHX_HIP(hipStreamWaitEvent(stream_, ev_pre_join_, 0));   // (whatever the stream does to the builder's arrays next comes behind them)
    return PHX_OK;
}

int DeviceSolver::build_schedule_device(const float4* d_bodies, int nb, const phx_contact_joint* d_joints, int nj, bool want_islands, bool* fallback)
{
    bool from_manifolds = false;
    if (prelabel_pending_) {                                 // the components and the bins are on their way (prelabel_components): wait for them and use them
        prelabel_pending_ = false;
        PHX_HIP(hipStreamWaitEvent(stream_, ev_pre_join_, 0));
        from_manifolds = prelabel_nb_ == nb && build_cps_ && pre_manifolds_ && spec_build_applies(want_islands, nj) && pre_grid_ == spec_grid() && pre_lanes_ == spec_lanes_;
    }
    *fallback = false;
    RoctxRange range("GatherIslands + PrepareIndices (schedule build)");          // ref: Solver.cpp:77, 135, 217, 285
    // the topology fingerprint (already queued on the stream) rides along with the first readback of the build
    auto with_fingerprint = [&]() -> int { if (fp_wanted_) { PHX_TRY(rb_.add(fp_wanted_, hash_.p + hash_slot_, sizeof *fp_wanted_, stream_)); fp_wanted_ = nullptr; } return PHX_OK; };
    const bool trace = opt_.trace_schedule;
    auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) { if (!trace) return; (void)hipStreamSynchronize(stream_); auto n = std::chrono::steady_clock::now(); fprintf(stderr, "[schedule/gpu] %-18s %.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t0).count()); t0 = n; };
    const int nbs = std::max(nb, 1), njs = std::max(nj, 1);
    if (from_manifolds) bld_.cur ^= 1;                       // (the side stream filled the other set of counters and bin tables)
    PHX_TRY(bld_.cc_parent.reserve(nbs)); PHX_TRY(bld_.cc_static.reserve(nbs)); PHX_TRY(bld_.cc_flags.reserve(nbs + 1)); PHX_TRY(bld_.comp_size().reserve(nbs + 1));
    PHX_TRY(bld_.joint_comp.reserve(njs)); PHX_TRY(bld_.sb_small.reserve(8));
    PHX_TRY(bld_.rec_a.reserve(njs)); PHX_TRY(bld_.rec_b.reserve(njs));
    PHX_TRY(hbm_.order.reserve(njs));
    PHX_TRY(bld_.partner.reserve(njs)); PHX_TRY(bld_.comp_units().reserve(nbs + 1));
    if (!from_manifolds) {
        // units (schedule.h): contact point -> first joint carrying it (reset by k_cc_init); the partners are found by the linking pass
        const size_t had = bld_.partner_first.cap;
        PHX_TRY(bld_.partner_first.reserve(std::max(ncp_, 1)));
        if (bld_.partner_first.cap != had || bld_.partner_tag <= 1) {          // a new table, or the tags ran out: every entry reads 'nobody' again
            PHX_HIP(hipMemsetAsync(bld_.partner_first.p, 0xFF, bld_.partner_first.cap * sizeof(unsigned long long), stream_));
            bld_.partner_tag = 0xFFFFFFFEu;
        } else --bld_.partner_tag;
        hipLaunchKernelGGL(k_cc_init, dim3(grid_for(std::max(nb, nj))), dim3(256), 0, stream_, d_bodies, nb, bld_.cc_parent.p, bld_.cc_static.p, bld_.sb_small.p,
                           d_joints, nj, ncp_, bld_.partner_first.p, bld_.partner_tag);
    }
    Schedule sc;
    sc.colour_offsets.assign(1, 0); sc.group_offsets.assign(1, 0); sc.group_first_colour.assign(1, 0); sc.group_body_offsets.assign(1, 0);
    sc.islands = want_islands; sc.lds_on_host = false;
    int nbins = 0, lds_slots = 0, where = 0, ncomp_total = 0;
    bool any_partitioned = false;        // some component has more than COLOUR_B_MAX_JOINTS joints (schedule.h)
    spec_bins_pending_ = false;
    if (spec_build_applies(want_islands, nj)) {
        PHX_TRY(build_bins_speculative(d_bodies, nb, d_joints, nj, sc, from_manifolds));
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
    for (int k = 0; k < 2; ++k) { PHX_TRY(bld_.sort_keys[k].reserve(njs)); PHX_TRY(bld_.sort_vals[k].reserve(njs)); }
    PHX_TRY(bld_.sort_hist.reserve(radix_hist_words(nj)));
    {
        hipLaunchKernelGGL(k_cc_link, dim3(grid_for(nj)), dim3(256), 0, stream_, d_joints, nj, nb, bld_.cc_parent.p, (const unsigned char*)bld_.cc_static.p,
                           (const unsigned long long*)bld_.partner_first.p, bld_.partner_tag, ncp_, bld_.partner.p, sched_.has_hbm_group() ? 1 : 0);
        hipLaunchKernelGGL(k_cc_compress_window, dim3(std::max(1, div_up(nb, CCW_BODIES))), dim3(CCW_T), 0, stream_, bld_.cc_parent.p, nb);
        hipLaunchKernelGGL(k_cc_compress, dim3(grid_for(nb)), dim3(256), 0, stream_, bld_.cc_parent.p, nb, bld_.sb_small.p);
        PHX_TRY(device_exclusive_scan_of(RootFlagLoad{(const int*)bld_.cc_parent.p, nb, bld_.comp_size().p, bld_.comp_units().p}, bld_.cc_flags.p, nb + 1,
                                         reinterpret_cast<unsigned*>(bld_.sb_small.p + 1), bld_.sort_scan, stream_));
        hipLaunchKernelGGL(k_joint_components, dim3(std::max(1, std::min(div_up(nj, JC_T), 1024))), dim3(JC_T), 0, stream_, d_joints, nj, nb, (const int*)bld_.cc_parent.p,
                           (const unsigned*)bld_.cc_flags.p, (const int*)bld_.partner.p, bld_.joint_comp.p, bld_.comp_size().p, bld_.comp_units().p, bld_.sb_small.p);
        // fetch as many sizes as the previous build needed (+25 %); the rest, if any, in a second trip
        guess = std::min(nb, std::max(1024, ncomp_guess_ + ncomp_guess_ / 4));
        comp_size.assign(std::max(guess, 1), 0u); comp_units.assign(std::max(guess, 1), 0u);
        PHX_TRY(with_fingerprint());
        int pair[2] = {0, 0};                              // {labels disagree, component count}: adjacent words, one copy
        PHX_TRY(rb_.add(pair, bld_.sb_small.p, sizeof pair, stream_));
        PHX_TRY(rb_.add(comp_size.data(), bld_.comp_size().p, (size_t)guess * sizeof(unsigned), stream_));
        PHX_TRY(rb_.add(comp_units.data(), bld_.comp_units().p, (size_t)guess * sizeof(unsigned), stream_));
        PHX_TRY(rb_.wait(stream_));
        if (pair[0]) { set_error("connected components: a joint's bodies carry different labels"); return PHX_ERR_STATE; }
        ncomp_u = (unsigned)pair[1];
    }
    lap("components+count");
    const int ncomp = (int)ncomp_u;
    ncomp_total = ncomp;
    if (ncomp > guess) {
        comp_size.resize(ncomp); comp_units.resize(ncomp);
        PHX_TRY(rb_.add(comp_size.data() + guess, bld_.comp_size().p + guess, (size_t)(ncomp - guess) * sizeof(unsigned), stream_));
        PHX_TRY(rb_.add(comp_units.data() + guess, bld_.comp_units().p + guess, (size_t)(ncomp - guess) * sizeof(unsigned), stream_));
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
    PHX_TRY(bld_.bin_tables().reserve(table_words)); PHX_TRY(bld_.bin_tables_host.reserve(table_words));
    std::copy(bin_of.begin(), bin_of.end(), bld_.bin_tables_host.p);
    std::copy(rank_of.begin(), rank_of.end(), bld_.bin_tables_host.p + nc1);
    std::copy(sc.group_offsets.begin(), sc.group_offsets.begin() + nbins + 1, bld_.bin_tables_host.p + 2 * nc1);
    if (table_words <= 65536)
        hipLaunchKernelGGL(k_upload_words, dim3(std::max(1, std::min(div_up((int)table_words, 256), 64))), dim3(256), 0, stream_,
                           reinterpret_cast<unsigned*>(bld_.bin_tables().p), reinterpret_cast<const unsigned*>(bld_.bin_tables_host.p), (int)table_words);
    else PHX_HIP(hipMemcpyAsync(bld_.bin_tables().p, bld_.bin_tables_host.p, table_words * sizeof(int), hipMemcpyHostToDevice, stream_));
    const int* bin_of_comp = bld_.bin_tables().p; const int* rank_of_comp = bld_.bin_tables().p + nc1; const int* grp_goff = bld_.bin_tables().p + 2 * nc1;
    lap("bin");

    // 4. the units dealt to their bins (k_joint_scatter: a fill counter per bin, k_build_bin sorts its own), and the HBM group's joints
    //    — components too big for a workgroup, joints between static bodies — compacted in joint order behind the bins' slots
    const int rest_n = nj - lds_slots;
    PHX_TRY(bld_.bin_cursor.reserve((size_t)nbins + 2));
    PHX_HIP(hipMemsetAsync(bld_.sb_small.p + 2, 0, 2 * sizeof(int), stream_));      // 'a bin was rejected', the dealer's fail bits
    if (nbins) {
        PHX_HIP(hipMemsetAsync(bld_.bin_cursor.p, 0, ((size_t)nbins + 1) * sizeof(unsigned), stream_));
        ScatterView sv{};
        sv.joints = d_joints; sv.nj = nj; sv.nb = nb; sv.parent = bld_.cc_parent.p; sv.joint_comp = bld_.joint_comp.p; sv.partner = bld_.partner.p;
        sv.bin_of = bin_of_comp; sv.rank_of = rank_of_comp; sv.goff = grp_goff; sv.result = nullptr; sv.nbins = nbins; sv.max_bins = nbins; sv.ncomp_cap = std::max(ncomp, 1);
        sv.cursor = bld_.bin_cursor.p; sv.rec_a = bld_.rec_a.p; sv.rec_b = bld_.rec_b.p; sv.rejected = bld_.sb_small.p + 2; sv.spoil = bld_.sb_small.p + 3;
        hipLaunchKernelGGL(k_joint_scatter, dim3(std::max(1, std::min(div_up(nj, 256), 2048))), dim3(256), 0, stream_, sv);
    }
    if (rest_n > 0) {
        if (nbins) {
            PHX_TRY(bld_.sort_keys[0].reserve((size_t)njs + 1));
            hipLaunchKernelGGL(k_rest_flags, dim3(grid_for(nj + 1)), dim3(256), 0, stream_, (const int*)bld_.joint_comp.p, bin_of_comp, nj, nbins, std::max(ncomp, 1), bld_.sort_keys[0].p);
            PHX_TRY(device_exclusive_scan(bld_.sort_keys[0].p, nj + 1, nullptr, bld_.sort_scan, stream_));
            hipLaunchKernelGGL(k_compact_flagged, dim3(grid_for(nj)), dim3(256), 0, stream_, (const unsigned*)bld_.sort_keys[0].p, nj, reinterpret_cast<int*>(bld_.sort_vals[0].p + lds_slots));
        } else hipLaunchKernelGGL(k_iota, dim3(grid_for(nj)), dim3(256), 0, stream_, bld_.sort_vals[0].p, nj);      // (Single mode, or nothing fits a workgroup: every joint, in joint order)
    }
    where = 0;
    lap("deal");

    // 5. one workgroup per bin: body table, colouring, slot arrays
    PHX_TRY(isl_.desc.reserve(std::max(nbins, 1))); PHX_TRY(isl_.ncol.reserve(std::max(nbins, 1)));
    PHX_TRY(isl_.bodies.reserve((size_t)std::max(nbins, 1) * cap_bodies));
    PHX_TRY(isl_.slot_local.reserve(std::max(lds_slots, 1))); PHX_TRY(isl_.slot_colour.reserve(std::max(lds_slots, 1)));
    if (nbins) {
        BinBuildView bv{};
        PHX_TRY(isl_.units.reserve(nbins)); PHX_TRY(isl_.unit_recs.reserve(2 * (size_t)nbins * cap_units));
        bv.rec_a = bld_.rec_a.p; bv.rec_b = bld_.rec_b.p; bv.group_offsets = grp_goff; bv.cursor = bld_.bin_cursor.p; bv.spoil = bld_.sb_small.p + 3;
        bv.nb = nb; bv.max_static = 1 << 30;
        bv.order = hbm_.order.p; bv.slot_local = isl_.slot_local.p; bv.slot_colour = isl_.slot_colour.p; bv.desc = isl_.desc.p; bv.ncol = isl_.ncol.p;
        bv.units = isl_.units.p; bv.unit_recs = isl_.unit_recs.p;
        bv.bodies = isl_.bodies.p; bv.rejected = bld_.sb_small.p + 2; bv.poison = hash_.p + hash_slot_;
        if (cap_units > ISL_T) hipLaunchKernelGGL((k_build_bin<ISL_T_BIG, ISL_B_BIG>), dim3(nbins), dim3(ISL_T_BIG), 0, stream_, bv);
        else hipLaunchKernelGGL((k_build_bin<ISL_T, ISL_B>), dim3(nbins), dim3(ISL_T), 0, stream_, bv);
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
        jv.joint_comp = bld_.joint_comp.p; jv.partner = bld_.partner.p; jv.kind = bld_.jp_kind.p; jv.ncomp = ncomp_total; jv.comp_size = bld_.comp_size().p;
        jv.seen_a = bld_.jp_seen.p; jv.seen_b = bld_.jp_seen.p + ncomp_total + 1; jv.seen_c = bld_.jp_seen.p + 2 * ((size_t)ncomp_total + 1); jv.bad_b = bld_.jp_bad_b.p;
        jv.counts = bld_.jp_counts.p; jv.flags = bld_.jp_small.p; jv.hist = reinterpret_cast<unsigned*>(bld_.jp_small.p + 4);
        const unsigned* perm = nullptr;                     // the entries sorted by part (partitioned components only)
        const int parts = parts_total(nb);                    // over both levels (schedule.h)
        unsigned h_hist[2 * JP_MAX_COLOURS], h_touched = 0, h_nstatic = 0;
        int h_flags[3] = {0, 0, 0};                            // [0] bit 1: the interior + other classes exceed the device builder's 64; [1] KI0; [2] KI1
        // (round 6 tried to take the look at the frontier sizes below with the build's LAST readback — as many rounds queued as the previous
        //  build needed, + 2: a plain step was no faster (the host's wait overlaps queued work) and six of 120 steps of the settling 200k world
        //  needed more rounds than that and were built twice, 0.4 ms each: removed)
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
        // (round 6: when the previous build walked few entries, ONE workgroup walks all the rounds in one launch and the host never looks at
        //  the frontier sizes: k_jp_walk_one — its {rounds, entries, 'ran out of rounds'} come back with the build's last readback)
        const bool walk_one = !trace && !opt_.no_jp_walk_one && jp_walk_entries_ >= 0 && jp_walk_entries_ <= JP_WALK_ONE_MAX;
        int walk_result[3] = {0, 0, 0};
        long long walked = 0;
        if (walk_one) {
            PHX_TRY(bld_.jp_walk_result.reserve(4));
            hipLaunchKernelGGL(k_jp_walk_one, dim3(1), dim3(JP_WALK_T), 0, stream_, jv, JP_ROUNDS_MAX, bld_.jp_list[0].p, bld_.jp_list[1].p, bld_.jp_walk_result.p);
        }
        for (bool done = walk_one; !done;) {
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
                else if (k < batch) walked += n;                                               // (the batch's last entry is the next batch's first)
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
        PHX_TRY(rb_.add(h_hist, hist, sizeof h_hist, stream_));
        PHX_TRY(rb_.add(&h_touched, bld_.jp_touched.p + nb, sizeof h_touched, stream_));
        // static slots (only the HBM path indexes the global static-tag tables)
        // (a table of its own: the readback batch reads its sources at wait(), so bld_.jp_touched must stay as it is until then)
        PHX_TRY(hbm_.static_slot.reserve(nbs));
        unsigned* sflags = bld_.jp_degree.p;                       // per body + 1; the colouring is done with it
        hipLaunchKernelGGL(k_static_flags, dim3(grid_for(nb + 1)), dim3(256), 0, stream_, (const unsigned char*)bld_.cc_static.p, nb, sflags);
        PHX_TRY(device_exclusive_scan(sflags, nb + 1, nullptr, bld_.sort_scan, stream_));
        hipLaunchKernelGGL(k_static_slots, dim3(grid_for(nb)), dim3(256), 0, stream_, (const unsigned char*)bld_.cc_static.p, (const unsigned*)sflags, nb, hbm_.static_slot.p);
        PHX_TRY(rb_.add(&h_nstatic, sflags + nb, sizeof h_nstatic, stream_));
        PHX_TRY(rb_.add(h_flags, bld_.jp_small.p, sizeof h_flags, stream_));
        if (walk_one) { PHX_TRY(with_fingerprint()); PHX_TRY(rb_.add(walk_result, bld_.jp_walk_result.p, sizeof walk_result, stream_)); }
        PHX_TRY(rb_.wait(stream_));
        if (walk_one) {                                        // what the rounds' loop checks between its batches
            if (h_flags[0] & 1) { set_error("a joint references a body out of range"); return PHX_ERR_INVALID; }
            if ((h_flags[0] & 4) || walk_result[2]) { *fallback = true; return PHX_OK; }       // a body in thousands of joints, or a pathological dependency chain: host builder
            jp_rounds_guess_ = walk_result[0]; walked = walk_result[1];
        }
        jp_walk_entries_ = walked;
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
        {
            int interior_leaders = 0;
            PHX_TRY(upload_class_tab(sc, &interior_leaders));      // (k_solve_parts' and k_solve_tail's table)
        }
        if (sc.hbm_interior_classes > 0) {
            const int ki = sc.hbm_interior_classes;
            if (!perm || ki >= (int)sc.hbm_class_leaders.size() + 1 || ki > JP_MAX_COLOURS) { set_error("interior classes out of range"); return PHX_ERR_STATE; }
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
int DeviceSolver::build_bins_speculative(const float4* d_bodies, int nb, const phx_contact_joint* d_joints, int nj, Schedule& sc, bool from_manifolds)
{
    const int grid = spec_grid();
    const int cap_units = spec_lanes_, cap_bodies = cap_units > ISL_T ? ISL_B_BIG : ISL_B;
    PHX_TRY(bld_.bin_tables().reserve(2 * (size_t)BINC_MAX + (size_t)grid + 2));
    PHX_TRY(bld_.bin_result.reserve(16)); PHX_TRY(bld_.bin_cursor.reserve((size_t)grid + 2)); PHX_TRY(bld_.side_flags.reserve(4));
    gate_expected_ = 0x5EED000000000000ull | (++gate_serial_ & 0xFFFFFFFFFFFFull);
    int* const bin_of = bld_.bin_tables().p; int* const rank_of = bld_.bin_tables().p + BINC_MAX; int* const goff = bld_.bin_tables().p + 2 * BINC_MAX;
    BinBuildView bv{};
    if (from_manifolds) {
        // components, counts, bins and the units' slots stand (side stream): the bins are built — ONE launch
        ++lite_builds_;
        bv.unit_m = bld_.unit_m.p; bv.manifolds = pre_manifolds_; bv.cps = build_cps_; bv.joints = d_joints; bv.nj = nj;
        bv.side_flags = bld_.side_flags.p; bv.result = bld_.bin_result.p; bv.gate = gate_expected_;
    } else {
        ++full_builds_;
        hipLaunchKernelGGL(k_cc_link, dim3(grid_for(nj)), dim3(256), 0, stream_, d_joints, nj, nb, bld_.cc_parent.p, (const unsigned char*)bld_.cc_static.p,
                           (const unsigned long long*)bld_.partner_first.p, bld_.partner_tag, ncp_, bld_.partner.p, sched_.has_hbm_group() ? 1 : 0);
        hipLaunchKernelGGL(k_cc_compress_window, dim3(std::max(1, div_up(nb, CCW_BODIES))), dim3(CCW_T), 0, stream_, bld_.cc_parent.p, nb);
        hipLaunchKernelGGL(k_cc_compress, dim3(grid_for(nb)), dim3(256), 0, stream_, bld_.cc_parent.p, nb, bld_.sb_small.p);
        PHX_TRY(device_exclusive_scan_of(RootFlagLoad{(const int*)bld_.cc_parent.p, nb, bld_.comp_size().p, bld_.comp_units().p}, bld_.cc_flags.p, nb + 1,
                                         reinterpret_cast<unsigned*>(bld_.sb_small.p + 1), bld_.sort_scan, stream_));
        hipLaunchKernelGGL(k_joint_components, dim3(std::max(1, std::min(div_up(nj, JC_T), 1024))), dim3(JC_T), 0, stream_, d_joints, nj, nb, (const int*)bld_.cc_parent.p,
                           (const unsigned*)bld_.cc_flags.p, (const int*)bld_.partner.p, bld_.joint_comp.p, bld_.comp_size().p, bld_.comp_units().p, bld_.sb_small.p);
        BinCompView cv{};
        cv.comp_size = bld_.comp_size().p; cv.comp_units = bld_.comp_units().p; cv.cc_small = bld_.sb_small.p; cv.nj = nj;
        cv.cap_units = cap_units; cv.small_units = ISL_T; cv.max_bins = grid;
        cv.bin_of = bin_of; cv.rank_of = rank_of; cv.goff = goff;
        cv.result = bld_.bin_result.p; cv.cursor = bld_.bin_cursor.p;
        cv.fingerprint = hash_.p + hash_slot_; cv.hash_out = reinterpret_cast<unsigned long long*>(bld_.bin_result.p + 8);
        cv.gate = gate_expected_;
        // (a workgroup packs 16 chunks of 64 components at a time: as many workgroups as last build's component count asks for — from
        //  four rounds on; below that the hand-over through memory costs more than it saves: 15.7 against 9.6 us at cfg 2's 16 chunks)
        const int bin_chunks = div_up(ncomp_guess_ + ncomp_guess_ / 4, BIN_CHUNK);
        const int bin_groups = bin_chunks <= 3 * (BINC_T / 64) ? 1 : std::min(16, div_up(bin_chunks, BINC_T / 64));
        if (!bld_.bin_scratch.p) {
            PHX_TRY(bld_.bin_scratch.reserve(2 * (size_t)BINC_T + 2));
            PHX_HIP(hipMemsetAsync(bld_.bin_scratch.p, 0, bld_.bin_scratch.cap * sizeof(unsigned long long), stream_));
        }
        cv.scratch = bld_.bin_scratch.p;
        hipLaunchKernelGGL(k_bin_components, dim3(bin_groups), dim3(BINC_T), 0, stream_, cv);
        ScatterView sv{};
        sv.joints = d_joints; sv.nj = nj; sv.nb = nb; sv.parent = bld_.cc_parent.p; sv.joint_comp = bld_.joint_comp.p; sv.partner = bld_.partner.p;
        sv.bin_of = bin_of; sv.rank_of = rank_of; sv.goff = goff; sv.result = bld_.bin_result.p; sv.nbins = 0; sv.max_bins = grid; sv.ncomp_cap = BINC_MAX;
        sv.cursor = bld_.bin_cursor.p; sv.rec_a = bld_.rec_a.p; sv.rec_b = bld_.rec_b.p; sv.rejected = bld_.sb_small.p + 2; sv.spoil = bld_.bin_result.p + 7;
        hipLaunchKernelGGL(k_joint_scatter, dim3(std::max(1, std::min(div_up(nj, 256), 2048))), dim3(256), 0, stream_, sv);
        bv.rec_a = bld_.rec_a.p; bv.rec_b = bld_.rec_b.p; bv.spoil = bld_.bin_result.p + 7;
    }
    PHX_TRY(isl_.desc.reserve(grid)); PHX_TRY(isl_.ncol.reserve(grid));
    PHX_TRY(isl_.bodies.reserve((size_t)grid * cap_bodies));
    PHX_TRY(isl_.slot_local.reserve(nj)); PHX_TRY(isl_.slot_colour.reserve(nj));
    PHX_TRY(isl_.units.reserve(grid)); PHX_TRY(isl_.unit_recs.reserve(2 * (size_t)grid * cap_units));
    bv.group_offsets = goff; bv.cursor = bld_.bin_cursor.p;
    bv.nb = nb; bv.max_static = 1 << 30;
    bv.order = hbm_.order.p; bv.slot_local = isl_.slot_local.p; bv.slot_colour = isl_.slot_colour.p; bv.desc = isl_.desc.p; bv.ncol = isl_.ncol.p;
    bv.units = isl_.units.p; bv.unit_recs = isl_.unit_recs.p;
    bv.bodies = isl_.bodies.p; bv.rejected = bld_.sb_small.p + 2; bv.poison = hash_.p + hash_slot_;
    bv.nbins_dev = bld_.bin_result.p;
    if (cap_units > ISL_T) hipLaunchKernelGGL((k_build_bin<ISL_T_BIG, ISL_B_BIG>), dim3(grid), dim3(ISL_T_BIG), 0, stream_, bv);
    else hipLaunchKernelGGL((k_build_bin<ISL_T, ISL_B>), dim3(grid), dim3(ISL_T), 0, stream_, bv);
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
    PHX_TRY(fetch_build_tables());
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
    if (!tab.empty()) PHX_HIP(hipMemcpyAsync(parts_.class_tab.p, tab.data(), tab.size() * sizeof(int4), hipMemcpyHostToDevice, stream_));
    class_tab_ok_ = true;
    return PHX_OK;
}

// host-built schedules: the interior units by part as the builder left them (schedule.hip build_part_tables)
int DeviceSolver::upload_part_tables()
{
    parts_.count = 0;
    const int ki = sched_.hbm_interior_classes;
    int interior_leaders = 0;
    PHX_TRY(upload_class_tab(sched_, &interior_leaders));         // (k_solve_parts' and k_solve_tail's table)
    if (ki <= 0 || sched_.part_begin.empty()) return PHX_OK;      // (more than 64 interior classes: no tables, one launch per class)
    if (interior_leaders != sched_.part_begin.back()) { set_error("part tables do not match the interior classes"); return PHX_ERR_STATE; }
    const size_t parts = sched_.part_begin.size() - 1;
    PHX_TRY(parts_.ranges.reserve(parts * PARTS_CLASS_STRIDE)); PHX_TRY(parts_.begin.reserve(parts + 2));
    PHX_HIP(hipMemcpyAsync(parts_.ranges.p, sched_.part_ranges.data(), sched_.part_ranges.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    PHX_HIP(hipMemcpyAsync(parts_.begin.p, sched_.part_begin.data(), sched_.part_begin.size() * sizeof(int), hipMemcpyHostToDevice, stream_));
    PHX_HIP(hipStreamSynchronize(stream_));
    parts_.count = (int)parts;
    return PHX_OK;
}

} // namespace phx
