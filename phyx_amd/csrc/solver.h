// solver.h — host side of the device solver: schedule (colouring / islands), residency, launches.
#pragma once

#include "common.h"
#include "body_view.h"
#include "schedule.h"
#include "comm.h"

namespace phx {

// device-side view of the solver state, passed by value to every kernel (layout: solver_kernels.h)
struct SolverView {
    int nb, nj, ncp, nstatic, ncolours;
    const unsigned long long* fingerprint;       // topology fingerprint computed for this call (device word)
    unsigned long long expected_fingerprint;     // the one the schedule in use was built for
    float4* sb_imp;
    float4* sb_disp;
    const float4* sb_par;  // = the resident mpos array of the solve's bodies (body_view.h)
    float4* q0;
    float4* q1;
    float4* q2;
    int4* q3;
    float2* acc;
    float2* dd;
    float* qn;             // per slot: q2.x again (what a unit's follower reads instead of q2 and q3)
    const int* order;      // slot -> joint index
    unsigned* sw_imp;      // static-body productive words, [2][nstatic] (see static_word())
    unsigned* sw_disp;
    unsigned long long* stamps;   // [0] min clock of the solve's first kernel, [1] max clock of its last (100 MHz wall clock)
    int* imp_active;       // [iter] 1 if any joint was productive in sweep `iter`
    int* disp_active;
};

// the interior units of partitioned components by part (schedule.h; solver_kernels.h k_solve_parts)
struct PartsView {
    const int4* ranges;         // [part * 64 + class]: the part's slots of that interior class = {first, end} of its leaders with a
                                // follower, {first, end} of its single leaders (an interior class is laid out part by part, schedule.h)
    const int* part_begin;      // parts + 1: interior units before each part (an empty part is skipped)
    const int4* class_tab;      // per class of the HBM group: {first slot, leaders, followers, leaders of the classes before}
    int first_part, parts;      // the parts of ONE level (schedule.h): a launch sweeps them, workgroup w = part first_part + w
    int c0, c1;                 // ... and that level's classes [c0, c1)
};
constexpr int PARTS_CLASS_STRIDE = 64;      // = JP_MAX_COLOURS, the device schedule builder's class limit

class DeviceSolver {
public:
    explicit DeviceSolver(int device) : device_(device) {}
    ~DeviceSolver();
    int init();

    int solve_host(phx_rigid_body* bodies, int nb, const phx_contact_point* cps, int ncp, phx_contact_joint* joints, int nj, const phx_config& cfg);
    // the C-ABI edge: device-resident 128-byte records (converted to the resident arrays and back around the solve, body_view.h)
    // topology_changed: the caller knows the joint list differs from the previous solve's (skips one host round trip)
    int solve_device(void* d_bodies, int nb, const void* d_cps, int ncp, void* d_joints, int nj, const phx_config& cfg, bool topology_changed = false);
    // the resident form (what the World and bench() use): reads vel / dvel / mpos, writes vel / dvel in place
    int solve_resident(const BodyView& bodies, int nb, const void* d_cps, int ncp, void* d_joints, int nj, const phx_config& cfg, bool topology_changed = false);
    // (`while_waiting`, may be null: queued-work hook of the settling round trip, Readback::wait — called at most once, and only if a
    //  solve is pending; the caller checks whether it ran)
    // The components, their joint counts and the BINS from the MANIFOLDS, on the side stream, while the caller still works on its joint
    // list (schedule_kernels.h k_cc_link_manifolds, k_manifold_components, k_bin_components): a manifold with n contact points has n
    // joints once RefreshContactJoints is through.  The next rebuild waits for them and is then two launches — the joints dealt to their
    // bins, the bins built (PHX_NO_PRELABEL=1: never).  cancel_prelabel(): no rebuild followed, or the manifolds moved under the side stream.
    // prelabel_mark(): the manifolds are final from HERE on the stream (an event); prelabel_components(): queue the side stream's work
    // behind that point — called later, once the stream has the caller's next kernels to run while the host queues these.
    int prelabel_mark();
    int prelabel_components(const float4* d_mpos, int nb, const phx_manifold* d_manifolds, int nm);
    int cancel_prelabel();
    bool prelabel_pending() const { return prelabel_pending_; }
    void build_counts(int64_t out2[2]) const { out2[0] = lite_builds_; out2[1] = full_builds_; }      // device rebuilds whose components and bins came from the manifolds (side stream) / from the joints
    int synchronize(const std::function<int()>* while_waiting = nullptr, const MailCarrier* carrier = nullptr);      // (`carrier`: Readback::wait)
    int get_stats(phx_solve_stats* out);
    int get_schedule(int* order, int order_cap, int* offsets, int offsets_cap, int* ncolours);
    int set_body_state_bits(int bits);
    int set_shard(int shard, int count);
    void set_schedule_reuse(bool on) { reuse_schedule_ = on; }
    void set_trace(int level) { trace_islands_ = level != 0; trace_waves_ = level != 1; drop_graphs(); }      // 1: phase stamps only; else also per-wave step cycles
    int get_island_trace(unsigned long long* out, int cap_groups, int* groups);
    int get_wave_trace(unsigned long long* out, int cap_words, int* waves_per_group);
    int get_groups(int* offsets, int cap, int* count, int* lds_count);
    int get_lanes(int* leader_slot, int* lane, int cap, int* count);      // the LDS groups' units: leader slot -> lane of the island kernel (schedule.h LANES)
    int get_partition(int* interior_classes, int* parts, int* sweep_launches)
    {
        PHX_TRY(synchronize());
        if (interior_classes) *interior_classes = sched_.hbm_interior_classes;
        if (parts) *parts = parts_in_use() ? parts_.count : 0;
        if (sweep_launches) *sweep_launches = sweep_launches_;
        return PHX_OK;
    }
    int get_refreshed(int joint, float out30[30]);
    int bench_stage(const void* d_bodies, int nb, const void* d_joints, int nj, int steps);
    int bench(const void* d_bodies, int nb, const void* d_cps, int ncp, const void* d_joints, int nj,
              const phx_config& cfg, int warmup, int steps, phx_bench_result* out, phx_step_hook hook = nullptr, void* user = nullptr);
    // 64-bit checksum of what the LAST step of the last bench() left in its copy of the input: velocities, displacing velocities and
    // accumulated impulses, position-sensitive (bench.py compares the timed blocks' with the warm-up solve's, outside its clock)
    int bench_checksum(unsigned long long* out);

    // island-sharded solves: pack this rank's results / scatter the other ranks' (exchange.h); the all-gather in between
    // belongs to the caller (RCCL on stream())
    int set_exchange_buffers(void* d_send, void* d_recv, size_t segment_capacity_bytes);
    int exchange_pack(const void* d_bodies, const void* d_joints, size_t* segment_bytes, int status_word = 0);      // 128-byte records (C-ABI edge)
    int exchange_unpack(void* d_bodies, void* d_joints);
    int exchange_pack_resident(const BodyView* bodies, const void* d_joints, size_t* segment_bytes, int status_word = 0);   // null bodies: header only
    int exchange_unpack_resident(const BodyView& bodies, void* d_joints);
    int exchange_status(int* out);
    size_t exchange_segment_bytes() const { return (size_t)xch_seg_words_ * 4; }
    int shard_count() const { return shard_count_; }
    void set_comm(Comm* c) { comm_ = c; }              // bench(): the all-gather between pack and unpack runs natively (comm.hip)
    int exchange_all_gather();                        // the segment of the last pack, every rank's into the recv buffer, on stream()

    void set_wait_timeout(double seconds) { rb_.set_timeout(seconds); }      // (a sharded World: the stream carries collectives, common.h Readback)
    hipStream_t stream() const { return stream_; }
    // run on a caller-owned stream from now on (the World puts broadphase, step kernels and solver on one stream so that
    // consecutive phases need no host synchronisation)
    int adopt_stream(hipStream_t s);
    bool has_pending() const { return pending_.active; }
    // the gate of an unverified solve (world.hip queues the integrator behind it under the same gate)
    const unsigned long long* fingerprint_word() const { return hash_.p + hash_slot_; }
    unsigned long long expected_fingerprint() const { return gate_expected_; }
    // the C-ABI edge's arrays after a solve_device / exchange call on records (tests, bench plumbing)
    const BodyView& current_view() const { return cur_.view; }
    unsigned replays() const { return replays_; }
    int device() const { return device_; }

private:
    // the arrays of one solve: the resident view, and — for a solve that came through the C-ABI edge — the caller's records
    struct Arrays { BodyView view{nullptr, nullptr, nullptr}; phx_rigid_body* aos = nullptr; };
    int solve_common(const Arrays& a, int nb, const void* d_cps, int ncp, void* d_joints, int nj, const phx_config& cfg, bool topology_changed);
    int edge_view(const void* d_bodies_aos, int nb, Arrays* out);      // converts the records into this handle's edge arrays
    int ensure_schedule(const float4* d_mpos, int nb, const phx_contact_joint* d_joints, int nj, int ncp, const phx_config& cfg, bool force_rebuild,
                        bool known_changed = false);
    int build_schedule_device(const float4* d_mpos, int nb, const phx_contact_joint* d_joints, int nj, bool want_islands, bool* fallback);
    int build_bins_speculative(const float4* d_mpos, int nb, const phx_contact_joint* d_joints, int nj, Schedule& sc, bool from_manifolds);
    int spec_grid() const;                // the launch grid of a speculative build's per-bin kernels
    int materialise_schedule();
    int launch_fingerprint(const float4* d_mpos, int nb, const phx_contact_joint* d_joints, int nj, int ncp);
    // How the solve being queued is gated (island_view.h): by the hash pass / the build (ISL_GATED) or by the island kernel's own
    // check of the cached schedule (ISL_VERIFY).  arm_cached_solve picks one for a solve on the cached schedule; false = neither is
    // possible (no hash on record and the launch is not eligible for ISL_VERIFY): the caller rebuilds.
    bool arm_cached_solve(const float4* d_mpos, int nb, const phx_contact_joint* d_joints, int nj, int ncp, int* status);
    bool verify_eligible(int groups, bool big_shape) const;
    bool spec_build_applies(bool want_islands, int nj) const;
    void begin_set(bool hash_runs);                   // the solve being queued takes the other control set
    int complete_partial();                           // ISL_COMPLETE for the groups a verified launch left uncommitted
    struct GraphKey {
        const void *bodies = nullptr, *cps = nullptr, *joints = nullptr;
        int nb = 0, nj = 0, ncp = 0, ci = 0, pi = 0;
        long long schedule_version = -1;
        bool valid = false;
        bool operator==(const GraphKey& o) const
        {
            return bodies == o.bodies && cps == o.cps && joints == o.joints && nb == o.nb && nj == o.nj && ncp == o.ncp && ci == o.ci && pi == o.pi &&
                   schedule_version == o.schedule_version;
        }
    };
    int enqueue(const Arrays& a, int nb, const phx_contact_point* d_cps, phx_contact_joint* d_joints, int nj, const phx_config& cfg);
    int enqueue_pre(const BodyView& bodies, int nb, const phx_contact_point* d_cps, phx_contact_joint* d_joints, int nj);
    int enqueue_sweeps(const BodyView& bodies, const phx_contact_point* d_cps, phx_contact_joint* d_joints, int nj, int ci, int pi, int mode_override = -1);
    int enqueue_post(const BodyView& bodies, int nb, phx_contact_joint* d_joints, int nj);
    int capture_graphs(const GraphKey& key, const BodyView& bodies, const phx_contact_point* d_cps, phx_contact_joint* d_joints);
    void drop_graphs();
    int collect_stats(unsigned long long* extra = nullptr, const unsigned long long* extra_src = nullptr, const std::function<int()>* while_waiting = nullptr,
                      const MailCarrier* carrier = nullptr);
    SolverView view() const;

    int device_;
    hipStream_t stream_ = nullptr;
    hipStream_t side_stream_ = nullptr;      // the LDS islands of a schedule that also has an HBM group (enqueue_sweeps)
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
    hipEvent_t ev_pre_fork_ = nullptr, ev_pre_join_ = nullptr;      // prelabel_components: stream_ -> side stream, side stream -> the rebuild
    bool prelabel_marked_ = false, prelabel_pending_ = false; int prelabel_nb_ = 0;
    const phx_manifold* pre_manifolds_ = nullptr; int pre_nm_ = 0, pre_grid_ = 0, pre_lanes_ = 0;      // what the side stream binned from, and for which launch grid / shape
    const phx_contact_point* build_cps_ = nullptr;      // the contact points of the solve being queued (the manifold build pairs the joints through them)
    hipEvent_t ev_begin_ = nullptr, ev_end_ = nullptr, ev_sweep_begin_ = nullptr, ev_sweep_end_ = nullptr;


    // ---- what the handle owns, by path.  Every buffer is a DevBuf (freed with the handle); init() reads the environment once ----
    // the PHX_* knobs (measurement and debugging; README.md lists them)
    struct Options {
        bool no_side_stream = false;      // PHX_NO_SIDE_STREAM=1: everything on the one stream
        bool no_jp_walk_one = false;      // PHX_NO_JP_WALK_ONE=1: the HBM group's colouring walk always one launch per round (k_jp_front) with the host's look in between
        bool no_tail = false;             // PHX_NO_TAIL=1: the HBM group's trailing tiny classes one launch each (k_solve_colour) instead of one workgroup's launch (k_solve_tail)
        bool no_parts = false;            // PHX_NO_PARTS=1: sweep the interior classes one launch each
        bool no_fused_verify = false;     // PHX_NO_FUSED_VERIFY=1: always the hash pass (also set for good once a verified launch timed out)
        bool use_graphs = false;          // PHX_GRAPHS=1
        bool gpu_builder = true;          // PHX_SCHEDULE_BUILDER=host clears it
        bool speculate = true;            // PHX_NO_SPECULATION=1 clears it
        bool no_islands = false;          // PHX_NO_ISLANDS=1
        bool no_spec_bins = false;        // PHX_NO_SPEC_BINS=1
        bool no_prelabel = false;         // PHX_NO_PRELABEL=1: the World's rebuilds take their components from the joints (A/B, tests)
        bool trace_schedule = false;      // PHX_TRACE_SCHEDULE
        int isl_wait_polls = 0;           // PHX_ISL_WAIT_POLLS
        static Options from_env();
    } opt_;
    // the HBM path (any island size; everything in Single mode): solver-side body state, joint constants in schedule order,
    // per-sweep 'productive' flags, static tags, the bodies the HBM group touches
    struct HbmPath {
        DevBuf<float4> sb_imp, sb_disp, q0, q1, q2;
        DevBuf<float> qn;
        DevBuf<int4> q3;
        DevBuf<float2> acc, dd;
        DevBuf<int> order, static_slot, flags, hbm_body_list;
        DevBuf<unsigned> sw;
    } hbm_;
    // the island path: per-group tables the island kernel addresses by the group number alone (island_view.h), its counters and
    // control sets
    struct IslandPath {
        DevBuf<int4> desc;
        DevBuf<int> ncol, units, bodies, stats;
        DevBuf<int4> unit_recs;           // per LDS group (stride = lanes of the kernel shape), two words per unit: joints, contact points, local bodies, class, slots
        DevBuf<unsigned> slot_local;
        DevBuf<unsigned char> slot_colour;
        DevBuf<unsigned long long> visits;
        DevBuf<unsigned> done;            // per LDS group: epoch of the last verified solve that committed it
        DevBuf<unsigned long long> shards;    // two control sets of ISL_SHARDS arrival counters
        DevBuf<unsigned long long> trace;
    } isl_;
    // the interior units of partitioned components by part (schedule.h, k_solve_parts)
    struct PartsPath {
        DevBuf<int> begin;
        DevBuf<int4> ranges;
        DevBuf<unsigned> keys[2], vals[2];    // the HBM group's entries sorted by part (k_colour_parts)
        DevBuf<int4> class_tab;
        std::vector<int4> class_tab_host;
        int count = 0;                        // workgroups of k_solve_parts (0: the schedule has no interior classes)
    } parts_;
    // scratch of the device schedule builder (schedule_kernels.h): components, units, binning, the sort by bin, the colouring of the HBM group
    struct ScheduleBuilder {
        DevBuf<int> cc_parent, joint_comp, sb_small;
        // The per-component counters and the bin tables (component -> bin | component -> rank in its bin | bin -> first slot) exist TWICE:
        // the side stream fills the set the last build did NOT use, so that what the last build left for the statistics
        // (fetch_build_tables) stands until a rebuild really replaces it (a prelabel that no rebuild followed zeroed them).
        DevBuf<int> bin_tables_s[2];
        DevBuf<unsigned> comp_size_s[2], comp_units_s[2];
        int cur = 0;                          // the set of the build in hand
        DevBuf<int>& bin_tables() { return bin_tables_s[cur]; }
        DevBuf<unsigned>& comp_size() { return comp_size_s[cur]; }
        DevBuf<unsigned>& comp_units() { return comp_units_s[cur]; }
        PinnedBuf<int> bin_tables_host;
        DevBuf<unsigned char> cc_static;
        DevBuf<unsigned> cc_flags, sort_keys[2], sort_vals[2], sort_hist;
        DevBuf<int4> rec_a;                   // the joints' records, bin by bin (schedule_kernels.h BinRecord)
        DevBuf<int2> rec_b;
        DevBuf<int2> unit_m;                  // the World's units, bin by bin: {manifold, component rank | static bits} (k_manifold_slots)
        DevBuf<unsigned> bin_cursor;          // per bin: units dealt (k_joint_scatter / k_manifold_slots)
        DevBuf<int> side_flags;               // k_manifold_components' flags
        ScanScratch sort_scan, prelabel_scan;      // (the side stream's scan keeps its own state: it runs beside the joint list's scans)
        DevBuf<int> partner;                  // joint -> the other joint of its unit (schedule.h)
        DevBuf<unsigned long long> partner_first;      // contact point -> tag << 32 | first joint carrying it (k_cc_init)
        unsigned partner_tag = 0;             // this build's tag: counts down, 0 = the table has to be cleared first
        DevBuf<int> bin_result;               // k_bin_components' results: 8 ints, then the topology hash (8 bytes)
        DevBuf<unsigned long long> bin_scratch;        // k_bin_components launched as several workgroups: what they hand to the last one
        DevBuf<unsigned long long> jp_used, jp_used_b, jp_seen;
        DevBuf<uint4> jp_ent, jp_adj;
        DevBuf<uint2> jp_succ;
        DevBuf<unsigned> jp_offset, jp_cursor, jp_pred, jp_ent_comp, jp_seed, jp_touched, jp_keys[2], jp_vals[2], jp_degree, jp_colour_b, jp_list[2];
        DevBuf<int> jp_walk_result;         // k_jp_walk_one: {rounds, entries walked, ran out of rounds}
        DevBuf<unsigned char> jp_bad_b, jp_kind;
        DevBuf<int> jp_small, jp_counts;
    } bld_;

    // device state
    DevBuf<float4> edge_vel_, edge_dvel_, edge_mpos_;      // resident form of the records a C-ABI edge call handed over
    Arrays cur_;                                          // arrays of the solve being queued (view() reads the resident mpos from it)
    int upload_class_tab(const Schedule& sc, int* interior_leaders);
    int tail_first_class(int from) const;                     // k_solve_tail's share of the HBM group's classes (solver.hip)
    bool class_tab_ok_ = false;                               // parts_.class_tab describes sched_'s HBM classes
    int upload_part_tables();            // host-built schedules: part_units_ / part_class_begin_ from sched_
    bool parts_in_use() const { return !opt_.no_parts && parts_.count > 0 && sched_.hbm_interior_classes > 0; }
    int part_levels() const { return sched_.hbm_interior_classes > sched_.hbm_interior_classes0 ? 2 : 1; }
    PartsView parts_view(int level, int nb) const
    {
        const int P = parts_per_level(nb), ki0 = sched_.hbm_interior_classes0, ki = sched_.hbm_interior_classes;
        return level == 0 ? PartsView{parts_.ranges.p, parts_.begin.p, parts_.class_tab.p, 0, P, 0, ki0}
                          : PartsView{parts_.ranges.p, parts_.begin.p, parts_.class_tab.p, P, P + 1, ki0, ki};
    }
    int jp_rounds_guess_ = 0;
    long long jp_walk_entries_ = -1;      // entries the last build's colouring walk visited (-1: no build yet): few -> k_jp_walk_one
    // a device-built schedule whose 'did every bin fit' flag has not been read yet (build_schedule_device, collect_stats)
    bool build_unverified_ = false, build_was_unverified_ = false, force_host_builder_ = false, defer_build_check_ = true;
    int unverified_bins_ = 0;
    // speculative binning (build_bins_speculative): the bins are made on the device and the build has no host round trip at all;
    // what the host would have read — bin count, offsets, GatherIslands' numbers, the topology hash — comes back when the solve is settled
    bool spec_bins_ok_ = false, spec_bins_pending_ = false, spec_bins_failed_ = false;
    bool tables_pending_ = false;         // the last speculative build's tables are still on the device only (fetch_build_tables)
    int tables_bins_ = 0, tables_comps_ = 0, tables_set_ = 0;
    int fetch_build_tables();
    int spec_bins_guess_ = 0, spec_lanes_ = 0;
    unsigned long long gate_expected_ = 0, gate_serial_ = 0;      // what the gates of the solve in flight compare the fingerprint word with
    bool time_sweeps_ = false, timed_sweeps_ = false;   // bench(): the event pair brackets the sweeps instead of the whole solve
    const unsigned* sw_cleared_ = nullptr;          // the static-tag table launch_fingerprint's kernel cleared for the solve in flight
    size_t sw_cleared_words_ = 0;
    unsigned replays_ = 0;                          // solves repeated because nothing could be committed (stale or spoiled schedule)
    phx_step_hook step_hook_ = nullptr;  // bench(): called with phase 1 between a step's local preparation and its sweeps
    void* step_hook_user_ = nullptr;
    int step_hook_step_ = 0;
    int hash_slot_ = 1;                  // which of the two control sets (control word = fingerprint accumulator, island counters, stamps) the solve in flight uses
    bool island_clears_next_ = false;    // no hash pass runs in front of this solve: its island launch clears the next solve's control set
    bool spec_hash_ran_ = false;         // the hash pass ran in front of the speculative build that is being settled
    bool have_hash_ = false;             // raw_fingerprint_ is the cached schedule's topology hash (a rebuild without a hash pass leaves none)
    int isl_mode_ = 0;                   // ISL_GATED / ISL_VERIFY of the solve being queued (island_view.h)
    unsigned isl_nexpect_ = 0, solve_epoch_ = 0;
    int cu_count_ = 0;
    unsigned long long* fp_wanted_ = nullptr;   // set by ensure_schedule: deliver the fingerprint with the builder's first readback
    int ncomp_guess_ = 0;               // component count of the previous device build (sizes its readback)
    DevBuf<unsigned long long> hash_;
    Readback rb_;                        // pinned staging for every small device->host readback of this handle
    // staging for the host-pointer entry point
    DevBuf<phx_rigid_body> st_bodies_;
    DevBuf<phx_contact_point> st_cps_;
    DevBuf<phx_contact_joint> st_joints_;
    // bench snapshots (resident form)
    DevBuf<float4> snap_vel_, snap_dvel_, snap_mpos_;
    DevBuf<phx_contact_joint> snap_joints_;
    long long lite_builds_ = 0, full_builds_ = 0;                                  // (statistics: device builds from the manifolds / from the joints)
    BodyView bench_last_b_{}; phx_contact_joint* bench_last_j_ = nullptr; int bench_last_nb_ = 0, bench_last_nj_ = 0;      // bench_checksum
    // bench_stage(): private copies of the input, one per timed step, made BEFORE the timed region (the input of every step is
    // then resident in HBM — in the resident layout — when the clock starts, and no restore copy runs between the solves)
    DevBuf<float4> stage_vel_, stage_dvel_, stage_mpos_;      // (mpos is read-only: one copy serves every step)
    DevBuf<phx_contact_joint> stage_joints_;
    const void* staged_src_bodies_ = nullptr; const void* staged_src_joints_ = nullptr;
    int staged_nb_ = 0, staged_nj_ = 0, staged_steps_ = 0;
    bool in_bench_loop_ = false;                      // bench()'s timed loop: unverified solves are chained on purpose
    bool bench_trusted_ = false;                      // bench(): the timed solves run on copies of the input the schedule was built (and verified) for

    Schedule sched_;
    std::vector<int> h_static_slot_;
    int nstatic_ = 0, nb_ = 0, nj_ = 0;
    int max_iters_ = 0;
    phx_solve_stats stats_{};
    bool stats_pending_ = false, have_solve_ = false;
    int last_ci_ = 0, last_pi_ = 0, last_island_mode_ = 0;
    long long sweep_launches_ = 0, graph_sweep_launches_ = 0, schedule_version_ = 0;
    hipGraphExec_t graph_[3] = {nullptr, nullptr, nullptr};
    GraphKey graph_key_, last_key_;
    bool half_state_ = false, reuse_schedule_ = true;
    bool owns_stream_ = true;
    int shard_ = 0, shard_count_ = 1;    // this handle sweeps groups g with g % shard_count_ == shard_ (the HBM group counts as group lds_groups)
    // (the deal of the groups to the ranks: exchange.h exchange_partition; shard_count_ == 1 owns everything)
    bool owns_hbm_group() const { return sched_.has_hbm_group() && (shard_count_ == 1 || (partition_ok() && owner_host_[(size_t)sched_.lds_groups] == shard_)); }
    bool partition_ok() const { return partition_version_ == schedule_version_ && partition_shards_ == shard_count_ && partition_shard_ == shard_ && owner_host_.size() > (size_t)sched_.lds_groups; }
    int ensure_partition();
    std::vector<int> owner_host_;                  // per group (entry lds_groups: the HBM group): the rank that solves it
    DevBuf<int> grp_owner_, grp_mine_;             // the same on the device; the LDS groups this rank owns, ascending
    int mine_count_ = 0, partition_shards_ = 0, partition_shard_ = -1;
    long long partition_version_ = -1;
    // a solve enqueued on the cached schedule before its fingerprint was checked; verified in synchronize()
    struct Pending { bool active = false; int count = 0; Arrays arrays; const void* cps = nullptr; void* joints = nullptr; int nb = 0, ncp = 0, nj = 0; phx_config cfg{};
                     int mode = 0; unsigned nexpect = 0; } pending_;
    void register_pending(const Arrays& a, int nb, const void* cps, int ncp, void* joints, int nj, const phx_config& cfg, bool repeat);
    bool same_as_pending(const Arrays& a, int nb, const void* cps, int ncp, const void* joints, int nj, const phx_config& cfg) const;
    int ncp_ = 0;
    unsigned long long raw_fingerprint_ = 0;
    // exchange of an island-sharded solve (exchange.h)
    int ensure_exchange_layout();
    std::vector<int> grp_body_count_;              // per LDS group: entries of its body table (both builders fill it)
    Comm* comm_ = nullptr;
    unsigned* xch_send_ = nullptr;                 // caller-owned: one segment
    unsigned* xch_recv_ = nullptr;                 // caller-owned: shard_count segments, rank-major
    long long xch_cap_words_ = 0, xch_seg_words_ = 0, xch_layout_version_ = -1;
    int xch_layout_shards_ = 0;
    unsigned xch_serial_ = 0;
    DevBuf<long long> xch_off_;
    std::vector<long long> xch_off_host_;
    DevBuf<int> xch_err_;
    bool trace_islands_ = false, trace_waves_ = true;
    std::vector<hipEvent_t> bench_events_;
};

} // namespace phx
