// world.hip — World: the whole step of ref: src/World.cpp:19-37 on HBM-resident arrays.
//
// Bodies live on the device as the RESIDENT structure of arrays (body_view.h: velocities, displacing velocities,
// {invMass, invInertia, pos}, frame, AABB, size — every kernel of the step reads the 16-byte granules it needs, coalesced);
// the reference's 128-byte records exist at the C-ABI edge only (construction, phx_world_get_bodies).  Manifolds / contact
// points / joints live on the device in the reference's POD layouts; the host keeps only their counts.  Per step: IntegrateVelocity (kernel) -> UpdateBroadphase + UpdatePairs (DeviceBroadphase) ->
// manifold creation + UpdateManifolds (narrowphase kernel) -> PackManifolds -> RefreshContactJoints (scan-based,
// order-preserving, world_kernels.h) -> SolveJoints (DeviceSolver) -> IntegratePosition (kernel).  What crosses
// PCIe per step is a handful of counters (new pairs, dead manifolds, new / dead joints) and, only when the joint
// topology changed, the body-pair list the host schedule builder needs.
#include "reslab.h"
#include "handles.h"
#include "device_scan.h"
#include "world_kernels.h"

#include <algorithm>
#include <chrono>
#include <cmath>

namespace phx {

static inline int wgrid(int n) { return std::max(1, std::min(div_up(n, 256), 4096)); }
constexpr int PRELABEL_EARLY_MANIFOLDS = 400000;      // worlds from this size on queue the side stream's share of the schedule rebuild before the joint match (refresh_contact_joints)

class World {
public:
    explicit World(int device) : broadphase_h(device), solver_h(device), device_(device), broadphase_(broadphase_h.impl), solver_(solver_h.impl) {}
    ~World();
    int init();

    int add_body(float px, float py, float angle, float sx, float sy);
    int set_static(int body);
    int set_inverse_mass(int body, float inv_mass, float inv_inertia);
    int synchronize();
    int update(float dt, const phx_config& cfg);
    int pre_solve(float dt);
    int finish_step(float dt, const phx_config& cfg);
    int step_begin(float dt, const phx_config& cfg, size_t* segment_bytes);
    int step_end(float dt);
    int set_comm(Comm* c);
    int step_sharded(float dt, const phx_config& cfg);
    int step_sharded_inner(float dt, const phx_config& cfg);
    int check_exchange();
    int x_extent(float out[2]);
    hipStream_t stream() const { return stream_; }
    int download_bodies(phx_rigid_body* out, int cap);
    int download_manifolds(phx_manifold* out, int cap);
    int download_contact_points(phx_contact_point* out, int cap);
    int download_joints(phx_contact_joint* out, int cap);
    int set_state(const phx_rigid_body* bodies, int body_count, const phx_manifold* manifolds, int manifold_count,
                  const phx_contact_point* cps, int cp_count, const phx_contact_joint* joints, int joint_count);
    int get_slab_state(const long long* global_index, int count, SlabState* out);      // this world as a re-slab hands it over (reslab.h)

    int nb() const { return (int)host_bodies_.size(); }
    int nm = 0, nj = 0;
    float gravity = 0.f;
    int shard = 0, shard_count = 1;
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool phase_timing = false;          // per-phase host timers: one stream synchronisation per phase, off by default
    int dropped_points = 0;
    long long deferred_packs = 0, deferred_pack_retries = 0;      // PackManifolds counts settled with the joint counts / of those, the ones that found dead manifolds
    phx_broadphase broadphase_h;     // world-owned handles, also reachable through phx_world_broadphase()/phx_world_solver()
    phx_solver solver_h;
    DeviceBroadphase& broadphase() { return broadphase_; }
    DeviceSolver& solver() { return solver_; }

private:
    int sync_bodies_to_device();
    int refresh_records();               // resident arrays -> the 128-byte records (getters)
    WorldBodies resident() const { return WorldBodies{BodyView{vel_.p, dvel_.p, mpos_.p}, frame_.p, aabb_.p, size_.p}; }
    int update_pairs();
    bool fuse_velocity_ = false; float step_dt_ = 0.f;      // IntegrateVelocity rides on the broadphase's key build (update_pairs)
    int fresh_manifolds_ = 0;           // pairs UpdatePairs found this step: their manifolds are created by UpdateManifolds' kernel
    int manifolds_updated_ = 0;         // manifolds [0, this) were updated during update_pairs' round trip (update_manifolds() does the rest)
    int update_manifolds();
    int finish_pack(int dead, int dropped);
    int pack_manifolds();
    int refresh_contact_joints();
    int solve(const phx_config& cfg, bool settle);
    int solve_and_integrate(float dt, const phx_config& cfg);
    int scratch_for(int n);

    int device_;
    DeviceBroadphase& broadphase_;
    DeviceSolver& solver_;
    hipStream_t stream_ = nullptr;
    std::vector<phx_rigid_body> host_bodies_;     // construction-time staging; the device copy is authoritative after upload
    bool bodies_dirty_ = false;
    DevBuf<phx_rigid_body> d_bodies_;             // the records: uploaded once, refreshed from the resident arrays only when a getter asks
    DevBuf<float4> vel_, dvel_, mpos_, frame_, aabb_;      // the resident body state (body_view.h)
    DevBuf<float2> size_;
    bool records_stale_ = false;                  // a step has run since the records were last refreshed
    DevBuf<float4> accel_; bool accel_pending_ = false;      // accelerations the uploaded records came with: consumed by the next IntegrateVelocity
    DevBuf<phx_manifold> d_manifolds_;
    DevBuf<phx_contact_point> d_cps_;
    DevBuf<phx_contact_joint> d_joints_;
    DevBuf<unsigned> flags_, dead_flags_, counters_, joint_seen_;      // joint_seen_[j] == joint_epoch_: a contact point re-attached joint j this step
    DevBuf<unsigned> pack_flags_;       // dead-manifold flags, then their scan (a table of its own: the joint match reuses flags_ while the pack may still be pending)
    // PackManifolds' count is not waited for when the previous step found no dead manifold (the steady state of a stack): the
    // joint match is queued behind the scan on the assumption that nothing dies, both counts come back in ONE round trip, and
    // only if a manifold did die is the pack run then and the match repeated (a new epoch makes the first one void)
    bool pack_pending_ = false, expect_no_dead_manifolds_ = false;
    unsigned joint_epoch_ = 0;
    bool packed_this_step_ = false;          // PackManifolds moved manifolds in this step
    const MailRide* old_ride_ = nullptr;     // update_pairs: the post the old manifolds' UpdateManifolds launch carries
    ScanScratch scan_tiles_;     // counters_: [0] new joints, [1] dead joints, [2] dead manifolds, [3] dropped points
    Readback rb_;
    bool joints_changed_ = true;          // joints were created / destroyed (or a body's mass changed) since the last solve
    DevBuf<int> mover_pos_;
    DevBuf<uint2> erased_;
    // native transport (comm.hip): the world owns — and grows — the exchange buffers
    Comm* comm_ = nullptr;
    DevBuf<unsigned> xch_send_, xch_recv_;
    size_t xch_capacity_ = 0;
    unsigned sharded_steps_ = 0;
    size_t agreed_seg_ = 0;                  // segment bytes of the last host-read agreement (0: none): step_sharded
    int ensure_exchange_capacity(size_t bytes);
};

World::~World()
{
    if (hipSetDevice(device_) != hipSuccess) return;
    if (stream_) (void)hipStreamSynchronize(stream_);
    // (device buffers are DevBuf members: freed with the object)
    // (stream_ belongs to the broadphase handle, which is destroyed after this body and after the solver handle)
}

int World::init()
{
    PHX_TRY(broadphase_.init());
    PHX_TRY(solver_.init());
    // one stream for the whole step: broadphase, narrowphase, contact cache, solver and integrators are ordered by it, the
    // host waits only where it needs a count to size the next launch
    stream_ = broadphase_.stream();
    PHX_TRY(solver_.adopt_stream(stream_));
    PHX_TRY(counters_.reserve(8));
    return PHX_OK;
}

// ref: World.cpp:11-17, RigidBody.h:15-36, Coords2.h:10-17 (cos/sin resolve to the double overloads)
int World::add_body(float px, float py, float angle, float sx, float sy)
{
    if (!bodies_dirty_ && !host_bodies_.empty() && d_bodies_.p) PHX_TRY(download_bodies(host_bodies_.data(), (int)host_bodies_.size()));
    phx_rigid_body b;
    std::memset(&b, 0, sizeof b);
    const float pi = 3.141592f;
    const float quarter = angle + pi / 2.0f;
    b.xvector.x = (float)std::cos((double)angle);   b.xvector.y = (float)std::sin((double)angle);
    b.yvector.x = (float)std::cos((double)quarter); b.yvector.y = (float)std::sin((double)quarter);
    b.pos.x = px; b.pos.y = py;
    b.geom_size.x = sx; b.geom_size.y = sy;
    const float density = 1e-5f;
    const float mass = density * (sx * sy);
    const float inertia = mass * (sx * sx + sy * sy);
    b.inv_mass = 1.0f / mass;
    b.inv_inertia = 1.0f / inertia;
    update_geom(b);
    b.index = (uint32_t)host_bodies_.size();
    host_bodies_.push_back(b);
    bodies_dirty_ = true;
    return (int)host_bodies_.size() - 1;
}

int World::set_static(int body)
{
    if (!bodies_dirty_ && d_bodies_.p) PHX_TRY(download_bodies(host_bodies_.data(), (int)host_bodies_.size()));
    host_bodies_[body].inv_mass = 0.f;
    host_bodies_[body].inv_inertia = 0.f;
    bodies_dirty_ = true;
    return PHX_OK;
}

int World::set_inverse_mass(int body, float inv_mass, float inv_inertia)
{
    if (!bodies_dirty_ && d_bodies_.p) PHX_TRY(download_bodies(host_bodies_.data(), (int)host_bodies_.size()));
    host_bodies_[body].inv_mass = inv_mass;
    host_bodies_[body].inv_inertia = inv_inertia;
    bodies_dirty_ = true;
    joints_changed_ = true;              // which bodies are static is part of the schedule's topology
    return PHX_OK;
}

int World::sync_bodies_to_device()
{
    if (!bodies_dirty_) return PHX_OK;
    PHX_TRY(use_device(device_));
    PHX_TRY(d_bodies_.reserve(std::max<size_t>(host_bodies_.size(), 1)));
    const size_t n = std::max<size_t>(host_bodies_.size(), 1);
    PHX_TRY(vel_.reserve(n)); PHX_TRY(dvel_.reserve(n)); PHX_TRY(mpos_.reserve(n)); PHX_TRY(frame_.reserve(n)); PHX_TRY(aabb_.reserve(n)); PHX_TRY(size_.reserve(n));
    if (!host_bodies_.empty()) {
        PHX_HIP(hipMemcpyAsync(d_bodies_.p, host_bodies_.data(), host_bodies_.size() * sizeof(phx_rigid_body), hipMemcpyHostToDevice, stream_));
        hipLaunchKernelGGL(k_bodies_to_world, dim3(wgrid(nb())), dim3(256), 0, stream_, (const phx_rigid_body*)d_bodies_.p, nb(), resident());
        PHX_HIP(hipGetLastError());
        // records that come with accelerations (a handed-over state; AddBody leaves none): the next IntegrateVelocity applies them,
        // once, like the reference (ref: World.cpp:44-53)
        accel_pending_ = false;
        for (const phx_rigid_body& b : host_bodies_) if (b.acceleration.x != 0.f || b.acceleration.y != 0.f || b.angular_acceleration != 0.f) { accel_pending_ = true; break; }
        if (accel_pending_) {
            std::vector<float4> acc(host_bodies_.size());
            for (size_t i = 0; i < acc.size(); ++i) acc[i] = make_float4(host_bodies_[i].acceleration.x, host_bodies_[i].acceleration.y, host_bodies_[i].angular_acceleration, 0.f);
            PHX_TRY(accel_.reserve(acc.size()));
            PHX_HIP(hipMemcpyAsync(accel_.p, acc.data(), acc.size() * sizeof(float4), hipMemcpyHostToDevice, stream_));
            PHX_HIP(hipStreamSynchronize(stream_));      // (`acc` is a local)
        }
    }
    PHX_HIP(hipStreamSynchronize(stream_));
    bodies_dirty_ = false;
    records_stale_ = false;
    return PHX_OK;
}

int World::refresh_records()
{
    if (!records_stale_ || !nb()) return PHX_OK;
    hipLaunchKernelGGL(k_world_to_bodies, dim3(wgrid(nb())), dim3(256), 0, stream_, resident(), nb(), d_bodies_.p);
    PHX_HIP(hipGetLastError());
    records_stale_ = false;
    return PHX_OK;
}

int World::scratch_for(int n)
{
    PHX_TRY(flags_.reserve((size_t)n + 2));
    PHX_TRY(mover_pos_.reserve((size_t)n + 2));
    return PHX_OK;
}

int World::update_pairs()                                                   // ref: Collider.cpp:251-345
{
    const DeviceBroadphase::StepPrologue prologue{gravity, step_dt_, counters_.p, vel_.p, mpos_.p, accel_pending_ ? (const float4*)accel_.p : nullptr};
    // UpdateManifolds of the manifolds that exist already needs nothing from this update (positions, rotations and AABBs do not
    // change in it): it is queued behind the mailbox post of the new-pair count and runs while that round trip is under way — the
    // GPU used to idle through it.  The new pairs' manifolds follow in update_manifolds().  (Not with per-phase timing: the phases
    // would overlap.)
    manifolds_updated_ = 0;
    packed_this_step_ = false;
    const std::function<int()> old_manifolds = [this]() -> int {
        if (!nm) return PHX_OK;
        PHX_TRY(scratch_for(nm));
        PHX_TRY(pack_flags_.reserve((size_t)nm + 2));
        hipLaunchKernelGGL(k_update_manifolds, dim3(wgrid(nm)), dim3(256), 0, stream_, d_manifolds_.p, nm, resident(), d_cps_.p,
                           pack_flags_.p, reinterpret_cast<int*>(counters_.p + 3), nm, (const uint2*)nullptr, 0, counters_.p + 2, old_ride_ ? *old_ride_ : MailRide{});
        PHX_HIP(hipGetLastError());
        manifolds_updated_ = nm;
        return PHX_OK;
    };
    // (that launch also CARRIES the post of the new-pair count: its first workgroup posts before it updates — a dispatch fewer per step)
    const MailCarrier old_manifolds_with = [&](const MailRide* ride) -> int { old_ride_ = ride; const int st = old_manifolds(); old_ride_ = nullptr; return st; };
    PHX_TRY(broadphase_.update_resident(aabb_.p, nb(), fuse_velocity_ ? &prologue : nullptr, phase_timing ? nullptr : &old_manifolds,
                                        phase_timing || !nm ? nullptr : &old_manifolds_with));      // same stream; returns once the new-pair count is known
    fuse_velocity_ = false;
    if (accel_pending_) {                  // IntegrateVelocity of this step has consumed the uploaded accelerations (ref: World.cpp:50, 53)
        accel_pending_ = false;
        if (nb()) hipLaunchKernelGGL(k_clear_accelerations, dim3(wgrid(nb())), dim3(256), 0, stream_, d_bodies_.p, nb());
        PHX_HIP(hipGetLastError());
    }
    const int fresh = broadphase_.new_pair_count();
    if (!fresh) return PHX_OK;
    PHX_TRY(d_manifolds_.reserve_keep((size_t)nm + fresh, nm, stream_));
    PHX_TRY(d_cps_.reserve_keep(2 * ((size_t)nm + fresh), 2 * (size_t)nm, stream_));
    nm += fresh;                                                            // (created by UpdateManifolds' kernel, in the lane that updates them)
    fresh_manifolds_ = fresh;
    return PHX_OK;
}

int World::update_manifolds()                                               // ref: Collider.cpp:368-377
{
    const int first = manifolds_updated_;                                   // (the old manifolds may have been updated during update_pairs' round trip)
    manifolds_updated_ = 0;
    if (nm <= first) { fresh_manifolds_ = 0; return PHX_OK; }
    PHX_TRY(scratch_for(nm));
    PHX_TRY(pack_flags_.reserve_keep((size_t)nm + 2, (size_t)first, stream_));      // (a pending pack keeps its scan in it until refresh_contact_joints settles it)
    hipLaunchKernelGGL(k_update_manifolds, dim3(wgrid(nm - first)), dim3(256), 0, stream_, d_manifolds_.p, nm, resident(), d_cps_.p,
                       pack_flags_.p, reinterpret_cast<int*>(counters_.p + 3), nm - fresh_manifolds_, broadphase_.new_pairs_device(), first, counters_.p + 2, MailRide{});
    fresh_manifolds_ = 0;
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

int World::pack_manifolds()                                                 // ref: Collider.cpp:379-416
{
    pack_pending_ = false;
    if (!nm) return PHX_OK;
    // counters_[2] holds the number of dead manifolds already (k_update_manifolds counts them); the scan of their flags — what the
    // clean-up places its movers by — runs only when there are any: at once if the last step had some, else (the bet) after
    // refresh_contact_joints' round trip has shown the count
    if (expect_no_dead_manifolds_ && !phase_timing) { pack_pending_ = true; return PHX_OK; }
    PHX_TRY(device_exclusive_scan(pack_flags_.p, nm, counters_.p + 2, scan_tiles_, stream_));
    unsigned host[2] = {0, 0};                                              // [0] dead manifolds, [1] dropped points
    PHX_TRY(rb_.add(host, counters_.p + 2, sizeof host, stream_));
    PHX_TRY(rb_.wait(stream_));
    return finish_pack((int)host[0], (int)host[1]);
}

int World::finish_pack(int dead, int dropped)
{
    dropped_points += dropped;
    expect_no_dead_manifolds_ = dead == 0;
    if (!dead) return PHX_OK;
    packed_this_step_ = true;
    PHX_TRY(erased_.reserve(dead));
    hipLaunchKernelGGL(k_compact_movers, dim3(wgrid(dead)), dim3(256), 0, stream_, (const unsigned*)pack_flags_.p, (const unsigned*)(counters_.p + 2), nm, nm, mover_pos_.p);
    hipLaunchKernelGGL(k_pack_manifolds, dim3(wgrid(nm)), dim3(256), 0, stream_, d_manifolds_.p, d_cps_.p, nm, (const unsigned*)pack_flags_.p,
                       (const unsigned*)(counters_.p + 2), (const int*)mover_pos_.p, erased_.p);
    PHX_HIP(hipGetLastError());
    nm -= dead;
    return broadphase_.erase_pairs_device(erased_.p, dead);                 // ref: Collider.cpp:391 manifoldMap.erase
}

int World::refresh_contact_joints()                                         // ref: World.cpp:72-149
{
    PHX_TRY(scratch_for(std::max(nm, nj + 2 * nm)));
    PHX_TRY(dead_flags_.reserve((size_t)nj + 2));
    const size_t seen_cap = joint_seen_.cap;
    PHX_TRY(joint_seen_.reserve((size_t)nj + 2));
    if (joint_seen_.cap != seen_cap) {                                      // new (uninitialised) table: stale stamps could alias
        PHX_HIP(hipMemsetAsync(joint_seen_.p, 0, joint_seen_.cap * sizeof(unsigned), stream_));
        joint_epoch_ = 0;
    }
    // One host round trip for the counts.  A joint is dead iff no contact point re-attached it (the match), which is
    // known before the new joints exist; the new joints are appended behind the old ones and are alive by construction.
    unsigned host[4] = {0, 0, 0, 0};                                        // [0] new joints, [1] dead joints, [2] dead manifolds, [3] dropped points
    for (;;) {
        if (++joint_epoch_ == 0) {                                          // the epoch wrapped: stale stamps could alias
            PHX_HIP(hipMemsetAsync(joint_seen_.p, 0, joint_seen_.cap * sizeof(unsigned), stream_));
            joint_epoch_ = 1;
        }
        // (the side stream's share of the rebuild is queued BEHIND the match and its scans in a small world — the host needs ~30 us for its
        //  eight launches, more than those kernels run, and the GPU would idle — but IN FRONT of them in a large one, whose match alone runs
        //  longer than that: at 1M boxes the side stream's 140 us were the critical path of the step, started 100 us after they could have)
        if (nm >= PRELABEL_EARLY_MANIFOLDS) PHX_TRY(solver_.prelabel_components((const float4*)mpos_.p, nb(), (const phx_manifold*)d_manifolds_.p, nm));
        if (nm) {
            hipLaunchKernelGGL(k_joints_match, dim3(wgrid(nm)), dim3(256), 0, stream_, (const phx_manifold*)d_manifolds_.p, nm, (const phx_contact_point*)d_cps_.p,
                               d_joints_.p, joint_seen_.p, joint_epoch_, flags_.p);
            PHX_TRY(device_exclusive_scan(flags_.p, nm, counters_.p, scan_tiles_, stream_));
        }
        if (nj) PHX_TRY(device_exclusive_scan_of(JointDeadLoad{(const unsigned*)joint_seen_.p, joint_epoch_}, dead_flags_.p, nj, counters_.p + 1, scan_tiles_, stream_));
        if (nm || nj || pack_pending_) {                                    // counters_[0 .. 3]: adjacent words, one copy
            unsigned got[4] = {0, 0, 0, 0};
            PHX_TRY(rb_.add(got, counters_.p, sizeof got, stream_));
            PHX_TRY(solver_.prelabel_components((const float4*)mpos_.p, nb(), (const phx_manifold*)d_manifolds_.p, nm));      // (once: the mark is consumed)
            PHX_TRY(rb_.wait(stream_));
            host[0] = got[0]; host[1] = got[1];
            if (pack_pending_) { host[2] = got[2]; host[3] = got[3]; }      // (else PackManifolds has settled them already)
            if (!nm) host[0] = 0;                                           // (not written this step)
            if (!nj) host[1] = 0;
        }
        if (!pack_pending_) break;
        // PackManifolds' count came back with the joints': normally 0 (that was the bet) — if not, pack now and match again
        pack_pending_ = false;
        const int nm_before = nm;
        ++deferred_packs;
        if (host[2]) {
            // the bet was lost: now the flags are scanned, and the manifolds MOVE — under the side stream, which is reading them to count
            // and bin the components: what it makes of them is void, and this step's rebuild takes its components from the joints
            PHX_TRY(solver_.cancel_prelabel());
            PHX_TRY(device_exclusive_scan(pack_flags_.p, nm, counters_.p + 2, scan_tiles_, stream_));
        }
        PHX_TRY(finish_pack((int)host[2], (int)host[3]));
        if (nm == nm_before) break;
        ++deferred_pack_retries;                                            // the bet was lost: match again under a new epoch (one more pass: nothing is pending now)
    }
    const int fresh = (int)host[0], dead = (int)host[1], old = nj;
    const int total = nj + fresh;
    if (dead) PHX_TRY(mover_pos_.reserve((size_t)total + 2));
    if (fresh) {
        joints_changed_ = true;
        PHX_TRY(d_joints_.reserve_keep((size_t)nj + fresh, nj, stream_));
        const int mover_blocks = dead ? wgrid(dead) : 0;                    // the clean-up's movers ride along (independent of the new joints)
        hipLaunchKernelGGL(k_joints_create, dim3(wgrid(nm) + mover_blocks), dim3(256), 0, stream_, (const phx_manifold*)d_manifolds_.p, nm, d_cps_.p, d_joints_.p, nj,
                           (const unsigned*)flags_.p, mover_blocks, (const unsigned*)dead_flags_.p, (const unsigned*)(counters_.p + 1), total, mover_pos_.p);
    }
    if (dead) {                                                             // cleanup (ref: World.cpp:125-143): holes take movers from the tail
        joints_changed_ = true;
        if (!fresh) hipLaunchKernelGGL(k_compact_movers, dim3(wgrid(dead)), dim3(256), 0, stream_, (const unsigned*)dead_flags_.p, (const unsigned*)(counters_.p + 1), total, old,
                                       mover_pos_.p);
        hipLaunchKernelGGL(k_joints_fill, dim3(wgrid(total)), dim3(256), 0, stream_, d_joints_.p, total, old, (const unsigned*)dead_flags_.p,
                           (const unsigned*)(counters_.p + 1), (const int*)mover_pos_.p, d_cps_.p);
    }
    nj = total - dead;
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

int World::solve(const phx_config& cfg, bool settle)                        // ref: World.cpp:34
{
    // island sharding: the solver sweeps only this rank's groups (DeviceSolver::set_shard); the other groups' bodies
    // keep their velocities here
    if (!joints_changed_) PHX_TRY(solver_.cancel_prelabel());              // (no rebuild will pick the side stream's bins up)
    PHX_TRY(solver_.solve_resident(resident().s, nb(), d_cps_.p, 2 * nm, d_joints_.p, nj, cfg, joints_changed_));
    joints_changed_ = false;
    // a solve that is still unverified (it ran speculatively on the cached schedule, or on a device-built schedule whose 'every
    // bin fits' flag has not been read) must be settled before anything consumes its result unconditionally
    return settle && solver_.has_pending() ? solver_.synchronize() : PHX_OK;
}

// IntegratePosition behind the solve.  An unverified solve is NOT waited for first: the integrator is queued behind it gated by
// the same fingerprint word (it integrates nothing if the solve committed nothing), THEN the host settles the solve — its
// round trip overlaps the island kernel and the integrator instead of idling the GPU — and repeats the integrator only if the
// solve had to be repeated.
int World::solve_and_integrate(float dt, const phx_config& cfg)
{
    { RoctxRange r("SolveJoints"); PHX_TRY(solve(cfg, false)); }
    RoctxRange r("IntegratePosition");                                      // ref: World.cpp:57-70
    const bool pending = solver_.has_pending();
    const unsigned replays = solver_.replays();
    // (what the settle reads is final when the solve's last kernel ends, so its mailbox post goes in FRONT of the integrator and the
    //  integrator is queued while the post crosses the link: the host wakes up a kernel earlier)
    bool integrated = false;
    // (the integrator also CARRIES the settle's post: its first workgroup posts before it integrates — a dispatch fewer per step)
    const MailCarrier integrate_with = [&](const MailRide* ride) -> int {
        integrated = true;
        if (nb()) hipLaunchKernelGGL(k_integrate_position, dim3(wgrid(nb())), dim3(256), 0, stream_, resident(), nb(), dt,
                                     pending ? solver_.fingerprint_word() : (const unsigned long long*)nullptr, solver_.expected_fingerprint(), ride ? *ride : MailRide{});
        PHX_HIP(hipGetLastError());
        return PHX_OK;
    };
    const std::function<int()> integrate = [&]() -> int { return integrate_with(nullptr); };
    if (!pending) PHX_TRY(integrate());
    if (pending) {
        PHX_TRY(solver_.synchronize(&integrate, nb() ? &integrate_with : nullptr));
        if (!integrated) { set_error("the settle did not queue the integrator"); return PHX_ERR_STATE; }
        if (solver_.replays() != replays && nb())
            hipLaunchKernelGGL(k_integrate_position, dim3(wgrid(nb())), dim3(256), 0, stream_, resident(), nb(), dt, (const unsigned long long*)nullptr, 0ull, MailRide{});
        PHX_HIP(hipGetLastError());
    }
    return PHX_OK;
}

int World::pre_solve(float dt)
{
    using clk = std::chrono::steady_clock;
    PHX_TRY(use_device(device_));
    PHX_TRY(sync_bodies_to_device());
    auto t = clk::now();
    auto lap = [&](int phase) { if (!phase_timing) return; (void)hipStreamSynchronize(stream_); auto n = clk::now(); phase_ms[phase] = std::chrono::duration<double, std::milli>(n - t).count(); t = n; };
    RoctxRange update_range("Update (before SolveJoints)");                 // ref: World.cpp:21
    {
        RoctxRange r("IntegrateVelocity");                                  // ref: World.cpp:39-55
        // (normally fused into the broadphase's first kernel, update_pairs(); a kernel of its own only when the phases are timed one by one)
        fuse_velocity_ = !phase_timing;
        step_dt_ = dt;
        records_stale_ = true;
        if (nb() && !fuse_velocity_) hipLaunchKernelGGL(k_integrate_velocity, dim3(wgrid(nb())), dim3(256), 0, stream_, vel_.p, (const float4*)mpos_.p, nb(), gravity, dt, counters_.p, accel_pending_ ? (const float4*)accel_.p : nullptr);
        PHX_HIP(hipGetLastError());
        lap(0);
    }
    // the device runs sort and sweep back to back; the host clock cannot split them, so the whole device broadphase
    // is booked under UpdatePairs and UpdateBroadphase reads 0 (phx_broadphase_stats has the device time)
    phase_ms[1] = 0.0;
    { RoctxRange r("UpdateBroadphase + UpdatePairs"); PHX_TRY(update_pairs()); lap(2); }            // ref: Collider.cpp:253, 288
    { RoctxRange r("UpdateManifolds"); PHX_TRY(update_manifolds()); lap(3); }                        // ref: Collider.cpp:370
    // the manifolds say which bodies hang together: the solver labels the connected components on its side stream while the joint
    // list is still being matched, extended and compacted (solver.h prelabel_components) — not with per-phase timing: the phases overlap
    // (marked here — the manifolds are final from this point of the stream on — and queued from refresh_contact_joints, once the stream has
    //  the match and its scans to run while the host queues the side stream's kernels)
    { RoctxRange r("PackManifolds"); PHX_TRY(pack_manifolds()); lap(4); }                            // ref: Collider.cpp:381
    // (marked BEHIND PackManifolds: a pack that runs here moves manifolds, and the side stream must read them where they end up —
    //  ADVICE r5; a pack that is only found necessary later, with the joint counts, voids the side stream's work: refresh_contact_joints)
    if (!phase_timing) PHX_TRY(solver_.prelabel_mark());
    { RoctxRange r("RefreshContactJoints"); PHX_TRY(refresh_contact_joints()); lap(5); }             // ref: World.cpp:74
    return PHX_OK;
}

// A sharded world (island sharding across ranks): this rank solves only its own groups, so the step has
// two halves around the caller's all-gather of the ranks' results (exchange.h).  The unsplit entry points would integrate
// the other ranks' bodies with unsolved velocities, so they refuse to run sharded.
int World::step_begin(float dt, const phx_config& cfg, size_t* segment_bytes)
{
    PHX_TRY(pre_solve(dt));
    { RoctxRange r("SolveJoints (this rank's groups)"); PHX_TRY(solve(cfg, true)); }
    RoctxRange r("Exchange: pack");
    const BodyView bodies = resident().s;
    return solver_.exchange_pack_resident(&bodies, d_joints_.p, segment_bytes);
}

int World::step_end(float dt)
{
    PHX_TRY(use_device(device_));
    { RoctxRange r("Exchange: unpack"); PHX_TRY(solver_.exchange_unpack_resident(resident().s, d_joints_.p)); }
    RoctxRange r("IntegratePosition");
    records_stale_ = true;
    if (nb()) hipLaunchKernelGGL(k_integrate_position, dim3(wgrid(nb())), dim3(256), 0, stream_, resident(), nb(), dt, (const unsigned long long*)nullptr, 0ull, MailRide{});            // ref: World.cpp:57-70
    PHX_HIP(hipGetLastError());
    return PHX_OK;
}

// ---- the sharded step with the native transport (comm.hip) -------------------------------------------------------------
int World::ensure_exchange_capacity(size_t bytes)
{
    if (bytes <= xch_capacity_ && xch_send_.p) return PHX_OK;
    PHX_TRY(solver_.synchronize());
    PHX_HIP(hipStreamSynchronize(stream_));
    const size_t cap = (std::max<size_t>(2 * bytes, 1u << 20) + 255) / 256 * 256;
    PHX_TRY(xch_send_.reserve(cap / 4));
    PHX_TRY(xch_recv_.reserve(cap / 4 * (size_t)std::max(shard_count, 1)));
    xch_capacity_ = cap;
    return solver_.set_exchange_buffers(xch_send_.p, xch_recv_.p, cap);
}

int World::set_comm(Comm* c)
{
    PHX_TRY(use_device(device_));
    comm_ = c;
    agreed_seg_ = 0;
    // from here on the stream carries collectives: every host wait on it is bounded (a peer that dies or stops stepping must end
    // this rank's step with an error, not hang it — ADVICE r5)
    const double bound = c ? Comm::timeout_s() : 0.0;
    rb_.set_timeout(bound); solver_.set_wait_timeout(bound); broadphase_.set_wait_timeout(bound);
    if (!c) return PHX_OK;
    shard = c->rank(); shard_count = c->size();
    PHX_TRY(solver_.set_shard(shard, shard_count));
    xch_capacity_ = 0;                                                      // (the recv buffer is sized by the rank count)
    return ensure_exchange_capacity(1u << 20);
}

// (Any error return forgets the agreed segment size, on every rank alike — the failed one included: the next step's ranks then all
//  wait for their agreement again, whatever this step left behind.)
int World::step_sharded(float dt, const phx_config& cfg)
{
    const int st = step_sharded_inner(dt, cfg);
    if (st != PHX_OK) agreed_seg_ = 0;
    return st;
}

int World::step_sharded_inner(float dt, const phx_config& cfg)
{
    if (!comm_) { set_error("phx_world_step_sharded needs a communicator (phx_world_set_comm)"); return PHX_ERR_STATE; }
    size_t seg = 0;
    int st = step_begin(dt, cfg, &seg);
    if (st == PHX_ERR_CAPACITY) {
        // the layout is a pure function of the schedule: every rank finds the same segment size too big and grows the same way;
        // the solve is done, only the pack is repeated
        PHX_TRY(ensure_exchange_capacity(solver_.exchange_segment_bytes()));
        const BodyView bodies = resident().s;
        st = solver_.exchange_pack_resident(&bodies, d_joints_.p, &seg);
    }
    // Before the collective the ranks agree on {did anybody fail in this step, the segment size} — one 16-byte all-reduce (max) QUEUED
    // on the stream in every step.  Its result is waited for on the host only by a rank that NEEDS it: one whose segment size is
    // not the size of the last agreement (the layout pads segments to 64 KB, exchange.h: in a running world that is rare), or one
    // that failed.  The steady step is therefore stream-ordered end to end again (round 4 waited for the agreement in every step);
    // healthy replicas all wait or all do not (the size is a pure function of the schedule), and:
    //   * a rank that failed learns its healthy peers' size from the result, hence whether THEY waited: if they did, they have seen
    //     its failure and nobody enters the all-gather; if they did not, they are in the all-gather already and it joins them with a
    //     header-only segment of the agreed size that carries its status (the unpack and check_exchange report it) — either way the
    //     step ends on every rank with an error and no collective is left half-entered;
    //   * sizes that differ among healthy ranks (replicas that diverged) end the step on every rank that waited.
    const std::string why = st != PHX_OK ? last_error() : std::string();
    const bool failed = st != PHX_OK;
    const int posted = comm_->agree_post(failed ? 1 : 0, failed ? 0ll : (long long)seg, stream_);
    if (posted != PHX_OK && posted != PHX_ERR_INVALID) return posted;      // (the collective itself could not be queued)
    if (failed || posted == PHX_ERR_INVALID) {
        const std::string cause = failed ? why : std::string(last_error());
        int worst = 0; long long lo = 0, hi = 0;
        PHX_TRY(comm_->agree_read(&worst, &lo, &hi, stream_));
        const bool peers_waited = lo != hi || hi != (long long)agreed_seg_;      // (no healthy peer at all: lo = 2^31 - 1, hi = 0)
        if (!peers_waited && agreed_seg_ > 0) {
            size_t hdr = 0;
            (void)solver_.exchange_pack_resident(nullptr, nullptr, &hdr, 1);      // header only: magic, serial, status
            (void)comm_->all_gather(xch_send_.p, xch_recv_.p, agreed_seg_, stream_);
        }
        set_error("%s", cause.c_str());
        return failed ? st : PHX_ERR_INVALID;
    }
    if (seg != agreed_seg_) {
        int worst = 0; long long lo = 0, hi = 0;
        agreed_seg_ = 0;
        PHX_TRY(comm_->agree_read(&worst, &lo, &hi, stream_));
        if (worst) { set_error("island-sharded step: a peer failed before the exchange (this rank's half of the step is done, nothing was exchanged)"); return PHX_ERR_STATE; }
        if (lo != hi) { set_error("island-sharded step: the ranks' segment sizes differ (%lld .. %lld bytes): the replicas diverged", lo, hi); return PHX_ERR_STATE; }
        agreed_seg_ = seg;
    }
    { RoctxRange r("Exchange: all-gather (RCCL)"); PHX_TRY(comm_->all_gather(xch_send_.p, xch_recv_.p, seg, stream_)); }
    PHX_TRY(step_end(dt));
    if ((++sharded_steps_ & 15u) == 0) PHX_TRY(check_exchange());
    return PHX_OK;
}

// the peers' headers of every exchange so far (exchange.h) and the communicator's asynchronous error state
int World::check_exchange()
{
    int bits = 0;
    PHX_TRY(solver_.exchange_status(&bits));
    if (bits) { set_error("island-sharded exchange inconsistent: PHX_XCH bits %d (1 peer error, 2 step serial, 4 topology, 8 segment never written)", bits); return PHX_ERR_STATE; }
    if (comm_) {
        int e = 0;
        PHX_TRY(comm_->async_error(&e));
        if (e) { set_error("RCCL asynchronous error %d", e); return PHX_ERR_STATE; }
    }
    return PHX_OK;
}

// [min x, max x] over the AABBs of the dynamic bodies (an ownership-sharded caller checks it against its slab, dist.py SlabWorld)
int World::x_extent(float out[2])
{
    PHX_TRY(use_device(device_));
    PHX_TRY(sync_bodies_to_device());
    PHX_TRY(flags_.reserve(4));
    const unsigned init[2] = {0xFFFFFFFFu, 0u};
    PHX_HIP(hipMemcpyAsync(flags_.p, init, sizeof init, hipMemcpyHostToDevice, stream_));
    PHX_HIP(hipStreamSynchronize(stream_));                                 // (`init` is on the stack)
    if (nb()) hipLaunchKernelGGL(k_x_extent, dim3(wgrid(nb())), dim3(256), 0, stream_, resident(), nb(), flags_.p);
    PHX_HIP(hipGetLastError());
    unsigned keys[2] = {0, 0};
    PHX_TRY(rb_.add(keys, flags_.p, sizeof keys, stream_));
    PHX_TRY(rb_.wait(stream_));
    auto unkey = [](unsigned k) { const unsigned u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k; float f; std::memcpy(&f, &u, 4); return f; };
    if (keys[0] == 0xFFFFFFFFu) { out[0] = 0.f; out[1] = 0.f; }            // no dynamic body
    else { out[0] = unkey(keys[0]); out[1] = unkey(keys[1]); }
    return PHX_OK;
}

int World::finish_step(float dt, const phx_config& cfg)
{
    using clk = std::chrono::steady_clock;
    if (shard_count > 1) { set_error("a sharded world steps through phx_world_step_begin / all-gather / phx_world_step_end"); return PHX_ERR_STATE; }
    PHX_TRY(use_device(device_));
    PHX_TRY(sync_bodies_to_device());
    auto t = clk::now();
    auto lap = [&](int phase) { if (!phase_timing) return; (void)hipStreamSynchronize(stream_); auto n = clk::now(); phase_ms[phase] = std::chrono::duration<double, std::milli>(n - t).count(); t = n; };
    if (phase_timing) {                                                     // (per-phase host timing: settle the solve before the integrator)
        { RoctxRange r("SolveJoints"); PHX_TRY(solve(cfg, true)); lap(6); }
        RoctxRange r("IntegratePosition");
        if (nb()) hipLaunchKernelGGL(k_integrate_position, dim3(wgrid(nb())), dim3(256), 0, stream_, resident(), nb(), dt, (const unsigned long long*)nullptr, 0ull, MailRide{});
        PHX_HIP(hipGetLastError());
    } else PHX_TRY(solve_and_integrate(dt, cfg));
    records_stale_ = true;
    lap(7);
    return PHX_OK;
}

int World::synchronize()
{
    PHX_TRY(use_device(device_));
    PHX_TRY(solver_.synchronize());
    if (comm_) return comm_->wait_stream(stream_, "World::synchronize");      // (bounded: the stream carries collectives)
    PHX_HIP(hipStreamSynchronize(stream_));
    return PHX_OK;
}

int World::update(float dt, const phx_config& cfg)
{
    if (shard_count > 1) { set_error("a sharded world steps through phx_world_step_begin / all-gather / phx_world_step_end"); return PHX_ERR_STATE; }
    PHX_TRY(pre_solve(dt));
    return finish_step(dt, cfg);
}

#define PHX_DOWNLOAD(fn, type, buf, count_expr)                                                                    \
    int World::fn(type* out, int cap)                                                                                \
    {                                                                                                                \
        const int n = (count_expr);                                                                                  \
        if (cap < n) { set_error(#fn ": buffer too small"); return PHX_ERR_CAPACITY; }                             \
        PHX_TRY(use_device(device_));                                                                                \
        PHX_HIP(hipStreamSynchronize(stream_));                                                                      \
        if (n) PHX_HIP(hipMemcpy(out, buf.p, (size_t)n * sizeof(type), hipMemcpyDeviceToHost));                     \
        return PHX_OK;                                                                                               \
    }
PHX_DOWNLOAD(download_manifolds, phx_manifold, d_manifolds_, nm)
PHX_DOWNLOAD(download_contact_points, phx_contact_point, d_cps_, 2 * nm)
PHX_DOWNLOAD(download_joints, phx_contact_joint, d_joints_, nj)

// Restore (or hand over) a whole world: what the four getters return, put back.  The contact cache is the state the reference
// carries from step to step (ref: Collider.h:57-58 manifolds + manifoldMap, World.h:33 contactJoints with their warm-start
// impulses, ContactPoint::solverIndex linking the two); everything else a step needs is rebuilt by the step.  Used by
// checkpoint / resume and by the hand-over of bodies between the ranks of an ownership-sharded world.
int World::set_state(const phx_rigid_body* bodies, int body_count, const phx_manifold* manifolds, int manifold_count,
                     const phx_contact_point* cps, int cp_count, const phx_contact_joint* joints, int joint_count)
{
    PHX_TRY(use_device(device_));
    PHX_REQUIRE(body_count >= 0 && manifold_count >= 0 && joint_count >= 0 && cp_count == 2 * manifold_count, "bad counts (two contact-point slots per manifold)");
    PHX_REQUIRE((body_count == 0 || bodies) && (manifold_count == 0 || (manifolds && cps)) && (joint_count == 0 || joints), "null array");
    PHX_REQUIRE(shard_count == 1 && !comm_, "a sharded world cannot be restored");
    // the invariants the step's kernels rely on
    for (int i = 0; i < manifold_count; ++i) {
        const phx_manifold& m = manifolds[i];
        PHX_REQUIRE((unsigned)m.body1 < (unsigned)body_count && (unsigned)m.body2 < (unsigned)body_count, "manifold: body index out of range");
        PHX_REQUIRE(m.point_index == 2 * i && m.point_count >= 0 && m.point_count <= 2, "manifold: its contact points are slots 2i, 2i + 1");
    }
    for (int j = 0; j < joint_count; ++j) {
        const phx_contact_joint& q = joints[j];
        PHX_REQUIRE((unsigned)q.contact_point_index < (unsigned)cp_count, "joint: contact point out of range");
        const phx_manifold& m = manifolds[q.contact_point_index / 2];
        PHX_REQUIRE(q.body1 == m.body1 && q.body2 == m.body2, "joint: bodies differ from its manifold's");
        PHX_REQUIRE(cps[q.contact_point_index].solver_index == j, "joint: its contact point does not point back at it");
    }
    PHX_TRY(synchronize());
    host_bodies_.assign(bodies, bodies + body_count);
    bodies_dirty_ = true;
    PHX_TRY(sync_bodies_to_device());
    nm = manifold_count; nj = joint_count;
    PHX_TRY(d_manifolds_.reserve(std::max<size_t>(nm, 1))); PHX_TRY(d_cps_.reserve(std::max<size_t>(2 * (size_t)nm, 1))); PHX_TRY(d_joints_.reserve(std::max<size_t>(nj, 1)));
    if (nm) {
        PHX_HIP(hipMemcpyAsync(d_manifolds_.p, manifolds, (size_t)nm * sizeof(phx_manifold), hipMemcpyHostToDevice, stream_));
        PHX_HIP(hipMemcpyAsync(d_cps_.p, cps, 2 * (size_t)nm * sizeof(phx_contact_point), hipMemcpyHostToDevice, stream_));
    }
    if (nj) PHX_HIP(hipMemcpyAsync(d_joints_.p, joints, (size_t)nj * sizeof(phx_contact_joint), hipMemcpyHostToDevice, stream_));
    PHX_HIP(hipStreamSynchronize(stream_));
    std::vector<uint2> pairs((size_t)nm);
    for (int i = 0; i < nm; ++i) pairs[i] = make_uint2((unsigned)manifolds[i].body1, (unsigned)manifolds[i].body2);
    PHX_TRY(broadphase_.reset_pairs(pairs.data(), nm));
    // nothing of the old world's bookkeeping survives
    joints_changed_ = true; pack_pending_ = false; expect_no_dead_manifolds_ = false; fresh_manifolds_ = 0; manifolds_updated_ = 0; fuse_velocity_ = false;
    if (joint_seen_.p) { PHX_HIP(hipMemsetAsync(joint_seen_.p, 0, joint_seen_.cap * sizeof(unsigned), stream_)); }
    joint_epoch_ = 0;
    PHX_HIP(hipStreamSynchronize(stream_));
    return PHX_OK;
}

int World::get_slab_state(const long long* global_index, int count, SlabState* out)
{
    if (count != nb()) { set_error("re-slab: %d scene indices for a world of %d bodies", count, nb()); return PHX_ERR_INVALID; }
    out->global_index.assign(global_index, global_index + count);
    out->bodies.resize((size_t)nb()); out->manifolds.resize((size_t)nm); out->cps.resize(2 * (size_t)nm); out->joints.resize((size_t)nj);
    PHX_TRY(download_bodies(out->bodies.data(), nb()));
    PHX_TRY(download_manifolds(out->manifolds.data(), nm));
    PHX_TRY(download_contact_points(out->cps.data(), 2 * nm));
    PHX_TRY(download_joints(out->joints.data(), nj));
    return PHX_OK;
}

int World::download_bodies(phx_rigid_body* out, int cap)
{
    const int n = nb();
    if (cap < n) { set_error("download_bodies: buffer too small"); return PHX_ERR_CAPACITY; }
    if (bodies_dirty_ || !d_bodies_.p) { if (n && out != host_bodies_.data()) std::memcpy(out, host_bodies_.data(), (size_t)n * sizeof(phx_rigid_body)); return PHX_OK; }
    PHX_TRY(use_device(device_));
    PHX_TRY(solver_.synchronize());                                         // (an unverified solve is settled before its results are read)
    PHX_TRY(refresh_records());
    PHX_HIP(hipStreamSynchronize(stream_));
    if (n) PHX_HIP(hipMemcpy(out, d_bodies_.p, (size_t)n * sizeof(phx_rigid_body), hipMemcpyDeviceToHost));
    return PHX_OK;
}

} // namespace phx

// ---- C ABI ------------------------------------------------------------------------------------------------
struct phx_world {
    phx::World impl;
    explicit phx_world(int d) : impl(d) {}
};

extern "C" {

int phx_world_create(phx_world** out, int device)
{
    PHX_REQUIRE(out, "null out");
    *out = nullptr;
    PHX_TRY(phx::use_device(device));
    phx_world* w = new (std::nothrow) phx_world(device);
    PHX_REQUIRE(w, "out of host memory");
    int st = w->impl.init();
    if (st != PHX_OK) { delete w; return st; }
    *out = w;
    return PHX_OK;
}

void phx_world_destroy(phx_world* w) { delete w; }

int phx_world_add_body(phx_world* w, float px, float py, float angle, float hx, float hy)
{
    PHX_REQUIRE(w, "null handle");
    PHX_REQUIRE(hx > 0.f && hy > 0.f, "half sizes must be positive");
    return w->impl.add_body(px, py, angle, hx, hy);
}

int phx_world_set_body_static(phx_world* w, int32_t body)
{
    PHX_REQUIRE(w, "null handle");
    PHX_REQUIRE(body >= 0 && body < w->impl.nb(), "body index out of range");
    return w->impl.set_static(body);
}

int phx_world_set_body_inverse_mass(phx_world* w, int32_t body, float inv_mass, float inv_inertia)
{
    PHX_REQUIRE(w, "null handle");
    PHX_REQUIRE(body >= 0 && body < w->impl.nb(), "body index out of range");
    PHX_REQUIRE(inv_mass >= 0.f && inv_inertia >= 0.f, "inverse mass / inertia must not be negative");
    return w->impl.set_inverse_mass(body, inv_mass, inv_inertia);
}

int phx_world_set_gravity(phx_world* w, float g) { PHX_REQUIRE(w, "null handle"); w->impl.gravity = g; return PHX_OK; }

int phx_world_set_shard(phx_world* w, int32_t shard, int32_t count)
{
    PHX_REQUIRE(w, "null handle");
    PHX_REQUIRE(count >= 1 && shard >= 0 && shard < count, "bad shard");
    w->impl.shard = shard; w->impl.shard_count = count;
    PHX_TRY(w->impl.solver().set_shard(shard, count));
    return PHX_OK;
}

int phx_world_update(phx_world* w, float dt, const phx_config* cfg)
{
    PHX_REQUIRE(w && cfg, "null handle / config");
    return w->impl.update(dt, *cfg);
}

int phx_world_pre_solve(phx_world* w, float dt)
{
    PHX_REQUIRE(w, "null handle");
    return w->impl.pre_solve(dt);
}

int phx_world_finish_step(phx_world* w, float dt, const phx_config* cfg)
{
    PHX_REQUIRE(w && cfg, "null handle / config");
    return w->impl.finish_step(dt, *cfg);
}

int phx_world_step_begin(phx_world* w, float dt, const phx_config* cfg, size_t* segment_bytes)
{
    PHX_REQUIRE(w && cfg, "null handle / config");
    return w->impl.step_begin(dt, *cfg, segment_bytes);
}

int phx_world_step_end(phx_world* w, float dt)
{
    PHX_REQUIRE(w, "null handle");
    return w->impl.step_end(dt);
}

void* phx_world_stream(phx_world* w) { return w ? (void*)w->impl.stream() : nullptr; }

int phx_world_set_comm(phx_world* w, phx_comm* c)
{
    PHX_REQUIRE(w, "null handle");
    return w->impl.set_comm(c ? &c->impl : nullptr);
}

int phx_world_step_sharded(phx_world* w, float dt, const phx_config* cfg)
{
    PHX_REQUIRE(w && cfg, "null handle / config");
    return w->impl.step_sharded(dt, *cfg);
}

int phx_world_check_exchange(phx_world* w)
{
    PHX_REQUIRE(w, "null handle");
    return w->impl.check_exchange();
}

int phx_world_counts(phx_world* w, int32_t* nb, int32_t* nm, int32_t* ncp, int32_t* nj)
{
    PHX_REQUIRE(w, "null handle");
    if (nb) *nb = w->impl.nb();
    if (nm) *nm = w->impl.nm;
    if (ncp) *ncp = 2 * w->impl.nm;
    if (nj) *nj = w->impl.nj;
    return PHX_OK;
}

int phx_world_get_bodies(phx_world* w, phx_rigid_body* out, int32_t cap) { PHX_REQUIRE(w && out, "null handle / buffer"); return w->impl.download_bodies(out, cap); }
int phx_world_get_manifolds(phx_world* w, phx_manifold* out, int32_t cap) { PHX_REQUIRE(w && out, "null handle / buffer"); return w->impl.download_manifolds(out, cap); }
int phx_world_get_contact_points(phx_world* w, phx_contact_point* out, int32_t cap) { PHX_REQUIRE(w && out, "null handle / buffer"); return w->impl.download_contact_points(out, cap); }
int phx_world_get_joints(phx_world* w, phx_contact_joint* out, int32_t cap) { PHX_REQUIRE(w && out, "null handle / buffer"); return w->impl.download_joints(out, cap); }

int phx_world_set_state(phx_world* w, const phx_rigid_body* bodies, int32_t body_count, const phx_manifold* manifolds, int32_t manifold_count,
                        const phx_contact_point* contact_points, int32_t contact_point_count, const phx_contact_joint* joints, int32_t joint_count)
{
    PHX_REQUIRE(w, "null handle");
    return w->impl.set_state(bodies, body_count, manifolds, manifold_count, contact_points, contact_point_count, joints, joint_count);
}

// ---- re-slab (reslab.hip) ----
static int slab_transport(const phx_slab_transport* t, phx::World& world, phx::SlabTransport* out)
{
    out->rank = 0; out->size = 1; out->stream = world.stream();
    if (!t) return PHX_OK;
    PHX_REQUIRE(t->size >= 1 && t->rank >= 0 && t->rank < t->size, "bad rank / size");
    out->rank = t->rank; out->size = t->size; out->user = t->user;
    out->gather_fn = t->all_gather; out->max_fn = reinterpret_cast<int (*)(void*, long long*)>(t->all_reduce_max);
    if (t->comm) {
        PHX_REQUIRE(t->comm->impl.size() == t->size && t->comm->impl.rank() == t->rank, "the communicator's rank / size differ from the transport's");
        out->comm = &t->comm->impl;
    }
    return PHX_OK;
}

int phx_world_reslab_intervals(phx_world* w, const int64_t* global_index, int32_t body_count, int64_t* gi, double* lo, double* hi, int32_t cap, int32_t* count)
{
    PHX_REQUIRE(w && global_index && count, "null handle / arguments");
    phx::SlabState st;
    PHX_TRY(w->impl.get_slab_state(reinterpret_cast<const long long*>(global_index), body_count, &st));
    std::vector<long long> g; std::vector<double> l, h;
    PHX_TRY(phx::reslab_intervals(st, g, l, h));
    *count = (int32_t)g.size();
    if ((int)g.size() > cap) { phx::set_error("re-slab: %zu dynamic bodies, room for %d", g.size(), cap); return PHX_ERR_CAPACITY; }
    PHX_REQUIRE(g.empty() || (gi && lo && hi), "null output");
    for (size_t k = 0; k < g.size(); ++k) { gi[k] = g[k]; lo[k] = l[k]; hi[k] = h[k]; }
    return PHX_OK;
}

int phx_reslab_plan(int64_t* gi, double* lo, double* hi, int32_t n, int32_t nranks, double margin, int32_t* owner, double* bounds)
{
    PHX_REQUIRE(n >= 0 && nranks >= 1 && bounds && (n == 0 || (gi && lo && hi && owner)), "bad arguments");
    std::vector<long long> g(gi, gi + n); std::vector<double> l(lo, lo + n), h(hi, hi + n), b;
    std::vector<int> o;
    phx::reslab_plan(g, l, h, nranks, margin, o, b);
    for (int k = 0; k < n; ++k) { gi[k] = g[(size_t)k]; lo[k] = l[(size_t)k]; hi[k] = h[(size_t)k]; owner[k] = o[(size_t)k]; }
    for (int r = 0; r < 2 * nranks; ++r) bounds[r] = b[(size_t)r];
    return PHX_OK;
}

int phx_world_reslab(phx_world* w, const phx_slab_transport* transport, int64_t* global_index, int32_t capacity, int32_t* body_count, int32_t scene_size, double margin,
                     double bounds[2], int32_t* moved)
{
    PHX_REQUIRE(w && global_index && body_count && bounds && moved, "null handle / arguments");
    PHX_REQUIRE(scene_size >= *body_count && capacity >= *body_count, "bad sizes");
    phx::SlabTransport tp;
    PHX_TRY(slab_transport(transport, w->impl, &tp));
    phx::SlabState st;
    PHX_TRY(w->impl.get_slab_state(reinterpret_cast<const long long*>(global_index), *body_count, &st));
    int mv = 0;
    PHX_TRY(phx::reslab(tp, st, scene_size, margin, bounds, &mv));
    *moved = mv;
    if (!mv) return PHX_OK;                                   // nobody changes owner: the world stays as it is, only its slab's bounds are new
    if ((int)st.bodies.size() > capacity) { phx::set_error("re-slab: the new slab holds %zu bodies, room for %d scene indices", st.bodies.size(), capacity); return PHX_ERR_CAPACITY; }
    PHX_TRY(w->impl.set_state(st.bodies.data(), (int)st.bodies.size(), st.manifolds.data(), (int)st.manifolds.size(), st.cps.data(), (int)st.cps.size(),
                              st.joints.data(), (int)st.joints.size()));
    for (size_t k = 0; k < st.global_index.size(); ++k) global_index[k] = st.global_index[k];
    *body_count = (int32_t)st.bodies.size();
    return PHX_OK;
}

int phx_world_get_solve_stats(phx_world* w, phx_solve_stats* out) { PHX_REQUIRE(w, "null handle"); return w->impl.solver().get_stats(out); }
int phx_world_get_broadphase_stats(phx_world* w, phx_broadphase_stats* out) { PHX_REQUIRE(w, "null handle"); return w->impl.broadphase().get_stats(out); }

phx_solver* phx_world_solver(phx_world* w) { return w ? &w->impl.solver_h : nullptr; }
phx_broadphase* phx_world_broadphase(phx_world* w) { return w ? &w->impl.broadphase_h : nullptr; }

int phx_world_synchronize(phx_world* w)
{
    PHX_REQUIRE(w, "null handle");
    return w->impl.synchronize();
}

int phx_world_debug_counters(phx_world* w, int64_t out4[4])
{
    PHX_REQUIRE(w && out4, "null handle / buffer");
    out4[0] = w->impl.deferred_packs; out4[1] = w->impl.deferred_pack_retries;
    out4[2] = (int64_t)w->impl.solver().replays(); out4[3] = (int64_t)w->impl.dropped_points;
    return PHX_OK;
}

int phx_world_build_counts(phx_world* w, int64_t out2[2])
{
    PHX_REQUIRE(w && out2, "null handle / buffer");
    w->impl.solver().build_counts(out2);
    return PHX_OK;
}

int phx_world_x_extent(phx_world* w, float out2[2])
{
    PHX_REQUIRE(w && out2, "null handle / buffer");
    return w->impl.x_extent(out2);
}

int phx_world_set_phase_timing(phx_world* w, int32_t on)
{
    PHX_REQUIRE(w, "null handle");
    w->impl.phase_timing = on != 0;
    return PHX_OK;
}

int phx_world_get_phase_ms(phx_world* w, double out8[8])
{
    PHX_REQUIRE(w && out8, "null handle / buffer");
    for (int i = 0; i < 8; ++i) out8[i] = w->impl.phase_ms[i];
    return PHX_OK;
}

} // extern "C"
