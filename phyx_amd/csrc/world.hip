// world.hip — World: step orchestration in the reference's order (ref: src/World.cpp:19-37).
//
// The two hot halves run on the device (DeviceBroadphase, DeviceSolver).  The stages between them —
// narrowphase + manifold cache (ref: Collider.cpp:368-416), joint matching (ref: World.cpp:72-149) and
// the integrators (ref: World.cpp:39-70) — are host C++ in this round (SURVEY.md §8(f) rows 1-3 are the
// next ones to move to HIP); they run on all host cores where the reference uses parallelFor.
#include "handles.h"
#include "narrowphase.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <thread>

namespace phx {

template <typename F>
static void parallel_for(int count, int grain, F&& fn)
{
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    const int workers = std::max(1, std::min(hw, count / std::max(grain, 1)));
    if (workers <= 1) { fn(0, count); return; }
    std::vector<std::thread> pool;
    const int chunk = div_up(count, workers);
    for (int w = 0; w < workers; ++w) {
        const int b = w * chunk, e = std::min(count, b + chunk);
        if (b >= e) break;
        pool.emplace_back([&fn, b, e] { fn(b, e); });
    }
    for (auto& t : pool) t.join();
}

class World {
public:
    explicit World(int device) : broadphase_h(device), solver_h(device), broadphase_(broadphase_h.impl), solver_(solver_h.impl) {}
    int init() { PHX_TRY(broadphase_.init()); return solver_.init(); }

    int add_body(float px, float py, float angle, float sx, float sy);
    int update(float dt, const phx_config& cfg);
    int pre_solve(float dt);
    int finish_step(float dt, const phx_config& cfg);

    std::vector<phx_rigid_body> bodies;
    std::vector<phx_manifold> manifolds;
    std::vector<phx_contact_point> contact_points;
    std::vector<phx_contact_joint> joints;
    float gravity = 0.f;
    int shard = 0, shard_count = 1;
    double phase_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int dropped_points = 0;
    phx_broadphase broadphase_h;     // world-owned handles, also reachable through phx_world_broadphase()/phx_world_solver()
    phx_solver solver_h;
    DeviceBroadphase& broadphase() { return broadphase_; }
    DeviceSolver& solver() { return solver_; }

private:
    void integrate_velocity(float dt);
    void integrate_position(float dt);
    int update_pairs();
    void update_manifolds();
    int pack_manifolds();
    void refresh_contact_joints();
    int solve(const phx_config& cfg);

    DeviceBroadphase& broadphase_;
    DeviceSolver& solver_;
    std::vector<uint32_t> pair_scratch_;
};

// ref: World.cpp:11-17, RigidBody.h:15-36, Coords2.h:10-17 (cos/sin resolve to the double overloads)
int World::add_body(float px, float py, float angle, float sx, float sy)
{
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
    b.index = (uint32_t)bodies.size();
    bodies.push_back(b);
    return (int)bodies.size() - 1;
}

void World::integrate_velocity(float dt)                                   // ref: World.cpp:39-55
{
    const float g = gravity;
    parallel_for((int)bodies.size(), 16384, [&](int b, int e) {
        for (int i = b; i < e; ++i) {
            phx_rigid_body& body = bodies[i];
            if (body.inv_mass > 0.0f) body.acceleration.y += g;
            body.velocity.x += body.acceleration.x * dt; body.velocity.y += body.acceleration.y * dt;
            body.acceleration.x = 0.f; body.acceleration.y = 0.f;
            body.angular_velocity += body.angular_acceleration * dt;
            body.angular_acceleration = 0.f;
        }
    });
}

static inline void rotate(phx_vec2& v, float c, float s)                   // ref: Vector2.h:48-56
{
    const V2 x = v2(v), y = perp(x);
    const V2 delta = (x * c + y * s) - x;
    v.x = v.x + delta.x; v.y = v.y + delta.y;
}

void World::integrate_position(float dt)                                   // ref: World.cpp:57-70
{
    parallel_for((int)bodies.size(), 8192, [&](int b, int e) {
        for (int i = b; i < e; ++i) {
            phx_rigid_body& body = bodies[i];
            body.pos.x += body.displacing_velocity.x + body.velocity.x * dt;
            body.pos.y += body.displacing_velocity.y + body.velocity.y * dt;
            const float ang = -(body.displacing_angular_velocity + body.angular_velocity * dt);
            const float c = (float)std::cos((double)ang), s = (float)std::sin((double)ang);
            rotate(body.xvector, c, s);
            rotate(body.yvector, c, s);
            body.displacing_velocity.x = 0.f; body.displacing_velocity.y = 0.f;
            body.displacing_angular_velocity = 0.f;
            update_geom(body);
        }
    });
}

int World::update_pairs()                                                   // ref: Collider.cpp:251-345
{
    int count = 0;
    PHX_TRY(broadphase_.update_host(bodies.data(), (int)bodies.size(), nullptr, 0, &count));
    pair_scratch_.resize(2 * (size_t)std::max(count, 1));
    PHX_TRY(broadphase_.get_new_pairs(pair_scratch_.data(), count, &count));
    for (int k = 0; k < count; ++k) {                                       // ref: Collider.cpp:313-316
        phx_manifold m;
        m.body1 = (int)pair_scratch_[2 * k]; m.body2 = (int)pair_scratch_[2 * k + 1];
        m.point_count = 0; m.point_index = (int)manifolds.size() * 2;
        manifolds.push_back(m);
    }
    return PHX_OK;
}

void World::update_manifolds()                                              // ref: Collider.cpp:368-377
{
    const size_t old = contact_points.size();
    contact_points.resize(manifolds.size() * 2);
    for (size_t k = old; k < contact_points.size(); ++k) { std::memset(&contact_points[k], 0, sizeof(phx_contact_point)); contact_points[k].solver_index = -1; }
    std::vector<int> dropped(64, 0);
    parallel_for((int)manifolds.size(), 2048, [&](int b, int e) {
        int d = 0;
        for (int i = b; i < e; ++i) d += update_manifold(manifolds[i], bodies.data(), contact_points.data() + manifolds[i].point_index) ? 1 : 0;
        if (d) __atomic_fetch_add(&dropped[0], d, __ATOMIC_RELAXED);
    });
    dropped_points += dropped[0];
}

int World::pack_manifolds()                                                 // ref: Collider.cpp:379-416
{
    std::vector<uint32_t> erased;
    for (size_t i = 0; i < manifolds.size();) {
        phx_manifold& m = manifolds[i];
        if (m.point_count == 0 && !aabb_intersects(bodies[m.body1], bodies[m.body2])) {
            erased.push_back((uint32_t)m.body1); erased.push_back((uint32_t)m.body2);
            const phx_manifold last = manifolds.back();
            const int slot = m.point_index;
            for (int k = 0; k < last.point_count; ++k) contact_points[slot + k] = contact_points[last.point_index + k];
            m = last;
            m.point_index = slot;
            manifolds.pop_back();
        } else ++i;
    }
    contact_points.resize(manifolds.size() * 2);
    if (!erased.empty()) PHX_TRY(broadphase_.erase_pairs(erased.data(), (int)erased.size() / 2));
    return PHX_OK;
}

void World::refresh_contact_joints()                                        // ref: World.cpp:72-149
{
    for (auto& j : joints) j.contact_point_index = -1;
    for (const phx_manifold& m : manifolds)
        for (int k = 0; k < m.point_count; ++k) {
            const int cpi = m.point_index + k;
            phx_contact_point& cp = contact_points[cpi];
            if (cp.solver_index < 0) {
                cp.solver_index = (int)joints.size();
                phx_contact_joint j;
                j.contact_point_index = cpi; j.body1 = m.body1; j.body2 = m.body2;
                j.normal_accumulated_impulse = 0.f; j.friction_accumulated_impulse = 0.f;
                joints.push_back(j);
            } else {
                joints[cp.solver_index].contact_point_index = cpi;
            }
        }
    for (size_t k = 0; k < joints.size();) {
        if (joints[k].contact_point_index < 0) { joints[k] = joints.back(); joints.pop_back(); }
        else { contact_points[joints[k].contact_point_index].solver_index = (int)k; ++k; }
    }
}

int World::solve(const phx_config& cfg)                                     // ref: World.cpp:34
{
    if (shard_count <= 1)
        return solver_.solve_host(bodies.data(), (int)bodies.size(), contact_points.data(), (int)contact_points.size(),
                                  joints.data(), (int)joints.size(), cfg);
    // island sharding: this rank solves the joints of islands whose index % shard_count == shard; islands are
    // body-disjoint (static bodies aside), so the other shards' bodies simply keep their velocities here
    const int nj = (int)joints.size(), nb = (int)bodies.size();
    std::vector<int> b1(nj), b2(nj), joint_island, island_size;
    std::vector<unsigned char> is_static(nb);
    for (int j = 0; j < nj; ++j) { b1[j] = joints[j].body1; b2[j] = joints[j].body2; }
    for (int i = 0; i < nb; ++i) is_static[i] = (bodies[i].inv_mass == 0.f && bodies[i].inv_inertia == 0.f);
    gather_islands(b1.data(), b2.data(), nj, is_static.data(), nb, joint_island, island_size);
    std::vector<phx_contact_joint> mine;
    std::vector<int> where;
    for (int j = 0; j < nj; ++j)
        if (joint_island[j] >= 0 && joint_island[j] % shard_count == shard) { mine.push_back(joints[j]); where.push_back(j); }
    PHX_TRY(solver_.solve_host(bodies.data(), nb, contact_points.data(), (int)contact_points.size(), mine.data(), (int)mine.size(), cfg));
    for (size_t k = 0; k < mine.size(); ++k) joints[where[k]] = mine[k];
    return PHX_OK;
}

int World::pre_solve(float dt)
{
    using clk = std::chrono::steady_clock;
    auto t = clk::now();
    auto lap = [&](int phase) { auto n = clk::now(); phase_ms[phase] = std::chrono::duration<double, std::milli>(n - t).count(); t = n; };
    integrate_velocity(dt); lap(0);
    // the device runs sort and sweep back to back; the host clock cannot split them, so the whole device
    // broadphase is booked under UpdatePairs and UpdateBroadphase reads 0 (phx_broadphase_stats has device_ms)
    phase_ms[1] = 0.0;
    PHX_TRY(update_pairs()); lap(2);
    update_manifolds(); lap(3);
    PHX_TRY(pack_manifolds()); lap(4);
    refresh_contact_joints(); lap(5);
    return PHX_OK;
}

int World::finish_step(float dt, const phx_config& cfg)
{
    using clk = std::chrono::steady_clock;
    auto t = clk::now();
    auto lap = [&](int phase) { auto n = clk::now(); phase_ms[phase] = std::chrono::duration<double, std::milli>(n - t).count(); t = n; };
    PHX_TRY(solve(cfg)); lap(6);
    integrate_position(dt); lap(7);
    return PHX_OK;
}

int World::update(float dt, const phx_config& cfg)
{
    PHX_TRY(pre_solve(dt));
    return finish_step(dt, cfg);
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
    PHX_REQUIRE(body >= 0 && body < (int)w->impl.bodies.size(), "body index out of range");
    w->impl.bodies[body].inv_mass = 0.f;
    w->impl.bodies[body].inv_inertia = 0.f;
    return PHX_OK;
}

int phx_world_set_gravity(phx_world* w, float g) { PHX_REQUIRE(w, "null handle"); w->impl.gravity = g; return PHX_OK; }

int phx_world_set_shard(phx_world* w, int32_t shard, int32_t count)
{
    PHX_REQUIRE(w, "null handle");
    PHX_REQUIRE(count >= 1 && shard >= 0 && shard < count, "bad shard");
    w->impl.shard = shard; w->impl.shard_count = count;
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

int phx_world_counts(phx_world* w, int32_t* nb, int32_t* nm, int32_t* ncp, int32_t* nj)
{
    PHX_REQUIRE(w, "null handle");
    if (nb) *nb = (int)w->impl.bodies.size();
    if (nm) *nm = (int)w->impl.manifolds.size();
    if (ncp) *ncp = (int)w->impl.contact_points.size();
    if (nj) *nj = (int)w->impl.joints.size();
    return PHX_OK;
}

#define PHX_WORLD_GETTER(name, member, type)                                                         \
    int name(phx_world* w, type* out, int32_t cap)                                                   \
    {                                                                                                \
        PHX_REQUIRE(w && out, "null handle / buffer");                                               \
        const size_t n = w->impl.member.size();                                                      \
        if ((size_t)cap < n) { phx::set_error(#name ": buffer too small"); return PHX_ERR_CAPACITY; } \
        if (n) std::memcpy(out, w->impl.member.data(), n * sizeof(type));                            \
        return PHX_OK;                                                                               \
    }
PHX_WORLD_GETTER(phx_world_get_bodies, bodies, phx_rigid_body)
PHX_WORLD_GETTER(phx_world_get_manifolds, manifolds, phx_manifold)
PHX_WORLD_GETTER(phx_world_get_contact_points, contact_points, phx_contact_point)
PHX_WORLD_GETTER(phx_world_get_joints, joints, phx_contact_joint)

int phx_world_get_solve_stats(phx_world* w, phx_solve_stats* out) { PHX_REQUIRE(w, "null handle"); return w->impl.solver().get_stats(out); }
int phx_world_get_broadphase_stats(phx_world* w, phx_broadphase_stats* out) { PHX_REQUIRE(w, "null handle"); return w->impl.broadphase().get_stats(out); }

phx_solver* phx_world_solver(phx_world* w) { return w ? &w->impl.solver_h : nullptr; }
phx_broadphase* phx_world_broadphase(phx_world* w) { return w ? &w->impl.broadphase_h : nullptr; }

int phx_world_get_phase_ms(phx_world* w, double out8[8])
{
    PHX_REQUIRE(w && out8, "null handle / buffer");
    for (int i = 0; i < 8; ++i) out8[i] = w->impl.phase_ms[i];
    return PHX_OK;
}

} // extern "C"
