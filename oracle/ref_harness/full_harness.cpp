// full_harness.cpp — extern "C" driver over the REAL reference World / Collider / Solver (the hot path's .cpp files).
//
// Test infrastructure.  NOT buildable in this container today: /root/reference/src/{World,Solver,Collider}.cpp include
// "microprofile.h" (base/Parallel.h:7, Collider.cpp:6, World.cpp:4), an un-vendored submodule (.gitmodules: src/microprofile),
// and no stand-in is written for it.  oracle/Makefile's `ref_full` target builds this file together with those sources AS THEY
// LIE, straight from $(REF)/src, the day $(REF)/src/microprofile/microprofile.h exists, into oracle/_ref/libphyx_ref_full.so
// (git-ignored; it is a container-only checker and never travels).  tests/golden/make_reference_goldens.py then dumps the
// fixtures SURVEY.md §8(c) lists and tests/test_reference_goldens.py (skipped until they exist) pins the oracle on them.
// This translation unit itself only needs the reference's headers, so its syntax is checked here today (`make -C oracle
// ref_full_syntax`).
//
// Every step of World::Update (ref: World.cpp:19-37) is reachable through public members, so the harness can stop between
// the stages and hand out the solver's inputs, the grouping, and the outputs.  workers = 0 => deterministic.
#include <cstddef>
#include <cstdint>
#include <cstring>

#include "World.h"
#include "Configuration.h"
#include "base/WorkQueue.h"

namespace {
struct Harness {
    World world;
    WorkQueue queue;
    Harness() : queue(0) {}
};

Configuration make_config(int solve_mode, int island_mode, int contact_iters, int penetration_iters)
{
    Configuration c;
    c.solveMode = (Configuration::SolveMode)solve_mode;
    c.islandMode = (Configuration::IslandMode)island_mode;
    c.contactIterationsCount = contact_iters;
    c.penetrationIterationsCount = penetration_iters;
    return c;
}

template <typename T>
int copy_out(const AlignedArray<T>& a, void* out, int cap_elements)
{
    if (out && cap_elements >= a.size && a.size > 0) memcpy(out, a.data, (size_t)a.size * sizeof(T));
    return a.size;
}
} // namespace

extern "C" {

void* reff_world_create(float gravity)
{
    Harness* h = new Harness();
    h->world.gravity = gravity;
    return h;
}

void reff_world_destroy(void* p) { delete static_cast<Harness*>(p); }

// World::AddBody (ref: World.cpp:11-17); is_static: main.cpp:91-93 (invMass = invInertia = 0)
int reff_world_add_body(void* p, float px, float py, float angle, float sx, float sy, int is_static)
{
    Harness* h = static_cast<Harness*>(p);
    RigidBody* b = h->world.AddBody(Coords2f(Vector2f(px, py), angle), Vector2f(sx, sy));
    if (is_static) { b->invMass = 0.f; b->invInertia = 0.f; }
    return (int)b->index;
}

// the whole step (ref: World.cpp:19-37)
void reff_world_update(void* p, float dt, int solve_mode, int island_mode, int contact_iters, int penetration_iters)
{
    Harness* h = static_cast<Harness*>(p);
    h->world.Update(h->queue, dt, make_config(solve_mode, island_mode, contact_iters, penetration_iters));
}

// everything of World::Update that precedes Solver::SolveJoints (ref: World.cpp:25-32)
void reff_world_pre_solve(void* p, float dt)
{
    Harness* h = static_cast<Harness*>(p);
    World& w = h->world;
    w.IntegrateVelocity(h->queue, dt);
    w.collider.UpdateBroadphase(w.bodies.data, w.bodies.size);
    w.collider.UpdatePairs(h->queue, w.bodies.data, w.bodies.size);
    w.collider.UpdateManifolds(h->queue, w.bodies.data);
    w.collider.PackManifolds(w.bodies.data);
    w.RefreshContactJoints();
}

// Solver::SolveJoints alone (ref: World.cpp:34)
void reff_world_solve(void* p, int solve_mode, int island_mode, int contact_iters, int penetration_iters)
{
    Harness* h = static_cast<Harness*>(p);
    World& w = h->world;
    w.solver.SolveJoints(h->queue, w.bodies.data, w.bodies.size, w.collider.contactPoints.data, make_config(solve_mode, island_mode, contact_iters, penetration_iters));
}

void reff_world_integrate_position(void* p, float dt)
{
    Harness* h = static_cast<Harness*>(p);
    h->world.IntegratePosition(h->queue, dt);
}

// state getters: return the element count; copy when `out` holds at least that many elements
int reff_bodies(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.bodies, out, cap); }
int reff_manifolds(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.collider.manifolds, out, cap); }
int reff_contact_points(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.collider.contactPoints, out, cap); }
int reff_joints(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.solver.contactJoints, out, cap); }
int reff_joint_index(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.solver.joint_index, out, cap); }
int reff_island_offset(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.solver.island_offset, out, cap); }
int reff_island_size(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.solver.island_size, out, cap); }
int reff_broadphase_entries(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.collider.broadphase, out, cap); }
int reff_broadphase_sorted(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.collider.broadphaseSort[1], out, cap); }
int reff_solve_bodies_impulse(void* p, void* out, int cap) { return copy_out(static_cast<Harness*>(p)->world.solver.solveBodiesImpulse, out, cap); }
// the refreshed / solved joint blocks (ContactJointPacked<N>: 35 words per joint, AoSoA of N lanes)
int reff_joint_packed(void* p, int n, void* out, int cap)
{
    Solver& s = static_cast<Harness*>(p)->world.solver;
    return n == 8 ? copy_out(s.joint_packed8, out, cap) : n == 4 ? copy_out(s.joint_packed4, out, cap) : copy_out(s.joint_packed1, out, cap);
}
void reff_island_stats(void* p, int* count, int* max_size)
{
    Solver& s = static_cast<Harness*>(p)->world.solver;
    *count = s.islandCount; *max_size = s.islandMaxSize;
}

} // extern "C"
