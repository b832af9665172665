// body_view.h — the RESIDENT layout of body state in HBM: structure of arrays, one 16-byte granule per
// body and array, so that every kernel of the step reads exactly the fields it needs with coalesced 16-byte loads.
//
// The reference keeps a 128-byte AoS RigidBody (ref: src/RigidBody.h:12-57) and stages a solver-side copy on every call
// (PrepareBodies / FinishBodies, ref: Solver.cpp:456-494: SolveBody {velocity, angularVelocity, lastIteration} and
// SolveBodyParams {invMass, invInertia, coords}).  Here that staged form IS the resident form: the World keeps bodies as
// the arrays below for the whole simulation, the solver reads and writes them in place, and the 128-byte records exist
// only at the C-ABI edge (phx_world_get_bodies, phx_solver_solve*: converted by the two kernels at the bottom).
//
//   vel[b]   = {velocity.x, velocity.y, angularVelocity, 0}                          solver in/out, integrators
//   dvel[b]  = {displacingVelocity.x, .y, displacingAngularVelocity, 0}              solver in/out, IntegratePosition
//   mpos[b]  = {invMass, invInertia, pos.x, pos.y}                                   solver in (ref SolveBodyParams), narrowphase
//   (World only)  frame[b] = {xVector.x, xVector.y, yVector.x, yVector.y},  aabb[b] = {min.x, min.y, max.x, max.y},
//                 size[b]  = {geom.size.x, geom.size.y}
#pragma once

#include "common.h"

namespace phx {

struct BodyView {
    float4* vel;
    float4* dvel;
    float4* mpos;
};

// everything the World keeps per body: the solver's view + frame, AABB and half size
struct WorldBodies {
    BodyView s;
    float4* frame;      // {xVector.x, xVector.y, yVector.x, yVector.y}
    float4* aabb;       // {min.x, min.y, max.x, max.y}
    float2* size;       // geom.size
};

// 16-byte non-temporal store (results that nobody re-reads before the kernel ends: they should not wait in L2 for the
// end-of-kernel write-back)
typedef float f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_nt(float4* p, float x, float y, float z, float w)
{
    f4v v; v.x = x; v.y = y; v.z = z; v.w = w;
    __builtin_nontemporal_store(v, reinterpret_cast<f4v*>(p));
}

// ---- the C-ABI edge: 128-byte records <-> resident arrays (solver fields only) ------------------------------------
// PrepareBodies (ref: Solver.cpp:456-480) for callers that hand over the reference's records
__attribute__((unused)) static __global__ void __launch_bounds__(256) k_bodies_to_view(const phx_rigid_body* __restrict__ bodies, int n, BodyView out)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const phx_rigid_body& b = bodies[i];
        out.vel[i] = make_float4(b.velocity.x, b.velocity.y, b.angular_velocity, 0.f);
        out.dvel[i] = make_float4(b.displacing_velocity.x, b.displacing_velocity.y, b.displacing_angular_velocity, 0.f);
        out.mpos[i] = make_float4(b.inv_mass, b.inv_inertia, b.pos.x, b.pos.y);
    }
}

// FinishBodies (ref: Solver.cpp:482-494): the four velocity fields back into the records.  `gate` (may be null): the control
// word of the solve queued in front; if it differs from `expected` that solve committed nothing and neither does this.
__attribute__((unused)) static __global__ void __launch_bounds__(256) k_view_to_bodies(BodyView in, int n, phx_rigid_body* __restrict__ bodies,
                                                               const unsigned long long* __restrict__ gate, unsigned long long expected)
{
    if (gate && *gate != expected) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 a = in.vel[i], d = in.dvel[i];
        phx_rigid_body& b = bodies[i];
        b.velocity.x = a.x; b.velocity.y = a.y; b.angular_velocity = a.z;
        b.displacing_velocity.x = d.x; b.displacing_velocity.y = d.y; b.displacing_angular_velocity = d.z;
    }
}

} // namespace phx
