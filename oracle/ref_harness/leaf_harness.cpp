// leaf_harness.cpp — thin extern "C" shims over the REAL reference headers.
//
// Test infrastructure.  Built only where /root/reference exists (this container), by
// oracle/Makefile, straight from the headers where they lie (-I/root/reference/src); nothing of
// the reference is copied into the repo and no stand-in header is supplied.  Output:
// oracle/_ref/libphyx_ref_leaf.so (git-ignored, travels to the GPU box as a prebuilt file).
//
// Only the header-only part of the reference is reachable this way.  Solver.cpp, Collider.cpp and
// World.cpp include "microprofile.h" (an un-vendored submodule) and are therefore unbuildable
// here; see oracle/phx_oracle.h for the resulting pinned / unpinned split.
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <cstdio>
#include <cmath>
#include <utility>

#include "RigidBody.h"        // Vector2.h, Coords2.h, AABB2.h, Geom.h
#include "Manifold.h"
#include "Joints.h"
#include "Collider.h"         // BroadphaseSortEntry/BroadphaseEntry, std::hash<pair>, DenseHashSet
#include "Solver.h"           // ContactJointPacked<N>, Solver::SolveBody/SolveBodyParams layouts
#include "Configuration.h"
#include "base/RadixSort.h"
#include "base/SIMD.h"

template <int N> static int packed_offset(int field)
{
    typedef ContactJointPacked<N> P;
    switch (field) {
    case 0: return (int)offsetof(P, body1Index) / 4;
    case 1: return (int)offsetof(P, body2Index) / 4;
    case 2: return (int)offsetof(P, contactPointIndex) / 4;
    case 3: return (int)(offsetof(P, normalLimiter) + offsetof(ContactLimiterPacked<N>, normalProjector1X)) / 4;
    case 4: return (int)(offsetof(P, normalLimiter) + offsetof(ContactLimiterPacked<N>, compInvMass)) / 4;
    case 5: return (int)offsetof(P, normalLimiter_accumulatedImpulse) / 4;
    case 6: return (int)offsetof(P, normalLimiter_dstVelocity) / 4;
    case 7: return (int)offsetof(P, normalLimiter_dstDisplacingVelocity) / 4;
    case 8: return (int)offsetof(P, normalLimiter_accumulatedDisplacingImpulse) / 4;
    case 9: return (int)(offsetof(P, frictionLimiter) + offsetof(ContactLimiterPacked<N>, normalProjector1X)) / 4;
    case 10: return (int)(offsetof(P, frictionLimiter) + offsetof(ContactLimiterPacked<N>, compInvMass)) / 4;
    case 11: return (int)offsetof(P, frictionLimiter_accumulatedImpulse) / 4;
    case 12: return (int)sizeof(P) / 4;
    }
    return -1;
}

extern "C" {

// struct sizes/offsets the C-ABI and the oracle rely on
int ref_sizeof(int what)
{
    switch (what) {
    case 0: return (int)sizeof(RigidBody);
    case 1: return (int)sizeof(ContactPoint);
    case 2: return (int)sizeof(ContactJoint);
    case 3: return (int)sizeof(Manifold);
    case 4: return (int)sizeof(Collider::BroadphaseEntry);
    case 5: return (int)sizeof(Collider::BroadphaseSortEntry);
    case 6: return (int)sizeof(ContactJointPacked<1>);
    case 7: return (int)sizeof(Solver::SolveBody);
    case 8: return (int)sizeof(Solver::SolveBodyParams);
    case 9: return (int)sizeof(ContactJointPacked<8>);
    case 10: return (int)offsetof(RigidBody, velocity);
    case 11: return (int)offsetof(RigidBody, invMass);
    case 12: return (int)offsetof(RigidBody, coords);
    case 13: return (int)offsetof(RigidBody, lastIteration);
    case 14: return (int)offsetof(ContactPoint, solverIndex);
    case 15: return (int)offsetof(RigidBody, geom);
    case 16: return (int)offsetof(RigidBody, displacingVelocity);
    case 17: return (int)offsetof(RigidBody, angularVelocity);
    case 18: return (int)offsetof(RigidBody, displacingAngularVelocity);
    }
    return -1;
}

int ref_config_enum(int which)
{
    switch (which) {
    case 0: return Configuration::Solve_Scalar;
    case 1: return Configuration::Solve_SSE2;
    case 2: return Configuration::Solve_AVX2;
    case 3: return Configuration::Island_Single;
    case 4: return Configuration::Island_Multiple;
    case 5: return Configuration::Island_SingleSloppy;
    case 6: return Configuration::Island_MultipleSloppy;
    }
    return -1;
}

unsigned ref_radix_float(float v) { return radixFloat(v); }

// sorts n {value,index} entries; result copied back into e0
void ref_radix_sort3(Collider::BroadphaseSortEntry* e0, Collider::BroadphaseSortEntry* scratch, size_t n)
{
    Collider::BroadphaseSortEntry* r =
        radixSort3(e0, scratch, n, [](const Collider::BroadphaseSortEntry& e) { return e.value; });
    if (r != e0) memcpy(e0, r, n * sizeof(*e0));
}

unsigned ref_pair_hash(unsigned a, unsigned b)
{
    return (unsigned)std::hash<std::pair<unsigned, unsigned>>()(std::make_pair(a, b));
}

void ref_body_init(RigidBody* out, float px, float py, float angle, float sx, float sy, float density)
{
    memset(out, 0, sizeof(RigidBody));
    RigidBody b(Coords2f(Vector2f(px, py), angle), Vector2f(sx, sy), density);
    b.index = 0; b.lastIteration = 0; b.lastDisplacementIteration = 0;
    *out = b;
}

void ref_recompute_aabb(RigidBody* b) { b->geom.RecomputeAABB(); }

void ref_update_geom(RigidBody* b) { b->UpdateGeom(); }

void ref_rotate(float* v2, float angle)
{
    Vector2f v(v2[0], v2[1]);
    v.Rotate(angle);
    v2[0] = v.x; v2[1] = v.y;
}

void ref_coords_rotate(RigidBody* b, float angle) { b->coords.Rotate(angle); }

int ref_support_points(RigidBody* b, float ax, float ay, float* out4)
{
    Vector2f pts[2];
    pts[0] = Vector2f(0, 0); pts[1] = Vector2f(0, 0);
    int n = b->geom.GetSupportPointSet(Vector2f(ax, ay), pts);
    out4[0] = pts[0].x; out4[1] = pts[0].y; out4[2] = pts[1].x; out4[3] = pts[1].y;
    return n;
}

int ref_aabb_intersects(const RigidBody* a, const RigidBody* b) { return a->geom.aabb.Intersects(b->geom.aabb) ? 1 : 0; }

int ref_contact_equals(const ContactPoint* a, const ContactPoint* b, float tol) { return a->Equals(*b, tol) ? 1 : 0; }

void ref_contact_point_make(ContactPoint* out, float p1x, float p1y, float p2x, float p2y, float nx, float ny, RigidBody* b1, RigidBody* b2)
{
    memset(out, 0, sizeof(ContactPoint));
    ContactPoint c(Vector2f(p1x, p1y), Vector2f(p2x, p2y), Vector2f(nx, ny), b1, b2);
    out->delta1 = c.delta1; out->delta2 = c.delta2; out->normal = c.normal;
    out->isMerged = c.isMerged; out->isNewlyCreated = c.isNewlyCreated; out->solverIndex = c.solverIndex;
}

void ref_project_point_to_line(float px, float py, float qx, float qy, float nx, float ny, float dx, float dy, float* out2)
{
    Vector2f r;
    ProjectPointToLine(Vector2f(px, py), Vector2f(qx, qy), Vector2f(nx, ny), Vector2f(dx, dy), r);
    out2[0] = r.x; out2[1] = r.y;
}

// insert-only run of the pair set (tombstone-free, so the DenseHash erase defect is not exercised):
// result[i] = 1 if pairs[i] was newly inserted
void ref_pairset_insert_run(const unsigned* pairs, size_t n, unsigned char* result)
{
    DenseHashSet<std::pair<unsigned, unsigned>> set;
    for (size_t i = 0; i < n; ++i)
        result[i] = set.insert(std::make_pair(pairs[2 * i], pairs[2 * i + 1])) ? 1 : 0;
}

// scalar SIMD wrapper semantics used by the solve loops (base/SIMD_Scalar.h)
float ref_flipsign1(float x, float y) { return simd::flipsign(simd::V1f(x), simd::V1f(y)).v; }
float ref_max1(float l, float r) { return simd::max(simd::V1f(l), simd::V1f(r)).v; }
// the SSE2 / AVX2 wrappers of the same three operations (base/SIMD_SSE2.h, base/SIMD_AVX2.h), lane 0 of a splat:
// op 0 flipsign(x, y), 1 max(x, y), 2 abs(x)
float ref_simd4_lane0(int op, float x, float y)
{
    simd::V4f a = simd::V4f::one(x), b = simd::V4f::one(y);
    simd::V4f r = op == 0 ? simd::flipsign(a, b) : op == 1 ? simd::max(a, b) : simd::abs(a);
    float out[4];
    _mm_storeu_ps(out, r.v);
    return out[0];
}
#ifdef __AVX2__
float ref_simd8_lane0(int op, float x, float y)
{
    simd::V8f a = simd::V8f::one(x), b = simd::V8f::one(y);
    simd::V8f r = op == 0 ? simd::flipsign(a, b) : op == 1 ? simd::max(a, b) : simd::abs(a);
    float out[8];
    _mm256_storeu_ps(out, r.v);
    return out[0];
}
#endif

// ContactJointPacked<N> field offsets in 32-bit words (ref: Solver.h:7-45): what the 8-wide CPU baseline's pack8 mirrors.
// field: 0 body1Index 1 body2Index 2 contactPointIndex 3 normalLimiter.normalProjector1X 4 normalLimiter.compInvMass
//        5 normalLimiter_accumulatedImpulse 6 normalLimiter_dstVelocity 7 normalLimiter_dstDisplacingVelocity
//        8 normalLimiter_accumulatedDisplacingImpulse 9 frictionLimiter.normalProjector1X 10 frictionLimiter.compInvMass
//        11 frictionLimiter_accumulatedImpulse 12 sizeof
int ref_packed_offset(int n, int field) { return n == 8 ? packed_offset<8>(field) : n == 4 ? packed_offset<4>(field) : packed_offset<1>(field); }

#ifdef __AVX2__
// the gather / scatter of eight 16-byte SolveBody records the AVX2 solve loops use (ref: base/SIMD_AVX2.h:324-377):
// out32 = {lane0..7 of v0, of v1, of v2, of v3}
void ref_loadindexed4_v8(const void* base, const int* indices, unsigned stride, float* out32)
{
    simd::V8f v0, v1, v2, v3;
    simd::loadindexed4(v0, v1, v2, v3, base, indices, stride);
    _mm256_storeu_ps(out32, v0.v); _mm256_storeu_ps(out32 + 8, v1.v); _mm256_storeu_ps(out32 + 16, v2.v); _mm256_storeu_ps(out32 + 24, v3.v);
}
void ref_storeindexed4_v8(const float* in32, void* base, const int* indices, unsigned stride)
{
    simd::V8f v0, v1, v2, v3;
    v0.v = _mm256_loadu_ps(in32); v1.v = _mm256_loadu_ps(in32 + 8); v2.v = _mm256_loadu_ps(in32 + 16); v3.v = _mm256_loadu_ps(in32 + 24);
    simd::storeindexed4(v0, v1, v2, v3, base, indices, stride);
}
#endif

// insert / erase / insert run of the pair set in which no inserted key is ever already present (so the first-tombstone
// insertion defect of base/DenseHash.h:152-162 cannot create a duplicate): ops[i] = 0 insert, 1 erase; afterwards
// member[i] = contains(pairs[i]) and the return value is size()
size_t ref_pairset_mixed_run(const unsigned* pairs, const unsigned char* ops, size_t n, unsigned char* member)
{
    DenseHashSet<std::pair<unsigned, unsigned>> set;
    for (size_t i = 0; i < n; ++i) {
        const std::pair<unsigned, unsigned> key(pairs[2 * i], pairs[2 * i + 1]);
        if (ops[i] == 0) set.insert(key); else set.erase(key);
    }
    for (size_t i = 0; i < n; ++i) member[i] = set.contains(std::make_pair(pairs[2 * i], pairs[2 * i + 1])) ? 1 : 0;
    return set.size();
}

} // extern "C"
