// narrowphase.h — box/box SAT, contact generation and manifold merging as __host__ __device__ functions.
//
// This is the step between the two hot halves of the path; it restates
// ref: src/Collider.cpp:8-245 and src/Geom.h:10-85 on the fields the resident arrays hold (body_view.h): a body is
// {pos, xVector, yVector, size} here (the reference's Geom keeps a copy of the frame, refreshed by UpdateGeom,
// RigidBody.h:38-42: the same values).  The functions are
// plain IEEE float arithmetic (no libm beyond fabsf) so the host loop and a HIP kernel that call them
// produce identical bits under -ffp-contract=off.
#pragma once

#include "common.h"

namespace phx {

struct V2 { float x, y; };
__host__ __device__ __attribute__((always_inline)) inline V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
__host__ __device__ __attribute__((always_inline)) inline V2 v2(const phx_vec2& a) { return v2(a.x, a.y); }
__host__ __device__ __attribute__((always_inline)) inline V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
__host__ __device__ __attribute__((always_inline)) inline V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
__host__ __device__ __attribute__((always_inline)) inline V2 operator-(V2 a) { return v2(-a.x, -a.y); }
__host__ __device__ __attribute__((always_inline)) inline V2 operator*(V2 a, float s) { return v2(a.x * s, a.y * s); }
__host__ __device__ __attribute__((always_inline)) inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }           // ref: Vector2.h operator*(Vector2)
__host__ __device__ __attribute__((always_inline)) inline float sqlen(V2 a) { return a.x * a.x + a.y * a.y; }
__host__ __device__ __attribute__((always_inline)) inline V2 perp(V2 a) { return v2(-a.y, a.x); }                            // ref: Vector2.h GetPerpendicular
__host__ __device__ __attribute__((always_inline)) inline phx_vec2 pv(V2 a) { phx_vec2 r; r.x = a.x; r.y = a.y; return r; }

// what the narrowphase reads of a body
struct NpBody { V2 pos, xv, yv, size; };

// ref: Geom.h:79-85 RecomputeAABB: {min.x, min.y, max.x, max.y}
__host__ __device__ __attribute__((always_inline)) inline void geom_aabb(V2 pos, V2 xv, V2 yv, V2 size, float& minx, float& miny, float& maxx, float& maxy)
{
    const float dx = fabsf(xv.x) * size.x + fabsf(yv.x) * size.y;
    const float dy = fabsf(xv.y) * size.x + fabsf(yv.y) * size.y;
    minx = pos.x - dx; miny = pos.y - dy;
    maxx = pos.x + dx; maxy = pos.y + dy;
}

// ref: Geom.h:79-85 on a 128-byte record (uses the Geom copy of the frame, refreshed by UpdateGeom, RigidBody.h:38-42)
__host__ __device__ __attribute__((always_inline)) inline void update_geom(phx_rigid_body& b)
{
    b.geom_xvector = b.xvector; b.geom_yvector = b.yvector; b.geom_pos = b.pos;
    geom_aabb(v2(b.geom_pos), v2(b.geom_xvector), v2(b.geom_yvector), v2(b.geom_size), b.aabb_min.x, b.aabb_min.y, b.aabb_max.x, b.aabb_max.y);
}

// ref: Geom.h:66-77 with GetClippingEdge (:22-64) and GetClippingVertex (:10-20)
struct Support { V2 a, b; int n; };      // (returned by value: results written through references ended up behind a pointer phi, i.e. in scratch)
__host__ __device__ __attribute__((always_inline)) inline Support support_points(const NpBody& b, V2 axis)
{
    Support out;
    out.a = v2(0.f, 0.f); out.b = out.a;
    const V2 xv = b.xv, yv = b.yv, pos = b.pos;
    const V2 xdim = xv * b.size.x, ydim = yv * b.size.y;
    const float xdiff = dot(axis, xv), ydiff = dot(axis, yv);
    if (fabsf(xdiff) < 0.1f || fabsf(ydiff) < 0.1f) {
        V2 p1 = pos, p2 = pos, off = v2(0.f, 0.f);
        if (fabsf(xdiff) < fabsf(ydiff)) {
            if (dot(axis, ydim) > 0.0f) { off = off + ydim; p1 = p1 + xdim; p2 = p2 - xdim; }
            else                        { off = off - ydim; p1 = p1 - xdim; p2 = p2 + xdim; }
        } else {
            if (dot(axis, xdim) > 0.0f) { off = off + xdim; p1 = p1 - ydim; p2 = p2 + ydim; }
            else                        { off = off - xdim; p1 = p1 + ydim; p2 = p2 - ydim; }
        }
        out.a = p1 + off;
        out.b = p2 + off;
        out.n = 2;
        return out;
    }
    const float xs = dot(xv, axis) < 0.0f ? -1.0f : 1.0f;
    const float ys = dot(yv, axis) < 0.0f ? -1.0f : 1.0f;
    out.a = (pos + xdim * xs) + ydim * ys;
    out.n = 1;
    return out;
}

// ref: Collider.cpp:8-56 — returns false when a separating axis exists
__host__ __device__ __attribute__((always_inline)) inline bool least_penetration_axis(const NpBody& b1, const NpBody& b2, V2& axis)
{
    const V2 a00 = b1.xv, a01 = b1.yv, a10 = b2.xv, a11 = b2.yv;
    const V2 e0 = b1.size, e1 = b2.size;
    const V2 d = b1.pos - b2.pos;
    const float ad00 = fabsf(dot(a00, a10)), ad01 = fabsf(dot(a00, a11));
    const float r0 = e0.x + e1.x * ad00 + e1.y * ad01;
    const float d0 = fabsf(dot(a00, d)) - r0;
    if (d0 > 0) return false;
    float best = d0; V2 bestaxis = a00;
    const float ad10 = fabsf(dot(a01, a10)), ad11 = fabsf(dot(a01, a11));
    const float r1 = e0.y + e1.x * ad10 + e1.y * ad11;
    const float d1 = fabsf(dot(a01, d)) - r1;
    if (d1 > 0) return false;
    if (d1 > best) { best = d1; bestaxis = a01; }
    const float r2 = e1.x + e0.x * ad00 + e0.y * ad10;
    const float d2 = fabsf(dot(a10, d)) - r2;
    if (d2 > 0) return false;
    if (d2 > best) { best = d2; bestaxis = a10; }
    const float r3 = e1.y + e0.x * ad01 + e0.y * ad11;
    const float d3 = fabsf(dot(a11, d)) - r3;
    if (d3 > 0) return false;
    if (d3 > best) { best = d3; bestaxis = a11; }
    axis = bestaxis;
    return true;
}

// The manifold's working set of at most four contact points (two cached + two new, ref: Collider.cpp:215 newpoints[kMaxContactPoints * 2]).
// Four named members with static accessors instead of an array: an array indexed by a run-time count lives in scratch memory
// on the GPU (224 bytes per lane in round 2's kernel — a private-memory round trip for every point touched), members live in
// registers.
// (a contact point in registers: the record's four flag / padding bytes travel as one word — is_merged in bits 0-7,
//  is_newly_created in bits 8-15 — so that nothing is addressed byte-wise)
struct CpR { V2 d1, d2, n; unsigned flags; int si; };
__host__ __device__ __attribute__((always_inline)) inline CpR cp_load(const phx_contact_point& p)
{
    CpR r;
    r.d1 = v2(p.delta1); r.d2 = v2(p.delta2); r.n = v2(p.normal);
    r.flags = (unsigned)p.is_merged | ((unsigned)p.is_newly_created << 8) | ((unsigned)p.pad_[0] << 16) | ((unsigned)p.pad_[1] << 24);
    r.si = p.solver_index;
    return r;
}
__host__ __device__ __attribute__((always_inline)) inline void cp_store(phx_contact_point& p, const CpR& r)
{
    p.delta1 = pv(r.d1); p.delta2 = pv(r.d2); p.normal = pv(r.n);
    p.is_merged = (uint8_t)(r.flags & 0xFFu); p.is_newly_created = (uint8_t)((r.flags >> 8) & 0xFFu);
    p.pad_[0] = (uint8_t)((r.flags >> 16) & 0xFFu); p.pad_[1] = (uint8_t)(r.flags >> 24);
    p.solver_index = r.si;
}
struct Pts4 { CpR a, b, c, d; };
// (word by word: whole-struct copies under a run-time index keep the set in memory)
template <typename T> __host__ __device__ __attribute__((always_inline)) inline T sel4(int i, T a, T b, T c, T d) { return i == 0 ? a : (i == 1 ? b : (i == 2 ? c : d)); }
__host__ __device__ __attribute__((always_inline)) inline CpR pts_get(const Pts4& s, int i)
{
    CpR r;
    r.d1.x = sel4(i, s.a.d1.x, s.b.d1.x, s.c.d1.x, s.d.d1.x); r.d1.y = sel4(i, s.a.d1.y, s.b.d1.y, s.c.d1.y, s.d.d1.y);
    r.d2.x = sel4(i, s.a.d2.x, s.b.d2.x, s.c.d2.x, s.d.d2.x); r.d2.y = sel4(i, s.a.d2.y, s.b.d2.y, s.c.d2.y, s.d.d2.y);
    r.n.x = sel4(i, s.a.n.x, s.b.n.x, s.c.n.x, s.d.n.x); r.n.y = sel4(i, s.a.n.y, s.b.n.y, s.c.n.y, s.d.n.y);
    r.flags = sel4(i, s.a.flags, s.b.flags, s.c.flags, s.d.flags); r.si = sel4(i, s.a.si, s.b.si, s.c.si, s.d.si);
    return r;
}
__host__ __device__ __attribute__((always_inline)) inline void cp_put(CpR& dst, const CpR& v, bool on)
{
    dst.d1.x = on ? v.d1.x : dst.d1.x; dst.d1.y = on ? v.d1.y : dst.d1.y;
    dst.d2.x = on ? v.d2.x : dst.d2.x; dst.d2.y = on ? v.d2.y : dst.d2.y;
    dst.n.x = on ? v.n.x : dst.n.x; dst.n.y = on ? v.n.y : dst.n.y;
    dst.flags = on ? v.flags : dst.flags; dst.si = on ? v.si : dst.si;
}
__host__ __device__ __attribute__((always_inline)) inline void pts_set(Pts4& s, int i, const CpR& v)
{
    cp_put(s.a, v, i == 0); cp_put(s.b, v, i == 1); cp_put(s.c, v, i == 2); cp_put(s.d, v, i >= 3);
}

// ref: Collider.cpp:58-92 with ContactPoint::Equals (Manifold.h:31-38)
__host__ __device__ __attribute__((always_inline)) inline void merge_point(Pts4& pts, int& count, V2 p1, V2 p2, V2 n,
                                            const NpBody& b1, const NpBody& b2)
{
    const V2 d1 = p1 - b1.pos, d2 = p2 - b2.pos;                      // ref: Manifold.h:20-21
    int closest = -1;
    float bestdepth = 3.402823466e+38f;
    auto consider = [&](const CpR& q, int i) {                        // the reference's loop over the points, in index order
        if (i >= count) return;
        const float s1 = sqlen(q.d1 - d1), s2 = sqlen(q.d2 - d2);
        if (s1 > 2.0f * 2.0f && s2 > 2.0f * 2.0f) return;             // !Equals(col, 2.0f)
        const float depth = sqlen(d1 - q.d1) + sqlen(d2 - q.d2);
        if (depth < bestdepth) { bestdepth = depth; closest = i; }
    };
    consider(pts.a, 0); consider(pts.b, 1); consider(pts.c, 2); consider(pts.d, 3);
    if (closest >= 0) {
        CpR c = pts_get(pts, closest);
        c.flags = (c.flags & 0xFFFF0000u) | 1u;                        // isMerged = true, isNewlyCreated = false
        c.n = n; c.d1 = d1; c.d2 = d2;
        pts_set(pts, closest, c);
    } else {
        CpR c;
        c.d1 = d1; c.d2 = d2; c.n = n;
        c.flags = 1u | (1u << 8);                                      // merged, newly created
        c.si = -1;
        pts_set(pts, count, c);
        ++count;
    }
}

// ref: Vector2.h ProjectPointToLine(point, planePoint, planeNormal, projectionDirection, out)
__host__ __device__ __attribute__((always_inline)) inline V2 project_to_line(V2 point, V2 plane_point, V2 plane_normal, V2 dir)
{
    const float mult = 1.0f / dot(dir, plane_normal);
    const float s = dot(plane_point, plane_normal) - dot(point, plane_normal);
    return point + (dir * s) * mult;
}

__host__ __device__ __attribute__((always_inline)) inline bool within_segment(V2 p, V2 a, V2 b)
{
    return dot(p - a, b - a) >= 0.0f && dot(p - b, a - b) >= 0.0f;
}

// ref: Collider.cpp:94-209
__host__ __device__ __attribute__((always_inline)) inline void generate_contacts(const NpBody& b1, const NpBody& b2, Pts4& pts, int& count, V2 axis)
{
    if (dot(axis, b1.pos - b2.pos) < 0.0f) axis = -axis;
    const Support sp1 = support_points(b1, -axis), sp2 = support_points(b2, axis);
    V2 s10 = sp1.a, s11 = sp1.b, s20 = sp2.a, s21 = sp2.b;          // (named, not indexed: see Pts4)
    int n1 = sp1.n, n2 = sp2.n;
    const float tol = 2.0f;
    if (n1 == 2 && sqlen(s10 - s11) < tol * tol) { s10 = (s10 + s11) * 0.5f; n1 = 1; }
    if (n2 == 2 && sqlen(s20 - s21) < tol * tol) { s20 = (s20 + s21) * 0.5f; n2 = 1; }

    if (n1 == 1 && n2 == 1) {
        if (dot(s20 - s10, axis) >= 0.0f) merge_point(pts, count, s10, s20, axis, b1, b2);
    } else if (n1 == 1 && n2 == 2) {
        const V2 p = project_to_line(s10, s20, perp(s21 - s20), axis);
        if (within_segment(p, s20, s21)) merge_point(pts, count, s10, p, axis, b1, b2);
    } else if (n1 == 2 && n2 == 1) {
        const V2 p = project_to_line(s20, s10, perp(s11 - s10), axis);
        if (within_segment(p, s10, s11)) merge_point(pts, count, p, s20, axis, b1, b2);
    } else {
        // up to four candidate pairs in the reference's tempCol[4] (ref: Collider.cpp:160-200); only the first two found are ever
        // merged, so only those two are kept (named, not indexed: see Pts4)
        // the four candidates in the reference's order, each with its 'found' flag; the first two found are picked by selects
        const V2 nrm2 = perp(s21 - s20), nrm1 = perp(s11 - s10);
        const V2 p0 = project_to_line(s10, s20, nrm2, axis), p1 = project_to_line(s11, s20, nrm2, axis);
        const V2 p2 = project_to_line(s20, s10, nrm1, axis), p3 = project_to_line(s21, s10, nrm1, axis);
        const bool f0 = dot(s10 - s20, nrm2) >= 0.0f && within_segment(p0, s20, s21);
        const bool f1 = dot(s11 - s20, nrm2) >= 0.0f && within_segment(p1, s20, s21);
        const bool f2 = dot(s20 - s10, nrm1) >= 0.0f && within_segment(p2, s10, s11);
        const bool f3 = dot(s21 - s10, nrm1) >= 0.0f && within_segment(p3, s10, s11);
        const int tc = (f0 ? 1 : 0) + (f1 ? 1 : 0) + (f2 ? 1 : 0) + (f3 ? 1 : 0);
        const int k0 = f0 ? 0 : (f1 ? 1 : (f2 ? 2 : 3));                                   // first found
        const int k1 = k0 == 0 ? (f1 ? 1 : (f2 ? 2 : 3)) : (k0 == 1 ? (f2 ? 2 : 3) : 3);   // second found (if there is one)
        // candidate k = (point on body 1, point on body 2): {s10, p0}, {s11, p1}, {p2, s20}, {p3, s21}
        const V2 ta0 = v2(sel4(k0, s10.x, s11.x, p2.x, p3.x), sel4(k0, s10.y, s11.y, p2.y, p3.y));
        const V2 tb0 = v2(sel4(k0, p0.x, p1.x, s20.x, s21.x), sel4(k0, p0.y, p1.y, s20.y, s21.y));
        const V2 ta1 = v2(sel4(k1, s10.x, s11.x, p2.x, p3.x), sel4(k1, s10.y, s11.y, p2.y, p3.y));
        const V2 tb1 = v2(sel4(k1, p0.x, p1.x, s20.x, s21.x), sel4(k1, p0.y, p1.y, s20.y, s21.y));
        if (tc == 1) merge_point(pts, count, ta0, tb0, axis, b1, b2);
        if (tc >= 2) {
            merge_point(pts, count, ta0, tb0, axis, b1, b2);
            merge_point(pts, count, ta1, tb1, axis, b1, b2);
        }
    }
}

// ref: Collider.cpp:211-245.  Returns true if a third merged point had to be dropped: the reference
// would write it past the manifold's two slots (only an assert guards it).
__host__ __device__ __attribute__((always_inline)) inline bool update_manifold(phx_manifold& m, const NpBody& b1, const NpBody& b2, phx_contact_point* pts)
{
    Pts4 np;
    CpR blank;
    blank.d1 = blank.d2 = blank.n = v2(0.f, 0.f); blank.flags = 0u; blank.si = -1;
    np.a = np.b = np.c = np.d = blank;                                  // (slots at and beyond `count` are never looked at)
    if (m.point_count > 0) { np.a = cp_load(pts[0]); np.a.flags &= 0xFFFF0000u; }      // isMerged = isNewlyCreated = false
    if (m.point_count > 1) { np.b = cp_load(pts[1]); np.b.flags &= 0xFFFF0000u; }
    int count = m.point_count;
    V2 axis;
    if (least_penetration_axis(b1, b2, axis)) generate_contacts(b1, b2, np, count, axis);
    m.point_count = 0;
    bool dropped = false;
    auto keep = [&](const CpR& q, int i) {
        if (i >= count || !(q.flags & 0xFFu)) return;
        if (m.point_count < 2) cp_store(pts[m.point_count++], q);
        else dropped = true;
    };
    keep(np.a, 0); keep(np.b, 1); keep(np.c, 2); keep(np.d, 3);
    return dropped;
}

// ref: AABB2.h:19-24 on {min.x, min.y, max.x, max.y}
__host__ __device__ __attribute__((always_inline)) inline bool aabb_intersects(const float4& a, const float4& b)
{
    if (a.x > b.z || b.x > a.z) return false;
    if (a.y > b.w || b.y > a.w) return false;
    return true;
}

} // namespace phx
