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
__host__ __device__ inline V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
__host__ __device__ inline V2 v2(const phx_vec2& a) { return v2(a.x, a.y); }
__host__ __device__ inline V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
__host__ __device__ inline V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
__host__ __device__ inline V2 operator-(V2 a) { return v2(-a.x, -a.y); }
__host__ __device__ inline V2 operator*(V2 a, float s) { return v2(a.x * s, a.y * s); }
__host__ __device__ inline float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }           // ref: Vector2.h operator*(Vector2)
__host__ __device__ inline float sqlen(V2 a) { return a.x * a.x + a.y * a.y; }
__host__ __device__ inline V2 perp(V2 a) { return v2(-a.y, a.x); }                            // ref: Vector2.h GetPerpendicular
__host__ __device__ inline phx_vec2 pv(V2 a) { phx_vec2 r; r.x = a.x; r.y = a.y; return r; }

// what the narrowphase reads of a body
struct NpBody { V2 pos, xv, yv, size; };

// ref: Geom.h:79-85 RecomputeAABB: {min.x, min.y, max.x, max.y}
__host__ __device__ inline void geom_aabb(V2 pos, V2 xv, V2 yv, V2 size, float& minx, float& miny, float& maxx, float& maxy)
{
    const float dx = fabsf(xv.x) * size.x + fabsf(yv.x) * size.y;
    const float dy = fabsf(xv.y) * size.x + fabsf(yv.y) * size.y;
    minx = pos.x - dx; miny = pos.y - dy;
    maxx = pos.x + dx; maxy = pos.y + dy;
}

// ref: Geom.h:79-85 on a 128-byte record (uses the Geom copy of the frame, refreshed by UpdateGeom, RigidBody.h:38-42)
__host__ __device__ inline void update_geom(phx_rigid_body& b)
{
    b.geom_xvector = b.xvector; b.geom_yvector = b.yvector; b.geom_pos = b.pos;
    geom_aabb(v2(b.geom_pos), v2(b.geom_xvector), v2(b.geom_yvector), v2(b.geom_size), b.aabb_min.x, b.aabb_min.y, b.aabb_max.x, b.aabb_max.y);
}

// ref: Geom.h:66-77 with GetClippingEdge (:22-64) and GetClippingVertex (:10-20)
__host__ __device__ inline int support_points(const NpBody& b, V2 axis, V2 out[2])
{
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
        out[0] = p1 + off;
        out[1] = p2 + off;
        return 2;
    }
    const float xs = dot(xv, axis) < 0.0f ? -1.0f : 1.0f;
    const float ys = dot(yv, axis) < 0.0f ? -1.0f : 1.0f;
    out[0] = (pos + xdim * xs) + ydim * ys;
    return 1;
}

// ref: Collider.cpp:8-56 — returns false when a separating axis exists
__host__ __device__ inline bool least_penetration_axis(const NpBody& b1, const NpBody& b2, V2& axis)
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

// ref: Collider.cpp:58-92 with ContactPoint::Equals (Manifold.h:31-38)
__host__ __device__ inline void merge_point(phx_contact_point* pts, int& count, V2 p1, V2 p2, V2 n,
                                            const NpBody& b1, const NpBody& b2)
{
    const V2 d1 = p1 - b1.pos, d2 = p2 - b2.pos;                      // ref: Manifold.h:20-21
    int closest = -1;
    float bestdepth = 3.402823466e+38f;
    for (int i = 0; i < count; ++i) {
        const float s1 = sqlen(v2(pts[i].delta1) - d1), s2 = sqlen(v2(pts[i].delta2) - d2);
        if (s1 > 2.0f * 2.0f && s2 > 2.0f * 2.0f) continue;            // !Equals(col, 2.0f)
        const float depth = sqlen(d1 - v2(pts[i].delta1)) + sqlen(d2 - v2(pts[i].delta2));
        if (depth < bestdepth) { bestdepth = depth; closest = i; }
    }
    if (closest >= 0) {
        phx_contact_point& c = pts[closest];
        c.is_merged = 1; c.is_newly_created = 0;
        c.normal = pv(n); c.delta1 = pv(d1); c.delta2 = pv(d2);
    } else {
        phx_contact_point c;
        c.delta1 = pv(d1); c.delta2 = pv(d2); c.normal = pv(n);
        c.is_merged = 1; c.is_newly_created = 1; c.pad_[0] = 0; c.pad_[1] = 0; c.solver_index = -1;
        pts[count++] = c;
    }
}

// ref: Vector2.h ProjectPointToLine(point, planePoint, planeNormal, projectionDirection, out)
__host__ __device__ inline V2 project_to_line(V2 point, V2 plane_point, V2 plane_normal, V2 dir)
{
    const float mult = 1.0f / dot(dir, plane_normal);
    const float s = dot(plane_point, plane_normal) - dot(point, plane_normal);
    return point + (dir * s) * mult;
}

__host__ __device__ inline bool within_segment(V2 p, V2 a, V2 b)
{
    return dot(p - a, b - a) >= 0.0f && dot(p - b, a - b) >= 0.0f;
}

// ref: Collider.cpp:94-209
__host__ __device__ inline void generate_contacts(const NpBody& b1, const NpBody& b2, phx_contact_point* pts, int& count, V2 axis)
{
    if (dot(axis, b1.pos - b2.pos) < 0.0f) axis = -axis;
    V2 s1[2], s2[2];
    int n1 = support_points(b1, -axis, s1);
    int n2 = support_points(b2, axis, s2);
    const float tol = 2.0f;
    if (n1 == 2 && sqlen(s1[0] - s1[1]) < tol * tol) { s1[0] = (s1[0] + s1[1]) * 0.5f; n1 = 1; }
    if (n2 == 2 && sqlen(s2[0] - s2[1]) < tol * tol) { s2[0] = (s2[0] + s2[1]) * 0.5f; n2 = 1; }

    if (n1 == 1 && n2 == 1) {
        if (dot(s2[0] - s1[0], axis) >= 0.0f) merge_point(pts, count, s1[0], s2[0], axis, b1, b2);
    } else if (n1 == 1 && n2 == 2) {
        const V2 p = project_to_line(s1[0], s2[0], perp(s2[1] - s2[0]), axis);
        if (within_segment(p, s2[0], s2[1])) merge_point(pts, count, s1[0], p, axis, b1, b2);
    } else if (n1 == 2 && n2 == 1) {
        const V2 p = project_to_line(s2[0], s1[0], perp(s1[1] - s1[0]), axis);
        if (within_segment(p, s1[0], s1[1])) merge_point(pts, count, p, s2[0], axis, b1, b2);
    } else {
        V2 t1[4], t2[4];
        int tc = 0;
        const V2 nrm2 = perp(s2[1] - s2[0]);
        for (int i = 0; i < 2; ++i)
            if (dot(s1[i] - s2[0], nrm2) >= 0.0f) {
                const V2 p = project_to_line(s1[i], s2[0], nrm2, axis);
                if (within_segment(p, s2[0], s2[1])) { t1[tc] = s1[i]; t2[tc] = p; ++tc; }
            }
        const V2 nrm1 = perp(s1[1] - s1[0]);
        for (int i = 0; i < 2; ++i)
            if (dot(s2[i] - s1[0], nrm1) >= 0.0f) {
                const V2 p = project_to_line(s2[i], s1[0], nrm1, axis);
                if (within_segment(p, s1[0], s1[1])) { t1[tc] = p; t2[tc] = s2[i]; ++tc; }
            }
        if (tc == 1) merge_point(pts, count, t1[0], t2[0], axis, b1, b2);
        if (tc >= 2) {
            merge_point(pts, count, t1[0], t2[0], axis, b1, b2);
            merge_point(pts, count, t1[1], t2[1], axis, b1, b2);
        }
    }
}

// ref: Collider.cpp:211-245.  Returns true if a third merged point had to be dropped: the reference
// would write it past the manifold's two slots (only an assert guards it).
__host__ __device__ inline bool update_manifold(phx_manifold& m, const NpBody& b1, const NpBody& b2, phx_contact_point* pts)
{
    phx_contact_point np[4];
    for (int i = 0; i < m.point_count; ++i) { np[i] = pts[i]; np[i].is_merged = 0; np[i].is_newly_created = 0; }
    int count = m.point_count;
    V2 axis;
    if (least_penetration_axis(b1, b2, axis)) generate_contacts(b1, b2, np, count, axis);
    m.point_count = 0;
    bool dropped = false;
    for (int i = 0; i < count; ++i)
        if (np[i].is_merged) {
            if (m.point_count < 2) pts[m.point_count++] = np[i];
            else dropped = true;
        }
    return dropped;
}

// ref: AABB2.h:19-24 on {min.x, min.y, max.x, max.y}
__host__ __device__ inline bool aabb_intersects(const float4& a, const float4& b)
{
    if (a.x > b.z || b.x > a.z) return false;
    if (a.y > b.w || b.y > a.w) return false;
    return true;
}

} // namespace phx
