/*
 * phx_oracle.c — CPU restatement of the zeux/phyx simulation step (see phx_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY.  Pinned/unpinned status per function is listed in phx_oracle.h;
 * in short: leaf header-only functions are pinned against the compiled reference headers
 * (oracle/_ref), everything from Solver.cpp / Collider.cpp / World.cpp is "parity unpinned"
 * (those files need the absent microprofile.h and the reference has no tests).
 *
 * All "ref:" citations are relative to /root/reference/src/.
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 */
#define _POSIX_C_SOURCE 200809L
#include "phx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                               */

static inline float dot2(phxo_vec2 a, phxo_vec2 b) { return a.x * b.x + a.y * b.y; }           /* ref: Vector2.h:133 operator* */
static inline phxo_vec2 add2(phxo_vec2 a, phxo_vec2 b) { phxo_vec2 r = {a.x + b.x, a.y + b.y}; return r; }
static inline phxo_vec2 sub2(phxo_vec2 a, phxo_vec2 b) { phxo_vec2 r = {a.x - b.x, a.y - b.y}; return r; }
static inline phxo_vec2 mul2(phxo_vec2 a, float s) { phxo_vec2 r = {a.x * s, a.y * s}; return r; }
static inline phxo_vec2 neg2(phxo_vec2 a) { phxo_vec2 r = {-a.x, -a.y}; return r; }
static inline phxo_vec2 perp2(phxo_vec2 a) { phxo_vec2 r = {-a.y, a.x}; return r; }             /* ref: Vector2.h GetPerpendicular */
static inline float sqlen2(phxo_vec2 a) { return a.x * a.x + a.y * a.y; }
static inline float maxf_ref(float l, float r) { return l > r ? l : r; }                         /* ref: base/SIMD_Scalar.h:275-278 */

/* IEEE binary16 round trip (round to nearest even), for the fp16 body-state ablation of BASELINE config 5.
 * Not part of the reference: it models what the HIP island kernel does when asked to keep body velocities in half. */
static uint16_t f32_to_f16(float f)
{
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, a = x & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (a < 0x33000001u) return (uint16_t)sign;
    if (a < 0x38800000u) {
        int e = (int)(a >> 23);
        uint32_t m = (a & 0x7fffffu) | 0x800000u;
        int shift = 126 - e;
        uint32_t r = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (r & 1u))) r++;
        return (uint16_t)(sign | r);
    }
    uint32_t r = (a - 0x38000000u) >> 13, rem = a & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
    return (uint16_t)(sign | r);
}
static float f16_to_f32(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else { int sh = 0; while (!(m & 0x400u)) { m <<= 1; ++sh; } m &= 0x3ffu; x = sign | ((uint32_t)(113 - sh) << 23) | (m << 13); }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &x, 4); return f;
}
static inline float qh(int half, float x) { return half ? f16_to_f32(f32_to_f16(x)) : x; }
float phxo_round_f16(float x) { return f16_to_f32(f32_to_f16(x)); }

#define GROW(ptr, cap, need, type)                                           \
    do {                                                                     \
        if ((size_t)(need) > (size_t)(cap)) {                                \
            size_t nc_ = (cap) ? (size_t)(cap) : 16;                         \
            while (nc_ < (size_t)(need)) nc_ += nc_ / 2 + 8;                 \
            (ptr) = (type*)realloc((ptr), nc_ * sizeof(type));               \
            (cap) = nc_;                                                     \
        }                                                                    \
    } while (0)

/* ------------------------------------------------------------------------------------------ */
/* leaf functions                                                                              */

/* ref: base/RadixSort.h:19-26 — sign-magnitude float bits -> monotone unsigned key */
uint32_t phxo_radix_float(float v)
{
    int32_t f;
    memcpy(&f, &v, 4);
    uint32_t mask = (uint32_t)(f >> 31) | 0x80000000u; /* arithmetic shift: all-ones for negatives */
    return (uint32_t)f ^ mask;
}

/* ref: base/RadixSort.h:28-95 — stable LSD radix sort, digits 11/11/10 bits, one histogram pass
 * for all three digit tables, ping-pong e0->e1->e0->e1; the sorted sequence ends up in e1. */
void phxo_radix_sort3(phxo_sort_entry* e0, phxo_sort_entry* e1, size_t n)
{
    enum { B = 2048 };
    static const int shift[3] = {0, 11, 22};
    uint32_t* hist = (uint32_t*)calloc(3 * B, sizeof(uint32_t));
    for (size_t i = 0; i < n; ++i) {
        uint32_t k = e0[i].value;
        hist[0 * B + (k & 2047u)]++;
        hist[1 * B + ((k >> 11) & 2047u)]++;
        hist[2 * B + (k >> 22)]++;
    }
    for (int d = 0; d < 3; ++d) {
        uint32_t run = 0;
        for (int b = 0; b < B; ++b) {
            uint32_t c = hist[d * B + b];
            hist[d * B + b] = run;
            run += c;
        }
    }
    phxo_sort_entry* src = e0;
    phxo_sort_entry* dst = e1;
    for (int d = 0; d < 3; ++d) {
        uint32_t* h = hist + d * B;
        for (size_t i = 0; i < n; ++i) {
            uint32_t digit = (src[i].value >> shift[d]) & (d == 2 ? 1023u : 2047u);
            dst[h[digit]++] = src[i];
        }
        phxo_sort_entry* t = src; src = dst; dst = t;
    }
    free(hist);
}

/* ref: Collider.h:10-18 */
uint32_t phxo_pair_hash(uint32_t lb, uint32_t rb) { return lb ^ (rb + 0x9e3779b9u + (lb << 6) + (lb >> 2)); }

/* ref: Geom.h:79-85 — uses Geom::coords (a copy of RigidBody::coords, RigidBody.h:38-41) */
void phxo_recompute_aabb(phxo_body* b)
{
    float dx = fabsf(b->geom_xv.x) * b->geom_size.x + fabsf(b->geom_yv.x) * b->geom_size.y;
    float dy = fabsf(b->geom_xv.y) * b->geom_size.x + fabsf(b->geom_yv.y) * b->geom_size.y;
    b->aabb_min.x = b->geom_pos.x - dx; b->aabb_min.y = b->geom_pos.y - dy;
    b->aabb_max.x = b->geom_pos.x + dx; b->aabb_max.y = b->geom_pos.y + dy;
}

static void update_geom(phxo_body* b) /* ref: RigidBody.h:38-42 */
{
    b->geom_xv = b->xv; b->geom_yv = b->yv; b->geom_pos = b->pos;
    phxo_recompute_aabb(b);
}

/* ref: RigidBody.h:15-36 + Coords2.h:10-17.  The reference calls unqualified cos()/sin() on a
 * float, which resolves to the double overload under libstdc++, then narrows to float. */
void phxo_body_init(phxo_body* b, float px, float py, float angle, float sx, float sy, float density)
{
    memset(b, 0, sizeof *b);
    float pi = 3.141592f;
    float quarter = angle + pi / 2.0f;
    b->xv.x = (float)cos((double)angle);   b->xv.y = (float)sin((double)angle);
    b->yv.x = (float)cos((double)quarter); b->yv.y = (float)sin((double)quarter);
    b->pos.x = px; b->pos.y = py;
    b->geom_size.x = sx; b->geom_size.y = sy;
    float mass = density * (sx * sy);
    float inertia = mass * (sx * sx + sy * sy);
    b->inv_mass = 1.0f / mass;
    b->inv_inertia = 1.0f / inertia;
    update_geom(b);
}

/* ref: Vector2.h:48-56 — v += v*cos(a) + perp(v)*sin(a) - v, cos/sin in double then narrowed */
void phxo_rotate_vec(phxo_vec2* v, float angle)
{
    float c = (float)cos((double)angle), s = (float)sin((double)angle);
    phxo_vec2 x = *v, y = perp2(x);
    phxo_vec2 delta = sub2(add2(mul2(x, c), mul2(y, s)), x);
    v->x = v->x + delta.x;
    v->y = v->y + delta.y;
}

/* ref: Geom.h:10-20 */
static phxo_vec2 clipping_vertex(const phxo_body* b, phxo_vec2 axis)
{
    phxo_vec2 xdim = mul2(b->geom_xv, b->geom_size.x), ydim = mul2(b->geom_yv, b->geom_size.y);
    float xs = dot2(b->geom_xv, axis) < 0.0f ? -1.0f : 1.0f;
    float ys = dot2(b->geom_yv, axis) < 0.0f ? -1.0f : 1.0f;
    return add2(add2(b->geom_pos, mul2(xdim, xs)), mul2(ydim, ys));
}

/* ref: Geom.h:22-64 */
static void clipping_edge(const phxo_body* b, phxo_vec2 axis, phxo_vec2* e1, phxo_vec2* e2)
{
    phxo_vec2 p1 = b->geom_pos, p2 = b->geom_pos, off = {0.f, 0.f};
    phxo_vec2 xdim = mul2(b->geom_xv, b->geom_size.x), ydim = mul2(b->geom_yv, b->geom_size.y);
    float xdiff = dot2(axis, b->geom_xv), ydiff = dot2(axis, b->geom_yv);
    if (fabsf(xdiff) < fabsf(ydiff)) {
        if (dot2(axis, ydim) > 0.0f) { off = add2(off, ydim); p1 = add2(p1, xdim); p2 = sub2(p2, xdim); }
        else                         { off = sub2(off, ydim); p1 = sub2(p1, xdim); p2 = add2(p2, xdim); }
    } else {
        if (dot2(axis, xdim) > 0.0f) { off = add2(off, xdim); p1 = sub2(p1, ydim); p2 = add2(p2, ydim); }
        else                         { off = sub2(off, xdim); p1 = add2(p1, ydim); p2 = sub2(p2, ydim); }
    }
    *e1 = add2(p1, off);
    *e2 = add2(p2, off);
}

/* ref: Geom.h:66-77 */
int phxo_support_points(const phxo_body* b, float ax, float ay, phxo_vec2 out[2])
{
    phxo_vec2 axis = {ax, ay};
    if (fabsf(dot2(axis, b->geom_xv)) < 0.1f || fabsf(dot2(axis, b->geom_yv)) < 0.1f) {
        clipping_edge(b, axis, &out[0], &out[1]);
        return 2;
    }
    out[0] = clipping_vertex(b, axis);
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* persistent pair set — set semantics of DenseHashSet<pair<u32,u32>> (ref: base/DenseHash.h     */
/* 208-236) without its tombstone defect: linear probing + backshift.   */

typedef struct { uint64_t* slot; size_t cap, count; } pairset;
#define PS_EMPTY 0xFFFFFFFFFFFFFFFFull

static inline uint64_t ps_key(uint32_t a, uint32_t b) { return ((uint64_t)a << 32) | b; }
static inline size_t ps_home(const pairset* s, uint64_t k) { return (size_t)phxo_pair_hash((uint32_t)(k >> 32), (uint32_t)k) * 0x9E3779B1u & (s->cap - 1); }

static void ps_rehash(pairset* s, size_t ncap)
{
    uint64_t* old = s->slot; size_t ocap = s->cap;
    s->slot = (uint64_t*)malloc(ncap * sizeof(uint64_t));
    for (size_t i = 0; i < ncap; ++i) s->slot[i] = PS_EMPTY;
    s->cap = ncap;
    for (size_t i = 0; i < ocap; ++i)
        if (old[i] != PS_EMPTY) {
            size_t p = ps_home(s, old[i]);
            while (s->slot[p] != PS_EMPTY) p = (p + 1) & (s->cap - 1);
            s->slot[p] = old[i];
        }
    free(old);
}

static int ps_insert(pairset* s, uint32_t a, uint32_t b) /* 1 if newly inserted */
{
    if (s->cap == 0) ps_rehash(s, 1024);
    if ((s->count + 1) * 2 > s->cap) ps_rehash(s, s->cap * 2);
    uint64_t k = ps_key(a, b);
    size_t p = ps_home(s, k);
    while (s->slot[p] != PS_EMPTY) {
        if (s->slot[p] == k) return 0;
        p = (p + 1) & (s->cap - 1);
    }
    s->slot[p] = k;
    s->count++;
    return 1;
}

static void ps_erase(pairset* s, uint32_t a, uint32_t b)
{
    if (!s->cap) return;
    uint64_t k = ps_key(a, b);
    size_t p = ps_home(s, k), m = s->cap - 1;
    while (s->slot[p] != PS_EMPTY && s->slot[p] != k) p = (p + 1) & m;
    if (s->slot[p] == PS_EMPTY) return;
    size_t hole = p;
    for (size_t q = (p + 1) & m; s->slot[q] != PS_EMPTY; q = (q + 1) & m) {
        size_t h = ps_home(s, s->slot[q]);
        /* can the entry at q move into the hole?  only if its home is cyclically outside (hole, q] */
        int between = hole <= q ? (h > hole && h <= q) : (h > hole || h <= q);
        if (!between) { s->slot[hole] = s->slot[q]; hole = q; }
    }
    s->slot[hole] = PS_EMPTY;
    s->count--;
}

/* ------------------------------------------------------------------------------------------ */
/* broadphase                                                                                  */

/* ref: Collider.cpp:251-284 */
void phxo_broadphase_build(const phxo_body* bodies, size_t n, phxo_sort_entry* keys_unsorted,
                           phxo_sort_entry* sorted, phxo_bp_entry* entries)
{
    phxo_sort_entry* tmp = (phxo_sort_entry*)malloc((n ? n : 1) * sizeof *tmp);
    for (size_t i = 0; i < n; ++i) {
        tmp[i].value = phxo_radix_float(bodies[i].aabb_min.x);
        tmp[i].index = (uint32_t)i;
    }
    if (keys_unsorted) memcpy(keys_unsorted, tmp, n * sizeof *tmp);
    phxo_radix_sort3(tmp, sorted, n);
    for (size_t i = 0; i < n; ++i) {
        const phxo_body* b = &bodies[sorted[i].index];
        entries[i].minx = b->aabb_min.x;
        entries[i].maxx = b->aabb_max.x;
        entries[i].centery = (b->aabb_min.y + b->aabb_max.y) * 0.5f;
        entries[i].extenty = (b->aabb_max.y - b->aabb_min.y) * 0.5f;
        entries[i].index = sorted[i].index;
    }
    free(tmp);
}

/* ref: Collider.cpp:296-318 / 347-366 minus the pair-set test: every (i<j in sorted order) with
 * minx_j <= maxx_i and |cy_j-cy_i| <= ey_i+ey_j, keyed (index_i, index_j) — NOT min/max canonical. */
size_t phxo_sweep_candidates(const phxo_bp_entry* e, size_t n, uint32_t* pairs, size_t cap, uint64_t* tests)
{
    size_t count = 0; uint64_t t = 0;
    for (size_t i = 0; i < n; ++i) {
        float maxx = e[i].maxx;
        for (size_t j = i + 1; j < n; ++j) {
            if (e[j].minx > maxx) break;
            t++;
            if (fabsf(e[j].centery - e[i].centery) <= e[i].extenty + e[j].extenty) {
                if (count < cap) { pairs[2 * count] = e[i].index; pairs[2 * count + 1] = e[j].index; }
                count++;
            }
        }
    }
    if (tests) *tests = t;
    return count;
}

/* ------------------------------------------------------------------------------------------ */
/* solver                                                                                      */

typedef struct { float vx, vy, w; int32_t tag; } sbody;               /* ref: Solver.h:95-101 SolveBody       */
typedef struct { float im, ii; phxo_vec2 pos, xv, yv; } sparams;      /* ref: Solver.h:86-93  SolveBodyParams */

typedef struct {            /* ref: Solver.h:7-24 ContactLimiterPacked<1> */
    float p1x, p1y, p2x, p2y, a1, a2;
    float c1x, c1y, c2x, c2y, c1a, c2a, cim;
} limiter;

typedef struct {            /* ref: Solver.h:26-45 ContactJointPacked<1> */
    int32_t b1, b2, cp;
    limiter n;
    float n_acc, n_dst, n_dst_disp, n_acc_disp;
    limiter f;
    float f_acc;
} pjoint;

typedef struct {
    int nb, nj;
    sbody *imp, *disp;
    sparams* par;
    pjoint* pj;            /* indexed by slot (position in joint_index) */
    int32_t* joint_index;  /* slot -> joint, -1 for alignment padding */
    int slots;
    const phxo_contact_point* cps;
    phxo_contact_joint* joints;
    /* static-tag bookkeeping for PHXO_STAG_COLOUR_SYNC / event counting */
    int stag_mode;
    int fp16;                     /* body velocities are rounded to binary16 on every store (ablation) */
    const int32_t* slot_colour;   /* slot -> colour or NULL */
    uint8_t* is_static;
    /* colour-sync bookkeeping for static bodies, two words per body selected by iteration parity:
     * sw_iter[p][b] = the iteration (parity p) in which a joint on b was last productive,
     * sw_col[p][b]  = the smallest colour index that was productive in that iteration. */
    int32_t *sw_iter[2], *sw_col[2];
    phxo_solve_stats* st;
} sctx;

/* ref: Solver.cpp:456-480 */
static void prepare_bodies(sctx* c, const phxo_body* bodies)
{
    for (int i = 0; i < c->nb; ++i) {
        const phxo_body* b = &bodies[i];
        c->par[i].im = b->inv_mass; c->par[i].ii = b->inv_inertia;
        c->par[i].pos = b->pos; c->par[i].xv = b->xv; c->par[i].yv = b->yv;
        c->imp[i].vx = b->velocity.x; c->imp[i].vy = b->velocity.y; c->imp[i].w = b->angular_velocity; c->imp[i].tag = -1;
        c->disp[i].vx = b->displacing_velocity.x; c->disp[i].vy = b->displacing_velocity.y;
        c->disp[i].w = b->displacing_angular_velocity; c->disp[i].tag = -1;
        c->is_static[i] = (b->inv_mass == 0.f && b->inv_inertia == 0.f); /* ref: Solver.cpp:304 */
    }
}

/* ref: Solver.cpp:482-494 */
static void finish_bodies(const sctx* c, phxo_body* bodies)
{
    for (int i = 0; i < c->nb; ++i) {
        bodies[i].velocity.x = c->imp[i].vx; bodies[i].velocity.y = c->imp[i].vy; bodies[i].angular_velocity = c->imp[i].w;
        bodies[i].displacing_velocity.x = c->disp[i].vx; bodies[i].displacing_velocity.y = c->disp[i].vy;
        bodies[i].displacing_angular_velocity = c->disp[i].w;
    }
}

/* ref: Solver.cpp:549-590 RefreshLimiter */
static void refresh_limiter(limiter* L, float n1x, float n1y, float n2x, float n2y,
                            float w1x, float w1y, float w2x, float w2y, float im1, float ii1, float im2, float ii2)
{
    L->p1x = n1x; L->p1y = n1y; L->p2x = n2x; L->p2y = n2y;
    L->a1 = n1x * w1y - n1y * w1x;
    L->a2 = n2x * w2y - n2y * w2x;
    L->c1x = L->p1x * im1; L->c1y = L->p1y * im1; L->c1a = L->a1 * ii1;
    L->c2x = L->p2x * im2; L->c2y = L->p2y * im2; L->c2a = L->a2 * ii2;
    float m1 = L->p1x * L->c1x + L->p1y * L->c1y + L->a1 * L->c1a;
    float m2 = L->p2x * L->c2x + L->p2y * L->c2y + L->a2 * L->c2a;
    float m = m1 + m2;
    L->cim = fabsf(m) > 0.f ? 1.0f / m : 0.f;
}

/* ref: Solver.cpp:592-695 RefreshJoints<1,1> body */
static void refresh_one(pjoint* J, const sbody* imp, const sparams* par, const phxo_contact_point* cps)
{
    const sbody *v1 = &imp[J->b1], *v2 = &imp[J->b2];
    const sparams *q1 = &par[J->b1], *q2 = &par[J->b2];
    const phxo_contact_point* cp = &cps[J->cp];

    float p1x = cp->delta1.x + q1->pos.x, p1y = cp->delta1.y + q1->pos.y;
    float p2x = cp->delta2.x + q2->pos.x, p2y = cp->delta2.y + q2->pos.y;
    float w1x = cp->delta1.x, w1y = cp->delta1.y;
    float w2x = p1x - q2->pos.x, w2y = p1y - q2->pos.y;   /* ref: :649-650 — body-1's point, on purpose (Appendix C.1) */
    float nx = cp->normal.x, ny = cp->normal.y;

    refresh_limiter(&J->n, nx, ny, -nx, -ny, w1x, w1y, w2x, w2y, q1->im, q1->ii, q2->im, q2->ii);

    const float bounce = 0.f, delta_velocity = 1.f, max_pen_velocity = 0.1f, delta_depth = 1.f, error_reduction = 0.1f; /* :658-662 */
    float pv1x = (q1->pos.y - p1y) * v1->w + v1->vx, pv1y = (p1x - q1->pos.x) * v1->w + v1->vy;
    float pv2x = (q2->pos.y - p2y) * v2->w + v2->vx, pv2y = (p2x - q2->pos.x) * v2->w + v2->vy;
    float rvx = pv1x - pv2x, rvy = pv1y - pv2y;
    float dv = -bounce * (rvx * nx + rvy * ny);
    float depth = (p2x - p1x) * nx + (p2y - p1y) * ny;
    float dst = maxf_ref(dv - delta_velocity, 0.f);
    J->n_dst = depth < delta_depth ? dst - max_pen_velocity : dst;
    J->n_dst_disp = error_reduction * maxf_ref(0.f, depth - 2.0f * delta_depth);
    J->n_acc_disp = 0.f;

    float tx = -ny, ty = nx;
    refresh_limiter(&J->f, tx, ty, -tx, -ty, w1x, w1y, w2x, w2y, q1->im, q1->ii, q2->im, q2->ii);
}

void phxo_refresh_joint(const phxo_body* bodies, const phxo_contact_point* cps, const phxo_contact_joint* j, float out[30])
{
    sbody imp[2]; sparams par[2];
    const phxo_body* bb[2] = {&bodies[j->body1], &bodies[j->body2]};
    for (int k = 0; k < 2; ++k) {
        imp[k].vx = bb[k]->velocity.x; imp[k].vy = bb[k]->velocity.y; imp[k].w = bb[k]->angular_velocity; imp[k].tag = -1;
        par[k].im = bb[k]->inv_mass; par[k].ii = bb[k]->inv_inertia; par[k].pos = bb[k]->pos; par[k].xv = bb[k]->xv; par[k].yv = bb[k]->yv;
    }
    pjoint J; memset(&J, 0, sizeof J);
    J.b1 = 0; J.b2 = 1; J.cp = 0;
    refresh_one(&J, imp, par, &cps[j->contact_point_index]);
    memcpy(out, &J.n, 13 * sizeof(float));
    out[13] = 0.f; out[14] = J.n_dst; out[15] = J.n_dst_disp; out[16] = J.n_acc_disp;
    memcpy(out + 17, &J.f, 13 * sizeof(float));
}

/* ---- the sweeps' arithmetic form ------------------------------------------------------------------------------------
 * The reference writes `dV -= projector * velocity` and `velocity += compMass * dImpulse` (ref: Solver.cpp:736-750, 833-858,
 * 866-889, 973-996) and is built -ffast-math -mfma (ref: Makefile:11, 17-24): whether such a pair is one fused multiply-add
 * or a rounded product and a rounded sum is its compiler's choice.  The oracle restates both, in the reference's source order:
 *   PHXO_ARITH_SOURCE (0, default)  product and sum rounded separately — what this file's -ffp-contract=off gives the plain text;
 *   PHXO_ARITH_FUSED  (1)           every such pair is one fmaf() (correctly rounded once, C99 7.12.13.1).
 * Nothing else differs (RefreshJoints, the clamps, the productive tests, the order of the joints).  Process-wide, set by the
 * tests to the form the library under test reports (phx_arith_mode). */
static int g_arith = 0;
void phxo_set_arith(int fused) { g_arith = fused ? 1 : 0; }
int phxo_get_arith(void) { return g_arith; }
static inline float mul_add(float a, float b, float acc) { return g_arith ? fmaf(a, b, acc) : acc + a * b; }      /* acc + a * b */
static inline float mul_sub(float a, float b, float acc) { return g_arith ? fmaf(-a, b, acc) : acc - a * b; }     /* acc - a * b */

/* ref: Solver.cpp:697-758 PreStepJoints<1,1> body */
static void prestep_one(const pjoint* J, sbody* imp, int half, const uint8_t* is_static)
{
    sbody *b1 = &imp[J->b1], *b2 = &imp[J->b2];
    float v1x = b1->vx, v1y = b1->vy, w1 = b1->w, v2x = b2->vx, v2y = b2->vy, w2 = b2->w;
    v1x = mul_add(J->n.c1x, J->n_acc, v1x); v1y = mul_add(J->n.c1y, J->n_acc, v1y); w1 = mul_add(J->n.c1a, J->n_acc, w1);
    v2x = mul_add(J->n.c2x, J->n_acc, v2x); v2y = mul_add(J->n.c2y, J->n_acc, v2y); w2 = mul_add(J->n.c2a, J->n_acc, w2);
    v1x = mul_add(J->f.c1x, J->f_acc, v1x); v1y = mul_add(J->f.c1y, J->f_acc, v1y); w1 = mul_add(J->f.c1a, J->f_acc, w1);
    v2x = mul_add(J->f.c2x, J->f_acc, v2x); v2y = mul_add(J->f.c2y, J->f_acc, v2y); w2 = mul_add(J->f.c2a, J->f_acc, w2);
    /* a static body's compMass is 0, so its velocity is unchanged; the fp16 form (like the device) does not store it */
    if (!(half && is_static[J->b1])) { b1->vx = qh(half, v1x); b1->vy = qh(half, v1y); b1->w = qh(half, w1); }
    if (!(half && is_static[J->b2])) { b2->vx = qh(half, v2x); b2->vy = qh(half, v2y); b2->w = qh(half, w2); }
}

static inline float flipsign_scalar(float x, float y) { return y < 0.f ? -x : x; }             /* ref: base/SIMD_Scalar.h:265-268 */
static inline float flipsign_simd(float x, float y)                                             /* ref: base/SIMD_AVX2.h:272-275   */
{
    uint32_t xb, yb; memcpy(&xb, &x, 4); memcpy(&yb, &y, 4);
    xb ^= yb & 0x80000000u; memcpy(&x, &xb, 4); return x;
}

/* exported so tests can pin the two flipsign forms and max against the reference's scalar / SSE2 / AVX2 wrappers */
float phxo_flipsign(float x, float y, int simd) { return simd ? flipsign_simd(x, y) : flipsign_scalar(x, y); }
float phxo_max(float l, float r) { return maxf_ref(l, r); }

/* ref: Solver.cpp:790-798 skip test for one body: lastIteration > iterationIndex - 2.
 * PHXO_STAG_COLOUR_SYNC changes only how a STATIC body's tag is observed: a productive joint of the
 * current iteration is visible only to joints of later colours.  Tags start at -1
 * (ref: Solver.cpp:474,478), which passes the test in iteration 0 only. */
static inline int body_productive(const sctx* c, const sbody* arr, int body, int slot, int iter)
{
    if (c->stag_mode == PHXO_STAG_COLOUR_SYNC && c->slot_colour && c->is_static[body]) {
        if (iter == 0) return 1;
        if (c->sw_iter[(iter - 1) & 1][body] == iter - 1) return 1;
        return c->sw_iter[iter & 1][body] == iter && c->sw_col[iter & 1][body] < c->slot_colour[slot];
    }
    return arr[body].tag > iter - 2;
}

static inline void mark_productive(sctx* c, sbody* arr, int body, int slot, int iter)
{
    if (c->slot_colour && c->is_static[body]) {
        int p = iter & 1, col = c->slot_colour[slot];
        if (c->sw_iter[p][body] != iter) { c->sw_iter[p][body] = iter; c->sw_col[p][body] = col; }
        else if (col < c->sw_col[p][body]) c->sw_col[p][body] = col;
    }
    arr[body].tag = iter;
}

static void reset_static_words(sctx* c)
{
    for (int p = 0; p < 2; ++p)
        for (int i = 0; i < c->nb; ++i) { c->sw_iter[p][i] = -100; c->sw_col[p][i] = 0; }
}

/* one joint of ref: Solver.cpp:800-911; returns productive */
static int impulse_one(pjoint* J, sbody* imp, int simd_flipsign, float* out_dn, float* out_df, int half, const uint8_t* is_static)
{
    sbody *b1 = &imp[J->b1], *b2 = &imp[J->b2];
    float v1x = b1->vx, v1y = b1->vy, w1 = b1->w, v2x = b2->vx, v2y = b2->vy, w2 = b2->w;
    const limiter *N = &J->n, *F = &J->f;

    float dv = J->n_dst;
    dv = mul_sub(N->p1x, v1x, dv); dv = mul_sub(N->p1y, v1y, dv); dv = mul_sub(N->a1, w1, dv);
    dv = mul_sub(N->p2x, v2x, dv); dv = mul_sub(N->p2y, v2y, dv); dv = mul_sub(N->a2, w2, dv);
    float dn = dv * N->cim;
    dn = maxf_ref(dn, -J->n_acc);
    v1x = mul_add(N->c1x, dn, v1x); v1y = mul_add(N->c1y, dn, v1y); w1 = mul_add(N->c1a, dn, w1);
    v2x = mul_add(N->c2x, dn, v2x); v2y = mul_add(N->c2y, dn, v2y); w2 = mul_add(N->c2a, dn, w2);
    J->n_acc += dn;

    float fv = 0.f;
    fv = mul_sub(F->p1x, v1x, fv); fv = mul_sub(F->p1y, v1y, fv); fv = mul_sub(F->a1, w1, fv);
    fv = mul_sub(F->p2x, v2x, fv); fv = mul_sub(F->p2y, v2y, fv); fv = mul_sub(F->a2, w2, fv);
    float df = fv * F->cim;
    float reaction = J->n_acc, acc = J->f_acc;
    float force = acc + df;
    float limit = reaction * 0.3f;                                 /* kFrictionCoefficient, ref: Solver.cpp:9 */
    float signed_limit = simd_flipsign ? flipsign_simd(limit, force) : flipsign_scalar(limit, force);
    float adjusted = signed_limit - acc;
    if (fabsf(force) > limit) df = adjusted;
    J->f_acc += df;
    v1x = mul_add(F->c1x, df, v1x); v1y = mul_add(F->c1y, df, v1y); w1 = mul_add(F->c1a, df, w1);
    v2x = mul_add(F->c2x, df, v2x); v2y = mul_add(F->c2y, df, v2y); w2 = mul_add(F->c2a, df, w2);

    if (!(half && is_static[J->b1])) { b1->vx = qh(half, v1x); b1->vy = qh(half, v1y); b1->w = qh(half, w1); }
    if (!(half && is_static[J->b2])) { b2->vx = qh(half, v2x); b2->vy = qh(half, v2y); b2->w = qh(half, w2); }
    *out_dn = dn; *out_df = df;
    return maxf_ref(fabsf(dn), fabsf(df)) > 1e-4f;                  /* kProductiveImpulse, ref: Solver.cpp:8 */
}

/* one joint of ref: Solver.cpp:973-1015 */
static int displacement_one(pjoint* J, sbody* disp, int half, const uint8_t* is_static)
{
    sbody *b1 = &disp[J->b1], *b2 = &disp[J->b2];
    const limiter* N = &J->n;
    float v1x = b1->vx, v1y = b1->vy, w1 = b1->w, v2x = b2->vx, v2y = b2->vy, w2 = b2->w;
    float dv = J->n_dst_disp;
    dv = mul_sub(N->p1x, v1x, dv); dv = mul_sub(N->p1y, v1y, dv); dv = mul_sub(N->a1, w1, dv);
    dv = mul_sub(N->p2x, v2x, dv); dv = mul_sub(N->p2y, v2y, dv); dv = mul_sub(N->a2, w2, dv);
    float di = dv * N->cim;
    di = maxf_ref(di, -J->n_acc_disp);
    v1x = mul_add(N->c1x, di, v1x); v1y = mul_add(N->c1y, di, v1y); w1 = mul_add(N->c1a, di, w1);
    v2x = mul_add(N->c2x, di, v2x); v2y = mul_add(N->c2y, di, v2y); w2 = mul_add(N->c2a, di, w2);
    J->n_acc_disp += di;
    if (!(half && is_static[J->b1])) { b1->vx = qh(half, v1x); b1->vy = qh(half, v1y); b1->w = qh(half, w1); }
    if (!(half && is_static[J->b2])) { b2->vx = qh(half, v2x); b2->vy = qh(half, v2y); b2->w = qh(half, w2); }
    return fabsf(di) > 1e-4f;
}

/* sweep slots [begin,end) of one iteration with vector width vn (1 = scalar; N = group-granular
 * skip, ref: Solver.cpp:798 `if (none(body_productive)) continue;` over N lanes).
 * which: 0 = impulse, 1 = displacement.  Returns any-productive. */
static int sweep(sctx* c, int begin, int end, int iter, int vn, int which)
{
    sbody* arr = which ? c->disp : c->imp;
    int any = 0;
    for (int g = begin; g < end; g += vn) {
        int lanes = vn;
        int go = 0;
        for (int l = 0; l < lanes; ++l) {
            int s = g + l;
            if (c->joint_index[s] < 0) continue;
            pjoint* J = &c->pj[s];
            if (body_productive(c, arr, J->b1, s, iter) || body_productive(c, arr, J->b2, s, iter)) go = 1;
        }
        if (!which && c->st) c->st->joint_visits += lanes;
        if (!go) continue;
        /* lanes of one group are body-disjoint (PrepareIndices), so lane order is immaterial */
        for (int l = 0; l < lanes; ++l) {
            int s = g + l;
            if (c->joint_index[s] < 0) continue;
            pjoint* J = &c->pj[s];
            int productive;
            if (which) productive = displacement_one(J, arr, c->fp16, c->is_static);
            else { float dn, df; productive = impulse_one(J, arr, vn > 1, &dn, &df, c->fp16, c->is_static); if (c->st) c->st->joints_computed++; }
            if (productive) {
                mark_productive(c, arr, J->b1, s, iter);
                mark_productive(c, arr, J->b2, s, iter);
                any = 1;
            }
        }
    }
    return any;
}

/* ref: Solver.cpp:217-273 */
static int prepare_indices(const phxo_contact_joint* joints, int32_t* joint_index, int32_t* group_bodies /*nb, zeroed by caller once*/,
                           int32_t* work, int begin, int end, int group)
{
    if (group == 1) return end;
    for (int i = begin; i < end; ++i) work[i] = joint_index[i];
    int tag = 0;
    /* NOTE: like the reference, tags restart at 0 for every island while group_bodies is only
     * zeroed once per SolveJoints (ref: Solver.cpp:83-84,105-106) — islands are body-disjoint
     * except for static bodies, which is the race the reference's own TODO at :244 mentions. */
    int remaining = end - begin, out = begin;
    while (remaining >= group) {
        int got = 0;
        ++tag;
        for (int i = 0; i < remaining && got < group;) {
            int ji = work[begin + i];
            const phxo_contact_joint* j = &joints[ji];
            if (group_bodies[j->body1] < tag && group_bodies[j->body2] < tag) {
                group_bodies[j->body1] = tag; group_bodies[j->body2] = tag;
                joint_index[out + got++] = ji;
                work[begin + i] = work[begin + remaining - 1];
                --remaining;
            } else ++i;
        }
        out += got;
        if (got < group) break;
    }
    for (int i = 0; i < remaining; ++i) joint_index[out + i] = work[begin + i];
    return out & ~(group - 1);
}

int phxo_prepare_indices(const phxo_contact_joint* joints, int nb, int32_t* joint_index, int begin, int end, int group_size)
{
    int32_t* gb = (int32_t*)calloc(nb > 0 ? nb : 1, sizeof(int32_t));
    int32_t* work = (int32_t*)malloc((end > 0 ? end : 1) * sizeof(int32_t));
    int r = prepare_indices(joints, joint_index, gb, work, begin, end, group_size);
    free(gb); free(work);
    return r;
}

static int uf_find(int32_t* t, int i) /* ref: Solver.cpp:275-283 */
{
    int r = i;
    while (r != t[r]) r = t[r];
    return t[i] = r;
}

/* ref: Solver.cpp:285-454 */
int phxo_gather_islands(const phxo_body* bodies, int nb, const phxo_contact_joint* joints, int nj, int group,
                        int32_t* joint_index, int cap, int32_t* island_offset, int32_t* island_size,
                        int32_t* island_count_out, int32_t* island_max_out)
{
    int32_t* root = (int32_t*)malloc((nb + 1) * sizeof(int32_t));
    int32_t* number = (int32_t*)malloc((nb + 1) * sizeof(int32_t));
    int32_t* merged = (int32_t*)malloc((nb + 1) * sizeof(int32_t));
    int32_t* cursor = (int32_t*)malloc((nb + 1) * sizeof(int32_t));

    for (int i = 0; i < nb; ++i) root[i] = (bodies[i].inv_mass == 0.f && bodies[i].inv_inertia == 0.f) ? -1 : i;   /* :302-305 */
    for (int k = 0; k < nj; ++k) {                                                                              /* :311-323 */
        int a = root[joints[k].body1], b = root[joints[k].body2];
        if ((a | b) < 0) continue;
        int ra = uf_find(root, a), rb = uf_find(root, b);
        root[ra] = rb;
    }
    int count = 0;
    for (int i = 0; i < nb; ++i) number[i] = -1;
    for (int i = 0; i < nb; ++i) if (root[i] >= 0) root[i] = uf_find(root, i);                                    /* :336-342 */
    for (int i = 0; i < nb; ++i) {                                                                              /* :344-356 */
        if (root[i] < 0) continue;
        if (number[root[i]] < 0) number[root[i]] = count++;
    }
    for (int i = 0; i < count; ++i) island_offset[i] = 0;
    for (int k = 0; k < nj; ++k) {                                                                              /* :367-379 */
        int a = root[joints[k].body1], b = root[joints[k].body2];
        if ((a & b) < 0) continue;                 /* both static: joint belongs to no island */
        int isl = a < 0 ? b : a;
        island_offset[number[isl]]++;
    }
    /* coalesce consecutive islands until >= kIslandMinSize (256) joints, pad to `group` (:382-413) */
    int run_index = 0, run_count = 0, total = 0;
    for (int i = 0; i < count; ++i) {
        run_count += island_offset[i];
        merged[i] = run_index;
        if (run_count >= 256 || (run_count > 0 && i == count - 1)) {
            int aligned = (run_count + group - 1) & ~(group - 1);
            island_size[run_index] = run_count;
            island_offset[run_index] = total;
            total += aligned;
            run_count = 0;
            run_index++;
        }
    }
    count = run_index;
    if (total > cap) { free(root); free(number); free(merged); free(cursor); return -total; }
    for (int i = 0; i < total; ++i) joint_index[i] = -1;
    for (int i = 0; i < count; ++i) cursor[i] = island_offset[i];
    for (int k = 0; k < nj; ++k) {                                                                              /* :425-438 */
        int a = root[joints[k].body1], b = root[joints[k].body2];
        if ((a & b) < 0) continue;
        int isl = a < 0 ? b : a;
        joint_index[cursor[merged[number[isl]]]++] = k;
    }
    int mx = 0;
    for (int i = 0; i < count; ++i) if (island_size[i] > mx) mx = island_size[i];
    *island_count_out = count; *island_max_out = mx;
    free(root); free(number); free(merged); free(cursor);
    return total;
}

static void ctx_alloc(sctx* c, int nb, int slots)
{
    memset(c, 0, sizeof *c);
    c->nb = nb; c->slots = slots;
    c->imp = (sbody*)malloc((nb + 1) * sizeof(sbody));
    c->disp = (sbody*)malloc((nb + 1) * sizeof(sbody));
    c->par = (sparams*)malloc((nb + 1) * sizeof(sparams));
    c->is_static = (uint8_t*)calloc(nb + 1, 1);
    for (int p = 0; p < 2; ++p) {
        c->sw_iter[p] = (int32_t*)malloc((nb + 1) * sizeof(int32_t));
        c->sw_col[p] = (int32_t*)malloc((nb + 1) * sizeof(int32_t));
    }
    reset_static_words(c);
    c->pj = (pjoint*)calloc(slots + 8, sizeof(pjoint));
    c->joint_index = (int32_t*)malloc((slots + 8) * sizeof(int32_t));
}

static void ctx_free(sctx* c)
{
    free(c->imp); free(c->disp); free(c->par); free(c->is_static); for (int p = 0; p < 2; ++p) { free(c->sw_iter[p]); free(c->sw_col[p]); }
    free(c->pj); free(c->joint_index);
}

/* ref: Solver.cpp:509-521 CopyJoints into packed slots */
static void copy_joints_in(sctx* c, int begin, int end)
{
    for (int s = begin; s < end; ++s) {
        int ji = c->joint_index[s];
        if (ji < 0) continue;
        const phxo_contact_joint* j = &c->joints[ji];
        pjoint* J = &c->pj[s];
        J->b1 = j->body1; J->b2 = j->body2; J->cp = j->contact_point_index;
        J->n_acc = j->normal_acc; J->f_acc = j->friction_acc;
    }
}

/* ref: Solver.cpp:527-547 */
static void copy_joints_out(sctx* c, int begin, int end)
{
    for (int s = begin; s < end; ++s) {
        int ji = c->joint_index[s];
        if (ji < 0) continue;
        c->joints[ji].normal_acc = c->pj[s].n_acc;
        c->joints[ji].friction_acc = c->pj[s].f_acc;
    }
}

/* ref: Solver.cpp:130-215 SolveJointIsland<N>, workers = 0.  With workers = 0 the Sloppy batch
 * split (:138-139) runs the batches back to back in index order, which is the same sequence as
 * the single whole-island batch, so island_mode's sloppy bit does not change the result here. */
static void solve_island(sctx* c, int begin, int end, int group_offset, int n, int contact_iters, int pen_iters)
{
    int vec_end = group_offset < end ? group_offset : end;
    int tail_begin = group_offset > begin ? group_offset : begin;

    for (int s = begin; s < end; ++s) if (c->joint_index[s] >= 0) refresh_one(&c->pj[s], c->imp, c->par, c->cps);
    for (int s = begin; s < end; ++s) if (c->joint_index[s] >= 0) prestep_one(&c->pj[s], c->imp, c->fp16, c->is_static);

    int it;
    reset_static_words(c);
    for (it = 0; it < contact_iters; ++it) {
        int p = sweep(c, begin, vec_end, it, n, 0);
        p |= sweep(c, tail_begin, end, it, 1, 0);
        if (!p) { ++it; break; }                                      /* ref: Solver.cpp:189 */
    }
    if (c->st && it > c->st->impulse_iterations) c->st->impulse_iterations = it;
    reset_static_words(c);
    for (it = 0; it < pen_iters; ++it) {
        int p = sweep(c, begin, vec_end, it, n, 1);
        p |= sweep(c, tail_begin, end, it, 1, 1);
        if (!p) { ++it; break; }                                      /* ref: Solver.cpp:210 */
    }
    if (c->st && it > c->st->displacement_iterations) c->st->displacement_iterations = it;
}

void phxo_solver_solve(phxo_body* bodies, int nb, const phxo_contact_point* cps, phxo_contact_joint* joints, int nj,
                       int solve_mode, int island_mode, int contact_iters, int pen_iters,
                       int32_t* order_out, int order_cap, phxo_solve_stats* stats)
{
    int n = solve_mode == PHXO_SOLVE_AVX2 ? 8 : solve_mode == PHXO_SOLVE_SSE2 ? 4 : 1;   /* ref: Solver.cpp:19-40 */
    int split = (island_mode == PHXO_ISLAND_MULTIPLE || island_mode == PHXO_ISLAND_MULTIPLE_SLOPPY);
    phxo_solve_stats local; if (!stats) stats = &local;
    memset(stats, 0, sizeof *stats);

    int slots_cap = nj + (nj / 256 + 2) * 8 + 16;
    sctx c; ctx_alloc(&c, nb, slots_cap);
    c.cps = cps; c.joints = joints; c.nj = nj; c.st = stats; c.stag_mode = PHXO_STAG_SEQUENTIAL;
    prepare_bodies(&c, bodies);

    int32_t* group_bodies = (int32_t*)calloc(nb + 1, sizeof(int32_t));
    int32_t* work = (int32_t*)malloc((slots_cap + 8) * sizeof(int32_t));

    if (split) {
        int32_t* off = (int32_t*)malloc((nb + 1) * sizeof(int32_t));
        int32_t* siz = (int32_t*)malloc((nb + 1) * sizeof(int32_t));
        int32_t cnt = 0, mx = 0;
        int total = phxo_gather_islands(bodies, nb, joints, nj, n, c.joint_index, slots_cap, off, siz, &cnt, &mx);
        if (total < 0) { fprintf(stderr, "phx_oracle: island slot overflow\n"); abort(); }
        c.slots = total;
        stats->island_count = cnt; stats->island_max_size = mx;
        for (int i = 0; i < cnt; ++i) {                           /* ref: Solver.cpp:86-91, workers = 0 => index order */
            int b = off[i], e = b + siz[i];
            int go = prepare_indices(joints, c.joint_index, group_bodies, work, b, e, n);
            copy_joints_in(&c, b, e);
            solve_island(&c, b, e, go, n, contact_iters, pen_iters);
            copy_joints_out(&c, b, e);
            stats->group_offset = go;
        }
        free(off); free(siz);
    } else {
        for (int i = 0; i < nj; ++i) c.joint_index[i] = i;          /* ref: Solver.cpp:102-103 */
        c.slots = nj;
        stats->island_count = 1; stats->island_max_size = nj;
        int go = prepare_indices(joints, c.joint_index, group_bodies, work, 0, nj, n);
        copy_joints_in(&c, 0, nj);
        solve_island(&c, 0, nj, go, n, contact_iters, pen_iters);
        copy_joints_out(&c, 0, nj);
        stats->group_offset = go;
    }
    if (order_out) for (int i = 0; i < c.slots && i < order_cap; ++i) order_out[i] = c.joint_index[i];
    finish_bodies(&c, bodies);
    free(group_bodies); free(work);
    ctx_free(&c);
}

void phxo_solver_solve_ordered(phxo_body* bodies, int nb, const phxo_contact_point* cps, phxo_contact_joint* joints, int nj,
                               const int32_t* order, const int32_t* colour_offsets, int ncolours,
                               int contact_iters, int pen_iters, int stag_mode, phxo_solve_stats* stats)
{
    phxo_solve_stats local; if (!stats) stats = &local;
    memset(stats, 0, sizeof *stats);
    sctx c; ctx_alloc(&c, nb, nj);
    c.cps = cps; c.joints = joints; c.nj = nj; c.st = stats; c.stag_mode = stag_mode;
    prepare_bodies(&c, bodies);
    for (int i = 0; i < nj; ++i) c.joint_index[i] = order ? order[i] : i;
    int32_t* colour = NULL;
    if (colour_offsets && ncolours > 0) {
        colour = (int32_t*)malloc((nj + 1) * sizeof(int32_t));
        for (int k = 0; k < ncolours; ++k)
            for (int s = colour_offsets[k]; s < colour_offsets[k + 1]; ++s) colour[s] = k;
        c.slot_colour = colour;
    }
    stats->island_count = 1; stats->island_max_size = nj; stats->group_offset = nj;
    copy_joints_in(&c, 0, nj);

    if (stag_mode == PHXO_STAG_SEQUENTIAL && colour) {
        /* run twice: once in colour-sync mode on a scratch copy just to count how many skip
         * decisions differ between the two visibility rules (reported as stag_events). */
        phxo_body* b2 = (phxo_body*)malloc((nb + 1) * sizeof(phxo_body));
        phxo_contact_joint* j2 = (phxo_contact_joint*)malloc((nj + 1) * sizeof(phxo_contact_joint));
        memcpy(b2, bodies, nb * sizeof(phxo_body)); memcpy(j2, joints, nj * sizeof(phxo_contact_joint));
        phxo_solve_stats s2;
        phxo_solver_solve_ordered(b2, nb, cps, j2, nj, order, colour_offsets, ncolours, contact_iters, pen_iters, PHXO_STAG_COLOUR_SYNC, &s2);
        solve_island(&c, 0, nj, nj, 1, contact_iters, pen_iters);
        stats->stag_events = stats->joints_computed - s2.joints_computed;
        free(b2); free(j2);
    } else {
        solve_island(&c, 0, nj, nj, 1, contact_iters, pen_iters);
    }
    copy_joints_out(&c, 0, nj);
    finish_bodies(&c, bodies);
    free(colour);
    ctx_free(&c);
}

/* The schedule form the HIP path uses in the island-aware modes: `order` is cut into GROUPS of consecutive
 * slots; each group is an independent SolveJointIsland (Refresh, PreStep, sweeps with its own early exit —
 * ref: Solver.cpp:130-215) and observes its OWN copy of every static body's lastIteration tag, i.e. the tags
 * are reset at each group start.  (In the reference a static body's tag is one shared word that a later island
 * inherits from the previous one, an artefact its own TODO at Solver.cpp:244 files under races; giving each
 * island a private copy is the deterministic reading.)  Inside a group the scalar (N=1) loop runs in slot
 * order, colours only matter for PHXO_STAG_COLOUR_SYNC. */
void phxo_solver_solve_grouped(phxo_body* bodies, int nb, const phxo_contact_point* cps, phxo_contact_joint* joints, int nj,
                               const int32_t* order, const int32_t* colour_offsets, int ncolours,
                               const int32_t* group_offsets, int ngroups,
                               int contact_iters, int pen_iters, int stag_mode, phxo_solve_stats* stats)
{
    phxo_solver_solve_grouped_fp16(bodies, nb, cps, joints, nj, order, colour_offsets, ncolours, group_offsets, ngroups,
                                   contact_iters, pen_iters, stag_mode, 0, stats);
}

/* Same, with the first `fp16_groups` groups keeping their body velocities in binary16 between joint updates (what the
 * device's island kernel does under phx_solver_set_body_state_bits(16)); arithmetic stays fp32. */
void phxo_solver_solve_grouped_fp16(phxo_body* bodies, int nb, const phxo_contact_point* cps, phxo_contact_joint* joints, int nj,
                                    const int32_t* order, const int32_t* colour_offsets, int ncolours,
                                    const int32_t* group_offsets, int ngroups,
                                    int contact_iters, int pen_iters, int stag_mode, int fp16_groups, phxo_solve_stats* stats)
{
    phxo_solve_stats local; if (!stats) stats = &local;
    memset(stats, 0, sizeof *stats);
    sctx c; ctx_alloc(&c, nb, nj);
    c.cps = cps; c.joints = joints; c.nj = nj; c.st = stats; c.stag_mode = stag_mode;
    prepare_bodies(&c, bodies);
    for (int i = 0; i < nj; ++i) c.joint_index[i] = order ? order[i] : i;
    int32_t* colour = NULL;
    if (colour_offsets && ncolours > 0) {
        colour = (int32_t*)malloc((nj + 1) * sizeof(int32_t));
        for (int k = 0; k < ncolours; ++k)
            for (int s = colour_offsets[k]; s < colour_offsets[k + 1]; ++s) colour[s] = k;
        c.slot_colour = colour;
    }
    stats->island_count = ngroups; stats->group_offset = nj;
    copy_joints_in(&c, 0, nj);
    for (int g = 0; g < ngroups; ++g) {
        int b = group_offsets[g], e = group_offsets[g + 1];
        if (e - b > stats->island_max_size) stats->island_max_size = e - b;
        for (int i = 0; i < nb; ++i) if (c.is_static[i]) { c.imp[i].tag = -1; c.disp[i].tag = -1; }
        c.fp16 = g < fp16_groups;
        if (c.fp16) {       /* the group's working copy of its bodies is binary16 from the start (the statics' copy is private) */
            for (int s = b; s < e; ++s)
                for (int side = 0; side < 2; ++side) {
                    int body = side ? c.pj[s].b2 : c.pj[s].b1;
                    if (c.is_static[body]) continue;
                    c.imp[body].vx = qh(1, c.imp[body].vx); c.imp[body].vy = qh(1, c.imp[body].vy); c.imp[body].w = qh(1, c.imp[body].w);
                    c.disp[body].vx = qh(1, c.disp[body].vx); c.disp[body].vy = qh(1, c.disp[body].vy); c.disp[body].w = qh(1, c.disp[body].w);
                }
            /* static bodies: rounded copies for the duration of the group, originals restored afterwards */
            sbody* keep_i = (sbody*)malloc((nb + 1) * sizeof(sbody));
            sbody* keep_d = (sbody*)malloc((nb + 1) * sizeof(sbody));
            memcpy(keep_i, c.imp, nb * sizeof(sbody)); memcpy(keep_d, c.disp, nb * sizeof(sbody));
            for (int i = 0; i < nb; ++i) if (c.is_static[i]) {
                c.imp[i].vx = qh(1, c.imp[i].vx); c.imp[i].vy = qh(1, c.imp[i].vy); c.imp[i].w = qh(1, c.imp[i].w);
                c.disp[i].vx = qh(1, c.disp[i].vx); c.disp[i].vy = qh(1, c.disp[i].vy); c.disp[i].w = qh(1, c.disp[i].w);
            }
            solve_island(&c, b, e, e, 1, contact_iters, pen_iters);
            for (int i = 0; i < nb; ++i) if (c.is_static[i]) { c.imp[i] = keep_i[i]; c.disp[i] = keep_d[i]; }
            free(keep_i); free(keep_d);
        } else {
            solve_island(&c, b, e, e, 1, contact_iters, pen_iters);
        }
        c.fp16 = 0;
    }
    copy_joints_out(&c, 0, nj);
    finish_bodies(&c, bodies);
    free(colour);
    ctx_free(&c);
}

/* ------------------------------------------------------------------------------------------ */
/* narrowphase (ref: Collider.cpp:8-245) — the step between the two hot halves                  */

/* ref: Collider.cpp:8-56 — box/box SAT, 4 axes, returns the axis of least penetration */
static int separating_axis(const phxo_body* b1, const phxo_body* b2, phxo_vec2* axis)
{
    phxo_vec2 a0[2] = {b1->xv, b1->yv}, a1[2] = {b2->xv, b2->yv};
    phxo_vec2 e0 = b1->geom_size, e1 = b2->geom_size;
    phxo_vec2 d = sub2(b1->pos, b2->pos);
    float ad[2][2];
    ad[0][0] = fabsf(dot2(a0[0], a1[0])); ad[0][1] = fabsf(dot2(a0[0], a1[1]));
    float r0 = e0.x + e1.x * ad[0][0] + e1.y * ad[0][1];
    float d0 = fabsf(dot2(a0[0], d)) - r0;
    if (d0 > 0) return 0;
    float best = d0; phxo_vec2 bestaxis = a0[0];
    ad[1][0] = fabsf(dot2(a0[1], a1[0])); ad[1][1] = fabsf(dot2(a0[1], a1[1]));
    float r1 = e0.y + e1.x * ad[1][0] + e1.y * ad[1][1];
    float d1 = fabsf(dot2(a0[1], d)) - r1;
    if (d1 > 0) return 0;
    if (d1 > best) { best = d1; bestaxis = a0[1]; }
    float r2 = e1.x + e0.x * ad[0][0] + e0.y * ad[1][0];
    float d2 = fabsf(dot2(a1[0], d)) - r2;
    if (d2 > 0) return 0;
    if (d2 > best) { best = d2; bestaxis = a1[0]; }
    float r3 = e1.y + e0.x * ad[0][1] + e0.y * ad[1][1];
    float d3 = fabsf(dot2(a1[1], d)) - r3;
    if (d3 > 0) return 0;
    if (d3 > best) { best = d3; bestaxis = a1[1]; }
    *axis = bestaxis;
    return 1;
}

/* ref: Manifold.h:31-38 */
static int cp_equals(const phxo_contact_point* a, const phxo_contact_point* o, float tol)
{
    if (sqlen2(sub2(o->delta1, a->delta1)) > tol * tol && sqlen2(sub2(o->delta2, a->delta2)) > tol * tol) return 0;
    return 1;
}

/* ref: Manifold.h:18-27 */
static phxo_contact_point make_point(phxo_vec2 p1, phxo_vec2 p2, phxo_vec2 n, const phxo_body* b1, const phxo_body* b2)
{
    phxo_contact_point c; memset(&c, 0, sizeof c);
    c.delta1 = sub2(p1, b1->pos); c.delta2 = sub2(p2, b2->pos); c.normal = n;
    c.is_merged = 0; c.is_newly_created = 1; c.solver_index = -1;
    return c;
}

/* exported forms of the header-resident narrowphase leaves, so tests can pin them against oracle/_ref */
int phxo_contact_equals(const phxo_contact_point* a, const phxo_contact_point* o, float tol) { return cp_equals(a, o, tol); }
void phxo_contact_point_make(phxo_contact_point* out, float p1x, float p1y, float p2x, float p2y, float nx, float ny,
                             const phxo_body* b1, const phxo_body* b2)
{
    phxo_vec2 p1 = {p1x, p1y}, p2 = {p2x, p2y}, n = {nx, ny};
    *out = make_point(p1, p2, n, b1, b2);
}

/* ref: Collider.cpp:58-92 */
static void add_point(phxo_contact_point* pts, int* count, phxo_contact_point* nb)
{
    phxo_contact_point* closest = NULL;
    float bestdepth = 3.402823466e+38f;
    for (int i = 0; i < *count; ++i) {
        phxo_contact_point* col = &pts[i];
        if (cp_equals(nb, col, 2.0f)) {
            float depth = sqlen2(sub2(nb->delta1, col->delta1)) + sqlen2(sub2(nb->delta2, col->delta2));
            if (depth < bestdepth) { bestdepth = depth; closest = col; }
        }
    }
    if (closest) {
        closest->is_merged = 1; closest->is_newly_created = 0;
        closest->normal = nb->normal; closest->delta1 = nb->delta1; closest->delta2 = nb->delta2;
    } else {
        nb->is_merged = 1; nb->is_newly_created = 1;
        pts[(*count)++] = *nb;
    }
}

/* ref: Vector2.h ProjectPointToLine(point, planePoint, planeNormal, projectionDirection, out) */
static phxo_vec2 project_to_line(phxo_vec2 point, phxo_vec2 plane_point, phxo_vec2 plane_normal, phxo_vec2 dir)
{
    float mult = 1.0f / dot2(dir, plane_normal);
    float s = dot2(plane_point, plane_normal) - dot2(point, plane_normal);
    return add2(point, mul2(mul2(dir, s), mult));
}

void phxo_project_point_to_line(float px, float py, float qx, float qy, float nx, float ny, float dx, float dy, float out[2])
{
    phxo_vec2 p = {px, py}, q = {qx, qy}, n = {nx, ny}, d = {dx, dy};
    phxo_vec2 r = project_to_line(p, q, n, d);
    out[0] = r.x; out[1] = r.y;
}

static int within_segment(phxo_vec2 p, phxo_vec2 a, phxo_vec2 b)
{
    return dot2(sub2(p, a), sub2(b, a)) >= 0.0f && dot2(sub2(p, b), sub2(a, b)) >= 0.0f;
}

/* ref: Collider.cpp:94-209 */
static void generate_contacts(const phxo_body* b1, const phxo_body* b2, phxo_contact_point* pts, int* count, phxo_vec2 axis)
{
    if (dot2(axis, sub2(b1->pos, b2->pos)) < 0.0f) axis = neg2(axis);
    phxo_vec2 s1[2], s2[2];
    const float lin_tol = 2.0f;
    int n1 = phxo_support_points(b1, -axis.x, -axis.y, s1);
    int n2 = phxo_support_points(b2, axis.x, axis.y, s2);
    if (n1 == 2 && sqlen2(sub2(s1[0], s1[1])) < lin_tol * lin_tol) { s1[0] = mul2(add2(s1[0], s1[1]), 0.5f); n1 = 1; }
    if (n2 == 2 && sqlen2(sub2(s2[0], s2[1])) < lin_tol * lin_tol) { s2[0] = mul2(add2(s2[0], s2[1]), 0.5f); n2 = 1; }

    if (n1 == 1 && n2 == 1) {
        phxo_vec2 delta = sub2(s2[0], s1[0]);
        if (dot2(delta, axis) >= 0.0f) {
            phxo_contact_point c = make_point(s1[0], s2[0], axis, b1, b2);
            add_point(pts, count, &c);
        }
    } else if (n1 == 1 && n2 == 2) {
        phxo_vec2 n = perp2(sub2(s2[1], s2[0]));
        phxo_vec2 p = project_to_line(s1[0], s2[0], n, axis);
        if (within_segment(p, s2[0], s2[1])) {
            phxo_contact_point c = make_point(s1[0], p, axis, b1, b2);
            add_point(pts, count, &c);
        }
    } else if (n1 == 2 && n2 == 1) {
        phxo_vec2 n = perp2(sub2(s1[1], s1[0]));
        phxo_vec2 p = project_to_line(s2[0], s1[0], n, axis);
        if (within_segment(p, s1[0], s1[1])) {
            phxo_contact_point c = make_point(p, s2[0], axis, b1, b2);
            add_point(pts, count, &c);
        }
    } else if (n1 == 2 && n2 == 2) {
        phxo_vec2 t1[4], t2[4]; int tc = 0;
        for (int i = 0; i < 2; ++i) {
            phxo_vec2 n = perp2(sub2(s2[1], s2[0]));
            if (dot2(sub2(s1[i], s2[0]), n) >= 0.0f) {
                phxo_vec2 p = project_to_line(s1[i], s2[0], n, axis);
                if (within_segment(p, s2[0], s2[1])) { t1[tc] = s1[i]; t2[tc] = p; tc++; }
            }
        }
        for (int i = 0; i < 2; ++i) {
            phxo_vec2 n = perp2(sub2(s1[1], s1[0]));
            if (dot2(sub2(s2[i], s1[0]), n) >= 0.0f) {
                phxo_vec2 p = project_to_line(s2[i], s1[0], n, axis);
                if (within_segment(p, s1[0], s1[1])) { t1[tc] = p; t2[tc] = s2[i]; tc++; }
            }
        }
        if (tc == 1) {
            phxo_contact_point c = make_point(t1[0], t2[0], axis, b1, b2);
            add_point(pts, count, &c);
        }
        if (tc >= 2) {
            phxo_contact_point c1 = make_point(t1[0], t2[0], axis, b1, b2);
            add_point(pts, count, &c1);
            phxo_contact_point c2 = make_point(t1[1], t2[1], axis, b1, b2);
            add_point(pts, count, &c2);
        }
    }
}

/* ref: Collider.cpp:211-245.  Returns 1 if more than kMaxContactPoints merged points had to be
 * clamped (the reference would write past the manifold's slots there). */
static int update_manifold(phxo_manifold* m, const phxo_body* bodies, phxo_contact_point* pts)
{
    phxo_contact_point np[4];
    for (int i = 0; i < m->point_count; ++i) { np[i] = pts[i]; np[i].is_merged = 0; np[i].is_newly_created = 0; }
    int count = m->point_count;
    const phxo_body *b1 = &bodies[m->body1], *b2 = &bodies[m->body2];
    phxo_vec2 axis;
    if (separating_axis(b1, b2, &axis)) generate_contacts(b1, b2, np, &count, axis);
    m->point_count = 0;
    int overflow = 0;
    for (int i = 0; i < count; ++i)
        if (np[i].is_merged) {
            if (m->point_count < 2) pts[m->point_count++] = np[i];
            else overflow = 1;
        }
    return overflow;
}

/* ------------------------------------------------------------------------------------------ */
/* world                                                                                       */

struct phxo_world {
    phxo_body* bodies; size_t nb, cap_b;
    phxo_manifold* manifolds; size_t nm, cap_m;
    phxo_contact_point* cps; size_t ncp, cap_cp;
    phxo_contact_joint* joints; size_t nj, cap_j;
    pairset set;
    float gravity;
    phxo_sort_entry* sorted; phxo_bp_entry* entries; size_t cap_bp;
    uint32_t* new_pairs; size_t n_new, cap_new;
    uint64_t sweep_tests;
    int point_overflows;
    phxo_solve_stats stats;
};

phxo_world* phxo_world_create(void) { return (phxo_world*)calloc(1, sizeof(phxo_world)); }

void phxo_world_destroy(phxo_world* w)
{
    if (!w) return;
    free(w->bodies); free(w->manifolds); free(w->cps); free(w->joints); free(w->set.slot);
    free(w->sorted); free(w->entries); free(w->new_pairs); free(w);
}

int phxo_world_add_body(phxo_world* w, float px, float py, float angle, float sx, float sy) /* ref: World.cpp:11-17 */
{
    GROW(w->bodies, w->cap_b, w->nb + 1, phxo_body);
    phxo_body* b = &w->bodies[w->nb];
    phxo_body_init(b, px, py, angle, sx, sy, 1e-5f);
    b->index = (uint32_t)w->nb;
    return (int)w->nb++;
}

void phxo_world_set_gravity(phxo_world* w, float g) { w->gravity = g; }

static void integrate_velocity(phxo_world* w, float dt) /* ref: World.cpp:39-55 */
{
    for (size_t i = 0; i < w->nb; ++i) {
        phxo_body* b = &w->bodies[i];
        if (b->inv_mass > 0.0f) b->acceleration.y += w->gravity;
        b->velocity.x += b->acceleration.x * dt; b->velocity.y += b->acceleration.y * dt;
        b->acceleration.x = 0.f; b->acceleration.y = 0.f;
        b->angular_velocity += b->angular_acceleration * dt;
        b->angular_acceleration = 0.f;
    }
}

void phxo_world_integrate_position(phxo_world* w, float dt) /* ref: World.cpp:57-70 */
{
    for (size_t i = 0; i < w->nb; ++i) {
        phxo_body* b = &w->bodies[i];
        b->pos.x += b->displacing_velocity.x + b->velocity.x * dt;
        b->pos.y += b->displacing_velocity.y + b->velocity.y * dt;
        float ang = -(b->displacing_angular_velocity + b->angular_velocity * dt);
        phxo_rotate_vec(&b->xv, ang);
        phxo_rotate_vec(&b->yv, ang);
        b->displacing_velocity.x = 0.f; b->displacing_velocity.y = 0.f;
        b->displacing_angular_velocity = 0.f;
        update_geom(b);
    }
}

static void update_pairs(phxo_world* w) /* ref: Collider.cpp:296-318 (workers = 0 path) */
{
    const phxo_bp_entry* e = w->entries; size_t n = w->nb;
    w->n_new = 0; w->sweep_tests = 0;
    for (size_t i = 0; i < n; ++i) {
        float maxx = e[i].maxx;
        for (size_t j = i + 1; j < n; ++j) {
            if (e[j].minx > maxx) break;
            w->sweep_tests++;
            if (fabsf(e[j].centery - e[i].centery) <= e[i].extenty + e[j].extenty) {
                if (ps_insert(&w->set, e[i].index, e[j].index)) {
                    GROW(w->manifolds, w->cap_m, w->nm + 1, phxo_manifold);
                    phxo_manifold m = {(int32_t)e[i].index, (int32_t)e[j].index, 0, (int32_t)(w->nm * 2)};
                    w->manifolds[w->nm++] = m;
                    GROW(w->new_pairs, w->cap_new, 2 * (w->n_new + 1), uint32_t);
                    w->new_pairs[2 * w->n_new] = e[i].index; w->new_pairs[2 * w->n_new + 1] = e[j].index;
                    w->n_new++;
                }
            }
        }
    }
}

static int aabb_intersects(const phxo_body* a, const phxo_body* b) /* ref: AABB2.h:19-24 */
{
    if (a->aabb_min.x > b->aabb_max.x || b->aabb_min.x > a->aabb_max.x) return 0;
    if (a->aabb_min.y > b->aabb_max.y || b->aabb_min.y > a->aabb_max.y) return 0;
    return 1;
}

int phxo_aabb_intersects(const phxo_body* a, const phxo_body* b) { return aabb_intersects(a, b); }

static void pack_manifolds(phxo_world* w) /* ref: Collider.cpp:379-416 */
{
    for (size_t i = 0; i < w->nm;) {
        phxo_manifold* m = &w->manifolds[i];
        if (m->point_count == 0 && !aabb_intersects(&w->bodies[m->body1], &w->bodies[m->body2])) {
            ps_erase(&w->set, (uint32_t)m->body1, (uint32_t)m->body2);
            phxo_manifold last = w->manifolds[w->nm - 1];
            int32_t slot = m->point_index;
            for (int k = 0; k < last.point_count; ++k) w->cps[slot + k] = w->cps[last.point_index + k];
            *m = last;
            m->point_index = slot;
            w->nm--;
        } else ++i;
    }
    w->ncp = w->nm * 2;
}

static void refresh_contact_joints(phxo_world* w) /* ref: World.cpp:72-149 */
{
    for (size_t k = 0; k < w->nj; ++k) w->joints[k].contact_point_index = -1;
    for (size_t mi = 0; mi < w->nm; ++mi) {
        const phxo_manifold* m = &w->manifolds[mi];
        for (int k = 0; k < m->point_count; ++k) {
            int32_t cpi = m->point_index + k;
            phxo_contact_point* cp = &w->cps[cpi];
            if (cp->solver_index < 0) {
                cp->solver_index = (int32_t)w->nj;
                GROW(w->joints, w->cap_j, w->nj + 1, phxo_contact_joint);
                phxo_contact_joint j = {cpi, m->body1, m->body2, 0.f, 0.f};
                w->joints[w->nj++] = j;
            } else {
                w->joints[cp->solver_index].contact_point_index = cpi;
            }
        }
    }
    for (size_t k = 0; k < w->nj;) {
        if (w->joints[k].contact_point_index < 0) { w->joints[k] = w->joints[w->nj - 1]; w->nj--; }
        else { w->cps[w->joints[k].contact_point_index].solver_index = (int32_t)k; ++k; }
    }
}

void phxo_world_pre_solve(phxo_world* w, float dt) /* ref: World.cpp:25-32 */
{
    integrate_velocity(w, dt);
    if (w->nb > w->cap_bp) {
        w->cap_bp = w->nb + w->nb / 2 + 16;
        w->sorted = (phxo_sort_entry*)realloc(w->sorted, w->cap_bp * sizeof(phxo_sort_entry));
        w->entries = (phxo_bp_entry*)realloc(w->entries, w->cap_bp * sizeof(phxo_bp_entry));
    }
    phxo_broadphase_build(w->bodies, w->nb, NULL, w->sorted, w->entries);
    update_pairs(w);
    /* UpdateManifolds, ref: Collider.cpp:368-377 */
    GROW(w->cps, w->cap_cp, w->nm * 2 + 2, phxo_contact_point);
    for (size_t k = w->ncp; k < w->nm * 2; ++k) { memset(&w->cps[k], 0, sizeof(phxo_contact_point)); w->cps[k].solver_index = -1; }
    w->ncp = w->nm * 2;
    for (size_t mi = 0; mi < w->nm; ++mi)
        w->point_overflows += update_manifold(&w->manifolds[mi], w->bodies, w->cps + w->manifolds[mi].point_index);
    pack_manifolds(w);
    refresh_contact_joints(w);
}

void phxo_world_solve_and_integrate(phxo_world* w, float dt, int solve_mode, int island_mode, int ci, int pi)
{
    phxo_solver_solve(w->bodies, (int)w->nb, w->cps, w->joints, (int)w->nj, solve_mode, island_mode, ci, pi, NULL, 0, &w->stats);
    phxo_world_integrate_position(w, dt);
}

void phxo_world_update(phxo_world* w, float dt, int solve_mode, int island_mode, int ci, int pi) /* ref: World.cpp:19-37 */
{
    phxo_world_pre_solve(w, dt);
    phxo_world_solve_and_integrate(w, dt, solve_mode, island_mode, ci, pi);
}

phxo_body* phxo_world_bodies(phxo_world* w, int* n) { if (n) *n = (int)w->nb; return w->bodies; }
phxo_manifold* phxo_world_manifolds(phxo_world* w, int* n) { if (n) *n = (int)w->nm; return w->manifolds; }
phxo_contact_point* phxo_world_contact_points(phxo_world* w, int* n) { if (n) *n = (int)w->ncp; return w->cps; }
phxo_contact_joint* phxo_world_joints(phxo_world* w, int* n) { if (n) *n = (int)w->nj; return w->joints; }
const phxo_sort_entry* phxo_world_sorted(phxo_world* w, int* n) { if (n) *n = (int)w->nb; return w->sorted; }
const phxo_bp_entry* phxo_world_bp_entries(phxo_world* w, int* n) { if (n) *n = (int)w->nb; return w->entries; }
const uint32_t* phxo_world_new_pairs(phxo_world* w, int* np) { if (np) *np = (int)w->n_new; return w->new_pairs; }
const phxo_solve_stats* phxo_world_stats(phxo_world* w) { return &w->stats; }
uint64_t phxo_world_sweep_tests(phxo_world* w) { return w->sweep_tests; }
int phxo_world_point_overflows(phxo_world* w) { return w->point_overflows; }

/* ------------------------------------------------------------------------------------------ */
/* multi-threaded timing harness (cpu_baseline leg of bench.py only)                           */
/* Persistent worker threads meeting at a spin barrier once per sweep, 512-joint batches dealt  */
/* round-robin — the shape of the reference's Single Sloppy mode (ref: Solver.cpp:138-139,      */
/* 181-187; base/Parallel.h:27-103), including its data races on shared bodies.                 */

typedef struct {
    sctx* c;
    int nj, iters, threads;
    volatile int arrived, generation, stop_after;   /* barrier state; stop_after = sweeps actually run */
    volatile int any[2];
} tshared;

typedef struct { tshared* sh; int tid; } tjob;

static void spin_barrier(tshared* sh, int* local_gen)
{
    int gen = *local_gen;
    if (__atomic_add_fetch(&sh->arrived, 1, __ATOMIC_ACQ_REL) == sh->threads) {
        sh->arrived = 0;
        __atomic_store_n(&sh->generation, gen + 1, __ATOMIC_RELEASE);
    } else {
        int spins = 0;
        while (__atomic_load_n(&sh->generation, __ATOMIC_ACQUIRE) == gen)
            if (++spins > 2000) { sched_yield(); spins = 0; }
    }
    *local_gen = gen + 1;
}

static void* tworker(void* p)
{
    tjob* j = (tjob*)p;
    tshared* sh = j->sh;
    int gen = 0;
    for (int it = 0; it < sh->iters; ++it) {
        int any = 0;
        for (int b = j->tid * 512; b < sh->nj; b += sh->threads * 512) {
            int e = b + 512 < sh->nj ? b + 512 : sh->nj;
            any |= sweep(sh->c, b, e, it, 1, 0);
        }
        if (any) sh->any[it & 1] = 1;
        spin_barrier(sh, &gen);
        int go = sh->any[it & 1];
        spin_barrier(sh, &gen);
        if (j->tid == 0) { sh->any[(it + 1) & 1] = 0; sh->stop_after = it + 1; }
        if (!go) break;                                   /* ref: Solver.cpp:189 */
    }
    return NULL;
}

double phxo_time_impulse_loop(phxo_body* bodies, int nb, const phxo_contact_point* cps, phxo_contact_joint* joints, int nj,
                              int iters, int threads, int64_t* joint_visits)
{
    if (threads < 1) threads = 1;
    sctx c; ctx_alloc(&c, nb, nj);
    c.cps = cps; c.joints = joints; c.nj = nj; c.st = NULL; c.stag_mode = PHXO_STAG_SEQUENTIAL;
    prepare_bodies(&c, bodies);
    for (int i = 0; i < nj; ++i) c.joint_index[i] = i;
    copy_joints_in(&c, 0, nj);
    for (int s = 0; s < nj; ++s) refresh_one(&c.pj[s], c.imp, c.par, c.cps);
    for (int s = 0; s < nj; ++s) prestep_one(&c.pj[s], c.imp, 0, c.is_static);

    tshared sh; memset(&sh, 0, sizeof sh);
    sh.c = &c; sh.nj = nj; sh.iters = iters; sh.threads = threads;
    pthread_t* th = (pthread_t*)malloc(threads * sizeof(pthread_t));
    tjob* jobs = (tjob*)malloc(threads * sizeof(tjob));
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 1; t < threads; ++t) { jobs[t].sh = &sh; jobs[t].tid = t; pthread_create(&th[t], NULL, tworker, &jobs[t]); }
    jobs[0].sh = &sh; jobs[0].tid = 0;
    tworker(&jobs[0]);                                     /* the caller works too (ref: base/Parallel.h:94) */
    for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (joint_visits) *joint_visits = (int64_t)sh.stop_after * nj;
    free(th); free(jobs);
    ctx_free(&c);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
