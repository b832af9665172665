/*
 * cpu_baseline.c — the CPU baseline that stands beside the GPU numbers: an 8-wide, AVX2-order
 * restatement of the reference's hot path, built with the reference's own flags (-O3 -DNDEBUG -ffast-math -mavx2 -mfma,
 * ref: /root/reference/Makefile:11,17-24) into oracle/libcpubaseline_fast.so, and once more with the oracle's strict flags
 * into oracle/libcpubaseline_strict.so so that tests can check it BIT FOR BIT against phx_oracle.c's AVX2 mode.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY: bench.py's cpu_baseline leg and tests/ use it; the product (phyx_amd/) never does.
 *
 * What it restates (ref = /root/reference/src):
 *   Solver::SolveJoints<8>, Single and Single-Sloppy island modes (Solver.cpp:68-119), i.e.
 *     PrepareBodies (:456-480), PrepareIndices greedy 8-grouping (:217-273), PrepareJoints (:496-521),
 *     RefreshJoints<8>/<1> (:592-695), PreStepJoints (:697-758), SolveJointsImpulses (:760-914) with the group-granular
 *     skip (:797), SolveJointsDisplacement (:916-1018), FinishJoints (:527-547), FinishBodies (:482-494),
 *     on the AoSoA layout ContactJointPacked<8> (Solver.h:7-45: 35 words per joint) and SolveBody / SolveBodyParams
 *     (Solver.h:86-103), 512-joint batches handed to a pool of persistent threads in the Sloppy mode (:138-139);
 *   Collider::UpdateBroadphase (Collider.cpp:251-284) with radixSort3's 11/11/10 split (base/RadixSort.h:28-95) and
 *   Collider::UpdatePairs (Collider.cpp:286-366) against a persistent pair set (steady state: lookups only).
 * The 8-wide paths are written with AVX2 intrinsics, operation for operation what the reference writes with its own V8f
 * wrapper (base/SIMD_AVX2.h: 128-bit loads of the 16-byte SolveBody records + an 8x4 transpose for the gathers, :328-377);
 * multiplies and adds are separate intrinsics, so the strict build rounds every operation (bit-equal to the scalar oracle)
 * and the fast build may contract them into FMAs exactly as the reference's build does.  The scalar tails (vn = 1) and the
 * refresh stage are plain C.  Parity is unpinned in the same sense as the oracle's: the
 * reference's .cpp files cannot be built here, so the timing cannot be calibrated against them.
 */
#define _GNU_SOURCE
#include <immintrin.h>
#include <limits.h>
#define _GNU_SOURCE
#include <linux/futex.h>
#include <sched.h>
#include <stdio.h>
#include <math.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "phx_oracle.h"

#define N 8

typedef struct { float vx, vy, w; int32_t tag; } sbody;                                 /* ref: Solver.h:95-101 SolveBody */
typedef struct { float im, ii, px, py, xvx, xvy, yvx, yvy; } sparam;                    /* ref: Solver.h:86-93 SolveBodyParams */
typedef struct { float p1x[N], p1y[N], p2x[N], p2y[N], a1[N], a2[N], c1x[N], c1y[N], c2x[N], c2y[N], c1a[N], c2a[N], cim[N]; } lim8;
typedef struct {                                                                         /* ref: Solver.h:26-45 ContactJointPacked<8> */
    int32_t b1[N], b2[N], cp[N];
    lim8 n;
    float n_cim_dup[N], n_acc[N], n_dst[N], n_dstd[N], n_accd[N];
    lim8 f;
    float f_acc[N];
} pack8;

typedef struct {
    double prepare_bodies, prepare_indices, prepare_joints, refresh, prestep, impulse, displacement, finish, total;
    int32_t impulse_iterations, displacement_iterations, group_offset, threads;
    int64_t joint_visits;
} phxb_phases;

typedef struct {
    double update_broadphase, update_pairs;
    int64_t candidate_tests, overlapping_pairs;
    int32_t threads, reps;
} phxb_broadphase_phases;

/* The reference's executable is linked with -ffast-math, so it runs flush-to-zero / denormals-are-zero (crtfastmath.o).  This
 * library must not change the loading process's mode, so the fast build sets the two MXCSR bits around its own timed calls
 * (threads created inside inherit them) and restores the caller's MXCSR afterwards. */
#ifdef __FAST_MATH__
#define FAST_MODE_ENTER() const unsigned saved_mxcsr_ = _mm_getcsr(); _mm_setcsr(saved_mxcsr_ | 0x8040u)
#define FAST_MODE_LEAVE() _mm_setcsr(saved_mxcsr_)
#else
#define FAST_MODE_ENTER() do { } while (0)
#define FAST_MODE_LEAVE() do { } while (0)
#endif

static double now_s(void)
{
    struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

static inline float maxf_ref(float l, float r) { return l > r ? l : r; }                 /* ref: base/SIMD_Scalar.h:275-278 */
static inline float flipsign_simd(float x, float y)                                      /* ref: base/SIMD_AVX2.h:272-275 */
{
    uint32_t xb, yb; memcpy(&xb, &x, 4); memcpy(&yb, &y, 4);
    xb ^= yb & 0x80000000u; memcpy(&x, &xb, 4); return x;
}
static inline float flipsign_scalar(float x, float y) { return y < 0.f ? -x : x; }       /* ref: base/SIMD_Scalar.h:265-268 */

/* ---- solver state ---------------------------------------------------------------------------------------------- */
typedef struct {
    int nb, nj;
    sbody *imp, *disp;
    sparam* par;
    pack8* packs;
    int32_t* joint_index;
    const phxo_contact_point* cps;
    phxo_contact_joint* joints;
    int group_offset;
} ctx;

/* ref: Solver.cpp:549-590 RefreshLimiter, lanes [ip, ip + vn) */
static inline void refresh_limiter(lim8* L, int ip, int vn, const float* n1x, const float* n1y, const float* w1x, const float* w1y,
                                   const float* w2x, const float* w2y, const float* im1, const float* ii1, const float* im2, const float* ii2)
{
    for (int l = 0; l < vn; ++l) {
        const float p1x = n1x[l], p1y = n1y[l], p2x = -n1x[l], p2y = -n1y[l];
        const float a1 = p1x * w1y[l] - p1y * w1x[l];
        const float a2 = p2x * w2y[l] - p2y * w2x[l];
        const float c1x = p1x * im1[l], c1y = p1y * im1[l], c1a = a1 * ii1[l];
        const float c2x = p2x * im2[l], c2y = p2y * im2[l], c2a = a2 * ii2[l];
        const float m1 = p1x * c1x + p1y * c1y + a1 * c1a;
        const float m2 = p2x * c2x + p2y * c2y + a2 * c2a;
        const float m = m1 + m2;
        const int k = ip + l;
        L->p1x[k] = p1x; L->p1y[k] = p1y; L->p2x[k] = p2x; L->p2y[k] = p2y; L->a1[k] = a1; L->a2[k] = a2;
        L->c1x[k] = c1x; L->c1y[k] = c1y; L->c2x[k] = c2x; L->c2y[k] = c2y; L->c1a[k] = c1a; L->c2a[k] = c2a;
        L->cim[k] = fabsf(m) > 0.f ? 1.0f / m : 0.f;
    }
}

/* ref: Solver.cpp:592-695 RefreshJoints<VN,8>, joints [jb, je) in steps of vn */
static void refresh_joints(ctx* c, int jb, int je, int vn)
{
    for (int j = jb; j < je; j += vn) {
        pack8* P = &c->packs[(unsigned)j / N];
        const int ip = vn == N ? 0 : (j & (N - 1));
        float d1x[N], d1y[N], d2x[N], d2y[N], nx[N], ny[N], tx[N], ty[N], w1x[N], w1y[N], w2x[N], w2y[N];
        float im1[N], ii1[N], im2[N], ii2[N], dst[N], dstd[N];
        for (int l = 0; l < vn; ++l) {
            const sparam *q1 = &c->par[P->b1[ip + l]], *q2 = &c->par[P->b2[ip + l]];
            const sbody *v1 = &c->imp[P->b1[ip + l]], *v2 = &c->imp[P->b2[ip + l]];
            const phxo_contact_point* cp = &c->cps[P->cp[ip + l]];
            d1x[l] = cp->delta1.x; d1y[l] = cp->delta1.y; d2x[l] = cp->delta2.x; d2y[l] = cp->delta2.y;
            nx[l] = cp->normal.x; ny[l] = cp->normal.y;
            im1[l] = q1->im; ii1[l] = q1->ii; im2[l] = q2->im; ii2[l] = q2->ii;
            const float p1x = d1x[l] + q1->px, p1y = d1y[l] + q1->py;
            const float p2x = d2x[l] + q2->px, p2y = d2y[l] + q2->py;
            w1x[l] = d1x[l]; w1y[l] = d1y[l];
            w2x[l] = p1x - q2->px; w2y[l] = p1y - q2->py;                                 /* ref: :649-650, body-1's point (sic) */
            const float bounce = 0.f, delta_velocity = 1.f, max_pen_velocity = 0.1f, delta_depth = 1.f, error_reduction = 0.1f;
            const float pv1x = (q1->py - p1y) * v1->w + v1->vx, pv1y = (p1x - q1->px) * v1->w + v1->vy;
            const float pv2x = (q2->py - p2y) * v2->w + v2->vx, pv2y = (p2x - q2->px) * v2->w + v2->vy;
            const float rvx = pv1x - pv2x, rvy = pv1y - pv2y;
            const float dv = -bounce * (rvx * nx[l] + rvy * ny[l]);
            const float depth = (p2x - p1x) * nx[l] + (p2y - p1y) * ny[l];
            const float d = maxf_ref(dv - delta_velocity, 0.f);
            dst[l] = depth < delta_depth ? d - max_pen_velocity : d;
            dstd[l] = error_reduction * maxf_ref(0.f, depth - 2.0f * delta_depth);
            tx[l] = -ny[l]; ty[l] = nx[l];
        }
        refresh_limiter(&P->n, ip, vn, nx, ny, w1x, w1y, w2x, w2y, im1, ii1, im2, ii2);
        refresh_limiter(&P->f, ip, vn, tx, ty, w1x, w1y, w2x, w2y, im1, ii1, im2, ii2);
        for (int l = 0; l < vn; ++l) { P->n_dst[ip + l] = dst[l]; P->n_dstd[ip + l] = dstd[l]; P->n_accd[ip + l] = 0.f; }
    }
}

/* ---- 8-wide AVX2 forms ------------------------------------------------------------------------------------------- */
/* 8 SolveBody records -> {vx, vy, w, tag} lanes (ref: base/SIMD_AVX2.h:328-335 loadindexed4) */
static inline void gather8(const sbody* base, const int32_t* idx, __m256* x, __m256* y, __m256* z, __m256* t)
{
    const __m256 a0 = _mm256_insertf128_ps(_mm256_castps128_ps256(_mm_load_ps((const float*)&base[idx[0]])), _mm_load_ps((const float*)&base[idx[4]]), 1);
    const __m256 a1 = _mm256_insertf128_ps(_mm256_castps128_ps256(_mm_load_ps((const float*)&base[idx[1]])), _mm_load_ps((const float*)&base[idx[5]]), 1);
    const __m256 a2 = _mm256_insertf128_ps(_mm256_castps128_ps256(_mm_load_ps((const float*)&base[idx[2]])), _mm_load_ps((const float*)&base[idx[6]]), 1);
    const __m256 a3 = _mm256_insertf128_ps(_mm256_castps128_ps256(_mm_load_ps((const float*)&base[idx[3]])), _mm_load_ps((const float*)&base[idx[7]]), 1);
    const __m256 t0 = _mm256_unpacklo_ps(a0, a1), t1 = _mm256_unpackhi_ps(a0, a1), t2 = _mm256_unpacklo_ps(a2, a3), t3 = _mm256_unpackhi_ps(a2, a3);
    *x = _mm256_shuffle_ps(t0, t2, 0x44); *y = _mm256_shuffle_ps(t0, t2, 0xEE); *z = _mm256_shuffle_ps(t1, t3, 0x44); *t = _mm256_shuffle_ps(t1, t3, 0xEE);
}
/* ref: base/SIMD_AVX2.h:370-377 storeindexed4 */
static inline void scatter8(sbody* base, const int32_t* idx, __m256 x, __m256 y, __m256 z, __m256 t)
{
    const __m256 t0 = _mm256_unpacklo_ps(x, y), t1 = _mm256_unpackhi_ps(x, y), t2 = _mm256_unpacklo_ps(z, t), t3 = _mm256_unpackhi_ps(z, t);
    const __m256 r0 = _mm256_shuffle_ps(t0, t2, 0x44), r1 = _mm256_shuffle_ps(t0, t2, 0xEE), r2 = _mm256_shuffle_ps(t1, t3, 0x44), r3 = _mm256_shuffle_ps(t1, t3, 0xEE);
    _mm_store_ps((float*)&base[idx[0]], _mm256_castps256_ps128(r0)); _mm_store_ps((float*)&base[idx[4]], _mm256_extractf128_ps(r0, 1));
    _mm_store_ps((float*)&base[idx[1]], _mm256_castps256_ps128(r1)); _mm_store_ps((float*)&base[idx[5]], _mm256_extractf128_ps(r1, 1));
    _mm_store_ps((float*)&base[idx[2]], _mm256_castps256_ps128(r2)); _mm_store_ps((float*)&base[idx[6]], _mm256_extractf128_ps(r2, 1));
    _mm_store_ps((float*)&base[idx[3]], _mm256_castps256_ps128(r3)); _mm_store_ps((float*)&base[idx[7]], _mm256_extractf128_ps(r3, 1));
}
#define LD(p) _mm256_loadu_ps(p)
#define MUL _mm256_mul_ps
#define ADD _mm256_add_ps
#define SUB _mm256_sub_ps
static inline __m256 abs8(__m256 v) { return _mm256_andnot_ps(_mm256_set1_ps(-0.0f), v); }
static inline int productive_lanes(__m256 t1, __m256 t2, int iter)             /* ref: Solver.cpp:790-798 */
{
    const __m256i lim = _mm256_set1_epi32(iter - 2);
    const __m256i p = _mm256_or_si256(_mm256_cmpgt_epi32(_mm256_castps_si256(t1), lim), _mm256_cmpgt_epi32(_mm256_castps_si256(t2), lim));
    return _mm256_movemask_ps(_mm256_castsi256_ps(p));
}

/* ref: Solver.cpp:697-758 PreStepJoints<8,8> */
static void prestep_joints8(ctx* c, int jb, int je)
{
    for (int j = jb; j < je; j += N) {
        pack8* P = &c->packs[(unsigned)j / N];
        __m256 v1x, v1y, w1, t1, v2x, v2y, w2, t2;
        gather8(c->imp, P->b1, &v1x, &v1y, &w1, &t1);
        gather8(c->imp, P->b2, &v2x, &v2y, &w2, &t2);
        const __m256 an = LD(P->n_acc), af = LD(P->f_acc);
        v1x = ADD(v1x, MUL(LD(P->n.c1x), an)); v1y = ADD(v1y, MUL(LD(P->n.c1y), an)); w1 = ADD(w1, MUL(LD(P->n.c1a), an));
        v2x = ADD(v2x, MUL(LD(P->n.c2x), an)); v2y = ADD(v2y, MUL(LD(P->n.c2y), an)); w2 = ADD(w2, MUL(LD(P->n.c2a), an));
        v1x = ADD(v1x, MUL(LD(P->f.c1x), af)); v1y = ADD(v1y, MUL(LD(P->f.c1y), af)); w1 = ADD(w1, MUL(LD(P->f.c1a), af));
        v2x = ADD(v2x, MUL(LD(P->f.c2x), af)); v2y = ADD(v2y, MUL(LD(P->f.c2y), af)); w2 = ADD(w2, MUL(LD(P->f.c2a), af));
        scatter8(c->imp, P->b1, v1x, v1y, w1, t1);
        scatter8(c->imp, P->b2, v2x, v2y, w2, t2);
    }
}

/* ref: Solver.cpp:760-914 SolveJointsImpulses<8,8> */
static int solve_impulses8(ctx* c, int jb, int je, int iter)
{
    int any = 0;
    const __m256 iterf = _mm256_castsi256_ps(_mm256_set1_epi32(iter));
    for (int j = jb; j < je; j += N) {
        pack8* P = &c->packs[(unsigned)j / N];
        __m256 v1x, v1y, w1, t1, v2x, v2y, w2, t2;
        gather8(c->imp, P->b1, &v1x, &v1y, &w1, &t1);
        gather8(c->imp, P->b2, &v2x, &v2y, &w2, &t2);
        if (!productive_lanes(t1, t2, iter)) continue;                                    /* group-granular skip, ref: :797 */
        __m256 dv = LD(P->n_dst);
        dv = SUB(dv, MUL(LD(P->n.p1x), v1x)); dv = SUB(dv, MUL(LD(P->n.p1y), v1y)); dv = SUB(dv, MUL(LD(P->n.a1), w1));
        dv = SUB(dv, MUL(LD(P->n.p2x), v2x)); dv = SUB(dv, MUL(LD(P->n.p2y), v2y)); dv = SUB(dv, MUL(LD(P->n.a2), w2));
        __m256 nacc = LD(P->n_acc);
        __m256 dn = MUL(dv, LD(P->n.cim));
        dn = _mm256_max_ps(dn, _mm256_xor_ps(nacc, _mm256_set1_ps(-0.0f)));
        v1x = ADD(v1x, MUL(LD(P->n.c1x), dn)); v1y = ADD(v1y, MUL(LD(P->n.c1y), dn)); w1 = ADD(w1, MUL(LD(P->n.c1a), dn));
        v2x = ADD(v2x, MUL(LD(P->n.c2x), dn)); v2y = ADD(v2y, MUL(LD(P->n.c2y), dn)); w2 = ADD(w2, MUL(LD(P->n.c2a), dn));
        nacc = ADD(nacc, dn);
        _mm256_storeu_ps(P->n_acc, nacc);
        __m256 fv = _mm256_setzero_ps();
        fv = SUB(fv, MUL(LD(P->f.p1x), v1x)); fv = SUB(fv, MUL(LD(P->f.p1y), v1y)); fv = SUB(fv, MUL(LD(P->f.a1), w1));
        fv = SUB(fv, MUL(LD(P->f.p2x), v2x)); fv = SUB(fv, MUL(LD(P->f.p2y), v2y)); fv = SUB(fv, MUL(LD(P->f.a2), w2));
        __m256 df = MUL(fv, LD(P->f.cim));
        const __m256 facc = LD(P->f_acc);
        const __m256 force = ADD(facc, df);
        const __m256 limit = MUL(nacc, _mm256_set1_ps(0.3f));
        const __m256 signed_limit = _mm256_xor_ps(limit, _mm256_and_ps(force, _mm256_set1_ps(-0.0f)));      /* flipsign, ref: base/SIMD_AVX2.h:272-275 */
        const __m256 adjusted = SUB(signed_limit, facc);
        df = _mm256_blendv_ps(df, adjusted, _mm256_cmp_ps(abs8(force), limit, _CMP_GT_OQ));
        _mm256_storeu_ps(P->f_acc, ADD(facc, df));
        v1x = ADD(v1x, MUL(LD(P->f.c1x), df)); v1y = ADD(v1y, MUL(LD(P->f.c1y), df)); w1 = ADD(w1, MUL(LD(P->f.c1a), df));
        v2x = ADD(v2x, MUL(LD(P->f.c2x), df)); v2y = ADD(v2y, MUL(LD(P->f.c2y), df)); w2 = ADD(w2, MUL(LD(P->f.c2a), df));
        const __m256 prod = _mm256_cmp_ps(_mm256_max_ps(abs8(dn), abs8(df)), _mm256_set1_ps(1e-4f), _CMP_GT_OQ);
        t1 = _mm256_blendv_ps(t1, iterf, prod); t2 = _mm256_blendv_ps(t2, iterf, prod);
        any |= _mm256_movemask_ps(prod);
        scatter8(c->imp, P->b1, v1x, v1y, w1, t1);
        scatter8(c->imp, P->b2, v2x, v2y, w2, t2);
    }
    return any != 0;
}

/* ref: Solver.cpp:916-1018 SolveJointsDisplacement<8,8> */
static int solve_displacement8(ctx* c, int jb, int je, int iter)
{
    int any = 0;
    const __m256 iterf = _mm256_castsi256_ps(_mm256_set1_epi32(iter));
    for (int j = jb; j < je; j += N) {
        pack8* P = &c->packs[(unsigned)j / N];
        __m256 v1x, v1y, w1, t1, v2x, v2y, w2, t2;
        gather8(c->disp, P->b1, &v1x, &v1y, &w1, &t1);
        gather8(c->disp, P->b2, &v2x, &v2y, &w2, &t2);
        if (!productive_lanes(t1, t2, iter)) continue;
        __m256 dv = LD(P->n_dstd);
        dv = SUB(dv, MUL(LD(P->n.p1x), v1x)); dv = SUB(dv, MUL(LD(P->n.p1y), v1y)); dv = SUB(dv, MUL(LD(P->n.a1), w1));
        dv = SUB(dv, MUL(LD(P->n.p2x), v2x)); dv = SUB(dv, MUL(LD(P->n.p2y), v2y)); dv = SUB(dv, MUL(LD(P->n.a2), w2));
        __m256 accd = LD(P->n_accd);
        __m256 di = MUL(dv, LD(P->n.cim));
        di = _mm256_max_ps(di, _mm256_xor_ps(accd, _mm256_set1_ps(-0.0f)));
        v1x = ADD(v1x, MUL(LD(P->n.c1x), di)); v1y = ADD(v1y, MUL(LD(P->n.c1y), di)); w1 = ADD(w1, MUL(LD(P->n.c1a), di));
        v2x = ADD(v2x, MUL(LD(P->n.c2x), di)); v2y = ADD(v2y, MUL(LD(P->n.c2y), di)); w2 = ADD(w2, MUL(LD(P->n.c2a), di));
        _mm256_storeu_ps(P->n_accd, ADD(accd, di));
        const __m256 prod = _mm256_cmp_ps(abs8(di), _mm256_set1_ps(1e-4f), _CMP_GT_OQ);
        t1 = _mm256_blendv_ps(t1, iterf, prod); t2 = _mm256_blendv_ps(t2, iterf, prod);
        any |= _mm256_movemask_ps(prod);
        scatter8(c->disp, P->b1, v1x, v1y, w1, t1);
        scatter8(c->disp, P->b2, v2x, v2y, w2, t2);
    }
    return any != 0;
}

/* gather / scatter of the 16-byte SolveBody records of `vn` lanes (ref: base/SIMD_AVX2.h:328-377 loadindexed4 / storeindexed4) */
#define GATHER(arr, idx)                                                                                              \
    float vx[N], vy[N], w[N]; int32_t tag[N];                                                                          \
    for (int l = 0; l < vn; ++l) { const sbody* b_ = &(arr)[(idx)[ip + l]]; vx[l] = b_->vx; vy[l] = b_->vy; w[l] = b_->w; tag[l] = b_->tag; }

/* ref: Solver.cpp:697-758 PreStepJoints */
static void prestep_joints(ctx* c, int jb, int je, int vn)
{
    for (int j = jb; j < je; j += vn) {
        pack8* P = &c->packs[(unsigned)j / N];
        const int ip = vn == N ? 0 : (j & (N - 1));
        float v1x[N], v1y[N], w1[N], v2x[N], v2y[N], w2[N]; int32_t t1[N], t2[N];
        for (int l = 0; l < vn; ++l) {
            const sbody *a = &c->imp[P->b1[ip + l]], *b = &c->imp[P->b2[ip + l]];
            v1x[l] = a->vx; v1y[l] = a->vy; w1[l] = a->w; t1[l] = a->tag; v2x[l] = b->vx; v2y[l] = b->vy; w2[l] = b->w; t2[l] = b->tag;
        }
        for (int l = 0; l < vn; ++l) {
            const int k = ip + l;
            const float an = P->n_acc[k], af = P->f_acc[k];
            v1x[l] += P->n.c1x[k] * an; v1y[l] += P->n.c1y[k] * an; w1[l] += P->n.c1a[k] * an;
            v2x[l] += P->n.c2x[k] * an; v2y[l] += P->n.c2y[k] * an; w2[l] += P->n.c2a[k] * an;
            v1x[l] += P->f.c1x[k] * af; v1y[l] += P->f.c1y[k] * af; w1[l] += P->f.c1a[k] * af;
            v2x[l] += P->f.c2x[k] * af; v2y[l] += P->f.c2y[k] * af; w2[l] += P->f.c2a[k] * af;
        }
        for (int l = 0; l < vn; ++l) {
            sbody *a = &c->imp[P->b1[ip + l]], *b = &c->imp[P->b2[ip + l]];
            a->vx = v1x[l]; a->vy = v1y[l]; a->w = w1[l]; a->tag = t1[l]; b->vx = v2x[l]; b->vy = v2y[l]; b->w = w2[l]; b->tag = t2[l];
        }
    }
}

/* ref: Solver.cpp:760-914 SolveJointsImpulses<VN,8>; returns any(productive) */
static int solve_impulses(ctx* c, int jb, int je, int vn, int iter)
{
    int any = 0;
    for (int j = jb; j < je; j += vn) {
        pack8* P = &c->packs[(unsigned)j / N];
        const int ip = vn == N ? 0 : (j & (N - 1));
        float v1x[N], v1y[N], w1[N], v2x[N], v2y[N], w2[N]; int32_t t1[N], t2[N];
        int go = 0;
        for (int l = 0; l < vn; ++l) {
            const sbody *a = &c->imp[P->b1[ip + l]], *b = &c->imp[P->b2[ip + l]];
            v1x[l] = a->vx; v1y[l] = a->vy; w1[l] = a->w; t1[l] = a->tag; v2x[l] = b->vx; v2y[l] = b->vy; w2[l] = b->w; t2[l] = b->tag;
            go |= (t1[l] > iter - 2) | (t2[l] > iter - 2);                               /* ref: :790-798 none(...) over the lanes */
        }
        if (!go) continue;
        int prod[N];
        for (int l = 0; l < vn; ++l) {
            const int k = ip + l;
            float dv = P->n_dst[k];
            dv -= P->n.p1x[k] * v1x[l]; dv -= P->n.p1y[k] * v1y[l]; dv -= P->n.a1[k] * w1[l];
            dv -= P->n.p2x[k] * v2x[l]; dv -= P->n.p2y[k] * v2y[l]; dv -= P->n.a2[k] * w2[l];
            float dn = dv * P->n.cim[k];
            dn = maxf_ref(dn, -P->n_acc[k]);
            v1x[l] += P->n.c1x[k] * dn; v1y[l] += P->n.c1y[k] * dn; w1[l] += P->n.c1a[k] * dn;
            v2x[l] += P->n.c2x[k] * dn; v2y[l] += P->n.c2y[k] * dn; w2[l] += P->n.c2a[k] * dn;
            P->n_acc[k] += dn;
            float fv = 0.f;
            fv -= P->f.p1x[k] * v1x[l]; fv -= P->f.p1y[k] * v1y[l]; fv -= P->f.a1[k] * w1[l];
            fv -= P->f.p2x[k] * v2x[l]; fv -= P->f.p2y[k] * v2y[l]; fv -= P->f.a2[k] * w2[l];
            float df = fv * P->f.cim[k];
            const float reaction = P->n_acc[k], acc = P->f_acc[k];
            const float force = acc + df;
            const float limit = reaction * 0.3f;                                          /* kFrictionCoefficient, ref: :9 */
            const float signed_limit = vn > 1 ? flipsign_simd(limit, force) : flipsign_scalar(limit, force);
            const float adjusted = signed_limit - acc;
            if (fabsf(force) > limit) df = adjusted;
            P->f_acc[k] += df;
            v1x[l] += P->f.c1x[k] * df; v1y[l] += P->f.c1y[k] * df; w1[l] += P->f.c1a[k] * df;
            v2x[l] += P->f.c2x[k] * df; v2y[l] += P->f.c2y[k] * df; w2[l] += P->f.c2a[k] * df;
            prod[l] = maxf_ref(fabsf(dn), fabsf(df)) > 1e-4f;                             /* kProductiveImpulse, ref: :8 */
        }
        for (int l = 0; l < vn; ++l) {
            sbody *a = &c->imp[P->b1[ip + l]], *b = &c->imp[P->b2[ip + l]];
            if (prod[l]) { t1[l] = iter; t2[l] = iter; any = 1; }
            a->vx = v1x[l]; a->vy = v1y[l]; a->w = w1[l]; a->tag = t1[l]; b->vx = v2x[l]; b->vy = v2y[l]; b->w = w2[l]; b->tag = t2[l];
        }
    }
    return any;
}

/* ref: Solver.cpp:916-1018 SolveJointsDisplacement */
static int solve_displacement(ctx* c, int jb, int je, int vn, int iter)
{
    int any = 0;
    for (int j = jb; j < je; j += vn) {
        pack8* P = &c->packs[(unsigned)j / N];
        const int ip = vn == N ? 0 : (j & (N - 1));
        float v1x[N], v1y[N], w1[N], v2x[N], v2y[N], w2[N]; int32_t t1[N], t2[N];
        int go = 0;
        for (int l = 0; l < vn; ++l) {
            const sbody *a = &c->disp[P->b1[ip + l]], *b = &c->disp[P->b2[ip + l]];
            v1x[l] = a->vx; v1y[l] = a->vy; w1[l] = a->w; t1[l] = a->tag; v2x[l] = b->vx; v2y[l] = b->vy; w2[l] = b->w; t2[l] = b->tag;
            go |= (t1[l] > iter - 2) | (t2[l] > iter - 2);
        }
        if (!go) continue;
        int prod[N];
        for (int l = 0; l < vn; ++l) {
            const int k = ip + l;
            float dv = P->n_dstd[k];
            dv -= P->n.p1x[k] * v1x[l]; dv -= P->n.p1y[k] * v1y[l]; dv -= P->n.a1[k] * w1[l];
            dv -= P->n.p2x[k] * v2x[l]; dv -= P->n.p2y[k] * v2y[l]; dv -= P->n.a2[k] * w2[l];
            float di = dv * P->n.cim[k];
            di = maxf_ref(di, -P->n_accd[k]);
            v1x[l] += P->n.c1x[k] * di; v1y[l] += P->n.c1y[k] * di; w1[l] += P->n.c1a[k] * di;
            v2x[l] += P->n.c2x[k] * di; v2y[l] += P->n.c2y[k] * di; w2[l] += P->n.c2a[k] * di;
            P->n_accd[k] += di;
            prod[l] = fabsf(di) > 1e-4f;
        }
        for (int l = 0; l < vn; ++l) {
            sbody *a = &c->disp[P->b1[ip + l]], *b = &c->disp[P->b2[ip + l]];
            if (prod[l]) { t1[l] = iter; t2[l] = iter; any = 1; }
            a->vx = v1x[l]; a->vy = v1y[l]; a->w = w1[l]; a->tag = t1[l]; b->vx = v2x[l]; b->vy = v2y[l]; b->w = w2[l]; b->tag = t2[l];
        }
    }
    return any;
}

/* ref: Solver.cpp:217-273 PrepareIndices, group size 8, over joint_index[0, nj) */
static int prepare_indices8(const phxo_contact_joint* joints, int32_t* joint_index, int32_t* group_bodies, int32_t* work, int nj)
{
    for (int i = 0; i < nj; ++i) work[i] = joint_index[i];
    int tag = 0, remaining = nj, out = 0;
    while (remaining >= N) {
        int got = 0;
        ++tag;
        for (int i = 0; i < remaining && got < N;) {
            const int ji = work[i];
            const phxo_contact_joint* j = &joints[ji];
            if (group_bodies[j->body1] < tag && group_bodies[j->body2] < tag) {
                group_bodies[j->body1] = tag; group_bodies[j->body2] = tag;
                joint_index[out + got++] = ji;
                work[i] = work[remaining - 1];
                --remaining;
            } else ++i;
        }
        out += got;
        if (got < N) break;
    }
    for (int i = 0; i < remaining; ++i) joint_index[out + i] = work[i];
    return out & ~(N - 1);
}

/* ---- persistent thread pool: like the reference's WorkQueue + parallelFor (ref: base/Parallel.h:27-103) a phase is a
 *      shared atomic batch counter that every worker — the caller included, :94 — pulls from, and it is over when every BATCH
 *      is done (:96-102 waits for ready == groupCount), not when every THREAD has arrived: a worker that wakes up late (or not at
 *      all: more threads than free cores) finds nothing to pull and delays nobody.  (Round 3's pool met at a full barrier of all
 *      threads twice per phase: one descheduled thread held everybody, and 128 / 256 threads ran 10-100x slower than 64.) ---- */
typedef struct pool pool;
typedef void (*phase_fn)(pool*, int batch, int worker);
/* One cursor per worker (round 6; a cache line each): phase << 48 | end of the worker's share << 24 | next batch of it.  A phase's
 * batches are dealt to the workers in contiguous shares — worker w sweeps the same joint packs in every phase of every sweep, which
 * it also touched first (ph_zero), so they sit in its core's cache hierarchy and on its socket's memory — and ONE fetch-and-add on a
 * cursor hands out a batch and says by itself whether that batch exists and which phase it belongs to.  A worker that has finished
 * its share pulls from the others' cursors (the phase is over when every BATCH is done, not when every thread has arrived: a worker
 * that wakes up late finds its share taken and delays nobody).  Round 5 pulled everything from one shared ticket with unpinned
 * threads: the rate moved 4x with the thread count (515 M visits/s at 128 threads between 2077 at 64 and 1296 at 256). */
typedef struct { _Atomic unsigned long long v; char pad[56]; } cursor;
struct pool {
    int threads;
    cursor* cur;                            /* [threads] */
    atomic_int phase_word;                  /* the phase number: what sleeping workers wait on (futex) */
    atomic_int done, sleepers, quit;
    phase_fn fn[2];                         /* what phase e runs: fn[e & 1] (a worker holding a batch of phase e keeps the phase open) */
    ctx* c;
    int batch_size, iter;
    atomic_int productive;
    pthread_t* th;
    const int* cpus; int ncpus;             /* the process's CPUs, one hardware thread of every core first (pool_cpus); worker w is pinned to cpus[w % ncpus] */
};

/* the CPUs this process may run on, ordered so that the first entries are one hardware thread of every core (a CPU that leads its
 * thread_siblings_list), the SMT siblings behind them: T <= cores workers then sit on T different cores */
static int pool_cpus(int* out, int cap)
{
    cpu_set_t set; CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) != 0) return 0;
    int n = 0;
    for (int pass = 0; pass < 2; ++pass)
        for (int cpu = 0; cpu < CPU_SETSIZE && n < cap; ++cpu) {
            if (!CPU_ISSET(cpu, &set)) continue;
            int first = cpu;
            char path[96]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
            FILE* f = fopen(path, "r");
            if (f) { if (fscanf(f, "%d", &first) != 1) first = cpu; fclose(f); }
            if ((first == cpu) == (pass == 0)) out[n++] = cpu;
        }
    return n;
}
static void pool_pin(const pool* p, int worker)
{
    if (!p->ncpus) return;
    cpu_set_t one; CPU_ZERO(&one); CPU_SET(p->cpus[worker % p->ncpus], &one);
    (void)sched_setaffinity(0, sizeof one, &one);      /* (the calling thread) */
}

/* the worker's own share, then whatever the others have left (a worker that arrives late simply helps the phase it finds) */
static void pool_pull(pool* p, int worker)
{
    for (int k = 0; k < p->threads; ++k) {
        cursor* cu = &p->cur[(worker + k) % p->threads];
        for (;;) {
            const unsigned long long seen = atomic_load_explicit(&cu->v, memory_order_relaxed);
            if (((unsigned)seen & 0xFFFFFFu) >= ((unsigned)(seen >> 24) & 0xFFFFFFu)) break;      /* (nothing left there: no write to somebody else's line) */
            const unsigned long long t = atomic_fetch_add_explicit(&cu->v, 1ull, memory_order_acq_rel);
            const unsigned e = (unsigned)(t >> 48), n = (unsigned)(t >> 24) & 0xFFFFFFu, b = (unsigned)t & 0xFFFFFFu;
            if (b >= n) break;
            p->fn[e & 1](p, (int)b, worker);
            atomic_fetch_add_explicit(&p->done, 1, memory_order_release);
        }
    }
}

typedef struct { pool* p; int worker; } warg;
/* A worker spins for a few tens of microseconds for the next phase (the gap between two sweeps of the impulse loop), then sleeps on
 * the futex: during the serial phases (PrepareIndices runs on the main thread only, like the reference) idle workers must not
 * burn the cores — and their SMT siblings — the main thread is working on. */
static void* pool_thread(void* a)
{
    pool* p = ((warg*)a)->p; const int worker = ((warg*)a)->worker;
    pool_pin(p, worker);
    unsigned seen = 0;
    for (;;) {
        unsigned now;
        for (int spins = 0; (now = (unsigned)atomic_load_explicit(&p->phase_word, memory_order_acquire)) == seen; ++spins) {
            if (atomic_load_explicit(&p->quit, memory_order_acquire)) return NULL;
            if (spins < 20000) { _mm_pause(); continue; }
            atomic_fetch_add(&p->sleepers, 1);
            syscall(SYS_futex, &p->phase_word, FUTEX_WAIT_PRIVATE, (int)seen, NULL, NULL, 0);
            atomic_fetch_sub(&p->sleepers, 1);
            spins = 0;
        }
        if (atomic_load_explicit(&p->quit, memory_order_acquire)) return NULL;
        seen = now;
        pool_pull(p, worker);
    }
}

/* run one phase: the caller is worker 0 (ref: base/Parallel.h:94) and waits for the batches, not for the threads */
static void pool_run(pool* p, int* phase_counter, phase_fn fn, int batches)
{
    const unsigned phase = (unsigned)++*phase_counter;
    p->fn[phase & 1] = fn;
    atomic_store_explicit(&p->done, 0, memory_order_relaxed);
    for (int w = 0; w < p->threads; ++w) {      /* contiguous shares: worker w = batches [w B / T, (w + 1) B / T) */
        const unsigned long long b0 = (unsigned long long)w * (unsigned)batches / (unsigned)p->threads, b1 = (unsigned long long)(w + 1) * (unsigned)batches / (unsigned)p->threads;
        atomic_store_explicit(&p->cur[w].v, ((unsigned long long)(phase & 0xFFFFu) << 48) | (b1 << 24) | b0, memory_order_release);
    }
    if (p->threads > 1) {
        /* sequentially consistent store, then the (seq_cst) load of `sleepers`: with a release store the load could pass it (x86
         * store -> load reordering), main would read sleepers == 0 while a worker that has just counted itself in still sees the
         * old phase word and sleeps through the phase — never a deadlock (phases complete by batches), but silently fewer threads */
        atomic_store(&p->phase_word, (int)phase);
        if (atomic_load(&p->sleepers)) syscall(SYS_futex, &p->phase_word, FUTEX_WAKE_PRIVATE, INT_MAX, NULL, NULL, 0);
    }
    pool_pull(p, 0);
    while (atomic_load_explicit(&p->done, memory_order_acquire) < batches) _mm_pause();
}

static void pool_stop(pool* p)
{
    atomic_store_explicit(&p->quit, 1, memory_order_release);
    atomic_fetch_add(&p->phase_word, 1);
    syscall(SYS_futex, &p->phase_word, FUTEX_WAKE_PRIVATE, INT_MAX, NULL, NULL, 0);
    for (int t = 1; t < p->threads; ++t) pthread_join(p->th[t], NULL);
}

static inline void batch_range(pool* p, int batch, int* vb, int* ve, int* tb, int* te)
{
    const int nj = p->c->nj, go = p->c->group_offset;
    const int b = batch * p->batch_size, e = b + p->batch_size < nj ? b + p->batch_size : nj;
    *vb = b; *ve = go < e ? go : e; *tb = go > b ? go : b; *te = e;          /* ref: :150-151 vector part, scalar tail */
}
/* first touch: a worker zeroes the joint packs of its share before anybody reads them (the pages land on its socket) */
static void ph_zero(pool* p, int batch, int w)
{
    (void)w; ctx* c = p->c;
    const int b = batch * p->batch_size, e = b + p->batch_size < c->nj ? b + p->batch_size : c->nj;
    if (e > b) memset(&c->packs[(unsigned)b / N], 0, (size_t)((unsigned)(e - 1) / N - (unsigned)b / N + 1) * sizeof(pack8));
}
static void ph_copy_in(pool* p, int batch, int w)
{
    (void)w; ctx* c = p->c;
    const int b = batch * p->batch_size, e = b + p->batch_size < c->nj ? b + p->batch_size : c->nj;
    for (int s = b; s < e; ++s) {                                                        /* ref: :509-521 */
        const phxo_contact_joint* j = &c->joints[c->joint_index[s]];
        pack8* P = &c->packs[(unsigned)s / N]; const int k = s & (N - 1);
        P->b1[k] = j->body1; P->b2[k] = j->body2; P->cp[k] = j->contact_point_index;
        P->n_acc[k] = j->normal_acc; P->f_acc[k] = j->friction_acc;
    }
}
static void ph_copy_out(pool* p, int batch, int w)
{
    (void)w; ctx* c = p->c;
    const int b = batch * p->batch_size, e = b + p->batch_size < c->nj ? b + p->batch_size : c->nj;
    for (int s = b; s < e; ++s) {                                                        /* ref: :527-547 */
        phxo_contact_joint* j = &c->joints[c->joint_index[s]];
        const pack8* P = &c->packs[(unsigned)s / N]; const int k = s & (N - 1);
        j->normal_acc = P->n_acc[k]; j->friction_acc = P->f_acc[k];
    }
}
static void ph_refresh(pool* p, int batch, int w) { (void)w; int vb, ve, tb, te; batch_range(p, batch, &vb, &ve, &tb, &te); if (ve > vb) refresh_joints(p->c, vb, ve, N); if (te > tb) refresh_joints(p->c, tb, te, 1); }
static void ph_prestep(pool* p, int batch, int w) { (void)w; int vb, ve, tb, te; batch_range(p, batch, &vb, &ve, &tb, &te); if (ve > vb) prestep_joints8(p->c, vb, ve); if (te > tb) prestep_joints(p->c, tb, te, 1); }
static void ph_impulse(pool* p, int batch, int w)
{
    (void)w; int vb, ve, tb, te; batch_range(p, batch, &vb, &ve, &tb, &te);
    int any = 0;
    if (ve > vb) any |= solve_impulses8(p->c, vb, ve, p->iter);
    if (te > tb) any |= solve_impulses(p->c, tb, te, 1, p->iter);
    if (any) atomic_store_explicit(&p->productive, 1, memory_order_relaxed);
}
static void ph_displacement(pool* p, int batch, int w)
{
    (void)w; int vb, ve, tb, te; batch_range(p, batch, &vb, &ve, &tb, &te);
    int any = 0;
    if (ve > vb) any |= solve_displacement8(p->c, vb, ve, p->iter);
    if (te > tb) any |= solve_displacement(p->c, tb, te, 1, p->iter);
    if (any) atomic_store_explicit(&p->productive, 1, memory_order_relaxed);
}

/* Solver::SolveJoints<8> in place (ref: Solver.cpp:68-119), island mode Single (sloppy = 0: one batch, one thread does
 * the solve) or Single Sloppy (sloppy = 1: 512-joint batches over `threads` persistent threads, racy like the reference).
 * Returns 0, fills the per-phase seconds. */
int phxb_solve(phxo_body* bodies, int nb, const phxo_contact_point* cps, phxo_contact_joint* joints, int nj,
               int contact_iters, int pen_iters, int threads, int sloppy, phxb_phases* out)
{
    phxb_phases ph; memset(&ph, 0, sizeof ph);
    if (threads < 1) threads = 1;
    FAST_MODE_ENTER();
    ctx c; memset(&c, 0, sizeof c);
    c.nb = nb; c.nj = nj; c.cps = cps; c.joints = joints;
    c.imp = (sbody*)aligned_alloc(64, ((size_t)(nb + 1) * sizeof(sbody) + 63) & ~(size_t)63);
    c.disp = (sbody*)aligned_alloc(64, ((size_t)(nb + 1) * sizeof(sbody) + 63) & ~(size_t)63);
    c.par = (sparam*)malloc((size_t)(nb + 1) * sizeof(sparam));
    c.packs = (pack8*)aligned_alloc(64, ((size_t)(nj / N + 2) * sizeof(pack8) + 63) & ~(size_t)63);
    memset(&c.packs[nj / N], 0, 2 * sizeof(pack8));                                      /* (the rest: ph_zero, by the workers that sweep it) */
    c.joint_index = (int32_t*)malloc((size_t)(nj + N) * sizeof(int32_t));
    int32_t* group_bodies = (int32_t*)malloc((size_t)(nb + 1) * sizeof(int32_t));
    int32_t* work = (int32_t*)malloc((size_t)(nj + N) * sizeof(int32_t));

    pool p; memset(&p, 0, sizeof p);
    p.threads = threads; p.c = &c;
    p.batch_size = sloppy ? 512 : (nj > 0 ? nj : 1);                                      /* ref: :138-139 */
    const int batches = nj ? (nj + p.batch_size - 1) / p.batch_size : 0;
    warg* args = (warg*)malloc((size_t)threads * sizeof(warg));
    p.th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
    p.cur = (cursor*)aligned_alloc(64, (size_t)threads * sizeof(cursor));
    memset(p.cur, 0, (size_t)threads * sizeof(cursor));
    int cpus[CPU_SETSIZE];
    cpu_set_t caller_set; CPU_ZERO(&caller_set);
    const int have_set = sched_getaffinity(0, sizeof caller_set, &caller_set) == 0;
    p.cpus = cpus; p.ncpus = threads > 1 ? pool_cpus(cpus, CPU_SETSIZE) : 0;             /* (one thread: wherever the scheduler puts it, like the reference's main thread) */
    for (int t = 1; t < threads; ++t) { args[t].p = &p; args[t].worker = t; pthread_create(&p.th[t], NULL, pool_thread, &args[t]); }
    pool_pin(&p, 0);
    int sense = 0;
    pool_run(&p, &sense, ph_zero, batches);                                              /* (outside the timed phases: the reference's arrays persist from step to step) */

    const double t_begin = now_s();
    double t0 = t_begin, t1;
    for (int i = 0; i < nb; ++i) {                                                       /* PrepareBodies, ref: :456-480 */
        const phxo_body* b = &bodies[i];
        c.par[i].im = b->inv_mass; c.par[i].ii = b->inv_inertia; c.par[i].px = b->pos.x; c.par[i].py = b->pos.y;
        c.par[i].xvx = b->xv.x; c.par[i].xvy = b->xv.y; c.par[i].yvx = b->yv.x; c.par[i].yvy = b->yv.y;
        c.imp[i].vx = b->velocity.x; c.imp[i].vy = b->velocity.y; c.imp[i].w = b->angular_velocity; c.imp[i].tag = -1;
        c.disp[i].vx = b->displacing_velocity.x; c.disp[i].vy = b->displacing_velocity.y; c.disp[i].w = b->displacing_angular_velocity; c.disp[i].tag = -1;
    }
    t1 = now_s(); ph.prepare_bodies = t1 - t0; t0 = t1;
    for (int i = 0; i < nj; ++i) c.joint_index[i] = i;                                   /* ref: :102-106 */
    for (int i = 0; i < nb; ++i) group_bodies[i] = 0;
    c.group_offset = prepare_indices8(joints, c.joint_index, group_bodies, work, nj);    /* serial in the reference too */
    t1 = now_s(); ph.prepare_indices = t1 - t0; t0 = t1;
    pool_run(&p, &sense, ph_copy_in, batches);
    t1 = now_s(); ph.prepare_joints = t1 - t0; t0 = t1;
    pool_run(&p, &sense, ph_refresh, batches);
    t1 = now_s(); ph.refresh = t1 - t0; t0 = t1;
    pool_run(&p, &sense, ph_prestep, batches);
    t1 = now_s(); ph.prestep = t1 - t0; t0 = t1;
    int it;
    for (it = 0; it < contact_iters; ++it) {                                             /* ref: :171-190 */
        p.iter = it; atomic_store(&p.productive, 0);
        pool_run(&p, &sense, ph_impulse, batches);
        ph.joint_visits += nj;
        if (!atomic_load(&p.productive)) { ++it; break; }
    }
    ph.impulse_iterations = it;
    t1 = now_s(); ph.impulse = t1 - t0; t0 = t1;
    for (it = 0; it < pen_iters; ++it) {                                                 /* ref: :193-211 */
        p.iter = it; atomic_store(&p.productive, 0);
        pool_run(&p, &sense, ph_displacement, batches);
        if (!atomic_load(&p.productive)) { ++it; break; }
    }
    ph.displacement_iterations = it;
    t1 = now_s(); ph.displacement = t1 - t0; t0 = t1;
    pool_run(&p, &sense, ph_copy_out, batches);
    for (int i = 0; i < nb; ++i) {                                                       /* FinishBodies, ref: :482-494 */
        bodies[i].velocity.x = c.imp[i].vx; bodies[i].velocity.y = c.imp[i].vy; bodies[i].angular_velocity = c.imp[i].w;
        bodies[i].displacing_velocity.x = c.disp[i].vx; bodies[i].displacing_velocity.y = c.disp[i].vy;
        bodies[i].displacing_angular_velocity = c.disp[i].w;
    }
    t1 = now_s(); ph.finish = t1 - t0;
    ph.total = t1 - t_begin;
    ph.group_offset = c.group_offset; ph.threads = threads;

    if (threads > 1) pool_stop(&p);
    if (p.ncpus && have_set) (void)sched_setaffinity(0, sizeof caller_set, &caller_set);   /* the caller's thread gets its CPUs back */
    free(args); free(p.th); free(p.cur); free(group_bodies); free(work);
    free(c.imp); free(c.disp); free(c.par); free(c.packs); free(c.joint_index);
    FAST_MODE_LEAVE();
    if (out) *out = ph;
    return 0;
}

/* ---- test hooks (tests/test_oracle_pins.py pins them against the reference's headers) ---------------------------------- */
/* 32-bit word offset of a field of the pack8 block = ContactJointPacked<8> (ref: Solver.h:26-45); same field numbering as
 * oracle/ref_harness/leaf_harness.cpp::ref_packed_offset */
#include <stddef.h>
int phxb_pack8_offset(int field)
{
    switch (field) {
    case 0: return (int)offsetof(pack8, b1) / 4;
    case 1: return (int)offsetof(pack8, b2) / 4;
    case 2: return (int)offsetof(pack8, cp) / 4;
    case 3: return (int)(offsetof(pack8, n) + offsetof(lim8, p1x)) / 4;
    case 4: return (int)(offsetof(pack8, n) + offsetof(lim8, cim)) / 4;
    case 5: return (int)offsetof(pack8, n_acc) / 4;
    case 6: return (int)offsetof(pack8, n_dst) / 4;
    case 7: return (int)offsetof(pack8, n_dstd) / 4;
    case 8: return (int)offsetof(pack8, n_accd) / 4;
    case 9: return (int)(offsetof(pack8, f) + offsetof(lim8, p1x)) / 4;
    case 10: return (int)(offsetof(pack8, f) + offsetof(lim8, cim)) / 4;
    case 11: return (int)offsetof(pack8, f_acc) / 4;
    case 12: return (int)sizeof(pack8) / 4;
    }
    return -1;
}
/* the 8-lane gather / scatter of 16-byte SolveBody records (counterparts of ref: base/SIMD_AVX2.h:324-377) */
void phxb_test_gather8(const void* base, const int32_t* idx, float* out32)
{
    __m256 x, y, z, t;
    gather8((const sbody*)base, idx, &x, &y, &z, &t);
    _mm256_storeu_ps(out32, x); _mm256_storeu_ps(out32 + 8, y); _mm256_storeu_ps(out32 + 16, z); _mm256_storeu_ps(out32 + 24, t);
}
void phxb_test_scatter8(const float* in32, void* base, const int32_t* idx)
{
    scatter8((sbody*)base, idx, _mm256_loadu_ps(in32), _mm256_loadu_ps(in32 + 8), _mm256_loadu_ps(in32 + 16), _mm256_loadu_ps(in32 + 24));
}

/* ---- broadphase -------------------------------------------------------------------------------------------------- */
typedef struct { uint32_t value, index; } sort_entry;                                     /* ref: Collider.h:52-56 */
typedef struct { float minx, maxx, centery, extenty; uint32_t index; } bp_entry;          /* ref: Collider.h:45-50 */

static inline uint32_t radix_float(float f)                                               /* ref: base/RadixSort.h:19-26 */
{
    uint32_t u; memcpy(&u, &f, 4);
    return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}

/* ref: base/RadixSort.h:28-95 radixSort3: one histogram pass for the three digits (11/11/10 bits), three scatters */
static sort_entry* radix_sort3(sort_entry* e0, sort_entry* e1, size_t count)
{
    static _Thread_local uint32_t h[3][2048];
    memset(h, 0, sizeof h);
    for (size_t i = 0; i < count; ++i) {
        const uint32_t v = e0[i].value;
        h[0][v & 2047]++; h[1][(v >> 11) & 2047]++; h[2][v >> 22]++;
    }
    uint32_t s0 = 0, s1 = 0, s2 = 0;
    for (int i = 0; i < 2048; ++i) {
        const uint32_t a = h[0][i], b = h[1][i], c = h[2][i];
        h[0][i] = s0; h[1][i] = s1; h[2][i] = s2; s0 += a; s1 += b; s2 += c;
    }
    for (size_t i = 0; i < count; ++i) e1[h[0][e0[i].value & 2047]++] = e0[i];
    for (size_t i = 0; i < count; ++i) e0[h[1][(e1[i].value >> 11) & 2047]++] = e1[i];
    for (size_t i = 0; i < count; ++i) e1[h[2][e0[i].value >> 22]++] = e0[i];
    return e1;
}

static inline uint64_t pair_hash(uint32_t lb, uint32_t rb) { return lb ^ (rb + 0x9e3779b9u + (lb << 6) + (lb >> 2)); }   /* ref: Collider.h:7-20 */

typedef struct { uint64_t* keys; size_t mask; } pairset;                                   /* set semantics of DenseHashSet */
static int ps_find(const pairset* s, uint32_t a, uint32_t b)
{
    const uint64_t key = ((uint64_t)a << 32) | b;
    size_t h = (size_t)pair_hash(a, b) & s->mask;
    for (size_t probe = 0;; ++probe) {                                                     /* triangular probing, ref: base/DenseHash.h:111-150 */
        if (s->keys[h] == key) return 1;
        if (s->keys[h] == ~0ull) return 0;
        h = (h + probe + 1) & s->mask;
    }
}
static void ps_insert(pairset* s, uint32_t a, uint32_t b)
{
    const uint64_t key = ((uint64_t)a << 32) | b;
    size_t h = (size_t)pair_hash(a, b) & s->mask;
    for (size_t probe = 0;; ++probe) {
        if (s->keys[h] == key) return;
        if (s->keys[h] == ~0ull) { s->keys[h] = key; return; }
        h = (h + probe + 1) & s->mask;
    }
}

typedef struct {
    const bp_entry* e; size_t n; const pairset* set;
    atomic_size_t next;
    int64_t tests[256], overlaps[256], fresh[256];
} sweep_job;
typedef struct { sweep_job* j; int worker; } sweep_arg;

/* ref: Collider.cpp:296-318 (serial) / :320-345 (parallel: blocks of 128 rows pulled by the workers) */
static void* sweep_worker(void* a)
{
    sweep_job* j = ((sweep_arg*)a)->j; const int w = ((sweep_arg*)a)->worker;
    int64_t tests = 0, overlaps = 0, fresh = 0;
    for (;;) {
        const size_t b = atomic_fetch_add(&j->next, 128);
        if (b >= j->n) break;
        const size_t e = b + 128 < j->n ? b + 128 : j->n;
        for (size_t i = b; i < e; ++i) {
            const bp_entry* A = &j->e[i];
            for (size_t k = i + 1; k < j->n; ++k) {
                const bp_entry* B = &j->e[k];
                if (B->minx > A->maxx) break;
                ++tests;
                if (fabsf(B->centery - A->centery) <= A->extenty + B->extenty) {
                    ++overlaps;
                    if (!ps_find(j->set, A->index, B->index)) ++fresh;
                }
            }
        }
    }
    j->tests[w] = tests; j->overlaps[w] = overlaps; j->fresh[w] = fresh;
    return NULL;
}

/* UpdateBroadphase + UpdatePairs timed `reps` times on `threads` threads (UpdateBroadphase is serial in the reference even
 * with workers).  The pair set is filled by an untimed first pass, so the timed passes are the steady state (lookups only). */
int phxb_broadphase(const phxo_body* bodies, int nb, int threads, int reps, phxb_broadphase_phases* out)
{
    phxb_broadphase_phases ph; memset(&ph, 0, sizeof ph);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    FAST_MODE_ENTER();
    sort_entry* s0 = (sort_entry*)malloc((size_t)(nb + 1) * sizeof(sort_entry));
    sort_entry* s1 = (sort_entry*)malloc((size_t)(nb + 1) * sizeof(sort_entry));
    bp_entry* ent = (bp_entry*)malloc((size_t)(nb + 1) * sizeof(bp_entry));
    pairset set; size_t cap = 1024;
    while (cap < (size_t)nb * 8) cap <<= 1;
    set.mask = cap - 1; set.keys = (uint64_t*)malloc(cap * sizeof(uint64_t));
    memset(set.keys, 0xFF, cap * sizeof(uint64_t));
    pthread_t* th = (pthread_t*)malloc((size_t)threads * sizeof(pthread_t));
    sweep_arg* args = (sweep_arg*)malloc((size_t)threads * sizeof(sweep_arg));
    sweep_job* job = (sweep_job*)calloc(1, sizeof(sweep_job));

    for (int rep = -1; rep < reps; ++rep) {
        const double t0 = now_s();
        for (int i = 0; i < nb; ++i) { s0[i].value = radix_float(bodies[i].aabb_min.x); s0[i].index = (uint32_t)i; }   /* ref: Collider.cpp:259-265 */
        const sort_entry* sorted = radix_sort3(s0, s1, (size_t)nb);
        for (int i = 0; i < nb; ++i) {                                                                              /* ref: :269-283 */
            const phxo_body* b = &bodies[sorted[i].index];
            ent[i].minx = b->aabb_min.x; ent[i].maxx = b->aabb_max.x;
            ent[i].centery = (b->aabb_min.y + b->aabb_max.y) * 0.5f; ent[i].extenty = (b->aabb_max.y - b->aabb_min.y) * 0.5f;
            ent[i].index = sorted[i].index;
        }
        const double t1 = now_s();
        if (rep < 0) {                                                                   /* fill the persistent set (untimed) */
            for (int i = 0; i < nb; ++i)
                for (int k = i + 1; k < nb && ent[k].minx <= ent[i].maxx; ++k)
                    if (fabsf(ent[k].centery - ent[i].centery) <= ent[i].extenty + ent[k].extenty) ps_insert(&set, ent[i].index, ent[k].index);
            continue;
        }
        job->e = ent; job->n = (size_t)nb; job->set = &set; atomic_store(&job->next, 0);
        for (int t = 1; t < threads; ++t) { args[t].j = job; args[t].worker = t; pthread_create(&th[t], NULL, sweep_worker, &args[t]); }
        args[0].j = job; args[0].worker = 0; sweep_worker(&args[0]);
        for (int t = 1; t < threads; ++t) pthread_join(th[t], NULL);
        const double t2 = now_s();
        ph.update_broadphase += t1 - t0; ph.update_pairs += t2 - t1;
        ph.candidate_tests = 0; ph.overlapping_pairs = 0;
        for (int t = 0; t < threads; ++t) { ph.candidate_tests += job->tests[t]; ph.overlapping_pairs += job->overlaps[t]; }
    }
    ph.threads = threads; ph.reps = reps;
    free(s0); free(s1); free(ent); free(set.keys); free(th); free(args); free(job);
    FAST_MODE_LEAVE();
    if (out) *out = ph;
    return 0;
}
