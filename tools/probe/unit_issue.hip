// unit_issue.hip — what bounds one class step of the island kernel on MI355X: issue rate or dependent latency?
// One wave per workgroup.  Variants: (a) pure dependent mul/add chains, 1 / 2 / 4 of them interleaved in one wave;
// (b) the island kernel's own unit update (isl_impulse on leader + follower, bodies in LDS), P units one after the other
// ("plain", the shipped branchy form) against P units interleaved statement by statement and committed by selects ("ilp").
// Grids of 256 / 1024 / 2048 / 4096 one-wave workgroups = ~1 wave per CU, per SIMD, 2 and 4 per SIMD.
#include "../../phyx_amd/csrc/island_kernel.h"
#include <cstdio>
#include <vector>
using namespace phx;
// one wave per workgroup: program order orders its LDS operations; the compiler only has to be told that other lanes' stores count
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

template <int C>
__global__ void __launch_bounds__(64) k_chain(float* out, unsigned long long* cyc, int steps, float a, float b)
{
    float x[C];
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = threadIdx.x * 0.001f + c;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = x[c] * a;
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = x[c] + b;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0; for (int c = 0; c < C; ++c) r += x[c];
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// P units interleaved; one impulse visit of joint q[p] of every unit, statement by statement
template <int P>
__device__ __forceinline__ void ilp_impulse(IslJoint (&q)[P], float4 (&B1)[P], float4 (&B2)[P], const float (&im1)[P], const float (&ii1)[P],
                                            const float (&im2)[P], const float (&ii2)[P], const bool (&on)[P], int it, bool (&prod)[P])
{
    bool ev[P]; float dv[P], dn[P], fv[P], df[P], x1[P], y1[P], z1[P], x2[P], y2[P], z2[P], an[P], af[P];
#pragma unroll
    for (int p = 0; p < P; ++p) ev[p] = on[p] && (__float_as_int(B1[p].w) > it - 2 || __float_as_int(B2[p].w) > it - 2);
#pragma unroll
    for (int p = 0; p < P; ++p) { x1[p] = B1[p].x; y1[p] = B1[p].y; z1[p] = B1[p].z; x2[p] = B2[p].x; y2[p] = B2[p].y; z2[p] = B2[p].z; dv[p] = q[p].dstV; }
#pragma unroll
    for (int p = 0; p < P; ++p) dv[p] -= q[p].nx * x1[p];
#pragma unroll
    for (int p = 0; p < P; ++p) dv[p] -= q[p].ny * y1[p];
#pragma unroll
    for (int p = 0; p < P; ++p) dv[p] -= q[p].aN1 * z1[p];
#pragma unroll
    for (int p = 0; p < P; ++p) dv[p] -= (-q[p].nx) * x2[p];
#pragma unroll
    for (int p = 0; p < P; ++p) dv[p] -= (-q[p].ny) * y2[p];
#pragma unroll
    for (int p = 0; p < P; ++p) dv[p] -= q[p].aN2 * z2[p];
#pragma unroll
    for (int p = 0; p < P; ++p) { dn[p] = dv[p] * q[p].cimN; dn[p] = max_ref(dn[p], -q[p].accN); }
#pragma unroll
    for (int p = 0; p < P; ++p) { x1[p] += (q[p].nx * im1[p]) * dn[p]; y1[p] += (q[p].ny * im1[p]) * dn[p]; z1[p] += (q[p].aN1 * ii1[p]) * dn[p]; }
#pragma unroll
    for (int p = 0; p < P; ++p) { x2[p] += ((-q[p].nx) * im2[p]) * dn[p]; y2[p] += ((-q[p].ny) * im2[p]) * dn[p]; z2[p] += (q[p].aN2 * ii2[p]) * dn[p]; }
#pragma unroll
    for (int p = 0; p < P; ++p) { an[p] = q[p].accN + dn[p]; fv[p] = 0.f; }
#pragma unroll
    for (int p = 0; p < P; ++p) fv[p] -= (-q[p].ny) * x1[p];
#pragma unroll
    for (int p = 0; p < P; ++p) fv[p] -= q[p].nx * y1[p];
#pragma unroll
    for (int p = 0; p < P; ++p) fv[p] -= q[p].aF1 * z1[p];
#pragma unroll
    for (int p = 0; p < P; ++p) fv[p] -= q[p].ny * x2[p];          // -(-ny): tx = -ny, -tx = ny
#pragma unroll
    for (int p = 0; p < P; ++p) fv[p] -= (-q[p].nx) * y2[p];
#pragma unroll
    for (int p = 0; p < P; ++p) fv[p] -= q[p].aF2 * z2[p];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        df[p] = fv[p] * q[p].cimF;
        const float force = q[p].accF + df[p];
        const float limit = an[p] * 0.3f;
        const float signed_limit = force < 0.f ? -limit : limit;
        const float adjusted = signed_limit - q[p].accF;
        df[p] = fabsf(force) > limit ? adjusted : df[p];
        af[p] = q[p].accF + df[p];
    }
#pragma unroll
    for (int p = 0; p < P; ++p) { x1[p] += ((-q[p].ny) * im1[p]) * df[p]; y1[p] += (q[p].nx * im1[p]) * df[p]; z1[p] += (q[p].aF1 * ii1[p]) * df[p]; }
#pragma unroll
    for (int p = 0; p < P; ++p) { x2[p] += (q[p].ny * im2[p]) * df[p]; y2[p] += ((-q[p].nx) * im2[p]) * df[p]; z2[p] += (q[p].aF2 * ii2[p]) * df[p]; }
#pragma unroll
    for (int p = 0; p < P; ++p) {
        prod[p] = ev[p] && max_ref(fabsf(dn[p]), fabsf(df[p])) > 1e-4f;
        q[p].accN = ev[p] ? an[p] : q[p].accN; q[p].accF = ev[p] ? af[p] : q[p].accF;
        B1[p].x = ev[p] ? x1[p] : B1[p].x; B1[p].y = ev[p] ? y1[p] : B1[p].y; B1[p].z = ev[p] ? z1[p] : B1[p].z;
        B2[p].x = ev[p] ? x2[p] : B2[p].x; B2[p].y = ev[p] ? y2[p] : B2[p].y; B2[p].z = ev[p] ? z2[p] : B2[p].z;
        B1[p].w = prod[p] ? __int_as_float(it) : B1[p].w; B2[p].w = prod[p] ? __int_as_float(it) : B2[p].w;
    }
}

template <int P, int MODE>       // MODE 0: plain (P units one after the other, shipped code), 1: ilp (interleaved, selects)
__global__ void __launch_bounds__(64) k_units(const float* __restrict__ seed, float4* out, unsigned long long* cyc, int steps, int it0)
{
    __shared__ float4 imp[64 * 2 * P + 64];
    const int tid = threadIdx.x;
    IslJoint q0[P], q1[P]; float im1[P], ii1[P], im2[P], ii2[P]; int l1[P], l2[P]; bool on[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const float s = seed[(tid * P + p) & 1023];
        q0[p] = IslJoint{0.6f + 0.01f * s, 0.8f - 0.01f * s, 1.5f + s, -1.25f + s, 0.5f * s, 0.25f - s, 0.001f + 0.0001f * s, 0.002f, -0.1f, 0.f, 0.5f + s, 0.01f * s, 0.f};
        q1[p] = IslJoint{0.6f + 0.01f * s, 0.8f - 0.01f * s, -1.5f + s, 1.25f + s, 0.7f * s, 0.35f - s, 0.001f + 0.0002f * s, 0.002f, -0.1f, 0.f, 0.4f + s, 0.02f * s, 0.f};
        im1[p] = 4000.f; ii1[p] = 80.f; im2[p] = 4000.f; ii2[p] = 80.f;
        l1[p] = tid * 2 * P + 2 * p; l2[p] = l1[p] + 1; on[p] = true;
        imp[l1[p]] = make_float4(s, -s, 0.1f * s, __int_as_float(-1)); imp[l2[p]] = make_float4(-0.5f * s, 0.25f * s, -0.1f * s, __int_as_float(-1));
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        const int it = it0;        // (0: every joint is evaluated, like the first sweeps of a solve)
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                float4 B1 = imp[l1[p]], B2 = imp[l2[p]];
                bool pr0 = false, pr1 = false;
                bool t = isl_impulse(q0[p], B1, B2, im1[p], ii1[p], im2[p], ii2[p], false, false, false, false, it, pr0);
                t |= isl_impulse(q1[p], B1, B2, im1[p], ii1[p], im2[p], ii2[p], false, false, false, false, it, pr1);
                if (t) { imp[l1[p]] = B1; imp[l2[p]] = B2; }
                WAVE_SYNC();
            }
        } else {
            float4 B1[P], B2[P]; bool pr0[P], pr1[P];
#pragma unroll
            for (int p = 0; p < P; ++p) { B1[p] = imp[l1[p]]; B2[p] = imp[l2[p]]; }
            ilp_impulse<P>(q0, B1, B2, im1, ii1, im2, ii2, on, it, pr0);
            ilp_impulse<P>(q1, B1, B2, im1, ii1, im2, ii2, on, it, pr1);
#pragma unroll
            for (int p = 0; p < P; ++p) { imp[l1[p]] = B1[p]; imp[l2[p]] = B2[p]; }
            WAVE_SYNC();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float4 r = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < P; ++p) { const float4 a = imp[l1[p]]; r.x += a.x + q0[p].accN + q1[p].accF; r.y += a.y; r.z += a.z; }
    out[blockIdx.x * 64 + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}


typedef float v2f __attribute__((ext_vector_type(2)));

template <int C>
__global__ void __launch_bounds__(64) k_chain_pk(float* out, unsigned long long* cyc, int steps, float a, float b)
{
    v2f x[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { x[c].x = threadIdx.x * 0.001f + c; x[c].y = threadIdx.x * 0.002f + c; }
    const v2f va = {a, a}, vb = {b, b * 2.f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = x[c] * va;
#pragma unroll
            for (int c = 0; c < C; ++c) x[c] = x[c] + vb;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0; for (int c = 0; c < C; ++c) r += x[c].x + x[c].y;
    out[blockIdx.x * 64 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// the unit update on PAIRS of units: every arithmetic statement is one packed operation on {unit A, unit B}; compares and selects per half
struct PkJoint { v2f nx, ny, aN1, aN2, aF1, aF2, cimN, cimF, dstV, dstD, accN, accF, accD; };
__device__ __forceinline__ v2f sel2(bool a, bool b, v2f t, v2f f) { v2f r; r.x = a ? t.x : f.x; r.y = b ? t.y : f.y; return r; }

template <int PP>
__device__ __forceinline__ void pk_impulse(PkJoint (&q)[PP], v2f (&X1)[PP], v2f (&Y1)[PP], v2f (&Z1)[PP], int (&T1)[PP][2], v2f (&X2)[PP], v2f (&Y2)[PP], v2f (&Z2)[PP], int (&T2)[PP][2],
                                           const v2f (&im1)[PP], const v2f (&ii1)[PP], const v2f (&im2)[PP], const v2f (&ii2)[PP], int it)
{
    bool ea[PP], eb[PP]; v2f dv[PP], dn[PP], fv[PP], df[PP], x1[PP], y1[PP], z1[PP], x2[PP], y2[PP], z2[PP], an[PP], af[PP];
#pragma unroll
    for (int p = 0; p < PP; ++p) { ea[p] = T1[p][0] > it - 2 || T2[p][0] > it - 2; eb[p] = T1[p][1] > it - 2 || T2[p][1] > it - 2; }
#pragma unroll
    for (int p = 0; p < PP; ++p) { x1[p] = X1[p]; y1[p] = Y1[p]; z1[p] = Z1[p]; x2[p] = X2[p]; y2[p] = Y2[p]; z2[p] = Z2[p]; dv[p] = q[p].dstV; }
#pragma unroll
    for (int p = 0; p < PP; ++p) dv[p] -= q[p].nx * x1[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) dv[p] -= q[p].ny * y1[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) dv[p] -= q[p].aN1 * z1[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) dv[p] -= (-q[p].nx) * x2[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) dv[p] -= (-q[p].ny) * y2[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) dv[p] -= q[p].aN2 * z2[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) { dn[p] = dv[p] * q[p].cimN; const v2f m = -q[p].accN; dn[p] = sel2(dn[p].x > m.x, dn[p].y > m.y, dn[p], m); }
#pragma unroll
    for (int p = 0; p < PP; ++p) { x1[p] += (q[p].nx * im1[p]) * dn[p]; y1[p] += (q[p].ny * im1[p]) * dn[p]; z1[p] += (q[p].aN1 * ii1[p]) * dn[p]; }
#pragma unroll
    for (int p = 0; p < PP; ++p) { x2[p] += ((-q[p].nx) * im2[p]) * dn[p]; y2[p] += ((-q[p].ny) * im2[p]) * dn[p]; z2[p] += (q[p].aN2 * ii2[p]) * dn[p]; }
#pragma unroll
    for (int p = 0; p < PP; ++p) { an[p] = q[p].accN + dn[p]; fv[p] = (v2f){0.f, 0.f}; }
#pragma unroll
    for (int p = 0; p < PP; ++p) fv[p] -= (-q[p].ny) * x1[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) fv[p] -= q[p].nx * y1[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) fv[p] -= q[p].aF1 * z1[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) fv[p] -= q[p].ny * x2[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) fv[p] -= (-q[p].nx) * y2[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) fv[p] -= q[p].aF2 * z2[p];
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        df[p] = fv[p] * q[p].cimF;
        const v2f force = q[p].accF + df[p];
        const v2f limit = an[p] * (v2f){0.3f, 0.3f};
        const v2f signed_limit = sel2(force.x < 0.f, force.y < 0.f, -limit, limit);
        const v2f adjusted = signed_limit - q[p].accF;
        df[p] = sel2(fabsf(force.x) > limit.x, fabsf(force.y) > limit.y, adjusted, df[p]);
        af[p] = q[p].accF + df[p];
    }
#pragma unroll
    for (int p = 0; p < PP; ++p) { x1[p] += ((-q[p].ny) * im1[p]) * df[p]; y1[p] += (q[p].nx * im1[p]) * df[p]; z1[p] += (q[p].aF1 * ii1[p]) * df[p]; }
#pragma unroll
    for (int p = 0; p < PP; ++p) { x2[p] += (q[p].ny * im2[p]) * df[p]; y2[p] += ((-q[p].nx) * im2[p]) * df[p]; z2[p] += (q[p].aF2 * ii2[p]) * df[p]; }
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        const bool pa = ea[p] && max_ref(fabsf(dn[p].x), fabsf(df[p].x)) > 1e-4f, pb = eb[p] && max_ref(fabsf(dn[p].y), fabsf(df[p].y)) > 1e-4f;
        q[p].accN = sel2(ea[p], eb[p], an[p], q[p].accN); q[p].accF = sel2(ea[p], eb[p], af[p], q[p].accF);
        X1[p] = sel2(ea[p], eb[p], x1[p], X1[p]); Y1[p] = sel2(ea[p], eb[p], y1[p], Y1[p]); Z1[p] = sel2(ea[p], eb[p], z1[p], Z1[p]);
        X2[p] = sel2(ea[p], eb[p], x2[p], X2[p]); Y2[p] = sel2(ea[p], eb[p], y2[p], Y2[p]); Z2[p] = sel2(ea[p], eb[p], z2[p], Z2[p]);
        T1[p][0] = pa ? it : T1[p][0]; T2[p][0] = pa ? it : T2[p][0]; T1[p][1] = pb ? it : T1[p][1]; T2[p][1] = pb ? it : T2[p][1];
    }
}

template <int PP>
__global__ void __launch_bounds__(64) k_units_pk(const float* __restrict__ seed, float4* out, unsigned long long* cyc, int steps, int it0)
{
    __shared__ float4 imp[64 * 4 * PP + 64];
    const int tid = threadIdx.x;
    PkJoint q0[PP], q1[PP]; v2f im1[PP], ii1[PP], im2[PP], ii2[PP]; int l1[PP][2], l2[PP][2];
#pragma unroll
    for (int p = 0; p < PP; ++p) {
        const float sa = seed[(tid * 2 * PP + 2 * p) & 1023], sb = seed[(tid * 2 * PP + 2 * p + 1) & 1023];
        const v2f s = {sa, sb}, one = {1.f, 1.f};
        q0[p] = PkJoint{0.6f * one + 0.01f * s, 0.8f * one - 0.01f * s, 1.5f * one + s, -1.25f * one + s, 0.5f * s, 0.25f * one - s, 0.001f * one + 0.0001f * s, 0.002f * one, -0.1f * one, 0.f * one, 0.5f * one + s, 0.01f * s, 0.f * one};
        q1[p] = PkJoint{0.6f * one + 0.01f * s, 0.8f * one - 0.01f * s, -1.5f * one + s, 1.25f * one + s, 0.7f * s, 0.35f * one - s, 0.001f * one + 0.0002f * s, 0.002f * one, -0.1f * one, 0.f * one, 0.4f * one + s, 0.02f * s, 0.f * one};
        im1[p] = 4000.f * one; ii1[p] = 80.f * one; im2[p] = 4000.f * one; ii2[p] = 80.f * one;
        for (int h = 0; h < 2; ++h) {
            l1[p][h] = tid * 4 * PP + 4 * p + 2 * h; l2[p][h] = l1[p][h] + 1;
            const float ss = h ? sb : sa;
            imp[l1[p][h]] = make_float4(ss, -ss, 0.1f * ss, __int_as_float(-1)); imp[l2[p][h]] = make_float4(-0.5f * ss, 0.25f * ss, -0.1f * ss, __int_as_float(-1));
        }
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        const int it = it0;
        v2f X1[PP], Y1[PP], Z1[PP], X2[PP], Y2[PP], Z2[PP]; int T1[PP][2], T2[PP][2];
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            const float4 a1 = imp[l1[p][0]], a2 = imp[l2[p][0]], b1 = imp[l1[p][1]], b2 = imp[l2[p][1]];
            X1[p] = (v2f){a1.x, b1.x}; Y1[p] = (v2f){a1.y, b1.y}; Z1[p] = (v2f){a1.z, b1.z}; T1[p][0] = __float_as_int(a1.w); T1[p][1] = __float_as_int(b1.w);
            X2[p] = (v2f){a2.x, b2.x}; Y2[p] = (v2f){a2.y, b2.y}; Z2[p] = (v2f){a2.z, b2.z}; T2[p][0] = __float_as_int(a2.w); T2[p][1] = __float_as_int(b2.w);
        }
        pk_impulse<PP>(q0, X1, Y1, Z1, T1, X2, Y2, Z2, T2, im1, ii1, im2, ii2, it);
        pk_impulse<PP>(q1, X1, Y1, Z1, T1, X2, Y2, Z2, T2, im1, ii1, im2, ii2, it);
#pragma unroll
        for (int p = 0; p < PP; ++p) {
            imp[l1[p][0]] = make_float4(X1[p].x, Y1[p].x, Z1[p].x, __int_as_float(T1[p][0])); imp[l2[p][0]] = make_float4(X2[p].x, Y2[p].x, Z2[p].x, __int_as_float(T2[p][0]));
            imp[l1[p][1]] = make_float4(X1[p].y, Y1[p].y, Z1[p].y, __int_as_float(T1[p][1])); imp[l2[p][1]] = make_float4(X2[p].y, Y2[p].y, Z2[p].y, __int_as_float(T2[p][1]));
        }
        WAVE_SYNC();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float4 r = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < PP; ++p) { const float4 a = imp[l1[p][0]], b = imp[l1[p][1]]; r.x += a.x + b.x + q0[p].accN.x + q1[p].accF.y; r.y += a.y + b.y; r.z += a.z + b.z; }
    out[blockIdx.x * 64 + tid] = r;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
static void report(const char* name, int blocks, int steps, double per, F launch)
{
    float4* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 64 * sizeof(float4));
    hipMalloc(&cyc, blocks * sizeof(unsigned long long));
    for (int rep = 0; rep < 2; ++rep) launch(out, cyc);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    double sum = 0; unsigned long long mx = 0;
    for (auto v : h) { sum += (double)v; mx = v > mx ? v : mx; }
    printf("%-44s waves %5d: %8.1f cycles per %s (mean), %8.1f (slowest wave)\n", name, blocks, sum / blocks / steps / per, per == 1.0 ? "step" : "op", (double)mx / steps / per);
    hipFree(out); hipFree(cyc);
}

int main()
{
    float* seed; hipMalloc(&seed, 1024 * sizeof(float));
    std::vector<float> hs(1024); for (int i = 0; i < 1024; ++i) hs[i] = 0.001f * (float)((i * 37) % 101);
    hipMemcpy(seed, hs.data(), 1024 * sizeof(float), hipMemcpyHostToDevice);
    const int steps = 400;
    for (int blocks : {256, 1024, 2048, 4096}) {
        report("dependent mul+add chain x1", blocks, steps, 64.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(64), 0, 0, (float*)o, c, steps, 1.0001f, 0.0001f); });
        report("mul+add chains x2 interleaved (per op)", blocks, steps, 128.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(64), 0, 0, (float*)o, c, steps, 1.0001f, 0.0001f); });
        report("mul+add chains x4 interleaved (per op)", blocks, steps, 256.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL(k_chain<4>, dim3(blocks), dim3(64), 0, 0, (float*)o, c, steps, 1.0001f, 0.0001f); });
        report("packed mul+add chain x1 (per packed op)", blocks, steps, 64.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL(k_chain_pk<1>, dim3(blocks), dim3(64), 0, 0, (float*)o, c, steps, 1.0001f, 0.0001f); });
        report("packed chains x2 interleaved (per packed op)", blocks, steps, 128.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL(k_chain_pk<2>, dim3(blocks), dim3(64), 0, 0, (float*)o, c, steps, 1.0001f, 0.0001f); });
        report("packed chains x4 interleaved (per packed op)", blocks, steps, 256.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL(k_chain_pk<4>, dim3(blocks), dim3(64), 0, 0, (float*)o, c, steps, 1.0001f, 0.0001f); });
        report("unit step packed, 1 pair (2 units)", blocks, steps, 1.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL((k_units_pk<1>), dim3(blocks), dim3(64), 0, 0, seed, o, c, steps, 0); });
        report("unit step packed, 2 pairs (4 units)", blocks, steps, 1.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL((k_units_pk<2>), dim3(blocks), dim3(64), 0, 0, seed, o, c, steps, 0); });
        report("unit step plain, 1 unit", blocks, steps, 1.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL((k_units<1, 0>), dim3(blocks), dim3(64), 0, 0, seed, o, c, steps, 0); });
        report("unit step plain, 2 units in sequence", blocks, steps, 1.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL((k_units<2, 0>), dim3(blocks), dim3(64), 0, 0, seed, o, c, steps, 0); });
        report("unit step ilp, 1 unit (selects)", blocks, steps, 1.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL((k_units<1, 1>), dim3(blocks), dim3(64), 0, 0, seed, o, c, steps, 0); });
        report("unit step ilp, 2 units interleaved", blocks, steps, 1.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL((k_units<2, 1>), dim3(blocks), dim3(64), 0, 0, seed, o, c, steps, 0); });
        report("unit step ilp, 3 units interleaved", blocks, steps, 1.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL((k_units<3, 1>), dim3(blocks), dim3(64), 0, 0, seed, o, c, steps, 0); });
        report("unit step ilp, 4 units interleaved", blocks, steps, 1.0, [&](float4* o, unsigned long long* c) { hipLaunchKernelGGL((k_units<4, 1>), dim3(blocks), dim3(64), 0, 0, seed, o, c, steps, 0); });
    }
    return 0;
}
