// mvrl_device.hpp - device-side building blocks shared by the environment kernels (gfx950 / CDNA4).
//
// One wavefront lane = one environment instance.  All per-env state lives in HBM as SoA planes
// (plane k of env i at base[k * n + i]) so that every state load/store of a wave is one fully coalesced
// 256-B transaction; model constants are wave-uniform kernel arguments (scalar loads -> SGPRs).
#pragma once
#ifndef __HIPCC_RTC__   /* hiprtc brings its own runtime declarations and fixed-width types */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#else
using __hip_internal::uint8_t;
using __hip_internal::int32_t;
using __hip_internal::uint32_t;
using __hip_internal::int64_t;
using __hip_internal::uint64_t;
typedef unsigned long uintptr_t;
#ifndef offsetof
#define offsetof(t, m) __builtin_offsetof(t, m)
#endif
#endif

namespace mvrl {
template <class A, class B> struct same_type { static constexpr bool value = false; };
template <class A> struct same_type<A, A> { static constexpr bool value = true; };
}  // namespace mvrl

#define MVRL_BLOCK 256
// Launch bounds of the rigid-body step kernels.  The second argument (min waves per SIMD) caps the register
// budget; tuned on hardware (see DESIGN.md "occupancy").
// Block size of the rigid-body step kernels.  Their lanes never cooperate, so this only sets the dispatch granularity:
// with one wave per block a SIMD gets its next wave as soon as one retires instead of when a whole 4-wave block has
// retired, which de-synchronises the waves' memory phases (+2 % measured against 256).
#ifndef MVRL_STEP_BLOCK
#define MVRL_STEP_BLOCK 64
#endif
#ifdef MVRL_MIN_WAVES
#define MVRL_STEP_BOUNDS __launch_bounds__(MVRL_STEP_BLOCK, MVRL_MIN_WAVES)
#else
#define MVRL_STEP_BOUNDS __launch_bounds__(MVRL_STEP_BLOCK)
#endif

namespace mvrl {

// ---- fp32 math helpers -------------------------------------------------------------------------
__device__ __forceinline__ float fsign(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(hi, fmaxf(lo, x)); }

// This file (like mvrl_rov6/rov3/auv.hip and mvrl_kernels.hpp) is written for fp32; tools/gen_f64.py derives the fp64
// build (namespace mvrl64, every `float` -> `double`, f-suffixed literals and libm names widened) from the same text.
// The few places where the two precisions must differ are under MVRL_F64.
#ifndef MVRL_F64
#define MVRL_F64 0
#endif
// 2*pi split for Cody-Waite style reduction: TWO_PI_HI + TWO_PI_LO == 2*pi beyond working precision
#if MVRL_F64
#define MVRL_TWO_PI_HI 6.283185307179586
#define MVRL_TWO_PI_LO 2.4492935982947064e-16
#define MVRL_INV_TWO_PI 0.15915494309189535
#define MVRL_PI 3.141592653589793
#else
#define MVRL_TWO_PI_HI 6.2831855f
#define MVRL_TWO_PI_LO (-1.7484555e-7f)
#define MVRL_INV_TWO_PI 0.15915494f
#define MVRL_PI 3.14159265f
#endif

// Python float modulo x % (2*pi) -> [0, 2*pi)   (resources.py:92-93, 6DoF.py:560)
__device__ __forceinline__ float mod_two_pi(float x) {
    float q = floorf(x * MVRL_INV_TWO_PI);
    float r = fmaf(-q, MVRL_TWO_PI_HI, x);
#if !MVRL_F64
    r = fmaf(-q, MVRL_TWO_PI_LO, r);   // fp32: HI + LO is the reference's fp64 2*pi.  fp64: see angle_error
#endif
    // q may be off by one when x sits next to a multiple of 2*pi
    r = (r < 0.f) ? r + MVRL_TWO_PI_HI : r;
    r = (r >= MVRL_TWO_PI_HI) ? r - MVRL_TWO_PI_HI : r;
    return r;
}

// resources.angleError (resources.py:75-95): signed difference in [-pi, pi).  The reference takes the smaller of
// (d mod 2pi) and -((-d) mod 2pi); that is d minus the nearest multiple of 2pi, with the tie d = pi (mod 2pi)
// resolved to -pi.
__device__ __forceinline__ float angle_error(float psi_d, float psi) {
    const float d = psi_d - psi;
    const float q = rintf(d * MVRL_INV_TWO_PI);
    float r = fmaf(-q, MVRL_TWO_PI_HI, d);
#if !MVRL_F64
    r = fmaf(-q, MVRL_TWO_PI_LO, r);
#endif
    // fp64 build: the reference's `%` is the EXACT remainder by the fp64 number 2*pi (not by the real 2 pi); the fma above is
    // that remainder exactly, and adding the LO correction would decide the +-pi tie differently from the reference
    // (golden G1 holds 4 such cases; tests/test_gpu_edge.py)
    r = (r >= MVRL_PI) ? r - MVRL_TWO_PI_HI : r;
    r = (r < -MVRL_PI) ? r + MVRL_TWO_PI_HI : r;
    return r;
}

// sin & cos with ~1 ulp accuracy for |x| up to a few thousand radians, branch-free (Cody-Waite reduction by
// pi/2 in three exact-product pieces + Cephes-style minimax polynomials on [-pi/4, pi/4]).
__device__ __forceinline__ void sincos_f32(float x, float& s, float& c) {
#if MVRL_F64
    ::sincos(x, &s, &c);  // fp64 build: the library routine
#else
#ifdef MVRL_NATIVE_TRIG
    // hardware v_sin_f32 / v_cos_f32 (argument in revolutions): ~4x fewer issue slots, ~1e-6 absolute accuracy
    const float rev = x * MVRL_INV_TWO_PI;
    const float fr = rev - floorf(rev);
    s = __builtin_amdgcn_sinf(fr);
    c = __builtin_amdgcn_cosf(fr);
    return;
#endif
    float q = rintf(x * 0.63661977f);
    float r = fmaf(-q, 1.5703125f, x);
    r = fmaf(-q, 4.837512969970703125e-4f, r);
    r = fmaf(-q, 7.54978995489188216e-8f, r);
    float r2 = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float sn = fmaf(ps * r2, r, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    float cn = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
    int n = (int)q;
    float s1 = (n & 1) ? cn : sn;
    float c1 = (n & 1) ? sn : cn;
    s = (n & 2) ? -s1 : s1;
    c = ((n + 1) & 2) ? -c1 : c1;
#endif
}

// ---- binary angles (fp32 build; DESIGN.md section 2 "Euler angles as binary angles") ---------------------------------------
// The reference keeps its Euler angles wrapped to [0, 2 pi) (6DoF.py:560, 3DoF.py:480).  As an fp32 number such an angle is resolved to
// 4.8e-7 rad near 2 pi - and on a whole 250-step episode that storage rounding, re-injected every step, is what separates an fp32
// trajectory from the fp64 one (tests/audit/episode_audit.py: of the envs that an fp64 computation with fp32 state storage loses, 95 % are
// lost to the three angle words).  The state planes of the fp32 build therefore hold the angles as 32-bit BINARY ANGLES: angle = b * 2 pi / 2^32, uniform
// resolution 1.5e-9 rad, the wrap is integer overflow (exact), a step ADDS its small increment, and the one full sincos of a step reduces
// its argument exactly (top two bits = quadrant).  Same 4 bytes per angle; mvrl_get_state hands out the bit patterns like iStep's.
#if !MVRL_F64
#define MVRL_BAM 1
#define MVRL_BAM_RAD 1.4629180792671596e-9f      /* 2 pi / 2^32, rounded to fp32 ...                                  */
#define MVRL_BAM_RAD_LO (-4.0709404e-17f)        /* ... and the rest of it: 2 pi / 2^32 minus that fp32 number            */
#define MVRL_RAD_BAM 683565275.57643159f         /* 2^32 / (2 pi) */
// signed value in [-pi, pi) as ONE fp32 number (two roundings, <= 2.4e-7): where an absolute angle of ordinary fp32 quality is enough
// (the step kernels pass the conversion factors in: a value they pinned to a register next to its uses, so that LLVM does not hoist a
// dozen literals out of the fused-launch loop and keep them in VGPRs across the whole RK4 loop)
__device__ __forceinline__ float bam_to_rad(uint32_t b, float c_rad = MVRL_BAM_RAD) { return (float)(int32_t)b * c_rad; }
// the same angle in the reference's convention [0, 2 pi) (set-point = a * scale + angle is reported that way; 6DoF.py:545-552)
__device__ __forceinline__ float bam_to_rad_pos(uint32_t b, float c_rad = MVRL_BAM_RAD) { const float a = bam_to_rad(b, c_rad); return (a < 0.f) ? a + MVRL_TWO_PI_HI : a; }
// signed value as hi + lo (|lo| <~ 1e-7, the pair exact to ~1e-11): top 24 bits through an exact product error, low 8 bits on top
__device__ __forceinline__ void bam_to_rad2(uint32_t b, float& hi, float& lo) {
    const float th = (float)((int32_t)b >> 8);                  // exact: 24 significant bits
    const float tl = (float)(b & 0xffu);
    const float k_hi = 256.0f * MVRL_BAM_RAD, k_lo = 256.0f * MVRL_BAM_RAD_LO;
    hi = th * k_hi;
    asm("" : "+v"(hi));   // -ffast-math must not see fma(th, k_hi, -(th * k_hi)) as zero: it is the product's rounding error
    lo = fmaf(tl, MVRL_BAM_RAD, fmaf(th, k_lo, fmaf(th, k_hi, -hi)));
}
// b + (angle increment d [rad]); the wrap to [0, 2 pi) of 6DoF.py:560 is the integer overflow.  |d| of any size: reduced first.
__device__ __forceinline__ uint32_t bam_add(uint32_t b, float d, float c_bam = MVRL_RAD_BAM) {
    d = fmaf(-rintf(d * MVRL_INV_TWO_PI), MVRL_TWO_PI_HI, d);   // |d| <= pi (1 + 1e-7): d * 2^32 / (2 pi) is within 2^31 (1 + 1e-7) ...
    // ... so the product is clamped to the largest fp32 below 2^31 before the conversion (an out-of-range float -> int conversion is
    // undefined in C++, whatever v_cvt_i32_f32 does): an increment of exactly half a turn comes out 1.9e-7 rad short at worst
    const float t = fminf(fmaxf(rintf(d * c_bam), -2147483520.f), 2147483520.f);
    return b + (uint32_t)(int32_t)t;
}
// sin and cos of a binary angle: the top two bits (after rounding to the nearest quadrant) ARE the Cody-Waite quotient, the remainder is
// exact, and its conversion to radians carries its rounding error along (first-order correction): ~1 ulp of the RESULT for any angle
__device__ __forceinline__ void sincos_bam(uint32_t b, float& s, float& c, float c_rad = MVRL_BAM_RAD) {
    const uint32_t k = (b + 0x20000000u) >> 30;                 // nearest multiple of pi/2 (4 wraps to 0 with b)
    const int32_t rem = (int32_t)(b - (k << 30));               // [-2^29, 2^29)
    const int32_t rh = rem & ~0x3f;                             // 24 significant bits: exact as fp32
    const float fh = (float)rh, fl = (float)(rem - rh);
    float r = fh * c_rad;
    asm("" : "+v"(r));    // as in bam_to_rad2: the next line's innermost fma is the rounding error of this product
    const float lo = fmaf(fl, c_rad, fmaf(fh, MVRL_BAM_RAD_LO, fmaf(fh, c_rad, -r)));
    const float r2 = r * r;
    const float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float sn = fmaf(ps * r2, r, r);
    const float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    float cn = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
    const float sn0 = sn;
    sn = fmaf(cn, lo, sn);
    cn = fmaf(-sn0, lo, cn);
    const float s1 = (k & 1u) ? cn : sn, c1 = (k & 1u) ? sn : cn;
    s = (k & 2u) ? -s1 : s1;
    c = ((k + 1u) & 2u) ? -c1 : c1;
}
#else
#define MVRL_BAM 0
#endif

// the step counter shares the SoA state buffer with the real-valued planes: stored as an integer bit pattern
#if MVRL_F64
__device__ __forceinline__ int unpack_int(double v) { return (int)__double_as_longlong(v); }
__device__ __forceinline__ double pack_int(int i) { return __longlong_as_double((long long)i); }
#else
__device__ __forceinline__ int unpack_int(float v) { return __float_as_int(v); }
__device__ __forceinline__ float pack_int(int i) { return __int_as_float(i); }
#endif

// ---- Philox4x32-10 counter-based RNG (Salmon et al. 2011) -----------------------------------------
// Streams are keyed by (seed) and counted by (global env id, the env's episode counter, slot) so results do not depend on how
// the batch is sharded over GPUs.
struct Philox4 {
    uint32_t v[4];
};
__device__ __host__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                          uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
// uniform in [0, 1) with 24 random bits (exactly representable in fp32)
__device__ __host__ __forceinline__ float u01(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// ---- turbulence table -----------------------------------------------------------------------------
struct FlowDev {
    // [n_t][n_y][n_x] cells of 64 bytes: cell (t, y, x) holds the whole 2 x 2 x 2 interpolation stencil,
    // [dt][dy][dx] x (u, v) - exactly ONE cache line per lookup.  The library builds this from the caller's plain
    // [n_t][n_y][n_x][2] table at mvrl_set_flow (8 x the memory: 320 MB for the 2000-snapshot table).  In the plain
    // table a lookup touches four grid rows = 4.5 lines per lane, and scattered line requests, not bytes, are what the
    // gathers cost: 9.5 us per million of them on a chip that is streaming the state planes at the same time
    // (tools/plane_layout.hip) - a third of an AuvEnv step.
    const float4* table;
    int n_t, n_y, n_x;
    float inv_dt, inv_dx, inv_dy;
    // the sample TIME is formed in fp64 in both builds (flow_time_index): 1 / dt of the table and 1 / (2 (n_t - 1)) as host-side doubles
    double inv_dt64, inv_period2;
    float t_quarter;      // flow.time[n_t // 4]  (verySimpleAuv.py:245)
    // 0: ReconstructedFlow.interp as it is (AuvEnv, mvrl_flow_interp): indices clamped, weights not - linear EXTRAPOLATION outside
    //    the table.  AuvEnv ends an episode 1 m from the origin after at most 5 s, so it never gets far outside.
    // 1: the 3/6-DoF + turbulence composition (no reference counterpart).  Those vehicles are free to leave the
    //    3.3 m x 2.2 m table and their 50-s episodes outlast its 44 s; extrapolated linearly the "current" grows without bound -
    //    measured in fp64 on BASELINE configs[3]: 11 % of the envs non-finite after 100 steps, all of them after 224
    //    (DESIGN.md section 1).  Outside the table the composition therefore HOLDS the boundary value in space and REFLECTS time
    //    (t -> triangle wave over the table's duration: continuous, unsteady for any episode length); inside it is interp exactly.
    int bounded;
};

// ReconstructedFlow.interp restricted to (u, v) (tag/flowGenerator.py:97-136): cell index clamped, weights NOT
// clamped (linear extrapolation outside the table), origin ignored - all as the reference.
// Split in two so that a kernel can issue the gathers early and consume them late (the loads are a dependent HBM /
// Infinity-Cache round trip): flow_gather = index arithmetic + the loads, flow_combine = the interpolation.
struct FlowTap {
    float2 c000, c001, c010, c011, c100, c101, c110, c111;
    float ft, fx, fy;
};
// Sample time in table units, split into slice index and fraction - in fp64 SCALARS of the lane, whatever the build's precision.
// The env's time is an integer step count times the host's fp64 dt (verySimpleAuv.py:266-267, 6DoF.py:533-534) plus the episode's
// offset; at step 250 of a rigid-body episode time / dt_table is ~2 700, whose fp32 ulp is 2.4e-4 of a slice: formed in fp32 the
// sample time was off by ~5e-5 slices, i.e. 1e-7 .. 6e-7 m/s of current per step (VERDICT r4 "weak 2") - several times the rounding
// of a velocity word, injected every step.  In fp64 the index is exact and the fraction is rounded once, to fp32, at its own size.
__device__ __forceinline__ void flow_time_index(const FlowDev& f, int istep, double dt64, float toff, int& kk, float& ft) {
#if defined(MVRL_FLOW_TIME_F32) && !MVRL_F64   /* attribution build only (tests/audit/episode_audit.py): the round-4 fp32 sample time */
    double tt = (double)(((float)istep * (float)dt64 + toff) * f.inv_dt);
#else
    double tt = ((double)istep * dt64 + (double)toff) * f.inv_dt64;
#endif
    if (f.bounded) {   // wave-uniform: triangle wave over [0, n_t - 1] (see FlowDev::bounded)
        const double per = (double)(f.n_t - 1);
        const double m = tt - 2.0 * per * floor(tt * f.inv_period2);   // [0, 2 per)
        tt = per - fabs(m - per);
    }
    kk = min(f.n_t - 2, max(0, (int)floor(tt)));
    ft = (float)(tt - (double)kk);
}
__device__ __forceinline__ FlowTap flow_gather(const FlowDev& f, int kk, float ft, float x, float y) {
    FlowTap g;
    float xx = x * f.inv_dx, yy = y * f.inv_dy;
    if (f.bounded) {   // wave-uniform
        xx = clampf(xx, 0.f, (float)(f.n_x - 1));
        yy = clampf(yy, 0.f, (float)(f.n_y - 1));
    }
    int ii = min(f.n_x - 2, max(0, (int)floorf(xx)));
    int jj = min(f.n_y - 2, max(0, (int)floorf(yy)));
    g.ft = ft; g.fx = xx - (float)ii; g.fy = yy - (float)jj;
    // one line: (x0, x1) of row y0 at t0 | row y1 at t0 | row y0 at t1 | row y1 at t1
    const float4* q = f.table + (((size_t)kk * f.n_y + jj) * f.n_x + ii) * 4;
    const float4 a = q[0], b = q[1], c = q[2], d = q[3];
    g.c000 = make_float2(a.x, a.y); g.c001 = make_float2(a.z, a.w);
    g.c010 = make_float2(b.x, b.y); g.c011 = make_float2(b.z, b.w);
    g.c100 = make_float2(c.x, c.y); g.c101 = make_float2(c.z, c.w);
    g.c110 = make_float2(d.x, d.y); g.c111 = make_float2(d.z, d.w);
    return g;
}
__device__ __forceinline__ float2 flow_combine(const FlowTap& g) {
    const float wt0 = 1.f - g.ft, wx0 = 1.f - g.fx, wy0 = 1.f - g.fy, fx = g.fx, fy = g.fy, ft = g.ft;
    float u0 = wy0 * (g.c000.x * wx0 + g.c001.x * fx) + fy * (g.c010.x * wx0 + g.c011.x * fx);
    float v0 = wy0 * (g.c000.y * wx0 + g.c001.y * fx) + fy * (g.c010.y * wx0 + g.c011.y * fx);
    float u1 = wy0 * (g.c100.x * wx0 + g.c101.x * fx) + fy * (g.c110.x * wx0 + g.c111.x * fx);
    float v1 = wy0 * (g.c100.y * wx0 + g.c101.y * fx) + fy * (g.c110.y * wx0 + g.c111.y * fx);
    return make_float2(u0 * wt0 + u1 * ft, v0 * wt0 + v1 * ft);
}
__device__ __forceinline__ float2 flow_interp_uv(const FlowDev& f, int istep, double dt64, float toff, float x, float y) {
    int kk;
    float ft;
    flow_time_index(f, istep, dt64, toff, kk, ft);
    return flow_combine(flow_gather(f, kk, ft, x, y));
}

// ---- device mirrors of the model constants (fp32) -------------------------------------------------
struct Rov6Dev {
    float m, wb;            // mass, W - B
    float cg[3];
    float I[9];
    float gw[3];            // (xg*W - xb*B), (yg*W - yb*B), (zg*W - zb*B)
    float added[6];
    float minv[36];
    float dlin[36];
    float dquad[36];
    float A[48];
    float Ainv[48];
    // sign-symmetric thruster layout (valid when the host selected the SYM kernel):
    //   sym_a    = |A[0,0]| |A[1,0]| |A[2,4]| |A[3,0]| |A[3,4]| |A[4,0]| |A[4,4]| |A[5,0]|
    //   sym_ainv = |Ainv[0,0]| |Ainv[0,1]| |Ainv[0,5]| |Ainv[4,0]| |Ainv[4,1]| |Ainv[4,2]| |Ainv[4,3]| |Ainv[4,4]|
    float sym_a[8];
    float sym_ainv[8];
    // Coriolis constants of the structured (SYM) right-hand side, grouped by velocity product (dynamics6):
    //   m z_g,  m - added[2],  m - added[1],  m - added[0],
    //   Izz - Iyy + added[4] - added[5],  Ixx - Izz + added[5] - added[3],  Iyy - Ixx + added[3] - added[4],  0
    float sym_c[8];
    float thrust_k, inv_thrust_k, rpm_max, rpm_dead, f_max, f_dead;
    float kp[6], ki[6], kd[6], windup[6], umax[6];
    float act_scale[6];
    float inv_obs_pos, inv_obs_ang;
};

struct Rov3Dev {
    float m, cgx, cgy;
    float xud, yvd;
    float minv[9], dlin[9], dquad[9];
    float Ainv[12];
    float thrust_k, inv_thrust_k, rpm_max, rpm_dead, f_max, f_dead;
    float cos_a, sin_a, yaw_arm, inv_jet_area_k, jet_c1, jet_k1, jet_c2, jet_k2, jet_drag_k;
    float kp[3], ki[3], kd[3], windup[3], umax[3];
    float act_scale[3];
    float inv_obs_pos, inv_obs_ang;
};

struct AuvDev {
    float m, izz, xuu, yvv, nrr, xu, yv, nr, max_force, max_moment;
    float x_min, x_max, y_min, y_max;
    float noise_coeffs, noise_act;
    int stop_on_bounds;
    int n_wp;               // > 0: AuvEnvCyl way-point following
    float obs_scale[9];
    float wp_thr;
    float wp[96];           // x, y, target heading per way-point
};

// Model constants are read through a CONSTANT-address-space pointer so that they come in through the scalar
// cache (s_load_dwordxN -> SGPR operands, no VALU or VGPR cost).  A 6-DoF RHS touches ~140 distinct constants,
// more than the ~100 SGPRs a wave owns; left alone, LLVM hoists all of them out of the RK4 loop and spills the
// excess into VGPR lanes (v_writelane/v_readlane = one VALU slot per use).  `launder` hides the pointer's
// provenance at the start of each phase (PID / allocation / dynamics), so each phase re-issues a few wide
// scalar loads (scalar-cache hits) instead.
typedef const __attribute__((address_space(4))) Rov6Dev* CP6;
typedef const __attribute__((address_space(4))) Rov3Dev* CP3;
typedef const __attribute__((address_space(4))) AuvDev* CPA;
template <class T>
__device__ __forceinline__ const __attribute__((address_space(4))) T* as_const(const T* p) {
    return (const __attribute__((address_space(4))) T*)(uintptr_t)p;
}
template <class PT>
__device__ __forceinline__ PT launder(PT p) {
    asm volatile("" : "+s"(p));
    return p;
}
// Row-by-row variant for the dense (generic) forms: the pointer is re-hidden together with a value the previous row
// produced, so that the loads of row i cannot be issued (and their destinations kept alive) before row i - 1 is done -
// an asm without such a data dependence floats to the top of the block, all rows' loads behind it.
template <class PT>
__device__ __forceinline__ PT launder_after(PT p, float& x) {
    asm volatile("" : "+s"(p), "+v"(x));
    return p;
}
}  // namespace mvrl
#include "mvrl_baked.inc"
namespace mvrl {
// A wave-uniform run-time scalar used as a VALU operand lives in an SGPR, and on gfx950 a VALU instruction with an SGPR
// operand issues at HALF rate (tools/valu_operands.hip: 0.55 vs 1.0 wave-instr/ns/SIMD; literals and inline constants
// are free).  `in_vgpr` pins such a value to a VGPR once, outside the hot loop.
__device__ __forceinline__ float in_vgpr(float x) {
#if MVRL_F64
    return x;
#else
#ifndef MVRL_NO_VGPR_SCALARS
    asm volatile("" : "+v"(x));
#endif
    return x;
#endif
}

// "ctrl" flavour: the reference's vehicle (literals) with a run-time controller - PID gains, wind-up and output limits,
// action / observation scales are what users retune, the hull is not.  Rov6BakedVeh's ordinary members mirror the tail of
// Rov6Dev from `kp` on, so the pointer is simply &params->kp in the constant address space (38 scalar-loaded words).
typedef const __attribute__((address_space(4))) Rov6BakedVeh* CPV6;
static_assert(sizeof(Rov6BakedVeh) == sizeof(Rov6Dev) - offsetof(Rov6Dev, kp), "Rov6BakedVeh must mirror the tail of Rov6Dev");

// Baked flavour: `p->field` resolves to a static constexpr member, i.e. an instruction literal; nothing to launder.
__device__ __forceinline__ const Rov6Baked* launder(const Rov6Baked* p) { return p; }
__device__ __forceinline__ const Rov6Baked* launder_after(const Rov6Baked* p, float&) { return p; }
__device__ __forceinline__ const Rov3Baked* launder(const Rov3Baked* p) { return p; }
template <class PP, class T>
__device__ __forceinline__ PP param_ptr(const T* pg) {
    if constexpr (same_type<PP, const Rov6Baked*>::value || same_type<PP, const Rov3Baked*>::value) {
        return nullptr;  // never dereferenced: every member is static
    } else if constexpr (same_type<PP, CPV6>::value) {
        return (PP)(uintptr_t)(&pg->kp[0]);
    } else {
        return (PP)(uintptr_t)pg;
    }
}

// ---- common launch arguments ----------------------------------------------------------------------
struct StepIO {
    float* state;           // SoA planes [words][n]
    const float* actions;   // [n][act_dim]
    float* obs;             // [n][obs_dim]
    float* reward;          // [n]
    uint8_t* done;          // [n]
    float* term_obs;        // [n][obs_dim] or nullptr
    float* aux;             // [n][aux_dim] or nullptr
    int* nfev;              // [n] RHS evaluations of the adaptive integrator, or nullptr
    int64_t n;
    int64_t env_offset;
    uint64_t seed;
    int n_sub;
    int max_steps;
    int fixed_sp;
    int auto_reset;
    float dt;
    double dt64;            // the host's dt unrounded: the turbulence sample time is istep * dt64 (flow_time_index)
    int k_steps;            // > 1: fused multi-step launch (actions / obs / reward / done are [k_steps][n][...])
    // lanes [lane0, lane_end) of the n-lane batch are stepped by this launch (mvrl_step_range_dev: independent chains of
    // sub-batches on their own streams); every array is still indexed by the lane's position in the whole batch
    int64_t lane0, lane_end;
};

}  // namespace mvrl
