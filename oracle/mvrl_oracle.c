/*
 * mvrl_oracle.c - CPU restatement of the reference's environment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity ORACLE: a plain-C restatement of the algorithm of
 * UnnamedMoose/MarineVehicleReinforcementLearning for the path named in BASELINE.json.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it - as the checker / the CPU baseline,
 * never as (or behind) the product path.  libmvrl.so does not link, call or fall back to anything here.
 *
 * Pinning: PINNED.  The reference ships no tests for this path, so the oracle is pinned against golden
 * vectors produced by importing the reference itself in the build container (oracle/gen/gen_golden_*.py,
 * fixtures under tests/golden/; numpy 2.2.6 / scipy 1.15.3 recorded in each fixture) plus the one
 * known-answer triple the reference holds (example_temp.py:19-28 -> g14).  tests/test_oracle_*.py checks
 * every function below against them.  The 3/6-DoF "+ turbulence current" branches (cur != 0) have no
 * reference counterpart (dead code behind np.zeros, 6DoF.py:258 / 3DoF.py:183): parity by construction
 * only; outside the table that composition holds the boundary value in space and
 * reflects time (orc_flow_sample_bounded) instead of extrapolating, see DESIGN.md section 1.
 *
 * Compiled twice (oracle/Makefile): -DREAL=double -DSUF=_f64 and -DREAL=float -DSUF=_f32.  Time variables
 * stay double in both builds (they are host-side Python floats in the reference).
 *
 * File abbreviations: 6DoF.py = dynamicsModel_BlueROV2_Heavy_6DoF.py, 3DoF.py = dynamicsModel_BlueROV2_Heavy_3DoF.py,
 * tag/ = tag_00_Dec2023_simpleControlTurbulence/.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/mvrl.h"

#ifndef REAL
#define REAL double
#define SUF _f64
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

typedef REAL real;

#define TWO_PI 6.283185307179586476925286766559

/* Distance-to-discontinuity bookkeeping (tests only): while tl_margin points at a record, every RHS evaluation lowers
 * the record's entries to the smallest distance seen between the trajectory and a point where the reference's
 * right-hand side JUMPS (or, entry 4, amplifies rounding without bound):
 *   [0] |e - eOld| / (h * S) of a PID axis at a call with t - tOld <= 1e-9 (6DoF.py:64-66: dedt = +-(e - eOld)/1e-9 -> the
 *       demand sits on the +-umax rail given by the SIGN of a difference of two nearby errors).  The difference is an
 *       increment of the pose over (a fraction of) the sub-step h, i.e. h x a combination of pose rates sum_j J_ij nu_j;
 *       S = sum_j |J_ij nu_j| is the size of the terms that rate is made of, so the ratio says how completely they
 *       cancel - which is what another precision must reproduce to get the sign right.  Exact zeros (at rest) excluded
 *   [1] | |rpm| / deadband - 1 |  of a thruster after saturation (limit(), 6DoF.py:271-275: |rpm| < 300 -> 0)
 *   [2] | |e| - windup |  of an axis whose integral is non-zero (eInt[|e| > windup] = 0, 6DoF.py:68)
 *   [3] pi - |yaw error|  (angleError's branch at +-pi, resources.py:92-95)
 *   [4] |cos(theta)|      (J2 ~ 1/cos(theta), resources.py:116-132; below 1e-6 the guard itself switches)
 * A trajectory computed in another precision may legitimately leave the reference trajectory only where one of these
 * was within that precision's rounding of the quantity concerned. */
#define ORC_N_MARGIN 5
static __thread double* FN(tl_margin) = 0;
static __thread double FN(tl_h) = 1.0;          /* sub-step of the RK4 harness */
static __thread double FN(tl_rate)[6] = {1, 1, 1, 1, 1, 1};   /* S_i of the current RHS evaluation */
static inline void margin_note(int k, double v) {
    double* m = FN(tl_margin);
    if (m && v < m[k]) m[k] = v;
}

static inline real r_abs(real x) { return x < 0 ? -x : x; }
static inline real r_sign(real x) { return (real)((x > 0) - (x < 0)); }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_min(real a, real b) { return a < b ? a : b; }
static inline real r_sqrt(real x) { return (real)sqrt((double)x); }
static inline real r_sin(real x) { return sizeof(real) == 4 ? (real)sinf((float)x) : (real)sin((double)x); }
static inline real r_cos(real x) { return sizeof(real) == 4 ? (real)cosf((float)x) : (real)cos((double)x); }
static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }

/* Python / numpy float modulo: fmod, then move the result to the sign of the divisor (CPython float_rem). */
static inline real py_mod(real a, real b) {
    real m = sizeof(real) == 4 ? (real)fmodf((float)a, (float)b) : (real)fmod((double)a, (double)b);
    if (m != 0) {
        if ((b < 0) != (m < 0)) m += b;
    } else {
        m = (real)copysign(0.0, (double)b);
    }
    return m;
}

/* resources.angleError (resources.py:75-95) == tag/resources.headingError (tag/resources.py:26-46) */
real FN(orc_angle_error)(real psi_d, real psi) {
    real a = py_mod(psi_d - psi, (real)TWO_PI);
    real b = py_mod(psi - psi_d, (real)TWO_PI);
    return a < b ? a : -b;
}

/* Body axes of BlueROV2Heavy6DoF.updateMovingCoordSystem (6DoF.py:238-242): iHat,jHat,kHat = columns of
 * R = Rx(phi) Ry(theta) Rz(psi) (scipy 'XYZ' intrinsic).  axes = [iHat; jHat; kHat] row-major, so
 * globalToVehicle(v) (6DoF.py:244-248) = axes . v */
void FN(orc_body_axes)(const real ang[3], real axes[9]) {
    real sp = r_sin(ang[0]), cp = r_cos(ang[0]);
    real st = r_sin(ang[1]), ct = r_cos(ang[1]);
    real ss = r_sin(ang[2]), cs = r_cos(ang[2]);
    axes[0] = ct * cs;  axes[1] = cp * ss + sp * st * cs;  axes[2] = sp * ss - cp * st * cs;
    axes[3] = -ct * ss; axes[4] = cp * cs - sp * st * ss;  axes[5] = sp * cs + cp * st * ss;
    axes[6] = st;       axes[7] = -sp * ct;                axes[8] = cp * ct;
}

static inline void g2v(const real axes[9], const real v[3], real out[3]) {
    for (int i = 0; i < 3; i++) out[i] = v[0] * axes[3 * i] + v[1] * axes[3 * i + 1] + v[2] * axes[3 * i + 2];
}

/* resources.coordinateTransform(phi, theta, psi, dof=6) (resources.py:98-143), INCLUDING the J1[0,2] typo
 * (sin(phi) where Fossen has cos(phi), resources.py:123) and the cos(theta) guard (:116-120). */
void FN(orc_coord_transform6)(real phi, real theta, real psi, real J[36]) {
    real sp = r_sin(phi), cp = r_cos(phi), st = r_sin(theta), ct = r_cos(theta), ss = r_sin(psi), cs = r_cos(psi);
    real cd = ct;
    margin_note(4, (double)r_abs(ct));
    if (r_abs(cd) < (real)1e-12) cd = (real)1e-6;
    else if (r_abs(cd) < (real)1e-6) cd = (real)1e-6 * r_sign(cd);
    memset(J, 0, 36 * sizeof(real));
    J[0] = cs * ct;  J[1] = -ss * cp + cs * st * sp;  J[2] = ss * sp + cs * st * sp;
    J[6] = ss * ct;  J[7] = cs * cp + ss * st * sp;   J[8] = -cs * sp + ss * st * cp;
    J[12] = -st;     J[13] = ct * sp;                 J[14] = ct * cp;
    J[21] = 1;       J[22] = sp * st / cd;            J[23] = cp * st / cd;
    J[27] = 0;       J[28] = cp;                      J[29] = -sp;
    J[33] = 0;       J[34] = sp / cd;                 J[35] = cp / cd;
}

/* ------------------------------------------------------------------------------------------------
 * PID state: eOld (None until the first call), eInt, tOld   (6DoF.py:37-41)                        */
typedef struct {
    real eold[6];
    real eint[6];
    double told;
    int32_t has_old;
} FN(orc_pid);
typedef FN(orc_pid) pid_t_;

/* Shared PID law: 6DoF.py:62-71 == 3DoF.py:147-157 */
static void pid_law(int n, const real* e, double t, pid_t_* s, const double* kp, const double* ki, const double* kd,
                    const double* windup, const double* umax, real* out) {
    if (!s->has_old) {
        for (int i = 0; i < n; i++) s->eold[i] = e[i];
        s->has_old = 1;
    }
    double dtp = t - s->told;
    real den = (real)(dtp > 1e-9 ? dtp : 1e-9);
    for (int i = 0; i < n; i++) {
        real dedt = (e[i] - s->eold[i]) / den;
        if (dtp <= 1e-9 && e[i] != s->eold[i])
            margin_note(0, (double)r_abs(e[i] - s->eold[i]) / (FN(tl_h) * (FN(tl_rate)[i] > 1e-12 ? FN(tl_rate)[i] : 1e-12)));
        s->eint[i] += (real)0.5 * (s->eold[i] + e[i]) * (real)dtp;
        if (s->eint[i] != 0) margin_note(2, (double)r_abs(r_abs(e[i]) - (real)windup[i]));
        if (r_abs(e[i]) > (real)windup[i]) s->eint[i] = 0;
        real u = (real)kp[i] * e[i] + (real)kd[i] * dedt + (real)ki[i] * s->eint[i];
        u = r_max(-(real)umax[i], r_min((real)umax[i], u));
        out[i] = u;
        s->eold[i] = e[i];
    }
    s->told = t;
}

/* BlueROV2Heavy6DoF_PID_controller.computeControlForces (6DoF.py:43-73) */
void FN(orc_pid6)(const mvrl_rov6_params* p, const real sp[6], const real pose[6], double t, pid_t_* s, real out[6]) {
    real e[6];
    e[0] = sp[0] - pose[0]; e[1] = sp[1] - pose[1]; e[2] = sp[2] - pose[2];
    e[3] = sp[3] - pose[3]; e[4] = sp[4] - pose[4];
    e[5] = FN(orc_angle_error)(sp[5], pose[5]);
    margin_note(3, 3.14159265358979323846 - (double)r_abs(e[5]));
    pid_law(6, e, t, s, p->kp, p->ki, p->kd, p->windup, p->umax, out);
}

/* allocateThrust (6DoF.py:220-231): body-frame demand -> Ainv -> rpm */
void FN(orc_alloc6)(const mvrl_rov6_params* p, const real axes[9], const real gcf[6], real rpm[8]) {
    real b[6];
    g2v(axes, gcf, b);
    g2v(axes, gcf + 3, b + 3);
    for (int i = 0; i < 8; i++) {
        real cv = 0;
        for (int j = 0; j < 6; j++) cv += (real)p->alloc_inv[6 * i + j] * b[j];
        rpm[i] = r_sign(cv) * r_sqrt(r_abs(cv) / (real)p->thrust_k) * 60;
    }
}

static inline real limit_rpm(real x, real rmax, real dead) { /* 6DoF.py:271-275 */
    real r = r_max(-rmax, r_min(rmax, x));
    margin_note(1, (double)r_abs(r_abs(r) / dead - 1));
    if (r_abs(r) < dead) r = 0;
    return r;
}

/* forceModel (6DoF.py:253-404).  cur_body: current velocity already in the body frame (6 comps; zeros in the
 * reference).  Outputs RHS[6]; comp (may be NULL) = retComp layout [6][5]: -Crb v, -Ca v, -D v, G, H. */
void FN(orc_force_model6)(const mvrl_rov6_params* P, const real ang[3], const real vel[6], const real rpms[8],
                          const real cur_body[6], real RHS[6], real* comp) {
    real phi = ang[0], theta = ang[1];
    real u = vel[0], v = vel[1], w = vel[2], p = vel[3], q = vel[4], r = vel[5];
    real m = (real)P->m;
    real xg = (real)P->cg[0], yg = (real)P->cg[1], zg = (real)P->cg[2];
    const double* I = P->inertia;
    real Ixx = (real)I[0], Ixy = (real)I[1], Ixz = (real)I[2], Iyy = (real)I[4], Iyz = (real)I[5], Izz = (real)I[8];
    real velRel[6];
    for (int i = 0; i < 6; i++) velRel[i] = vel[i] - (cur_body ? cur_body[i] : 0);

    real H[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 8; i++) {
        real rp = limit_rpm(rpms[i], (real)P->rpm_max, (real)P->rpm_deadband);
        real F = (real)P->thrust_k * (rp / 60) * (rp / 60) * r_sign(rp); /* thrusterModel 6DoF.py:233-236 */
        for (int k = 0; k < 6; k++) H[k] += F * (real)P->alloc[8 * k + i];
    }
    real Crb[36] = {
        0, 0, 0, m * (yg * q + zg * r), -m * (xg * q - w), -m * (xg * r + v),
        0, 0, 0, -m * (yg * p + w), m * (zg * r + xg * p), -m * (yg * r - u),
        0, 0, 0, -m * (zg * p - v), -m * (zg * q + u), m * (xg * p + yg * q),
        -m * (yg * q + zg * r), m * (yg * p + w), m * (zg * p - v), 0, -Iyz * q - Ixz * p + Izz * r, Iyz * r + Ixy * p - Iyy * q,
        m * (xg * q - w), -m * (zg * r + xg * p), m * (zg * q + u), Iyz * q + Ixz * p - Izz * r, 0, -Ixz * r - Ixy * q + Ixx * p,
        m * (xg * r + v), m * (yg * r - u), -m * (xg * p + yg * q), -Iyz * r - Ixy * p + Iyy * q, Ixz * r + Ixy * q - Ixx * p, 0};
    real Xud = (real)P->added[0], Yvd = (real)P->added[1], Zwd = (real)P->added[2];
    real Kpd = (real)P->added[3], Mqd = (real)P->added[4], Nrd = (real)P->added[5];
    real Ca[36] = {
        0, 0, 0, 0, -Zwd * w, Yvd * v,
        0, 0, 0, Zwd * w, 0, -Xud * u,
        0, 0, 0, -Yvd * v, Xud * u, 0,
        0, -Zwd * w, Yvd * v, 0, -Nrd * r, Mqd * q,
        Zwd * w, 0, -Xud * u, Nrd * r, 0, -Kpd * p,
        -Yvd * v, Xud * u, 0, -Mqd * q, Kpd * p, 0};
    real D[36];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) D[6 * i + j] = (real)P->dlin[6 * i + j] + (real)P->dquad[6 * i + j] * r_abs(vel[j]);
    real W = (real)P->weight, B = (real)P->buoyancy;
    real xb = (real)P->cb[0], yb = (real)P->cb[1], zb = (real)P->cb[2];
    real st = r_sin(theta), ct = r_cos(theta), sp = r_sin(phi), cp = r_cos(phi);
    real G[6] = {(W - B) * st,
                 -(W - B) * ct * sp,
                 -(W - B) * ct * cp,
                 -(yg * W - yb * B) * ct * cp + (zg * W - zb * B) * ct * sp,
                 (zg * W - zb * B) * st + (xg * W - xb * B) * ct * cp,
                 -(xg * W - xb * B) * ct * sp - (yg * W - yb * B) * st};
    for (int i = 0; i < 6; i++) {
        real c1 = 0, c2 = 0, ca_v = 0, d_v = 0;
        for (int j = 0; j < 6; j++) {
            c1 += Crb[6 * i + j] * vel[j];
            c2 += (Ca[6 * i + j] + D[6 * i + j]) * velRel[j];
            ca_v += Ca[6 * i + j] * vel[j];
            d_v += D[6 * i + j] * vel[j];
        }
        RHS[i] = -c1 - c2 - G[i] + H[i]; /* 6DoF.py:396 (E = 0) */
        if (comp) {
            comp[5 * i + 0] = -c1; comp[5 * i + 1] = -ca_v; comp[5 * i + 2] = -d_v; comp[5 * i + 3] = G[i]; comp[5 * i + 4] = H[i];
        }
    }
}

/* Everything of derivs after the controller (6DoF.py:424-442) for given rpm. cur_glob = (u_c, v_c) global. */
/* Perturbation ensemble (tests only; tests/parity_util.ensemble_sensitive).  With noise > 0 the RK4 harness multiplies the
 * state by 1 + noise * U(-1, 1) after every sub-step, the set-point once per step and every thruster's rpm at every RHS
 * evaluation (the thrusters pull against each other: a force or moment can be a small difference of eight large terms, so
 * a relative rounding of the terms is a much larger relative rounding of the sum - the one place where noise on the state
 * alone would under-state what fp32 arithmetic does, in particular from rest).  A stand-in for the rounding an fp32
 * implementation commits, used to ask whether the fp64 algorithm ITSELF is stable at fp32 resolution for a given env.
 * Off (0) unless a test switches it on; every golden-vector check runs with it off. */
static double FN(g_noise) = 0.0;
static uint64_t FN(g_noise_seed) = 0;
void FN(orc_set_noise)(double noise, uint64_t seed) { FN(g_noise) = noise; FN(g_noise_seed) = seed; }
static __thread double FN(tl_noise) = 0.0;
static __thread uint64_t FN(tl_rng) = 0;
static double noise_u(uint64_t* st) {   /* splitmix64 -> U(-1, 1) */
    uint64_t z = (*st += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    z ^= z >> 31;
    return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}
/* Rounding emulation (tests/audit/attribution_cpu.py only; off for every golden-vector check): switches that make the fp64 build
 * commit ONE class of fp32 rounding, so that its effect on a whole episode can be measured alone.
 *   1: every RHS output (dy) rounded to fp32 - the least any fp32 right-hand side commits (one rounding of each result)
 *   2: the turbulence sample time formed in fp32, as the round-4 kernels did (flow_gather of that round)
 *   4: the sampled current (u_c, v_c) rounded to fp32
 *   8: the state rounded to fp32 after every RK4 sub-step (instead of once per env step, which the callers do on the arrays)
 *  16: forceModel + M^-1 + J (everything of derivs after the allocation) evaluated in fp32 ARITHMETIC by the fp32 build of this
 *      file on fp32-rounded inputs, its result widened again: what the plain fp32 formulation of the right-hand side commits
 *      with exact state, exact controller and exact allocation
 *  32: the same for allocateThrust (body-frame demand, pinv(A) product, rpm map) */
static int FN(g_emulate) = 0;
void FN(orc_set_emulate)(int mask) { FN(g_emulate) = mask; }
static inline real emu_round(real x) { return (real)(float)x; }

static const real* noisy_rpm(const real* rpm, int n, real* buf) {
    if (!(FN(tl_noise) > 0)) return rpm;
    for (int k = 0; k < n; k++) buf[k] = rpm[k] * (real)(1.0 + FN(tl_noise) * noise_u(&FN(tl_rng)));
    return buf;
}

static void rhs6_given_rpm(const mvrl_rov6_params* P, const real y[12], const real axes[9], const real rpm_in[8],
                           const real cur_glob[2], real dy[12]) {
    real rpm_buf[8];
    const real* rpm = noisy_rpm(rpm_in, 8, rpm_buf);
    real cur_body[6] = {0, 0, 0, 0, 0, 0};
    if (cur_glob && (cur_glob[0] != 0 || cur_glob[1] != 0)) {
        real cg3[3] = {cur_glob[0], cur_glob[1], 0};
        g2v(axes, cg3, cur_body); /* SURVEY 9.5 */
    }
    real RHS[6];
    FN(orc_force_model6)(P, y + 3, y + 6, rpm, cur_body, RHS, NULL);
    for (int i = 0; i < 6; i++) {
        real a = 0;
        for (int j = 0; j < 6; j++) a += (real)P->minv[6 * i + j] * RHS[j]; /* np.linalg.solve(M, RHS) 6DoF.py:428 */
        dy[6 + i] = a;
    }
    real J[36];
    FN(orc_coord_transform6)(y[3], y[4], y[5], J);
    for (int i = 0; i < 6; i++) {
        real a = 0;
        for (int j = 0; j < 6; j++) a += J[6 * i + j] * y[6 + j];
        dy[i] = a;
    }
}

void FN(orc_rhs6_given_rpm)(const mvrl_rov6_params* P, const real y[12], const real axes[9], const real rpm_in[8], const real cur_glob[2],
                            real dy[12]) {
    rhs6_given_rpm(P, y, axes, rpm_in, cur_glob, dy);
}
#if defined(ORC_IS_F64)
void orc_rhs6_given_rpm_f32(const mvrl_rov6_params*, const float*, const float*, const float*, const float*, float*);
void orc_alloc6_f32(const mvrl_rov6_params*, const float*, const float*, float*);
void orc_body_axes_f32(const float*, float*);
#endif

/* BlueROV2Heavy6DoF.derivs (6DoF.py:406-442) */
void FN(orc_derivs6)(const mvrl_rov6_params* P, double t, const real y[12], const real sp[6], pid_t_* pid,
                     const real cur_glob[2], real dy[12], real gcf[6], real rpm[8]) {
    real axes[9];
    FN(orc_body_axes)(y + 3, axes);
    if (FN(tl_margin)) {   /* size of the terms each pose rate is made of (margin [0]) */
        real J[36];
        double* save = FN(tl_margin);
        FN(tl_margin) = 0;
        FN(orc_coord_transform6)(y[3], y[4], y[5], J);
        FN(tl_margin) = save;
        for (int i = 0; i < 6; i++) {
            double sabs = 0;
            for (int j = 0; j < 6; j++) sabs += (double)r_abs(J[6 * i + j] * y[6 + j]);
            FN(tl_rate)[i] = sabs;
        }
    }
    FN(orc_pid6)(P, sp, y, t, pid, gcf);
#if defined(ORC_IS_F64)
    if (FN(g_emulate) & (16 | 32)) {   /* attribution only: parts of the right-hand side in fp32 arithmetic (see orc_set_emulate) */
        float yf[12], af[9], gf[6], rf[8], cf[2] = {0, 0}, df[12];
        for (int i = 0; i < 12; i++) yf[i] = (float)y[i];
        for (int i = 0; i < 6; i++) gf[i] = (float)gcf[i];
        orc_body_axes_f32(yf + 3, af);
        if (FN(g_emulate) & 32) {
            orc_alloc6_f32(P, af, gf, rf);
            for (int i = 0; i < 8; i++) rpm[i] = rf[i];
        } else {
            FN(orc_alloc6)(P, axes, gcf, rpm);
            for (int i = 0; i < 8; i++) rf[i] = (float)rpm[i];
        }
        if (FN(g_emulate) & 16) {
            if (cur_glob) { cf[0] = (float)cur_glob[0]; cf[1] = (float)cur_glob[1]; }
            orc_rhs6_given_rpm_f32(P, yf, af, rf, cf, df);
            for (int i = 0; i < 12; i++) dy[i] = df[i];
        } else {
            rhs6_given_rpm(P, y, axes, rpm, cur_glob, dy);
        }
        return;
    }
#endif
    FN(orc_alloc6)(P, axes, gcf, rpm);
    rhs6_given_rpm(P, y, axes, rpm, cur_glob, dy);
}

/* ------------------------------------------------------------------------------------------------
 * 3-DoF: BlueROV2Heavy3DoF.derivs (3DoF.py:128-296)                                                */
static void ctrl3(const mvrl_rov3_params* P, double t, const real y[6], const real sp[3], pid_t_* pid, real gcf[3],
                  real rpm[4]) {
    real psi = y[2];
    real e[3] = {sp[0] - y[0], sp[1] - y[1], FN(orc_angle_error)(sp[2], psi)};
    margin_note(3, 3.14159265358979323846 - (double)r_abs(e[2]));
    if (FN(tl_margin)) {   /* eta_dot = J(psi) nu (3DoF.py:288): sizes of the terms of each pose rate (margin [0]) */
        double cp = fabs((double)r_cos(psi)), sn = fabs((double)r_sin(psi));
        FN(tl_rate)[0] = cp * fabs((double)y[3]) + sn * fabs((double)y[4]);
        FN(tl_rate)[1] = sn * fabs((double)y[3]) + cp * fabs((double)y[4]);
        FN(tl_rate)[2] = fabs((double)y[5]);
    }
    real cvv[3];
    pid_law(3, e, t, pid, P->kp, P->ki, P->kd, P->windup, P->umax, cvv);
    real c = r_cos(psi), s = r_sin(psi);
    gcf[0] = cvv[0] * c + cvv[1] * s;   /* 3DoF.py:160-163 */
    gcf[1] = -cvv[0] * s + cvv[1] * c;
    gcf[2] = cvv[2];
    for (int i = 0; i < 4; i++) {
        real cv = 0;
        for (int j = 0; j < 3; j++) cv += (real)P->alloc_inv[3 * i + j] * gcf[j];
        rpm[i] = r_sign(cv) * r_sqrt(r_abs(cv) / (real)P->thrust_k) * 60; /* :166-168 */
    }
}

static void rhs3_given_rpm(const mvrl_rov3_params* P, const real y[6], const real rpm_in[4], const real cur_glob[2],
                           real dy[6]) {
    real rpm_buf[4];
    const real* rpm = noisy_rpm(rpm_in, 4, rpm_buf);
    real psi = y[2], u = y[3], v = y[4], r = y[5];
    real vel[3] = {u, v, r};
    real c = r_cos(psi), s = r_sin(psi);
    real cur[3] = {0, 0, 0};
    if (cur_glob && (cur_glob[0] != 0 || cur_glob[1] != 0)) { /* 3DoF.py:186-188: pinv(J) = J^T */
        cur[0] = c * cur_glob[0] + s * cur_glob[1];
        cur[1] = -s * cur_glob[0] + c * cur_glob[1];
    }
    real velRel[3] = {u - cur[0], v - cur[1], r - cur[2]};
    real uRel = velRel[0], vRel = velRel[1];
    real m = (real)P->m, xg = (real)P->cg[0], yg = (real)P->cg[1];
    real Crb[9] = {0, 0, -m * (xg * r + v), 0, 0, -m * (yg * r - u), m * (xg * r + v), m * (yg * r - u), 0};
    real Xud = (real)P->added[0], Yvd = (real)P->added[1];
    real Ca[9] = {0, 0, Yvd * vRel, 0, 0, -Xud * uRel, -Yvd * vRel, Xud * uRel, 0};
    real av[3] = {r_abs(uRel), r_abs(vRel), r_abs(r)};
    real D[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) D[3 * i + j] = (real)P->dlin[3 * i + j] + (real)P->dquad[3 * i + j] * av[j];
    real F[4], X[4];
    for (int i = 0; i < 4; i++) { /* thrusterModel 3DoF.py:114-126 */
        real rp = limit_rpm(rpm[i], (real)P->rpm_max, (real)P->rpm_deadband);
        F[i] = (real)P->thrust_k * (rp / 60) * (rp / 60) * r_sign(rp);
        real uJet = r_sqrt(r_abs(F[i]) / (real)P->jet_area_k);
        real den = r_max((real)1e-5, uJet);
        real dCd = (real)P->jet_c1 * r_exp(-(real)P->jet_k1 * r_abs(u) / den) + (real)P->jet_c2 * r_exp(-(real)P->jet_k2 * r_abs(u) / den);
        X[i] = dCd * -(real)P->jet_drag_k * r_abs(u) * u;
    }
    /* order FP, AP, FS, AS = rpm[0..3] (3DoF.py:177-180, 244-247) */
    real Xh = X[0] + X[1] + X[2] + X[3] + (F[0] + F[1] - F[2] - F[3]) * (real)P->cos_alpha;
    real Yh = (F[0] - F[1] + F[2] - F[3]) * (real)P->sin_alpha;
    real Nh = (real)P->yaw_arm * (F[0] + F[1] + F[2] + F[3]);
    real Hh[3] = {Xh, Yh, Nh};
    real RHS[3];
    for (int i = 0; i < 3; i++) {
        real c1 = 0, c2 = 0;
        for (int j = 0; j < 3; j++) {
            c1 += Crb[3 * i + j] * vel[j];
            c2 += (Ca[3 * i + j] + D[3 * i + j]) * velRel[j];
        }
        RHS[i] = -c1 - c2 + Hh[i];
    }
    for (int i = 0; i < 3; i++) {
        real a = 0;
        for (int j = 0; j < 3; j++) a += (real)P->minv[3 * i + j] * RHS[j];
        dy[3 + i] = a;
    }
    dy[0] = c * u - s * v;
    dy[1] = s * u + c * v;
    dy[2] = r;
}

void FN(orc_derivs3)(const mvrl_rov3_params* P, double t, const real y[6], const real sp[3], pid_t_* pid,
                     const real cur_glob[2], real dy[6], real gcf[3], real rpm[4]) {
    ctrl3(P, t, y, sp, pid, gcf, rpm);
    rhs3_given_rpm(P, y, rpm, cur_glob, dy);
}

/* ------------------------------------------------------------------------------------------------
 * Generic RHS closure used by the two integrators                                                  */
typedef struct {
    int dof;      /* 3 or 6 */
    int zoh;      /* 1: rpm held, controller not evaluated inside f */
    const mvrl_rov6_params* p6;
    const mvrl_rov3_params* p3;
    const real* sp;
    pid_t_* pid;
    const real* cur;
    real gcf[6];
    real rpm[8];
    int64_t nfev;
} rhs_ctx;

static void rhs_eval(rhs_ctx* c, double t, const real* y, real* dy) {
    c->nfev++;
    if (c->dof == 6) {
        if (c->zoh) {
            real axes[9];
            FN(orc_body_axes)(y + 3, axes);
            rhs6_given_rpm(c->p6, y, axes, c->rpm, c->cur, dy);
        } else {
            FN(orc_derivs6)(c->p6, t, y, c->sp, c->pid, c->cur, dy, c->gcf, c->rpm);
        }
    } else {
        if (c->zoh) rhs3_given_rpm(c->p3, y, c->rpm, c->cur, dy);
        else FN(orc_derivs3)(c->p3, t, y, c->sp, c->pid, c->cur, dy, c->gcf, c->rpm);
    }
    if (FN(g_emulate) & 1)
        for (int i = 0; i < 2 * c->dof; i++) dy[i] = emu_round(dy[i]);
}

static void zoh_control(rhs_ctx* c, double t, const real* y) {
    if (c->dof == 6) {
        real axes[9];
        FN(orc_body_axes)(y + 3, axes);
        FN(orc_pid6)(c->p6, c->sp, y, t, c->pid, c->gcf);
        FN(orc_alloc6)(c->p6, axes, c->gcf, c->rpm);
    } else {
        ctrl3(c->p3, t, y, c->sp, c->pid, c->gcf, c->rpm);
    }
}

/* Fixed-step classic RK4 over [t0, t0+dt] in n_sub sub-steps: OUR harness (the reference has no fixed-step
 * integrator), identical to oracle/gen/gen_golden_root.py::rk4_env_step. */
static void integrate_rk4(rhs_ctx* c, double t0, double dt, int n_sub, real* y) {
    int n = 2 * c->dof;
    double h = dt / n_sub;
    real hh = (real)h;
    FN(tl_h) = h;
    real k1[12], k2[12], k3[12], k4[12], yt[12];
    for (int k = 0; k < n_sub; k++) {
        double tk = t0 + k * h;
        if (c->zoh) zoh_control(c, tk, y);
        rhs_eval(c, tk, y, k1);
        for (int i = 0; i < n; i++) yt[i] = y[i] + (real)0.5 * hh * k1[i];
        rhs_eval(c, tk + 0.5 * h, yt, k2);
        for (int i = 0; i < n; i++) yt[i] = y[i] + (real)0.5 * hh * k2[i];
        rhs_eval(c, tk + 0.5 * h, yt, k3);
        for (int i = 0; i < n; i++) yt[i] = y[i] + hh * k3[i];
        rhs_eval(c, tk + h, yt, k4);
        for (int i = 0; i < n; i++) y[i] = y[i] + (hh / 6) * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        if (FN(tl_noise) > 0)
            for (int i = 0; i < n; i++) y[i] *= (real)(1.0 + FN(tl_noise) * noise_u(&FN(tl_rng)));
        if (FN(g_emulate) & 8)
            for (int i = 0; i < n; i++) y[i] = emu_round(y[i]);
    }
}

/* scipy.integrate.solve_ivp(method='RK45', t_eval=[t_bound], max_step=dt, rtol=atol=1e-3) as called at
 * 6DoF.py:555-557 / 3DoF.py:475-477, restated from scipy 1.15.3 (scipy/integrate/_ivp/rk.py: rk_step :14-75,
 * RungeKutta.__init__ :85-106, _step_impl :111-179, RK45 tableau :377-405, RkDenseOutput :552-574;
 * common.py: norm :63-65, select_initial_step :68-134).  A fresh solver per env step: f0 and the initial-step
 * probe both hit the stateful PID. */
static real rms_norm(const real* x, int n) {
    double s = 0;
    for (int i = 0; i < n; i++) s += (double)x[i] * (double)x[i];
    return (real)(sqrt(s) / sqrt((double)n));
}

static const double RK45_C[6] = {0, 1. / 5, 3. / 10, 4. / 5, 8. / 9, 1};
static const double RK45_A[6][5] = {{0, 0, 0, 0, 0},
                                    {1. / 5, 0, 0, 0, 0},
                                    {3. / 40, 9. / 40, 0, 0, 0},
                                    {44. / 45, -56. / 15, 32. / 9, 0, 0},
                                    {19372. / 6561, -25360. / 2187, 64448. / 6561, -212. / 729, 0},
                                    {9017. / 3168, -355. / 33, 46732. / 5247, 49. / 176, -5103. / 18656}};
static const double RK45_B[6] = {35. / 384, 0, 500. / 1113, 125. / 192, -2187. / 6784, 11. / 84};
static const double RK45_E[7] = {-71. / 57600, 0, 71. / 16695, -71. / 1920, 17253. / 339200, -22. / 525, 1. / 40};
static const double RK45_P[7][4] = {
    {1, -8048581381. / 2820520608, 8663915743. / 2820520608, -12715105075. / 11282082432},
    {0, 0, 0, 0},
    {0, 131558114200. / 32700410799, -68118460800. / 10900136933, 87487479700. / 32700410799},
    {0, -1754552775. / 470086768, 14199869525. / 1410260304, -10690763975. / 1880347072},
    {0, 127303824393. / 49829197408, -318862633887. / 49829197408, 701980252875. / 199316789632},
    {0, -282668133. / 205662961, 2019193451. / 616988883, -1453857185. / 822651844},
    {0, 40617522. / 29380423, -110615467. / 29380423, 69997945. / 29380423}};

static int integrate_rk45(rhs_ctx* c, double t0, double t_bound, double max_step, double rtol, double atol, real* y) {
    int n = 2 * c->dof;
    real f[12], K[7][12], y_new[12], f_new[12], scale[12], tmp[12], y_old[12];
    double t = t0;
    rhs_eval(c, t, y, f); /* RungeKutta.__init__: self.f = fun(t0, y0) */
    /* select_initial_step(fun, t0, y0, t_bound, max_step, f0, direction=+1, order=4, rtol, atol) */
    double h_abs;
    {
        double interval = fabs(t_bound - t0);
        if (interval == 0.0) {
            h_abs = 0.0;
        } else {
            for (int i = 0; i < n; i++) scale[i] = (real)atol + r_abs(y[i]) * (real)rtol;
            for (int i = 0; i < n; i++) tmp[i] = y[i] / scale[i];
            double d0 = rms_norm(tmp, n);
            for (int i = 0; i < n; i++) tmp[i] = f[i] / scale[i];
            double d1 = rms_norm(tmp, n);
            double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
            if (h0 > interval) h0 = interval;
            real y1[12], f1[12];
            for (int i = 0; i < n; i++) y1[i] = y[i] + (real)h0 * f[i];
            rhs_eval(c, t0 + h0, y1, f1);
            for (int i = 0; i < n; i++) tmp[i] = (f1[i] - f[i]) / scale[i];
            double d2 = rms_norm(tmp, n) / h0;
            double h1;
            if (d1 <= 1e-15 && d2 <= 1e-15) h1 = fmax(1e-6, h0 * 1e-3);
            else h1 = pow(0.01 / fmax(d1, d2), 1.0 / 5.0);
            h_abs = fmin(fmin(100 * h0, h1), fmin(interval, max_step));
        }
    }
    int have_step = 0;
    double h_last = 0;
    while (t != t_bound) { /* OdeSolver.step until finished */
        double min_step = 10 * fabs(nextafter(t, INFINITY) - t);
        double ha;
        if (h_abs > max_step) ha = max_step;
        else if (h_abs < min_step) ha = min_step;
        else ha = h_abs;
        int accepted = 0, rejected = 0;
        double t_new = t, h = 0;
        while (!accepted) {
            if (ha < min_step) return -1; /* TOO_SMALL_STEP */
            h = ha;
            t_new = t + h;
            if (t_new - t_bound > 0) t_new = t_bound;
            h = t_new - t;
            ha = fabs(h);
            /* rk_step */
            for (int i = 0; i < n; i++) K[0][i] = f[i];
            for (int s = 1; s < 6; s++) {
                for (int i = 0; i < n; i++) {
                    real a = 0;
                    for (int j = 0; j < s; j++) a += K[j][i] * (real)RK45_A[s][j];
                    tmp[i] = y[i] + a * (real)h;
                }
                rhs_eval(c, t + RK45_C[s] * h, tmp, K[s]);
            }
            for (int i = 0; i < n; i++) {
                real a = 0;
                for (int j = 0; j < 6; j++) a += K[j][i] * (real)RK45_B[j];
                y_new[i] = y[i] + (real)h * a;
            }
            rhs_eval(c, t + h, y_new, f_new);
            for (int i = 0; i < n; i++) K[6][i] = f_new[i];
            for (int i = 0; i < n; i++) {
                real sc = (real)atol + r_max(r_abs(y[i]), r_abs(y_new[i])) * (real)rtol;
                real e = 0;
                for (int j = 0; j < 7; j++) e += K[j][i] * (real)RK45_E[j];
                tmp[i] = e * (real)h / sc;
            }
            double err = rms_norm(tmp, n);
            if (err < 1) {
                double factor = (err == 0) ? 10.0 : fmin(10.0, 0.9 * pow(err, -0.2));
                if (rejected) factor = fmin(1.0, factor);
                ha *= factor;
                accepted = 1;
            } else {
                ha *= fmax(0.2, 0.9 * pow(err, -0.2));
                rejected = 1;
            }
        }
        for (int i = 0; i < n; i++) { y_old[i] = y[i]; y[i] = y_new[i]; f[i] = f_new[i]; }
        t = t_new;
        h_abs = ha;
        h_last = h;
        have_step = 1;
    }
    /* t_eval = [t_bound] is served by the dense output of the final step at x = 1 (ivp.py + rk.py:552-574):
     * y = y_old + h * sum_j K[j] * (sum_k P[j][k] * 1^k) */
    if (have_step) {
        for (int i = 0; i < n; i++) {
            real acc = 0;
            for (int j = 0; j < 7; j++) {
                real pj = (real)(((RK45_P[j][0] + RK45_P[j][1]) + RK45_P[j][2]) + RK45_P[j][3]);
                acc += K[j][i] * pj;
            }
            y[i] = y_old[i] + (real)h_last * acc;
        }
    }
    return 0;
}

/* dataToState (6DoF.py:467-483 / 3DoF.py:397-409); iWp is always 0 in the reference. */
static inline real clip1(real x) { return r_max((real)-1, r_min((real)1, x)); }

void FN(orc_obs6)(const mvrl_rov6_params* P, const real y[12], const real path[6], const real sp[6], real obs[9]) {
    real s = (real)P->obs_pos_scale, a = (real)P->obs_ang_scale;
    for (int k = 0; k < 3; k++) obs[k] = clip1((path[k] - y[k]) / s);
    for (int k = 0; k < 3; k++) obs[3 + k] = clip1((path[3 + k] - y[k]) / s);
    for (int k = 0; k < 3; k++) obs[6 + k] = clip1(FN(orc_angle_error)(sp[3 + k], y[3 + k]) / a);
}

void FN(orc_obs3)(const mvrl_rov3_params* P, const real y[6], const real path[4], const real sp[3], real obs[5]) {
    real s = (real)P->obs_pos_scale, a = (real)P->obs_ang_scale;
    obs[0] = clip1((path[0] - y[0]) / s);
    obs[1] = clip1((path[1] - y[1]) / s);
    obs[2] = clip1((path[2] - y[0]) / s);
    obs[3] = clip1((path[3] - y[1]) / s);
    obs[4] = clip1(FN(orc_angle_error)(sp[2], y[2]) / a);
}

/* ------------------------------------------------------------------------------------------------
 * ReconstructedFlow.interp (tag/flowGenerator.py:97-136). table [n_t][n_y][n_x][n_comp]              */
void FN(orc_flow_interp)(const real* table, int n_t, int n_y, int n_x, int n_comp, double fdt, double fdx, double fdy,
                         real time, real x, real y, real* out) {
    real tt = time / (real)fdt, xx = x / (real)fdx, yy = y / (real)fdy;
    int kk = (int)floor((double)tt), ii = (int)floor((double)xx), jj = (int)floor((double)yy);
    kk = kk < 0 ? 0 : (kk > n_t - 2 ? n_t - 2 : kk);
    ii = ii < 0 ? 0 : (ii > n_x - 2 ? n_x - 2 : ii);
    jj = jj < 0 ? 0 : (jj > n_y - 2 ? n_y - 2 : jj);
    real wt[2] = {1 - (tt - kk), tt - kk};
    real wx[2] = {1 - (xx - ii), xx - ii};
    real wy[2] = {1 - (yy - jj), yy - jj};
    for (int k = 0; k < n_comp; k++) {
        real res = 0;
        for (int a = 0; a < 2; a++) {
            const real* tb = table + ((size_t)(kk + a) * n_y) * n_x * n_comp;
            /* yy^T (F xx): F[j][i] */
            real r0 = tb[((size_t)(jj)*n_x + ii) * n_comp + k] * wx[0] + tb[((size_t)(jj)*n_x + ii + 1) * n_comp + k] * wx[1];
            real r1 = tb[((size_t)(jj + 1) * n_x + ii) * n_comp + k] * wx[0] + tb[((size_t)(jj + 1) * n_x + ii + 1) * n_comp + k] * wx[1];
            res += (wy[0] * r0 + wy[1] * r1) * wt[a];
        }
        out[k] = res;
    }
}

/* The 3/6-DoF + turbulence composition (no reference counterpart): inside the table exactly
 * interp; outside it the boundary value is HELD in space and time is REFLECTED (triangle wave over the table's duration), because
 * those vehicles leave the 3.3 m x 2.2 m table and their 50-s episodes outlast its 44 s - extrapolated linearly the current grows
 * without bound and every env ends non-finite.  Same arithmetic as flow_gather (mvrl_device.hpp) with f.bounded = 1. */
void FN(orc_flow_sample_bounded)(const real* table, int n_t, int n_y, int n_x, int n_comp, double fdt, double fdx, double fdy,
                                 real time, real x, real y, real* out) {
    real tt = time / (real)fdt, xx = x / (real)fdx, yy = y / (real)fdy;
    const real per = (real)(n_t - 1);
    const real m = tt - 2 * per * (real)floor((double)(tt / (2 * per)));
    tt = per - (real)fabs((double)(m - per));
    xx = xx < 0 ? 0 : (xx > (real)(n_x - 1) ? (real)(n_x - 1) : xx);
    yy = yy < 0 ? 0 : (yy > (real)(n_y - 1) ? (real)(n_y - 1) : yy);
    FN(orc_flow_interp)(table, n_t, n_y, n_x, n_comp, fdt, fdx, fdy, tt * (real)fdt, xx * (real)fdx, yy * (real)fdy, out);
}

void FN(orc_flow_interp_batch)(const real* table, int n_t, int n_y, int n_x, int n_comp, double fdt, double fdx,
                               double fdy, const real* t, const real* x, const real* y, int64_t n, real* out) {
    for (int64_t i = 0; i < n; i++) FN(orc_flow_interp)(table, n_t, n_y, n_x, n_comp, fdt, fdx, fdy, t[i], x[i], y[i], out + i * n_comp);
}

/* ------------------------------------------------------------------------------------------------
 * Batched env steps (AoS arrays of per-env fields; OpenMP over envs).
 * integrator: 0 = RK4 harness (n_sub, control_mode), 1 = scipy RK45 (the reference's env.step).
 * flow: table==NULL -> no current.  time[] is advanced like the reference (time += dt in fp64).      */
int FN(orc_rov_step)(int dof, const mvrl_rov6_params* p6, const mvrl_rov3_params* p3, int64_t n, double dt,
                     int integrator, int n_sub, int control_mode, int fixed_sp, int max_steps,
                     const real* actions, real* y, real* sp, const real* path, real* eold, real* eint, double* told,
                     int32_t* has_old, int32_t* istep, double* time,
                     const real* flow_table, int f_nt, int f_ny, int f_nx, double f_dt, double f_dx, double f_dy,
                     const real* toffset,
                     real* obs, real* reward, uint8_t* done, real* gcf_out, real* rpm_out, int64_t* nfev_out,
                     double* margins_out /* [n][ORC_N_MARGIN] smallest distances of this step, or NULL */) {
    int ns = 2 * dof, nthr = dof == 6 ? 8 : 4, nobs = dof == 6 ? 9 : 5, npath = dof == 6 ? 6 : 4;
    int status = 0;
#pragma omp parallel for schedule(static) reduction(min : status)
    for (int64_t e = 0; e < n; e++) {
        real* ye = y + e * ns;
        real* spe = sp + e * dof;
        pid_t_ pid;
        for (int i = 0; i < dof; i++) { pid.eold[i] = eold[e * dof + i]; pid.eint[i] = eint[e * dof + i]; }
        pid.told = told[e];
        pid.has_old = has_old[e];
        istep[e] += 1;   /* 6DoF.py:533 */
        time[e] += dt;   /* 6DoF.py:534 */
        if (!fixed_sp) { /* 6DoF.py:545-552 / 3DoF.py:469-472 */
            const real* a = actions + e * dof;
            if (dof == 6) for (int i = 0; i < 6; i++) spe[i] = a[i] * (real)p6->act_scale[i] + ye[i];
            else for (int i = 0; i < 3; i++) spe[i] = a[i] * (real)p3->act_scale[i] + ye[i];
        }
        real cur[2] = {0, 0};
        if (flow_table) { /* sampled once per step at the pre-step position, as verySimpleAuv.py:291 */
            real res[2];
            real tsample = (real)time[e] + toffset[e];
            if (FN(g_emulate) & 2)   /* the round-4 kernels: ((float)istep * dt + toff) * (1 / dt_table), all in fp32 */
                tsample = (real)(((float)istep[e] * (float)dt + (float)toffset[e]) * (float)(1.0 / f_dt)) * (real)f_dt;
            FN(orc_flow_sample_bounded)(flow_table, f_nt, f_ny, f_nx, 2, f_dt, f_dx, f_dy, tsample, ye[0], ye[1], res);
            cur[0] = res[0]; cur[1] = res[1];
            if (FN(g_emulate) & 4) { cur[0] = emu_round(cur[0]); cur[1] = emu_round(cur[1]); }
        }
        rhs_ctx c;
        memset(&c, 0, sizeof(c));
        c.dof = dof; c.zoh = (integrator == 0 && control_mode == MVRL_CTRL_ZOH); c.p6 = p6; c.p3 = p3; c.sp = spe;
        c.pid = &pid; c.cur = cur;
        FN(tl_noise) = FN(g_noise);
        if (FN(tl_noise) > 0) {
            FN(tl_rng) = FN(g_noise_seed) ^ ((uint64_t)e * 0xd1342543de82ef95ull) ^ ((uint64_t)istep[e] << 40);
            if (!fixed_sp) for (int i = 0; i < dof; i++) spe[i] *= (real)(1.0 + FN(tl_noise) * noise_u(&FN(tl_rng)));
        }
        double t0 = time[e] - dt;
        if (margins_out) {
            for (int i = 0; i < ORC_N_MARGIN; i++) margins_out[e * ORC_N_MARGIN + i] = 1e300;
            FN(tl_margin) = margins_out + e * ORC_N_MARGIN;
        }
        if (integrator == 0) integrate_rk4(&c, t0, dt, n_sub, ye);
        else if (integrate_rk45(&c, t0, time[e], dt, 1e-3, 1e-3, ye) != 0) status = -1;
        FN(tl_margin) = 0;
        FN(tl_noise) = 0.0;
        if (dof == 6) for (int i = 3; i < 6; i++) ye[i] = py_mod(ye[i], (real)TWO_PI); /* 6DoF.py:560 */
        else ye[2] = py_mod(ye[2], (real)TWO_PI);                                      /* 3DoF.py:480 */
        if (dof == 6) FN(orc_obs6)(p6, ye, path + e * npath, spe, obs + e * nobs);
        else FN(orc_obs3)(p3, ye, path + e * npath, spe, obs + e * nobs);
        done[e] = istep[e] >= max_steps; /* 6DoF.py:569-571 */
        reward[e] = 0;                   /* 6DoF.py:575 */
        for (int i = 0; i < dof; i++) { eold[e * dof + i] = pid.eold[i]; eint[e * dof + i] = pid.eint[i]; }
        told[e] = pid.told;
        has_old[e] = pid.has_old;
        if (gcf_out) for (int i = 0; i < dof; i++) gcf_out[e * dof + i] = c.gcf[i];
        if (rpm_out) for (int i = 0; i < nthr; i++) rpm_out[e * nthr + i] = c.rpm[i];
        if (nfev_out) nfev_out[e] = c.nfev;
    }
    return status;
}

/* ------------------------------------------------------------------------------------------------
 * AuvEnv (tag/verySimpleAuv.py).  Per-env fields:
 *   pose[6] = x, y, heading, vx, vy, r (velocities are GLOBAL-frame, :321-326)
 *   tgt[1]  = headingTarget ; err_o[3] = herr_o, perr_o[2] ; mult[11] ; toffset ; hist[10][3] newest first ; istep */
void FN(orc_auv_obs)(const mvrl_auv_params* P, const real pose[6], const real target[3], real* err_o, int has_err_o,
                     real obs[11]) {
    /* dataToState: "V3" of AuvEnv (verySimpleAuv.py:201-212, positionTarget = 0 :241) or "V0" of AuvEnvCyl
     * (verySimpleAuv_cyl.py:100-111) - the two differ only in the divisors, carried by P->obs_scale as reciprocals */
    real perr[2] = {target[0] - pose[0], target[1] - pose[1]};
    real herr = FN(orc_angle_error)(target[2], pose[2]);
    if (!has_err_o) { err_o[0] = herr; err_o[1] = perr[0]; err_o[2] = perr[1]; }
    const double* sc = P->obs_scale;
    obs[0] = clip1(perr[0] * (real)sc[0]);
    obs[1] = clip1(perr[1] * (real)sc[1]);
    obs[2] = clip1(herr * (real)sc[2]);
    obs[3] = clip1((herr - err_o[0]) * (real)sc[3]);
    obs[4] = clip1((perr[0] - err_o[1]) * (real)sc[4]);
    obs[5] = clip1((perr[1] - err_o[2]) * (real)sc[5]);
    obs[6] = clip1(pose[3] * (real)sc[6]); obs[7] = clip1(pose[4] * (real)sc[7]); obs[8] = clip1(pose[5] * (real)sc[8]);
    obs[9] = 0; obs[10] = 0;
}

void FN(orc_auv_step)(const mvrl_auv_params* P, int64_t n, double dt, int max_steps, const real* actions, real* pose,
                      real* tgt /* [n][3]: positionTarget, headingTarget */, int32_t* iwp, real* err_o, const real* mult,
                      const real* toffset, real* hist, int32_t* istep,
                      const real* flow_table, int f_nt, int f_ny, int f_nx, double f_dt, double f_dx, double f_dy,
                      real* obs, real* reward, uint8_t* done, real* aux /* [n][11]: Fhydro3, velCurrent2, rmsAc, terms5 */) {
    const real PI = (real)3.14159265358979323846;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; e++) {
        real* ps = pose + e * 6;
        const real* a = actions + e * 3;
        const real* mu = mult + e * 11; /* m I Xuu Yvv Nrr Xu Yv Nr Xact Yact Nact */
        istep[e] += 1;                  /* :266 */
        real time = (real)(istep[e] * dt); /* :267 (time += dt from 0) */
        int dn = istep[e] >= max_steps; /* :270-272 */
        real* hs = hist + e * 30;       /* deque.appendleft :275 */
        memmove(hs + 3, hs, 27 * sizeof(real));
        hs[0] = a[0]; hs[1] = a[1]; hs[2] = a[2];
        int nh = istep[e] < 10 ? istep[e] : 10;
        real Fset0 = a[0] * (real)P->max_force * mu[8], Fset1 = a[1] * (real)P->max_force * mu[9]; /* :278 */
        real Nset = a[2] * (real)P->max_moment * mu[10];                                            /* :279 */
        real c = r_cos(ps[2]), s = r_sin(ps[2]);
        real cur[2] = {0, 0};
        if (flow_table) {
            real res[2];
            FN(orc_flow_interp)(flow_table, f_nt, f_ny, f_nx, 2, f_dt, f_dx, f_dy, time + toffset[e], ps[0], ps[1], res); /* :291 */
            cur[0] = res[0]; cur[1] = res[1];
        }
        real dvx = ps[3] - cur[0], dvy = ps[4] - cur[1];
        real vr0 = c * dvx + s * dvy, vr1 = -s * dvx + c * dvy; /* :298 invJ = J^T */
        real rr = ps[5];
        real Fh0 = ((real)P->xu * mu[5] + (real)P->xuu * mu[2] * r_abs(vr0)) * vr0; /* :303-307 */
        real Fh1 = ((real)P->yv * mu[6] + (real)P->yvv * mu[3] * r_abs(vr1)) * vr1;
        real Fh2 = ((real)P->nr * mu[7] + (real)P->nrr * mu[4] * r_abs(rr)) * rr;
        real Fg0 = c * Fh0 - s * Fh1, Fg1 = s * Fh0 + c * Fh1, Fg2 = Fh2; /* :310 */
        real acc0 = (Fg0 + Fset0) / ((real)P->m * mu[0]);                 /* :314-318 */
        real acc1 = (Fg1 + Fset1) / ((real)P->m * mu[0]);
        real acc2 = (Fg2 + Nset) / ((real)P->izz * mu[1]);
        real h = (real)dt;
        real nx = ps[0] + ps[3] * h, ny = ps[1] + ps[4] * h;             /* :321-326 Euler */
        real nh_ = py_mod(ps[2] + ps[5] * h, (real)TWO_PI);
        real nvx = ps[3] + acc0 * h, nvy = ps[4] + acc1 * h, nr = ps[5] + acc2 * h;
        real npose[6] = {nx, ny, nh_, nvx, nvy, nr};
        real* eo = err_o + e * 3;
        real* tg = tgt + e * 3;
        FN(orc_auv_obs)(P, npose, tg, eo, 1, obs + e * 11); /* :329 (herr_o/perr_o from the previous step) */
        real bonus = 0;
        if (nx < (real)P->x_min || nx > (real)P->x_max) { if (P->stop_on_bounds) dn = 1; bonus += -100; } /* :335-342 */
        if (ny < (real)P->y_min || ny > (real)P->y_max) { if (P->stop_on_bounds) dn = 1; bonus += -100; }
        real perr0 = tg[0] - nx, perr1 = tg[1] - ny;
        real herr = FN(orc_angle_error)(tg[2], nh_);
        if (P->n_waypoints > 0 && r_sqrt(perr0 * perr0 + perr1 * perr1) < (real)P->wp_threshold) { /* _cyl.py:249-253 */
            iwp[e] = iwp[e] + 1 < P->n_waypoints ? iwp[e] + 1 : P->n_waypoints - 1;
            tg[0] = (real)P->waypoints[3 * iwp[e]]; tg[1] = (real)P->waypoints[3 * iwp[e] + 1]; tg[2] = (real)P->waypoints[3 * iwp[e] + 2];
        }
        eo[0] = herr; eo[1] = perr0; eo[2] = perr1; /* :349-350 */
        real rms = 0; /* :353-355 */
        for (int k = 0; k < 3; k++) {
            real mean = 0;
            for (int j = 0; j < nh; j++) mean += hs[3 * j + k];
            mean /= nh;
            real ss = 0;
            for (int j = 0; j < nh; j++) ss += (hs[3 * j + k] - mean) * (hs[3 * j + k] - mean);
            rms += r_sqrt(ss / nh);
        }
        rms /= 3;
        real hdeg = herr / PI * 180;
        real t0 = r_exp(-5 * r_sqrt(perr0 * perr0 + perr1 * perr1));
        real t1 = r_abs(herr) < PI / 2 ? r_exp((real)-0.1 * r_abs(hdeg)) : -r_exp((real)-0.1 * (180 - r_abs(hdeg)));
        real t2 = r_exp((real)-0.6 * rms);
        real t3 = (real)-0.1 * (a[0] * a[0] + a[1] * a[1] + a[2] * a[2]) / 3;
        reward[e] = (((t0 + t1) + t2) + t3) + bonus; /* :357-381 */
        done[e] = (uint8_t)dn;
        for (int k = 0; k < 6; k++) ps[k] = npose[k]; /* :384-386 */
        if (aux) {
            real* ax = aux + e * 11;
            ax[0] = Fg0; ax[1] = Fg1; ax[2] = Fg2; ax[3] = cur[0]; ax[4] = cur[1]; ax[5] = rms;
            ax[6] = t0; ax[7] = t1; ax[8] = t2; ax[9] = t3; ax[10] = bonus;
        }
    }
}

int FN(orc_real_size)(void) { return (int)sizeof(real); }
