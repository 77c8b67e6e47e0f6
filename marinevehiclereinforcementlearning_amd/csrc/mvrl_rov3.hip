// mvrl_rov3.hip - BlueROV2 Heavy 3-DoF environment step / reset kernels for gfx950.
//
// Replaces BlueROV2Heavy3DoFEnv.step/reset/dataToState and BlueROV2Heavy3DoF.derivs with its inlined PID,
// 4-thruster allocation and jet-drag augment (dynamicsModel_BlueROV2_Heavy_3DoF.py:114-296, :397-514).
// Same structure as the 6-DoF kernel: one lane per env, SoA state read/written once per step.
#include "mvrl_kernels.hpp"
#if MVRL_F64
#include "mvrl_rk45.hpp"
#endif

namespace mvrl {

// where the timeHistory side outputs of a call go, if anywhere; `on` is wave-uniform: a scalar branch (AuxRow in mvrl_rov6.hip)
struct AuxRow3 {
    bool on;
    float* row;
};

struct Pid3 {
    float eold[3];
    float eint[3];
};

// PID + body-frame resolution + allocation + saturation (3DoF.py:141-180) -> limited thruster forces F[4]
// USE_INC / dpose: see pid6 in mvrl_rov6.hip - the error difference of two nearby RK stages is taken from the
// stage slopes, not from the rounded fp32 states.
// z: the pose in ERROR coordinates, z = setPoint - pose (yaw: the unwrapped difference) - see rov6_step_kernel in mvrl_rov6.hip.
// e0 / fixed: with a fixed set-point the integrated variable is the displacement since the start of the step and the error is
// e0 + z (see pid6 in mvrl_rov6.hip); `fixed` is wave-uniform.
template <bool HAS_DT, bool USE_INC, class PP>
__device__ __forceinline__ void control3(PP p, const float* z, Pid3& s, float dtp, float inv_den,
                                         const float* dpose, bool inc_valid, float c, float sn, float* F, const AuxRow3& aux,
                                         bool fixed, const float* e0) {
    p = launder(p);
    float e[3] = {z[0], z[1], 0.f}, z2 = z[2];
    // e0: the offset of the controller's error against the integrated variable - setPoint - pose_start with a fixed set-point, ZERO in the
    // action mode (the caller's `ec`): added unconditionally (z + 0 is z), where a test of the wave-uniform `fixed` became three
    // add-and-select pairs per call
    e[0] += e0[0]; e[1] += e0[1]; z2 += e0[2];
    (void)fixed;
#if !defined(MVRL_NO_YAW_INC)
    // yaw error carried from call to call inside an env step (see pid6): previous error minus the yaw increment, wrapped
    float yaw_w = 0.f;
    if (USE_INC && inc_valid) {
        const float r1 = s.eold[2] - dpose[2];
        yaw_w = (r1 >= MVRL_PI) ? -MVRL_TWO_PI_HI : ((r1 < -MVRL_PI) ? MVRL_TWO_PI_HI : 0.f);
        e[2] = r1 + yaw_w;
        // (one turn of correction is all the carried error can need here: the heading rate r is a state of this model - a few rad/s, bounded
        // by the yaw damping - and moves the heading by h/2 * r << pi per stage.  The 6-DoF model's Euler-angle rates are not bounded that
        // way: see pid6 in mvrl_rov6.hip.)
    } else {
        e[2] = angle_error(z2, 0.f);
    }
#define MVRL_YAW_INC3_ON 1
#else
    e[2] = angle_error(z2, 0.f);
#endif
    float u[3], dev[3];
#if defined(MVRL_YAW_INC3_ON) && !defined(MVRL_INC_SELECT)
    if (USE_INC && inc_valid) {   // wave-uniform: a scalar branch (see pid6 in mvrl_rov6.hip)
        asm volatile("");
        dev[0] = -dpose[0]; dev[1] = -dpose[1]; dev[2] = yaw_w - dpose[2];
    } else {
#pragma unroll
        for (int i = 0; i < 3; i++) dev[i] = e[i] - s.eold[i];
    }
#else
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float de = e[i] - s.eold[i];
        if (USE_INC) {
            const float di = -dpose[i];
#ifdef MVRL_YAW_INC3_ON
            de = inc_valid ? ((i < 2) ? di : yaw_w + di) : de;
#else
            const bool use = (i < 2) ? inc_valid : (inc_valid && fabsf(de - di) <= 1e-5f);
            de = use ? di : de;
#endif
        }
        dev[i] = de;
    }
#endif
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float de = dev[i];
        // dtp / 2 and K_D / dt are the same for every call of a step: loop-invariant products (literals x one register in the
        // baked flavour), hoisted by the compiler
        if (HAS_DT) s.eint[i] = fmaf(s.eold[i] + e[i], 0.5f * dtp, s.eint[i]);
        s.eint[i] = (fabsf(e[i]) > p->windup[i]) ? 0.f : s.eint[i];
        float v = fmaf(p->ki[i], s.eint[i], fmaf(p->kd[i] * inv_den, de, p->kp[i] * e[i]));
        u[i] = clampf(v, -p->umax[i], p->umax[i]);
        s.eold[i] = e[i];
    }
    float Xd = u[0] * c + u[1] * sn, Yd = -u[0] * sn + u[1] * c, Nd = u[2];
    float cvs[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        float cv = fmaf(p->Ainv[3 * i + 2], Nd, fmaf(p->Ainv[3 * i + 1], Yd, p->Ainv[3 * i] * Xd));
        float f = clampf(cv, -p->f_max, p->f_max);   // rpm clamp +-3500 and dead-band 300 in force space (3DoF.py:171-180)
        F[i] = (fabsf(f) < p->f_dead) ? 0.f : f;
        cvs[i] = cv;
    }
    if (aux.on) {
        float* const aux_row = aux.row;
        aux_row[0] = Xd; aux_row[1] = Yd; aux_row[2] = Nd;   // timeHistory F0..F2 (3DoF.py:498-507)
#pragma unroll
        for (int i = 0; i < 4; i++) aux_row[3 + i] = fsign(cvs[i]) * sqrtf(fabsf(cvs[i]) * p->inv_thrust_k) * 60.f;  // u0..u3 [rpm]
    }
}

// forces + solve + kinematics (3DoF.py:182-296) for given limited thruster forces
template <bool FLOW, class PP>
__device__ __forceinline__ void dynamics3(PP p, const float* y, float c, float sn, const float* F, float2 cur,
                                          float* dy) {
    p = launder(p);
    const float u = y[3], v = y[4], r = y[5];
    float uRel = u, vRel = v;
    if (FLOW) {  // pinv(J) = J^T (3DoF.py:186-188)
        uRel -= c * cur.x + sn * cur.y;
        vRel -= -sn * cur.x + c * cur.y;
    }
    const float vr[3] = {uRel, vRel, r};
    const float av[3] = {fabsf(uRel), fabsf(vRel), fabsf(r)};
    // thruster jet-drag augment (3DoF.py:114-126)
    float au = fabsf(u);
    float drag = -p->jet_drag_k * au * u;
    float Xsum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; i++) {
#if MVRL_F64
        float uJet = sqrtf(fabsf(F[i]) * p->inv_jet_area_k);
        float q = au / fmaxf(1e-5f, uJet);
#else
        // |u| / max(1e-5, uJet) = |u| * min(1e5, 1 / sqrt(|F| / k)): one v_rsq instead of v_sqrt + v_rcp (rsq(0) = inf -> 1e5)
        float q = au * fminf(1e5f, __builtin_amdgcn_rsqf(fabsf(F[i]) * p->inv_jet_area_k));
#endif
#if MVRL_F64 || defined(MVRL_LIB_EXP)
        float dCd = p->jet_c1 * expf(-p->jet_k1 * q) + p->jet_c2 * expf(-p->jet_k2 * q);
#else
        // exp(-k q) = exp2(q * (-k log2 e)): ONE multiply (the constant product folds in the baked flavour) and v_exp_f32.  The library
        // expf wraps the same v_exp_f32 in six more instructions per call (compare, two selects, a scaled second multiply ...) so that
        // results below 2^-126 come out as denormals; here they are flushed (|error| < 1.2e-38 on a factor of order one) - and a lone
        // wave per SIMD (C2) pays 2.2 ns for every instruction: 48 fewer per RK stage, -22 % of the 3-DoF stage.
        float dCd = p->jet_c1 * __builtin_amdgcn_exp2f(q * (-1.44269504088896341f * p->jet_k1))
                  + p->jet_c2 * __builtin_amdgcn_exp2f(q * (-1.44269504088896341f * p->jet_k2));
#endif
        Xsum += dCd * drag;
    }
    float H[3];
    H[0] = Xsum + (F[0] + F[1] - F[2] - F[3]) * p->cos_a;  // 3DoF.py:255-263
    H[1] = (F[0] - F[1] + F[2] - F[3]) * p->sin_a;
    H[2] = p->yaw_arm * (F[0] + F[1] + F[2] + F[3]);
    const float m = p->m;
    float ka = m * (p->cgx * r + v), kb = m * (p->cgy * r - u);
    float c1[3] = {-ka * r, -kb * r, ka * u + kb * v};                      // Crb . vel   (3DoF.py:210-214)
    float ca[3] = {p->yvd * vRel * r, -p->xud * uRel * r, -p->yvd * vRel * uRel + p->xud * uRel * vRel};  // Ca . velRel
    float R[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float d = 0.f;
#pragma unroll
        for (int j = 0; j < 3; j++) d = fmaf(fmaf(p->dquad[3 * i + j], av[j], p->dlin[3 * i + j]), vr[j], d);
        R[i] = -c1[i] - (ca[i] + d) + H[i];
    }
#pragma unroll
    for (int i = 0; i < 3; i++) dy[3 + i] = fmaf(p->minv[3 * i + 2], R[2], fmaf(p->minv[3 * i + 1], R[1], p->minv[3 * i] * R[0]));
    dy[0] = c * u - sn * v;
    dy[1] = sn * u + c * v;
    dy[2] = r;
}

struct Trig1 {
    float s, c;
};
__device__ __forceinline__ Trig1 trig1(float a) {
    Trig1 t;
    sincos_f32(a, t.s, t.c);
    return t;
}
// sin / cos of the heading at an RK stage that differs from a known one by the small increment d (stage_trig in mvrl_rov6.hip):
// rotation by short Taylor polynomials; a lane with |d| > 0.25 (and the fp64 build) evaluates sp_psi - z_psi in full.
__device__ __forceinline__ Trig1 stage_trig1(const Trig1& b, float d, float sp_psi, float z_psi) {
#if defined(MVRL_FULL_STAGE_TRIG)
    return trig1(sp_psi - z_psi);
#else
    const float r2 = d * d;
#if MVRL_F64
    // fp64: Taylor to d^11 / d^12 - truncation 2.4e-18 / 4e-20 at |d| = 0.25 (stage_trig in mvrl_rov6.hip)
    const float ps = fmaf(fmaf(fmaf(fmaf(-2.5052108385441720e-8f, r2, 2.7557319223985893e-6f), r2, -1.9841269841269841e-4f), r2, 8.3333333333333332e-3f), r2, -1.6666666666666666e-1f);
    const float sd = fmaf(ps * r2, d, d);
    const float pc = fmaf(fmaf(fmaf(fmaf(2.0876756987868099e-9f, r2, -2.7557319223985888e-7f), r2, 2.4801587301587302e-5f), r2, -1.3888888888888889e-3f), r2, 4.1666666666666664e-2f);
    const float cd = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
#else
    const float ps = fmaf(8.333333333e-3f, r2, -1.666666667e-1f);
    const float sd = fmaf(ps * r2, d, d);
    const float pc = fmaf(-1.388888889e-3f, r2, 4.166666667e-2f);
    const float cd = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
#endif
    Trig1 t;
    t.s = fmaf(b.c, sd, b.s * cd);
    t.c = fmaf(-b.s, sd, b.c * cd);
#ifndef MVRL_NO_TRIG_VOTE
    if (__builtin_expect(__any(fabsf(d) > 0.25f) != 0, 0)) {   // the common case falls through one not-taken scalar branch
        if (fabsf(d) > 0.25f) t = trig1(sp_psi - z_psi);
    }
#else
    if (fabsf(d) > 0.25f) t = trig1(sp_psi - z_psi);
#endif
    return t;
#endif
}

// y: [error coordinates of the pose (3) | body velocities (3)]; t: sin / cos of the stage's heading
template <bool FLOW, bool HAS_DT, class PP>
__device__ __forceinline__ void derivs3(PP p, const float* y, const Trig1& t, Pid3& pid, float dtp, float inv_den,
                                        const float* dpose, bool inc_valid, float2 cur, float* dy, const AuxRow3& aux, bool fixed,
                                        const float* e0) {
    float F[4];
    control3<HAS_DT, true>(p, y, pid, dtp, inv_den, dpose, inc_valid, t.c, t.s, F, aux, fixed, e0);
    dynamics3<FLOW>(p, y, t.c, t.s, F, cur, dy);
}

// e_psi = angleError(setPoint[2], heading) (3DoF.py:407): the step kernel has it from its error coordinates
template <class PP>
__device__ __forceinline__ void observe3e(PP p, const float* y, const float* path, float e_psi, float* o) {
    o[0] = clampf((path[0] - y[0]) * p->inv_obs_pos, -1.f, 1.f);  // 3DoF.py:397-409
    o[1] = clampf((path[1] - y[1]) * p->inv_obs_pos, -1.f, 1.f);
    o[2] = clampf((path[2] - y[0]) * p->inv_obs_pos, -1.f, 1.f);
    o[3] = clampf((path[3] - y[1]) * p->inv_obs_pos, -1.f, 1.f);
    o[4] = clampf(e_psi * p->inv_obs_ang, -1.f, 1.f);
}
template <class PP>
__device__ __forceinline__ void observe3(PP p, const float* y, const float* path, const float* sp, float* o) {
    observe3e(p, y, path, angle_error(sp[2], y[2]), o);
}

// 3DoF.py:423-424: path = (rand(4).reshape(2,2) - 0.5) * 10, heading = rand() * 2 pi
__device__ __forceinline__ void random_init3(uint64_t seed, int64_t gid, uint32_t epoch, float t_quarter, float* path,
                                             float& heading, float& toffset) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t g0 = (uint32_t)gid, g1 = (uint32_t)((uint64_t)gid >> 32);
    Philox4 r0 = philox4x32_10(g0, g1, epoch, 0u, k0, k1);
    Philox4 r1 = philox4x32_10(g0, g1, epoch, 1u, k0, k1);
#pragma unroll
    for (int q = 0; q < 4; q++) path[q] = (u01(r0.v[q]) - 0.5f) * 10.f;
    heading = u01(r1.v[0]) * MVRL_TWO_PI_HI;
    toffset = u01(r1.v[1]) * t_quarter;
}

enum { R3_Y = 0, R3_EOLD = 6, R3_EINT = 9, R3_SP = 12, R3_PATH = 15, R3_EPISODE = 19 /* see mvrl_rov6.hip */, R3_TOLD = 20, R3_TIME = 21,
       R3_TOFF = 22, R3_ISTEP = 23, R3_WORDS = 24 };

// BlueROV2Heavy3DoF.derivs (3DoF.py:128-296) with run-time t - tOld: the RHS functor of the adaptive solver and of
// rov3_derivs_kernel
template <bool FLOW, class PP>
struct Rhs3 {
    PP p;
    const float* sp;
    Pid3* pid;
    float* told;
    float2 cur;
    float* aux_row;
    __device__ void operator()(float t, const float* y, float* dy) {
        float sn, c;
        sincos_f32(y[2], sn, c);
        float e[3] = {sp[0] - y[0], sp[1] - y[1], angle_error(sp[2], y[2])};
        const float dtp = t - *told;
        const float den = fmaxf(1e-9f, dtp);
        float u[3];
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float dedt = (e[i] - pid->eold[i]) / den;
            pid->eint[i] += 0.5f * (pid->eold[i] + e[i]) * dtp;
            pid->eint[i] = (fabsf(e[i]) > p->windup[i]) ? 0.f : pid->eint[i];
            const float v = p->kp[i] * e[i] + p->kd[i] * dedt + p->ki[i] * pid->eint[i];
            u[i] = clampf(v, -p->umax[i], p->umax[i]);
            pid->eold[i] = e[i];
        }
        *told = t;
        const float Xd = u[0] * c + u[1] * sn, Yd = -u[0] * sn + u[1] * c, Nd = u[2];
        if (aux_row) { aux_row[0] = Xd; aux_row[1] = Yd; aux_row[2] = Nd; }
        float F[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float cv = p->Ainv[3 * i] * Xd + p->Ainv[3 * i + 1] * Yd + p->Ainv[3 * i + 2] * Nd;
            const float f = clampf(cv, -p->f_max, p->f_max);
            F[i] = (fabsf(f) < p->f_dead) ? 0.f : f;
            if (aux_row) aux_row[3 + i] = fsign(cv) * sqrtf(fabsf(cv) * p->inv_thrust_k) * 60.f;
        }
        dynamics3<FLOW>(p, y, c, sn, F, cur, dy);
    }
};


// MULTI: io.k_steps consecutive env steps per launch (mvrl_rollout_dev), see rov6_step_kernel
template <class PP, bool ZOH, bool FLOW, int INTEG, bool MULTI = false>
__global__ MVRL_STEP_BOUNDS void rov3_step_kernel(const Rov3Dev* __restrict__ pg, const StepIO io, const FlowDev fl) {
    const PP p = param_ptr<PP>(pg);
    const uint32_t i_in = (uint32_t)io.lane0 + blockIdx.x * MVRL_STEP_BLOCK + threadIdx.x;
    if (i_in >= (uint32_t)io.lane_end) return;
    const uint32_t n32 = (uint32_t)io.n;  // see mvrl_rov6.hip: 32-bit byte offsets -> saddr addressing
    char* const stb = reinterpret_cast<char*>(io.state);
#define ST(k) (*reinterpret_cast<float*>(stb + ((uint32_t)(k) * (n32 * (uint32_t)sizeof(float)) + LANE * (uint32_t)sizeof(float))))
#define LANE i_k
    const int k_steps = MULTI ? io.k_steps : 1;
    // state: loaded before the first step of a launch, stored after the last, in registers in between (mvrl_rov6.hip)
    float y[6], sp[3], path[4];
    Pid3 pid;
    int istep = 0;
    float toff = 0.f;
#pragma nounroll
    for (int kstep = 0; kstep < k_steps; kstep++) {
    const float* const actions_k = (MULTI && io.actions) ? io.actions + (size_t)kstep * (size_t)io.n * 3 : io.actions;
    float* const obs_k = MULTI ? io.obs + (size_t)kstep * (size_t)io.n * 5 : io.obs;
    float* const reward_k = MULTI ? io.reward + (size_t)kstep * (size_t)io.n : io.reward;
    uint8_t* const done_k = MULTI ? io.done + (size_t)kstep * (size_t)io.n : io.done;
    uint32_t i_k = i_in;
    if (MULTI) asm volatile("" : "+v"(i_k));
    // load order as in mvrl_rov6.hip: what the turbulence gathers depend on first, everything else behind it,
    // branch-free up to the RK4 loop so that the gathers leave before the bulk of the state is waited on
    if (!MULTI || kstep == 0) {
        y[0] = ST(R3_Y + 0); y[1] = ST(R3_Y + 1);
        istep = unpack_int(ST(R3_ISTEP));
        if (FLOW) toff = ST(R3_TOFF);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 2; k < 6; k++) y[k] = ST(R3_Y + k);
#pragma unroll
        for (int k = 0; k < 3; k++) { pid.eold[k] = ST(R3_EOLD + k); pid.eint[k] = ST(R3_EINT + k); }
    }
    float spin[3];
    {
        const float* arow = io.fixed_sp ? nullptr : actions_k + (size_t)i_in * 3;
#pragma unroll
        for (int k = 0; k < 3; k++) spin[k] = io.fixed_sp ? ST(R3_SP + k) : arow[k];
    }
    const bool first = (istep == 0);
    istep += 1;
    FlowTap tap;
    if (FLOW) {   // the sample time in fp64 (flow_time_index)
        int kk;
        float ft;
        flow_time_index(fl, istep, io.dt64, toff, kk, ft);
        tap = flow_gather(fl, kk, ft, y[0], y[1]);
    }
    // error coordinates (see rov6_step_kernel): the RK4 loop integrates z = setPoint - pose, which starts the step at a * scale
    // (fixed set-point: z = pose_start - pose, starting at 0, and the controller adds e0 = setPoint - pose_start)
    const bool fixed = io.fixed_sp != 0;
    float z0[3], e0[3], org[3];
#if MVRL_BAM
    // the heading plane holds a binary angle (mvrl_device.hpp "binary angles"; y[2] carries its bit pattern, also from one step of a
    // fused launch to the next): an ordinary fp32 copy for the origin of the rare full-sincos lanes, the bits for everything else
    const uint32_t bpsi = (uint32_t)unpack_int(y[2]);
    float ys[2];
#endif
#pragma unroll
    for (int k = 0; k < 3; k++) {  // 3DoF.py:469-472
        const float da = spin[k] * p->act_scale[k];
#if MVRL_BAM
        const float yk = (k == 2) ? bam_to_rad(bpsi) : y[k];
#else
        const float yk = y[k];
#endif
        sp[k] = fixed ? spin[k] : da + yk;
        e0[k] = fixed ? spin[k] - yk : da;             // the error at the start of the step (yaw: unwrapped)
        org[k] = fixed ? yk : sp[k];                   // pose = org - z
        z0[k] = fixed ? 0.f : e0[k];
#if MVRL_BAM
        if (k < 2) ys[k] = yk;                         // start position: the step's displacement z_start - z_end is added to it at the end
#endif
    }
#if MVRL_BAM
    if (fixed) {   // the yaw error against a fixed set-point from the binary angle's hi + lo pair (wrapped by the controller)
        float hi, lo;
        bam_to_rad2(bpsi, hi, lo);
        e0[2] = (spin[2] - hi) - lo;
    }
#endif
    float2 cur = make_float2(0.f, 0.f);
    if (FLOW) cur = flow_combine(tap);
    if (first) { pid.eold[0] = e0[0]; pid.eold[1] = e0[1]; pid.eold[2] = angle_error(e0[2], 0.f); }
    const float ec[3] = {fixed ? e0[0] : 0.f, fixed ? e0[1] : 0.f, fixed ? e0[2] : 0.f};   // what the controller adds to z (see control3)

    const float h_s = io.dt / (float)io.n_sub;
    const float h = in_vgpr(h_s), hh = in_vgpr(0.5f * h_s), h6 = in_vgpr(h_s / 6.f), inv_hh = in_vgpr(1.0f / (0.5f * h_s));
    float* const aux_row = io.aux ? io.aux + (size_t)i_in * 7 : nullptr;
    float inc_prev[3] = {0.f, 0.f, 0.f};  // see mvrl_rov6.hip
#if MVRL_BAM
    float e_psi = 0.f;
    uint32_t bpsi_new = 0u;
#endif
#if MVRL_F64
    if (INTEG == 1) {  // the reference's own integrator (3DoF.py:475-477), see mvrl_rk45.hpp
        float told = ST(R3_TOLD), time = ST(R3_TIME);
        if (first) { told = 0.f; time = 0.f; }
        time += io.dt;
        Rhs3<FLOW, PP> rhs{p, sp, &pid, &told, cur, aux_row};
        int nfev = 0;
        rk45_solve<6>(rhs, time - io.dt, time, io.dt, 1e-3, 1e-3, y, &nfev);
        ST(R3_TOLD) = told;
        ST(R3_TIME) = time;
        if (io.nfev) io.nfev[i_in] = nfev;
    } else
#endif
    {
#if MVRL_BAM
    Trig1 tb;                 // heading at the start of the step: the one full sincos of the step, of the binary angle
    sincos_bam(bpsi, tb.s, tb.c);
#else
    Trig1 tb = trig1(y[2]);   // heading at the start of the step: the one full sincos of the step
#endif
#pragma unroll
    for (int q = 0; q < 3; q++) y[q] = z0[q];          // from here to the end of the loop y[0..2] is the ERROR setPoint - pose
#define MVRL_AX3(c_, q_) ((q_) < 3 ? -(c_) : (c_))
    for (int ks = 0; ks < io.n_sub; ks++) {
        float k[6], acc[6], yt[6];
        const AuxRow3 aux_last{io.aux != nullptr && ks == io.n_sub - 1, aux_row}, aux_none{false, nullptr};
#if MVRL_BAM
        // ZOH: every sub-step; FAITHFUL: re-anchored every fourth - at the binary start heading + what the step has turned so far
        if (ks > 0 && (ZOH || (ks & 3) == 0)) sincos_bam(bam_add(bpsi, z0[2] - y[2]), tb.s, tb.c);
#else
        if (ks > 0 && (ZOH || (ks & 3) == 0)) tb = trig1(org[2] - y[2]);   // ZOH: every sub-step; FAITHFUL: re-anchored every fourth
#endif
        if (ZOH) {
            float F[4];
            if (first && ks == 0) control3<false, false>(p, y, pid, 0.f, 1e9f, nullptr, false, tb.c, tb.s, F, aux_last, fixed, ec);
            else control3<true, true>(p, y, pid, h, 1.0f / h, inc_prev, ks > 0, tb.c, tb.s, F, aux_last, fixed, ec);
            dynamics3<FLOW>(p, y, tb.c, tb.s, F, cur, k);
#pragma unroll
            for (int q = 0; q < 6; q++) { acc[q] = k[q]; yt[q] = fmaf(MVRL_AX3(hh, q), k[q], y[q]); }
            Trig1 t = stage_trig1(tb, hh * k[2], org[2], yt[2]);
            dynamics3<FLOW>(p, yt, t.c, t.s, F, cur, k);
#pragma unroll
            for (int q = 0; q < 6; q++) { acc[q] = fmaf(2.f, k[q], acc[q]); yt[q] = fmaf(MVRL_AX3(hh, q), k[q], y[q]); }
            t = stage_trig1(tb, hh * k[2], org[2], yt[2]);
            dynamics3<FLOW>(p, yt, t.c, t.s, F, cur, k);
#pragma unroll
            for (int q = 0; q < 6; q++) { acc[q] = fmaf(2.f, k[q], acc[q]); yt[q] = fmaf(MVRL_AX3(h, q), k[q], y[q]); }
            t = stage_trig1(tb, h * k[2], org[2], yt[2]);
            dynamics3<FLOW>(p, yt, t.c, t.s, F, cur, k);
#pragma unroll
            for (int q = 0; q < 3; q++) inc_prev[q] = h6 * (acc[q] + k[q]);
        } else {
            float dp[3], d2[3], d3[3];
#pragma unroll
            for (int q = 0; q < 3; q++) dp[q] = inc_prev[q];
            derivs3<FLOW, false>(p, y, tb, pid, 0.f, 1e9f, dp, ks > 0, cur, k, aux_none, fixed, ec);
#pragma unroll
            for (int q = 0; q < 6; q++) { acc[q] = k[q]; yt[q] = fmaf(MVRL_AX3(hh, q), k[q], y[q]); }
#pragma unroll
            for (int q = 0; q < 3; q++) dp[q] = hh * k[q];
            derivs3<FLOW, true>(p, yt, stage_trig1(tb, dp[2], org[2], yt[2]), pid, hh, inv_hh, dp, true, cur, k, aux_none, fixed, ec);
#pragma unroll
            for (int q = 0; q < 3; q++) { dp[q] = hh * (k[q] - acc[q]); d2[q] = hh * k[q]; }
#pragma unroll
            for (int q = 0; q < 6; q++) { acc[q] = fmaf(2.f, k[q], acc[q]); yt[q] = fmaf(MVRL_AX3(hh, q), k[q], y[q]); }
            derivs3<FLOW, false>(p, yt, stage_trig1(tb, d2[2], org[2], yt[2]), pid, 0.f, 1e9f, dp, true, cur, k, aux_none, fixed, ec);
#pragma unroll
            for (int q = 0; q < 3; q++) { d3[q] = h * k[q]; dp[q] = d3[q] - d2[q]; }
#pragma unroll
            for (int q = 0; q < 6; q++) { acc[q] = fmaf(2.f, k[q], acc[q]); yt[q] = fmaf(MVRL_AX3(h, q), k[q], y[q]); }
            derivs3<FLOW, true>(p, yt, stage_trig1(tb, d3[2], org[2], yt[2]), pid, hh, inv_hh, dp, true, cur, k, aux_last, fixed, ec);
#pragma unroll
            for (int q = 0; q < 3; q++) inc_prev[q] = h6 * (acc[q] + k[q]) - d3[q];
        }
        const float dpsi = h6 * (acc[2] + k[2]);      // the sub-step's own heading increment
#pragma unroll
        for (int q = 0; q < 6; q++) y[q] = fmaf(MVRL_AX3(h6, q), acc[q] + k[q], y[q]);
        if (!ZOH && ((ks + 1) & 3) != 0 && ks + 1 < io.n_sub) tb = stage_trig1(tb, dpsi, org[2], y[2]);   // base of the next sub-step
    }
#undef MVRL_AX3
#if MVRL_BAM
    // the heading leaves the loop as the error z_end; the step moved it by z_start - z_end, which is ADDED to the binary start angle
    // (the wrap of 3DoF.py:480 is the integer overflow); the observation's heading error is z_end itself (+ E0 with a fixed set-point)
    e_psi = angle_error((fixed ? e0[2] : 0.f) + y[2], 0.f);
    if (!fixed) sp[2] = z0[2] + bam_to_rad_pos(bpsi);                  // 3DoF.py:469-472 with the heading in [0, 2 pi)
    bpsi_new = bam_add(bpsi, z0[2] - y[2]);
#pragma unroll
    for (int q = 0; q < 2; q++) y[q] = ys[q] + (z0[q] - y[q]);   // back to the position: one rounding at the size of the position
#else
#pragma unroll
    for (int q = 0; q < 3; q++) y[q] = org[q] - y[q];   // back to the pose
#endif
    }
#if !MVRL_BAM
    y[2] = mod_two_pi(y[2]);  // 3DoF.py:480
#endif
    // The epilogue addresses the same SoA planes as the prologue.  Left alone, LLVM keeps all ~40 prologue
    // addresses alive in VGPR pairs across the whole RK4 loop (~75 registers, the difference between 2 and 3 waves
    // per SIMD) instead of recomputing them; hiding the lane index behind an empty asm makes it recompute.
    uint32_t i = i_in;
    asm volatile("" : "+v"(i));
#undef LANE
#define LANE i
#pragma unroll
    for (int k = 0; k < 4; k++) path[k] = ST(R3_PATH + k);
    float o[5];
#if MVRL_BAM
    observe3e(p, y, path, e_psi, o);
    y[2] = pack_int((int)bpsi_new);      // the heading word is a bit pattern again
#else
    observe3(p, y, path, sp, o);
#endif
    const bool done = istep >= io.max_steps;
    reward_k[i] = 0.f;
    done_k[i] = done ? 3 : 0;  // bit 0 = done, bit 1 = time limit
    if (done && io.auto_reset) {
        if (io.term_obs) {
#pragma unroll
            for (int q = 0; q < 5; q++) io.term_obs[(size_t)i * 5 + q] = o[q];
        }
        const int episode = unpack_int(ST(R3_EPISODE)) + 1;
        ST(R3_EPISODE) = pack_int(episode);
        if (!io.fixed_sp) {
            float heading;
            random_init3(io.seed, io.env_offset + (int64_t)i, (uint32_t)episode, fl.t_quarter, path, heading, toff);
#pragma unroll
            for (int q = 0; q < 4; q++) ST(R3_PATH + q) = path[q];
            ST(R3_TOFF) = toff;
            sp[0] = path[0]; sp[1] = path[1]; sp[2] = heading;
        }
#pragma unroll
        for (int q = 0; q < 6; q++) y[q] = 0.f;
#pragma unroll
        for (int q = 0; q < 3; q++) { pid.eold[q] = 0.f; pid.eint[q] = 0.f; }
        istep = 0;
        observe3(p, y, path, sp, o);
    }
#pragma unroll
    for (int q = 0; q < 5; q++) obs_k[(size_t)i * 5 + q] = o[q];
    if (!MULTI || kstep == k_steps - 1) {
#pragma unroll
        for (int k = 0; k < 6; k++) ST(R3_Y + k) = y[k];
#pragma unroll
        for (int k = 0; k < 3; k++) { ST(R3_EOLD + k) = pid.eold[k]; ST(R3_EINT + k) = pid.eint[k]; }
        if (!io.fixed_sp) {
#pragma unroll
            for (int k = 0; k < 3; k++) ST(R3_SP + k) = sp[k];
        }
        ST(R3_ISTEP) = pack_int(istep);
    }
    }  // kstep
#undef ST
#undef LANE
}

__global__ __launch_bounds__(MVRL_BLOCK) void rov3_reset_kernel(const Rov3Dev* __restrict__ pg, float* state, int64_t n, const uint8_t* mask,
                                                                const float* init, float* obs, uint64_t seed,
                                                                int64_t env_offset, float t_quarter) {
    const int64_t i = (int64_t)blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    const CP3 p = as_const(pg);
    float* st = state + i;
    const int episode = unpack_int(st[R3_EPISODE * n]) + 1;
    st[R3_EPISODE * n] = pack_int(episode);
    float path[4], sp[3], y[6] = {0, 0, 0, 0, 0, 0}, toff = 0.f;
    if (init) {
#pragma unroll
        for (int q = 0; q < 4; q++) path[q] = init[i * 5 + q];
        sp[2] = init[i * 5 + 4];
    } else {
        random_init3(seed, env_offset + i, (uint32_t)episode, t_quarter, path, sp[2], toff);
    }
    sp[0] = path[0]; sp[1] = path[1];
#pragma unroll
    for (int q = 0; q < 6; q++) st[(R3_Y + q) * n] = 0.f;
#pragma unroll
    for (int q = 0; q < 3; q++) { st[(R3_EOLD + q) * n] = 0.f; st[(R3_EINT + q) * n] = 0.f; st[(R3_SP + q) * n] = sp[q]; }
#pragma unroll
    for (int q = 0; q < 4; q++) st[(R3_PATH + q) * n] = path[q];
    st[R3_TOLD * n] = 0.f;
    st[R3_TIME * n] = 0.f;
    st[R3_TOFF * n] = toff;
    st[R3_ISTEP * n] = pack_int(0);
    if (obs) {
        float o[5];
        observe3(p, y, path, sp, o);
#pragma unroll
        for (int q = 0; q < 5; q++) obs[i * 5 + q] = o[q];
    }
}

// dataToState(systemState) of every env's current state (3DoF.py:397-409) through observe3 (mvrl_observe); nothing is modified.
__global__ __launch_bounds__(MVRL_BLOCK) void rov3_observe_kernel(const Rov3Dev* __restrict__ pg, const float* state, int64_t n, float* obs) {
    const int64_t i = (int64_t)blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const CP3 p = as_const(pg);
    const float* st = state + i;
    float y[6], path[4], sp[3], o[5];
#pragma unroll
    for (int q = 0; q < 6; q++) y[q] = st[(R3_Y + q) * n];
#if MVRL_BAM
    y[2] = bam_to_rad_pos((uint32_t)unpack_int(y[2]));
#endif
#pragma unroll
    for (int q = 0; q < 4; q++) path[q] = st[(R3_PATH + q) * n];
#pragma unroll
    for (int q = 0; q < 3; q++) sp[q] = st[(R3_SP + q) * n];
    observe3(p, y, path, sp, o);
#pragma unroll
    for (int q = 0; q < 5; q++) obs[i * 5 + q] = o[q];
}

hipError_t launch_rov3_observe(const Rov3Dev* p, const float* state, int64_t n, float* obs, hipStream_t stream) {
    dim3 grid((unsigned)((n + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    hipLaunchKernelGGL(rov3_observe_kernel, grid, block, 0, stream, p, state, n, obs);
    return hipGetLastError();
}

// One evaluation of vehicle.derivs(t, y) for n independent tuples (see rov6_derivs_kernel)
template <class PP, bool FLOW>
__global__ __launch_bounds__(MVRL_STEP_BLOCK) void rov3_derivs_kernel(const Rov3Dev* __restrict__ pg, int64_t n, const float* t,
                                                                      const float* y_in, const float* sp_in, const float* cur_in, float* eold,
                                                                      float* eint, float* told, const uint8_t* has_old,
                                                                      float* dy_out, float* aux_out) {
    const PP p = param_ptr<PP>(pg);
    const int64_t i = (int64_t)blockIdx.x * MVRL_STEP_BLOCK + threadIdx.x;
    if (i >= n) return;
    float y[6], sp[3], dy[6];
    Pid3 pid;
#pragma unroll
    for (int k = 0; k < 6; k++) y[k] = y_in[i * 6 + k];
#pragma unroll
    for (int k = 0; k < 3; k++) { sp[k] = sp_in[i * 3 + k]; pid.eold[k] = eold[i * 3 + k]; pid.eint[k] = eint[i * 3 + k]; }
    if (!has_old[i]) {  // eOld is None (3DoF.py:144-145)
        pid.eold[0] = sp[0] - y[0]; pid.eold[1] = sp[1] - y[1]; pid.eold[2] = angle_error(sp[2], y[2]);
    }
    float to = told[i];
    // cur_in (FLOW; mvrl_derivs_cur): global-frame current per tuple - the velCurrent of 3DoF.py:182-191 (golden G21)
    const float2 cur = FLOW ? make_float2(cur_in[i * 2], cur_in[i * 2 + 1]) : make_float2(0.f, 0.f);
    Rhs3<FLOW, PP> rhs{p, sp, &pid, &to, cur, aux_out + i * 7};
    rhs(t[i], y, dy);
#pragma unroll
    for (int k = 0; k < 6; k++) dy_out[i * 6 + k] = dy[k];
#pragma unroll
    for (int k = 0; k < 3; k++) { eold[i * 3 + k] = pid.eold[k]; eint[i * 3 + k] = pid.eint[k]; }
    told[i] = to;
}

hipError_t launch_rov3_derivs(const Rov3Dev* p, bool baked, int64_t n, const float* t, const float* y, const float* sp, const float* cur,
                              float* eold, float* eint, float* told, const uint8_t* has_old, float* dy, float* aux, hipStream_t stream) {
    dim3 grid((unsigned)((n + MVRL_STEP_BLOCK - 1) / MVRL_STEP_BLOCK)), block(MVRL_STEP_BLOCK);
#define MVRL_D3(PPT, F) hipLaunchKernelGGL((rov3_derivs_kernel<PPT, F>), grid, block, 0, stream, p, n, t, y, sp, cur, eold, eint, told, has_old, dy, aux)
    if (cur) { if (baked) MVRL_D3(const Rov3Baked*, true); else MVRL_D3(CP3, true); }
    else { if (baked) MVRL_D3(const Rov3Baked*, false); else MVRL_D3(CP3, false); }
#undef MVRL_D3
    return hipGetLastError();
}

// (Capping the 75-89-VGPR 3-DoF kernels at four waves per SIMD with a dynamic-LDS allocation was measured: 94.0 vs 93.1 us
// per step at 1 048 576 envs, 65.9 vs 63.7 us with ZOH - no gain, not adopted; gpurun_out/r2_ab14.log.)
hipError_t launch_rov3_step(const Rov3Dev* p, const StepIO& io, const FlowDev& fl, bool baked, bool zoh, bool flow,
                            bool rk45, hipStream_t stream) {
    dim3 grid((unsigned)((io.lane_end - io.lane0 + MVRL_STEP_BLOCK - 1) / MVRL_STEP_BLOCK)), block(MVRL_STEP_BLOCK);
#define MVRL_L3(PPT, Z, F) hipLaunchKernelGGL((rov3_step_kernel<PPT, Z, F, 0>), grid, block, 0, stream, p, io, fl)
#if MVRL_F64
    if (rk45) {
        if (flow) hipLaunchKernelGGL((rov3_step_kernel<CP3, false, true, 1>), grid, block, 0, stream, p, io, fl);
        else hipLaunchKernelGGL((rov3_step_kernel<CP3, false, false, 1>), grid, block, 0, stream, p, io, fl);
        return hipGetLastError();
    }
#endif
    // one instance for single steps and fused multi-step launches (see launch_rov6_step): same binary, same roundings
#if !MVRL_F64 && !defined(MVRL_SEPARATE_SINGLE)
    {
#else
    if (io.k_steps > 1) {
#endif
#define MVRL_L3M(PPT, Z, F) hipLaunchKernelGGL((rov3_step_kernel<PPT, Z, F, 0, true>), grid, block, 0, stream, p, io, fl)
        if (baked) {
            if (zoh) { if (flow) MVRL_L3M(const Rov3Baked*, true, true); else MVRL_L3M(const Rov3Baked*, true, false); }
            else { if (flow) MVRL_L3M(const Rov3Baked*, false, true); else MVRL_L3M(const Rov3Baked*, false, false); }
        } else {
            if (zoh) { if (flow) MVRL_L3M(CP3, true, true); else MVRL_L3M(CP3, true, false); }
            else { if (flow) MVRL_L3M(CP3, false, true); else MVRL_L3M(CP3, false, false); }
        }
#undef MVRL_L3M
        return hipGetLastError();
    }
#if MVRL_F64 || defined(MVRL_SEPARATE_SINGLE)   /* fp32: everything was dispatched above */
    if (baked) {
        if (zoh) { if (flow) MVRL_L3(const Rov3Baked*, true, true); else MVRL_L3(const Rov3Baked*, true, false); }
        else { if (flow) MVRL_L3(const Rov3Baked*, false, true); else MVRL_L3(const Rov3Baked*, false, false); }
    } else {
        if (zoh) { if (flow) MVRL_L3(CP3, true, true); else MVRL_L3(CP3, true, false); }
        else { if (flow) MVRL_L3(CP3, false, true); else MVRL_L3(CP3, false, false); }
    }
#endif
#undef MVRL_L3
    return hipGetLastError();
}

hipError_t launch_rov3_reset(const Rov3Dev* p, float* state, int64_t n, const uint8_t* mask, const float* init, float* obs,
                             uint64_t seed, int64_t env_offset, float t_quarter, hipStream_t stream) {
    dim3 grid((unsigned)((n + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    hipLaunchKernelGGL(rov3_reset_kernel, grid, block, 0, stream, p, state, n, mask, init, obs, seed, env_offset,
                       t_quarter);
    return hipGetLastError();
}

}  // namespace mvrl
