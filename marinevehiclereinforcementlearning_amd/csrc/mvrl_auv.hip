// mvrl_auv.hip - simplified 3-DoF AUV environment (explicit Euler + turbulence current + shaped reward).
//
// Replaces AuvEnv.step / reset / dataToState (tag_00_Dec2023_simpleControlTurbulence/verySimpleAuv.py:147-410)
// including the flow lookup (flowGenerator.py:97-136) in one fused kernel per env step.  This one IS
// bandwidth-shaped: ~150 flop against ~390 B per env step (53 state words, most of them the 10-deep action
// ring the reward's smoothness term needs), so the layout work - SoA planes, one coalesced read + one
// coalesced write per word - is what matters here.
#include "mvrl_kernels.hpp"
#ifndef MVRL_AUV_LDS_OBS
#define MVRL_AUV_LDS_OBS 1
#endif

namespace mvrl {

enum {
    AV_X = 0, AV_Y = 1, AV_PSI = 2, AV_VX = 3, AV_VY = 4, AV_R = 5, AV_TGT = 6, AV_HERR_O = 7, AV_PERR_O = 8,
    AV_MULT = 10,   // m I Xuu Yvv Nrr Xu Yv Nr Xact Yact Nact  (verySimpleAuv.py:222-229)
    AV_TOFF = 21, AV_HIST = 22, AV_ISTEP = 52,
    AV_IWP = 53,    // AuvEnvCyl: way-point index (integer bit pattern); survives reset() like the reference's self.iWp
    AV_EPISODE = 54,  // number of resets of this env = counter of its Philox stream (see mvrl_rov6.hip)
    // Origin of the action ring: the action of step `istep` goes to slot (istep - 1 + phase) % 10.  A new episode starts
    // where the old one stopped (phase' = next slot) instead of at slot 0, so the lanes of a wave - whose episodes end at
    // different times - keep writing the SAME ring plane every step: one coalesced 256-B store instead of 64 scattered
    // 4-byte ones, each of which costs a 32-B HBM sector (measured: 65 B/env of write traffic for 12 B of data).
    // recentActions (verySimpleAuv.py:275, :353-355) is only ever reduced to per-component mean / std: order-free.
    AV_PHASE = 55,
    AV_WORDS = 56
};

// dataToState "V3" (verySimpleAuv.py:201-212); positionTarget = 0 (:241)
// obs_scale: all ones except 1/(45 deg) on the heading error for AuvEnv's "V3"; AuvEnvCyl's "V0" scaling otherwise
// (tag/verySimpleAuv_cyl.py:100-111)
__device__ __forceinline__ void observe_auv(const AuvDev& p, float x, float y, float psi, float vx, float vy, float r,
                                            float tx, float ty, float tgt, float herr_o, float perr_ox, float perr_oy,
                                            float* o) {
    float perr0 = tx - x, perr1 = ty - y;
    float herr = angle_error(tgt, psi);
    o[0] = clampf(perr0 * p.obs_scale[0], -1.f, 1.f);
    o[1] = clampf(perr1 * p.obs_scale[1], -1.f, 1.f);
    o[2] = clampf(herr * p.obs_scale[2], -1.f, 1.f);
    o[3] = clampf((herr - herr_o) * p.obs_scale[3], -1.f, 1.f);
    o[4] = clampf((perr0 - perr_ox) * p.obs_scale[4], -1.f, 1.f);
    o[5] = clampf((perr1 - perr_oy) * p.obs_scale[5], -1.f, 1.f);
    o[6] = clampf(vx * p.obs_scale[6], -1.f, 1.f);
    o[7] = clampf(vy * p.obs_scale[7], -1.f, 1.f);
    o[8] = clampf(r * p.obs_scale[8], -1.f, 1.f);
    o[9] = 0.f;
    o[10] = 0.f;
}

// positionTarget / headingTarget of way-point k (verySimpleAuv_cyl.py:141-142)
__device__ __forceinline__ void waypoint(const AuvDev& p, int k, float& tx, float& ty, float& th) {
    tx = p.wp[3 * k]; ty = p.wp[3 * k + 1]; th = p.wp[3 * k + 2];
}

// reset draws (verySimpleAuv.py:222-245), order: 8 coefficient multipliers, 3 actuation multipliers,
// position (2), heading, headingTarget, flow time offset
__device__ __forceinline__ void random_init_auv(const AuvDev& p, uint64_t seed, int64_t gid, uint32_t epoch, float t_quarter,
                                                float* v /*16: x y psi tgt toff mult[11]*/) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t g0 = (uint32_t)gid, g1 = (uint32_t)((uint64_t)gid >> 32);
    float u[16];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        Philox4 r = philox4x32_10(g0, g1, epoch, (uint32_t)b, k0, k1);
#pragma unroll
        for (int q = 0; q < 4; q++) u[4 * b + q] = u01(r.v[q]);
    }
#pragma unroll
    for (int q = 0; q < 8; q++) v[5 + q] = 1.f + 0.5f * p.noise_coeffs - u[q] * p.noise_coeffs;
#pragma unroll
    for (int q = 0; q < 3; q++) v[13 + q] = 1.f + 0.5f * p.noise_act - u[8 + q] * p.noise_act;
    v[0] = (u[11] - 0.5f) * 0.5f * (p.x_max - p.x_min);
    v[1] = (u[12] - 0.5f) * 0.5f * (p.y_max - p.y_min);
    v[2] = u[13] * MVRL_TWO_PI_HI;
    v[3] = u[14] * MVRL_TWO_PI_HI;
    v[4] = u[15] * t_quarter;
}

// Everything one AuvEnv lane carries between steps, in registers
struct AuvLane {
    float x, y, psi, vx, vy, r;
    float tgt, tx, ty;              // headingTarget, positionTarget (0 for AuvEnv, waypoints[iWp] for AuvEnvCyl)
    float herr_o, perr_ox, perr_oy;
    float mu[11];
    float hist[30];
    float toff;
    int istep, iwp, phase;
};
struct AuvStepOut {
    float o[11];
    float reward;
    bool done, time_up;
    int slot;
    float Fg0, Fg1, Fh2, curx, cury, rms, t0, t1, t2, t3, bonus;   // timeHistory side outputs (:389-403)
};

// AuvEnv.step (verySimpleAuv.py:264-410) / AuvEnvCyl.step for one lane: shared by the step kernel and the fused
// PD-episode kernel, so both run the same arithmetic.
template <bool FLOW>
__device__ __forceinline__ void auv_step_core(const AuvDev& p, const FlowDev& fl, AuvLane& s, float a0, float a1, float a2,
                                              float dt, double dt64, int max_steps, AuvStepOut& out) {
    const bool cyl = p.n_wp > 0;
    s.istep += 1;                                  // verySimpleAuv.py:266
    // time = iStep * dt (:267): only the turbulence lookup reads it, and forms it in fp64 from the step count (flow_time_index)
    const bool time_up = s.istep >= max_steps;     // :270-272
    bool done = time_up;
    const int slot = (s.istep - 1 + s.phase) % 10; // recentActions.appendleft (:275) as a ring starting at slot `phase`
    const int nh = s.istep < 10 ? s.istep : 10;
#pragma unroll
    for (int j = 0; j < 10; j++) {
        s.hist[3 * j + 0] = (j == slot) ? a0 : s.hist[3 * j + 0];
        s.hist[3 * j + 1] = (j == slot) ? a1 : s.hist[3 * j + 1];
        s.hist[3 * j + 2] = (j == slot) ? a2 : s.hist[3 * j + 2];
    }
    const float Fset0 = a0 * p.max_force * s.mu[8], Fset1 = a1 * p.max_force * s.mu[9];   // :278
    const float Nset = a2 * p.max_moment * s.mu[10];                                        // :279
    float sn, c;
    sincos_f32(s.psi, sn, c);
    float2 cur = make_float2(0.f, 0.f);
    if (FLOW) cur = flow_interp_uv(fl, s.istep, dt64, s.toff, s.x, s.y);                    // :291
    const float dvx = s.vx - cur.x, dvy = s.vy - cur.y;
    const float vr0 = c * dvx + sn * dvy, vr1 = -sn * dvx + c * dvy;                        // :298 (pinv(J) = J^T)
    const float Fh0 = (p.xu * s.mu[5] + p.xuu * s.mu[2] * fabsf(vr0)) * vr0;                // :303-307
    const float Fh1 = (p.yv * s.mu[6] + p.yvv * s.mu[3] * fabsf(vr1)) * vr1;
    const float Fh2 = (p.nr * s.mu[7] + p.nrr * s.mu[4] * fabsf(s.r)) * s.r;
    const float Fg0 = c * Fh0 - sn * Fh1, Fg1 = sn * Fh0 + c * Fh1;                         // :310
    const float acc0 = (Fg0 + Fset0) / (p.m * s.mu[0]);                                     // :314-318
    const float acc1 = (Fg1 + Fset1) / (p.m * s.mu[0]);
    const float acc2 = (Fh2 + Nset) / (p.izz * s.mu[1]);
    const float h = dt;                                                                     // :321-326 explicit Euler
    s.x = fmaf(s.vx, h, s.x); s.y = fmaf(s.vy, h, s.y);
    s.psi = mod_two_pi(fmaf(s.r, h, s.psi));
    s.vx = fmaf(acc0, h, s.vx); s.vy = fmaf(acc1, h, s.vy); s.r = fmaf(acc2, h, s.r);

    observe_auv(p, s.x, s.y, s.psi, s.vx, s.vy, s.r, s.tx, s.ty, s.tgt, s.herr_o, s.perr_ox, s.perr_oy, out.o);    // :329
    float bonus = 0.f;                                                                      // :335-342
    if (s.x < p.x_min || s.x > p.x_max) { if (p.stop_on_bounds) done = true; bonus += -100.f; }
    if (s.y < p.y_min || s.y > p.y_max) { if (p.stop_on_bounds) done = true; bonus += -100.f; }
    const float perr0 = s.tx - s.x, perr1 = s.ty - s.y;
    const float herr = angle_error(s.tgt, s.psi);
    if (cyl && sqrtf(perr0 * perr0 + perr1 * perr1) < p.wp_thr) {                           // _cyl.py:249-253
        s.iwp = min(p.n_wp - 1, s.iwp + 1);
        waypoint(p, s.iwp, s.tx, s.ty, s.tgt);
    }
    s.herr_o = herr; s.perr_ox = perr0; s.perr_oy = perr1;                                  // :349-350
    float rms = 0.f;                                                                        // :353-355
    const float inv_nh = 1.0f / (float)nh;
    bool valid[10];                                // the episode's last nh actions sit in slots phase .. phase + nh - 1 (mod 10)
#pragma unroll
    for (int j = 0; j < 10; j++) {
        const int d = j - s.phase;
        valid[j] = (d < 0 ? d + 10 : d) < nh;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float mean = 0.f;
#pragma unroll
        for (int j = 0; j < 10; j++) mean += valid[j] ? s.hist[3 * j + k] : 0.f;
        mean *= inv_nh;
        float ss = 0.f;
#pragma unroll
        for (int j = 0; j < 10; j++) { float d = s.hist[3 * j + k] - mean; ss += valid[j] ? d * d : 0.f; }
        rms += sqrtf(ss * inv_nh);
    }
    rms *= (1.0f / 3.0f);
    const float PI = 3.14159265358979323846f;
    const float hdeg = herr * (180.0f / PI);
    const float t0 = expf(-5.f * sqrtf(perr0 * perr0 + perr1 * perr1));                     // :357-381
    const float t1 = (fabsf(herr) < 0.5f * PI) ? expf(-0.1f * fabsf(hdeg)) : -expf(-0.1f * (180.f - fabsf(hdeg)));
    const float t2 = expf(-0.6f * rms);
    const float t3 = -0.1f * (a0 * a0 + a1 * a1 + a2 * a2) * (1.0f / 3.0f);
    out.reward = (((t0 + t1) + t2) + t3) + bonus;
    out.done = done; out.time_up = time_up; out.slot = slot;
    out.Fg0 = Fg0; out.Fg1 = Fg1; out.Fh2 = Fh2; out.curx = cur.x; out.cury = cur.y; out.rms = rms;
    out.t0 = t0; out.t1 = t1; out.t2 = t2; out.t3 = t3; out.bonus = bonus;
}

#define MVRL_AUV_LOAD_LANE(s)                                                                                         \
    do {                                                                                                              \
        (s).x = ST(AV_X); (s).y = ST(AV_Y); (s).psi = ST(AV_PSI); (s).vx = ST(AV_VX); (s).vy = ST(AV_VY); (s).r = ST(AV_R); \
        (s).tgt = ST(AV_TGT); (s).tx = 0.f; (s).ty = 0.f; (s).iwp = 0;  /* positionTarget = 0 for AuvEnv (:241) */        \
        if (p.n_wp > 0) { (s).iwp = unpack_int(ST(AV_IWP)); waypoint(p, (s).iwp, (s).tx, (s).ty, (s).tgt); }              \
        (s).herr_o = ST(AV_HERR_O); (s).perr_ox = ST(AV_PERR_O); (s).perr_oy = ST(AV_PERR_O + 1);                       \
        _Pragma("unroll") for (int q_ = 0; q_ < 11; q_++) (s).mu[q_] = ST(AV_MULT + q_);                               \
        _Pragma("unroll") for (int q_ = 0; q_ < 30; q_++) (s).hist[q_] = ST(AV_HIST + q_);                             \
        (s).istep = unpack_int(ST(AV_ISTEP));                                                                         \
        (s).phase = unpack_int(ST(AV_PHASE));                                                                         \
        (s).toff = FLOW ? ST(AV_TOFF) : 0.f;                                                                          \
    } while (0)

template <bool FLOW>
__global__ __launch_bounds__(MVRL_BLOCK) void auv_step_kernel(const AuvDev p, const StepIO io, const FlowDev fl) {
    const uint32_t i = (uint32_t)io.lane0 + blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= (uint32_t)io.lane_end) return;
    const uint32_t n32 = (uint32_t)io.n;
    char* const stb = reinterpret_cast<char*>(io.state);
#define ST(k) (*reinterpret_cast<float*>(stb + ((uint32_t)(k) * (n32 * (uint32_t)sizeof(float)) + i * (uint32_t)sizeof(float))))
    const bool cyl = p.n_wp > 0;
    AuvLane s;
    MVRL_AUV_LOAD_LANE(s);
    const float* ap = io.actions + (size_t)i * 3;
    const float a0 = ap[0], a1 = ap[1], a2 = ap[2];
    AuvStepOut out;
    auv_step_core<FLOW>(p, fl, s, a0, a1, a2, io.dt, io.dt64, io.max_steps, out);
    // the kernel body below keeps its historical local names
    float x = s.x, y = s.y, psi = s.psi, vx = s.vx, vy = s.vy, r = s.r, tgt = s.tgt, tx = s.tx, ty = s.ty;
    float herr_o = s.herr_o, perr_ox = s.perr_ox, perr_oy = s.perr_oy;
    int istep = s.istep, iwp = s.iwp;
    const bool done = out.done, time_up = out.time_up;
    const int slot = out.slot;
    float o[11];
#pragma unroll
    for (int q = 0; q < 11; q++) o[q] = out.o[q];
#ifndef MVRL_AUV_SKIP   /* traffic-attribution probe builds (tools/variants.py auvskip*): bit 0 hist, 1 obs, 2 reward/done, 3 state */
#define MVRL_AUV_SKIP 0
#endif
    if (!(MVRL_AUV_SKIP & 4)) {
    io.reward[i] = out.reward;
    io.done[i] = done ? (time_up ? 3 : 1) : 0;  // bit 0 = done, bit 1 = time limit (else: bounds exceeded)
    }
    if (io.aux) {  // timeHistory: Fx Fy N u_current v_current rmsAc r0..r4 (:389-403)
        float* ax = io.aux + (size_t)i * 11;
        ax[0] = out.Fg0; ax[1] = out.Fg1; ax[2] = out.Fh2; ax[3] = out.curx; ax[4] = out.cury; ax[5] = out.rms;
        ax[6] = out.t0; ax[7] = out.t1; ax[8] = out.t2; ax[9] = out.t3; ax[10] = out.bonus;
    }
    if (done && io.auto_reset) {
        if (io.term_obs) {
#pragma unroll
            for (int q = 0; q < 11; q++) io.term_obs[(size_t)i * 11 + q] = o[q];
        }
        float v[16];
        const int episode = unpack_int(ST(AV_EPISODE)) + 1;
        ST(AV_EPISODE) = pack_int(episode);
        random_init_auv(p, io.seed, io.env_offset + (int64_t)i, (uint32_t)episode, fl.t_quarter, v);
        x = v[0]; y = v[1]; psi = v[2]; vx = 0.f; vy = 0.f; r = 0.f;
        if (!cyl) tgt = v[3];   // AuvEnvCyl: the target stays waypoints[iWp] - iWp is not reset (_cyl.py:41,141-142)
        ST(AV_TOFF) = v[4];
#pragma unroll
        for (int q = 0; q < 11; q++) ST(AV_MULT + q) = v[5 + q];
        perr_ox = tx - x; perr_oy = ty - y; herr_o = angle_error(tgt, psi);            // herr_o = None -> first call (:160-162)
        istep = 0;
        ST(AV_PHASE) = pack_int((slot + 1) % 10);     // the new episode's ring continues at the next slot (see AV_PHASE)
        observe_auv(p, x, y, psi, vx, vy, r, tx, ty, tgt, herr_o, perr_ox, perr_oy, o);
    } else {
        // only the ring slot that changed is written back
        if (!(MVRL_AUV_SKIP & 1)) {
#pragma unroll
        for (int k = 0; k < 3; k++) ST(AV_HIST + 3 * slot + k) = (k == 0) ? a0 : ((k == 1) ? a1 : a2);
        }
    }
#if MVRL_AUV_LDS_OBS
    if (!(MVRL_AUV_SKIP & 2))
    {   // Row-major [n, 11] observations: written lane-by-lane each store instruction scatters 4-byte pieces over 44-byte
        // strides (partial cache lines).  Transposing the wave's 64 x 11 tile through LDS turns them into 11 stores of
        // 256 contiguous bytes.  Wave-private region, stride 11 (odd) -> conflict-free; only the tail wave of the grid
        // can be partial, and it takes the scattered path.
        __shared__ float tile[(MVRL_BLOCK / 64) * 64 * 11];
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const uint32_t wave_base = (uint32_t)io.lane0 + (blockIdx.x * MVRL_BLOCK + wave * 64u);
        float* t = tile + wave * 704u;
        if (wave_base + 64u <= (uint32_t)io.lane_end) {
#pragma unroll
            for (int q = 0; q < 11; q++) t[lane * 11u + q] = o[q];
            __builtin_amdgcn_wave_barrier();
            float* dst = io.obs + (size_t)wave_base * 11;
#pragma unroll
            for (int q = 0; q < 11; q++) dst[q * 64u + lane] = t[q * 64u + lane];
        } else {
#pragma unroll
            for (int q = 0; q < 11; q++) io.obs[(size_t)i * 11 + q] = o[q];
        }
    }
#else
#pragma unroll
    for (int q = 0; q < 11; q++) io.obs[(size_t)i * 11 + q] = o[q];
#endif
    if (!(MVRL_AUV_SKIP & 8)) {
    ST(AV_X) = x; ST(AV_Y) = y; ST(AV_PSI) = psi; ST(AV_VX) = vx; ST(AV_VY) = vy; ST(AV_R) = r;
    ST(AV_HERR_O) = herr_o; ST(AV_PERR_O) = perr_ox; ST(AV_PERR_O + 1) = perr_oy;
    }
    if (cyl || (done && io.auto_reset)) ST(AV_TGT) = tgt;   // the target only changes on way-point switches / new episodes
    if (cyl) ST(AV_IWP) = pack_int(iwp);
    ST(AV_ISTEP) = pack_int(istep);
#undef ST
}

// Whole episodes of the PD baseline in one launch: evaluate_agent(PDController, AuvEnv) (tag/resources.py:49-102 driving
// verySimpleAuv.py:22-50 and :264-410) for every env of the batch, from its current state until `done` or n_steps.
// The lane's state never leaves its registers between steps: no state round trip through HBM, no observation or
// action arrays, no launch per step - per env the kernel reads 55 words once, gathers the current at every step and
// writes a return, a length and the terminal state.  Same device functions as the step-by-step path
// (pd_policy_kernel + auv_step_kernel), so the two agree to fp32 rounding of the return sum.
template <bool FLOW>
__global__ __launch_bounds__(MVRL_BLOCK) void auv_pd_episode_kernel(const AuvDev p, const FlowDev fl, float* __restrict__ state,
                                                                    int64_t n, double dt64, int max_steps, int n_steps,
                                                                    float inv_pdt, float p0, float p1, float p2, float d0,
                                                                    float d1, float d2, float* __restrict__ returns,
                                                                    int32_t* __restrict__ lengths) {
    const uint32_t i = blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= (uint32_t)n) return;
    const uint32_t n32 = (uint32_t)n;
    char* const stb = reinterpret_cast<char*>(state);
#define ST(k) (*reinterpret_cast<float*>(stb + ((uint32_t)(k) * (n32 * (uint32_t)sizeof(float)) + i * (uint32_t)sizeof(float))))
    AuvLane s;
    MVRL_AUV_LOAD_LANE(s);
    const float P[3] = {p0, p1, p2}, D[3] = {d0, d1, d2};
    float o[11], old[3];
    observe_auv(p, s.x, s.y, s.psi, s.vx, s.vy, s.r, s.tx, s.ty, s.tgt, s.herr_o, s.perr_ox, s.perr_oy, o);
#pragma unroll
    for (int k = 0; k < 3; k++) old[k] = o[k];     // PDController.oldObs is None on the first call
    float ret = 0.f;
    int len = 0;
    bool alive = true;
    for (int t = 0; t < n_steps; t++) {
        if (!alive) continue;                       // lanes that hit the bounds early idle until the wave is done
        float a[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {               // PDController.predict (verySimpleAuv.py:32-50), noise-free
            const float xk = o[k];
            float ak = clampf(xk * P[k] + (xk - old[k]) * inv_pdt * D[k], -1.f, 1.f);
            a[k] = clampf(ak + 0.f, -1.f, 1.f);
            old[k] = xk;
        }
        AuvStepOut out;
        auv_step_core<FLOW>(p, fl, s, a[0], a[1], a[2], (float)dt64, dt64, max_steps, out);
        ret += out.reward;
        len += 1;
#pragma unroll
        for (int q = 0; q < 11; q++) o[q] = out.o[q];
        alive = !out.done;
    }
    returns[i] = ret;
    lengths[i] = len;
    ST(AV_X) = s.x; ST(AV_Y) = s.y; ST(AV_PSI) = s.psi; ST(AV_VX) = s.vx; ST(AV_VY) = s.vy; ST(AV_R) = s.r;
    ST(AV_HERR_O) = s.herr_o; ST(AV_PERR_O) = s.perr_ox; ST(AV_PERR_O + 1) = s.perr_oy;
    ST(AV_TGT) = s.tgt;
    if (p.n_wp > 0) ST(AV_IWP) = pack_int(s.iwp);
#pragma unroll
    for (int q = 0; q < 30; q++) ST(AV_HIST + q) = s.hist[q];
    ST(AV_ISTEP) = pack_int(s.istep);
#undef ST
}

hipError_t launch_auv_pd_episodes(const AuvDev& p, const FlowDev& fl, bool flow, float* state, int64_t n, double dt, int max_steps,
                                  int n_steps, float policy_dt, const float* P, const float* D, float* returns, int32_t* lengths,
                                  hipStream_t stream) {
    dim3 grid((unsigned)((n + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    if (flow) hipLaunchKernelGGL((auv_pd_episode_kernel<true>), grid, block, 0, stream, p, fl, state, n, dt, max_steps, n_steps,
                                 1.0f / policy_dt, P[0], P[1], P[2], D[0], D[1], D[2], returns, lengths);
    else hipLaunchKernelGGL((auv_pd_episode_kernel<false>), grid, block, 0, stream, p, fl, state, n, dt, max_steps, n_steps,
                            1.0f / policy_dt, P[0], P[1], P[2], D[0], D[1], D[2], returns, lengths);
    return hipGetLastError();
}

__global__ __launch_bounds__(MVRL_BLOCK) void auv_reset_kernel(const AuvDev p, float* state, int64_t n, const uint8_t* mask,
                                                               const float* init, float* obs, uint64_t seed,
                                                               int64_t env_offset, float t_quarter) {
    const int64_t i = (int64_t)blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (mask && !mask[i]) return;
    float* st = state + i;
    const int episode = unpack_int(st[AV_EPISODE * n]) + 1;
    st[AV_EPISODE * n] = pack_int(episode);
    float v[16];
    if (init) {
#pragma unroll
        for (int q = 0; q < 16; q++) v[q] = init[i * 16 + q];
    } else {
        random_init_auv(p, seed, env_offset + i, (uint32_t)episode, t_quarter, v);
    }
    const float x = v[0], y = v[1], psi = v[2];
    float tgt = v[3], tx = 0.f, ty = 0.f;
    if (p.n_wp > 0) {
        // explicit initial values carry iWp in slot 3; a random reset keeps the env's current iWp (_cyl.py:41)
        int iwp = init ? max(0, min(p.n_wp - 1, (int)v[3])) : unpack_int(st[AV_IWP * n]);
        iwp = max(0, min(p.n_wp - 1, iwp));
        st[AV_IWP * n] = pack_int(iwp);
        waypoint(p, iwp, tx, ty, tgt);
    }
    st[AV_X * n] = x; st[AV_Y * n] = y; st[AV_PSI * n] = psi;
    st[AV_VX * n] = 0.f; st[AV_VY * n] = 0.f; st[AV_R * n] = 0.f;
    st[AV_TGT * n] = tgt;
    const float herr_o = angle_error(tgt, psi), perr_ox = tx - x, perr_oy = ty - y;
    st[AV_HERR_O * n] = herr_o; st[AV_PERR_O * n] = perr_ox; st[(AV_PERR_O + 1) * n] = perr_oy;
#pragma unroll
    for (int q = 0; q < 11; q++) st[(AV_MULT + q) * n] = v[5 + q];
    st[AV_TOFF * n] = v[4];
#pragma unroll
    for (int q = 0; q < 30; q++) st[(AV_HIST + q) * n] = 0.f;
    // the ring of the new episode starts where this env would have written next: envs that shared a slot keep sharing it
    int ph = (unpack_int(st[AV_ISTEP * n]) + unpack_int(st[AV_PHASE * n])) % 10;
    st[AV_PHASE * n] = pack_int(ph < 0 ? 0 : ph);
    st[AV_ISTEP * n] = pack_int(0);
    if (obs) {
        float o[11];
        observe_auv(p, x, y, psi, 0.f, 0.f, 0.f, tx, ty, tgt, herr_o, perr_ox, perr_oy, o);
#pragma unroll
        for (int q = 0; q < 11; q++) obs[i * 11 + q] = o[q];
    }
}

// io.k_steps consecutive AuvEnv steps per launch (mvrl_rollout_dev): actions[k] -> obs[k] / reward[k] / done[k], the
// lane's 55 words in registers from the first step to the last.  Same auv_step_core and the same reset sequence as
// auv_step_kernel; being a different kernel it may round the last bit differently (fast-math), so its results are those of
// k_steps single-step launches to fp32 rounding, not bit for bit (the rigid-body kernels, one template for both, are).
template <bool FLOW>
__global__ __launch_bounds__(MVRL_BLOCK) void auv_rollout_kernel(const AuvDev p, const StepIO io, const FlowDev fl) {
    const uint32_t i = (uint32_t)io.lane0 + blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= (uint32_t)io.lane_end) return;
    const uint32_t n32 = (uint32_t)io.n;
    char* const stb = reinterpret_cast<char*>(io.state);
#define ST(k) (*reinterpret_cast<float*>(stb + ((uint32_t)(k) * (n32 * (uint32_t)sizeof(float)) + i * (uint32_t)sizeof(float))))
    const bool cyl = p.n_wp > 0;
    AuvLane s;
    MVRL_AUV_LOAD_LANE(s);
    const size_t n = (size_t)io.n;
#pragma nounroll
    for (int k = 0; k < io.k_steps; k++) {
        const float* ap = io.actions + ((size_t)k * n + i) * 3;
        AuvStepOut out;
        auv_step_core<FLOW>(p, fl, s, ap[0], ap[1], ap[2], io.dt, io.dt64, io.max_steps, out);
        io.reward[(size_t)k * n + i] = out.reward;
        io.done[(size_t)k * n + i] = out.done ? (out.time_up ? 3 : 1) : 0;
        if (out.done && io.auto_reset) {
            if (io.term_obs) {
#pragma unroll
                for (int q = 0; q < 11; q++) io.term_obs[(size_t)i * 11 + q] = out.o[q];
            }
            float v[16];
            const int episode = unpack_int(ST(AV_EPISODE)) + 1;
            ST(AV_EPISODE) = pack_int(episode);
            random_init_auv(p, io.seed, io.env_offset + (int64_t)i, (uint32_t)episode, fl.t_quarter, v);
            s.x = v[0]; s.y = v[1]; s.psi = v[2]; s.vx = 0.f; s.vy = 0.f; s.r = 0.f;
            if (!cyl) s.tgt = v[3];
            s.toff = v[4];
#pragma unroll
            for (int q = 0; q < 11; q++) s.mu[q] = v[5 + q];
            s.perr_ox = s.tx - s.x; s.perr_oy = s.ty - s.y; s.herr_o = angle_error(s.tgt, s.psi);
            s.istep = 0;
            s.phase = (out.slot + 1) % 10;
            observe_auv(p, s.x, s.y, s.psi, s.vx, s.vy, s.r, s.tx, s.ty, s.tgt, s.herr_o, s.perr_ox, s.perr_oy, out.o);
        }
        float* orow = io.obs + ((size_t)k * n + i) * 11;
#pragma unroll
        for (int q = 0; q < 11; q++) orow[q] = out.o[q];
    }
    ST(AV_X) = s.x; ST(AV_Y) = s.y; ST(AV_PSI) = s.psi; ST(AV_VX) = s.vx; ST(AV_VY) = s.vy; ST(AV_R) = s.r;
    ST(AV_HERR_O) = s.herr_o; ST(AV_PERR_O) = s.perr_ox; ST(AV_PERR_O + 1) = s.perr_oy;
    ST(AV_TGT) = s.tgt;
    if (cyl) ST(AV_IWP) = pack_int(s.iwp);
#pragma unroll
    for (int q = 0; q < 11; q++) ST(AV_MULT + q) = s.mu[q];
    if (FLOW) ST(AV_TOFF) = s.toff;
#pragma unroll
    for (int q = 0; q < 30; q++) ST(AV_HIST + q) = s.hist[q];
    ST(AV_ISTEP) = pack_int(s.istep);
    ST(AV_PHASE) = pack_int(s.phase);
#undef ST
}

hipError_t launch_auv_step(const AuvDev& p, const StepIO& io, const FlowDev& fl, bool flow, hipStream_t stream) {
    const int64_t lanes = io.lane_end - io.lane0;
    dim3 grid_r((unsigned)((lanes + MVRL_BLOCK - 1) / MVRL_BLOCK)), block_r(MVRL_BLOCK);
    if (io.k_steps > 1) {
        if (flow) hipLaunchKernelGGL((auv_rollout_kernel<true>), grid_r, block_r, 0, stream, p, io, fl);
        else hipLaunchKernelGGL((auv_rollout_kernel<false>), grid_r, block_r, 0, stream, p, io, fl);
        return hipGetLastError();
    }
    dim3 grid((unsigned)((lanes + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    if (flow) hipLaunchKernelGGL((auv_step_kernel<true>), grid, block, 0, stream, p, io, fl);
    else hipLaunchKernelGGL((auv_step_kernel<false>), grid, block, 0, stream, p, io, fl);
    return hipGetLastError();
}

// dataToState(position, heading, velocities) of every env's current state (verySimpleAuv.py:147-214, _cyl.py:100-132) through
// observe_auv with the stored herr_o / perr_o (mvrl_observe); nothing is modified.
__global__ __launch_bounds__(MVRL_BLOCK) void auv_observe_kernel(const AuvDev p, const float* state, int64_t n, float* obs) {
    const int64_t i = (int64_t)blockIdx.x * MVRL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* st = state + i;
    float tx = 0.f, ty = 0.f, tgt = st[AV_TGT * n];
    if (p.n_wp > 0) waypoint(p, unpack_int(st[AV_IWP * n]), tx, ty, tgt);
    float o[11];
    observe_auv(p, st[AV_X * n], st[AV_Y * n], st[AV_PSI * n], st[AV_VX * n], st[AV_VY * n], st[AV_R * n], tx, ty, tgt,
                st[AV_HERR_O * n], st[AV_PERR_O * n], st[(AV_PERR_O + 1) * n], o);
#pragma unroll
    for (int q = 0; q < 11; q++) obs[i * 11 + q] = o[q];
}

hipError_t launch_auv_observe(const AuvDev& p, const float* state, int64_t n, float* obs, hipStream_t stream) {
    dim3 grid((unsigned)((n + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    hipLaunchKernelGGL(auv_observe_kernel, grid, block, 0, stream, p, state, n, obs);
    return hipGetLastError();
}

hipError_t launch_auv_reset(const AuvDev& p, float* state, int64_t n, const uint8_t* mask, const float* init, float* obs,
                            uint64_t seed, int64_t env_offset, float t_quarter, hipStream_t stream) {
    dim3 grid((unsigned)((n + MVRL_BLOCK - 1) / MVRL_BLOCK)), block(MVRL_BLOCK);
    hipLaunchKernelGGL(auv_reset_kernel, grid, block, 0, stream, p, state, n, mask, init, obs, seed, env_offset,
                       t_quarter);
    return hipGetLastError();
}

}  // namespace mvrl
