/*
 * mvrl.h - C ABI of libmvrl.so: MI355X-native vectorised marine-vehicle environments.
 *
 * This is the drop-in boundary for the reference's environment hot path.  Every entry point
 * cites the reference interface it replaces (paths relative to the reference checkout;
 * "6DoF.py" = dynamicsModel_BlueROV2_Heavy_6DoF.py, "3DoF.py" = dynamicsModel_BlueROV2_Heavy_3DoF.py,
 * "tag/" = tag_00_Dec2023_simpleControlTurbulence/).
 *
 * Conventions
 *   - plain C symbols, plain pointers and sizes, no C++/torch types;
 *   - every function returns 0 on success or a negative MVRL_E* code; the message is
 *     available through mvrl_last_error(); no exception crosses the ABI;
 *   - one handle = one device, one internal HIP stream, one host thread at a time; several devices in one process: mvrl_group_*
 *     (one shard handle per device, a grouped RCCL send / recv to gather the outputs; ABI 4);
 *   - host-pointer entry points (`mvrl_step`, `mvrl_reset`, ...) are synchronous on return;
 *     `*_dev` entry points take DEVICE pointers, enqueue on the caller's hipStream_t, taken
 *     literally (NULL is HIP's null stream - what torch's default stream is), and return
 *     after enqueue; an event recorded behind each such enqueue orders any later
 *     host-pointer call (which runs on the handle's internal stream) after it, so e.g.
 *     mvrl_get_state after mvrl_step_dev needs no explicit synchronisation (one such event per
 *     caller stream, all joined); work the caller spreads over several of its own streams is
 *     ordered among those streams by the caller;
 *   - actions / observations are row-major [n_envs, dim] float32 at the ABI (what the
 *     reference's Gym API and SB3's VecEnv exchange); internal state is SoA in HBM;
 *   - the library owns all device buffers; it never retains a caller pointer past a call.
 *
 * There is NO CPU fallback: without a HIP device every compute entry point fails with
 * MVRL_ENODEV.
 */
#ifndef MVRL_H
#define MVRL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVRL_ABI_VERSION 4

/* ---- models (which reference environment the handle replaces) ------------------------------ */
#define MVRL_MODEL_AUV 0  /* AuvEnv, explicit Euler + turbulence current   tag/verySimpleAuv.py:76-416 */
#define MVRL_MODEL_ROV3 1 /* BlueROV2Heavy3DoFEnv                           3DoF.py:375-514             */
#define MVRL_MODEL_ROV6 2 /* BlueROV2Heavy6DoFEnv                           6DoF.py:445-594             */

/* ---- where the PID sits relative to the integrator ---------------------------------------- */
#define MVRL_CTRL_FAITHFUL 0 /* PID inside the RHS, evaluated (and mutated) at every RK stage - as 6DoF.py:418 */
#define MVRL_CTRL_ZOH 1      /* PID + allocation once per sub-step, thruster rpm held over the stages          */

/* ---- arithmetic / storage precision of a handle ---------------------------------------------------------- */
#define MVRL_PREC_F32 0 /* fp32 state, arithmetic and ABI arrays (float*): the throughput path (1.0e10 env-steps/s on 6-DoF +
                           turbulence; 1e-5 against the reference over tens of steps, 59 % of envs over a whole episode)  */
#define MVRL_PREC_F64 1 /* fp64 everywhere (double*; use the *_f64 entry points): the mode that follows the reference for WHOLE
                           episodes - same kernel text widened (tools/gen_f64.py), 4.0e9 env-steps/s, single calls to ~1e-12,
                           every env of 65 536 within 1e-5 after 250 steps under random actions      */

/* ---- time integrator of the 3/6-DoF models ------------------------------------------------------------- */
#define MVRL_INTEG_RK4 0  /* classic RK4, n_substeps sub-steps per env step (this build's integrator)           */
#define MVRL_INTEG_RK45 1 /* the reference's own: scipy solve_ivp(method="RK45", max_step=dt, rtol=atol=1e-3)
                             restated per lane (6DoF.py:555-557, 3DoF.py:475-477); needs MVRL_PREC_F64         */

/* ---- error codes --------------------------------------------------------------------------- */
#define MVRL_OK 0
#define MVRL_EINVAL (-1)  /* bad argument / configuration     */
#define MVRL_ENODEV (-2)  /* no usable HIP device             */
#define MVRL_ENOMEM (-3)  /* device or host allocation failed */
#define MVRL_EHIP (-4)    /* a HIP runtime call failed        */
#define MVRL_ESTATE (-5)  /* call sequence error (e.g. step_wait without step_async) */

/* 6-DoF vehicle + PID constants.  Values are produced on the host in fp64 from the reference's
 * literals (6DoF.py:83-218 vehicle, :43-53 PID) - see marinevehiclereinforcementlearning_amd/params.py -
 * and narrowed to fp32 when uploaded. Matrices are row-major. */
typedef struct mvrl_rov6_params {
    double m;             /* 11.4                                  6DoF.py:88  */
    double length;        /* 0.457                                 6DoF.py:90  */
    double cg[3];         /* [0,0,0.05]                            6DoF.py:95  */
    double cb[3];         /* [0,0,0]                               6DoF.py:94  */
    double inertia[9];    /* 0.16*I3                               6DoF.py:97  */
    double weight;        /* m*9.81                                6DoF.py:372 */
    double buoyancy;      /* dispVol*rho*9.81                      6DoF.py:373 */
    double added[6];      /* Xudot,Yvdot,Zwdot,Kpdot,Mqdot,Nrdot (Coriolis Ca)  6DoF.py:334-341 */
    double minv[36];      /* inverse of M = Mrb + Ma               6DoF.py:286-299,428 */
    double mass[36];      /* M itself (returned by forceModel)     6DoF.py:299 */
    double dlin[36];      /* Dl  = -[linear coeffs]                6DoF.py:345-352 */
    double dquad[36];     /* -[quadratic coeffs]; entry (i,j) multiplies |nu_j|  6DoF.py:354-368 */
    double alloc[48];     /* A    6x8                              resources.py:27-32 */
    double alloc_inv[48]; /* Ainv 8x6 = pinv(A)                    resources.py:33 */
    double thrust_k;      /* rho*D^4*Kt : F = thrust_k*(rpm/60)^2*sign(rpm)   6DoF.py:228,235 */
    double rpm_max;       /* 3500                                  6DoF.py:272 */
    double rpm_deadband;  /* 300                                   6DoF.py:273 */
    double kp[6], ki[6], kd[6], windup[6], umax[6]; /* 6DoF.py:46-53 */
    double act_scale[6];  /* [2L,2L,2L,pi/4,pi/4,pi/4]             6DoF.py:545-551 */
    double obs_pos_scale; /* 3L                                    6DoF.py:472 */
    double obs_ang_scale; /* pi/4                                  6DoF.py:480 */
} mvrl_rov6_params;

/* 3-DoF vehicle (PID inlined in derivs) - 3DoF.py:26-126, :141-157 */
typedef struct mvrl_rov3_params {
    double m, length;
    double cg[3];
    double izz;
    double added[3];       /* Xudot, Yvdot, Nrdot                   3DoF.py:58-60 */
    double minv[9];        /* inverse of M                          3DoF.py:198-206,283 */
    double mass[9];
    double dlin[9];        /* 3DoF.py:225-229 */
    double dquad[9];       /* 3DoF.py:231-239 ; entry (i,j) multiplies [|uRel|,|vRel|,|r|][j] */
    double alloc_inv[12];  /* Ainv 4x3                              3DoF.py:104-112 */
    double thrust_k;       /* rho*D^4*Kt                            3DoF.py:118 */
    double rpm_max, rpm_deadband;
    double cos_alpha, sin_alpha; /* 45 deg                           3DoF.py:90 */
    double yaw_arm;        /* sqrt(l_x^2+l_y^2)                     3DoF.py:263 */
    double jet_area_k;     /* 0.5*rho*pi*D^2                        3DoF.py:121 */
    double jet_c1, jet_k1, jet_c2, jet_k2; /* 0.56599,7.60891,0.05654,0.89679  3DoF.py:122-123 */
    double jet_drag_k;     /* 0.5*rho*dispVol^(2/3)                 3DoF.py:124 */
    double kp[3], ki[3], kd[3], windup[3], umax[3]; /* 3DoF.py:142-154 */
    double act_scale[3];   /* [2L,2L,pi/4]                          3DoF.py:469-471 */
    double obs_pos_scale, obs_ang_scale;
} mvrl_rov3_params;

#define MVRL_MAX_WAYPOINTS 32
/* Simplified AUV - tag/verySimpleAuv.py:106-127 ; its way-point variant AuvEnvCyl - tag/verySimpleAuv_cyl.py:22-344 */
typedef struct mvrl_auv_params {
    double m, izz;
    double xuu, yvv, nrr, xu, yv, nr;
    double max_force, max_moment;
    double x_min, x_max, y_min, y_max;   /* :106-107 */
    double noise_mag_coeffs, noise_mag_actuation; /* :124-125 */
    int32_t stop_on_bounds;              /* :88 */
    int32_t n_waypoints;                 /* 0: AuvEnv (target = origin, random target heading).  > 0: AuvEnvCyl - the target
                                            follows waypoints[iWp] and advances when closer than wp_threshold
                                            (tag/verySimpleAuv_cyl.py:30-40, :249-253) */
    double obs_scale[9];                 /* observation = clip(raw * scale): all 1 except herr/(45 deg) for AuvEnv "V3"
                                            (verySimpleAuv.py:201-212); the "V0" scaling of AuvEnvCyl (_cyl.py:100-111) */
    double wp_threshold;                 /* Rcyl * 0.05 */
    double waypoints[3 * MVRL_MAX_WAYPOINTS]; /* x, y, target heading per way-point */
} mvrl_auv_params;

/* Turbulence table as produced by ReconstructedFlow.scale (tag/flowGenerator.py:53-95):
 * float32 [n_t][n_y][n_x][2] = (u, v), spacing dt/dx/dy AFTER scaling; the interpolation ignores the
 * origin (as the reference does, flowGenerator.py:118-120). */
typedef struct mvrl_flow_desc {
    int32_t n_t, n_y, n_x, _pad;
    double dt, dx, dy;
} mvrl_flow_desc;

typedef struct mvrl_config {
    int32_t abi_version;    /* MVRL_ABI_VERSION */
    int32_t model;          /* MVRL_MODEL_* */
    int32_t device;         /* HIP device ordinal */
    int32_t n_substeps;     /* RK4 sub-steps per env step (3/6-DoF; >= 1).  AUV: ignored (Euler) */
    int64_t n_envs;         /* environments owned by this handle (this shard) */
    int64_t env_offset;     /* global index of the first env: RNG streams are keyed by the GLOBAL id */
    double dt;              /* env step (6DoF.py:446 dt=0.2 ; verySimpleAuv.py:77 dt=0.02) */
    int32_t max_steps;      /* episode length (250) */
    int32_t control_mode;   /* MVRL_CTRL_* (3/6-DoF) */
    int32_t fixed_setpoint; /* 1: ignore actions, hold the reset set-point (the reference's fixedSp, 6DoF.py:536-541) */
    int32_t auto_reset;     /* 1: SB3 VecEnv semantics - done lanes are re-initialised inside step */
    uint64_t seed;          /* counter-based RNG key for resets without explicit initial values */
    int32_t use_flow;       /* 1: sample the turbulence table each step (AUV: as the reference; 3/6-DoF: SURVEY 9.5) */
    int32_t precision;      /* MVRL_PREC_* */
    int32_t integrator;     /* MVRL_INTEG_* (3/6-DoF) */
    int32_t _pad;
    mvrl_rov6_params rov6;
    mvrl_rov3_params rov3;
    mvrl_auv_params auv;
} mvrl_config;

typedef struct mvrl_handle mvrl_handle;

/* The configuration the reference's constructors amount to - BlueROV2Heavy6DoFEnv() / BlueROV2Heavy3DoFEnv() / AuvEnv() with their default
 * arguments (6DoF.py:83-218, :445-465; 3DoF.py:26-126, :375-395; verySimpleAuv.py:76-127): every constant of the three parameter blocks
 * (incl. M^-1 and pinv(A), which a C caller would otherwise have to compute), dt 0.2 / 0.02 s, 250-step episodes, and this build's choices
 * for what the reference does not have: RK4 with 4 sub-steps, PID at every stage (FAITHFUL), fp32, auto-reset on, turbulence for AuvEnv
 * only, device 0, seed 0.  Edit the fields you want to differ, then mvrl_create.  (Python: params.make_config is the same thing.) */
int mvrl_default_config(int32_t model, int64_t n_envs, mvrl_config* out);

/* ---- introspection ------------------------------------------------------------------------- */
int mvrl_abi_version(void);
/* number of visible HIP devices (0 when none; never fails) */
int mvrl_device_count(void);
/* Last error message of `h`, or of the last failed call without a handle when h == NULL. */
const char* mvrl_last_error(const mvrl_handle* h);
/* dims for a model: returns 0 / MVRL_EINVAL */
int mvrl_model_dims(int32_t model, int32_t* act_dim, int32_t* obs_dim, int32_t* init_dim, int32_t* state_words);
/* width of one row of the aux side outputs (see mvrl_enable_aux), or MVRL_EINVAL */
int mvrl_aux_dim(int32_t model);
/* Which kernel flavour the handle dispatches to, e.g. "rov6/baked/faithful+flow":
 *   baked   - runtime constants equal the reference's defaults bit-for-bit -> literals in the instruction stream
 *   ctrl    - (6-DoF, fp32) the reference's vehicle with other PID gains / limits / action and observation scales: vehicle
 *             constants as literals, the controller's 38 numbers read at run time
 *   sym     - BlueROV2-Heavy structure (sparse, sign-symmetric thruster layout), constants read at run time
 *   generic - arbitrary constants, dense matrices */
const char* mvrl_variant(const mvrl_handle* h);

/* ---- lifetime: replaces Env.__init__ (6DoF.py:446-465, 3DoF.py:376-395, verySimpleAuv.py:77-145) ---- */
int mvrl_create(const mvrl_config* cfg, mvrl_handle** out);
void mvrl_destroy(mvrl_handle* h);

/* Upload the scaled turbulence table (host float32 [n_t][n_y][n_x][2]).
 * Replaces the table held by ReconstructedFlow after scale() (flowGenerator.py:76-95).  The handle keeps its own copy in
 * the layout its kernels read - the 2 x 2 x 2 interpolation stencil of every cell in one 64-byte cache line: eight times
 * the table's size in device memory (320 MB for 2000 snapshots of 41 x 61), one instead of 4.5 scattered cache lines per
 * lookup.
 * How the step kernels sample it: AuvEnv / AuvEnvCyl handles exactly as ReconstructedFlow.interp does (flowGenerator.py:97-136:
 * cell index clamped, weights not - linear extrapolation outside the table).  3-/6-DoF handles (the "+ turbulence" composition,
 * which has no reference counterpart) sample it like interp INSIDE the table and, outside it, hold the boundary value in space and
 * reflect time over the table's duration: those vehicles leave the table and their episodes outlast it, and extrapolated
 * linearly the current grows without bound.  mvrl_flow_interp below is interp itself. */
int mvrl_set_flow(mvrl_handle* h, const float* table_host, const mvrl_flow_desc* desc);
int mvrl_set_flow_f64(mvrl_handle* h, const double* table_host, const mvrl_flow_desc* desc);
/* Same, table already resident on the handle's device in the handle's precision (read once, here: later changes to the
 * caller's table are not seen, and the caller may free it after the call). */
int mvrl_set_flow_dev(mvrl_handle* h, const void* table_dev, const mvrl_flow_desc* desc);

/* ---- reset: replaces Env.reset (6DoF.py:485-529, 3DoF.py:411-453, verySimpleAuv.py:216-262) ----------
 * mask : n_envs bytes, non-zero = reset this env; NULL = all.
 * init : [n_envs, init_dim] explicit initial values, NULL = draw from the counter-based RNG.
 *        ROV6: wp0(3) wp1(3) target angles(3)   (reset(initialSetpoint=sp) == wp0=wp1=sp[:3], angles=sp[3:])
 *        ROV3: wp0(2) wp1(2) target heading(1)
 *        AUV : x y heading headingTarget flowTimeOffset  mMult IMult XuuMult YvvMult NrrMult XuMult YvMult NrMult
 *              XactMult YactMult NactMult   (fixedInitialValues + the multipliers of verySimpleAuv.py:222-229,245);
 *              with way-points (AuvEnvCyl) slot 3 carries the way-point index iWp instead of the target heading
 * obs  : [n_envs, obs_dim] out (rows of envs that were not reset are left untouched); may be NULL. */
int mvrl_reset(mvrl_handle* h, const uint8_t* mask, const float* init, float* obs);
int mvrl_reset_f64(mvrl_handle* h, const uint8_t* mask, const double* init, double* obs);
/* device pointers in the handle's precision (float* for MVRL_PREC_F32, double* for MVRL_PREC_F64) */
int mvrl_reset_dev(mvrl_handle* h, const uint8_t* mask_dev, const void* init_dev, void* obs_dev, void* stream);

/* ---- step: replaces Env.step (6DoF.py:531-594, 3DoF.py:455-514, verySimpleAuv.py:264-410) -------------
 * and SB3 VecEnv.step_async/step_wait (called at tag/main_00_sbl.py:145-161 through agent.learn).
 * actions [n_envs, act_dim] f32 ; obs [n_envs, obs_dim] f32 ; reward [n_envs] f32 ; done [n_envs] u8:
 * 0 = running, non-zero = done; bit 1 (value 2) is set when the episode ended on the time limit, clear when it
 * ended on a bounds violation (AuvEnv, verySimpleAuv.py:335-342).  The reference's envs do not tell the two apart
 * (info = {}); the bit is extra information a caller may surface as SB3's info["TimeLimit.truncated"]
 * (MarineVecEnv(report_truncation=True)). */
int mvrl_step(mvrl_handle* h, const float* actions, float* obs, float* reward, uint8_t* done);
int mvrl_step_async(mvrl_handle* h, const float* actions);
int mvrl_step_wait(mvrl_handle* h, float* obs, float* reward, uint8_t* done);
int mvrl_step_f64(mvrl_handle* h, const double* actions, double* obs, double* reward, uint8_t* done);
int mvrl_step_dev(mvrl_handle* h, const void* actions_dev, void* obs_dev, void* reward_dev, uint8_t* done_dev,
                  void* stream);

/* The same step for the lanes [first_env, first_env + n_range) only (first_env a multiple of 64).  Environments are
 * independent, so a caller may run sub-batches as separate CHAINS, each on its own stream - e.g. two half-batches whose
 * policy/step sequences overlap (policy(A) while step(B)), which also hides the gap between dependent launches and the
 * ramp/tail of every launch behind the other chain's kernel.  Pointers are the bases of the FULL [n_envs, dim] arrays
 * (rows outside the range are not touched).  Replaces stepping a subset of SubprocVecEnv's workers
 * (SB3 VecEnv.step_async on `indices`; the reference steps all of them, tag/main_00_sbl.py:145). */
int mvrl_step_range_dev(mvrl_handle* h, int64_t first_env, int64_t n_range, const void* actions_dev, void* obs_dev,
                        void* reward_dev, uint8_t* done_dev, void* stream);

/* The handle's own pinned, device-visible staging block of the host-buffer step: actions[n_envs, act_dim], obs[n_envs, obs_dim],
 * reward[n_envs] in the handle's precision, done[n_envs] u8 - valid for the handle's lifetime (freed by mvrl_destroy).  A caller that
 * writes its actions THERE and passes these very pointers to mvrl_step / mvrl_step_async / mvrl_step_wait saves the two staging
 * copies of a step (SB3's numpy buffers are ordinary pageable memory, which is why the copies exist: 25 + 43 MB per step at
 * 1 048 576 6-DoF envs); outputs are overwritten by the next step.  The ACTION block must not be written between mvrl_step_async
 * and the matching mvrl_step_wait: for small batches (n_envs <= 4096) the kernel in flight reads it directly (any OTHER caller
 * buffer handed to mvrl_step_async is free again when the call returns). */
int mvrl_host_buffers(mvrl_handle* h, void** actions, void** obs, void** reward, uint8_t** done);
/* Observation of the step on which an env finished (SB3 infos[i]["terminal_observation"]); rows of envs
 * that did not finish on the last step are unspecified.  Only meaningful with auto_reset = 1. */
int mvrl_get_terminal_obs(mvrl_handle* h, float* obs);
int mvrl_get_terminal_obs_f64(mvrl_handle* h, double* obs);
/* k_steps consecutive env steps in one call: actions_dev [k][n][act_dim], obs_dev [k][n][obs_dim], reward_dev [k][n],
 * done_dev [k][n] (device pointers, the handle's precision) - the results of k mvrl_step_dev calls, auto-resets included.
 * For open-loop roll-outs (random / recorded / action-repeat actions: BASELINE's "random-action rollouts"); the fp32 kernels
 * (6-DoF: those with the reference's structure) run all k steps in ONE launch with the env state in registers between
 * the steps; everything else is stepped launch by launch.  Rigid-body results equal k single steps bit for bit, AuvEnv's
 * to fp32 rounding.  The terminal-observation buffer keeps the LAST termination of each env. */
int mvrl_rollout_dev(mvrl_handle* h, const void* actions_dev, void* obs_dev, void* reward_dev, uint8_t* done_dev, int32_t k_steps,
                     void* stream);
int mvrl_get_terminal_obs_dev(mvrl_handle* h, void* obs_dev, void* stream);

/* ---- raw state (checkpoint / parity tests): SoA [state_words][n_envs] in the handle's precision.  Planes:
 *   ROV6 (41): y[12] eOld[6] eInt[6] setPoint[6] path[6] episode tOld time flowTimeOffset iStep
 *   ROV3 (24): y[6] eOld[3] eInt[3] setPoint[3] path[4] episode tOld time flowTimeOffset iStep
 *   AUV  (56): pose[6] headingTarget herr_o perr_o[2] mult[11] flowTimeOffset actionRing[30] iStep iWp episode ringPhase
 *              (the action of step iStep sits in ring slot (iStep - 1 + ringPhase) % 10: a new episode's ring continues
 *              at the slot where the previous one stopped, so the envs of a wave keep writing one plane per step)
 * BINARY ANGLES (ABI 3): in an fp32 handle the Euler-angle words of the rigid-body models - y[3..5] (phi, theta, psi) of ROV6, y[2] (psi)
 * of ROV3 - are uint32 bit patterns b with angle = b * 2 pi / 2^32 in [0, 2 pi): the reference wraps its angles into that interval
 * (6DoF.py:560, 3DoF.py:480), where an fp32 NUMBER is only resolved to 4.8e-7 rad; the binary angle resolves 1.5e-9 rad everywhere,
 * wraps exactly (integer overflow) and lets a step ADD its small increment.  Encode: b = llrint(fmod(angle, 2 pi) *
 * 2^32 / (2 pi)) mod 2^32; decode: (double)b * 2 pi / 2^32.  fp64 handles keep plain doubles in [0, 2 pi).  AuvEnv is unchanged.
 * iStep / iWp / episode are integer bit patterns (int32 in a float slot / int64 in a double slot).  `episode` counts
 * the env's resets and is the counter of its Philox stream: random resets depend on (seed, global env id, episode)
 * only, not on how many launches the handle has issued - restoring a state restores the RNG position, and a sequence
 * of mvrl_step_dev launches can be captured into a HIP graph and replayed. ---- */
int mvrl_get_state(mvrl_handle* h, float* buf, size_t n_elems);
int mvrl_set_state(mvrl_handle* h, const float* buf, size_t n_elems);
int mvrl_get_state_f64(mvrl_handle* h, double* buf, size_t n_elems);
int mvrl_set_state_f64(mvrl_handle* h, const double* buf, size_t n_elems);

/* obs[n_envs, obs_dim] = dataToState of every env's CURRENT state (6DoF.py:467-483, 3DoF.py:397-409, verySimpleAuv.py:147-214 with
 * the stored herr_o / perr_o; scripts call it directly: tag/script_4_compareRLandPID.py:100-112), e.g. after mvrl_set_state.
 * Evaluated by the device function the step and reset kernels use; the state is not modified. */
int mvrl_observe(mvrl_handle* h, float* obs);
int mvrl_observe_f64(mvrl_handle* h, double* obs);

/* ---- one evaluation of the vehicle's derivs(t, y) for n independent tuples (unit-level parity, system identification):
 * BlueROV2Heavy6DoF.derivs 6DoF.py:406-442 (PID :43-73 -> allocateThrust :220 -> thrusterModel/limit -> forceModel
 * :253-404 -> solve :428 -> J :430) / BlueROV2Heavy3DoF.derivs 3DoF.py:128-296, with the handle's constants.
 * Row-major host arrays: t[n], y[n,2*dof], sp[n,dof]; controller memory eold[n,dof], eint[n,dof], told[n] is read AND
 * updated exactly as the call mutates it; has_old[n] = 0 means controller.eOld is None (first call).  Outputs dy[n,2*dof]
 * and, if non-NULL, the side outputs gcf[n,dof] (controller demand; 3-DoF: resolved into the body frame as the
 * reference's timeHistory stores it) and rpm[n,8|4].  Zero current.  Rigid-body models only.
 * t and told are DOUBLE for both precisions (ABI 2): the controller uses t - tOld, which two fp32 times near t = 40 s resolve
 * to 4e-6 only; an fp32 handle forms the difference in fp64 on the host and computes everything else in fp32. ---- */
int mvrl_derivs(mvrl_handle* h, int64_t n, const double* t, const float* y, const float* sp, float* eold, float* eint, double* told,
                const uint8_t* has_old, float* dy, float* gcf, float* rpm);
int mvrl_derivs_f64(mvrl_handle* h, int64_t n, const double* t, const double* y, const double* sp, double* eold, double* eint,
                    double* told, const uint8_t* has_old, double* dy, double* gcf, double* rpm);
/* The same call with a water current: cur[n,2] = (u_c, v_c) in the GLOBAL frame per tuple (NULL = zero = mvrl_derivs) - the
 * `velCurrent` the reference declares and leaves at zero ("TODO add a current model", 3DoF.py:182-191, 6DoF.py:257-267): resolved
 * into the body frame (3-DoF: pinv(J) = J^T; 6-DoF: globalToVehicle of the translational part), velRel = vel - velCurrent enters
 * Ca / Dq (3-DoF) and the (Ca + D) velRel product (both).  What the step kernels apply when a turbulence table is set
 * (SURVEY 9.5); pinned by goldens G21 / G22, which EXECUTE those reference lines with a non-zero current (ABI 4). */
int mvrl_derivs_cur(mvrl_handle* h, int64_t n, const double* t, const float* y, const float* sp, const float* cur, float* eold,
                    float* eint, double* told, const uint8_t* has_old, float* dy, float* gcf, float* rpm);
int mvrl_derivs_cur_f64(mvrl_handle* h, int64_t n, const double* t, const double* y, const double* sp, const double* cur, double* eold,
                        double* eint, double* told, const uint8_t* has_old, double* dy, double* gcf, double* rpm);

/* ---- groups: the GPUs of one node behind ONE object in ONE host process (ABI 4) ---------------------------------------------
 * Replaces SB3's SubprocVecEnv - one Python process per env, pipe send / recv around env.step (tag/main_00_sbl.py:145-146) - for a
 * caller without torch.distributed (BASELINE configs[4] from C).  The batch of cfg->n_envs environments is cut into contiguous
 * shards (mvrl_group_shard_range: the first n_envs % n_devices shards own one env more); shard i is an ordinary handle on
 * devices[i] with env_offset = cfg->env_offset + its first env, so random resets - Philox keyed by the GLOBAL env id - make the
 * shards together bit-identical to the unsharded batch.  A step is one launch per device with no host synchronisation between
 * them; the ONE exchange of the path, returning (observation, reward, done) to the root device, is a grouped ncclSend / ncclRecv
 * over RCCL / xGMI (librccl is dlopen-ed at group creation: the library itself loads without it).  A group of one device, one with
 * a repeated device (RCCL refuses duplicates: rehearsal on a 1-GPU box) or MVRL_GROUP_TRANSPORT=copy moves the messages with
 * device-to-device / peer copies instead; MVRL_GROUP_TRANSPORT=rccl makes even a one-device group go through RCCL (a one-rank
 * communicator sending to itself).  fp32 handles only (the message is an fp32 format).  Not re-entrant: one host thread
 * at a time per group.
 * Message per shard (the step kernel writes its outputs straight into it; = distributed.OutputGather's format):
 *   obs[cmax, obs_dim] f32 | reward[cmax] f32 (AuvEnv only: the rigid-body reward is identically 0, 6DoF.py:575) | done[cmax] u8,
 *   padded to 16 B; cmax = the largest shard.  The root holds [n_shards][msg_bytes], twice (gather k overlaps step k + 1). */
typedef struct mvrl_group mvrl_group;
typedef struct mvrl_group_layout {
    int64_t n_global;
    int32_t n_shards, obs_dim, reward_plane;
    int32_t transport;      /* 0: device-to-device / peer copies, 1: RCCL (filled by mvrl_group_info) */
    int64_t cmax, off_reward, off_done, msg_bytes;
} mvrl_group_layout;
int mvrl_group_shard_range(int64_t n_global, int32_t shard, int32_t n_shards, int64_t* first, int64_t* count);   /* no GPU needed */
int mvrl_group_message_layout(int64_t n_global, int32_t n_shards, int32_t obs_dim, int32_t reward_plane, mvrl_group_layout* out); /* no GPU needed */
/* cfg: as for mvrl_create with n_envs = the GLOBAL env count (cfg->device is ignored); root = index INTO devices[] of the gather's target */
int mvrl_group_create(const mvrl_config* cfg, const int32_t* devices, int32_t n_devices, int32_t root, mvrl_group** out);
void mvrl_group_destroy(mvrl_group* g);
const char* mvrl_group_last_error(const mvrl_group* g);
int mvrl_group_info(const mvrl_group* g, mvrl_group_layout* out);
mvrl_handle* mvrl_group_shard(mvrl_group* g, int32_t shard);            /* the shard's handle: mvrl_get_state, mvrl_reset with explicit values ... */
int mvrl_group_set_flow(mvrl_group* g, const float* table_host, const mvrl_flow_desc* desc);   /* replicated on every device */
int mvrl_group_reset(mvrl_group* g);                                     /* random resets everywhere; first observations into the current message */
/* actions_dev[i]: pointer ON DEVICE i to shard i's rows [count_i, act_dim] f32; NULL (array or entry) = the group's own action
 * buffer of that shard (mvrl_group_scatter_actions_dev / mvrl_group_fill_actions); nothing is read with a fixed set-point */
int mvrl_group_step_dev(mvrl_group* g, const void* const* actions_dev);
int mvrl_group_gather_dev(mvrl_group* g);                               /* the last step's (or reset's) messages -> root; asynchronous */
int mvrl_group_wait(mvrl_group* g);                                     /* host blocks until the last gather has arrived */
int mvrl_group_gathered_event(mvrl_group* g, void** hip_event);         /* ... or order a consumer stream behind it: hipStreamWaitEvent */
/* root-device pointers to one shard's rows of the last gather (reward NULL for the rigid-body models); first / count: its global range */
int mvrl_group_root_views(mvrl_group* g, int32_t shard, const float** obs, const float** reward, const uint8_t** done, int64_t* first, int64_t* count);
int mvrl_group_download(mvrl_group* g, float* obs, float* reward, uint8_t* done);   /* last gather to host arrays in global env order */
int mvrl_group_scatter_actions_dev(mvrl_group* g, const float* actions_root_dev);   /* [n_global, act_dim] on the root device -> the shards */
int mvrl_group_fill_actions(mvrl_group* g, uint64_t seed, uint64_t counter, float lo, float hi);   /* synthetic roll-outs: uniform actions per shard */
int mvrl_group_synchronize(mvrl_group* g);

/* ---- run-time specialisation (6-DoF, fp32, RK4 harness).  libmvrl.so carries its fastest step kernel - model constants
 * as instruction literals - for the reference's default vehicle only (6DoF.py:83-218); a vehicle with other constants
 * (the reference is edited or subclassed for that) runs kernels that read them at run time, 1.13x (same structure) to 1.85x
 * (arbitrary constants) slower.  mvrl_specialize compiles the step kernel once more for THIS handle's constants - the
 * library carries its kernel sources; compiler: the ROCm installation's hipcc as a child process ($MVRL_HIPCC, else
 * /opt/rocm/bin/hipcc, else hipcc on PATH; 2-3 s), falling back to in-process hiprtc; MVRL_JIT_COMPILER=hipcc|hiprtc pins
 * it - and switches the handle to it: same arithmetic, same arguments, mvrl_variant() gains "jit-".  Structured constants
 * then run at the default vehicle's speed, arbitrary ones 1.3x slower than it.  No-op for handles that already run the
 * literal-constant kernel.  On failure (no compiler) returns MVRL_EHIP with the compiler's log in mvrl_last_error and the
 * handle keeps its ahead-of-time kernel.
 * mvrl_jit_compile_check performs the compilation alone (no GPU, nothing loaded): build and CI check; log_buf receives the
 * two kernel names and which compiler produced them. ---- */
int mvrl_specialize(mvrl_handle* h);
int mvrl_jit_compile_check(const mvrl_rov6_params* params, int control_mode, size_t* code_size, char* log_buf, size_t log_cap);
/* What mvrl_specialize built for this handle, read from the AMDGPU metadata note of the code object it loaded: which compiler
 * ("hipcc", "hiprtc"; "none" = the handle runs an ahead-of-time kernel), the register budget it was compiled for, and the worse of
 * the two instances' (turbulence off / on) register, spill and scratch figures (-1 = not in the note).  A build with spills
 * (the in-process hiprtc of a process that loaded an older comgr produces them: DESIGN.md section 5) runs 8-20 % slower than the
 * ahead-of-time kernel; MarineVecEnv warns about it.  mvrl_jit_compile_check2 = mvrl_jit_compile_check + that report (no GPU).
 * mvrl_jit_child_env: the environment the hipcc child process is started with, one KEY=value per line - the parent's minus
 * LD_PRELOAD / LD_AUDIT / ROCP_* / ROCPROFILER_* / HSA_TOOLS_* ..., so that a profiler attached to the host process does not
 * follow into the compiler. */
typedef struct mvrl_jit_report {
    int32_t specialized;        /* 1 = a run-time compiled kernel is in use */
    int32_t min_waves_per_simd; /* launch bound of the build that was kept (4, or 3 / 2 when the 128-VGPR build spilled) */
    int32_t vgprs, sgprs;
    int32_t vgpr_spills, sgpr_spills;
    int32_t scratch_bytes, lds_bytes;
    int64_t code_bytes;
    char compiler[16];
} mvrl_jit_report;
int mvrl_jit_info(const mvrl_handle* h, mvrl_jit_report* out);
int mvrl_jit_compile_check2(const mvrl_rov6_params* params, int control_mode, mvrl_jit_report* report, char* log_buf, size_t log_cap);
int mvrl_jit_child_env(char* buf, size_t cap);

/* ---- unit-level operators of the 6-DoF vehicle for n independent tuples (host arrays; NULL inputs / outputs are skipped), each
 * evaluated by the device functions the step kernel runs - the public methods example_trialTrajectories.py:100-134 and the
 * demos call on the vehicle object:
 *   axes[n,9]     rows iHat, jHat, kHat of updateMovingCoordSystem(angles)   6DoF.py:238-242  (globalToVehicle(v) = axes . v, :244-248)
 *   rpm_out[n,8]  allocateThrust() for generalisedControlForces = gcf[n,6] at `angles`   6DoF.py:220-231
 *   rhs[n,6]      forceModel(pos, angles, vel, rpm)[1]                        6DoF.py:253-404
 *   thruster_h[n,6]  the thruster column H of forceModel(..., retComp=True) = A . thrusterModel(limit(rpm))   :233-236, :271-282
 * with rpm = rpm_in[n,8] if given, else the allocation of gcf.  vel NULL = zero velocity.  (M itself is constant:
 * mvrl_rov6_params.mass.) ---- */
int mvrl_vehicle_ops(mvrl_handle* h, int64_t n, const float* angles, const float* gcf, const float* rpm_in, const float* vel,
                     float* axes, float* rpm_out, float* rhs, float* thruster_h);
int mvrl_vehicle_ops_f64(mvrl_handle* h, int64_t n, const double* angles, const double* gcf, const double* rpm_in, const double* vel,
                         double* axes, double* rpm_out, double* rhs, double* thruster_h);

/* forceModel(pos, angles, vel, rpms, retComp=True) (6DoF.py:253, :401-402): comp[n, 6, 5] = columns -Crb.vel, -Ca.vel, -D.vel, G, H
 * (H = A . thrusterModel(limit(rpm))) for n independent tuples; angles[n,3], vel[n,6], rpm_in[n,8]. */
int mvrl_force_components(mvrl_handle* h, int64_t n, const float* angles, const float* vel, const float* rpm_in, float* comp);
int mvrl_force_components_f64(mvrl_handle* h, int64_t n, const double* angles, const double* vel, const double* rpm_in, double* comp);

/* acc[n,6] = np.linalg.solve(M, RHS) for n right-hand sides rhs[n,6] (6DoF.py:428; the reference's own known answer is
 * example_temp.py:19-28), through the constant M^-1 the handle's step kernel applies (10 non-zeros for x_g = y_g = 0, dense otherwise). */
int mvrl_mass_solve(mvrl_handle* h, int64_t n, const float* rhs, float* acc);
int mvrl_mass_solve_f64(mvrl_handle* h, int64_t n, const double* rhs, double* acc);

/* Per-step side outputs the reference keeps in timeHistory (6DoF.py:578-587: F0..F5, u0..u7; 3DoF: F0..F2,u0..u3;
 * verySimpleAuv.py:389-403: Fx,Fy,N,u_current,v_current,rmsAc,r0..r4).  Enable BEFORE stepping;
 * aux row = [n_envs, aux_dim] f32 with aux_dim = 14 (ROV6) / 7 (ROV3) / 11 (AUV). */
int mvrl_enable_aux(mvrl_handle* h, int32_t enable);
int mvrl_get_aux(mvrl_handle* h, float* aux);
int mvrl_get_aux_f64(mvrl_handle* h, double* aux);
/* MVRL_INTEG_RK45 only: RHS evaluations each env spent in its last step (scipy's nfev). */
int mvrl_get_nfev(mvrl_handle* h, int32_t* nfev);

/* ---- stand-alone turbulence-field operators ----------------------------------------------------------- */
/* ReconstructedFlow.interp (flowGenerator.py:97-136) for n query points; table is host f32
 * [n_t][n_y][n_x][n_comp]; out [n, n_comp]. */
int mvrl_flow_interp(int32_t device, const float* table_host, const mvrl_flow_desc* desc, int32_t n_comp,
                     const float* t, const float* x, const float* y, int64_t n, float* out);
/* ReconstructedFlow.__init__ + scale (flowGenerator.py:19-23, 76-92): out[t,j,i,c] =
 * affine_c( sum_k Re(modes[j,i,c,k]*coeffs[k,t]) + ltm[j,i,c] ).  modes_re/modes_im [n_y*n_x*3, K],
 * coeffs_re/coeffs_im [K, n_t], ltm [n_y*n_x*3]; scale_mul/scale_add [3]; out host f32 [n_t, n_y*n_x*3]. */
int mvrl_flow_reconstruct(int32_t device, const float* modes_re, const float* modes_im, const float* coeffs_re,
                          const float* coeffs_im, const float* ltm, int32_t n_space3, int32_t n_modes, int32_t n_t,
                          const float* scale_mul, const float* scale_add, float* out);

/* ---- baseline policies of the reference, batched on the device -------------------------------
 * MVRL_POLICY_PD  : PDController.predict   tag/verySimpleAuv.py:22-50   (P, D gains on obs[:3], oldObs memory, optional noise)
 * MVRL_POLICY_LOS : LOSNavigation.predict + lineOfSight   3DoF.py:517-607   (obs = [p0(2), p1(2), psi_e], Rnav = 0.5)
 * obs [n_envs, obs_dim] f32, actions [n_envs, 3] f32. */
#define MVRL_POLICY_PD 0
#define MVRL_POLICY_LOS 1
typedef struct mvrl_policy mvrl_policy;
int mvrl_policy_create(int32_t kind, int32_t device, int64_t n_envs, int32_t obs_dim, double dt, const double* P, const double* D,
                       double noise_sigma, double r_nav, uint64_t seed, mvrl_policy** out);
void mvrl_policy_destroy(mvrl_policy* p);
int mvrl_policy_reset(mvrl_policy* p);
int mvrl_policy_predict(mvrl_policy* p, const float* obs, float* actions);
int mvrl_policy_predict_dev(mvrl_policy* p, const float* obs_dev, float* actions_dev, void* stream);

/* ---- learner-side consumer: CustomReplayBuffer.add with symmetry augmentation ----------------
 * tag/main_02_sbl_contrib_customBuffer.py:76-160.  All pointers are DEVICE pointers.  obs/next_obs [n_envs, 11],
 * actions [n_envs, 3], reward [n_envs] f32, done [n_envs] u8 as produced by mvrl_step_dev.  record_timeouts = 0 writes
 * `timeouts` = 0 - the reference's behaviour: its envs return info = {} (verySimpleAuv.py:410), so
 * info.get("TimeLimit.truncated", False) at :154 is always False and a time-limit `done` is a true terminal;
 * record_timeouts = 1 copies bit 1 of the done byte (time limit) into `timeouts` for learners that bootstrap through
 * truncations.  The ring buffers are [buffer_size, n_envs, dim].  Writes n_transforms (5, or 1 once the
 * buffer has rolled over more than twice, :143) consecutive slots starting at `pos`; the caller advances pos. */
int mvrl_replay_add_sym_dev(int32_t device, const float* obs, const float* next_obs, const float* actions, const float* reward,
                            const uint8_t* done, int64_t n_envs, float* buf_obs, float* buf_next_obs, float* buf_actions,
                            float* buf_reward, uint8_t* buf_done, uint8_t* buf_timeout, int64_t buffer_size, int64_t pos,
                            int32_t n_transforms, int32_t record_timeouts, void* stream);

/* ---- benchmark helpers -------------------------------------------------------------------------------- */
/* Whole episodes of the PD baseline in ONE launch - evaluate_agent(PDController(policy_dt, P, D), AuvEnv)
 * (tag/resources.py:49-102 driving tag/verySimpleAuv.py:22-50 and :264-410) for every env of the handle: from the env's
 * current state (call mvrl_reset first) a fresh noise-free PDController acts until `done` or n_steps; the env state
 * stays in registers between steps.  returns_dev[n] = sum of rewards, lengths_dev[n] = steps taken; the handle is left
 * in the terminal states (no auto-reset).  fp32 AuvEnv / AuvEnvCyl handles only. */
int mvrl_auv_pd_episodes_dev(mvrl_handle* h, const double* P, const double* D, double policy_dt, int32_t n_steps, float* returns_dev,
                             int32_t* lengths_dev, void* stream);

/* Fill a DEVICE buffer with uniform(lo,hi) f32 from the counter-based generator (key seed, stream `counter`). */
int mvrl_fill_uniform_dev(mvrl_handle* h, float* dst_dev, int64_t n, uint64_t seed, uint64_t counter, float lo,
                          float hi, void* stream);
/* HIP-event timing of step kernels launched through mvrl_step_dev on `stream`:
 * begin records an event, end records another, synchronises it and returns elapsed ms and launch count. */
int mvrl_timing_begin(mvrl_handle* h, void* stream);
/* Occupy `stream` for about `microseconds` (<= 10 000) with a single idle wave: phases independent chains of lane ranges
 * against each other without a cross-stream dependency (chains.ChainStepper.phase_delay). */
int mvrl_delay_dev(mvrl_handle* h, int32_t microseconds, void* stream);
/* step-kernel launches (env steps x lane ranges) enqueued through this handle since mvrl_create */
int mvrl_launch_count(mvrl_handle* h, int64_t* n_launches);
int mvrl_timing_end(mvrl_handle* h, void* stream, float* elapsed_ms, int64_t* n_launches);
/* device allocation helpers so that torch-free callers can own device buffers */
int mvrl_dev_alloc(mvrl_handle* h, size_t bytes, void** out_dev);
int mvrl_dev_free(mvrl_handle* h, void* dev);
int mvrl_dev_upload(mvrl_handle* h, void* dst_dev, const void* src_host, size_t bytes);
int mvrl_dev_download(mvrl_handle* h, void* dst_host, const void* src_dev, size_t bytes);
int mvrl_synchronize(mvrl_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* MVRL_H */
