// mvrl_abi.hip - the C ABI of libmvrl.so (include/mvrl.h): handle management, parameter narrowing,
// kernel-variant selection, host<->device staging.  No compute happens on the host and there is no CPU
// fallback: every entry point that needs the GPU fails with MVRL_ENODEV / MVRL_EHIP when HIP does.
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <new>
#include <string>

#include "../../include/mvrl.h"
#include "mvrl_kernels.hpp"

using namespace mvrl;

namespace {

thread_local std::string g_last_error;

struct Dims {
    int act, obs, init, words, aux;
};
bool model_dims(int model, Dims* d) {
    switch (model) {
        case MVRL_MODEL_AUV: *d = {3, 11, 16, 53, 11}; return true;
        case MVRL_MODEL_ROV3: *d = {3, 5, 5, 21, 7}; return true;
        case MVRL_MODEL_ROV6: *d = {6, 9, 9, 38, 14}; return true;
    }
    return false;
}

}  // namespace

struct mvrl_handle {
    mvrl_config cfg;
    Dims dims;
    int device;
    hipStream_t stream;
    float* state;
    Rov6Dev h6;
    Rov3Dev h3;
    AuvDev ha;
    void* params_dev;
    bool baked, sym;
    FlowDev flow;
    float* flow_owned;
    float *d_actions, *d_obs, *d_reward, *d_term_obs, *d_aux, *d_init;
    uint8_t *d_done, *d_mask;
    float *p_actions, *p_obs, *p_reward;  // pinned host staging
    uint8_t* p_done;
    uint32_t epoch;
    bool async_pending;
    bool aux_enabled;
    hipEvent_t ev0, ev1;
    int64_t launches;
    std::string err;
    char variant[64];
};

namespace {

int fail(mvrl_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_last_error = msg;
    return code;
}
#define HIP_TRY(h, call)                                                                         \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail(h, e_ == hipErrorOutOfMemory ? MVRL_ENOMEM : MVRL_EHIP,                  \
                        std::string(#call) + ": " + hipGetErrorString(e_));                      \
    } while (0)

// ---- fp64 host params -> fp32 device params (same arithmetic as marinevehiclereinforcementlearning_amd/devparams.py)
bool sym_layout(const mvrl_rov6_params& p, double* sa, double* sb, double tol = 1e-9) {
    const double* A = p.alloc;
    const double* Ai = p.alloc_inv;
    sa[0] = fabs(A[0]); sa[1] = fabs(A[8]); sa[2] = fabs(A[16 + 4]); sa[3] = fabs(A[24]); sa[4] = fabs(A[24 + 4]);
    sa[5] = fabs(A[32]); sa[6] = fabs(A[32 + 4]); sa[7] = fabs(A[40]);
    sb[0] = fabs(Ai[0]); sb[1] = fabs(Ai[1]); sb[2] = fabs(Ai[5]);
    sb[3] = fabs(Ai[24]); sb[4] = fabs(Ai[25]); sb[5] = fabs(Ai[26]); sb[6] = fabs(Ai[27]); sb[7] = fabs(Ai[28]);
    const double pA[4] = {1, 1, -1, -1}, pB[4] = {-1, 1, -1, 1}, pC[4] = {-1, 1, 1, -1}, vB[4] = {-1, -1, 1, 1}, vC[4] = {1, -1, 1, -1};
    double Ar[48] = {0}, Air[48] = {0};
    for (int i = 0; i < 4; i++) {
        Ar[0 + i] = sa[0] * pA[i]; Ar[8 + i] = sa[1] * pB[i]; Ar[16 + 4 + i] = sa[2] * pC[i];
        Ar[24 + i] = -sa[3] * pB[i]; Ar[24 + 4 + i] = sa[4] * vB[i];
        Ar[32 + i] = sa[5] * pA[i]; Ar[32 + 4 + i] = sa[6] * vC[i]; Ar[40 + i] = sa[7] * pC[i];
        Air[6 * i + 0] = sb[0] * pA[i]; Air[6 * i + 1] = sb[1] * pB[i]; Air[6 * i + 5] = sb[2] * pC[i];
        Air[6 * (4 + i) + 0] = sb[3] * pB[i]; Air[6 * (4 + i) + 1] = sb[4] * vB[i]; Air[6 * (4 + i) + 2] = sb[5] * pC[i];
        Air[6 * (4 + i) + 3] = sb[6] * vB[i]; Air[6 * (4 + i) + 4] = sb[7] * vC[i];
    }
    double ma = 1.0, mi = 1.0, da = 0, di = 0;
    for (int k = 0; k < 48; k++) {
        ma = fmax(ma, fabs(A[k])); mi = fmax(mi, fabs(Ai[k]));
        da = fmax(da, fabs(A[k] - Ar[k])); di = fmax(di, fabs(Ai[k] - Air[k]));
    }
    return da <= tol * ma && di <= tol * mi;
}

bool rov6_structured(const mvrl_rov6_params& p, double tol = 1e-12) {
    if (fabs(p.cg[0]) > tol || fabs(p.cg[1]) > tol || fabs(p.cb[0]) > tol || fabs(p.cb[1]) > tol) return false;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            if (i != j && fabs(p.inertia[3 * i + j]) > tol) return false;
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            if (i == j || (i == 4 && j == 2)) continue;
            if (fabs(p.dlin[6 * i + j]) > tol || fabs(p.dquad[6 * i + j]) > tol) return false;
        }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            bool allowed = (i == j) || (i == 0 && j == 4) || (i == 4 && j == 0) || (i == 1 && j == 3) || (i == 3 && j == 1);
            if (!allowed && fabs(p.minv[6 * i + j]) > tol) return false;
        }
    double sa[8], sb[8];
    return sym_layout(p, sa, sb);
}

void to_dev(const mvrl_rov6_params& p, Rov6Dev* d) {
    memset(d, 0, sizeof(*d));
    const double W = p.weight, B = p.buoyancy;
    d->m = (float)p.m;
    d->wb = (float)(W - B);
    for (int i = 0; i < 3; i++) { d->cg[i] = (float)p.cg[i]; d->gw[i] = (float)(p.cg[i] * W - p.cb[i] * B); }
    for (int i = 0; i < 9; i++) d->I[i] = (float)p.inertia[i];
    for (int i = 0; i < 6; i++) d->added[i] = (float)p.added[i];
    for (int i = 0; i < 36; i++) { d->minv[i] = (float)p.minv[i]; d->dlin[i] = (float)p.dlin[i]; d->dquad[i] = (float)p.dquad[i]; }
    for (int i = 0; i < 48; i++) { d->A[i] = (float)p.alloc[i]; d->Ainv[i] = (float)p.alloc_inv[i]; }
    double sa[8], sb[8];
    if (sym_layout(p, sa, sb)) {
        for (int i = 0; i < 8; i++) { d->sym_a[i] = (float)sa[i]; d->sym_ainv[i] = (float)sb[i]; }
    }
    const double k = p.thrust_k;
    d->thrust_k = (float)k;
    d->inv_thrust_k = (float)(1.0 / k);
    d->rpm_max = (float)p.rpm_max;
    d->rpm_dead = (float)p.rpm_deadband;
    d->f_max = (float)(k * pow(p.rpm_max / 60., 2));
    d->f_dead = (float)(k * pow(p.rpm_deadband / 60., 2));
    for (int i = 0; i < 6; i++) {
        d->kp[i] = (float)p.kp[i]; d->ki[i] = (float)p.ki[i]; d->kd[i] = (float)p.kd[i];
        d->windup[i] = (float)p.windup[i]; d->umax[i] = (float)p.umax[i]; d->act_scale[i] = (float)p.act_scale[i];
    }
    d->inv_obs_pos = (float)(1.0 / p.obs_pos_scale);
    d->inv_obs_ang = (float)(1.0 / p.obs_ang_scale);
}

void to_dev(const mvrl_rov3_params& p, Rov3Dev* d) {
    memset(d, 0, sizeof(*d));
    d->m = (float)p.m; d->cgx = (float)p.cg[0]; d->cgy = (float)p.cg[1];
    d->xud = (float)p.added[0]; d->yvd = (float)p.added[1];
    for (int i = 0; i < 9; i++) { d->minv[i] = (float)p.minv[i]; d->dlin[i] = (float)p.dlin[i]; d->dquad[i] = (float)p.dquad[i]; }
    for (int i = 0; i < 12; i++) d->Ainv[i] = (float)p.alloc_inv[i];
    const double k = p.thrust_k;
    d->thrust_k = (float)k; d->inv_thrust_k = (float)(1.0 / k);
    d->rpm_max = (float)p.rpm_max; d->rpm_dead = (float)p.rpm_deadband;
    d->f_max = (float)(k * pow(p.rpm_max / 60., 2));
    d->f_dead = (float)(k * pow(p.rpm_deadband / 60., 2));
    d->cos_a = (float)p.cos_alpha; d->sin_a = (float)p.sin_alpha; d->yaw_arm = (float)p.yaw_arm;
    d->inv_jet_area_k = (float)(1.0 / p.jet_area_k);
    d->jet_c1 = (float)p.jet_c1; d->jet_k1 = (float)p.jet_k1; d->jet_c2 = (float)p.jet_c2; d->jet_k2 = (float)p.jet_k2;
    d->jet_drag_k = (float)p.jet_drag_k;
    for (int i = 0; i < 3; i++) {
        d->kp[i] = (float)p.kp[i]; d->ki[i] = (float)p.ki[i]; d->kd[i] = (float)p.kd[i];
        d->windup[i] = (float)p.windup[i]; d->umax[i] = (float)p.umax[i]; d->act_scale[i] = (float)p.act_scale[i];
    }
    d->inv_obs_pos = (float)(1.0 / p.obs_pos_scale);
    d->inv_obs_ang = (float)(1.0 / p.obs_ang_scale);
}

void to_dev(const mvrl_auv_params& p, AuvDev* d) {
    memset(d, 0, sizeof(*d));
    d->m = (float)p.m; d->izz = (float)p.izz; d->xuu = (float)p.xuu; d->yvv = (float)p.yvv; d->nrr = (float)p.nrr;
    d->xu = (float)p.xu; d->yv = (float)p.yv; d->nr = (float)p.nr;
    d->max_force = (float)p.max_force; d->max_moment = (float)p.max_moment;
    d->x_min = (float)p.x_min; d->x_max = (float)p.x_max; d->y_min = (float)p.y_min; d->y_max = (float)p.y_max;
    d->noise_coeffs = (float)p.noise_mag_coeffs; d->noise_act = (float)p.noise_mag_actuation;
    d->stop_on_bounds = p.stop_on_bounds;
}

int check_flow_desc(mvrl_handle* h, const mvrl_flow_desc* d) {
    if (!d || d->n_t < 2 || d->n_y < 2 || d->n_x < 2 || !(d->dt > 0) || !(d->dx > 0) || !(d->dy > 0))
        return fail(h, MVRL_EINVAL, "flow descriptor: need n_t,n_y,n_x >= 2 and positive spacings");
    return MVRL_OK;
}

void set_flow_dev(mvrl_handle* h, const float* table_dev, const mvrl_flow_desc* d) {
    h->flow.table = reinterpret_cast<const float2*>(table_dev);
    h->flow.n_t = d->n_t; h->flow.n_y = d->n_y; h->flow.n_x = d->n_x;
    h->flow.inv_dt = (float)(1.0 / d->dt); h->flow.inv_dx = (float)(1.0 / d->dx); h->flow.inv_dy = (float)(1.0 / d->dy);
    h->flow.t_quarter = (float)((d->n_t / 4) * d->dt);  // flow.time[nT // 4] (verySimpleAuv.py:245)
}

StepIO make_io(mvrl_handle* h, const float* actions, float* obs, float* reward, uint8_t* done) {
    StepIO io;
    io.state = h->state; io.actions = actions; io.obs = obs; io.reward = reward; io.done = done;
    io.term_obs = h->cfg.auto_reset ? h->d_term_obs : nullptr;
    io.aux = h->aux_enabled ? h->d_aux : nullptr;
    io.n = h->cfg.n_envs; io.env_offset = h->cfg.env_offset; io.seed = h->cfg.seed; io.epoch = ++h->epoch;
    io.n_sub = h->cfg.n_substeps; io.max_steps = h->cfg.max_steps; io.fixed_sp = h->cfg.fixed_setpoint;
    io.auto_reset = h->cfg.auto_reset; io.dt = (float)h->cfg.dt;
    return io;
}

int launch_step(mvrl_handle* h, const StepIO& io, hipStream_t s) {
    const bool flow = h->cfg.use_flow != 0;
    if (flow && !h->flow.table) return fail(h, MVRL_ESTATE, "use_flow = 1 but no turbulence table was set (mvrl_set_flow)");
    const bool zoh = h->cfg.control_mode == MVRL_CTRL_ZOH;
    hipError_t e;
    switch (h->cfg.model) {
        case MVRL_MODEL_ROV6:
            e = launch_rov6_step((const Rov6Dev*)h->params_dev, io, h->flow, h->baked, h->sym, zoh, flow, s);
            break;
        case MVRL_MODEL_ROV3:
            e = launch_rov3_step((const Rov3Dev*)h->params_dev, io, h->flow, h->baked, zoh, flow, s);
            break;
        default:
            e = launch_auv_step(h->ha, io, h->flow, flow, s);
    }
    if (e != hipSuccess) return fail(h, MVRL_EHIP, std::string("step kernel launch: ") + hipGetErrorString(e));
    h->launches++;
    return MVRL_OK;
}

int launch_reset(mvrl_handle* h, const uint8_t* mask_dev, const float* init_dev, float* obs_dev, hipStream_t s) {
    const uint32_t epoch = ++h->epoch;
    hipError_t e;
    const int64_t n = h->cfg.n_envs;
    switch (h->cfg.model) {
        case MVRL_MODEL_ROV6:
            e = launch_rov6_reset((const Rov6Dev*)h->params_dev, h->state, n, mask_dev, init_dev, obs_dev, h->cfg.seed,
                                  h->cfg.env_offset, epoch, h->flow.t_quarter, s);
            break;
        case MVRL_MODEL_ROV3:
            e = launch_rov3_reset((const Rov3Dev*)h->params_dev, h->state, n, mask_dev, init_dev, obs_dev, h->cfg.seed,
                                  h->cfg.env_offset, epoch, h->flow.t_quarter, s);
            break;
        default:
            e = launch_auv_reset(h->ha, h->state, n, mask_dev, init_dev, obs_dev, h->cfg.seed, h->cfg.env_offset, epoch,
                                 h->flow.t_quarter, s);
    }
    if (e != hipSuccess) return fail(h, MVRL_EHIP, std::string("reset kernel launch: ") + hipGetErrorString(e));
    return MVRL_OK;
}

int use_device(mvrl_handle* h) {
    HIP_TRY(h, hipSetDevice(h->device));
    return MVRL_OK;
}

}  // namespace

extern "C" {

int mvrl_abi_version(void) { return MVRL_ABI_VERSION; }

int mvrl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* mvrl_last_error(const mvrl_handle* h) { return h ? h->err.c_str() : g_last_error.c_str(); }

const char* mvrl_variant(const mvrl_handle* h) { return h ? h->variant : ""; }

int mvrl_model_dims(int32_t model, int32_t* act_dim, int32_t* obs_dim, int32_t* init_dim, int32_t* state_words) {
    Dims d;
    if (!model_dims(model, &d)) return fail(nullptr, MVRL_EINVAL, "unknown model");
    if (act_dim) *act_dim = d.act;
    if (obs_dim) *obs_dim = d.obs;
    if (init_dim) *init_dim = d.init;
    if (state_words) *state_words = d.words;
    return MVRL_OK;
}

int mvrl_aux_dim(int32_t model) {
    Dims d;
    return model_dims(model, &d) ? d.aux : MVRL_EINVAL;
}

int mvrl_create(const mvrl_config* cfg, mvrl_handle** out) {
    if (!cfg || !out) return fail(nullptr, MVRL_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->abi_version != MVRL_ABI_VERSION) return fail(nullptr, MVRL_EINVAL, "ABI version mismatch");
    Dims d;
    if (!model_dims(cfg->model, &d)) return fail(nullptr, MVRL_EINVAL, "unknown model");
    if (cfg->n_envs < 1) return fail(nullptr, MVRL_EINVAL, "n_envs must be >= 1");
    if ((int64_t)d.words * cfg->n_envs >= ((int64_t)1 << 30))
        return fail(nullptr, MVRL_EINVAL, "n_envs too large for 32-bit state offsets (state words * n_envs must be < 2^30)");
    if (cfg->model != MVRL_MODEL_AUV && cfg->n_substeps < 1) return fail(nullptr, MVRL_EINVAL, "n_substeps must be >= 1");
    if (!(cfg->dt > 0)) return fail(nullptr, MVRL_EINVAL, "dt must be > 0");
    if (cfg->max_steps < 1) return fail(nullptr, MVRL_EINVAL, "max_steps must be >= 1");
    if (cfg->control_mode != MVRL_CTRL_FAITHFUL && cfg->control_mode != MVRL_CTRL_ZOH)
        return fail(nullptr, MVRL_EINVAL, "unknown control_mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, MVRL_ENODEV, "no HIP device available (libmvrl has no CPU fallback)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, MVRL_ENODEV, "device ordinal out of range");

    mvrl_handle* h = new (std::nothrow) mvrl_handle();
    if (!h) return fail(nullptr, MVRL_ENOMEM, "host allocation failed");
    h->cfg = *cfg;
    h->dims = d;
    h->device = cfg->device;
    h->epoch = 0;
    h->launches = 0;
    memset(&h->flow, 0, sizeof(h->flow));
#define CREATE_TRY(call)                                                                       \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            int rc_ = fail(nullptr, e_ == hipErrorOutOfMemory ? MVRL_ENOMEM : MVRL_EHIP,      \
                           std::string(#call) + ": " + hipGetErrorString(e_));                 \
            mvrl_destroy(h);                                                                   \
            return rc_;                                                                        \
        }                                                                                      \
    } while (0)
    CREATE_TRY(hipSetDevice(h->device));
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_TRY(hipEventCreate(&h->ev0));
    CREATE_TRY(hipEventCreate(&h->ev1));
    const size_t n = (size_t)cfg->n_envs;
    CREATE_TRY(hipMalloc(&h->state, n * d.words * sizeof(float)));
    CREATE_TRY(hipMemsetAsync(h->state, 0, n * d.words * sizeof(float), h->stream));
    CREATE_TRY(hipMalloc(&h->d_actions, n * d.act * sizeof(float)));
    CREATE_TRY(hipMalloc(&h->d_obs, n * d.obs * sizeof(float)));
    CREATE_TRY(hipMalloc(&h->d_term_obs, n * d.obs * sizeof(float)));
    CREATE_TRY(hipMemsetAsync(h->d_term_obs, 0, n * d.obs * sizeof(float), h->stream));
    CREATE_TRY(hipMalloc(&h->d_reward, n * sizeof(float)));
    CREATE_TRY(hipMalloc(&h->d_done, n));
    CREATE_TRY(hipMalloc(&h->d_mask, n));
    CREATE_TRY(hipMalloc(&h->d_init, n * d.init * sizeof(float)));
    CREATE_TRY(hipHostMalloc(&h->p_actions, n * d.act * sizeof(float)));
    CREATE_TRY(hipHostMalloc(&h->p_obs, n * d.obs * sizeof(float)));
    CREATE_TRY(hipHostMalloc(&h->p_reward, n * sizeof(float)));
    CREATE_TRY(hipHostMalloc(&h->p_done, n));

    h->baked = false;
    h->sym = false;
    if (cfg->model == MVRL_MODEL_ROV6) {
        to_dev(cfg->rov6, &h->h6);
        h->sym = rov6_structured(cfg->rov6);
        h->baked = h->sym && memcmp(&h->h6, &kRov6Default, sizeof(Rov6Dev)) == 0;
        CREATE_TRY(hipMalloc(&h->params_dev, sizeof(Rov6Dev)));
        CREATE_TRY(hipMemcpyAsync(h->params_dev, &h->h6, sizeof(Rov6Dev), hipMemcpyHostToDevice, h->stream));
    } else if (cfg->model == MVRL_MODEL_ROV3) {
        to_dev(cfg->rov3, &h->h3);
        h->baked = memcmp(&h->h3, &kRov3Default, sizeof(Rov3Dev)) == 0;
        CREATE_TRY(hipMalloc(&h->params_dev, sizeof(Rov3Dev)));
        CREATE_TRY(hipMemcpyAsync(h->params_dev, &h->h3, sizeof(Rov3Dev), hipMemcpyHostToDevice, h->stream));
    } else {
        to_dev(cfg->auv, &h->ha);
    }
    snprintf(h->variant, sizeof(h->variant), "%s/%s/%s%s",
             cfg->model == MVRL_MODEL_ROV6 ? "rov6" : (cfg->model == MVRL_MODEL_ROV3 ? "rov3" : "auv"),
             h->baked ? "baked" : (h->sym ? "sym" : "generic"),
             cfg->control_mode == MVRL_CTRL_ZOH ? "zoh" : "faithful", cfg->use_flow ? "+flow" : "");
    CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
    *out = h;
    return MVRL_OK;
}

void mvrl_destroy(mvrl_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void* dev[] = {h->state, h->params_dev, h->flow_owned, h->d_actions, h->d_obs, h->d_reward, h->d_term_obs, h->d_aux,
                   h->d_init, h->d_done, h->d_mask};
    for (void* p : dev)
        if (p) (void)hipFree(p);
    void* pin[] = {h->p_actions, h->p_obs, h->p_reward, h->p_done};
    for (void* p : pin)
        if (p) (void)hipHostFree(p);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int mvrl_set_flow(mvrl_handle* h, const float* table_host, const mvrl_flow_desc* desc) {
    if (!h || !table_host) return fail(h, MVRL_EINVAL, "null argument");
    int rc = check_flow_desc(h, desc);
    if (rc) return rc;
    if ((rc = use_device(h))) return rc;
    const size_t bytes = (size_t)desc->n_t * desc->n_y * desc->n_x * 2 * sizeof(float);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->flow_owned) { (void)hipFree(h->flow_owned); h->flow_owned = nullptr; h->flow.table = nullptr; }
    HIP_TRY(h, hipMalloc(&h->flow_owned, bytes));
    HIP_TRY(h, hipMemcpy(h->flow_owned, table_host, bytes, hipMemcpyHostToDevice));
    set_flow_dev(h, h->flow_owned, desc);
    return MVRL_OK;
}

int mvrl_set_flow_dev(mvrl_handle* h, const float* table_dev, const mvrl_flow_desc* desc) {
    if (!h || !table_dev) return fail(h, MVRL_EINVAL, "null argument");
    int rc = check_flow_desc(h, desc);
    if (rc) return rc;
    if ((rc = use_device(h))) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->flow_owned) { (void)hipFree(h->flow_owned); h->flow_owned = nullptr; }
    set_flow_dev(h, table_dev, desc);
    return MVRL_OK;
}

int mvrl_reset_dev(mvrl_handle* h, const uint8_t* mask_dev, const float* init_dev, float* obs_dev, void* stream) {
    if (!h) return fail(h, MVRL_EINVAL, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    return launch_reset(h, mask_dev, init_dev, obs_dev, stream ? (hipStream_t)stream : h->stream);
}

int mvrl_reset(mvrl_handle* h, const uint8_t* mask, const float* init, float* obs) {
    if (!h) return fail(h, MVRL_EINVAL, "null handle");
    if (h->async_pending) return fail(h, MVRL_ESTATE, "reset while a step_async is pending");
    int rc = use_device(h);
    if (rc) return rc;
    const size_t n = (size_t)h->cfg.n_envs;
    if (mask) HIP_TRY(h, hipMemcpyAsync(h->d_mask, mask, n, hipMemcpyHostToDevice, h->stream));
    if (init) HIP_TRY(h, hipMemcpyAsync(h->d_init, init, n * h->dims.init * sizeof(float), hipMemcpyHostToDevice, h->stream));
    if (obs && mask)  // rows of envs that are not reset must keep the caller's values
        HIP_TRY(h, hipMemcpyAsync(h->d_obs, obs, n * h->dims.obs * sizeof(float), hipMemcpyHostToDevice, h->stream));
    rc = launch_reset(h, mask ? h->d_mask : nullptr, init ? h->d_init : nullptr, obs ? h->d_obs : nullptr, h->stream);
    if (rc) return rc;
    if (obs) HIP_TRY(h, hipMemcpyAsync(obs, h->d_obs, n * h->dims.obs * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MVRL_OK;
}

int mvrl_step_dev(mvrl_handle* h, const float* actions_dev, float* obs_dev, float* reward_dev, uint8_t* done_dev,
                  void* stream) {
    if (!h || !obs_dev || !reward_dev || !done_dev) return fail(h, MVRL_EINVAL, "null argument");
    if (!actions_dev && !h->cfg.fixed_setpoint) return fail(h, MVRL_EINVAL, "actions required unless fixed_setpoint");
    int rc = use_device(h);
    if (rc) return rc;
    StepIO io = make_io(h, actions_dev, obs_dev, reward_dev, done_dev);
    return launch_step(h, io, stream ? (hipStream_t)stream : h->stream);
}

int mvrl_step_async(mvrl_handle* h, const float* actions) {
    if (!h) return fail(h, MVRL_EINVAL, "null handle");
    if (h->async_pending) return fail(h, MVRL_ESTATE, "step_async called twice without step_wait");
    if (!actions && !h->cfg.fixed_setpoint) return fail(h, MVRL_EINVAL, "actions required unless fixed_setpoint");
    int rc = use_device(h);
    if (rc) return rc;
    const size_t n = (size_t)h->cfg.n_envs;
    if (actions) {
        memcpy(h->p_actions, actions, n * h->dims.act * sizeof(float));  // caller's buffer is free after return
        HIP_TRY(h, hipMemcpyAsync(h->d_actions, h->p_actions, n * h->dims.act * sizeof(float), hipMemcpyHostToDevice, h->stream));
    }
    StepIO io = make_io(h, h->d_actions, h->d_obs, h->d_reward, h->d_done);
    rc = launch_step(h, io, h->stream);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->p_obs, h->d_obs, n * h->dims.obs * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->p_reward, h->d_reward, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->p_done, h->d_done, n, hipMemcpyDeviceToHost, h->stream));
    h->async_pending = true;
    return MVRL_OK;
}

int mvrl_step_wait(mvrl_handle* h, float* obs, float* reward, uint8_t* done) {
    if (!h) return fail(h, MVRL_EINVAL, "null handle");
    if (!h->async_pending) return fail(h, MVRL_ESTATE, "step_wait without step_async");
    int rc = use_device(h);
    if (rc) return rc;
    h->async_pending = false;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t n = (size_t)h->cfg.n_envs;
    if (obs) memcpy(obs, h->p_obs, n * h->dims.obs * sizeof(float));
    if (reward) memcpy(reward, h->p_reward, n * sizeof(float));
    if (done) memcpy(done, h->p_done, n);
    return MVRL_OK;
}

int mvrl_step(mvrl_handle* h, const float* actions, float* obs, float* reward, uint8_t* done) {
    int rc = mvrl_step_async(h, actions);
    if (rc) return rc;
    return mvrl_step_wait(h, obs, reward, done);
}

int mvrl_get_terminal_obs_dev(mvrl_handle* h, float* obs_dev, void* stream) {
    if (!h || !obs_dev) return fail(h, MVRL_EINVAL, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(obs_dev, h->d_term_obs, (size_t)h->cfg.n_envs * h->dims.obs * sizeof(float),
                              hipMemcpyDeviceToDevice, stream ? (hipStream_t)stream : h->stream));
    return MVRL_OK;
}

int mvrl_get_terminal_obs(mvrl_handle* h, float* obs) {
    if (!h || !obs) return fail(h, MVRL_EINVAL, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(obs, h->d_term_obs, (size_t)h->cfg.n_envs * h->dims.obs * sizeof(float), hipMemcpyDeviceToHost,
                              h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MVRL_OK;
}

int mvrl_get_state(mvrl_handle* h, float* buf, size_t n_floats) {
    if (!h || !buf) return fail(h, MVRL_EINVAL, "null argument");
    const size_t need = (size_t)h->cfg.n_envs * h->dims.words;
    if (n_floats != need) return fail(h, MVRL_EINVAL, "state buffer must hold state_words * n_envs floats");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(buf, h->state, need * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MVRL_OK;
}

int mvrl_set_state(mvrl_handle* h, const float* buf, size_t n_floats) {
    if (!h || !buf) return fail(h, MVRL_EINVAL, "null argument");
    const size_t need = (size_t)h->cfg.n_envs * h->dims.words;
    if (n_floats != need) return fail(h, MVRL_EINVAL, "state buffer must hold state_words * n_envs floats");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(h->state, buf, need * sizeof(float), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MVRL_OK;
}

int mvrl_enable_aux(mvrl_handle* h, int32_t enable) {
    if (!h) return fail(h, MVRL_EINVAL, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    if (enable && !h->d_aux) {
        HIP_TRY(h, hipMalloc(&h->d_aux, (size_t)h->cfg.n_envs * h->dims.aux * sizeof(float)));
        HIP_TRY(h, hipMemsetAsync(h->d_aux, 0, (size_t)h->cfg.n_envs * h->dims.aux * sizeof(float), h->stream));
    }
    h->aux_enabled = enable != 0;
    return MVRL_OK;
}

int mvrl_get_aux(mvrl_handle* h, float* aux) {
    if (!h || !aux) return fail(h, MVRL_EINVAL, "null argument");
    if (!h->d_aux) return fail(h, MVRL_ESTATE, "aux outputs were never enabled (mvrl_enable_aux)");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(aux, h->d_aux, (size_t)h->cfg.n_envs * h->dims.aux * sizeof(float), hipMemcpyDeviceToHost,
                              h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MVRL_OK;
}

int mvrl_flow_interp(int32_t device, const float* table_host, const mvrl_flow_desc* desc, int32_t n_comp, const float* t,
                     const float* x, const float* y, int64_t n, float* out) {
    if (!table_host || !t || !x || !y || !out || n < 0 || n_comp < 1 || n_comp > 4) return fail(nullptr, MVRL_EINVAL, "bad argument");
    int rc = check_flow_desc(nullptr, desc);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(nullptr, MVRL_ENODEV, "no such HIP device");
    if (n == 0) return MVRL_OK;
    HIP_TRY(nullptr, hipSetDevice(device));
    const size_t tb = (size_t)desc->n_t * desc->n_y * desc->n_x * n_comp * sizeof(float);
    float *d_tab = nullptr, *d_q = nullptr, *d_out = nullptr;
    hipError_t e = hipMalloc(&d_tab, tb);
    if (e == hipSuccess) e = hipMalloc(&d_q, (size_t)n * 3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d_out, (size_t)n * n_comp * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_tab, table_host, tb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_q, t, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_q + n, x, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_q + 2 * n, y, (size_t)n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launch_flow_interp(d_tab, desc->n_t, desc->n_y, desc->n_x, n_comp, (float)(1.0 / desc->dt), (float)(1.0 / desc->dx),
                               (float)(1.0 / desc->dy), d_q, d_q + n, d_q + 2 * n, n, d_out, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d_out, (size_t)n * n_comp * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_tab); (void)hipFree(d_q); (void)hipFree(d_out);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorOutOfMemory ? MVRL_ENOMEM : MVRL_EHIP, std::string("flow_interp: ") + hipGetErrorString(e));
    return MVRL_OK;
}

int mvrl_flow_reconstruct(int32_t device, const float* modes_re, const float* modes_im, const float* coeffs_re,
                          const float* coeffs_im, const float* ltm, int32_t n_space3, int32_t n_modes, int32_t n_t,
                          const float* scale_mul, const float* scale_add, float* out) {
    if (!modes_re || !modes_im || !coeffs_re || !coeffs_im || !ltm || !scale_mul || !scale_add || !out || n_space3 < 1 ||
        n_modes < 1 || n_t < 1 || n_space3 % 3 != 0 || n_modes > 8192)
        return fail(nullptr, MVRL_EINVAL, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(nullptr, MVRL_ENODEV, "no such HIP device");
    HIP_TRY(nullptr, hipSetDevice(device));
    const size_t mb = (size_t)n_space3 * n_modes * sizeof(float), cb = (size_t)n_modes * n_t * sizeof(float);
    const size_t ob = (size_t)n_t * n_space3 * sizeof(float);
    float *d_mr = nullptr, *d_mi = nullptr, *d_cr = nullptr, *d_ci = nullptr, *d_l = nullptr, *d_o = nullptr;
    hipError_t e = hipMalloc(&d_mr, mb);
    if (e == hipSuccess) e = hipMalloc(&d_mi, mb);
    if (e == hipSuccess) e = hipMalloc(&d_cr, cb);
    if (e == hipSuccess) e = hipMalloc(&d_ci, cb);
    if (e == hipSuccess) e = hipMalloc(&d_l, (size_t)n_space3 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&d_o, ob);
    if (e == hipSuccess) e = hipMemcpy(d_mr, modes_re, mb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_mi, modes_im, mb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_cr, coeffs_re, cb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_ci, coeffs_im, cb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_l, ltm, (size_t)n_space3 * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_flow_reconstruct(d_mr, d_mi, d_cr, d_ci, d_l, n_space3, n_modes, n_t, scale_mul, scale_add, d_o, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out, d_o, ob, hipMemcpyDeviceToHost);
    (void)hipFree(d_mr); (void)hipFree(d_mi); (void)hipFree(d_cr); (void)hipFree(d_ci); (void)hipFree(d_l); (void)hipFree(d_o);
    if (e != hipSuccess) return fail(nullptr, e == hipErrorOutOfMemory ? MVRL_ENOMEM : MVRL_EHIP, std::string("flow_reconstruct: ") + hipGetErrorString(e));
    return MVRL_OK;
}

int mvrl_fill_uniform_dev(mvrl_handle* h, float* dst_dev, int64_t n, uint64_t seed, uint64_t counter, float lo, float hi,
                          void* stream) {
    if (!h || !dst_dev || n < 0) return fail(h, MVRL_EINVAL, "bad argument");
    int rc = use_device(h);
    if (rc) return rc;
    if (n == 0) return MVRL_OK;
    hipError_t e = launch_fill_uniform(dst_dev, n, seed, counter, lo, hi, stream ? (hipStream_t)stream : h->stream);
    if (e != hipSuccess) return fail(h, MVRL_EHIP, std::string("fill_uniform: ") + hipGetErrorString(e));
    return MVRL_OK;
}

int mvrl_timing_begin(mvrl_handle* h, void* stream) {
    if (!h) return fail(h, MVRL_EINVAL, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    h->launches = 0;
    HIP_TRY(h, hipEventRecord(h->ev0, stream ? (hipStream_t)stream : h->stream));
    return MVRL_OK;
}

int mvrl_timing_end(mvrl_handle* h, void* stream, float* elapsed_ms, int64_t* n_launches) {
    if (!h) return fail(h, MVRL_EINVAL, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipEventRecord(h->ev1, stream ? (hipStream_t)stream : h->stream));
    HIP_TRY(h, hipEventSynchronize(h->ev1));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    if (elapsed_ms) *elapsed_ms = ms;
    if (n_launches) *n_launches = h->launches;
    return MVRL_OK;
}

int mvrl_dev_alloc(mvrl_handle* h, size_t bytes, void** out_dev) {
    if (!h || !out_dev) return fail(h, MVRL_EINVAL, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipMalloc(out_dev, bytes));
    return MVRL_OK;
}
int mvrl_dev_free(mvrl_handle* h, void* dev) {
    if (!h) return fail(h, MVRL_EINVAL, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipFree(dev));
    return MVRL_OK;
}
int mvrl_dev_upload(mvrl_handle* h, void* dst_dev, const void* src_host, size_t bytes) {
    if (!h || !dst_dev || !src_host) return fail(h, MVRL_EINVAL, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MVRL_OK;
}
int mvrl_dev_download(mvrl_handle* h, void* dst_host, const void* src_dev, size_t bytes) {
    if (!h || !dst_host || !src_dev) return fail(h, MVRL_EINVAL, "null argument");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MVRL_OK;
}
int mvrl_synchronize(mvrl_handle* h) {
    if (!h) return fail(h, MVRL_EINVAL, "null handle");
    int rc = use_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return MVRL_OK;
}

}  // extern "C"
