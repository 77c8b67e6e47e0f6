// mvrl_group.hip - several GPUs of one node behind ONE object in ONE host process (SURVEY.md 8(b) "multi-GPU handled inside one
// handle", 8(e) "all driven from one host process"; BASELINE configs[4]).
//
// Replaces SB3's SubprocVecEnv (tag/main_00_sbl.py:145-146: one Python process per env, pipe send / recv around env.step) for a C
// caller: the batch is cut into contiguous shards (the same partition as distributed.shard_range), each shard is an ordinary
// mvrl_handle on its device with env_offset = its first global env (random resets are Philox-keyed by the GLOBAL env id, so the
// shards together are bit-identical to the unsharded batch), a step is one launch per device with no host synchronisation between
// them, and the ONE exchange of the path - returning (observation, reward, done) rows to the root device for a single-process
// consumer - is a grouped ncclSend / ncclRecv over RCCL / xGMI.  librccl is dlopen-ed on first use (like hiprtc), so the library
// loads without it; a group of one device, or one with a repeated device (rehearsal on a 1-GPU box: RCCL refuses duplicates),
// moves its messages with device-to-device copies instead.
//
// Built ONLY on the public C ABI (include/mvrl.h) plus the HIP runtime: a group is a composition of handles, nothing else.
// Message format = distributed.OutputGather's: per shard obs[cmax, obs_dim] f32 | reward[cmax] f32 (absent for the rigid-body
// models, whose reward is identically 0, 6DoF.py:575) | done[cmax] u8, padded to 16 B; the step kernel writes its outputs straight
// into the message (no pack kernel); the root's receive buffer is [n_shards][msg_bytes]; two message buffers so that the gather of
// step k overlaps step k + 1.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/mvrl.h"

namespace {

// ---- librccl, resolved at run time ----------------------------------------------------------------------------------------
typedef void* ncclComm_t;
struct RcclApi {
    void* lib = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string why;
};
RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return &api;
    tried = true;
    const char* names[] = {getenv("MVRL_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (api.lib) break;
        api.why = dlerror();
    }
    if (!api.lib) return &api;
#define MVRL_SYM(field, name)                                                     \
    *(void**)(&api.field) = dlsym(api.lib, name);                                 \
    if (!api.field) { api.why = std::string("librccl: missing ") + name; dlclose(api.lib); api.lib = nullptr; return &api; }
    MVRL_SYM(CommInitAll, "ncclCommInitAll")
    MVRL_SYM(CommDestroy, "ncclCommDestroy")
    MVRL_SYM(GroupStart, "ncclGroupStart")
    MVRL_SYM(GroupEnd, "ncclGroupEnd")
    MVRL_SYM(Send, "ncclSend")
    MVRL_SYM(Recv, "ncclRecv")
    MVRL_SYM(GetErrorString, "ncclGetErrorString")
#undef MVRL_SYM
    return &api;
}
const int kNcclUint8 = 1;   // rccl.h: ncclUint8

thread_local std::string g_group_error;

}  // namespace

struct mvrl_group {
    mvrl_group_layout lay;
    std::vector<int> devices;
    std::vector<mvrl_handle*> h;
    std::vector<int64_t> first, count;
    int root = 0, act_dim = 0;
    bool use_rccl = false, fixed_sp = false;
    std::vector<ncclComm_t> comm;
    struct Shard {
        void* msg[2] = {nullptr, nullptr};       // the message the step kernel writes (device i)
        void* rew_scratch = nullptr;             // where the all-zero reward of the rigid-body models goes
        void* actions = nullptr;                 // [count, act_dim] f32: target of mvrl_group_scatter_actions_dev / fill
        hipStream_t s_step = nullptr, s_comm = nullptr;
        hipEvent_t ev_step[2] = {nullptr, nullptr}, ev_sent[2] = {nullptr, nullptr};
    };
    std::vector<Shard> sh;
    void* recv[2] = {nullptr, nullptr};          // root device: [n_shards][msg_bytes]
    hipEvent_t ev_gathered[2] = {nullptr, nullptr};
    int64_t k = 0;                               // steps gathered so far: message buffer of the current step = k & 1
    bool stepped = false, gathered_once = false;
    int last_gathered = 0;
    std::string err;
};

namespace {

int gfail(mvrl_group* g, int code, const std::string& msg) {
    if (g) g->err = msg;
    g_group_error = msg;
    return code;
}
#define G_HIP(g, call)                                                                                            \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) return gfail(g, e_ == hipErrorOutOfMemory ? MVRL_ENOMEM : MVRL_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define G_MVRL(g, i, call)                                                                                        \
    do {                                                                                                          \
        int rc_ = (call);                                                                                         \
        if (rc_) return gfail(g, rc_, std::string("shard ") + std::to_string(i) + ": " + mvrl_last_error((g)->h[i])); \
    } while (0)
#define G_NCCL(g, call)                                                                                           \
    do {                                                                                                          \
        int r_ = (call);                                                                                          \
        if (r_ != 0) return gfail(g, MVRL_EHIP, std::string(#call) + ": " + rccl()->GetErrorString(r_));          \
    } while (0)

void views(const mvrl_group* g, void* msg, void* rew_scratch, void** obs, void** rew, void** done) {
    char* b = (char*)msg;
    *obs = b;
    *rew = g->lay.reward_plane ? (void*)(b + g->lay.off_reward) : rew_scratch;
    *done = b + g->lay.off_done;
}

}  // namespace

extern "C" {

// distributed.shard_range: contiguous blocks, the first (n_global % n_shards) shards own one env more
int mvrl_group_shard_range(int64_t n_global, int32_t shard, int32_t n_shards, int64_t* first, int64_t* count) {
    if (n_global < 1 || n_shards < 1 || shard < 0 || shard >= n_shards || !first || !count) return gfail(nullptr, MVRL_EINVAL, "bad shard arguments");
    const int64_t base = n_global / n_shards, extra = n_global % n_shards;
    *count = base + (shard < extra ? 1 : 0);
    *first = shard * base + (shard < extra ? shard : extra);
    return MVRL_OK;
}

// distributed.OutputGather's message: offsets and size for a batch cut into n_shards (no GPU needed)
int mvrl_group_message_layout(int64_t n_global, int32_t n_shards, int32_t obs_dim, int32_t reward_plane, mvrl_group_layout* out) {
    if (n_global < 1 || n_shards < 1 || obs_dim < 1 || !out) return gfail(nullptr, MVRL_EINVAL, "bad layout arguments");
    memset(out, 0, sizeof(*out));
    out->n_global = n_global; out->n_shards = n_shards; out->obs_dim = obs_dim; out->reward_plane = reward_plane ? 1 : 0;
    out->cmax = n_global / n_shards + (n_global % n_shards ? 1 : 0);
    out->off_reward = out->cmax * obs_dim * 4;
    out->off_done = out->off_reward + (reward_plane ? out->cmax * 4 : 0);
    out->msg_bytes = (out->off_done + out->cmax + 15) / 16 * 16;
    return MVRL_OK;
}

const char* mvrl_group_last_error(const mvrl_group* g) { return g ? g->err.c_str() : g_group_error.c_str(); }

void mvrl_group_destroy(mvrl_group* g) {
    if (!g) return;
    for (size_t i = 0; i < g->h.size(); i++) {
        if (!g->h[i]) continue;
        (void)hipSetDevice(g->devices[i]);
        mvrl_group::Shard& s = g->sh[i];
        if (s.s_step) (void)hipStreamSynchronize(s.s_step);
        if (s.s_comm) (void)hipStreamSynchronize(s.s_comm);
    }
    if (g->use_rccl)
        for (ncclComm_t c : g->comm)
            if (c) rccl()->CommDestroy(c);
    for (size_t i = 0; i < g->h.size(); i++) {
        if (!g->h[i]) continue;
        (void)hipSetDevice(g->devices[i]);
        mvrl_group::Shard& s = g->sh[i];
        for (int b = 0; b < 2; b++) {
            if (s.msg[b]) (void)hipFree(s.msg[b]);
            if (s.ev_step[b]) (void)hipEventDestroy(s.ev_step[b]);
            if (s.ev_sent[b]) (void)hipEventDestroy(s.ev_sent[b]);
        }
        if (s.rew_scratch) (void)hipFree(s.rew_scratch);
        if (s.actions) (void)hipFree(s.actions);
        if (s.s_step) (void)hipStreamDestroy(s.s_step);
        if (s.s_comm) (void)hipStreamDestroy(s.s_comm);
        if ((int)i == g->root)
            for (int b = 0; b < 2; b++) {
                if (g->recv[b]) (void)hipFree(g->recv[b]);
                if (g->ev_gathered[b]) (void)hipEventDestroy(g->ev_gathered[b]);
            }
        mvrl_destroy(g->h[i]);
    }
    delete g;
}

int mvrl_group_create(const mvrl_config* cfg, const int32_t* devices, int32_t n_devices, int32_t root, mvrl_group** out) {
    if (!cfg || !devices || !out || n_devices < 1 || n_devices > 64 || root < 0 || root >= n_devices)
        return gfail(nullptr, MVRL_EINVAL, "mvrl_group_create: need a config, 1..64 devices and a root index inside the list");
    if (cfg->abi_version != MVRL_ABI_VERSION) return gfail(nullptr, MVRL_EINVAL, "ABI version mismatch");
    if (cfg->precision != MVRL_PREC_F32) return gfail(nullptr, MVRL_EINVAL, "mvrl_group: the gather message is an fp32 format (precision F32 only)");
    if (cfg->n_envs < n_devices) return gfail(nullptr, MVRL_EINVAL, "mvrl_group_create: fewer envs than devices");
    int32_t act = 0, obs = 0, ini = 0, words = 0;
    if (mvrl_model_dims(cfg->model, &act, &obs, &ini, &words)) return gfail(nullptr, MVRL_EINVAL, "unknown model");
    const int n_visible = mvrl_device_count();
    if (n_visible < 1) return gfail(nullptr, MVRL_ENODEV, "no HIP device visible (there is no CPU fallback)");
    bool dup = false;
    for (int i = 0; i < n_devices; i++) {
        if (devices[i] < 0 || devices[i] >= n_visible) return gfail(nullptr, MVRL_EINVAL, "mvrl_group_create: device ordinal out of range");
        for (int j = 0; j < i; j++) dup |= devices[j] == devices[i];
    }
    mvrl_group* g = new mvrl_group();
    const bool reward_plane = cfg->model == MVRL_MODEL_AUV;   // the rigid-body models' reward is identically 0 (6DoF.py:575, 3DoF.py:495)
    mvrl_group_message_layout(cfg->n_envs, n_devices, obs, reward_plane, &g->lay);
    g->root = root; g->act_dim = act; g->fixed_sp = cfg->fixed_setpoint != 0;
    g->devices.assign(devices, devices + n_devices);
    g->h.assign(n_devices, nullptr);
    g->sh.resize(n_devices);
    g->first.resize(n_devices); g->count.resize(n_devices);
    const char* tr = getenv("MVRL_GROUP_TRANSPORT");
    const bool want_copy = tr && !strcmp(tr, "copy"), want_rccl = tr && !strcmp(tr, "rccl");
    // RCCL needs distinct devices; a one-device group uses it only on request (MVRL_GROUP_TRANSPORT=rccl: a one-rank communicator
    // sending to itself - the same calls, datatypes and stream protocol as the 8-GPU gather, runnable on a 1-GPU box)
    g->use_rccl = !dup && !want_copy && (n_devices > 1 || want_rccl);
#define G_CREATE_TRY(code_, msg_) do { const int c__ = (code_); const std::string m__ = (msg_); mvrl_group_destroy(g); return gfail(nullptr, c__, m__); } while (0)
    for (int i = 0; i < n_devices; i++) {
        mvrl_group_shard_range(cfg->n_envs, i, n_devices, &g->first[i], &g->count[i]);
        mvrl_config c = *cfg;
        c.device = devices[i]; c.n_envs = g->count[i]; c.env_offset = cfg->env_offset + g->first[i];
        int rc = mvrl_create(&c, &g->h[i]);
        if (rc) G_CREATE_TRY(rc, std::string("shard ") + std::to_string(i) + ": " + mvrl_last_error(nullptr));
        mvrl_group::Shard& s = g->sh[i];
        hipError_t e = hipSetDevice(devices[i]);
        for (int b = 0; b < 2 && e == hipSuccess; b++) {
            e = hipMalloc(&s.msg[b], (size_t)g->lay.msg_bytes);
            if (e == hipSuccess) e = hipMemset(s.msg[b], 0, (size_t)g->lay.msg_bytes);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.ev_step[b], hipEventDisableTiming);
        }
        // "message b has left its buffer": recorded on the shard's own communication stream under RCCL, on the ROOT's under the copy
        // transport (the root pulls) - an event is recorded on a stream of the device it was created on, waited for from any device
        if (e == hipSuccess && !g->use_rccl) e = hipSetDevice(devices[root]);
        for (int b = 0; b < 2 && e == hipSuccess; b++) e = hipEventCreateWithFlags(&s.ev_sent[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipSetDevice(devices[i]);
        if (e == hipSuccess && !reward_plane) e = hipMalloc(&s.rew_scratch, (size_t)g->lay.cmax * 4);
        if (e == hipSuccess) e = hipMalloc(&s.actions, (size_t)g->count[i] * act * 4);
        if (e == hipSuccess) e = hipMemset(s.actions, 0, (size_t)g->count[i] * act * 4);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.s_step, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&s.s_comm, hipStreamNonBlocking);
        if (e == hipSuccess && i == root)
            for (int b = 0; b < 2 && e == hipSuccess; b++) {
                e = hipMalloc(&g->recv[b], (size_t)g->lay.msg_bytes * n_devices);
                if (e == hipSuccess) e = hipMemset(g->recv[b], 0, (size_t)g->lay.msg_bytes * n_devices);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&g->ev_gathered[b], hipEventDisableTiming);
            }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) G_CREATE_TRY(e == hipErrorOutOfMemory ? MVRL_ENOMEM : MVRL_EHIP, std::string("group buffers on device ") + std::to_string(devices[i]) + ": " + hipGetErrorString(e));
    }
    if (g->use_rccl) {
        RcclApi* api = rccl();
        if (!api->lib) G_CREATE_TRY(MVRL_EHIP, "librccl could not be loaded (" + api->why + "); MVRL_GROUP_TRANSPORT=copy moves the messages with device-to-device copies instead");
        g->comm.assign(n_devices, nullptr);
        const int r = api->CommInitAll(g->comm.data(), n_devices, g->devices.data());
        if (r != 0) G_CREATE_TRY(MVRL_EHIP, std::string("ncclCommInitAll: ") + api->GetErrorString(r));
    } else if (n_devices > 1) {
        for (int i = 0; i < n_devices; i++)   // copy transport between distinct devices: peer access where the topology offers it
            if (i != root && devices[i] != devices[root]) {
                (void)hipSetDevice(devices[root]);
                (void)hipDeviceEnablePeerAccess(devices[i], 0);   // already-enabled / unsupported are not errors: the copy is staged then
                (void)hipGetLastError();
            }
    }
#undef G_CREATE_TRY
    g->lay.transport = g->use_rccl ? 1 : 0;
    *out = g;
    return MVRL_OK;
}

int mvrl_group_info(const mvrl_group* g, mvrl_group_layout* out) {
    if (!g || !out) return gfail(nullptr, MVRL_EINVAL, "null argument");
    *out = g->lay;
    return MVRL_OK;
}
mvrl_handle* mvrl_group_shard(mvrl_group* g, int32_t shard) { return (g && shard >= 0 && shard < (int)g->h.size()) ? g->h[shard] : nullptr; }

int mvrl_group_set_flow(mvrl_group* g, const float* table_host, const mvrl_flow_desc* desc) {
    if (!g) return gfail(g, MVRL_EINVAL, "null group");
    for (size_t i = 0; i < g->h.size(); i++) G_MVRL(g, i, mvrl_set_flow(g->h[i], table_host, desc));   // replicated: 60 MB per device
    return MVRL_OK;
}

// Every shard starts new episodes (random: Philox keyed by the GLOBAL env id; explicit initial values are set per shard through
// mvrl_group_shard + mvrl_reset); the first observations land in the current message buffer with done = 0, so a gather hands them
// to the root like a step's.
int mvrl_group_reset(mvrl_group* g) {
    if (!g) return gfail(g, MVRL_EINVAL, "null group");
    const int b = (int)(g->k & 1);
    for (size_t i = 0; i < g->h.size(); i++) {
        mvrl_group::Shard& s = g->sh[i];
        G_HIP(g, hipSetDevice(g->devices[i]));
        G_HIP(g, hipStreamWaitEvent(s.s_step, s.ev_sent[b], 0));
        void *o, *r, *d;
        views(g, s.msg[b], s.rew_scratch, &o, &r, &d);
        G_MVRL(g, i, mvrl_reset_dev(g->h[i], nullptr, nullptr, o, s.s_step));
        G_HIP(g, hipMemsetAsync(d, 0, (size_t)g->count[i], s.s_step));
        G_HIP(g, hipEventRecord(s.ev_step[b], s.s_step));
    }
    g->stepped = true;
    return MVRL_OK;
}

// One env step of the whole batch: one launch per device on that device's stream, no host synchronisation between them.
// actions_dev[i]: device pointer ON DEVICE i to shard i's rows [count_i, act_dim] f32; NULL (the array or an entry) = the group's own
// action buffer of that shard (mvrl_group_scatter_actions_dev / mvrl_group_fill_actions), or nothing with a fixed set-point.
int mvrl_group_step_dev(mvrl_group* g, const void* const* actions_dev) {
    if (!g) return gfail(g, MVRL_EINVAL, "null group");
    const int b = (int)(g->k & 1);
    for (size_t i = 0; i < g->h.size(); i++) {
        mvrl_group::Shard& s = g->sh[i];
        G_HIP(g, hipSetDevice(g->devices[i]));
        // the message buffer may only be overwritten once its previous gather (two steps ago) has left it
        G_HIP(g, hipStreamWaitEvent(s.s_step, s.ev_sent[b], 0));
        void *o, *r, *d;
        views(g, s.msg[b], s.rew_scratch, &o, &r, &d);
        const void* a = (actions_dev && actions_dev[i]) ? actions_dev[i] : (g->fixed_sp ? nullptr : s.actions);
        G_MVRL(g, i, mvrl_step_dev(g->h[i], a, o, r, (uint8_t*)d, s.s_step));
        G_HIP(g, hipEventRecord(s.ev_step[b], s.s_step));
    }
    g->stepped = true;
    return MVRL_OK;
}

// The ONE exchange of the path: every shard's message of the last step -> the root device's receive buffer, on the communication
// streams (the next mvrl_group_step_dev overlaps it: it writes the OTHER message buffer).  RCCL: one ncclGroupStart / End holding an
// ncclSend per shard and the root's ncclRecv from every shard (root included: a self-send is legal inside a group).
int mvrl_group_gather_dev(mvrl_group* g) {
    if (!g) return gfail(g, MVRL_EINVAL, "null group");
    if (!g->stepped) return gfail(g, MVRL_ESTATE, "mvrl_group_gather_dev: nothing stepped or reset since the last gather");
    const int b = (int)(g->k & 1), n = (int)g->h.size();
    const size_t mb = (size_t)g->lay.msg_bytes;
    mvrl_group::Shard& rs = g->sh[g->root];
    for (int i = 0; i < n; i++) {
        G_HIP(g, hipSetDevice(g->devices[i]));
        G_HIP(g, hipStreamWaitEvent(g->sh[i].s_comm, g->sh[i].ev_step[b], 0));
    }
    if (g->use_rccl) {
        RcclApi* api = rccl();
        G_NCCL(g, api->GroupStart());
        for (int i = 0; i < n; i++) {
            G_HIP(g, hipSetDevice(g->devices[i]));
            G_NCCL(g, api->Send(g->sh[i].msg[b], mb, kNcclUint8, g->root, g->comm[i], g->sh[i].s_comm));
        }
        G_HIP(g, hipSetDevice(g->devices[g->root]));
        for (int i = 0; i < n; i++) G_NCCL(g, api->Recv((char*)g->recv[b] + (size_t)i * mb, mb, kNcclUint8, i, g->comm[g->root], rs.s_comm));
        G_NCCL(g, api->GroupEnd());
        for (int i = 0; i < n; i++) {
            G_HIP(g, hipSetDevice(g->devices[i]));
            G_HIP(g, hipEventRecord(g->sh[i].ev_sent[b], g->sh[i].s_comm));
        }
    } else {
        // copy transport: the root's communication stream pulls every message (device-to-device / peer copies)
        G_HIP(g, hipSetDevice(g->devices[g->root]));
        for (int i = 0; i < n; i++) {
            if (i != g->root) G_HIP(g, hipStreamWaitEvent(rs.s_comm, g->sh[i].ev_step[b], 0));
            if (g->devices[i] == g->devices[g->root])
                G_HIP(g, hipMemcpyAsync((char*)g->recv[b] + (size_t)i * mb, g->sh[i].msg[b], mb, hipMemcpyDeviceToDevice, rs.s_comm));
            else
                G_HIP(g, hipMemcpyPeerAsync((char*)g->recv[b] + (size_t)i * mb, g->devices[g->root], g->sh[i].msg[b], g->devices[i], mb, rs.s_comm));
        }
        for (int i = 0; i < n; i++) G_HIP(g, hipEventRecord(g->sh[i].ev_sent[b], rs.s_comm));
    }
    G_HIP(g, hipSetDevice(g->devices[g->root]));
    G_HIP(g, hipEventRecord(g->ev_gathered[b], rs.s_comm));
    g->last_gathered = b;
    g->gathered_once = true;
    g->stepped = false;
    g->k += 1;
    return MVRL_OK;
}

// Block the host until the last gather has arrived at the root.
int mvrl_group_wait(mvrl_group* g) {
    if (!g) return gfail(g, MVRL_EINVAL, "null group");
    if (!g->gathered_once) return gfail(g, MVRL_ESTATE, "mvrl_group_wait: no gather yet");
    G_HIP(g, hipSetDevice(g->devices[g->root]));
    G_HIP(g, hipEventSynchronize(g->ev_gathered[g->last_gathered]));
    return MVRL_OK;
}

// Root-device pointers to shard `shard`'s rows of the LAST gather: obs [count, obs_dim] f32, reward [count] f32 (NULL for the
// rigid-body models: identically 0), done [count] u8 (bit 0 done, bit 1 time limit); first / count = the shard's global env range.
// Valid until the gather after the next one (two buffers); order a consumer stream behind the gather with mvrl_group_wait, or -
// without blocking the host - with mvrl_group_gathered_event.
int mvrl_group_root_views(mvrl_group* g, int32_t shard, const float** obs, const float** reward, const uint8_t** done, int64_t* first, int64_t* count) {
    if (!g || shard < 0 || shard >= (int)g->h.size()) return gfail(g, MVRL_EINVAL, "bad shard");
    if (!g->gathered_once) return gfail(g, MVRL_ESTATE, "mvrl_group_root_views: no gather yet");
    char* m = (char*)g->recv[g->last_gathered] + (size_t)shard * (size_t)g->lay.msg_bytes;
    if (obs) *obs = (const float*)m;
    if (reward) *reward = g->lay.reward_plane ? (const float*)(m + g->lay.off_reward) : nullptr;
    if (done) *done = (const uint8_t*)(m + g->lay.off_done);
    if (first) *first = g->first[shard];
    if (count) *count = g->count[shard];
    return MVRL_OK;
}
// hipEvent_t (as void*) recorded behind the last gather on the root's communication stream: hipStreamWaitEvent(consumer, ev)
int mvrl_group_gathered_event(mvrl_group* g, void** event) {
    if (!g || !event) return gfail(g, MVRL_EINVAL, "null argument");
    if (!g->gathered_once) return gfail(g, MVRL_ESTATE, "no gather yet");
    *event = (void*)g->ev_gathered[g->last_gathered];
    return MVRL_OK;
}

// The mirror direction (SB3's pipe send): actions_root_dev [n_global, act_dim] f32 on the ROOT device -> every shard's own action
// buffer, which the next mvrl_group_step_dev(g, NULL) reads.  Enqueued on the step streams (ordered before that step); the caller's
// buffer must be complete (synchronise the stream that produced it, or produce it on the root's step stream).
int mvrl_group_scatter_actions_dev(mvrl_group* g, const float* actions_root_dev) {
    if (!g || !actions_root_dev) return gfail(g, MVRL_EINVAL, "null argument");
    const int n = (int)g->h.size();
    const size_t row = (size_t)g->act_dim * 4;
    if (g->use_rccl) {
        RcclApi* api = rccl();
        G_NCCL(g, api->GroupStart());
        G_HIP(g, hipSetDevice(g->devices[g->root]));
        for (int i = 0; i < n; i++)
            G_NCCL(g, api->Send((const char*)actions_root_dev + (size_t)g->first[i] * row, (size_t)g->count[i] * row, kNcclUint8, i, g->comm[g->root], g->sh[g->root].s_step));
        for (int i = 0; i < n; i++) {
            G_HIP(g, hipSetDevice(g->devices[i]));
            G_NCCL(g, api->Recv(g->sh[i].actions, (size_t)g->count[i] * row, kNcclUint8, g->root, g->comm[i], g->sh[i].s_step));
        }
        G_NCCL(g, api->GroupEnd());
    } else {
        for (int i = 0; i < n; i++) {
            G_HIP(g, hipSetDevice(g->devices[i]));
            const char* src = (const char*)actions_root_dev + (size_t)g->first[i] * row;
            if (g->devices[i] == g->devices[g->root]) G_HIP(g, hipMemcpyAsync(g->sh[i].actions, src, (size_t)g->count[i] * row, hipMemcpyDeviceToDevice, g->sh[i].s_step));
            else G_HIP(g, hipMemcpyPeerAsync(g->sh[i].actions, g->devices[i], src, g->devices[g->root], (size_t)g->count[i] * row, g->sh[i].s_step));
        }
    }
    return MVRL_OK;
}

// Synthetic roll-outs: uniform(lo, hi) actions into every shard's own buffer (mvrl_fill_uniform_dev, Philox keyed by seed and by
// the stream counter * n_shards + shard: reproducible for a given sharding).
int mvrl_group_fill_actions(mvrl_group* g, uint64_t seed, uint64_t counter, float lo, float hi) {
    if (!g) return gfail(g, MVRL_EINVAL, "null group");
    for (size_t i = 0; i < g->h.size(); i++)
        G_MVRL(g, i, mvrl_fill_uniform_dev(g->h[i], (float*)g->sh[i].actions, g->count[i] * g->act_dim, seed, counter * g->h.size() + i, lo, hi,
                                           g->sh[i].s_step));
    return MVRL_OK;
}

// Block the host until every device's step and communication streams have drained.
int mvrl_group_synchronize(mvrl_group* g) {
    if (!g) return gfail(g, MVRL_EINVAL, "null group");
    for (size_t i = 0; i < g->h.size(); i++) {
        G_HIP(g, hipSetDevice(g->devices[i]));
        G_HIP(g, hipStreamSynchronize(g->sh[i].s_step));
        G_HIP(g, hipStreamSynchronize(g->sh[i].s_comm));
    }
    return MVRL_OK;
}

// Host copy of the last gather in GLOBAL env order: obs [n_global, obs_dim], reward [n_global] (zeros for the rigid-body models),
// done [n_global]; any pointer may be NULL.  Blocks until the gather has arrived.
int mvrl_group_download(mvrl_group* g, float* obs, float* reward, uint8_t* done) {
    int rc = mvrl_group_wait(g);
    if (rc) return rc;
    G_HIP(g, hipSetDevice(g->devices[g->root]));
    for (size_t i = 0; i < g->h.size(); i++) {
        const float *o, *r;
        const uint8_t* d;
        int64_t first, count;
        mvrl_group_root_views(g, (int32_t)i, &o, &r, &d, &first, &count);
        if (obs) G_HIP(g, hipMemcpy(obs + first * g->lay.obs_dim, o, (size_t)count * g->lay.obs_dim * 4, hipMemcpyDeviceToHost));
        if (reward) {
            if (r) G_HIP(g, hipMemcpy(reward + first, r, (size_t)count * 4, hipMemcpyDeviceToHost));
            else memset(reward + first, 0, (size_t)count * 4);
        }
        if (done) G_HIP(g, hipMemcpy(done + first, d, (size_t)count, hipMemcpyDeviceToHost));
    }
    return MVRL_OK;
}

}  // extern "C"
