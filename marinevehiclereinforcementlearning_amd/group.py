"""Several GPUs of one node behind ONE object in ONE host process (`mvrl_group_*`, csrc/mvrl_group.hip): the arrangement that
replaces SB3's SubprocVecEnv (tag/main_00_sbl.py:145-146) for a caller without torch.distributed - BASELINE configs[4] from a
single process.  Contiguous shards, one launch per device and step, one grouped RCCL send / recv of (obs, reward, done) to the
root device per step (device-to-device copies for a single or repeated device).  `distributed.py` is the one-process-per-GPU
arrangement of the same path; both cut the batch with the same `shard_range` and exchange the same message format."""
import ctypes as C

import numpy as np

from . import _lib, params as P


class GroupLayout(C.Structure):
    _fields_ = [("n_global", C.c_int64), ("n_shards", C.c_int32), ("obs_dim", C.c_int32), ("reward_plane", C.c_int32),
                ("transport", C.c_int32), ("cmax", C.c_int64), ("off_reward", C.c_int64), ("off_done", C.c_int64),
                ("msg_bytes", C.c_int64)]


def _declare(lib):
    if getattr(lib, "_mvrl_group_declared", False):
        return lib
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.mvrl_group_shard_range.argtypes = [i64, i32, i32, C.POINTER(i64), C.POINTER(i64)]
    lib.mvrl_group_message_layout.argtypes = [i64, i32, i32, i32, C.POINTER(GroupLayout)]
    lib.mvrl_group_create.argtypes = [vp, C.POINTER(i32), i32, i32, C.POINTER(vp)]
    lib.mvrl_group_destroy.argtypes = [vp]
    lib.mvrl_group_destroy.restype = None
    lib.mvrl_group_last_error.argtypes = [vp]
    lib.mvrl_group_last_error.restype = C.c_char_p
    lib.mvrl_group_info.argtypes = [vp, C.POINTER(GroupLayout)]
    lib.mvrl_group_shard.argtypes = [vp, i32]
    lib.mvrl_group_shard.restype = vp
    lib.mvrl_group_set_flow.argtypes = [vp, vp, vp]
    lib.mvrl_group_reset.argtypes = [vp]
    lib.mvrl_group_step_dev.argtypes = [vp, vp]
    lib.mvrl_group_gather_dev.argtypes = [vp]
    lib.mvrl_group_wait.argtypes = [vp]
    lib.mvrl_group_gathered_event.argtypes = [vp, C.POINTER(vp)]
    lib.mvrl_group_root_views.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(i64), C.POINTER(i64)]
    lib.mvrl_group_download.argtypes = [vp, vp, vp, vp]
    lib.mvrl_group_scatter_actions_dev.argtypes = [vp, vp]
    lib.mvrl_group_fill_actions.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_float, C.c_float]
    lib.mvrl_group_synchronize.argtypes = [vp]
    lib._mvrl_group_declared = True
    return lib


def shard_range(n_global, shard, n_shards):
    """(first, count) of a shard as the C side cuts the batch (== distributed.shard_range; needs no GPU)."""
    lib = _declare(_lib.load())
    a, b = C.c_int64(), C.c_int64()
    if lib.mvrl_group_shard_range(n_global, shard, n_shards, C.byref(a), C.byref(b)):
        raise _lib.MvrlError(lib.mvrl_group_last_error(None).decode())
    return a.value, b.value


def message_layout(n_global, n_shards, obs_dim, reward_plane=True):
    """Offsets / size of one shard's gather message as the C side lays it out (== distributed.message_layout; needs no GPU)."""
    lib = _declare(_lib.load())
    lay = GroupLayout()
    if lib.mvrl_group_message_layout(n_global, n_shards, obs_dim, int(bool(reward_plane)), C.byref(lay)):
        raise _lib.MvrlError(lib.mvrl_group_last_error(None).decode())
    return dict(cmax=lay.cmax, off_reward=lay.off_reward, off_done=lay.off_done, msg_bytes=lay.msg_bytes)


class _ShardHandle(_lib.Handle):
    """A shard's mvrl_handle wrapped WITHOUT ownership (the group destroys it): get_state / set_state / reset with explicit values."""

    def __init__(self, group, ptr, cfg):  # noqa: D401 - deliberately not calling Handle.__init__ (no mvrl_create)
        self.lib = group.lib
        self.cfg = cfg
        self.model = int(cfg.model)
        self.n = int(cfg.n_envs)
        self.f64 = False
        self.dtype, self.itype, self._sfx = np.float32, np.int32, ""
        self.act_dim, self.obs_dim, self.init_dim, self.state_words, self.aux_dim = P.MODEL_DIMS[self.model]
        self.h = C.c_void_p(ptr)
        self._group = group          # keeps the owner alive
        pa, po, pr, pd = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.mvrl_host_buffers(self.h, C.byref(pa), C.byref(po), C.byref(pr), C.byref(pd)), self.h)

        def view(ptr_, shape, ctype, dtype):
            return np.frombuffer((ctype * int(np.prod(shape))).from_address(ptr_.value), dtype=dtype).reshape(shape)
        self._act_in = view(pa, (self.n, self.act_dim), C.c_float, np.float32)
        self._obs = view(po, (self.n, self.obs_dim), C.c_float, np.float32)
        self._rew = view(pr, (self.n,), C.c_float, np.float32)
        self._done = view(pd, (self.n,), C.c_uint8, np.uint8)

    def close(self):                 # the group owns the handle
        self.h = None

    def __del__(self):
        pass


class DeviceGroup:
    """`DeviceGroup(cfg, devices)`: cfg as for a handle with n_envs = the GLOBAL env count; devices = HIP ordinals (a repeated ordinal
    is allowed: messages then move by device-to-device copies - rehearsal on a 1-GPU box)."""

    def __init__(self, cfg, devices, root=0):
        self.lib = _declare(_lib.load())
        self.cfg = cfg
        self.devices = [int(d) for d in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        g = C.c_void_p()
        rc = self.lib.mvrl_group_create(C.byref(cfg), arr, len(self.devices), int(root), C.byref(g))
        if rc:
            raise _lib.MvrlError(f"mvrl_group_create: {_lib.ERRORS.get(rc, rc)}: {self.lib.mvrl_group_last_error(None).decode()}")
        self.g = g
        self.root = int(root)
        self.layout = GroupLayout()
        self._check(self.lib.mvrl_group_info(self.g, C.byref(self.layout)))
        self.n_global = int(cfg.n_envs)
        self.act_dim, self.obs_dim = P.MODEL_DIMS[int(cfg.model)][:2]
        self.ranges = [shard_range(self.n_global, i, len(self.devices)) for i in range(len(self.devices))]

    def _check(self, rc):
        if rc:
            raise _lib.MvrlError(f"{_lib.ERRORS.get(rc, rc)}: {self.lib.mvrl_group_last_error(self.g).decode()}")

    @property
    def transport(self):
        return "rccl" if self.layout.transport else "copy"

    def shard(self, i):
        first, count = self.ranges[i]
        c = type(self.cfg).from_buffer_copy(self.cfg)
        c.n_envs, c.env_offset, c.device = count, int(self.cfg.env_offset) + first, self.devices[i]
        return _ShardHandle(self, self.lib.mvrl_group_shard(self.g, i), c)

    def set_flow(self, table_uv, dt, dx, dy):
        t = np.ascontiguousarray(table_uv, np.float32)
        d = _lib.flow_desc(t.shape[0], t.shape[1], t.shape[2], dt, dx, dy)
        self._check(self.lib.mvrl_group_set_flow(self.g, t.ctypes.data, C.byref(d)))

    def reset(self):
        self._check(self.lib.mvrl_group_reset(self.g))

    def step_dev(self, action_ptrs=None):
        """action_ptrs: per-shard device pointers (ints) or None = the group's own action buffers (scatter_actions_dev / fill_actions)."""
        arr = None
        if action_ptrs is not None:
            arr = (C.c_void_p * len(self.devices))(*[C.c_void_p(int(p) if p else 0) for p in action_ptrs])
        self._check(self.lib.mvrl_group_step_dev(self.g, arr))

    def gather_dev(self):
        self._check(self.lib.mvrl_group_gather_dev(self.g))

    def wait(self):
        self._check(self.lib.mvrl_group_wait(self.g))

    def scatter_actions_dev(self, ptr):
        self._check(self.lib.mvrl_group_scatter_actions_dev(self.g, C.c_void_p(int(ptr))))

    def fill_actions(self, seed, counter, lo=-1.0, hi=1.0):
        self._check(self.lib.mvrl_group_fill_actions(self.g, int(seed), int(counter), float(lo), float(hi)))

    def synchronize(self):
        self._check(self.lib.mvrl_group_synchronize(self.g))

    def root_views(self, i):
        """(obs_ptr, reward_ptr or None, done_ptr, first, count): root-device pointers into the last gather for shard i."""
        o, r, d = C.c_void_p(), C.c_void_p(), C.c_void_p()
        a, b = C.c_int64(), C.c_int64()
        self._check(self.lib.mvrl_group_root_views(self.g, i, C.byref(o), C.byref(r), C.byref(d), C.byref(a), C.byref(b)))
        return o.value, r.value, d.value, a.value, b.value

    def download(self):
        """The last gather as host arrays in global env order: obs [N, obs_dim] f32, reward [N] f32, done bits [N] u8."""
        obs = np.zeros((self.n_global, self.obs_dim), np.float32)
        rew = np.zeros(self.n_global, np.float32)
        done = np.zeros(self.n_global, np.uint8)
        self._check(self.lib.mvrl_group_download(self.g, obs.ctypes.data, rew.ctypes.data, done.ctypes.data))
        return obs, rew, done

    def close(self):
        if getattr(self, "g", None):
            self.lib.mvrl_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class GroupVecEnv:
    """SB3 `VecEnv` calling convention over a `DeviceGroup`: ONE Python process, the node's GPUs, the whole batch in global env order -
    what `SubprocVecEnv([make_env(i) for i in range(nProc)])` (tag/main_00_sbl.py:145) is to its caller, without the processes.

        venv = GroupVecEnv("rov6", 8 * 1048576, devices=range(8), flow=flow)
        obs = venv.reset()
        obs, rewards, dones, infos = venv.step(actions)        # numpy [N, ...] in, numpy out; auto-reset + terminal_observation

    Per step: the action batch is uploaded to the root device and scattered to the shards, every device launches its shard's step,
    the shards' (obs, reward, done) messages are gathered to the root (grouped RCCL send / recv; device-to-device copies for a single
    or repeated device) and downloaded in one piece per plane.  `infos` are lean (a list-like that builds a dict on access)."""

    def __init__(self, model, num_envs, devices, *, seed=0, dt=None, maxSteps=250, n_substeps=4, control_mode="faithful", flow=None,
                 currentVelScale=1.0, currentTurbScale=2.0, noiseMagCoeffs=0.0, noiseMagActuation=0.0, stopOnBoundsExceeded=True,
                 report_truncation=False, root=0):
        from .spaces import unit_box
        cyl = model == "auv_cyl"
        if cyl:
            model, maxSteps = "auv", (1200 if maxSteps == 250 else maxSteps)
        self.model = P.MODEL_NAMES[model] if isinstance(model, str) else int(model)
        self.num_envs = int(num_envs)
        act, obs = P.MODEL_DIMS[self.model][:2]
        self.action_space, self.observation_space = unit_box(act), unit_box(obs)
        self.report_truncation = bool(report_truncation)
        if self.model == P.MODEL_AUV and flow is None:
            raise ValueError("AuvEnv needs a turbulence field (flow=ReconstructedFlow(...)): verySimpleAuv.py:102-104")
        kw = {"auv": P.auv_params(noiseMagCoeffs, noiseMagActuation, stopOnBoundsExceeded, cyl=cyl)} if self.model == P.MODEL_AUV else {}
        cm = {"faithful": P.CTRL_FAITHFUL, "zoh": P.CTRL_ZOH}[control_mode] if isinstance(control_mode, str) else control_mode
        self.cfg = P.make_config(self.model, self.num_envs, dt=dt, n_substeps=n_substeps, max_steps=maxSteps, control_mode=cm,
                                 auto_reset=True, seed=seed or 0, use_flow=flow is not None, **kw)
        self.group = DeviceGroup(self.cfg, list(devices), root=root)
        if flow is not None:
            if self.model == P.MODEL_AUV:
                flow.scale(11., currentVelScale, currentTurbScale, translate=(-1.65, -1.1))      # verySimpleAuv.py:104
            self.group.set_flow(flow.table_uv(), flow.dt, flow.dx, flow.dy)
        self._shards = [self.group.shard(i) for i in range(len(self.group.devices))]
        self._root = self._shards[self.group.root]
        self._act_ptr = self._root.dev_alloc(self.num_envs * act * 4)
        self._pending = False

    def reset(self):
        self.group.reset()
        self.group.gather_dev()
        return self.group.download()[0]

    def step_async(self, actions):
        a = np.ascontiguousarray(np.asarray(actions, np.float32).reshape(self.num_envs, self.action_space.shape[0]))
        self._root.dev_upload(self._act_ptr, a)                 # synchronous: complete before the scatter is enqueued
        self.group.scatter_actions_dev(self._act_ptr)
        self.group.step_dev()
        self.group.gather_dev()
        self._pending = True

    def step_wait(self):
        obs, rew, bits = self.group.download()
        self._pending = False
        dones = bits != 0
        infos = [{} for _ in range(self.num_envs)] if self.num_envs <= 65536 else _LazyInfos(self.num_envs)
        idx = np.nonzero(dones)[0]
        if len(idx):
            term = np.concatenate([s.terminal_obs() for s in self._shards], axis=0)
            for i in idx:
                infos[i] = {"terminal_observation": term[i].copy()}
                if self.report_truncation:
                    infos[i]["TimeLimit.truncated"] = bool(bits[i] & 2)
        return obs, rew, dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def get_state(self, raw=True):
        """The shards' state planes side by side = the planes of the unsharded batch (raw: the C ABI's bit patterns)."""
        return np.concatenate([s.get_state(raw=raw) for s in self._shards], axis=1)

    def close(self):
        if getattr(self, "group", None) is not None:
            try:
                self._root.dev_free(self._act_ptr)
            except Exception:  # noqa: BLE001
                pass
            self._shards = []
            self.group.close()
            self.group = None

    def seed(self, seed=None):
        return [None] * self.num_envs

    def render(self, mode="human"):
        return None

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False] * self.num_envs

    def get_attr(self, attr_name, indices=None):
        return [getattr(self, attr_name)] * self.num_envs


class _LazyInfos(dict):
    """infos of a very large batch: a mapping index -> dict that answers {} for envs that did not finish (a million empty dicts
    per step cost more than the step)."""

    def __init__(self, n):
        super().__init__()
        self._n = n

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return dict.get(self, i, {})
