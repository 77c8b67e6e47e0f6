"""Vectorised environment with the stable-baselines3 `VecEnv` calling convention, backed by libmvrl.so.

Replaces `SubprocVecEnv([make_env(i, env_kwargs) for i in range(nProc)])` of the reference's training
drivers (tag_00_Dec2023_simpleControlTurbulence/main_00_sbl.py:145-146): instead of one Python process per
environment talking over pipes, all N environments are lanes of one HIP kernel launch on one GPU.

SB3 is not a dependency (it is not installed in the build image): the class duck-types the VecEnv interface
of SB3 1.6-1.8 (gym 0.21 API, which is what the reference uses):
    num_envs, observation_space, action_space
    reset() -> obs[N, obs_dim] f32
    step_async(actions) ; step_wait() -> (obs, rewards[N] f32, dones[N] bool, infos: list[dict])
    step(actions) ; close() ; seed() ; get_attr / set_attr / env_method / env_is_wrapped ; render()
with SB3's auto-reset semantics: on a `done` the returned row of `obs` is the first observation of the next
episode and the last observation of the finished one is in infos[i]["terminal_observation"].  The reference's envs are not
TimeLimit-wrapped and return info = {} (verySimpleAuv.py:410), so `info.get("TimeLimit.truncated", False)` at
main_02_sbl_contrib_customBuffer.py:154 is always False there: a time-limit `done` is a true terminal.  That is the
default here too; `report_truncation=True` adds infos[i]["TimeLimit.truncated"] (from the done byte's time-limit bit) for
learners that want to bootstrap through truncations - a deliberate, opt-in deviation.

`infos="lean"` returns one shared, lazily-evaluated infos object instead of N dicts - at 1e5..1e7 environments the
list of dicts alone costs more than the physics.  `step_tensors` keeps actions/observations on the GPU
(torch tensors, zero-copy through raw device pointers).
"""
import numpy as np

from . import _lib, params as P
from .spaces import unit_box


class _LeanInfos:
    """Sequence-like infos for huge batches: infos[i] builds the dict on demand."""

    def __init__(self, env, done_bits, term_obs_fetch, report_truncation=False):
        self._env, self._bits, self._fetch, self._trunc = env, done_bits, term_obs_fetch, report_truncation
        self._term = None

    def __len__(self):
        return len(self._bits)

    def __getitem__(self, i):
        b = int(self._bits[i])
        if not b:
            return {}
        if self._term is None:
            self._term = self._fetch()
        info = {"terminal_observation": self._term[i]}
        if self._trunc:
            info["TimeLimit.truncated"] = bool(b & 2)
        return info

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    @property
    def done_indices(self):
        return np.nonzero(self._bits)[0]


class MarineVecEnv:
    """N BlueROV2 / AUV environments on one GPU.

    model: "rov6" | "rov3" | "auv" | "auv_cyl".  Remaining keyword arguments mirror the reference constructors
    (`dt`, `maxSteps` - 6DoF.py:446 / 3DoF.py:376; `noiseMagCoeffs`, `noiseMagActuation`, `currentVelScale`,
    `currentTurbScale`, `stopOnBoundsExceeded` - verySimpleAuv.py:77-78) plus the integrator settings that are
    this build's (`n_substeps`, `control_mode`).  `specialize=True` (6-DoF with `vehicle_params` other than the
    reference's): compile the step kernel for those constants at construction (the ROCm installation's hipcc as a child
    process, 2-3 s; mvrl_specialize) - structured constants then run at the speed of the default vehicle, arbitrary ones
    30 % faster than the run-time-constant kernel.  `specialize="auto"` (default): whenever the handle
    is a fp32 6-DoF RK4 handle with non-default constants, keeping the ahead-of-time kernel (with a warning) if no compiler is
    available; `True`: insist (raise on failure); `False`: never.
    """

    metadata = {"render.modes": []}

    def __init__(self, model, num_envs, *, seed=0, dt=None, maxSteps=250, n_substeps=4, control_mode="faithful",
                 fixed_setpoint=False, flow=None, currentVelScale=1.0, currentTurbScale=2.0, noiseMagCoeffs=0.0,
                 noiseMagActuation=0.0, stopOnBoundsExceeded=True, device=0, env_offset=0, infos="dict",
                 vehicle_params=None, precision="f32", integrator="rk4", report_truncation=False, specialize="auto"):
        cyl = model == "auv_cyl"          # AuvEnvCyl: AuvEnv with way-points (tag/verySimpleAuv_cyl.py)
        if cyl:
            model = "auv"
            maxSteps = 1200 if maxSteps == 250 else maxSteps
        self.model = P.MODEL_NAMES[model] if isinstance(model, str) else int(model)
        self.model_name = {v: k for k, v in P.MODEL_NAMES.items()}[self.model]
        self.num_envs = int(num_envs)
        act, obs, init, words, aux = P.MODEL_DIMS[self.model]
        self.action_space = unit_box(act)
        self.observation_space = unit_box(obs)
        self.infos_mode = infos
        self.report_truncation = bool(report_truncation)
        cm = {"faithful": P.CTRL_FAITHFUL, "zoh": P.CTRL_ZOH}[control_mode] if isinstance(control_mode, str) else control_mode
        use_flow = flow is not None
        if self.model == P.MODEL_AUV and flow is None:
            raise ValueError("AuvEnv needs a turbulence field (flow=ReconstructedFlow(...)): verySimpleAuv.py:102-104")
        kw = {}
        if self.model == P.MODEL_AUV:
            kw["auv"] = P.auv_params(noiseMagCoeffs, noiseMagActuation, stopOnBoundsExceeded, cyl=cyl)
        elif vehicle_params is not None:
            kw["rov6" if self.model == P.MODEL_ROV6 else "rov3"] = vehicle_params
        self.cfg = P.make_config(self.model, self.num_envs, dt=dt, n_substeps=n_substeps, max_steps=maxSteps,
                                 control_mode=cm, fixed_setpoint=fixed_setpoint, auto_reset=True, seed=seed or 0,
                                 use_flow=use_flow, device=device, env_offset=env_offset, precision=precision,
                                 integrator=integrator, **kw)
        self.dt = self.cfg.dt
        self._h = _lib.Handle(self.cfg)
        if specialize is True:
            self._h.specialize()
        elif (specialize == "auto" and self.model == P.MODEL_ROV6 and "/baked/" not in self._h.variant and precision == "f32"
              and integrator == "rk4"):
            try:
                self._h.specialize()
            except _lib.MvrlError as e:   # no compiler on this machine: the ahead-of-time kernel keeps running
                import warnings
                warnings.warn(f"mvrl_specialize failed, keeping the run-time-constant kernel ({self._h.variant}): {e}")
        self.has_reward = self.model == P.MODEL_AUV      # the rigid-body environments return reward = 0. (6DoF.py:575, 3DoF.py:495)
        self.jit = self._h.jit_info()     # compiler / registers / spills of a run-time compiled kernel ("none": ahead of time)
        # a handful of SGPRs parked in VGPR lanes are harmless; scratch memory, VGPR spills or dozens of SGPR spills are not
        if self.jit["specialized"] and (self.jit["scratch_bytes"] > 0 or self.jit["sgpr_spills"] > 16 or self.jit["vgpr_spills"] > 0):
            import warnings
            warnings.warn(f"mvrl_specialize: the {self.jit['compiler']} build of the step kernel spills ({self.jit['sgpr_spills']} SGPR, "
                          f"{self.jit['vgpr_spills']} VGPR, {self.jit['scratch_bytes']} B scratch at {self.jit['min_waves_per_simd']} waves per SIMD) - "
                          "expect it 8-20 % slower than the ahead-of-time kernels; an in-process hiprtc older than the ROCm "
                          "installation's hipcc does that (set MVRL_HIPCC / MVRL_JIT_COMPILER=hipcc)")
        self.flow = flow
        if use_flow:
            if self.model == P.MODEL_AUV:
                flow.scale(11., currentVelScale, currentTurbScale, translate=(-1.65, -1.1))  # verySimpleAuv.py:104
            self._h.set_flow(flow.table_uv(), flow.dt, flow.dx, flow.dy)
        self._pending = False
        self._tensors = None

    # ---- VecEnv API -----------------------------------------------------------------------------
    def reset(self, init=None):
        """All environments start a new episode.  `init` [N, init_dim] gives explicit initial values (see
        include/mvrl.h); otherwise they are drawn on the device (counter-based RNG keyed by seed and env id)."""
        return self._h.reset(init=init)

    def step_async(self, actions):
        # not clipped here: the reference envs apply the raw action (6DoF.py:545-551)
        if self._h.f64:
            self._deferred = actions          # fp64 handles step synchronously in step_wait
        else:
            self._h.step_async(actions)
        self._pending = True

    def step_wait(self):
        obs, rew, done = self._h.step(self._deferred, copy=False) if self._h.f64 else self._h.step_wait(copy=False)   # copied below
        self._pending = False
        bits = done.copy()
        dones = bits != 0
        if self.infos_mode == "lean":
            infos = _LeanInfos(self, bits, self._h.terminal_obs, self.report_truncation)
        else:
            infos = [{} for _ in range(self.num_envs)]
            idx = np.nonzero(dones)[0]
            if len(idx):
                term = self._h.terminal_obs()
                for i in idx:
                    infos[i] = {"terminal_observation": term[i].copy()}
                    if self.report_truncation:
                        infos[i]["TimeLimit.truncated"] = bool(bits[i] & 2)
        return obs.copy(), rew.copy(), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        if self._h is not None:
            self._h.close()
            self._h = None

    def seed(self, seed=None):
        return [seed] * self.num_envs

    def render(self, mode="human"):
        return None

    def get_attr(self, attr_name, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [getattr(self, attr_name)] * n

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [getattr(self, method_name)(*method_args, **method_kwargs)] * n

    def env_is_wrapped(self, wrapper_class, indices=None):
        n = self.num_envs if indices is None else len(np.atleast_1d(indices))
        return [False] * n

    @property
    def unwrapped(self):
        return self

    # ---- extras -----------------------------------------------------------------------------------
    @property
    def variant(self):
        return self._h.variant

    @property
    def handle(self):
        return self._h

    def get_state(self, raw=False):
        """The SoA state planes (include/mvrl.h).  raw=True: verbatim, as the C ABI exchanges them - what a CHECKPOINT should keep: an
        fp32 handle stores the Euler angles as 32-bit binary angles, and the default (decoded to fp32 radians, the reference's
        convention) loses up to 2.4e-7 rad when it is encoded again by a later set_state on another handle or after another
        get_state; the closed loop is chaotic, so only set_state(get_state(raw=True), raw=True) reproduces a run bit for bit."""
        return self._h.get_state(raw=raw)

    def set_state(self, st, raw=False):
        self._h.set_state(st, raw=raw)

    def _ensure_tensors(self):
        if self._tensors is None:
            import torch
            dev = torch.device("cuda", self.cfg.device)
            n = self.num_envs
            rt = torch.float64 if self._h.f64 else torch.float32
            self._tensors = (torch.empty((n, self.observation_space.shape[0]), dtype=rt, device=dev),
                             torch.empty((n,), dtype=rt, device=dev),
                             torch.empty((n,), dtype=torch.uint8, device=dev))
        return self._tensors

    def reset_tensors(self):
        import torch
        obs, _, _ = self._ensure_tensors()
        self._h.reset_dev(None, None, obs.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return obs

    supports_out = True

    def rollout_tensors(self, actions, out=None):
        """K consecutive env steps on device: `actions` [K, N, act_dim] (contiguous) -> (obs [K, N, obs_dim], reward [K, N],
        done_bits [K, N]) - the results of K step_tensors calls (auto-resets included), in one launch for the fp32 6-DoF
        kernels (mvrl_rollout_dev).  For open-loop roll-outs: recorded / random / repeated actions."""
        import torch
        rt = torch.float64 if self._h.f64 else torch.float32
        assert actions.is_cuda and actions.is_contiguous() and actions.dtype == rt and actions.dim() == 3
        k, n, od = actions.shape[0], self.num_envs, self.observation_space.shape[0]
        assert tuple(actions.shape[1:]) == (n, self.action_space.shape[0])
        if out is None:
            out = (torch.empty((k, n, od), dtype=rt, device=actions.device), torch.empty((k, n), dtype=rt, device=actions.device),
                   torch.empty((k, n), dtype=torch.uint8, device=actions.device))
        obs, rew, done = out
        assert obs.is_contiguous() and rew.is_contiguous() and done.is_contiguous()
        assert tuple(obs.shape) == (k, n, od) and tuple(rew.shape) == (k, n) and tuple(done.shape) == (k, n)
        self._h.rollout_dev(actions.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(), k,
                            torch.cuda.current_stream().cuda_stream)
        return obs, rew, done

    def step_range_tensors(self, first, count, actions, out=None):
        """Step only the lanes [first, first + count) (first a multiple of 64) on torch's current stream.  `actions` and the
        returned / `out` tensors are the FULL-batch tensors; rows outside the range are left alone.  Sub-batches stepped on
        different streams form independent chains (see `chains.ChainStepper`)."""
        import torch
        rt = torch.float64 if self._h.f64 else torch.float32
        assert actions.is_cuda and actions.is_contiguous() and actions.dtype == rt
        assert tuple(actions.shape) == (self.num_envs, self.action_space.shape[0])
        obs, rew, done = out if out is not None else self._ensure_tensors()
        self._h.step_range_dev(first, count, actions.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(),
                               torch.cuda.current_stream().cuda_stream)
        return obs, rew, done

    def step_tensors(self, actions, out=None):
        """Device-resident step: `actions` is a contiguous float32 CUDA(HIP) tensor [N, act_dim]; returns
        (obs, reward, done_bits) tensors that are overwritten by the next call.  Enqueued on torch's current stream;
        nothing crosses PCIe.  `out=(obs, reward, done)` makes the kernel write into caller-owned contiguous tensors
        instead (e.g. the views of a gather message, `distributed.OutputGather.out_views()`)."""
        import torch
        assert actions.is_cuda and actions.is_contiguous()
        rt = torch.float64 if self._h.f64 else torch.float32
        assert actions.dtype == rt
        if out is not None:
            obs, rew, done = out
            n, od = self.num_envs, self.observation_space.shape[0]
            assert obs.is_cuda and obs.is_contiguous() and obs.dtype == rt and tuple(obs.shape) == (n, od)
            assert rew.is_cuda and rew.is_contiguous() and rew.dtype == rt and tuple(rew.shape) == (n,)
            assert done.is_cuda and done.is_contiguous() and done.dtype == torch.uint8 and tuple(done.shape) == (n,)
        else:
            obs, rew, done = self._ensure_tensors()
        self._h.step_dev(actions.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr(),
                         torch.cuda.current_stream().cuda_stream)
        return obs, rew, done
