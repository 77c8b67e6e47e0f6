"""The reference's hand-written baseline "agents", with their SB3-like `predict(obs, deterministic=True)` signature,
evaluated on the GPU for whole batches of environments:

    PDController    tag_00_Dec2023_simpleControlTurbulence/verySimpleAuv.py:22-50
    LOSNavigation   dynamicsModel_BlueROV2_Heavy_3DoF.py:584-607 (lineOfSight :517-581)

`predict` takes one observation [obs_dim] or a batch [n, obs_dim] (numpy) and returns `(actions, states)` like the
reference; `predict_tensors` takes/returns torch device tensors so closed-loop roll-outs with MarineVecEnv.step_tensors
never cross PCIe.
"""
import ctypes as C

import numpy as np

from . import _lib

POLICY_PD, POLICY_LOS = 0, 1


class _DevicePolicy(object):
    def __init__(self, kind, num_envs, obs_dim, dt=0.02, P=None, D=None, noise_sigma=0.0, r_nav=0.5, seed=0, device=0):
        self.lib = _lib.load()
        self.n, self.obs_dim, self.device = int(num_envs), int(obs_dim), device
        P3 = (C.c_double * 3)(*([1., 1., 1.] if P is None else [float(v) for v in P]))
        D3 = (C.c_double * 3)(*([0., 0., 0.] if D is None else [float(v) for v in D]))
        h = C.c_void_p()
        _lib.check(self.lib.mvrl_policy_create(kind, device, self.n, self.obs_dim, float(dt), P3, D3, float(noise_sigma),
                                               float(r_nav), int(seed), C.byref(h)))
        self.h = h
        self._act = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.mvrl_policy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _lib.check(self.lib.mvrl_policy_reset(self.h))

    def predict(self, obs, deterministic=True):
        states = obs
        o = np.ascontiguousarray(obs, dtype=np.float32)
        single = o.ndim == 1
        o = o.reshape(-1, o.shape[-1])
        if o.shape != (self.n, self.obs_dim):
            raise ValueError(f"expected obs of shape ({self.n}, {self.obs_dim}) (or ({self.obs_dim},) for 1 env), got {o.shape}")
        a = np.zeros((self.n, 3), np.float32)
        _lib.check(self.lib.mvrl_policy_predict(self.h, o.ctypes.data, a.ctypes.data))
        return (a[0] if single else a), states

    def predict_tensors(self, obs):
        import torch
        assert obs.is_cuda and obs.dtype == torch.float32 and obs.is_contiguous() and tuple(obs.shape) == (self.n, self.obs_dim)
        if self._act is None:
            self._act = torch.empty((self.n, 3), dtype=torch.float32, device=obs.device)
        _lib.check(self.lib.mvrl_policy_predict_dev(self.h, obs.data_ptr(), self._act.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream))
        return self._act


class PDController(_DevicePolicy):
    """PDController(dt, P, D, noiseSigma) of the reference; `num_envs`/`obs_dim` size the batch (1 x 11 by default).
    With noiseSigma the deviates come from the device's counter-based generator, not numpy's global one."""

    def __init__(self, dt, P=[1., 1., 1.], D=[0.05, 0.05, 0.01], noiseSigma=None, num_envs=1, obs_dim=11, seed=0, device=0):
        self.P, self.D, self.dt, self.noiseSigma = np.array(P), np.array(D), dt, noiseSigma
        super().__init__(POLICY_PD, num_envs, obs_dim, dt=dt, P=P, D=D, noise_sigma=noiseSigma or 0.0, seed=seed,
                         device=device)


    def run_episodes(self, venv, n_steps=None):
        """One whole episode of this (noise-free) controller in every env of `venv` (a MarineVecEnv("auv" | "auv_cyl")), fused
        into one kernel launch: the device-side counterpart of resources.evaluate_agent(PDController, env)
        (tag/resources.py:49-102).  Starts from the envs' current states (call venv.reset() / reset_tensors() first), stops
        each env at `done` or after n_steps (default: the env's episode length).  Returns (returns, lengths) torch tensors."""
        import torch
        if self.noiseSigma:
            raise ValueError("run_episodes evaluates the deterministic controller (noiseSigma=None)")
        n = venv.num_envs
        dev = torch.device("cuda", venv.cfg.device)
        ret = torch.empty((n,), dtype=torch.float32, device=dev)
        length = torch.empty((n,), dtype=torch.int32, device=dev)
        venv.handle.auv_pd_episodes_dev(self.P, self.D, self.dt, int(n_steps or venv.cfg.max_steps), ret.data_ptr(),
                                        length.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return ret, length


class LOSNavigation(_DevicePolicy):
    """LOSNavigation() of the reference (Rnav = 0.5, observations of the 3-DoF env)."""

    def __init__(self, num_envs=1, obs_dim=5, Rnav=0.5, device=0):
        super().__init__(POLICY_LOS, num_envs, obs_dim, r_nav=Rnav, device=device)
