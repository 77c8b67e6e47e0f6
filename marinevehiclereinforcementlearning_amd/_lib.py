"""ctypes binding of libmvrl.so (include/mvrl.h) - the only door between Python and the HIP kernels.

The library is loaded lazily on first use.  If it has not been built, or no HIP device is present, the call
raises: there is deliberately no CPU fallback in the product path.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import params as P

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmvrl.so")

_lib = None

ERRORS = {-1: "EINVAL", -2: "ENODEV", -3: "ENOMEM", -4: "EHIP", -5: "ESTATE"}

# every symbol include/mvrl.h declares (checked by tests/test_abi.py against the header text)
SYMBOLS = [
    "mvrl_abi_version", "mvrl_device_count", "mvrl_last_error", "mvrl_model_dims", "mvrl_aux_dim", "mvrl_variant",
    "mvrl_create", "mvrl_destroy", "mvrl_set_flow", "mvrl_set_flow_dev", "mvrl_reset", "mvrl_reset_dev", "mvrl_step",
    "mvrl_step_async", "mvrl_step_wait", "mvrl_step_dev", "mvrl_step_range_dev", "mvrl_get_terminal_obs", "mvrl_get_terminal_obs_dev",
    "mvrl_get_state", "mvrl_set_state", "mvrl_enable_aux", "mvrl_get_aux", "mvrl_flow_interp", "mvrl_flow_reconstruct",
    "mvrl_fill_uniform_dev", "mvrl_timing_begin", "mvrl_timing_end", "mvrl_launch_count", "mvrl_delay_dev", "mvrl_dev_alloc", "mvrl_dev_free",
    "mvrl_dev_upload", "mvrl_dev_download", "mvrl_synchronize",
    # fp64 twins of the host-buffer entry points + RK45 diagnostics
    "mvrl_set_flow_f64", "mvrl_reset_f64", "mvrl_step_f64", "mvrl_get_terminal_obs_f64", "mvrl_get_state_f64",
    "mvrl_set_state_f64", "mvrl_get_aux_f64", "mvrl_get_nfev", "mvrl_derivs", "mvrl_derivs_f64", "mvrl_derivs_cur", "mvrl_derivs_cur_f64", "mvrl_vehicle_ops", "mvrl_vehicle_ops_f64", "mvrl_specialize", "mvrl_jit_compile_check",
    "mvrl_jit_info", "mvrl_jit_compile_check2", "mvrl_jit_child_env", "mvrl_force_components", "mvrl_force_components_f64", "mvrl_mass_solve", "mvrl_mass_solve_f64", "mvrl_observe", "mvrl_observe_f64", "mvrl_host_buffers", "mvrl_default_config",
    # several devices behind one object in one process (csrc/mvrl_group.hip; Python: group.DeviceGroup)
    "mvrl_group_shard_range", "mvrl_group_message_layout", "mvrl_group_create", "mvrl_group_destroy", "mvrl_group_last_error", "mvrl_group_info",
    "mvrl_group_shard", "mvrl_group_set_flow", "mvrl_group_reset", "mvrl_group_step_dev", "mvrl_group_gather_dev", "mvrl_group_wait",
    "mvrl_group_gathered_event", "mvrl_group_root_views", "mvrl_group_download", "mvrl_group_scatter_actions_dev", "mvrl_group_fill_actions",
    "mvrl_group_synchronize",
    "mvrl_auv_pd_episodes_dev", "mvrl_rollout_dev", "mvrl_replay_add_sym_dev", "mvrl_policy_create", "mvrl_policy_destroy", "mvrl_policy_reset", "mvrl_policy_predict", "mvrl_policy_predict_dev",
]


class MvrlError(RuntimeError):
    pass


def load(path=None):
    """dlopen libmvrl.so and declare the prototypes.  Raises MvrlError if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("MVRL_LIB") or LIB_PATH
    # One HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.  If libmvrl.so pulled in the
    # system copy first, a later `import torch` would bring up a second runtime that sees no GPU.  Importing torch
    # first (when it is installed) makes both share torch's copy (same soname).
    if "torch" not in sys.modules and os.environ.get("MVRL_NO_TORCH_PRELOAD") is None:
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001 - torch is optional for the host-buffer API
            pass
    if not os.path.exists(path):
        raise MvrlError(f"{path} not found - build it with `python -m marinevehiclereinforcementlearning_amd.build` "
                        "(needs hipcc; there is no CPU fallback)")
    lib = C.CDLL(path)
    vp, i32, i64, u64, fp = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float
    lib.mvrl_abi_version.restype = C.c_int
    lib.mvrl_device_count.restype = C.c_int
    lib.mvrl_last_error.restype = C.c_char_p
    lib.mvrl_last_error.argtypes = [vp]
    lib.mvrl_variant.restype = C.c_char_p
    lib.mvrl_variant.argtypes = [vp]
    lib.mvrl_model_dims.argtypes = [i32] + [C.POINTER(i32)] * 4
    lib.mvrl_aux_dim.argtypes = [i32]
    lib.mvrl_create.argtypes = [C.POINTER(P.Config), C.POINTER(vp)]
    lib.mvrl_destroy.argtypes = [vp]
    lib.mvrl_destroy.restype = None
    lib.mvrl_set_flow.argtypes = [vp, vp, C.POINTER(P.FlowDesc)]
    lib.mvrl_set_flow_dev.argtypes = [vp, vp, C.POINTER(P.FlowDesc)]
    lib.mvrl_reset.argtypes = [vp, vp, vp, vp]
    lib.mvrl_reset_dev.argtypes = [vp, vp, vp, vp, vp]
    lib.mvrl_step.argtypes = [vp, vp, vp, vp, vp]
    lib.mvrl_step_async.argtypes = [vp, vp]
    lib.mvrl_step_wait.argtypes = [vp, vp, vp, vp]
    lib.mvrl_step_dev.argtypes = [vp, vp, vp, vp, vp, vp]
    lib.mvrl_step_range_dev.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp]
    lib.mvrl_get_terminal_obs.argtypes = [vp, vp]
    lib.mvrl_get_terminal_obs_dev.argtypes = [vp, vp, vp]
    lib.mvrl_get_state.argtypes = [vp, vp, C.c_size_t]
    lib.mvrl_set_state.argtypes = [vp, vp, C.c_size_t]
    lib.mvrl_enable_aux.argtypes = [vp, i32]
    lib.mvrl_get_aux.argtypes = [vp, vp]
    lib.mvrl_flow_interp.argtypes = [i32, vp, C.POINTER(P.FlowDesc), i32, vp, vp, vp, i64, vp]
    lib.mvrl_flow_reconstruct.argtypes = [i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]
    lib.mvrl_fill_uniform_dev.argtypes = [vp, vp, i64, u64, u64, fp, fp, vp]
    lib.mvrl_timing_begin.argtypes = [vp, vp]
    lib.mvrl_launch_count.argtypes = [vp, C.POINTER(i64)]
    lib.mvrl_delay_dev.argtypes = [vp, i32, vp]
    lib.mvrl_timing_end.argtypes = [vp, vp, C.POINTER(fp), C.POINTER(i64)]
    lib.mvrl_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.mvrl_dev_free.argtypes = [vp, vp]
    lib.mvrl_dev_upload.argtypes = [vp, vp, vp, C.c_size_t]
    lib.mvrl_dev_download.argtypes = [vp, vp, vp, C.c_size_t]
    lib.mvrl_synchronize.argtypes = [vp]
    lib.mvrl_set_flow_f64.argtypes = [vp, vp, C.POINTER(P.FlowDesc)]
    lib.mvrl_reset_f64.argtypes = [vp, vp, vp, vp]
    lib.mvrl_step_f64.argtypes = [vp, vp, vp, vp, vp]
    lib.mvrl_get_terminal_obs_f64.argtypes = [vp, vp]
    lib.mvrl_get_state_f64.argtypes = [vp, vp, C.c_size_t]
    lib.mvrl_set_state_f64.argtypes = [vp, vp, C.c_size_t]
    lib.mvrl_get_aux_f64.argtypes = [vp, vp]
    lib.mvrl_get_nfev.argtypes = [vp, vp]
    lib.mvrl_derivs.argtypes = [vp, i64] + [vp] * 10
    lib.mvrl_derivs_f64.argtypes = [vp, i64] + [vp] * 10
    lib.mvrl_derivs_cur.argtypes = [vp, i64] + [vp] * 11
    lib.mvrl_derivs_cur_f64.argtypes = [vp, i64] + [vp] * 11
    lib.mvrl_vehicle_ops.argtypes = [vp, i64] + [vp] * 8
    lib.mvrl_specialize.argtypes = [vp]
    lib.mvrl_jit_compile_check.argtypes = [vp, C.c_int, vp, vp, C.c_size_t]
    lib.mvrl_jit_info.argtypes = [vp, C.POINTER(P.JitReport)]
    lib.mvrl_jit_compile_check2.argtypes = [vp, C.c_int, C.POINTER(P.JitReport), vp, C.c_size_t]
    lib.mvrl_jit_child_env.argtypes = [vp, C.c_size_t]
    lib.mvrl_vehicle_ops_f64.argtypes = [vp, i64] + [vp] * 8
    lib.mvrl_force_components.argtypes = [vp, i64] + [vp] * 4
    lib.mvrl_force_components_f64.argtypes = [vp, i64] + [vp] * 4
    lib.mvrl_mass_solve.argtypes = [vp, i64, vp, vp]
    lib.mvrl_mass_solve_f64.argtypes = [vp, i64, vp, vp]
    lib.mvrl_observe.argtypes = [vp, vp]
    lib.mvrl_host_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    lib.mvrl_default_config.argtypes = [i32, i64, vp]
    lib.mvrl_observe_f64.argtypes = [vp, vp]
    lib.mvrl_rollout_dev.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    lib.mvrl_auv_pd_episodes_dev.argtypes = [vp, vp, vp, C.c_double, i32, vp, vp, vp]
    lib.mvrl_replay_add_sym_dev.argtypes = [i32] + [vp] * 5 + [i64] + [vp] * 6 + [i64, i64, i32, i32, vp]
    lib.mvrl_policy_create.argtypes = [i32, i32, i64, i32, C.c_double, vp, vp, C.c_double, C.c_double, u64, C.POINTER(vp)]
    lib.mvrl_policy_destroy.argtypes = [vp]
    lib.mvrl_policy_destroy.restype = None
    lib.mvrl_policy_reset.argtypes = [vp]
    lib.mvrl_policy_predict.argtypes = [vp, vp, vp]
    lib.mvrl_policy_predict_dev.argtypes = [vp, vp, vp, vp]
    if lib.mvrl_abi_version() != P.ABI_VERSION:
        raise MvrlError("libmvrl.so ABI version does not match the Python package")
    _lib = lib
    return lib


def check(rc, handle=None):
    if rc == 0:
        return
    msg = load().mvrl_last_error(handle)
    raise MvrlError(f"libmvrl {ERRORS.get(rc, rc)}: {msg.decode() if msg else ''}")


def device_count():
    return int(load().mvrl_device_count())


def _real(a, dtype, shape=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def _f32(a, shape=None):
    return _real(a, np.float32, shape)


def flow_desc(n_t, n_y, n_x, dt, dx, dy):
    d = P.FlowDesc()
    d.n_t, d.n_y, d.n_x, d.dt, d.dx, d.dy = int(n_t), int(n_y), int(n_x), float(dt), float(dx), float(dy)
    return d


class Handle:
    """Owning wrapper of one mvrl_handle (one shard of environments on one GPU).

    The handle's precision (cfg.precision) decides the dtype of every real-valued array that crosses the ABI
    (float32 -> the plain entry points, float64 -> the *_f64 ones)."""

    def __init__(self, cfg):
        self.lib = load()
        self.cfg = cfg
        self.model = int(cfg.model)
        self.n = int(cfg.n_envs)
        self.f64 = int(cfg.precision) == P.PREC_F64
        self.dtype = np.float64 if self.f64 else np.float32
        self.itype = np.int64 if self.f64 else np.int32   # view of the step-counter plane
        self._sfx = "_f64" if self.f64 else ""
        self.act_dim, self.obs_dim, self.init_dim, self.state_words, self.aux_dim = P.MODEL_DIMS[self.model]
        h = C.c_void_p()
        check(self.lib.mvrl_create(C.byref(cfg), C.byref(h)))
        self.h = h
        # numpy views of the handle's pinned staging block (mvrl_host_buffers): the step writes its outputs there and reads the
        # actions from there - no staging copies inside the library
        pa, po, pr, pd = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        check(self.lib.mvrl_host_buffers(self.h, C.byref(pa), C.byref(po), C.byref(pr), C.byref(pd)), self.h)
        ct = C.c_double if self.f64 else C.c_float

        def view(ptr, shape, ctype, dtype):
            cnt = int(np.prod(shape))
            return np.frombuffer((ctype * cnt).from_address(ptr.value), dtype=dtype).reshape(shape)
        self._act_in = view(pa, (self.n, self.act_dim), ct, self.dtype)
        self._obs = view(po, (self.n, self.obs_dim), ct, self.dtype)
        self._rew = view(pr, (self.n,), ct, self.dtype)
        self._done = view(pd, (self.n,), C.c_uint8, np.uint8)

    def _fn(self, name):
        return getattr(self.lib, name + self._sfx)

    # -- lifetime ---------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self._act_in = self._obs = self._rew = self._done = None     # views of memory the handle owns
            self.lib.mvrl_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def variant(self):
        return self.lib.mvrl_variant(self.h).decode()

    # -- flow -------------------------------------------------------------------------------------
    def set_flow(self, table_uv, dt, dx, dy):
        t = _real(table_uv, self.dtype)
        if t.ndim != 4 or t.shape[3] != 2:
            raise ValueError("flow table must be [n_t, n_y, n_x, 2]")
        d = flow_desc(t.shape[0], t.shape[1], t.shape[2], dt, dx, dy)
        check(self._fn("mvrl_set_flow")(self.h, t.ctypes.data, C.byref(d)), self.h)

    def set_flow_dev(self, ptr, n_t, n_y, n_x, dt, dx, dy):
        d = flow_desc(n_t, n_y, n_x, dt, dx, dy)
        check(self.lib.mvrl_set_flow_dev(self.h, ptr, C.byref(d)), self.h)

    # -- host-buffer API --------------------------------------------------------------------------
    def reset(self, mask=None, init=None, obs_out=None, copy=True):
        """copy=False hands out the handle's pinned staging block itself (see `step`); a caller-supplied obs_out is returned as it is."""
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.n)
        ini = None if init is None else _real(init, self.dtype, (self.n, self.init_dim))
        obs = self._obs if obs_out is None else obs_out
        assert obs.dtype == self.dtype
        check(self._fn("mvrl_reset")(self.h, None if m is None else m.ctypes.data,
                                     None if ini is None else ini.ctypes.data, obs.ctypes.data), self.h)
        return obs.copy() if (copy and obs_out is None) else obs

    def _stage_actions(self, actions):
        """actions -> the handle's pinned action block (one copy, with dtype conversion if needed); returns its address or None"""
        if actions is None:
            return None
        np.copyto(self._act_in, np.asarray(actions).reshape(self.n, self.act_dim), casting="same_kind")
        return self._act_in.ctypes.data

    def step(self, actions, copy=True):
        """One env step through host buffers: (obs [n, obs_dim], reward [n], done bits [n]) as FRESH arrays, like the reference's step
        (SURVEY 8(b) "Ownership").  copy=False: zero-copy VIEWS of the handle's pinned staging block (mvrl_host_buffers) - overwritten
        by the next step / reset and INVALID once the handle is closed or collected (the block is freed with it); for callers that
        consume the outputs at once (MarineVecEnv, which hands out its own copies; the PCIe-inclusive benchmarks)."""
        check(self._fn("mvrl_step")(self.h, self._stage_actions(actions), self._obs.ctypes.data,
                                    self._rew.ctypes.data, self._done.ctypes.data), self.h)
        if copy:
            return self._obs.copy(), self._rew.copy(), self._done.copy()
        return self._obs, self._rew, self._done

    def step_async(self, actions):
        if self.f64:
            raise MvrlError("step_async/step_wait are fp32-only; fp64 handles use step()")
        if getattr(self, "_async_pending", False):
            # refused BEFORE the actions are staged: for small batches the in-flight kernel reads the pinned action block directly, and
            # staging a second batch over it would change the pending step's actions (the C side would only then answer MVRL_ESTATE)
            raise MvrlError("step_async: a step is already pending (call step_wait first)")
        check(self.lib.mvrl_step_async(self.h, self._stage_actions(actions)), self.h)
        self._async_pending = True

    def step_wait(self, copy=True):
        check(self.lib.mvrl_step_wait(self.h, self._obs.ctypes.data, self._rew.ctypes.data, self._done.ctypes.data),
              self.h)
        self._async_pending = False
        if copy:
            return self._obs.copy(), self._rew.copy(), self._done.copy()
        return self._obs, self._rew, self._done

    def terminal_obs(self):
        out = np.zeros((self.n, self.obs_dim), self.dtype)
        check(self._fn("mvrl_get_terminal_obs")(self.h, out.ctypes.data), self.h)
        return out

    def get_state(self, raw=False):
        """The SoA state planes [state_words, n_envs] (include/mvrl.h).  An fp32 handle keeps the Euler angles of the rigid-body
        models as binary angles (uint32 bit patterns); by default they are handed out DECODED - fp32 radians in [0, 2 pi), the
        reference's convention (6DoF.py:560) - and `set_state` encodes them again.  Angle entries that come back to `set_state`
        unchanged are restored bit for bit (the raw planes of the last `get_state` are remembered), so get -> edit other planes ->
        set is an exact round trip.  raw=True: the planes verbatim, as the C ABI exchanges them."""
        buf = np.zeros((self.state_words, self.n), self.dtype)
        check(self._fn("mvrl_get_state")(self.h, buf.ctypes.data, buf.size), self.h)
        planes = () if self.f64 else P.ANGLE_PLANES[self.model]
        if raw or not planes:
            return buf
        idx = list(planes)
        bits = buf[idx].view(np.uint32).copy()
        dec = (bits.astype(np.float64) * (2.0 * np.pi / 4294967296.0)).astype(np.float32)
        dec[dec >= np.float32(2.0 * np.pi)] = 0.0          # 2 pi - 1e-9 rounds up to fp32(2 pi): that is the angle 0
        buf[idx] = dec
        self._angle_cache = (dec.copy(), bits)
        return buf

    def set_state(self, buf, raw=False):
        b = _real(buf, self.dtype, (self.state_words, self.n))
        planes = () if self.f64 else P.ANGLE_PLANES[self.model]
        if planes and not raw:
            b = b.copy()
            idx = list(planes)
            ang = b[idx]
            bits = np.round(np.mod(ang.astype(np.float64), 2.0 * np.pi) * (4294967296.0 / (2.0 * np.pi))).astype(np.uint64)
            bits = (bits & 0xFFFFFFFF).astype(np.uint32)
            cache = getattr(self, "_angle_cache", None)
            if cache is not None and cache[0].shape == ang.shape:
                same = ang == cache[0]
                bits[same] = cache[1][same]
            b[idx] = bits.view(np.float32)
        check(self._fn("mvrl_set_state")(self.h, b.ctypes.data, b.size), self.h)

    def step_counter(self, state=None):
        st = self.get_state() if state is None else state
        return st[P.STATE_PLANES[self.model]["istep"]].view(self.itype)

    def episode_counter(self, state=None):
        """Number of resets each env has gone through = the counter of its Philox stream."""
        st = self.get_state() if state is None else state
        return st[P.STATE_PLANES[self.model]["episode"]].view(self.itype)

    def derivs(self, t, y, sp, eold=None, eint=None, told=None, has_old=None, cur=None):
        """vehicle.derivs(t, y) for n tuples (6DoF.py:406-442 / 3DoF.py:128-296).  Returns dict(dy, eold, eint, told, gcf,
        rpm) - eold/eint/told are the controller memory AFTER the call.  cur [n, 2]: a global-frame water current per tuple."""
        dof = 6 if self.model == P.MODEL_ROV6 else 3
        nthr = 8 if dof == 6 else 4
        y = np.ascontiguousarray(np.atleast_2d(y), self.dtype)
        n = y.shape[0]
        t = np.ascontiguousarray(np.broadcast_to(np.asarray(t, np.float64), (n,)))      # times are fp64 at the ABI for both precisions
        sp = np.ascontiguousarray(np.atleast_2d(sp), self.dtype)
        eo = np.zeros((n, dof), self.dtype) if eold is None else np.array(np.atleast_2d(eold), self.dtype)
        ei = np.zeros((n, dof), self.dtype) if eint is None else np.array(np.atleast_2d(eint), self.dtype)
        to = np.zeros(n, np.float64) if told is None else np.array(np.broadcast_to(np.asarray(told, np.float64), (n,)))
        ho = (np.zeros(n, np.uint8) if (has_old is None and eold is None) else
              np.ones(n, np.uint8) if has_old is None else np.ascontiguousarray(np.broadcast_to(has_old, (n,)), np.uint8))
        assert y.shape == (n, 2 * dof) and sp.shape == (n, dof) and eo.shape == (n, dof) and ei.shape == (n, dof)
        dy = np.zeros((n, 2 * dof), self.dtype); gcf = np.zeros((n, dof), self.dtype); rpm = np.zeros((n, nthr), self.dtype)
        if cur is None:
            check(self._fn("mvrl_derivs")(self.h, n, t.ctypes.data, y.ctypes.data, sp.ctypes.data, eo.ctypes.data, ei.ctypes.data,
                                          to.ctypes.data, ho.ctypes.data, dy.ctypes.data, gcf.ctypes.data, rpm.ctypes.data), self.h)
        else:   # water current (u_c, v_c) in the global frame per tuple: the reference's velCurrent hook (mvrl_derivs_cur)
            cu = _real(np.atleast_2d(cur), self.dtype, (n, 2))
            check(self._fn("mvrl_derivs_cur")(self.h, n, t.ctypes.data, y.ctypes.data, sp.ctypes.data, cu.ctypes.data, eo.ctypes.data,
                                              ei.ctypes.data, to.ctypes.data, ho.ctypes.data, dy.ctypes.data, gcf.ctypes.data,
                                              rpm.ctypes.data), self.h)
        return dict(dy=dy, eold=eo, eint=ei, told=to, gcf=gcf, rpm=rpm)

    def vehicle_ops(self, angles, gcf=None, rpm=None, vel=None, want=("axes", "rpm", "rhs", "thruster_h")):
        """Body axes / allocateThrust / forceModel of the 6-DoF vehicle for n tuples (mvrl_vehicle_ops).  Returns a dict with
        the requested outputs that the inputs allow: axes [n,3,3] always; rpm [n,8] needs gcf; rhs / thruster_h [n,6] use
        rpm if given, else the allocation of gcf."""
        ang = np.ascontiguousarray(np.atleast_2d(angles), self.dtype)
        n = ang.shape[0]

        def arr(x, w):
            return None if x is None else _real(np.atleast_2d(x), self.dtype, (n, w))
        g, r, v = arr(gcf, 6), arr(rpm, 8), arr(vel, 6)
        out = {}
        if "axes" in want:
            out["axes"] = np.zeros((n, 3, 3), self.dtype)
        if "rpm" in want and g is not None:
            out["rpm"] = np.zeros((n, 8), self.dtype)
        if g is not None or r is not None or v is not None:
            for k in ("rhs", "thruster_h"):
                if k in want:
                    out[k] = np.zeros((n, 6), self.dtype)

        def p(a):
            return None if a is None else a.ctypes.data
        check(self._fn("mvrl_vehicle_ops")(self.h, n, ang.ctypes.data, p(g), p(r), p(v), p(out.get("axes")), p(out.get("rpm")),
                                           p(out.get("rhs")), p(out.get("thruster_h"))), self.h)
        return out

    def force_components(self, angles, vel, rpm):
        """forceModel(..., retComp=True) for n tuples: [n, 6, 5] = columns -Crb.vel, -Ca.vel, -D.vel, G, H (mvrl_force_components)."""
        ang = np.ascontiguousarray(angles, self.dtype).reshape(-1, 3)
        n = len(ang)
        v = np.ascontiguousarray(vel, self.dtype).reshape(n, 6)
        r = np.ascontiguousarray(rpm, self.dtype).reshape(n, 8)
        out = np.zeros((n, 6, 5), self.dtype)
        check(self._fn("mvrl_force_components")(self.h, n, ang.ctypes.data, v.ctypes.data, r.ctypes.data, out.ctypes.data), self.h)
        return out

    def observe(self):
        """dataToState of every env's current state (mvrl_observe): [n_envs, obs_dim]."""
        out = np.zeros((self.n, self.obs_dim), self.dtype)
        check(self._fn("mvrl_observe")(self.h, out.ctypes.data), self.h)
        return out

    def mass_solve(self, rhs):
        """acc = np.linalg.solve(M, RHS) for n right-hand sides through the M^-1 of the handle's step kernel (mvrl_mass_solve)."""
        r = np.ascontiguousarray(rhs, self.dtype).reshape(-1, 6)
        out = np.zeros_like(r)
        check(self._fn("mvrl_mass_solve")(self.h, len(r), r.ctypes.data, out.ctypes.data), self.h)
        return out

    def specialize(self):
        """Compile the 6-DoF step kernel for this handle's constants (hipcc child process, hiprtc fallback) and switch to it (mvrl_specialize)."""
        check(self.lib.mvrl_specialize(self.h), self.h)
        return self.variant

    def jit_info(self):
        """What mvrl_specialize built (mvrl_jit_info): dict(compiler, specialized, min_waves_per_simd, vgprs, sgprs, vgpr_spills,
        sgpr_spills, scratch_bytes, lds_bytes, code_bytes); compiler "none" for an ahead-of-time kernel."""
        rep = P.JitReport()
        check(self.lib.mvrl_jit_info(self.h, C.byref(rep)), self.h)
        return rep.as_dict()

    def enable_aux(self, on=True):
        check(self.lib.mvrl_enable_aux(self.h, 1 if on else 0), self.h)

    def get_aux(self):
        out = np.zeros((self.n, self.aux_dim), self.dtype)
        check(self._fn("mvrl_get_aux")(self.h, out.ctypes.data), self.h)
        return out

    def get_nfev(self):
        out = np.zeros(self.n, np.int32)
        check(self.lib.mvrl_get_nfev(self.h, out.ctypes.data), self.h)
        return out

    # -- device-pointer API (ints = raw device addresses, e.g. torch.Tensor.data_ptr()) -------------
    def step_dev(self, actions_ptr, obs_ptr, reward_ptr, done_ptr, stream=None):
        check(self.lib.mvrl_step_dev(self.h, actions_ptr, obs_ptr, reward_ptr, done_ptr, stream), self.h)

    def step_range_dev(self, first, count, actions_ptr, obs_ptr, reward_ptr, done_ptr, stream=None):
        """Step lanes [first, first + count) only; pointers are the bases of the full-batch arrays."""
        check(self.lib.mvrl_step_range_dev(self.h, int(first), int(count), actions_ptr, obs_ptr, reward_ptr, done_ptr, stream), self.h)

    def rollout_dev(self, actions_ptr, obs_ptr, reward_ptr, done_ptr, k_steps, stream=None):
        check(self.lib.mvrl_rollout_dev(self.h, actions_ptr, obs_ptr, reward_ptr, done_ptr, int(k_steps), stream), self.h)

    def reset_dev(self, mask_ptr, init_ptr, obs_ptr, stream=None):
        check(self.lib.mvrl_reset_dev(self.h, mask_ptr, init_ptr, obs_ptr, stream), self.h)

    def terminal_obs_dev(self, obs_ptr, stream=None):
        check(self.lib.mvrl_get_terminal_obs_dev(self.h, obs_ptr, stream), self.h)

    def fill_uniform_dev(self, ptr, n, seed, counter, lo=-1.0, hi=1.0, stream=None):
        check(self.lib.mvrl_fill_uniform_dev(self.h, ptr, n, seed, counter, lo, hi, stream), self.h)

    def auv_pd_episodes_dev(self, P_gain, D_gain, policy_dt, n_steps, returns_ptr, lengths_ptr, stream=None):
        Pa = (C.c_double * 3)(*[float(v) for v in P_gain])
        Da = (C.c_double * 3)(*[float(v) for v in D_gain])
        check(self.lib.mvrl_auv_pd_episodes_dev(self.h, Pa, Da, float(policy_dt), int(n_steps), returns_ptr, lengths_ptr, stream),
              self.h)

    def count_launches(self, k):
        """Step launches replayed from a captured graph do not pass through the library: account for them here so
        that timing_end() reports the right launch count."""
        self._graph_launches = getattr(self, "_graph_launches", 0) + int(k)

    def delay_dev(self, microseconds, stream=None):
        check(self.lib.mvrl_delay_dev(self.h, int(microseconds), stream), self.h)

    def launch_count(self):
        nl = C.c_int64()
        check(self.lib.mvrl_launch_count(self.h, C.byref(nl)), self.h)
        return nl.value

    def timing_begin(self, stream=None):
        self._graph_launches = 0
        check(self.lib.mvrl_timing_begin(self.h, stream), self.h)

    def timing_end(self, stream=None):
        ms, nl = C.c_float(), C.c_int64()
        check(self.lib.mvrl_timing_end(self.h, stream, C.byref(ms), C.byref(nl)), self.h)
        return ms.value, nl.value + getattr(self, "_graph_launches", 0)

    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        check(self.lib.mvrl_dev_alloc(self.h, nbytes, C.byref(p)), self.h)
        return p.value

    def dev_free(self, ptr):
        check(self.lib.mvrl_dev_free(self.h, ptr), self.h)

    def dev_upload(self, ptr, arr):
        a = np.ascontiguousarray(arr)
        check(self.lib.mvrl_dev_upload(self.h, ptr, a.ctypes.data, a.nbytes), self.h)

    def dev_download(self, ptr, arr):
        assert arr.flags["C_CONTIGUOUS"]
        check(self.lib.mvrl_dev_download(self.h, arr.ctypes.data, ptr, arr.nbytes), self.h)
        return arr

    def synchronize(self):
        check(self.lib.mvrl_synchronize(self.h), self.h)


def flow_interp(table, dt, dx, dy, t, x, y, device=0):
    """ReconstructedFlow.interp on the GPU for arrays of query points (tag/flowGenerator.py:97-136)."""
    lib = load()
    tab = _f32(table)
    if tab.ndim != 4 or not (1 <= tab.shape[3] <= 4):
        raise ValueError("table must be [n_t, n_y, n_x, n_comp<=4]")
    t, x, y = _f32(np.ravel(t)), _f32(np.ravel(x)), _f32(np.ravel(y))
    out = np.zeros((len(t), tab.shape[3]), np.float32)
    d = flow_desc(tab.shape[0], tab.shape[1], tab.shape[2], dt, dx, dy)
    check(lib.mvrl_flow_interp(device, tab.ctypes.data, C.byref(d), tab.shape[3], t.ctypes.data, x.ctypes.data,
                               y.ctypes.data, len(t), out.ctypes.data))
    return out


def flow_reconstruct(modes, coeffs, ltm, scale_mul, scale_add, device=0):
    """flowData[t,j,i,c] = mul[c] * (Re(modes[j,i,c,:] @ coeffs[:,t]) + ltm[j,i,c]) + add[c] on the GPU
    (tag/flowGenerator.py:19-23 fused with the affine map of scale(), :76-92)."""
    lib = load()
    ny, nx, nc, K = modes.shape
    assert nc == 3 and coeffs.shape[0] == K
    nT = coeffs.shape[1]
    m2 = modes.reshape(ny * nx * 3, K)
    mre, mim = _f32(m2.real), _f32(m2.imag)
    cre, cim = _f32(coeffs.real), _f32(coeffs.imag)
    l = _f32(ltm.reshape(-1))
    mul, add = _f32(scale_mul, (3,)), _f32(scale_add, (3,))
    out = np.zeros((nT, ny, nx, 3), np.float32)
    check(lib.mvrl_flow_reconstruct(device, mre.ctypes.data, mim.ctypes.data, cre.ctypes.data, cim.ctypes.data,
                                    l.ctypes.data, ny * nx * 3, K, nT, mul.ctypes.data, add.ctypes.data,
                                    out.ctypes.data))
    return out
