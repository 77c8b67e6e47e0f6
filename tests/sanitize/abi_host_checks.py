"""Run under the host-ASan build of libmvrl (tests/sanitize/test_sanitizers.py: LD_PRELOAD of clang's ASan runtime, MVRL_LIB=...hostasan.so,
no GPU): every argument-checking and host-only path of the C ABI - bad configurations (the cases of tests/test_abi.py), NULL and
mis-sized arguments, parameter narrowing for all three models, the JIT driver with both compilers, the code-object note parser, the
child-environment scrubber.  Any ASan / UBSan finding aborts the process."""
import ctypes as C
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
os.environ["MVRL_NO_TORCH_PRELOAD"] = "1"
from marinevehiclereinforcementlearning_amd import _lib, params as P  # noqa: E402

assert "hostasan" in os.environ["MVRL_LIB"]
lib = _lib.load()
assert lib.mvrl_abi_version() == P.ABI_VERSION
h = C.c_void_p()

# ---- configurations that must be rejected before anything touches a device (tests/test_abi.py::test_bad_config_is_rejected)
bad = []
c = P.make_config("rov6", 16); c.abi_version = 99; bad.append(c)
bad.append(P.make_config("rov6", 0))
bad.append(P.make_config("rov6", 40_000_000))
bad.append(P.make_config("rov3", 8, n_substeps=0))
c = P.make_config("auv", 8); c.model = 7; bad.append(c)
c = P.make_config("rov6", 8); c.dt = -0.1; bad.append(c)
c = P.make_config("rov6", 8); c.precision = 5; bad.append(c)
for cfg in bad:
    rc = lib.mvrl_create(C.byref(cfg), C.byref(h))
    assert rc != 0, "accepted a bad configuration"
    assert lib.mvrl_last_error(None)
assert lib.mvrl_create(None, C.byref(h)) != 0 and lib.mvrl_create(C.byref(P.make_config("rov6", 8)), None) != 0

# ---- valid configurations of every model / flavour: parameter narrowing and variant selection run, then ENODEV (no GPU here)
n_dev = lib.mvrl_device_count()
for cfg in (P.make_config("rov6", 64), P.make_config("rov6", 64, rov6=P.rov6_params(m=12.0, Xuu=-19.0)),
            P.make_config("rov6", 64, rov6=P.rov6_params(CG=[0.01, -0.015, 0.04], Yr=-0.3)), P.make_config("rov3", 64),
            P.make_config("rov3", 64, rov3=P.rov3_params(m=12.0)), P.make_config("auv", 64), P.make_config("auv", 64, auv=P.auv_params(cyl=True)),
            P.make_config("rov6", 64, precision="f64", integrator="rk45")):
    rc = lib.mvrl_create(C.byref(cfg), C.byref(h))
    if n_dev == 0:
        assert rc != 0 and b"" != lib.mvrl_last_error(None)
    elif rc == 0:
        lib.mvrl_destroy(h)

# ---- NULL handles and arguments on the entry points
one = (C.c_float * 64)()
for call in (lambda: lib.mvrl_step(None, one, one, one, one), lambda: lib.mvrl_reset(None, None, None, one),
             lambda: lib.mvrl_get_state(None, one, 64), lambda: lib.mvrl_set_state(None, one, 64), lambda: lib.mvrl_specialize(None),
             lambda: lib.mvrl_jit_info(None, None), lambda: lib.mvrl_enable_aux(None, 1), lambda: lib.mvrl_synchronize(None),
             lambda: lib.mvrl_step_dev(None, None, None, None, None, None), lambda: lib.mvrl_jit_child_env(None, 0),
             lambda: lib.mvrl_force_components(None, 1, one, one, one, one), lambda: lib.mvrl_derivs(None, 1, *([one] * 10)),
             lambda: lib.mvrl_mass_solve(None, 1, one, one), lambda: lib.mvrl_observe(None, one),
             lambda: lib.mvrl_host_buffers(None, None, None, None, None), lambda: lib.mvrl_default_config(9, 1, C.byref(P.Config())),
             lambda: lib.mvrl_default_config(2, 1, None)):
    assert call() != 0
assert lib.mvrl_model_dims(7, None, None, None, None) != 0 and lib.mvrl_aux_dim(9) < 0
a, o_, i_, w = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
for model in (0, 1, 2):
    assert lib.mvrl_model_dims(model, C.byref(a), C.byref(o_), C.byref(i_), C.byref(w)) == 0
for name, model in P.MODEL_NAMES.items():       # the generated default blocks, copied through the ABI
    c = P.Config()
    assert lib.mvrl_default_config(model, 77, C.byref(c)) == 0 and bytes(c) == bytes(P.make_config(name, 77))
lib.mvrl_destroy(None)
assert lib.mvrl_variant(None) == b""

# ---- the JIT driver: text generation, hipcc child (scrubbed environment, private TMPDIR), hiprtc, and the note parser
buf = C.create_string_buffer(1 << 18)
assert lib.mvrl_jit_child_env(buf, len(buf)) == 0 and b"PATH=" in buf.value
small = C.create_string_buffer(8)
assert lib.mvrl_jit_child_env(small, len(small)) != 0          # too small a buffer is reported, not overrun
for compiler in ("hipcc", "hiprtc"):
    os.environ["MVRL_JIT_COMPILER"] = compiler
    for p6, mode in ((P.rov6_params(m=12.0, Xuu=-19.0), P.CTRL_FAITHFUL), (P.rov6_params(CG=[0.01, -0.015, 0.04], Yr=-0.3, m=12.0), P.CTRL_ZOH)):
        rep, log = P.JitReport(), C.create_string_buffer(16)   # a log buffer shorter than the log: truncated, not overrun
        assert lib.mvrl_jit_compile_check2(C.addressof(p6), mode, C.byref(rep), log, len(log)) == 0
        r = rep.as_dict()
        assert r["compiler"] == compiler and r["vgprs"] > 0 and r["lds_bytes"] == 10240, r
os.environ["MVRL_JIT_COMPILER"] = "hipcc"
os.environ["MVRL_HIPCC"] = "/nonexistent/hipcc"
log = C.create_string_buffer(4096)
p6 = P.rov6_params(m=12.0)      # (kept alive across the call: C.addressof of a temporary is a dangling pointer - ASan says so)
assert lib.mvrl_jit_compile_check2(C.addressof(p6), P.CTRL_FAITHFUL, None, log, len(log)) != 0
print("HOST-ASAN-CHECKS-OK")
