"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol the header
declares, and refuses to compute without a GPU (no silent CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from .conftest import REPO
from marinevehiclereinforcementlearning_amd import _lib, build, params as P


@pytest.fixture(scope="module")
def lib():
    build.build_lib()
    return _lib.load()


def header_symbols():
    txt = open(os.path.join(REPO, "include", "mvrl.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mvrl_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(lib):
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"libmvrl.so does not export {s}"
    assert sorted(_lib.SYMBOLS) == syms  # the Python binding covers exactly the header


def test_abi_version_and_dims(lib):
    assert lib.mvrl_abi_version() == P.ABI_VERSION
    for model, (act, obs, init, words, aux) in P.MODEL_DIMS.items():
        a, o, i, w = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        assert lib.mvrl_model_dims(model, C.byref(a), C.byref(o), C.byref(i), C.byref(w)) == 0
        assert (a.value, o.value, i.value, w.value) == (act, obs, init, words)
        assert lib.mvrl_aux_dim(model) == aux
    assert lib.mvrl_model_dims(7, None, None, None, None) == -1


def test_struct_layouts_match_header(lib):
    """sizeof() of the ctypes mirrors == the C structs (checked by compiling a probe with gcc)."""
    import subprocess
    import tempfile
    src = '#include <stdio.h>\n#include "mvrl.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(mvrl_rov6_params), ' \
          'sizeof(mvrl_rov3_params), sizeof(mvrl_auv_params), sizeof(mvrl_flow_desc), sizeof(mvrl_config));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        out = subprocess.check_output([os.path.join(d, "p")]).split()
    sizes = [int(x) for x in out]
    assert sizes == [C.sizeof(P.Rov6Params), C.sizeof(P.Rov3Params), C.sizeof(P.AuvParams), C.sizeof(P.FlowDesc),
                     C.sizeof(P.Config)]


def test_bad_config_is_rejected(lib):
    cfg = P.make_config("rov6", 16)
    cfg.abi_version = 99
    h = C.c_void_p()
    assert lib.mvrl_create(C.byref(cfg), C.byref(h)) == -1
    assert b"ABI" in lib.mvrl_last_error(None)
    cfg = P.make_config("rov6", 0)
    assert lib.mvrl_create(C.byref(cfg), C.byref(h)) == -1
    cfg = P.make_config("rov6", 40_000_000)  # 41 words * n * 4 B >= 2^32
    assert lib.mvrl_create(C.byref(cfg), C.byref(h)) == -1
    cfg = P.make_config("rov3", 8, n_substeps=0)
    assert lib.mvrl_create(C.byref(cfg), C.byref(h)) == -1


def test_no_cpu_fallback_without_gpu(lib):
    """On a box without a HIP device the product path must fail loudly (MVRL_ENODEV), never compute on the CPU."""
    if lib.mvrl_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.MvrlError, match="ENODEV"):
        _lib.Handle(P.make_config("rov6", 8))
    with pytest.raises(_lib.MvrlError, match="ENODEV"):
        _lib.flow_interp(np.zeros((2, 2, 2, 3), np.float32), 1, 1, 1, [0.], [0.], [0.])


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under the package may import/reference it."""
    pkg = os.path.join(REPO, "marinevehiclereinforcementlearning_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".inc")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.lower() or f == "_never_.py", f"{f} mentions the oracle"


def test_baked_table_matches_params():
    """csrc/mvrl_baked.inc is regenerated from params.py; SYM structure detection agrees with the defaults."""
    from marinevehiclereinforcementlearning_amd import devparams as D
    assert D.rov6_structured(P.rov6_params())
    assert not D.rov6_structured(P.rov6_params(CG=[0.01, 0., 0.05]))
    assert not D.rov6_structured(P.rov6_params(l_x=0.15, l_y=0.101, Yr=0.3))
    sa, sb = D.sym_layout(P.rov6_params().alloc, P.rov6_params().alloc_inv)
    assert abs(sa[0] - 0.838671) < 1e-6 and abs(sb[7] - 2.083333) < 1e-6


def test_jit_specialisation_compiles_without_a_gpu():
    """mvrl_specialize's compilation step (hipcc or hiprtc, gfx950) for the default, a BlueROV2-structured and an arbitrary set of
    constants, both controller placements: the library carries its own kernel sources and they build at run time."""
    import ctypes as C
    from marinevehiclereinforcementlearning_amd import _lib, params as P
    lib = _lib.load()
    cases = [(P.rov6_params(), P.CTRL_FAITHFUL, "Lb1ELb0E"), (P.rov6_params(m=12.0, Xuu=-19.0), P.CTRL_ZOH, "Lb1ELb1E"),
             (P.rov6_params(CG=[0.01, -0.015, 0.04], Yr=-0.3, m=12.0), P.CTRL_FAITHFUL, "Lb0ELb0E")]
    for p6, mode, tag in cases:
        size, log = C.c_size_t(0), C.create_string_buffer(8192)
        rc = lib.mvrl_jit_compile_check(C.addressof(p6), mode, C.addressof(size), log, len(log))
        assert rc == 0, log.value.decode()
        names = log.value.decode().split("\n")
        assert size.value > 10000 and len(names) == 3 and all("rov6_step_kernel" in nm and tag in nm for nm in names[:2]), names
        assert names[2] in ("compiled by hipcc", "compiled by hiprtc")


def test_jit_specialisation_with_either_compiler(monkeypatch):
    """The two compilers mvrl_specialize can use: the ROCm installation's hipcc as a child process (preferred: a process that
    imports PyTorch has PyTorch's older libhiprtc / comgr loaded, whose code for this kernel spills) and in-process hiprtc."""
    import ctypes as C
    from marinevehiclereinforcementlearning_amd import _lib, params as P
    lib = _lib.load()
    p6 = P.rov6_params(m=12.0, Xuu=-19.0)
    for compiler in ("hipcc", "hiprtc"):
        monkeypatch.setenv("MVRL_JIT_COMPILER", compiler)
        size, log = C.c_size_t(0), C.create_string_buffer(8192)
        rc = lib.mvrl_jit_compile_check(C.addressof(p6), P.CTRL_FAITHFUL, C.addressof(size), log, len(log))
        assert rc == 0 and size.value > 10000 and log.value.decode().endswith("compiled by " + compiler), log.value.decode()
    monkeypatch.setenv("MVRL_JIT_COMPILER", "hipcc")
    monkeypatch.setenv("MVRL_HIPCC", "/nonexistent/hipcc")
    size, log = C.c_size_t(0), C.create_string_buffer(8192)
    assert lib.mvrl_jit_compile_check(C.addressof(p6), P.CTRL_FAITHFUL, C.addressof(size), log, len(log)) != 0
    assert "cannot start" in log.value.decode() or "failed" in log.value.decode()


def test_jit_child_environment_is_scrubbed(monkeypatch, tmp_path):
    """The hipcc child of mvrl_specialize must not inherit a profiler: under rocprofv3 the host process carries LD_PRELOAD /
    ROCP_TOOL_LIBRARIES / HSA_TOOLS_LIB, the tool library would initialise the GPU inside hipcc, and hipcc execs clang and lld -
    exec after GPU initialisation.  mvrl_jit_child_env shows the environment the child gets; a compilation still succeeds with a
    (dummy) profiler environment set, because the child never sees it."""
    import ctypes as C
    from marinevehiclereinforcementlearning_amd import _lib, params as P
    lib = _lib.load()
    for k, v in (("LD_PRELOAD", "/nonexistent/librocprofiler-sdk-tool.so"), ("ROCP_TOOL_LIBRARIES", "/nonexistent/tool.so"),
                 ("HSA_TOOLS_LIB", "/nonexistent/libtool.so"), ("ROCPROFILER_LIBRARY_CTOR", "1"), ("LD_AUDIT", "/nonexistent/audit.so"),
                 ("MVRL_KEEP_ME", "yes")):
        # straight into the C environment (os.putenv): the test process itself must not start children with LD_PRELOAD set
        C.CDLL(None).setenv(k.encode(), v.encode(), 1)
    try:
        buf = C.create_string_buffer(1 << 18)
        assert lib.mvrl_jit_child_env(buf, len(buf)) == 0
        keys = [ln.split("=", 1)[0] for ln in buf.value.decode().splitlines()]
        assert "MVRL_KEEP_ME" in keys and "PATH" in keys
        for k in keys:
            assert not k.startswith(("LD_PRELOAD", "LD_AUDIT", "ROCP_", "ROCPROFILER_", "HSA_TOOLS_")), k
        C.CDLL(None).setenv(b"MVRL_JIT_COMPILER", b"hipcc", 1)
        C.CDLL(None).setenv(b"TMPDIR", str(tmp_path).encode(), 1)      # the scratch directory honours $TMPDIR ...
        p6 = P.rov6_params(m=12.0, Xuu=-19.0)
        rep, log = P.JitReport(), C.create_string_buffer(8192)
        rc = lib.mvrl_jit_compile_check2(C.addressof(p6), P.CTRL_FAITHFUL, C.byref(rep), log, len(log))
        assert rc == 0, log.value.decode()
        assert os.listdir(tmp_path) == []                                # ... and is removed afterwards
    finally:
        for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB", "ROCPROFILER_LIBRARY_CTOR", "LD_AUDIT", "MVRL_KEEP_ME",
                  "MVRL_JIT_COMPILER", "TMPDIR"):
            C.CDLL(None).unsetenv(k.encode())


def test_jit_report_reads_the_code_object_notes(monkeypatch):
    """mvrl_jit_compile_check2 / mvrl_jit_info: compiler, registers, spills and scratch of what was built, parsed from the
    AMDGPU metadata note.  The installation's hipcc fits the structured kernel into the 128-VGPR budget without scratch or VGPR spills."""
    import ctypes as C
    from marinevehiclereinforcementlearning_amd import _lib, params as P
    lib = _lib.load()
    p6 = P.rov6_params(m=12.0, Xuu=-19.0)
    got = {}
    for compiler in ("hipcc", "hiprtc"):
        monkeypatch.setenv("MVRL_JIT_COMPILER", compiler)
        rep, log = P.JitReport(), C.create_string_buffer(8192)
        assert lib.mvrl_jit_compile_check2(C.addressof(p6), P.CTRL_FAITHFUL, C.byref(rep), log, len(log)) == 0, log.value.decode()
        r = rep.as_dict()
        got[compiler] = r
        assert r["compiler"] == compiler and r["specialized"] == 1 and r["min_waves_per_simd"] == 4
        assert 64 <= r["vgprs"] <= 128 and 16 <= r["sgprs"] <= 112 and r["lds_bytes"] == 10240 and r["code_bytes"] > 10000
        assert r["vgpr_spills"] >= 0 and r["sgpr_spills"] >= 0 and r["scratch_bytes"] >= 0
    # (a couple of SGPRs parked in VGPR lanes are harmless - the installation's compiler shows 2 for this kernel; scratch is not)
    assert got["hipcc"]["scratch_bytes"] == 0 and got["hipcc"]["vgpr_spills"] == 0 and got["hipcc"]["sgpr_spills"] <= 16, got


def test_library_does_not_link_hiprtc_at_load_time(lib):
    """hiprtc is the fallback compiler of an optional feature: dlopen-ed on first use, not a DT_NEEDED of libmvrl.so."""
    import subprocess
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    needed = [ln for ln in out.splitlines() if "NEEDED" in ln]
    assert needed and not any("hiprtc" in ln for ln in needed), needed


def test_default_config_equals_make_config(lib):
    """mvrl_default_config (what a C caller starts from) fills the very bytes params.make_config does, for all three models:
    the reference's default constants incl. M^-1 and pinv(A), generated from params.py at build time (csrc/gen/mvrl_defaults.inc)."""
    for name, model in P.MODEL_NAMES.items():
        c = P.Config()
        assert lib.mvrl_default_config(model, 4096, C.byref(c)) == 0
        assert bytes(c) == bytes(P.make_config(name, 4096)), name
    assert lib.mvrl_default_config(7, 1, C.byref(P.Config())) != 0 and lib.mvrl_default_config(2, 1, None) != 0


def _build_c_example(tmp_path, name="step_rov6"):
    import subprocess
    exe = str(tmp_path / name)
    pkg = os.path.join(REPO, "marinevehiclereinforcementlearning_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "examples", name + ".c"),
                           "-L", pkg, "-lmvrl", "-lm", f"-Wl,-rpath,{pkg}", "-o", exe])
    return exe


def test_c_example_builds_against_the_header_and_fails_loudly_without_a_gpu(lib, tmp_path):
    """examples/step_rov6.c: plain C against include/mvrl.h + libmvrl.so (no Python, no torch).  Without a HIP device it must say so and
    exit 2 - no CPU fallback."""
    import subprocess
    exe = _build_c_example(tmp_path)
    if lib.mvrl_device_count() > 0:
        pytest.skip("a GPU is present: tests/test_gpu_api.py runs the example for real")
    r = subprocess.run([exe, "64", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no HIP device" in r.stderr and "step" not in r.stdout
    # examples/group_c5.c: BASELINE configs[4] from one C process (mvrl_group_*): same contract
    exe = _build_c_example(tmp_path, "group_c5")
    r = subprocess.run([exe, "64", "2"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "no HIP device" in r.stderr and "env-steps/s" not in r.stdout


def test_handle_outputs_are_copies_by_default():
    """Lifetime contract of the Python wrapper: Handle.reset / step / step_wait hand out FRESH arrays unless the caller asks
    for the zero-copy views of the handle's pinned staging block (copy=False), which die with the handle.  Checked on the signatures
    here (no GPU); tests/test_gpu_api.py::test_outputs_outlive_the_handle checks the behaviour."""
    import inspect
    from marinevehiclereinforcementlearning_amd import _lib
    for name in ("reset", "step", "step_wait"):
        sig = inspect.signature(getattr(_lib.Handle, name))
        assert sig.parameters["copy"].default is True, name


def test_group_partition_and_message_layout_match_the_python_side(lib):
    """mvrl_group_* cuts the batch and lays the gather message out exactly as distributed.shard_range / OutputGather do (both sides
    of BASELINE configs[4] - one process per GPU over torch.distributed, or one process for all GPUs over the C ABI - exchange the
    same bytes); without a GPU a group cannot be created and says so (no CPU fallback)."""
    from marinevehiclereinforcementlearning_amd import distributed as D, group as G
    for n, w in [(67, 8), (8 * 1048576, 8), (5, 5), (100, 3), (1, 1), (1000003, 7)]:
        assert [G.shard_range(n, i, w) for i in range(w)] == [D.shard_range(n, i, w) for i in range(w)]
        for obs_dim, rp in [(9, False), (11, True), (5, False)]:
            assert G.message_layout(n, w, obs_dim, rp) == D.message_layout(n, w, obs_dim, rp)
    # configs[4]: 8 x 1 048 576 6-DoF envs -> 37 B per env on the wire (obs 36 + done 1; the reward plane is identically 0)
    lay = G.message_layout(8 * 1048576, 8, 9, False)
    assert lay["msg_bytes"] == 1048576 * 37
    with pytest.raises(_lib.MvrlError):
        G.shard_range(10, 3, 3)
    if lib.mvrl_device_count() == 0:
        with pytest.raises(_lib.MvrlError, match="ENODEV"):
            G.DeviceGroup(P.make_config("rov6", 128, use_flow=False), [0])
    # bad arguments are refused before any device is touched
    g = C.c_void_p()
    cfg = P.make_config("rov6", 128, use_flow=False)
    G._declare(lib)
    assert lib.mvrl_group_create(C.byref(cfg), None, 1, 0, C.byref(g)) == -1
    dev = (C.c_int32 * 2)(0, 0)
    assert lib.mvrl_group_create(C.byref(cfg), dev, 2, 5, C.byref(g)) == -1
    cfg64 = P.make_config("rov6", 128, use_flow=False, precision="f64")
    assert lib.mvrl_group_create(C.byref(cfg64), dev, 2, 0, C.byref(g)) == -1
