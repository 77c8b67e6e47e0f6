"""Sanitizer runs of the CPU-side code (SURVEY.md section 5 "race detection / sanitizers"): AddressSanitizer + UndefinedBehaviorSanitizer
builds of (a) the C oracle, replaying golden RK4 / RK45 / fp32 / AuvEnv trajectories, and (b) the HOST half of libmvrl.so (argument
checking, parameter narrowing, JIT driver, code-object note parser), driven through the bad-configuration cases of tests/test_abi.py
and the host-only entry points.  CPU builds only - nothing here touches a GPU (GPU sanitizers are not available on the pool)."""
import os
import subprocess
import sys

import pytest

from ..conftest import REPO


def _run(driver, preload, extra_env):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", **extra_env)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "sanitize", driver)], capture_output=True, text=True, env=env, timeout=900)
    findings = [ln for ln in r.stderr.splitlines() if "AddressSanitizer" in ln or "runtime error:" in ln]
    assert r.returncode == 0 and not findings, (r.returncode, findings[:5], r.stderr[-1500:], r.stdout[-500:])
    return r.stdout


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "-s", "-f", os.path.join(REPO, "tests", "sanitize", "oracle_asan.mk")])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(libasan):
        pytest.skip("gcc has no libasan")
    out = _run("oracle_replay.py", libasan, {"MVRL_ORACLE_LIB": os.path.join(REPO, "oracle", "_build", "libmvrl_oracle_asan.so")})
    assert "SANITIZER-REPLAY-OK" in out and "rk45 f64" in out and "rk4 f32" in out


def test_libmvrl_host_side_under_asan_ubsan():
    from marinevehiclereinforcementlearning_amd import _lib
    from . import build_san as build
    if _lib.device_count() > 0:
        pytest.skip("CPU-container check: the sanitizer build is never loaded next to a GPU")
    rt = build.asan_runtime()
    if rt is None:
        pytest.skip("no clang ASan runtime in this ROCm installation")
    lib = build.build_host_asan()
    out = _run("abi_host_checks.py", rt, {"MVRL_LIB": lib})
    assert "HOST-ASAN-CHECKS-OK" in out
