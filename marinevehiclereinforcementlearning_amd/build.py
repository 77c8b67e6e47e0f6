"""Build libmvrl.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m marinevehiclereinforcementlearning_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so lands next to this file so that it travels with the tree.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmvrl.so")
SOURCES = ["mvrl_abi.hip", "mvrl_rov6.hip", "mvrl_rov3.hip", "mvrl_auv.hip", "mvrl_flow.hip", "mvrl_policy.hip", "mvrl_replay.hip",
           # fp64 twins, generated from the fp32 sources by tools/gen_f64.py at build time
           "gen/mvrl_rov6_f64.hip", "gen/mvrl_rov3_f64.hip", "gen/mvrl_auv_f64.hip"]
HEADERS = ["mvrl_device.hpp", "mvrl_kernels.hpp", "mvrl_baked.inc", "mvrl_rk45.hpp", "gen/mvrl_device_f64.hpp",
           "gen/mvrl_kernels_f64.hpp", "gen/mvrl_baked_f64.inc"]
ARCH = "gfx950"
# -fno-slp-vectorize: v_pk_* packing costs more constant moves than it saves here (measured -16 %);
# -ffast-math: reassociation + no-NaN/Inf folding (structural zeros of the baked constants disappear), +10 %; parity
# tests pass with the same margins with and without it (DESIGN.md "compiler flags").
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-ffast-math", f"--offload-arch={ARCH}", "-I", CSRC, "-I",
         os.path.join(CSRC, "gen"), "-Wall",
         "-Wno-unused-function"]


def _gen_baked():
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import gen_baked
    import gen_f64
    gen_baked.main()
    return gen_f64.main()


def source_hash():
    """Identity of the kernels a profile was taken from: sha256 over the hand-written kernel sources, the generators of the
    derived ones and the compiler flags.  Counter summaries under profiles/ carry it; bench.py drops them when it differs."""
    import hashlib
    hsh = hashlib.sha256()
    names = [s_ for s_ in SOURCES if not s_.startswith("gen/")] + [h_ for h_ in HEADERS if not h_.startswith("gen/") and h_ != "mvrl_baked.inc"]
    for name in sorted(names):
        with open(os.path.join(CSRC, name), "rb") as f:
            hsh.update(name.encode() + b"\0" + f.read())
    for tool in ("gen_baked.py", "gen_f64.py"):
        with open(os.path.join(REPO, "tools", tool), "rb") as f:
            hsh.update(f.read())
    with open(os.path.join(HERE, "params.py"), "rb") as f:
        hsh.update(f.read())
    hsh.update(" ".join(fl for fl in FLAGS if not fl.startswith("/") and fl != "-I").encode())
    return hsh.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force=False, verbose=False, extra_flags=None, out=None, drop_flags=()):
    """extra_flags / out / drop_flags: build a tuning variant next to the default library (tools/variants.py)."""
    _gen_baked()
    if extra_flags or out or drop_flags:
        return _build_variant(extra_flags or [], out or LIB, drop_flags)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        hipcc = "hipcc"
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(REPO, "include", "mvrl.h")]
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            flags = list(FLAGS)
            if src.startswith("gen/"):
                # the fp64 twins are the exactness path: IEEE arithmetic, no contraction beyond the explicit fma() calls
                flags = [f for f in flags if f != "-ffast-math"] + ["-ffp-contract=off"]
            jobs.append([hipcc] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(objdir, os.path.basename(s).replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def _build_variant(extra_flags, out, drop_flags):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = [f for f in FLAGS if f not in drop_flags] + list(extra_flags)
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    r = subprocess.run([hipcc] + flags + ["-shared", "-o", out] + srcs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr)
    return out


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
