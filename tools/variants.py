#!/usr/bin/env python3
"""Build tuning variants of libmvrl.so (different launch bounds / compiler flags) into gpurun_out/variants/ and,
with `run`, time the C4/C3 workloads with each of them (MVRL_LIB selects the library)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
OUT = os.path.join(REPO, "variants_build")  # *.so is git-ignored; gpurun_out/ does not travel to the GPU box
VARIANTS = {
    "base": dict(extra=[], drop=()),
    # the fp64 twins (the exactness path) under other floating-point models: IEEE without contraction (rounds 1-4), full fast-math
    "f64strict": dict(extra=[], drop=(), f64=["-ffp-contract=off"]),
    "f64fast": dict(extra=[], drop=(), f64=["-ffast-math"]),
    "f64nopark": dict(extra=["-DMVRL_NO_PARK"], drop=()),
    "f64ilp": dict(extra=[], drop=(), f64=["-ffp-contract=fast", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
    "f64occ": dict(extra=[], drop=(), f64=["-ffp-contract=fast", "-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]),
    "f64fulltrig": dict(extra=["-DMVRL_FULL_STAGE_TRIG"], drop=()),
    # round 5, second session: the PID's increment select as it was (six subtract-and-select pairs per first-stage call)
    "notrigvote": dict(extra=["-DMVRL_NO_TRIG_VOTE"], drop=()),   # rare per-lane fall-backs as plain exec guards (before this session: C2 +5 %)
    # the kernels as they were at the end of the round's first sitting (select forms, exec guards, library expf, fp64 twins with NaN / Inf /
    # signed-zero bookkeeping): the reference arm when a new seed of a sweep fails - is it this sitting's doing?
    "r5a": dict(extra=["-DMVRL_INC_SELECT", "-DMVRL_NO_TRIG_VOTE", "-DMVRL_LIB_EXP"], drop=(), f64=["-ffp-contract=fast"]),
    "noyawwrap": dict(extra=["-DMVRL_NO_YAW_FULL_WRAP"], drop=()),   # the carried yaw error with ONE turn of correction only (rounds 3-5: wrong through gimbal lock)
    "libexp": dict(extra=["-DMVRL_LIB_EXP"], drop=()),   # the 3-DoF jet-drag factors through the library expf (before this session: C2 +18 %)
    "incsel": dict(extra=["-DMVRL_INC_SELECT"], drop=()),
    # fp64 twins: no NaN / Inf / signed-zero bookkeeping, but IEEE division and the written order of operations
    "f64nofin": dict(extra=[], drop=(), f64=["-ffp-contract=fast"]),     # the fp64 twins WITH NaN / Inf / signed-zero bookkeeping (first session of round 5)
    "f64nolicm": dict(extra=[], drop=(), f64=["-ffp-contract=fast", "-mllvm", "-disable-machine-licm"]),
    # attribution (tests/audit/episode_audit.py): the round-4 fp32 turbulence sample time
    "flowt32": dict(extra=["-DMVRL_FLOW_TIME_F32"], drop=()),
    # profiling build: per-wave s_memtime stamps at the phase boundaries of the 6-DoF step kernel (tools/stamp_probe.py)
    "stamp": dict(extra=["-DMVRL_STAMP"], drop=()),
    # ablations of choices the default build makes
    "novs": dict(extra=["-DMVRL_NO_VGPR_SCALARS"], drop=()),          # RK step sizes left in SGPRs
    "auvscatter": dict(extra=["-DMVRL_AUV_LDS_OBS=0"], drop=()),      # AuvEnv observations stored row-per-lane
    # ("blk256", -DMVRL_STEP_BLOCK=256, measured +2 % in round 2, no longer builds: the LDS parking tiles are sized for one-wave blocks)
    # two / four waves per workgroup: fewer workgroup launches per grid (the dependent-launch gap grows with the grid: 3.3 us at 1 024 one-wave
    # workgroups, 5.8 us at 16 384); the parking tiles scale with the block
    "blk128": dict(extra=["-DMVRL_STEP_BLOCK=128", "-DMVRL_PARK_FLOAT4S=(1280*(4/MVRL_PARK_PER))"], drop=()),
    "blk256": dict(extra=["-DMVRL_STEP_BLOCK=256", "-DMVRL_PARK_FLOAT4S=(2560*(4/MVRL_PARK_PER))"], drop=()),
    "blk32": dict(extra=["-DMVRL_STEP_BLOCK=32"], drop=()),           # half-filled waves: twice the waves for a launch-bound batch (C2)
    "slp": dict(extra=[], drop=("-fno-slp-vectorize",)),
    "nofast": dict(extra=[], drop=("-ffast-math",)),
    "fulltrig": dict(extra=["-DMVRL_FULL_STAGE_TRIG"], drop=()),      # full sincos at every RK stage (round-1 behaviour)
    "native": dict(extra=["-DMVRL_NATIVE_TRIG"], drop=()),            # hardware v_sin/v_cos (1e-6 absolute accuracy)
    "nopark": dict(extra=["-DMVRL_NO_PARK"], drop=()),                # y / acc stay in registers: 156 VGPRs, three waves per SIMD
    "nopark_w4": dict(extra=["-DMVRL_NO_PARK", "-DMVRL_MIN_WAVES=4"], drop=()),   # 128-VGPR cap without parking: scratch spills
    "park6k": dict(extra=["-DMVRL_PARK_FLOAT4S=384"], drop=()),       # 6 KB of LDS per wave: lets a fifth wave in where VGPRs allow
    "auvskip1": dict(extra=["-DMVRL_AUV_SKIP=1"], drop=()),  # AuvEnv write-traffic attribution: no action-ring stores
    "auvskip2": dict(extra=["-DMVRL_AUV_SKIP=2"], drop=()),  #   no observation stores
    "auvskip4": dict(extra=["-DMVRL_AUV_SKIP=4"], drop=()),  #   no reward / done stores
    "auvskip8": dict(extra=["-DMVRL_AUV_SKIP=8"], drop=()),  #   no pose / error-memory stores
    "trigwave": dict(extra=["-DMVRL_TRIG_WAVE_FALLBACK"], drop=()),   # a wave with one large angle increment evaluates the stage in full for all lanes
    "w2": dict(extra=["-DMVRL_MIN_WAVES=2"], drop=()),
    "w3": dict(extra=["-DMVRL_MIN_WAVES=3"], drop=()),
    "w4": dict(extra=["-DMVRL_MIN_WAVES=4"], drop=()),                # spills
    "rt3": dict(extra=["-DMVRL_RT_WAVES=3"], drop=()),                # run-time-constant structured flavours (ctrl, sym) at 3 waves per SIMD
    "tb4": dict(extra=["-DMVRL_TB_FROM_STAGE4"], drop=()),            # experiment (not adopted): next base attitude rotated from the fourth stage's
    "nos3": dict(extra=["-DMVRL_NO_STAGE3_SMALL"], drop=()),          # stage-3 attitude rotated from the base attitude like the other stages
    "novote": dict(extra=["-DMVRL_NO_WINDUP_VOTE"], drop=()),         # per-axis wind-up compare-and-select at every PID call (round-2 behaviour)
    "noyaw": dict(extra=["-DMVRL_NO_YAW_INC"], drop=()),              # fresh angle reduction of the yaw error at every PID call (round-2 behaviour)
    "nofb": dict(extra=["-DMVRL_TRIG_NO_FALLBACK"], drop=()),         # attribution only: no full sincos for lanes with large angle increments
    "sc1st": dict(extra=["-DMVRL_STORE_SC1=1"], drop=()),              # state planes stored write-through (sc1): nothing dirty in L2 at the kernel boundary
    "sc1all": dict(extra=["-DMVRL_STORE_SC1=3"], drop=()),             # ... and the observations
    "ilp": dict(extra=["-mllvm", "-amdgpu-sched-strategy=max-ilp"], drop=()),
    "bias0": dict(extra=["-mllvm", "-amdgpu-schedule-metric-bias=0"], drop=()),
    "ilpw3": dict(extra=["-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DMVRL_MIN_WAVES=3"], drop=()),
}


def build_all(names):
    from marinevehiclereinforcementlearning_amd import build
    os.makedirs(OUT, exist_ok=True)

    def one(name):
        v = VARIANTS[name]
        return build.build_lib(extra_flags=v["extra"], out=os.path.join(OUT, f"libmvrl_{name}.so"), drop_flags=v["drop"], f64_flags=v.get("f64"))
    with ThreadPoolExecutor(max_workers=2) as ex:   # each build compiles its sources four at a time
        print(list(ex.map(one, names)))


def run_all(names):
    for name in names:
        env = dict(os.environ, MVRL_LIB=os.path.join(OUT, f"libmvrl_{name}.so"))
        for wlk in (os.environ.get("MVRL_VARIANT_WORKLOADS", "c4,c3").split(",")):
            r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", wlk, "--steps", "3000", "--warmup",
                                "500", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
            import json
            try:
                j = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"{name:8s} {wlk}: {j['value']:.3e} env-steps/s  {j['roofline']['kernel_us_per_launch']:.1f} us/launch", flush=True)
            except Exception:  # noqa: BLE001
                print(name, wlk, "FAILED", r.stderr[-400:], flush=True)


if __name__ == "__main__":
    names = [a for a in sys.argv[2:]] or list(VARIANTS)
    (build_all if sys.argv[1] == "build" else run_all)(names)
