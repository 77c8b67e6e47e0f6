"""bench.py's one-line JSON contract, exercised end to end on the GPU with a small batch: every field the driver and the
judge read is present and self-consistent (value = envs x steps / time of the median repeat, roofline.frac = achieved /
peak, cpu_baseline from the oracle, the launch plans that were timed)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*extra):
    env = dict(os.environ, MVRL_CPU_THREADS="8")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--steps", "16", "--warmup", "8", "--repeats", "3",
                        "--prewarm-s", "0.05", "--envs-per-gpu", "262144"] + list(extra), capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]          # ONE JSON line
    return json.loads(lines[0])


def test_default_line_has_every_contract_field():
    j = _run()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "timing"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 16 and j["warmup"] == 8 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "f32" and j["data"] == "synthetic"
    assert "workload" in j["config"] and "model" not in j["config"]
    assert abs(j["value"] - 262144 * 16 / (j["ms_per_step"] * 1e-3 * 16)) / j["value"] < 1e-9
    rf = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["algorithmic_bytes_per_env_step"] == 365
    # achieved = algorithmic bytes x envs / GPU-side time per step
    assert abs(rf["achieved"] - 365 * 262144 / (rf["kernel_us_per_step"] * 1e-6) / 1e9) / rf["achieved"] < 1e-9
    assert rf["traffic"] is None        # the committed counters describe 1 048 576 envs per GPU: dropped for another batch size
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 8 and cb["value"] > 1e4 and "sample" in cb and cb["unit"] == "env-steps/s"
    t = j["timing"]
    assert t["repeats"] == 3 and len(t["ms_per_step_repeats"]) == 3 and t["prewarm_s"] >= 0.05
    assert set(t["launch_plans"]) == {"chains", "single"}               # --launch auto timed both
    # the headline plan is fixed by rule (chains for a batch of >= 4 waves per SIMD), not the minimum of the two medians
    assert j["launch_plan"] == "chains"
    assert abs(t["launch_plans"]["chains"]["ms_per_step"] - j["ms_per_step"]) < 1e-12
    assert j["outputs_finite"] is True


def test_other_workload_and_plan():
    j = _run("--workload", "auv", "--launch", "single", "--no-cpu-baseline")
    assert "cpu_baseline" not in j and j["roofline"]["algorithmic_bytes_per_env_step"] == 389
    assert "one launch per step" in j["config"]["launch"] and list(j["timing"]["launch_plans"]) == ["single"]
    assert j["config"]["kernel"].startswith("auv/")


@pytest.mark.parametrize("world", [2, 4])
def test_plain_command_launches_its_own_ranks(world):
    """`python bench.py --gpus N` (N = 2 and 4: four ranks are within the box's limit of six GPU processes) - no torchrun, no wrapper script - starts its ranks itself (bench.self_launch, before
    anything touches the GPU), prints ONE JSON line and returns 0.  On the one-GPU box the two ranks share the card over gloo
    (the rehearsal knobs; RCCL refuses duplicate devices) - everything else is the path the driver's N = 2/4/8 runs take:
    rendezvous, sharded step, gather pipeline, max-over-ranks timing."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(MVRL_BENCH_BACKEND="gloo", MVRL_BENCH_SAME_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(world), "--steps", "16", "--warmup", "8", "--repeats", "3",
                        "--prewarm-s", "0.05", "--envs-per-gpu", "131072"], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-500:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == world and j["config"]["global_envs"] == world * 131072
    assert j["rccl"]["world_size_seen"] == world and not j["rccl"]["ranks_on_distinct_devices"]      # the rehearsal shares one card
    assert j["with_gather"]["value"] is not None and j["with_gather"]["value"] > 0
    assert "cpu_baseline" not in j          # rank 0 at N = 1 only
    assert abs(j["value"] - world * 131072 * 16 / (j["ms_per_step"] * 1e-3 * 16)) / j["value"] < 1e-9
    assert j["with_gather"]["bytes_per_step_at_root"] == (131072 * 37 + 15) // 16 * 16 * world      # obs | done per rank, padded

