"""`python bench.py --gpus N` started plainly launches its own N ranks (bench.self_launch) - the parts of that which need no
GPU: the launcher runs before torch is imported, hands every rank the torchrun environment, relays rank 0's stdout only and
returns the worst exit code.  The complete two-rank run is tests/test_gpu_bench_contract.py::test_plain_command_launches_its_own_ranks."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}


def test_bad_arguments_fail_in_every_rank_and_in_the_launcher():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--workload", "nosuch"], capture_output=True,
                       text=True, env=_env(), timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])          # argparse's exit code, before any rank exists
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_ranks_get_the_torchrun_environment_and_the_worst_exit_code_wins(tmp_path):
    """self_launch with a stand-in rank program: every rank sees RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT, only
    rank 0's stdout reaches the launcher's stdout, and one failing rank (exit 3, the gather watchdog's code) decides the result."""
    prog = tmp_path / "launch_probe.py"
    prog.write_text(
        "import os, sys, importlib.util\n"
        f"spec = importlib.util.spec_from_file_location('bench', {os.path.join(REPO, 'bench.py')!r})\n"
        "bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)\n"
        "if 'WORLD_SIZE' not in os.environ:\n"
        "    bench.__file__ = os.path.abspath(__file__)          # the ranks re-run THIS file\n"
        "    sys.exit(bench.self_launch(3, argv=[]))\n"
        "assert 'torch' not in sys.modules\n"
        "r = int(os.environ['RANK'])\n"
        "print('rank', r, os.environ['LOCAL_RANK'], os.environ['WORLD_SIZE'], os.environ['MASTER_ADDR'], int(os.environ['MASTER_PORT']) > 0, flush=True)\n"
        "sys.exit(3 if r == 1 else 0)\n")
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, env=_env(), timeout=300)
    assert r.returncode == 3, (r.returncode, r.stderr[-800:])
    assert r.stdout.strip() == "rank 0 0 3 127.0.0.1 True", r.stdout
    assert "rank 1 1 3 127.0.0.1 True" in r.stderr and "rank 2 2 3 127.0.0.1 True" in r.stderr


def test_eight_ranks_rendezvous_and_the_line_reports_all_of_them(tmp_path):
    """bench.py's own launcher at the world size of BASELINE configs[4]: self_launch(8) starts eight ranks; each joins the process group the
    way bench.py does (distributed.init_from_env - gloo here, there is no GPU) and takes part in rccl_info, the collective whose result
    rides in the bench line as `rccl`; rank 0's line must report world_size_seen == 8 and eight device entries."""
    import json
    prog = tmp_path / "world8_probe.py"
    prog.write_text(
        "import os, sys, json, importlib.util\n"
        f"sys.path.insert(0, {REPO!r})\n"
        f"spec = importlib.util.spec_from_file_location('bench', {os.path.join(REPO, 'bench.py')!r})\n"
        "bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)\n"
        "if 'WORLD_SIZE' not in os.environ:\n"
        "    bench.__file__ = os.path.abspath(__file__)\n"
        "    sys.exit(bench.self_launch(8, argv=[]))\n"
        "import torch.distributed as dist\n"
        "from marinevehiclereinforcementlearning_amd import distributed as D\n"
        "rank, world, local_rank = D.init_from_env('gloo')\n"
        "off, cnt = D.shard_range(8388608, rank, world)\n"
        "info = D.rccl_info('gloo', local_rank)\n"
        "dist.barrier()\n"
        "if rank == 0:\n"
        "    print(json.dumps({'n_gpus': world, 'rccl': info, 'shard0': [off, cnt]}), flush=True)\n"
        "dist.destroy_process_group()\n")
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 0, (r.returncode, r.stderr[-1500:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # ONE JSON line, rank 0's
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["rccl"]["world_size_seen"] == 8 and len(j["rccl"]["devices"]) == 8 and j["rccl"]["backend"] == "gloo"
    assert j["shard0"] == [0, 1048576]                                 # configs[4]: 8 388 608 envs = 8 x 1 048 576


def test_a_dead_rank_takes_the_hung_ones_with_it(tmp_path):
    """A rank that dies while the others wait for it (a broken rendezvous) must not hang the job: the launcher kills the survivors
    after its grace period and reports the failure."""
    import time
    prog = tmp_path / "hang_probe.py"
    prog.write_text(
        "import os, sys, time, importlib.util\n"
        f"spec = importlib.util.spec_from_file_location('bench', {os.path.join(REPO, 'bench.py')!r})\n"
        "bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)\n"
        "if 'WORLD_SIZE' not in os.environ:\n"
        "    bench.__file__ = os.path.abspath(__file__)\n"
        "    sys.exit(bench.self_launch(2, argv=[]))\n"
        "if os.environ['RANK'] == '0':\n"
        "    sys.exit(7)\n"
        "time.sleep(600)\n")
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, str(prog)], capture_output=True, text=True, env=dict(_env(), MVRL_LAUNCH_GRACE_S="1.5"), timeout=120)
    assert r.returncode != 0 and time.monotonic() - t0 < 60, (r.returncode, time.monotonic() - t0)
    assert r.returncode == 128 + 9        # the worst code: the SIGKILL the launcher handed to the hung rank (rank 0's own was 7)


def test_no_gpu_no_result():
    """Without a GPU the plain N = 2 command fails loudly in both ranks (no CPU fallback) and prints no JSON line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-container check")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       capture_output=True, text=True, env=_env(), timeout=600)
    assert r.returncode == 1, (r.returncode, r.stderr[-500:])
    assert r.stderr.count("needs a GPU") == 2
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
