"""bench.py's launch logic, checked without a GPU: `--gpus N` never falls back to fewer GPUs."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH, *args], capture_output=True, text=True, timeout=300, env=env)


def test_gpus_above_visible_devices_is_refused():
    import torch
    n = torch.cuda.device_count()
    p = run(["--gpus", str(n + 2), "--steps", "1", "--warmup", "0"])
    assert p.returncode == 2 and "refusing to run" in p.stderr and p.stdout.strip() == ""


def test_gpus_must_match_world_size_under_an_external_launcher():
    p = run(["--gpus", "1", "--steps", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode == 2 and "does not match WORLD_SIZE=2" in p.stderr and p.stdout.strip() == ""


def test_parent_of_a_multi_rank_run_makes_no_gpu_call():
    """The spawn path is taken before HipPath / torch.cuda are touched: the parent function only
    counts devices (static check of the code order)."""
    src = open(BENCH).read()
    main = src[src.index("def main():"):]
    assert main.index("spawn_ranks(args)") < main.index("HipPath(")
    spawn = src[src.index("def spawn_ranks"):src.index("def cpu_baseline")]
    assert "device_count()" in spawn and "HipPath" not in spawn and "set_device" not in spawn
    assert "torch.distributed.run" in spawn and "subprocess.call" in spawn
