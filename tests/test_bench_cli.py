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


import json

import pytest


@pytest.mark.gpu
def test_bench_contract_line_on_the_gpu():
    """A short run of bench.py (reduced batch; the real shape per pair) prints exactly ONE JSON line on
    stdout with the contract's keys, a roofline object, a cpu_baseline object, and verified results."""
    p = run(["--batch", "512", "--slots", "32", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1.5",
             "--no-cpu-all-cores", "--secondary-seconds", "120"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[:500]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "secondary"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "mult/s" and d["dtype"] == "u64"
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["verified_vs_oracle"] is True and len(d["config"]["verified_slots"]) >= 2
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0.3 < r["frac"] < 1.0
    assert r["kernel"] == "k_touch+k_mul_flat"
    c = d["cpu_baseline"]
    assert c["cores"] == 1 and c["kind"] in ("reference", "port") and c["value"] > 0 and "sample" in c
    assert abs(d["value"] - 512 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6
    # the secondary suite: every case timed and checked against the oracle, none skipped or failed
    sec = {row["name"]: row for row in d["secondary"]}
    for name in ("config5_graph_compiled_n4096", "config5_graph_compiled_n1247", "config5_graph_tape_n4096",
                 "config5_graph_tape_n1247", "mul_1x1", "add_1024", "decrypt_1024", "permute_1m", "encrypt_keyed_1m",
                 "mul_ragged_mean8_kernel", "mul_ragged_mean8_async", "mul_ragged_mean16_kernel", "mul_ragged_mean16_async",
                 "compact_0pct", "compact_50pct", "compact_large", "add_ragged_singles", "add_ragged_singles_bounded", "decrypt_ragged_singles",
                 "decrypt_ragged_singles_bounded"):
        assert name in sec, name
        row = sec[name]
        assert "error" not in row and "skipped" not in row, row
        assert row["verified"] is True and row["ms"] > 0 and row["bytes"] > 0, row
        assert abs(row["frac"] - row["bytes"] / (row["ms"] / 1e3) / 8.0e12) < 1e-9 and row["frac"] < 1.0, row
    assert sec["add_ragged_singles_bounded"]["ms"] < sec["add_ragged_singles"]["ms"]
    # the compiled config-5 graph moves about a third of the tape's bytes and takes less than two thirds of its time
    assert sec["config5_graph_compiled_n4096"]["bytes"] * 2 < sec["config5_graph_tape_n4096"]["bytes"]
    assert sec["config5_graph_compiled_n4096"]["ms"] * 1.5 < sec["config5_graph_tape_n4096"]["ms"]


@pytest.mark.gpu
def test_bench_native_ranks_mode_prints_the_same_contract_line():
    """--native-ranks: the measurement runs in tools/bin/bench_native -- one process, one host
    thread per GPU over the C ABI, no torch, the RCCL libcsgn_shard.so was built against under the STRICT version
    check -- and prints bench.py's contract line.  With --force-collective the native all-gather of the term counts
    is inside the timed region at world 1; every arena slot is verified on the GPU by the tool itself."""
    p = run(["--native-ranks", "--force-collective", "--batch", "512", "--slots", "32", "--steps", "2", "--warmup", "1",
             "--cpu-seconds", "1.5"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[:500]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["unit"] == "mult/s" and d["dtype"] == "u64" and d["scaling"] == "weak"
    assert "CSGN_COMM_STRICT" in d["config"]["collective"] and "ncclAllGather" in d["config"]["collective"]
    assert "no torch" in d["config"]["collective"] and d["config"]["verified_slots"] == 32
    r = d["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0.3 < r["frac"] < 1.0
    assert r["kernel"] == "k_touch+k_mul_flat"
    assert abs(d["value"] - 512 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-3
    # the tool is what it says: no torch, no python in its link map
    import subprocess
    tool = os.path.join(ROOT, "tools", "bin", "bench_native")
    ldd = subprocess.run(["ldd", tool], capture_output=True, text=True).stdout
    assert "libcsgn_hip.so" in ldd and "libcsgn_shard.so" in ldd and "torch" not in ldd and "python" not in ldd


@pytest.mark.gpu
def test_bench_spawned_rank_with_the_rccl_gather():
    """--spawn: the parent starts one rank through torch.distributed.run; --force-collective puts the
    native RCCL all-gather of the term counts into the timed region."""
    p = run(["--gpus", "1", "--spawn", "--force-collective", "--batch", "256", "--slots", "32", "--steps", "2",
             "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[:500]                     # RCCL's banner must not reach stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and "ncclAllGather" in d["config"]["collective"] and d["config"]["verified_vs_oracle"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_bench_world_size_logic_rehearsed_on_one_gpu(world):
    """The N > 1 bench path -- ranks started by bench.py, shard ranges from the GLOBAL pair index, host
    barrier, max-over-ranks time, gathered counts verified on rank 0 -- with every rank on GPU 0 and the
    exchange on gloo (RCCL refuses two ranks on one device; --dev-ranks-share-gpu is a development
    rehearsal and says so in config.collective).  value must be the whole job's."""
    p = run(["--gpus", str(world), "--dev-ranks-share-gpu", "--batch", "256", "--slots", "32", "--steps", "2",
             "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["config"]["verified_vs_oracle"] is True
    assert "gloo" in d["config"]["collective"] and "DEVELOPMENT" in d["config"]["collective"]
    assert abs(d["value"] - world * 256 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6
