"""bench.py --gpus N must start N ranks by itself when no launcher did (the flag was parsed and ignored)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra)
    return env


@pytest.mark.parametrize("n", [2, 4])
def test_gpus_flag_spawns_one_rank_per_gpu(n):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                         env=_clean_env(BENCH_LAUNCH_ONLY="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == list(range(n))
    assert all(l["n_gpus"] == n and l["local_rank"] == l["rank"] for l in lines)
    assert len({l["master"] for l in lines}) == 1 and lines[0]["master"].startswith("127.0.0.1:")


def test_under_a_launcher_the_flag_does_not_spawn_again():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"],
                         env=_clean_env(BENCH_LAUNCH_ONLY="1", RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["rank"] == 1 and lines[0]["n_gpus"] == 2


def test_mismatched_world_size_is_refused():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                         env=_clean_env(BENCH_LAUNCH_ONLY="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2"), capture_output=True, text=True, timeout=120)
    assert out.returncode != 0


def test_a_dead_rank_takes_the_launch_down_instead_of_hanging_it():
    """one rank exits non-zero while the others are blocked (here: asleep) -- the launcher must terminate them and return
    non-zero promptly, like torch.distributed.run, instead of waiting for ranks that will never finish."""
    import time
    t0 = time.monotonic()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                         env=_clean_env(BENCH_LAUNCH_ONLY="1", BENCH_LAUNCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode != 0
    assert time.monotonic() - t0 < 60


def test_the_launch_has_a_wall_clock_limit():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=_clean_env(BENCH_LAUNCH_ONLY="1", BENCH_LAUNCH_FAIL_RANK="7", BENCH_LAUNCH_TIMEOUT_S="2"), capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 124


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_report_n_gpus_2():
    """The real N=2 bench path on a single-GPU box: both ranks on cuda:0, reductions staged through gloo (numbers meaningless)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "16384",
                          "--features", "16", "--no-cpu-baseline", "--large-ensemble", "0"],
                         env=_clean_env(BENCH_SHARE_DEVICE="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["value"] > 0
    # the row-sharded exchange really ran: on a shared device the transport is the torch.distributed hooks over gloo
    assert "torch.distributed" in (lines[0]["config"]["exchange"] or ""), lines[0]["config"]
    assert lines[0]["config"]["sharding"] == "rows x2" and lines[0]["scaling"] == "weak"
    assert lines[0]["collective"]["calls_per_step"] >= 10, lines[0]["collective"]


@pytest.mark.gpu
def test_two_ranks_grow_the_trees_of_one_process_on_the_same_rows(tmp_path):
    """`bench.py --gpus 2` end to end (two ranks sharing cuda:0, reductions through gloo) must grow, tree for tree,
    what ONE process grows on the concatenation of the two ranks' rows (--emulate-ranks 2): structure bit-identical, leaf values equal
    (integer sums: the row partition does not enter any result)."""
    import numpy as np
    common = ["--steps", "2", "--warmup", "1", "--rows", "66000", "--features", "16",   # >= 65 536 rows per rank: the fused transpose + its uint16 digit counts through the exchange
              "--no-cpu-baseline", "--large-ensemble", "0", "--no-extra-legs"]
    a, b = str(tmp_path / "two.npz"), str(tmp_path / "one.npz")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dump-ensemble", a] + common,
                         env=_clean_env(BENCH_SHARE_DEVICE="1"), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--emulate-ranks", "2", "--dump-ensemble", b] + common,
                         env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    ea, eb = np.load(a), np.load(b)
    assert len(ea["tree_indices"]) == 3
    for k in ea.files:
        assert np.array_equal(ea[k], eb[k]), k


@pytest.mark.gpu
def test_the_bench_line_keeps_its_contract():
    """One JSON line with the driver's keys, the `roofline` object of the dominant kernel and the `cpu_baseline` object (kind "reference" or
    "port", cores, sample), at a reduced shape so that the test takes seconds (the numbers are not the headline's)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--rows", "131072", "--features", "32",
                          "--no-extra-legs", "--large-ensemble", "0", "--cpu-sample-rows", "4096"],
                         env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, "exactly ONE json line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 3 * (131072 / float(1 << 20)) / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0
