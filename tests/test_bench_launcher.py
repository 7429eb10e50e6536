"""bench.py --gpus N started plainly must start its N ranks itself: the launcher half is exercised here
without a GPU (VRT_BENCH_LAUNCH_ONLY=1: the ranks rendezvous over gloo and agree on a sum, rank 0 prints the JSON line)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, extra_env, timeout=120):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_plain_start_spawns_one_rank_per_gpu():
    r = _run(["--gpus", "3", "--steps", "1", "--warmup", "0"], {"VRT_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]     # (gloo itself chats on stdout)
    assert len(lines) == 1                                   # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out == {"launcher_test": True, "n_gpus": 3, "rank_sum": 1 + 2 + 3}


def test_a_failing_rank_fails_the_run():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"VRT_BENCH_LAUNCH_ONLY": "1", "VRT_BENCH_FAIL_RANK": "1"}, timeout=60)
    assert r.returncode == 3
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]   # no line that could be mistaken for a result


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "4"], {"VRT_BENCH_LAUNCH_ONLY": "1", "RANK": "0", "WORLD_SIZE": "2", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "does not match" in r.stderr


def test_torchrun_form_still_works():
    env = {"VRT_BENCH_LAUNCH_ONLY": "1"}
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29713", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=e, capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["rank_sum"] == 3
