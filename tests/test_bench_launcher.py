"""`python bench.py --gpus N` must start N ranks by itself (the driver's N=1 call and its torch.distributed.run call both
keep working).  The rank body is stubbed (`--stub-body`): gloo on the CPU, no GPU, so this runs in the build container."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in env_drop}
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=300)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def test_gpus_2_self_launches_two_ranks_and_relays_one_json_line():
    res = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--stub-body", "ok")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = _json_lines(res.stdout)
    assert len(lines) == 1, res.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["steps"] == 3 and lines[0]["warmup"] == 1
    assert lines[0]["value"] == 3.0  # 1 + 2: both ranks took part in the all-reduce


def test_failing_rank_gives_nonzero_exit_code():
    res = _run("--gpus", "2", "--stub-body", "fail")
    assert res.returncode != 0
    assert not _json_lines(res.stdout)


def test_single_rank_runs_in_process():
    res = _run("--gpus", "1", "--stub-body", "ok")
    assert res.returncode == 0, res.stderr[-2000:]
    assert _json_lines(res.stdout)[0]["n_gpus"] == 1
    assert "launcher:" not in res.stderr
