"""bench.py's entry logic without a GPU: `python bench.py --gpus 2` launched PLAINLY (no torchrun environment) must start
its two ranks itself, rendezvous over gloo on 127.0.0.1, average gradients across ranks, take the MAX of the ranks'
times and have rank 0 print exactly one JSON line with the contract's keys.  `--device cpu` runs the same main() on CPU
plumbing (module path, BASELINE config C1) -- a rehearsal of the launch / exchange / timing code, never a measurement."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config"}


def _run(*args, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--device", "cpu", "--config", "c1",
                           "--no-cpu-baseline", *args], capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def _json_lines(stdout):
    return [json.loads(l) for l in stdout.splitlines() if l.startswith("{")]


def test_plain_launch_with_two_ranks_prints_one_line():
    one = _run("--steps", "2", "--warmup", "1")
    assert one.returncode == 0, one.stderr[-2000:]
    two = _run("--gpus", "2", "--steps", "2", "--warmup", "1")
    assert two.returncode == 0, two.stderr[-2000:]
    (a,), (b,) = _json_lines(one.stdout), _json_lines(two.stdout)
    for line, n in ((a, 1), (b, 2)):
        assert REQUIRED <= set(line), REQUIRED - set(line)
        assert line["n_gpus"] == n and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak"
        assert line["config"]["global_batch"] == 128 * n and line["config"]["parallelism"] == f"dp{n}"
        assert line["config"]["model_file"] == "configs/model/c1_core_vae.yaml"
        assert line["metric"].startswith("cells/sec") and line["unit"] == "cells/s" and line["vs_baseline"] is None
        assert abs(line["value"] - 128 * n * 2 / (line["ms_per_step"] * 2e-3)) <= 1e-6 * line["value"]
    assert b["dist_backend"] == "gloo" and b["rccl_ranks"] == 0 and a["dist_backend"] == "none"
    # the same synthetic cells on both ranks' shards would give the same losses; different shards must not, while the
    # averaged update keeps the ranks' parameters -- and so the next loss of rank 0 -- finite and reproducible
    again = _json_lines(_run("--gpus", "2", "--steps", "2", "--warmup", "1").stdout)[0]
    assert again["last_losses"] == b["last_losses"]


def test_plain_launch_with_eight_ranks_rehearses_the_scaling_run():
    """The driver's N = 8 launch shape (one rank per GPU of a node), rehearsed on the gloo backend with CPU plumbing:
    eight ranks rendezvous, exchange gradients every step, agree on the timing (max over ranks) and rank 0 alone prints
    the line -- the world size the first real 8-GPU run will have.  Never a measurement."""
    r = _run("--gpus", "8", "--steps", "2", "--warmup", "1", timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    (line,) = _json_lines(r.stdout)
    assert REQUIRED <= set(line)
    assert line["n_gpus"] == 8 and line["config"]["global_batch"] == 128 * 8 and line["config"]["parallelism"] == "dp8"
    assert line["dist_backend"] == "gloo" and line["rccl_ranks"] == 0 and line["scaling"] == "weak"
    assert abs(line["value"] - 128 * 8 * 2 / (line["ms_per_step"] * 2e-3)) <= 1e-6 * line["value"]
    assert line["finite"]["last_losses"] is True


def test_a_failing_rank_fails_the_launch():
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--config", "c9")
    assert r.returncode != 0 and not _json_lines(r.stdout)


def test_under_torchrun_style_environment_world_must_match():
    r = _run("--gpus", "2", "--steps", "1", "--warmup", "0", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
