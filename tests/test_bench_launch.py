"""`python bench.py --gpus N` must work as typed (the driver runs exactly that form): without WORLD_SIZE in the
environment bench.py starts one worker per rank through torch.distributed.run itself.  Checked here without a
GPU through the --dry-run plumbing path (gloo): launch, rendezvous on 127.0.0.1, row-block planning, max-over-ranks
reduction, ONE JSON line from rank 0, exit status of the workers."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=300)


def json_lines(out):
    return [json.loads(l) for l in out.decode().splitlines() if l.startswith("{")]


def test_plain_command_with_two_ranks_self_launches():
    p = run_bench("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout.decode()
    assert lines[0]["dry_run"] is True and lines[0]["value"] is None  # a dry run reports no metric
    assert lines[0]["n_gpus"] == 2 and lines[0]["rows_total"] == 2048 and lines[0]["max_rank_seen"] == 1
    assert lines[0]["ghost_depth"] == [9, 8]  # 3 passes of the three-iteration kernel between two exchanges (profiles/r03_rank_share_bandwidth.txt)


def test_worker_failure_is_reported_by_the_exit_status():
    # the workers refuse a world size that does not match --gpus; the parent must pass the failure on
    p = run_bench("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert p.returncode != 0
    assert b"WORLD_SIZE (3) != --gpus (2)" in p.stderr + p.stdout


def test_without_a_gpu_the_real_bench_fails_loudly():
    import torch

    if torch.cuda.is_available():
        return
    p = run_bench("--gpus", "1", "--steps", "1", "--warmup", "0")
    assert p.returncode != 0 and b"no CPU fallback" in p.stderr + p.stdout
