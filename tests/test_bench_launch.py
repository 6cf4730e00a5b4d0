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
    assert lines[0]["ghost_depth"] == [8, 7]  # 2 passes of the four-iteration kernel between two exchanges


def test_eight_ranks_plan_the_real_2048_blocks():
    """the N = 8 run of the scaling record, as far as it goes without GPUs: eight processes through the launcher path, the real
    2048 x 2048 row blocks with the default passes per exchange (a FIXED 2: ghost depth 8 / 7 with the four-iteration
    kernel), every rank's entry in the line, the rows tile the grid, and the bytes of an exchange are what the geometry says"""
    p = run_bench("--gpus", "8", "--dry-run", "--steps", "1", "--warmup", "0")
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = json_lines(p.stdout)
    assert len(lines) == 1, p.stdout.decode()
    L = lines[0]
    assert L["n_gpus"] == 8 and L["rows_total"] == 2048 and L["max_rank_seen"] == 7 and L["value"] is None
    assert L["ghost_depth"] == [8, 7] and L["passes_per_exchange"] == 2 and L["subiterations_per_pass"] == 4
    ranks = L["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8))
    first = 0
    for r in ranks:
        assert r["row_first"] == first and r["rows_owned"] == 256
        first += r["rows_owned"]
        inner_below, inner_above = r["rank"] > 0, r["rank"] < 7
        assert r["ghost_rows_below"] == (8 if inner_below else 0) and r["ghost_rows_above"] == (7 if inner_above else 0)
        assert r["rows_local"] == 256 + r["ghost_rows_below"] + r["ghost_rows_above"]
        assert r["neighbour_below"] == (r["rank"] - 1 if inner_below else None) and r["neighbour_above"] == (r["rank"] + 1 if inner_above else None)
        # 8 stress rows of 3 components (32 tiles x 8 coefficients x 64 elements x 8 B) + 16 node rows of u and v (4097 nodes) upwards,
        # 7 stress rows + 15 node rows downwards; 120 sub-iterations = 30 passes of 4 = 15 groups of 2
        assert r["mevp_exchange_bytes_up"] == (8 * 3 * 131072 + 16 * 2 * 4097 * 8 if inner_above else 0)
        assert r["mevp_exchange_bytes_down"] == (7 * 3 * 131072 + 15 * 2 * 4097 * 8 if inner_below else 0)
        assert r["mevp_exchanges_per_step"] == 15
        assert r["transport_exchange_bytes_up"] == (8 * 2048 * 96 if inner_above else 0)
    assert first == 2048


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


def test_ranks_summary_of_an_n_gpu_line():
    """the `ranks` object that makes a SCALE record self-explaining: min / max over ranks, the slowest rank, reports in rank order"""
    import bench

    reports = [{"rank": r, "rows_owned": 256, "rows_local": 273, "ghost_rows_below": 9, "ghost_rows_above": 8, "cycle_ms": 5.0 + 0.1 * r,
                "prepare_ms": 0.1, "transport_ms": 0.3, "step_gpu_ms": 6.0 + (0.5 if r == 2 else 0.0), "step_wall_ms": 6.6,
                "mevp_exchange_ms_per_step": 1.0 + 0.2 * r, "transport_exchange_ms_per_step": 0.05} for r in (3, 0, 2, 1)]
    s = bench.ranks_summary(reports)
    assert [r["rank"] for r in s["per_rank"]] == [0, 1, 2, 3]
    assert s["cycle_ms_min"] == 5.0 and abs(s["cycle_ms_max"] - 5.3) < 1e-12 and s["slowest_rank"] == 2
    assert s["step_gpu_ms_max"] == 6.5 and s["step_gpu_ms_min"] == 6.0 and abs(s["exchange_ms_per_step_max"] - 1.65) < 1e-12
    # a rank without native-driver statistics (python driver / torch halo) still summarises
    assert bench.ranks_summary([{"rank": 0, "cycle_ms": 1.0, "step_gpu_ms": 2.0}])["exchange_ms_per_step_max"] == 0.0


def test_failure_line_and_optional_diagnostics():
    """a run that fails in its measured part still has a line (value null + the error); an optional diagnostic that fails
    costs a field, not the line"""
    import argparse

    import bench

    a = argparse.Namespace(steps=5, warmup=2, nx=2048, ny=2048, nsub=120)
    L = bench.failure_line(a, 8, "timed region", RuntimeError("ncclRecv: peer gone"))
    assert L["value"] is None and L["n_gpus"] == 8 and L["failed_in"] == "timed region" and "peer gone" in L["error"]
    for key in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert key in L
    json.dumps(L)
    del bench.DIAGNOSTICS_FAILED[:]
    assert bench.optional("works", lambda x: x + 1, 1) == 2 and not bench.DIAGNOSTICS_FAILED
    assert bench.optional("copy_peak_gbs", lambda: 1 / 0) is None
    assert len(bench.DIAGNOSTICS_FAILED) == 1 and "copy_peak_gbs" in bench.DIAGNOSTICS_FAILED[0] and "ZeroDivisionError" in bench.DIAGNOSTICS_FAILED[0]
    del bench.DIAGNOSTICS_FAILED[:]


def test_roofline_labels_say_what_was_measured():
    """round-4 review, weak point 3: the line must not call fabric traffic HBM traffic, must not call a scaled CPU sample
    'not extrapolated', and carries a flop-side entry next to the byte-side one.  The line itself needs a GPU
    (tests/test_gpu_configs.py checks the keys of a real one); here: the constants and the source of the labels."""
    import bench

    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "hbm_physical_frac\":" not in src.replace("called the first figure hbm_physical_frac", "")
    for key in ('"fabric_traffic_frac"', '"fabric_traffic_frac_of_copy_peak"', '"fp64_flops_per_launch"', '"frac_fp64_valu"', '"scaled_from_sample"',
                '"bound": "fp64-valu/power"', '"frac_definition"', '"survey_8d_ratio"', '"wasted_traffic_ratio"'):
        assert key in src, key
    assert '"extrapolated": False' not in src
    # the flop count is the ISA's: re-counted when the compiler is here (3 s)
    if os.path.exists("/opt/rocm/bin/hipcc"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import isa_flops

        text = isa_flops.listing()
        for adaptive in (False, True):  # uniform alpha, beta / the local, solution-adaptive form (two instantiations of the kernel)
            loops = isa_flops.loops(text, adaptive)
            assert len(loops) == 2
            for _, ins in loops:  # the stages 1-3 and the loader run the same arithmetic
                c = isa_flops.flops(ins)
                assert c["flops"] == bench.FP64_FLOPS_PER_ELEMENT_SUBITER[adaptive], c
                assert c["fma"] == bench.FP64_FMA_PER_ELEMENT_SUBITER[adaptive], c
                assert c["fma"] + c["addmul"] + 4 * c["trans"] + c["other_f64"] == bench.FP64_ISSUE_SLOTS_PER_ELEMENT_SUBITER[adaptive], c
