#!/usr/bin/env python3
"""bench.py -- element-steps/s of the dynamics core (mEVP sub-cycle + DG2 transport) on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU.  Either the caller starts them (`python -m torch.distributed.run --nproc-per-node N
... bench.py --gpus N`, RANK / LOCAL_RANK / WORLD_SIZE in the environment) or bench.py starts them itself: the
plain command above, run without WORLD_SIZE, spawns torch.distributed.run as a child BEFORE anything touches a
GPU, passes rank 0's JSON line through and exits with the children's status.

Workload (BASELINE.json metric "element-steps/sec (dynamics+transport)"; the >=40 % HBM-roofline
target is stated on the DG2 mEVP inner loop at 2048x2048): 2048 x 2048 elements, DG2 advected H and A,
CG2 velocity, 8-coefficient stress, 120 mEVP sub-iterations + one SSP-RK3 transport step per model
step, fp64, synthetic 512 km box test.  One "step" = one model time step of the whole grid; an
"element-step" is everything done to one element in it.  With N GPUs the SAME grid
is split into N row blocks (strong scaling) with ghost-row send/recv over RCCL.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (dominant kernel: one pass
of the fused mEVP kernel, HIP-event timed in this run; `algorithmic_bytes_per_launch` is the figure of SURVEY.md
section 8(d), 896 B per element and sub-iteration x the sub-iterations of a launch, and `survey_8d_ratio` its rate
against 8 TB/s -- NOT a fraction of anything physical for a kernel that keeps intermediate iterates on chip, it exceeds
1; `frac` = `frac_compulsory` is bounded by 1: the bytes a pass MUST move, 776 B per element, against 8 TB/s) and
`cpu_baseline` (the CPU
oracle = this repo's own restatement, "port", timed on a bounded sample of the same 2048 x 2048 workload: a few of the
step's 120 sub-iterations and its transport step).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from nextsimdg_amd import basis, abi, rowblock, synthetic  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_ELEM_SUBITER = 896  # byte model of ONE mEVP sub-iteration per HBM round trip
BYTES_COMPULSORY_PER_PASS = 776  # what a pass of a fused kernel must move per element whatever it computes on chip
SHADER_CLOCK_PEAK_HZ = 2.4e9  # MI355X_MICROARCH.md: peak engine clock
BYTES_TRANSPORT = 1008  # DG2, 2 fields, RK3
# fp64 flops of one element-sub-iteration, counted in the ISA of mevp_fused4_kernel (tools/isa_flops.py: 373 v_fma_f64 x 2 + 336
# v_add / v_mul + 17 v_rcp / v_rsq per lane and march step; min / max / compares / moves count 0); tests/test_bench_launch.py
# re-counts it when hipcc is present
# ISA counts of one march step of mevp_fused4_kernel (tools/isa_flops.py; re-counted by tests/test_bench_launch.py): uniform alpha, beta /
# the adaptive form (index: adaptive)
FP64_FLOPS_PER_ELEMENT_SUBITER = {False: 1099, True: 1133}
FP64_FMA_PER_ELEMENT_SUBITER = {False: 373, True: 379}
FP64_ISSUE_SLOTS_PER_ELEMENT_SUBITER = {False: 373 + 336 + 4 * 17 + 6, True: 379 + 356 + 4 * 19 + 32}  # full-rate fp64 instructions + the quarter-rate rcp / rsq as four slots each
# v_fma_f64 flat out with ONE wave per SIMD -- this kernel's occupancy -- on an MI355X of this pool: 63.6 TFLOP/s at 2.4 GHz and
# 1245 W (profiles/r04_fp64_energy_valu_vs_mfma.txt; 70.6 / 73.2 with 2 / 4 waves per SIMD); the arithmetic peak is
# 256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz = 78.6
FP64_VALU_CEILING_TFLOPS = 63.6
FP64_VALU_PEAK_TFLOPS = 78.6
COMM_DEADLINE_S = float(os.environ.get("NSDG_COMM_TIMEOUT_S", "300"))  # a rank that has died must not block the others for ever; 0 = wait for ever
# the same number for torch's process group, where 0 would mean "fail at once": the library's "for ever" becomes a day
TORCH_TIMEOUT_S = COMM_DEADLINE_S if COMM_DEADLINE_S > 0 else 86400.0


class stdout_to_stderr:
    """RCCL prints a start-up banner (version, hostname, library path) on STDOUT when its first communicator is made;
    stdout of this program carries exactly one JSON line, so file descriptor 1 points at stderr while communicators
    are being created"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


DIAGNOSTICS_FAILED = []


def optional(name, fn, *a, **kw):
    """an optional diagnostic of the line (exchange statistics, copy ceiling, offline counters, CPU baseline): a failure
    degrades its field to null and is named in `diagnostics_failed` -- it never costs the run its JSON line"""
    try:
        return fn(*a, **kw)
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 -- anything: the measurement itself is already complete
        DIAGNOSTICS_FAILED.append("%s: %s: %s" % (name, type(e).__name__, str(e)[:300]))
        sys.stderr.write("bench.py: optional diagnostic %s failed: %r\n" % (name, e))
        return None


def failure_line(args, world, where, err):
    """the line of a run that failed in its measured part: the contract's keys with value null and the error text, so
    that a BENCH / SCALE record says what happened instead of being empty"""
    return {"metric": "element-steps/sec (dynamics+transport)", "value": None, "unit": "element-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "%dx%d DG2 transport + mEVP (%d sub-iterations)" % (args.nx, args.ny, args.nsub)},
            "error": "%s: %s" % (type(err).__name__, str(err)[:1000]), "failed_in": where}


def host_cores():
    """usable host cores: the scheduler affinity, capped by the cgroup CPU quota of the box (a one-GPU box
    shows all hardware threads but grants a share of them; oversubscribing that share with one OpenMP thread
    per visible CPU made the all-cores baseline 3x instead of 10x the single-thread figure)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                quota = int(txt[0])
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if quota > 0:
                    n = min(n, max(1, int(quota / period + 0.5)))
            break
        except Exception:
            continue
    return n


def set_omp_threads(n):
    """thread count of the OpenMP build of the oracle: a launcher (torch.distributed.run) exports OMP_NUM_THREADS=1 and
    torch has initialised libgomp with it by the time this runs, so the environment variable alone is too late"""
    import ctypes

    os.environ["OMP_NUM_THREADS"] = str(n)
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass
    return int(n)


def copy_peak_gbs(ctx, device, mib=1024, reps=10):
    """measured device-copy ceiling of THIS box: a 16-byte-per-lane streaming copy (nsdg_copy_f64) of `mib` MiB -- four
    times the 256 MiB Infinity Cache -- timed with HIP events on the context's stream; GB/s counts read + write"""
    n = mib * (1 << 20) // 8
    src = torch.ones(n, dtype=torch.float64, device=device)
    dst = torch.empty_like(src)
    for _ in range(3):
        ctx.copy_f64(dst, src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream)
    for _ in range(reps):
        ctx.copy_f64(dst, src)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ok = bool((dst[::4097] == 1.0).all())
    del src, dst
    torch.cuda.empty_cache()
    if not ok:
        raise SystemExit("device copy produced wrong values: invalid run")
    return 2.0 * n * 8 / (ms * 1e-3) / 1e9


def cpu_baseline(nsub_full, nx, ny, budget_s=5.0, adaptive=False):
    """Time the CPU oracle (the tests' checker; here only as the reported baseline) on a BOUNDED sample of the SAME workload, in about
    `budget_s` seconds (round-5 review: the CPU leg dominated the run and hid the GPU's 0.5 s from the driver's monitor): the bench's own
    nx x ny box test (capped at 2048 x 2048), a BAND of its element rows in the middle of the domain -- the oracle's entry points take
    row ranges, every row costs the same arithmetic -- as many mEVP sub-iterations on the band as fit (at least two) and the three
    Runge-Kutta stages of the DG2 transport of H and A on it, single thread (the reference itself is single-threaded);
    the step's cost per element is t_step = nsub * t_subiteration + t_transport.  Then the same with the OpenMP build on the host cores
    the box grants.  adaptive: the sub-iterations in the adaptive form of alpha, beta, as the device runs them."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cores = set_omp_threads(host_cores())
    nx, ny = min(nx, 2048), min(ny, 2048)
    ny_full = ny
    bt = synthetic.BoxTest(nx, ny)
    p = O.mevp_params(**({k: v for k, v in bt.subcycle_parameters(120.0).items() if k.startswith("aevp")} if adaptive else {}))
    # the band, as an array of its own (rows k0 .. k1 of the box test: the same cells, the same fields, evaluated where the band lies;
    # the oracle treats the band's edges as walls, which costs the same arithmetic)
    band = min(ny, max(8, (1 << 18) // nx))  # ~0.26 M elements: a sub-iteration takes ~0.15-0.3 s on one core
    y0 = ((ny - band) // 2) * bt.hy
    H = basis.project_dg(lambda x, y: bt.H0(x, y + y0), nx, band, bt.L, band * bt.hy, 6, nq=3)
    A = basis.project_dg(lambda x, y: bt.A0(x, y + y0), nx, band, bt.L, band * bt.hy, 6, nq=3)
    A[1:] = 0.0
    X, Y = basis.node_coords(nx, band, bt.L, band * bt.hy)
    Y = Y + y0
    uo, vo = np.ascontiguousarray(0.01 * (2 * Y - bt.L) / bt.L), np.ascontiguousarray(0.01 * (bt.L - 2 * X) / bt.L)  # BoxTest.ocean
    ua, va = np.ascontiguousarray(5.0 + 0 * X), np.ascontiguousarray(-3.0 + 0 * X)  # a wind of the cyclone's strength
    k0, k1, ny = 0, band, band
    pg = O.ice_strength(nx, ny, p, H, A)
    cgh, cga = O.dg_to_cg(nx, ny, H), O.dg_to_cg(nx, ny, A)
    tax, tay = O.wind_stress(p, ua, va)
    u, v = np.zeros_like(uo), np.zeros_like(uo)
    un, vn = np.zeros_like(uo), np.zeros_like(uo)
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    alpha_e = np.zeros((ny, nx))
    n_el = nx * band
    adv = O.prepare_advection(nx, ny, 2, u, v)
    # one untimed sub-iteration: first touch of the arrays, the oracle's tables
    O.mevp_stress(nx, ny, k0, k1, bt.hx, bt.hy, p, u, v, pg, *s, dt=120.0, cgh=cgh, cga=cga, alpha_e=alpha_e)
    O.mevp_velocity(nx, ny, k0, k1, bt.hx, bt.hy, 120.0, p, s, (u, v), (un, vn), (u, v), (tax, tay), (uo, vo), cgh, cga, alpha_e=alpha_e)
    u, un, v, vn = un, u, vn, v
    out = {}
    for omp in (False, True):
        try:
            O.lib(omp)
        except Exception:
            continue
        share = 0.45 if not omp else 0.12  # of the budget: sub-iterations; the transport stages come on top
        t0 = time.perf_counter()
        k = 0
        while k < 2 or time.perf_counter() - t0 < budget_s * share:
            O.mevp_stress(nx, ny, k0, k1, bt.hx, bt.hy, p, u, v, pg, *s, omp=omp, dt=120.0, cgh=cgh, cga=cga, alpha_e=alpha_e)
            O.mevp_velocity(nx, ny, k0, k1, bt.hx, bt.hy, 120.0, p, s, (u, v), (un, vn), (u, v), (tax, tay), (uo, vo), cgh, cga, omp=omp, alpha_e=alpha_e)
            u, un, v, vn = un, u, vn, v
            k += 1
        t_sub = (time.perf_counter() - t0) / (k * n_el)
        t0 = time.perf_counter()
        for f in (H, A):  # SSP-RK3: three stages per field
            t1, t2 = np.zeros_like(f), np.zeros_like(f)
            O.transport_stage(nx, ny, k0, k1, bt.hx, bt.hy, 2, 120.0, 0.0, 1.0, f, f, t1, adv, omp=omp)
            O.transport_stage(nx, ny, k0, k1, bt.hx, bt.hy, 2, 120.0, 0.75, 0.25, f, t1, t2, adv, omp=omp)
            O.transport_stage(nx, ny, k0, k1, bt.hx, bt.hy, 2, 120.0, 1.0 / 3.0, 2.0 / 3.0, f, t2, t1, adv, omp=omp)
        t_tr = (time.perf_counter() - t0) / n_el
        out[omp] = (1.0 / (nsub_full * t_sub + t_tr), 1.0 / t_sub, k)
    what = "oracle/dyn_oracle.c on a band of %d element rows in the middle of the bench's own %dx%d box test: %d mEVP sub-iterations%s + the three Runge-Kutta " \
           "stages of the DG2 transport of H and A on the band, the cost per element scaled to %d sub-iterations per step (a bounded sample of the same " \
           "workload: every row costs the same arithmetic); own CPU restatement -- the reference snapshot has no dynamics code to time"
    res = {"value": out[False][0], "unit": "element-steps/s", "cores": 1, "kind": "port", "extrapolated": True,
           "scaled_from_sample": {"subiterations_timed": out[False][2], "subiterations_per_step": nsub_full, "rows_timed": band, "rows": ny_full,
                                  "how": "value = 1 / (subiterations_per_step x measured seconds per element-sub-iteration + measured seconds per "
                                         "element of the transport step), both measured on the band of rows"},
           "sample": what % (band, nx, ny_full, out[False][2], " (adaptive alpha, beta)" if adaptive else "", nsub_full), "subiters_per_s": out[False][1]}
    if True in out:
        res["all_cores"] = {"value": out[True][0], "cores": cores, "subiters_per_s": out[True][1], "extrapolated": True,
                            "scaled_from_sample": {"subiterations_timed": out[True][2], "subiterations_per_step": nsub_full, "rows_timed": band, "rows": ny_full},
                            "sample": "OpenMP build, the same band of %d rows of the %dx%d grid, %d sub-iterations + the transport stages" % (band, nx, ny_full, out[True][2])}
    return res


KERNEL_SOURCES = ("mevp_fused4.hip", "mevp_p2p.h", "mevp_pipeline.h", "mevp_fused.hip", "mevp_common.h", "transport.hip")


def kernel_source_hash():
    """identifies the kernel sources an offline profile belongs to"""
    import hashlib

    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, "nextsimdg_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def offline_counters(nx, ny, kernel):
    """Counter figures of the dominant kernel from the committed rocprofv3 --pmc passes (bench.py cannot run the
    profiler on itself, so these are OFFLINE numbers of the same command, labelled as such): HBM-side bytes per
    launch (FETCH_SIZE doubled as the gfx950 correction requires, WRITE_SIZE as is) and VALU wave-instructions
    per launch.  `stale` tells whether the kernel sources changed since the profile was taken."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic_latest.json")))
        if d.get("nx") == nx and d.get("ny") == ny and kernel in d["kernels"]:
            k = d["kernels"][kernel]
            return {"traffic": k["total_bytes"], "valu_insts": k.get("valu_wave_insts"),
                    "source": "profiles/%s (offline rocprofv3 --pmc passes of this command, %s)" % (
                        d.get("file", "hbm_traffic_latest.json"), d.get("config", "")),
                    "stale": d.get("kernel_source_hash") != kernel_source_hash()}
    except Exception:
        pass
    return None


def column_bench(args, device):
    """Secondary workload: the reference's column-physics step (the only per-element path the snapshot
    contains) on nx*ny seeded elements; 160 B / element-step algorithmic traffic."""
    n = args.nx * args.ny
    ctx = abi.Context(device)
    state, forcing, newice = synthetic.column_fields(n)
    put = lambda a: torch.from_numpy(a).to(device)
    ds, df, dn = {k: put(v) for k, v in state.items()}, {k: put(v) for k, v in forcing.items()}, put(newice)
    for _ in range(max(args.warmup, 1)):
        ctx.column_step(600.0, ds, df, dn)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(ctx.stream)
    for _ in range(args.steps):
        ctx.column_step(600.0, ds, df, dn)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / args.steps
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    m = 1 << 20
    st, fo, ni = synthetic.column_fields(m)
    O.column_step(O.column_params(), 600.0, st, fo, ni)
    c0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - c0 < 10.0:
        O.column_step(O.column_params(), 600.0, st, fo, ni)
        reps += 1
    cpu = reps * m / (time.perf_counter() - c0)
    cores = set_omp_threads(host_cores())
    big = 1 << 24  # large enough for every host core to have work
    st, fo, ni = synthetic.column_fields(big)
    O.column_step(O.column_params(), 600.0, st, fo, ni, omp=True)
    c0 = time.perf_counter()
    reps_omp = 0
    while time.perf_counter() - c0 < 5.0:
        O.column_step(O.column_params(), 600.0, st, fo, ni, omp=True)
        reps_omp += 1
    cpu_all = reps_omp * big / (time.perf_counter() - c0)
    achieved = n * 160 / (ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "element-steps/sec (column physics)", "value": n * args.steps / elapsed, "unit": "element-steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "column-physics step (DevStep::iterate equivalent) on %d seeded elements, dt=600 s, default modules" % n},
        "roofline": {"bound": "hbm", "kernel": "column_step_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": n * 160, "avg_launch_ms": ms},
        "cpu_baseline": {"value": cpu, "unit": "element-steps/s", "cores": 1, "kind": "port",
                         "sample": "oracle/column_oracle.c, %d steps of 2^20 seeded elements, single thread" % reps,
                         "all_cores": {"value": cpu_all, "cores": cores, "sample": "OpenMP build, %d steps of 2^24 elements" % reps_omp}}}), flush=True)


def transport_bench(args, device):
    """BASELINE config 2: DG advection-only rotating patch (default 512x512 DG1, SSP-RK2); a step = one RK
    step of one field.  Algorithmic traffic per element-step: stages x (read phi + write phi'
    + DG velocity + owned edge-normal velocities) = 256 B for DG1, 504 B for DG2 (one field)."""
    order = args.order
    n = args.nx
    ctx = abi.Context(device)
    if args.transport_variant is not None:
        ctx.set_transport_variant(args.transport_variant)
    ctx.set_grid(n, n, 1.0 / n, 1.0 / n)
    phi, u, v, _ = synthetic.rotating_patch(n, n, order)
    nc, ng = {0: 1, 1: 3, 2: 6}[order], order + 1
    z = lambda *sh: torch.zeros(*sh, dtype=torch.float64, device=device)
    put = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    adv = (z(nc, n, n), z(nc, n, n), z(ng, n, n + 1), z(ng, n + 1, n))
    ctx.prepare_advection(order, put(u), put(v), *adv)
    d = put(phi)
    scratch = z(2 * phi.size)
    dt = 0.1 / (2 * order + 1) * (1.0 / n) / np.pi
    m0 = float(d[0].sum())
    # --transport-fused (default for this workload): all stages of a step in ONE launch (nsdg_transport_step_oop), the field
    # ping-pongs between two buffers; --no-transport-fused: one launch per stage (nsdg_transport_step, in place)
    fused = not args.no_transport_fused
    d2 = torch.empty_like(d)
    if fused:
        # the two paths are bit-identical: checked here on the live field before anything is timed
        chk = d.clone()
        ctx.transport_step(order, dt, [chk], adv, scratch)
        ctx.transport_step_oop(order, dt, [d], [d2], adv)
        if not torch.equal(chk, d2):
            raise SystemExit("transport bench: the fused-stages step differs from the staged step: invalid run")
        del chk

    def step(k):
        if fused:
            a, b = (d, d2) if k % 2 == 0 else (d2, d)
            ctx.transport_step_oop(order, dt, [a], [b], adv)
        else:
            ctx.transport_step(order, dt, [d], adv, scratch)

    nwarm = max(args.warmup, 1) + (max(args.warmup, 1) % 2)  # an even number of steps: the field is back in d
    nsteps = args.steps + (args.steps % 2 if fused else 0)
    for k in range(nwarm):
        step(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(ctx.stream)
    for k in range(nsteps):
        step(k)
    e1.record(ctx.stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / nsteps
    if abs(float(d[0].sum()) - m0) > 1e-11 * abs(m0):
        raise SystemExit("transport bench lost mass: invalid run")
    args.steps = nsteps
    stages = order + 1
    per_stage = 8 * nc * 2 + 2 * 8 * nc + 2 * ng * 8
    alg = n * n * stages * per_stage
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    m = 256
    ph, uu, vv, _ = synthetic.rotating_patch(m, m, order)
    adv_o = O.prepare_advection(m, m, order, uu, vv)
    c0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - c0 < 8.0:
        O.transport_step(m, m, 1.0 / m, 1.0 / m, order, dt, ph, adv_o)
        reps += 1
    cpu = reps * m * m / (time.perf_counter() - c0)
    # every stage reads the field (as phis and phi0), the element and edge velocities and writes the field.  The
    # fused march reads all of that ONCE per step and writes once: its compulsory bytes are one stage's, and `frac` is priced on
    # them (a fraction of the HBM peak above 1 would otherwise appear: the 8(d) model is no lower bound for a fused step)
    compulsory = n * n * per_stage if fused else alg
    achieved = compulsory / (ms * 1e-3) / 1e9
    print(json.dumps({
        "metric": "element-steps/sec (DG%d transport)" % order, "value": n * n * args.steps / elapsed, "unit": "element-steps/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%dx%d DG%d advection-only rotating patch, SSP-RK%d, 1 field" % (n, n, order, stages)},
        "roofline": {"bound": "hbm", "kernel": ("transport_march_kernel<%d> (all %d stages in one launch)" if fused else "transport_stage_kernel<%d> x %d") % (order, stages),
                     "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "frac_definition": "compulsory bytes of a step (field read once and written once, velocities read once: %d B per element) / time / HBM peak" % per_stage
                     if fused else "SURVEY 8(d) bytes of the stages / time / HBM peak",
                     "compulsory_bytes_per_launch": compulsory if fused else alg / stages,
                     "algorithmic_bytes_per_launch": alg if fused else alg / stages,
                     "survey_8d_ratio": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "working_set_bytes": int(2 * phi.size * 8 + sum(a.numel() for a in adv) * 8),
                     "working_set_note": ("field in + field out + velocities of this grid fit into the 256 MB Infinity Cache: `frac` compares a cache-resident "
                                          "launch with the HBM peak -- a throughput figure, not a statement about HBM") if (2 * phi.size * 8 + sum(a.numel() for a in adv) * 8) < 256e6
                     else "larger than the 256 MB Infinity Cache: the bytes come from HBM",
                     "avg_launch_ms": ms if fused else ms / stages, "launches_per_step": 1 if fused else stages,
                     "self_check": "fused-stages step == staged step bitwise on the live field" if fused else None},
        "cpu_baseline": {"value": cpu, "unit": "element-steps/s", "cores": 1, "kind": "port",
                         "sample": "oracle/dyn_oracle.c, %d RK steps of a 256x256 grid, single thread" % reps}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", type=int, default=1, help="DG order of the transport workload")
    ap.add_argument("--workload", choices=["dynamics", "column", "coupled", "transport"], default="dynamics",
                    help="dynamics (default, BASELINE metric), column physics only, or dynamics + column thermodynamics (config 5)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nx", type=int, default=2048)
    ap.add_argument("--ny", type=int, default=2048)
    ap.add_argument("--nsub", type=int, default=120)
    ap.add_argument("--variant", type=int, default=None, help="mEVP kernel variant (default: library default)")
    ap.add_argument("--strip-rows", type=int, default=None, help="rows per strip of the fused mEVP kernel")
    ap.add_argument("--transport-variant", type=int, default=None, help="transport stage kernel: 0 one element per lane, 2 two elements per lane (default: library default = 2)")
    ap.add_argument("--occupancy", type=int, default=None, help="waves/SIMD budget of the fused mEVP kernel (1 or 2)")
    ap.add_argument("--passes-per-exchange", type=int, default=2,
                    help="N > 1: mEVP kernel passes (v = 4 sub-iterations each with the default kernel) between two ghost-row exchanges "
                         "(ghost depth v k / v k - 1 rows); default 2: the best of the rehearsed 8-block runs at both modelled link rates "
                         "(DESIGN.md section 8; rounds 4-5 used 3)")
    ap.add_argument("--tune-passes", type=str, default="",
                    help="N > 1, explicit opt-in: comma-separated candidates for --passes-per-exchange, e.g. 2,3,6: each is run for two un-timed "
                         "steps in the warm-up (a context and a communicator of its own) and the fastest is kept.  Off by default: a plain "
                         "N-rank run creates exactly ONE library communicator")
    ap.add_argument("--no-transport-fused", action="store_true", help="transport workload: one launch per Runge-Kutta stage instead of all stages in one launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--halo", choices=["native", "torch"], default="native",
                    help="N > 1: ghost-row exchange through the C ABI (nsdg_halo_*, RCCL calls and pack kernels in libnsdg.so) or through torch.distributed P2P ops")
    ap.add_argument("--driver", choices=["native", "python"], default="native",
                    help="sub-cycle and transport of a step as one C call each (nsdg_rb_*_run) or as the Python sequence of the same launches")
    ap.add_argument("--graph", action="store_true", help="native driver: replay the launches between two exchanges as one hipGraph")
    ap.add_argument("--no-guard", action="store_true", help="skip the fused-pass == single-iterations check (timing experiments with "
                    "deliberately wrong diagnostic builds only; the line then says 'finite fields' as its self-check)")
    ap.add_argument("--no-closure", action="store_true", help="the bare scheme of rounds 1-4: no ridging cap / scaling limiter in the transport, no "
                    "ice-free-node rule (A/B of what the closure costs; the default run has it ON, as the hosts do)")
    ap.add_argument("--subcycle", choices=["adaptive", "adaptive_converged", "keep_alpha", "keep_delta_min"], default="adaptive",
                    help="how the sub-cycle satisfies its stability bound (nsdg_mevp_stable_params): adaptive (default since round 6) = local, "
                         "solution-adaptive alpha and beta (Kimmritz et al. 2016) at the literature's Delta_min = 2e-9, alpha_min = 50; adaptive_converged = the same "
                         "with the alpha_min that lets 120 sub-iterations converge on the mesh; keep_alpha (round 5) = uniform "
                         "alpha = beta = 1500 with the Delta_min the mesh needs for it; keep_delta_min (rounds 1-4) = uniform alpha = beta from the "
                         "bound for Delta_min = 2e-9")
    ap.add_argument("--delta-min", type=float, default=None, help="regularisation of Delta [1/s] instead of 2e-9 (adaptive, keep_delta_min) / as a floor (keep_alpha)")
    ap.add_argument("--dry-run", action="store_true", help="plumbing check without a GPU (gloo): launch, rendezvous, planning; no metric")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("NSDG_BENCH_SELF_LAUNCH")):
        return self_launch(args.gpus)  # NSDG_BENCH_SELF_LAUNCH: rehearse the launcher path with one rank on a one-GPU box
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if args.dry_run:
        return dry_run(args, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the dynamics core has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or bool(os.environ.get("NSDG_FORCE_DIST"))  # NSDG_FORCE_DIST: rehearse the RCCL set-up with one rank
    from nextsimdg_amd import build

    if rank == 0:
        build.build_lib(verbose=False)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        with stdout_to_stderr():
            import datetime

            # ONE RCCL communicator per rank -- the library's, which moves the ghost rows.  What torch.distributed does here is control
            # plane only (the 128 bytes of the RCCL id, the barriers around the timed region, MAX / MIN over the ranks of a few doubles,
            # the per-rank reports): the gloo backend on CPU tensors.  Two communicators on one device were ordered against each other
            # by convention only (round-5 review); --halo torch (ghost rows as torch P2P ops on device tensors) still needs nccl.
            if args.halo == "torch":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=datetime.timedelta(seconds=TORCH_TIMEOUT_S))
            else:
                dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=TORCH_TIMEOUT_S))
            dist.barrier()  # the first collective creates torch's communicator (and waits for rank 0's build)
            torch.cuda.synchronize()
    red_device = device if (use_dist and args.halo == "torch") else torch.device("cpu")  # where the tensors of torch's reductions live

    if args.workload in ("column", "transport"):
        if world != 1:
            raise SystemExit("the %s workload is a single-GPU measurement" % args.workload)
        return column_bench(args, device) if args.workload == "column" else transport_bench(args, device)

    nx, ny, nsub = args.nx, args.ny, args.nsub
    L = 512e3
    dt = 120.0
    bt = synthetic.BoxTest(nx, ny, L)
    # The sub-cycle's parameters (nsdg_mevp_stable_params through synthetic.BoxTest.subcycle_parameters).  Default: local, solution-adaptive
    # alpha and beta at the standard Delta_min = 2e-9 that SURVEY 8(d) names; --subcycle keep_alpha: the uniform alpha = beta = 1500 of
    # SURVEY 8(d) with the regularisation this mesh needs for it (round 5); keep_delta_min: uniform alpha from the bound (28 875 at 2048^2)
    sub = bt.subcycle_parameters(dt, mode=args.subcycle, delta_min=args.delta_min)
    alpha = sub["alpha"]
    coupled = args.workload == "coupled"
    # One-GPU rehearsal of the WHOLE N-rank code path (NSDG_BENCH_LOOPBACK_WORLD=W): this process plays the interior block W/2
    # of W, both neighbours are the rank itself, every exchange a real RCCL send/recv group -- the values wrap around, so
    # the line carries timings and the `ranks` object but NO metric value.
    loop_world = int(os.environ.get("NSDG_BENCH_LOOPBACK_WORLD", "0"))
    if loop_world and (world != 1 or loop_world < 3):
        raise SystemExit("NSDG_BENCH_LOOPBACK_WORLD needs a single process and a world of at least 3 (an interior block)")
    eff_world, eff_rank = (loop_world, loop_world // 2) if loop_world else (world, rank)
    H, A = bt.dg_fields()
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    column = None
    if coupled:
        # thermodynamic forcing held constant in time: smooth analytic fields in the ranges of SURVEY.md section 8(d)
        # with the mixed layer at the freezing point (synthetic.column_fields_smooth explains why not the per-element
        # random fields of the column-kernel bench: those blow the coupled model up within ~10 steps)
        cs, cf = synthetic.column_fields_smooth(nx, ny, L)
        column = {**cs, **cf}
        del cs, cf

    def build_core(kpass):
        """context, row block, exchanger and driver for `kpass` passes per exchange, fields loaded"""
        c = abi.Context(device)
        if args.variant is not None:
            c.set_mevp_variant(args.variant)
        if args.strip_rows is not None:
            c.set_mevp_strip_rows(args.strip_rows)
        if args.occupancy is not None:
            c.set_mevp_occupancy(args.occupancy)
        if args.transport_variant is not None:
            c.set_transport_variant(args.transport_variant)
        c.set_mevp_params(c.mevp_default_params(**sub, **(dict(min_conc=0.0, min_thick=0.0) if args.no_closure else {})))
        b, d = plan_blocks(c.mevp_variant, kpass, nx, ny, eff_rank, eff_world)
        ex = None
        if eff_world > 1 or os.environ.get("NSDG_FORCE_DIST"):
            with stdout_to_stderr():  # the library's own RCCL communicator
                ex = make_exchanger(args.halo, c, b, device, loopback=bool(loop_world))
            if hasattr(c, "comm_deadline"):
                c.comm_deadline(COMM_DEADLINE_S)
        nat = args.driver == "native" and (ex is None or isinstance(ex, rowblock.NativeHaloExchanger))
        co = (rowblock.CoupledCore if coupled else rowblock.DynamicsCore)(c, b, L / nx, L / ny, dt, nsub, device, exchanger=ex, native=nat,
                                                                          use_graph=args.graph, closure=not args.no_closure)
        if coupled:
            co.load_column(column)
        co.load_global(H, A, uo, vo, ua, va)
        return c, b, d, ex, nat, co

    def drain(c, ex):
        # drain this rank's own work first, then meet the others on the host (torch's barrier is gloo: no device work, the library's RCCL
        # communicator is the only one this rank owns).  With neighbours the drain is BOUNDED
        # (nsdg_ctx_synchronize polls the streams against the communicator's deadline): a rank whose neighbour has died
        # leaves with a non-zero status instead of waiting in ncclRecv for ever; the launcher then ends the others.
        if ex is not None:
            try:
                c.synchronize()
            except abi.NsdgError as e:
                sys.stderr.write("bench.py rank %d: %s\n" % (rank, e))
                sys.stderr.flush()
                os._exit(3)  # no destructors: they would synchronise a device that cannot drain
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Passes per exchange: a fixed 2 by default (with a transfer time proportional to the bytes few passes
    # per exchange win -- 2 was the best of the rehearsed 8-block runs at both modelled link rates -- and the first runs on real links
    # should be boring: ONE context, ONE library communicator).  What a
    # transfer costs on the machine -- mostly latency or mostly bandwidth -- is only known there, so --tune-passes K1,K2,...
    # tries the candidates during the warm-up, two un-timed steps each, and keeps the fastest (max over ranks).
    tune = None
    kpass = max(args.passes_per_exchange, 1)
    cands = [int(x) for x in args.tune_passes.split(",") if x.strip()] if eff_world > 1 else []
    if cands:
        tune = {}
        for cand in cands:
            c_, b_, d_, ex_, nat_, co_ = build_core(cand)
            if d_ in [t["ghost_depth"] for t in tune.values()]:  # short blocks cap the depth: both candidates are the same plan
                co_.close()
                co_ = ex_ = None
                c_.close()
                continue
            co_.step()
            drain(c_, ex_)
            t0 = time.perf_counter()
            for _ in range(2):
                co_.step()
            drain(c_, ex_)
            tt = torch.tensor([(time.perf_counter() - t0) / 2], dtype=torch.float64, device=red_device)
            if use_dist:
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tune[cand] = {"ms_per_step": 1e3 * float(tt[0]), "ghost_depth": d_}
            co_.close()  # driver plans first, then the context with the library's communicator
            co_ = ex_ = None
            c_.close()
        kpass = min(tune, key=lambda k_: tune[k_]["ms_per_step"])
    ctx, blk, depth, exchanger, native, core = build_core(kpass)
    del H, A, uo, vo, ua, va, column

    def sync():
        drain(ctx, exchanger)

    where = "warm-up"
    try:
        for _ in range(args.warmup):
            core.step()
        sync()
        optional("exchange_stats(reset)", exchange_stats, core, reset=True)  # count the exchanges of the timed region only
        # Timed region: EXACTLY what core.step() does (column thermodynamics when coupled, per-step preparation, the
        # sub-cycle, transport), with HIP events on the context's stream around the sub-cycle for the dominant kernel
        where = "timed region"
        ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(args.steps)]
        epoch0 = time.time()  # wall-clock start of the timed region: lets a monitor's samples (rocm-smi) be matched to it
        t0 = time.perf_counter()
        for k in range(args.steps):
            ev[k][0].record(ctx.stream)
            core._set_grid()
            if coupled:
                core.thermodynamics()
            core.prepare()
            ev[k][1].record(ctx.stream)
            core.subcycle()
            ev[k][2].record(ctx.stream)
            core.transport()
            ev[k][3].record(ctx.stream)
        sync()
        elapsed = time.perf_counter() - t0
        epoch1 = time.time()
        where = "reduction over the ranks"
        own_elapsed = elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_device)
        if use_dist:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        cycle_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in ev]))  # one sub-cycle (nsub sub-iterations), this rank
        rank_report = {"rank": eff_rank, "rows_owned": blk.r1 - blk.r0, "rows_local": blk.ny, "ghost_rows_below": blk.gb, "ghost_rows_above": blk.gt,
                       "cycle_ms": cycle_ms, "prepare_ms": float(np.mean([e[0].elapsed_time(e[1]) for e in ev])),
                       "transport_ms": float(np.mean([e[2].elapsed_time(e[3]) for e in ev])),
                       "step_gpu_ms": float(np.mean([e[0].elapsed_time(e[3]) for e in ev])), "step_wall_ms": 1e3 * own_elapsed / args.steps}
        rank_report.update(optional("exchange_stats", exchange_stats, core, steps=args.steps) or {})
        reports = [rank_report]
        if use_dist:
            reports = [None] * world
            dist.all_gather_object(reports, rank_report)

        where = "validity checks"
        # ---- validity of the run: finite, non-trivial, and the fused pass still equals single sub-iterations bit for bit
        ok = torch.tensor([float(bool(torch.isfinite(core.u).all() and torch.isfinite(core.H).all())), float(core.u.abs().max())],
                          dtype=torch.float64, device=red_device)
        guard = None if args.no_guard else fused_pass_guard(ctx, core)
        gave_up = ctx.pipeline_waits_given_up()  # bounded waits of the stage-per-wave pipeline that hit their bound: must be none
        if gave_up:
            guard = False
            sys.stderr.write("bench.py rank %d: %d wait(s) of the mEVP pipeline gave up\n" % (rank, gave_up))
        ok = torch.cat([ok, torch.tensor([float(guard is not False)], dtype=torch.float64, device=red_device)])
        if use_dist:
            lo = ok.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN)
            dist.all_reduce(ok, op=dist.ReduceOp.MAX)
            ok[0], ok[2] = lo[0], lo[2]
        if (ok[0] == 0.0 or ok[1] == 0.0) and not args.no_guard:  # --no-guard: a timing-only build computes on garbage
            raise SystemExit("bench produced non-finite or trivial fields: invalid run")
        if ok[2] == 0.0:
            raise SystemExit("bench: one pass of the fused kernel differs from single sub-iterations on the live state (or a wait of its pipeline "
                         "gave up): invalid run")

    except (abi.NsdgError, RuntimeError) as e:
        # a failed launch, a broken communicator, a device error, a collective of the reduction / validity part that a dead rank
        # never joins (torch raises after its timeout): rank 0 still prints a line (value null + the error), every rank
        # leaves with a non-zero status WITHOUT destructors (they would synchronise a device that may never drain)
        sys.stderr.write("bench.py rank %d failed in the %s: %r\n" % (rank, where, e))
        if rank == 0:
            print(json.dumps(failure_line(args, world, where, e)), flush=True)
        sys.stderr.flush()
        os._exit(3)
    full, twos, ones = core.passes_per_step()
    per_launch = core.per_pass if full else (2 if twos else 1)
    launches = full if full else (twos if twos else ones)  # launches of the dominant kernel per sub-cycle
    fused_kernel = "mevp_fused4_kernel" if per_launch >= 2 else "mevp_fused_kernel"  # passes of 2, 3 and 4 sub-iterations are one kernel
    if rank == 0:
        n_elem = nx * ny
        value = n_elem * args.steps / elapsed
        own_elems = (blk.r1 - blk.r0) * nx
        # every sub-iteration costs the same arithmetic whichever kernel runs it: a launch's share of the sub-cycle
        # time is its share of the sub-iterations (remainder launches of 2 / 1 sub-iterations are the minority)
        launch_ms = cycle_ms * per_launch / nsub
        compulsory = own_elems * BYTES_COMPULSORY_PER_PASS
        algorithmic = own_elems * BYTES_PER_ELEM_SUBITER * per_launch  # SURVEY.md section 8(d) x the sub-iterations of a launch
        achieved = compulsory / (launch_ms * 1e-3) / 1e9
        survey_gbs = algorithmic / (launch_ms * 1e-3) / 1e9
        off = optional("offline_counters", offline_counters, nx, ny, fused_kernel) if eff_world == 1 else None
        copy_peak = optional("copy_peak_gbs", copy_peak_gbs, ctx, device)
        traffic_gbs = (off["traffic"] / (launch_ms * 1e-3) / 1e9) if off else None  # L2 -> fabric bytes per second (Infinity-Cache hits included)
        adaptive = sub["aevp_c"] > 0
        flops = own_elems * per_launch * FP64_FLOPS_PER_ELEMENT_SUBITER[adaptive]
        tflops = flops / (launch_ms * 1e-3) / 1e12
        roof = {"bound": "fp64-valu/power",
                "bound_note": "what the counters and the power probe support (DESIGN.md section 5): vector-ALU issue of fp64 arithmetic at the socket's power "
                              "cap (1380 W of 1400, shader clock 2.22 of 2.4 GHz: profiles/r05_power_probe.txt) with the L2 -> fabric traffic at `fabric_traffic_frac_of_copy_peak` of the measured "
                              "copy ceiling; HBM bytes alone (`frac`) are not what limits the pass.  `achieved` / `peak` / `frac` stay the HBM figures "
                              "BASELINE.json's target is stated in; the flop side is `fp64_tflops` / `frac_fp64_valu`",
                "kernel": fused_kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_definition": "frac = frac_compulsory = compulsory_bytes_per_launch / avg_launch_ms / 8 TB/s: the bytes one pass must move whatever it "
                                   "keeps on chip; a pass performs subiterations_per_launch sub-iterations on them, so fractions of kernels with "
                                   "different numbers of sub-iterations per pass do not compare -- ms_per_subiteration does",
                "frac_compulsory": achieved / HBM_PEAK_GBS,
                "survey_8d_ratio": survey_gbs / HBM_PEAK_GBS,
                "survey_8d_note": "algorithmic_bytes_per_launch / avg_launch_ms / 8 TB/s with the one-HBM-round-trip-per-sub-iteration byte model of "
                                  "SURVEY.md 8(d); the fused kernel keeps the intermediate stress and velocity of %d sub-iterations on chip, so "
                                  "this ratio is NOT a roofline fraction (it exceeds 1)" % per_launch,
                "copy_peak_GBs": copy_peak, "frac_of_copy_peak": (achieved / copy_peak) if copy_peak else None,
                "copy_peak_note": "measured in this run on this box: nsdg_copy_f64, 16 bytes per lane, 1 GiB read + 1 GiB written per launch "
                                  "(SURVEY.md 8(d): 'additionally against a measured device-copy peak on the box')",
                "traffic": off["traffic"] if off else None,
                "traffic_source": (off["source"] + (" -- STALE: kernel sources changed since" if off["stale"] else "")) if off else None,
                "wasted_traffic_ratio": (off["traffic"] / compulsory) if off else None,
                "algorithmic_bytes_per_launch": algorithmic,
                "algorithmic_bytes_model": "SURVEY.md 8(d): %d B per element and sub-iteration x %d sub-iterations per launch x owned elements" % (
                    BYTES_PER_ELEM_SUBITER, per_launch),
                "compulsory_bytes_per_launch": compulsory,
                "compulsory_bytes_model": "%d B per element and launch: u,v of the 4 owned nodes 64 + ice strength 72 + stress in 192 + nodal coefficients 192 "
                                          "+ stress out 192 + u,v out 64" % BYTES_COMPULSORY_PER_PASS,
                "avg_launch_ms": launch_ms, "launches_per_step": launches, "subiterations_per_launch": per_launch,
                "ms_per_subiteration": launch_ms / per_launch,
                "fabric_traffic_frac": (traffic_gbs / HBM_PEAK_GBS) if off else None,
                "fabric_traffic_frac_of_copy_peak": (traffic_gbs / copy_peak) if (off and copy_peak) else None,
                "fabric_traffic_note": "`traffic` is FETCH_SIZE x 2 + WRITE_SIZE of the offline counter passes: bytes between the XCDs' L2 and the fabric.  "
                                       "Reads served by the 256 MB Infinity Cache are counted, so this is NOT HBM traffic and may exceed the copy ceiling "
                                       "(round 4 called the first figure hbm_physical_frac: renamed)",
                "fp64_flops_per_launch": flops, "fp64_tflops": tflops,
                "fp64_flops_model": "%d fp64 flops per element and sub-iteration (ISA count, tools/isa_flops.py: v_fma_f64 = 2) x %d sub-iterations x owned "
                                    "elements; the lanes and rows recomputed at the edges of a strip are not counted" % (FP64_FLOPS_PER_ELEMENT_SUBITER[adaptive], per_launch),
                "fp64_valu_ceiling_tflops": FP64_VALU_CEILING_TFLOPS,
                "frac_fp64_valu": tflops / FP64_VALU_CEILING_TFLOPS,
                "frac_fp64_valu_note": "against v_fma_f64 flat out at this kernel's occupancy (one wave per SIMD) measured on this pool: 63.6 TFLOP/s at 2.4 GHz, "
                                       "1245 W (profiles/r04_fp64_energy_valu_vs_mfma.txt); arithmetic peak 78.6.  Only %d of the kernel's %d fp64 issue slots "
                                       "per element-sub-iteration are FMAs: a flat-out run of ITS instruction mix would reach %.1f" % (
                                           FP64_FMA_PER_ELEMENT_SUBITER[adaptive], FP64_ISSUE_SLOTS_PER_ELEMENT_SUBITER[adaptive],
                                           FP64_VALU_PEAK_TFLOPS * FP64_FLOPS_PER_ELEMENT_SUBITER[adaptive] / (2.0 * FP64_ISSUE_SLOTS_PER_ELEMENT_SUBITER[adaptive])),
                "valu_issue_frac": valu_issue_frac(off["valu_insts"], launch_ms, ctx) if off and off.get("valu_insts") else None}
        line = {
            "metric": "element-steps/sec (dynamics+transport)", "value": None if (loop_world or args.no_guard) else value, "unit": "element-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("column thermodynamics + " if coupled else "") + "%dx%d DG2 transport (H,A; SSP-RK3) + mEVP (%d sub-iterations, CG2 velocity, DG8 stress), "
                                   "512 km box test, dt=120 s, %s, Delta_min=%.2e 1/s (creep below %.3g %% per day)" % (
                                       nx, ny, nsub,
                                       ("local, solution-adaptive alpha and beta (Kimmritz et al. 2016: alpha_e^2 = max(%.0f^2, %.2f zeta_e dt / (m_e |K|)), beta_n = max(alpha_min, max over "
                                        "the adjacent elements of alpha_e m_e / m_n)) at the standard parameters of SURVEY 8(d)" % (sub["aevp_alpha_min"], sub["aevp_c"])) if sub["aevp_c"] > 0 else
                                       ("uniform alpha=beta=%.0f (%s; the %d sub-iterations move the sub-cycle %.1f %% of the way per model step)" % (
                                           alpha, "SURVEY 8(d)'s alpha with the smallest regularisation for which it satisfies the sub-cycle's stability bound on this mesh"
                                           if args.subcycle == "keep_alpha" else "alpha from the stability bound of the sub-cycle on this mesh", nsub, 100.0 * min(1.0, nsub / alpha))),
                                       sub["delta_min"], sub["delta_min"] * 8.64e6),
                       "subcycle": args.subcycle,
                       "decomposition": "%d row block(s), ghost-row send/recv" % eff_world + (
                           ", ghost depth %d/%d rows, one exchange per %d mEVP passes%s, halo=%s" % (
                               depth[0], depth[1], core.group_passes,
                               " (--tune-passes, chosen in the warm-up: %s)" % ", ".join("%d passes %.3f ms/step" % (k_, t_["ms_per_step"]) for k_, t_ in sorted(tune.items())) if tune else "",
                               args.halo) if eff_world > 1 else ""),
                       "mevp_passes": "%s sub-iteration%s per kernel pass" % ({4: "four", 3: "three", 2: "two", 1: "one"}[per_launch], "s" if per_launch > 1 else ""),
                       "mevp_variant": args.variant if args.variant is not None else "default",
                       "closure": "off (--no-closure: the bare scheme)" if args.no_closure else
                                  "on: ridging cap + scaling limiter in the transport epilogue, free drift at ice-free nodes (include/nsdg.h)",
                       "driver": ("native (nsdg_rb_mevp_run / nsdg_rb_transport_run)" + (" + hipGraph replay" if args.graph else "")) if native else "python sequence",
                       "parity": "dynamics parity unpinned (the reference snapshot has no DG/mEVP code); self-check of this run: "
                                 + ("fused pass == single sub-iterations bitwise on the live state" if guard else "finite fields")},
            "roofline": roof,
            "timed_region_epoch": {"start": epoch0, "end": epoch1, "note": "seconds since the Unix epoch on rank 0 around the %d timed steps: the GPU is busy "
                                   "between them and nowhere else for long (the CPU baseline leg that follows takes ~5 s on the host cores)" % args.steps},
            "mevp_element_subiters_per_s": own_elems * nsub / (cycle_ms * 1e-3),
        }
        if use_dist or loop_world:  # N > 1 (and the one-rank rehearsals, which exercise the same code)
            line["ranks"] = ranks_summary(reports)
        if loop_world:
            line["rehearsal"] = ("NOT A MEASUREMENT OF THE METRIC: one GPU plays the interior block %d of %d, both neighbours are the rank itself (real "
                                 "RCCL send/recv groups, values wrap around); ms_per_step is this block's share of the step" % (eff_rank, eff_world))
        if eff_world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = optional("cpu_baseline", cpu_baseline, nsub, nx, ny, adaptive=sub["aevp_c"] > 0)
        if args.no_guard:
            line["timing_only"] = "--no-guard: NOT A MEASUREMENT OF THE METRIC (value null): the validity checks of the run were skipped (timing experiments with builds that compute on wrong values)"
        if DIAGNOSTICS_FAILED:
            line["diagnostics_failed"] = DIAGNOSTICS_FAILED
        print(json.dumps(line), flush=True)
    sync()
    core.close()
    core = exchanger = None
    ctx.close()  # the library's communicator goes before torch's process group
    if use_dist:
        dist.destroy_process_group()


def ranks_summary(reports):
    """the `ranks` object of an N-GPU line: where each rank's time went, so that a SCALE record explains itself"""
    cyc = [r["cycle_ms"] for r in reports]
    exch = [(r.get("mevp_exchange_ms_per_step") or 0.0) + (r.get("transport_exchange_ms_per_step") or 0.0) for r in reports]
    return {"cycle_ms_min": min(cyc), "cycle_ms_max": max(cyc), "step_gpu_ms_min": min(r["step_gpu_ms"] for r in reports),
            "step_gpu_ms_max": max(r["step_gpu_ms"] for r in reports), "exchange_ms_per_step_max": max(exch),
            "slowest_rank": max(reports, key=lambda r: r["step_gpu_ms"])["rank"],
            "note": "per rank: GPU time of a step and of its parts (HIP events on the rank's stream), and the ghost exchanges -- "
                    "time on the communication stream from 'data ready' to 'ghost rows written' (pack + transfer + unpack + "
                    "waiting for a late neighbour; it overlaps with the interior launch of the overlap split)",
            "per_rank": sorted(reports, key=lambda r: r["rank"])}


def exchange_stats(core, steps=None, reset=False):
    """ghost-exchange statistics of the native row-block drivers (nsdg_rb_*_stats): per model step when `steps` is given"""
    out = {}
    for name, run in (("mevp", getattr(core, "_run_mevp", None)), ("transport", getattr(core, "_run_transport", None))):
        st = getattr(run, "stats", None)
        if st is None:
            continue
        d = st(reset)
        if steps and d["exchanges"]:
            out["%s_exchanges_per_step" % name] = d["exchanges"] / steps
            out["%s_exchange_ms_per_step" % name] = d["ms"] / steps
            out["%s_exchange_ms_each" % name] = d["ms"] / max(d["exchanges"] - d["untimed"], 1)
            out["%s_exchange_bytes_sent" % name] = d["bytes_sent"]
            out["%s_exchanges_untimed" % name] = d["untimed"]
    return out


def plan_blocks(variant, passes_per_exchange, nx, ny, rank, world):
    """row block of `rank`: ghost element rows below / above are (v k, v k - 1) for k passes of v sub-iterations
    between two exchanges"""
    kpass = max(1, min(passes_per_exchange, (ny // world) // 16)) if world > 1 else 1
    vpass = min(variant, 4)
    depth = (vpass * kpass, vpass * kpass - 1) if vpass >= 2 else (1, 1)
    return rowblock.RowBlock(nx, ny, rank, world, *depth), depth


def make_exchanger(kind, ctx, blk, device, loopback=False):
    loop = loopback or (bool(os.environ.get("NSDG_FORCE_DIST")) and blk.world == 1)
    if kind == "native":
        return rowblock.NativeHaloExchanger(ctx, blk, device, loopback=loop)
    return rowblock.HaloExchanger(blk, loopback=loop)


def fused_pass_guard(ctx, core):
    """In-bench correctness guard: on the LIVE state after the timed region, one pass of the multi-iteration kernel
    must equal the same number of single sub-iterations bit for bit (stress and velocity), treating the rank's
    local array as a domain of its own.  Returns True (checked, equal), False (differs) or None (the run uses
    the single-iteration kernel only: nothing to compare)."""
    v = core.per_pass
    if v < 2 or core.nsub < v:
        return None
    ny = core.blk.ny
    core._set_grid()
    new = lambda: ([torch.empty_like(x) for x in core.s], (torch.zeros_like(core.u), torch.zeros_like(core.v)))
    (sx, ux), (sy, uy) = new(), new()
    getattr(ctx, "mevp_iterate%d" % v)(0, ny, core.s, core.sb, (core.u, core.v), (core.ub, core.vb), core.packed, core.pg)
    src, usrc = core.s, (core.u, core.v)
    for i in range(v):
        dst, udst = (sx, ux) if i % 2 == 0 else (sy, uy)
        ctx.mevp_iterate(0, 0, ny, src, dst, usrc, udst, core.packed, core.pg)
        src, usrc = dst, udst
    same = all(torch.equal(a, b) for a, b in zip(src, core.sb)) and torch.equal(usrc[0], core.ub) and torch.equal(usrc[1], core.vb)
    return bool(same)


def valu_issue_frac(valu_wave_insts, launch_ms, ctx):
    """VALU wave-instructions of a launch (offline SQ_INSTS_VALU) x 4 cycles (a 64-wide fp64 instruction on a
    16-lane SIMD) / (SIMDs x peak shader clock x launch time): the share of the chip's peak VALU issue slots the
    kernel fills"""
    simds = 4 * ctx.num_cus()
    return valu_wave_insts * 4.0 / (simds * SHADER_CLOCK_PEAK_HZ * launch_ms * 1e-3)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (this
    process has not touched the GPU and never will), let the children's output through and exit with their status"""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.exit(subprocess.call(cmd, env=env))


def exchange_plan(blk, nx, nsub, per_pass):
    """what one rank's ghost exchanges of a model step move (pure arithmetic from the block geometry; the native plans report
    the same sizes through nsdg_halo_counts): bytes per mEVP ghost-zone exchange and direction -- tiled stress rows of the
    three components + node rows of u and v -- the number of such exchanges per step, and the bytes of the one exchange of
    the advected fields"""
    tiled_row = ((nx + 63) // 64) * 8 * 64 * 8  # bytes of one element row of one stress component (tiles of 64 elements x 8 coefficients)
    node_row = (2 * nx + 1) * 8
    up = (blk.depth_below * 3 * tiled_row + 2 * blk.depth_below * 2 * node_row) if blk.above is not None else 0
    down = (blk.depth_above * 3 * tiled_row + (2 * blk.depth_above + 1) * 2 * node_row) if blk.below is not None else 0
    k = max(blk.depth_below // per_pass, 1)
    groups = -(-(nsub // per_pass) // k) + (1 if nsub % per_pass else 0)
    field_row = nx * 6 * 8 * 2  # H and A, 6 DG2 coefficients
    return {"mevp_exchange_bytes_up": up, "mevp_exchange_bytes_down": down, "mevp_exchanges_per_step": groups if blk.world > 1 else 0,
            "transport_exchange_bytes_up": blk.depth_below * field_row if blk.above is not None else 0,
            "transport_exchange_bytes_down": blk.depth_above * field_row if blk.below is not None else 0}


def dry_run(args, rank, world):
    """Plumbing check without a GPU (tests/test_bench_launch.py, gloo): the launch, the rendezvous, the row-block
    planning of the REAL run (ghost depths, rows, bytes per exchange: every rank's entry reaches rank 0 as in the `ranks`
    object of a real line) and the max-over-ranks reduction -- no kernel runs, NO metric is reported."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29512")
    import datetime

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=TORCH_TIMEOUT_S))
    v = abi.DEFAULT_MEVP_VARIANT
    blk, depth = plan_blocks(v, max(args.passes_per_exchange, 1), args.nx, args.ny, rank, world)
    rows = torch.tensor([float(blk.r1 - blk.r0)], dtype=torch.float64)
    dist.all_reduce(rows, op=dist.ReduceOp.SUM)
    dist.barrier()
    t = torch.tensor([float(rank)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    entry = {"rank": rank, "rows_owned": blk.r1 - blk.r0, "rows_local": blk.ny, "row_first": blk.r0, "ghost_rows_below": blk.gb, "ghost_rows_above": blk.gt,
             "neighbour_below": blk.below, "neighbour_above": blk.above}
    entry.update(exchange_plan(blk, args.nx, args.nsub, min(v, 4)))
    entries = [None] * world
    dist.all_gather_object(entries, entry)
    if rank == 0:
        print(json.dumps({"dry_run": True, "metric": None, "value": None, "n_gpus": world, "rows_total": int(rows[0]),
                          "ghost_depth": list(depth), "passes_per_exchange": depth[0] // min(v, 4) if world > 1 else None,
                          "subiterations_per_pass": min(v, 4), "max_rank_seen": int(t[0]),
                          "ranks": sorted(entries, key=lambda e: e["rank"])}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
