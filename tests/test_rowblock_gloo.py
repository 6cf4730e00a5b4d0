"""Multi-rank path on CPU: world_size 2 and 3 over `gloo`.  The row-block driver and halo exchange of
nextsimdg_amd/rowblock.py are the product code under test; the numerical kernels are replaced by the
oracle (tests/oracle_ops.py) because no GPU exists here.  The decomposed run must reproduce the
single-domain run bit for bit (gather formulation + redundant ghost-row stress update)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from nextsimdg_amd import rowblock, synthetic  # noqa: E402

NX, NY, NSUB, NSTEPS = 20, 29, 5, 2


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def column_fields_2d(seed=5):
    state, forcing, _ = synthetic.column_fields(NX * NY, seed)
    f = {k: v.reshape(NY, NX) for k, v in {**state, **forcing}.items()}
    f["wind"] = 0.2 * f["wind"]
    return f


def run_core(rank, world, overlap=True, coupled=False, variant=1, group=1, nsub=NSUB, steps=NSTEPS, resume=None, adaptive=False):
    from oracle_ops import OracleOps

    bt = synthetic.BoxTest(NX, NY)
    rng = np.random.default_rng(41)
    H, A = bt.dg_fields()
    A[0] -= 0.3 * rng.random((NY, NX))
    H[1:3] += 0.02 * rng.standard_normal((2, NY, NX))
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    depth = (variant * group, variant * group - 1) if variant >= 2 else (1, 1)  # `group` passes of `variant` sub-iterations between two ghost exchanges
    blk = rowblock.RowBlock(NX, NY, rank, world, *depth)
    cls = rowblock.CoupledCore if coupled else rowblock.DynamicsCore
    pk = dict(aevp_c=(2.4 * np.pi) ** 2, aevp_alpha_min=3.0, delta_min=2e-7) if adaptive else dict(alpha=200.0, beta=200.0)
    core = cls(OracleOps(mevp_variant=variant, **pk), blk, bt.hx, bt.hy, 120.0, nsub, torch.device("cpu"), overlap=overlap)
    core.load_global(H, A, uo, vo, 3.0 * ua, 3.0 * va)
    if coupled:
        core.load_column(column_fields_2d())
    if resume is not None:  # the state of the whole domain (DynamicsCore.merge_states of the ranks' state_dict()s)
        core.load_state_dict(resume)
    for _ in range(steps):
        core.step()
    return core


def worker(rank, world, port, outdir, overlap=True, coupled=False, variant=1, group=1, nsub=NSUB, adaptive=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        core = run_core(rank, world, overlap, coupled, variant, group, nsub, adaptive=adaptive)
        if variant >= 2 and world > 1:
            assert core.per_pass == variant and core.group_passes == group
        out = {k: core.owned(getattr(core, k)).clone() for k in ("H", "A", "u", "v")}
        out["s11"] = core.owned(core.s[0]).clone()
        if coupled:
            b = core.blk
            out["tice0"] = core.col["tice0"][b.j0:b.j1].clone()
        torch.save(out, os.path.join(outdir, "rank%d.pt" % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def worker_checkpoint(rank, world, port, outdir, variant, adaptive):
    """half of the steps, checkpoint (every rank's state_dict, gathered and merged), a FRESH core resumed from it, the other half"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        group = 1
        first = run_core(rank, world, variant=variant, group=group, nsub=9, steps=1, adaptive=adaptive)
        states = [None] * world
        dist.all_gather_object(states, first.state_dict())
        merged = rowblock.DynamicsCore.merge_states(states)
        assert merged["rows"] == (0, NY) and merged["u"].shape == (2 * NY + 1, 2 * NX + 1) and merged["s12"].shape == (8, NY, NX)
        dist.barrier()
        core = run_core(rank, world, variant=variant, group=group, nsub=9, steps=1, resume=merged, adaptive=adaptive)
        out = {k: core.owned(getattr(core, k)).clone() for k in ("H", "A", "u", "v")}
        out["s11"] = core.owned(core.s[0]).clone()
        torch.save(out, os.path.join(outdir, "rank%d.pt" % rank))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,variant,adaptive", [(2, 4, False), (3, 2, True)])
def test_checkpoint_and_resume_is_exact(world, variant, adaptive, tmp_path):
    """2 steps == 1 step + checkpoint + resume on fresh cores + 1 step, bit for bit, across ranks (the Python driver's state_dict /
    load_state_dict: H, A with their higher DG coefficients, the velocity, the stress -- round-5 review: the driver had no checkpoint);
    adaptive: with local, solution-adaptive alpha and beta, which carry no state of their own from step to step"""
    ref = run_core(0, 1, variant=variant, nsub=9, steps=2, adaptive=adaptive)
    assert float(ref.u.abs().max()) > 1e-5
    port = free_port()
    mp.spawn(worker_checkpoint, args=(world, port, str(tmp_path), variant, adaptive), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for key, full in (("H", ref.H), ("A", ref.A), ("s11", ref.s[0])):
        assert torch.equal(torch.cat([p[key] for p in parts], dim=1), full), key
    for key, full in (("u", ref.u), ("v", ref.v)):
        assert torch.equal(torch.cat([p[key] for p in parts], dim=0), full), key
    # a resume from cell means alone (what the hosts wrote before round 6) is NOT the same run
    lossy = rowblock.DynamicsCore.merge_states([run_core(0, 1, variant=variant, nsub=9, steps=1, adaptive=adaptive).state_dict()])
    lossy["u"][:] = 0.0
    lossy["s11"][:] = 0.0
    other = run_core(0, 1, variant=variant, nsub=9, steps=1, resume=lossy, adaptive=adaptive)
    assert not torch.equal(other.u, ref.u)


@pytest.mark.parametrize("world,variant", [(2, 1), (2, 4)])
def test_adaptive_row_blocks_equal_single_domain_bitwise(world, variant, tmp_path):
    """local, solution-adaptive alpha and beta (round 6): an element's alpha depends on its own nodes, a node's beta on its adjacent
    elements -- the decomposed run still equals the single domain bit for bit"""
    ref = run_core(0, 1, variant=variant, nsub=9, adaptive=True)
    uniform = run_core(0, 1, variant=variant, nsub=9)
    assert float(ref.u.abs().max()) > 1e-5 and not torch.equal(ref.u, uniform.u)
    port = free_port()
    mp.spawn(worker, args=(world, port, str(tmp_path), True, False, variant, 1, 9, True), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for key, full in (("H", ref.H), ("A", ref.A), ("s11", ref.s[0])):
        assert torch.equal(torch.cat([p[key] for p in parts], dim=1), full), key
    for key, full in (("u", ref.u), ("v", ref.v)):
        assert torch.equal(torch.cat([p[key] for p in parts], dim=0), full), key


def test_rowblock_index_bookkeeping():
    for world in (1, 2, 3, 8):
        rows = []
        for r in range(world):
            d = rowblock.RowBlock(16, 37, r, world, 2, 1)
            assert (d.gb, d.gt) == (2 if r > 0 else 0, 1 if r < world - 1 else 0) and d.k0 == max(d.j0 - 1, 0)
            b = rowblock.RowBlock(16, 37, r, world)
            assert b.ny == (b.r1 - b.r0) + b.gb + b.gt
            assert (b.j0, b.j1) == (b.gb, b.ny - b.gt)
            assert (b.below is None) == (r == 0) and (b.above is None) == (r == world - 1)
            rows.append((b.r0, b.r1))
        assert rows[0][0] == 0 and rows[-1][1] == 37
        assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        rowblock.RowBlock(4, 2, 0, 3)


@pytest.mark.parametrize("world,overlap,variant", [(2, True, 1), (3, True, 1), (3, False, 1), (2, True, 2), (3, True, 2), (3, False, 2)])
def test_row_block_run_equals_single_domain_bitwise(world, overlap, variant, tmp_path):
    """overlap=True: boundary rows are computed and sent first, the interior follows (3 launches per
    sub-iteration); overlap=False: one launch then a blocking exchange.  Both must equal the 1-rank run."""
    ref = run_core(0, 1, variant=variant)  # NSUB = 5: two double passes + one single sub-iteration for variant 2
    assert float(ref.u.abs().max()) > 1e-5
    if variant == 2:
        one = run_core(0, 1, variant=1)  # pass structure does not change the arithmetic
        assert torch.equal(one.u, ref.u) and torch.equal(one.s[0], ref.s[0]) and torch.equal(one.H, ref.H)
    port = free_port()
    mp.spawn(worker, args=(world, port, str(tmp_path), overlap, False, variant), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for key, full in (("H", ref.H), ("A", ref.A), ("s11", ref.s[0])):  # oracle ops keep [nc, ny, nx] planes
        got = torch.cat([p[key] for p in parts], dim=1)
        assert torch.equal(got, full), key
    for key, full in (("u", ref.u), ("v", ref.v)):
        got = torch.cat([p[key] for p in parts], dim=0)
        assert got.shape == full.shape
        assert torch.equal(got, full), key


@pytest.mark.parametrize("world,overlap,group,nsub,variant", [(2, True, 2, 9, 2), (3, True, 2, 9, 2), (2, False, 3, 15, 2), (2, True, 3, 7, 2),
                                                              (2, True, 1, 8, 3), (2, True, 2, 14, 3), (3, False, 1, 7, 3),
                                                              (2, True, 1, 11, 4), (3, False, 1, 10, 4), (2, True, 1, 9, 4)])
def test_grouped_passes_with_deep_ghost_zones_bitwise(world, overlap, group, nsub, variant, tmp_path):
    """latency-avoiding halo: `group` passes of `variant` (2, 3 or 4) sub-iterations between two exchanges on ghost
    zones of depth (variant*group, variant*group - 1); the ghost rows are advanced redundantly.  nsub leaves a
    remainder (a trailing pass of fewer sub-iterations and / or a single sub-iteration) and is not a multiple of the
    group (a shorter last group).  Must equal the 1-rank run."""
    ref = run_core(0, 1, variant=variant, nsub=nsub)
    assert float(ref.u.abs().max()) > 1e-5
    if variant >= 3:
        one = run_core(0, 1, variant=1, nsub=nsub)  # pass structure does not change the arithmetic
        assert torch.equal(one.u, ref.u) and torch.equal(one.s[0], ref.s[0]) and torch.equal(one.H, ref.H)
    port = free_port()
    mp.spawn(worker, args=(world, port, str(tmp_path), overlap, False, variant, group, nsub), nprocs=world, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(world)]
    for key, full in (("H", ref.H), ("A", ref.A), ("s11", ref.s[0])):
        assert torch.equal(torch.cat([p[key] for p in parts], dim=1), full), key
    for key, full in (("u", ref.u), ("v", ref.v)):
        assert torch.equal(torch.cat([p[key] for p in parts], dim=0), full), key


def test_coupled_thermodynamics_dynamics_row_blocks(tmp_path):
    """BASELINE config 5 in miniature: column physics + dynamics per step, 2 ranks == 1 rank bitwise"""
    ref = run_core(0, 1, coupled=True, variant=2)
    assert float((ref.A[0] - 1.0).abs().max()) > 1e-3  # the thermodynamics changed the concentration
    port = free_port()
    mp.spawn(worker, args=(2, port, str(tmp_path), True, True, 2), nprocs=2, join=True)
    parts = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r)) for r in range(2)]
    for key, full in (("H", ref.H), ("A", ref.A), ("s11", ref.s[0])):
        assert torch.equal(torch.cat([p[key] for p in parts], dim=1), full), key
    assert torch.equal(torch.cat([p["u"] for p in parts], dim=0), ref.u)
    assert torch.equal(torch.cat([p["tice0"] for p in parts], dim=0), ref.col["tice0"])
