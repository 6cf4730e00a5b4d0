"""BASELINE.json configs 3, 4 and 5 AS STATED, on one MI355X, through the C ABI.

The CPU oracle cannot run these sizes in seconds, so the checks are the size-independent ones the discretisation
offers: the kernel variants agree bit for bit (three sub-iterations per pass == three single passes), mass is
conserved in the closed box, the Dirichlet rows stay zero, and a row-block decomposition reproduces the
single-domain run bit for bit.  The multi-rank driver runs with one thread per rank on the one GPU of the test
box (tests/thread_ranks.py): real kernels, real ghost-row layouts, in-process transport instead of RCCL.
"""
import gc

import numpy as np
import pytest
import torch

from nextsimdg_amd import abi, rowblock, synthetic

V = abi.DEFAULT_MEVP_VARIANT  # the library default: four sub-iterations per kernel pass
from thread_ranks import fields, gather, run_world

pytestmark = pytest.mark.gpu


def free():
    gc.collect()
    torch.cuda.empty_cache()


def check_physical(res, mass0, nx, ny):
    u, v, H, A = res["u"], res["v"], res["H"], res["A"]
    for f in (u, v, H, A):
        assert bool(torch.isfinite(f).all())
    for f in (u, v):  # closed box: v = 0 on all four walls
        assert float(f[0].abs().max()) == 0 and float(f[-1].abs().max()) == 0
        assert float(f[:, 0].abs().max()) == 0 and float(f[:, -1].abs().max()) == 0
    assert 1e-6 < float(u.abs().max()) < 1.0
    assert float(A[0].max()) <= 1.0  # the ridging cap of the closure (cell means; on by default in the drivers)
    if mass0 is not None:  # transport alone conserves the cell means in the closed box: the volume (H) exactly; the area (A) up to
        # what the cap turned into thickness where the cover closed -- a loss, never a gain, and small over a few steps
        assert abs(float(H[0].sum()) - mass0[0]) <= 1e-12 * abs(mass0[0])
        assert -1e-6 * abs(mass0[1]) <= float(A[0].sum()) - mass0[1] <= 1e-12 * abs(mass0[1])


SUBCYCLE = ["keep_delta_min", "adaptive"]  # uniform alpha = beta of the stability bound (rounds 1-4) / the hosts' adaptive form (round 6)


def subcycle(bt, mode):
    """what run_world takes as `alpha`: the uniform value, or the parameter set of the adaptive form"""
    return bt.stable_alpha(120.0) if mode == "keep_delta_min" else bt.subcycle_parameters(120.0, mode=mode)


@pytest.mark.parametrize("mode", SUBCYCLE)
def test_config3_1024_transport_and_mevp_120_subiterations(gpu, mode):
    """config 3: 1024x1024 DG2 transport (H, A; SSP-RK3) + mEVP with 120 sub-iterations, 3 model steps, default
    kernel (four sub-iterations per pass, one pipeline stage per wave) against one sub-iteration per pass: bit-identical; mass, walls"""
    n, nsub, nsteps = 1024, 120, 3
    data = fields(n, n, wind_scale=1.0)
    alpha = subcycle(data[0], mode)
    mass0 = (float(np.sum(data[1][0])), float(np.sum(data[2][0])))
    res = {}
    for variant in (V, 1):
        res[variant] = run_world(1, variant, False, n, n, nsub, nsteps, data=data, alpha=alpha)[0]
        check_physical(res[variant], mass0, n, n)
    for k in ("u", "v", "H", "A", "s11"):
        assert torch.equal(res[V][k], res[1][k]), k
    assert float((res[V]["H"][0] - torch.from_numpy(data[1][0]).cuda()).abs().max()) > 1e-9  # the ice did move
    del res
    free()


@pytest.mark.parametrize("mode", SUBCYCLE)
def test_config4_2048_four_row_blocks_equal_single_domain_bitwise(gpu, mode):
    """config 4: 2048x2048 DG2 full dynamics, 4 row blocks of 512 rows, 120 sub-iterations, 3 passes of the default
    (four-iteration) kernel between two ghost-row exchanges (ghost depth 12 / 11: the default of bench.py and of the C++
    host; config 5 below runs 6 passes per exchange), overlap split on, 2 model steps:
    every owned row of every block equals the single-domain run (Python sequence of launches) bit for bit"""
    n, nsub, nsteps, world, group = 2048, 120, 2, 4, (3 if mode == "keep_delta_min" else 2)  # 2 passes per exchange: the hosts' default since round 6
    data = fields(n, n, wind_scale=1.0)
    alpha = subcycle(data[0], mode)
    mass0 = (float(np.sum(data[1][0])), float(np.sum(data[2][0])))
    ref = run_world(1, V, False, n, n, nsub, nsteps, data=data, alpha=alpha)[0]
    check_physical(ref, mass0, n, n)
    # the product path: native row-block drivers (nsdg_rb_*_run) and the exchange behind the C ABI (nsdg_halo_*) on its
    # in-process transport -- everything of the 4-GPU run except RCCL itself
    parts = run_world(world, V, False, n, n, nsub, nsteps, group=group, data=data, alpha=alpha, transport="native", native=True)
    for k in ("H", "A", "u", "v", "s11"):
        got = gather(parts, world, k)
        assert got.shape == ref[k].shape, k
        assert torch.equal(got, ref[k]), k
    del ref, parts
    free()


@pytest.mark.parametrize("mode", SUBCYCLE)
def test_config5_4096_coupled_eight_row_blocks_equal_single_domain_bitwise(gpu, mode):
    """config 5: 4096x4096 DG2 dynamics + column thermodynamics, 8 row blocks of 512 rows, 120 sub-iterations,
    3 model steps with the smooth winter forcing of the coupled bench: bit-identical to the single domain, fields
    in their physical ranges.  (The one-day run of this configuration is tools/soak_coupled.py; its record is in
    profiles/.)"""
    n, nsub, nsteps, world, group = 4096, 120, 3, 8, 6
    bt = synthetic.BoxTest(n, n)
    H, A = bt.dg_fields()
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    data = (bt, H, A, uo, vo, ua, va)
    cs, cf = synthetic.column_fields_smooth(n, n)
    column = {**cs, **cf}
    alpha = subcycle(bt, mode)
    keep = ("H", "A", "u", "v")
    ref = run_world(1, V, True, n, n, nsub, nsteps, data=data, column=column, alpha=alpha, keep=keep)[0]
    check_physical(ref, None, n, n)
    assert 0.25 < float(ref["H"][0].min()) and float(ref["H"][0].max()) < 0.45
    assert 0.9 < float(ref["A"][0].min()) and float(ref["A"][0].max()) < 1.01
    assert -40.0 < float(ref["tice0"].min()) and float(ref["tice0"].max()) <= 0.0
    assert float((ref["tice0"] - torch.from_numpy(column["tice0"]).cuda()).abs().max()) > 1e-3  # the column step ran
    parts = run_world(world, V, True, n, n, nsub, nsteps, group=group, data=data, column=column, alpha=alpha, keep=keep,
                      transport="native", native=True)
    for k in keep + ("hsnow", "tice0"):
        got = gather(parts, world, k)
        assert got.shape == ref[k].shape, k
        assert torch.equal(got, ref[k]), k
    del ref, parts
    free()
