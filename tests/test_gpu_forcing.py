"""Device-side forcing providers (csrc/forcing.hip) against their numpy restatement
(tests/forcing_ref.py) and the host-side generators of nextsimdg_amd/synthetic.py.  Tolerance: 1e-13 relative --
device sin / cos / exp against numpy's."""
import numpy as np
import pytest
import torch

import forcing_ref as R
from nextsimdg_amd import abi, rowblock, synthetic
from thread_ranks import gather, run_world

pytestmark = pytest.mark.gpu

PLANES = ("tair", "tdew", "slp", "qsw", "qlw", "mld", "snowfall")


@pytest.fixture()
def ctx(gpu):
    c = abi.Context(gpu)
    yield c
    c.close()


def close(a, b, rtol=1e-13, atol=1e-13):
    a, b = np.asarray(a), np.asarray(b)
    return np.all(np.abs(a - b) <= atol + rtol * np.abs(b))


def test_dummy_forcing_is_the_reference_constants(ctx):
    nx, ny = 70, 37
    ctx.set_grid(nx, ny, 1.0, 1.0)
    f = {k: torch.full((ny, nx), 7.0, dtype=torch.float64, device="cuda") for k in PLANES}
    ctx.column_forcing("dummy", 1234.0, f)
    ref = R.column_forcing("dummy", nx, ny, 1234.0)
    for k in PLANES:
        assert np.array_equal(f[k].cpu().numpy(), ref[k]), k  # DummyExternalData.hpp:22-34, exactly


@pytest.mark.parametrize("t", [0.0, 6 * 3600.0, 3.7 * 86400.0])
def test_winter_forcing_matches_numpy_and_row_blocks_are_bitwise_consistent(ctx, t):
    nx, ny = 150, 96
    ctx.set_grid(nx, ny, 1.0, 1.0)
    ctx.set_block(0, 0)
    f = {k: torch.zeros(ny, nx, dtype=torch.float64, device="cuda") for k in PLANES}
    ctx.column_forcing("winter", t, f)
    ref = R.column_forcing("winter", nx, ny, t)
    for k in PLANES:
        assert close(f[k].cpu().numpy(), ref[k]), k
    assert float(f["qsw"].min()) >= 0.0 and (t != 6 * 3600.0 or float(f["qsw"].max()) > 50.0)  # noon sun, no negative flux
    # a row block (rows 40..71 of the same domain) evaluates the same expressions: bit-identical
    lo, hi = 40, 72
    ctx.set_grid(nx, hi - lo, 1.0, 1.0)
    ctx.set_block(lo, ny)
    g = {k: torch.zeros(hi - lo, nx, dtype=torch.float64, device="cuda") for k in PLANES}
    ctx.column_forcing("winter", t, g)
    for k in PLANES:
        assert torch.equal(g[k], f[k][lo:hi]), k
    ctx.set_block(0, 0)


def test_boxtest_wind_on_a_row_block_and_column_wind(ctx):
    nx, ny, L, t = 96, 80, 512e3, 7200.0
    bt = synthetic.BoxTest(nx, ny, L)
    ctx.set_grid(nx, ny, bt.hx, bt.hy)
    ctx.set_block(0, 0)
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
    ua, va, uo, vo = (z(2 * ny + 1, 2 * nx + 1) for _ in range(4))
    ctx.boxtest_forcing(L, t, wind=(ua, va), ocean=(uo, vo))
    wa, wo = bt.wind(t), bt.ocean()
    scale = float(np.abs(wa[0]).max())
    for got, want in ((ua, wa[0]), (va, wa[1]), (uo, wo[0]), (vo, wo[1])):
        assert np.abs(got.cpu().numpy() - want).max() <= 1e-12 * scale
    wind = z(ny, nx)
    ctx.column_wind(ua, va, wind)
    assert close(wind.cpu().numpy(), R.column_wind(ua.cpu().numpy(), va.cpu().numpy()), 1e-15, 0.0)
    assert 1.0 < float(wind.max()) < 30.0
    # rows 24..55 as a block of the same domain
    lo, hi = 24, 56
    ctx.set_grid(nx, hi - lo, bt.hx, bt.hy)
    ctx.set_block(lo, ny)
    ub, vb = z(2 * (hi - lo) + 1, 2 * nx + 1), z(2 * (hi - lo) + 1, 2 * nx + 1)
    ctx.boxtest_forcing(L, t, wind=(ub, vb))
    assert torch.equal(ub, ua[2 * lo:2 * hi + 1]) and torch.equal(vb, va[2 * lo:2 * hi + 1])
    ctx.set_block(0, 0)
    # error behaviour: a placement the local array does not fit into
    ctx.set_block(60, ny)
    assert ctx.lib.nsdg_boxtest_forcing(ctx.h, L, t, abi._ptr(ub), abi._ptr(vb), None, None) == -1
    ctx.set_block(0, 0)


def test_coupled_model_with_device_forcing_row_blocks_bitwise(gpu):
    """config 5 in small with the forcing generated on the device every step (winter fields + wind speed from the
    dynamics' wind): 3 row blocks on the native driver == single domain bit for bit; the wind coupling is active"""
    nx, ny, nsub, nsteps = 150, 128, 13, 4
    kw = dict(forcing="winter")
    ref = run_world(1, 3, True, nx, ny, nsub, nsteps, core_kw=kw)[0]
    parts = run_world(3, 3, True, nx, ny, nsub, nsteps, group=2, transport="native", native=True, core_kw=kw)
    for key in ("H", "A", "u", "v", "tice0", "hsnow"):
        assert torch.equal(gather(parts, 3, key), ref[key]), key
    assert bool(torch.isfinite(ref["tice0"]).all())
