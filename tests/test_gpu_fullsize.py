"""Full-size (BASELINE.json configs) property tests on the GPU.  The CPU oracle cannot run these sizes
in seconds, so correctness is checked through size-independent properties of the discretisation:
mass conservation, boundary conditions, point symmetry, element independence (permutation
equivariance) and agreement between the two mEVP kernel variants."""
import numpy as np
import pytest
import torch

from nextsimdg_amd import abi, basis, rowblock, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu):
    from nextsimdg_amd import build

    build.build_lib(verbose=False)
    c = abi.Context(gpu)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _no_closure_left_on_the_shared_context(ctx):
    """a DynamicsCore states the closure's bounds (H, A) on its context; tests that call the transport entry points with other field
    lists on the SAME context must not inherit them"""
    yield
    ctx.set_transport_bounds(())


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_config2_512_dg1_rotating_patch_mass_and_bounds(ctx):
    """BASELINE config 2: 512x512 DG1 advection-only rotating patch (velocity cut off to zero at the wall)"""
    n, order = 512, 1
    phi, u, v, _ = synthetic.rotating_patch(n, n, order)
    ctx.set_grid(n, n, 1.0 / n, 1.0 / n)
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
    adv = (z(3, n, n), z(3, n, n), z(2, n, n + 1), z(2, n + 1, n))
    ctx.prepare_advection(order, dev(u), dev(v), *adv)
    d = dev(phi)
    scratch = z(2 * phi.size)
    m0 = float(d[0].sum())
    dt = 0.1 / 3 * (1.0 / n) / np.pi
    for _ in range(50):
        ctx.transport_step(order, dt, [d], adv, scratch)
    m1 = float(d[0].sum())
    assert abs(m1 - m0) <= 1e-12 * abs(m0)
    assert float(d[0].min()) > -1e-3 and float(d[0].max()) < 1.0 + 1e-3
    assert float((d - dev(phi)).abs().max()) > 1e-4  # it did move


def test_column_step_4096_is_element_independent(ctx):
    """config 5's thermodynamic coupling size: 4096^2 elements; elements never see their neighbours
    (core/src/DevStep.cpp:17-22), so permuting the inputs permutes the outputs bit for bit"""
    n = 4096 * 4096
    state, forcing, newice = synthetic.column_fields(n)
    ds, df, dn = {k: dev(v) for k, v in state.items()}, {k: dev(v) for k, v in forcing.items()}, dev(newice)
    perm = torch.randperm(n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3))
    ps = {k: v[perm].contiguous() for k, v in ds.items()}
    pf = {k: v[perm].contiguous() for k, v in df.items()}
    pn = dn[perm].contiguous()
    for _ in range(3):
        ctx.column_step(600.0, ds, df, dn)
        ctx.column_step(600.0, ps, pf, pn)
    for k in abi.STATE:
        assert torch.equal(ds[k][perm], ps[k]), k
        assert bool(torch.isfinite(ds[k]).all())
    assert torch.equal(dn[perm], pn)
    c = ds["cice"]
    assert float(c.min()) >= 0.0 and float(c.max()) < 1.5
    assert float((c == 0).double().mean()) > 0.05  # the open-water / vanish branches are exercised


def box_core(ctx, n, nsub, uniform=False):
    bt = synthetic.BoxTest(n, n)
    alpha = bt.stable_alpha(120.0)  # 1.4e4 at 2048^2: with alpha = 1500 the sub-cycle amplifies round-off x150 per iteration
    ctx.set_mevp_params(ctx.mevp_default_params(alpha=alpha, beta=alpha))
    core = rowblock.DynamicsCore(ctx, rowblock.RowBlock(n, n), bt.hx, bt.hy, 120.0, nsub, torch.device("cuda"))
    H, A = bt.dg_fields()
    if uniform:
        H[:] = 0
        H[0] = 0.3
        A[:] = 0
        A[0] = 0.9
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    core.load_global(H, A, uo, vo, ua, va)
    return core


def test_config4_2048_dynamics_step_properties(ctx):
    """2048x2048 DG2 transport + mEVP: boundary conditions, mass conservation of H and A in the closed
    box, finite fields, agreement of the fused and the two-kernel mEVP variants at full size"""
    n = 2048
    results = {}
    for variant in (3, 2, 1, 0):
        ctx.set_mevp_variant(variant)
        core = box_core(ctx, n, nsub=12)
        mH, mA = float(core.H[0].sum()), float(core.A[0].sum())
        core.step()
        for f in (core.u, core.v):
            assert bool(torch.isfinite(f).all())
            assert float(f[0].abs().max()) == 0 and float(f[-1].abs().max()) == 0
            assert float(f[:, 0].abs().max()) == 0 and float(f[:, -1].abs().max()) == 0
        assert float(core.u.abs().max()) > 1e-6
        assert abs(float(core.H[0].sum()) - mH) <= 1e-12 * abs(mH)
        # the area is conserved up to what the ridging cap (closure, on by default) turned into thickness: a loss, never a gain
        assert -1e-6 * abs(mA) <= float(core.A[0].sum()) - mA <= 1e-12 * abs(mA) and float(core.A[0].max()) <= 1.0
        results[variant] = (core.u.clone(), core.v.clone(), core.s[0].clone(), core.H.clone())
        del core
    ctx.set_mevp_variant(abi.DEFAULT_MEVP_VARIANT)
    ctx.set_mevp_params(ctx.mevp_default_params())
    ctx.set_transport_bounds(())  # the cores set the closure's bounds on the shared context
    for a, b in zip(results[0], results[1]):
        scale = float(a.abs().max())
        assert float((a - b).abs().max()) <= 1e-10 * scale
    for other in (2, 3):  # one, two and three sub-iterations per pass: bit-identical velocities and thickness
        for a, b in zip(results[1][:2] + results[1][3:], results[other][:2] + results[other][3:]):
            assert torch.equal(a, b), other


def test_config3_1024_point_symmetry(ctx):
    """1024x1024: with a uniform ice cover the box set-up is invariant under the point reflection about
    the centre of the box (cyclone centred, circular current): u(x) = -u(L - x) to round-off"""
    n = 1024
    ctx.set_mevp_variant(2)
    core = box_core(ctx, n, nsub=20, uniform=True)
    core._set_grid()
    core.momentum()
    for f in (core.u, core.v):
        scale = float(f.abs().max())
        assert scale > 1e-6
        assert float((f + torch.flip(f, (0, 1))).abs().max()) <= 1e-11 * scale
    ctx.set_mevp_params(ctx.mevp_default_params())


def test_config2_convergence_rate_one_revolution(ctx):
    """BASELINE config 2 / SURVEY.md section 8(d): DG1 rotating patch, ONE full revolution, L2 error against the
    initial field and its convergence rate over 128^2 / 256^2 / 512^2 (expected ~2 for DG1)"""
    order = 1
    errs = []
    for n in (128, 256, 512):
        phi, u, v, phi0 = synthetic.rotating_patch(n, n, order, kind="narrow")
        ctx.set_grid(n, n, 1.0 / n, 1.0 / n)
        z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
        adv = (z(3, n, n), z(3, n, n), z(2, n, n + 1), z(2, n + 1, n))
        ctx.prepare_advection(order, dev(u), dev(v), *adv)
        d = dev(phi)
        scratch = z(2 * phi.size)
        steps = int(np.ceil(1.0 / (0.15 / 3 * (1.0 / n) / np.pi)))
        dt = 1.0 / steps
        m0 = float(d[0].sum())
        for _ in range(steps):
            ctx.transport_step(order, dt, [d], adv, scratch)
        assert abs(float(d[0].sum()) - m0) <= 1e-11 * abs(m0)
        errs.append(basis.l2_error(d.cpu().numpy(), phi0, 1.0, 1.0))
    rates = [np.log2(errs[i] / errs[i + 1]) for i in range(2)]
    assert errs[2] < errs[1] < errs[0]
    assert rates[1] > 1.7, (errs, rates)


def test_config5_coupled_run_stays_physical(ctx):
    """BASELINE config 5 in small: column thermodynamics + dynamics for 240 model steps (8 hours) with the smooth
    winter forcing of synthetic.column_fields_smooth.  The fields must stay finite and physical: drift of a few
    cm/s, thickness and concentration inside their initial ranges plus slow thermodynamic growth.  (With the
    per-element random forcing of the column-kernel tests, or with a mixed layer above freezing -- which the
    reference never cools, SURVEY App. A.7 quirk 4 -- the same model melts / blows up within ~10 steps.)"""
    nx = ny = 256
    L, dt = 512e3, 120.0
    bt = synthetic.BoxTest(nx, ny, L)
    alpha = bt.stable_alpha(dt)
    ctx.set_mevp_variant(abi.DEFAULT_MEVP_VARIANT)
    ctx.set_mevp_params(ctx.mevp_default_params(alpha=alpha, beta=alpha))
    core = rowblock.CoupledCore(ctx, rowblock.RowBlock(nx, ny, 0, 1), L / nx, L / ny, dt, 120, torch.device("cuda"))
    cs, cf = synthetic.column_fields_smooth(nx, ny, L)
    core.load_column({**cs, **cf})
    H, A = bt.dg_fields()
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    core.load_global(H, A, uo, vo, ua, va)
    for _ in range(240):
        core.step()
    for f in (core.u, core.v, core.H, core.A, core.col["tice0"], core.col["hsnow"]):
        assert bool(torch.isfinite(f).all())
    assert 1e-3 < float(core.u.abs().max()) < 0.5
    assert 0.25 < float(core.H[0].min()) and float(core.H[0].max()) < 0.45
    assert 0.9 < float(core.A[0].min()) and float(core.A[0].max()) < 1.01  # convergent drift piles concentration up slightly above 1
    assert -40.0 < float(core.col["tice0"].min()) and float(core.col["tice0"].max()) <= 0.0
    ctx.set_mevp_params(ctx.mevp_default_params())
    ctx.set_transport_bounds(())
