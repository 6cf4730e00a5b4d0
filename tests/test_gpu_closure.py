"""The closure that keeps the dynamics inside the physical range (include/nsdg.h "INPUT DOMAIN AND CLOSURE", DESIGN.md section
3.3), through the C ABI against oracle/dyn_oracle.c and by its own properties:

  * ridging cap + Zhang-Shu scaling limiter at the end of a transport step (nsdg_transport_bounds_set / nsdg_transport_limit,
    the epilogue of the marching launch);
  * free drift at ice-free nodes (nsdg_mevp_params.min_conc / min_thick), a property of the packed nodal coefficients.

PARITY UNPINNED: the reference snapshot has no dynamics (/root/reference/CMakeLists.txt:43-46) and no cap on the concentration
either (physics/src/modules/HiblerConcentration.cpp:32-47); the oracle is this repository's own restatement, held on the CPU to
the independent numpy restatement tests/dyn_independent.py (tests/golden/dyn_independent_v3.npz), which the HIP path meets below.
"""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
from nextsimdg_amd import abi, basis, rowblock, synthetic

pytestmark = pytest.mark.gpu

INF = float("inf")


@pytest.fixture()
def ctx(gpu):
    from nextsimdg_amd import build

    build.build_lib(verbose=False)
    c = abi.Context(gpu)
    yield c
    c.close()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.cpu().numpy()


def rough_field(rng, nc, ny, nx, mean, spread, slopes):
    f = np.zeros((nc, ny, nx))
    f[0] = mean + spread * rng.standard_normal((ny, nx))
    f[1:] = slopes * rng.standard_normal((nc - 1, ny, nx))
    return f


def point_values(f):
    """values of a DG1 / DG2 field at the scheme's quadrature points: (p+1)^2 volume Gauss points and p+1 Gauss points per edge"""
    nc = f.shape[0]
    g = {3: [-0.5 / np.sqrt(3.0), 0.5 / np.sqrt(3.0)], 6: [-0.5 * np.sqrt(0.6), 0.0, 0.5 * np.sqrt(0.6)]}[nc]
    pts = [(x, y) for y in g for x in g] + [(0.5, s) for s in g] + [(-0.5, s) for s in g] + [(s, 0.5) for s in g] + [(s, -0.5) for s in g]
    pts += [(x, y) for y in (-0.5, 0.5) for x in (-0.5, 0.5)]  # ... and the corners
    psi = lambda x, y: (1.0, x, y, x * x - 1.0 / 12.0, y * y - 1.0 / 12.0, x * y)[:nc]
    P = np.array([psi(x, y) for (x, y) in pts])  # [npts, nc]
    return np.einsum("pc,cyx->pyx", P, f)


@pytest.mark.parametrize("order", [0, 1, 2])
def test_limit_matches_oracle_and_keeps_its_promises(ctx, order):
    """nsdg_transport_limit against oracle_transport_limit on rough fields (a third of the elements out of range), for a field bounded
    below (the thickness) and one bounded on both sides with a capped mean (the concentration), on a ragged grid and on a row range"""
    nx, ny = 70, 37
    nc = basis.NCOEF[order]
    rng = np.random.default_rng(90 + order)
    ctx.set_grid(nx, ny, 1.0, 1.3)
    H = rough_field(rng, nc, ny, nx, 0.15, 0.1, 0.08)
    A = rough_field(rng, nc, ny, nx, 0.93, 0.06, 0.05)
    bounds = ((0.0, INF, False), (0.0, 1.0, True))
    ctx.set_transport_bounds(bounds)
    dH, dA = dev(H), dev(A)
    ctx.transport_limit(order, 0, ny, [dH, dA])
    oH, oA = H.copy(), A.copy()
    O.transport_limit(nx, ny, order, oH, 0.0, INF, False)
    O.transport_limit(nx, ny, order, oA, 0.0, 1.0, True)
    for got, want, name in ((host(dH), oH, "H"), (host(dA), oA, "A")):
        assert np.max(np.abs(got - want)) <= 1e-13, (name, order, np.max(np.abs(got - want)))
    # promises: the limiter never touches a cell mean; the cap only lowers means above 1; the point values are in range
    assert np.array_equal(host(dH)[0], H[0])
    assert np.array_equal(host(dA)[0], np.minimum(A[0], 1.0)) and int((A[0] > 1.0).sum()) > 0
    if order > 0:
        vH, vA = point_values(host(dH)), point_values(host(dA))
        pos = H[0] >= 0  # an element whose MEAN is negative is flattened, not lifted: the mean is the conserved quantity
        assert vH[:, pos].min() >= -1e-15 and np.all(host(dH)[1:][:, ~pos] == 0.0) and int((~pos).sum()) > 0
        assert vA.min() >= -1e-15 and vA.max() <= 1.0 + 1e-15
        changed = lambda a, b: int((np.abs(a[1:] - b[1:]).max(axis=0) > 0).sum())
        assert nx * ny // 10 < changed(host(dH), H) < nx * ny and nx * ny // 10 < changed(host(dA), A) < nx * ny
        # applied again it changes nothing beyond round-off (the binding point sits ON the bound)
        again = [dH.clone(), dA.clone()]
        ctx.transport_limit(order, 0, ny, again)
        assert float((again[0] - dH).abs().max()) <= 1e-15 and float((again[1] - dA).abs().max()) <= 1e-15
    # a row range touches its rows only
    part = [dev(H), dev(A)]
    ctx.transport_limit(order, 5, 9, part)
    for p, full, orig in zip(part, (dH, dA), (H, A)):
        assert torch.equal(p[:, 5:9], full[:, 5:9]) and torch.equal(p[:, :5], dev(orig)[:, :5]) and torch.equal(p[:, 9:], dev(orig)[:, 9:])
    # errors: no bounds, a field count that differs from the bounds'
    with pytest.raises(abi.NsdgError, match="different number of fields"):
        ctx.transport_limit(order, 0, ny, [dH])
    ctx.set_transport_bounds(())
    with pytest.raises(abi.NsdgError, match="no bounds set"):
        ctx.transport_limit(order, 0, ny, [dH, dA])
    with pytest.raises(abi.NsdgError, match="lo <= hi"):
        ctx.set_transport_bounds(((1.0, 0.0, False),))
    with pytest.raises(abi.NsdgError, match="finite upper bound"):
        ctx.set_transport_bounds(((0.0, INF, True),))


@pytest.mark.parametrize("order", [0, 1, 2])
def test_step_entry_points_apply_the_closure_identically(ctx, order):
    """with bounds set, the marching launch (closure in its epilogue), the staged step (closure as a pass of its own) and a step
    composed by hand from stage calls + nsdg_transport_limit agree BIT FOR BIT, over several steps and on a row range; and the
    result is the oracle's step followed by the oracle's limiter"""
    nc = basis.NCOEF[order]
    for (nx, ny) in ((70, 37), (130, 9), (5, 3)):
        rng = np.random.default_rng(300 + order + nx)
        ctx.set_grid(nx, ny, 1.0 / nx, 1.3 / ny)
        u, v = 0.3 * rng.standard_normal((2 * ny + 1, 2 * nx + 1)), 0.3 * rng.standard_normal((2 * ny + 1, 2 * nx + 1))
        z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
        adv = (z(nc, ny, nx), z(nc, ny, nx), z(order + 1, ny, nx + 1), z(order + 1, ny + 1, nx))
        ctx.prepare_advection(order, dev(u), dev(v), *adv)
        adv_o = O.prepare_advection(nx, ny, order, u, v)
        H = rough_field(rng, nc, ny, nx, 0.1, 0.05, 0.06)
        A = rough_field(rng, nc, ny, nx, 0.97, 0.03, 0.04)
        bounds = ((0.0, INF, False), (0.0, 1.0, True))
        ctx.set_transport_bounds(bounds)
        dt = 0.02 / max(nx, ny)
        staged = [dev(H), dev(A)]
        a, b = [dev(H), dev(A)], [z(nc, ny, nx), z(nc, ny, nx)]
        oH, oA = H.copy(), A.copy()
        scratch = z(4 * nc * nx * ny)
        limited = 0
        for step in range(3):
            ctx.transport_step(order, dt, staged, adv, scratch)
            ctx.transport_step_oop(order, dt, a, b, adv)
            a, b = b, a
            for k in range(2):
                assert torch.equal(a[k], staged[k]), (order, nx, ny, step, k, float((a[k] - staged[k]).abs().max()))
            for f, (lo, hi, cap) in zip((oH, oA), bounds):
                before = None
                O.transport_step(nx, ny, 1.0 / nx, 1.3 / ny, order, dt, f, adv_o)
                before = f.copy()
                O.transport_limit(nx, ny, order, f, lo, hi, cap)
                limited += int((np.abs(f - before).max(axis=0) > 0).sum())
        assert limited > 0  # the closure was active in these steps
        for got, want in ((host(a[0]), oH), (host(a[1]), oA)):
            assert np.max(np.abs(got - want)) <= 1e-12, (order, nx, ny, np.max(np.abs(got - want)))
        if ny >= 9:
            part = [torch.full_like(x, 7.0) for x in a]
            full = [torch.zeros_like(x) for x in a]
            ctx.transport_step_oop(order, dt, a, full, adv)
            ctx.transport_step_oop_rows(order, 2, ny - 3, dt, a, part, adv)
            for k in range(2):
                assert torch.equal(part[k][:, 2:ny - 3], full[k][:, 2:ny - 3])
                assert bool((part[k][:, :2] == 7.0).all()) and bool((part[k][:, ny - 3:] == 7.0).all())
            # a step composed by hand: three (order + 1) stages, then the limiter
            if order == 2:
                t1, t2, out = [torch.zeros_like(x) for x in a], [torch.zeros_like(x) for x in a], [torch.zeros_like(x) for x in a]
                ctx.transport_stage(2, 0, ny, dt, 0.0, 1.0, a, a, t1, adv)
                ctx.transport_stage(2, 0, ny, dt, 0.75, 0.25, a, t1, t2, adv)
                ctx.transport_stage(2, 0, ny, dt, 1.0 / 3.0, 2.0 / 3.0, a, t2, out, adv)
                ctx.transport_limit(2, 0, ny, out)
                for k in range(2):
                    assert torch.equal(out[k], full[k])
        # a call that advances another number of fields than the bounds describe is refused, not silently unlimited -- and refused
        # BEFORE it advances anything (the staged step too, whose closure is a pass of its own at the end)
        with pytest.raises(abi.NsdgError, match="different number of fields"):
            ctx.transport_step_oop(order, dt, [a[0]], [b[0]], adv)
        keep = a[0].clone()
        with pytest.raises(abi.NsdgError, match="different number of fields"):
            ctx.transport_step(order, dt, [a[0]], adv, scratch)
        assert torch.equal(a[0], keep)
        ctx.set_transport_bounds(())


def thin_box(ctx, nx, ny, rng, **pk):
    """box test with patches of (almost) no ice: ice-free nodes, nodes below the mass floor, and ordinary ones"""
    bt = synthetic.BoxTest(nx, ny)
    po = O.mevp_params(**pk)
    ctx.set_mevp_params(ctx.mevp_default_params(**pk))
    ctx.set_grid(nx, ny, bt.hx, bt.hy)
    H, A = bt.dg_fields()
    A[0] -= 0.2 * rng.random((ny, nx))
    H[1:] += 0.01 * rng.standard_normal(H[1:].shape)
    thin = rng.random((ny, nx)) < 0.15
    H[0][thin] = 0.004 * rng.random(int(thin.sum()))  # true thickness below a centimetre
    H[1:, thin] *= 0.01
    open_water = rng.random((ny, nx)) < 0.05
    H[:, open_water] = 0.0
    A[:, open_water] = 0.0
    return bt, po, H, A


@pytest.mark.parametrize("variant", [0, 1, abi.DEFAULT_MEVP_VARIANT])
def test_ice_free_nodes_drift_freely_and_match_the_oracle(ctx, variant):
    """the ice-free-node rule through the packed coefficients: sub-iterations of every kind of kernel against the oracle's, on a
    box with thin and open patches; flagged nodes ignore the stress around them (their update does not change when the stress
    does), unflagged ones do not; with the rule switched off (min_conc = min_thick = 0) the bare scheme of rounds 1-4 returns"""
    ctx.set_mevp_variant(variant)
    nx, ny = 67, 21
    rng = np.random.default_rng(17)
    bt, po, H, A = thin_box(ctx, nx, ny, rng)
    shape = (2 * ny + 1, 2 * nx + 1)
    u, v = 0.05 * rng.standard_normal(shape), 0.05 * rng.standard_normal(shape)
    for a in (u, v):
        a[0] = a[-1] = 0
        a[:, 0] = a[:, -1] = 0
    s = [1e3 * rng.standard_normal((8, ny, nx)) for _ in range(3)]
    pg = O.ice_strength(nx, ny, po, H, A)
    cgh, cga = O.dg_to_cg(nx, ny, H), O.dg_to_cg(nx, ny, A)
    uo, vo = [np.ascontiguousarray(a) for a in bt.ocean()]
    tax, tay = O.wind_stress(po, *[np.ascontiguousarray(a) for a in bt.wind(0.0)])
    free = (cga < po.min_conc) | (cgh < po.min_thick * cga) | (cgh <= po.h_min)
    inner = np.zeros(shape, bool)
    inner[1:-1, 1:-1] = True
    assert 50 < int((free & inner).sum()) < inner.sum() // 2
    nsub = 4
    packed = torch.zeros(8 * u.size, dtype=torch.float64, device="cuda")
    # the fused preparation packs the same coefficients as the separate calls (nodal means and wind stress from the device, so that
    # both see the same operands), flags included: the flagged nodes carry coefficients scaled by 2^100
    dcgh, dcga, dtx, dty = (torch.zeros(shape, dtype=torch.float64, device="cuda") for _ in range(4))
    ua, va = [np.ascontiguousarray(a) for a in bt.wind(0.0)]
    ctx.dg_to_cg(dev(H), dcgh)
    ctx.dg_to_cg(dev(A), dcga)
    ctx.wind_stress(dev(ua), dev(va), dtx, dty)
    ctx.mevp_pack_nodal(120.0, (dev(0.9 * u), dev(0.9 * v)), (dtx, dty), (dev(uo), dev(vo)), dcgh, dcga, packed)
    p2 = torch.zeros_like(packed)
    ctx.mevp_prepare(120.0, dev(H), dev(A), (dev(ua), dev(va)), (dev(uo), dev(vo)), (dev(0.9 * u), dev(0.9 * v)), p2)
    assert torch.equal(packed, p2)
    hp = packed[:2 * u.size].view(shape[0], shape[1], 2)[:, :, 0].cpu().numpy()  # h' of every node: first entry of pair plane 0
    assert np.array_equal(hp > 1e20, free) and np.allclose(hp[free] * 2.0 ** -100, np.maximum(cgh[free], po.h_min), rtol=1e-13)
    ctx.mevp_pack_nodal(120.0, (dev(0.9 * u), dev(0.9 * v)), (dev(tax), dev(tay)), (dev(uo), dev(vo)), dev(cgh), dev(cga), packed)

    def run(stress):
        ds, du, dv = [abi.tile(dev(x)) for x in stress], dev(u), dev(v)
        scratch = torch.zeros(10 * u.size + 3 * ds[0].numel(), dtype=torch.float64, device="cuda")
        ctx.mevp_subcycle(120.0, nsub, ds, du, dv, dev(0.9 * u), dev(0.9 * v), dev(tax), dev(tay), dev(uo), dev(vo), dev(cgh), dev(cga),
                          abi.tile(dev(pg)), scratch)
        return host(du), host(dv), [abi.untile(x, nx).cpu().numpy() for x in ds]

    gu, gv, gs = run(s)
    so, ou, ov = [x.copy() for x in s], u.copy(), v.copy()
    O.mevp_subcycle(nx, ny, bt.hx, bt.hy, 120.0, nsub, po, so, ou, ov, 0.9 * u, 0.9 * v, tax, tay, uo, vo, cgh, cga, pg)
    scale = max(np.max(np.abs(ou)), np.max(np.abs(ov)))
    assert np.all(np.isfinite(gu)) and np.max(np.abs(gu - ou)) <= 1e-10 * scale and np.max(np.abs(gv - ov)) <= 1e-10 * scale
    for a, b in zip(gs, so):
        assert np.max(np.abs(a - b)) <= 1e-10 * np.max(np.abs(b))
    # ONE sub-iteration with a different stress: flagged nodes do not notice, the others do
    nsub = 1
    one_u, one_v, _ = run(s)
    other_u, other_v, _ = run([3.0 * x for x in s])
    moved = (np.abs(one_u - other_u) + np.abs(one_v - other_v)) > 0
    assert not moved[free & inner].any() and moved[~free & inner].mean() > 0.99
    # rule off: the bare scheme, in which those nodes feel the stress over their floor mass
    ctx.set_mevp_params(ctx.mevp_default_params(min_conc=0.0, min_thick=0.0))
    ctx.mevp_pack_nodal(120.0, (dev(0.9 * u), dev(0.9 * v)), (dev(tax), dev(tay)), (dev(uo), dev(vo)), dev(cgh), dev(cga), packed)
    bare_u, _, _ = run(s)
    pb = O.mevp_params(min_conc=0.0, min_thick=0.0)
    so, ou, ov = [x.copy() for x in s], u.copy(), v.copy()
    O.mevp_subcycle(nx, ny, bt.hx, bt.hy, 120.0, 1, pb, so, ou, ov, 0.9 * u, 0.9 * v, tax, tay, uo, vo, cgh, cga, pg)
    assert np.max(np.abs(bare_u - ou)) <= 1e-10 * np.max(np.abs(ou))
    assert np.max(np.abs(bare_u - one_u)[free & inner]) > 1e-2  # what the rule prevents: floor-mass nodes pushed by the stress around them
    assert np.array_equal(bare_u[~free], one_u[~free])  # ... and nothing else changes
    ctx.set_mevp_variant(abi.DEFAULT_MEVP_VARIANT)


def test_hip_closure_matches_the_independent_restatement(ctx):
    """the closure outputs of tests/golden/dyn_independent_v3.npz (cap + limiter on the case's H and A, and the velocity of ONE
    sub-iteration with its ice-free nodes) from the HIP path"""
    import dyn_independent as D

    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dyn_independent_v3.npz"))
    c, nx, ny = D.CASE, D.CASE["nx"], D.CASE["ny"]
    ctx.set_grid(nx, ny, c["hx"], c["hy"])
    ctx.set_transport_bounds(abi.H_A_BOUNDS)
    f = [dev(fix["in_H"]), dev(fix["in_A"])]
    ctx.transport_limit(2, 0, ny, f)
    for got, name in zip(f, ("H_limited", "A_limited")):
        assert np.max(np.abs(host(got) - fix["out_" + name])) <= 1e-13, name
    ctx.set_transport_bounds(())


@pytest.mark.parametrize("world", [1, 3])
def test_closure_is_active_and_decomposition_independent(gpu, world):
    """a coupled run on rough fields (open water, thin ice, a cover that closes) in 1 and in 3 row blocks with the native drivers:
    bit-identical, and the closure did something -- the same run without it leaves [0, 1]"""
    from thread_ranks import gather, run_world

    nx, ny, nsub, nsteps = 150, 96, 12, 3
    rng = np.random.default_rng(5)
    bt = synthetic.BoxTest(nx, ny)
    H, A = bt.dg_fields()
    A[0] = 1.0 - 0.05 * rng.random((ny, nx))
    A[1:3] = 0.03 * rng.standard_normal((2, ny, nx))
    H[1:3] += 0.05 * rng.standard_normal((2, ny, nx))
    hole = rng.random((ny, nx)) < 0.1
    H[:, hole] = 0.0
    A[:, hole] = 0.0
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    data = (bt, H, A, uo, vo, 3.0 * ua, 3.0 * va)
    kw = dict(data=data, alpha=300.0, keep=("H", "A", "u", "v"))
    ref = run_world(1, abi.DEFAULT_MEVP_VARIANT, False, nx, ny, nsub, nsteps, native=True, **kw)[0]
    vA, vH = point_values(host(ref["A"])), point_values(host(ref["H"]))
    assert vA.min() >= -1e-15 and vA.max() <= 1.0 + 1e-15 and float(ref["A"][0].max()) <= 1.0
    assert vH[:, host(ref["H"])[0] >= 0].min() >= -1e-15 and bool(torch.isfinite(ref["u"]).all())
    bare = run_world(1, abi.DEFAULT_MEVP_VARIANT, False, nx, ny, nsub, nsteps, native=True, core_kw=dict(closure=False), **kw)[0]
    bA = point_values(host(bare["A"]))
    assert bA.max() > 1.0 + 1e-3 and bA.min() < -1e-3  # what the bare transport does to these fields
    if world > 1:
        parts = run_world(world, abi.DEFAULT_MEVP_VARIANT, False, nx, ny, nsub, nsteps, group=2, transport="native", native=True, **kw)
        for k in ("H", "A", "u", "v"):
            assert torch.equal(gather(parts, world, k), ref[k]), k


def test_compressible_cover_1024_stays_in_range_for_37_hours(gpu):
    """The run that left the physical range in rounds 3-5 (profiles/r04_soak_divergence_cause.md, profiles/r05_closure.md): a uniform cover
    A0 = 0.9, H0 = 0.3 on 1024 x 1024, winter forcing, dynamics + column thermodynamics, dt = 120 s, 120 sub-iterations -- with alpha
    = beta from the stability bound at Delta_min = 2e-9 (14 438) it fails between step 1000 and 1100 with or without the closure.
    With the hosts' sub-cycle parameters (alpha = beta = 1500 and the regularisation this mesh needs for it) and the closure it
    must pass that point inside the physical range: 1100 steps = 36.7 model hours (the full 1600 steps: tools/soak_coupled.py)."""
    nx = ny = 1024
    L, dt, nsub, steps = 512e3, 120.0, 120, 1100
    c = abi.Context(gpu)
    bt = synthetic.BoxTest(nx, ny, L)
    sub = bt.subcycle_parameters(dt, mode="keep_alpha")  # round 5's policy; the adaptive form of round 6: test_compressible_cover_1024_adaptive below
    assert sub["alpha"] == 1500.0 and sub["aevp_c"] == 0.0 and 1.5e-7 < sub["delta_min"] < 2.5e-7 and abs(bt.stable_alpha(dt, delta_min=sub["delta_min"]) - 1500.0) < 1e-6
    c.set_mevp_params(c.mevp_default_params(**sub))
    core = rowblock.CoupledCore(c, rowblock.RowBlock(nx, ny, 0, 1), L / nx, L / ny, dt, nsub, gpu, native=True, forcing="winter")
    cs, cf = synthetic.column_fields_smooth(nx, ny, L)
    cs = {"hsnow": np.full((ny, nx), 0.05), "tice0": np.full((ny, nx), -8.0)}
    cf["sst"], cf["sss"] = np.full((ny, nx), -1.76), np.full((ny, nx), 32.0)
    core.load_column({**cs, **cf})
    H, A = np.zeros((6, ny, nx)), np.zeros((6, ny, nx))
    H[0], A[0] = 0.3, 0.9
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    core.load_global(H, A, uo, vo, ua, va)
    for step in range(steps):
        core.device_wind(L, step * dt)
        core.step()
    torch.cuda.synchronize()
    for f in (core.u, core.v, core.H, core.A):
        assert bool(torch.isfinite(f).all())
    assert 0.01 < float(core.u.abs().max()) < 0.2
    assert 0.2 < float(core.H[0].min()) and float(core.H[0].max()) < 0.45
    assert 0.7 < float(core.A[0].min()) and float(core.A[0].max()) <= 1.0
    vA = point_values(host(core.A[:, ::8, ::8]))
    assert vA.min() >= -1e-15 and vA.max() <= 1.0 + 1e-15
    core.close()
    c.close()


def test_a_transport_plan_with_its_own_bounds_neither_reads_nor_changes_the_contexts(gpu):
    """nsdg_rb_transport_desc.own_bounds (round 6; advisor of round 5: the closure's bounds were sticky state of the shared context, so a
    later step of two OTHER fields on that context was silently limited to H's and A's ranges): a plan that carries its own bounds applies
    them whatever nsdg_transport_bounds_set says -- and a plain step call on the same context afterwards sees the context's bounds only"""
    from nextsimdg_amd import rowblock, synthetic

    nx, ny, dt = 96, 64, 120.0
    c = abi.Context(gpu)
    bt = synthetic.BoxTest(nx, ny)
    c.set_grid(nx, ny, bt.hx, bt.hy)
    rng = np.random.default_rng(77)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    X, Y = np.meshgrid(np.arange(2 * nx + 1), np.arange(2 * ny + 1))
    u = 0.3 * np.sin(0.11 * X) * np.cos(0.07 * Y)
    v = 0.3 * np.cos(0.05 * X) * np.sin(0.13 * Y)
    for a in (u, v):
        a[0] = a[-1] = 0
        a[:, 0] = a[:, -1] = 0
    adv = tuple(torch.zeros(*s, dtype=torch.float64, device="cuda") for s in ((6, ny, nx), (6, ny, nx), (3, ny, nx + 1), (3, ny + 1, nx)))
    c.prepare_advection(2, dev(u), dev(v), *adv)
    F = np.zeros((6, ny, nx))
    F[0] = 0.95 + 0.2 * rng.random((ny, nx))  # cell means above 1: the cap and the limiter have work to do
    F[1:3] = 0.1 * rng.standard_normal((2, ny, nx))
    G = F.copy()
    blk = rowblock.RowBlock(nx, ny, 0, 1)
    z = lambda: torch.zeros(6, ny, nx, dtype=torch.float64, device="cuda")

    def run(plan_bounds, ctx_bounds):
        c.set_transport_bounds(ctx_bounds)
        phi, t1, t2 = [dev(F), dev(G)], [z(), z()], [z(), z()]
        tr = c.rb_transport(blk, (None, None), phi, t1, t2, adv, bounds=plan_bounds)
        assert tr(dt, 0) == 1
        torch.cuda.synchronize()
        tr.close()
        return [x.clone() for x in t1]

    HA = abi.H_A_BOUNDS
    own = run(HA, ())  # the plan's own bounds, nothing on the context
    ctxb = run(None, HA)  # no own bounds: the context's (ABI 5 behaviour)
    assert all(torch.equal(a, b) for a, b in zip(own, ctxb))
    other = ((0.0, float("inf"), False), (0.0, float("inf"), False))
    shielded = run(HA, other)  # the context says something else: the plan does not care ...
    assert all(torch.equal(a, b) for a, b in zip(own, shielded))
    assert c.transport_bounds == other  # ... and leaves it alone
    bare = run((), HA)  # a plan that states NO closure is not limited by the context's bounds either
    assert float(bare[1][0].max()) > 1.0 and float(own[1][0].max()) <= 1.0 and not torch.equal(bare[1], own[1])
    with pytest.raises(abi.NsdgError, match="own bounds"):
        c.rb_transport(blk, (None, None), [dev(F), dev(G)], [z(), z()], [z(), z()], adv, bounds=(HA[0],))
    c.set_transport_bounds(())
    c.close()
