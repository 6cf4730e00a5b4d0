"""The mEVP sub-cycle with LOCAL, SOLUTION-ADAPTIVE alpha and beta (round 6; nsdg_mevp_params.aevp_c > 0; after Kimmritz, Danilov & Losch
2016) through the C ABI: against the oracle and the independent restatement (both hold the same definition: oracle/dyn_oracle.h,
tests/dyn_independent.py), bitwise across the kernel variants and the decompositions, and with the same converged limit as the
uniform form.  Parity unpinned like the rest of the dynamics (the reference snapshot has no dynamics code, CMakeLists.txt:43-46)."""
import os

import numpy as np
import pytest
import torch

import oracle_lib as O
from nextsimdg_amd import abi, synthetic
from test_gpu_parity import Box, assert_close, dev, host, mevp_state, pack, tdev, thost

pytestmark = pytest.mark.gpu
AD = dict(aevp_c=(2.4 * np.pi) ** 2, aevp_alpha_min=50.0)


@pytest.fixture(scope="module")
def ctx(gpu):
    c = abi.Context(gpu)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _defaults_after_each_test(ctx):
    yield
    ctx.set_mevp_variant(abi.DEFAULT_MEVP_VARIANT)
    ctx.set_mevp_strip_rows(0)
    ctx.set_mevp_params(ctx.mevp_default_params())


def box_inputs(b):
    nx, ny = b.nx, b.ny
    pg = O.ice_strength(nx, ny, b.po, b.H, b.A)
    cgh, cga = O.dg_to_cg(nx, ny, b.H), O.dg_to_cg(nx, ny, b.A)
    tax, tay = O.wind_stress(b.po, b.ua, b.va)
    return pg, cgh, cga, tax, tay


def test_stable_params_is_the_one_copy_of_the_rule(ctx):
    """nsdg_mevp_stable_params: the three ways to satisfy the sub-cycle's stability bound (adaptive; keep alpha, raise Delta_min; keep
    Delta_min, raise alpha) and what the hosts print beside them"""
    bt = synthetic.BoxTest(2048, 2048)
    p = abi.stable_mevp_params(ctx.mevp_default_params(), abi.SUBCYCLE_ADAPTIVE, bt.hx, 120.0)
    assert abs(p.aevp_c - (2.4 * np.pi) ** 2) < 1e-12 and p.aevp_alpha_min == 50.0 and p.delta_min == 2e-9
    # the converging form: alpha_min = the bound's alpha at a strain rate of 1.67e-6 1/s on the mesh (1000 at 250 m), never below 50
    for n, want in ((2048, 1000.0), (1024, 500.0), (4096, 2000.0), (64, 50.0)):
        q = abi.stable_mevp_params(ctx.mevp_default_params(), abi.SUBCYCLE_ADAPTIVE_CONVERGED, 512e3 / n, 120.0)
        assert abs(q.aevp_alpha_min / want - 1) < 0.01 and q.aevp_c == p.aevp_c and q.delta_min == 2e-9, (n, q.aevp_alpha_min)
    p = abi.stable_mevp_params(ctx.mevp_default_params(alpha=1500.0), abi.SUBCYCLE_KEEP_ALPHA, bt.hx, 120.0)
    assert p.aevp_c == 0.0 and p.alpha == p.beta == 1500.0 and abs(p.delta_min / bt.stable_delta_min(120.0) - 1) < 1e-12
    assert abs(abi.creep_percent_per_day(p) - p.delta_min * 8.64e6) < 1e-12 and 5.0 < abi.creep_percent_per_day(p) < 8.0  # 7.4e-7 1/s = 6.4 % per day
    p = abi.stable_mevp_params(ctx.mevp_default_params(), abi.SUBCYCLE_KEEP_DELTA_MIN, bt.hx, 120.0)
    assert p.aevp_c == 0.0 and abs(p.alpha / bt.stable_alpha(120.0) - 1) < 1e-12 and p.alpha == p.beta and p.delta_min == 2e-9
    with pytest.raises(abi.NsdgError):
        abi.stable_mevp_params(ctx.mevp_default_params(), 7, bt.hx, 120.0)
    with pytest.raises(abi.NsdgError, match="aevp"):
        ctx.set_mevp_params(ctx.mevp_default_params(aevp_c=-1.0))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
def test_adaptive_subcycle_matches_oracle(ctx, variant):
    """25 sub-iterations with adaptive alpha / beta through nsdg_mevp_subcycle against the oracle's 25, for every kernel variant (variant 0
    runs the marching kernel of variant 1: the two-kernel form has no adaptive path and says so when it is called directly)"""
    ctx.set_mevp_variant(variant)
    b = Box(ctx, 48, 40, **AD)
    nx, ny = b.nx, b.ny
    pg, cgh, cga, tax, tay = box_inputs(b)
    shape = (2 * ny + 1, 2 * nx + 1)
    u, v = np.zeros(shape), np.zeros(shape)
    s = [np.zeros((8, ny, nx)) for _ in range(3)]
    du, dv, ds = dev(u), dev(v), [tdev(x) for x in s]
    scratch = torch.zeros(10 * u.size + 3 * ds[0].numel(), dtype=torch.float64, device="cuda")
    ctx.mevp_subcycle(120.0, 25, ds, du, dv, dev(u), dev(v), dev(tax), dev(tay), dev(b.uo), dev(b.vo), dev(cgh), dev(cga), tdev(pg), scratch)
    O.mevp_subcycle(nx, ny, b.bt.hx, b.bt.hy, 120.0, 25, b.po, s, u, v, u.copy(), v.copy(), tax, tay, b.uo, b.vo, cgh, cga, pg)
    assert np.max(np.abs(u)) > 1e-4 and np.all(np.isfinite(u))
    assert_close(host(du), u, 1e-9, 1e-10 * np.max(np.abs(u)), "u after the adaptive subcycle")
    assert_close(host(dv), v, 1e-9, 1e-10 * np.max(np.abs(v)), "v after the adaptive subcycle")
    for d, o in zip(ds, s):
        assert_close(thost(d, nx), o, 1e-9, 1e-10 * np.max(np.abs(o)), "stress after the adaptive subcycle")
    # the alphas of the last sub-iteration are local: from the lower bound to far above it
    al = np.zeros((ny, nx))
    O.mevp_stress(nx, ny, 0, ny, b.bt.hx, b.bt.hy, b.po, u, v, pg, *[x.copy() for x in s], dt=120.0, cgh=cgh, cga=cga, alpha_e=al)
    assert float(al.max()) > 4 * float(al.min())
    if variant == 0:
        with pytest.raises(abi.NsdgError, match="no adaptive"):
            ctx.mevp_stress(0, ny, du, dv, tdev(pg), *ds)


def test_adaptive_hip_path_matches_the_independent_restatement(ctx):
    """ONE adaptive sub-iteration on the 6 x 5 case of tests/dyn_independent.py against tests/golden/dyn_independent_v3.npz (elements
    at the lower bound, elements well above it, an element whose centre node is ice-free), variants 1 and 4"""
    import dyn_independent as D

    fix = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dyn_independent_v3.npz"))
    c, nx, ny = D.CASE, D.CASE["nx"], D.CASE["ny"]
    I = lambda k: np.ascontiguousarray(fix["in_" + k])
    W = lambda k: fix["out_" + k]
    for variant in (1, abi.DEFAULT_MEVP_VARIANT):
        ctx.set_grid(nx, ny, c["hx"], c["hy"])
        ctx.set_mevp_params(ctx.mevp_default_params(**dict(D.PARAMS, **D.ADAPTIVE)))
        ctx.set_mevp_variant(variant)
        u, v = dev(I("u")), dev(I("v"))
        S = [tdev(np.ascontiguousarray(x)) for x in I("S")]
        scratch = torch.zeros(10 * u.numel() + 3 * S[0].numel(), dtype=torch.float64, device="cuda")
        ctx.mevp_subcycle(c["dt"], 1, S, u, v, dev(I("u0")), dev(I("v0")), dev(W("tax")), dev(W("tay")), dev(I("uo")), dev(I("vo")),
                          dev(W("cgh")), dev(W("cga")), tdev(np.ascontiguousarray(W("pg"))), scratch)
        for k, name in enumerate(("ad_s11", "ad_s12", "ad_s22")):
            assert_close(thost(S[k], nx), W(name), 1e-11, 1e-12 * np.max(np.abs(W(name))), name)
        assert_close(host(u), W("ad_u_new"), 1e-11, 1e-12 * np.max(np.abs(W("ad_u_new"))), "u after one adaptive sub-iteration")
        assert_close(host(v), W("ad_v_new"), 1e-11, 1e-12 * np.max(np.abs(W("ad_v_new"))), "v after one adaptive sub-iteration")


def test_adaptive_passes_equal_single_sub_iterations_bitwise(ctx):
    """adaptive form: a pass of 2 / 3 / 4 sub-iterations of the stage-per-wave pipeline == that many launches of the single-iteration
    kernel, bit for bit, for several strip heights, widths around the 57 owned columns, a row sub-range and two ranges per launch"""
    for (nx, ny) in ((130, 45), (57, 9), (58, 13), (7, 5), (115, 22)):
        b = Box(ctx, nx, ny, **AD)
        rng = np.random.default_rng(67)
        u, v, s = mevp_state(b, rng)
        u, v = 0.01 * u, 0.01 * v  # strain rates for which the alphas spread over the elements
        pg_o, cgh, cga, tax, tay = box_inputs(b)
        packed = pack(ctx, 120.0, 0.5 * u, 0.5 * v, tax, tay, b.uo, b.vo, cgh, cga)
        pg = tdev(pg_o)
        s_in = [tdev(x) for x in s]
        ctx.set_mevp_variant(1)
        ctx.set_mevp_strip_rows(0)
        cur = s_in + [dev(u), dev(v)]
        refs = {}
        for it in range(1, 5):
            nxt = [torch.zeros_like(x) for x in s_in] + [torch.zeros_like(dev(u)), torch.zeros_like(dev(v))]
            ctx.mevp_iterate(0, 0, ny, cur[:3], nxt[:3], (cur[3], cur[4]), (nxt[3], nxt[4]), packed, pg)
            cur = nxt
            refs[it] = cur
        ctx.set_mevp_variant(4)
        for passes in (2, 3, 4):
            for rows in (1, 5, 0):
                ctx.set_mevp_strip_rows(rows)
                out = [torch.zeros_like(x) for x in s_in] + [torch.full_like(dev(u), 7.0), torch.full_like(dev(v), 7.0)]
                getattr(ctx, "mevp_iterate%d" % passes)(0, ny, s_in, out[:3], (dev(u), dev(v)), (out[3], out[4]), packed, pg)
                for k, (a, c) in enumerate(zip(refs[passes], out)):
                    assert torch.equal(a, c), (nx, ny, passes, rows, k, float((a - c).abs().max()))
        ref = refs[4]
        if ny >= 13:  # a sub-range with ghost rows on both sides
            ctx.set_mevp_strip_rows(3)
            out = [torch.zeros_like(x) for x in s_in] + [torch.zeros_like(dev(u)), torch.zeros_like(dev(v))]
            ctx.mevp_iterate4(4, ny - 3, s_in, out[:3], (dev(u), dev(v)), (out[3], out[4]), packed, pg)
            assert torch.equal(abi.untile(out[0], nx)[:, 4:ny - 3], abi.untile(ref[0], nx)[:, 4:ny - 3])
            assert torch.equal(out[3][8:2 * (ny - 3)], ref[3][8:2 * (ny - 3)])
        if ny >= 22:  # two disjoint ranges in one launch
            ctx.set_mevp_strip_rows(0)
            ra, rb = (ny - 7, ny - 3), (4, 9)
            one = [torch.zeros_like(x) for x in s_in] + [torch.zeros_like(dev(u)), torch.zeros_like(dev(v))]
            ctx.mevp_iterate4_pair(ra, rb, s_in, one[:3], (dev(u), dev(v)), (one[3], one[4]), packed, pg)
            assert torch.equal(abi.untile(one[1], nx)[:, 4:9], abi.untile(ref[1], nx)[:, 4:9])
            assert torch.equal(one[4][2 * (ny - 7):2 * (ny - 3)], ref[4][2 * (ny - 7):2 * (ny - 3)])
        # and the alphas of this state are spread (the test is not the uniform form in disguise)
        al = np.zeros((ny, nx))
        O.mevp_stress(nx, ny, 0, ny, b.bt.hx, b.bt.hy, b.po, u, v, pg_o, *[x.copy() for x in s], dt=120.0, cgh=cgh, cga=cga, alpha_e=al)
        assert nx < 57 or float(al.max()) > 2 * float(al.min()), (al.min(), al.max())  # (the 7 x 5 grid has 73 km cells: every alpha is the lower bound)


def test_adaptive_row_block_equals_full_domain_bitwise(ctx):
    """adaptive form on a row-block sub-domain (4 ghost element rows below, 3 above): the pass of four reproduces the full-domain pass bit
    for bit on the rows it owns -- an element's alpha depends on its own nodes only, a node's beta on its adjacent elements"""
    ctx.set_mevp_variant(4)
    b = Box(ctx, 90, 40, **AD)
    nx, ny = b.nx, b.ny
    rng = np.random.default_rng(71)
    u, v, s = mevp_state(b, rng)
    u, v = 0.01 * u, 0.01 * v
    pg, cgh, cga, tax, tay = box_inputs(b)
    packed = pack(ctx, 120.0, 0.5 * u, 0.5 * v, tax, tay, b.uo, b.vo, cgh, cga)
    full = [torch.zeros_like(tdev(x)) for x in s] + [torch.zeros_like(dev(u)), torch.zeros_like(dev(v))]
    ctx.mevp_iterate4(0, ny, [tdev(x) for x in s], full[:3], (dev(u), dev(v)), (full[3], full[4]), packed, tdev(pg))
    r0, r1 = 14, 27
    lo, hi = r0 - 4, r1 + 3
    sl_e = lambda a: np.ascontiguousarray(a[:, lo:hi])
    sl_n = lambda a: np.ascontiguousarray(a[2 * lo:2 * hi + 1])
    ctx.set_grid(nx, hi - lo, b.bt.hx, b.bt.hy)
    ppacked = pack(ctx, 120.0, *[sl_n(x) for x in (0.5 * u, 0.5 * v, tax, tay, b.uo, b.vo, cgh, cga)])
    part = [torch.zeros_like(tdev(sl_e(x))) for x in s] + [torch.zeros_like(dev(sl_n(u))), torch.zeros_like(dev(sl_n(v)))]
    ctx.mevp_iterate4(4, 4 + r1 - r0, [tdev(sl_e(x)) for x in s], part[:3], (dev(sl_n(u)), dev(sl_n(v))), (part[3], part[4]), ppacked, tdev(sl_e(pg)))
    for k in range(3):
        assert torch.equal(full[k][r0:r1], part[k][4:4 + r1 - r0])
    for k in (3, 4):
        assert torch.equal(full[k][2 * r0:2 * r1], part[k][8:8 + 2 * (r1 - r0)])


def test_adaptive_converged_subcycle_solves_the_implicit_vp_step(ctx):
    """the defining property of mEVP with adaptive alpha / beta on the device: 2500 sub-iterations of the pipelined kernel, then ONE Picard
    sweep of the implicit viscous-plastic step by the independent restatement returns the same stress and velocity -- the same check
    as for the uniform form (tests/test_oracle_dynamics.py), unchanged: the limit does not depend on alpha and beta"""
    from test_oracle_dynamics import check_implicit_vp_fixed_point, implicit_vp_case

    c = implicit_vp_case(adaptive=True)
    nx, ny = c["nx"], c["ny"]
    ctx.set_mevp_params(ctx.mevp_default_params(**c["pk"]))
    ctx.set_grid(nx, ny, c["hx"], c["hy"])
    u, v = dev(c["u0"]), dev(c["v0"])
    s = [tdev(np.zeros((8, ny, nx))) for _ in range(3)]
    scratch = torch.zeros(10 * u.numel() + 3 * s[0].numel(), dtype=torch.float64, device="cuda")
    ctx.mevp_subcycle(c["dt"], 2500, s, u, v, dev(c["u0"]), dev(c["v0"]), dev(c["tax"]), dev(c["tay"]), dev(c["uo"]), dev(c["vo"]),
                      dev(c["cgh"]), dev(c["cga"]), tdev(c["pg"]), scratch)
    check_implicit_vp_fixed_point(c, host(u), host(v), [thost(x, nx) for x in s])


def test_checkpoint_and_resume_on_the_device_is_exact(gpu):
    """the Python driver's checkpoint (DynamicsCore.state_dict / load_state_dict) with the real kernels and the native row-block driver:
    4 steps == 2 steps + checkpoint + a FRESH context and core resumed from it + 2 steps, bit for bit -- H and A with their higher DG2
    coefficients, velocity and stress (tiled on the device, coefficient planes in the checkpoint)"""
    from nextsimdg_amd import rowblock

    nx, ny, dt, nsub = 150, 96, 120.0, 24
    bt = synthetic.BoxTest(nx, ny)
    H, A = bt.dg_fields()
    A[0] -= 0.1
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)

    def fresh():
        c = abi.Context(gpu)
        c.set_mevp_params(c.mevp_default_params(**bt.subcycle_parameters(dt)))
        core = rowblock.DynamicsCore(c, rowblock.RowBlock(nx, ny, 0, 1), bt.hx, bt.hy, dt, nsub, gpu, native=True)
        core.load_global(H, A, uo, vo, 3.0 * ua, 3.0 * va)
        return c, core

    c0, ref = fresh()
    for _ in range(4):
        ref.step()
    c1, first = fresh()
    for _ in range(2):
        first.step()
    state = rowblock.DynamicsCore.merge_states([first.state_dict()])
    assert state["H"].shape == (6, ny, nx) and state["s22"].shape == (8, ny, nx) and float(np.abs(state["H"][1:]).max()) > 0 and float(np.abs(state["s11"]).max()) > 0
    first.close()
    c1.close()
    c2, second = fresh()
    second.load_state_dict(state)
    for _ in range(2):
        second.step()
    torch.cuda.synchronize()
    for name in ("H", "A", "u", "v"):
        assert torch.equal(getattr(second, name), getattr(ref, name)), name
    for a, b in zip(second.s, ref.s):
        assert torch.equal(a, b)
    assert float(ref.u.abs().max()) > 1e-4
    for core, c in ((ref, c0), (second, c2)):
        core.close()
        c.close()


def test_the_converging_form_converges_where_the_published_floor_is_noisy(gpu):
    """What the round's own convergence check found (profiles/r06_adaptive_noise.md), as a regression: the momentum equation of the box test on
    1024 x 1024 (500 m; H, A fixed at 0.3 / 0.9, cyclone wind of t = 0), 150 model steps of 120 sub-iterations from rest.  With alpha_min from
    the mesh (NSDG_SUBCYCLE_ADAPTIVE_CONVERGED: 500 here) the velocity then changes by < 0.5 % of its maximum per model step and by more than 1 mm/s
    nowhere; with the published alpha_min = 50 (the default) every deforming element sits at its stability limit and the step-to-step change stays
    above a tenth of the maximum -- bounded (nothing grows), which is what the coupled runs live on."""
    n, L, dt = 1024, 512e3, 120.0
    bt = synthetic.BoxTest(n, n, L)
    H, A = np.zeros((6, n, n)), np.zeros((6, n, n))
    H[0], A[0] = 0.3, 0.9
    dH, dA = dev(H), dev(A)
    uo, vo = [dev(a) for a in bt.ocean()]
    ua, va = [dev(a) for a in bt.wind(0.0)]
    z = lambda: torch.zeros((2 * n + 1, 2 * n + 1), dtype=torch.float64, device=gpu)
    change = {}
    for mode in ("adaptive_converged", "adaptive"):
        c = abi.Context(gpu)
        sub = bt.subcycle_parameters(dt, mode=mode)
        assert sub["aevp_alpha_min"] == (50.0 if mode == "adaptive" else pytest.approx(500.0, rel=0.01))
        c.set_mevp_params(c.mevp_default_params(**sub))
        c.set_grid(n, n, bt.hx, bt.hy)
        cgh, cga, tax, tay = z(), z(), z(), z()
        c.dg_to_cg(dH, cgh)
        c.dg_to_cg(dA, cga)
        c.wind_stress(ua, va, tax, tay)
        pg = c.private_zeros(9, n, n, gpu)
        c.ice_strength(dH, dA, pg)
        u, v, u0, v0 = z(), z(), z(), z()
        s = [c.private_zeros(8, n, n, gpu) for _ in range(3)]
        scratch = torch.zeros(10 * u.numel() + 3 * s[0].numel(), dtype=torch.float64, device=gpu)
        for _ in range(150):
            u0.copy_(u)
            v0.copy_(v)
            c.mevp_subcycle(dt, 120, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, scratch)
        c.synchronize()
        d = torch.maximum((u - u0).abs(), (v - v0).abs())
        umax = float(torch.maximum(u.abs().max(), v.abs().max()))
        assert bool(torch.isfinite(u).all()) and 0.05 < umax < 0.12
        change[mode] = (float(d.max()) / umax, int((d > 1e-3).sum()))
        c.close()
    assert change["adaptive_converged"][0] < 5e-3 and change["adaptive_converged"][1] == 0, change
    assert 0.1 < change["adaptive"][0] < 1.0 and change["adaptive"][1] > 10000, change


def test_compressible_cover_1024_at_the_literatures_delta_min(gpu):
    """The run that left the physical range in rounds 3-5 (profiles/r04_soak_divergence_cause.md, profiles/r05_closure.md): a uniform cover
    A0 = 0.9, H0 = 0.3 on 1024 x 1024 (500 m), winter forcing, dynamics + column thermodynamics, dt = 120 s, 120 sub-iterations.  With the
    uniform alpha = beta of the stability bound at the literature's Delta_min = 2e-9 (14 438) it fails between step 1000 and 1100 with or
    without the closure; round 5 kept alpha = 1500 and raised Delta_min 90-fold instead (tests/test_gpu_closure.py).  With LOCAL, solution-
    adaptive alpha and beta (round 6, the hosts' default) the run passes that point at Delta_min = 2e-9 itself: 1100 steps = 36.7 model
    hours here, the full 1600 steps in profiles/r06_soak_1024_A09_adaptive_1600_steps.txt."""
    from nextsimdg_amd import rowblock
    from test_gpu_closure import point_values

    nx = ny = 1024
    L, dt, nsub, steps = 512e3, 120.0, 120, 1100
    c = abi.Context(gpu)
    bt = synthetic.BoxTest(nx, ny, L)
    sub = bt.subcycle_parameters(dt)
    assert sub["delta_min"] == 2e-9 and sub["aevp_c"] > 50.0 and sub["aevp_alpha_min"] == 50.0
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
    c.synchronize()  # (also the status of the pipeline's bounded waits)
    for f in (core.u, core.v, core.H, core.A):
        assert bool(torch.isfinite(f).all())
    assert 0.01 < float(core.u.abs().max()) < 0.3
    assert 0.2 < float(core.H[0].min()) and float(core.H[0].max()) < 0.45
    assert 0.6 < float(core.A[0].min()) and float(core.A[0].max()) <= 1.0
    vA = point_values(host(core.A[:, ::8, ::8]))
    assert vA.min() >= -1e-15 and vA.max() <= 1.0 + 1e-15
    core.close()
    c.close()
