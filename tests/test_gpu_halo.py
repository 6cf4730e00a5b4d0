"""The ghost-row exchange behind the C ABI (csrc/halo.hip, nsdg_comm_* / nsdg_halo_*) on one GPU:
plans and pack / unpack kernels on hand-made segments, the RCCL transport in loopback (a communicator of one rank
whose neighbours are the rank itself: real ncclSend / ncclRecv groups), and the in-process transport under the
whole multi-rank driver, bit for bit against the single-domain run."""
import ctypes as C

import pytest
import torch

from nextsimdg_amd import abi
from thread_ranks import gather, run_world

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ctx(gpu):
    c = abi.Context(gpu)
    yield c
    c.close()


def rnd(*shape, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return torch.rand(*shape, dtype=torch.float64, device="cuda", generator=g)


@pytest.mark.parametrize("transport", ["rccl", "local"])
def test_loopback_exchange_delivers_every_block_in_order(ctx, transport):
    """several blocks per direction with odd lengths and odd (8-byte) alignments: what is sent upwards arrives as
    'from below' block by block, what is sent downwards as 'from above'; untouched memory stays untouched"""
    if transport == "rccl":
        ctx.comm_init_rccl(0, 1)
    else:
        ctx.comm_init_local(77, 0, 1)
    a, b, c = rnd(40, 257, seed=1), rnd(9, 3, 512, seed=2), rnd(1001, seed=3)
    ra, rb, rc = torch.zeros_like(a), torch.zeros_like(b), torch.zeros_like(c)
    up = [a[30:37], b[5:8], c[1:400]]  # c[1:] starts 8 bytes off a 16-byte boundary
    down = [a[2:3], c[500:503]]
    from_below = [ra[0:7], rb[0:3], rc[3:402]]
    from_above = [ra[39:40], rc[900:903]]
    plan = ctx.halo_plan(0, 0, up, down, from_above, from_below)
    assert plan.counts()[0] >= sum(v.numel() for v in up)
    for it in range(3):  # the plan is reusable; new data every time
        a.add_(1.0), b.add_(1.0), c.add_(1.0)
        plan.start()
        plan.finish()
        torch.cuda.synchronize()
        for s, r in list(zip(up, from_below)) + list(zip(down, from_above)):
            assert torch.equal(s, r)
    assert float(ra[7:39].abs().max()) == 0 and float(rb[3:].abs().max()) == 0
    assert float(rc[:3].abs().max()) == 0 and float(rc[402:900].abs().max()) == 0 and float(rc[903:].abs().max()) == 0
    plan.close()
    ctx.comm_finalize()


def test_halo_call_sequence_errors(ctx):
    """error behaviour of the boundary: negative status + nsdg_last_error, never an exception from the library"""
    lib = ctx.lib
    h = abi.VP()
    seg = (abi.HaloSeg * 1)()
    x = rnd(16)
    seg[0].ptr, seg[0].count = x.data_ptr(), 16
    # no communicator yet
    assert lib.nsdg_halo_plan_create(ctx.h, 0, 0, 1, seg, 0, seg, 0, seg, 1, seg, C.byref(h)) == -3
    ctx.comm_init_local(78, 0, 1)
    assert lib.nsdg_comm_init_local(ctx.h, 78, 0, 1) == -3  # second communicator on the same context
    assert lib.nsdg_halo_plan_create(ctx.h, 5, 0, 1, seg, 0, seg, 0, seg, 1, seg, C.byref(h)) == -1  # rank out of range
    assert lib.nsdg_halo_plan_create(ctx.h, -1, 0, 1, seg, 1, seg, 0, seg, 0, seg, C.byref(h)) == -1  # segments towards a wall
    plan = ctx.halo_plan(0, 0, [x[0:4]], [], [], [x[8:12]])
    assert lib.nsdg_halo_finish(ctx.h, plan.h) == -3  # finish without start
    plan.start()
    assert lib.nsdg_halo_start(ctx.h, plan.h) == -3  # start twice
    assert b"not finished" in lib.nsdg_last_error()
    plan.finish()
    torch.cuda.synchronize()
    assert torch.equal(x[8:12], x[0:4])
    plan.close()


@pytest.mark.parametrize("world,group,nsub,coupled,variant", [(3, 2, 20, False, 3), (4, 3, 13, True, 2), (2, 1, 7, False, 1)])
def test_driver_on_native_halo_equals_single_domain_bitwise(gpu, world, group, nsub, coupled, variant):
    """the whole multi-rank driver with the exchange of the product (nsdg_halo_*: pack kernel, transport, unpack
    kernel on a communication stream, event-ordered against the compute stream), thread-ranks on the in-process
    transport: bit-identical to the single-domain run"""
    nx, ny, nsteps = 150, 128, 2
    ref = run_world(1, variant, coupled, nx, ny, nsub, nsteps)[0]
    assert float(ref["u"].abs().max()) > 1e-5
    parts = run_world(world, variant, coupled, nx, ny, nsub, nsteps, group=group, transport="native")
    for key in ("H", "A", "u", "v", "s11"):
        assert torch.equal(gather(parts, world, key), ref[key]), (key, world, group)


@pytest.mark.parametrize("world,group,nsub,coupled,variant,graph", [(1, 1, 20, False, 3, False), (1, 1, 41, True, 3, True), (3, 2, 20, False, 3, False),
                                                                    (4, 4, 41, True, 3, True), (4, 3, 13, True, 2, False), (3, 1, 7, False, 1, False),
                                                                    (2, 1, 8, False, 3, True), (4, 5, 31, False, 3, False),
                                                                    (3, 3, 17, True, 2, True), (2, 8, 53, False, 3, False)])
def test_native_row_block_driver_equals_single_domain_bitwise(gpu, world, group, nsub, coupled, variant, graph):
    """the row-block drivers behind the C ABI (nsdg_rb_mevp_run / nsdg_rb_transport_run: one call per step each, with
    and without hipGraph replay of the launches between two exchanges) against the Python sequence on a single
    domain: bit-identical, for groups with remainders, all three kernels, coupled and uncoupled; (4, 5, ...) has blocks too
    short for the overlap split (32 owned rows against 15 + 14 ghost rows + 5)"""
    nx, ny, nsteps = 150, 128, 3
    ref = run_world(1, variant, coupled, nx, ny, nsub, nsteps)[0]
    assert float(ref["u"].abs().max()) > 1e-5
    parts = run_world(world, variant, coupled, nx, ny, nsub, nsteps, group=group, transport="native", native=True, use_graph=graph)
    for key in ("H", "A", "u", "v", "s11"):
        assert torch.equal(gather(parts, world, key), ref[key]), (key, world, group)


@pytest.mark.parametrize("overlap", [True, False])
def test_native_driver_equals_python_driver_in_rccl_loopback(gpu, overlap):
    """an interior block of eight whose neighbours are the rank itself, every exchange a real RCCL send/recv group:
    the values wrap around (meaningless physically) but are deterministic, so the native driver and the Python
    sequence -- same kernels, same exchanges, same order -- must agree bit for bit"""
    from nextsimdg_amd import rowblock, synthetic

    nx, ny, nsub, k = 200, 8 * 48, 23, 2
    bt = synthetic.BoxTest(nx, ny)
    H, A = bt.dg_fields()
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    res = []
    for native in (False, True):
        c = abi.Context(gpu)
        c.set_mevp_params(c.mevp_default_params(alpha=300.0, beta=300.0))
        blk = rowblock.RowBlock(nx, ny, 4, 8, 3 * k, 3 * k - 1)
        ex = rowblock.NativeHaloExchanger(c, blk, loopback=True)
        core = rowblock.DynamicsCore(c, blk, bt.hx, bt.hy, 120.0, nsub, gpu, exchanger=ex, overlap=overlap, native=native, use_graph=native)
        core.load_global(H, A, uo, vo, 3.0 * ua, 3.0 * va)
        for _ in range(2):
            core.step()
        torch.cuda.synchronize()
        res.append([x.clone() for x in (core.u, core.v, core.H, core.A, core.s[0], core.s[2])])
        if native:
            # the sizes bench.py's dry run plans from the geometry alone are the sizes the native plans really move
            import bench

            plan = bench.exchange_plan(blk, nx, nsub, core.per_pass)
            st = core._run_mevp.stats(False)
            # (the packed buffers start every block at a 16-byte boundary: an odd node row adds 8 bytes of padding per block)
            assert 0 <= st["bytes_sent"] - (plan["mevp_exchange_bytes_up"] + plan["mevp_exchange_bytes_down"]) <= 8 * 10, (st, plan)
            assert st["exchanges"] == 2 * plan["mevp_exchanges_per_step"], (st, plan)
            tr = core._run_transport.stats(False)
            assert 0 <= tr["bytes_sent"] - (plan["transport_exchange_bytes_up"] + plan["transport_exchange_bytes_down"]) <= 8 * 24, (tr, plan)
        del core, ex
        c.close()
    assert float(res[0][0].abs().max()) > 1e-6
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_native_driver_long_run_stays_bitwise(gpu):
    """40 model steps (840 sub-iterations, 280 ghost exchanges per rank) of four row blocks on the native driver with
    hipGraph replay: any lost ordering between the compute and the communication streams, or a reused packed buffer,
    would show up as a difference from the single-domain run -- there is none, bit for bit"""
    from nextsimdg_amd import synthetic

    nx, ny, nsub, nsteps = 130, 256, 21, 40
    bt = synthetic.BoxTest(nx, ny)
    H, A = bt.dg_fields()
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    cs, cf = synthetic.column_fields_smooth(nx, ny)
    kw = dict(data=(bt, H, A, uo, vo, ua, va), column={**cs, **cf}, alpha=bt.stable_alpha(120.0), core_kw=dict(forcing="winter"))
    ref = run_world(1, 3, True, nx, ny, nsub, nsteps, **kw)[0]
    assert bool(torch.isfinite(ref["u"]).all()) and 1e-4 < float(ref["u"].abs().max()) < 1.0
    parts = run_world(4, 3, True, nx, ny, nsub, nsteps, group=2, transport="native", native=True, use_graph=True, **kw)
    for key in ("H", "A", "u", "v", "s11", "tice0", "hsnow"):
        assert torch.equal(gather(parts, 4, key), ref[key]), key


def test_row_block_driver_argument_and_sequence_errors(ctx):
    """nsdg_rb_*_create / _run: geometry that does not describe a block, a block with neighbours on a context without a
    communicator, a grid that does not match the plan -- status codes and messages, no launch"""
    from nextsimdg_amd import rowblock

    nx, ny = 70, 40
    ctx.set_grid(nx, ny, 1.0, 1.0)
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
    s2 = ([ctx.private_zeros(8, ny, nx, "cuda") for _ in range(3)], [ctx.private_zeros(8, ny, nx, "cuda") for _ in range(3)])
    uv2 = ((z(2 * ny + 1, 2 * nx + 1), z(2 * ny + 1, 2 * nx + 1)), (z(2 * ny + 1, 2 * nx + 1), z(2 * ny + 1, 2 * nx + 1)))
    packed, pg = z(8 * (2 * ny + 1) * (2 * nx + 1)), ctx.private_zeros(9, ny, nx, "cuda")
    interior = rowblock.RowBlock(nx, 3 * 28, 1, 3, 6, 5)  # 28 owned + 6 + 5 ghost rows = 39 local rows: not this array
    with pytest.raises(abi.NsdgError, match="needs nsdg_comm_init"):
        blk = rowblock.RowBlock(nx, 3 * 29, 1, 3, 6, 5)  # 29 + 11 = 40 local rows, neighbours on both sides, no communicator
        assert blk.ny == ny
        ctx.rb_mevp(blk, (0, 2), 9, True, False, s2, uv2, packed, pg)
    with pytest.raises(abi.NsdgError, match="owned rows must be"):
        bad = rowblock.RowBlock(nx, ny, 0, 1)
        bad.j0 = 3  # owned rows that do not start at the array edge although there is no neighbour below
        ctx.rb_mevp(bad, (None, None), 9, True, False, s2, uv2, packed, pg)
    single = rowblock.RowBlock(nx, ny, 0, 1)
    run, per_pass, group = ctx.rb_mevp(single, (None, None), 7, True, False, s2, uv2, packed, pg)
    assert (per_pass, group) == (abi.DEFAULT_MEVP_VARIANT, 1)
    with pytest.raises(abi.NsdgError, match="pack_nodal was not called"):
        run(0)  # the coefficients of this step were never packed
    ctx.set_grid(nx, ny - 1, 1.0, 1.0)
    with pytest.raises(abi.NsdgError, match="does not match the plan"):
        run(0)
    ctx.set_grid(nx, ny, 1.0, 1.0)
    out = C.c_int32()
    assert ctx.lib.nsdg_rb_mevp_run(ctx.h, run.handle, 2, C.byref(out)) == -1  # parity must be 0 or 1
    f = [z(6, ny, nx) for _ in range(6)]
    adv = (z(6, ny, nx), z(6, ny, nx), z(3, ny, nx + 1), z(3, ny + 1, nx))
    tr = ctx.rb_transport(single, (None, None), f[0:2], f[2:4], f[4:6], adv)
    assert tr(120.0, 0) == 1 and tr(120.0, 1) == 0  # the state alternates between the two array sets
    torch.cuda.synchronize()
    assert interior.ny == 39


def test_two_row_ranges_in_one_launch_equal_two_launches_bitwise(ctx):
    """nsdg_mevp_iterate3_pair (the two bands of rows a block sends to its neighbours, one launch) against two
    nsdg_mevp_iterate3 calls, for band heights 1 .. 30 and both strip-height choices; rows outside the bands untouched"""
    from nextsimdg_amd import synthetic

    nx, ny = 150, 96
    bt = synthetic.BoxTest(nx, ny)
    ctx.set_grid(nx, ny, bt.hx, bt.hy)
    ctx.set_mevp_params(ctx.mevp_default_params(alpha=300.0, beta=300.0))
    dev = lambda a: torch.from_numpy(a.copy()).cuda()
    H, A = (dev(x) for x in bt.dg_fields())
    uo, vo = (dev(x) for x in bt.ocean())
    ua, va = (dev(3.0 * x) for x in bt.wind(0.0))
    g = torch.Generator(device="cuda").manual_seed(5)
    u = 0.05 * torch.rand(2 * ny + 1, 2 * nx + 1, dtype=torch.float64, device="cuda", generator=g)
    v = 0.05 * torch.rand(2 * ny + 1, 2 * nx + 1, dtype=torch.float64, device="cuda", generator=g)
    s = [1e3 * torch.rand(ctx.private_zeros(8, ny, nx, "cuda").shape, dtype=torch.float64, device="cuda", generator=g) for _ in range(3)]
    pg = ctx.private_zeros(9, ny, nx, "cuda")
    packed = torch.zeros(8 * u.numel(), dtype=torch.float64, device="cuda")
    ctx.ice_strength(H, A, pg)
    ctx.mevp_prepare(120.0, H, A, (ua, va), (uo, vo), (u, v), packed)
    for ra, rb, strip in (((72, 96), (0, 24), 0), ((60, 61), (10, 40), 0), ((3, 9), (50, 94), 5), ((70, 94), (24, 48), 1)):
        ctx.set_mevp_strip_rows(strip)
        outs = []
        for pair in (True, False):
            so = [torch.full_like(x, -7.0) for x in s]
            un, vn = torch.full_like(u, -7.0), torch.full_like(v, -7.0)
            if pair:
                ctx.mevp_iterate3_pair(ra, rb, s, so, (u, v), (un, vn), packed, pg)
            else:
                ctx.mevp_iterate3(ra[0], ra[1], s, so, (u, v), (un, vn), packed, pg)
                ctx.mevp_iterate3(rb[0], rb[1], s, so, (u, v), (un, vn), packed, pg)
            outs.append(so + [un, vn])
        torch.cuda.synchronize()
        for a, b in zip(*outs):
            assert torch.equal(a, b), (ra, rb, strip)
        written = torch.zeros(ny, dtype=torch.bool, device="cuda")
        written[ra[0]:ra[1]] = True
        written[rb[0]:rb[1]] = True
        assert bool((outs[0][0][~written] == -7.0).all()) and bool((outs[0][0][written] != -7.0).any())
    ctx.set_mevp_strip_rows(0)
    with pytest.raises(abi.NsdgError, match="disjoint"):
        ctx.mevp_iterate3_pair((10, 40), (30, 60), s, [torch.zeros_like(x) for x in s], (u, v), (torch.zeros_like(u), torch.zeros_like(v)), packed, pg)
    ctx.set_mevp_params(ctx.mevp_default_params())


def test_graph_replay_follows_parameter_time_step_and_strip_changes(gpu):
    """a captured group of launches bakes in the launch constants of the context (K = rho beta / dt, 1 / alpha, Delta_min^2,
    the strip height): after nsdg_mevp_params_set, a new time step or another strip height the cached graphs must be
    dropped, not replayed -- the replaying plan follows the same sequence of changes bit for bit as the plain one"""
    from nextsimdg_amd import rowblock, synthetic

    nx, ny, nsub = 150, 64, 14
    bt = synthetic.BoxTest(nx, ny)
    dev = lambda a: torch.from_numpy(a.copy()).cuda()
    H, A = (dev(x) for x in bt.dg_fields())
    uo, vo = (dev(x) for x in bt.ocean())
    ua, va = (dev(3.0 * x) for x in bt.wind(0.0))
    results = []
    for use_graph in (False, True):
        c = abi.Context(gpu)
        c.set_grid(nx, ny, bt.hx, bt.hy)
        z = lambda: torch.zeros(2 * ny + 1, 2 * nx + 1, dtype=torch.float64, device="cuda")
        s2 = ([c.private_zeros(8, ny, nx, "cuda") for _ in range(3)], [c.private_zeros(8, ny, nx, "cuda") for _ in range(3)])
        uv2 = ((z(), z()), (z(), z()))
        pg = c.private_zeros(9, ny, nx, "cuda")
        packed = torch.zeros(8 * uv2[0][0].numel(), dtype=torch.float64, device="cuda")
        c.ice_strength(H, A, pg)
        run, per_pass, _ = c.rb_mevp(rowblock.RowBlock(nx, ny, 0, 1), (None, None), nsub, True, use_graph, s2, uv2, packed, pg)
        assert per_pass == abi.DEFAULT_MEVP_VARIANT
        par = 0
        for alpha, beta, dt, strip in ((300.0, 300.0, 120.0, 0), (300.0, 300.0, 120.0, 0), (450.0, 300.0, 120.0, 0), (450.0, 520.0, 120.0, 0),
                                       (450.0, 520.0, 90.0, 0), (450.0, 520.0, 90.0, 7), (300.0, 300.0, 120.0, 0)):
            c.set_mevp_params(c.mevp_default_params(alpha=alpha, beta=beta))
            c.set_mevp_strip_rows(strip)
            c.mevp_prepare(dt, H, A, (ua, va), (uo, vo), uv2[par], packed)
            par = run(par)
        torch.cuda.synchronize()
        results.append([x.clone() for x in list(s2[par]) + list(uv2[par])])
        del run
        c.close()
    assert float(results[0][3].abs().max()) > 1e-6
    for a, b in zip(*results):
        assert torch.equal(a, b)


def test_block_with_fewer_owned_rows_than_it_sends_is_rejected(ctx):
    """depth_below element rows travel upwards and depth_above + 1 downwards: a block with neighbours that owns fewer rows
    would pack rows of its own ghost zone -- nsdg_rb_*_create refuses it (the Python RowBlock refuses such a split too)"""
    import types

    nx, own, db, da = 70, 5, 6, 5
    ny = own + db + da
    ctx.comm_init_local(91, 1, 3)
    ctx.set_grid(nx, ny, 1.0, 1.0)
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
    blk = types.SimpleNamespace(nx=nx, ny=ny, j0=db, j1=ny - da, depth_below=db, depth_above=da)
    s2 = ([ctx.private_zeros(8, ny, nx, "cuda") for _ in range(3)], [ctx.private_zeros(8, ny, nx, "cuda") for _ in range(3)])
    uv2 = ((z(2 * ny + 1, 2 * nx + 1), z(2 * ny + 1, 2 * nx + 1)), (z(2 * ny + 1, 2 * nx + 1), z(2 * ny + 1, 2 * nx + 1)))
    packed, pg = z(8 * (2 * ny + 1) * (2 * nx + 1)), ctx.private_zeros(9, ny, nx, "cuda")
    with pytest.raises(abi.NsdgError, match="must own at least"):
        ctx.rb_mevp(blk, (0, 2), 9, True, False, s2, uv2, packed, pg)
    f = [z(6, ny, nx) for _ in range(3)]
    adv = (z(6, ny, nx), z(6, ny, nx), z(3, ny, nx + 1), z(3, ny + 1, nx))
    with pytest.raises(abi.NsdgError, match="must own at least"):
        ctx.rb_transport(blk, (0, 2), f[0:1], f[1:2], f[2:3], adv)
    ctx.comm_finalize()


@pytest.mark.parametrize("nfields", [3, 4])
def test_transport_plan_of_an_interior_block_with_four_fields(gpu, nfields):
    """NSDG_RB_MAX_FIELDS = 4 DG2 fields on an interior block: 6 planes per field and direction = 48 row blocks in one
    plan (the segment table held 32 in round 2).  In RCCL-free loopback (both neighbours are the rank itself) what goes
    up must arrive from below, plane by plane"""
    import types

    nx, own, db, da = 66, 20, 3, 2
    ny = own + db + da
    c = abi.Context(gpu)
    c.comm_init_local(92, 0, 1)
    c.set_grid(nx, ny, 1.0, 1.0)
    g = torch.Generator(device="cuda").manual_seed(11)
    r = lambda *s: torch.rand(*s, dtype=torch.float64, device="cuda", generator=g)
    z = lambda *s: torch.zeros(*s, dtype=torch.float64, device="cuda")
    blk = types.SimpleNamespace(nx=nx, ny=ny, j0=db, j1=ny - da, depth_below=db, depth_above=da)
    phi = [r(6, ny, nx) for _ in range(nfields)]
    t1, t2 = [z(6, ny, nx) for _ in range(nfields)], [z(6, ny, nx) for _ in range(nfields)]
    adv = (z(6, ny, nx), z(6, ny, nx), z(3, ny, nx + 1), z(3, ny + 1, nx))  # velocity zero: a step leaves the owned rows as they are
    tr = c.rb_transport(blk, (0, 0), phi, t1, t2, adv)
    assert tr(120.0, 0) == 1
    torch.cuda.synchronize()
    for f in range(nfields):
        assert torch.allclose(t1[f][:, db:ny - da], phi[f][:, db:ny - da], rtol=1e-14, atol=0)  # (1/3) x + (2/3) x: round-off only
        # loopback: the top depth_below owned rows arrive as the ghost rows below, the bottom depth_above as those above
        assert torch.equal(t1[f][:, :db], t1[f][:, ny - da - db:ny - da]), f
        assert torch.equal(t1[f][:, ny - da:], t1[f][:, db:db + da]), f
    del tr
    c.close()


def test_a_rank_that_exits_early_does_not_block_its_neighbours(gpu):
    """three thread-ranks on the in-process transport with the native row-block driver; rank 1 leaves after its first
    model step (as a process that died would).  Its neighbours must not wait for ever: within the communicator's
    deadline (2 s here) their next exchange reports NSDG_ERR_COMM, and so does every later call on the broken group."""
    import threading
    import time

    from nextsimdg_amd import rowblock
    from thread_ranks import fields

    nx, ny, nsub, world, group_id = 100, 96, 6, 3, 4242
    data = fields(nx, ny)
    bt = data[0]
    out, t_fail = {}, {}
    barrier = threading.Barrier(world)

    def rank_main(rank):
        try:
            c = abi.Context(torch.device("cuda:0"))
            c.set_mevp_params(c.mevp_default_params(alpha=300.0, beta=300.0))
            blk = rowblock.RowBlock(nx, ny, rank, world, 3, 2)
            ex = rowblock.NativeHaloExchanger(c, blk, local_group=group_id)
            c.comm_deadline(2.0)
            core = rowblock.DynamicsCore(c, blk, bt.hx, bt.hy, 120.0, nsub, torch.device("cuda"), exchanger=ex, native=True)
            core.load_global(*data[1:])
            core.step()
            c.synchronize()
            barrier.wait(timeout=60)  # everybody has finished step 1
            if rank == 1:
                out[rank] = "left"
                return  # "dies": never posts another exchange
            t0 = time.perf_counter()
            try:
                for _ in range(3):
                    core.step()
                c.synchronize()
                out[rank] = "no error"
            except abi.NsdgError as e:
                t_fail[rank] = time.perf_counter() - t0
                out[rank] = str(e)
        except BaseException as e:  # noqa: BLE001
            out[rank] = e

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a rank is still blocked"
    assert out[1] == "left"
    for r in (0, 2):
        assert isinstance(out[r], str) and ("did not answer within" in out[r] or "failed" in out[r]), out[r]
        assert t_fail[r] < 30.0, t_fail


def test_bounded_drain_reports_a_stalled_communication_stream(ctx):
    """nsdg_ctx_synchronize on a context with a communicator polls the streams against the deadline instead of blocking:
    with the communication stream held up (the rehearsal aid nsdg_comm_simulate_wire spins for 3 s where a transfer would be --
    the stand-in for an ncclRecv whose sender has died; it ends by itself, nothing is left hanging) and a 0.5 s deadline the
    call returns NSDG_ERR_COMM in time, the communicator is marked broken, and finalising it does not wait either"""
    import time

    ctx.comm_init_local(93, 0, 1)
    x = rnd(4096, seed=9)
    plan = ctx.halo_plan(0, 0, [x[0:1024]], [], [], [x[2048:3072]])
    plan.start()
    plan.finish()
    ctx.synchronize()  # a healthy exchange drains at once
    assert torch.equal(x[2048:3072], x[0:1024])
    assert plan.stats()["exchanges"] == 1 and plan.stats()["ms"] > 0
    ctx.comm_deadline(0.5)
    ctx.comm_simulate_wire(delay_us=3e6)
    plan.start()
    plan.finish()
    ctx.comm_simulate_wire(0.0, 0.0)
    # the host runs ahead of the stalled stream: MORE exchanges than the statistics ring has slots (32) are posted behind
    # the stalled one -- nsdg_halo_start must not wait for the exchange whose ring slot comes round (it never finishes with
    # a dead neighbour), it gives the slot up as untimed
    t0 = time.perf_counter()
    for _ in range(40):
        plan.start()
        plan.finish()
    assert time.perf_counter() - t0 < 0.4, "posting exchanges behind a stalled one blocked the host"
    t0 = time.perf_counter()
    with pytest.raises(abi.NsdgError, match="did not drain within 0.5 s"):
        ctx.synchronize()
    waited = time.perf_counter() - t0
    assert 0.4 < waited < 2.0, waited
    with pytest.raises(abi.NsdgError, match="broken"):
        ctx.synchronize()
    with pytest.raises(abi.NsdgError, match="broken"):  # and no further exchange is posted on a broken communicator
        plan.start()
    t0 = time.perf_counter()
    plan.close()
    ctx.comm_finalize()  # does not drain a broken communicator
    assert time.perf_counter() - t0 < 1.0
    torch.cuda.synchronize()  # the stand-in ends by itself: the device is clean for the next test
