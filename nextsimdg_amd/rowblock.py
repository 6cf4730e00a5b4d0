"""Row-block decomposition of the structured mesh over the ranks of one node and the per-step driver
of the dynamics core (mEVP sub-cycle + DG2 transport of H and A).

One process per GPU; rank r owns the contiguous element rows [r*ny/R, (r+1)*ny/R) of the x-major
index e = iy*nx + ix (the reference's restart order i*nx + j, core/src/DevGridIO.cpp:107-109) and
keeps one ghost element row on each interior side.  The only communication is nearest-neighbour
send/recv of ghost rows through torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in the CPU tests) -- there is no collective on the data path:

  per mEVP sub-iteration : velocity node rows   up: 2 rows x (u,v) x (2nx+1)     down: 1 row x (u,v)
   (single-iteration       (the stress of the ghost row below is NOT exchanged: it is updated
    kernel)                 redundantly by the rank itself, bit-identically to its owner)
  per GROUP of k passes  : kernels with v = 2, 3 or 4 sub-iterations per pass and ghost depth (d, d-1), d = v k:
   (v k sub-iterations)    d stress rows + 2d node rows travel up, d-1 stress rows + 2d-1 node rows down.
                           Between two exchanges a rank runs k passes on a row range that shrinks by v rows
                           on each side per pass -- the ghost rows are advanced redundantly, bit-identically
                           to their owners -- so the number of messages per step drops by v k
                           (latency-avoiding halo)
  per RK stage           : the ghost element rows of each advected field in both directions (ghost depth
                           >= 3: once per step, the first two stages advance ghost rows redundantly)

Ownership of CG2 nodes is bottom-left: rank r owns node rows [2*r0, 2*r1); the global top node row is a
Dirichlet boundary.  Everything that travels to one neighbour in one exchange (row blocks of several
arrays) is packed into one buffer and sent with one P2P op (HaloExchanger).

The numerical kernels are reached through an `ops` object with the method names of
nextsimdg_amd.abi.Context (the C-ABI binding).  Nothing here computes on the host.
"""
import os

import torch
import torch.distributed as dist


def split_rows(ny, world, rank):
    return (rank * ny) // world, ((rank + 1) * ny) // world


class RowBlock:
    """index bookkeeping of one rank's local array.  `depth_below` / `depth_above` are the numbers of ghost
    element rows kept on the interior sides: (1, 1) for one mEVP sub-iteration per pass, (v k, v k - 1) for the
    kernels with v = 2 or 3 sub-iterations per pass and k passes between two ghost-row exchanges.  Such a
    kernel reads the stress of v element rows below and v - 1 above the rows it updates and the velocity up to
    the bottom node row of the v-th element row above (owned by that row), so complete rows shrink by v per
    pass on both sides; k = 1 gives the (2, 1) / (3, 2) of an exchange after every pass."""

    def __init__(self, nx, ny, rank=0, world=1, depth_below=1, depth_above=1):
        if ny < world * max(depth_below, depth_above, 1) * 2:
            raise ValueError("too few element rows per rank for the ghost depth")
        self.nx, self.ny_glob, self.rank, self.world = nx, ny, rank, world
        self.r0, self.r1 = split_rows(ny, world, rank)
        self.gb = depth_below if rank > 0 else 0  # ghost element rows below
        self.gt = depth_above if rank < world - 1 else 0  # ghost element rows above
        self.depth_below, self.depth_above = depth_below, depth_above
        self.lo, self.hi = self.r0 - self.gb, self.r1 + self.gt  # global rows held locally
        self.ny = self.hi - self.lo  # local element rows
        self.j0, self.j1 = self.gb, self.ny - self.gt  # owned local rows
        self.k0 = max(self.j0 - 1, 0)  # single-iteration kernel: stress is updated from one ghost row below
        self.below = rank - 1 if rank > 0 else None
        self.above = rank + 1 if rank < world - 1 else None

    def elem_slice(self):
        return slice(self.lo, self.hi)

    def node_slice(self):
        return slice(2 * self.lo, 2 * self.hi + 1)


class HaloPlan:
    """what one ghost-row exchange moves: lists of tensor views (row blocks) per direction"""

    def __init__(self):
        self.up_send, self.down_send, self.from_above, self.from_below = [], [], [], []
        self.buffers = None  # transport-private (packed send / receive buffers)
        self.ops = self.flat_up = self.flat_down = None  # transport-private (P2P ops and flattened views, built once)


class HaloExchanger:
    """nearest-neighbour ghost-row exchange.  Planning (which row blocks travel) is separate from transport:
    subclasses that move the data differently (the in-process exchanger of tests/test_gpu_multiblock.py, the
    no-op exchanger of tools/rank_share_timing.py) override _start / _finish only.

    Transport: everything that travels to one neighbour is packed into ONE buffer (torch.cat) and sent with ONE
    P2P op, so a batch has at most 2 sends + 2 receives whatever the number of fields -- the host cost of
    torch.distributed P2P ops (about 13 us each on the GPU box; 20 ops per batch made the 8-block sub-cycle
    host-bound) matters more than the extra device copy of a few MB."""

    def __init__(self, blk, group=None, loopback=False):
        self.blk, self.group = blk, group
        self._cache = {}
        # loopback (rehearsal on one GPU, tools/rank_share_timing.py --rccl-loopback): both neighbours are this
        # rank itself, so a send upwards must meet the receive from below -- receives are posted in that order
        self.loopback = loopback

    # ------------------------------------------------------------------ planning
    def _plan(self, key, build):
        plan = self._cache.get(key)
        if plan is None:
            plan = self._cache[key] = HaloPlan()
            if self.blk.world > 1:
                build(plan)
        return plan

    def _add_nodal(self, plan, fields, rows_down):
        """CG2 nodal arrays [2ny+1, 2nx+1]: 2*depth_below node rows travel upwards, `rows_down` rows (1 for the
        single-iteration kernel, 2*depth_above + 1 for the multi-iteration kernels) downwards"""
        b = self.blk
        up = 2 * b.depth_below
        for f in fields:
            if b.above is not None:
                plan.up_send.append(f[2 * b.j1 - up:2 * b.j1])
                plan.from_above.append(f[2 * b.j1:2 * b.j1 + rows_down])
            if b.below is not None:
                plan.down_send.append(f[2 * b.j0:2 * b.j0 + rows_down])
                plan.from_below.append(f[2 * b.j0 - up:2 * b.j0])

    def _add_rows(self, plan, fields, rows_of):
        """element-row arrays: depth_below rows travel upwards, depth_above rows downwards"""
        b = self.blk
        for f in fields:
            if b.above is not None:
                plan.up_send.append(rows_of(f, b.j1 - b.depth_below, b.j1))
                if b.gt:
                    plan.from_above.append(rows_of(f, b.j1, b.j1 + b.gt))
            if b.below is not None:
                if b.depth_above:
                    plan.down_send.append(rows_of(f, b.j0, b.j0 + b.depth_above))
                plan.from_below.append(rows_of(f, b.j0 - b.gb, b.j0))

    # ------------------------------------------------------------------ transport
    def _start(self, plan):
        b = self.blk
        if b.world == 1 or not (plan.up_send or plan.down_send or plan.from_above or plan.from_below):
            return None
        if plan.buffers is None:
            size = lambda views: sum(v.numel() for v in views)
            like = (plan.up_send or plan.down_send or plan.from_above or plan.from_below)[0]
            new = lambda n: torch.empty(n, dtype=like.dtype, device=like.device) if n else None
            plan.buffers = tuple(new(size(v)) for v in (plan.up_send, plan.down_send, plan.from_above, plan.from_below))
            # the P2P ops and the flattened source views are built once per plan
            s_up, s_down, r_above, r_below = plan.buffers
            ops = []
            if s_up is not None:
                ops.append(dist.P2POp(dist.isend, s_up, b.above, self.group))
            if s_down is not None:
                ops.append(dist.P2POp(dist.isend, s_down, b.below, self.group))
            recv_above = dist.P2POp(dist.irecv, r_above, b.above, self.group) if r_above is not None else None
            recv_below = dist.P2POp(dist.irecv, r_below, b.below, self.group) if r_below is not None else None
            ops += [r for r in ((recv_below, recv_above) if self.loopback else (recv_above, recv_below)) if r is not None]
            plan.ops = ops
            plan.flat_up = [v.reshape(-1) for v in plan.up_send] if all(v.is_contiguous() for v in plan.up_send) else None
            plan.flat_down = [v.reshape(-1) for v in plan.down_send] if all(v.is_contiguous() for v in plan.down_send) else None
        s_up, s_down = plan.buffers[0], plan.buffers[1]
        if s_up is not None:
            torch.cat(plan.flat_up if plan.flat_up is not None else [v.reshape(-1) for v in plan.up_send], out=s_up)
        if s_down is not None:
            torch.cat(plan.flat_down if plan.flat_down is not None else [v.reshape(-1) for v in plan.down_send], out=s_down)
        return dist.batch_isend_irecv(plan.ops), plan

    def _finish(self, handle):
        if handle is None:
            return
        works, plan = handle
        for w in works:
            w.wait()
        for views, buf in ((plan.from_above, plan.buffers[2]), (plan.from_below, plan.buffers[3])):
            if not views:
                continue
            chunks = buf.split([v.numel() for v in views])
            if all(v.is_contiguous() for v in views):  # one multi-tensor launch
                torch._foreach_copy_([v.view(-1) for v in views], list(chunks))
            else:
                for v, c in zip(views, chunks):
                    v.copy_(c.view(v.shape))

    # ------------------------------------------------------------------ the exchanges of the driver
    def nodal_start(self, fields, rows_down=1):
        """post the exchange of the ghost node rows of `fields`"""
        key = ("n", rows_down) + tuple(f.data_ptr() for f in fields)
        return self._start(self._plan(key, lambda p: self._add_nodal(p, fields, rows_down)))

    def finish(self, handle):
        self._finish(handle)

    def nodal(self, fields, rows_down=1):
        self.finish(self.nodal_start(fields, rows_down))

    def rows_exchange_start(self, fields, rows_of, nodal_fields=(), rows_down=1):
        """post ONE batch with the ghost rows of the private arrays `fields` (rows taken with ops.private_rows():
        for the tiled device layout a row range is one contiguous block) and, optionally, the ghost node rows of
        `nodal_fields`"""
        key = ("r", rows_down) + tuple(f.data_ptr() for f in fields) + tuple(f.data_ptr() for f in nodal_fields)

        def build(p):
            self._add_rows(p, fields, rows_of)
            self._add_nodal(p, nodal_fields, rows_down)

        return self._start(self._plan(key, build)), ()

    def rows_exchange_finish(self, handle, unused=()):
        self._finish(handle)

    def element(self, fields):
        """refresh the ghost element rows of DG arrays [nc, ny, nx] (after a transport stage)"""
        key = ("e",) + tuple(f.data_ptr() for f in fields)
        plan = self._plan(key, lambda p: self._add_rows(p, fields, lambda f, a, c: f[:, a:c, :]))
        self._finish(self._start(plan))


class NativeHaloExchanger(HaloExchanger):
    """the planning of HaloExchanger with the transport behind the C ABI (csrc/halo.hip): pack kernel, RCCL
    send/recv group on the context's communication stream, unpack kernel -- three C calls and no torch op per
    exchange.  `local_group`: in-process transport between thread-ranks instead of RCCL (one-GPU tests)."""

    def __init__(self, ctx, blk, device=None, group=None, loopback=False, local_group=None):
        super().__init__(blk, group, loopback)
        self.ctx = ctx
        if loopback:
            # rehearsal of an interior block on one GPU: a communicator of ONE rank, both neighbours are that rank
            # (what goes up arrives from below); the values wrap around, the calls and sizes are the real ones
            if local_group is not None:
                ctx.comm_init_local(local_group, 0, 1)
            else:
                ctx.comm_init_rccl(0, 1)
            self.peer_below = 0 if blk.below is not None else None
            self.peer_above = 0 if blk.above is not None else None
        else:
            if local_group is not None:
                ctx.comm_init_local(local_group, blk.rank, blk.world)
            else:
                ctx.comm_init_rccl(blk.rank, blk.world, group)
            self.peer_below, self.peer_above = blk.below, blk.above

    def _plan(self, key, build):
        plan = self._cache.get(key)
        if plan is None:
            plan = self._cache[key] = HaloPlan()
            if self.blk.world > 1:
                build(plan)
                if plan.up_send or plan.down_send or plan.from_above or plan.from_below:
                    plan.native = self.ctx.halo_plan(self.peer_below, self.peer_above, plan.up_send, plan.down_send,
                                                     plan.from_above, plan.from_below)
        return plan

    def _start(self, plan):
        native = getattr(plan, "native", None)
        if native is None:
            return None
        native.start()
        return native

    def _finish(self, handle):
        if handle is not None:
            handle.finish()


class DynamicsCore:
    """State and time step of the dynamics core on one rank's row block.

    step() = one model time step: nodal means of H and A, ice strength at the Gauss points, wind
    stress, `nsub` mEVP sub-iterations (velocity halo after each), advection-velocity preparation and
    one SSP-RK3 DG2 transport step of H and A (element halo after each stage)."""

    ORDER = 2
    # closure of the transported fields (include/nsdg.h "INPUT DOMAIN AND CLOSURE"): mean thickness H >= 0; concentration
    # 0 <= A <= 1 at the quadrature points with the cell mean capped at 1 (ridging) -- (lo, hi, cap_mean) per field
    BOUNDS = ((0.0, float("inf"), False), (0.0, 1.0, True))

    def __init__(self, ops, blk, hx, hy, dt, nsub, device, exchanger=None, overlap=True, native=False, use_graph=False, closure=True):
        self.ops, self.blk, self.hx, self.hy, self.dt, self.nsub = ops, blk, hx, hy, dt, nsub
        self.overlap = overlap
        # closure: cap + scaling limiter at the end of every transport step (the ice-free-node rule is a parameter of the
        # sub-cycle, on by default).  False: the bare scheme of rounds 1-4 (frozen fixtures of that scheme)
        self.closure = closure
        # the native driver's plan carries the bounds itself (nsdg_rb_transport_desc.own_bounds); the Python sequence of step calls reads
        # them from the context: stated here, put back by close() (they apply to EVERY later step call on the context: include/nsdg.h)
        self._bounds_before = getattr(ops, "transport_bounds", ())
        ops.set_transport_bounds(self.BOUNDS if closure else ())
        # native: the sub-cycle and the transport of a step are ONE C call each (csrc/rowblock.hip runs the same
        # sequence of passes and exchanges as subcycle() / transport() below); needs the C-ABI ops and, with
        # neighbours, a NativeHaloExchanger (it owns the communicator).  use_graph: replay the launches between
        # two exchanges as one hipGraph.
        self.native, self.use_graph = native, use_graph
        self._calls = {}
        # v sub-iterations per kernel pass (v = 4: variant 4, v = 3: variant 3, v = 2: variant 2) need a (v k, v k - 1) ghost depth
        # for k passes between two exchanges; a single domain has no ghosts at all.  All variants produce
        # bit-identical results, so a variant-3 context on a (2k, 2k-1) block simply uses the two-iteration kernel.
        variant = getattr(ops, "mevp_variant", None)
        self.per_pass = 1
        for v in (4, 3, 2):
            deep = blk.depth_below >= v and blk.depth_below % v == 0 and blk.depth_above == blk.depth_below - 1
            if variant is not None and variant >= v and (blk.world == 1 or deep):
                self.per_pass = v
                break
        self.two_per_pass = self.per_pass >= 2  # the ghost zones hold stress rows as well as velocity rows
        self.group_passes = blk.depth_below // self.per_pass if (self.per_pass >= 2 and blk.world > 1) else 1  # passes between two exchanges
        self.halo = exchanger if exchanger is not None else HaloExchanger(blk)
        nx, ny = blk.nx, blk.ny
        z = lambda *s: torch.zeros(*s, dtype=torch.float64, device=device)
        nodal = (2 * ny + 1, 2 * nx + 1)
        self.H, self.A = z(6, ny, nx), z(6, ny, nx)
        # stress and ice strength are private to the sub-cycle: the ops object chooses their layout
        self.s = [ops.private_zeros(8, ny, nx, device) for _ in range(3)]
        self.sb = [ops.private_zeros(8, ny, nx, device) for _ in range(3)]
        self.pg = ops.private_zeros(9, ny, nx, device)
        self.u, self.v, self.ub, self.vb = z(*nodal), z(*nodal), z(*nodal), z(*nodal)
        self.ua, self.va = z(*nodal), z(*nodal)
        self.uo, self.vo = z(*nodal), z(*nodal)
        self.packed = z(nodal[0] * nodal[1] * 8)  # per-step momentum coefficients, 8 per node
        self.adv = (z(6, ny, nx), z(6, ny, nx), z(3, ny, nx + 1), z(3, ny + 1, nx))
        self.t1 = [z(6, ny, nx), z(6, ny, nx)]
        self.t2 = [z(6, ny, nx), z(6, ny, nx)]
        if native:
            self._init_native()

    def _init_native(self):
        b = self.blk
        if b.world > 1 and not isinstance(self.halo, NativeHaloExchanger):
            raise ValueError("the native driver exchanges ghost rows through the C ABI: pass a NativeHaloExchanger")
        peers = (self.halo.peer_below, self.halo.peer_above) if b.world > 1 else (None, None)
        self._sbuf, self._uvbuf, self._par = (self.s, self.sb), ((self.u, self.v), (self.ub, self.vb)), 0
        self._run_mevp, per_pass, group = self.ops.rb_mevp(b, peers, self.nsub, self.overlap, self.use_graph, self._sbuf, self._uvbuf,
                                                           self.packed, self.pg)
        assert (per_pass, group) == (self.per_pass, self.group_passes), "native plan and driver disagree on the pass structure"
        self._fbuf, self._tpar = ((self.H, self.A), (self.t1[0], self.t1[1])), 0
        self._run_transport = self.ops.rb_transport(b, peers, self._fbuf[0], self._fbuf[1], self.t2, self.adv, bounds=self.BOUNDS if self.closure else ())

    def close(self):
        """releases the native driver plans (device buffers and events of their ghost exchanges); call it before the
        context is closed"""
        for name in ("_run_mevp", "_run_transport"):
            run = getattr(self, name, None)
            if run is not None and hasattr(run, "close"):
                run.close()
            setattr(self, name, None)
        if self._bounds_before is not None:  # the context is as this core found it
            self.ops.set_transport_bounds(self._bounds_before)
            self._bounds_before = None

    def load_global(self, H, A, uo, vo, ua, va, u=None, v=None):
        """fill the local arrays (ghost rows included) from global numpy arrays"""
        es, ns = self.blk.elem_slice(), self.blk.node_slice()
        put = lambda dst, src: dst.copy_(torch.from_numpy(src).to(dst.device))
        import numpy as np

        put(self.H, np.ascontiguousarray(H[:, es]))
        put(self.A, np.ascontiguousarray(A[:, es]))
        for dst, src in ((self.uo, uo), (self.vo, vo), (self.ua, ua), (self.va, va)):
            put(dst, np.ascontiguousarray(src[ns]))
        if u is not None:
            put(self.u, np.ascontiguousarray(u[ns]))
            put(self.v, np.ascontiguousarray(v[ns]))

    def _set_grid(self):
        self.ops.set_grid(self.blk.nx, self.blk.ny, self.hx, self.hy)
        place = getattr(self.ops, "set_block", None)
        if place is not None:  # where the local array sits in the global domain (device-side forcing providers)
            place(self.blk.lo, self.blk.ny_glob)

    def device_wind(self, domain_size, t):
        """cyclone wind of the box test at model time t, evaluated on the device for this rank's rows"""
        self._set_grid()
        self.ops.boxtest_forcing(domain_size, t, wind=(self.ua, self.va))

    def momentum(self):
        self.prepare()
        self.subcycle()

    def prepare(self):
        """once per model step: ice strength at the Gauss points, nodal means of H and A, wind stress and the packed
        momentum coefficients (one launch); the velocity at the start of the step is read from the current iterate
        (it is only needed inside the packing)"""
        ops, b = self.ops, self.blk
        ops.ice_strength(self.H, self.A, self.pg, 0, b.ny)
        ops.mevp_prepare(self.dt, self.H, self.A, (self.ua, self.va), (self.uo, self.vo), (self.u, self.v), self.packed)

    def passes_per_step(self):
        """kernel launches of the dominant kernel per sub-cycle on an unsplit block: (launches with per_pass
        sub-iterations, two-iteration remainders, single sub-iterations)"""
        n, v = self.nsub, self.per_pass
        if v == 1:
            return 0, 0, n
        full, rest = n // v, n % v  # the remainder runs through the kernels with fewer sub-iterations per pass: 3 -> one
        if rest == 3:  # three-iteration pass (counted with the "twos": a minority launch), 2 -> one two-iteration pass, 1 -> a single
            return full, 1, 0
        return full, rest // 2, rest % 2

    def subcycle(self):
        """the nsub mEVP sub-iterations of one model step (ghost rows exchanged as the ghost depth requires)"""
        ops, b = self.ops, self.blk
        if self.native:
            par = self._par = self._run_mevp(self._par)
            self.s, self.sb = self._sbuf[par], self._sbuf[1 - par]
            (self.u, self.v), (self.ub, self.vb) = self._uvbuf[par], self._uvbuf[1 - par]
            return
        it = 0
        if self.per_pass >= 2:
            # v sub-iterations per pass (the intermediate stress / velocity stay on chip).  With several ranks the
            # passes run in groups of k = group_passes: pass i of a group of m covers the owned rows plus
            # v(m-i) ghost rows on each side (what the remaining passes of the group will read), and only
            # after the last pass the ghost rows of the new stress and velocity are exchanged.  What is left
            # of nsub after the v-passes is done by a two-iteration pass and / or single sub-iterations.
            k = self.group_passes
            split_ok = self.overlap and b.world > 1 and (b.j1 - b.j0) >= b.depth_below + b.depth_above + 5
            for v in range(self.per_pass, 1, -1):
                while self.nsub - it >= v:
                    m = min(k, (self.nsub - it) // v)
                    for i in range(1, m + 1):
                        last = i == m
                        calls = self._pass_calls(v, split_ok and last, m - i)
                        for c in calls[:-1]:
                            c()
                        pending = self._ghost_exchange_start() if (split_ok and last) else None
                        calls[-1]()
                        if last:
                            if pending is None:
                                pending = self._ghost_exchange_start()
                            self._ghost_exchange_finish(pending)
                        self.u, self.ub = self.ub, self.u
                        self.v, self.vb = self.vb, self.v
                        self.s, self.sb = self.sb, self.s
                        it += v
        split = self.overlap and b.world > 1 and (b.j1 - b.j0) >= 4 and not self.two_per_pass
        for _ in range(self.nsub - it):
            uvn = (self.ub, self.vb)
            calls = self._iterate_calls(split)
            if not split:
                calls[0]()
                if self.two_per_pass:  # keep the deeper ghost zones (stress as well as velocity) of the multi-iteration passes consistent
                    self._ghost_exchange_finish(self._ghost_exchange_start())
                else:
                    self.halo.nodal(uvn)
            else:
                # boundary rows first, so that their node rows travel while the interior is computed:
                # the exchange is posted after the boundary launches and before the interior launch, the
                # communication stream therefore waits only for the former
                for c in calls[:-1]:
                    c()
                reqs = self.halo.nodal_start(uvn)
                calls[-1]()
                self.halo.finish(reqs)
            self.u, self.ub = self.ub, self.u
            self.v, self.vb = self.vb, self.v
            self.s, self.sb = self.sb, self.s

    def _ghost_exchange_start(self):
        """ghost zones of the multi-iteration passes, depth (d, d-1) with d = v k: velocity node rows (2d up, 2d-1
        down) and stress rows (d up, d-1 down) in one batch"""
        if self.blk.world == 1:
            return None
        return self.halo.rows_exchange_start(self.sb, self.ops.private_rows, nodal_fields=(self.ub, self.vb),
                                             rows_down=2 * self.blk.depth_above + 1)

    def _ghost_exchange_finish(self, pending):
        if pending is not None:
            self.halo.rows_exchange_finish(*pending)

    def _pass_calls(self, v, split, ext=0):
        """launches of one pass of v (2, 3 or 4) sub-iterations for the current ping-pong parity (bound once, cached).
        ext > 0: a pass inside a group, one launch over the owned rows extended by v*ext ghost rows on each
        side.  ext == 0 and split: the rows whose results travel to the neighbours first, the interior last"""
        key = (self.u.data_ptr(), self.s[0].data_ptr(), split, v, ext)
        calls = self._calls.get(key)
        if calls is not None:
            return calls
        ops, b = self.ops, self.blk
        uv, uvn = (self.u, self.v), (self.ub, self.vb)
        name = "mevp_iterate%d" % v
        bind = getattr(ops, "bind_" + name, None)
        if bind is None:  # ops without a binding fast path (the CPU test stand-in)
            bind = lambda *a: (lambda: getattr(ops, name)(*a))
        rng = []
        lo, hi = max(b.j0 - v * ext, 0), min(b.j1 + v * ext, b.ny)
        if split and ext == 0:
            if b.above is not None:  # top owned element rows: depth_below stress rows + 2*depth_below node rows go up
                rng.append((b.j1 - b.depth_below, b.j1))
                hi = b.j1 - b.depth_below
            if b.below is not None:  # bottom owned element rows: depth_above stress rows + 2*depth_above+1 node rows go down
                rng.append((b.j0, b.j0 + b.depth_above + 1))
                lo = b.j0 + b.depth_above + 1
        rng.append((lo, hi))
        calls = [bind(j0, j1, self.s, self.sb, uv, uvn, self.packed, self.pg) for (j0, j1) in rng]
        self._calls[key] = calls
        return calls

    def _iterate_calls(self, split):
        """the launches of one sub-iteration for the current ping-pong parity, bound once and cached"""
        key = (self.u.data_ptr(), self.s[0].data_ptr(), split)
        calls = self._calls.get(key)
        if calls is not None:
            return calls
        ops, b = self.ops, self.blk
        uv, uvn = (self.u, self.v), (self.ub, self.vb)
        bind = getattr(ops, "bind_mevp_iterate", None)
        if bind is None:  # ops without a binding fast path (the CPU test stand-in)
            bind = lambda *a: (lambda: ops.mevp_iterate(*a))
        rng = []
        if not split:
            rng.append((b.k0, b.j0, b.j1))  # k0 = j0 - 1: the ghost row just below is updated redundantly
        else:
            lo, hi = b.j0, b.j1
            if b.above is not None:  # top owned element row -> the two node rows sent upwards
                rng.append((b.j1 - 2, b.j1 - 1, b.j1))
                hi = b.j1 - 1
            if b.below is not None:  # bottom owned element row (+ redundant ghost-row stress) -> node row sent downwards
                rng.append((b.j0 - 1, b.j0, b.j0 + 1))
                lo = b.j0 + 1
            rng.append((lo - 1 if lo > 0 else 0, lo, hi))  # interior, launched after the exchange is posted
        calls = [bind(k0, j0, j1, self.s, self.sb, uv, uvn, self.packed, self.pg) for (k0, j0, j1) in rng]
        self._calls[key] = calls
        return calls

    def transport(self):
        ops, b = self.ops, self.blk
        ops.prepare_advection(self.ORDER, self.u, self.v, *self.adv)
        if self.native:
            par = self._tpar = self._run_transport(self.dt, self._tpar)
            (self.H, self.A), (self.t1[0], self.t1[1]) = self._fbuf[par], self._fbuf[1 - par]
            return
        f = [self.H, self.A]
        # Shu-Osher SSP-RK3: out = a*phi0 + b*(phis + dt L(phis)).  A stage reads one element row on each side
        # of the rows it updates: with at least 3 ghost rows per interior side the first two stages also
        # advance 2 / 1 ghost rows redundantly (bit-identically to their owners) and the ghost rows are
        # exchanged once per step instead of after every stage.
        deep = b.world > 1 and min(b.depth_below, b.depth_above) >= 3
        ext = (lambda e: (max(b.j0 - e, 0), min(b.j1 + e, b.ny))) if deep else (lambda e: (b.j0, b.j1))
        ops.transport_stage(self.ORDER, *ext(2), self.dt, 0.0, 1.0, f, f, self.t1, self.adv)
        if not deep:
            self.halo.element(self.t1)
        ops.transport_stage(self.ORDER, *ext(1), self.dt, 0.75, 0.25, f, self.t1, self.t2, self.adv)
        if not deep:
            self.halo.element(self.t2)
        ops.transport_stage(self.ORDER, b.j0, b.j1, self.dt, 1.0 / 3.0, 2.0 / 3.0, f, self.t2, self.t1, self.adv)
        if self.closure:  # the step entry points of the library do this themselves; a step composed of stages calls it
            ops.transport_limit(self.ORDER, b.j0, b.j1, self.t1)
        self.halo.element(self.t1)
        # the new state is t1 (ghost rows refreshed); swap buffers instead of copying
        self.H, self.t1[0] = self.t1[0], self.H
        self.A, self.t1[1] = self.t1[1], self.A

    def step(self):
        self._set_grid()
        self.momentum()
        self.transport()

    def owned(self, f):
        """owned element rows of a DG array / owned node rows of a nodal array (for gathering)"""
        b = self.blk
        if any(f is x for x in self.s + self.sb + [self.pg]):
            return self.ops.private_rows(f, b.j0, b.j1)
        if f.dim() == 3:
            return f[:, b.j0:b.j1]
        top = 2 * b.j1 + (1 if b.above is None else 0)
        return f[2 * b.j0:top]


    # ---- checkpoint / resume (round 6).  The state a run carries from one step to the next: the DG2 fields H and A, the velocity and the
    # stress -- what the C++ host writes into its restart file (host/include/RectGrid.hpp: hice, cice, hice_dg, cice_dg, u, v, s11, s12,
    # s22), which the reference's restart files are the model for (core/src/DevGridIO.cpp:169-201: every prognostic field it has).
    def state_dict(self):
        """numpy arrays of the rows this rank OWNS (H, A: [6, rows, nx]; u, v: owned node rows; s11, s12, s22: [8, rows, nx] coefficient
        planes) and the position of the rows in the global domain: concatenating the ranks' dictionaries along the row axis gives the
        global state, load_state_dict takes either"""
        b = self.blk
        nx = b.nx
        out = {"rows": (b.r0, b.r1), "ny_global": b.ny_glob, "nx": nx}
        for name, f in (("H", self.H), ("A", self.A), ("u", self.u), ("v", self.v)):
            out[name] = self.owned(f).detach().cpu().numpy().copy()
        for name, f in zip(("s11", "s12", "s22"), self.s):
            out[name] = self.ops.private_to_planes(f, nx)[:, b.j0:b.j1].detach().cpu().numpy().copy()
        return out

    def load_state_dict(self, state):
        """resume from a GLOBAL state (rows (0, ny_global): e.g. the ranks' state_dict()s concatenated): fills the local arrays, ghost rows
        included, and the current ping-pong buffers; forcing (load_global) and column fields are loaded as for a fresh run"""
        import numpy as np

        b = self.blk
        if tuple(state["rows"]) != (0, b.ny_glob) or state["nx"] != b.nx:
            raise ValueError("load_state_dict needs the state of the whole domain (rows (0, %d)), got rows %s" % (b.ny_glob, (state["rows"],)))
        es, ns = b.elem_slice(), b.node_slice()
        put = lambda dst, src: dst.copy_(torch.from_numpy(np.ascontiguousarray(src)).to(dst.device))
        put(self.H, state["H"][:, es])
        put(self.A, state["A"][:, es])
        put(self.u, state["u"][ns])
        put(self.v, state["v"][ns])
        for f, name in zip(self.s, ("s11", "s12", "s22")):
            f.copy_(self.ops.planes_to_private(torch.from_numpy(np.ascontiguousarray(state[name][:, es])).to(f.device)))

    @staticmethod
    def merge_states(states):
        """the ranks' state_dict()s (any order) -> the state of the whole domain"""
        import numpy as np

        states = sorted(states, key=lambda s: s["rows"][0])
        out = {"rows": (states[0]["rows"][0], states[-1]["rows"][1]), "ny_global": states[0]["ny_global"], "nx": states[0]["nx"]}
        for k in ("H", "A", "s11", "s12", "s22"):
            out[k] = np.concatenate([s[k] for s in states], axis=1)
        for k in ("u", "v"):
            out[k] = np.concatenate([s[k] for s in states], axis=0)
        return out


class CoupledCore(DynamicsCore):
    """BASELINE config 5: dynamics + column thermodynamics.  Each model step first advances the column
    physics of every owned element (the reference's DevStep::iterate, core/src/DevStep.cpp:14-23) and then
    the dynamics.  The coupling is zero-copy: the cell means of the DG fields ARE the column model's
    prognostic variables -- plane 0 of H is `hice`, plane 0 of A is `cice` -- so the column kernel reads
    and writes those planes in place; higher DG coefficients are left as they are (a thermodynamic
    source changes the mean, not the sub-cell shape).  The column step needs no exchange (elements are
    independent); it runs on the ghost rows too, redundantly, so that they stay consistent without a
    message."""

    COLUMN_STATE = ("hsnow", "tice0")
    COLUMN_FORCING = ("sst", "sss", "tair", "tdew", "slp", "qsw", "qlw", "mld", "snowfall", "wind")

    def __init__(self, ops, blk, hx, hy, dt, nsub, device, forcing=None, **kw):
        """forcing: None = the forcing planes are whatever load_column() put there (constant in time);
        "dummy" / "winter" = regenerated on the device at every step's model time (nsdg_column_forcing) and the
        column wind speed is |u_a| of the dynamics' wind (nsdg_column_wind) -- the replacement of the reference's
        DummyExternalData (core/src/include/DummyExternalData.hpp:22-34) and of its never-set windSpeed"""
        super().__init__(ops, blk, hx, hy, dt, nsub, device, **kw)
        z = lambda: torch.zeros(blk.ny, blk.nx, dtype=torch.float64, device=device)
        self.col = {k: z() for k in self.COLUMN_STATE + self.COLUMN_FORCING}
        self.newice = z()
        self.forcing, self.time = forcing, 0.0

    def load_column(self, fields):
        """fields: dict name -> global [ny, nx] numpy array for hsnow, tice0 and the 10 forcing fields"""
        import numpy as np

        es = self.blk.elem_slice()
        for k, dst in self.col.items():
            dst.copy_(torch.from_numpy(np.ascontiguousarray(fields[k][es])).to(dst.device))

    def thermodynamics(self):
        state = {"hice": self.H[0], "cice": self.A[0], "hsnow": self.col["hsnow"], "tice0": self.col["tice0"]}
        forcing = {k: self.col[k] for k in self.COLUMN_FORCING}
        self.ops.column_step(self.dt, state, forcing, self.newice)

    def external_forcing(self):
        if self.forcing is not None:
            self.ops.column_forcing(self.forcing, self.time, self.col)
            self.ops.column_wind(self.ua, self.va, self.col["wind"])

    def step(self):
        self._set_grid()
        self.external_forcing()
        self.thermodynamics()
        self.momentum()
        self.transport()
        self.time += self.dt
