"""Thread-rank harness shared by the one-GPU multi-block tests: every "rank" is a thread with its own nsdg
context and its own row block; ghost rows travel through an in-process mailbox instead of RCCL.  The real HIP
kernels run on real ghost-row layouts with the same DynamicsCore / RowBlock code the bench uses."""
import threading

import numpy as np
import torch

from nextsimdg_amd import abi, rowblock, synthetic


class Mailbox:
    def __init__(self):
        self.cv = threading.Condition()
        self.box = {}
        self.error = None

    def put(self, key, tensor):
        with self.cv:
            self.box.setdefault(key, []).append(tensor)
            self.cv.notify_all()

    def get(self, key):
        with self.cv:
            ok = self.cv.wait_for(lambda: self.error is not None or self.box.get(key), timeout=300)
            if self.error is not None or not ok:
                raise RuntimeError("peer failed or timed out")
            return self.box[key].pop(0)


class ThreadExchanger(rowblock.HaloExchanger):
    """the planning of HaloExchanger (which row blocks travel) with an in-process transport: the views sent to
    a neighbour are cloned into a mailbox, the receiver copies them out in the same order"""

    def __init__(self, blk, mailbox):
        super().__init__(blk)
        self.mb = mailbox

    def _start(self, plan):
        b = self.blk
        if b.world == 1:
            return None
        for views, peer in ((plan.up_send, b.above), (plan.down_send, b.below)):
            for v in views:
                self.mb.put((b.rank, peer), v.clone())
        return plan

    def _finish(self, plan):
        if plan is None:
            return
        b = self.blk
        for views, peer in ((plan.from_above, b.above), (plan.from_below, b.below)):
            for v in views:
                v.copy_(self.mb.get((peer, b.rank)))


def fields(nx, ny, wind_scale=3.0):
    """box-test fields with element-wise noise in A and in the slopes of H (so that no symmetry hides an indexing error)"""
    bt = synthetic.BoxTest(nx, ny)
    rng = np.random.default_rng(71)
    H, A = bt.dg_fields()
    A[0] -= 0.3 * rng.random((ny, nx))
    H[1:3] += 0.02 * rng.standard_normal((2, ny, nx))
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    return bt, H, A, uo, vo, wind_scale * ua, wind_scale * va


_local_group = [1000]  # ids of the in-process communicator groups of csrc/halo.hip, one per run_world()


def run_rank(rank, world, variant, coupled, nx, ny, nsub, nsteps, mailbox, out, overlap, group=1, data=None, column=None,
             alpha=300.0, keep=("H", "A", "u", "v", "s11"), transport="mailbox", local_group=None, native=False, use_graph=False, core_kw=None):
    try:
        ctx = abi.Context(torch.device("cuda:0"))
        ctx.set_mevp_variant(variant)
        # alpha: a number = uniform alpha = beta; a dict = keyword arguments of mevp_default_params (e.g. BoxTest.subcycle_parameters(dt): the
        # hosts' adaptive form)
        ctx.set_mevp_params(ctx.mevp_default_params(**alpha) if isinstance(alpha, dict) else ctx.mevp_default_params(alpha=alpha, beta=alpha))
        bt, H, A, uo, vo, ua, va = data if data is not None else fields(nx, ny)
        depth = (variant * group, variant * group - 1) if variant >= 2 else (1, 1)  # `group` passes of `variant` sub-iterations between two exchanges
        blk = rowblock.RowBlock(nx, ny, rank, world, *depth)
        cls = rowblock.CoupledCore if coupled else rowblock.DynamicsCore
        if transport == "native":  # the exchange of the product (csrc/halo.hip) on its in-process transport
            exchanger = rowblock.NativeHaloExchanger(ctx, blk, local_group=local_group) if world > 1 else None
        else:
            exchanger = ThreadExchanger(blk, mailbox)
        # native: sub-cycle and transport are one C call each (csrc/rowblock.hip) instead of the Python sequence
        core = cls(ctx, blk, bt.hx, bt.hy, 120.0, nsub, torch.device("cuda"), exchanger=exchanger, overlap=overlap, native=native,
                   use_graph=use_graph, **(core_kw or {}))
        core.load_global(H, A, uo, vo, ua, va)
        if coupled:
            if column is None:
                st, fo, _ = synthetic.column_fields(nx * ny, 5)
                column = {k: v.reshape(ny, nx) for k, v in {**st, **fo}.items()}
                column["wind"] = 0.2 * column["wind"]
            core.load_column(column)
        for _ in range(nsteps):
            core.step()
        torch.cuda.synchronize()
        res = {k: core.owned(getattr(core, k)).clone() for k in keep if k != "s11"}
        if "s11" in keep:
            res["s11"] = core.owned(core.s[0]).clone()
        if coupled:
            for k in ("hsnow", "tice0"):
                res[k] = core.col[k][blk.j0:blk.j1].clone()
        out[rank] = res
    except BaseException as e:  # noqa: BLE001 -- wake the peers up, then re-raise in the main thread
        with mailbox.cv:
            mailbox.error = e
            mailbox.cv.notify_all()
        out[rank] = e


def run_world(world, variant, coupled, nx, ny, nsub, nsteps, overlap=True, group=1, **kw):
    mailbox, out = Mailbox(), {}
    _local_group[0] += 1
    kw.setdefault("local_group", _local_group[0])
    threads = [threading.Thread(target=run_rank, args=(r, world, variant, coupled, nx, ny, nsub, nsteps, mailbox, out, overlap, group), kwargs=kw)
               for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r in range(world):
        if isinstance(out[r], BaseException):
            raise out[r]
    return out


def gather(parts, world, key):
    """owned rows of all ranks -> the global array: the DG arrays H, A are [nc, ny, nx] (rows along dim 1), nodal,
    tiled-stress and column arrays have their rows along dim 0"""
    return torch.cat([parts[r][key] for r in range(world)], dim=1 if key in ("H", "A") else 0)
