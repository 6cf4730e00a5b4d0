"""The multi-rank driver on ONE GPU: every "rank" is a thread with its own nsdg context and its own row
block; ghost rows travel through an in-process mailbox instead of RCCL.  This runs the real HIP kernels on
real ghost-row layouts (depth (2,1) for the two-iterations-per-pass kernel, (1,1) otherwise), with the same
DynamicsCore / RowBlock code the bench uses, and must reproduce the single-domain run bit for bit.  (What it
cannot cover is RCCL itself: there is one GPU per test box.)"""
import threading

import numpy as np
import pytest
import torch

from nextsimdg_amd import abi, rowblock, synthetic

pytestmark = pytest.mark.gpu


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
            ok = self.cv.wait_for(lambda: self.error is not None or self.box.get(key), timeout=120)
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


def fields(nx, ny):
    bt = synthetic.BoxTest(nx, ny)
    rng = np.random.default_rng(71)
    H, A = bt.dg_fields()
    A[0] -= 0.3 * rng.random((ny, nx))
    H[1:3] += 0.02 * rng.standard_normal((2, ny, nx))
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    return bt, H, A, uo, vo, 3.0 * ua, 3.0 * va


def run_rank(rank, world, variant, coupled, nx, ny, nsub, nsteps, mailbox, out, overlap, group=1):
    try:
        ctx = abi.Context(torch.device("cuda:0"))
        ctx.set_mevp_variant(variant)
        ctx.set_mevp_params(ctx.mevp_default_params(alpha=300.0, beta=300.0))
        bt, H, A, uo, vo, ua, va = fields(nx, ny)
        depth = (variant * group, variant * group - 1) if variant >= 2 else (1, 1)  # `group` passes of `variant` sub-iterations between two exchanges
        blk = rowblock.RowBlock(nx, ny, rank, world, *depth)
        cls = rowblock.CoupledCore if coupled else rowblock.DynamicsCore
        core = cls(ctx, blk, bt.hx, bt.hy, 120.0, nsub, torch.device("cuda"), exchanger=ThreadExchanger(blk, mailbox),
                   overlap=overlap)
        core.load_global(H, A, uo, vo, ua, va)
        if coupled:
            st, fo, _ = synthetic.column_fields(nx * ny, 5)
            col = {k: v.reshape(ny, nx) for k, v in {**st, **fo}.items()}
            col["wind"] = 0.2 * col["wind"]
            core.load_column(col)
        for _ in range(nsteps):
            core.step()
        torch.cuda.synchronize()
        out[rank] = {k: core.owned(getattr(core, k)).clone() for k in ("H", "A", "u", "v")}
        out[rank]["s11"] = core.owned(core.s[0]).clone()
    except BaseException as e:  # noqa: BLE001 -- wake the peers up, then re-raise in the main thread
        with mailbox.cv:
            mailbox.error = e
            mailbox.cv.notify_all()
        out[rank] = e


def run_world(world, variant, coupled, nx, ny, nsub, nsteps, overlap=True, group=1):
    mailbox, out = Mailbox(), {}
    threads = [threading.Thread(target=run_rank, args=(r, world, variant, coupled, nx, ny, nsub, nsteps, mailbox, out, overlap, group))
               for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for r in range(world):
        if isinstance(out[r], BaseException):
            raise out[r]
    return out


@pytest.mark.parametrize("world,variant,coupled", [(2, 2, False), (3, 2, False), (4, 2, True), (3, 1, False), (2, 1, True)])
def test_row_blocks_on_one_gpu_equal_single_domain_bitwise(gpu, world, variant, coupled):
    nx, ny, nsub, nsteps = 150, 64, 7, 2  # odd nsub: double passes + one single sub-iteration
    ref = run_world(1, variant, coupled, nx, ny, nsub, nsteps)[0]
    assert float(ref["u"].abs().max()) > 1e-5
    parts = run_world(world, variant, coupled, nx, ny, nsub, nsteps)
    for key, dim in (("H", 1), ("A", 1), ("u", 0), ("v", 0), ("s11", 0)):
        got = torch.cat([parts[r][key] for r in range(world)], dim=dim)
        assert got.shape == ref[key].shape, key
        assert torch.equal(got, ref[key]), (key, world, variant)


@pytest.mark.parametrize("world,group,nsub,coupled,variant", [(2, 2, 9, False, 2), (3, 4, 19, False, 2), (4, 3, 13, True, 2),
                                                              (2, 1, 8, False, 3), (3, 2, 20, False, 3), (4, 4, 41, True, 3)])
def test_grouped_passes_with_deep_ghost_zones_bitwise(gpu, world, group, nsub, coupled, variant):
    """latency-avoiding halo on the real kernels: `group` passes of `variant` sub-iterations between two ghost
    exchanges on ghost zones of depth (variant*group, variant*group - 1), ghost rows advanced redundantly; nsub
    with a remainder and a shorter last group.  Bit-identical to the single-domain run."""
    nx, ny, nsteps = 150, 128, 2
    ref = run_world(1, variant, coupled, nx, ny, nsub, nsteps)[0]
    assert float(ref["u"].abs().max()) > 1e-5
    parts = run_world(world, variant, coupled, nx, ny, nsub, nsteps, group=group)
    for key, dim in (("H", 1), ("A", 1), ("u", 0), ("v", 0), ("s11", 0)):
        got = torch.cat([parts[r][key] for r in range(world)], dim=dim)
        assert torch.equal(got, ref[key]), (key, world, group)


def test_overlap_split_does_not_change_results(gpu):
    a = run_world(3, 2, False, 150, 64, 6, 1, overlap=True)
    b = run_world(3, 2, False, 150, 64, 6, 1, overlap=False)
    for r in range(3):
        for k in a[r]:
            assert torch.equal(a[r][k], b[r][k])


def test_rccl_rehearsals_on_one_gpu(gpu):
    """what can be exercised of the RCCL path with one device: batched ncclSend / ncclRecv to self of the row-block
    views the driver exchanges (tools/rccl_p2p_selftest.py), and the whole multi-rank driver of an interior block
    with both neighbours mapped to the rank itself, every exchange a real RCCL batch
    (tools/rank_share_timing.py --rccl-loopback; timing only, the wrapped values are meaningless)"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_p2p_selftest.py")], env=dict(env, MASTER_PORT="29561"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0 and b"row-block views: ok" in p.stdout, p.stdout.decode()[-2000:]
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "rank_share_timing.py"), "--rccl-loopback", "--k", "2", "8"],
                       env=dict(env, MASTER_PORT="29562", NSDG_SHARE_GRID="512", NSDG_SHARE_NSUB="12"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0 and b"RCCL loopback  world 8" in p.stdout, p.stdout.decode()[-2000:]
