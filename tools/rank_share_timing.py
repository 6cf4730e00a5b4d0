"""Diagnostic: GPU time of ONE rank's share of the 2048x2048 dynamics step for a world of W row blocks, measured
on a single GPU with the ghost exchanges replaced by no-ops (the values in the ghost rows go stale, only the
timing is meaningful).  Gives the compute-side bound on the strong-scaling speed-up that the 8-GPU run of the
driver can reach: T(1 block) / T(share).  usage: python tools/rank_share_timing.py [--native [--graph]] [--halo native|torch] [--rccl-loopback] [--no-overlap] [--k K,...] [W ...]"""
import gc
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch

from nextsimdg_amd import abi, rowblock, synthetic


class NullExchanger(rowblock.HaloExchanger):
    def _start(self, plan):
        return None

    def _finish(self, handle):
        return


GRID = int(os.environ.get("NSDG_SHARE_GRID", "2048"))  # smaller grid / fewer sub-iterations for the smoke test
NSUB = int(os.environ.get("NSDG_SHARE_NSUB", "120"))


def check_loopback_values(core):
    """one ghost exchange of an interior block whose neighbours are the rank itself, on distinct random data:
    what was sent upwards must arrive as the ghost rows from below and vice versa, block by block -- this pins the
    pack / send / receive / unpack ordering of the exchanger on the device"""
    g = torch.Generator(device="cuda").manual_seed(17)
    for f in core.sb + [core.ub, core.vb]:
        f.copy_(torch.rand(f.shape, dtype=f.dtype, device=f.device, generator=g))
    core._ghost_exchange_finish(core._ghost_exchange_start())
    torch.cuda.synchronize()
    plans = [p for k, p in core.halo._cache.items() if k[0] == "r"]
    assert plans, "no ghost-zone plan was built"
    n = 0
    for p in plans:
        assert len(p.up_send) == len(p.from_below) and len(p.down_send) == len(p.from_above)
        for a, b in list(zip(p.up_send, p.from_below)) + list(zip(p.down_send, p.from_above)):
            assert a.shape == b.shape and torch.equal(a, b), "loopback exchange delivered wrong values"
            n += a.numel()
    return n


def run(world, kpass, nx=GRID, ny=GRID, nsub=NSUB, steps=3, overlap=True, loopback=False, halo="native", native=False, graph=False):
    dev = torch.device("cuda:0")
    ctx = abi.Context(dev)
    L, dt = 512e3, 120.0
    bt = synthetic.BoxTest(nx, ny, L)
    ctx.set_mevp_params(ctx.mevp_default_params(**bt.subcycle_parameters(dt, mode=os.environ.get("NSDG_SHARE_SUBCYCLE", "adaptive"))))  # the hosts' policy: adaptive alpha / beta (keep_alpha: round 5)
    if os.environ.get("NSDG_SHARE_VARIANT"):  # A/B of the mEVP kernel variants
        ctx.set_mevp_variant(int(os.environ["NSDG_SHARE_VARIANT"]))
    rank = world // 2
    v = min(ctx.mevp_variant, 4)  # sub-iterations per kernel pass
    depth = (v * kpass, v * kpass - 1)
    blk = rowblock.RowBlock(nx, ny, rank, world, *depth)
    exchanger = None
    if world > 1 and loopback:  # real RCCL send/recv, both neighbours = this rank (periodic wrap: values meaningless)
        if blk.below is None or blk.above is None:
            raise SystemExit("--rccl-loopback needs a block with two neighbours (world >= 3)")
        if halo == "native":  # pack kernel + ncclSend/ncclRecv group + unpack kernel behind the C ABI
            exchanger = rowblock.NativeHaloExchanger(ctx, blk, loopback=True)
        else:  # torch.distributed P2P ops
            blk.below = blk.above = 0
            exchanger = rowblock.HaloExchanger(blk, loopback=True)
    elif world > 1:
        exchanger = NullExchanger(blk)
    if native and world > 1 and not isinstance(exchanger, rowblock.NativeHaloExchanger):
        raise SystemExit("--native needs --rccl-loopback with the native halo (or world 1)")
    core = rowblock.DynamicsCore(ctx, blk, L / nx, L / ny, dt, nsub, dev, exchanger=exchanger, overlap=overlap, native=native, use_graph=graph)
    H, A = bt.dg_fields()
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    core.load_global(H, A, uo, vo, ua, va)
    core.step()
    torch.cuda.synchronize()
    gc.collect()  # contexts and plans of earlier runs die here (hipFree synchronises), not inside the timed loop
    t0 = time.perf_counter()
    for _ in range(steps):
        core.step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    res = (time.perf_counter() - t0) / steps * 1e3, host / steps * 1e3, blk.ny
    if world > 1 and loopback and core.per_pass >= 2:
        print("loopback values: ok (%d doubles compared)" % check_loopback_values(core), flush=True)
    return res


if __name__ == "__main__":
    ks = (1, 4)
    args = sys.argv[1:]
    overlap, loopback, halo, native, graph = True, False, "native", False, False
    if args and args[0] == "--native":  # sub-cycle and transport as one C call each (csrc/rowblock.hip)
        native = True
        args = args[1:]
    if args and args[0] == "--graph":  # ... with the launches between two exchanges replayed as one hipGraph
        graph = True
        args = args[1:]
    if args and args[0] == "--halo":  # native (default): exchanges behind the C ABI; torch: torch.distributed P2P ops
        halo = args[1]
        args = args[2:]
    if args and args[0] == "--rccl-loopback":  # exchanges are real RCCL send/recv to self (one rank on one GPU)
        torch.cuda.set_device(0)
        if halo == "torch":
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29551")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        loopback = True
        args = args[1:]
    if args and args[0] == "--no-overlap":  # one launch per pass, the exchange would follow it un-overlapped
        overlap = False
        args = args[1:]
    if args and args[0] == "--k":
        ks = tuple(int(x) for x in args[1].split(","))
        args = args[2:]
    worlds = [int(a) for a in args] or [1, 2, 4, 8]
    base = None
    for w in worlds:
        for k in ((1,) if w == 1 else ks):
            ms, host_ms, rows = run(w, k, overlap=overlap, loopback=loopback, halo=halo, native=native, graph=graph)
            base = ms if w == 1 else base
            print(("native driver%s  " % (" + graphs" if graph else "") if native else "") + ("RCCL loopback (%s halo)  " % halo if loopback else "") + ("" if overlap else "no-overlap  ") + "world %d  passes/exchange %d  local rows %4d  step %7.3f ms  (host issue %6.3f ms)  speed-up bound %s"
                  % (w, k, rows, ms, host_ms, "%.2f" % (base / ms) if base else "-"), flush=True)
    if loopback and halo == "torch":
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()
