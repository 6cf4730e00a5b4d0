"""Diagnostic: who waits for whom in the stage-per-wave mEVP pipeline (csrc/mevp_fused4.hip).  Needs the -DNSDG_P2P_SPINSTAT build:
    bash tools/ab_build.sh spin -DNSDG_P2P_SPINSTAT && python tools/p2p_spinstat.py [n=2048] [passes=30]
prints, per stage, the polls per row spent waiting for the previous stage's hand-over, for a free hand-over slot and (loader) for
a free ring slot.  A poll is an s_sleep 1 plus two LDS reads (~0.15 us); the stage the others wait for is the one that waits least."""
import ctypes
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["NSDG_LIB"] = os.path.join(root, "nextsimdg_amd", "lib", "alt", "spin", "libnsdg.so")
sys.path.insert(0, root)
import torch

from nextsimdg_amd import abi, rowblock, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda:0")
ctx = abi.Context(dev)
bt = synthetic.BoxTest(n, n)
ctx.set_mevp_params(ctx.mevp_default_params(**bt.subcycle_parameters(120.0)))
core = rowblock.DynamicsCore(ctx, rowblock.RowBlock(n, n), bt.hx, bt.hy, 120.0, 4 * passes, dev)
H, A = bt.dg_fields()
uo, vo = bt.ocean()
ua, va = bt.wind(0.0)
core.load_global(H, A, uo, vo, ua, va)
core.step()  # warm-up (and a non-trivial velocity)
out = (ctypes.c_ulonglong * 48)()
abi._lib.nsdg_debug_p2p_spinstat.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
abi._lib.nsdg_debug_p2p_spinstat(out)  # reset
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
core._set_grid()
core.prepare()
e0.record(ctx.stream)
core.subcycle()
e1.record(ctx.stream)
torch.cuda.synchronize()
assert abi._lib.nsdg_debug_p2p_spinstat(out) == 0
ms = e0.elapsed_time(e1) / passes
print("%d x %d, %d passes of four sub-iterations, %.4f ms per pass" % (n, n, passes, ms))
print("stage  rows/wave-pass   polls per row: previous stage's hand-over / free slot / free ring slot")
for s in range(4):
    v = [out[4 * s + k] for k in range(4)]
    rows = max(v[3], 1)
    print("  %d    %10.1f      %8.3f  %8.3f  %8.3f" % (s, v[3] / passes, v[0] / rows, v[1] / rows, v[2] / rows))
names = ("inputs (wait for the previous stage, LDS reads)", "projected stress", "ring / stress hand-over, requests of P, u, v", "relaxation, stress request",
         "contributions, node updates (wait for c)", "request of c", "outputs (slot wait, LDS writes / stores)")
print("shader cycles per row (s_memtime stamps, fenced: the stamped build is slower than the product):")
print("phase                                                       " + "".join("  stage %d" % s for s in range(4)))
for k, name in enumerate(names):
    print("%-60s" % name + "".join("  %7.0f" % (out[16 + 8 * s + k] / max(out[4 * s + 3], 1)) for s in range(4)))
print("%-60s" % "sum" + "".join("  %7.0f" % (sum(out[16 + 8 * s + k] for k in range(8)) / max(out[4 * s + 3], 1)) for s in range(4)))
