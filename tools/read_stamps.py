"""Diagnostic: runs a few mEVP passes at 2048x2048 with the -DNSDG_STAMPS build (tools/ab_build.sh stamps -DNSDG_STAMPS)
and prints the shader cycles per march step spent in each phase of mevp_fused2_kernel (median over sampled waves)."""
import ctypes
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.environ["NSDG_LIB"] = os.path.join(root, "nextsimdg_amd", "lib", "alt", sys.argv[1] if len(sys.argv) > 1 else "stamps", "libnsdg.so")
VARIANT = int(sys.argv[2]) if len(sys.argv) > 2 else None  # default: the library default
import numpy as np
import torch

from nextsimdg_amd import abi, rowblock, synthetic

nx = 2048
ny = int(os.environ.get("NSDG_STAMPS_NY", "2048"))  # 273: the local array of an 8-way row block (strips of 10 rows, 15 march steps per wave)
L, dt, nsub = 512e3, 120.0, 12
dev = torch.device("cuda:0")
ctx = abi.Context(dev)
if VARIANT is not None:
    ctx.set_mevp_variant(VARIANT)
bt = synthetic.BoxTest(nx, ny, L)
alpha = bt.subcycle_parameters(dt)["alpha"]
ctx.set_mevp_params(ctx.mevp_default_params(alpha=alpha, beta=alpha))
blk = rowblock.RowBlock(nx, ny, 0, 1)
core = rowblock.DynamicsCore(ctx, blk, L / nx, L / ny, dt, nsub, dev)
H, A = bt.dg_fields()
uo, vo = bt.ocean()
ua, va = bt.wind(0.0)
core.load_global(H, A, uo, vo, ua, va)
for _ in range(2):
    core.step()
torch.cuda.synchronize()
lib = abi.load_library()
out = (ctypes.c_uint * (64 * 16))()
if ctx.mevp_variant == 4:
    # one record per (sampled workgroup, pipeline stage): the four waves of a workgroup printed side by side
    rc = lib.nsdg_debug_read_stamps4(out)
    a = np.array(out[:], dtype=np.float64).reshape(16, 4, 16)
    a = a[a[:, 0, 12] > 0]
    order = [(0, "loop overhead"), (1, "Q0 inputs: LDS reads / fetched values, node gather"), (2, "  barrier (wave 1)"), (3, "Q1 projected stress"),
             (9, "   requests: P (+ u, v in stage 0) of the next row"), (10, "   relaxation (+ request: stress of the next row in stage 0)"), (4, "  barrier (wave 2)"),
             (5, "Q2 contributions, node updates, carries"), (11, "   request: nodal coefficients of the next row"), (6, "  barrier (wave 3)"),
             (7, "Q3 outputs: LDS writes / global stores"), (8, "  barrier (wave 0)")]
    per = a[:, :, :12] / a[:, :, 12:13]
    med = np.median(per, axis=0)  # [stage, phase]
    print("variant 4: workgroups sampled %d, march steps %s; cycles per march step, stages 0..3" % (len(a), sorted(set(a[:, 0, 12].astype(int)))))
    for k, name in order:
        print("%-62s %s" % (name, "  ".join("%7.0f" % med[s, k] for s in range(4))))
    print("%-62s %s" % ("sum", "  ".join("%7.0f" % med[s].sum() for s in range(4))))
    clk = a[:, :, 13] / (a[:, :, 14] * 10e-9) / 1e9
    print("in-kernel shader clock: median %.2f GHz (min %.2f, max %.2f); march of one workgroup %.3f ms" % (np.median(clk), clk.min(), clk.max(), np.median(a[:, :, 14]) * 10e-6))
    sys.exit(0)
if ctx.mevp_variant == 3:
    rc = lib.nsdg_debug_read_stamps3(out)
    nph, isteps, icyc, irt = 11, 11, 12, 13
    names = ["loop overhead", "A: loads issued (addressing)", "A: stress update", "A: nodal contributions + node updates + carries",
             "B: P loads issued, node gather", "B: stress update", "B: contributions, node updates, parking in LDS",
             "C: LDS reads, P/coefficient loads issued, node gather", "C: stress update", "C: stress stores", "C: contributions, node updates, stores"]
else:
    rc = lib.nsdg_debug_read_stamps(out)
    nph, isteps, icyc, irt = 9, 9, 10, 11
    names = ["step start -> A loads issued (B tail of previous step included)", "A: loads issued (addressing)", "A: stress update", "A: nodal contributions",
             "A: node updates + carries", "B: stress update", "B: stress stores", "B: nodal contributions", "B: node updates + stores"]
a = np.array(out[:], dtype=np.float64).reshape(64, 16)
a = a[a[:, isteps] > 0]
steps = a[:, isteps:isteps + 1]
per = a[:, :nph] / steps
print("variant %d: waves sampled %d, march steps per wave %s" % (ctx.mevp_variant, len(a), sorted(set(a[:, isteps].astype(int)))))
med = np.median(per, axis=0)
for k in range(nph):
    print("%-70s %8.0f cycles/step  (min %6.0f max %6.0f)" % (names[k], med[k], per[:, k].min(), per[:, k].max()))
print("%-70s %8.0f" % ("sum", med.sum()))
clk = a[:, icyc] / (a[:, irt] * 10e-9) / 1e9
print("in-kernel shader clock (s_memtime / s_memrealtime): median %.2f GHz (min %.2f, max %.2f); march of one wave %.3f ms"
      % (np.median(clk), clk.min(), clk.max(), np.median(a[:, irt]) * 10e-6))
