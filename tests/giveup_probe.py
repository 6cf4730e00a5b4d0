"""Run by tests/test_gpu_giveup.py in a process of its own with NSDG_LIB = the diagnostic build `giveup` (nextsimdg_amd/build.py: every
wait of the mEVP pipelines gives up after one poll).  Prints one JSON line: what a host sees when a wait has given up."""
import json
import sys

import numpy as np
import torch

from nextsimdg_amd import abi, synthetic


def main(variant):
    dev = torch.device("cuda:0")
    ctx = abi.Context(dev)
    nx, ny = 256, 96
    bt = synthetic.BoxTest(nx, ny)
    ctx.set_grid(nx, ny, bt.hx, bt.hy)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    H, A = bt.dg_fields()
    dH, dA = d(H), d(A)
    shape = (2 * ny + 1, 2 * nx + 1)
    cgh, cga = torch.zeros(shape, dtype=torch.float64, device=dev), torch.zeros(shape, dtype=torch.float64, device=dev)
    ctx.dg_to_cg(dH, cgh)
    ctx.dg_to_cg(dA, cga)
    pg = ctx.private_zeros(9, ny, nx, dev)
    ctx.ice_strength(dH, dA, pg, 0, ny)
    uo, vo = [d(a) for a in bt.ocean()]
    ua, va = [d(a) for a in bt.wind(0.0)]
    tax, tay = torch.zeros_like(uo), torch.zeros_like(uo)
    ctx.wind_stress(ua, va, tax, tay)
    out = {}

    def subcycle(v):
        ctx.set_mevp_variant(v)
        u, w = torch.zeros(shape, dtype=torch.float64, device=dev), torch.zeros(shape, dtype=torch.float64, device=dev)
        s = [ctx.private_zeros(8, ny, nx, dev) for _ in range(3)]
        scratch = torch.zeros(10 * u.numel() + 3 * s[0].numel(), dtype=torch.float64, device=dev)
        ctx.mevp_subcycle(120.0, 16, s, u, w, u.clone(), w.clone(), tax, tay, uo, vo, cgh, cga, pg, scratch)
        return u, s

    ref_u, ref_s = subcycle(1)  # the single-iteration kernel has no pipeline: the reference of this run
    ctx.synchronize()
    out["status_before"] = "ok"
    u, s = subcycle(variant)  # the launches return; their waits give up
    out["launch_returned"] = True
    try:
        ctx.synchronize()
        out["synchronize"] = "ok"
    except abi.NsdgError as e:
        out["synchronize"] = str(e)
    try:
        subcycle(variant)
        out["next_subcycle"] = "ok"
    except abi.NsdgError as e:
        out["next_subcycle"] = str(e)
    torch.cuda.synchronize()
    out["fields_differ"] = bool(not torch.equal(u, ref_u) or not torch.equal(s[0], ref_s[0]))
    out["given_up"] = ctx.pipeline_waits_given_up()
    try:
        ctx.synchronize()
        out["synchronize_after_health"] = "ok"
    except abi.NsdgError as e:
        out["synchronize_after_health"] = str(e)
    out["given_up_again"] = ctx.pipeline_waits_given_up()
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]))
