"""Rehearsal of the halo exchange's RCCL usage on ONE GPU: torch.distributed 'nccl' (= RCCL) with one rank,
batched send/recv to self of the same kind of tensor views the row-block driver exchanges (contiguous row
blocks of node lattices and of tiled stress arrays, several ops per batch, posted while a kernel runs).
What it cannot show is two devices talking over xGMI -- that is the driver's 8-GPU run."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
ok = True
try:
    nx, ny = 300, 40
    u = torch.randn(2 * ny + 1, 2 * nx + 1, dtype=torch.float64, device=dev)
    s = torch.randn(ny, (nx + 63) // 64, 8 * 64, dtype=torch.float64, device=dev)
    u2, s2 = torch.zeros_like(u), torch.zeros_like(s)
    big = torch.randn(4096, 4096, device=dev)
    for it in range(3):
        _ = big @ big  # something in flight on the compute stream while the batch is posted
        ops = [dist.P2POp(dist.isend, s[ny - 12:ny], 0), dist.P2POp(dist.isend, u[2 * ny - 24:2 * ny], 0),
               dist.P2POp(dist.irecv, s2[0:12], 0), dist.P2POp(dist.irecv, u2[0:24], 0)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        torch.cuda.synchronize()
        ok = ok and torch.equal(s2[0:12], s[ny - 12:ny]) and torch.equal(u2[0:24], u[2 * ny - 24:2 * ny])
        s2.zero_(), u2.zero_()
    dist.barrier()
finally:
    dist.destroy_process_group()
print("RCCL self send/recv of row-block views:", "ok" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
