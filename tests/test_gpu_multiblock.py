"""The multi-rank driver on ONE GPU: every "rank" is a thread with its own nsdg context and its own row
block; ghost rows travel through an in-process mailbox instead of RCCL (tests/thread_ranks.py).  This runs the
real HIP kernels on real ghost-row layouts (depth (2,1) for the two-iterations-per-pass kernel, (1,1) otherwise),
with the same DynamicsCore / RowBlock code the bench uses, and must reproduce the single-domain run bit for bit.
(What it cannot cover is RCCL between two devices: there is one GPU per test box.)"""
import pytest
import torch

from thread_ranks import run_world

pytestmark = pytest.mark.gpu


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
    (tools/rank_share_timing.py --rccl-loopback), followed by one exchange of distinct random data whose received
    ghost rows must equal the rows sent in the opposite direction, block by block"""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "rccl_p2p_selftest.py")], env=dict(env, MASTER_PORT="29561"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0 and b"row-block views: ok" in p.stdout, p.stdout.decode()[-2000:]
    for halo in ("native", "torch"):  # the exchange behind the C ABI (csrc/halo.hip) and the torch.distributed one
        p = subprocess.run([sys.executable, os.path.join(root, "tools", "rank_share_timing.py"), "--halo", halo, "--rccl-loopback", "--k", "2", "8"],
                           env=dict(env, MASTER_PORT="29562", NSDG_SHARE_GRID="512", NSDG_SHARE_NSUB="12"),
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        out = p.stdout.decode()
        assert p.returncode == 0 and ("RCCL loopback (%s halo)  world 8" % halo) in out and "loopback values: ok" in out, out[-2000:]
