"""Error behaviour of the C ABI on a live context: bad arguments and call-sequence errors are reported
through the status code + nsdg_last_error() (-> NsdgError in the Python binding), never ignored."""
import pytest
import torch

from nextsimdg_amd import abi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(gpu):
    c = abi.Context(gpu)
    yield c
    c.close()


def z(*shape):
    return torch.zeros(*shape, dtype=torch.float64, device="cuda")


def test_grid_must_be_set_first(gpu):
    c = abi.Context(gpu)
    with pytest.raises(abi.NsdgError, match="nsdg_grid_set was not called"):
        c.dg_to_cg(z(6, 4, 4), z(9, 9))
    c.close()


def test_bad_grid_and_params(ctx):
    for bad in ((0, 4, 1.0, 1.0), (4, -1, 1.0, 1.0), (4, 4, 0.0, 1.0)):
        with pytest.raises(abi.NsdgError):
            ctx.set_grid(*bad)
    with pytest.raises(abi.NsdgError, match="alpha and beta"):
        ctx.set_mevp_params(ctx.mevp_default_params(alpha=0.0))
    with pytest.raises(abi.NsdgError):
        ctx.set_mevp_variant(5)
    with pytest.raises(abi.NsdgError):
        ctx.set_mevp_occupancy(4)


def test_row_ranges_aliasing_and_sequence(ctx):
    nx, ny = 70, 12
    ctx.set_grid(nx, ny, 1.0, 1.0)
    s = [ctx.private_zeros(8, ny, nx, "cuda") for _ in range(3)]
    so = [torch.zeros_like(x) for x in s]
    pg = ctx.private_zeros(9, ny, nx, "cuda")
    u, v, un, vn = (z(2 * ny + 1, 2 * nx + 1) for _ in range(4))
    packed = z(8 * u.numel())
    fresh = abi.Context(ctx.device)
    fresh.set_grid(nx, ny, 1.0, 1.0)
    with pytest.raises(abi.NsdgError, match="nsdg_mevp_pack_nodal was not called"):
        fresh.mevp_iterate(0, 0, ny, s, so, (u, v), (un, vn), packed, pg)
    fresh.close()
    nodal = [(u, v), (u, v), (u, v), u, v]
    ctx.mevp_pack_nodal(120.0, *nodal, packed)
    with pytest.raises(abi.NsdgError, match="row range|need 0 <= k0"):
        ctx.mevp_iterate(0, 0, ny + 1, s, so, (u, v), (un, vn), packed, pg)
    with pytest.raises(abi.NsdgError, match="k0 == j0 - 1"):
        ctx.mevp_iterate(0, 3, ny, s, so, (u, v), (un, vn), packed, pg)
    with pytest.raises(abi.NsdgError, match="must not alias"):
        ctx.mevp_iterate(0, 0, ny, s, s, (u, v), (un, vn), packed, pg)
    with pytest.raises(abi.NsdgError, match="must not alias"):
        ctx.mevp_iterate(0, 0, ny, s, so, (u, v), (u, vn), packed, pg)
    ctx.set_mevp_variant(2)
    with pytest.raises(abi.NsdgError, match="two ghost rows"):
        ctx.mevp_iterate2(1, ny, s, so, (u, v), (un, vn), packed, pg)
    ctx.set_mevp_variant(1)
    with pytest.raises(abi.NsdgError, match="variant 2"):
        ctx.mevp_iterate2(0, ny, s, so, (u, v), (un, vn), packed, pg)
    with pytest.raises(abi.NsdgError, match="variant 3"):
        ctx.mevp_iterate3(0, ny, s, so, (u, v), (un, vn), packed, pg)
    ctx.set_mevp_variant(3)
    with pytest.raises(abi.NsdgError, match="three ghost rows"):
        ctx.mevp_iterate3(2, ny, s, so, (u, v), (un, vn), packed, pg)
    with pytest.raises(abi.NsdgError, match="two ghost rows above"):
        ctx.mevp_iterate3(0, ny - 1, s, so, (u, v), (un, vn), packed, pg)
    with pytest.raises(abi.NsdgError, match="variant 4"):
        ctx.mevp_iterate4(0, ny, s, so, (u, v), (un, vn), packed, pg)
    ctx.set_mevp_variant(4)
    with pytest.raises(abi.NsdgError, match="four ghost rows"):
        ctx.mevp_iterate4(3, ny, s, so, (u, v), (un, vn), packed, pg)
    with pytest.raises(abi.NsdgError, match="three ghost rows above"):
        ctx.mevp_iterate4(0, ny - 2, s, so, (u, v), (un, vn), packed, pg)
    with pytest.raises(abi.NsdgError, match="disjoint"):
        ctx.mevp_iterate4_pair((0, ny), (4, ny), s, so, (u, v), (un, vn), packed, pg)
    with pytest.raises(abi.NsdgError, match="must not alias"):
        ctx.mevp_iterate4(0, ny, s, s, (u, v), (un, vn), packed, pg)
    ctx.mevp_iterate3(0, ny, s, so, (u, v), (un, vn), packed, pg)  # a variant-4 context still serves the three-iteration pass (remainders)
    ctx.set_mevp_variant(abi.DEFAULT_MEVP_VARIANT)
    with pytest.raises(abi.NsdgError, match="order must be"):
        ctx.prepare_advection(3, u, v, z(6, ny, nx), z(6, ny, nx), z(3, ny, nx + 1), z(3, ny + 1, nx))
    with pytest.raises(abi.NsdgError, match="ncoef"):
        ctx.dg_to_cg(z(4, ny, nx), u)
    # wrong dtype / device never reaches the library
    with pytest.raises(abi.NsdgError, match="float64 CUDA"):
        ctx.wind_stress(u.float(), v, un, vn)
    with pytest.raises(abi.NsdgError, match="float64 CUDA"):
        ctx.wind_stress(u.cpu(), v, un, vn)
    # empty row ranges are no-ops
    ctx.mevp_iterate2(4, 4, s, so, (u, v), (un, vn), packed, pg)
    ctx.transport_stage(2, 5, 5, 1.0, 0.0, 1.0, [z(6, ny, nx)], [z(6, ny, nx)], [z(6, ny, nx)],
                        (z(6, ny, nx), z(6, ny, nx), z(3, ny, nx + 1), z(3, ny + 1, nx)))
    torch.cuda.synchronize()


def test_parameter_or_grid_change_invalidates_the_packed_coefficients(ctx):
    """the launch constants of the velocity update (K1, K2 from beta, rho_ice and dt) belong to the packing: changing the
    mEVP parameters or the grid between nsdg_mevp_pack_nodal and an iterate must be an error, not a silent mix of two
    parameter sets; repacking makes the call valid again"""
    nx, ny = 70, 12
    ctx.set_grid(nx, ny, 1.0, 1.0)
    s = [ctx.private_zeros(8, ny, nx, "cuda") for _ in range(3)]
    so = [torch.zeros_like(x) for x in s]
    pg = ctx.private_zeros(9, ny, nx, "cuda")
    u, v, un, vn = (z(2 * ny + 1, 2 * nx + 1) for _ in range(4))
    packed = z(8 * u.numel())
    nodal = [(u, v), (u, v), (u, v), u, v]
    ctx.mevp_pack_nodal(120.0, *nodal, packed)
    ctx.mevp_iterate(0, 0, ny, s, so, (u, v), (un, vn), packed, pg)  # fine
    ctx.set_mevp_params(ctx.mevp_default_params(beta=777.0))
    for call in (lambda: ctx.mevp_iterate(0, 0, ny, s, so, (u, v), (un, vn), packed, pg),
                 lambda: ctx.mevp_iterate3(0, ny, s, so, (u, v), (un, vn), packed, pg),
                 lambda: ctx.mevp_velocity(0, ny, s, (u, v), (un, vn), packed)):
        with pytest.raises(abi.NsdgError, match="pack_nodal was not called"):
            call()
    ctx.mevp_pack_nodal(120.0, *nodal, packed)
    ctx.mevp_iterate3(0, ny, s, so, (u, v), (un, vn), packed, pg)
    ctx.set_grid(nx, ny, 2.0, 1.0)  # another cell size: the coefficients belong to the old grid
    with pytest.raises(abi.NsdgError, match="pack_nodal was not called"):
        ctx.mevp_iterate(0, 0, ny, s, so, (u, v), (un, vn), packed, pg)
    ctx.set_grid(nx, ny, 2.0, 1.0)  # setting the same grid again changes nothing ...
    ctx.mevp_pack_nodal(120.0, *nodal, packed)
    ctx.set_grid(nx, ny, 2.0, 1.0)  # ... and does not invalidate a packing
    ctx.mevp_iterate(0, 0, ny, s, so, (u, v), (un, vn), packed, pg)
    # another SHAPE with the same cell size: the packed coefficients (and the arrays) belong to the old local array
    ctx.set_grid(nx, ny - 2, 2.0, 1.0)
    with pytest.raises(abi.NsdgError, match="pack_nodal was not called"):
        ctx.mevp_iterate(0, 0, ny - 2, s, so, (u, v), (un, vn), packed, pg)
    ctx.set_grid(nx - 6, ny, 2.0, 1.0)
    with pytest.raises(abi.NsdgError, match="pack_nodal was not called"):
        ctx.mevp_iterate3(0, ny, s, so, (u, v), (un, vn), packed, pg)
    ctx.set_grid(nx, ny, 2.0, 1.0)  # back to the first shape: still invalid until repacked
    with pytest.raises(abi.NsdgError, match="pack_nodal was not called"):
        ctx.mevp_iterate(0, 0, ny, s, so, (u, v), (un, vn), packed, pg)
    ctx.set_mevp_params(ctx.mevp_default_params())
    torch.cuda.synchronize()
