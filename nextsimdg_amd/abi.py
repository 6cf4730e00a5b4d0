"""ctypes binding of the C ABI in include/nsdg.h (libnsdg.so) for Python callers (tests, bench,
multi-rank driver).  PyTorch is used only as the owner of device memory and streams: every call
below passes raw device pointers through the extern "C" boundary.

There is deliberately NO fallback: if the HIP library is missing or fails to load, importing the
handle raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
GIVEN_UP_UNSEEN = 0  # waits of the mEVP pipeline that gave up on contexts that were closed without anybody asking (Context.close)
LIB_PATH = os.environ.get("NSDG_LIB", os.path.join(HERE, "lib", "libnsdg.so"))  # NSDG_LIB: A/B builds of the same ABI

c_double_p = C.POINTER(C.c_double)
NDIAG = 15
DIAG = ["rho", "qa", "qw", "qi", "cspec", "tau", "hi", "hs", "cnew", "qia", "qio", "subl", "dqdt", "hifroms", "qow"]
STATE = ["hice", "cice", "hsnow", "tice0"]
FORCING = ["sst", "sss", "tair", "tdew", "slp", "qsw", "qlw", "mld", "snowfall", "wind"]
ALBEDO = {"smu": 0, "smu2": 1, "ccsm": 2}
FREEZING = {"linear": 0, "unesco": 1}


class ColumnParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "drag_ocean_q", "drag_ocean_t", "drag_ice_t", "ocean_albedo", "i0", "min_conc", "min_thick",
        "ks", "h0", "phi_m", "ccsm_ice_albedo", "ccsm_snow_albedo")] + [
        ("flooding", C.c_int32), ("albedo_kind", C.c_int32), ("freezing_kind", C.c_int32), ("reserved", C.c_int32)]


class MevpParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in (
        "rho_ice", "rho_atm", "rho_ocean", "c_atm", "c_ocean", "pstar", "compaction", "delta_min", "fc",
        "alpha", "beta", "h_min", "min_conc", "min_thick", "aevp_c", "aevp_alpha_min")]


SUBCYCLE_ADAPTIVE, SUBCYCLE_KEEP_ALPHA, SUBCYCLE_KEEP_DELTA_MIN, SUBCYCLE_ADAPTIVE_CONVERGED = 0, 1, 2, 3  # modes of nsdg_mevp_stable_params


class FieldBounds(C.Structure):
    """nsdg_field_bounds: closure of a transport step for one advected field"""
    _fields_ = [("lo", C.c_double), ("hi", C.c_double), ("cap_mean", C.c_int32), ("reserved", C.c_int32)]


# the closure of the dynamics' two advected fields (include/nsdg.h "INPUT DOMAIN AND CLOSURE"): mean thickness H >= 0; concentration
# 0 <= A <= 1 with the cell mean capped at 1 (ridging: further convergence raises the true thickness H / A)
H_A_BOUNDS = ((0.0, float("inf"), False), (0.0, 1.0, True))


class HaloSeg(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("count", C.c_int64)]


class HaloStats(C.Structure):
    _fields_ = [("exchanges", C.c_int64), ("untimed", C.c_int64), ("ms", C.c_double), ("bytes_sent", C.c_int64), ("bytes_received", C.c_int64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


COMM_ID_BYTES = 128
RB_MAX_FIELDS = 4


class RbMevpDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nx", "ny", "j0", "j1", "depth_below", "depth_above", "rank_below", "rank_above", "nsub",
                                          "overlap", "use_graph", "reserved")] + [
        (n, C.c_void_p * 2) for n in ("s11", "s12", "s22", "u", "v")] + [("packed", C.c_void_p), ("pg", C.c_void_p)]


class RbTransportDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nx", "ny", "j0", "j1", "depth_below", "depth_above", "rank_below", "rank_above", "order",
                                          "nfields")] + [
        (n, C.c_void_p * RB_MAX_FIELDS) for n in ("phi", "t1", "t2")] + [(n, C.c_void_p) for n in ("vx_dg", "vy_dg", "un_x", "un_y")] + [
        ("own_bounds", C.c_int32), ("nbounds", C.c_int32), ("bounds", FieldBounds * RB_MAX_FIELDS)]


class NsdgError(RuntimeError):
    pass


# name -> (restype, argtypes); kept in one table so tests can check every declared symbol exists
VP = C.c_void_p
I32, I64, D = C.c_int32, C.c_int64, C.c_double
DEFAULT_TRANSPORT_VARIANT = 2  # nsdg_ctx_create's default transport stage kernel (csrc/nsdg_ctx.hip)
DEFAULT_MEVP_VARIANT = 4  # nsdg_ctx_create's default (NSDG_MEVP_DEFAULT_VARIANT): four sub-iterations per kernel pass (csrc/mevp_fused4.hip)

SYMBOLS = {
    "nsdg_abi_version": (C.c_int, []),
    "nsdg_last_error": (C.c_char_p, []),
    "nsdg_ctx_create": (C.c_int, [C.c_int, VP, C.POINTER(VP)]),
    "nsdg_ctx_destroy": (C.c_int, [VP]),
    "nsdg_ctx_synchronize": (C.c_int, [VP]),
    "nsdg_copy_f64": (C.c_int, [VP, VP, VP, I64]),
    "nsdg_column_default_params": (None, [C.POINTER(ColumnParams)]),
    "nsdg_column_params_set": (C.c_int, [VP, C.POINTER(ColumnParams)]),
    "nsdg_column_step": (C.c_int, [VP, I64, D] + [VP] * 16),
    "nsdg_mevp_default_params": (None, [C.POINTER(MevpParams)]),
    "nsdg_mevp_params_set": (C.c_int, [VP, C.POINTER(MevpParams)]),
    "nsdg_mevp_stable_params": (C.c_int, [C.POINTER(MevpParams), I32, D, D]),
    "nsdg_mevp_creep_percent_per_day": (C.c_double, [C.POINTER(MevpParams)]),
    "nsdg_tiled_len": (C.c_int64, [I32, I32, I32]),
    "nsdg_grid_set": (C.c_int, [VP, I32, I32, D, D]),
    "nsdg_mevp_variant_set": (C.c_int, [VP, I32]),
    "nsdg_prepare_advection": (C.c_int, [VP, I32] + [VP] * 6),
    "nsdg_transport_variant_set": (C.c_int, [VP, I32, I32]),
    "nsdg_transport_stage": (C.c_int, [VP, I32, I32, I32, D, D, D, I32, C.POINTER(VP), C.POINTER(VP), C.POINTER(VP)] + [VP] * 4),
    "nsdg_transport_step": (C.c_int, [VP, I32, D, I32, C.POINTER(VP)] + [VP] * 5),
    "nsdg_transport_step_oop": (C.c_int, [VP, I32, D, I32, C.POINTER(VP), C.POINTER(VP)] + [VP] * 4),
    "nsdg_transport_step_oop_rows": (C.c_int, [VP, I32, I32, I32, D, I32, C.POINTER(VP), C.POINTER(VP)] + [VP] * 4),
    "nsdg_transport_bounds_set": (C.c_int, [VP, I32, C.POINTER(FieldBounds)]),
    "nsdg_transport_limit": (C.c_int, [VP, I32, I32, I32, I32, C.POINTER(VP)]),
    "nsdg_dg_to_cg": (C.c_int, [VP, I32, VP, VP]),
    "nsdg_ice_strength": (C.c_int, [VP, I32, I32, VP, VP, VP]),
    "nsdg_boxtest_forcing": (C.c_int, [VP, D, D, VP, VP, VP, VP]),
    "nsdg_block_set": (C.c_int, [VP, I32, I32]),
    "nsdg_column_forcing": (C.c_int, [VP, I32, D] + [VP] * 7),
    "nsdg_column_wind": (C.c_int, [VP, VP, VP, VP]),
    "nsdg_wind_stress": (C.c_int, [VP, I64, VP, VP, VP, VP]),
    "nsdg_mevp_stress": (C.c_int, [VP, I32, I32] + [VP] * 6),
    "nsdg_mevp_pack_nodal": (C.c_int, [VP, D] + [VP] * 9),
    "nsdg_mevp_prepare": (C.c_int, [VP, D] + [VP] * 9),
    "nsdg_mevp_velocity": (C.c_int, [VP, I32, I32] + [VP] * 8),
    "nsdg_mevp_iterate": (C.c_int, [VP, I32, I32, I32] + [VP] * 12),
    "nsdg_mevp_iterate2": (C.c_int, [VP, I32, I32] + [VP] * 12),
    "nsdg_mevp_iterate3": (C.c_int, [VP, I32, I32] + [VP] * 12),
    "nsdg_mevp_iterate3_pair": (C.c_int, [VP, I32, I32, I32, I32] + [VP] * 12),
    "nsdg_mevp_iterate4": (C.c_int, [VP, I32, I32] + [VP] * 12),
    "nsdg_mevp_iterate4_pair": (C.c_int, [VP, I32, I32, I32, I32] + [VP] * 12),
    "nsdg_mevp_pipeline_health": (C.c_int, [VP, C.POINTER(C.c_uint32)]),
    "nsdg_mevp_strip_rows_set": (C.c_int, [VP, I32]),
    "nsdg_mevp_occupancy_set": (C.c_int, [VP, I32]),
    "nsdg_mevp_subcycle": (C.c_int, [VP, D, I32] + [VP] * 15),
    "nsdg_comm_unique_id": (C.c_int, [VP]),
    "nsdg_comm_init": (C.c_int, [VP, I32, I32, VP]),
    "nsdg_comm_init_local": (C.c_int, [VP, I64, I32, I32]),
    "nsdg_comm_finalize": (C.c_int, [VP]),
    "nsdg_comm_rank": (C.c_int, [VP, C.POINTER(I32), C.POINTER(I32)]),
    "nsdg_comm_deadline_set": (C.c_int, [VP, D]),
    "nsdg_comm_simulate_wire": (C.c_int, [VP, D, D]),
    "nsdg_halo_plan_create": (C.c_int, [VP, I32, I32, I32, C.POINTER(HaloSeg), I32, C.POINTER(HaloSeg), I32, C.POINTER(HaloSeg), I32,
                                        C.POINTER(HaloSeg), C.POINTER(VP)]),
    "nsdg_halo_plan_destroy": (C.c_int, [VP]),
    "nsdg_halo_counts": (C.c_int, [VP] + [C.POINTER(I64)] * 4),
    "nsdg_halo_start": (C.c_int, [VP, VP]),
    "nsdg_halo_finish": (C.c_int, [VP, VP]),
    "nsdg_halo_stats_get": (C.c_int, [VP, VP, C.POINTER(HaloStats), I32]),
    "nsdg_rb_mevp_create": (C.c_int, [VP, C.POINTER(RbMevpDesc), C.POINTER(VP)]),
    "nsdg_rb_mevp_destroy": (C.c_int, [VP]),
    "nsdg_rb_mevp_info": (C.c_int, [VP, C.POINTER(I32), C.POINTER(I32)]),
    "nsdg_rb_mevp_run": (C.c_int, [VP, VP, I32, C.POINTER(I32)]),
    "nsdg_rb_mevp_stats": (C.c_int, [VP, VP, C.POINTER(HaloStats), I32]),
    "nsdg_rb_transport_create": (C.c_int, [VP, C.POINTER(RbTransportDesc), C.POINTER(VP)]),
    "nsdg_rb_transport_destroy": (C.c_int, [VP]),
    "nsdg_rb_transport_run": (C.c_int, [VP, VP, D, I32, C.POINTER(I32)]),
    "nsdg_rb_transport_stats": (C.c_int, [VP, VP, C.POINTER(HaloStats), I32]),
}

_lib = None


def load_library(path=LIB_PATH):
    """dlopen libnsdg.so and declare every entry point of include/nsdg.h.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # Device pointers and streams come from PyTorch, so the kernels must be launched through the
    # SAME HIP runtime instance torch uses: import torch first so that its libamdhip64 is the one
    # already mapped when libnsdg.so's dependency on that soname is resolved.
    import torch  # noqa: F401

    if not os.path.exists(path):
        raise NsdgError("HIP library %s is missing: build it with `python -m nextsimdg_amd.build` "
                        "(there is no CPU fallback)" % path)
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


TILE = 64


def stable_mevp_params(p, mode, h, dt):
    """nsdg_mevp_stable_params: the sub-cycle's stability rule (one copy, in the library) applied to the MevpParams `p` in place for cells
    of size h and a model time step dt; mode = SUBCYCLE_ADAPTIVE / SUBCYCLE_ADAPTIVE_CONVERGED / SUBCYCLE_KEEP_ALPHA / SUBCYCLE_KEEP_DELTA_MIN"""
    rc = load_library().nsdg_mevp_stable_params(C.byref(p), int(mode), float(h), float(dt))
    if rc != 0:
        raise NsdgError("nsdg error %d: %s" % (rc, load_library().nsdg_last_error().decode()))
    return p


def creep_percent_per_day(p):
    """strain rate below which the ice creeps instead of staying rigid (= delta_min), in percent per day"""
    return float(load_library().nsdg_mevp_creep_percent_per_day(C.byref(p)))


def tile(a):
    """[nc, ny, nx] coefficient planes -> the tiled layout of include/nsdg.h (stress coefficients and Gauss-point
    ice strength), returned as [ny, ceil(nx/64), nc*64]: per tile of 64 elements the coefficients in pairs
    interleaved by element, (c/2)*128 + 2*l + c%2, an odd last coefficient at (nc/2)*128 + l; padding elements
    are zero"""
    import torch

    nc, ny, nx = a.shape
    ntx = (nx + TILE - 1) // TILE
    b = torch.nn.functional.pad(a, (0, ntx * TILE - nx)).reshape(nc, ny, ntx, TILE)
    npair = nc // 2
    parts = [b[:2 * npair].reshape(npair, 2, ny, ntx, TILE).permute(2, 3, 0, 4, 1).reshape(ny, ntx, npair * 2 * TILE)]
    if nc % 2:
        parts.append(b[nc - 1].reshape(ny, ntx, TILE))
    return torch.cat(parts, dim=2).contiguous()


def untile(t, nx):
    """inverse of tile(): [ny, ntx, nc*64] -> [nc, ny, nx]"""
    import torch

    ny, ntx, w = t.shape
    nc = w // TILE
    npair = nc // 2
    pairs = t[:, :, :npair * 2 * TILE].reshape(ny, ntx, npair, TILE, 2).permute(2, 4, 0, 1, 3).reshape(2 * npair, ny, ntx * TILE)
    planes = [pairs]
    if nc % 2:
        planes.append(t[:, :, npair * 2 * TILE:].reshape(1, ny, ntx * TILE))
    return torch.cat(planes, dim=0)[:, :, :nx].contiguous()


def _ptr(t):
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def _check_f64(*tensors):
    import torch

    for t in tensors:
        if t is None:
            continue
        if t.dtype != torch.float64 or not t.is_contiguous() or not t.is_cuda:
            raise NsdgError("expected contiguous float64 CUDA tensors")


def _ptr_array(tensors):
    arr = (VP * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class HaloPlanHandle:
    """an nsdg_halo plan; start() / finish() are pre-bound zero-argument calls"""

    def __init__(self, ctx, below, above, up_send, down_send, from_above, from_below):
        def pieces(v):  # a row block of a [nc, ny, nx] array is one contiguous block per coefficient plane
            return [v] if v.is_contiguous() else [p for sub in v.unbind(0) for p in pieces(sub)]

        lists = tuple([p for v in views for p in pieces(v)] for views in (up_send, down_send, from_above, from_below))
        for views in lists:
            _check_f64(*views)
        self.keep = [list(v) for v in lists]  # the tensors must outlive the plan
        arrs = []
        for views in lists:
            a = (HaloSeg * max(len(views), 1))()
            for i, v in enumerate(views):
                a[i].ptr, a[i].count = v.data_ptr(), v.numel()
            arrs.append(a)
        h = VP()
        nb = lambda r: -1 if r is None else int(r)
        ctx._call(ctx.lib.nsdg_halo_plan_create(ctx.h, nb(below), nb(above), len(lists[0]), arrs[0], len(lists[1]), arrs[1],
                                                len(lists[2]), arrs[2], len(lists[3]), arrs[3], C.byref(h)))
        self.ctx, self.h = ctx, h
        lib, ch = ctx.lib, ctx.h

        def start():
            rc = lib.nsdg_halo_start(ch, h)
            if rc != 0:
                ctx._call(rc)

        def finish():
            rc = lib.nsdg_halo_finish(ch, h)
            if rc != 0:
                ctx._call(rc)

        self.start, self.finish = start, finish

    def stats(self, reset=False):
        """what the exchanges of this plan cost so far (nsdg_halo_stats_get)"""
        st = HaloStats()
        self.ctx._call(self.ctx.lib.nsdg_halo_stats_get(self.ctx.h, self.h, C.byref(st), int(reset)))
        return st.as_dict()

    def counts(self):
        c = [I64() for _ in range(4)]
        self.ctx._call(self.ctx.lib.nsdg_halo_counts(self.h, *[C.byref(x) for x in c]))
        return tuple(x.value for x in c)

    def close(self):
        if self.h:
            self.ctx.lib.nsdg_halo_plan_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if self.ctx.h:  # plans die with their context's communicator otherwise
                self.close()
        except Exception:
            pass


class Context:
    """One nsdg_ctx bound to a torch device and (by default) torch's current stream on it, so that
    torch.cuda.Event timing and torch.distributed collectives order correctly with the kernels."""

    def __init__(self, device=None, stream=None):
        import torch

        self.lib = load_library()
        if not torch.cuda.is_available():
            raise NsdgError("no HIP device visible: the nextsimdg_amd kernels need a GPU (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        with torch.cuda.device(self.device):
            self.stream = torch.cuda.current_stream() if stream is None else stream
        h = VP()
        self._call(self.lib.nsdg_ctx_create(self.device.index or 0, VP(self.stream.cuda_stream), C.byref(h)))
        self.h = h
        self.nx = self.ny = 0
        self.mevp_variant = DEFAULT_MEVP_VARIANT  # the library default (four sub-iterations per pass)

    def _call(self, rc):
        if rc != 0:
            raise NsdgError("nsdg error %d: %s" % (rc, self.lib.nsdg_last_error().decode()))

    def close(self):
        """destroys the context; a wait of the mEVP pipeline that gave up on it and was never taken (pipeline_waits_given_up) is
        added to the process-wide tally abi.GIVEN_UP_UNSEEN first: a session that ignored wrong results can still be told"""
        global GIVEN_UP_UNSEEN
        if getattr(self, "h", None):
            try:
                GIVEN_UP_UNSEEN += self.pipeline_waits_given_up()
            except Exception:
                pass
            self.lib.nsdg_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        """drains the context's streams; with a communicator the wait is bounded by its deadline (NsdgError on expiry)"""
        self._call(self.lib.nsdg_ctx_synchronize(self.h))

    def copy_f64(self, dst, src):
        """streaming device copy, 16 bytes per lane (the measured copy ceiling of the roofline)"""
        _check_f64(dst, src)
        if dst.numel() != src.numel():
            raise NsdgError("copy_f64: sizes differ")
        self._call(self.lib.nsdg_copy_f64(self.h, dst.data_ptr(), src.data_ptr(), src.numel()))

    def comm_deadline(self, seconds):
        """upper bound on any wait for a neighbour rank (0 = for ever)"""
        self._call(self.lib.nsdg_comm_deadline_set(self.h, float(seconds)))

    def comm_simulate_wire(self, delay_us=0.0, gbs=0.0):
        """rehearsal aid: every exchange spins for delay_us + bytes / gbs on the communication stream (0, 0 = off)"""
        self._call(self.lib.nsdg_comm_simulate_wire(self.h, float(delay_us), float(gbs)))

    def num_cus(self):
        import torch

        return torch.cuda.get_device_properties(self.device).multi_processor_count

    # ---- row-block communicator and ghost-row exchange plans (csrc/halo.hip)
    def comm_init_rccl(self, rank, world, group=None):
        """RCCL communicator of this context; the ncclUniqueId of rank 0 travels through torch.distributed
        (any backend: it is 128 bytes, once)"""
        import torch.distributed as dist

        buf = C.create_string_buffer(COMM_ID_BYTES)
        if rank == 0:
            self._call(self.lib.nsdg_comm_unique_id(buf))
        if world > 1:
            box = [buf.raw if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            buf = C.create_string_buffer(box[0], COMM_ID_BYTES)
        self._call(self.lib.nsdg_comm_init(self.h, rank, world, buf))

    def comm_init_local(self, group_id, rank, world):
        """in-process communicator: the ranks of `group_id` are contexts driven by one thread each"""
        self._call(self.lib.nsdg_comm_init_local(self.h, int(group_id), rank, world))

    def comm_finalize(self):
        self._call(self.lib.nsdg_comm_finalize(self.h))

    def halo_plan(self, below, above, up_send, down_send, from_above, from_below):
        """plan of one kind of exchange; the arguments are lists of CONTIGUOUS tensor views (row blocks)"""
        return HaloPlanHandle(self, below, above, up_send, down_send, from_above, from_below)

    # ---- row-block drivers (csrc/rowblock.hip): one call per model step
    def _geometry(self, desc, blk, peers):
        desc.nx, desc.ny, desc.j0, desc.j1 = blk.nx, blk.ny, blk.j0, blk.j1
        desc.depth_below, desc.depth_above = blk.depth_below, blk.depth_above
        desc.rank_below = -1 if peers[0] is None else peers[0]
        desc.rank_above = -1 if peers[1] is None else peers[1]

    def rb_mevp(self, blk, peers, nsub, overlap, use_graph, s2, uv2, packed, pg):
        """native sub-cycle driver of a row block; s2 = (s, sb): two lists of three tiled stress arrays, uv2 =
        ((u, v), (ub, vb)).  Returns run(parity) -> parity of the result and the plan's (per_pass, group_passes)."""
        d = RbMevpDesc()
        self._geometry(d, blk, peers)
        d.nsub, d.overlap, d.use_graph = nsub, int(bool(overlap)), int(bool(use_graph))
        ts = list(s2[0]) + list(s2[1]) + list(uv2[0]) + list(uv2[1]) + [packed, pg]
        _check_f64(*ts)
        for k in range(2):
            d.s11[k], d.s12[k], d.s22[k] = (s2[k][i].data_ptr() for i in range(3))
            d.u[k], d.v[k] = uv2[k][0].data_ptr(), uv2[k][1].data_ptr()
        d.packed, d.pg = packed.data_ptr(), pg.data_ptr()
        h = VP()
        self._call(self.lib.nsdg_rb_mevp_create(self.h, C.byref(d), C.byref(h)))
        per_pass, group = I32(), I32()
        self._call(self.lib.nsdg_rb_mevp_info(h, C.byref(per_pass), C.byref(group)))
        out, fn, ch = I32(), self.lib.nsdg_rb_mevp_run, self.h

        def run(parity):
            rc = fn(ch, h, parity, C.byref(out))
            if rc != 0:
                self._call(rc)
            return out.value

        def stats(reset=False):
            st = HaloStats()
            self._call(self.lib.nsdg_rb_mevp_stats(ch, h, C.byref(st), int(reset)))
            return st.as_dict()

        def close():  # destroys the plan (its halo plans, packed buffers and events); idempotent, before the context goes
            if run.handle is not None and self.h:
                self.lib.nsdg_rb_mevp_destroy(run.handle)
            run.handle = None

        run.keep = ts
        run.handle = h
        run.stats = stats
        run.close = close
        return run, per_pass.value, group.value

    def rb_transport(self, blk, peers, phi, t1, t2, adv, bounds=None):
        """native SSP-RK3 transport driver of a row block; returns run(dt, parity) -> parity of the new state.  bounds: the closure of the
        step as the plan's OWN (one (lo, hi, cap_mean) per field, () = none), whatever set_transport_bounds says on the context; None:
        the context's bounds of the moment the plan runs"""
        d = RbTransportDesc()
        self._geometry(d, blk, peers)
        d.order, d.nfields = 2, len(phi)
        if bounds is not None:
            d.own_bounds, d.nbounds = 1, len(bounds)
            for k, (lo, hi, cap) in enumerate(bounds):
                d.bounds[k].lo, d.bounds[k].hi, d.bounds[k].cap_mean = float(lo), float(hi), int(bool(cap))
        ts = list(phi) + list(t1) + list(t2) + list(adv)
        _check_f64(*ts)
        for i in range(len(phi)):
            d.phi[i], d.t1[i], d.t2[i] = phi[i].data_ptr(), t1[i].data_ptr(), t2[i].data_ptr()
        d.vx_dg, d.vy_dg, d.un_x, d.un_y = (a.data_ptr() for a in adv)
        h = VP()
        self._call(self.lib.nsdg_rb_transport_create(self.h, C.byref(d), C.byref(h)))
        out, fn, ch = I32(), self.lib.nsdg_rb_transport_run, self.h

        def run(dt, parity):
            rc = fn(ch, h, float(dt), parity, C.byref(out))
            if rc != 0:
                self._call(rc)
            return out.value

        def stats(reset=False):
            st = HaloStats()
            self._call(self.lib.nsdg_rb_transport_stats(ch, h, C.byref(st), int(reset)))
            return st.as_dict()

        def close():
            if run.handle is not None and self.h:
                self.lib.nsdg_rb_transport_destroy(run.handle)
            run.handle = None

        run.keep = ts
        run.handle = h
        run.stats = stats
        run.close = close
        return run

    # ---- arrays private to the mEVP sub-cycle (stress, ice strength) live in the tiled layout
    def private_zeros(self, nc, ny, nx, device):
        import torch

        return torch.zeros(ny, (nx + TILE - 1) // TILE, nc * TILE, dtype=torch.float64, device=device)

    @staticmethod
    def private_rows(f, j0, j1):
        return f[j0:j1]

    @staticmethod
    def private_to_planes(f, nx):
        """tiled private array -> coefficient planes [nc, ny, nx] (checkpoints, gathers)"""
        return untile(f, nx)

    @staticmethod
    def planes_to_private(a):
        return tile(a)

    # ---- column physics
    def column_default_params(self, **kw):
        p = ColumnParams()
        self.lib.nsdg_column_default_params(C.byref(p))
        for k, v in kw.items():
            if k == "albedo":
                p.albedo_kind = ALBEDO[v]
            elif k == "freezing":
                p.freezing_kind = FREEZING[v]
            else:
                if not hasattr(p, k):
                    raise NsdgError("unknown column parameter " + k)
                setattr(p, k, v)
        return p

    def set_column_params(self, p):
        self._call(self.lib.nsdg_column_params_set(self.h, C.byref(p)))

    def column_step(self, dt, state, forcing, newice, diag=None):
        ts = [state[k] for k in STATE] + [forcing[k] for k in FORCING] + [newice]
        _check_f64(*ts, diag)
        n = ts[0].numel()
        self._call(self.lib.nsdg_column_step(self.h, n, float(dt), *[_ptr(t) for t in ts], _ptr(diag)))

    # ---- dynamics
    def mevp_default_params(self, **kw):
        p = MevpParams()
        self.lib.nsdg_mevp_default_params(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise NsdgError("unknown mEVP parameter " + k)
            setattr(p, k, v)
        return p

    def set_mevp_params(self, p):
        self._call(self.lib.nsdg_mevp_params_set(self.h, C.byref(p)))

    def set_grid(self, nx, ny, hx, hy):
        self._call(self.lib.nsdg_grid_set(self.h, nx, ny, float(hx), float(hy)))
        self.nx, self.ny = nx, ny

    def set_mevp_variant(self, variant):
        self._call(self.lib.nsdg_mevp_variant_set(self.h, variant))
        self.mevp_variant = variant

    def set_mevp_strip_rows(self, rows):
        self._call(self.lib.nsdg_mevp_strip_rows_set(self.h, rows))

    def set_mevp_occupancy(self, waves_per_simd):
        self._call(self.lib.nsdg_mevp_occupancy_set(self.h, waves_per_simd))

    def set_transport_variant(self, variant, strip_rows=0):
        self._call(self.lib.nsdg_transport_variant_set(self.h, variant, strip_rows))

    def prepare_advection(self, order, u, v, vx, vy, unx, uny):
        _check_f64(u, v, vx, vy, unx, uny)
        self._call(self.lib.nsdg_prepare_advection(self.h, order, *[_ptr(t) for t in (u, v, vx, vy, unx, uny)]))

    def transport_stage(self, order, j0, j1, dt, a, b, phi0, phis, out, adv):
        _check_f64(*phi0, *phis, *out, *adv)
        self._call(self.lib.nsdg_transport_stage(self.h, order, j0, j1, float(dt), float(a), float(b), len(phis),
                                                 _ptr_array(phi0), _ptr_array(phis), _ptr_array(out),
                                                 *[_ptr(t) for t in adv]))

    def transport_step(self, order, dt, fields, adv, scratch):
        _check_f64(*fields, *adv, scratch)
        need = 2 * sum(f.numel() for f in fields)
        if scratch.numel() < need:
            raise NsdgError("transport scratch too small: need %d doubles" % need)
        self._call(self.lib.nsdg_transport_step(self.h, order, float(dt), len(fields), _ptr_array(fields),
                                                *[_ptr(t) for t in adv], _ptr(scratch)))

    def transport_step_oop(self, order, dt, fields_in, fields_out, adv):
        """one SSP-RK step of every field, out of place, all stages in ONE launch (fields_out must not alias fields_in)"""
        _check_f64(*fields_in, *fields_out, *adv)
        self._call(self.lib.nsdg_transport_step_oop(self.h, order, float(dt), len(fields_in), _ptr_array(fields_in), _ptr_array(fields_out),
                                                    *[_ptr(t) for t in adv]))

    def transport_step_oop_rows(self, order, j0, j1, dt, fields_in, fields_out, adv):
        """the fused step on the rows [j0, j1) only (a row block's own rows; fields_in valid order + 2 rows around them)"""
        _check_f64(*fields_in, *fields_out, *adv)
        self._call(self.lib.nsdg_transport_step_oop_rows(self.h, order, j0, j1, float(dt), len(fields_in), _ptr_array(fields_in),
                                                         _ptr_array(fields_out), *[_ptr(t) for t in adv]))

    def pipeline_waits_given_up(self):
        """waits of the point-to-point mEVP pipelines (csrc/mevp_p2p.h) on THIS context that gave up since the last call: 0 in a correct
        program (a non-zero count means wrong results of the launches in between, and synchronize / mevp_subcycle / the row-block
        run raise NsdgError until this call has taken the events); waits for the context's stream"""
        n = C.c_uint32(0)
        self._call(self.lib.nsdg_mevp_pipeline_health(self.h, C.byref(n)))
        return int(n.value)

    def set_transport_bounds(self, bounds):
        """closure of a transport step: one (lo, hi, cap_mean) per advected field, in the order of the step calls' field lists;
        () or None = none.  abi.H_A_BOUNDS: the dynamics' H and A"""
        bounds = tuple(bounds or ())
        arr = (FieldBounds * max(len(bounds), 1))()
        for k, (lo, hi, cap) in enumerate(bounds):
            arr[k].lo, arr[k].hi, arr[k].cap_mean = float(lo), float(hi), int(bool(cap))
        self._call(self.lib.nsdg_transport_bounds_set(self.h, len(bounds), arr))
        self.transport_bounds = bounds

    def transport_limit(self, order, j0, j1, fields):
        """cap + scaling limiter in place on the rows [j0, j1) (what the step entry points apply themselves)"""
        _check_f64(*fields)
        self._call(self.lib.nsdg_transport_limit(self.h, order, j0, j1, len(fields), _ptr_array(fields)))

    def dg_to_cg(self, f_dg, f_cg):
        _check_f64(f_dg, f_cg)
        if f_dg.dim() != 3:
            raise NsdgError("dg_to_cg expects coefficient planes [nc, ny, nx]")
        self._call(self.lib.nsdg_dg_to_cg(self.h, f_dg.shape[0], _ptr(f_dg), _ptr(f_cg)))

    def ice_strength(self, H, A, pg, j0=0, j1=None):
        _check_f64(H, A, pg)
        self._call(self.lib.nsdg_ice_strength(self.h, j0, self.ny if j1 is None else j1, _ptr(H), _ptr(A), _ptr(pg)))

    def boxtest_forcing(self, domain_size, t, wind=None, ocean=None):
        ts = list(wind or (None, None)) + list(ocean or (None, None))
        _check_f64(*ts)
        self._call(self.lib.nsdg_boxtest_forcing(self.h, float(domain_size), float(t), *[_ptr(x) for x in ts]))

    def set_block(self, row0, ny_global):
        """placement of the local array in the global domain (analytic forcing providers)"""
        self._call(self.lib.nsdg_block_set(self.h, int(row0), int(ny_global)))

    def column_forcing(self, kind, t, forcing):
        """thermodynamic forcing planes at model time t; kind: "dummy" (the reference's constants) or "winter" """
        ts = [forcing[k] for k in ("tair", "tdew", "slp", "qsw", "qlw", "mld", "snowfall")]
        _check_f64(*ts)
        self._call(self.lib.nsdg_column_forcing(self.h, {"dummy": 0, "winter": 1}[kind], float(t), *[_ptr(x) for x in ts]))

    def column_wind(self, ua, va, wind):
        _check_f64(ua, va, wind)
        self._call(self.lib.nsdg_column_wind(self.h, _ptr(ua), _ptr(va), _ptr(wind)))

    def wind_stress(self, ua, va, tax, tay):
        _check_f64(ua, va, tax, tay)
        self._call(self.lib.nsdg_wind_stress(self.h, ua.numel(), _ptr(ua), _ptr(va), _ptr(tax), _ptr(tay)))

    def mevp_stress(self, k0, k1, u, v, pg, s11, s12, s22):
        _check_f64(u, v, pg, s11, s12, s22)
        self._call(self.lib.nsdg_mevp_stress(self.h, k0, k1, *[_ptr(t) for t in (u, v, pg, s11, s12, s22)]))

    def mevp_pack_nodal(self, dt, u0v0, tau, ocean, cgh, cga, packed):
        ts = [u0v0[0], u0v0[1], tau[0], tau[1], ocean[0], ocean[1], cgh, cga, packed]
        _check_f64(*ts)
        if packed.numel() < 8 * cgh.numel():
            raise NsdgError("packed nodal buffer too small: need %d doubles" % (8 * cgh.numel()))
        self._call(self.lib.nsdg_mevp_pack_nodal(self.h, float(dt), *[_ptr(t) for t in ts]))

    def mevp_prepare(self, dt, H, A, wind, ocean, u0v0, packed):
        """nodal means of H and A + wind stress + coefficient packing in one launch"""
        ts = [H, A, wind[0], wind[1], ocean[0], ocean[1], u0v0[0], u0v0[1], packed]
        _check_f64(*ts)
        self._call(self.lib.nsdg_mevp_prepare(self.h, float(dt), *[_ptr(t) for t in ts]))

    def mevp_velocity(self, j0, j1, s, uv_old, uv_new, packed):
        ts = [s[0], s[1], s[2], uv_old[0], uv_old[1], uv_new[0], uv_new[1], packed]
        _check_f64(*ts)
        self._call(self.lib.nsdg_mevp_velocity(self.h, j0, j1, *[_ptr(t) for t in ts]))

    def mevp_iterate(self, k0, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        ts = [s_in[0], s_in[1], s_in[2], s_out[0], s_out[1], s_out[2], uv_old[0], uv_old[1], uv_new[0], uv_new[1], packed, pg]
        _check_f64(*ts)
        self._call(self.lib.nsdg_mevp_iterate(self.h, k0, j0, j1, *[_ptr(t) for t in ts]))

    def mevp_iterate2(self, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        """two sub-iterations in one pass on the owned rows [j0, j1) (variant 2)"""
        self.bind_mevp_iterate2(j0, j1, s_in, s_out, uv_old, uv_new, packed, pg)()

    def mevp_iterate3(self, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        """three sub-iterations in one pass on the owned rows [j0, j1) (variant 3)"""
        self.bind_mevp_iterate2(j0, j1, s_in, s_out, uv_old, uv_new, packed, pg, passes=3)()

    def mevp_iterate3_pair(self, ra, rb, s_in, s_out, uv_old, uv_new, packed, pg):
        """three sub-iterations on two disjoint row ranges ra = (j0, j1), rb = (j0, j1) in one launch"""
        ts = [s_in[0], s_in[1], s_in[2], s_out[0], s_out[1], s_out[2], uv_old[0], uv_old[1], uv_new[0], uv_new[1], packed, pg]
        _check_f64(*ts)
        self._call(self.lib.nsdg_mevp_iterate3_pair(self.h, ra[0], ra[1], rb[0], rb[1], *[_ptr(t) for t in ts]))

    def bind_mevp_iterate3(self, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        return self.bind_mevp_iterate2(j0, j1, s_in, s_out, uv_old, uv_new, packed, pg, passes=3)

    def mevp_iterate4(self, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        """four sub-iterations in one pass on the owned rows [j0, j1) (variant 4: one pipeline stage per wave)"""
        self.bind_mevp_iterate2(j0, j1, s_in, s_out, uv_old, uv_new, packed, pg, passes=4)()

    def mevp_iterate4_pair(self, ra, rb, s_in, s_out, uv_old, uv_new, packed, pg):
        """four sub-iterations on two disjoint row ranges ra = (j0, j1), rb = (j0, j1) in one launch"""
        ts = [s_in[0], s_in[1], s_in[2], s_out[0], s_out[1], s_out[2], uv_old[0], uv_old[1], uv_new[0], uv_new[1], packed, pg]
        _check_f64(*ts)
        self._call(self.lib.nsdg_mevp_iterate4_pair(self.h, ra[0], ra[1], rb[0], rb[1], *[_ptr(t) for t in ts]))

    def bind_mevp_iterate4(self, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        return self.bind_mevp_iterate2(j0, j1, s_in, s_out, uv_old, uv_new, packed, pg, passes=4)

    def bind_mevp_iterate2(self, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg, passes=2):
        ts = [s_in[0], s_in[1], s_in[2], s_out[0], s_out[1], s_out[2], uv_old[0], uv_old[1], uv_new[0], uv_new[1], packed, pg]
        _check_f64(*ts)
        fn = {2: self.lib.nsdg_mevp_iterate2, 3: self.lib.nsdg_mevp_iterate3, 4: self.lib.nsdg_mevp_iterate4}[passes]
        args = (self.h, I32(j0), I32(j1)) + tuple(_ptr(t) for t in ts)
        keep = ts

        def call():
            rc = fn(*args)
            if rc != 0:
                self._call(rc)
            return keep is None

        return call

    def bind_mevp_iterate(self, k0, j0, j1, s_in, s_out, uv_old, uv_new, packed, pg):
        """Pre-validated, pre-marshalled form of mevp_iterate for inner loops: returns a zero-argument
        callable.  (Argument checking and ctypes marshalling cost ~20 us per call in Python -- comparable
        to a sub-iteration of a 256-row block -- so the 120-iteration sub-cycle binds its calls once.)"""
        ts = [s_in[0], s_in[1], s_in[2], s_out[0], s_out[1], s_out[2], uv_old[0], uv_old[1], uv_new[0], uv_new[1], packed, pg]
        _check_f64(*ts)
        fn, h = self.lib.nsdg_mevp_iterate, self.h
        args = (h, I32(k0), I32(j0), I32(j1)) + tuple(_ptr(t) for t in ts)
        keep = ts  # the tensors must outlive the binding

        def call():
            rc = fn(*args)
            if rc != 0:
                self._call(rc)
            return keep is None

        return call

    def mevp_subcycle(self, dt, nsub, s, u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, scratch):
        ts = [s[0], s[1], s[2], u, v, u0, v0, tax, tay, uo, vo, cgh, cga, pg, scratch]
        _check_f64(*ts)
        need = 10 * u.numel() + 3 * s[0].numel()  # s[k] are tiled arrays: numel() == nsdg_tiled_len(nx, ny, 8)
        if scratch.numel() < need:
            raise NsdgError("mEVP scratch too small: need %d doubles" % need)
        self._call(self.lib.nsdg_mevp_subcycle(self.h, float(dt), int(nsub), *[_ptr(t) for t in ts]))
