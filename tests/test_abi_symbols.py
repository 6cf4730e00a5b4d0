"""CPU-side checks of the C-ABI shared library: it loads, exports every symbol include/nsdg.h
declares, and fails loudly (never falls back) when no GPU is present."""
import ctypes as C
import os
import re

import pytest

from nextsimdg_amd import abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    build.build_lib(verbose=False)
    return abi.load_library()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "nsdg.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(nsdg_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libnsdg.so does not export " + n
        assert n in abi.SYMBOLS, "nextsimdg_amd/abi.py does not bind " + n
    assert sorted(abi.SYMBOLS) == names


def test_header_is_plain_c():
    """include/nsdg.h is the boundary: it must compile as C99 on its own (no C++ types, no torch types)"""
    import subprocess

    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "nsdg.h")],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert p.returncode == 0, p.stdout.decode()


def test_abi_version_and_default_params(lib):
    assert lib.nsdg_abi_version() == 6  # 6: a pipeline wait that gives up is an error status (nsdg_ctx_synchronize, nsdg_mevp_subcycle, nsdg_rb_mevp_run), per-context count; 5: nsdg_transport_bounds_set / nsdg_transport_limit, nsdg_mevp_params.min_conc / min_thick; 2: nsdg_comm_* / nsdg_halo_* (row-block ghost exchange); 3: bounded waits, exchange statistics, nsdg_copy_f64; 4: nsdg_mevp_iterate4*, nsdg_comm_simulate_wire
    p = abi.ColumnParams()
    lib.nsdg_column_default_params(C.byref(p))
    # defaults of the reference: NextsimPhysics.cpp:76-82, ThermoIce0.cpp:30-31, HiblerConcentration.cpp:28-29
    assert (p.drag_ocean_q, p.drag_ocean_t, p.drag_ice_t, p.ocean_albedo, p.i0) == (1.5e-3, 0.83e-3, 1.3e-3, 0.07, 0.17)
    assert (p.min_conc, p.min_thick, p.ks, p.flooding, p.h0, p.phi_m) == (1e-12, 0.01, 0.3096, 1, 0.25, 0.5)
    assert (p.ccsm_ice_albedo, p.ccsm_snow_albedo, p.albedo_kind, p.freezing_kind) == (0.538, 0.8256, 0, 0)
    m = abi.MevpParams()
    lib.nsdg_mevp_default_params(C.byref(m))
    assert (m.alpha, m.beta, m.pstar, m.delta_min) == (1500.0, 1500.0, 27.5e3, 2e-9)
    # the ice-free-node rule is ON by default, with the column model's cut-off values (physics/src/modules/NextsimPhysics.cpp:81-82)
    assert (m.h_min, m.min_conc, m.min_thick) == (1e-4, p.min_conc, p.min_thick)
    # closure entry points without a context: argument errors, never a crash
    assert lib.nsdg_transport_bounds_set(None, 0, None) == -1 and lib.nsdg_transport_limit(None, 2, 0, 0, 1, None) == -1
    assert lib.nsdg_mevp_pipeline_health(None, None) == -1


def test_param_struct_layout_matches_oracle():
    import oracle_lib as O

    for a, b in ((abi.ColumnParams, O.ColumnParams), (abi.MevpParams, O.MevpParams)):
        assert [f[0] for f in a._fields_] == [f[0] for f in b._fields_]
        assert C.sizeof(a) == C.sizeof(b)


def test_comm_entry_points_fail_cleanly_without_a_device(lib):
    """the ghost-exchange entry points on a machine without a GPU: argument / state errors, never a crash"""
    assert lib.nsdg_comm_init(None, 0, 1, None) == -1
    assert lib.nsdg_comm_finalize(None) == -1
    assert lib.nsdg_halo_plan_destroy(None) == 0
    assert lib.nsdg_halo_start(None, None) == -1
    assert lib.nsdg_halo_finish(None, None) == -1


def test_no_silent_cpu_fallback(lib):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = lib.nsdg_ctx_create(0, None, C.byref(h))
    assert rc == -4  # NSDG_ERR_NODEVICE
    assert b"no HIP device" in lib.nsdg_last_error()
    with pytest.raises(abi.NsdgError):
        abi.Context()


def test_product_never_imports_oracle():
    # the oracle is test infrastructure: nothing under nextsimdg_amd/ may reference it
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nextsimdg_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"import\s+oracle|from\s+oracle|liboracle|#include\s+\".*oracle", txt):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad
