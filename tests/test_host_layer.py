"""C++ host layer (nextsimdg_amd/host): plugin registry, configuration, structure, HipStep, Model.
The C++ test program restates the behaviour pinned by the reference's own Catch2 tests
(core/test/*_test.cpp, physics/test/NextsimPhysics_test.cpp); pytest only builds and runs it."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "nextsimdg_amd", "host")


@pytest.fixture(scope="module")
def host_build():
    from nextsimdg_amd import build

    build.build_lib(verbose=False)
    subprocess.check_call(["make", "-s", "-C", HOST])
    return os.path.join(HOST, "build")


GOLDEN = os.path.join(ROOT, "tests", "golden")


def run(cmd, cwd=None):
    env = dict(os.environ, NSDG_GOLDEN_DIR=GOLDEN)  # where host_tests finds the reference's run/dev1.res.nc
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=cwd, timeout=300, env=env)
    return p.returncode, p.stdout.decode()


def test_host_cpu_cases(host_build):
    rc, out = run([os.path.join(host_build, "host_tests")])
    assert rc == 0, out
    assert re.search(r"host CPU tests: \d+ checks, 0 failures", out), out


def test_written_restart_file_is_read_by_the_hdf5_library(host_build, tmp_path):
    """the HDF5 file written by RectGrid::dump is handed to the HDF5 library's own `h5dump` (present in this
    image under /opt/conda/bin; skipped where it is not): structure, dimensions, datatype and values must
    come back as written (6 x 9 grid, 2 ice layers, index pattern of core/test/DevGrid_test.cpp:36-96)"""
    import shutil

    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if not os.path.exists(h5dump):
        pytest.skip("no h5dump in this image")
    path = os.path.join(str(tmp_path), "kept.nc")
    env = dict(os.environ, NSDG_GOLDEN_DIR=GOLDEN, NSDG_KEEP_RESTART=path)
    p = subprocess.run([os.path.join(host_build, "host_tests")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=300)
    assert p.returncode == 0 and os.path.exists(path), p.stdout.decode()
    rc, head = run([h5dump, "-H", path])
    assert rc == 0, head
    for needle in ('GROUP "structure"', 'ATTRIBUTE "type"', 'GROUP "data"', 'DATASET "hice"', 'DATASET "tice"', "H5T_IEEE_F64LE",
                   "( 6, 9 ) / ( 6, 9 )", "( 6, 9, 2 ) / ( 6, 9, 2 )"):
        assert needle in head, (needle, head)
    rc, attr = run([h5dump, "-a", "/structure/type", path])
    assert rc == 0 and '"rectgrid"' in attr, attr
    # NetCDF-4 named dimensions x, y, nLayers (core/src/DevGridIO.cpp:169-201): the HDF5 library resolves the object
    # references of DIMENSION_LIST / REFERENCE_LIST (global heap, compound type) exactly as in the reference's own file
    rc, att = run([h5dump, "-A", path])
    assert rc == 0, att
    rc, ref = run([h5dump, "-A", os.path.join(GOLDEN, "dev1.res.nc")])
    assert rc == 0, ref

    def attributes(text):  # {dataset: sorted attribute names}, and the resolved reference targets
        out, cur = {}, None
        for ln in text.splitlines():
            m = re.search(r'DATASET "(\w+)" \{', ln)
            if m:
                cur = m.group(1)
                out[cur] = []
            m = re.search(r'ATTRIBUTE "(\w+)"', ln)
            if m and cur:
                out[cur].append(m.group(1))
        return {k: sorted(v) for k, v in out.items()}

    assert attributes(att) == attributes(ref)  # same datasets (x, y, nLayers included), same attributes on each
    assert re.search(r"\(DATASET \d+ /data/x \), \(DATASET \d+ /data/y \),\s+\(2\): \(DATASET \d+ /data/nLayers \)", att)  # tice: (x, y, nLayers)
    assert att.count("/data/x ), (DATASET") >= 6 and '"This is a netCDF dimension but not a netCDF variable.         6"' in att
    for needle in ('"DIMENSION_SCALE"', "H5T_IEEE_F32BE", 'H5T_STD_I32LE "dimension"', "(0): 0, 1, 2"):
        assert needle in att, needle
    rc, data = run([h5dump, "-d", "/data/hice", "-w", "400", path])
    assert rc == 0, data
    row4 = [ln for ln in data.splitlines() if ln.strip().startswith("(4,0):")]
    assert row4 and "1.0403" in row4[0], data  # element (4, 3) of the pattern 1 + 0.01 i + 0.0001 j


def test_help_lists_the_reference_options(host_build):
    rc, out = run([os.path.join(host_build, "nextsim_amd"), "--help"])
    assert rc == 0
    for opt in ("--help", "--config-file", "--config-files"):  # core/src/CommandLineParser.cpp:30-37
        assert opt in out


def test_executable_fails_loudly_without_gpu(host_build):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    rc, out = run([os.path.join(host_build, "nextsim_amd"), "--config-file", os.path.join(ROOT, "run", "dev1.cfg")], cwd=GOLDEN)
    assert rc != 0 and "no HIP device" in out


def test_unreadable_init_file_stops_the_run(host_build, tmp_path):
    """the reference's Model::configure builds the structure from the named restart file unconditionally and throws
    (core/src/Model.cpp:59-62): a typo in model.init_file, a truncated or a foreign file must end the run with a
    non-zero status and no restart file -- never a plausible run from constants"""
    exe = os.path.join(host_build, "nextsim_amd")
    cfg = os.path.join(ROOT, "run", "dev1.cfg")
    final = os.path.join(str(tmp_path), "restart.nc")
    rc, out = run([exe, "--config-file", cfg, "--model.final_file=" + final], cwd=str(tmp_path))  # dev1.res.nc is not here
    assert rc != 0 and "cannot open dev1.res.nc" in out, out
    assert not os.path.exists(final)
    whole = open(os.path.join(GOLDEN, "dev1.res.nc"), "rb").read()
    trunc = os.path.join(str(tmp_path), "truncated.nc")
    with open(trunc, "wb") as f:
        f.write(whole[:len(whole) // 3])
    rc, out = run([exe, "--config-file", cfg, "--model.init_file=" + trunc, "--model.final_file=" + final], cwd=str(tmp_path))
    assert rc != 0 and "nextsim_amd:" in out, out
    assert not os.path.exists(final)
    junk = os.path.join(str(tmp_path), "junk.nc")
    with open(junk, "wb") as f:
        f.write(b"this is not a restart file\n" * 10)
    rc, out = run([exe, "--config-file", cfg, "--model.init_file=" + junk, "--model.final_file=" + final], cwd=str(tmp_path))
    assert rc != 0 and "no structure type" in out, out


@pytest.mark.gpu
def test_host_gpu_cases(host_build, gpu):
    rc, out = run([os.path.join(host_build, "host_tests"), "--gpu"])
    assert rc == 0, out
    assert re.search(r"host GPU tests: \d+ checks, 0 failures", out), out


@pytest.mark.gpu
def test_dev1_cfg_end_to_end(host_build, gpu, tmp_path):
    """BASELINE config 1: ./nextsim --config-file dev1.cfg (run/dev1.sh:5 of the reference) -> one
    iterate(1) on 100 elements; expected state = SURVEY.md Appendix C row 'dev1'."""
    rc, out = run([os.path.join(host_build, "nextsim_amd"), "--config-file", os.path.join(ROOT, "run", "dev1.cfg"), "--model.init_file=",
                   "--model.final_file=%s" % os.path.join(str(tmp_path), "restart.nsdg")], cwd=str(tmp_path))
    assert rc == 0, out
    m = re.search(r"elements=(\d+) launches=(\d+) hice=(\S+) cice=(\S+) hsnow=(\S+) tice0=(\S+) sst=(\S+)", out)
    assert m, out
    assert int(m.group(1)) == 100 and int(m.group(2)) == 1
    got = [float(m.group(i)) for i in range(3, 8)]
    want = [0.04668325240678619, 0.36670813101696548, 0.0, -1.444501803353837, -1.0]
    for g, w in zip(got, want):
        assert abs(g - w) <= 1e-12 * abs(w), (got, want)
    assert os.path.exists(os.path.join(str(tmp_path), "restart.nsdg"))
    assert "constant initial state" in out  # init_file explicitly empty: the [init] constants were used
    # the same run started from the reference's own NetCDF-4 restart file (tests/golden/dev1.res.nc is a copy of
    # run/dev1.res.nc; `init_file = dev1.res.nc` is relative to the working directory as in run/dev1.sh) and
    # writing its restart file in the same format
    rc, out = run([os.path.join(host_build, "nextsim_amd"), "--config-file", os.path.join(ROOT, "run", "dev1.cfg"),
                   "--model.final_file=%s" % os.path.join(str(tmp_path), "restart.nc")], cwd=GOLDEN)
    assert rc == 0 and "Initial state read from dev1.res.nc (structure devgrid, 10 x 10)" in out, out
    m2 = re.search(r"elements=(\d+) launches=(\d+) hice=(\S+) cice=(\S+) hsnow=(\S+) tice0=(\S+) sst=(\S+)", out)
    assert m2 and [float(m2.group(i)) for i in range(3, 8)] == got, out
    with open(os.path.join(str(tmp_path), "restart.nc"), "rb") as f:
        assert f.read(8) == b"\x89HDF\r\n\x1a\n"
    # command-line override of a config value (first-wins precedence): 3 steps instead of 1
    rc, out = run([os.path.join(host_build, "nextsim_amd"), "--config-file", os.path.join(ROOT, "run", "dev1.cfg"), "--model.init_file=",
                   "--model.stop=3", "--model.final_file=%s" % os.path.join(str(tmp_path), "r2.nsdg")], cwd=str(tmp_path))
    assert rc == 0 and "launches=3" in out, out


@pytest.mark.gpu
def test_dynamics_step_executable_matches_python_driver(host_build, gpu, tmp_path):
    """the C++ DynamicsStep plugin and the Python row-block driver call the same ABI in the same order:
    same inputs (uniform ice, analytic box-test forcing) -> same diagnostics"""
    import numpy as np
    import torch

    from nextsimdg_amd import abi, rowblock, synthetic

    nslow, nfast, nsub = 48, 64, 8
    cfg = os.path.join(str(tmp_path), "dyn.cfg")
    with open(cfg, "w") as f:
        f.write("[Modules]\nNextsim::IModelStep = Nextsim::DynamicsStep\n[model]\nstructure = rectgrid\nstart = 0\nstop = 360\n"
                "time_step = 120\nfinal_file = %s\n[rectgrid]\nnx = %d\nny = %d\n[init]\nhice = 0.3\ncice = 0.9\n[dynamics]\nnsub = %d\n"
                % (os.path.join(str(tmp_path), "r.nsdg"), nslow, nfast, nsub))
    rc, out = run([os.path.join(host_build, "nextsim_amd"), "--config-file", cfg], cwd=str(tmp_path))
    assert rc == 0, out
    m = re.search(r"dynamics umax=(\S+) sumH=(\S+) sumA=(\S+)", out)
    assert m and "launches=3" in out, out
    got = [float(m.group(i)) for i in (1, 2, 3)]

    nx, ny = nfast, nslow  # the dynamics ABI calls the fast dimension nx
    bt = synthetic.BoxTest(nx, ny)
    ctx = abi.Context(gpu)
    ctx.set_mevp_params(ctx.mevp_default_params(**bt.subcycle_parameters(120.0)))  # the hosts' policy (DynamicsStep: stableDeltaMin)
    core = rowblock.DynamicsCore(ctx, rowblock.RowBlock(nx, ny), bt.hx, bt.hy, 120.0, nsub, torch.device("cuda"))
    H = np.zeros((6, ny, nx)); H[0] = 0.3
    A = np.zeros((6, ny, nx)); A[0] = 0.9
    uo, vo = bt.ocean()
    ua, va = bt.wind(0.0)
    core.load_global(H, A, uo, vo, ua, va)
    for k in range(3):
        ctx.set_grid(nx, ny, bt.hx, bt.hy)
        ctx.boxtest_forcing(bt.L, 120.0 * k, wind=(core.ua, core.va))  # the cyclone moves with model time
        if k == 1:  # the device forcing provider reproduces the numpy formulae
            wa, wb = bt.wind(120.0)
            assert float((core.ua.cpu() - torch.from_numpy(wa)).abs().max()) < 1e-13 * float(np.abs(wa).max())
            assert float((core.va.cpu() - torch.from_numpy(wb)).abs().max()) < 1e-13 * float(np.abs(wb).max())
        core.step()
    want = [float(core.u.abs().max()), float(core.H[0].sum()), float(core.A[0].sum())]
    assert want[0] > 1e-6
    for g, w in zip(got, want):
        assert abs(g - w) <= 1e-10 * abs(w), (got, want)


@pytest.mark.gpu
def test_timing_report(host_build, gpu, tmp_path):
    """model.timing = true prints the hierarchical timer tree (Timer::report format of the reference,
    core/src/Timer.cpp:141-198) with device work charged to the node that enqueued it"""
    rc, out = run([os.path.join(host_build, "nextsim_amd"), "--config-file", os.path.join(ROOT, "run", "dev1.cfg"), "--model.timing=true", "--model.init_file=",
                   "--model.stop=4", "--model.final_file=%s" % os.path.join(str(tmp_path), "r.nsdg")], cwd=str(tmp_path))
    assert rc == 0, out
    assert re.search(r"Total: ticks = 1", out) and re.search(r"run: ticks = 1", out), out
    assert re.search(r"iterate: ticks = 4 .*ms/tick", out), out
    assert "configure" in out and "% of parent" in out
