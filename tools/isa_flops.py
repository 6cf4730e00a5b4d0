"""fp64 flops of ONE march step (= one element-sub-iteration per lane) of mevp_fused4_kernel, counted in the ISA:
    python tools/isa_flops.py [listing.s]        (without an argument the listing is produced with the product's flags)
The kernel has two loops: the stages 1-3 (first `Inner Loop Header` of the listing) and the loader; the body of a loop is every
instruction between its header label and the backward branch, so wave-uniform side paths (first row of a stage, boundary
stores) are counted although most steps skip them: they hold no fp64 arithmetic worth mentioning.  v_fma_f64 = 2 flops,
v_add / v_mul = 1, v_rcp / v_rsq = 1 (they feed Newton steps that are counted as what they are); min / max / compares / moves = 0.
bench.py's `fp64_flops_per_launch` = the stage-loop figure x 64 lanes x element-sub-iterations: the numbers are kept in bench.py
(FP64_FLOPS_PER_ELEMENT_SUBITER, one for the uniform and one for the adaptive form of alpha and beta) and re-counted by
tests/test_bench_launch.py whenever the compiler is present."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fno-signed-zeros", "-ffp-contract=on", "-Wno-unused-function"]


def listing(path=None):
    if path:
        return open(path).read()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "fused4.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-S", "--cuda-device-only", os.path.join(ROOT, "nextsimdg_amd", "csrc", "mevp_fused4.hip"), "-o", out],
                              stderr=subprocess.DEVNULL)
        return open(out).read()


def loops(text, adaptive=False):
    """[(header label, [instructions])] of the depth-1 loops of one instantiation of the kernel: uniform alpha / beta (mevp_fused4_kernel<false>)
    or the adaptive form (<true>)"""
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*mevp_fused4_kernelILb%dE\w*:" % int(adaptive), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
    out = []
    for h in heads:
        label = body[h].split(":")[0].strip()
        # the loop's blocks are tagged "in Loop: Header=<label>"; the body = header block .. last tagged block's end
        tagged = [i for i, l in enumerate(body) if "Header=%s " % label.lstrip(".L") in l or i == h]
        first = min(tagged)
        last = max(tagged)
        nxt = next((i for i in range(last + 1, len(body)) if re.match(r"^\.LBB\d+_\d+:", body[i])), len(body))
        ins = [l.strip() for l in body[first:nxt] if l.strip() and not l.strip().startswith((";", ".")) and not l.strip().split()[0].endswith(":")]
        out.append((label, ins))
    return out


def flops(ins):
    c = collections.Counter()
    for t in ins:
        op = t.split()[0]
        if op.startswith("v_fma_f64") or op.startswith("v_fmac_f64"):
            c["fma"] += 1
        elif op.startswith(("v_add_f64", "v_mul_f64")):
            c["addmul"] += 1
        elif op.startswith(("v_rcp_f64", "v_rsq_f64")):
            c["trans"] += 1
        elif "f64" in op and op.startswith("v_"):
            c["other_f64"] += 1
        if op.startswith("v_"):
            c["valu"] += 1
    c["flops"] = 2 * c["fma"] + c["addmul"] + c["trans"]
    return c


if __name__ == "__main__":
    text = listing(sys.argv[1] if len(sys.argv) > 1 else None)
    for adaptive in (False, True):
      print("mevp_fused4_kernel<%s> (%s alpha, beta)" % (str(adaptive).lower(), "local, solution-adaptive" if adaptive else "uniform"))
      for label, ins in loops(text, adaptive):
        c = flops(ins)
        print("%-10s %5d instructions, %4d VALU: fma %d, add/mul %d, rcp/rsq %d, other f64 (min/max/cmp/cvt) %d -> %d fp64 flops per lane and march step"
              % (label, len(ins), c["valu"], c["fma"], c["addmul"], c["trans"], c["other_f64"], c["flops"]))
