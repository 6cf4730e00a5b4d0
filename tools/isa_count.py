"""Instruction mix of a kernel in an assembly listing: python tools/isa_count.py file.s [kernel-substring]
(whole kernel body; static counts -- a loop body that is executed N times counts once)."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
sub = sys.argv[2] if len(sys.argv) > 2 else "_kernel"
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % sub, l)][0]
c = collections.Counter()
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if not t or t.startswith(";") or t.startswith(".") or t.split()[0].endswith(":"):
        continue
    op = t.split()[0]
    if "accvgpr" in op:
        k = "accvgpr"
    elif op.startswith(("v_rsq_f64", "v_rcp_f64")):
        k = "trans64"
    elif op.startswith("v_") and "f64" in op:
        k = "fp64"
    elif "dpp" in t:
        k = "dpp"
    elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        k = "lane(sgpr spill)"
    elif op.startswith("v_mov") or op.startswith("v_pk_mov"):
        k = "v_mov"
    elif op.startswith("v_"):
        k = "valu other"
    elif op.startswith("scratch_"):
        k = "scratch"
    elif op.startswith("global_load"):
        k = "global_load"
    elif op.startswith("global_store"):
        k = "global_store"
    elif op.startswith("ds_"):
        k = "lds"
    elif op.startswith("s_waitcnt"):
        k = "s_waitcnt"
    elif op.startswith("s_barrier"):
        k = "s_barrier"
    elif op.startswith("s_"):
        k = "salu"
    else:
        k = op
    c[k] += 1
tot = sum(c.values())
for k, v in c.most_common():
    print("%-18s %6d  %5.1f %%" % (k, v, 100. * v / tot))
print("total", tot)
