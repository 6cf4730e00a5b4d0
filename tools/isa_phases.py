"""Counts instructions between the s_memtime stamps of a -DNSDG_STAMPS assembly listing (diagnostic)."""
import sys

lines = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN16nsdg_mevp_detail18mevp_fused2_kernel")][0]
keys = ["valu", "f64", "trans", "acc", "dpp", "lane", "vmem", "salu", "wait", "nop"]
cur = {k: 0 for k in keys}
segs = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if not t or t.startswith(";") or t.startswith(".") or t.split()[0].endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_memtime":
        segs.append(cur)
        cur = {k: 0 for k in keys}
        continue
    if op.startswith("v_"):
        cur["valu"] += 1
        if op.startswith(("v_rsq_f64", "v_rcp_f64")):
            cur["trans"] += 1
        elif "f64" in op:
            cur["f64"] += 1
        if "accvgpr" in op:
            cur["acc"] += 1
        if "dpp" in op:
            cur["dpp"] += 1
        if "readlane" in op or "writelane" in op:
            cur["lane"] += 1
    elif op.startswith(("global_", "flat_", "scratch_")):
        cur["vmem"] += 1
    elif op.startswith("s_waitcnt"):
        cur["wait"] += 1
    elif op.startswith("s_nop"):
        cur["nop"] += 1
    elif op.startswith("s_"):
        cur["salu"] += 1
segs.append(cur)
print("seg " + " ".join("%6s" % k for k in keys) + "   est.cycles")
for i, sg in enumerate(segs):
    est = (sg["valu"] - sg["trans"]) * 4.5 + sg["trans"] * 16 + (sg["salu"] + sg["nop"] + sg["vmem"]) * 4
    print("%3d " % i + " ".join("%6d" % sg[k] for k in keys) + "   %8.0f" % est)
