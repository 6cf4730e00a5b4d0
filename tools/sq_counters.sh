#!/bin/bash
# SQ counters of the dominant mEVP kernel: bash tools/sq_counters.sh TAG VARIANT [NAME of an alt build]   (on the GPU box)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; V=$2; N=${3:-default}
. "$ROOT/tools/rocprof_guard.sh"  # every pass under its own time limit; a failed pass ends the script (profiles/r05_pmc_pass_hang_cause.md)
PY="$(command -v python3)"
OUT="$ROOT/gpurun_out/${NSDG_ROUND:-r06}/sq_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
if [ "$N" != default ]; then export NSDG_LIB="$ROOT/nextsimdg_amd/lib/alt/$N/libnsdg.so"; else unset NSDG_LIB; fi
i=0; failed=0
# SQ / GRBM counters only, at most 8 per pass (the MFMA counters of the round-4 version were never validated on this pool: dropped)
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_CYCLES"; do
  i=$((i+1))
  guarded_rocprof p$i "$OUT" 180 --kernel-trace --pmc $set -- "$PY" "$ROOT/bench.py" --variant $V --steps 1 --warmup 0 --nsub 12 --no-cpu-baseline || { failed=1; break; }
done
"$PY" - "$OUT" "$TAG" <<'PYEOF'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mevp_fused" in r["Kernel_Name"] and "kernel" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-22:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%s %-24s %-28s %.5g  (n=%d)" % (name, k, c, sum(v) / len(v), len(v)))
PYEOF
find "$OUT" -name "*_kernel_trace.csv" -delete
exit $failed
