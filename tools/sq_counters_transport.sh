#!/bin/bash
# (a pass with TCP_*_STALL / TA_* counters stopped answering on this pool once: keep custom sets to SQ / SQC / TCC counters)
# SQ counters of the fused transport kernel: bash tools/sq_counters_transport.sh TAG BUILD [bench args]   (on the GPU box;
# BUILD = default or the name of an alt build under nextsimdg_amd/lib/alt)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; N=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
PY="$(command -v python3)"
OUT="$ROOT/gpurun_out/r04/sqt_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
if [ "$N" != default ]; then export NSDG_LIB="$ROOT/nextsimdg_amd/lib/alt/$N/libnsdg.so"; else unset NSDG_LIB; fi
i=0
if [ -n "$NSDG_PMC_SETS" ]; then mapfile -t SETS < "$ROOT/$NSDG_PMC_SETS"; else SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_CYCLES" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"); fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- "$PY" "$ROOT/bench.py" --workload transport --steps 20 --warmup 2 --no-cpu-baseline "$@" > "$OUT/p$i.json" 2> "$OUT/p$i.err" || { echo "pass $i failed"; tail -3 "$OUT/p$i.err"; }
done
"$PY" - "$OUT" "$TAG" <<'PYEOF'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "transport_fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print("%s %-32s %.5g  (n=%d)" % (name, c, sum(v) / len(v), len(v)))
PYEOF
find "$OUT" -name "*_kernel_trace.csv" -delete
