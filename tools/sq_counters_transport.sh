#!/bin/bash
# Counter sets: SQ / SQC / TCC counters only, at most 8 per pass.  Round 4: a pass that also asked for four TCP_*_STALL, three
# TA_* and one TD_* counter was REJECTED by the profiler ("error code 38: Request exceeds the capabilities of the hardware to
# collect": more TA counters than a TA block has), the program was aborted (SIGABRT) and rocprofv3's signal handler then never
# returned from its finalisation -- the call sat idle until gpurun's watchdog ended it seven minutes later.  That is the "hang":
# profiles/r05_pmc_pass_hang_cause.md.  Every pass therefore runs under its own `timeout -k`, and a failed pass fails the script.
# SQ counters of the fused transport kernel: bash tools/sq_counters_transport.sh TAG BUILD [bench args]   (on the GPU box;
# BUILD = default or the name of an alt build under nextsimdg_amd/lib/alt)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; N=$2; shift; shift
. "$ROOT/tools/rocprof_guard.sh"
PY="$(command -v python3)"
OUT="$ROOT/gpurun_out/${NSDG_ROUND:-r06}/sqt_$TAG"; rm -rf "$OUT"; mkdir -p "$OUT"
if [ "$N" != default ]; then export NSDG_LIB="$ROOT/nextsimdg_amd/lib/alt/$N/libnsdg.so"; else unset NSDG_LIB; fi
i=0; failed=0
if [ -n "$NSDG_PMC_SETS" ]; then mapfile -t SETS < "$ROOT/$NSDG_PMC_SETS"; else SETS=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE SQ_CYCLES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"); fi
for set in "${SETS[@]}"; do
  i=$((i+1))
  guarded_rocprof p$i "$OUT" 180 --kernel-trace --pmc $set -- "$PY" "$ROOT/bench.py" --workload transport --steps 20 --warmup 2 --no-cpu-baseline "$@" || { failed=1; break; }
done
"$PY" - "$OUT" "$TAG" <<'PYEOF'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/p*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "transport_march" in r["Kernel_Name"] or "transport_fused" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print("%s %-32s %.5g  (n=%d)" % (name, c, sum(v) / len(v), len(v)))
PYEOF
find "$OUT" -name "*_kernel_trace.csv" -delete
exit $failed
