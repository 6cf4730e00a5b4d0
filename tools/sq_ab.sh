#!/bin/bash
# SQ counters of the dominant kernel for two builds of libnsdg.so on one box.  usage: bash tools/sq_ab.sh NAME... (alt builds or `default`)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp
PY="$(command -v python3)"
for n in "$@"; do
  OUT="$ROOT/gpurun_out/sq_ab/$n"; rm -rf "$OUT"; mkdir -p "$OUT"
  if [ "$n" != default ]; then export NSDG_LIB="$ROOT/nextsimdg_amd/lib/alt/$n/libnsdg.so"; else unset NSDG_LIB; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES \
    --output-format csv -d "$OUT" -- "$PY" "$ROOT/bench.py" --steps 1 --warmup 0 --nsub 12 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/err.txt" || { tail -5 "$OUT/err.txt"; exit 1; }
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "$PY" "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_stats.json" 2> "$OUT/err2.txt"
  "$PY" - "$OUT" "$n" <<'PYEOF'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "mevp_fused3_kernel" in r["Kernel_Name"]:
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(name, {k: "%.4g" % (sum(v) / len(v)) for k, v in sorted(agg.items())})
for s in glob.glob(d + "/stats/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(s)):
        if "mevp_fused3" in r["Name"]:
            print(name, "rocprof avg ns", r["AverageNs"], "calls", r["Calls"])
PYEOF
  find "$OUT" -name "*_kernel_trace.csv" -delete
done
