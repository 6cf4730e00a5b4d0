#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel statistics, HBM-side traffic (two separate PMC passes, as the
# MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit into one pass) and SQ counters.
# usage (from the repo root on the box): bash tools/profile_bench.sh <tag> [bench options]
# Output under gpurun_out/prof_<tag>/; tools/summarize_pmc.py turns it into profiles/<tag>_*.
# Every pass is guarded (tools/rocprof_guard.sh): its own time limit, a failed pass ends the script with a non-zero status.
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:-prof}"; shift
OUT="$ROOT/gpurun_out/prof_$TAG"
rm -rf "$OUT"
mkdir -p "$OUT"
. "$ROOT/tools/rocprof_guard.sh"
PY="$(command -v python3)"
timeout -k 10 300 "$PY" "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err" || exit 1
guarded_rocprof stats "$OUT" 300 --kernel-trace --stats -- "$PY" "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" || exit 1
for pass in FETCH_SIZE WRITE_SIZE; do
  guarded_rocprof $pass "$OUT" 240 --kernel-trace --pmc $pass -- "$PY" "$ROOT/bench.py" --steps 1 --warmup 0 --nsub 8 --no-cpu-baseline "$@" || exit 1
done
guarded_rocprof SQ "$OUT" 240 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES \
  -- "$PY" "$ROOT/bench.py" --steps 1 --warmup 0 --nsub 8 --no-cpu-baseline "$@" || exit 1
# keep only what the summaries need (the traces are large)
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
ls -R "$OUT" | head -50
