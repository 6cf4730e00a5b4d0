#!/bin/bash
# Profiles `bench.py` on the GPU box: kernel statistics, HBM-side traffic (two separate PMC passes, as the
# MI355X guide prescribes: FETCH_SIZE and WRITE_SIZE do not fit into one pass) and SQ counters.
# usage (from the repo root on the box): bash tools/profile_bench.sh <tag> [bench options]
# Output under gpurun_out/prof_<tag>/; tools/summarize_pmc.py turns it into profiles/<tag>_*.
set -o pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:-prof}"; shift
OUT="$ROOT/gpurun_out/prof_$TAG"
rm -rf "$OUT"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PY="$(command -v python3)"
"$PY" "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_unprofiled.json" 2> "$OUT/bench_unprofiled.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- "$PY" "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench_stats.json" 2> "$OUT/stats.err" || exit 1
for pass in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $pass --output-format csv -d "$OUT/$pass" -- "$PY" "$ROOT/bench.py" --steps 1 --warmup 0 --nsub 8 --no-cpu-baseline "$@" > "$OUT/$pass.json" 2> "$OUT/$pass.err" || exit 1
done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES \
  --output-format csv -d "$OUT/SQ" -- "$PY" "$ROOT/bench.py" --steps 1 --warmup 0 --nsub 8 --no-cpu-baseline "$@" > "$OUT/SQ.json" 2> "$OUT/SQ.err" || exit 1
# keep only what the summaries need (the traces are large)
find "$OUT" -name "*_kernel_trace.csv" -size +8M -delete
ls -R "$OUT" | head -50
