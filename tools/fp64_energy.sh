#!/bin/bash
# Socket power (rocm-smi) while the vector / the matrix fp64 pipe run flat out: bash tools/fp64_energy.sh  -> gpurun_out/r04/fp64_energy.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/r04/fp64_energy.txt"; mkdir -p "$ROOT/gpurun_out/r04"; : > "$OUT"
BIN="$ROOT/tools/microbench/fp64_energy"
for run in "idle 6 4" "fma 10 4" "fma 10 8" "fma 10 16" "mfma 10 4" "mfma 10 8" "mfma 10 16" "idle 4 4"; do
  set -- $run
  echo "=== $1, $3 waves per CU" >> "$OUT"
  "$BIN" $1 $2 $3 >> "$OUT" 2>&1 &
  PID=$!
  sleep 4
  for i in 1 2 3 4; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk" | sed 's/^.*: //' | tr '\n' ' ' >> "$OUT"; echo >> "$OUT"
    sleep 1
  done
  wait $PID
done
cat "$OUT"
