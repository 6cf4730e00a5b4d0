#!/bin/bash
# Samples rocm-smi (socket power, shader clock, temperature) while bench.py runs: is the sub-cycle clock- / power-limited?
# usage (on the GPU box): bash tools/power_probe.sh [bench options]  -> gpurun_out/power_probe.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/power_probe.txt"
mkdir -p "$ROOT/gpurun_out"
rocm-smi --showmaxpower --showpower --showclocks --showperflevel > "$OUT" 2>&1
python3 "$ROOT/bench.py" --steps 300 --warmup 2 --no-cpu-baseline "$@" > "$ROOT/gpurun_out/power_probe_bench.json" 2> /dev/null &
BPID=$!
sleep 6
for i in 1 2 3 4 5 6; do
  echo "--- sample $i" >> "$OUT"
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" >> "$OUT"
  sleep 1
done
wait $BPID
echo "--- idle" >> "$OUT"
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" >> "$OUT"
cut -c 1-200 "$ROOT/gpurun_out/power_probe_bench.json" >> "$OUT"
cat "$OUT"
