#!/bin/bash
# tools/ab_variants.sh TAG "V1 V2 ..." [bench args]: alternating bench.py runs of mEVP kernel variants (or alt builds
# "name:variant" under nextsimdg_amd/lib/alt/) on ONE box; writes gpurun_out/${NSDG_ROUND:-r05}/TAG_*.json and prints a summary
tag=$1; shift; list=$1; shift
mkdir -p gpurun_out/${NSDG_ROUND:-r05}
for item in $list; do
  name=${item%%:*}; v=${item##*:}
  lib=""; [ "$name" != "$v" ] && lib="nextsimdg_amd/lib/alt/$name/libnsdg.so"
  f=gpurun_out/${NSDG_ROUND:-r05}/${tag}_${name}_$RANDOM
  if [ -n "$lib" ]; then export NSDG_LIB=$lib; else unset NSDG_LIB; fi
  timeout -k 10 300 python bench.py --variant $v --steps 8 --warmup 2 --no-cpu-baseline "$@" > $f.json 2> $f.err || { echo "$item FAILED"; tail -3 $f.err; continue; }
  python - "$f.json" "$item" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = j["roofline"]
print(sys.argv[2], "ms/step %.3f" % j["ms_per_step"], "pass ms %.4f" % r["avg_launch_ms"], r.get("kernel"), "guard", j.get("fused_pass_guard", j.get("checks")), flush=True)
PY
done
