#!/bin/bash
# tools/ab_transport.sh "BUILD ..." [bench args]: alternating transport bench.py runs of alt builds (nextsimdg_amd/lib/alt/BUILD;
# "default" = the in-tree library) on ONE box; prints microseconds per step
list=$1; shift
mkdir -p gpurun_out/${NSDG_ROUND:-r06}
for rep in 1 2; do
for name in $list; do
  if [ "$name" != default ]; then export NSDG_LIB=nextsimdg_amd/lib/alt/$name/libnsdg.so; else unset NSDG_LIB; fi
  f=gpurun_out/${NSDG_ROUND:-r06}/abt_${name}_$RANDOM
  timeout -k 10 300 python bench.py --workload transport --no-cpu-baseline "$@" > $f.json 2> $f.err || { echo "$name FAILED"; tail -3 $f.err; continue; }
  python - "$f.json" "$name" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[2], "us/step %.2f" % (1e3 * j["ms_per_step"]), "%.3e" % j["value"], j["roofline"].get("self_check"), flush=True)
PY
done
done
