#!/bin/bash
# The bench lines and profiles of a round at HEAD, on the GPU box: bash tools/profile_round.sh TAG
# writes gpurun_out/TAG/*.json (+ prof_TAG/ from tools/profile_bench.sh); copy / summarise into profiles/ afterwards.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
run() { name=$1; shift; echo "== $name: $*"; timeout -k 10 400 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name FAILED"; tail -3 $O/$name.err; }; }
run bench_default python bench.py --steps 20 --warmup 3
run config3_1024 python bench.py --nx 1024 --ny 1024 --steps 20 --warmup 3
run coupled_4096 python bench.py --workload coupled --nx 4096 --ny 4096 --steps 3 --warmup 1
run column python bench.py --workload column --nx 4096 --ny 4096 --steps 20 --warmup 2
run transport_dg1_512 python bench.py --workload transport --order 1 --nx 512 --ny 512 --steps 4000 --warmup 50
run transport_dg2_2048 python bench.py --workload transport --order 2 --nx 2048 --ny 2048 --steps 200 --warmup 10
NSDG_BENCH_LOOPBACK_WORLD=8 run rehearsal_loop8 python bench.py --steps 5 --warmup 2
bash tools/profile_bench.sh $TAG > $O/profile_bench.log 2>&1 || echo "profile_bench FAILED"
for f in $O/*.json; do python - "$f" <<'PY'
import json, sys
try:
    j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); r = j.get("roofline", {})
    print(sys.argv[1], "%.4g" % j["value"], "%.4f ms" % j["ms_per_step"], "frac", r.get("frac"), "traffic", r.get("traffic"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
