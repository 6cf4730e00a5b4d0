"""A/B timing of alternative builds of libnsdg.so on one box (box-to-box spread is 5-15 %, so variants are
only comparable within one call).  usage: python tools/ab_bench.py [--rounds 2] [--args "..."] NAME...
where NAME is a directory under nextsimdg_amd/lib/alt/ or 'default'."""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = 2
extra = "--steps 2 --warmup 1 --no-cpu-baseline"
while args and args[0].startswith("--"):
    if args[0] == "--rounds":
        rounds = int(args[1])
    elif args[0] == "--args":
        extra = args[1]
    args = args[2:]
res = {n: [] for n in args}
for r in range(rounds):
    for n in args:
        env = dict(os.environ)
        if n != "default":
            env["NSDG_LIB"] = os.path.join(root, "nextsimdg_amd", "lib", "alt", n, "libnsdg.so")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra.split(), env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(n, "FAILED", out.stderr[-500:], flush=True)
            continue
        j = json.loads(line[-1])
        res[n].append((j["ms_per_step"], j.get("roofline", {}).get("avg_launch_ms")))
        print(n, "ms/step %.3f" % j["ms_per_step"], "launch ms", j.get("roofline", {}).get("avg_launch_ms"), flush=True)
print(json.dumps(res))
