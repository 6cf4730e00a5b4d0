#!/bin/bash
# tools/gpu_steps.sh FILE : runs the steps listed in FILE on the GPU box, one per line as  NAME LIMIT_SECONDS COMMAND...
# Every step runs under its own `timeout -k 10`, its output goes to gpurun_out/$NSDG_ROUND/NAME.log, a progress line is printed
# per step -- and a step that TIMES OUT (or is killed) ends the call: no further GPU step is started after one that hung.
# A step that merely fails (a red test, a run that leaves the physical range) does not stop the others.
ROUND=${NSDG_ROUND:-r05}
mkdir -p gpurun_out/$ROUND
while IFS= read -r line; do
  [ -z "$line" ] && continue
  case "$line" in \#*) continue;; esac
  name=${line%% *}; rest=${line#* }; limit=${rest%% *}; cmd=${rest#* }
  echo "== $name (limit ${limit}s): $cmd"
  start=$(date +%s)
  timeout -k 10 "$limit" bash -c "$cmd" > gpurun_out/$ROUND/$name.log 2>&1
  rc=$?
  echo "   $name rc=$rc  $(( $(date +%s) - start )) s; last line: $(tail -n 1 gpurun_out/$ROUND/$name.log | cut -c1-300)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "   TIMEOUT in $name: no further GPU step in this call"; exit 1; fi
done < "$1"
exit 0
