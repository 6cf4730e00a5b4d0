# tools/rocprof_guard.sh -- sourced by the profiling scripts: ONE guarded way to run rocprofv3 on the GPU box.
# Round 4: a pass whose counter set the profiler rejected ("error code 38: Request exceeds the capabilities of the hardware to
# collect") aborted the program (SIGABRT) and rocprofv3's signal handler never returned -- the call sat idle until gpurun's watchdog
# ended it seven minutes later (profiles/r05_pmc_pass_hang_cause.md).  Hence: every pass under its own `timeout -k`, sets with TCP / TA /
# TD counters refused, the error line of a failed pass printed, and a non-zero status so that the calling script stops.
#   guarded_rocprof NAME OUTDIR LIMIT_SECONDS <rocprofv3 options ...> -- <program> <args ...>
# writes the program's stdout to OUTDIR/NAME.json and its stderr to OUTDIR/NAME.err; the profiler's files go under OUTDIR/NAME/.
cd /tmp && export TMPDIR=/tmp
guarded_rocprof() {
  local name=$1 out=$2 limit=$3; shift 3
  case " $* " in *" TCP_"*|*" TA_"*|*" TD_"*) echo "$name: refusing a counter set with TCP / TA / TD counters"; return 2;; esac
  timeout -k 10 "$limit" rocprofv3 --output-format csv -d "$out/$name" "$@" > "$out/$name.json" 2> "$out/$name.err"
  local rc=$?
  if [ $rc -ne 0 ]; then
    echo "$name FAILED (status $rc): $(grep -m1 -i 'error code\|Could not\|rocprofv3: error' "$out/$name.err")"
    tail -3 "$out/$name.err"
  fi
  return $rc
}
