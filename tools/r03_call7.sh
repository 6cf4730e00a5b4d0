set -o pipefail
bash tools/profile_bench.sh r03 > gpurun_out/r03_profile.log 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r03_share8" -- "$(command -v python3)" "$GRAFT_REPO_ROOT/tools/rank_share_timing.py" --native --halo native --rccl-loopback --k 3 8 > "$GRAFT_REPO_ROOT/gpurun_out/r03_share8_prof.txt" 2>&1 || exit 1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof_r03_share8 -name "*_kernel_trace.csv" -size +8M -delete
python bench.py --steps 20 --warmup 5 > gpurun_out/r03_bench_final.json 2> gpurun_out/r03_bench_final.err || exit 1
