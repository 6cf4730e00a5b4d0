set -o pipefail
A=nextsimdg_amd/lib/alt
for r in 1 2; do for n in default stagger2k stagger4k; do
  if [ $n = default ]; then unset NSDG_LIB; else export NSDG_LIB=$PWD/$A/$n/libnsdg.so; fi
  echo "## $n" >> gpurun_out/r03_stagger.txt
  python tools/rank_share_timing.py --native --halo native --rccl-loopback --k 3 8 2>&1 | grep "step " >> gpurun_out/r03_stagger.txt || exit 1
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('single block ms/step', j['ms_per_step'], 'pass ms', j['roofline']['avg_launch_ms'], 'copy peak', j['roofline']['copy_peak_GBs'])" >> gpurun_out/r03_stagger.txt || exit 1
done; done
