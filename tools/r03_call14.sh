set -o pipefail
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_b14_plain.json 2> gpurun_out/r03_b14_plain.err || exit 1
NSDG_FORCE_DIST=1 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_b14_forcedist.json 2> gpurun_out/r03_b14_forcedist.err || exit 1
NSDG_BENCH_LOOPBACK_WORLD=8 python bench.py --steps 5 --warmup 2 > gpurun_out/r03_b14_loop8.json 2> gpurun_out/r03_b14_loop8.err || exit 1
NSDG_HALO_DELAY_US=10 NSDG_HALO_SIM_GBS=50 NSDG_BENCH_LOOPBACK_WORLD=8 python bench.py --steps 5 --warmup 2 > gpurun_out/r03_b14_loop8_bw50.json 2> gpurun_out/r03_b14_loop8_bw50.err || exit 1
NSDG_HALO_DELAY_US=200 NSDG_BENCH_LOOPBACK_WORLD=8 python bench.py --steps 5 --warmup 2 > gpurun_out/r03_b14_loop8_fix200.json 2> gpurun_out/r03_b14_loop8_fix200.err || exit 1
NSDG_BENCH_LOOPBACK_WORLD=4 python bench.py --steps 5 --warmup 2 --workload coupled > gpurun_out/r03_b14_loop4_coupled.json 2> gpurun_out/r03_b14_loop4_coupled.err || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t6.log 2>&1 || exit 1
