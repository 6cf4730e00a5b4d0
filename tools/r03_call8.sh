set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t4.log 2>&1 || exit 1
bash tools/r03_call7.sh || exit 1
NSDG_FORCE_DIST=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench_forcedist.json 2> gpurun_out/r03_bench_forcedist.err || exit 1
python bench.py --workload column --nx 4096 --ny 4096 --steps 20 --warmup 3 > gpurun_out/r03_column_bench.json 2> gpurun_out/r03_column_bench.err || exit 1
python bench.py --workload coupled --nx 4096 --ny 4096 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r03_coupled_4096.json 2> gpurun_out/r03_coupled_4096.err || exit 1
python bench.py --workload transport --nx 512 --order 1 --steps 200 --warmup 20 > gpurun_out/r03_transport_dg1_512.json 2> gpurun_out/r03_transport_dg1_512.err || exit 1
python bench.py --workload transport --nx 2048 --order 2 --steps 30 --warmup 5 > gpurun_out/r03_transport_dg2_2048.json 2> gpurun_out/r03_transport_dg2_2048.err || exit 1
python bench.py --nx 1024 --ny 1024 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03_config3_1024.json 2> gpurun_out/r03_config3_1024.err || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1 || exit 1
