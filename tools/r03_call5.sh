set -o pipefail
tools/microbench/copy_peak > gpurun_out/r03_copy_peak.txt 2>&1 || exit 1
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t3.log 2>&1 || exit 1
./nextsimdg_amd/host/build/host_tests --gpu > gpurun_out/r03_host_gpu.log 2>&1 || exit 1
python tools/rank_share_timing.py --native --k 3 1 > gpurun_out/r03_share_k3.txt 2>&1 || exit 1
for cfg in "0 0" "10 50"; do set -- $cfg; echo "=== latency $1 us, bandwidth $2 GB/s per direction (0 = none)" >> gpurun_out/r03_share_k3.txt; NSDG_HALO_DELAY_US=$1 NSDG_HALO_SIM_GBS=$2 python tools/rank_share_timing.py --native --halo native --rccl-loopback --k 3,8 4 8 >> gpurun_out/r03_share_k3.txt 2>&1 || exit 1; done
NSDG_SOAK_EVERY=60 python tools/soak_coupled.py 720 4096 winter > gpurun_out/r03_soak_4096.txt 2>&1 || exit 1
