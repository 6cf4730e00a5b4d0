set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t5.log 2>&1 || exit 1
bash tools/r03_config5_day.sh 512 2400 > gpurun_out/r03_config5_smoke.txt 2>&1 || exit 1
bash tools/r03_config5_day.sh 4096 86400 > gpurun_out/r03_config5_day.txt 2>&1 || exit 1
