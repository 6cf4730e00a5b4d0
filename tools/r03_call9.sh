set -o pipefail
bash tools/r03_config5_day.sh 512 2400 > gpurun_out/r03_config5_smoke.txt 2>&1 || exit 1
bash tools/r03_config5_day.sh 4096 86400 > gpurun_out/r03_config5_day.txt 2>&1 || exit 1
