# BASELINE config 5 as a run through the C++ host: 4096 x 4096 DG2 dynamics + column thermodynamics, ONE MODEL DAY (720 steps),
# once as a single block and once as 8 row blocks (threads of one process, in-process transport, one GPU): the two restart
# files must be identical byte for byte, and the fields finite.  usage: [CICE=1.0] bash tools/config5_day.sh [n=4096] [stop=86400]
# (rounds 3-4: CICE=0.9 on 4096 x 4096 left the physical range after 22-23 model hours with alpha = beta from the stability bound at
# Delta_min = 2e-9; round 5 ran alpha = beta = 1500 with the mesh's Delta_min and the closure (--dynamics.subcycle=keep_alpha), round 6 runs local, adaptive alpha and beta at Delta_min = 2e-9 (the default):
# profiles/r05_closure.md, profiles/r05_config5_one_day_host.txt.  --dynamics.delta_min=2e-9 restores the old configuration)
set -o pipefail
N=${1:-4096}; STOP=${2:-86400}
BIN=nextsimdg_amd/host/build/nextsim_amd
COMMON="--Modules.Nextsim::IModelStep=Nextsim::DynamicsStep --model.structure=rectgrid --model.init_file= --rectgrid.nx=$N --rectgrid.ny=$N --init.hice=0.3 --init.cice=${CICE:-1.0} --init.sst=-1.76 --init.hsnow=0.05 --init.tice=-8 --dynamics.thermodynamics=true --dynamics.forcing=winter --model.start=0 --model.stop=$STOP --model.time_step=120 --model.timing=true"
for B in 1 8; do
  echo "=== row_blocks = $B"
  T0=$(date +%s)
  $BIN $COMMON --dynamics.row_blocks=$B --model.final_file=/tmp/nsdg_cfg5_rb$B.nsdg 2>&1 | tee /tmp/nsdg_cfg5_out_$B.txt | tail -14 || exit 1
  echo "wall $(( $(date +%s) - T0 )) s"
done
ls -l /tmp/nsdg_cfg5_rb1.nsdg /tmp/nsdg_cfg5_rb8.nsdg
if grep -q "sumH=nan\|sumH=-nan\|sumH=inf" /tmp/nsdg_cfg5_out_*.txt; then echo "NON-FINITE fields"; exit 1; fi
if cmp /tmp/nsdg_cfg5_rb1.nsdg /tmp/nsdg_cfg5_rb8.nsdg; then echo "restart files of 1 block and 8 blocks: IDENTICAL byte for byte"; else echo "restart files DIFFER"; exit 1; fi
rm -f /tmp/nsdg_cfg5_rb1.nsdg /tmp/nsdg_cfg5_rb8.nsdg
