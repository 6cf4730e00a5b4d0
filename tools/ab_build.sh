#!/bin/bash
# tools/ab_build.sh NAME [extra hipcc flags...] : builds nextsimdg_amd/lib/alt/NAME/libnsdg.so with the
# extra flags applied to the mEVP sources only (A/B experiments; see tools/ab_bench.py)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=nextsimdg_amd/lib/alt/$name
mkdir -p $out
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -fno-signed-zeros -ffp-contract=on -Wno-unused-function"
for s in column_step transport mevp mevp_fused mevp_fused4; do
  hipcc $FLAGS "$@" -c nextsimdg_amd/csrc/$s.hip -o $out/$s.o &
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libnsdg.so nextsimdg_amd/lib/nsdg_ctx.o nextsimdg_amd/lib/halo.o nextsimdg_amd/lib/rowblock.o \
  nextsimdg_amd/lib/forcing.o $out/column_step.o $out/transport.o $out/mevp.o $out/mevp_fused.o $out/mevp_fused4.o -ldl -lpthread
echo built $out/libnsdg.so
