#!/bin/bash
# The C++ host layer's CPU tests under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are not available
# on the test pool).  usage: bash tools/asan_host_tests.sh   -> "host CPU tests: N checks, 0 failures" and no sanitizer report
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
H=$ROOT/nextsimdg_amd/host
OUT=${TMPDIR:-/tmp}/nsdg_asan
mkdir -p $OUT
g++ -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -std=c++17 -D__HIP_PLATFORM_AMD__ -I$H/include -I/opt/rocm/include \
    $H/src/Configurator.cpp $H/src/ModuleLoader.cpp $H/src/Iterator.cpp $H/src/RectGrid.cpp $H/src/Hdf5Subset.cpp $H/src/PhysicsModules.cpp \
    $H/src/HipStep.cpp $H/src/DynamicsStep.cpp $H/src/Rendezvous.cpp $H/src/Timer.cpp $H/test/host_tests.cpp -o $OUT/host_tests_asan \
    -L$ROOT/nextsimdg_amd/lib -lnsdg -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,$ROOT/nextsimdg_amd/lib -Wl,-rpath,/opt/rocm/lib
NSDG_GOLDEN_DIR=$ROOT/tests/golden ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 $OUT/host_tests_asan
