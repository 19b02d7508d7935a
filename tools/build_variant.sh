#!/bin/bash
# Builds a variant of libptmi.so with extra compiler flags (ablation / A-B builds) into tools/bin/libptmi_<name>.so;
# run anything against it with PTMI_LIB=tools/bin/libptmi_<name>.so.  usage: tools/build_variant.sh <name> <flags...>
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=/tmp/ptmi_variant_$NAME; mkdir -p "$B" "$ROOT/tools/bin"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -mllvm -amdgpu-sched-strategy=max-ilp -Wall -Wno-unused-function $*"
cd "$ROOT/pathtrace_amd/csrc"
for f in ptmi_api.hip ptmi_render.hip ptmi_comm.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o "$B/${f%.hip}.o" & done
for f in bvh_build.cpp obj_load.cpp; do /opt/rocm/bin/hipcc $FLAGS -x hip -c $f -o "$B/${f%.cpp}.o" & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/bin/libptmi_$NAME.so" "$B"/*.o -ldl
echo "built tools/bin/libptmi_$NAME.so"
