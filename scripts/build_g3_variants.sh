#!/bin/bash
# Diagnostic / experiment builds of the ping-pong GEMM: gemm3.hip (or SRC=gemm4) recompiled with -D flags and linked with the shipped
# objects into merlot_reserve_amd/variants/<name>/libmreserve_hip.so.  Usage: [SRC=gemm4] scripts/build_g3_variants.sh name:-DFLAG[,-DFLAG] ...
set -e
cd "$(dirname "$0")/.."
python -m merlot_reserve_amd.build >/dev/null
B=merlot_reserve_amd/build
SRC=${SRC:-gemm3}
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  mkdir -p merlot_reserve_amd/variants/$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -x hip -c merlot_reserve_amd/csrc/$SRC.hip -o /tmp/${SRC}_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o merlot_reserve_amd/variants/$name/libmreserve_hip.so \
      $(ls $B/*.o | grep -v "/$SRC.o") /tmp/${SRC}_$name.o -ldl
  echo built variant $name "($flags)"
done
