#!/bin/bash
# Diagnostic builds of the ping-pong GEMM (timing-only / instrumented): gemm3.hip recompiled with -D flags and linked with the
# shipped objects into merlot_reserve_amd/variants/<name>/libmreserve_hip.so.  Usage: scripts/build_g3_variants.sh name:-DFLAG[,-DFLAG] ...
set -e
cd "$(dirname "$0")/.."
python -m merlot_reserve_amd.build >/dev/null
B=merlot_reserve_amd/build
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  mkdir -p merlot_reserve_amd/variants/$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -x hip -c merlot_reserve_amd/csrc/gemm3.hip -o $B/gemm3_$name.o
  objs=$(ls $B/*.o | grep -v "gemm3")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o merlot_reserve_amd/variants/$name/libmreserve_hip.so $objs $B/gemm3_$name.o -ldl
  echo built variant $name "($flags)"
done
