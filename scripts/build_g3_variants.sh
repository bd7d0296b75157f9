#!/bin/bash
# Diagnostic / experiment builds of the ping-pong GEMM: gemm3.hip recompiled with -D flags and linked with the shipped objects
# into merlot_reserve_amd/variants/<name>/libmreserve_hip.so.  Usage: scripts/build_g3_variants.sh name:-DFLAG[,-DFLAG] ...
set -e
cd "$(dirname "$0")/.."
python -m merlot_reserve_amd.build >/dev/null
B=merlot_reserve_amd/build
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; flags=${flags//,/ }
  mkdir -p merlot_reserve_amd/variants/$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -x hip -c merlot_reserve_amd/csrc/gemm3.hip -o /tmp/gemm3_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o merlot_reserve_amd/variants/$name/libmreserve_hip.so \
      $B/gemm.o $B/gemm256.o $B/attention.o $B/layernorm.o $B/rowops.o $B/adam.o $B/f32path.o $B/mr_error.o $B/comm.o /tmp/gemm3_$name.o -ldl
  echo built variant $name "($flags)"
done
