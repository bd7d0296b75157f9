#!/bin/bash
# scripts/build_diag.sh NAME SRC -DFLAG [-DFLAG..] : timing-only variant of the library with csrc/SRC.hip rebuilt under the flags
# -> merlot_reserve_amd/libdiag_NAME.so   (select it with MR_LIB=...; the other objects come from merlot_reserve_amd/build/)
cd "$(dirname "$0")/.." && name=$1 && src=$2 && shift 2
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -x hip -c merlot_reserve_amd/csrc/$src.hip -o /tmp/${src}_$name.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o merlot_reserve_amd/libdiag_$name.so /tmp/${src}_$name.o $(ls merlot_reserve_amd/build/*.o | grep -v "/$src.o") -ldl && echo built libdiag_$name.so
