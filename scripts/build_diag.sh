#!/bin/bash
# scripts/build_diag.sh NAME -DFLAG [-DFLAG..] : timing-only variant of the library -> merlot_reserve_amd/libdiag_NAME.so
cd "$(dirname "$0")/.." && name=$1 && shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -x hip -c merlot_reserve_amd/csrc/gemm256.hip -o /tmp/gemm256_$name.o &&
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o merlot_reserve_amd/libdiag_$name.so /tmp/gemm256_$name.o $(ls merlot_reserve_amd/build/*.o | grep -v gemm256.o) && echo built libdiag_$name.so
