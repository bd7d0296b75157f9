#!/bin/bash
# Timing-only ablation builds of the two-workgroups-per-CU GEMM (csrc/gemm5.hip): the library with gemm5.o compiled under one -D switch each,
# into merlot_reserve_amd/variants/<name>/libmreserve_hip.so (results are WRONG in these builds; only `gemm3_test g5time` is meaningful).
set -e
cd "$(dirname "$0")/.."
python -c "from merlot_reserve_amd.build import build; build()"
B=merlot_reserve_amd/build
OBJS=$(ls $B/*.o | grep -v gemm5.o)
for v in "base:" "noload:-DMR_DIAG_NOLOAD" "noread:-DMR_G5_NOREAD" "nobar:-DMR_G5_NOBAR" "nostore:-DMR_G3_NOSTORE" "noload_noread:-DMR_DIAG_NOLOAD -DMR_G5_NOREAD" "$@"; do
  name=${v%%:*}; flags=${v#*:}
  mkdir -p merlot_reserve_amd/variants/$name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 $flags -x hip -c merlot_reserve_amd/csrc/gemm5.hip -o merlot_reserve_amd/variants/$name/gemm5.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-z,defs -o merlot_reserve_amd/variants/$name/libmreserve_hip.so $OBJS merlot_reserve_amd/variants/$name/gemm5.o -ldl
  echo built $name "($flags)"
done
