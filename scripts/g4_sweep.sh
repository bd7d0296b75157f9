#!/bin/bash
# quick timing of the default library and variants
for v in "" $@; do
  if [ -z "$v" ]; then export LD_LIBRARY_PATH=merlot_reserve_amd; name=main; else export LD_LIBRARY_PATH=merlot_reserve_amd/variants/$v; name=$v; fi
  echo "== $name"; timeout -k 10 200 scripts/micro/gemm3_test ${MODE:-quick} 6 || exit 1
done
