#!/bin/bash
# time the same GEMM shapes with the timing-only library variants (scripts/build_diag.sh): which resource bounds the k-loop
for v in "" nomfma noload nofrag noloadfrag; do
  if [ -z "$v" ]; then lib=$GRAFT_REPO_ROOT/merlot_reserve_amd/libmreserve_hip.so; else lib=$GRAFT_REPO_ROOT/merlot_reserve_amd/libdiag_$v.so; fi
  echo "--- ${v:-full}"; MR_LIB=$lib python scripts/bench_gemm_shapes.py "$@" | awk '{print $1, $4, $5}' | tr "\n" ";"; echo
done
