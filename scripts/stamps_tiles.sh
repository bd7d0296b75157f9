#!/bin/bash
# per-item k-loop / epilogue cycles (s_memtime, 100 MHz) of one GEMM shape under each forced tile width
for bn in 128 192 256; do
  echo "=== bn $bn  shape $@"
  MR_G256_BN=$bn MR_LIB=$GRAFT_REPO_ROOT/merlot_reserve_amd/libdiag_stamps.so python scripts/stamps.py "$@"
done
