#!/bin/bash
# A/B two builds of the library in one gpurun call: scripts/ab.sh <script> [args]   (reference = libref_a.so)
for r in 1 2; do
  echo "--- ref"; MR_LIB=$GRAFT_REPO_ROOT/merlot_reserve_amd/libref_a.so python "$@" | awk '{print $1, $2, $3}' | tr "\n" ";"; echo
  echo "--- new"; python "$@" | awk '{print $1, $2, $3}' | tr "\n" ";"; echo
done
