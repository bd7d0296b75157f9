#!/bin/bash
# A/B an environment switch in one gpurun call: scripts/ab_env.sh "VAR=a" "VAR=b" <script> [args]
a=$1; b=$2; shift 2
for r in 1 2; do
  echo "--- $a"; env $a python "$@" | awk '{print $1, $2, $3}' | tr "\n" ";"; echo
  echo "--- $b"; env $b python "$@" | awk '{print $1, $2, $3}' | tr "\n" ";"; echo
done
