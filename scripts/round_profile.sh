#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command -> gpurun_out/prof_<tag>/ (+ summary text)
# usage: bash scripts/round_profile.sh <tag> [bench args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$root/gpurun_out/prof_$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $root/bench.py --no-cpu-baseline --no-calibration "$@" > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-400
f=$(ls /tmp/prof_$tag/*/*kernel_stats.csv | head -1)
cp $f $out/kernel_stats.csv
python3 $root/scripts/prof_summary.py $out/kernel_stats.csv "$@" --bench-log $out/bench.log > $out/summary.txt
head -40 $out/summary.txt
