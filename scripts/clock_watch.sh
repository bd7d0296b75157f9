#!/bin/bash
# sample GPU clock / power while the step benchmark runs (is the step power-limited?)
root=${GRAFT_REPO_ROOT:-$(pwd)}
python $root/bench.py --steps 150 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/cw_bench.log 2>&1 &
pid=$!
sleep 25
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (junction|edge)" | tr -s ' ' | tr '\n' ';'; echo
  sleep 1
done
wait $pid
tail -1 /tmp/cw_bench.log | python3 -c "import sys,json; print('ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr -s ' ' | tr '\n' ';'; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i power
