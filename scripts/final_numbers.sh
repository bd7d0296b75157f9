#!/bin/bash
# every measured configuration of DESIGN.md section 3 in one gpurun call -> gpurun_out/final_numbers.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}; out=$root/gpurun_out/final_numbers.txt; : > $out
j() { python3 -c "import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); c=d['config']; print('$1', 'ms', round(d['ms_per_step'],2), 'vseg/s', round(d['value'],1), 'mfma', round(c.get('step_mfma_frac',0),4), 'h2d', c.get('ms_per_step_inputs_from_host'), 'roof', round((d.get('roofline') or {}).get('frac',0),4))" >> $out; }
python3 $root/bench.py --no-cpu-baseline 2>/dev/null | j base_b4
python3 $root/bench.py --no-cpu-baseline --no-h2d --model large 2>/dev/null | j large_b4
python3 $root/bench.py --no-cpu-baseline --no-h2d --no-roofline --model large --records-per-gpu 8 2>/dev/null | j large_b8
python3 $root/bench.py --no-cpu-baseline --no-h2d --no-roofline --model large --resadapt --records-per-gpu 2 2>/dev/null | j large_resadapt_b2
python3 $root/bench.py --no-cpu-baseline --no-h2d --no-roofline --resadapt --records-per-gpu 2 2>/dev/null | j base_resadapt_b2
python3 $root/bench.py --no-cpu-baseline --no-h2d --no-roofline --force-comm 2>/dev/null | j base_forcecomm
python3 $root/bench.py --no-cpu-baseline --no-h2d --no-roofline --force-comm --model large 2>/dev/null | j large_forcecomm
python3 $root/bench.py --no-cpu-baseline --no-h2d --no-roofline --records-per-gpu 8 2>/dev/null | j base_b8
python3 $root/scripts/bench_vcr.py large 4 2>/dev/null | tail -2 >> $out
python3 $root/scripts/bench_vcr.py base 4 2>/dev/null | tail -2 >> $out
cat $out
