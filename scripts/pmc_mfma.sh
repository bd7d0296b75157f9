#!/bin/bash
# Matrix-core utilisation of the step's kernels from the SQ counters (one --pmc pass, eager launches so that every dispatch is
# attributed; counters only with --kernel-trace, as gpurun requires).  Output: gpurun_out/pmc_mfma_<tag>.json
# usage (GPU box, repo root):  bash scripts/pmc_mfma.sh <tag> [bench args]
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python3 $root/bench.py --no-graph --steps 2 --warmup 1 --no-cpu-baseline --no-calibration --no-roofline --no-h2d "$@" > /tmp/pmc_mfma.log 2>&1
tail -2 /tmp/pmc_mfma.log | cut -c1-300
python3 - "$@" <<'PY' > $out/pmc_mfma_$tag.json
import csv, glob, json, re, sys, collections
cc = glob.glob('/tmp/pmc_mfma/*/*counter_collection.csv')
kt = glob.glob('/tmp/pmc_mfma/*/*kernel_trace.csv')
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(cc[0])):
    k = re.sub(r"\((?!anonymous).*", "", r["Kernel_Name"]).replace("void ", "")
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r['Dispatch_Id'] not in seen[k]:
        seen[k].add(r['Dispatch_Id'])
        agg[k]['_ns'] += dur.get(r['Dispatch_Id'], 0)
        agg[k]['_launches'] += 1
res = {}
SIMDS = 256 * 4
for k, a in agg.items():
    ns = a['_ns']
    if ns <= 0: continue
    # (no clock figure: GRBM_GUI_ACTIVE / 8 / time reads 3-4 GHz on these sub-0.3-ms dispatches -- MI355X_MICROARCH.md, DVFS note: the quotient is only
    # usable for dispatches of ~10 ms; the utilisation ratios below are cycle / cycle and unaffected)
    mfma_busy = a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
    res[k] = {'launches': int(a['_launches']), 'ms_total': ns / 1e6,
              # busy cycles of the matrix pipes / (SIMDs x elapsed cycles): the fraction of the chip's MFMA issue slots in use
              'mfma_pipe_util': round(mfma_busy / (SIMDS * a.get('GRBM_GUI_ACTIVE', 1.0) / 8.0), 4) if a.get('GRBM_GUI_ACTIVE') else None,
              'mfma_tflops_executed': round(a.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0) * 512 / ns / 1e3, 1),
              'valu_insts_per_mfma_mop': round(a.get('SQ_INSTS_VALU', 0.0) / max(a.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0), 1.0), 3),
              'wave_cycles_waiting_frac': round(a.get('SQ_WAIT_ANY', 0.0) / max(a.get('SQ_WAVE_CYCLES', 0.0), 1.0), 3),
              'raw': {c: v for c, v in a.items() if not c.startswith('_')}}
print(json.dumps({'bench_args': sys.argv[1:], 'note': 'per kernel, summed over the dispatches of 4 eager steps (1 allocation + 1 warm-up + 2 timed); '
                  'mfma_pipe_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); mfma_tflops_executed = '
                  'SQ_INSTS_VALU_MFMA_MOPS_BF16 x 512 FLOP / kernel time (includes padded tile work)', 'kernels': res}, indent=1, sort_keys=True))
PY
python3 - <<PY
import json
d = json.load(open('$out/pmc_mfma_$tag.json'))['kernels']
for k, v in sorted(d.items(), key=lambda kv: -kv[1]['ms_total'])[:14]:
    print(f"{k[:60]:60s} ms={v['ms_total']:8.2f} util={v['mfma_pipe_util']} TF/s={v['mfma_tflops_executed']} wait={v['wave_cycles_waiting_frac']}")
PY
