"""Summarise a rocprofv3 kernel_stats.csv: per-kernel total / avg, normalised per step.
usage: prof_summary.py kernel_stats.csv [bench args as given to bench.py ...]   (steps are derived from --steps / --warmup /
--no-roofline: 1 eager + warm-up + timed + 3 x min(timed, 3) instrumented eager steps)"""
import csv, sys
path, args = sys.argv[1], sys.argv[2:]
def opt(name, default):
    return int(args[args.index(name) + 1]) if name in args else default
K, W = opt('--steps', 10), opt('--warmup', 3)
h2d = 0 if "--no-h2d" in args or "--no-graph" in args else max(3 * K, 30) + 2          # bench.py's host-fed leg
chk = 0 if "--no-graph" in args else 2                 # round 5: the one-shot graph == eager check (one eager step + one replay)
steps = 1 + W + K + (0 if "--no-roofline" in args else 3 * min(K, 3)) + h2d + chk
rows = list(csv.DictReader(open(path)))
tot = sum(float(r['TotalDurationNs']) for r in rows if 'cast_params' not in r['Name'])
if '--total-steps' in args:      # a run that is not bench.py (scripts/bench_vcr.py): the number of steps that launched kernels, given by the caller
    steps = opt('--total-steps', steps)
    print(f'# steps in the profiled run: {steps}')
else:
    print(f'# steps in the profiled run: {steps} (1 eager + {chk} of the graph == eager check + {W} warm-up + {K} timed graph replays' + ('' if '--no-roofline' in args else f' + 3 x {min(K, 3)} instrumented eager')
          + (f' + {h2d} replays fed from host memory' if h2d else '') + ')')
if '--bench-log' in args:      # the bench line of the SAME run: under rocprofv3 every dispatch is serialised and stamped, so the step is
    import json                # slower than un-profiled and ~equal to the kernel sum; the un-profiled step overlaps the towers' tails
    for line in open(args[args.index('--bench-log') + 1]):
        if line.startswith('{"metric"'):
            d = json.loads(line)
            print(f"# bench.py ms_per_step in this profiled run: {d['ms_per_step']:.2f}.  The kernel sum below need not equal it: launches of the two towers that "
                  f"overlap on two streams each count the time they share the GPU (sum > step), and rocprofv3's tracing adds ~10 us per dispatch (profiled step > un-profiled step, DESIGN.md section 3)")
print(f'total kernel ms/step (excl. one-off init): {tot / 1e6 / steps:.2f}')
fam = {}
def family(n):
    if 'gemm' in n or 'splitk' in n: return 'gemm'
    if 'attn' in n: return 'attention'
    if 'ln_' in n or 'colsum' in n or 'reduce_' in n: return 'layernorm+reductions'
    if 'adam' in n or 'nan_to_num' in n or 'cast_params' in n or 'transpose_leaves' in n: return 'optimizer'
    if 'ccl' in n.lower() or 'Reduce' in n: return 'rccl'
    return 'rowops+other'
for r in rows:
    if 'cast_params' in r['Name']:
        continue
    fam[family(r['Name'])] = fam.get(family(r['Name']), 0.0) + float(r['TotalDurationNs']) / 1e6 / steps
print('by family (ms/step): ' + ', '.join(f'{k} {v:.2f}' for k, v in sorted(fam.items(), key=lambda kv: -kv[1])))
for r in rows[:60]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('_ZN12_GLOBAL__N_1', '')[:70]
    print(f"{n:70s} calls/step={float(r['Calls']) / steps:7.1f} ms/step={float(r['TotalDurationNs']) / 1e6 / steps:7.2f} avg_us={float(r['AverageNs']) / 1e3:8.1f}")
