"""Concurrency of a replayed step from a rocprofv3 kernel trace (run on the GPU box):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4/tr -o tr -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-h2d --no-secondary
    python3 scripts/trace_overlap.py gpurun_out/r4/tr
Steps are cut at the Adam launches (five per step, the last one ends a step).  Per step: wall time, time with no kernel running, with exactly one, with two
or more, the summed kernel time, the gaps between consecutive dependent kernels of the busiest queue.  (The tracer adds several us per dispatch: read the
fractions, not the absolute step time.)"""
import csv
import glob
import json
import sys

root = sys.argv[1]
files = glob.glob(root + '/**/*kernel_trace.csv', recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Queue_Id', '0')))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[2]]
steps = []
for k in range(4, len(adam), 5):
    steps.append(adam[k])
out = []
for a, b in zip(steps[-8:-1], steps[-7:]):
    ev = rows[a + 1:b + 1]
    if len(ev) < 100:
        continue
    t0, t1 = min(e[0] for e in ev), max(e[1] for e in ev)
    pts = sorted([(e[0], 1) for e in ev] + [(e[1], -1) for e in ev])
    lvl, last, hist = 0, t0, {}
    for t, d in pts:
        hist[min(lvl, 3)] = hist.get(min(lvl, 3), 0) + (t - last)
        last = t
        lvl += d
    queues = {}
    for e in ev:
        queues.setdefault(e[3], []).append(e)
    qinfo = {}
    for q, es in queues.items():
        es.sort()
        gaps = [max(0, es[i + 1][0] - es[i][1]) for i in range(len(es) - 1)]
        small = [g for g in gaps if g < 20000]
        qinfo[q] = {'kernels': len(es), 'busy_ms': sum(e[1] - e[0] for e in es) / 1e6, 'gaps_lt20us_ms': sum(small) / 1e6, 'median_gap_us': sorted(gaps)[len(gaps) // 2] / 1e3 if gaps else 0}
    out.append({'kernels': len(ev), 'wall_ms': (t1 - t0) / 1e6, 'sum_kernel_ms': sum(e[1] - e[0] for e in ev) / 1e6, 'idle_ms': hist.get(0, 0) / 1e6,
                'one_ms': hist.get(1, 0) / 1e6, 'two_ms': hist.get(2, 0) / 1e6, 'three_plus_ms': hist.get(3, 0) / 1e6, 'queues': qinfo})
print(json.dumps(out[-3:], indent=1))

# ---- per family: how much of its kernel time ran while a kernel of ANOTHER queue was running too (last full step)
def fam(n):
    if 'adam' in n or 'transpose_leaves' in n or 'nan_to_num' in n:
        return 'optimizer'
    if 'gemm' in n or 'splitk' in n:
        return 'gemm'
    if 'attn' in n:
        return 'attention'
    if 'ln_' in n or 'colsum' in n or 'reduce_batch' in n:
        return 'layernorm+reductions'
    return 'other'


a, b = steps[-2], steps[-1]
ev = rows[a + 1:b + 1]
res = {}
for i, e in enumerate(ev):
    ov = 0
    for j, o in enumerate(ev):
        if j != i and o[3] != e[3] and o[0] < e[1] and o[1] > e[0]:
            ov = max(ov, min(e[1], o[1]) - max(e[0], o[0]))
    r = res.setdefault((fam(e[2]), e[3]), [0, 0.0, 0.0])
    r[0] += 1
    r[1] += (e[1] - e[0]) / 1e6
    r[2] += ov / 1e6
print(json.dumps({f'{k[0]} (queue {k[1]})': {'kernels': v[0], 'busy_ms': round(v[1], 3), 'of_which_beside_another_queue_ms': round(v[2], 3)} for k, v in sorted(res.items())}, indent=1))
