"""Timeline of ONE graph-replayed step from gpurun_out/trace/kernel_trace_small.csv (scripts/trace_step.sh): how long the GPU ran
0 / 1 / 2+ kernels at once, and which kernel family was running ALONE for how long (the serial part of the step)."""
import csv, sys, collections
path = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/trace/kernel_trace_small.csv'
rows = [(int(r['start_ns']), int(r['end_ns']), r['name']) for r in csv.DictReader(open(path))]
def fam(n):
    if 'gemm' in n or 'splitk' in n: return 'gemm'
    if 'attn' in n: return 'attention'
    if 'ln_' in n or 'colsum' in n or 'reduce_' in n: return 'ln+reductions'
    if 'adam' in n or 'nan_to_num' in n: return 'optimizer'
    return 'other'
# one step = from one pad_cols launch (first kernel of the audio forward, once per step) to the next
marks = [st for st, e, n in rows if 'pad_cols' in n]
if len(marks) < 2:
    print('need two step boundaries, got', len(marks)); sys.exit(1)
t0, t1 = marks[-2], marks[-1]
ev = []
for s, e, n in rows:
    if e <= t0 or s >= t1: continue
    ev.append((max(s, t0), 1, fam(n))); ev.append((min(e, t1), -1, fam(n)))
ev.sort(key=lambda x: (x[0], x[1]))
active = collections.Counter()
last = t0
conc = collections.Counter(); solo = collections.Counter(); mixed = collections.Counter()
for t, d, f in ev:
    n = sum(active.values())
    dt = t - last
    if dt > 0:
        conc[min(n, 3)] += dt
        fams = [k for k, v in active.items() if v > 0]
        if len(fams) == 1: solo[fams[0]] += dt
        elif len(fams) > 1: mixed['+'.join(sorted(fams))] += dt
    active[f] += d
    last = t
span = (t1 - t0) / 1e6
print(f'step span {span:.2f} ms;  idle {conc[0] / 1e6:.2f}  one kernel {conc[1] / 1e6:.2f}  two {conc[2] / 1e6:.2f}  three+ {conc[3] / 1e6:.2f} ms')
print('a single FAMILY running (ms): ' + ', '.join(f'{k} {v / 1e6:.2f}' for k, v in solo.most_common()))
print('families overlapping (ms): ' + ', '.join(f'{k} {v / 1e6:.2f}' for k, v in mixed.most_common(6)))
