"""Reference point, not a product path: torch.matmul (hipBLASLt / rocBLAS) on the step's plain GEMM shapes beside mr_gemm, rotating
cold-cache operand sets, SUSTAINED and INTERLEAVED (the contenders take turns for ~0.3 s after a warm-up of the same length, so
they share the clock the chip settles at under load: a cold 3-ms burst runs ~15 % faster than the same kernel inside a step).
The product's layout for each problem is listed first: forward and dgrad GEMMs read B as [N, K] (dgrads: the flax kernel as it is;
forward: its transposed working copy), weight gradients read A [K, M] and B [K, N].  torch.matmul gets the same operands in the
layout it is fastest with (both are timed)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
# (M, N, K, kind): 'nt' = A [M,K] . B[N,K]^T (forward / dgrad);  'tn' = A[K,M]^T . B[K,N] (weight gradient)
shapes = [(15424, 3072, 768, 'nt'), (15424, 768, 3072, 'nt'), (15424, 2304, 768, 'nt'), (15424, 768, 768, 'nt'), (15424, 768, 2304, 'nt'),
          (3072, 768, 15424, 'tn'), (15424, 4096, 1024, 'nt'), (15424, 1024, 4096, 'nt'), (4096, 1024, 15424, 'tn'), (8192, 8192, 8192, 'nt')]
for m, n, k, kind in shapes:
    per = (m * k + k * n + m * n) * 2
    nset = max(2, int(500e6 // per) + 1)
    sets = []
    for i in range(nset):
        if kind == 'nt':
            a = torch.randn(m, k, device=dev).to(torch.bfloat16)
            b = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        else:
            a = torch.randn(k, m, device=dev).to(torch.bfloat16)
            b = (torch.randn(k, n, device=dev) * 0.05).to(torch.bfloat16)
        sets.append((a, b, b.t().contiguous() if kind == 'nt' else None, torch.zeros(m, n, device=dev, dtype=torch.bfloat16)))
    if kind == 'nt':
        fns = {'mr_gemm': lambda a, b, bt, c: ops.gemm(a, b, c, transB=True, ws=WS),
               'torch.matmul(a, b.T)': lambda a, b, bt, c: torch.matmul(a, b.t(), out=c),
               'torch.matmul(a, b_kn)': lambda a, b, bt, c: torch.matmul(a, bt, out=c)}
    else:
        fns = {'mr_gemm': lambda a, b, bt, c: ops.gemm(a, b, c, transA=True, ws=WS),
               'torch.matmul(a.T, b)': lambda a, b, bt, c: torch.matmul(a.t(), b, out=c)}
    tot = {k_: [0.0, 0] for k_ in fns}
    for phase in range(2):
        spent = 0.0
        while spent < 300.0:
            evs = {}
            for name, fn in fns.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for s_ in sets: fn(*s_)
                e1.record()
                evs[name] = (e0, e1)
            torch.cuda.synchronize()
            for name, (e0, e1) in evs.items():
                ms = e0.elapsed_time(e1)
                spent += ms
                if phase:
                    tot[name][0] += ms; tot[name][1] += nset
    res = {k_: v[0] * 1e3 / v[1] for k_, v in tot.items()}
    best_blas = min(v for k_, v in res.items() if k_ != 'mr_gemm')
    print(f'{m}x{n}x{k} {kind}: ' + '  '.join(f'{k_} {v:7.1f} us {2.0 * m * n * k / v / 1e6:7.1f} TF/s' for k_, v in res.items())
          + f'   | mr_gemm / best vendor = {res["mr_gemm"] / best_blas:.3f}', flush=True)
