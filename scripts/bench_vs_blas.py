"""Reference point, not a product path: torch.matmul (hipBLASLt / rocBLAS) on the step's plain GEMM shapes beside mr_gemm, same
rotating cold-cache sets as scripts/bench_gemm_epi.py."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
shapes = [(15424, 3072, 768, 0, 0), (15424, 768, 3072, 0, 0), (15424, 2304, 768, 0, 0), (15424, 768, 768, 0, 0), (15424, 768, 3072, 0, 1),
          (3072, 768, 15424, 1, 0), (15424, 4096, 1024, 0, 0), (15424, 1024, 4096, 0, 0), (4096, 1024, 15424, 1, 0), (8192, 8192, 8192, 0, 0)]
for m, n, k, ta, tb in shapes:
    per = (m * k + k * n + m * n) * 2
    nset = max(2, int(500e6 // per) + 1)
    sets = []
    for i in range(nset):
        a = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
        b = torch.randn((n, k) if tb else (k, n), device=dev).to(torch.bfloat16) * 0.05
        c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
        sets.append((a, b, c))
    def mine(a, b, c): ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), ws=WS)
    def blas(a, b, c): torch.matmul(a.t() if ta else a, b.t() if tb else b, out=c)
    res = {}
    for name, fn in (('mr_gemm', mine), ('torch.matmul', blas)):
        for s_ in sets: fn(*s_)
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for s_ in sets: fn(*s_)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / nset)
        res[name] = best
    print(f'{m}x{n}x{k} ta={ta} tb={tb}: ' + '  '.join(f'{k_} {v:7.1f} us {2.0 * m * n * k / v / 1e6:7.1f} TF/s' for k_, v in res.items()), flush=True)
