"""Time arbitrary GEMM shapes: args are M,N,K,ta,tb tuples separated by spaces, e.g. 3072,2304,15424,1,0"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
for spec in sys.argv[1:]:
    m, n, k, ta, tb = [int(x) for x in spec.split(',')]
    a = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
    b = torch.randn((n, k) if tb else (k, n), device=dev).to(torch.bfloat16)
    c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), ws=WS)
    best = 0
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), ws=WS)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * m * n * k * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    print(f'{spec:28s} {best:8.1f} TF/s   {2.0*m*n*k/best/1e6:8.1f} us')
