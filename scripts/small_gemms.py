"""The step's small GEMMs (pooling / cls projections, embeddings) timed one by one in a replayed graph of 50 launches each: these sit on the
critical stream between the big launches, so their latency is what they cost."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
shapes = [(768, 768, 1152, 1, 0, 0), (1152, 768, 768, 0, 1, 0), (192, 768, 768, 0, 1, 1), (64, 64, 768, 0, 1, 0), (192, 192, 768, 0, 1, 0), (192, 768, 384, 0, 0, 0),
          (768, 768, 192, 1, 0, 0), (128, 768, 64, 1, 0, 0), (64, 768, 128, 0, 0, 0), (3840, 768, 768, 0, 1, 0), (768, 768, 3840, 1, 0, 0),
          (768, 768, 15424, 1, 0, 0), (768, 768, 5952, 1, 0, 0)]
for M, N, K, ta, tb, bias in shapes:
    a = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
    b = (torch.randn((N, K) if tb else (K, N), device=dev) * 0.05).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
    bv = torch.randn(N, device=dev).to(torch.bfloat16) if bias else None
    fn = lambda: ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), bias=bv, ws=WS)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(50): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 200 * 1e3
    print(f'{M:6d} x {N:5d} x {K:6d} ta={ta} tb={tb} bias={bias}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s', flush=True)
