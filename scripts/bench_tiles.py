"""Time model GEMM shapes under each forced tile width: python scripts/bench_tiles.py  (M,N,K,ta,tb specs optional)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops, _lib
dev = torch.device('cuda:0')
WS = torch.zeros(64 * 1024 * 1024, device=dev)
specs = sys.argv[1:] or ['15424,768,768,0,0', '15424,768,768,0,1', '15424,768,3072,0,0', '15424,768,3072,0,1', '15424,768,2304,0,1',
                         '15424,3072,768,0,0', '15424,3072,768,0,1', '15424,2304,768,0,0', '5952,768,3072,0,1', '5952,768,768,0,0',
                         '5952,3072,768,0,0', '15424,1024,1024,0,0', '15424,1024,4096,0,1']
flush = torch.zeros(128 * 1024 * 1024, device=dev)      # 512 MB: evicts L2 / MALL between timed launches
for spec in specs:
    m, n, k, ta, tb = [int(x) for x in spec.split(',')]
    a = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
    b = torch.randn((n, k) if tb else (k, n), device=dev).to(torch.bfloat16)
    c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    res = torch.randn(m, n, device=dev).to(torch.bfloat16)
    line = f'{spec:24s}'
    for bn in (0, 96, 128, 192, 256):
        _lib.load().mr_set_option(b'gemm_tile_n', bn)
        for with_res in (False, True):
            ts = []
            for rep in range(6):
                flush.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), residual=res if with_res else None, ws=WS)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            t = sorted(ts)[1]
            line += f' | bn{bn:3d}{"+r" if with_res else "  "} {t:6.1f}us {2.0 * m * n * k / t / 1e6:6.0f}TF'
    print(line, flush=True)
_lib.load().mr_set_option(b'gemm_tile_n', 0)
