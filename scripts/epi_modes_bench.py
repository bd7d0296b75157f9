"""One GEMM shape under each epilogue the step uses on it, cold rotating operand sets, with the kernel each launch is routed to:
    python scripts/epi_modes_bench.py M,N,K [M,N,K ...]        (NT operands: A [M, K], B [N, K])
plain | bias | bias + GELU + gelu' copy | residual | x aux | x aux + column sums.  (Round 6: why the VCR ViT's fc2 dgrad -- 2308 x 4096 x 1024, aux +
column sums -- takes twice its fc1 + GELU.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops, _lib
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
WS = torch.zeros(8 * 1024 * 1024, device=dev)
ops.set_option('gemm_trace', 1)
for a in [a for a in sys.argv[1:] if '=' in a]:          # NAME=VALUE: a library option for the whole run (e.g. gemm5=3: the 128 x 128-tile kernel for every NT problem)
    ops.set_option(a.split('=')[0], int(a.split('=')[1]))
sys.argv = [a for a in sys.argv if '=' not in a]
lib = _lib.load()
for spec in sys.argv[1:] or ['2308,4096,1024', '15424,3072,768', '5952,3072,768', '9216,4096,1024']:
    M, N, K = [int(v) for v in spec.split(',')]
    per = (M * K + N * K + 3 * M * N) * 2
    nset = max(3, int(600e6 // per) + 1)
    g = torch.Generator().manual_seed(0)
    sets = [dict(a=torch.randn(M, K, generator=g).to(BF16).to(dev), b=(torch.randn(N, K, generator=g) * 0.05).to(BF16).to(dev),
                 c=torch.zeros(M, N, dtype=BF16, device=dev), c2=torch.zeros(M, N, dtype=BF16, device=dev), x=torch.randn(M, N, generator=g).to(BF16).to(dev))
            for _ in range(nset)]
    bias = torch.randn(N, generator=g).to(BF16).to(dev)
    cs = torch.zeros(lib.mr_gemm_colsum_rows(M), N, device=dev)
    modes = [('plain', lambda s: {}), ('bias', lambda s: dict(bias=bias)), ('bias+gelu+c2', lambda s: dict(bias=bias, act=ops.ACT_GELU, c2=s['c2'])),
             ('residual', lambda s: dict(residual=s['x'])), ('aux', lambda s: dict(aux=s['x'])), ('aux+colsum', lambda s: dict(aux=s['x'], colsum=cs))]
    for name, kw in modes:
        def run():
            for s in sets:
                ops.gemm(s['a'], s['b'], s['c'], transB=True, ws=WS, **kw(s))
        run()
        route = lib.mr_last_gemm_kernel().decode()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / nset)
        print(f'{spec:18s} {name:14s} {route:34s} {best:7.1f} us {2.0 * M * N * K / best / 1e6:6.0f} TF/s', flush=True)
