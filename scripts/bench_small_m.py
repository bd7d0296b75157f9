"""Tile-width sweep for the short towers' GEMM shapes (audio M = 5952, span M = 3072): us per launch, rotating operand sets."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops, _lib
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
H = int(os.environ.get('H', '768'))
for M in (5952, 3072):
    for n, k, tb, kw in [(H, H, 0, {'residual': True}), (3 * H, H, 0, {'bias': True, 'rot_tab': True, 'rot_cols': 2 * H}), (4 * H, H, 0, {'bias': True, 'act': 1, 'c2': True}),
                         (H, 4 * H, 0, {'residual': True}), (H, H, 1, {}), (H, 3 * H, 1, {}), (4 * H, H, 1, {'aux': True}), (H, 4 * H, 1, {})]:
        nset = 6
        sets = []
        for i in range(nset):
            a = torch.randn(M, k, device=dev).to(torch.bfloat16)
            b = torch.randn((n, k) if tb else (k, n), device=dev).to(torch.bfloat16) * 0.05
            c = torch.zeros(M, n, device=dev, dtype=torch.bfloat16); c2 = torch.zeros_like(c); x = torch.randn(M, n, device=dev).to(torch.bfloat16)
            sets.append((a, b, c, c2, x))
        bias = torch.randn(n, device=dev).to(torch.bfloat16); tab = torch.rand(31, 32, device=dev)
        line = f'M={M} N={n} K={k} tb={tb} {"+".join(kw) or "plain":18s}'
        for bn in (0, 96, 128, 192, 256):
            _lib.load().mr_set_option(b'gemm_tile_n', bn)
            def call(a, b, c, c2, x):
                k2 = {key: ({'bias': bias, 'c2': c2, 'aux': x, 'residual': x, 'rot_tab': tab}.get(key, val) if val is True else val) for key, val in kw.items()}
                ops.gemm(a, b, c, transB=bool(tb), **k2, ws=WS)
            for s_ in sets: call(*s_)
            best = 1e9
            for rnd in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for s_ in sets: call(*s_)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / nset)
            line += f'  bn{bn}:{best:6.1f}'
        _lib.load().mr_set_option(b'gemm_tile_n', 0)
        print(line, flush=True)
