"""Cold-cache GEMM timing of one shape under different epilogues (what the training step really runs)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops, _lib
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
M, H = 15424, 768
def run(name, m, n, k, tb, variants):
    per = (m * k + k * n + 3 * m * n) * 2
    nset = max(2, int(700e6 // per) + 1)
    sets = []
    for i in range(nset):
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        b = torch.randn((n, k) if tb else (k, n), device=dev).to(torch.bfloat16) * 0.05
        c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
        c2 = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
        x = torch.randn(m, n, device=dev).to(torch.bfloat16)
        sets.append((a, b, c, c2, x))
    bias = torch.randn(n, device=dev).to(torch.bfloat16)
    tab = torch.rand(241, 32, device=dev)
    for vname, kw, bn in [(v, k_, b_) for v, k_ in variants for b_ in (0, 192, 256)]:
        _lib.load().mr_set_option(b'gemm_tile_n', bn)
        vname = f'{vname} bn={bn}'
        def call(a, b, c, c2, x):
            k2 = {}
            for key, val in kw.items():
                k2[key] = {'bias': bias, 'c2': c2, 'aux': x, 'residual': x, 'rot_tab': tab}.get(key, val) if val is True else val
            ops.gemm(a, b, c, transB=bool(tb), **k2, ws=WS)
        for s_ in sets:
            call(*s_)
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for s_ in sets:
                call(*s_)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / nset)
        print(f'{name:10s} {vname:22s} {best:7.1f} us  {2.0 * m * n * k / best / 1e6:7.1f} TF/s', flush=True)
run('fwd mlp1', M, 4 * H, H, 0, [('plain', {}), ('bias', {'bias': True}), ('bias+gelu', {'bias': True, 'act': 1}), ('bias+gelu+c2', {'bias': True, 'act': 1, 'c2': True})])
run('dgrad mlp2', M, 4 * H, H, 1, [('plain', {}), ('aux', {'aux': True})])
run('fwd qkv', M, 3 * H, H, 0, [('plain', {}), ('bias', {'bias': True}), ('bias+rot', {'bias': True, 'rot_tab': True, 'rot_cols': 2 * H})])
run('fwd proj', M, H, H, 0, [('plain', {}), ('residual', {'residual': True})])
run('fwd mlp2', M, H, 4 * H, 0, [('plain', {}), ('residual', {'residual': True})])
run('dgrad mlp1', M, H, 4 * H, 1, [('plain', {})])
