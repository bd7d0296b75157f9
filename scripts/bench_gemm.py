"""GEMM micro-benchmark on the shapes of the base / large pretraining step (random bf16 data, interleaved rounds)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
M = 15424
H = int(sys.argv[1]) if len(sys.argv) > 1 else 768
shapes = [  # name, M, N, K, ta, tb
    ('fwd qkv', M, 3 * H, H, 0, 0), ('fwd proj', M, H, H, 0, 0), ('fwd mlp1', M, 4 * H, H, 0, 0), ('fwd mlp2', M, H, 4 * H, 0, 0),
    ('dgrad qkv', M, H, 3 * H, 0, 1), ('dgrad mlp1', M, H, 4 * H, 0, 1), ('dgrad mlp2', M, 4 * H, H, 0, 1),
    ('wgrad qkv', H, 3 * H, M, 1, 0), ('wgrad proj', H, H, M, 1, 0), ('wgrad mlp1', H, 4 * H, M, 1, 0), ('wgrad mlp2', 4 * H, H, M, 1, 0),
    ('audio fwd mlp1', 5952, 4 * H, H, 0, 0), ('square 4096', 4096, 4096, 4096, 0, 1), ('square 4096 nn', 4096, 4096, 4096, 0, 0),
]
res = {}
bufs = {}
for name, m, n, k, ta, tb in shapes:
    a = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
    b = torch.randn((n, k) if tb else (k, n), device=dev).to(torch.bfloat16)
    c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    bufs[name] = (a, b, c, ta, tb, 2.0 * m * n * k)
for rnd in range(3):
    for name, (a, b, c, ta, tb, fl) in bufs.items():
        for _ in range(2):
            ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), ws=WS)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), ws=WS)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(name, []).append(fl * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e12)
for name, v in res.items():
    print(f'{name:18s} TF/s: ' + ' '.join(f'{x:7.1f}' for x in v))
