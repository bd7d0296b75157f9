"""GEMM micro-benchmark with CACHE-COLD operands: each shape cycles through enough distinct buffer sets (> 512 MB)
that nothing is served from the 256 MB Infinity Cache, as inside the training step."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
M = 15424
H = int(os.environ.get('H', 768))
shapes = [('fwd qkv', M, 3 * H, H, 0, 0), ('fwd proj', M, H, H, 0, 0), ('fwd mlp1', M, 4 * H, H, 0, 0), ('fwd mlp2', M, H, 4 * H, 0, 0),
          ('dgrad qkv', M, H, 3 * H, 0, 1), ('dgrad mlp1', M, H, 4 * H, 0, 1), ('dgrad mlp2', M, 4 * H, H, 0, 1), ('dgrad proj', M, H, H, 0, 1),
          ('wgrad mlp2', 4 * H, H, M, 1, 0), ('wgrad qkv', H, 3 * H, M, 1, 0)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0] in sys.argv[1:]]
for name, m, n, k, ta, tb in shapes:
    per = (m * k + k * n + m * n) * 2
    nset = max(2, int(600e6 // per) + 1)
    sets = []
    for i in range(nset):
        a = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
        b = torch.randn((n, k) if tb else (k, n), device=dev).to(torch.bfloat16)
        c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
        sets.append((a, b, c))
    for a, b, c in sets:
        ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), ws=WS)
    best = 0
    for rnd in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for a, b, c in sets:
            ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), ws=WS)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * m * n * k * nset / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    print(f'{name:12s} {best:7.1f} TF/s  ({2.0 * m * n * k / best / 1e6:6.1f} us, {nset} buffer sets)', flush=True)
    del sets
