"""One GEMM shape, a few launches (for rocprofv3 --pmc runs)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
WS = torch.zeros(32 * 1024 * 1024, device=dev)
m, n, k, ta, tb = [int(x) for x in sys.argv[1:6]]
a = torch.randn((k, m) if ta else (m, k), device=dev).to(torch.bfloat16)
b = torch.randn((n, k) if tb else (k, n), device=dev).to(torch.bfloat16)
c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    ops.gemm(a, b, c, transA=bool(ta), transB=bool(tb), ws=WS)
torch.cuda.synchronize()
