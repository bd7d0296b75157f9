"""fp32 GEMM (mr_f32_gemm) at the zero-shot forward's shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
for (m, n, k) in [(4616, 768, 768), (4616, 2304, 768), (4616, 3072, 768), (4616, 768, 3072), (1312, 2304, 768), (1312, 768, 3072), (744, 768, 768), (4616, 1024, 4096)]:
    a = torch.randn(m, k, device=dev); b = torch.randn(k, n, device=dev); c = torch.zeros(m, n, device=dev)
    for _ in range(3):
        ops.gemm(a, b, c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm(a, b, c)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print(f'{m}x{n}x{k}: {us:7.1f} us  {2.0 * m * n * k / us / 1e6:6.1f} TF/s')
