"""LayerNorm forward / backward at the step's shapes: us per launch and effective HBM rate."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
for rows, H in [(15424, 768), (15360, 768), (5952, 768), (15424, 1024)]:
    nset = 8
    xs = [torch.randn(rows, H, device=dev).to(torch.bfloat16) for _ in range(nset)]
    ys = [torch.zeros(rows, H, device=dev, dtype=torch.bfloat16) for _ in range(nset)]
    g = torch.randn(H, device=dev).to(torch.bfloat16); b = torch.randn(H, device=dev).to(torch.bfloat16)
    mean = torch.zeros(rows, device=dev); rstd = torch.zeros(rows, device=dev)
    for i in range(nset): ops.layernorm_fwd(xs[i], g, b, ys[i], mean, rstd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(10):
        for i in range(nset): ops.layernorm_fwd(xs[i], g, b, ys[i], mean, rstd)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (10 * nset)
    print(f'ln_fwd {rows}x{H}: {us:6.1f} us  {rows * H * 4 / us / 1e6:6.2f} TB/s')
