"""Fused bf16 Adam kernel alone: ms and TB/s for one gradient bucket's worth of parameters (20 B / parameter).  MR_LIB=<variant .so> to A/B."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
for n in (35 * 2048 * 512, 71 * 2048 * 512):
    master = torch.randn(n, device=dev); work = torch.zeros(n, device=dev, dtype=torch.bfloat16)
    grad = (torch.randn(n, device=dev) * 1e-3).to(torch.bfloat16); mu = torch.zeros_like(work); nu = torch.zeros_like(work)
    flags = torch.ones(n // 2048, dtype=torch.uint8, device=dev)
    f = lambda: ops.adam_bf16_update(master, work, grad, mu, nu, flags, 0.9, 0.98, 1e-6, 0.1, 1.0, -1e-4)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{n / 1e6:.1f} M parameters: {ms * 1e3:.1f} us  {20.0 * n / ms / 1e9:.2f} TB/s')
