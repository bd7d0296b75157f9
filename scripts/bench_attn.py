"""Attention kernels at the step's shapes (random data): time + optional single launch for PMC."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
shapes = [('joint', 24, 640, 12, True), ('vit', 64, 241, 12, False), ('audio', 192, 31, 12, False), ('span', 192, 16, 12, True)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if s[0] in sys.argv[1:]]
for name, nseq, S, nh, masked in shapes:
    H = nh * 64
    qkv = torch.randn(nseq * S, 3 * H, device=dev).to(torch.bfloat16)
    code = None
    if masked:
        c = torch.randint(0, 2, (nseq, S), device=dev)
        c[torch.rand(nseq, S, device=dev) < 0.1] = -1
        c[:, 0] = 0
        code = c.to(torch.int32).reshape(-1)
    out = torch.zeros(nseq * S, H, dtype=torch.bfloat16, device=dev)
    dout = torch.randn(nseq * S, H, device=dev).to(torch.bfloat16)
    lse = torch.zeros(nseq, nh, S, device=dev)
    delta = torch.zeros(nseq, nh, S, device=dev)
    dqkv = torch.zeros_like(qkv)
    fl = 4.0 * nseq * nh * S * S * 64
    for fn, mult, label in ((lambda: ops.attention_fwd(qkv, code, out, lse, nseq, S, nh), 1.0, 'fwd'),
                            (lambda: ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, None, nseq, S, nh), 2.5, 'bwd')):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n_it = int(os.environ.get('ATTN_ITERS', '10'))
        for _ in range(n_it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n_it
        print(f'{name:6s} {label} {us:8.1f} us  {fl * mult / us / 1e6:7.1f} TF/s (algorithmic)')
