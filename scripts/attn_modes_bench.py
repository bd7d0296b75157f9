"""Masked attention (joint tower shape) under the tile classification of csrc/attention.hip (option attn_tile_modes 1 / 0) for three code patterns:
the bench batch's runs (~18 % PAD in one gap / tail), every position valid (every tile FAST), half of every sequence PAD (many SKIP tiles)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
nseq, S, nh = 24, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 12
H = nh * 64
g = torch.Generator().manual_seed(0)
qkv = torch.randn(nseq * S, 3 * H, generator=g).to(BF16).to(dev)
dout_all = torch.randn(nseq * S, H, generator=g).to(BF16).to(dev)
out = torch.zeros(nseq * S, H, dtype=BF16, device=dev)
lse = torch.zeros(nseq, nh, S, device=dev)
delta = torch.zeros(nseq, nh, S, device=dev)
dqkv = torch.zeros_like(qkv)
pats = {}
c = torch.zeros(nseq, S, dtype=torch.int32)
for q in range(nseq):
    a, n = [(108, 52), (89, 71), (527, 113), (485, 155), (58, 102), (58, 102)][q % 6]
    c[q, a:a + n] = -1
pats['bench runs'] = c
pats['all valid'] = torch.zeros(nseq, S, dtype=torch.int32)
c = torch.zeros(nseq, S, dtype=torch.int32); c[:, 320:] = -1
pats['half PAD'] = c
c = torch.zeros(nseq, S, dtype=torch.int32); c[:, 320:] = 1
pats['two sources'] = c
for name, c in pats.items():
    code = c.reshape(-1).to(dev)
    dout = dout_all * (code >= 0).reshape(-1, 1).to(BF16)        # as in a training step: the PAD rows' upstream gradient is zero
    ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    res = {}
    for modes in (1, 0):
        ops.set_option('attn_tile_modes', modes)
        for kind, fn in (('fwd', lambda: ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)),
                         ('bwd', lambda: ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, None, nseq, S, nh))):
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                fn(); torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=s):
                    for _ in range(20): fn()
            gr.replay(); torch.cuda.synchronize()
            tot = 0.0
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            res[(modes, kind)] = tot / 5 / 20 * 1e3
    ops.set_option('attn_tile_modes', 1)
    print(f'{name:12s} nh {nh}: fwd {res[(1, "fwd")]:6.1f} us (general path {res[(0, "fwd")]:6.1f}) | bwd {res[(1, "bwd")]:6.1f} us (general path {res[(0, "bwd")]:6.1f})', flush=True)
