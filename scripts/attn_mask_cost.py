"""What the mask costs the joint tower's attention kernels: S = 640, 24 sequences, 12 heads -- unmasked (code = None), masked with one code for every
position (every pair allowed: the common case of the joint sequences), masked with the benchmark's pad pattern; forward and the two-pass backward, 20 launches
inside a replayed hipGraph each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
nseq, S, nh = 24, 640, 12
H = nh * 64
g = torch.Generator().manual_seed(0)
qkv = torch.randn(nseq * S, 3 * H, generator=g).to(BF16).to(dev)
dout = torch.randn(nseq * S, H, generator=g).to(BF16).to(dev)
out = torch.zeros(nseq * S, H, dtype=BF16, device=dev)
lse, delta = torch.zeros(nseq * nh * S, device=dev), torch.zeros(nseq * nh * S, device=dev)
dqkv = torch.zeros_like(qkv)
ws = torch.zeros((nseq * S // 16 + 64 + nseq) * 3 * H, device=dev)
bg = torch.zeros(3 * H, dtype=BF16, device=dev)
tab = (torch.rand(nseq * S, 32, generator=g) * 2 - 1).to(dev)
c_all = torch.zeros(nseq * S, dtype=torch.int32, device=dev)
c_pad = torch.zeros(nseq, S, dtype=torch.int32)
c_pad[:, 560:] = -1
c_pad = c_pad.reshape(-1).to(dev)
for name, code in (('unmasked', None), ('masked, all allowed', c_all), ('masked, 80 trailing PAD', c_pad)):
    fwd = lambda: ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    bwd = lambda: ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, tab, nseq, S, nh, colsum_ws=ws, bias_grad=bg, jobs=[])
    fwd(); bwd(); torch.cuda.synchronize()
    us = {}
    for nm, fn in (('fwd', fwd), ('bwd', bwd)):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(20): fn()
        gr.replay(); torch.cuda.synchronize()
        tot = 0.0
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            if rep: tot += e0.elapsed_time(e1)
        us[nm] = tot / 5 / 20 * 1e3
    print(f'{name:26s}: fwd {us["fwd"]:6.1f} us, bwd {us["bwd"]:6.1f} us', flush=True)
