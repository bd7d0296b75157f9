"""Short-sequence packing: P consecutive sequences of the audio (S = 31) / span (S = 16) towers handed to the attention kernels as ONE sequence of
P * S positions with a per-position code (the packed sequence's index inside the pack; PAD positions keep -1), i.e. the masked kernels' block-diagonal
case -- same memory, same results, 1 / P of the workgroups.  Forward and the two-pass backward, 20 launches inside a replayed hipGraph each; the
outputs of every packing are compared with the unpacked call's."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
for name, nseq, S, nh, masked in (('audio', 192, 31, 12, False), ('span', 192, 16, 12, True), ('audio large', 192, 31, 16, False)):
    H = nh * 64
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(nseq * S, 3 * H, generator=g)).to(BF16).to(dev)
    dout = (torch.randn(nseq * S, H, generator=g)).to(BF16).to(dev)
    base_code = torch.zeros(nseq, S, dtype=torch.int32)
    if masked:
        ln = torch.randint(3, S + 1, (nseq,), generator=g)
        base_code[torch.arange(S)[None] >= ln[:, None]] = -1
    tab = (torch.rand(S, 32, generator=g) * 2 - 1).to(dev)
    ref = None
    for P in (1, 2, 4, 8):
        if nseq % P or (P > 1 and P * S > 128):
            continue
        c = base_code.clone()
        if P > 1:
            idx = (torch.arange(nseq) % P)[:, None].expand(nseq, S).to(torch.int32)
            c = torch.where(c < 0, c, idx)
        code = c.reshape(-1).to(dev) if (masked or P > 1) else None
        ns, Sp = nseq // P, S * P
        out = torch.zeros(nseq * S, H, dtype=BF16, device=dev)
        lse = torch.zeros(nseq * nh * S, device=dev)
        delta = torch.zeros(nseq * nh * S, device=dev)
        dqkv = torch.zeros_like(qkv)
        ws = torch.zeros((nseq * S // 16 + 64 + nseq) * 3 * H, device=dev)
        bg = torch.zeros(3 * H, dtype=BF16, device=dev)
        fwd = lambda: ops.attention_fwd(qkv, code, out, lse, ns, Sp, nh)
        bwd = lambda: ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, tab, ns, Sp, nh, colsum_ws=ws, bias_grad=bg, jobs=[])
        fwd(); bwd(); torch.cuda.synchronize()
        cur = (out.clone(), dqkv.clone())
        if ref is None:
            ref = cur
            same = 'reference'
        else:
            same = f'out equal: {torch.equal(cur[0], ref[0])}, dqkv equal: {torch.equal(cur[1], ref[1])}, max |d dqkv| {float((cur[1].float() - ref[1].float()).abs().max()):.3e}'
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
        print(f'{name:12s} S {S:3d} x pack {P}: fwd {us["fwd"]:6.1f} us, bwd {us["bwd"]:6.1f} us | {same}', flush=True)
