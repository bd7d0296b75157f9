"""Attention backward: the dQ + dK / dV kernel pair (option attn_onepass = 0) against the one-pass kernel (1) on the step's shapes, each timed as
20 launches inside a replayed hipGraph (sustained, alternating), plus the forward for reference."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
shapes = [('base ViT', 64, 241, 12, False), ('large ViT', 64, 241, 16, False), ('base joint', 24, 640, 12, True), ('large joint', 24, 640, 16, True), ('audio', 192, 31, 12, False),
          ('span', 192, 16, 12, True), ('S=200 masked', 64, 200, 12, True), ('S=256', 64, 256, 12, False), ('S=130', 64, 130, 12, False),
          ('resadapt ViT', 32, 577, 16, False), ('resadapt joint', 12, 1312, 16, True), ('VCR joint', 32, 288, 16, True)]
for name, nseq, S, nh, masked in shapes:
    H = nh * 64
    g = torch.Generator().manual_seed(0)
    qkv = (torch.randn(nseq * S, 3 * H, generator=g)).to(BF16).to(dev)
    dout = (torch.randn(nseq * S, H, generator=g)).to(BF16).to(dev)
    code = None
    if masked:
        c = torch.zeros(nseq, S, dtype=torch.int32)
        if S in (640, 1312):      # the run structure of the bench batch's joint sequences (planner.joint_code: ~18 % PAD in one gap or tail per sequence)
            for q in range(nseq):
                a, n = [(108, 52), (89, 71), (527, 113), (485, 155), (58, 102), (58, 102)][q % 6]
                c[q, a * S // 640:(a + n) * S // 640] = -1
        else:
            c[:, S // 6:S // 6 + S // 10] = -1
        code = c.reshape(-1).to(dev)
        dout = dout * (code >= 0).reshape(-1, 1).to(BF16)        # as in a training step: the PAD rows' upstream gradient is zero
    out = torch.zeros(nseq * S, H, dtype=BF16, device=dev)
    lse = torch.zeros(nseq, nh, S, device=dev)
    delta = torch.zeros(nseq, nh, S, device=dev)
    dqkv = torch.zeros_like(qkv)
    rows = nseq * ((S + 15) // 16 + 4)
    ws = torch.zeros(rows * 3 * H, device=dev)
    bg = torch.zeros(3 * H, dtype=BF16, device=dev)
    rot = torch.rand(nseq * S if masked else S, 32, device=dev) * 2 - 1     # the step passes the towers' "rotary" scale tables: per position of the batch (joint) / of a sequence
    ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    res = {}
    graphs = {}
    for mode in ('fwd', 0, 1):
        if mode == 1 and S > 256:
            continue
        if mode != 'fwd':
            ops.set_option('attn_onepass', mode)
        fn = (lambda: ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)) if mode == 'fwd' else \
             (lambda: ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, rot, nseq, S, nh, colsum_ws=ws, bias_grad=bg, jobs=[]))
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            fn(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                for _ in range(20): fn()
        graphs[mode] = gr
    ops.set_option('attn_onepass', -1)
    for gr in graphs.values(): gr.replay()
    torch.cuda.synchronize()
    tot = {m: 0.0 for m in graphs}
    for rep in range(6):
        for m, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            if rep: tot[m] += e0.elapsed_time(e1)
    us = {m: tot[m] / 5 / 20 * 1e3 for m in tot}
    fl = 4.0 * S * S * 64 * nh * nseq
    line = f'{name:14s} nseq {nseq:4d} S {S:4d} nh {nh:2d} {"masked" if masked else "      "} | fwd {us["fwd"]:7.1f} us ({fl / us["fwd"] / 1e6:5.0f} TF/s) | bwd two-pass {us[0]:7.1f} us ({2.5 * fl / us[0] / 1e6:5.0f} TF/s)'
    if 1 in us:
        line += f' | one-pass {us[1]:7.1f} us ({2.5 * fl / us[1] / 1e6:5.0f} TF/s)  x{us[0] / us[1]:.2f}'
    print(line, flush=True)
