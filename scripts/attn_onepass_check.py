"""One-pass attention backward against the two-pass kernels (same inputs): largest element differences per output third, the share of
elements that differ by more than one bf16 ulp of the larger magnitude, and run-to-run bitwise reproducibility of the one-pass kernel."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0'); BF16 = torch.bfloat16
for nseq, S, nh, masked in [(2, 256, 2, True), (64, 241, 12, False), (8, 200, 3, True), (4, 129, 2, True), (5, 31, 2, False), (16, 256, 12, True)]:
    H = nh * 64
    g = torch.Generator().manual_seed(1)
    qkv = torch.randn(nseq * S, 3 * H, generator=g).to(BF16).to(dev)
    dout = torch.randn(nseq * S, H, generator=g).to(BF16).to(dev)
    code = None
    if masked:
        c = torch.randint(0, 2, (nseq, S), generator=g); pad = torch.rand(nseq, S, generator=g) < 0.2; c[pad] = -1; c[:, 0] = 0
        code = c.to(torch.int32).reshape(-1).to(dev)
    out = torch.zeros(nseq * S, H, dtype=BF16, device=dev); lse = torch.zeros(nseq, nh, S, device=dev); delta = torch.zeros(nseq, nh, S, device=dev)
    ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    res = {}
    for mode in (0, 1, 1):
        ops.set_option('attn_onepass', mode)
        d = torch.full_like(qkv, float('nan'))
        ops.attention_bwd(qkv, code, out, dout, lse, delta, d, None, nseq, S, nh)
        torch.cuda.synchronize()
        res.setdefault(mode, []).append(d.float())
    ops.set_option('attn_onepass', -1)
    two, one, one2 = res[0][0], res[1][0], res[1][1]
    line = f'nseq {nseq:3d} S {S:4d} nh {nh:2d} masked {int(masked)}: rerun bitwise {bool(torch.equal(one, one2))}'
    for name, sl in (('dq', slice(0, H)), ('dk', slice(H, 2 * H)), ('dv', slice(2 * H, 3 * H))):
        a, b = one[:, sl], two[:, sl]
        diff = (a - b).abs()
        ulp = torch.maximum(a.abs(), b.abs()) * 2 ** -7 + 1e-6
        line += f' | {name}: max|d| {diff.max().item():.3e} (scale {b.abs().mean().item():.3e}), >1ulp {(diff > ulp).float().mean().item() * 100:.3f}%, >4ulp {(diff > 4 * ulp).float().mean().item() * 100:.4f}%'
    print(line, flush=True)
