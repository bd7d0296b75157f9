import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
M, N, K = 241 * 4, 384, 128
a = torch.randn(M, K, device=dev).to(torch.bfloat16); w = (torch.randn(K, N, device=dev) * 0.1).to(torch.bfloat16)
bias = torch.randn(N, device=dev).to(torch.bfloat16)
tab = torch.rand(241, 32, device=dev) * 2 - 1
out = torch.zeros(M, N, dtype=torch.bfloat16, device=dev)
ops.gemm(a, w, out, bias=bias, rot_tab=tab, rot_cols=256)
ref = a.float() @ w.float() + bias.float()
scale = torch.ones(M, N, device=dev); rows = torch.arange(M, device=dev) % 241
for h in range(4): scale[:, h * 64:h * 64 + 32] = tab[rows]
ref = ref * scale
bad = ~torch.isfinite(out.float()) | ((out.float() - ref).abs() > 0.05 * ref.abs() + 0.05)
print('bad count', int(bad.sum()), 'of', bad.numel())
idx = bad.nonzero()
print('bad rows range', idx[:, 0].min().item(), idx[:, 0].max().item(), 'cols', sorted(set((idx[:, 1] // 16).tolist()))[:30])
print('rows mod 16 hist', torch.bincount(idx[:, 0] % 16, minlength=16).tolist())
print('rows // 64 hist', torch.bincount(idx[:, 0] // 64, minlength=16).tolist())
r, c = idx[0].tolist(); print('first bad', r, c, out[r, c].item(), ref[r, c].item(), 'ratio', (out[r, c].float() / ref[r, c]).item())
