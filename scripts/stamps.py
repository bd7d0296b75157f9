import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
ws = torch.zeros(32 * 1024 * 1024, device=dev)
WS = ws
m, n, k = [int(x) for x in sys.argv[1:4]]
a = torch.randn(m, k, device=dev).to(torch.bfloat16); b = torch.randn(k, n, device=dev).to(torch.bfloat16)
c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
bias = torch.randn(n, device=dev).to(torch.bfloat16)
for _ in range(3):
    ops.gemm(a, b, c, bias=bias, ws=WS)
torch.cuda.synchronize(); ws.zero_(); torch.cuda.synchronize()
ops.gemm(a, b, c, bias=bias, ws=WS)
torch.cuda.synchronize()
st = ws.view(torch.int64)[:256 * 64 * 4].view(256, 64, 4).cpu()
items = (st[:, :, 0] != 0).sum(1)
print('items per block: min %d max %d' % (items.min(), items.max()))
tot = []
for nm, x, y in (('k-loop (all but last)', 0, 1), ('last k-step', 1, 2), ('epilogue', 2, 3)):
    d = (st[:, :, y] - st[:, :, x]).float()[st[:, :, 0] != 0]
    print(f'{nm:24s} mean {d.mean():9.0f} cycles  ({d.mean() / 100:.1f} x 10ns @100MHz realtime?)  min {d.min():.0f} max {d.max():.0f}')
gap = (st[:, 1:, 0] - st[:, :-1, 3]).float()[(st[:, 1:, 0] != 0)]
print(f'{"item switch gap":24s} mean {gap.mean():9.0f}  min {gap.min():.0f} max {gap.max():.0f}')
whole = (st[:, :, 3] - st[:, :, 0]).float()[st[:, :, 0] != 0]
print('per item total', whole.mean().item(), ' block span', ((st[:, :, 3].max(1).values - st[:, 0, 0]).float().mean().item()))
