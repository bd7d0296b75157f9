"""Tile-walk A/B of the K = 768 wide-N forward GEMMs (round-5 review item 2): XCD partition px x (8 / px) of the tile grid (option gemm_xpx) and tile
columns per panel of a cell's walk (option gemm_xpanel; a panel is finished for every row tile of the cell before the next one starts: the
N-panel-stationary walk is a small panel on a partition that splits M).  One variant per process:
    python scripts/xcd_walk_ab.py XPX XPANEL [reps]        prints us per launch of fc1 + GELU, QKV + rotary, fc2-dgrad x gelu' on rotating (cold) operand sets
scripts/xcd_walk_ab.sh runs every variant plainly (time) and under rocprofv3 --pmc FETCH_SIZE (bytes out of L2 per launch)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
xpx, xpanel = int(sys.argv[1]), int(sys.argv[2])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
if len(sys.argv) > 4:          # force the ping-pong kernel's tile width (192 | 256) for every launch: the 192-wide aux / GELU instances on the 256-wide shapes
    ops.set_option('gemm3', int(sys.argv[4]))
ops.set_option('gemm_xpx', xpx)
ops.set_option('gemm_xpanel', xpanel)
ops.set_option('gemm_trace', 1)
M, H = 15424, 768
WS = torch.zeros(8 * 1024 * 1024, device=dev)
g = torch.Generator(device='cpu').manual_seed(0)


def rnd(*s, scale=1.0):
    return (torch.randn(*s, generator=g) * scale).to(BF16).to(dev)


NSET = 5
tab = (torch.rand(241, 32, generator=g) * 2 - 1).to(dev)
cases = []
# fc1 + GELU + gelu' copy: A [M, H] x W^T [4H, H]
cases.append(('fc1 + GELU', [dict(a=rnd(M, H), b=rnd(4 * H, H, scale=0.05), c=torch.zeros(M, 4 * H, dtype=BF16, device=dev), c2=torch.zeros(M, 4 * H, dtype=BF16, device=dev)) for _ in range(NSET)],
              lambda s, bias: ops.gemm(s['a'], s['b'], s['c'], transB=True, bias=bias, act=ops.ACT_GELU, c2=s['c2'], ws=WS), 4 * H))
cases.append(('QKV + rotary', [dict(a=rnd(M, H), b=rnd(3 * H, H, scale=0.05), c=torch.zeros(M, 3 * H, dtype=BF16, device=dev)) for _ in range(NSET)],
              lambda s, bias: ops.gemm(s['a'], s['b'], s['c'], transB=True, bias=bias, rot_tab=tab, rot_cols=2 * H, ws=WS), 3 * H))
cases.append(("fc2 dgrad x gelu'", [dict(a=rnd(M, H), b=rnd(4 * H, H, scale=0.05), c=torch.zeros(M, 4 * H, dtype=BF16, device=dev), x=rnd(M, 4 * H)) for _ in range(NSET)],
              lambda s, bias: ops.gemm(s['a'], s['b'], s['c'], transB=True, aux=s['x'], ws=WS), 4 * H))
for name, sets, fn, N in cases:
    bias = rnd(N)
    for s in sets:
        fn(s, bias)
    route = ops._lib.load().mr_last_gemm_kernel().decode()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for s in sets:
            fn(s, bias)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / NSET)
    print(f'xpx={xpx} xpanel={xpanel} {name:18s} {route:28s} {best:7.1f} us  {2.0 * M * N * H / best / 1e6:6.0f} TF/s', flush=True)
