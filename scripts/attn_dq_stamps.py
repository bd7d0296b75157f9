"""Phase cycles of the dQ kernel of the two-pass attention backward (joint tower shape) from the diagnostic build in which ONLY that kernel stamps
(bash scripts/build_diag.sh attnstamps2 attention -fno-slp-vectorize -DMR_ATTN_STAMPS -DMR_ATTN_STAMP_K=2; MR_LIB=merlot_reserve_amd/libdiag_attnstamps2.so).
Per key tile, wave 0 of the first 512 workgroups: 0-1 requests of the next tile + key codes (+ the vote on them) | 1-2 scores, P, dS | 2-3 dQ (transposed reads + MFMAs) |
3-4 store codes, wait for the next tile | 4-5 barrier."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops, _lib
dev = torch.device('cuda:0')
lib = C.CDLL(_lib.LIB_PATH)
nseq, S, nh = 24, 640, 12
H = nh * 64
g = torch.Generator().manual_seed(0)
qkv = torch.randn(nseq * S, 3 * H, generator=g).to(torch.bfloat16).to(dev)
c = torch.zeros(nseq, S, dtype=torch.int32)
for q in range(nseq):
    a, n = [(108, 52), (89, 71), (527, 113), (485, 155), (58, 102), (58, 102)][q % 6]
    c[q, a:a + n] = -1
code = c.reshape(-1).to(dev)
dout = torch.randn(nseq * S, H, generator=g).to(torch.bfloat16).to(dev) * (code >= 0).reshape(-1, 1).to(torch.bfloat16)
out = torch.zeros(nseq * S, H, dtype=torch.bfloat16, device=dev)
lse = torch.zeros(nseq, nh, S, device=dev)
delta = torch.zeros(nseq, nh, S, device=dev)
dqkv = torch.zeros_like(qkv)
rot = torch.rand(nseq * S, 32, device=dev) * 2 - 1
rows = nseq * ((S + 15) // 16 + 4)
ws = torch.zeros(rows * 3 * H, device=dev)
bg = torch.zeros(3 * H, dtype=torch.bfloat16, device=dev)
ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
fn = lambda: ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, rot, nseq, S, nh, colsum_ws=ws, bias_grad=bg, jobs=[])
for _ in range(3):
    fn()
torch.cuda.synchronize()
buf = (C.c_ulonglong * (512 * 16 * 8))()
lib.mr_diag_attn_stamps(buf)
st = torch.tensor(list(buf), dtype=torch.int64).view(512, 16, 8)
nt = (S + 63) // 64
wg, post = st[:, 15, :2], st[:, 13, :2]
tl = st[:, :nt]
names = ['requests + codes + vote', 'scores, P, dS', 'dQ', 'store codes + wait', 'barrier']
for k, nm in enumerate(names):
    d = (tl[:, :, k + 1] - tl[:, :, k]).float()
    print(f'   {nm:24s} mean {d.mean():7.0f}  per tile: ' + ' '.join(f'{d[:, j].mean():6.0f}' for j in range(nt)))
print(f'   prologue {(tl[:, 0, 0] - wg[:, 0]).float().mean():.0f} | tiles {(tl[:, nt - 1, 5] - tl[:, 0, 0]).float().mean():.0f} | dQ stores {(post[:, 1] - post[:, 0]).float().mean():.0f} | '
      f'column sums -> end {(wg[:, 1] - post[:, 1]).float().mean():.0f} | whole {(wg[:, 1] - wg[:, 0]).float().mean():.0f} cycles')
