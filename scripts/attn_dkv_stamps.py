"""Phase cycles of the dK / dV kernel of the two-pass attention backward (joint tower shape) from the -DMR_ATTN_STAMPS diagnostic build.
Wave 0 of the first 512 workgroups.  Per query tile: 0-1 requests of the tile two ahead | 1-2 compute (both halves) | 2-3 counted wait for the next tile | 3-4 barrier.
Round-5 history of one workgroup (cycles): 62 602 as found (9 000 prologue, 4 360 per tile of which ~1 000 the staging wave parked on its scalar loads at the top
of the tile) -> 59 642 (three-tile ring requested two ahead, asm row reads, raw barrier, every scalar of the sequence and every tile class in LDS from the prologue:
10 900 prologue, 3 780 per tile)."""
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
lib.mr_diag_attn_stamps(buf)           # (the dK / dV kernel runs last: its stamps are what is left in the buffer)
st = torch.tensor(list(buf), dtype=torch.int64).view(512, 16, 8)
nt = (S + 63) // 64
wg, post = st[:, 15, :2], st[:, 13, :2]
tl = st[:, :nt]
names = ['requests (two ahead)', 'compute', 'wait next tile', 'barrier']
for k, nm in enumerate(names):
    d = (tl[:, :, k + 1] - tl[:, :, k]).float()
    print(f'   {nm:20s} mean {d.mean():7.0f}  per tile: ' + ' '.join(f'{d[:, j].mean():6.0f}' for j in range(nt)))
print(f'   prologue {(tl[:, 0, 0] - wg[:, 0]).float().mean():.0f} | tiles {(tl[:, nt - 1, 4] - tl[:, 0, 0]).float().mean():.0f} | dK / dV stores {(post[:, 1] - post[:, 0]).float().mean():.0f} | '
      f'column sums -> end {(wg[:, 1] - post[:, 1]).float().mean():.0f} | whole {(wg[:, 1] - wg[:, 0]).float().mean():.0f} cycles')
