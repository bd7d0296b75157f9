"""Phase cycles of the forward attention kernel from the -DMR_ATTN_STAMPS diagnostic build (scripts/build_diag.sh attnstamps attention
-DMR_ATTN_STAMPS; run with MR_LIB=.../libdiag_attnstamps.so).  Phases per key tile, wave 0 of the first 512 workgroups:
0-1 issue of next tile's global loads | 1-2 S^T MFMAs issued | 2-3 softmax (waits for the MFMAs) | 3-4 PV MFMAs issued |
4-5 LDS stores of the next tile (wait for its global loads) | 5-6 barrier."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops, _lib
dev = torch.device('cuda:0')
lib = C.CDLL(_lib.LIB_PATH)
for name, nseq, S, nh, masked in [('joint', 24, 640, 12, True), ('vit', 64, 241, 12, False)]:
    H = nh * 64
    qkv = torch.randn(nseq * S, 3 * H, device=dev).to(torch.bfloat16)
    code = None
    if masked:
        c = torch.randint(0, 2, (nseq, S), device=dev); c[torch.rand(nseq, S, device=dev) < 0.1] = -1; c[:, 0] = 0
        code = c.to(torch.int32).reshape(-1)
    out = torch.zeros(nseq * S, H, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(nseq, nh, S, device=dev)
    for _ in range(3):
        ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.attention_fwd(qkv, code, out, lse, nseq, S, nh); e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (512 * 16 * 8))()
    lib.mr_diag_attn_stamps(buf)
    st = torch.tensor(list(buf), dtype=torch.int64).view(512, 16, 8)
    nt = (S + 63) // 64
    wg = st[:, 15, :2]
    st = st[:, :nt]
    names = ['load issue', 'S^T mfma issue', 'softmax', 'PV mfma issue', 'lds store', 'barrier']
    print(f'{name}: S={S} tiles={nt} kernel {e0.elapsed_time(e1) * 1e3:.1f} us')
    for k, nm in enumerate(names):
        d = (st[:, :, k + 1] - st[:, :, k]).float()
        print(f'   {nm:16s} mean {d.mean():7.0f}  (first tile {d[:, 0].mean():7.0f}, middle {d[:, 1:max(nt - 1, 2)].mean():7.0f}, last {d[:, nt - 1].mean():7.0f})')
    tile = (st[:, :, 6] - st[:, :, 0]).float()
    print(f'   workgroup: begin -> first tile {(st[:, 0, 0] - wg[:, 0]).float().mean():.0f}, last tile -> end {(wg[:, 1] - st[:, nt - 1, 6]).float().mean():.0f}, whole {(wg[:, 1] - wg[:, 0]).float().mean():.0f} cycles')
    print(f'   per tile total   mean {tile.mean():7.0f};  wave span first->last tile {(st[:, nt - 1, 6] - st[:, 0, 0]).float().mean():.0f} cycles')
