"""Phase cycles of the one-pass attention backward (attn_bwd1_kernel) from the -DMR_ATTN_STAMPS diagnostic build
(bash scripts/build_diag.sh attnstamps attention -fno-slp-vectorize -DMR_ATTN_STAMPS; run with MR_LIB=merlot_reserve_amd/libdiag_attnstamps.so).
Wave 0 of the first 512 workgroups.  Per query tile: 0-1 request of the next tile + per-query scalars | 1-2 dQ of the previous tile (transposed reads,
MFMAs, stores) | 2-3 first 32-query half (S, dP, P, dS, dS^T -> LDS, dV / dK) | 3-4 second half | 4-5 wait for the next tile | 5-6 barrier."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops, _lib
dev = torch.device('cuda:0')
lib = C.CDLL(_lib.LIB_PATH)
for name, nseq, S, nh in [('base ViT', 64, 241, 12), ('large ViT', 64, 241, 16)]:
    H = nh * 64
    qkv = torch.randn(nseq * S, 3 * H, device=dev).to(torch.bfloat16)
    dout = torch.randn(nseq * S, H, device=dev).to(torch.bfloat16)
    out = torch.zeros(nseq * S, H, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(nseq, nh, S, device=dev)
    delta = torch.zeros(nseq, nh, S, device=dev)
    dqkv = torch.zeros_like(qkv)
    rot = torch.rand(S, 32, device=dev) * 2 - 1
    rows = nseq * ((S + 15) // 16 + 4)
    ws = torch.zeros(rows * 3 * H, device=dev)
    bg = torch.zeros(3 * H, dtype=torch.bfloat16, device=dev)
    ops.attention_fwd(qkv, None, out, lse, nseq, S, nh)
    fn = lambda: ops.attention_bwd(qkv, None, out, dout, lse, delta, dqkv, rot, nseq, S, nh, colsum_ws=ws, bias_grad=bg, jobs=[])
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (512 * 16 * 8))()
    lib.mr_diag_attn_stamps(buf)
    st = torch.tensor(list(buf), dtype=torch.int64).view(512, 16, 8)
    nwg = min(512, nseq * nh)
    st = st[:nwg]
    nt = (S + 63) // 64
    wg, post = st[:, 15, :2], st[:, 13, :3]
    tl = st[:, :nt]
    names = ['stage + scalars issue', 'dQ of previous tile', 'half 0', 'half 1', 'wait next tile', 'barrier']
    print(f'{name}: S={S} tiles={nt} kernel {e0.elapsed_time(e1) * 1e3:.1f} us (stamped build)')
    for k, nm in enumerate(names):
        d = (tl[:, :, k + 1] - tl[:, :, k]).float()
        print(f'   {nm:22s} mean {d.mean():7.0f}  per tile: ' + ' '.join(f'{d[:, j].mean():7.0f}' for j in range(nt)))
    print(f'   prologue (begin -> first tile) {(tl[:, 0, 0] - wg[:, 0]).float().mean():.0f} | tiles {(tl[:, nt - 1, 6] - tl[:, 0, 0]).float().mean():.0f} | last dQ {(post[:, 1] - post[:, 0]).float().mean():.0f} | '
          f'dK / dV stores {(post[:, 2] - post[:, 1]).float().mean():.0f} | column sums -> end {(wg[:, 1] - post[:, 2]).float().mean():.0f} | whole {(wg[:, 1] - wg[:, 0]).float().mean():.0f} cycles')
