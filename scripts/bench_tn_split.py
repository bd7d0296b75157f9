"""Grouped weight-gradient launch (two base layers, 216 tiles) with and without the split tail (option "gemm_tn_split"), cold operands, us per launch."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops, _lib
dev = torch.device('cuda:0')
lib = _lib.load()
H = 768
ws = torch.zeros(32 << 20, device=dev)
for Mtok in (15424, 15360, 5952):
    shapes = [(4 * H, H), (H, 4 * H), (H, H), (H, 3 * H)] * 2
    sets = []
    for s_ in range(4):
        xs = [torch.randn(Mtok, m, device=dev).to(torch.bfloat16) for m, n in shapes]
        dys = [(torch.randn(Mtok, n, device=dev) * 0.05).to(torch.bfloat16) for m, n in shapes]
        outs = [torch.zeros(m, n, dtype=torch.bfloat16, device=dev) for m, n in shapes]
        sets.append([ops.gemm_args(x, dy, o, transA=True, ws=ws) for x, dy, o in zip(xs, dys, outs)] + [xs, dys, outs])
    fl = sum(2.0 * Mtok * m * n for m, n in shapes)
    for rep in range(2):
        for split in (0, 1):
            lib.mr_set_option(b'gemm_tn_split', split)
            for s_ in sets:
                ops.gemm_grouped(s_[:8])
            best = 1e9
            for r in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for s_ in sets:
                    ops.gemm_grouped(s_[:8])
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3 / len(sets))
            print(f'M={Mtok} split={split}: {best:7.1f} us  {fl / best / 1e6:7.1f} TF/s', flush=True)
lib.mr_set_option(b'gemm_tn_split', 1)
