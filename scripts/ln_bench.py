"""LayerNorm forward / backward at the step's shapes: option ln_impl = 0 (round 5) against 1 (round 6), each timed as 24 launches inside a
replayed hipGraph over FOUR rotating buffer sets (~380 MB at [15424, 768]: larger than the Infinity Cache, so every launch reads HBM as in the
step), algorithmic bytes / time in TB/s; plus the largest difference between the two kernels' outputs and the deferred reduction's time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
NSET, NL = 4, 24
shapes = [('base ViT', 15424, 768), ('base joint', 15360, 768), ('base audio', 5952, 768), ('base span', 3072, 768), ('large ViT', 15424, 1024),
          ('large resadapt ViT', 18464, 1024), ('VCR joint', 9216, 1024)]
for name, M, H in shapes:
    g = torch.Generator().manual_seed(0)
    sets = []
    for i in range(NSET):
        x = (torch.randn(M, H, generator=g) * 2 + 0.5).to(BF16).to(dev)
        dy = torch.randn(M, H, generator=g).to(BF16).to(dev)
        add = torch.randn(M, H, generator=g).to(BF16).to(dev)
        sets.append(dict(x=x, dy=dy, add=add, y=torch.zeros_like(x), dx=torch.zeros_like(x), mean=torch.zeros(M, device=dev), rstd=torch.zeros(M, device=dev)))
    gamma = (torch.randn(H, generator=g) + 1).to(BF16).to(dev)
    beta = torch.randn(H, generator=g).to(BF16).to(dev)
    dg, db = torch.zeros(H, dtype=BF16, device=dev), torch.zeros(H, dtype=BF16, device=dev)
    ws = [ops.layernorm_bwd_workspace(H, dev) for _ in range(NSET)]
    for s in sets:
        ops.layernorm_fwd(s['x'], gamma, beta, s['y'], s['mean'], s['rstd'])
    torch.cuda.synchronize()

    def fwd(i):
        s = sets[i % NSET]
        ops.layernorm_fwd(s['x'], gamma, beta, s['y'], s['mean'], s['rstd'])

    def bwd_add(i):
        s = sets[i % NSET]
        ops.layernorm_bwd(s['dy'], s['x'], gamma, s['mean'], s['rstd'], s['dx'], dg, db, ws[i % NSET], dx_add=s['add'], jobs=[])

    def bwd(i):
        s = sets[i % NSET]
        ops.layernorm_bwd(s['dy'], s['x'], gamma, s['mean'], s['rstd'], s['dx'], dg, db, ws[i % NSET], jobs=[])

    def red(i):          # a layer group's deferred reduction as the base step issues it: 4 LayerNorm jobs (+ bias jobs left out)
        jobs = []
        for k in range(4):
            s = sets[k % NSET]
            jobs.append(ops._lib.ReduceJob(ws[k % NSET].data_ptr(), ops._lib.load().mr_layernorm_bwd_nparts(M), 2 * H, H, dg.data_ptr(), db.data_ptr()))
        ops.reduce_partials(jobs)

    outs = {}
    graphs = {}
    for impl in (0, 1):
        ops.set_option('ln_impl', impl)
        for s in sets[:1]:
            ops.layernorm_bwd(s['dy'], s['x'], gamma, s['mean'], s['rstd'], s['dx'], dg, db, ws[0], dx_add=s['add'])
            outs[impl] = (s['dx'].float().clone(), dg.float().clone(), db.float().clone())
        for kind, fn in (('fwd', fwd), ('bwd+add', bwd_add), ('bwd', bwd), ('reduce4', red)):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                fn(0); torch.cuda.synchronize()
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr, stream=st):
                    for i in range(NL): fn(i)
            graphs[(impl, kind)] = gr
    ops.set_option('ln_impl', 1)
    for gr in graphs.values(): gr.replay()
    torch.cuda.synchronize()
    tot = {k: 0.0 for k in graphs}
    for rep in range(6):
        for k, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            if rep: tot[k] += e0.elapsed_time(e1)
    us = {k: tot[k] / 5 / NL * 1e3 for k in tot}
    nbytes = {'fwd': 2 * M * H * 2 + 8 * M, 'bwd+add': 4 * M * H * 2 + 8 * M, 'bwd': 3 * M * H * 2 + 8 * M}
    d = [float((a - b).abs().max()) for a, b in zip(outs[0], outs[1])]
    ref = [float(a.abs().max()) for a in outs[0]]
    line = f'{name:20s} [{M:6d},{H:5d}]'
    for kind in ('fwd', 'bwd+add', 'bwd'):
        line += f' | {kind} {us[(0, kind)]:6.1f} -> {us[(1, kind)]:6.1f} us ({nbytes[kind] / us[(0, kind)] / 1e6:4.2f} -> {nbytes[kind] / us[(1, kind)] / 1e6:4.2f} TB/s)'
    line += f' | reduce4 {us[(0, "reduce4")]:5.1f} -> {us[(1, "reduce4")]:5.1f} us | max diff dx {d[0]:.3g}/{ref[0]:.3g} dgamma {d[1]:.3g}/{ref[1]:.3g} dbeta {d[2]:.3g}/{ref[2]:.3g}'
    print(line, flush=True)
