"""Where do an eager step and a hipGraph replay of the SAME step diverge?  (diagnostic)  usage: det_check.py [base|large] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlot_reserve_amd.config import load_config
from merlot_reserve_amd.synthetic import make_batch
from merlot_reserve_amd.trainer import Trainer

name = sys.argv[1] if len(sys.argv) > 1 else 'base'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device('cuda:0')
cfg = load_config(name)
cfg['optimizer'].update(num_warmup_steps=2)
tr = Trainer(cfg, B, dev, seed=0)
p, eng = tr.params, tr.engine
batch = make_batch(cfg, B, seed=11, device=dev)
plan = tr.plan(batch)
tr.train_step(batch, plan=plan)
snap = {k: getattr(p, k).clone() for k in ('master', 'mu', 'nu')}
step0 = tr.state.step

def restore():
    for k, v in snap.items():
        getattr(p, k).copy_(v)
    p.refresh_work()
    tr.state.step = step0

def grab():
    torch.cuda.synchronize()
    out = {'E': eng.E.clone(), 'loss': eng.loss_acc.clone(), 'dE': eng.dE.clone(), 'grad': p.grad.clone(), 'master': p.master.clone()}
    for tname, st in (('tv', eng.tv), ('ta', eng.ta), ('tj', eng.tj), ('ts', eng.ts)):
        out[tname + '.xin'] = st.xin.clone()
        out[tname + '.X0'] = st.X[0].clone()
        out[tname + '.stats'] = st.stats.clone()
        for l in range(min(st.L, 3)):
            out[f'{tname}.L{l}.0ln1'] = st.ln1[l].clone()
            out[f'{tname}.L{l}.1qkv'] = st.qkv[l].clone()
            out[f'{tname}.L{l}.2att'] = st.att[l].clone()
            out[f'{tname}.L{l}.3xmid'] = st.xmid[l].clone()
            out[f'{tname}.L{l}.4ln2'] = st.ln2[l].clone()
            out[f'{tname}.L{l}.5hact'] = st.hact[l].clone()
            out[f'{tname}.L{l}.6hpre'] = st.hpre[l].clone()
            out[f'{tname}.L{l}.7Xnext'] = st.X[l + 1].clone()
        out[tname + '.xf'] = st.xf.clone()
    return out

def diff(a, b, label):
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    print(f'{label}: ' + ('IDENTICAL' if not bad else 'differ: ' + ', '.join(bad)))
    for k in bad[:12]:
        if a[k].dim() == 2:
            ne = (a[k] != b[k])
            rows = ne.any(1).nonzero().reshape(-1)
            cols = ne.any(0).nonzero().reshape(-1)
            print(f'     {k}: {int(ne.sum())} elements, max |d| {float((a[k].float() - b[k].float()).abs().max()):.3e} (max |v| {float(a[k].float().abs().max()):.2e}); '
                  f'rows {int(rows.min())}..{int(rows.max())} ({rows.numel()}), cols {int(cols.min())}..{int(cols.max())} ({cols.numel()})')
    if 'grad' in bad:      # which towers' gradient ranges differ
        for t, (lo, hi) in p.tower_ranges.items():
            d = (a['grad'][lo:hi] != b['grad'][lo:hi]).sum().item()
            if d:
                print(f'   grad[{t}]: {d} of {hi - lo} elements differ')

restore(); tr.train_step(batch, plan=plan); e1 = grab()
restore(); tr.capture(batch)
tr.train_step_graph(batch, plan); g1 = grab()
restore(); tr.train_step_graph(batch, plan); g2 = grab()
diff(g1, g2, 'replay vs replay')
diff(e1, g1, 'eager vs replay')

# which run is right?  host recompute of the first differing LayerNorm output from its grabbed input, with the CURRENT (restored) weights
W = p.w
cands = [('tv.X0', 'tv.xin', 'vision_encoder/transformer/pre_ln')]
for l in range(3):
    cands += [(f'tv.L{l}.0ln1', 'tv.X0' if l == 0 else f'tv.L{l-1}.7Xnext', f'vision_encoder/transformer/layer_{l:02d}/pre_attn_ln'),
              (f'tv.L{l}.4ln2', f'tv.L{l}.3xmid', f'vision_encoder/transformer/layer_{l:02d}/pre_mlp_ln')]
restore()
torch.cuda.synchronize()
for k, kin, wname in cands:
    if torch.equal(g1[k], g2[k]):
        continue
    ne = (g1[k] != g2[k])
    rows = ne.any(1).nonzero().reshape(-1).tolist()
    print('first differing LayerNorm output', k, 'input', kin, 'equal between runs:', bool(torch.equal(g1[kin], g2[kin])), 'rows', rows[:8])
    gam, bet = W[wname + '/scale'].float(), W[wname + '/bias'].float()
    for r in rows[:3]:
        for nm, G in (('run1', g1), ('run2', g2)):
            x = G[kin][r].float()
            mean = x.mean(); var = (x * x).mean() - mean * mean
            ref = ((x - mean) * (torch.rsqrt(var + 1e-5) * gam) + bet).to(torch.bfloat16)
            bad = (ref != G[k][r]).nonzero().reshape(-1)
            print(f'   row {r} {nm}: {bad.numel()} of {ref.numel()} outputs differ from the host recompute with the restored weights; cols {bad[:6].tolist()}..{bad[-3:].tolist() if bad.numel() else []}')
        cols = ne[r].nonzero().reshape(-1)
        print(f'      run1 vs run2 differ at cols {cols[:8].tolist()} ... ({cols.numel()}); row pair partner {r ^ 1} differs: {bool(ne[r ^ 1].any())}')
        print('      run1', [round(float(g1[k][r, c]), 5) for c in cols[:12]])
        print('      run2', [round(float(g2[k][r, c]), 5) for c in cols[:12]])
        print('      input x at those cols', [round(float(g1[kin][r, c]), 5) for c in cols[:12]], ' stats run1/run2:', g1['tv.stats'][:, :, r].flatten()[:6].tolist(), g2['tv.stats'][:, :, r].flatten()[:6].tolist())
    break
