"""A/B of the backward's reduction paths on the tiny configuration: per-leaf gradient error against autograd of the oracle
for (default) deferred reductions, MR_NO_TOWER_DEFER=1, MR_NO_ATTN_COLSUM=1 and MR_NO_BATCH_REDUCE=1.  Diagnostic only."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from tests.util import oracle_batch, oracle_draws, tiny_setup, tree_to
from oracle import ref_torch as R
from merlot_reserve_amd.config import Dims
from merlot_reserve_amd.engine import PretrainEngine
from merlot_reserve_amd.planner import build_plan

dev = torch.device('cuda:0')
B = 2
cfg, store, batch, splits, z = tiny_setup(B=B, seed=3, device=dev, hidden_size=128)
eng = PretrainEngine(cfg, B, store, dev)
plan = build_plan(batch, Dims(cfg, B), splits, z)
g = torch.Generator().manual_seed(1)
dE = (torch.randn(eng.R, eng.d.H, generator=g) * 1e-2).to(torch.bfloat16)

params = tree_to(store.work_tree(), torch.float32)
params = R.tree_map(lambda t: t.clone().requires_grad_(True), params)
osp, oz = oracle_draws(splits, z)
preds = R.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz)
SECTIONS = (('imgs_to_audio', 'x', 'i2a_x'), ('imgs_to_audio', 'y', 'i2a_y'), ('text_to_audio', 'x', 't2a_x'),
            ('text_to_audio', 'y', 't2a_y'), ('text_to_audio', 'y_extra', 't2a_ye'), ('stuff_to_span', 'x', 's2s_x'),
            ('stuff_to_span', 'y', 's2s_y'))
total = 0.0
for k, k2, name in SECTIONS:
    o, n = eng.sec[name]
    total = total + (preds[k][k2] * dE[o:o + n].float()).sum()
total.backward()
ref = {n: (t.grad if t.grad is not None else torch.zeros_like(t)).double() for n, t in R.tree_leaves(params)}

def run(env):
    for k in ('MR_NO_TOWER_DEFER', 'MR_NO_ATTN_COLSUM', 'MR_NO_BATCH_REDUCE'):
        os.environ.pop(k, None)
    os.environ.update(env)
    eng.forward(batch, plan=plan)
    eng.dE.copy_(dE.to(dev))
    eng.backward()
    torch.cuda.synchronize()
    gt = store.grad_tree()
    out = {}
    for name in ref:
        m = gt
        for part in name.split('/'):
            m = m[part]
        out[name] = m.double().cpu().clone()
    return out

modes = {'default': {}, 'notower': {'MR_NO_TOWER_DEFER': '1'}, 'noattncs': {'MR_NO_ATTN_COLSUM': '1'}, 'nobatch': {'MR_NO_BATCH_REDUCE': '1'}}
res = {k: run(v) for k, v in modes.items()}
res['default2'] = run({})
print('leaf-wise: rel err vs oracle per mode;  diff(default,nobatch)/|ref|')
rows = []
for name, r in ref.items():
    rn = float(r.norm()) + 1e-30
    e = {k: float((res[k][name] - r).norm()) / rn for k in res}
    d = float((res['default'][name] - res['nobatch'][name]).norm()) / rn
    d2 = float((res['default'][name] - res['default2'][name]).norm()) / rn
    rows.append((max(e['default'] / (e['nobatch'] + 1e-4), d), name, rn, e, d, d2))
rows.sort(key=lambda t: -t[0])
for _, name, rn, e, d, d2 in rows[:40]:
    print(f"{name:75s} |g|={rn:.2e} " + ' '.join(f'{k}={v:.2e}' for k, v in e.items()) + f' d={d:.2e} rerun={d2:.1e}')
tot = {k: sum(float((res[k][n] - ref[n]).norm()) ** 2 for n in ref) ** 0.5 / sum(float(ref[n].norm()) ** 2 for n in ref) ** 0.5 for k in res}
print('global rel err', tot)
