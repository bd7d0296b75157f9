"""Where do two reduction modes of the backward first differ inside the TRAINER's step?  Fresh Trainer per mode (same seed),
N eager steps, after each: per-leaf comparison of params.grad and params.work.  Diagnostic only."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from merlot_reserve_amd.config import tiny_config
from merlot_reserve_amd.synthetic import make_batch
from merlot_reserve_amd.trainer import Trainer
dev = torch.device('cuda:0')
B, N = 2, 4
KEYS = ('MR_NO_TOWER_DEFER', 'MR_NO_ATTN_COLSUM', 'MR_NO_BATCH_REDUCE')

def run(env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)
    cfg = tiny_config(seq_len=80, lang_seq_len=40)
    cfg['optimizer'].update(num_warmup_steps=10, learning_rate=1e-3, num_train_steps=400)
    tr = Trainer(cfg, B, dev, seed=1)
    batches = [make_batch(cfg, B, seed=70 + i, device=dev) for i in range(2)]
    plans = [tr.plan(b) for b in batches]
    snaps = []
    for i in range(N):
        tr.train_step(batches[i % 2], plan=plans[i % 2])
        torch.cuda.synchronize()
        snaps.append((tr.params.grad.float().cpu().clone(), tr.params.work.float().cpu().clone(), tr.engine.dE.float().cpu().clone(),
                      tr.loss_info()['loss']))
    return tr, snaps

modes = {'default': {}, 'default_again': {}, 'notower': {'MR_NO_TOWER_DEFER': '1'}, 'nobatch': {'MR_NO_BATCH_REDUCE': '1'}}
out = {k: run(v) for k, v in modes.items()}
tr0, base = out['default']
for k in list(modes)[1:]:
    _, s = out[k]
    print('=== default vs', k)
    for i in range(N):
        dg = (base[i][0] - s[i][0]); dw = (base[i][1] - s[i][1]); de = base[i][2] - s[i][2]
        print(f' step {i}: loss {base[i][3]:.5f} / {s[i][3]:.5f}  |dgrad|={float(dg.norm()):.3e} of {float(base[i][0].norm()):.3e}   |dwork|={float(dw.norm()):.3e}  |ddE|={float(de.norm()):.3e}')
        if float(dg.norm()) > 0 and i <= 1:
            rows = []
            for name, (o, n) in tr0.params.offsets.items():
                d = float(dg[o:o + n].norm())
                if d > 0:
                    rows.append((d / (float(base[i][0][o:o + n].norm()) + 1e-30), d, name, int((dg[o:o + n] != 0).sum()), n))
            rows.sort(reverse=True)
            for r in rows[:25]:
                print(f'    rel={r[0]:.3e} abs={r[1]:.3e} nz={r[3]}/{r[4]} {r[2]}')
