import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_vcr_gpu import setup, oracle_batch
from oracle import ref_torch as R
from tests.util import tree_to, relerr
dev = torch.device('cuda:0')
for variant in ('clean',):
    F, cfg, model, batch, params = setup(dev)
    if variant == 'clean':
        b2 = F.make_vcr_batch(cfg, 2, seed=0, device=dev)
        batch['answers'] = b2['answers']
    logits = model.apply({'params': params}, batch)
    eng, store = model.engine, model.params_store
    eng.loss_and_grad_logits()
    g = torch.Generator().manual_seed(9)
    inj = (torch.randn(16, generator=g) * 0.2).to(torch.bfloat16)
    eng.dlogits[:, 0] = inj.to(dev)
    eng.backward(); torch.cuda.synchronize()
    print('d_pooled check', relerr(eng.d_pooled, inj.float()[:, None] * store.work_tree()['proj']['kernel'].float().T))
    ob = oracle_batch(batch)
    wp = R.tree_map(lambda t: t.clone().requires_grad_(True), tree_to(store.work_tree(), torch.float32))
    ref = R.vcr_forward(wp, cfg, ob)
    dl = eng.dlogits[:, 0].float().cpu().view(2, 2, 4)
    (ref * dl).sum().backward()
    gt = store.grad_tree()
    print('=====', variant)
    for name, t in R.tree_leaves(wp):
        g = t.grad if t.grad is not None else torch.zeros_like(t)
        mine = gt
        for part in name.split('/'):
            mine = mine[part]
        gn = float(g.norm()); err = float((mine.double() - g.double()).norm())
        cos = float((mine.double().flatten() @ g.double().flatten()) / (mine.double().norm() * g.double().norm() + 1e-30))
        if True:
            print(f'{name:70s} |g|={gn:.3e} err={err:.3e} cos={cos:.4f}')
