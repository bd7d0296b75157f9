"""Per-leaf gradient error of the VCR step at full width vs the oracle: is the large model's miss (cos 0.98-0.99 on the
first joint layers) depth-dependent bf16 noise or a jump at some layer?  usage: python scripts/debug_vcr_large.py [large|base] [joint_layers]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from merlot_reserve_amd import finetune as F
from merlot_reserve_amd.config import load_config
from oracle import ref_torch as R
from tests.test_vcr_gpu import oracle_batch
from tests.util import relerr, tree_to
dev = torch.device('cuda:0')
name = sys.argv[1] if len(sys.argv) > 1 else 'large'
cfg = load_config(name)
if len(sys.argv) > 2:
    cfg['model']['joint_num_layers'] = int(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg['model']['output_grid'] = [18, 32]
cfg['data'].update(lang_seq_len=144, num_answers=4)
H = cfg['model']['hidden_size']
model = F.MerlotReserveVCR.from_config(cfg, device=dev, seed=0)
batch = F.make_vcr_batch(cfg, B, seed=0, device=dev)
params = model.init_from_dummy_batch(batch)
g = torch.Generator().manual_seed(1)
params['proj']['kernel'] = torch.randn(H, 1, generator=g) * 0.3
logits = model.apply({'params': params}, batch)
eng, store = model.engine, model.params_store
eng.loss_and_grad_logits()
inj = (torch.randn(8 * B, generator=g) * 0.2).to(torch.bfloat16)
eng.dlogits[:, 0] = inj.to(dev)
eng.backward(); torch.cuda.synchronize()
torch.set_num_threads(32)
ob = oracle_batch(batch)
wp = R.tree_map(lambda t: t.clone().requires_grad_(True), tree_to(store.work_tree(), torch.float32))
ref = R.vcr_forward(wp, cfg, ob)
print('logits rel err', relerr(logits, ref))
(ref * inj.float().view(B, 2, 4)).sum().backward()
gt = store.grad_tree()
leaves = [(n, t.grad if t.grad is not None else torch.zeros_like(t)) for n, t in R.tree_leaves(wp)]
gmax = max(float(gr.norm()) for _, gr in leaves)
print('gmax', gmax)
for n, gr in leaves:
    if 'kernel' not in n and 'embedding' not in n:
        continue
    mine = gt
    for part in n.split('/'):
        mine = mine[part]
    gn, err = float(gr.norm()), float((mine.double() - gr.double()).norm())
    cos = float((mine.double().flatten() @ gr.double().flatten()) / (mine.double().norm() * gr.double().norm() + 1e-30))
    print(f'{n:75s} |g|={gn:.3e} rel={err / (gn + 1e-30):.3e} cos={cos:.5f}')
