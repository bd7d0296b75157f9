import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.util import *
from merlot_reserve_amd.config import Dims
from merlot_reserve_amd.engine import PretrainEngine
from merlot_reserve_amd.planner import build_plan
from oracle import ref_torch as R
dev = torch.device('cuda:0')
B = 2
cfg, store, batch, splits, z = tiny_setup(B=B, seed=3, device=dev)
eng = PretrainEngine(cfg, B, store, dev)
plan = build_plan(batch, Dims(cfg, B), splits, z)
eng.forward(batch, plan=plan)
eng.loss_and_grad_outputs()
torch.cuda.synchronize()
dE = eng.dE.float().cpu().clone()
eng.backward()
torch.cuda.synchronize()
params = tree_to(store.work_tree(), torch.float32)
params = R.tree_map(lambda t: t.clone().requires_grad_(True), params)
osp, oz = oracle_draws(splits, z)
preds, dbg = R.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz, return_debug=True)
for k in preds:
    for k2 in ('x', 'y', 'y_extra'):
        if k2 in preds[k]:
            preds[k][k2].retain_grad()
for k in ('joint_head', 'imgs_seq', 'audio_seq', 'audio_cls', 'imgs_cls', 'joint_x'):
    dbg[k].retain_grad()
loss, info = R.loss_fn_given_preds([preds])
loss.backward()
sec = eng.sec
def s(k): return dE[sec[k][0]:sec[k][0] + sec[k][1]]
for (o, k2, name) in (('imgs_to_audio', 'x', 'i2a_x'), ('imgs_to_audio', 'y', 'i2a_y'), ('text_to_audio', 'x', 't2a_x'),
                      ('text_to_audio', 'y', 't2a_y'), ('text_to_audio', 'y_extra', 't2a_ye'), ('stuff_to_span', 'x', 's2s_x'),
                      ('stuff_to_span', 'y', 's2s_y')):
    g = preds[o][k2].grad
    print(name, 'dE rel', relerr(s(name), g), 'norms', float(s(name).norm()), float(g.norm()))
print('d_hj', relerr(eng.d_hj, dbg['joint_head'].grad.reshape(-1, eng.d.H)))
print('d_imgs_seq', relerr(eng.d_imgs_seq, dbg['imgs_seq'].grad.reshape(-1, eng.d.H)))
print('d_audio_seq', relerr(eng.d_audio_seq, dbg['audio_seq'].grad.reshape(-1, eng.d.H)))
print('d_a_cls', relerr(eng.d_a_cls, dbg['audio_cls'].grad.reshape(-1, eng.d.H)))
print('d_v_cls', relerr(eng.d_v_cls, dbg['imgs_cls'].grad.reshape(-1, eng.d.H)))
print('Dj (d joint_x)', relerr(eng.Dj, dbg['joint_x'].grad.reshape(-1, eng.d.H)))
print('dls', eng.dls.tolist(), params['contrastive_scales'].grad.tolist())
