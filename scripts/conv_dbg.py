import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
from merlot_reserve_amd.config import tiny_config
from merlot_reserve_amd.synthetic import make_batch
from merlot_reserve_amd.trainer import Trainer
dev = torch.device('cuda:0')
cfg = tiny_config(seq_len=80, lang_seq_len=40)
cfg['optimizer'].update(num_warmup_steps=10, learning_rate=1e-3, num_train_steps=400)
B = 2
tr = Trainer(cfg, B, dev, seed=1)
batches = [make_batch(cfg, B, seed=70 + i, device=dev) for i in range(2)]
plans = [tr.plan(b) for b in batches]
tr.train_step(batches[0], plan=plans[0])
first = tr.loss_info()['loss']
graph = os.environ.get('NOGRAPH') != '1'
if graph: tr.capture(batches[0])
hist = []
for i in range(int(os.environ.get("STEPS", "100"))):
    if graph: tr.train_step_graph(batches[i % 2], plans[i % 2])
    else: tr.train_step(batches[i % 2], plan=plans[i % 2])
    if i % 100 == 99: hist.append(round(tr.loss_info()['loss'], 3))
print(os.environ.get('TAG'), first, hist)
