"""VCR finetuning step (BASELINE config 5) timing on one MI355X: python scripts/bench_vcr.py [base|large] [B] [steps]
image grid 18x32 (ViT S = 577), answers [B, 2, 4, 144], joint [8B, 288].  Algorithmic FLOPs: SURVEY 8d (per example:
base fwd 0.529 / train 1.586 TF, large 1.842 / 5.526 TF)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlot_reserve_amd import finetune as F
from merlot_reserve_amd.config import load_config

name = sys.argv[1] if len(sys.argv) > 1 else 'large'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = load_config(name)
cfg['model']['output_grid'] = [18, 32]
cfg['data'].update(lang_seq_len=144, num_answers=4)
cfg['optimizer'] = {'beta_2': 0.98, 'eps': 1e-6, 'learning_rate': 5e-6, 'num_train_steps': 1000, 'num_warmup_steps': 100,
                    'use_bfloat16_adam': True, 'weight_decay_rate': 0.1, 'do_bias_correction': True}
dev = torch.device('cuda:0')
model = F.MerlotReserveVCR.from_config(cfg, device=dev)
batches = [F.make_vcr_batch(cfg, B, seed=i, device=dev) for i in range(2)]
model.init_from_dummy_batch(batches[0])
state, tx = F.construct_finetuning_train_state(cfg['optimizer'], model)
graph = '--eager' not in sys.argv
if graph:
    step = F.VCRGraphStep(state, batches[0])
    plans = [F.build_vcr_plan(b['answers'], model.engine.d) for b in batches]
    run = lambda i: step(batches[i % 2], plans[i % 2])
else:
    run = lambda i: F.finetune_train_step(state, batches[i % 2])
for i in range(3):
    run(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    run(i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
info = model.engine.loss_info()
H, Lv, Lj = cfg['model']['hidden_size'], cfg['model']['vit_num_layers'], cfg['model']['joint_num_layers']
enc = lambda n, S, L: n * S * L * (24 * H * H + 4 * S * H)
fwd = enc(1, 577, Lv) + enc(8, 288, Lj) + 576 * 2 * 768 * H + 144 * 20 * H * H
print(f'VCR {name} B={B} ({"hipGraph" if graph else "eager"}): {dt * 1e3:.2f} ms/step, {B / dt:.1f} examples/s, loss {info["loss"]:.4f}, '
      f'{3 * fwd * B / dt / 1e12:.1f} TFLOP/s algorithmic ({3 * fwd * B / dt / 2.5e15 * 100:.1f} % of bf16 MFMA peak)')
