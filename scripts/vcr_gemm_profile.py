"""Every GEMM launch of one eager VCR finetuning step (large, B = 4) with the kernel it was routed to: shape, us, TF/s -- sorted by time per kernel/shape."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from merlot_reserve_amd import finetune as F, ops
from merlot_reserve_amd.config import load_config

name = sys.argv[1] if len(sys.argv) > 1 else 'large'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = load_config(name)
cfg['model']['output_grid'] = [18, 32]
cfg['data'].update(lang_seq_len=144, num_answers=4)
cfg['optimizer'] = {'beta_2': 0.98, 'eps': 1e-6, 'learning_rate': 5e-6, 'num_train_steps': 1000, 'num_warmup_steps': 100,
                    'use_bfloat16_adam': True, 'weight_decay_rate': 0.1, 'do_bias_correction': True}
dev = torch.device('cuda:0')
model = F.MerlotReserveVCR.from_config(cfg, device=dev)
batch = F.make_vcr_batch(cfg, B, seed=0, device=dev)
model.init_from_dummy_batch(batch)
state, tx = F.construct_finetuning_train_state(cfg['optimizer'], model)
for _ in range(2):
    F.finetune_train_step(state, batch)
torch.cuda.synchronize()
ops.set_option('gemm_trace', 1)
ops.GEMM_PROFILE = []
for _ in range(3):
    F.finetune_train_step(state, batch)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for e0, e1, fl, shape, kern, _bytes in ops.GEMM_PROFILE:
    k = (kern, shape)
    a = agg.setdefault(k, [0, 0.0, fl])
    a[0] += 1
    a[1] += e0.elapsed_time(e1) * 1e3
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
tot = sum(v[1] for v in agg.values()) / 3
print(f'GEMM launches per step {sum(v[0] for v in agg.values()) / 3:.0f}, {tot / 1e3:.2f} ms per step (eager, every launch alone on its stream)')
for (kern, shape), (n, us, fl) in rows[:40]:
    print(f'{kern:46s} {str(shape):70s} n/step={n / 3:5.1f} avg={us / n:7.1f} us  ms/step={us / 3e3:6.2f}  {fl / (us / n) / 1e6:7.1f} TF/s')
