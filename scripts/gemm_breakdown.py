"""Per-shape GEMM time inside one eager training step (HIP events around every mr_gemm launch)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import ops
from merlot_reserve_amd.config import load_config
from merlot_reserve_amd.synthetic import make_batch
from merlot_reserve_amd.trainer import Trainer
dev = torch.device('cuda:0')
cfg = load_config(sys.argv[1] if len(sys.argv) > 1 else 'base')
B = 4
tr = Trainer(cfg, B, dev)
batch = make_batch(cfg, B, seed=1, device=dev)
plan = tr.plan(batch)
for _ in range(2):
    tr.train_step(batch, plan=plan)
torch.cuda.synchronize()
ops.GEMM_PROFILE = []
tr.train_step(batch, plan=plan)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for e0, e1, fl, sig in ops.GEMM_PROFILE:
    a = agg[sig]
    a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += fl
tot = sum(a[1] for a in agg.values())
print(f'total GEMM ms {tot:.2f}, TF {sum(a[2] for a in agg.values()) / 1e12:.2f}')
print('  M     N     K  tA tB bias rot c2 act res aux |   n   ms_total   us_each   TF/s  share')
for sig, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(sig, '| %3d %8.3f %9.1f %7.1f %5.1f%%' % (a[0], a[1], a[1] / a[0] * 1e3, a[2] / a[1] / 1e9, 100 * a[1] / tot))
