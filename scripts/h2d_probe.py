"""Where the host-fed step's extra time goes: the PrefetchLoader's host-side pieces timed one by one (pinned memcpy, H2D enqueue), and the
graph-replayed step fed from the loader vs from resident tensors."""
import os, sys, time, itertools, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from merlot_reserve_amd import config as cfg, synthetic
from merlot_reserve_amd.trainer import Trainer
from merlot_reserve_amd.loader import PrefetchLoader
dev = torch.device('cuda:0')
c = cfg.load_config('base')
B = 4
tr = Trainer(c, B, dev)
batches = [synthetic.make_batch(c, B, seed=1234 + i, device=dev) for i in range(2)]
plans = [tr.plan(b) for b in batches]
tr.train_step(batches[0], plan=plans[0])
tr.capture(batches[0])
def sync(): torch.cuda.synchronize()
for _ in range(3): tr.train_step_graph(batches[0], plans[0])
sync(); t = time.perf_counter()
for i in range(10): tr.train_step_graph(batches[i % 2], plans[i % 2])
sync(); print(f'resident: {(time.perf_counter() - t) * 100:.2f} ms / step')
host = [{k: (v.cpu() if torch.is_tensor(v) else v) for k, v in b.items()} for b in batches]
pin = {k: torch.empty(tuple(host[0][k].shape), dtype=torch.bfloat16, pin_memory=True) for k in ('images', 'audio_clips')}
t = time.perf_counter()
for _ in range(10):
    for k in pin: pin[k].copy_(host[0][k])
print(f'pageable -> pinned memcpy of one batch: {(time.perf_counter() - t) * 100:.2f} ms  ({sum(v.numel() * 2 for v in pin.values()) / 1e6:.1f} MB)')
d = {k: torch.empty_like(v, device=dev) for k, v in pin.items()}
sync(); t = time.perf_counter()
for _ in range(10):
    for k in pin: d[k].copy_(pin[k], non_blocking=True)
sync(); print(f'pinned -> device copy of one batch: {(time.perf_counter() - t) * 100:.2f} ms')
loader = PrefetchLoader(itertools.islice(itertools.cycle(host), 14), dev, depth=2)
ts = []
for i, b in enumerate(loader):
    if i == 2: sync(); t = time.perf_counter()
    t0 = time.perf_counter()
    tr.train_step_graph(b, plans[i % 2])
    ts.append(time.perf_counter() - t0)
sync(); print(f'loader-fed: {(time.perf_counter() - t) / 12 * 1e3:.2f} ms / step; host time inside train_step_graph: {sum(ts[2:]) / 12 * 1e3:.2f} ms')
