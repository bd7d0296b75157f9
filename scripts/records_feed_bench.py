"""The base pretraining step fed from TFRecord shards END TO END: synthetic shards with the reference's record layout on local disk ->
records.ShardFeeder (a feeder process with 16 parser processes, shared-memory batch slots) -> loader.PrefetchLoader (pinned ring + copy stream) -> Trainer.train_step_graph (hipGraph replay), against
the same trainer on resident batches.  python scripts/records_feed_bench.py [n_records=192] [steps=40] [workers=16]"""
import os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import records as R
from merlot_reserve_amd.config import load_config
from merlot_reserve_amd.loader import PrefetchLoader
from merlot_reserve_amd.synthetic import make_batch
from merlot_reserve_amd.trainer import Trainer


def main():
    if os.environ.get('FEED_TORCH_THREADS'):
        torch.set_num_threads(int(os.environ['FEED_TORCH_THREADS']))
    print('torch host threads:', torch.get_num_threads(), flush=True)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    tmp = tempfile.mkdtemp(prefix='mr_feed_')
    cfg = load_config('base')
    B = 4
    rng = np.random.default_rng(0)
    for s in range(4):
        R.write_tfrecord(os.path.join(tmp, f'train{s:05d}of00004.tfrecord'), [R.make_synthetic_record(cfg, rng, frame_hw=(288, 512)) for _ in range(n // 4)])     # 288 x 512: the corpus' stored size (data/process.py:418-423 of the reference)
    cfg['data'] = dict(cfg['data'], train_fns=os.path.join(tmp, 'train{:05d}of00004.tfrecord'), num_train_files=4)
    cfg['device'] = dict(cfg.get('device', {}), batch_size=B, shuffle_buffer_size=16, n_fns_per_cycle=4)
    dev = torch.device('cuda:0')
    tr = Trainer(cfg, B, dev, seed=0)
    b0 = make_batch(cfg, B, seed=1, device=dev)
    tr.train_step(b0, plan=tr.plan(b0))
    tr.capture(b0)
    res = [make_batch(cfg, B, seed=2 + i, device=dev) for i in range(2)]
    plans = [tr.plan(b) for b in res]
    for i in range(5):
        tr.train_step_graph(res[i % 2], plans[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.train_step_graph(res[i % 2], plans[i % 2])
    torch.cuda.synchronize()
    print(f'resident batches: {(time.perf_counter() - t0) / steps * 1e3:.2f} ms / step', flush=True)
    for fast in (False, True):                 # (False: the float resampler of rounds 1-5; True: the 8-bit path, the default for bf16 batches since round 6)
        cfg['data']['fast_image_resize'] = fast
        feeder = R.ShardFeeder(cfg, rank=0, world=1, seed=3, workers=workers, slots=4)       # the reader in its own process: no GIL shared with the step's host work
        feed = PrefetchLoader(feeder, dev, depth=2)
        t0, done = None, 0
        acc = {'next': 0.0, 'plan': 0.0, 'step': 0.0}
        it = iter(feed)
        i = 0
        while True:
            ta = time.perf_counter()
            try:
                b = next(it)
            except StopIteration:
                break
            tb = time.perf_counter()
            if i == 8:                      # the pool is up, the rings are full
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                acc = {'next': 0.0, 'plan': 0.0, 'step': 0.0}
            pl = tr.plan(b)
            tc = time.perf_counter()
            tr.train_step_graph(b, pl)
            td = time.perf_counter()
            acc['next'] += tb - ta; acc['plan'] += tc - tb; acc['step'] += td - tc
            done = i
            if i == 8 + steps:
                break
            i += 1
        print('   host ms / step:', {k: round(v / max(done - 8, 1) * 1e3, 2) for k, v in acc.items()}, flush=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f'fed from shards ({workers} parser processes{", fast_image_resize" if fast else ""}): {dt / (done - 8) * 1e3:.2f} ms / step, loss {tr.loss_info()["loss"]:.4f}', flush=True)
        feeder.close()


if __name__ == '__main__':
    main()
