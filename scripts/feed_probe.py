"""Where does the shard-fed step lose reader throughput?  The same ShardFeeder -> PrefetchLoader chain (A) alone, (B) with the host planner per batch,
(C) with GPU steps but no planner, (D) with both (= scripts/records_feed_bench.py)."""
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
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 14
    tmp = tempfile.mkdtemp(prefix='mr_feed_')
    cfg = load_config('base')
    rng = np.random.default_rng(0)
    for s in range(4):
        R.write_tfrecord(os.path.join(tmp, f'train{s:05d}of00004.tfrecord'), [R.make_synthetic_record(cfg, rng, frame_hw=(360, 640)) for _ in range(80)])
    cfg['data'] = dict(cfg['data'], train_fns=os.path.join(tmp, 'train{:05d}of00004.tfrecord'), num_train_files=4)
    cfg['device'] = dict(cfg.get('device', {}), batch_size=4, shuffle_buffer_size=16, n_fns_per_cycle=4)
    dev = torch.device('cuda:0')
    tr = Trainer(cfg, 4, dev, seed=0)
    b0 = make_batch(cfg, 4, seed=1, device=dev)
    tr.train_step(b0, plan=tr.plan(b0))
    tr.capture(b0)
    plan0 = tr.plan(b0)
    for label, do_plan, do_step in (('A reader alone', False, False), ('B + planner', True, False), ('C + GPU steps (fixed plan)', False, True), ('D + both', True, True)):
        with R.ShardFeeder(cfg, rank=0, world=1, seed=3, workers=workers, slots=4, epochs=1) as feeder:
            t0, n = None, 0
            for i, b in enumerate(PrefetchLoader(feeder, dev, depth=2)):
                if i == 10:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                pl = tr.plan(b) if do_plan else plan0
                if do_step:
                    tr.train_step_graph(b, pl)
                n = i
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f'{label}: {dt / (n - 10) * 1e3:.1f} ms / batch = {(n - 10) * 4 / dt:.0f} records / s', flush=True)


if __name__ == '__main__':
    main()
