"""The real-data reader on synthetic shards: writes a few TFRecord shards with the reference's record layout (records.make_synthetic_record: frames stored
at 288 x 512 -- the reference's data/process.py:418-423 resizes every extracted 360p frame to shorter side 288, longer side <= 512, before it is encoded;
round 5 timed 360 x 640 frames, 1.56 x the pixels; pass `hw=360x640` as the third argument for that), reads them back through records.make_dataset at several worker counts and prints records / s -- what one GPU's step
consumes is B / ms_per_step (base, B = 4: ~133 records / s).  CPU only.   python scripts/records_demo.py [n_records] [tmpdir]"""
import os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from merlot_reserve_amd import records as R
from merlot_reserve_amd.config import load_config


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    tmp = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix='mr_records_')
    hw = tuple(int(v) for v in sys.argv[3].split('x')) if len(sys.argv) > 3 else (288, 512)
    cfg = load_config('base')
    cfg['device'] = dict(cfg.get('device', {}), shuffle_buffer_size=8)
    rng = np.random.default_rng(0)
    t0 = time.time()
    fns = []
    for s in range(2):
        fn = os.path.join(tmp, f'train{s:05d}of00002.tfrecord')
        R.write_tfrecord(fn, [R.make_synthetic_record(cfg, rng, frame_hw=hw) for _ in range(n // 2)])
        fns.append(fn)
    size = sum(os.path.getsize(f) for f in fns)
    print(f'frames stored at {hw[0]} x {hw[1]}; wrote {n} records, {size / 1e6:.1f} MB ({size / n / 1e3:.0f} KB / record) in {time.time() - t0:.1f} s')
    t0 = time.time()
    recs = [r for f in fns for r in R.read_tfrecord(f)]
    print(f'container: {len(recs)} records scanned + checksummed in {(time.time() - t0) * 1e3:.1f} ms ({size / 1e6 / (time.time() - t0):.0f} MB/s)')
    t0 = time.time()
    ex = [R.parse_example(r) for r in recs]
    print(f'tf.train.Example parse: {(time.time() - t0) / len(recs) * 1e3:.2f} ms / record')
    for workers, procs, fast in ((0, False, False), (8, True, False), (16, True, False), (0, False, True), (8, True, True), (16, True, True)):
        cfg['data']['fast_image_resize'] = fast          # (False: the float resampler of rounds 1-5; True: the 8-bit path, the default for bf16 batches since round 6)
        t0 = time.time()
        nb, t_first = 0, None
        for _ in R.make_dataset(cfg, fns, 4, is_training=True, seed=1, workers=workers, processes=procs):
            nb += 1
            if t_first is None:
                t_first = time.time()          # (the pool has started and the first two chunks are parsed)
        dt = time.time() - t0
        sustained = (nb - 1) * 4 / max(time.time() - t_first, 1e-9)
        print(f'   sustained (first batch -> last batch, pool start-up excluded): {sustained:.1f} records / s') if nb > 1 else None
        print(f'workers={workers} {"processes" if procs else "threads"}{" fast_image_resize" if fast else ""}: {nb} batches of 4 in {dt:.2f} s = {nb * 4 / dt:.1f} records / s (incl. pool start-up)')


if __name__ == '__main__':        # (spawned parser workers re-import this file: nothing may run at import)
    main()
