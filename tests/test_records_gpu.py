"""The real-data path end to end on the GPU: shards on disk -> records.input_fn_builder -> loader.PrefetchLoader -> Trainer.train_step
(pretrain/dataloader.py:906-955 feeding pretrain/train.py's loop), and against the oracle's forward on the same record-fed batch."""
import numpy as np
import pytest
import torch

from util import oracle_batch, oracle_draws, relerr, tree_to

pytestmark = pytest.mark.gpu


def test_training_from_record_shards(dev, tmp_path):
    from merlot_reserve_amd import records as R
    from merlot_reserve_amd.config import tiny_config
    from merlot_reserve_amd.loader import PrefetchLoader
    from merlot_reserve_amd.synthetic import make_draws
    from merlot_reserve_amd.trainer import Trainer
    from oracle import ref_torch as O
    cfg = tiny_config()
    cfg['optimizer'].update(num_warmup_steps=1, learning_rate=1e-3)
    rng = np.random.default_rng(9)
    for s in range(2):
        R.write_tfrecord(tmp_path / f'train{s:05d}of00002.tfrecord', [R.make_synthetic_record(cfg, rng) for _ in range(4)])
    B = 2
    cfg['data'] = dict(cfg['data'], train_fns=str(tmp_path / 'train{:05d}of00002.tfrecord'), num_train_files=2)
    cfg['device'] = dict(cfg.get('device', {}), batch_size=B, shuffle_buffer_size=4, n_fns_per_cycle=2)
    tr = Trainer(cfg, B, dev, seed=0)
    feed = PrefetchLoader(R.input_fn_builder(cfg, rank=0, world=1, seed=4, epochs=1, workers=2), dev, depth=2)
    losses = []
    for i, batch in enumerate(feed):
        draws = make_draws(cfg, B, seed=20 + i)
        if i == 0:          # the forward on a record-fed batch against the oracle (bf16 program: the tolerance of tests/test_pretrain_gpu.py)
            params = tree_to(tree_to(tr.params.master_tree(), torch.bfloat16), torch.float32)
            tr.forward_and_loss(batch, draws=draws)
            loss = tr.loss_info()['loss']
            osp, oz = oracle_draws(*draws)
            with torch.no_grad():
                opreds = O.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz)
                oloss, _ = O.loss_fn_given_preds([opreds])
            assert abs(loss - float(oloss)) <= 2e-2 * abs(float(oloss)), (loss, float(oloss))
        tr.train_step(batch, draws=draws)
        losses.append(tr.loss_info()['loss'])
    torch.cuda.synchronize()
    assert len(losses) == 4 and all(np.isfinite(losses)) and tr.state.step == 4
