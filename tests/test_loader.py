"""Host -> device input prefetch (merlot_reserve_amd/loader.py; the reference's prefetch_to_device, pretrain/dataloader.py:957-958):
batches come out in order, bit-identical to a direct upload, integer streams untouched; on the GPU a training run fed through the
loader matches the same run fed with resident batches."""
import numpy as np
import pytest
import torch

from merlot_reserve_amd.config import tiny_config
from merlot_reserve_amd.loader import PrefetchLoader
from merlot_reserve_amd.synthetic import make_batch


def _host_batches(cfg, n, B=2):
    return [make_batch(cfg, B, seed=40 + i, device='cpu') for i in range(n)]


@pytest.mark.parametrize('depth', [2, 3])
def test_loader_order_and_content_cpu(depth):
    cfg = tiny_config(seq_len=80, lang_seq_len=40)
    host = _host_batches(cfg, 5)
    out = []
    for b in PrefetchLoader(iter(host), 'cpu', depth=depth):
        out.append({k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()})
    assert len(out) == len(host)
    for got, want in zip(out, host):
        assert set(got) == set(want)
        for k, v in want.items():
            if torch.is_tensor(v):
                assert torch.equal(got[k], v)
            else:
                assert got[k] is v                      # integer streams are passed through for the host-side planner


def test_loader_accepts_numpy_and_casts_to_wire_dtype():
    cfg = tiny_config(seq_len=80, lang_seq_len=40)
    b = make_batch(cfg, 1, seed=3, device='cpu', float_dtype=torch.float32)
    nb = dict(b, images=b['images'].numpy(), audio_clips=b['audio_clips'].numpy())
    got = next(PrefetchLoader(iter([nb]), 'cpu'))
    assert got['images'].dtype == torch.bfloat16 and torch.equal(got['images'], b['images'].to(torch.bfloat16))


@pytest.mark.gpu
def test_training_through_the_loader_matches_resident_batches(dev):
    from merlot_reserve_amd.trainer import Trainer
    cfg = tiny_config(seq_len=80, lang_seq_len=40)
    cfg['optimizer'].update(num_warmup_steps=1, learning_rate=1e-3)
    B = 2
    host = _host_batches(cfg, 6, B)
    res = []
    for mode in ('resident', 'loader'):
        tr = Trainer(cfg, B, dev, seed=0)
        first = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in host[0].items()}
        tr.train_step(first, plan=tr.plan(first))
        tr.capture(first)
        if mode == 'resident':
            feed = ({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()} for b in host[1:])
        else:
            feed = PrefetchLoader(iter(host[1:]), dev, depth=2)
        for b in feed:
            tr.train_step_graph(b, tr.plan(b))
        torch.cuda.synchronize()
        res.append(tr.params.master.detach().cpu())
    assert torch.equal(res[0], res[1])
