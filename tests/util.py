"""Shared helpers for the parity tests (test infrastructure: may import oracle/)."""
import numpy as np
import torch

from merlot_reserve_amd.config import Dims, tiny_config
from merlot_reserve_amd.params import ParamStore
from merlot_reserve_amd.synthetic import make_batch, make_draws

INT_KEYS = ['text2audio', 'audio2text', 'audio_text_matching', 'random_text']


def oracle_batch(batch, dtype=torch.float32):
    """synthetic.make_batch layout -> the dict the oracle's pretrain_forward consumes (int64 indices, float inputs)."""
    ob = {}
    for k, v in batch.items():
        if isinstance(v, np.ndarray):
            ob[k] = torch.from_numpy(v.astype(np.int64))
        else:
            ob[k] = v.detach().to('cpu').to(torch.float32).to(dtype)
    return ob


def oracle_draws(splits, z):
    return [torch.from_numpy(s.astype(np.int64)) for s in splits], torch.from_numpy(z)


def tree_to(tree, dtype):
    if isinstance(tree, dict):
        return {k: tree_to(v, dtype) for k, v in tree.items()}
    return tree.to(dtype)


def tiny_setup(B=2, seed=3, device='cpu', hidden_size=128, model_flags=None, data_flags=None):
    cfg = tiny_config(hidden_size=hidden_size, seq_len=80, lang_seq_len=40)
    cfg['model'].update(model_flags or {})
    cfg['data'].update(data_flags or {})
    store = ParamStore(cfg, device, seed=seed)
    batch = make_batch(cfg, B, seed=seed, device=device)
    splits, z = make_draws(cfg, B, seed=seed)
    # force at least one video-source split so the masking branch is exercised
    splits[0][0] = 1
    return cfg, store, batch, splits, z


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
