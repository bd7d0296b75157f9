"""Checkpoint I/O (merlot_reserve_amd/checkpoint.py) against the flax msgpack wire format (flax.serialization of
flax==0.3.4): a byte-level known answer derived by hand from the format, round trips of the dtypes the reference
writes (fp16 weights, bf16 Adam state), chunked arrays, the ckpt_<step> naming, and the train-state layout."""
import os

import numpy as np
import torch

from merlot_reserve_amd import checkpoint as C


def test_wire_format_known_answer():
    # {'a': float32[2] = [1, 2]}: fixmap(1) 'a' ext8(len 21, type 1) [ array3: array1(2), 'float32', bin8(8 bytes) ]
    raw = np.array([1.0, 2.0], dtype=np.float32).tobytes()
    expect = b'\x81\xa1a' + b'\xc7\x15\x01' + b'\x93' + b'\x91\x02' + b'\xa7float32' + b'\xc4\x08' + raw
    assert C.msgpack_serialize({'a': torch.tensor([1.0, 2.0])}) == expect
    back = C.msgpack_restore(expect)
    assert back['a'].dtype == torch.float32 and back['a'].tolist() == [1.0, 2.0]
    # 0-d int32 (optax `count`): shape () is an empty array
    b = C.msgpack_serialize({'count': torch.tensor(7, dtype=torch.int32)})
    assert b == b'\x81\xa5count' + b'\xc7\x0e\x01' + b'\x93\x90' + b'\xa5int32' + b'\xc4\x04' + np.int32(7).tobytes()


def test_dtypes_and_structure_roundtrip(tmp_path):
    g = torch.Generator().manual_seed(0)
    params = {'enc': {'kernel': torch.randn(5, 3, 64, generator=g), 'bias': torch.zeros(3, 64)}, 'scales': torch.ones(3)}
    mu = C.tree_map(lambda x: (x * 0.1).to(torch.bfloat16), params)
    state = {'step': 12, 'params': params,
             'opt_state': {'0': {'count': torch.tensor(12, dtype=torch.int32), 'mu': mu, 'nu': mu}, '1': {}, '2': {'count': torch.tensor(12, dtype=torch.int32)}, '3': {}}}
    fn = C.save_checkpoint(state, str(tmp_path))
    assert os.path.basename(fn) == 'ckpt_12'
    sd = C.load_checkpoint(str(tmp_path))                    # directory -> latest
    sd2 = C.load_checkpoint(fn)                              # file, as from_pretrained passes it
    for s in (sd, sd2):
        assert s['step'] == 12
        k = s['params']['enc']['kernel']
        assert k.dtype == torch.float32 and k.shape == (5, 3, 64)
        assert torch.equal(k, params['enc']['kernel'].to(torch.float16).float())      # fp16 on disk (checkpoint.py:26-37)
        m = s['opt_state']['0']['mu']['enc']['kernel']
        assert m.dtype == torch.bfloat16 and torch.equal(m, mu['enc']['kernel'])       # bf16 kept bit for bit
        assert s['opt_state']['1'] == {} and int(s['opt_state']['2']['count']) == 12
    C.save_checkpoint({'step': 13, 'params': params, 'opt_state': None}, str(tmp_path), keep=1, no_optimizer=True)
    assert sorted(os.listdir(tmp_path)) == ['ckpt_13']
    assert C.load_checkpoint(str(tmp_path))['opt_state'] is None
    bf = C.load_checkpoint(str(tmp_path), use_bfloat16_weights=True)
    assert bf['params']['scales'].dtype == torch.bfloat16


def test_chunked_arrays(monkeypatch):
    monkeypatch.setattr(C, 'MAX_CHUNK_SIZE', 64)
    t = torch.arange(100, dtype=torch.float32).reshape(10, 10)
    blob = C.msgpack_serialize({'w': t})
    raw = C.msgpack.unpackb(blob, ext_hook=C._ext_unpack, raw=False)
    assert raw['w']['__msgpack_chunked_array__'] is True and raw['w']['shape'] == {'0': 10, '1': 10}
    assert len(raw['w']['chunks']) == 7 and raw['w']['chunks']['0'].numel() == 16
    assert torch.equal(C.msgpack_restore(blob)['w'], t)


def test_train_state_layout_roundtrip(tmp_path):
    from merlot_reserve_amd.config import tiny_config
    from merlot_reserve_amd.params import ParamStore
    from merlot_reserve_amd.trainer import construct_train_state
    cfg = tiny_config(hidden_size=64)
    ps = ParamStore(cfg, 'cpu', seed=1)
    st = construct_train_state(cfg['optimizer'], ps)
    st.step = 5
    ps.mu.copy_(torch.randn(ps.total).to(torch.bfloat16))
    ps.nu.copy_(-torch.rand(ps.total).to(torch.bfloat16))
    sd = st.state_dict()
    assert set(sd) == {'step', 'params', 'opt_state'} and set(sd['opt_state']) == {'0', '1', '2', '3'}
    assert set(sd['params']) == {'audio_encoder', 'contrastive_scales', 'head', 'joint_transformer', 'span_encoder', 'token_encoder', 'vision_encoder'}
    assert tuple(sd['params']['joint_transformer']['layer_00']['attention_layer']['qkv']['kernel'].shape) == (64, 3, 64)
    C.save_checkpoint(st, str(tmp_path))
    ps2 = ParamStore(cfg, 'cpu', seed=2)
    st2 = construct_train_state(cfg['optimizer'], ps2)
    C.load_checkpoint(str(tmp_path), state=st2)
    assert st2.step == 5
    for name, *_ in ps.specs:                 # (the alignment gaps between leaves are not part of the state)
        o, n = ps.offsets[name]
        assert torch.equal(ps2.mu[o:o + n], ps.mu[o:o + n]) and torch.equal(ps2.nu[o:o + n], ps.nu[o:o + n]), name
        assert torch.equal(ps2.master[o:o + n], ps.master[o:o + n].to(torch.float16).float()), name   # fp16 on disk


def test_resadapt_restart(tmp_path):
    """train_fixres.py:78-113: per-process grid, re-derived joint length, short schedule, Adam moments kept, counts reset."""
    from merlot_reserve_amd.config import resadapt_config, tiny_config
    from merlot_reserve_amd.params import ParamStore
    from merlot_reserve_amd.trainer import construct_train_state
    c0, c1 = resadapt_config('base', rank=0), resadapt_config('base', rank=1)
    assert c0['model']['output_grid'] == [18, 32] and c1['model']['output_grid'] == [24, 24]
    assert c0['data']['seq_len'] == c1['data']['seq_len'] == 160 + 8 * 576 // 4 == 1312
    assert abs(c0['data']['random_scale_max'] - 1.1) < 1e-12 and abs(c1['data']['random_scale_max'] - (16 / 9 + 0.1)) < 1e-12
    o = c0['optimizer']
    assert (o['num_train_steps'], o['num_warmup_steps'], o['final_lr_scale']) == (75000, 15000, 0.0) and abs(o['learning_rate'] - 8e-6) < 1e-12
    cfg = tiny_config(hidden_size=64)
    ps = ParamStore(cfg, 'cpu', seed=1)
    st = construct_train_state(cfg['optimizer'], ps)
    st.step = 750
    ps.mu.copy_(torch.randn(ps.total).to(torch.bfloat16))
    C.save_checkpoint(st, str(tmp_path))
    ps2 = ParamStore(cfg, 'cpu', seed=2)
    st2 = construct_train_state(cfg['optimizer'], ps2)
    st2.load_state_dict(C.load_checkpoint(str(tmp_path)), reset_schedule=True)
    o, n = ps.offsets['head/kernel']
    assert st2.step == 0 and torch.equal(ps2.mu[o:o + n], ps.mu[o:o + n])
