"""Checkpoint I/O in the reference's on-disk format (mreserve/checkpoint.py:61-119 over flax.training.checkpoints /
flax.serialization, flax==0.3.4 -- requirements.txt:24): ONE msgpack file `<dir>/ckpt_<step>` holding the state dict
{'step', 'params', 'opt_state'}; every array leaf is a msgpack ExtType(1) whose payload is
msgpack((shape, dtype-name, raw C-order bytes)); numpy scalars are ExtType(3) with the same payload; arrays larger
than 2^30 bytes are stored as {'__msgpack_chunked_array__': True, 'shape': {'0': d0, ...}, 'chunks': {'0': a0, ...}}
(flat chunks).  Tuples / optax NamedTuples appear as dicts keyed '0', '1', ... / by field name.

Weights are fp32 in memory and fp16 on disk (`_compress_state`, checkpoint.py:26-37); the bf16 Adam moments are
stored as bfloat16.  The released gs://merlotreserve/ckpts/{base,large}[_resadapt] files are such files and load with
`load_checkpoint(path)['params']` into the tree `ParamStore.load_tree` / `MerlotReserve.apply` take.

Leaves are torch CPU tensors (numpy has no bfloat16)."""
import os
import re

import msgpack
import numpy as np
import torch

_EXT_NDARRAY, _EXT_NPSCALAR = 1, 3
MAX_CHUNK_SIZE = 2 ** 30

_TORCH_BY_NAME = {'float32': torch.float32, 'float16': torch.float16, 'bfloat16': torch.bfloat16, 'float64': torch.float64,
                  'int32': torch.int32, 'int64': torch.int64, 'uint8': torch.uint8, 'int8': torch.int8, 'int16': torch.int16,
                  'bool': torch.bool}
_NAME_BY_TORCH = {v: k for k, v in _TORCH_BY_NAME.items()}


def _as_tensor(x):
    if isinstance(x, torch.Tensor):
        return x.detach().to('cpu').contiguous()
    return torch.from_numpy(np.ascontiguousarray(x))


def _tensor_to_payload(t):
    t = _as_tensor(t)
    raw = t.view(torch.int16).numpy().tobytes() if t.dtype == torch.bfloat16 else t.numpy().tobytes()
    return msgpack.packb((tuple(t.shape), _NAME_BY_TORCH[t.dtype], raw), use_bin_type=True)


def _payload_to_tensor(data):
    shape, dtype_name, raw = msgpack.unpackb(data, raw=True)
    dtype_name = dtype_name.decode() if isinstance(dtype_name, bytes) else dtype_name
    shape = tuple(int(s) for s in shape)
    if dtype_name == 'bfloat16':
        t = torch.frombuffer(bytearray(raw), dtype=torch.int16).view(torch.bfloat16)
    else:
        t = torch.from_numpy(np.frombuffer(raw, dtype=np.dtype(dtype_name)).copy())
    return t.reshape(shape)


def _ext_pack(x):
    if isinstance(x, (torch.Tensor, np.ndarray)):
        return msgpack.ExtType(_EXT_NDARRAY, _tensor_to_payload(x))
    if isinstance(x, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _tensor_to_payload(np.asarray(x)))
    raise TypeError(f'cannot serialise {type(x)}')


def _ext_unpack(code, data):
    if code == _EXT_NDARRAY:
        return _payload_to_tensor(data)
    if code == _EXT_NPSCALAR:
        return _payload_to_tensor(data).reshape(()).item()
    return msgpack.ExtType(code, data)


def _chunk(tree):
    if isinstance(tree, dict):
        return {k: _chunk(v) for k, v in tree.items()}
    if isinstance(tree, (list, tuple)):
        return {str(i): _chunk(v) for i, v in enumerate(tree)}
    if isinstance(tree, (torch.Tensor, np.ndarray)):
        t = _as_tensor(tree)
        nbytes = t.numel() * t.element_size()
        if nbytes > MAX_CHUNK_SIZE:
            per = max(1, MAX_CHUNK_SIZE // t.element_size())
            flat = t.reshape(-1)
            return {'__msgpack_chunked_array__': True, 'shape': {str(i): int(s) for i, s in enumerate(t.shape)},
                    'chunks': {str(i): flat[o:o + per].clone() for i, o in enumerate(range(0, flat.numel(), per))}}
        return t
    return tree


def _unchunk(tree):
    if isinstance(tree, dict):
        if tree.get('__msgpack_chunked_array__', False):
            shape = tuple(tree['shape'][str(i)] for i in range(len(tree['shape'])))
            chunks = [tree['chunks'][str(i)] for i in range(len(tree['chunks']))]
            return torch.cat([c.reshape(-1) for c in chunks]).reshape(shape)
        return {k: _unchunk(v) for k, v in tree.items()}
    return tree


def msgpack_serialize(tree):
    return msgpack.packb(_chunk(tree), default=_ext_pack, strict_types=True)


def msgpack_restore(data):
    return _unchunk(msgpack.unpackb(data, ext_hook=_ext_unpack, raw=False, strict_map_key=False))


def tree_map(fn, tree):
    if isinstance(tree, dict):
        return {k: tree_map(fn, v) for k, v in tree.items()}
    return fn(tree)


def _treemap_cast(from_dtype, to_dtype, tree):
    """checkpoint.py:10-23"""
    return tree_map(lambda x: x.to(to_dtype) if isinstance(x, torch.Tensor) and x.dtype == from_dtype else x, tree)


def bf16_to_f32(params):
    return _treemap_cast(torch.bfloat16, torch.float32, params)


def f32_to_bf16(params):
    return _treemap_cast(torch.float32, torch.bfloat16, params)


def tree_map_nested_keys(f, params, prefix=''):
    """checkpoint.py:104-119: tree map that also passes the '/'-joined key."""
    if isinstance(params, dict):
        return {k: tree_map_nested_keys(f, v, f'{prefix}/{k}' if prefix else k) for k, v in params.items()}
    return f(prefix, params)


def log_param_shapes(params):
    total = 0
    rows = []

    def visit(k, v):
        nonlocal total
        total += v.numel()
        rows.append(f'{k:90s} {str(tuple(v.shape)):22s} {v.numel():>12,d}')
        return v
    tree_map_nested_keys(visit, params)
    print('\n'.join(rows) + f'\nTotal: {total:,d}', flush=True)
    return total


def _natural_key(name):
    return [int(s) if s.isdigit() else s for s in re.split(r'(\d+)', name)]


def latest_checkpoint(ckpt_dir, prefix='ckpt_'):
    names = [n for n in os.listdir(ckpt_dir) if n.startswith(prefix) and not n.endswith('tmp')]
    return os.path.join(ckpt_dir, sorted(names, key=_natural_key)[-1]) if names else None


def save_checkpoint(state, path, keep=None, overwrite=True, no_optimizer=False, prefix='ckpt_', rank=None):
    """state: {'step': int, 'params': tree, 'opt_state': tree or None} (or an object with those attributes / a
    .state_dict()).  fp32 leaves are written as fp16 (checkpoint.py:26-37), bf16 leaves unchanged.

    Data-parallel runs: with partitioned Adam moments (Trainer(shard_optimizer=True), MerlotReserveVCR(shard_optimizer=True))
    `state.state_dict()` gathers the moments from every rank -- a COLLECTIVE -- so EVERY rank must call save_checkpoint; pass `rank=`
    and only rank 0 writes the file (the others take part in the gather and return the path).  `if rank == 0: save_checkpoint(...)`
    around a sharded state would leave rank 0 waiting in the all-gather.  rank=None (the default) always writes."""
    if hasattr(state, 'state_dict'):
        state = state.state_dict()
    step = int(state['step'])
    if rank is not None and int(rank) != 0:
        return os.path.join(path, f'{prefix}{step}')
    sd = {'step': step, 'params': state['params'], 'opt_state': None if no_optimizer else state.get('opt_state')}
    sd = _treemap_cast(torch.float32, torch.float16, tree_map(lambda x: _as_tensor(x) if isinstance(x, (torch.Tensor, np.ndarray)) else x, sd))
    os.makedirs(path, exist_ok=True)
    fn = os.path.join(path, f'{prefix}{step}')
    if os.path.exists(fn) and not overwrite:
        raise ValueError(f'checkpoint {fn} exists')
    tmp = fn + 'tmp'
    with open(tmp, 'wb') as f:
        f.write(msgpack_serialize(sd))
    os.replace(tmp, fn)
    if keep is not None:
        names = sorted([n for n in os.listdir(path) if n.startswith(prefix) and not n.endswith('tmp')], key=_natural_key)
        for n in names[:-keep]:
            os.remove(os.path.join(path, n))
    return fn


def load_checkpoint(path, state=None, step=None, use_bfloat16_weights=False, prefix='ckpt_'):
    """checkpoint.py:80-96.  path: a checkpoint FILE (as PretrainedMerlotReserve.from_pretrained passes, modeling.py:991)
    or a directory of ckpt_<step> files (latest, or `step`).  Returns the state dict with fp16 leaves cast back to fp32;
    when `state` (an object with load_state_dict) is given it is filled and returned instead."""
    fn = path
    if os.path.isdir(path):
        fn = os.path.join(path, f'{prefix}{step}') if step is not None else latest_checkpoint(path, prefix)
        if fn is None:
            raise FileNotFoundError(f'no {prefix}* checkpoint under {path}')
    with open(fn, 'rb') as f:
        sd = msgpack_restore(f.read())
    sd = _treemap_cast(torch.float16, torch.float32, sd)
    if use_bfloat16_weights:
        sd['params'] = f32_to_bf16(sd['params'])
    if state is not None:
        state.load_state_dict(sd)
        return state
    return sd
