"""`mreserve.modeling` of the reference, name for name, on the MI355X kernels (forward / inference surface).

    reference (JAX / Flax, mreserve/modeling.py)                   here
    -------------------------------------------------------------  ---------------------------------------------------
    get_rotary_coordinates(_2d), multimodal_rotary_coords  :21-78  same names, numpy (host: they depend on shapes only)
    construct_rotary_sinusoids / apply_rotary             :81-144  numpy restatements + `rotary_scale_table` (what the
                                                                   QKV GEMM epilogue multiplies by: the reference's
                                                                   rotary is a per-position diagonal scaling)
    TransformerEncoder / VisionTransformer / AudioTransformer /
    SpanTransformer / TokenEmbedder                      :283-538  callables with the same arguments and returned dicts
                                                                   ('cls', 'seq', 'seq_attnpool')
    one_hot_pool, unit_normalize                         :541-578  device ops (segment sum / normalise kernels)
    MerlotReserve (.from_config, .apply, sub-encoders,
      prepare_multimodal_inputs, embed_* methods)        :581-931  same names, arguments, shapes
    PretrainedMerlotReserve (from_pretrained, __getattr__
      method cache, get_label_space)                     :933-1032 same; checkpoints are read from `cache_dir`
                                                                   (no network), `from_random` gives random weights

Arrays in: numpy or torch (any device); arrays out: torch tensors on the GPU in the model dtype -- fp32 unless
config['model']['use_bfloat16'] (M:594), like the reference off-TPU (M:999-1000).  All arithmetic runs in
libmreserve_hip.so (fp32 path: csrc/f32path.hip; bf16 path: the training kernels); there is no CPU fallback.
Integer logic (token / pointer / mask handling) is computed on the host in numpy, bit-exactly as the reference.
"""
import copy
import math
import os

import numpy as np
import torch

from . import ops
from .params import VOCAB, _get, param_specs
from .planner import AUDIOSPAN, PADDING, rot_scale_table, rotary_coords_1d, rotary_coords_2d

MASK, MASKAUDIO, LTOVPOOL = 3, 4, 6
BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


# ------------------------------------------------------------------------------------------------ coordinates (host)
def get_rotary_coordinates(seq_len, dtype=np.float32, center_origin=True):
    """M:21-35"""
    return rotary_coords_1d(seq_len, center_origin).astype(dtype)


def get_rotary_coordinates_2d(h, w, dtype=np.float32):
    """M:38-50: [h*w, 2], h first"""
    return rotary_coords_2d(h, w).astype(dtype)


def multimodal_rotary_coords(h=None, w=None, segment_idx=None, token_idx=None, dtype=np.float32, max_segment=16.0,
                             max_token=1024):
    """M:53-78: [B, L, 4] = (h, w, segment_idx / 16, token_idx / 1024), zeros where not given."""
    given = [np.asarray(x) for x in (h, w, segment_idx, token_idx) if x is not None]
    B, L = given[0].shape
    assert all(x.shape == (B, L) for x in given)
    z = np.zeros([B, L], dtype=np.float64)
    h_vec = z if h is None else np.asarray(h, dtype=np.float64)
    w_vec = z if w is None else np.asarray(w, dtype=np.float64)
    s_vec = z if segment_idx is None else np.asarray(segment_idx, dtype=np.float64) / max_segment
    t_vec = z if token_idx is None else np.asarray(token_idx, dtype=np.float64) / max_token
    return np.stack([h_vec, w_vec, s_vec, t_vec], -1).astype(dtype)


def construct_rotary_sinusoids(coords, rotary_hsize=32, max_freq=10.0):
    """M:81-113: coords [*b, L, nd] -> [*b, 2, L, rotary_hsize] = stack(cos, sin), each value repeated twice."""
    coords = np.asarray(coords, dtype=np.float64)
    *bd, L, nd = coords.shape
    assert rotary_hsize % (nd * 2) == 0
    de = rotary_hsize // (nd * 2)
    freqs = np.logspace(0.0, math.log2(max_freq / 2.0), de, base=2)
    rad = (coords[..., None] * freqs * np.pi).reshape(*bd, L, nd * de)
    return np.repeat(np.stack([np.cos(rad), np.sin(rad)], -3), 2, axis=-1).astype(np.float32)


def apply_rotary(query_key, sinusoids):
    """M:116-144 on numpy arrays (host utility; on the GPU this is the `rot_tab` multiply of the QKV GEMM epilogue).
    query_key [*b, L, nh, 64]; sinusoids [*b, 2, L, 32].  Note the reference's names: `sin = sinusoids[..., 0]` is the
    cosine table and vice versa, and each even/odd pair is rotated with itself."""
    qk = np.asarray(query_key, dtype=np.float64)
    sn = np.asarray(sinusoids, dtype=np.float64)
    rh = sn.shape[-1]
    qk_rope, qk_pass = qk[..., :rh], qk[..., rh:]
    sin = sn[..., 0, :, None, :]
    cos = sn[..., 1, :, None, :]
    sign = np.tile(np.array([-1.0, 1.0]), rh // 2)
    out = qk_rope * cos + (qk_rope * sign) * sin
    return np.concatenate([out, qk_pass], -1).astype(np.float32)


rotary_scale_table = rot_scale_table     # [*, nd] coords -> [*, 32] multipliers: even dims sin - cos, odd dims sin + cos


def _np(x, dtype=None):
    if isinstance(x, torch.Tensor):
        x = x.detach().cpu().numpy()
    x = np.asarray(x)
    return x if dtype is None else x.astype(dtype)


def _mask_to_code(mask):
    """A dense [n, L, L] boolean attention mask -> one int32 code per position such that
    mask[b, i, j] == (code[b, i] == code[b, j] >= 0).  Every mask the reference builds has this form (validity x
    validity, optionally ANDed with equality of a per-position source id: M:343-345, 743-756); anything else raises
    NotImplementedError here and TransformerEncoder falls back to the dense-mask kernel (ops.attention_fwd_dense_mask)."""
    m = _np(mask).astype(bool)
    n, L, _ = m.shape
    valid = m[:, np.arange(L), np.arange(L)]
    first = np.argmax(m, axis=-1)                              # lowest j with mask[i, j]
    code = np.where(valid, first, -1).astype(np.int32)
    rebuilt = (code[:, :, None] == code[:, None, :]) & (code[:, :, None] >= 0)
    if not np.array_equal(rebuilt, m):
        raise NotImplementedError('attention_mask is not of the block form valid[i] & valid[j] & (src[i] == src[j])')
    return code


# ------------------------------------------------------------------------------------------------ bound parameters
class _Weights:
    """The Flax-named parameter tree as device tensors in the model dtype, each Dense kernel viewed
    as row-major [in, out]."""

    def __init__(self, model_config, tree, device, dtype):
        self.device, self.dtype = torch.device(device), dtype
        self.w = {}
        for name, fshape, vshape, _kind, _fan in param_specs({'model': model_config}):
            leaf = _get(tree, name)
            leaf = leaf if isinstance(leaf, torch.Tensor) else torch.from_numpy(np.asarray(leaf))
            if tuple(leaf.shape) != tuple(fshape):
                raise ValueError(f'parameter {name}: shape {tuple(leaf.shape)}, expected {tuple(fshape)}')
            self.w[name] = leaf.reshape(vshape).to(device=self.device, dtype=dtype).contiguous()
        # optional leaves: learned position embeddings `pe` [L, H] of an encoder that was initialised WITHOUT rotary coordinates (M:335-341; no
        # released MERLOT Reserve model has them, param_specs does not list them)

        def walk(node, path):
            for k, v in (node.items() if isinstance(node, dict) else []):
                if k == 'pe' and not isinstance(v, dict):
                    leaf = v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))
                    self.w['/'.join(path + ['pe'])] = leaf.to(device=self.device, dtype=dtype).contiguous()
                else:
                    walk(v, path + [k])
        walk(tree, [])
        self.zero_scale = torch.zeros(1, device=self.device, dtype=dtype)       # log-temperature 0 -> factor 1


class _Module:
    def __init__(self, owner, prefix):
        self.owner, self.prefix = owner, prefix

    @property
    def W(self):
        return self.owner._weights().w

    @property
    def dtype(self):
        return self.owner.dtype

    @property
    def device(self):
        return self.owner.device

    def _in(self, x):
        x = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))
        return x.to(device=self.device, dtype=self.dtype)

    def _empty(self, *shape):
        return torch.empty(*shape, device=self.device, dtype=self.dtype)


class TransformerEncoder(_Module):
    """M:283-376"""

    def __init__(self, owner, prefix, hidden_size, num_layers, add_cls_token=False):
        super().__init__(owner, prefix)
        self.hidden_size, self.num_layers, self.add_cls_token = hidden_size, num_layers, add_cls_token

    def __call__(self, x, *, rotary_coords=None, attention_mask=None, is_valid=None, attention_code=None):
        x = self._in(x)
        *batch_dims, L, H = x.shape
        assert H == self.hidden_size
        nseq = int(np.prod(batch_dims)) if batch_dims else 1
        if self.add_cls_token and (attention_mask is not None or attention_code is not None):
            raise ValueError('Attention mask must not be provided if adding CLS token')
        if is_valid is not None and attention_mask is not None:
            raise ValueError('Provide only one of `is_valid` and `attention_mask` as we can use is_valid to construct attention mask')
        S = L + (1 if self.add_cls_token else 0)
        if self.add_cls_token:
            xin = self._empty(nseq * S, H)
            xin.view(nseq, S, H)[:, 1:].copy_(x.reshape(nseq, L, H))
        else:
            xin = x.reshape(nseq * L, H).contiguous()
        code = None
        if attention_code is not None:
            code = _np(attention_code, np.int32).reshape(nseq, L)
        elif is_valid is not None:
            code = np.where(_np(is_valid).astype(bool).reshape(nseq, L), 0, -1).astype(np.int32)
        dense = None
        elif_mask = attention_code is None and is_valid is None and attention_mask is not None
        if elif_mask:
            m = _np(attention_mask).astype(bool).reshape(nseq, L, L)
            try:
                code = _mask_to_code(m)             # the block form every mask of the model has: one code per position, the fast kernels
            except NotImplementedError:
                dense = torch.from_numpy(np.ascontiguousarray(m.astype(np.uint8))).to(self.device)      # anything else: mr_attention_fwd_dense_mask
        xf = self._run(xin, nseq, S, rotary_coords, code, dense_mask=dense)
        info = {}
        if self.add_cls_token:
            info['cls'] = self._cls_proj(xf, nseq, S).reshape(*batch_dims, H)
            info['seq'] = xf.view(nseq, S, H)[:, 1:].reshape(*batch_dims, L, H)
        else:
            info['seq'] = xf.view(*batch_dims, L, H)
        return info

    def _cls_proj(self, xf, nseq, S):
        H, W = self.hidden_size, self.W
        out = self._empty(nseq, H)
        ops.gemm(xf.view(nseq, S * H)[:, :H], W[f'{self.prefix}/cls_proj/kernel'], out, bias=W[f'{self.prefix}/cls_proj/bias'])
        return out

    def _run(self, xin, nseq, S, rotary_coords, code, dense_mask=None):
        """xin [nseq*S, H] with the CLS rows (if any) still to be filled; rotary_coords / code WITHOUT the CLS position.
        Returns final_ln(x) [nseq*S, H]."""
        W, H, p = self.W, self.hidden_size, self.prefix
        nh, M, dev = H // 64, nseq * S, self.device
        if self.add_cls_token:
            ops.fill_rows(W[f'{p}/cls'], xin, nseq, S, 0)
            if code is not None:
                code = np.concatenate([np.zeros((nseq, 1), np.int32), code], 1)  # CLS is always valid (M:321-322)
        if rotary_coords is None:
            # learned position embeddings instead of the "rotary" scales (M:335-341): x += pe[S, H], no scaling of q / k
            pe = W.get(f'{p}/pe')
            if pe is None:
                raise KeyError(f"rotary_coords is None and the parameters hold no '{p}/pe' (learned position embeddings, mreserve/modeling.py:335-341): "
                               'this encoder was initialised with rotary coordinates')
            assert tuple(pe.shape) == (S, H), f'pe has shape {tuple(pe.shape)}, the sequence (with CLS, if any) needs {(S, H)}'
            ops.add_rows_periodic(xin, pe)
            rot = None
        else:
            rc = _np(rotary_coords, np.float64)
            if self.add_cls_token:
                rc = np.concatenate([np.zeros_like(rc[..., :1, :]), rc], -2)         # CLS coordinates are 0 (M:324-326)
            assert rc.shape[-2] == S, f'rotary_coords cover {rc.shape[-2]} positions, sequence has {S}'
            rc = rc.reshape(-1, rc.shape[-1])
            assert rc.shape[0] in (S, M), 'rotary_coords batch dims must be absent or equal to the batch dims of x'
            rot = torch.from_numpy(rot_scale_table(rc)).to(dev)
        code_t = None if code is None else torch.from_numpy(np.ascontiguousarray(code.reshape(-1))).to(dev)
        e = self._empty
        xa, xb, ln, qkv, att, xmid, hid = e(M, H), e(M, H), e(M, H), e(M, 3 * H), e(M, H), e(M, H), e(M, 4 * H)
        lse = torch.empty(nseq, nh, S, device=dev, dtype=F32) if self.dtype == BF16 else None
        ops.layernorm_fwd(xin, W[f'{p}/pre_ln/scale'], W[f'{p}/pre_ln/bias'], xa)
        for l in range(self.num_layers):
            q = f'{p}/layer_{l:02d}'
            ops.layernorm_fwd(xa, W[f'{q}/pre_attn_ln/scale'], W[f'{q}/pre_attn_ln/bias'], ln)
            ops.gemm(ln, W[f'{q}/attention_layer/qkv/kernel'], qkv, bias=W[f'{q}/attention_layer/qkv/bias'], rot_tab=rot, rot_cols=2 * H if rot is not None else 0)
            if dense_mask is not None:
                ops.attention_fwd_dense_mask(qkv, dense_mask, att, nseq, S, nh)
            else:
                ops.attention_fwd(qkv, code_t, att, lse, nseq, S, nh)
            ops.gemm(att, W[f'{q}/attention_layer/attn_proj/kernel'], xmid, residual=xa)
            ops.layernorm_fwd(xmid, W[f'{q}/pre_mlp_ln/scale'], W[f'{q}/pre_mlp_ln/bias'], ln)
            ops.gemm(ln, W[f'{q}/mlp_layer/intermediate/kernel'], hid, bias=W[f'{q}/mlp_layer/intermediate/bias'], act=ops.ACT_GELU)
            ops.gemm(hid, W[f'{q}/mlp_layer/out/kernel'], xb, residual=xmid)
            xa, xb = xb, xa
        ops.layernorm_fwd(xa, W[f'{p}/final_ln/scale'], W[f'{p}/final_ln/bias'], xb)
        return xb


class _PooledEncoder(_Module):
    """Shared tail of VisionTransformer / AudioTransformer: attention pooling with a mean query (M:412-427, 464-472)."""

    def _attnpool(self, xf, key_rows, nseq, S):
        W, H, p = self.W, self.hidden_size, f'{self.prefix}/seq_attnpool'
        G = key_rows.shape[0]
        rows = torch.from_numpy(key_rows).to(self.device)
        e = self._empty
        qin, q, k, v, po, out = e(G, H), e(G, H), e(nseq * S, H), e(nseq * S, H), e(G, H), e(G, H)
        ops.rows_mean_fwd(xf, rows, qin)
        ops.gemm(qin, W[f'{p}/query/kernel'], q, bias=W[f'{p}/query/bias'])
        ops.gemm(xf, W[f'{p}/key/kernel'], k, bias=W[f'{p}/key/bias'])
        ops.gemm(xf, W[f'{p}/value/kernel'], v, bias=W[f'{p}/value/bias'])
        probs = torch.empty(G, H // 64, key_rows.shape[1], device=self.device, dtype=F32) if self.dtype == BF16 else None
        ops.poolattn_fwd(q, k, v, rows, po, probs, H // 64)
        ops.gemm(po, W[f'{p}/out/kernel'], out, bias=W[f'{p}/out/bias'])
        return out


class VisionTransformer(_PooledEncoder):
    """M:379-430"""

    def __init__(self, owner, prefix, num_layers, patch_size, pooling_ratio, output_grid_h, output_grid_w, hidden_size):
        super().__init__(owner, prefix)
        self.num_layers, self.patch_size, self.pooling_ratio, self.hidden_size = num_layers, patch_size, pooling_ratio, hidden_size
        self.output_grid_h, self.output_grid_w = output_grid_h, output_grid_w
        self.transformer = TransformerEncoder(owner, f'{prefix}/transformer', hidden_size, num_layers, add_cls_token=True)

    def __call__(self, x):
        x = self._in(x)
        *batch_dims, hw, pp3 = x.shape
        assert hw == self.output_grid_h * self.output_grid_w
        assert pp3 == (self.patch_size ** 2) * 3
        W, H, pr, gw = self.W, self.hidden_size, self.pooling_ratio, self.output_grid_w
        assert self.output_grid_h % pr == 0 and gw % pr == 0
        N = int(np.prod(batch_dims)) if batch_dims else 1
        S = hw + 1
        xin = self._empty(N * S, H)
        ops.gemm(x.reshape(N * hw, pp3), W[f'{self.prefix}/embedding/kernel'], xin, bias=W[f'{self.prefix}/embedding/bias'],
                 row_map=(hw, S, 1))
        coords = rotary_coords_2d(self.output_grid_h, gw)
        xf = self.transformer._run(xin, N, S, coords, None)
        h2, w2 = self.output_grid_h // pr, gw // pr
        n, i2, j2, di, dj = np.meshgrid(np.arange(N), np.arange(h2), np.arange(w2), np.arange(pr), np.arange(pr), indexing='ij')
        key_rows = (n * S + 1 + (i2 * pr + di) * gw + j2 * pr + dj).reshape(N * h2 * w2, pr * pr).astype(np.int32)
        return {'cls': self.transformer._cls_proj(xf, N, S).reshape(*batch_dims, H),
                'seq': xf.view(N, S, H)[:, 1:].reshape(*batch_dims, hw, H),
                'seq_attnpool': self._attnpool(xf, key_rows, N, S).reshape(*batch_dims, h2 * w2, H)}


class AudioTransformer(_PooledEncoder):
    """M:433-476: the stride-`patch_size` Conv over time is a Dense over `patch_size` consecutive hops."""

    def __init__(self, owner, prefix, num_layers, patch_size, pooling_ratio, hidden_size):
        super().__init__(owner, prefix)
        self.num_layers, self.patch_size, self.pooling_ratio, self.hidden_size = num_layers, patch_size, pooling_ratio, hidden_size
        self.transformer = TransformerEncoder(owner, f'{prefix}/transformer', hidden_size, num_layers, add_cls_token=True)

    def __call__(self, x):
        x = self._in(x)
        *batch_dims, raw_len, nm = x.shape
        assert nm == 65
        assert raw_len % self.patch_size == 0
        W, H = self.W, self.hidden_size
        seq_len = raw_len // self.patch_size
        assert seq_len % self.pooling_ratio == 0
        N = int(np.prod(batch_dims)) if batch_dims else 1
        S, K = seq_len + 1, self.patch_size * 65
        a = x.reshape(N * seq_len, K)
        if self.dtype == BF16:                           # 16-byte rows for the bf16 kernels: pad 130 -> 136 columns
            Kp = (K + 7) // 8 * 8
            a_pad = self._empty(N * seq_len, Kp)
            ops.pad_cols(a.contiguous(), a_pad)
            a = a_pad[:, :K]
        xin = self._empty(N * S, H)
        ops.gemm(a, W[f'{self.prefix}/embedding/kernel'], xin, bias=W[f'{self.prefix}/embedding/bias'], row_map=(seq_len, S, 1))
        coords = rotary_coords_1d(seq_len, True)[:, None] / seq_len
        xf = self.transformer._run(xin, N, S, coords, None)
        l2 = seq_len // self.pooling_ratio
        n, tt, r = np.meshgrid(np.arange(N), np.arange(l2), np.arange(self.pooling_ratio), indexing='ij')
        key_rows = (n * S + 1 + tt * self.pooling_ratio + r).reshape(N * l2, self.pooling_ratio).astype(np.int32)
        return {'cls': self.transformer._cls_proj(xf, N, S).reshape(*batch_dims, H),
                'seq': xf.view(N, S, H)[:, 1:].reshape(*batch_dims, seq_len, H),
                'seq_attnpool': self._attnpool(xf, key_rows, N, S).reshape(*batch_dims, l2, H)}


class SpanTransformer(_Module):
    """M:479-504"""

    def __init__(self, owner, prefix, num_layers, hidden_size, max_len=16):
        super().__init__(owner, prefix)
        self.num_layers, self.hidden_size, self.max_len = num_layers, hidden_size, max_len
        self.transformer = TransformerEncoder(owner, f'{prefix}/transformer', hidden_size, num_layers, add_cls_token=True)

    def __call__(self, x, x_isvalid):
        seq_len = x.shape[-2]
        assert seq_len < self.max_len
        coords = rotary_coords_1d(seq_len, False)[:, None] / self.max_len
        return self.transformer(x, is_valid=x_isvalid, rotary_coords=coords)['cls']


class TokenEmbedder(_Module):
    """M:507-538"""

    def __init__(self, owner, prefix, hidden_size, vocab_size=VOCAB):
        super().__init__(owner, prefix)
        self.hidden_size, self.vocab_size = hidden_size, vocab_size

    def __call__(self, token_dict):
        table = self.W[f'{self.prefix}/Embed_0/embedding']
        out = {}
        for k in sorted(token_dict.keys()):
            ids = _np(token_dict[k]).astype(np.int64)
            if ids.size and (ids.min() < 0 or ids.max() >= self.vocab_size):
                raise IndexError(f'token id out of range in {k!r}')
            flat = torch.from_numpy(ids.reshape(-1).astype(np.int32)).to(self.device)
            indptr = torch.arange(flat.numel() + 1, dtype=I32, device=self.device)
            dst = self._empty(max(flat.numel(), 1), self.hidden_size)
            if flat.numel():
                ops.segment_sum([table], indptr, flat, dst)
            out[k] = dst[:flat.numel()].reshape(*ids.shape, self.hidden_size)
        return out


def one_hot_pool(do_pool, idx, v, num_segments, real_bsize=None):
    """M:541-567 as a segment sum: out[b, s] = sum_l [do_pool[b,l] and idx[b,l] == s] v[b,l].  Returns the reference's
    dict {'x', 'idx_oh'} (idx_oh as a host numpy array: it only feeds integer bookkeeping)."""
    assert isinstance(v, torch.Tensor) and v.is_cuda, 'v must be a GPU tensor'
    B, L, H = v.shape
    do_pool, idx = _np(do_pool).astype(bool), _np(idx).astype(np.int64)
    assert do_pool.shape == (B, L) and idx.shape == (B, L)
    if real_bsize is not None:
        l2 = (L * B) // real_bsize
        do_pool, idx, v = do_pool.reshape(real_bsize, l2), idx.reshape(real_bsize, l2), v.reshape(real_bsize, l2, H)
        B, L = real_bsize, l2
    pointer = np.where(do_pool, idx, -1)
    ok = (pointer >= 0) & (pointer < num_segments)
    b_i, l_i = np.nonzero(ok)
    dst_rows = b_i * num_segments + pointer[b_i, l_i]
    order = np.argsort(dst_rows, kind='stable')
    counts = np.bincount(dst_rows, minlength=B * num_segments)
    indptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    src = (b_i * L + l_i)[order].astype(np.int32)
    out = torch.empty(B * num_segments, H, device=v.device, dtype=v.dtype)
    ops.segment_sum([v.reshape(B * L, H).contiguous()], torch.from_numpy(indptr).to(v.device),
                    torch.from_numpy(src if src.size else np.zeros(1, np.int32)).to(v.device), out)
    idx_oh = (pointer[..., None] == np.arange(num_segments)).astype(np.float32)
    return {'x': out.view(B, num_segments, H), 'idx_oh': idx_oh}


def unit_normalize(x):
    """M:570-578 (fp32 arithmetic, result in x's dtype)"""
    assert isinstance(x, torch.Tensor) and x.is_cuda
    H = x.shape[-1]
    x2 = x.reshape(-1, H).contiguous()
    y = torch.empty_like(x2)
    scale = None if x.dtype == F32 else torch.zeros(1, device=x.device, dtype=x.dtype)
    inv = None if x.dtype == F32 else torch.empty(x2.shape[0], device=x.device, dtype=F32)
    ops.unit_norm_scale_fwd(x2, scale, y, inv)
    return y.view(x.shape)


# ------------------------------------------------------------------------------------------------ the model
class MerlotReserve:
    """M:581-931.  `config` is config['model'] with config['data'] nested under 'data' (what from_config builds)."""

    def __init__(self, config, device='cuda:0'):
        self.config = config
        for k, v in config.items():                      # M:591-592
            setattr(self, k, v)
        self.dtype = BF16 if config.get('use_bfloat16', False) else F32          # M:594
        self.device = torch.device(device)
        self.output_grid_h, self.output_grid_w = config['output_grid']
        self.size_per_head = config.get('size_per_head', 64)
        assert self.size_per_head == 64, 'the attention kernels are specialised for 64-wide heads (every released config)'
        H = config['hidden_size']
        self.vision_encoder = VisionTransformer(self, 'vision_encoder', config['vit_num_layers'], config['vit_patch_size'],
                                                config['vit_pooling_ratio'], self.output_grid_h, self.output_grid_w, H)
        self.audio_encoder = AudioTransformer(self, 'audio_encoder', config['audio_num_layers'], config['audio_patch_size'],
                                              config['audio_seq_length'] // (config['audio_token_length'] * config['audio_patch_size']), H)
        self.token_encoder = TokenEmbedder(self, 'token_encoder', H)
        self.span_encoder = SpanTransformer(self, 'span_encoder', config['span_num_layers'], H)
        self.joint_transformer = TransformerEncoder(self, 'joint_transformer', H, config['joint_num_layers'], add_cls_token=False)
        self._bound = None
        self._bound_key = None

    @classmethod
    def from_config(cls, config, **kwargs):
        """M:584-588"""
        my_config = copy.deepcopy(config)
        my_config['model']['data'] = my_config['data']
        return cls(config=my_config['model'], **kwargs)

    # ---- parameters -----------------------------------------------------------------------------------------------
    def bind(self, params):
        """Upload a Flax-named parameter tree (numpy / torch leaves, flax shapes) in the model dtype."""
        if self._bound is None or self._bound_key is not params:
            self._bound = _Weights(self.config, params, self.device, self.dtype)
            self._bound_key = params
        return self

    def _weights(self):
        if self._bound is None:
            raise RuntimeError('no parameters bound: use model.apply({"params": params}, ..., method=...) or model.bind(params)')
        return self._bound

    def apply(self, variables, *args, method=None, **kwargs):
        """flax `Module.apply({'params': p}, *args, method=m)`: binds the parameters and calls the method."""
        self.bind(variables['params'] if 'params' in variables else variables)
        method = self.__call__ if method is None else method
        if isinstance(method, str):
            method = getattr(self, method)
        return method(*args, **kwargs)

    def init_from_dummy_batch(self, dummy_batch=None, seed=0):
        """M:636-649: the reference's initialisers (fp32 tree with the Flax names).  Shapes depend on the config only."""
        from .params import init_leaf, _set
        gen = torch.Generator().manual_seed(seed)
        tree = {}
        for name, fshape, _v, kind, fan in param_specs({'model': self.config}):
            _set(tree, name, init_leaf(kind, fshape, fan, self.config['hidden_size'], gen))
        return tree

    def joint_proj(self, x):
        """`head` Dense (M:631-632)"""
        W = self._weights().w
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        out = torch.empty_like(x2)
        ops.gemm(x2, W['head/kernel'], out, bias=W['head/bias'])
        return out.view(x.shape)

    @property
    def scale_params(self):
        return self._weights().w['contrastive_scales']

    # ---- multimodal input assembly --------------------------------------------------------------------------------
    def prepare_multimodal_inputs(self, tokens, token_segment_idx=None, token_embs=None, vision_input=None, audio_spans=None,
                                  audio_pointers=None, padding_len=None, video_src_idx=None):
        """M:651-758.  Returns {'x' [B, L', H], 'rotary_coords' [B, L', 4] (numpy), 'attention_mask' [B, L', L'] (bool
        tensor, the reference's output) and 'attention_code' [B, L'] (numpy int32: valid ? video_src : -1, the form the
        attention kernel consumes; TransformerEncoder prefers it over the dense mask)}.
        The embedding rows are assembled by ONE index-driven gather from [token table | audio spans | vision tokens]."""
        W = self._weights().w
        H, dev = self.hidden_size, self.device
        tokens = _np(tokens).astype(np.int64)
        B, L = tokens.shape
        tables = []
        if token_embs is None:
            if tokens.min() < 0 or tokens.max() >= VOCAB:
                raise IndexError('token id out of range')
            tables.append(W['token_encoder/Embed_0/embedding'])
            tok_code = tokens.copy()
        else:
            te = token_embs.to(device=dev, dtype=self.dtype) if isinstance(token_embs, torch.Tensor) else torch.from_numpy(np.asarray(token_embs)).to(device=dev, dtype=self.dtype)
            assert te.shape == (B, L, H)
            tables.append(te.reshape(B * L, H).contiguous())
            tok_code = np.arange(B * L, dtype=np.int64).reshape(B, L)
        n0 = tables[0].shape[0]
        if (audio_spans is not None) and (audio_pointers is not None):
            b_, num_audio_seqs, audio_token_length, h_ = audio_spans.shape
            assert b_ == B
            assert self.audio_token_length == audio_token_length
            audio_pointers = _np(audio_pointers).astype(np.int64)
            assert tokens.shape == audio_pointers.shape
            is_audio_src = tokens == AUDIOSPAN
            audio_ptr = np.maximum(audio_pointers, 0)
            audio_subpos = np.maximum(np.cumsum(is_audio_src.astype(np.int32), -1) - 1, 0) % audio_token_length     # M:691
            if audio_ptr.max() >= num_audio_seqs:
                raise IndexError('audio pointer beyond the provided audio spans')
            aud = audio_spans.to(device=dev, dtype=self.dtype) if isinstance(audio_spans, torch.Tensor) else torch.from_numpy(np.asarray(audio_spans)).to(device=dev, dtype=self.dtype)
            tables.append(aud.reshape(B * num_audio_seqs * audio_token_length, H).contiguous())
            a_code = n0 + (np.arange(B)[:, None] * num_audio_seqs + audio_ptr) * audio_token_length + audio_subpos
            tok_code = np.where(is_audio_src, a_code, tok_code)
        n01 = sum(t.shape[0] for t in tables)
        token_idx = np.tile(1.0 + np.arange(L, dtype=np.float64)[None], [B, 1])
        seg = None if token_segment_idx is None else _np(token_segment_idx).astype(np.int64)
        coords = multimodal_rotary_coords(segment_idx=seg, token_idx=token_idx, dtype=np.float64)
        codes = [tok_code]
        vis_seq_len, vis_segment_idx = 0, None
        if vision_input is not None:
            hpool, wpool = self.output_grid_h // self.vit_pooling_ratio, self.output_grid_w // self.vit_pooling_ratio
            img_coords_pool = rotary_coords_2d(hpool, wpool)
            vi = vision_input.to(device=dev, dtype=self.dtype) if isinstance(vision_input, torch.Tensor) else torch.from_numpy(np.asarray(vision_input)).to(device=dev, dtype=self.dtype)
            b_, vis_seq_len, h_ = vi.shape
            assert b_ == B
            num_pool_segments = vis_seq_len // (hpool * wpool)
            img_coords = np.tile(np.tile(img_coords_pool, [num_pool_segments, 1])[None], [B, 1, 1])
            vis_segment_idx = np.tile(np.arange(num_pool_segments, dtype=np.int64).repeat(hpool * wpool)[None], [B, 1])
            img_mm = multimodal_rotary_coords(segment_idx=vis_segment_idx, h=img_coords[..., 0], w=img_coords[..., 1], dtype=np.float64)
            assert img_mm.shape[-2] == vis_seq_len
            coords = np.concatenate([coords, img_mm], 1)
            tables.append(vi.reshape(B * vis_seq_len, H).contiguous())              # (the kernel takes the tables in order)
            codes.append(n01 + np.arange(B * vis_seq_len, dtype=np.int64).reshape(B, vis_seq_len))
        is_valid = tokens != PADDING
        if vis_seq_len > 0:
            is_valid = np.concatenate([is_valid, np.ones([B, vis_seq_len], dtype=bool)], 1)
        extra_len = 0
        if padding_len is not None:
            extra_len = padding_len - is_valid.shape[1]
            assert extra_len >= 0
            if extra_len > 0:
                is_valid = np.concatenate([is_valid, np.zeros([B, extra_len], dtype=bool)], 1)
                coords = np.concatenate([coords, np.zeros([B, extra_len, 4])], 1)
                codes.append(np.full([B, extra_len], -1, dtype=np.int64))            # -> empty gather list -> zero row
        code_all = np.concatenate(codes, 1)
        Lt = code_all.shape[1]
        flat = code_all.reshape(-1)
        has = flat >= 0
        indptr = np.concatenate([[0], np.cumsum(has)]).astype(np.int32)
        idx = flat[has].astype(np.int32) if has.any() else np.zeros(1, np.int32)
        x = torch.empty(B * Lt, H, device=dev, dtype=self.dtype)
        ops.segment_sum(tables, torch.from_numpy(indptr).to(dev), torch.from_numpy(idx).to(dev), x)
        attn_code = np.where(is_valid, 0, -1).astype(np.int32)
        if (video_src_idx is not None) and (token_segment_idx is not None):
            vsi = _np(video_src_idx).astype(np.int64)
            bi = np.arange(B)[:, None]
            video_src = [vsi[bi, seg]]
            if vis_segment_idx is not None:
                video_src.append(vsi[bi, vis_segment_idx])
            if extra_len > 0:
                video_src.append(np.full([B, extra_len], -1, dtype=np.int64))
            video_src = np.concatenate(video_src, -1)
            attn_code = np.where(is_valid, video_src, -1).astype(np.int32)
        code_t = torch.from_numpy(attn_code).to(dev)
        attn_mask = (code_t[:, :, None] == code_t[:, None, :]) & (code_t[:, :, None] >= 0)
        return {'x': x.view(B, Lt, H), 'rotary_coords': coords.astype(np.float32), 'attention_mask': attn_mask,
                'attention_code': attn_code}

    def __call__(self, batch):
        raise NotImplementedError()                      # M:760-761

    # ---- zero-shot / feature API (M:767-931) ----------------------------------------------------------------------
    def embed_text_spans_only(self, text_spans):
        text_spans = _np(text_spans).astype(np.int64)
        token_embs = self.token_encoder({'text_spans': text_spans})['text_spans']
        return unit_normalize(self.span_encoder(x=token_embs, x_isvalid=text_spans != PADDING))

    def embed_audio_only(self, audio_clips):
        *batch_dims, num_hops_per_audio, num_mels_plus_one = audio_clips.shape
        audio_enc = self.audio_encoder(_reshape(audio_clips, (-1, self.audio_seq_length, 65)))['cls']
        return unit_normalize(audio_enc).reshape(*batch_dims, self.hidden_size)

    def get_imgseq_only(self, imgs):
        *batch_dims, num_patch_per_img, pp3 = imgs.shape
        imgs_enc = self.vision_encoder(_reshape(imgs, (-1, num_patch_per_img, pp3)))['seq_attnpool']
        return imgs_enc.reshape(list(batch_dims) + [num_patch_per_img // 4, self.hidden_size])

    def get_audioseq_only(self, audio_clips):
        return self.audio_encoder(_reshape(audio_clips, (-1, self.audio_seq_length, 65)))['seq_attnpool']

    def _joint_embed(self, tokens, subseg_idxs, imgs_enc, audio_enc, audio_pointers):
        """tokens / subseg_idxs [B, L]; imgs_enc [B, V, H]; audio_enc [B, n, 6, H] or None -> [B, L, H]"""
        tokens, subseg_idxs = _np(tokens).astype(np.int64), _np(subseg_idxs).astype(np.int64)
        token_length = tokens.shape[1]
        mm = self.prepare_multimodal_inputs(tokens=tokens, token_segment_idx=subseg_idxs // 3,       # floor division (M:836)
                                            vision_input=imgs_enc, audio_pointers=audio_pointers, audio_spans=audio_enc)
        joint_enc = self.joint_transformer(mm['x'], rotary_coords=mm['rotary_coords'], attention_code=mm['attention_code'])['seq']
        return unit_normalize(self.joint_proj(joint_enc[:, :token_length].contiguous()))

    def embed_video(self, images, audio_clips, tokens, subseg_idxs):
        """M:806-843"""
        num_segments, num_patch_per_img, pp3 = images.shape
        assert pp3 == 768
        num_subsegments, num_hops_per_audio, num_mels_plus_one = audio_clips.shape
        assert num_subsegments == 3 * num_segments
        assert num_hops_per_audio == self.audio_seq_length
        assert num_mels_plus_one == 65
        token_length, = tokens.shape
        token_length_, = subseg_idxs.shape
        assert token_length_ == token_length
        return self.batch_embed_video(_expand(images), _expand(audio_clips), _expand(tokens), _expand(subseg_idxs))[0]

    def batch_embed_video(self, images, audio_clips, tokens, subseg_idxs):
        """M:845-846 (vmap of embed_video): here the batch is folded into the kernels' sequence dimension."""
        B, num_segments, num_patch_per_img, pp3 = images.shape
        imgs_enc = self.vision_encoder(_reshape(images, (B * num_segments, num_patch_per_img, pp3)))['seq_attnpool']
        imgs_enc = imgs_enc.reshape(B, num_segments * num_patch_per_img // 4, self.hidden_size)
        audio_enc = self.audio_encoder(_reshape(audio_clips, (-1, self.audio_seq_length, 65)))['seq_attnpool']
        audio_enc = audio_enc.reshape(B, -1, audio_enc.shape[-2], self.hidden_size)
        return self._joint_embed(tokens, subseg_idxs, imgs_enc, audio_enc, _np(subseg_idxs))

    def embed_singleimg_with_multiimg_prompt(self, images_prompt, images, tokens, subseg_idxs):
        """M:848-878"""
        ns0 = images_prompt.shape[0]
        ns1, num_patch_per_img, pp3 = images.shape
        assert (ns0 + ns1) <= 8
        imgs_enc = self.vision_encoder(images)['seq_attnpool']
        prompt = images_prompt.to(device=self.device, dtype=self.dtype) if isinstance(images_prompt, torch.Tensor) else torch.from_numpy(np.asarray(images_prompt)).to(device=self.device, dtype=self.dtype)
        imgs_enc = torch.cat([prompt, imgs_enc], 0).reshape(1, (ns0 + ns1) * num_patch_per_img // 4, self.hidden_size)
        assert tokens.shape == subseg_idxs.shape and len(tokens.shape) == 1
        return self._joint_embed(_expand(tokens), _expand(subseg_idxs), imgs_enc, None, None)[0]

    def embed_preencoded_noaudio(self, images_enc, tokens, subseg_idxs):
        """M:880-904"""
        ns, npp4, hidden_size = images_enc.shape
        assert tokens.shape == subseg_idxs.shape and len(tokens.shape) == 1
        return self._joint_embed(_expand(tokens), _expand(subseg_idxs), _reshape(images_enc, (1, ns * npp4, hidden_size)), None, None)[0]

    def embed_preencoded_audio(self, images_enc, audio_enc, tokens, subseg_idxs, audio_pointers):
        """M:906-931"""
        assert tokens.shape == subseg_idxs.shape and len(tokens.shape) == 1
        return self._joint_embed(_expand(tokens), _expand(subseg_idxs), _reshape(images_enc, (1, -1, self.hidden_size)),
                                 _expand(audio_enc), _expand(_np(audio_pointers)))[0]


def _expand(x):
    return x[None] if isinstance(x, (torch.Tensor, np.ndarray)) else np.asarray(x)[None]


def _reshape(x, shape):
    return x.reshape(shape) if isinstance(x, (torch.Tensor, np.ndarray)) else np.asarray(x).reshape(shape)


# ------------------------------------------------------------------------------------------------ pretrained wrapper
_PARAM_FN = {('base', (12, 20)): 'base', ('large', (12, 20)): 'large', ('base', (18, 32)): 'base_resadapt',
             ('large', (18, 32)): 'large_resadapt', ('base', (24, 24)): 'base_resadapt', ('large', (24, 24)): 'large_resadapt'}


def get_encoder(path=None):
    """mreserve/lowercase_encoder.py: the byte-level BPE tokenizer (`tokenizers.Tokenizer`).  Its vocabulary file
    (lowercase_encoder.json, a data asset of the reference release) is not redistributed here: pass its path, set
    MRESERVE_TOKENIZER_JSON, or put it next to this module."""
    from tokenizers import Tokenizer
    cands = [path, os.environ.get('MRESERVE_TOKENIZER_JSON'), os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lowercase_encoder.json')]
    for c in cands:
        if c and os.path.exists(c):
            return Tokenizer.from_file(c)
    raise FileNotFoundError('tokenizer vocabulary not found: pass path=, or set MRESERVE_TOKENIZER_JSON to the '
                            "reference release's mreserve/lowercase_encoder.json")


class PretrainedMerlotReserve:
    """M:933-1032"""

    def __init__(self, encoder, params, model, _method_cache=None):
        self.__dict__.update(encoder=encoder, params=params, model=model, _method_cache=_method_cache)

    @staticmethod
    def _check(model_name, image_grid_size):
        if model_name not in ('base', 'large'):
            raise ValueError("Must provide a model that is `base' or `large'")                    # M:954-955
        if tuple(image_grid_size) not in [(18, 32), (12, 20), (24, 24)]:
            raise ValueError('Invalid grid size {}'.format(image_grid_size))                      # M:957-958
        return _PARAM_FN[model_name, tuple(image_grid_size)]

    @staticmethod
    def _config(model_name, image_grid_size, use_bfloat16):
        from .config import load_config
        config = load_config(model_name)
        config['model']['output_grid'] = list(image_grid_size)
        config['model']['use_bfloat16'] = bool(use_bfloat16)       # M:999-1000: bf16 only on TPU; fp32 elsewhere
        return config

    @classmethod
    def from_pretrained(cls, model_name, image_grid_size=(18, 24), cache_dir=None, device='cuda:0', use_bfloat16=False,
                        encoder=None):
        """M:940-1003.  The checkpoint `<cache_dir>/<base|large|base_resadapt|large_resadapt>` (the released
        gs://merlotreserve/ckpts file, flax msgpack) must already be there: this build has no network path."""
        from .checkpoint import load_checkpoint
        param_fn = cls._check(model_name, image_grid_size)
        if cache_dir is None:
            cache_dir = os.path.join(os.path.expanduser('~'), '.cache', 'merlotreserve')
        cache_path = os.path.join(cache_dir, param_fn)
        if not os.path.exists(cache_path):
            raise FileNotFoundError(f'{cache_path} not found: download gs://merlotreserve/ckpts/{param_fn} there '
                                    '(https://storage.googleapis.com/merlotreserve/ckpts/...)')
        params = load_checkpoint(cache_path)['params']
        model = MerlotReserve.from_config(cls._config(model_name, image_grid_size, use_bfloat16), device=device)
        return cls(model=model, params=params, encoder=encoder if encoder is not None else get_encoder())

    @classmethod
    def from_random(cls, model_name, image_grid_size=(18, 32), seed=0, device='cuda:0', use_bfloat16=False, encoder=None):
        """Random weights of the named architecture (the reference's initialisers): plumbing and benchmarks."""
        cls._check(model_name, image_grid_size)
        model = MerlotReserve.from_config(cls._config(model_name, image_grid_size, use_bfloat16), device=device)
        return cls(model=model, params=model.init_from_dummy_batch(seed=seed), encoder=encoder)

    def __getattr__(self, name):
        """M:1005-1022: forwards to the model's method with the parameters bound (cached per name)."""
        if self._method_cache is None:
            self.__dict__['_method_cache'] = {}
        if name in self._method_cache:
            return self._method_cache[name]
        elif name in dir(self.model):
            fn = lambda *args, **kwargs: self.model.apply({'params': self.params}, *args, **kwargs, method=getattr(self.model, name))
            self._method_cache[name] = fn
            return fn
        else:
            raise ValueError(f'Unknown attribute {name}')

    def get_label_space(self, options):
        """M:1024-1032"""
        self.encoder.enable_padding(pad_token='<|PAD|>', length=15)
        answer_table_enc = np.array([x.ids[:15] for x in self.encoder.encode_batch(options)])
        self.encoder.no_padding()
        return self.embed_text_spans_only(answer_table_enc)
