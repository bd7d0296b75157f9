"""Torch (CPU) restatement of the reference's pretraining step -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py): the reference has no tests or golden
vectors and JAX/Flax cannot be imported here.  This file restates, function by
function, what the reference computes in its fp32 mode; dtype follows the params
(float32 = the parity oracle, float64 = the check on the oracle itself).  Autograd
of this restatement provides the gradient oracle; the same code timed on host
cores is bench.py's ``cpu_baseline`` (kind "port").

Every function cites the reference lines it follows (paths relative to the
upstream repo root: mreserve/modeling.py = M, pretrain/pretrain_model.py = P,
pretrain/optimization.py = O).  Third-party semantics (flax 0.3.4 Dense /
LayerNorm / dot_product_attention_weights / MultiHeadDotProductAttention / Conv /
Embed, optax 0.0.9 chain) are restated from their published behaviour.
"""
import math

import numpy as np
import torch

PADDING, START, END, MASK, MASKAUDIO, AUDIOSPAN, LTOVPOOL, RESETCTX = 0, 1, 2, 3, 4, 5, 6, 9


# ----------------------------------------------------------------------------- storage rounding (round 6)
# The bf16 training program keeps every tensor it writes to HBM in bf16 and accumulates in fp32 in between (the reference's bf16 mode:
# flax modules with dtype = bfloat16, P:323-324).  Inside `with bf16_storage():` this restatement rounds at exactly those points -- the
# output of a LayerNorm, of a Dense WITH its fused epilogue (bias + "rotary", bias + GELU, + residual: one rounding, after the epilogue,
# as merlot_reserve_amd/engine.py's launches store them), the softmax weights as the second product's operand, the attention output, pooled
# rows, the normalised embeddings -- so that what is left between it and the HIP program is summation order and 1-ulp rounding flips, an
# order of magnitude below the 2^-8 of the storage format.  Outside the context nothing changes (the exact fp32 / fp64 oracle).
_STORE = None


class bf16_storage:
    def __enter__(self):
        global _STORE
        self._prev, _STORE = _STORE, (lambda t: t.to(torch.bfloat16).to(t.dtype))
        return self

    def __exit__(self, *exc):
        global _STORE
        _STORE = self._prev


def _st(t):
    return t if _STORE is None else _STORE(t)


# ----------------------------------------------------------------------------- coordinates
def get_rotary_coordinates(seq_len, center_origin=True):
    """M:21-35. 1-D coordinates; the origin is skipped when centred."""
    if center_origin:
        sl0 = seq_len // 2
        nseq = np.arange(sl0, dtype=np.float64) - float(sl0)
        pseq = 1.0 + np.arange(seq_len - sl0, dtype=np.float64)
        return np.concatenate([nseq, pseq], 0)
    return 1.0 + np.arange(seq_len, dtype=np.float64)


def get_rotary_coordinates_2d(h, w):
    """M:38-50. [h*w, 2] (h first), scaled by 1/(max(h,w)+1)."""
    base_scale = 1.0 / (max(h, w) + 1.0)
    w_coords = base_scale * get_rotary_coordinates(w)
    h_coords = base_scale * get_rotary_coordinates(h)
    hh, ww = np.meshgrid(h_coords, w_coords, indexing='ij')
    return np.stack([hh, ww], -1).reshape(h * w, 2)


def multimodal_rotary_coords(B, L, h=None, w=None, segment_idx=None, token_idx=None,
                             max_segment=16.0, max_token=1024):
    """M:53-78. [B, L, 4] = (h, w, segment/16, token/1024); absent axes are zero."""
    z = np.zeros([B, L], dtype=np.float64)
    h_vec = z if h is None else np.asarray(h, dtype=np.float64)
    w_vec = z if w is None else np.asarray(w, dtype=np.float64)
    s_vec = z if segment_idx is None else np.asarray(segment_idx, dtype=np.float64) / max_segment
    t_vec = z if token_idx is None else np.asarray(token_idx, dtype=np.float64) / max_token
    return np.stack([h_vec, w_vec, s_vec, t_vec], -1)


def construct_rotary_sinusoids(coords, rotary_hsize=32, max_freq=10.0):
    """M:81-113. coords [..., S, nd] -> [..., 2 (cos, then sin), S, rotary_hsize]."""
    coords = np.asarray(coords, dtype=np.float64)
    *batch_dims, seq_length, num_dims = coords.shape
    assert rotary_hsize % (num_dims * 2) == 0
    dim_expansion = rotary_hsize // (num_dims * 2)
    freqs = np.logspace(0.0, math.log2(max_freq / 2.0), dim_expansion, base=2)
    radians = coords[..., None] * freqs * np.pi
    radians = radians.reshape(*batch_dims, seq_length, num_dims * dim_expansion)
    sinusoids = np.stack([np.cos(radians), np.sin(radians)], -3)
    return np.repeat(sinusoids, 2, axis=-1)


def apply_rotary(query_key, sinusoids):
    """M:116-144, literally -- including the swapped names at :133-134 (index 0 of the
    stack is cos but is bound to the name ``sin``) and the self-pairing at :139."""
    rotary_hsize = sinusoids.shape[-1]
    sin = sinusoids[..., 0, :, None, :]   # (this is cos(theta): reference quirk)
    cos = sinusoids[..., 1, :, None, :]   # (this is sin(theta))
    qk_rope = query_key[..., :rotary_hsize]
    qk_rotated_two = torch.stack([-qk_rope[..., ::2], qk_rope[..., 1::2]], -1).reshape(qk_rope.shape)
    qk_rope = qk_rope * cos + qk_rotated_two * sin
    return torch.cat([qk_rope, query_key[..., rotary_hsize:]], -1)


# ----------------------------------------------------------------------------- layers
def layer_norm(x, p, eps=1e-5):
    """flax 0.3.4 nn.LayerNorm: var = E[x^2] - E[x]^2; y = (x-mean)*rsqrt(var+eps)*scale + bias."""
    mean = x.mean(-1, keepdim=True)
    mean2 = (x * x).mean(-1, keepdim=True)
    var = mean2 - mean * mean
    mul = torch.rsqrt(var + eps) * p['scale']
    return _st((x - mean) * mul + p['bias'])


def dense(x, p):
    y = x @ p['kernel']
    if 'bias' in p:
        y = y + p['bias']
    return y


def my_gelu(x):
    """M:240-241."""
    return x * torch.sigmoid(1.702 * x)


def dot_product_attention_weights(query, key, bias=None):
    """flax 0.3.4: query / sqrt(depth) BEFORE the product; softmax over keys."""
    depth = query.shape[-1]
    query = query / math.sqrt(depth)
    w = torch.einsum('...qhd,...khd->...hqk', query, key)
    if bias is not None:
        w = w + bias
    return torch.softmax(w, dim=-1)


def attention_layer(p, x, sinusoids, attention_bias):
    """M:188-237. qkv heads: [0,nh) = Q, [nh,2nh) = K, [2nh,3nh) = V; attn_proj has no bias."""
    kq = p['qkv']['kernel']                                  # [H, 3nh, 64]
    nh = kq.shape[1] // 3
    qkv = torch.einsum('...h,hnd->...nd', x, kq) + p['qkv']['bias']
    query_key, value = qkv[..., :2 * nh, :], qkv[..., 2 * nh:, :]
    if sinusoids is not None:
        query_key = apply_rotary(query_key, sinusoids)
    query_key, value = _st(query_key), _st(value)            # (stored once, behind the bias + "rotary" epilogue)
    query, key = query_key[..., :nh, :], query_key[..., nh:, :]
    probs = _st(dot_product_attention_weights(query, key, attention_bias))
    o = _st(torch.einsum('...hqk,...khd->...qhd', probs, value))
    return torch.einsum('...nd,ndh->...h', o, p['attn_proj']['kernel'])       # (stored behind the residual add: transformer_layer)


def mlp_block(p, x):
    """M:244-256. out has no bias."""
    x1 = _st(my_gelu(dense(x, p['intermediate'])))
    return x1 @ p['out']['kernel']


def transformer_layer(p, x, sinusoids, attention_bias):
    """M:259-280."""
    x = _st(x + attention_layer(p['attention_layer'], layer_norm(x, p['pre_attn_ln']), sinusoids, attention_bias))
    x = _st(x + mlp_block(p['mlp_layer'], layer_norm(x, p['pre_mlp_ln'])))
    return x


def transformer_encoder(p, x, num_layers, rotary_coords=None, attention_mask=None, is_valid=None,
                        add_cls_token=False):
    """M:283-376. x [N, S, H]; rotary_coords [S, nd] or [N, S, nd] (numpy float64)."""
    N, S, H = x.shape
    if add_cls_token:
        if attention_mask is not None:
            raise ValueError("Attention mask must not be provided if adding CLS token")
        cls_tok = p['cls'][None, None].expand(N, 1, H)
        x = torch.cat([cls_tok.to(x.dtype), x], -2)
        if is_valid is not None:
            is_valid = torch.cat([torch.ones(N, 1, dtype=torch.bool), is_valid], -1)
        if rotary_coords is not None:
            rotary_coords = np.concatenate([np.zeros_like(rotary_coords[..., :1, :]), rotary_coords], -2)
    if rotary_coords is not None:
        sinusoids = torch.as_tensor(construct_rotary_sinusoids(rotary_coords)).to(x.dtype)
    else:                                  # M:335-341: learned position embeddings pe [S, H] (S includes the CLS position), no rotary
        sinusoids = None
        x = _st(x + p['pe'][None].to(x.dtype))

    if (is_valid is not None) and (attention_mask is None):
        attention_mask = is_valid[..., None] & is_valid[..., None, :]
    elif (is_valid is not None) and (attention_mask is not None):
        raise ValueError("Provide only one of `is_valid` and `attention_mask`")
    attention_bias = None
    if attention_mask is not None:
        am = attention_mask[..., None, :, :]
        attention_bias = torch.where(am, torch.zeros((), dtype=x.dtype), torch.full((), -1e10, dtype=x.dtype))

    x = layer_norm(x, p['pre_ln'])
    for i in range(num_layers):
        x = transformer_layer(p[f'layer_{i:02d}'], x, sinusoids, attention_bias)
    x_ln = layer_norm(x, p['final_ln'])
    info = {}
    if add_cls_token:
        info['cls'] = _st(dense(x_ln[..., 0, :], p['cls_proj']))
        info['seq'] = x_ln[..., 1:, :]
    else:
        info['seq'] = x_ln
    return info


def multihead_attnpool(p, inputs_q, inputs_kv):
    """flax 0.3.4 nn.MultiHeadDotProductAttention (biases on q/k/v/out)."""
    q = _st(torch.einsum('...h,hnd->...nd', inputs_q, p['query']['kernel']) + p['query']['bias'])
    k = _st(torch.einsum('...h,hnd->...nd', inputs_kv, p['key']['kernel']) + p['key']['bias'])
    v = _st(torch.einsum('...h,hnd->...nd', inputs_kv, p['value']['kernel']) + p['value']['bias'])
    w = dot_product_attention_weights(q, k)
    o = _st(torch.einsum('...hqk,...khd->...qhd', w, v))
    return _st(torch.einsum('...nd,ndh->...h', o, p['out']['kernel']) + p['out']['bias'])


def vision_transformer(p, x, num_layers, grid_h, grid_w, pooling_ratio=2):
    """M:379-430."""
    N, hw, pp3 = x.shape
    assert hw == grid_h * grid_w
    H = p['embedding']['kernel'].shape[-1]
    x = _st(dense(x, p['embedding']))
    coords = get_rotary_coordinates_2d(grid_h, grid_w)
    t_out = transformer_encoder(p['transformer'], x, num_layers, rotary_coords=coords, add_cls_token=True)
    h2, w2 = grid_h // pooling_ratio, grid_w // pooling_ratio
    b2 = N * h2
    seq = t_out['seq'].reshape(b2, pooling_ratio, w2, pooling_ratio, H).transpose(-4, -3)
    seq = seq.reshape(b2 * w2, pooling_ratio ** 2, H)
    pooled = multihead_attnpool(p['seq_attnpool'], _st(seq.mean(-2, keepdim=True)), seq)
    t_out['seq_attnpool'] = pooled.reshape(N, h2 * w2, H)
    return t_out


def audio_transformer(p, x, num_layers, pooling_ratio, patch_size=2):
    """M:433-476. Conv(kernel 2, stride 2, SAME) over time == linear on 2 consecutive hops."""
    N, raw_len, nm = x.shape
    assert nm == 65 and raw_len % patch_size == 0
    seq_len = raw_len // patch_size
    k = p['embedding']['kernel']                         # [2, 65, H]
    H = k.shape[-1]
    x = _st(x.reshape(N, seq_len, patch_size * nm) @ k.reshape(patch_size * nm, H) + p['embedding']['bias'])
    coords = get_rotary_coordinates(seq_len)[:, None] / seq_len
    t_out = transformer_encoder(p['transformer'], x, num_layers, rotary_coords=coords, add_cls_token=True)
    l2 = seq_len // pooling_ratio
    seq = t_out['seq'].reshape(-1, pooling_ratio, H)
    pooled = multihead_attnpool(p['seq_attnpool'], _st(seq.mean(-2, keepdim=True)), seq)
    t_out['seq_attnpool'] = pooled.reshape(N, l2, H)
    return t_out


def span_transformer(p, x, x_isvalid, num_layers, max_len=16):
    """M:479-504."""
    N, S, H = x.shape
    assert S < max_len
    coords = get_rotary_coordinates(S, center_origin=False)[:, None] / max_len
    return transformer_encoder(p['transformer'], x, num_layers, rotary_coords=coords, is_valid=x_isvalid,
                               add_cls_token=True)['cls']


def token_embedder(p, token_dict):
    """M:507-538: one gather for all streams."""
    emb = p['Embed_0']['embedding']
    return {k: emb[v] for k, v in token_dict.items()}


def one_hot_pool(do_pool, idx, v, num_segments, real_bsize=None):
    """M:541-567. idx == -1 (or do_pool False) contributes a zero row."""
    B, L, H = v.shape
    if real_bsize is not None:
        l2 = (L * B) // real_bsize
        do_pool = do_pool.reshape(real_bsize, l2)
        idx = idx.reshape(real_bsize, l2)
        v = v.reshape(real_bsize, l2, H)
    pointer = torch.where(do_pool, idx, torch.full_like(idx, -1))
    oh = (pointer[..., None] == torch.arange(num_segments)[None, None]).to(v.dtype)
    return {'x': torch.einsum('bls,blh->bsh', oh, v), 'idx_oh': oh}


def unit_normalize(x):
    """M:570-578."""
    return x / torch.sqrt((x * x).sum(-1, keepdim=True) + 1e-5)


# ----------------------------------------------------------------------------- model wiring
class Cfg:
    """The attributes MerlotReserve.setup derives from config['model'] (+ 'data')  (M:584-634)."""

    def __init__(self, config):
        m, d = config['model'], config['data']
        self.hidden_size = m['hidden_size']
        self.grid_h, self.grid_w = m['output_grid']
        self.vit_num_layers = m['vit_num_layers']
        self.vit_pooling_ratio = m['vit_pooling_ratio']
        self.audio_num_layers = m['audio_num_layers']
        self.audio_patch_size = m['audio_patch_size']
        self.audio_seq_length = m['audio_seq_length']
        self.audio_token_length = m['audio_token_length']
        self.audio_pooling_ratio = m['audio_seq_length'] // (m['audio_token_length'] * m['audio_patch_size'])
        self.joint_num_layers = m['joint_num_layers']
        self.span_num_layers = m['span_num_layers']
        self.text_span_length = m['text_span_length']
        self.data = d
        self.model = m


def prepare_multimodal_inputs(params, cfg, tokens, token_segment_idx=None, token_embs=None, vision_input=None,
                              audio_spans=None, audio_pointers=None, padding_len=None, video_src_idx=None):
    """M:651-758."""
    B, L = tokens.shape
    H = cfg.hidden_size
    if token_embs is None:
        token_embs = token_embedder(params['token_encoder'], {'k': tokens})['k']
    if (audio_spans is not None) and (audio_pointers is not None):
        assert audio_spans.shape[0] == B and audio_spans.shape[2] == cfg.audio_token_length
        is_audio_src = tokens == AUDIOSPAN
        audio_ptr = torch.clamp(audio_pointers, min=0)
        audio_subpos = torch.clamp(torch.cumsum(is_audio_src.to(torch.int64), -1) - 1, min=0) % cfg.audio_token_length
        audio_embs = audio_spans[torch.arange(B)[:, None], audio_ptr, audio_subpos]
        token_embs = torch.where(is_audio_src[..., None], audio_embs, token_embs)

    token_idx = np.tile(1.0 + np.arange(L, dtype=np.float64)[None], [B, 1])
    coords = multimodal_rotary_coords(
        B, L, segment_idx=token_segment_idx.numpy() if token_segment_idx is not None else None, token_idx=token_idx)

    vis_seq_len = 0
    vis_segment_idx = None
    if vision_input is not None:
        hpool, wpool = cfg.grid_h // cfg.vit_pooling_ratio, cfg.grid_w // cfg.vit_pooling_ratio
        img_coords_pool = get_rotary_coordinates_2d(hpool, wpool)
        vis_seq_len = vision_input.shape[1]
        num_pool_segments = vis_seq_len // (hpool * wpool)
        img_coords = np.tile(np.tile(img_coords_pool, [num_pool_segments, 1])[None], [B, 1, 1])
        vis_segment_idx = np.tile(np.arange(num_pool_segments).repeat(hpool * wpool)[None], [B, 1])
        img_mm = multimodal_rotary_coords(B, vis_seq_len, segment_idx=vis_segment_idx,
                                          h=img_coords[..., 0], w=img_coords[..., 1])
        coords = np.concatenate([coords, img_mm], 1)
        token_embs = torch.cat([token_embs, vision_input], 1)

    is_valid = tokens != PADDING
    if vis_seq_len > 0:
        is_valid = torch.cat([is_valid, torch.ones(B, vis_seq_len, dtype=torch.bool)], 1)
    extra_len = 0
    if padding_len is not None:
        extra_len = padding_len - is_valid.shape[1]
        assert extra_len >= 0
        if extra_len > 0:
            is_valid = torch.cat([is_valid, torch.zeros(B, extra_len, dtype=torch.bool)], 1)
            coords = np.concatenate([coords, np.zeros([B, extra_len, 4])], 1)
            token_embs = torch.cat([token_embs, torch.zeros(B, extra_len, H, dtype=token_embs.dtype)], 1)
    attn_mask = is_valid[:, None] & is_valid[:, :, None]
    if (video_src_idx is not None) and (token_segment_idx is not None):
        bi = torch.arange(B)[:, None]
        video_src = [video_src_idx[bi, token_segment_idx]]
        if vis_segment_idx is not None:
            video_src.append(video_src_idx[bi, torch.as_tensor(vis_segment_idx)])
        if extra_len > 0:
            video_src.append(torch.full([B, extra_len], -1, dtype=video_src_idx.dtype))
        video_src = torch.cat(video_src, -1)
        attn_mask = attn_mask & (video_src[:, None] == video_src[:, :, None])
    return {'x': token_embs, 'rotary_coords': coords, 'attention_mask': attn_mask}


def augment_video_src_idx(video_src_idx, split_from_here):
    """P:9-36 with the random draw injected: split_from_here in 1..L (L = no split)."""
    B, L = video_src_idx.shape
    if L == 1:
        return video_src_idx
    split_mask = split_from_here[:, None] <= torch.arange(L)[None]
    return torch.where(split_mask, video_src_idx + 4 * L, video_src_idx)


def pretrain_forward(params, config, batch, split_from_here, gumbel_z, return_debug=False):
    """P:38-259 (MerlotReservePretrainer.__call__) in fp32/fp64 mode.

    batch: the per-device dict of P:49-93 (integer tensors int64, images/audio float).
    split_from_here: two [2B] int tensors (the jax.random.choice draws of P:30, +1 applied);
    gumbel_z: [B, num_text_spans] float32 (the -log(-log(U)) of P:222-223).
    """
    cfg = Cfg(config)
    d = cfg.data
    H = cfg.hidden_size
    batch = dict(batch)
    num_segment_groups = d['num_segment_groups']
    nas = d['num_audio_subsegments']
    lang_seq_len, seq_len = d['lang_seq_len'], d['seq_len']

    batch_size = batch['images'].shape[0]
    nvpatch0 = cfg.grid_h * cfg.grid_w
    num_segments = batch['images'].shape[1] // nvpatch0
    nspg = num_segments // num_segment_groups

    imgs_enc = vision_transformer(params['vision_encoder'], batch['images'].reshape(batch_size * num_segments, nvpatch0, -1),
                                  cfg.vit_num_layers, cfg.grid_h, cfg.grid_w, cfg.vit_pooling_ratio)
    nvpatch1 = nvpatch0 // (cfg.vit_pooling_ratio ** 2)
    imgs_seq = imgs_enc['seq_attnpool'].reshape(batch_size, num_segment_groups, nspg * nvpatch1, H)
    if config['model'].get('no_vision', False):          # pretrain/pretrain_model.py:61-63
        imgs_seq = imgs_seq * 0.0
    vis_seq_length = imgs_seq.shape[-2]

    audio_enc = audio_transformer(params['audio_encoder'],
                                  batch['audio_clips'].reshape(batch_size * num_segments * nas, cfg.audio_seq_length, -1),
                                  cfg.audio_num_layers, cfg.audio_pooling_ratio, cfg.audio_patch_size)
    num_audio_spans = num_segments * nas
    audio_seq = audio_enc['seq_attnpool'].reshape(batch_size, num_audio_spans, cfg.audio_token_length, H)
    audio_cls = audio_enc['cls'].reshape(batch_size, num_audio_spans, H)

    for k1 in ['text2audio', 'audio2text']:
        for k2 in ['', '/audio_ptr', '/text_ptr']:
            batch[k1 + k2] = batch[k1 + k2].reshape(-1, lang_seq_len)
    for k in ['random_text', 'random_text/text_ptr', 'audio_text_matching', 'audio_text_matching/audio_ptr']:
        batch[k] = batch[k].reshape(-1, seq_len)
    batch['text_spans'] = batch['text_spans'].reshape(-1, cfg.text_span_length)

    txt_embs = token_embedder(params['token_encoder'],
                              {k: batch[k] for k in ['text2audio', 'audio2text', 'audio_text_matching', 'text_spans',
                                                     'random_text']})
    batch['video_src_index'] = batch['video_src_index'].reshape(-1, nspg)

    def floordiv(a, b):
        return torch.div(a, b, rounding_mode='floor')

    n_a2t, n_t2a = d['num_audio2text_seqs'], d['num_text2audio_seqs']
    mm = {}
    vsi3 = batch['video_src_index'].reshape(batch_size, num_segment_groups, nspg)
    tile_vis = lambda n: imgs_seq.repeat(1, n, 1, 1).reshape(-1, vis_seq_length, H)          # jnp.tile(imgs_seq, [1, n, 1, 1]) (P:104, 129)
    tile_vsi = lambda n: vsi3.repeat(1, n, 1).reshape(-1, nspg)                               # jnp.tile(..., [1, n, 1])         (P:109-110, 133-134)
    mm['audio2text'] = prepare_multimodal_inputs(
        params, cfg, tokens=batch['audio2text'],
        token_segment_idx=floordiv(batch['audio2text/audio_ptr'], nas) % nspg,
        token_embs=txt_embs['audio2text'],
        vision_input=tile_vis(n_a2t),
        audio_spans=audio_seq.repeat_interleave(num_segment_groups * n_a2t, dim=0),
        audio_pointers=batch['audio2text/audio_ptr'], padding_len=seq_len,
        video_src_idx=augment_video_src_idx(tile_vsi(n_a2t), split_from_here[0]))
    mm['audio_text_matching'] = prepare_multimodal_inputs(
        params, cfg, tokens=batch['audio_text_matching'],
        token_segment_idx=torch.cumsum((batch['audio_text_matching'] == LTOVPOOL).to(torch.int64), -1),
        token_embs=txt_embs['audio_text_matching'], audio_spans=audio_seq,
        audio_pointers=batch['audio_text_matching/audio_ptr'], padding_len=seq_len)
    mm['text2audio'] = prepare_multimodal_inputs(
        params, cfg, tokens=batch['text2audio'],
        token_segment_idx=floordiv(batch['text2audio/audio_ptr'], nas) % nspg,
        token_embs=txt_embs['text2audio'],
        vision_input=tile_vis(n_t2a),
        audio_pointers=batch['text2audio/audio_ptr'], padding_len=seq_len,
        video_src_idx=augment_video_src_idx(tile_vsi(n_t2a), split_from_here[1]))
    mm['random_text'] = prepare_multimodal_inputs(params, cfg, tokens=batch['random_text'], padding_len=seq_len,
                                                  token_embs=txt_embs['random_text'])
    # NOTE P:138 passes no token_embs for random_text, so the reference re-embeds the same tokens: same values.

    keys = sorted(mm.keys())
    x = torch.cat([mm[k]['x'] for k in keys], 0)
    coords = np.concatenate([mm[k]['rotary_coords'] for k in keys], 0)
    attnmask = torch.cat([mm[k]['attention_mask'] for k in keys], 0)
    real_bsizes = [mm[k]['x'].shape[0] for k in keys]

    if not config['model'].get('do_rotary', True):       # P:146-148: no coordinates -> the encoder's learned `pe` (M:335-341)
        coords = None
    joint_enc = transformer_encoder(params['joint_transformer'], x, cfg.joint_num_layers, rotary_coords=coords,
                                    attention_mask=attnmask)['seq']
    joint_enc = _st(dense(joint_enc, params['head']))
    mm_out = dict(zip(keys, torch.split(joint_enc, real_bsizes, dim=0)))
    mm_out['text2audio'] = mm_out['text2audio'][:, :lang_seq_len]
    mm_out['audio2text'] = mm_out['audio2text'][:, :lang_seq_len]

    is_pool = batch['audio_text_matching'] == LTOVPOOL
    v2a_idx = torch.cumsum(is_pool.to(torch.int64), -1) - 1
    a2v = _st(one_hot_pool(is_pool, v2a_idx, mm_out['audio_text_matching'], num_segments)['x']).reshape(
        batch_size * num_segments, H)

    t2a_pool = one_hot_pool(batch['text2audio'] == MASKAUDIO, batch['text2audio/audio_ptr'], mm_out['text2audio'],
                            num_segments * nas, real_bsize=batch_size)
    ntrg = int(num_audio_spans * d['mask_rate']) * n_t2a
    is_selected = t2a_pool['idx_oh'].sum(1)
    idx_sort = torch.argsort(-is_selected, dim=-1, stable=True)
    best_idxs = idx_sort[:, :ntrg].reshape(batch_size * ntrg)
    bi = torch.arange(batch_size).repeat_interleave(ntrg)
    t2a_sel = _st(t2a_pool['x'][bi, best_idxs])
    a2t_sel = audio_cls[bi, best_idxs]
    extra_idxs = idx_sort[:, ntrg:].reshape(batch_size * (num_audio_spans - ntrg))
    bi2 = torch.arange(batch_size).repeat_interleave(num_audio_spans - ntrg)
    a2t_extra = audio_cls[bi2, extra_idxs]

    num_text_spans = txt_embs['text_spans'].shape[0] // batch_size
    t2sp = {}
    for k in ['audio2text', 'text2audio', 'random_text']:
        t2sp[k] = one_hot_pool(batch[k] == MASK, batch[f'{k}/text_ptr'], mm_out[k], num_text_spans,
                               real_bsize=batch_size)
        t2sp[k]['count'] = t2sp[k].pop('idx_oh').sum(1)
    t2sp_sel = t2sp['text2audio']['x'] + t2sp['audio2text']['x'] + t2sp['random_text']['x']
    t2sp_ct = t2sp['text2audio']['count'] + t2sp['audio2text']['count'] + t2sp['random_text']['count']
    t2sp_src = torch.stack([torch.zeros_like(t2sp['text2audio']['count']), t2sp['text2audio']['count'],
                            t2sp['audio2text']['count'], t2sp['random_text']['count']], -1).argmax(-1) - 1

    is_valid = (batch['text_spans'] != PADDING).any(-1).reshape(batch_size, num_text_spans)
    is_valid = is_valid & (t2sp_ct > 0.0)
    # P:214-224 in float32 whatever the parity dtype, so the ordering is the reference's fp32-mode ordering
    logits_for_pred = is_valid.to(torch.float32) * np.float32(1e6) + np.float32(np.log(4)) * (
        t2sp['text2audio']['count'] + t2sp['audio2text']['count']).to(torch.float32)
    score = logits_for_pred + gumbel_z.to(torch.float32)
    n_inc = d['num_text_spans_to_include']
    assert n_inc <= num_text_spans
    # lax.top_k: descending, lower index first on ties
    best_sp = torch.sort(score.reshape(-1), descending=True, stable=True)[1][:n_inc * batch_size]

    t2sp_sel = _st(t2sp_sel.reshape(batch_size * num_text_spans, H)[best_sp])      # (one segment sum over the three streams' lists, stored once)
    t2sp_src = t2sp_src.reshape(batch_size * num_text_spans)[best_sp]
    sp2t_sel = span_transformer(params['span_encoder'], txt_embs['text_spans'][best_sp],
                                batch['text_spans'][best_sp] != PADDING, cfg.span_num_layers)

    log_scales = torch.clamp(params['contrastive_scales'], max=float(np.log(100.0)))
    outputs = {
        'imgs_to_audio': {'x': a2v, 'y': imgs_enc['cls'], 'log_scale': log_scales[0]},
        'text_to_audio': {'x': t2a_sel, 'y': a2t_sel, 'y_extra': a2t_extra, 'log_scale': log_scales[1]},
        'stuff_to_span': {'x': t2sp_sel, 'y': sp2t_sel, 'log_scale': log_scales[2], '_sources': t2sp_src},
    }
    for k in outputs:
        temp = torch.exp(outputs[k].pop('log_scale') / 2.0)
        for k2 in ['x', 'y', 'y_extra']:
            if k2 in outputs[k]:
                outputs[k][k2] = _st(unit_normalize(outputs[k][k2]) * temp)
    if return_debug:
        dbg = {'idx_sort': idx_sort, 'best_sp': best_sp, 'joint_x': x, 'joint_coords': coords,
               'joint_mask': attnmask, 'imgs_seq': imgs_seq, 'audio_seq': audio_seq, 'audio_cls': audio_cls,
               'imgs_cls': imgs_enc['cls'], 'joint_head': joint_enc}
        return outputs, dbg
    return outputs


def loss_fn_given_preds(preds_per_device, rank=0):
    """P:262-303 for device ``rank`` of a virtual pmap: all_gather(y) is the rank-major concat over
    ``preds_per_device`` (P:290).  Pass a 1-element list for the single-device case."""
    preds = dict(preds_per_device[rank])
    loss_info = {}
    if 'text_preds' in preds:                                # P:265-274: the mask-LM special case (no forward of the reference emits it)
        tp = preds.pop('text_preds')
        logits, labels = tp['logits'], tp['labels'].long()
        onehot = torch.nn.functional.one_hot(labels, logits.shape[1]).to(logits.dtype)
        logprobs = torch.log_softmax(logits, dim=-1)
        mask = (labels != 0).to(logits.dtype)
        loss_info['audio2text'] = -((logprobs * onehot).sum(-1) * mask).sum() / mask.sum()
    for c_type, c_dict in preds.items():
        numer = (c_dict['x'] * c_dict['y']).sum(-1)
        loss_info[c_type] = 0.0
        if '_sources' in c_dict:
            for k in ['text2audio', 'audio2text', 'random_text']:
                loss_info[f'_{c_type}_from_{k}'] = 0.0
        for k1, k2 in [('x', 'y'), ('y', 'x')]:
            x = c_dict[k1]
            ys = []
            for pd in preds_per_device:
                y = pd[c_type][k2]
                if f'{k2}_extra' in pd[c_type]:
                    y = torch.cat([y, pd[c_type][f'{k2}_extra']])
                ys.append(y)
            y_all = torch.cat(ys, 0)
            denom_lse = torch.logsumexp(x @ y_all.T, dim=-1)
            loss_info[c_type] = loss_info[c_type] + (denom_lse - numer).mean() / 2.0
            if '_sources' in c_dict:
                for i, type_i in enumerate(['text2audio', 'audio2text', 'random_text']):
                    does_match = (c_dict['_sources'] == i).to(x.dtype)
                    lm = ((denom_lse - numer) * does_match).sum() / (does_match.sum() + 1e-5)
                    loss_info[f'_{c_type}_from_{type_i}'] = loss_info[f'_{c_type}_from_{type_i}'] + lm / 2.0
    loss = sum(v for k, v in loss_info.items() if not k.startswith('_'))
    return loss, loss_info


# ----------------------------------------------------------------------------- optimizer
MISSING_PRECISION = 1 + (1 / 2 ** 9)          # O:36


def unsigned_bf16_decode(v_bf16):
    """O:38-41. v_bf16: torch.bfloat16 (sign bit = 'add 2^-9 of mantissa')."""
    v = v_bf16.to(torch.float32)
    v_abs = v.abs()
    v_abs = torch.where(v >= 0, v_abs, v_abs * np.float32(MISSING_PRECISION))
    return _cbrt(v_abs)


def _cbrt(x):
    return torch.as_tensor(np.cbrt(x.numpy()))


def unsigned_bf16_encode(v):
    """O:44-51. v float32 >= 0. Returns torch.bfloat16; strict '<' keeps '+' (so enc(0) = -0.0)."""
    v_pow = v * v * v
    v_bf = v_pow.to(torch.bfloat16)
    v_bf32 = v_bf.to(torch.float32)
    err0 = (v_bf32 - v_pow).abs()
    err1 = (v_bf32 * np.float32(MISSING_PRECISION) - v_pow).abs()
    return torch.where(err0 < err1, v_bf, -v_bf)


def lr_scale_linearwarmup_cosinedecay(step, num_warmup_steps, num_train_steps, final_lr_scale=0.1):
    """O:117-137 (float32 arithmetic on an int32 step, as jnp does)."""
    step = np.float32(step)
    warmup_scale = step / np.float32(num_warmup_steps)
    post = (step - np.float32(num_warmup_steps)) / np.float32(num_train_steps - num_warmup_steps + 1.0)
    post = np.minimum(post, np.float32(1.0))
    post = np.float32(1.0) - (np.float32(1.0) - np.cos(np.float32(np.pi) * post, dtype=np.float32)) / np.float32(2.0)
    post = np.float32(final_lr_scale) + np.float32(1.0 - final_lr_scale) * post
    return np.float32(warmup_scale if step < num_warmup_steps else post)


def lr_scale_linearwarmup_lineardecay(step, num_warmup_steps, num_train_steps):
    """O:140-155."""
    step = np.float32(step)
    warmup_scale = step / np.float32(num_warmup_steps)
    post = (step - np.float32(num_warmup_steps)) / np.float32(num_train_steps - num_warmup_steps + 1.0)
    post = np.float32(1.0) - np.minimum(post, np.float32(1.0))
    return np.float32(warmup_scale if step < num_warmup_steps else post)


def adam_bf16_apply(param, grad, mu_bf16, nu_bf16, count, opt_config):
    """O:54-114 + the optax chain of O:180-195 + apply_updates, for ONE leaf.

    param, grad float32; mu/nu torch.bfloat16; count = the schedule's step BEFORE increment
    (optax.scale_by_schedule evaluates at its own count, which starts at 0: first update is zero).
    Returns (new_param, new_mu, new_nu).
    """
    f32 = np.float32
    b1, b2 = opt_config.get('beta_1', 0.9), opt_config.get('beta_2', 0.98)
    eps = opt_config.get('eps', 1e-8)
    assert opt_config.get('use_bfloat16_adam', True)
    next_m = f32(1 - b1) * grad + f32(b1) * mu_bf16.to(torch.float32)
    next_v = f32(1 - b2) * grad * grad + f32(b2) * unsigned_bf16_decode(nu_bf16)
    new_mu = next_m.to(torch.bfloat16)
    new_nu = unsigned_bf16_encode(next_v)
    m_hat, v_hat = next_m, next_v
    if opt_config.get('do_bias_correction', False):
        c = count + 1
        m_hat = next_m / f32(1 - b1 ** c)
        v_hat = next_v / f32(1 - b2 ** c)
    u = m_hat / (torch.sqrt(v_hat) + f32(eps))
    if param.ndim > 1:                                   # O:182-184 weight-decay mask
        u = u + f32(opt_config['weight_decay_rate']) * param
    sched = lr_scale_linearwarmup_cosinedecay(count, opt_config['num_warmup_steps'], opt_config['num_train_steps'],
                                              opt_config.get('final_lr_scale', 0.02))
    u = u * sched
    u = u * f32(-opt_config['learning_rate'])
    return param + u, new_mu, new_nu


# ----------------------------------------------------------------------------- helpers for tests / bench
def tree_map(fn, tree):
    if isinstance(tree, dict):
        return {k: tree_map(fn, v) for k, v in tree.items()}
    return fn(tree)


def tree_leaves(tree, prefix=''):
    if isinstance(tree, dict):
        for k in sorted(tree):
            yield from tree_leaves(tree[k], f'{prefix}/{k}' if prefix else k)
    else:
        yield prefix, tree


def loss_and_grads(params, config, batch, split_from_here, gumbel_z):
    """value_and_grad of P:317-326 on one device (fp32/fp64 mode: no bf16 cast)."""
    params = tree_map(lambda t: t.detach().clone().requires_grad_(True), params)
    preds = pretrain_forward(params, config, batch, split_from_here, gumbel_z)
    loss, info = loss_fn_given_preds([preds])
    loss.backward()
    grads = tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), params)
    return loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in info.items()}, preds, grads


# ----------------------------------------------------------------------------- zero-shot API (M:767-931)
def embed_text_spans_only(params, config, text_spans):
    """M:767-774: text_spans [B, L] int64 -> [B, H]."""
    cfg = Cfg(config)
    token_embs = token_embedder(params['token_encoder'], {'text_spans': text_spans})['text_spans']
    return unit_normalize(span_transformer(params['span_encoder'], token_embs, text_spans != PADDING, cfg.span_num_layers))


def embed_audio_only(params, config, audio_clips):
    """M:776-785"""
    cfg = Cfg(config)
    enc = audio_transformer(params['audio_encoder'], audio_clips.reshape(-1, cfg.audio_seq_length, 65), cfg.audio_num_layers,
                            cfg.audio_pooling_ratio, cfg.audio_patch_size)['cls']
    return unit_normalize(enc).reshape(*audio_clips.shape[:-2], cfg.hidden_size)


def embed_video(params, config, images, audio_clips, tokens, subseg_idxs):
    """M:806-843: images [ns, P, 768], audio_clips [3 ns, 60, 65], tokens / subseg_idxs [L] int64 -> [L, H]."""
    cfg = Cfg(config)
    H = cfg.hidden_size
    num_segments, num_patch_per_img, pp3 = images.shape
    assert pp3 == 768 and audio_clips.shape[0] == 3 * num_segments
    token_length = tokens.shape[0]
    imgs_enc = vision_transformer(params['vision_encoder'], images, cfg.vit_num_layers, cfg.grid_h, cfg.grid_w,
                                  cfg.vit_pooling_ratio)['seq_attnpool'].reshape(num_segments * num_patch_per_img // 4, H)
    audio_enc = audio_transformer(params['audio_encoder'], audio_clips.reshape(-1, cfg.audio_seq_length, 65), cfg.audio_num_layers,
                                  cfg.audio_pooling_ratio, cfg.audio_patch_size)['seq_attnpool']
    mm = prepare_multimodal_inputs(params, cfg, tokens=tokens[None],
                                   token_segment_idx=torch.div(subseg_idxs[None], 3, rounding_mode='floor'),     # jnp // floors
                                   vision_input=imgs_enc[None], audio_pointers=subseg_idxs[None], audio_spans=audio_enc[None])
    joint = transformer_encoder(params['joint_transformer'], mm['x'], cfg.joint_num_layers, rotary_coords=mm['rotary_coords'],
                                attention_mask=mm['attention_mask'])['seq']
    return unit_normalize(dense(joint[0, :token_length], params['head']))


# ----------------------------------------------------------------------------- VCR finetuning (BASELINE config 5)
def vcr_forward(params, config, batch):
    """MerlotReserveVCR.__call__, finetune/vcr/qa_qar_joint_finetune.py:150-170.
    batch: 'image' [B, hw, 768] float, 'answers' [B, 2, A, T] int64 -> logits [B, 2, A]."""
    cfg = Cfg(config)
    H = cfg.hidden_size
    batch_size, two_, num_ans_per, token_length = batch['answers'].shape
    n = batch_size * 2 * num_ans_per
    answers2d = batch['answers'].reshape(n, token_length)
    imgs_enc = vision_transformer(params['vision_encoder'], batch['image'], cfg.vit_num_layers, cfg.grid_h, cfg.grid_w,
                                  cfg.vit_pooling_ratio)['seq_attnpool'].repeat_interleave(2 * num_ans_per, dim=0)
    mm = prepare_multimodal_inputs(params, cfg, tokens=answers2d, token_segment_idx=torch.zeros(n, token_length, dtype=torch.int64),
                                   vision_input=imgs_enc)
    joint = transformer_encoder(params['joint_transformer'], mm['x'], cfg.joint_num_layers, rotary_coords=mm['rotary_coords'],
                                attention_mask=mm['attention_mask'])['seq'][:, :token_length]
    pool_idx = torch.argmax((answers2d == MASK).to(torch.float32), 1)        # first MASK (0 when there is none)
    pooled_h = joint[torch.arange(n), pool_idx]
    return (pooled_h @ params['proj']['kernel']).reshape(batch_size, 2, num_ans_per)


def vcr_train_loss(logits, labels):
    """train_loss_fn, qa_qar_joint_finetune.py:188-195."""
    log_p = torch.log_softmax(logits, dim=-1)
    loss = -(torch.nn.functional.one_hot(labels, log_p.shape[-1]).to(log_p.dtype) * log_p).sum(-1).mean()
    is_right = (torch.argmax(log_p, -1) == labels).to(torch.float32).mean()
    return loss, {'is_right': is_right, 'loss': loss}


def finetune_adam_apply(param, orig_bf16, grad, mu_bf16, nu_bf16, count, opt_config):
    """finetune/optimization.py:56-105 (tx chain) + :151-185 for ONE leaf (replicated optimizer: the 8-way sharding of
    :148-171 only re-lays the same arithmetic).  Bias correction ON, eps 1e-6 by default, decay mask ndim > 1 and
    size > 4096, `- wd * bf16(initial param)` before `+ wd * param`, linear warmup / linear decay."""
    f32 = np.float32
    b1, b2 = opt_config.get('beta_1', 0.9), opt_config.get('beta_2', 0.98)
    eps = opt_config.get('eps', 1e-6)
    next_m = f32(1 - b1) * grad + f32(b1) * mu_bf16.to(torch.float32)
    next_v = f32(1 - b2) * grad * grad + f32(b2) * unsigned_bf16_decode(nu_bf16)
    new_mu, new_nu = next_m.to(torch.bfloat16), unsigned_bf16_encode(next_v)
    m_hat, v_hat = next_m, next_v
    if opt_config.get('do_bias_correction', True):
        c = count + 1
        m_hat, v_hat = next_m / f32(1 - b1 ** c), next_v / f32(1 - b2 ** c)
    u = m_hat / (torch.sqrt(v_hat) + f32(eps))
    if param.ndim > 1 and param.numel() > 4096:
        wd = f32(opt_config['weight_decay_rate'])
        u = u - wd * orig_bf16.to(torch.float32)
        u = u + wd * param
    u = u * lr_scale_linearwarmup_lineardecay(count, opt_config['num_warmup_steps'], opt_config['num_train_steps'])
    u = u * f32(-opt_config['learning_rate'])
    return param + u, new_mu, new_nu
