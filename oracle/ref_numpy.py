"""TEST INFRASTRUCTURE -- a SECOND, independent restatement of the reference's pretraining forward and loss, in float64 NumPy.

Purpose: `oracle/ref_torch.py` is the oracle every GPU parity test uses; it was the
only statement of the semantics.  This file restates the same reference lines again, by a different route -- no torch, no
einsum / batched matmul tricks: attention, LayerNorm, pooling, the one-hot gathers and the contrastive loss are written as
explicit per-sequence / per-head / per-row loops over 2-D `a @ b` products -- so that a transcription slip in either file
shows up as a disagreement (tests/test_oracle_numpy_crosscheck.py compares them to 1e-10 on the tiny configuration, and
finite-differences ref_torch's gradients).  PARITY IS STILL UNPINNED BY THE REFERENCE: it ships no tests or golden vectors
and JAX cannot be imported here; two restatements agreeing removes the single point of failure, no more.

Only tests/ may import this module.  Citations: M = /root/reference/mreserve/modeling.py, P = pretrain/pretrain_model.py.
Third-party semantics assumed (flax 0.3.4 / jax, requirements.txt): nn.Dense = x @ kernel + bias; nn.LayerNorm with
var = E[x^2] - E[x]^2; dot_product_attention_weights scales the QUERY by 1/sqrt(depth) before the product, adds the bias,
softmax over keys; MultiHeadDotProductAttention = Dense q/k/v [H, nh, 64] + the same attention + Dense out [nh, 64, H];
nn.Conv(kernel [k, in, out], stride k, SAME with no padding needed) = Dense over k consecutive frames; jnp.argsort stable;
lax.top_k descending, lower index first on ties; one_hot(-1) = zero row; integer // and % floor.
"""
import math

import numpy as np

PADDING, MASK, MASKAUDIO, AUDIOSPAN, LTOVPOOL = 0, 3, 4, 5, 6      # mreserve/lowercase_encoder.py:9-23
F = np.float64


# ------------------------------------------------------------------------------------------------ coordinates (M:21-113)
def rotary_coordinates(seq_len, center_origin=True):
    if center_origin:                                    # M:30-34: [-floor(L/2) .. -1, 1 .. L - floor(L/2)]
        sl0 = seq_len // 2
        neg = [float(i) - float(sl0) for i in range(sl0)]
        pos = [1.0 + i for i in range(seq_len - sl0)]
        return np.array(neg + pos, F)
    return np.array([1.0 + i for i in range(seq_len)], F)


def rotary_coordinates_2d(h, w):
    base = 1.0 / (max(h, w) + 1.0)                       # M:47
    hc, wc = base * rotary_coordinates(h), base * rotary_coordinates(w)
    out = np.zeros((h * w, 2), F)
    for i in range(h):                                   # meshgrid(h_coords, w_coords, indexing='ij') -> [h, w, 2] (M:50)
        for j in range(w):
            out[i * w + j] = (hc[i], wc[j])
    return out


def rotary_sinusoids(coords, rotary_hsize=32, max_freq=10.0):
    """coords [L, nd] -> (first, second) [L, 32] = sinusoids[0], sinusoids[1] of M:108 = (cos, sin), each repeated twice (M:112)."""
    L, nd = coords.shape
    assert rotary_hsize % (2 * nd) == 0
    dexp = rotary_hsize // (2 * nd)
    ex = np.linspace(0.0, math.log2(max_freq / 2.0), dexp)          # jnp.logspace(0, log2(5), d, base=2) (M:97)
    freqs = np.power(2.0, ex)
    first, second = np.zeros((L, rotary_hsize), F), np.zeros((L, rotary_hsize), F)
    for l in range(L):
        col = 0
        for a in range(nd):                              # radians.reshape(L, nd * dexp): dim-major (M:102-103)
            for k in range(dexp):
                r = coords[l, a] * freqs[k] * np.pi
                first[l, 2 * col] = first[l, 2 * col + 1] = math.cos(r)
                second[l, 2 * col] = second[l, 2 * col + 1] = math.sin(r)
                col += 1
    return first, second


def apply_rotary_head(x, first, second):
    """M:116-144 for one head: x [L, 64].  `sin` := sinusoids[0] (= cos!), `cos` := sinusoids[1] (= sin!) (M:133-134);
    rotated = [-x0, x1, -x2, x3, ...] (M:139); out = x * `cos` + rotated * `sin` on the first 32 dims."""
    L, D = x.shape
    R = first.shape[1]
    out = x.copy()
    for l in range(L):
        for c in range(R):
            rot = -x[l, c] if c % 2 == 0 else x[l, c]
            out[l, c] = x[l, c] * second[l, c] + rot * first[l, c]
    return out


# ------------------------------------------------------------------------------------------------ layers
def layer_norm(x, p, eps=1e-5):
    out = np.zeros_like(x)
    for r in range(x.shape[0]):
        row = x[r]
        mean = row.sum() / row.size
        mean2 = (row * row).sum() / row.size
        var = mean2 - mean * mean
        out[r] = (row - mean) * (1.0 / math.sqrt(var + eps)) * p['scale'] + p['bias']
    return out


def dense(x, p):
    k = p['kernel']
    y = x @ k.reshape(x.shape[-1], -1)
    if 'bias' in p:
        y = y + p['bias'].reshape(-1)
    return y


def softmax_rows(s):
    out = np.zeros_like(s)
    for r in range(s.shape[0]):
        m = s[r].max()
        e = np.exp(s[r] - m)
        out[r] = e / e.sum()
    return out


def attention_layer(p, x, sinus, bias):
    """M:205-237 + M:188-200 for ONE sequence: x [L, H]; bias [L, L] additive or None."""
    L, H = x.shape
    nh = H // 64
    qkv = dense(x, p['qkv']).reshape(L, 3 * nh, 64)          # heads 0..nh-1 = q, nh..2nh-1 = k, 2nh.. = v (M:191,196)
    ctx = np.zeros((L, nh, 64), F)
    for h in range(nh):
        q, k, v = qkv[:, h], qkv[:, nh + h], qkv[:, 2 * nh + h]
        if sinus is not None:
            q, k = apply_rotary_head(q, *sinus), apply_rotary_head(k, *sinus)
        s = (q / math.sqrt(64.0)) @ k.T                       # flax: query scaled BEFORE the product
        if bias is not None:
            s = s + bias
        ctx[:, h] = softmax_rows(s) @ v
    return ctx.reshape(L, H) @ p['attn_proj']['kernel'].reshape(H, H)      # no bias (M:235)


def mlp_block(p, x):
    h = dense(x, p['intermediate'])
    h = h * (1.0 / (1.0 + np.exp(-1.702 * h)))                # my_gelu (M:240-241)
    return h @ p['out']['kernel']


def transformer_encoder(p, x, num_layers, coords=None, mask=None, is_valid=None, add_cls=False):
    """M:283-376 for ONE sequence: x [L, H]; coords [L, nd]; mask [L, L] bool or is_valid [L] bool."""
    L, H = x.shape
    if add_cls:
        assert mask is None
        x = np.concatenate([p['cls'][None], x], 0)                                   # M:316-320
        if is_valid is not None:
            is_valid = np.concatenate([[True], is_valid])                              # M:321-322
        if coords is not None:
            coords = np.concatenate([np.zeros((1, coords.shape[1]), F), coords], 0)   # M:324-326
    sinus = rotary_sinusoids(coords) if coords is not None else None
    if sinus is None:                                                                  # M:335-341: no coordinates -> learned position embeddings
        assert p['pe'].shape == x.shape
        x = x + p['pe']
    if is_valid is not None:
        assert mask is None
        mask = np.outer(is_valid, is_valid)                                            # M:343-345
    bias = None if mask is None else np.where(mask, 0.0, -1e10)                      # M:350-356
    x = layer_norm(x, p['pre_ln'])
    for i in range(num_layers):
        lp = p[f'layer_{i:02d}']
        x = x + attention_layer(lp['attention_layer'], layer_norm(x, lp['pre_attn_ln']), sinus, bias)
        x = x + mlp_block(lp['mlp_layer'], layer_norm(x, lp['pre_mlp_ln']))
    x = layer_norm(x, p['final_ln'])
    if add_cls:
        return {'cls': dense(x[0:1], p['cls_proj'])[0], 'seq': x[1:]}
    return {'seq': x}


def attnpool(p, group):
    """flax MultiHeadDotProductAttention with one query (the group's mean, M:423) over the group's rows: group [R, H] -> [H]."""
    R, H = group.shape
    nh = H // 64
    qin = group.sum(0, keepdims=True) / R
    q = dense(qin, p['query']).reshape(1, nh, 64)
    k = dense(group, p['key']).reshape(R, nh, 64)
    v = dense(group, p['value']).reshape(R, nh, 64)
    ctx = np.zeros((nh, 64), F)
    for h in range(nh):
        s = (q[:, h] / math.sqrt(64.0)) @ k[:, h].T
        ctx[h] = (softmax_rows(s) @ v[:, h])[0]
    return ctx.reshape(1, H) @ p['out']['kernel'].reshape(H, H) + p['out']['bias']


def vision_transformer(p, patches, num_layers, gh, gw, pr=2):
    """M:379-430 for ONE image: patches [gh * gw, 768] -> cls [H], seq_attnpool [(gh/pr)(gw/pr), H]."""
    x = dense(patches, p['embedding'])
    t = transformer_encoder(p['transformer'], x, num_layers, coords=rotary_coordinates_2d(gh, gw), add_cls=True)
    seq = t['seq']
    pooled = []
    for i2 in range(gh // pr):                                # reshape [h2, pr, w2, pr, H] -> swap -> groups of pr*pr (M:419-421)
        for j2 in range(gw // pr):
            rows = [(i2 * pr + di) * gw + j2 * pr + dj for di in range(pr) for dj in range(pr)]
            pooled.append(attnpool(p['seq_attnpool'], seq[rows])[0])
    t['seq_attnpool'] = np.stack(pooled)
    return t


def audio_transformer(p, clip, num_layers, pooling_ratio, patch=2):
    """M:433-476 for ONE clip: clip [60, 65] -> cls [H], seq_attnpool [60 / patch / ratio, H]."""
    raw, nm = clip.shape
    assert nm == 65 and raw % patch == 0
    L = raw // patch
    k = p['embedding']['kernel']                              # [patch, 65, H]
    x = np.zeros((L, k.shape[2]), F)
    for l in range(L):                                        # strided conv = sum over the patch's frames
        for j in range(patch):
            x[l] += clip[l * patch + j] @ k[j]
        x[l] += p['embedding']['bias']
    coords = (rotary_coordinates(L) / L)[:, None]             # M:457
    t = transformer_encoder(p['transformer'], x, num_layers, coords=coords, add_cls=True)
    seq = t['seq']
    t['seq_attnpool'] = np.stack([attnpool(p['seq_attnpool'], seq[g * pooling_ratio:(g + 1) * pooling_ratio])[0]
                                  for g in range(L // pooling_ratio)])
    return t


def span_transformer(p, x, valid, num_layers, max_len=16):
    L = x.shape[0]
    assert L < max_len
    coords = (rotary_coordinates(L, center_origin=False) / max_len)[:, None]          # M:497
    return transformer_encoder(p['transformer'], x, num_layers, coords=coords, is_valid=valid, add_cls=True)['cls']


def one_hot_pool(do_pool, idx, v, num_segments):
    """M:541-567 after the optional re-batching: do_pool, idx [L]; v [L, H] -> x [S, H], count [S]."""
    out, cnt = np.zeros((num_segments, v.shape[1]), F), np.zeros(num_segments, F)
    for l in range(v.shape[0]):
        if do_pool[l] and 0 <= idx[l] < num_segments:         # one_hot(-1) (and any out-of-range class) is a zero row
            out[idx[l]] += v[l]
            cnt[idx[l]] += 1.0
    return out, cnt


def unit_normalize(x):
    out = np.zeros_like(x)
    for r in range(x.shape[0]):
        out[r] = x[r] / math.sqrt((x[r] * x[r]).sum() + 1e-5)                          # M:576-577
    return out


# ------------------------------------------------------------------------------------------------ multimodal assembly (M:651-758)
def prepare_multimodal_inputs(emb, cfg, tokens, token_segment_idx=None, token_embs=None, vision_input=None, audio_spans=None,
                              audio_pointers=None, padding_len=None, video_src_idx=None):
    """ONE sequence: tokens [L]; token_embs [L, H]; vision_input [V, H]; audio_spans [n, 6, H]; video_src_idx [nseg]."""
    m = cfg['model']
    H, atl = m['hidden_size'], m['audio_token_length']
    L = tokens.shape[0]
    x = token_embs.copy() if token_embs is not None else emb[tokens]
    if audio_spans is not None and audio_pointers is not None:
        n_audio = 0
        for l in range(L):                                    # M:685-695
            is_a = tokens[l] == AUDIOSPAN
            n_audio += int(is_a)
            if is_a:
                x[l] = audio_spans[max(int(audio_pointers[l]), 0), max(n_audio - 1, 0) % atl]
    coords = np.zeros((L, 4), F)
    for l in range(L):
        coords[l, 2] = 0.0 if token_segment_idx is None else token_segment_idx[l] / 16.0
        coords[l, 3] = (1.0 + l) / 1024.0                     # M:697-700, :54,77
    valid = [bool(t != PADDING) for t in tokens]
    src = None
    if video_src_idx is not None and token_segment_idx is not None:
        src = [int(video_src_idx[token_segment_idx[l]]) for l in range(L)]
    if vision_input is not None:
        hp, wp = m['output_grid'][0] // m['vit_pooling_ratio'], m['output_grid'][1] // m['vit_pooling_ratio']
        ic = rotary_coordinates_2d(hp, wp)
        V = vision_input.shape[0]
        vc = np.zeros((V, 4), F)
        for t in range(V):                                    # M:702-720: frame t // (hp wp), pooled-grid position t % (hp wp)
            seg, pos = divmod(t, hp * wp)
            vc[t] = (ic[pos, 0], ic[pos, 1], seg / 16.0, 0.0)
            valid.append(True)
            if src is not None:
                src.append(int(video_src_idx[seg]))
        coords = np.concatenate([coords, vc], 0)
        x = np.concatenate([x, vision_input], 0)
    if padding_len is not None:
        extra = padding_len - len(valid)
        assert extra >= 0
        valid += [False] * extra
        if src is not None:
            src += [-1] * extra
        coords = np.concatenate([coords, np.zeros((extra, 4), F)], 0)
        x = np.concatenate([x, np.zeros((extra, H), F)], 0)
    S = len(valid)
    mask = np.zeros((S, S), bool)
    for i in range(S):
        for j in range(S):
            mask[i, j] = valid[i] and valid[j] and (src is None or src[i] == src[j])       # M:743-756
    return x, coords, mask


# ------------------------------------------------------------------------------------------------ pretraining forward (P:38-259)
def pretrain_forward(params, cfg, batch, split_from_here, gumbel_z):
    """params: nested dict of float64 arrays (Flax names); batch: dict of numpy arrays, ONE device's slice;
    split_from_here: two int arrays [2B] (the draws of P:30, + 1); gumbel_z [B, num_text_spans]."""
    d, m = cfg['data'], cfg['model']
    H = m['hidden_size']
    gh, gw = m['output_grid']
    B = batch['images'].shape[0]
    hw = gh * gw
    nseg = batch['images'].shape[1] // hw
    ngr = d['num_segment_groups']
    nspg = nseg // ngr
    nas = d['num_audio_subsegments']
    nspans = nseg * nas
    lang, seq_len = d['lang_seq_len'], d['seq_len']
    hw4 = hw // (m['vit_pooling_ratio'] ** 2)
    a_ratio = m['audio_seq_length'] // (m['audio_token_length'] * m['audio_patch_size'])
    emb = params['token_encoder']['Embed_0']['embedding']

    imgs_cls, imgs_seq = [], []
    for b in range(B):
        for s in range(nseg):
            t = vision_transformer(params['vision_encoder'], batch['images'][b, s * hw:(s + 1) * hw].astype(F), m['vit_num_layers'], gh, gw,
                                   m['vit_pooling_ratio'])
            imgs_cls.append(t['cls'])
            imgs_seq.append(t['seq_attnpool'])
    audio_cls, audio_seq = [], []
    al = m['audio_seq_length']
    for b in range(B):
        for s in range(nspans):
            t = audio_transformer(params['audio_encoder'], batch['audio_clips'][b, s * al:(s + 1) * al].astype(F), m['audio_num_layers'],
                                  a_ratio, m['audio_patch_size'])
            audio_cls.append(t['cls'])
            audio_seq.append(t['seq_attnpool'])

    def vision_of(b, g):                                      # [nspg * hw4, H]: frames g*nspg .. of record b (P:58-59)
        return np.concatenate([imgs_seq[b * nseg + g * nspg + f] for f in range(nspg)], 0)

    def spans_of(b):                                          # [nspans, 6, H]
        return np.stack([audio_seq[b * nspans + s] for s in range(nspans)])

    def augmented_src(b, g, split):                           # P:9-36 on row (b, g) of video_src_index.reshape(-1, nspg)
        row = batch['video_src_index'][b].reshape(ngr, nspg)[g].astype(np.int64).copy()
        if nspg > 1:
            for j in range(nspg):
                if split <= j:
                    row[j] += 4 * nspg
        return row

    # the four kinds of joint sequences, concatenated in sorted-key order (P:140-144)
    seqs, kinds = [], []
    # more than one sequence per kind (P:99-110, 124-135): jnp.tile(x, [1, n, 1, 1]) repeats the GROUP axis n times, so row j of a record reads
    # group j mod ngr of the vision input and of video_src_index; the record's audio spans serve every row
    rows_a2t, rows_t2a, n_text = ngr * d['num_audio2text_seqs'], ngr * d['num_text2audio_seqs'], d.get('num_text_seqs', 1)
    use_coords = m.get('do_rotary', True)                     # P:146-148
    for b in range(B):
        for j in range(rows_a2t):
            g = j % ngr
            tok, ap = batch['audio2text'][b, j], batch['audio2text/audio_ptr'][b, j]
            seg = (ap // nas) % nspg                          # floor semantics: ptr = -1 -> segment nspg - 1 (P:102)
            seqs.append(prepare_multimodal_inputs(emb, cfg, tok, seg, emb[tok], vision_of(b, g), spans_of(b), ap, seq_len,
                                                  augmented_src(b, g, split_from_here[0][b * rows_a2t + j])))
            kinds.append(('audio2text', b, j))
    for b in range(B):
        tok, ap = batch['audio_text_matching'][b, 0], batch['audio_text_matching/audio_ptr'][b, 0]
        seg = np.cumsum(tok == LTOVPOOL)                      # inclusive cumsum (P:117)
        seqs.append(prepare_multimodal_inputs(emb, cfg, tok, seg, emb[tok], None, spans_of(b), ap, seq_len, None))
        kinds.append(('audio_text_matching', b, 0))
    for b in range(B):
        for j in range(n_text):
            tok = batch['random_text'][b, j]
            seqs.append(prepare_multimodal_inputs(emb, cfg, tok, None, None, None, None, None, seq_len, None))
            kinds.append(('random_text', b, j))
    for b in range(B):
        for j in range(rows_t2a):
            g = j % ngr
            tok, ap = batch['text2audio'][b, j], batch['text2audio/audio_ptr'][b, j]
            seg = (ap // nas) % nspg
            seqs.append(prepare_multimodal_inputs(emb, cfg, tok, seg, emb[tok], vision_of(b, g), None, ap, seq_len,
                                                  augmented_src(b, g, split_from_here[1][b * rows_t2a + j])))
            kinds.append(('text2audio', b, j))
    outs = {}
    for (x, coords, mask), key in zip(seqs, kinds):
        enc = transformer_encoder(params['joint_transformer'], x, m['joint_num_layers'], coords=coords if use_coords else None, mask=mask)['seq']
        outs[key] = dense(enc, params['head'])                # P:150-151

    # vision -> audio: rows at LTOVPOOL positions, slot = cumsum - 1 (P:160-165)
    a2v = []
    for b in range(B):
        tok = batch['audio_text_matching'][b, 0]
        is_pool = tok == LTOVPOOL
        x, _ = one_hot_pool(is_pool, np.cumsum(is_pool) - 1, outs[('audio_text_matching', b, 0)], nseg)
        a2v.append(x)
    a2v = np.concatenate(a2v, 0)
    # text -> audio (P:170-190): pool at MASKAUDIO by audio_ptr, the record's groups merged (real_bsize)
    ntrg = int(nspans * d['mask_rate']) * d['num_text2audio_seqs']
    t2a_sel, a2t_sel, a2t_extra = [], [], []
    for b in range(B):
        x, cnt = np.zeros((nspans, H), F), np.zeros(nspans, F)
        for g in range(rows_t2a):
            tok = batch['text2audio'][b, g]
            xg, cg = one_hot_pool(tok == MASKAUDIO, batch['text2audio/audio_ptr'][b, g], outs[('text2audio', b, g)][:lang], nspans)
            x, cnt = x + xg, cnt + cg
        order = sorted(range(nspans), key=lambda i: -cnt[i])   # python's sort is stable, like jnp.argsort (P:181)
        for i in order[:ntrg]:
            t2a_sel.append(x[i])
            a2t_sel.append(audio_cls[b * nspans + i])
        for i in order[ntrg:]:
            a2t_extra.append(audio_cls[b * nspans + i])
    # text spans (P:195-236)
    nts = batch['text_spans'].shape[1]
    pooled, counts = {}, {}
    for k in ('audio2text', 'text2audio', 'random_text'):
        for b in range(B):
            x, cnt = np.zeros((nts, H), F), np.zeros(nts, F)
            groups = range({'audio2text': rows_a2t, 'text2audio': rows_t2a, 'random_text': n_text}[k])
            for g in groups:
                tok, tp = batch[k][b, g], batch[f'{k}/text_ptr'][b, g]
                v = outs[(k, b, g)]
                if k != 'random_text':
                    v = v[:lang]
                xg, cg = one_hot_pool(tok == MASK, tp, v, nts)
                x, cnt = x + xg, cnt + cg
            pooled[(k, b)], counts[(k, b)] = x, cnt
    t2sp_sel, t2sp_src, score = [], [], []
    for b in range(B):
        for s in range(nts):
            c = [0.0, counts[('text2audio', b)][s], counts[('audio2text', b)][s], counts[('random_text', b)][s]]
            t2sp_src.append(int(np.argmax(c)) - 1)            # first maximum (P:208-209)
            t2sp_sel.append(pooled[('text2audio', b)][s] + pooled[('audio2text', b)][s] + pooled[('random_text', b)][s])
            valid = bool((batch['text_spans'][b, s] != PADDING).any()) and (c[1] + c[2] + c[3]) > 0.0
            # P:214-224, in float32 like the reference's fp32 mode (the ordering is what matters)
            sc = np.float32(1e6) * np.float32(valid) + np.float32(np.log(4)) * np.float32(c[1] + c[2])
            score.append(np.float32(sc) + np.float32(gumbel_z[b, s]))
    n_inc = d['num_text_spans_to_include']
    best = sorted(range(B * nts), key=lambda i: -score[i])[:n_inc * B]     # lax.top_k: descending, lower index first on ties
    spans_flat = batch['text_spans'].reshape(B * nts, -1)
    sp2t = [span_transformer(params['span_encoder'], emb[spans_flat[i]], spans_flat[i] != PADDING, m['span_num_layers']) for i in best]
    ls = np.minimum(params['contrastive_scales'].astype(F), math.log(100.0))
    temps = np.exp(ls / 2.0)
    N = lambda rows, t: unit_normalize(np.stack(rows)) * t
    return {'imgs_to_audio': {'x': unit_normalize(a2v) * temps[0], 'y': N(imgs_cls, temps[0])},
            'text_to_audio': {'x': N(t2a_sel, temps[1]), 'y': N(a2t_sel, temps[1]), 'y_extra': N(a2t_extra, temps[1])},
            'stuff_to_span': {'x': N([t2sp_sel[i] for i in best], temps[2]), 'y': N(sp2t, temps[2]),
                              '_sources': np.array([t2sp_src[i] for i in best])}}


def loss_fn_given_preds(preds_per_device, rank=0):
    """P:262-303 for device `rank` of a virtual pmap (all_gather = rank-major concatenation, P:290)."""
    preds = dict(preds_per_device[rank])
    info = {}
    if 'text_preds' in preds:                                # P:265-274: masked token cross-entropy, rows with label 0 left out
        tp = preds.pop('text_preds')
        tot, cnt = 0.0, 0
        for row, lab in zip(np.asarray(tp['logits'], dtype=F), np.asarray(tp['labels'])):
            if lab != 0:
                mx = row.max()
                tot += (mx + math.log(np.exp(row - mx).sum())) - row[int(lab)]
                cnt += 1
        info['audio2text'] = tot / cnt
    for ctype, cd in preds.items():
        info[ctype] = 0.0
        if '_sources' in cd:
            for k in ('text2audio', 'audio2text', 'random_text'):
                info[f'_{ctype}_from_{k}'] = 0.0
        for k1, k2 in (('x', 'y'), ('y', 'x')):
            x = cd[k1]
            ys = []
            for pd in preds_per_device:
                ys.append(pd[ctype][k2])
                if f'{k2}_extra' in pd[ctype]:
                    ys.append(pd[ctype][f'{k2}_extra'])
            y_all = np.concatenate(ys, 0)
            terms = []
            for l in range(x.shape[0]):
                logits = y_all @ x[l]
                mx = logits.max()
                lse = mx + math.log(np.exp(logits - mx).sum())
                numer = float((cd['x'][l] * cd['y'][l]).sum())                            # P:276: same row of x and y
                terms.append(lse - numer)
            info[ctype] += (sum(terms) / len(terms)) / 2.0
            if '_sources' in cd:
                for i, t in enumerate(('text2audio', 'audio2text', 'random_text')):
                    sel = [terms[l] for l in range(len(terms)) if cd['_sources'][l] == i]
                    info[f'_{ctype}_from_{t}'] += (sum(sel) / (len(sel) + 1e-5)) / 2.0
    loss = sum(v for k, v in info.items() if not k.startswith('_'))
    return loss, info
