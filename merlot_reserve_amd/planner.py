"""Host-side planner: every integer decision of the pretraining forward, computed once per batch from the integer
streams alone (they never depend on activations), so the device program is dense kernels + index-driven row moves.

It restates, bit-exactly, the index logic of
  * MerlotReserve.prepare_multimodal_inputs   mreserve/modeling.py:651-758   (audio-span substitution, coordinates,
                                                                             validity / video-source attention mask)
  * MerlotReservePretrainer.__call__          pretrain/pretrain_model.py:93-236 (segment indices, one_hot_pool targets,
                                                                             stable argsort, Gumbel top-k span choice)
  * _augment_video_src_idx                    pretrain/pretrain_model.py:9-36
and emits (a) CSR gather lists for the forward, (b) the inverted CSR lists that turn every scatter-add of the
backward into a fixed-order segment sum, (c) one int32 "attention code" per joint position (valid ? video_src : -1)
instead of the [S,S] mask, (d) the per-position "rotary" scale table (modeling.py:81-144 collapse to a diagonal
scaling: out[2i] = x[2i](sin - cos), out[2i+1] = x[2i+1](sin + cos)).
"""
import math

import numpy as np

PADDING, MASK, MASKAUDIO, AUDIOSPAN, LTOVPOOL = 0, 3, 4, 5, 6
VOCAB = 32768


# ------------------------------------------------------------------------------------------------ coordinates
def rotary_coords_1d(n, center_origin=True):
    """modeling.py:21-35."""
    if center_origin:
        sl0 = n // 2
        return np.concatenate([np.arange(sl0, dtype=np.float64) - float(sl0), 1.0 + np.arange(n - sl0, dtype=np.float64)])
    return 1.0 + np.arange(n, dtype=np.float64)


def rotary_coords_2d(h, w):
    """modeling.py:38-50: [h*w, 2], h first."""
    s = 1.0 / (max(h, w) + 1.0)
    hh, ww = np.meshgrid(s * rotary_coords_1d(h), s * rotary_coords_1d(w), indexing='ij')
    return np.stack([hh, ww], -1).reshape(h * w, 2)


def rot_scale_table(coords, rotary_hsize=32, max_freq=10.0):
    """coords [..., nd] -> [..., 32] fp32 multipliers for the first 32 dims of every q / k head.
    theta = coord * freq * pi with freq = 2^linspace(0, log2(max_freq/2), 32/(2 nd)), dims-major (modeling.py:97-103);
    even dims get sin - cos, odd dims sin + cos (modeling.py:133-142 with its swapped names and self-pairing)."""
    coords = np.asarray(coords, dtype=np.float64)
    nd = coords.shape[-1]
    assert rotary_hsize % (2 * nd) == 0
    de = rotary_hsize // (2 * nd)
    freqs = np.logspace(0.0, math.log2(max_freq / 2.0), de, base=2)
    rad = (coords[..., None] * freqs * np.pi).reshape(*coords.shape[:-1], nd * de)
    s, c = np.sin(rad), np.cos(rad)
    return np.stack([s - c, s + c], -1).reshape(*coords.shape[:-1], 2 * nd * de).astype(np.float32)


def static_tables(d):
    """Per-configuration constants: rotary scale tables of the ViT / audio / span encoders (CLS row = coordinate 0,
    modeling.py:324-326) and the key rows of the two attention pools (modeling.py:419-421, 467)."""
    t = {}
    t['vit_rot'] = rot_scale_table(np.concatenate([np.zeros((1, 2)), rotary_coords_2d(d.gh, d.gw)], 0))
    t['audio_rot'] = rot_scale_table(np.concatenate([np.zeros((1, 1)), rotary_coords_1d(d.a_len)[:, None] / d.a_len], 0))
    t['span_rot'] = rot_scale_table(np.concatenate([np.zeros((1, 1)), rotary_coords_1d(d.span_len, False)[:, None] / 16.0], 0))
    pr, h2, w2 = d.pr, d.gh // d.pr, d.gw // d.pr
    n, i2, j2, di, dj = np.meshgrid(np.arange(d.Nv), np.arange(h2), np.arange(w2), np.arange(pr), np.arange(pr), indexing='ij')
    t['vit_pool_rows'] = (n * d.Sv + 1 + (i2 * pr + di) * d.gw + j2 * pr + dj).reshape(d.Nv * h2 * w2, pr * pr).astype(np.int32)
    n, tt, r = np.meshgrid(np.arange(d.Na), np.arange(d.a_tok), np.arange(d.a_pool), indexing='ij')
    t['audio_pool_rows'] = (n * d.Sa + 1 + tt * d.a_pool + r).reshape(d.Na * d.a_tok, d.a_pool).astype(np.int32)
    return t


# ------------------------------------------------------------------------------------------------ CSR helpers
def csr_from_pairs(dst, src, n_dst):
    """Lists of src per dst, sources in ascending order of appearance (stable): fixed summation order."""
    dst = np.asarray(dst, dtype=np.int64)
    src = np.asarray(src, dtype=np.int64)
    order = np.argsort(dst, kind='stable')
    indptr = np.zeros(n_dst + 1, dtype=np.int32)
    np.cumsum(np.bincount(dst, minlength=n_dst), out=indptr[1:])
    return indptr, src[order].astype(np.int32)


def csr_gather(codes):
    """One-or-zero-element lists: code < 0 -> empty (zero row)."""
    codes = np.asarray(codes, dtype=np.int64).reshape(-1)
    has = codes >= 0
    indptr = np.zeros(len(codes) + 1, dtype=np.int32)
    np.cumsum(has, out=indptr[1:])
    return indptr, codes[has].astype(np.int32)


def augment_video_src_idx(video_src_idx, split_from_here):
    """pretrain_model.py:32-35 (the draw itself, :30, is an input)."""
    L = video_src_idx.shape[1]
    if L == 1:
        return video_src_idx
    split_mask = split_from_here[:, None] <= np.arange(L)[None]
    return np.where(split_mask, video_src_idx + 4 * L, video_src_idx)


# ------------------------------------------------------------------------------------------------ the plan
def build_plan(batch, d, split_from_here, gumbel_z):
    """batch: integer streams of one device (numpy int32, shapes of synthetic.make_batch).  Returns a dict of numpy
    arrays (int32 unless noted)."""
    B, G, lang, Sj = d.B, d.ngroups, d.lang, d.Sj
    BG = B * G
    n_audio_rows = d.Na * d.a_tok
    vis_base = VOCAB + n_audio_rows
    img2d = rotary_coords_2d(d.gh // d.pr, d.gw // d.pr)                     # pooled grid (modeling.py:704-706)
    vis_seg = np.arange(d.nspg).repeat(d.hw4)                                # :711
    vis_hw = np.tile(img2d, [d.nspg, 1])                                     # :710

    gcode = np.full((d.Nj, Sj), -1, dtype=np.int64)
    mcode = np.full((d.Nj, Sj), -1, dtype=np.int64)
    coords = np.zeros((d.Nj, Sj, 4), dtype=np.float64)
    tok_pos = 1.0 + np.arange(Sj, dtype=np.float64)

    vsi = batch['video_src_index'].astype(np.int64).reshape(BG, d.nspg)      # pretrain_model.py:93
    assert vsi.min() >= 0, 'video_src_index must be non-negative'

    def audio_codes(tokens, audio_ptr, rec):
        is_audio = tokens == AUDIOSPAN                                       # modeling.py:685-693
        ptr = np.maximum(audio_ptr, 0)
        subpos = np.maximum(np.cumsum(is_audio.astype(np.int64), -1) - 1, 0) % d.a_tok
        code = VOCAB + (rec[:, None] * d.nspans + ptr) * d.a_tok + subpos
        return is_audio, code

    def fill_grouped(row0, name, split, with_audio, nseq):
        # nseq sequences of this kind per record, each in G groups: rows (record, sequence, group) -- the reference tiles the vision input and
        # video_src_index nseq times along the group axis (pretrain_model.py:104, 109-110: row j of a record reads group j mod G) and repeats the
        # record's audio spans for every row (:106)
        per = G * nseq
        n = B * per
        tokens = batch[name].astype(np.int64).reshape(n, lang)
        ap = batch[name + '/audio_ptr'].astype(np.int64).reshape(n, lang)
        rows = np.arange(n)
        vrow = (rows // per) * G + (rows % per) % G                          # the row of the [B G, vis_len] pooled vision input this sequence reads
        sl = slice(row0, row0 + n)
        code = tokens.copy()
        if with_audio:
            is_audio, acode = audio_codes(tokens, ap, rows // per)
            code = np.where(is_audio, acode, tokens)
        gcode[sl, :lang] = code
        gcode[sl, lang:lang + d.vis_len] = vis_base + vrow[:, None] * d.vis_len + np.arange(d.vis_len)[None]
        seg = (ap // d.nas) % d.nspg                                         # pretrain_model.py:102 (floor semantics)
        vsrc = augment_video_src_idx(vsi[vrow], split.astype(np.int64))
        valid = tokens != PADDING
        mcode[sl, :lang] = np.where(valid, vsrc[rows[:, None], seg], -1)     # modeling.py:743-756
        mcode[sl, lang:lang + d.vis_len] = vsrc[rows[:, None], vis_seg[None]]
        coords[sl, :lang, 2] = seg / 16.0
        coords[sl, :lang, 3] = tok_pos[:lang] / 1024.0
        coords[sl, lang:lang + d.vis_len, 0] = vis_hw[None, :, 0]
        coords[sl, lang:lang + d.vis_len, 1] = vis_hw[None, :, 1]
        coords[sl, lang:lang + d.vis_len, 2] = vis_seg[None] / 16.0
        return tokens

    n_a2t, n_t2a, n_rand = B * d.rows_a2t, B * d.rows_t2a, B * d.n_text
    r_a2t, r_match, r_rand, r_t2a = 0, n_a2t, n_a2t + B, n_a2t + B + n_rand  # sorted-key order, pretrain_model.py:140-144
    tok_a2t = fill_grouped(r_a2t, 'audio2text', split_from_here[0], True, d.n_a2t)
    tok_t2a = fill_grouped(r_t2a, 'text2audio', split_from_here[1], False, d.n_t2a)

    tok_m = batch['audio_text_matching'].astype(np.int64).reshape(B, Sj)
    ap_m = batch['audio_text_matching/audio_ptr'].astype(np.int64).reshape(B, Sj)
    is_audio, acode = audio_codes(tok_m, ap_m, np.arange(B))
    gcode[r_match:r_match + B] = np.where(is_audio, acode, tok_m)
    is_pool = tok_m == LTOVPOOL
    seg_m = np.cumsum(is_pool.astype(np.int64), -1)                          # pretrain_model.py:117
    mcode[r_match:r_match + B] = np.where(tok_m != PADDING, 0, -1)
    coords[r_match:r_match + B, :, 2] = seg_m / 16.0
    coords[r_match:r_match + B, :, 3] = tok_pos[None] / 1024.0

    tok_r = batch['random_text'].astype(np.int64).reshape(n_rand, Sj)
    gcode[r_rand:r_rand + n_rand] = tok_r
    mcode[r_rand:r_rand + n_rand] = np.where(tok_r != PADDING, 0, -1)
    coords[r_rand:r_rand + n_rand, :, 3] = tok_pos[None] / 1024.0

    plan = {}
    plan['joint_gather_indptr'], plan['joint_gather_idx'] = csr_gather(gcode)
    plan['joint_code'] = mcode.reshape(-1).astype(np.int32)
    plan['joint_rot'] = rot_scale_table(coords).reshape(d.Nj * Sj, 32)       # float32

    # ---------------- pooling targets (rows of the head output, index r*Sj + p) ----------------
    def flat_rows(row0, nrows, L):
        return (row0 + np.arange(nrows))[:, None] * Sj + np.arange(L)[None]

    dst, src = [], []
    # vision -> audio: LTOVPOOL rows of the matching sequence to slot cumsum-1   (pretrain_model.py:160-165)
    slot = seg_m - 1
    ok = is_pool & (slot < d.nseg)
    rows_m = flat_rows(r_match, B, Sj)
    dst.append((np.arange(B)[:, None] * d.nseg + slot)[ok])
    src.append(rows_m[ok])
    off_t2a = B * d.nseg

    # text -> audio: MASKAUDIO rows by audio_ptr, the record's groups merged (real_bsize)   (:170-190)
    ap_t2a = batch['text2audio/audio_ptr'].astype(np.int64).reshape(n_t2a, lang)
    ok = (tok_t2a == MASKAUDIO) & (ap_t2a >= 0) & (ap_t2a < d.nspans)
    rec = (np.arange(n_t2a) // d.rows_t2a)[:, None].repeat(lang, 1)
    count = np.zeros((B, d.nspans), dtype=np.int64)
    np.add.at(count, (rec[ok], ap_t2a[ok]), 1)
    idx_sort = np.argsort(-count, axis=-1, kind='stable')                    # jnp.argsort is stable (:181)
    rank_of_span = np.argsort(idx_sort, axis=-1, kind='stable')              # inverse permutation
    rk = rank_of_span[rec[ok], ap_t2a[ok]]
    keep = rk < d.ntrg
    rows_t = flat_rows(r_t2a, n_t2a, lang)
    dst.append(off_t2a + rec[ok][keep] * d.ntrg + rk[keep])
    src.append(rows_t[ok][keep])
    off_sp = off_t2a + B * d.ntrg

    # text spans: MASK rows by text_ptr in three streams   (:195-209)
    nts = d.ntext_spans
    counts = {}
    mask_rows = {}
    for name, tokens, row0, nrow, L, per_rec in (('audio2text', tok_a2t, r_a2t, n_a2t, lang, d.rows_a2t),
                                                 ('text2audio', tok_t2a, r_t2a, n_t2a, lang, d.rows_t2a),
                                                 ('random_text', tok_r, r_rand, n_rand, Sj, d.n_text)):
        tp = batch[name + '/text_ptr'].astype(np.int64).reshape(nrow, L)
        ok = (tokens == MASK) & (tp >= 0) & (tp < nts)
        rec = (np.arange(nrow) // per_rec)[:, None].repeat(L, 1)
        c = np.zeros((B, nts), dtype=np.int64)
        np.add.at(c, (rec[ok], tp[ok]), 1)
        counts[name] = c
        mask_rows[name] = (rec[ok], tp[ok], flat_rows(row0, nrow, L)[ok])
    t2sp_ct = counts['text2audio'] + counts['audio2text'] + counts['random_text']
    t2sp_src = np.stack([np.zeros_like(t2sp_ct), counts['text2audio'], counts['audio2text'], counts['random_text']],
                        -1).argmax(-1) - 1                                   # first maximum, like jnp.argmax
    spans = batch['text_spans'].astype(np.int64).reshape(B, nts, d.span_len)
    is_valid = (spans != PADDING).any(-1) & (t2sp_ct > 0)                    # :212-213
    # :214-224 in float32
    logits_for_pred = is_valid.astype(np.float32) * np.float32(1e6) + np.float32(np.log(4)) * (
        counts['text2audio'] + counts['audio2text']).astype(np.float32)
    score = (logits_for_pred + gumbel_z.astype(np.float32)).astype(np.float32)
    best_sp = np.argsort(-score.reshape(-1), kind='stable')[:d.n_inc * B]    # lax.top_k: ties -> lower index first
    slot_of_flat = np.full(B * nts, -1, dtype=np.int64)
    slot_of_flat[best_sp] = np.arange(len(best_sp))
    for name in ('text2audio', 'audio2text', 'random_text'):                 # summation order of :206
        rec, tp, rows = mask_rows[name]
        sl = slot_of_flat[rec * nts + tp]
        keep = sl >= 0
        dst.append(off_sp + sl[keep])
        src.append(rows[keep])
    n_pool = off_sp + B * d.n_inc
    dst, src = np.concatenate(dst), np.concatenate(src)
    plan['pool_indptr'], plan['pool_idx'] = csr_from_pairs(dst, src, n_pool)
    plan['poolT_indptr'], plan['poolT_idx'] = csr_from_pairs(src, dst, d.Nj * Sj)
    plan['n_pool'] = n_pool
    plan['t2sp_src'] = t2sp_src.reshape(-1)[best_sp].astype(np.int32)
    plan['idx_sort'] = idx_sort.astype(np.int32)
    plan['best_sp'] = best_sp.astype(np.int32)

    # audio CLS targets: [ntrg selected | the other spans as extra negatives] per record   (:183-190)
    sel = (np.arange(B)[:, None] * d.nspans + idx_sort[:, :d.ntrg]).reshape(-1)
    ext = (np.arange(B)[:, None] * d.nspans + idx_sort[:, d.ntrg:]).reshape(-1)
    acls = np.concatenate([sel, ext])
    plan['acls_indptr'], plan['acls_idx'] = csr_gather(acls)
    plan['aclsT_indptr'], plan['aclsT_idx'] = csr_from_pairs(acls, np.arange(len(acls)), d.Na)

    # span encoder input: CLS row (filled separately) + 15 token rows per chosen span   (:233-236)
    sp_tok = spans.reshape(B * nts, d.span_len)[best_sp]
    scode = np.full((d.Ns, d.Ss), -1, dtype=np.int64)
    scode[:, 1:] = sp_tok
    plan['span_gather_indptr'], plan['span_gather_idx'] = csr_gather(scode)
    smask = np.zeros((d.Ns, d.Ss), dtype=np.int32)
    smask[:, 1:] = np.where(sp_tok != PADDING, 0, -1)
    plan['span_code'] = smask.reshape(-1)

    # ---------------- inverted lists for backward ----------------
    g = gcode.reshape(-1)
    jrows = np.arange(d.Nj * Sj)
    is_tok = (g >= 0) & (g < VOCAB) & (g != PADDING)      # PAD positions carry exactly-zero gradients
    s_flat = scode.reshape(-1)
    is_stok = (s_flat > 0)
    emb_dst = np.concatenate([g[is_tok], s_flat[is_stok]])
    emb_src = np.concatenate([jrows[is_tok], d.Nj * Sj + np.arange(d.Ns * d.Ss)[is_stok]])
    plan['embT_indptr'], plan['embT_idx'] = csr_from_pairs(emb_dst, emb_src, VOCAB)
    is_a = (g >= VOCAB) & (g < vis_base)
    plan['audT_indptr'], plan['audT_idx'] = csr_from_pairs(g[is_a] - VOCAB, jrows[is_a], n_audio_rows)
    is_v = g >= vis_base
    plan['visT_indptr'], plan['visT_idx'] = csr_from_pairs(g[is_v] - vis_base, jrows[is_v], BG * d.vis_len)
    return plan
