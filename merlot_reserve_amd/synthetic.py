"""Synthetic, structurally valid pretraining batches with the per-device layout of the reference's dataloader
(pretrain/dataloader.py:732-789 handle_batch, and mask_tokens :192-257 for how token streams, audio pointers and text
pointers relate).  There is no network, so benchmarks and tests run on these; seeds make them reproducible.

Per device (B records):  images [B, nseg*hw, 768] bf16 in [0,1);  audio_clips [B, nspans*60, 65] bf16 (64 log-mel
values in [0,5) + playback speed 1.0);  text2audio / audio2text [B, ngroups, lang] int32 with /audio_ptr and /text_ptr
twins;  audio_text_matching / random_text [B, 1, seq_len] with twins;  text_spans [B, 2*ntrg + budget, 15];
video_src_index [B, nseg].
"""
import numpy as np
import torch

from .config import Dims

PADDING, START, END, MASK, MASKAUDIO, AUDIOSPAN, LTOVPOOL, RESETCTX = 0, 1, 2, 3, 4, 5, 6, 9
VOCAB = 32768


def _pad_stream(rows, length):
    out = np.zeros((length, 3), dtype=np.int32)
    out[:, 1:] = -1
    rows = np.asarray(rows, dtype=np.int32).reshape(-1, 3)[:length]
    out[:len(rows)] = rows
    return out


def _record_text(d, rng):
    nspans, ntrg = d.nspans, d.ntrg1
    per_group = nspans // d.ngroups
    # every sequence of either kind masks its own ntrg spans (dataloader.py:537-545: one draw without replacement, cut into the sequences); text
    # pointers count through the text -> audio sequences first, then the audio -> text ones (:563, :593)
    trg = rng.permutation(nspans)[:(d.n_t2a + d.n_a2t) * ntrg]
    t2a_ranks = [{int(s): k + q * ntrg for k, s in enumerate(np.sort(trg[q * ntrg:(q + 1) * ntrg]))} for q in range(d.n_t2a)]
    a2t_ranks = [{int(s): k + (d.n_t2a + q) * ntrg for k, s in enumerate(np.sort(trg[(d.n_t2a + q) * ntrg:(d.n_t2a + q + 1) * ntrg]))}
                 for q in range(d.n_a2t)]

    def tok(n):
        return rng.integers(10, VOCAB, size=n)

    t2a, a2t = [], []
    for t2a_rank in t2a_ranks:                                 # rows (sequence, group): what `tokens_all.extend(output_groups)` builds
        for g in range(d.ngroups):
            rows_t = []
            for s in range(g * per_group, (g + 1) * per_group):
                if s in t2a_rank:
                    rows_t += [(MASK, s, t2a_rank[s]), (MASKAUDIO, s, t2a_rank[s])]
                else:
                    rows_t += [(int(t), s, -1) for t in tok(rng.integers(1, 5))]
            t2a.append(_pad_stream(rows_t, d.lang))
    for a2t_rank in a2t_ranks:
        for g in range(d.ngroups):
            rows_a = []
            for s in range(g * per_group, (g + 1) * per_group):
                if s in a2t_rank:
                    rows_a += [(MASK, s, a2t_rank[s])]
                elif rng.random() < 0.8:
                    rows_a += [(AUDIOSPAN, s, -1)] * d.a_tok
                else:
                    rows_a += [(int(t), s, -1) for t in tok(rng.integers(1, 5))]
            a2t.append(_pad_stream(rows_a, d.lang))

    rows_m = []
    use_audio = rng.random() < 0.5
    for i in range(d.nseg):
        rows_m.append((LTOVPOOL, i * d.nas, -1))
        if use_audio:
            for j in range(d.nas):
                rows_m += [(AUDIOSPAN, i * d.nas + j, -1)] * d.a_tok
        else:
            rows_m += [(int(t), i * d.nas, -1) for t in tok(rng.integers(3, 9))]
    room = max(d.seq_len - len(rows_m), 0)
    n_aux = int(rng.integers(0, room + 1)) if room > 0 else 0
    aux = [(int(t), -1, -1) for t in tok(n_aux)]
    matching = _pad_stream(aux + rows_m, d.seq_len)

    random_text = []
    for q in range(d.n_text):
        rows_r, seg = [], 0
        n_fill = int(d.seq_len * rng.uniform(0.7, 1.0))
        mask_at = set(np.sort(rng.permutation(max(n_fill, d.budget))[:d.budget]).tolist())
        k = 0
        for pos in range(max(n_fill, d.budget)):
            if pos in mask_at:
                rows_r.append((MASK, seg, (d.n_t2a + d.n_a2t) * ntrg + q * d.budget + k))
                k += 1
                seg += 1
            else:
                rows_r.append((int(tok(1)[0]), seg, -1))
        random_text.append(_pad_stream(rows_r, d.seq_len))

    spans = np.zeros((d.ntext_spans, d.span_len), dtype=np.int32)
    for i in range(d.ntext_spans):
        n = 0 if rng.random() < 0.05 else int(rng.integers(1, d.span_len + 1))
        spans[i, :n] = tok(n)
    vsrc = np.zeros(d.nseg, dtype=np.int32)
    if rng.random() < 0.1:
        vsrc[int(rng.integers(1, d.nseg)):] = 1
    return np.stack(t2a), np.stack(a2t), matching[None], np.stack(random_text), spans, vsrc


def make_batch(config, B, seed=1234, device='cpu', float_dtype=torch.bfloat16):
    """One device's batch.  Integer tensors are returned as numpy int32 (host side: the planner consumes them),
    images / audio as torch tensors on ``device``."""
    d = Dims(config, B)
    rng = np.random.default_rng(seed)
    recs = [_record_text(d, rng) for _ in range(B)]
    batch = {}
    for name, i in (('text2audio', 0), ('audio2text', 1), ('audio_text_matching', 2), ('random_text', 3)):
        x = np.stack([r[i] for r in recs])                      # [B, n, L, 3]
        batch[name] = np.ascontiguousarray(x[..., 0])
        batch[name + '/audio_ptr'] = np.ascontiguousarray(x[..., 1])
        batch[name + '/text_ptr'] = np.ascontiguousarray(x[..., 2])
    batch['text_spans'] = np.stack([r[4] for r in recs])
    batch['video_src_index'] = np.stack([r[5] for r in recs])
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(B, d.nseg * d.hw, d.pp3, generator=g)
    audio = torch.rand(B, d.nspans * d.a_raw, 65, generator=g) * 5.0
    audio[..., 64] = 1.0
    batch['images'] = images.to(float_dtype).to(device)
    batch['audio_clips'] = audio.to(float_dtype).to(device)
    return batch


def make_draws(config, B, seed=1234):
    """The random draws of pretrain_model.py:30 and :222-223, made injectable (JAX's threefry stream cannot be
    reproduced without JAX): split_from_here = 1 + choice(L, p=[p/(L-1)]*(L-1) + [1-p]) for the two augmented
    streams, and the Gumbel noise -log(-log(U)) for the span choice."""
    d = Dims(config, B)
    rng = np.random.default_rng(seed + 7919)
    L = d.nspg
    p = config['model'].get('_augment_video_src_idx_prob', 0.1)
    probs = np.array([p / (L - 1)] * (L - 1) + [1 - p]) if L > 1 else np.array([1.0])
    splits = [1 + rng.choice(L, size=B * rows, p=probs).astype(np.int32) for rows in (d.rows_a2t, d.rows_t2a)]
    u = rng.uniform(1e-9, 1.0, size=(B, d.ntext_spans)).astype(np.float32)
    z = (-np.log(-np.log(u))).astype(np.float32)
    return splits, z
