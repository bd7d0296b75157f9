"""The mreserve.modeling drop-in on the MI355X against the oracle (fp32 torch restatement of the reference).

fp32 model (the reference's dtype off-TPU, M:999-1000): stated tolerance 1e-3 relative (north-star forward-parity bar;
measured ~1e-5..1e-6).  bf16 model: 2e-2 relative (bf16 storage, 2^-8 per element).  Integer outputs bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from merlot_reserve_amd import modeling as M
from merlot_reserve_amd import preprocess as P
from merlot_reserve_amd.config import load_config, tiny_config
from oracle import ref_torch as R
from tests.test_modeling_host import FixtureTokenizer
from tests.util import relerr, tree_to

pytestmark = pytest.mark.gpu
FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tokenizer_ids.json')))


def make_model(dev, bf16, H=128, grid=(4, 6)):
    cfg = tiny_config(hidden_size=H, grid=grid)
    cfg['model']['use_bfloat16'] = bf16
    model = M.MerlotReserve.from_config(cfg, device=dev)
    params = model.init_from_dummy_batch(seed=5)
    # the reference's initialisers give near-zero biases / unit LN scales: perturb so every term is exercised
    g = torch.Generator().manual_seed(7)

    def jitter(t):
        return t + 0.05 * torch.randn(t.shape, generator=g) if t.dim() <= 2 and t.shape[-1] != 32768 else t
    params = jax_tree_map(jitter, params)
    if bf16:    # the oracle must read the same (bf16-representable) weights
        params = jax_tree_map(lambda t: t.to(torch.bfloat16).float(), params)
    return cfg, model, params


def jax_tree_map(fn, tree):
    return {k: jax_tree_map(fn, v) for k, v in tree.items()} if isinstance(tree, dict) else fn(tree)


def video_inputs(cfg, seed=0, n_seg=4):
    rng = np.random.default_rng(seed)
    gh, gw = cfg['model']['output_grid']
    segs = [{'patches': rng.random((gh * gw, 768)).astype(np.float32), 'text': rng.integers(10, 32768, size=9).tolist() + [3]}]
    for i in range(1, n_seg):
        spec = rng.random((3, 60, 65)).astype(np.float32) * 5
        spec[..., 64] = 1.0
        segs.append({'patches': rng.random((gh * gw, 768)).astype(np.float32), 'spectrogram': spec, 'use_text_as_input': i % 2 == 0,
                     'text': rng.integers(10, 32768, size=5).tolist()})
    return P.preprocess_video(segs, (gh, gw))


@pytest.mark.parametrize('bf16', [False, True])
def test_sub_encoders_match_oracle(dev, bf16):
    cfg, model, params = make_model(dev, bf16)
    tol = 2e-2 if bf16 else 1e-3
    model.bind(params)
    rng = np.random.default_rng(1)
    q = (lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()) if bf16 else (lambda a: a)
    imgs = q(rng.random((3, 24, 768)).astype(np.float32))
    out = model.vision_encoder(imgs)
    ref = R.vision_transformer(params['vision_encoder'], torch.from_numpy(imgs), 2, 4, 6, 2)
    for k in ('cls', 'seq', 'seq_attnpool'):
        assert out[k].shape == ref[k].shape and relerr(out[k], ref[k]) < tol, (k, relerr(out[k], ref[k]))
    aud = q((rng.random((2, 3, 60, 65)) * 4).astype(np.float32))
    out = model.audio_encoder(aud)                                        # extra batch dims are kept
    ref = R.audio_transformer(params['audio_encoder'], torch.from_numpy(aud).reshape(6, 60, 65), 2, 5, 2)
    assert out['seq_attnpool'].shape == (2, 3, 6, 128) and out['cls'].shape == (2, 3, 128)
    for k in ('cls', 'seq', 'seq_attnpool'):
        assert relerr(out[k].reshape(ref[k].shape), ref[k]) < tol, k
    spans = rng.integers(10, 32768, size=(5, 15))
    spans[0, 4:] = 0
    spans[3, 1:] = 0
    got = model.embed_text_spans_only(spans)
    ref = R.embed_text_spans_only(params, cfg, torch.from_numpy(spans))
    assert relerr(got, ref) < tol
    assert abs(float(got.float().norm(dim=-1).mean()) - 1.0) < 1e-2
    got = model.embed_audio_only(aud)
    assert relerr(got.reshape(6, -1), R.embed_audio_only(params, cfg, torch.from_numpy(aud)).reshape(6, -1)) < tol
    emb = model.token_encoder({'b': spans, 'a': spans[:2, :3]})
    assert emb['a'].shape == (2, 3, 128) and torch.equal(emb['a'].float().cpu(), params['token_encoder']['Embed_0']['embedding'][torch.from_numpy(spans[:2, :3])])


@pytest.mark.parametrize('bf16', [False, True])
def test_embed_video_matches_oracle(dev, bf16):
    cfg, model, params = make_model(dev, bf16)
    tol = 2e-2 if bf16 else 1e-3
    v = video_inputs(cfg)
    if bf16:
        v['images'] = torch.from_numpy(v['images']).to(torch.bfloat16).float().numpy()
        v['audio_clips'] = torch.from_numpy(v['audio_clips']).to(torch.bfloat16).float().numpy()
    got = model.apply({'params': params}, **v, method=model.embed_video)
    ref = R.embed_video(params, cfg, torch.from_numpy(v['images']), torch.from_numpy(v['audio_clips']),
                        torch.from_numpy(v['tokens'].astype(np.int64)), torch.from_numpy(v['subseg_idxs'].astype(np.int64)))
    valid = torch.from_numpy(v['tokens'] != 0)
    assert got.shape == ref.shape == (160, 128)
    e = relerr(got[valid.to(got.device)], ref[valid])
    assert e < tol, e
    # batch_embed_video == vmap(embed_video)  (M:845-846)
    v2 = video_inputs(cfg, seed=3)
    bat = model.batch_embed_video(*[np.stack([v[k], v2[k]]) for k in ('images', 'audio_clips', 'tokens', 'subseg_idxs')])
    one = model.embed_video(**v2)
    assert relerr(bat[0][valid.to(got.device)], got[valid.to(got.device)]) < (1e-5 if not bf16 else 1e-2)
    assert relerr(bat[1][torch.from_numpy(v2['tokens'] != 0).to(got.device)], one[torch.from_numpy(v2['tokens'] != 0).to(got.device)]) < (1e-5 if not bf16 else 1e-2)
    # pre-encoded variants agree with the end-to-end call (M:880-931)
    imgs_enc = model.get_imgseq_only(v['images'])
    aud_enc = model.get_audioseq_only(v['audio_clips'])
    pre = model.embed_preencoded_audio(imgs_enc, aud_enc, v['tokens'], v['subseg_idxs'], v['subseg_idxs'])
    assert relerr(pre[valid.to(got.device)], got[valid.to(got.device)]) < (1e-6 if not bf16 else 1e-2)


def test_prepare_multimodal_inputs_matches_oracle(dev):
    cfg, model, params = make_model(dev, False)
    model.bind(params)
    rng = np.random.default_rng(4)
    B, L, V = 3, 20, 12
    tokens = rng.integers(10, 32768, size=(B, L))
    tokens[:, 5:11] = 5
    tokens[1, 15:] = 0
    ap = rng.integers(-1, 4, size=(B, L))
    seg = rng.integers(0, 2, size=(B, L))
    vis = rng.standard_normal((B, V, 128)).astype(np.float32)
    aud = rng.standard_normal((B, 4, 6, 128)).astype(np.float32)
    vsi = np.array([[0, 1], [0, 0], [2, 0]])
    got = model.prepare_multimodal_inputs(tokens, token_segment_idx=seg, vision_input=vis, audio_spans=aud, audio_pointers=ap,
                                          padding_len=40, video_src_idx=vsi)
    ref = R.prepare_multimodal_inputs(params, R.Cfg(cfg), torch.from_numpy(tokens), token_segment_idx=torch.from_numpy(seg),
                                      vision_input=torch.from_numpy(vis), audio_spans=torch.from_numpy(aud),
                                      audio_pointers=torch.from_numpy(ap), padding_len=40, video_src_idx=torch.from_numpy(vsi))
    assert torch.equal(got['x'].cpu(), ref['x'])                                   # pure gather: bit-exact
    assert np.allclose(got['rotary_coords'], ref['rotary_coords'], atol=1e-7)
    assert torch.equal(got['attention_mask'].cpu(), ref['attention_mask'])         # bit-exact mask
    # the dense mask alone is enough for the encoder (converted back to codes on the host)
    a = model.joint_transformer(got['x'], rotary_coords=got['rotary_coords'], attention_mask=got['attention_mask'])['seq']
    b = model.joint_transformer(got['x'], rotary_coords=got['rotary_coords'], attention_code=got['attention_code'])['seq']
    r = R.transformer_encoder(params['joint_transformer'], ref['x'], 2, rotary_coords=ref['rotary_coords'], attention_mask=ref['attention_mask'])['seq']
    ok = torch.from_numpy(got['attention_code'] >= 0)
    assert relerr(a[ok.to(dev)], r[ok]) < 1e-3 and torch.equal(a, b)
    pooled = M.one_hot_pool(tokens == 5, np.cumsum(tokens == 5, -1) - 1, got['x'][:, :L].contiguous(), num_segments=8)
    rp = R.one_hot_pool(torch.from_numpy(tokens == 5), torch.from_numpy(np.cumsum(tokens == 5, -1) - 1), ref['x'][:, :L], 8)
    assert torch.allclose(pooled['x'].cpu(), rp['x'], atol=1e-6) and np.array_equal(pooled['idx_oh'], rp['idx_oh'].numpy())


def test_pretrained_wrapper_and_label_space(dev, tmp_path):
    from merlot_reserve_amd.checkpoint import save_checkpoint
    cfg, model, params = make_model(dev, False)
    pm = M.PretrainedMerlotReserve(encoder=FixtureTokenizer(), params=params, model=model)
    opts = ['making coffee', 'going backpacking']
    ls = pm.get_label_space(opts)
    ids = torch.tensor([FIX['encode_padded_15'][o] for o in opts])
    assert ls.shape == (2, 128) and relerr(ls, R.embed_text_spans_only(params, cfg, ids)) < 1e-3
    v = video_inputs(cfg)
    out_h = pm.embed_video(**v)
    out_h = out_h[torch.from_numpy(v['tokens'] == M.MASK).to(dev)]
    logits = 100.0 * out_h @ ls.T                                            # demo/demo_video.py:41
    assert logits.shape == (1, 2) and torch.isfinite(logits).all()
    # a released-format checkpoint file on disk -> from_pretrained (no network): weights go through fp16 on disk
    ck = {'step': 0, 'params': jax_tree_map(lambda t: t, M.MerlotReserve.from_config(
        M.PretrainedMerlotReserve._config('base', (12, 20), False), device='cpu').init_from_dummy_batch(seed=1)), 'opt_state': None}
    fn = save_checkpoint(ck, str(tmp_path / 'x'), no_optimizer=True)
    os.rename(fn, str(tmp_path / 'base'))
    pm2 = M.PretrainedMerlotReserve.from_pretrained('base', image_grid_size=(12, 20), cache_dir=str(tmp_path), device=dev, encoder=FixtureTokenizer())
    ls2 = pm2.get_label_space(opts)
    assert ls2.shape == (2, 768) and abs(float(ls2.norm(dim=-1)[0]) - 1.0) < 1e-3


def test_config1_zero_shot_forward_base_size(dev):
    """BASELINE config 1: demo/demo_video.py zero-shot forward, base size, random weights, 8 segments at grid (18, 32),
    fp32 -- against the oracle on the host cores (the reference's own run of this config is JAX on CPU)."""
    pm = M.PretrainedMerlotReserve.from_random('base', image_grid_size=(18, 32), seed=0, device=dev, encoder=FixtureTokenizer())
    cfg = M.PretrainedMerlotReserve._config('base', (18, 32), False)
    rng = np.random.default_rng(0)
    segs = [{'patches': rng.random((576, 768)).astype(np.float32), 'text': "in this video i'll be<|MASK|>"}]
    for i in range(1, 8):
        spec = (rng.random((3, 60, 65)) * 5).astype(np.float32)
        spec[..., 64] = 1.0
        segs.append({'patches': rng.random((576, 768)).astype(np.float32), 'spectrogram': spec, 'use_text_as_input': False})
    v = P.preprocess_video(segs, (18, 32), encoder=pm.encoder)
    got = pm.embed_video(**v)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    with torch.no_grad():
        ref = R.embed_video(pm.params, cfg, torch.from_numpy(v['images']), torch.from_numpy(v['audio_clips']),
                            torch.from_numpy(v['tokens'].astype(np.int64)), torch.from_numpy(v['subseg_idxs'].astype(np.int64)))
    valid = torch.from_numpy(v['tokens'] != 0)
    e = relerr(got[valid.to(dev)], ref[valid])
    assert got.shape == (160, 768) and e < 1e-3, e
    ls = pm.get_label_space(['making coffee', 'going backpacking'])
    probs = torch.softmax(100.0 * got[torch.from_numpy(v['tokens'] == M.MASK).to(dev)] @ ls.T, -1)
    assert probs.shape == (1, 2) and abs(float(probs.sum()) - 1.0) < 1e-5


def test_image_only_embedding_methods(dev):
    """embed_preencoded_noaudio / embed_singleimg_with_multiimg_prompt (M:848-904): vision-only joint encoding, against the
    oracle's composition of the same pieces, and against each other (a precomputed prompt + fresh images == all fresh)."""
    cfg, model, params = make_model(dev, False)
    model.bind(params)
    rng = np.random.default_rng(8)
    imgs = rng.random((4, 24, 768)).astype(np.float32)
    tokens = np.concatenate([rng.integers(10, 32768, size=20), [3], np.zeros(11, dtype=np.int64)]).astype(np.int32)
    subseg = np.concatenate([np.repeat(np.arange(4) * 3, 5), [9], -np.ones(11, dtype=np.int64)]).astype(np.int32)
    enc = model.get_imgseq_only(imgs)                                      # [4, 6, H]
    a = model.embed_preencoded_noaudio(enc, tokens, subseg)
    b = model.embed_singleimg_with_multiimg_prompt(enc[:2], imgs[2:], tokens, subseg)
    valid = torch.from_numpy(tokens != 0).to(dev)
    assert a.shape == (32, 128) and relerr(a[valid], b[valid]) < 1e-5
    # oracle composition
    C = R.Cfg(cfg)
    oenc = R.vision_transformer(params['vision_encoder'], torch.from_numpy(imgs), 2, 4, 6, 2)['seq_attnpool'].reshape(1, 24, 128)
    mm = R.prepare_multimodal_inputs(params, C, tokens=torch.from_numpy(tokens.astype(np.int64))[None],
                                     token_segment_idx=torch.div(torch.from_numpy(subseg.astype(np.int64))[None], 3, rounding_mode='floor'),
                                     vision_input=oenc)
    joint = R.transformer_encoder(params['joint_transformer'], mm['x'], 2, rotary_coords=mm['rotary_coords'], attention_mask=mm['attention_mask'])['seq']
    ref = R.unit_normalize(R.dense(joint[0, :32], params['head']))
    assert relerr(a[valid], ref[valid.cpu()]) < 1e-3


@pytest.mark.parametrize('bf16', [False, True])
def test_transformer_encoder_arbitrary_mask_and_learned_pe(dev, bf16):
    """The two branches of TransformerEncoder.__call__ no MERLOT Reserve encoder takes (mreserve/modeling.py:303, 335-341, 350-356): an attention_mask that
    is NOT of the block form valid x valid x same-source (here: causal, and a random one with an entirely masked row, which the reference's -1e10
    bias turns into a uniform row) -> the dense-mask attention kernel; and rotary_coords = None -> learned position embeddings `pe` added to x, no
    rotary scaling.  Against the oracle's encoder on the same weights."""
    cfg, model, params = make_model(dev, bf16)
    tol = 2e-2 if bf16 else 1e-3
    rng = np.random.default_rng(3)
    q = (lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()) if bf16 else (lambda a: a)
    N, L, H = 3, 37, 128
    x = q(rng.standard_normal((N, L, H)).astype(np.float32))
    coords = (M.get_rotary_coordinates(L, center_origin=False) / L).reshape(L, 1)
    enc = model.joint_transformer
    nl = cfg['model']['joint_num_layers']
    model.bind(params)
    # (a) causal mask and a random mask with an empty row: not expressible as one code per position
    causal = np.tril(np.ones((L, L), bool))[None].repeat(N, 0)
    rnd = rng.random((N, L, L)) < 0.6
    rnd[:, np.arange(L), np.arange(L)] = True
    rnd[1, 5, :] = False
    for mask in (causal, rnd):
        with pytest.raises(NotImplementedError):
            M._mask_to_code(mask)
        got = enc(x, rotary_coords=coords, attention_mask=mask)['seq']
        ref = R.transformer_encoder(params['joint_transformer'], torch.from_numpy(x), nl, rotary_coords=coords, attention_mask=torch.from_numpy(mask))['seq']
        assert got.shape == ref.shape and relerr(got, ref) < tol, relerr(got, ref)
    # a block-form mask still goes to the code-based kernels and agrees with the dense kernel on the same mask
    valid = rng.random((N, L)) < 0.8
    valid[:, 0] = True
    blockm = valid[:, :, None] & valid[:, None, :]
    a = enc(x, rotary_coords=coords, attention_mask=blockm)['seq']
    from merlot_reserve_amd import ops
    qkv = torch.randn(N * L, 3 * H, device=dev).to(a.dtype)
    o1, o2 = torch.zeros(N * L, H, device=dev, dtype=a.dtype), torch.zeros(N * L, H, device=dev, dtype=a.dtype)
    lse = torch.zeros(N, H // 64, L, device=dev)
    code = torch.from_numpy(M._mask_to_code(blockm).reshape(-1)).to(dev)
    ops.attention_fwd(qkv, code, o1, lse, N, L, H // 64)
    ops.attention_fwd_dense_mask(qkv, torch.from_numpy(blockm.astype(np.uint8)).to(dev), o2, N, L, H // 64)
    assert relerr(o2, o1) < (1e-2 if bf16 else 1e-5), relerr(o2, o1)
    # (b) learned position embeddings
    with pytest.raises(KeyError):
        enc(x, rotary_coords=None)
    g = torch.Generator().manual_seed(9)
    pe = torch.randn(L, H, generator=g) * 0.02
    if bf16:
        pe = pe.to(torch.bfloat16).float()
    params2 = jax_tree_map(lambda t: t, params)
    params2['joint_transformer'] = dict(params['joint_transformer'], pe=pe)
    model.bind(params2)
    got = model.joint_transformer(x, rotary_coords=None, is_valid=valid)['seq']
    ref = R.transformer_encoder(params2['joint_transformer'], torch.from_numpy(x), nl, rotary_coords=None, is_valid=torch.from_numpy(valid))['seq']
    assert relerr(got, ref) < tol, relerr(got, ref)
