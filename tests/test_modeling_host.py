"""Host side of the mreserve.modeling / mreserve.preprocess drop-ins (no GPU): rotary helpers against the hand-derived
known answers of SURVEY 8c, the token / sub-segment layout of preprocess_video (preprocess.py:482-551) against the
reference tokenizer's recorded ids, patch order, mask -> code conversion, and the reference's error conventions."""
import json
import os

import numpy as np
import pytest
import torch

from merlot_reserve_amd import modeling as M
from merlot_reserve_amd import preprocess as P
from merlot_reserve_amd.config import load_config
from oracle import ref_torch as R

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tokenizer_ids.json')))


class FixtureTokenizer:
    """Stands in for tokenizers.Tokenizer on the strings whose ids were recorded from the reference's tokenizer."""

    def __init__(self):
        self.pad = None

    class _Enc:
        def __init__(self, ids):
            self.ids = ids

    def encode(self, s):
        return self._Enc(list(FIX['encode'][s]))

    def enable_padding(self, pad_token, length):
        assert pad_token == '<|PAD|>'
        self.pad = length

    def no_padding(self):
        self.pad = None

    def encode_batch(self, opts):
        assert self.pad == 15
        return [self._Enc(list(FIX['encode_padded_15'][o])) for o in opts]


def test_rotary_coordinate_known_answers():
    assert M.get_rotary_coordinates(4).tolist() == [-2, -1, 1, 2]
    assert M.get_rotary_coordinates(5).tolist() == [-2, -1, 1, 2, 3]
    assert M.get_rotary_coordinates(3, center_origin=False).tolist() == [1, 2, 3]
    c = M.get_rotary_coordinates_2d(12, 20, dtype=np.float64)
    assert c.shape == (240, 2) and np.allclose(c[0], [-6 / 21, -10 / 21]) and np.allclose(c[1], [-6 / 21, -9 / 21])
    mm = M.multimodal_rotary_coords(segment_idx=np.array([[0, 8]]), token_idx=np.array([[1.0, 512.0]]))
    assert mm.shape == (1, 2, 4) and mm[0, 1].tolist() == [0.0, 0.0, 0.5, 0.5]


def test_rotary_scale_table_is_the_reference_rotary():
    """apply_rotary(x, construct_rotary_sinusoids(c)) == x * rotary_scale_table(c) on the first 32 dims (M:116-144)."""
    rng = np.random.default_rng(0)
    for nd in (1, 2, 4):
        coords = rng.uniform(-1, 1, size=(7, nd))
        sn = M.construct_rotary_sinusoids(coords)
        assert sn.shape == (2, 7, 32)
        qk = rng.standard_normal((7, 3, 64)).astype(np.float32)
        out = M.apply_rotary(qk, sn)
        tab = M.rotary_scale_table(coords)
        assert np.allclose(out[..., :32], qk[..., :32] * tab[:, None, :], atol=1e-6) and np.array_equal(out[..., 32:], qk[..., 32:])
        # and both agree with the oracle's restatement
        ref = R.apply_rotary(torch.from_numpy(qk), torch.as_tensor(R.construct_rotary_sinusoids(coords)).float())
        assert np.allclose(out, ref.numpy(), atol=1e-6)
    # theta = 0 -> [-x0, x1, -x2, ...]; theta = pi/2 -> x  
    assert np.allclose(M.rotary_scale_table(np.zeros((1, 1)))[0], np.tile([-1.0, 1.0], 16))


def test_mask_to_code_roundtrip_and_rejection():
    rng = np.random.default_rng(1)
    code = rng.integers(-1, 3, size=(4, 9)).astype(np.int32)
    mask = (code[:, :, None] == code[:, None, :]) & (code[:, :, None] >= 0)
    got = M._mask_to_code(mask)
    assert np.array_equal((got[:, :, None] == got[:, None, :]) & (got[:, :, None] >= 0), mask)
    causal = np.tril(np.ones((1, 5, 5), bool))
    with pytest.raises(NotImplementedError):
        M._mask_to_code(causal)


def test_patchify_is_space_to_depth():
    img = np.random.default_rng(2).random((2 * 16, 3 * 16, 3)).astype(np.float32)
    p = P.patchify(img, (2, 3))
    assert p.shape == (6, 768)
    for (i, j, dy, dx, c) in [(0, 0, 0, 0, 0), (0, 1, 2, 3, 1), (1, 2, 15, 15, 2), (1, 0, 7, 0, 1)]:
        assert p[i * 3 + j, (dy * 16 + dx) * 3 + c] == img[i * 16 + dy, j * 16 + dx, c]
    # same element order as torch's pixel_unshuffle generalised to NHWC (= tf.nn.space_to_depth)
    t = torch.from_numpy(img).permute(2, 0, 1)[None]
    ref = torch.nn.functional.unfold(t, kernel_size=16, stride=16)[0].T.reshape(6, 3, 16, 16).permute(0, 2, 3, 1).reshape(6, 768)
    assert np.array_equal(p, ref.numpy())


def test_preprocess_video_layout():
    enc = FixtureTokenizer()
    segs = [{'patches': np.zeros((6, 768)), 'text': "in this video i'll be<|MASK|>", 'use_text_as_input': True}]
    segs += [{'patches': np.ones((6, 768)), 'spectrogram': np.full((3, 60, 65), float(i)), 'use_text_as_input': False} for i in range(1, 8)]
    out = P.preprocess_video(segs, (2, 3), encoder=enc)
    assert out['images'].shape == (8, 6, 768) and out['audio_clips'].shape == (24, 60, 65)
    assert out['tokens'].dtype == np.int32 and out['subseg_idxs'].dtype == np.int32 and out['tokens'].shape == (160,)
    assert out['tokens'][:7].tolist() == [226, 246, 571, 187, 424, 238, 3] and out['subseg_idxs'][:7].tolist() == [0] * 7
    assert out['tokens'][7:7 + 18 * 7].tolist() == [5] * 126
    assert out['subseg_idxs'][7:25].tolist() == [3] * 6 + [4] * 6 + [5] * 6
    assert out['subseg_idxs'][7 + 18 * 6:7 + 18 * 7].tolist() == [21] * 6 + [22] * 6 + [23] * 6
    assert out['tokens'][133:].tolist() == [0] * 27 and out['subseg_idxs'][133:].tolist() == [-1] * 27
    assert float(out['audio_clips'][0].sum()) == 0.0 and float(out['audio_clips'][3, 0, 0]) == 1.0
    # truncation at 160 and the 8-segment limit
    long = P.preprocess_video([{'patches': np.zeros((6, 768)), 'text': list(range(10, 210))}], (2, 3))
    assert long['tokens'].tolist() == list(range(10, 170)) and long['subseg_idxs'].tolist() == [0] * 160
    with pytest.raises(ValueError):
        P.preprocess_video(segs + segs[:1], (2, 3), encoder=enc)
    # subseg // 3 floors: the padding pointer -1 stays -1 (M:836)
    assert (out['subseg_idxs'] // 3)[-1] == -1


def test_reference_error_conventions():
    with pytest.raises(ValueError):
        M.PretrainedMerlotReserve.from_pretrained('huge')
    with pytest.raises(ValueError):
        M.PretrainedMerlotReserve.from_pretrained('base', image_grid_size=(10, 10))
    with pytest.raises(FileNotFoundError):
        M.PretrainedMerlotReserve.from_pretrained('base', image_grid_size=(12, 20), cache_dir='/nonexistent_dir_xyz')
    model = M.MerlotReserve.from_config(load_config('base'), device='cpu')
    assert model.hidden_size == 768 and model.audio_encoder.pooling_ratio == 5 and model.dtype == torch.bfloat16
    assert model.data['seq_len'] == 640                                    # config['data'] nested under 'data' (M:587)
    with pytest.raises(NotImplementedError):
        model({})
    with pytest.raises(RuntimeError):
        model.joint_proj(torch.zeros(1, 768))                              # no parameters bound
    enc = model.joint_transformer
    with pytest.raises(ValueError):
        M.TransformerEncoder(model, 'span_encoder/transformer', 768, 1, add_cls_token=True)(
            torch.zeros(1, 4, 768), attention_mask=np.ones((1, 4, 4), bool), rotary_coords=np.zeros((4, 1)))
    with pytest.raises(ValueError):
        enc(torch.zeros(1, 4, 768), attention_mask=np.ones((1, 4, 4), bool), is_valid=np.ones((1, 4), bool), rotary_coords=np.zeros((4, 1)))
    pm = M.PretrainedMerlotReserve(encoder=FixtureTokenizer(), params={}, model=model)
    with pytest.raises(ValueError):
        pm.no_such_method
    assert pm.embed_video is pm.embed_video                                # method cache (M:1012-1015)
