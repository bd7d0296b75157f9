"""The reference's import names resolve to the build: demo/demo_video.py:7-50's call
sequence under `mreserve.*`, with random base weights, and the names pretrain/train.py and finetune/vcr import."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_demo_video_call_sequence_under_reference_import_names(dev):
    from mreserve.modeling import PretrainedMerlotReserve
    from mreserve.preprocess import MASK, preprocess_video
    from tests.test_modeling_gpu import FixtureTokenizer
    grid_size = (18, 32)
    model = PretrainedMerlotReserve.from_random('base', image_grid_size=grid_size, seed=0, device=dev, encoder=FixtureTokenizer())
    rng = np.random.default_rng(1)
    video_segments = []
    for i in range(8):                                  # demo_video.py:19-27 (decoded media: arrays instead of video_to_segments)
        spec = (rng.random((3, 60, 65)) * 5).astype(np.float32)
        spec[..., 64] = 1.0
        video_segments.append({'patches': rng.random((576, 768)).astype(np.float32), 'spectrogram': spec, 'use_text_as_input': False})
    video_segments[0]['text'] = "in this video i'll be<|MASK|>"
    video_segments[0]['use_text_as_input'] = True
    video_pre = preprocess_video(video_segments, output_grid_size=grid_size, encoder=model.encoder, verbose=False)
    out_h = model.embed_video(**video_pre)                                     # :33
    out_h = out_h[torch.from_numpy(video_pre['tokens'] == MASK).to(out_h.device)]
    options = ['making coffee', 'going backpacking']
    label_space = model.get_label_space(options)                               # :43
    logits = 100.0 * torch.einsum('bh,lh->bl', out_h, label_space)             # :46
    probs = torch.softmax(logits, -1)
    assert probs.shape == (1, 2) and torch.isfinite(probs).all() and abs(float(probs.sum()) - 1.0) < 1e-5
