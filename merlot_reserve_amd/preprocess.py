"""Integer / layout side of `mreserve.preprocess` (mreserve/preprocess.py:463-551): what turns decoded media into the
arrays `MerlotReserve.embed_video` takes.  Media decoding itself (ffmpeg, librosa mel spectrograms, TF image resizing:
preprocess.py:27-460) is out of scope: frames arrive here as float arrays already at the target resolution, spectrograms
as [3, 60, 65] arrays.

  patchify                       preprocess.py:478-479   tf.nn.space_to_depth(img, 16) + reshape -> [h*w, 768], patch vector
                                                          order (dy, dx, c) -- the same order as pretrain/dataloader.py:83-84
  preprocess_video               preprocess.py:482-551   token / sub-segment streams, bit-exact
"""
import numpy as np

from .planner import AUDIOSPAN

PADDING, START, END, MASK, MASKAUDIO, LTOVPOOL, RESETCTX = 0, 1, 2, 3, 4, 6, 9


_encoder = None


def __getattr__(name):
    """`from mreserve.preprocess import encoder` (preprocess.py:24): the tokenizer, loaded on first use from the vocabulary
    file the user supplies (modeling.get_encoder)."""
    global _encoder
    if name == 'encoder':
        if _encoder is None:
            from .modeling import get_encoder
            _encoder = get_encoder()
        return _encoder
    raise AttributeError(name)


def video_to_segments(*args, **kwargs):
    """mreserve/preprocess.py:352-460 decodes a video file with ffmpeg / librosa into frames and mel spectrograms: media
    decoding is outside this package -- pass decoded arrays to preprocess_video."""
    raise NotImplementedError('video decoding (ffmpeg / librosa) is out of scope: build the segment dicts from decoded frames '
                              "('frame' or 'patches') and spectrograms ('spectrogram' [3, 60, 65]) and call preprocess_video")


def patchify(img, output_grid_size, P=16):
    """img [h1*P, w1*P, 3] float -> [h1*w1, P*P*3]: space_to_depth in NHWC order (block row, block col, channel)."""
    h1, w1 = output_grid_size
    assert h1 <= 24 and w1 <= 32, "we didn't pretrain on anything bigger than 24x24 or 18x32"      # preprocess.py:474-475
    img = np.asarray(img, dtype=np.float32)
    assert img.shape == (h1 * P, w1 * P, 3), f'expected a {h1 * P}x{w1 * P}x3 frame, got {img.shape}'
    return img.reshape(h1, P, w1, P, 3).transpose(0, 2, 1, 3, 4).reshape(h1 * w1, P * P * 3)


def preprocess_video(video_segments, output_grid_size, encoder=None, verbose=False):
    """preprocess.py:482-551.  Each segment: 'frame' ([H, W, 3] float at output resolution, or 'patches' [h*w, 768]
    already patchified), 'spectrogram' [3, 60, 65], 'text' (str -- needs `encoder` -- or a list of token ids),
    'use_text_as_input' (default True).  Returns images [n, h*w, 768], audio_clips [3n, 60, 65], tokens [160] int32,
    subseg_idxs [160] int32."""
    if len(video_segments) > 8:
        raise ValueError('We only support videos of at most 8 segments right now')             # preprocess.py:497-498
    images = np.stack([np.asarray(s['patches'], dtype=np.float32) if 'patches' in s else patchify(s['frame'], output_grid_size)
                       for s in video_segments])
    subseg_idxs, audio_clips, tokens_out = [], [], []
    for i, segm_i in enumerate(video_segments):
        if segm_i.get('use_text_as_input', True):
            txt = segm_i.get('text', '')
            if isinstance(txt, str):
                if encoder is None:
                    try:
                        encoder = __getattr__('encoder')          # the module-level tokenizer, as in the reference
                    except FileNotFoundError as e:
                        raise ValueError('a tokenizer is needed for string text (pass encoder=..., set MRESERVE_TOKENIZER_JSON, '
                                         'or pass token ids)') from e
                txt_tok = encoder.encode(txt).ids
            else:
                txt_tok = list(txt)
            audio_clips.append(np.zeros([3, 60, 65], dtype=np.float32))       # dummy audio clip
            subseg_idxs.extend([i * 3] * len(txt_tok))
            tokens_out.extend(txt_tok)
        else:
            audio_clips.append(np.asarray(segm_i['spectrogram'], dtype=np.float32))
            tokens_out.extend([AUDIOSPAN] * 18)                               # 6 audio tokens per sub-segment
            subseg_idxs.extend((i * 3 + np.arange(18) // 6).tolist())
    if len(tokens_out) >= 160:
        if verbose:
            print(f'warning -- truncating tokens {len(tokens_out)} to be 160', flush=True)
        tokens_out, subseg_idxs = tokens_out[:160], subseg_idxs[:160]
    while len(tokens_out) < 160:
        tokens_out.append(0)
        subseg_idxs.append(-1)
    return {'images': images, 'audio_clips': np.stack(audio_clips).reshape(-1, 60, 65),
            'tokens': np.array(tokens_out, dtype=np.int32), 'subseg_idxs': np.array(subseg_idxs, dtype=np.int32)}
