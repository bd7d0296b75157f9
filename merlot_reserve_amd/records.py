"""Real-data input of the pretraining / resolution-adaptation drivers without TensorFlow: the reference's
`pretrain/dataloader.py:449-789` (TFRecord shards -> `dataset_parser` -> `handle_batch`) restated on numpy + PIL, with the
container format read by the library's native code (csrc/hostio.cpp: mr_tfrecord_scan, CRC-32C).

    shard file --mr_tfrecord_scan--> record bytes --parse_example--> {feature name: values}
               --dataset_parser(config, rng)--> one record's features (images, audio_clips, the four token streams, text_spans ...)
               --handle_batch--> the per-device batch dict of pretrain/dataloader.py:732-789 = what synthetic.make_batch emits,
                                 what planner.build_plan / Trainer.train_step / loader.PrefetchLoader consume

What is the reference's and what is not.  The record layout (feature names and types, dataloader.py:29-59), every integer rule of the
masking pipeline (token -> audio-span assignment, reassign_empty_tokens, increase_textmask, mask_tokens, select_tokens, the matching and
random-text streams, pointer columns) and the float rules of load_audio / resize_and_pad follow the cited lines.  The RANDOM DRAWS come from
a numpy Generator instead of TensorFlow's stateful ops: same distributions, different streams (the reference reseeds from the wall clock,
dataloader.py:921-923, so no stream was ever reproducible).  Image resampling uses PIL's filters where the reference draws one of TensorFlow's
eight `ResizeMethod`s with antialiasing (data_utils.py:116-125): BOX for AREA, HAMMING for GAUSSIAN, LANCZOS for both Lanczos radii, BICUBIC
for Mitchell -- augmentation noise, not arithmetic the model is checked against.  JPEG decoding is libjpeg in both.  `TOKEN_IS_VALID`
(dataloader.py:381-386: which vocabulary entries decode to [ A-Za-z0-9']*) needs the tokenizer's vocabulary file, a data asset of the
reference release: pass `token_is_valid` (or an encoder) -- without it every id > 10 counts as valid.

Shard paths are local files (the reference's configs name gs:// objects, which TensorFlow's file-system layer resolves: mount or copy them).

Parity note: the reference holds no recorded batch, so the parser is pinned by structure only (tests/test_records.py: the invariants mask_tokens
guarantees, hand-derived cases, acceptance by the planner and a training step).  The CONTAINER is pinned independently: CRC-32C by the RFC 3720 vectors,
the tf.train.Example codec in both directions against the protobuf runtime (the published .proto rebuilt from descriptors); DESIGN.md section 4.
"""
import io
import os
import struct

import numpy as np

# special token ids (mreserve/lowercase_encoder.py; the same constants as synthetic.py -- restated so that a parser worker process imports neither torch nor the library)
PADDING, START, END, MASK, MASKAUDIO, AUDIOSPAN, LTOVPOOL = 0, 1, 2, 3, 4, 5, 6

# encoder.encode('title:' / 'description:' / 'tags:').ids (dataloader.py:633-636; recorded from the reference tokenizer: tests/golden/tokenizer_ids.json)
TITLE_IDS, DESCRIPTION_IDS, TAGS_IDS = [3388, 35], [2026, 35], [10884, 35]

# per-segment features (dataloader.py:29-59): name -> kind ('bytes' scalar, 'int' scalar, 'float' scalar, 'ints' / 'floats' variable length)
SEGMENT_FEATURES = {
    'image/encoded': 'bytes', 'image/format': 'bytes', 'image/key/sha256': 'bytes', 'image/height': 'int', 'image/width': 'int',
    'spectrogram/encoded': 'bytes', 'spectrogram/format': 'bytes', 'spectrogram/key/sha256': 'bytes', 'spectrogram/height': 'int',
    'spectrogram/width': 'int', 'spectrogram/magic_number': 'float',
    'youtube_id': 'bytes', 'video_src_index': 'int',
    'title': 'ints', 'tags': 'ints', 'description': 'ints', 'meta': 'bytes',
    'playback_speed': 'ints', 'start_time': 'float', 'end_time': 'float',
    'tok_ids': 'ints', 'tok_start_times': 'floats', 'tok_end_times': 'floats', 'random_text': 'ints',
}
_DEFAULTS = {'image/format': b'jpeg', 'spectrogram/format': b'jpeg'}


# ------------------------------------------------------------------------------------------------ container: TFRecord framing
def _lib():
    from . import _lib as L
    return L.load()


def read_tfrecord(path_or_bytes, verify=True):
    """All records of one shard (a path, or the file's bytes) as a list of `bytes`.  Both checksums of every record are checked by the
    native scanner unless verify=False; a truncated or corrupted shard raises ValueError with the record number."""
    import ctypes as C
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        buf = np.frombuffer(path_or_bytes, dtype=np.uint8)
    else:
        buf = np.fromfile(path_or_bytes, dtype=np.uint8)
    lib = _lib()
    ptr = buf.ctypes.data_as(C.c_void_p)
    n = lib.mr_tfrecord_scan(ptr, buf.size, None, None, 0, 1 if verify else 0)       # pass 1: count (and check); pass 2: offsets
    if n < 0:
        raise ValueError(lib.mr_last_error().decode())
    offs, lens = np.zeros(max(n, 1), dtype=np.int64), np.zeros(max(n, 1), dtype=np.int64)
    n = lib.mr_tfrecord_scan(ptr, buf.size, offs.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), n, 0)
    if n < 0:
        raise ValueError(lib.mr_last_error().decode())
    return [buf[o:o + l].tobytes() for o, l in zip(offs[:n].tolist(), lens[:n].tolist())]


def iter_tfrecord(path, verify=True):
    """The records of one shard ONE AT A TIME (a shard of the corpus holds ~2 400 records of ~1.3 MB: make_dataset keeps several shards open and must not
    hold them whole).  Same checks as read_tfrecord, record by record."""
    lib = _lib()
    with open(path, 'rb') as f:
        n = 0
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError(f'{path}: truncated header of record {n}')
            length, lcrc = struct.unpack('<QI', head)
            if verify and lib.mr_crc32c_masked(head[:8], 8) != lcrc:
                raise ValueError(f'{path}: record {n}: corrupted length field')
            body = f.read(length + 4)
            if len(body) < length + 4:
                raise ValueError(f'{path}: record {n}: {length} data bytes run past the end of the file')
            data = body[:length]
            if verify and lib.mr_crc32c_masked(data, length) != struct.unpack('<I', body[length:])[0]:
                raise ValueError(f'{path}: record {n}: data checksum mismatch')
            yield data
            n += 1


def write_tfrecord(path, records):
    """The inverse (fixtures, re-sharding): length | masked crc | data | masked crc per record."""
    lib = _lib()
    with open(path, 'wb') as f:
        for rec in records:
            head = struct.pack('<Q', len(rec))
            f.write(head + struct.pack('<I', lib.mr_crc32c_masked(head, 8)) + rec + struct.pack('<I', lib.mr_crc32c_masked(rec, len(rec))))


# ------------------------------------------------------------------------------------------------ container: tf.train.Example (protobuf wire format)
# Example { Features features = 1 }   Features { map<string, Feature> feature = 1 }   Feature { oneof: BytesList bytes_list = 1,
# FloatList float_list = 2, Int64List int64_list = 3 }   *List { repeated value = 1 (floats / ints usually packed) }
def _varint(b, i):
    x, s = 0, 0
    while True:
        c = b[i]
        i += 1
        x |= (c & 0x7f) << s
        if c < 0x80:
            return x, i
        s += 7


def _fields(b):
    """(field number, wire type, value) triples of one message; length-delimited values as memoryviews."""
    b = memoryview(b)
    i, n = 0, len(b)
    while i < n:
        key, i = _varint(b, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(b, i)
        elif wt == 2:
            ln, i = _varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield fno, wt, v


def _int64(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def _decode_feature(msg):
    for fno, wt, v in _fields(msg):
        if fno == 1:                                   # BytesList
            return [bytes(x) for f, _w, x in _fields(v) if f == 1]
        if fno == 2:                                   # FloatList: packed (wire type 2) or one fixed32 per value
            out = []
            for f, w, x in _fields(v):
                if f == 1:
                    out.append(np.frombuffer(x, dtype='<f4'))
            return np.concatenate(out).astype(np.float32) if out else np.zeros(0, np.float32)
        if fno == 3:                                   # Int64List: packed varints, or one varint per value
            out = []
            for f, w, x in _fields(v):
                if f != 1:
                    continue
                if w == 0:
                    out.append(_int64(x))
                else:
                    i = 0
                    while i < len(x):
                        y, i = _varint(x, i)
                        out.append(_int64(y))
            return np.asarray(out, dtype=np.int64)
    return []                                          # a Feature with no list set


def parse_example(record):
    """{feature name: list of bytes | float32 array | int64 array} of one serialised tf.train.Example."""
    out = {}
    for fno, _wt, feats in _fields(record):
        if fno != 1:
            continue
        for f2, _w2, entry in _fields(feats):
            if f2 != 1:
                continue
            key, val = None, []
            for f3, _w3, x in _fields(entry):
                if f3 == 1:
                    key = bytes(x).decode()
                elif f3 == 2:
                    val = _decode_feature(x)
            out[key] = val
    return out


def _enc_varint(x):
    x &= (1 << 64) - 1
    out = bytearray()
    while True:
        if x < 0x80:
            out.append(x)
            return bytes(out)
        out.append((x & 0x7f) | 0x80)
        x >>= 7


def _ld(fno, payload):
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example(features):
    """Serialise {name: bytes | list of bytes | int / float scalars or arrays} as a tf.train.Example (floats / ints packed, like TensorFlow's writer)."""
    entries = b''
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        if isinstance(v, (list, tuple)) and len(v) > 0 and isinstance(v[0], (bytes, bytearray)):
            feat = _ld(1, b''.join(_ld(1, bytes(x)) for x in v))
        else:
            a = np.atleast_1d(np.asarray(v))
            if a.dtype.kind == 'f':
                feat = _ld(2, _ld(1, a.astype('<f4').tobytes()) if a.size else b'')
            else:
                feat = _ld(3, _ld(1, b''.join(_enc_varint(int(x)) for x in a.tolist())) if a.size else b'')
        entries += _ld(1, _ld(1, name.encode()) + _ld(2, feat))
    return _ld(1, entries)


# ------------------------------------------------------------------------------------------------ random helpers (pretrain/data_utils.py:263-318)
def uniform_random_select(rng, n, num_samples, sort_idx=True):
    assert num_samples <= n
    idx = np.argsort(rng.random(n), kind='stable')[:num_samples]
    return (np.sort(idx) if sort_idx else idx).astype(np.int32)


def random_categorical_without_replacement(rng, logits, num_samples):
    """Gumbel top-k (data_utils.py:281-290)."""
    logits = np.asarray(logits, dtype=np.float64)
    if num_samples > logits.shape[0]:
        raise ValueError(f'cannot draw {num_samples} distinct items out of {logits.shape[0]}')
    z = -np.log(-np.log(rng.uniform(1e-300, 1.0, size=logits.shape)))
    return np.argsort(-(logits + z), kind='stable')[:num_samples].astype(np.int32)


def cumulative_maximum_int(x):
    return np.maximum.accumulate(np.asarray(x, dtype=np.int32)) if len(x) else np.asarray(x, dtype=np.int32)


# ------------------------------------------------------------------------------------------------ media (dataloader.py:63-128, data_utils.py:25-147)
_PIL_METHODS = None


def _resize_f32(img, height, width, method):
    """[H, W, C] float32 -> [height, width, C] with PIL's antialiased filters on float planes (no 8-bit round trip).  A uint8 image takes PIL's 8-bit path
    (its SIMD resampler, ~4x faster) and comes back as float32 / 255: the `fast_image_resize` option, for batches that are rounded to bf16 anyway (one
    8-bit step is 1 / 255, a bf16 step between 0.5 and 1 is 1 / 256)."""
    from PIL import Image
    if img.dtype == np.uint8:
        return np.asarray(Image.fromarray(img).resize((width, height), method), dtype=np.float32) / np.float32(255.0)
    planes = [np.asarray(Image.fromarray(np.ascontiguousarray(img[:, :, c]), mode='F').resize((width, height), method), dtype=np.float32)
              for c in range(img.shape[2])]
    return np.stack(planes, -1)


def flip_if_vertical(image):
    """data_utils.py:25-39: a portrait frame (height >= 4/3 width) is rotated by 90 degrees (counter-clockwise, tf.image.rot90) and padded by four
    columns of 0.5 on either side."""
    h, w = image.shape[:2]
    if h >= 4.0 * w / 3.0:
        image = np.pad(np.rot90(image), [(0, 0), (4, 4), (0, 0)], mode='constant', constant_values=128 if image.dtype == np.uint8 else 0.5)
    return image


def resize_and_pad(image, desired_output_size, rng, random_scale_min=0.1, random_scale_max=2.0, do_random_scale=False, shrink_both_sides=True,
                   do_flip_if_vertical=True, resize_method='random', keep_u8=False):
    """data_utils.py:42-147 on a float32 [H, W, 3] image in [0, 1].  Returns (image [dh, dw, 3], image_info [7]).
    keep_u8 (a uint8 image, the 8-bit resampling path): the result stays uint8 -- resampled, cropped and zero-padded bytes; the caller scales by
    1 / 255 (or maps bytes to bf16 through a 256-entry table) once, on the final [dh, dw, 3] pixels only."""
    from PIL import Image
    if do_flip_if_vertical:
        image = flip_if_vertical(image)
    dh, dw = int(desired_output_size[0]), int(desired_output_size[1])
    height, width = float(image.shape[0]), float(image.shape[1])
    f32 = np.float32
    if do_random_scale:
        factor = f32(rng.uniform(random_scale_min, random_scale_max))
        if not shrink_both_sides:
            factor = min(max(f32(dw) / f32(width), f32(dh) / f32(height)), factor)
        scaled_y, scaled_x = int(factor * f32(dh)), int(factor * f32(dw))
        scale = min(f32(scaled_x) / f32(width), f32(scaled_y) / f32(height))
        scale = max(scale, f32(64.0) / f32(min(height, width)))
        scaled_height, scaled_width = int(f32(height) * scale), int(f32(width) * scale)
        offset_y = int(max(0.0, float(scaled_height - dh)) * rng.uniform(0, 1))
        offset_x = int(max(0.0, float(scaled_width - dw)) * rng.uniform(0, 1))
    else:
        scale = min(f32(dw) / f32(width), f32(dh) / f32(height))
        scaled_height, scaled_width = int(f32(height) * scale), int(f32(width) * scale)
        offset_y = offset_x = 0
    if resize_method == 'random' and do_random_scale:
        # sorted(tf.image.ResizeMethod): AREA, BICUBIC, BILINEAR, GAUSSIAN, LANCZOS3, LANCZOS5, MITCHELLCUBIC, NEAREST_NEIGHBOR
        R = Image.Resampling
        method = [R.BOX, R.BICUBIC, R.BILINEAR, R.HAMMING, R.LANCZOS, R.LANCZOS, R.BICUBIC, R.NEAREST][int(rng.integers(0, 8))]
    else:
        method = Image.Resampling.BILINEAR
    if keep_u8 and image.dtype == np.uint8:
        image = np.asarray(Image.fromarray(image).resize((max(scaled_width, 1), max(scaled_height, 1)), method))     # (8-bit output: already inside [0, 255])
    else:
        image = _resize_f32(image if image.dtype == np.uint8 else np.asarray(image, dtype=np.float32), max(scaled_height, 1), max(scaled_width, 1), method)
        image = np.clip(image, 0.0, 1.0)
    image = image[offset_y:offset_y + dh, offset_x:offset_x + dw]
    out = np.zeros((dh, dw, image.shape[2]), dtype=image.dtype)         # pad_to_bounding_box(image, 0, 0, dh, dw)
    out[:image.shape[0], :image.shape[1]] = image
    info = np.array([min(scaled_height, dh) / dh, min(scaled_width, dw) / dw, 1.0 / scale, height, width, offset_y / height, offset_x / width],
                    dtype=np.float32)
    return out, info


_U8_TO_BF16 = None


def u8_to_bf16_bits(x):
    """uint8 pixels -> the bf16 bit patterns of float32(v) / 255 (what bf16_bits(x / 255) gives), through a 256-entry table: the 8-bit image path
    never materialises float32 frames (16 frames of 192 x 320 x 3 per record: 11.8 MB of floats and their rounding pass, ~13 ms of a 97-ms record)."""
    global _U8_TO_BF16
    if _U8_TO_BF16 is None:
        t = bf16_bits(np.arange(256, dtype=np.float32) / np.float32(255.0)).astype(np.uint32)
        pair = np.arange(65536, dtype=np.uint32)                  # two pixels per lookup: a little-endian uint16 holds (first, second) as (low, high) byte
        _U8_TO_BF16 = (t, t[pair & 0xff] | (t[pair >> 8] << np.uint32(16)))
    t, t2 = _U8_TO_BF16
    x = np.ascontiguousarray(x)
    if x.size % 2 == 0 and x.dtype == np.uint8:
        return t2[x.reshape(-1).view(np.uint16)].view(np.uint16).reshape(x.shape)
    return t[x].astype(np.uint16)


def load_and_resize_img(encoded_jpg, config, rng, as_bf16_bits=False):
    """dataloader.py:63-85: encoded JPEG -> [(H / P) (W / P), P P 3] patches in space_to_depth order (dy, dx, c).
    data.fast_image_resize: the frame is resampled as 8-bit RGB (PIL's SIMD resampler, one call instead of three float planes) and only the final
    patches are scaled -- to float32, or with as_bf16_bits straight to bf16 bit patterns (uint16)."""
    from PIL import Image
    P = config['vit_patch_size']
    h1, w1 = config['output_grid']
    fast = config.get('fast_image_resize', False)
    im = Image.open(io.BytesIO(encoded_jpg))
    if fast:
        # DCT-domain downscale while decoding, when the stored frame is at least twice the largest size the random scale can ask for (never for the
        # corpus' 288 x 512 frames at the 192 x 320 grid; a no-op then)
        smax = config.get('random_scale_max', 1.1) if config.get('do_random_scale', True) else 1.0
        im.draft('RGB', (int(w1 * P * smax) + 1, int(h1 * P * smax) + 1))
    img = np.asarray(im.convert('RGB'))
    if not fast:                                             # the reference resamples floats (convert_image_dtype before resize_and_pad)
        img = img.astype(np.float32) / np.float32(255.0)
    img, _info = resize_and_pad(img, (h1 * P, w1 * P), rng, do_random_scale=config.get('do_random_scale', True),
                                random_scale_max=config.get('random_scale_max', 1.1), random_scale_min=config.get('random_scale_min', 1.05),
                                shrink_both_sides=config.get('shrink_both_sides', True), do_flip_if_vertical=config.get('do_flip_if_vertical', True),
                                resize_method='random', keep_u8=fast)
    img = np.ascontiguousarray(img.reshape(h1, P, w1, P, 3).transpose(0, 2, 1, 3, 4)).reshape(h1 * w1, P * P * 3)
    if img.dtype == np.uint8:
        return u8_to_bf16_bits(img) if as_bf16_bits else img.astype(np.float32) / np.float32(255.0)
    return bf16_bits(img) if as_bf16_bits else img


def load_audio(encoded_audio, magic_number, playback_speed, config, rng):
    """dataloader.py:88-128: the spectrogram travels as a grayscale JPEG [num_mels, spec_size]; num_audio_subsegments windows of audio_seq_length hops
    are cut at random offsets that keep their order, divided by the record's magic number (inverse of the 8-bit scaling), and get the playback
    speed as a 65th feature.  Returns (audio [nsub, T, num_mels + 1], start_t [nsub], end_t [nsub])."""
    from PIL import Image
    img = np.asarray(Image.open(io.BytesIO(encoded_audio)).convert('L'))
    if img.shape != (config['num_mels'], config['spec_size']):
        raise ValueError(f"spectrogram is {img.shape}, the config says {(config['num_mels'], config['spec_size'])}")
    img = img.T
    nsub, T = config['num_audio_subsegments'], config['audio_seq_length']
    content_len = nsub * T
    assert content_len < config['spec_size']
    paddings = rng.uniform(0.0, 1.0, size=nsub + 1).astype(np.float32)
    num_pad = config['spec_size'] - content_len
    paddings_int = (np.float32(num_pad) * np.cumsum(paddings / paddings.sum(), dtype=np.float32)).astype(np.int32)
    start_idx = paddings_int[:nsub] + np.arange(nsub, dtype=np.int32) * T
    seqs = np.stack([img[s:s + T] for s in start_idx]).astype(np.float32) / np.float32(magic_number)
    seqs = np.concatenate([seqs, np.full((nsub, T, 1), np.float32(playback_speed), dtype=np.float32)], -1)
    fft_window = config['fft_window_size'] / config['sample_rate']
    fft_to_time = config['fft_hop_length'] / config['sample_rate']
    start_t = start_idx.astype(np.float32) * np.float32(fft_to_time) - np.float32(fft_window / 2.0)
    end_t = start_t + np.float32(T * fft_to_time) + np.float32(fft_window)
    return seqs, start_t, end_t


# ------------------------------------------------------------------------------------------------ ragged token rows
class Ragged:
    """N rows of int32 tokens as (values, row_lengths): tf.RaggedTensor of rank 2, as far as the masking pipeline uses it."""

    def __init__(self, values, row_lengths):
        self.values = np.asarray(values, dtype=np.int32)
        self.row_lengths = np.asarray(row_lengths, dtype=np.int64)
        assert self.row_lengths.sum() == len(self.values) and (self.row_lengths >= 0).all()

    @classmethod
    def from_value_rowids(cls, values, rowids, nrows):
        rowids = np.asarray(rowids, dtype=np.int64)
        assert len(rowids) == 0 or ((np.diff(rowids) >= 0).all() and rowids[-1] < nrows and rowids[0] >= 0)
        return cls(values, np.bincount(rowids, minlength=nrows))

    @property
    def nrows(self):
        return len(self.row_lengths)

    def rowids(self):
        return np.repeat(np.arange(self.nrows), self.row_lengths)

    def rows(self):
        ends = np.cumsum(self.row_lengths)
        return [self.values[e - l:e] for e, l in zip(ends, self.row_lengths)]

    @classmethod
    def from_rows(cls, rows):
        return cls(np.concatenate([np.asarray(r, dtype=np.int32) for r in rows]) if rows else np.zeros(0, np.int32), [len(r) for r in rows])


def _one_hot(idx, N):
    m = np.zeros(N, dtype=bool)
    m[np.asarray(idx, dtype=np.int64)] = True
    return m


def shift_ragged_tokens_at_positions(tokens, positions, right_to_left=True):
    """dataloader.py:260-283: every listed NON-EMPTY row hands one token to a neighbour -- its first token to the row on its left
    (right_to_left) or its last token to the row on its right.  Only row lengths change."""
    N = tokens.nrows
    amt = (_one_hot(positions, N) & (tokens.row_lengths > 0)).astype(np.int64)
    if right_to_left:
        take = amt[1:]
        delta = np.concatenate([[0], -take]) + np.concatenate([take, [0]])
    else:
        take = amt[:-1]
        delta = np.concatenate([-take, [0]]) + np.concatenate([[0], take])
    return Ragged(tokens.values, tokens.row_lengths + delta)


def _both_directions(rng, f, tokens, **kw):
    """dataloader.py:286-296: right-to-left then left-to-right, or the other way round, with probability 1/2 each."""
    first = bool(rng.random() < 0.5)
    return f(f(tokens, right_to_left=first, **kw), right_to_left=not first, **kw)


def _reassign_empty_tokens(tokens, *, mask_idx, right_to_left):
    """dataloader.py:299-322: a masked row without tokens takes one from an unmasked neighbour that holds at least two."""
    N = tokens.nrows
    masked = _one_hot(mask_idx, N)
    needs = masked & (tokens.row_lengths == 0)
    can_donate = ~masked & (tokens.row_lengths >= 2)
    if right_to_left:
        return shift_ragged_tokens_at_positions(tokens, np.nonzero(can_donate[1:] & needs[:-1])[0] + 1, right_to_left=True)
    return shift_ragged_tokens_at_positions(tokens, np.nonzero(can_donate[:-1] & needs[1:])[0], right_to_left=False)


def reassign_empty_tokens(rng, tokens, mask_idx):
    return _both_directions(rng, _reassign_empty_tokens, tokens, mask_idx=mask_idx)


def _increase_textmask(tokens, *, mask_idx, tok_centroids_vals, audio_start_end, right_to_left, delta_thresh):
    """dataloader.py:325-377: a masked row takes the nearest token of an unmasked neighbour holding at least two when that token's centre lies
    within delta_thresh seconds of the masked span's audio boundary."""
    N = tokens.nrows
    mask_idx = np.asarray(mask_idx, dtype=np.int64)
    ends = np.cumsum(tokens.row_lengths)
    starts = ends - tokens.row_lengths
    cent = np.asarray(tok_centroids_vals, dtype=np.float32)
    masked = _one_hot(mask_idx, N)
    if right_to_left:
        nb = mask_idx + 1
        inside = nb < N
        nbc = np.minimum(nb, N - 1)
        t_out = np.array([cent[starts[r]:ends[r]].min() if tokens.row_lengths[r] > 1 else 10000.0 for r in nbc], dtype=np.float32)
        t_out = np.where(inside, t_out, np.float32(10000.0))
        delta = t_out - np.asarray(audio_start_end, dtype=np.float32)[mask_idx, 1]
        take = (delta < delta_thresh) & inside & ~masked[nbc]
        return shift_ragged_tokens_at_positions(tokens, nb[take], right_to_left=True)
    nb = mask_idx - 1
    inside = nb >= 0
    nbc = np.maximum(nb, 0)
    t_out = np.array([cent[starts[r]:ends[r]].max() if tokens.row_lengths[r] > 1 else -10000.0 for r in nbc], dtype=np.float32)
    t_out = np.where(inside, t_out, np.float32(-10000.0))
    delta = np.asarray(audio_start_end, dtype=np.float32)[mask_idx, 0] - t_out
    take = (delta < delta_thresh) & inside & ~masked[nbc]
    return shift_ragged_tokens_at_positions(tokens, nb[take], right_to_left=False)


def increase_textmask(rng, tokens, mask_idx, tok_centroids_vals, audio_start_end, delta_thresh=0.1):
    return _both_directions(rng, _increase_textmask, tokens, mask_idx=mask_idx, tok_centroids_vals=tok_centroids_vals,
                            audio_start_end=audio_start_end, delta_thresh=delta_thresh)


def pad_tokens_to_fixed_size(tokens, padded_seq_len):
    """dataloader.py:131-142: rows [0, -1, -1] appended, then truncated."""
    out = np.zeros((padded_seq_len, 3), dtype=np.int32)
    out[:, 1:] = -1
    t = np.asarray(tokens, dtype=np.int32).reshape(-1, 3)[:padded_seq_len]
    out[:len(t)] = t
    return out


def select_tokens(rng, tokens, padded_seq_len, num_segments):
    """dataloader.py:150-189: an over-long stream loses unmasked text at both ends first (never a MASK, and AUDIOSPAN only on the right), then
    keeps padded_seq_len rows drawn without replacement -- every MASK, and otherwise whole segments together (one random score per segment)."""
    L = len(tokens)
    amt = L - padded_seq_len
    is_mask = np.cumsum((tokens[:, 0] == MASK) | (tokens[:, 0] == MASKAUDIO))
    is_audiospan = np.cumsum(tokens[:, 0] == AUDIOSPAN)
    lhs_amt = int(((is_mask == 0) & (is_audiospan == 0)).sum())
    rhs_amt = int((is_mask == is_mask[-1]).sum()) - 1
    trunc_start = min(amt // 2, lhs_amt)
    trunc_end = min(amt - trunc_start, rhs_amt)
    trunc_start = min(amt - trunc_end, lhs_amt)
    tokens0 = tokens[trunc_start:L - trunc_end]
    if len(tokens0) <= padded_seq_len:
        return tokens0
    keep_logits = 1e7 * (tokens0[:, 0] == MASK).astype(np.float64)
    keep_logits += rng.uniform(-1e5, 1e5, size=num_segments)[tokens0[:, 1]]
    return tokens0[np.sort(random_categorical_without_replacement(rng, keep_logits, padded_seq_len))]


def mask_tokens(rng, tokens, mask_idx, do_audio_span=None, audio_token_length=6, text_span_start_counter=0, num_groups=1, padded_seq_len=None,
                do_audio_mask=False):
    """dataloader.py:192-257.  Returns (text spans = the original rows at the sorted mask indices, a list of num_groups [<= L, 3] streams with
    columns token id / audio-span pointer (the row index) / text-span pointer (rank of the masked row + text_span_start_counter, else -1))."""
    N = tokens.nrows
    mask_idx = np.sort(np.asarray(mask_idx, dtype=np.int64))
    rows = tokens.rows()
    text_spans = [rows[r].copy() for r in mask_idx]
    masked = _one_hot(mask_idx, N)
    if do_audio_span is not None:
        do_audio_span = np.asarray(do_audio_span, dtype=bool) & ~masked
        rows = [np.full(audio_token_length, AUDIOSPAN, dtype=np.int32) if do_audio_span[r] else rows[r] for r in range(N)]
    mask_tok = np.array([MASK, MASKAUDIO] if do_audio_mask else [MASK], dtype=np.int32)
    rows = [mask_tok if masked[r] else rows[r] for r in range(N)]
    text_ptr = np.where(masked, np.cumsum(masked) - 1 + text_span_start_counter, -1).astype(np.int32)
    grp = N // num_groups
    out = []
    for i in range(num_groups):
        rs = range(i * grp, (i + 1) * grp)
        vals = np.concatenate([rows[r] for r in rs]) if grp else np.zeros(0, np.int32)
        ptr = np.concatenate([np.full(len(rows[r]), r, dtype=np.int32) for r in rs]) if grp else np.zeros(0, np.int32)
        stream = np.stack([vals, ptr, text_ptr[ptr]], -1).astype(np.int32)
        if padded_seq_len is not None:
            stream = select_tokens(rng, stream, padded_seq_len, num_segments=N) if len(stream) > padded_seq_len else stream
            stream = pad_tokens_to_fixed_size(stream, padded_seq_len)
        out.append(stream)
    return text_spans, out


RAWTEXT_WEIGHTS_V1 = [0.0210583, 0.03984984, 0.06506665, 0.09467365, 0.12138153, 0.13305461, 0.12973022, 0.11296043, 0.09024, 0.06730134, 0.04789645,
                      0.03232633, 0.02123288, 0.01397406, 0.00925371]
RAWTEXT_WEIGHTS_V2 = [0.03233136, 0.05236081, 0.08763368, 0.11757072, 0.13737426, 0.13717706, 0.12541218, 0.10262764, 0.0771088, 0.05364242, 0.0342899,
                      0.0203823, 0.01177542, 0.00664939, 0.00366406]


def convert_rawtext_into_fake_segments(rng, tokens, desired_len, span_budget, use_v1_stats=False):
    """dataloader.py:400-447: plain text cut into pseudo 'audio segments' whose lengths follow the measured distribution of the video streams
    (1 + categorical over 15 weights).  Returns (Ragged rows, the unused tokens on the left, on the right)."""
    w = np.asarray(RAWTEXT_WEIGHTS_V1 if use_v1_stats else RAWTEXT_WEIGHTS_V2, dtype=np.float64)
    ev = float((np.arange(len(w)) * w).sum()) + 1
    L = min(desired_len + int((ev * 0.85 - 1) * span_budget), len(tokens))
    segm_lens = rng.choice(len(w), size=L, p=w / w.sum()).astype(np.int64) + 1
    segm_lens = segm_lens[np.cumsum(segm_lens) <= L]
    l_sel = int(segm_lens.sum())
    wiggle = len(tokens) - l_sel
    off = int(rng.integers(0, max(wiggle, 1)))
    tokens = np.asarray(tokens, dtype=np.int32)
    return Ragged(tokens[off:off + l_sel], segm_lens), tokens[:off], tokens[off + l_sel:]


def filter_out_tokens_not_in_youtube(spans, token_is_valid):
    return [s[token_is_valid[s]] for s in spans]


def make_token_is_valid(encoder=None, vocab_size=32768):
    """dataloader.py:381-386.  With the reference tokenizer: ids > 10 whose text is made of [ A-Za-z0-9'] only, minus seven listed ids.  Without
    it (its vocabulary file is not redistributed): every id > 10."""
    valid = np.arange(vocab_size) > 10
    if encoder is not None:
        import re
        ok = re.compile(r"^[ A-Za-z0-9']*$")
        valid = np.array([(i > 10) and bool(ok.match(encoder.decode([i]))) for i in range(encoder.get_vocab_size())])
    for i in (149, 4858, 9504, 15162, 22312, 22433, 32156):
        if i < len(valid):
            valid[i] = False
    return valid


# ------------------------------------------------------------------------------------------------ one record (dataloader.py:449-712)
def _segments_of(example, num_segments):
    segs = []
    for i in range(num_segments):
        seg = {}
        for k, kind in SEGMENT_FEATURES.items():
            v = example.get(f'c{i:02d}/{k}')
            missing = v is None or len(v) == 0
            if kind == 'bytes':
                seg[k] = _DEFAULTS.get(k, b'') if missing else v[0]
            elif kind == 'int':
                seg[k] = 1 if missing else int(v[0])
            elif kind == 'float':
                seg[k] = np.float32(1.0) if missing else np.float32(v[0])
            elif kind == 'ints':
                seg[k] = np.zeros(0, np.int32) if missing else np.asarray(v, dtype=np.int64).astype(np.int32)
            else:
                seg[k] = np.zeros(0, np.float32) if missing else np.asarray(v, dtype=np.float32)
        segs.append(seg)
    return segs


def merged_data_config(config):
    """make_dataset's merged_config (dataloader.py:876-877): the data section updated with the model section."""
    if 'data' not in config:
        return config
    c = dict(config['data'])
    c.update(config['model'])
    return c


def dataset_parser(record, config, rng=None, token_is_valid=None, as_bf16_bits=False):
    """dataloader.py:449-712 for one serialised record (or an already parsed Example dict).  `config` is the merged data + model section (or the
    whole config).  Returns numpy features: images [nseg, hw, P P 3] f32, audio_clips [nseg, nsub, T, 65] f32, text2audio / audio2text
    [groups x seqs, lang_seq_len, 3], audio_text_matching [seq_len, 3], random_text [num_text_seqs, seq_len, 3], text_spans [nspans, span_len],
    video_src_index [nseg], meta / youtube_id bytes."""
    config = merged_data_config(config)
    rng = np.random.default_rng() if rng is None else rng
    if token_is_valid is None:
        token_is_valid = make_token_is_valid()
    example = parse_example(record) if isinstance(record, (bytes, bytearray, memoryview)) else record
    nseg, nsub = config['num_segments'], config['num_audio_subsegments']
    segs = _segments_of(example, nseg)
    feats = {}

    # (as_bf16_bits: images and audio_clips leave as uint16 bf16 bit patterns, converted where they are produced -- the batch will be bf16 anyway)
    feats['images'] = np.stack([load_and_resize_img(s['image/encoded'], config, rng, as_bf16_bits=as_bf16_bits) for s in segs])
    if config.get('disable_imgs_dataloader', False):
        feats['images'] = np.zeros_like(feats['images'])
    audio, audio_start, audio_end = [], [], []
    for s in segs:
        if len(s['playback_speed']) != 1:
            raise ValueError('playback_speed must hold one value per segment')
        a, t0, t1 = load_audio(s['spectrogram/encoded'], s['spectrogram/magic_number'], s['playback_speed'][0], config, rng)
        audio.append(a)
        audio_start.append(t0)
        audio_end.append(t1)
    feats['audio_clips'] = np.stack(audio)
    if config.get('disable_audio_dataloader', False):
        feats['audio_clips'] *= 0.0
    if as_bf16_bits:
        feats['audio_clips'] = bf16_bits(feats['audio_clips'])

    num_audio_spans = nseg * nsub
    ntrg = int(num_audio_spans * config['mask_rate'])
    n_t2a, n_a2t = config['num_text2audio_seqs'], config['num_audio2text_seqs']

    # tokens -> audio spans: the nearest window centre, made monotone (dataloader.py:508-531)
    segment_idx, cents, start_end = [], [], []
    t_start = np.float32(0.0)
    for i, s in enumerate(segs):
        if not (len(s['tok_ids']) == len(s['tok_start_times']) == len(s['tok_end_times'])):
            raise ValueError(f'segment {i}: token ids and times differ in length')
        tok_c = (s['tok_start_times'] + s['tok_end_times']) / np.float32(2.0)
        aud_c = (audio_start[i] + audio_end[i]) / np.float32(2.0)
        assignment = np.abs(tok_c[:, None] - aud_c[None]).argmin(1).astype(np.int32) if len(tok_c) else np.zeros(0, np.int32)
        segment_idx.append(cumulative_maximum_int(assignment) + i * nsub)
        cents.append(tok_c + t_start)
        start_end.append(np.stack([audio_start[i], audio_end[i]], -1) + t_start)
        t_start = t_start + (s['end_time'] - s['start_time'])
    tokens = Ragged.from_value_rowids(np.concatenate([s['tok_ids'] for s in segs]), np.concatenate(segment_idx), nrows=num_audio_spans)
    tok_centroids_vals = np.concatenate(cents)
    audio_start_end = np.concatenate(start_end, 0)

    trg = uniform_random_select(rng, num_audio_spans, ntrg * (n_t2a + n_a2t), sort_idx=False)
    t2a_idx = trg[:ntrg * n_t2a].reshape(n_t2a, ntrg)
    a2t_idx = trg[ntrg * n_t2a:].reshape(n_a2t, ntrg)

    spans_all, t2a_streams = [], []
    for i in range(n_t2a):                                    # text -> audio (dataloader.py:547-567)
        tk = reassign_empty_tokens(rng, tokens, t2a_idx[i])
        tk = increase_textmask(rng, tk, t2a_idx[i], tok_centroids_vals, audio_start_end, delta_thresh=0.125)
        spans, groups = mask_tokens(rng, tk, t2a_idx[i], text_span_start_counter=i * ntrg, num_groups=config['num_segment_groups'],
                                    padded_seq_len=config['lang_seq_len'], do_audio_mask=True)
        spans_all += spans
        t2a_streams += groups
    feats['text2audio'] = np.stack(t2a_streams)

    a2t_streams = []
    for i in range(n_a2t):                                    # audio -> text (dataloader.py:573-597): spans beside a masked one usually become text
        masked = _one_hot(a2t_idx[i], num_audio_spans)
        ext = np.concatenate([[False], masked, [False]])
        should_textify = (ext[2:] | ext[:-2]) & ~masked
        should_textify &= rng.random(num_audio_spans) < config.get('convert_extra_span_to_text_prob', 0.8)
        spans, groups = mask_tokens(rng, tokens, a2t_idx[i], do_audio_span=~should_textify, audio_token_length=config['audio_token_length'],
                                    padded_seq_len=config['lang_seq_len'], text_span_start_counter=(i + n_t2a) * ntrg,
                                    num_groups=config['num_segment_groups'])
        spans_all += spans
        a2t_streams += groups
    feats['audio2text'] = np.stack(a2t_streams)

    max_text_seq_len = config.get('max_text_seq_len', config['seq_len'])

    # audio / text <-> frames matching stream (dataloader.py:606-641)
    use_audio_tokens = bool(rng.random() < config.get('use_audio_token_prob', 1.0))
    rows = []
    for i, s in enumerate(segs):
        rows.append(np.array([[LTOVPOOL, i * nsub, -1]], dtype=np.int32))
        if use_audio_tokens:
            ptr = np.repeat(np.arange(nsub, dtype=np.int32) + i * nsub, config['audio_token_length'])
            rows.append(np.stack([np.full_like(ptr, AUDIOSPAN), ptr, np.full_like(ptr, -1)], 1))
        else:
            t = s['tok_ids']
            rows.append(np.stack([t, np.full_like(t, i * nsub), np.full_like(t, -1)], 1))
    matching = np.concatenate(rows, 0)
    aux = np.concatenate([[START], TITLE_IDS, segs[0]['title'], [START], DESCRIPTION_IDS, segs[0]['description'], [START], TAGS_IDS, segs[0]['tags'],
                          [END]]).astype(np.int32)
    aux = aux[:max(max_text_seq_len - len(matching), 0)]
    aux = np.stack([aux, np.full_like(aux, -1), np.full_like(aux, -1)], 1)
    feats['audio_text_matching'] = pad_tokens_to_fixed_size(np.concatenate([aux, matching], 0), config['seq_len'])

    # plain text with pseudo segments (dataloader.py:645-697)
    n_in_record = config['num_text_seqs_in_record']
    assert config['num_text_seqs'] <= n_in_record
    random_text = [segs[i]['random_text'] for i in range(min(n_in_record, nseg))]
    counter = ntrg * (n_a2t + n_t2a)
    rt_streams = []
    for j in uniform_random_select(rng, n_in_record, config['num_text_seqs']):
        budget = config['text_span_budget'] if 'text_span_budget' in config else int(max_text_seq_len / (5.5 / config['mask_rate'] - 5.5 + 1.0))
        tk, extra_lhs, extra_rhs = convert_rawtext_into_fake_segments(rng, random_text[j], desired_len=max_text_seq_len, span_budget=budget,
                                                                      use_v1_stats='ytt180m' in config.get('train_fns', ''))
        mask_w = np.array([0.2 + 0.8 * float(token_is_valid[r].all()) for r in tk.rows()])
        do_mask = np.sort(random_categorical_without_replacement(rng, np.log(mask_w), budget))       # (too little text for the budget: ValueError)
        spans, streams = mask_tokens(rng, tk, do_mask, text_span_start_counter=counter, num_groups=1)
        stream = streams[0]
        needed = max(max_text_seq_len - len(stream), 0)
        amt_lhs = min(len(extra_lhs), needed // 2)
        lhs = extra_lhs[len(extra_lhs) - amt_lhs:]
        lhs = np.stack([lhs, np.zeros_like(lhs), np.full_like(lhs, -1)], 1)
        amt_rhs = min(len(extra_rhs), (needed + 1) // 2)
        rhs = extra_rhs[:amt_rhs]
        rhs = np.stack([rhs, np.full_like(rhs, stream[-1, 1] + 1), np.full_like(rhs, -1)], 1)
        rt_streams.append(pad_tokens_to_fixed_size(np.concatenate([lhs, stream, rhs], 0), config['seq_len']))
        spans_all += filter_out_tokens_not_in_youtube(spans, token_is_valid)
        counter += budget
    if config['num_text_seqs'] > 0:
        feats['random_text'] = np.stack(rt_streams)

    span_len = config['text_span_length']
    ts = np.full((len(spans_all), span_len), PADDING, dtype=np.int32)
    for r, s in enumerate(spans_all):
        ts[r, :min(len(s), span_len)] = s[:span_len]
    feats['text_spans'] = ts
    feats['video_src_index'] = np.array([s['video_src_index'] for s in segs], dtype=np.int32)
    feats['meta'], feats['youtube_id'] = segs[0]['meta'], segs[0]['youtube_id']
    return feats


def bf16_bits(x):
    """float32 array -> uint16 array holding the bfloat16 (round to nearest even) of every value: what `.to(torch.bfloat16)` produces, computed where
    the record is parsed so that half the bytes cross the process boundary."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return ((u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def handle_batch(records, use_bfloat16=True, device='cpu', as_numpy=False, out=None):
    """dataloader.py:732-789 for ONE device (`records`: a list of dataset_parser outputs): the batch dict of synthetic.make_batch -- images
    [B, nseg hw, P P 3] and audio_clips [B, nseg nsub T, 65] as torch tensors (bf16 when use_bfloat16), the token streams split into the id /
    audio_ptr / text_ptr planes [B, n, L] (numpy int32: the planner reads them on the host), text_spans [B, nspans, span_len], video_src_index.
    as_numpy (the feeder process: no torch there): the two float arrays stay numpy -- uint16 bf16 bit patterns when use_bfloat16, float32 otherwise --
    written straight into `out[k]` (views of a shared-memory slot) when given."""
    B = len(records)
    batch = {}
    if as_numpy:
        for k in ('images', 'audio_clips'):
            dst = None if out is None else out[k]
            for i, r in enumerate(records):
                x = r[k].reshape(-1, r[k].shape[-1])
                if use_bfloat16 and x.dtype != np.uint16:
                    x = bf16_bits(x)
                elif not use_bfloat16 and x.dtype == np.uint16:
                    x = (x.astype(np.uint32) << np.uint32(16)).view(np.float32)
                if dst is None:
                    dst = np.empty((B,) + x.shape, dtype=x.dtype)
                dst[i] = x
            batch[k] = dst
    else:
        import torch
        dt = torch.bfloat16 if use_bfloat16 else torch.float32
    for k in ('images', 'audio_clips') if not as_numpy else ():
        x = np.stack([r[k] for r in records])
        x = np.ascontiguousarray(x.reshape(B, -1, x.shape[-1]))
        if x.dtype == np.uint16:                                # bf16 bit patterns from a parser worker (bf16_bits)
            t = torch.from_numpy(x.view(np.int16)).view(torch.bfloat16)
            batch[k] = (t if use_bfloat16 else t.to(torch.float32)).to(device)
        else:
            batch[k] = torch.from_numpy(x).to(dt).to(device)
    batch['text_spans'] = np.stack([r['text_spans'] for r in records]).astype(np.int32)
    batch['video_src_index'] = np.stack([r['video_src_index'] for r in records]).astype(np.int32)
    for k in ('text2audio', 'audio2text', 'audio_text_matching', 'random_text'):
        if k not in records[0]:
            continue
        x = np.stack([r[k] for r in records]).astype(np.int32)
        x = x.reshape(B, -1, x.shape[-2], 3)                    # [B, prod(middle dims), L, 3]
        batch[k] = np.ascontiguousarray(x[..., 0])
        batch[k + '/audio_ptr'] = np.ascontiguousarray(x[..., 1])
        batch[k + '/text_ptr'] = np.ascontiguousarray(x[..., 2])
    return batch


# ------------------------------------------------------------------------------------------------ shards -> batches (dataloader.py:864-955)
_ATTACHED = {}          # worker process: shared-memory segments of the parent's record slots, by name


def _slot_arrays(name, offset, rec_shapes):
    """This process' views of one record slot of the segment `name` (attached once per worker.  The pool's workers are children of the creating
    process and share its resource tracker, whose books are a set: attaching registers the name a second time without effect, and the creator's unlink
    clears it)."""
    m = _ATTACHED.get(name)
    if m is None:
        from multiprocessing import shared_memory
        m = shared_memory.SharedMemory(name=name)
        _ATTACHED.clear()                                         # (a new cycle's segment replaces the previous one)
        _ATTACHED[name] = m
    return _rec_views(m.buf, offset, rec_shapes)


def _rec_views(buf, offset, rec_shapes):
    out, off = {}, offset
    for k in ('images', 'audio_clips'):
        shape, dt = rec_shapes[k]
        out[k] = np.ndarray(shape, dtype=dt, buffer=buf, offset=off)
        off += (int(np.prod(shape)) * np.dtype(dt).itemsize + 4095) // 4096 * 4096
    return out


def _parse_job(args):
    """One record in a parser worker (module level: picklable).  Floats leave as bf16 bit patterns when the batch will be bf16 anyway -- and, when the
    job names a record slot (round 6: `_RecordSlots`), through shared memory: the worker writes images / audio_clips into the slot and returns only the
    token streams (a few KB) through the pool's pipe.  6.3 MB per base record crossed that pipe pickled before, which held a 16-process pool to
    136 records / s whatever the workers could parse."""
    rec, seed, merged, token_is_valid, as_bf16 = args[:5]
    slot = args[5] if len(args) > 5 else None
    if as_bf16 and 'fast_image_resize' not in merged:             # a bf16 batch: 8-bit resampling by default (one 8-bit step is 1 / 255, a bf16 step
        merged = dict(merged, fast_image_resize=True)             # between 0.5 and 1 is 1 / 256); data.fast_image_resize: false keeps the float resampler
    try:
        f = dataset_parser(rec, merged, rng=np.random.default_rng(seed), token_is_valid=token_is_valid, as_bf16_bits=as_bf16)
    except Exception as e:                                        # noqa: BLE001 -- the reference's iterator wraps the cycle in `except Exception`, logs and goes on
        # (dataloader.py:948-951): a corrupt JPEG raises PIL.UnidentifiedImageError / OSError, a truncated protobuf IndexError or struct.error,
        # a record with a missing feature KeyError -- one bad record must cost one record, not the cycle's shards and not the feeder
        print(f'records: skipping a record: {type(e).__name__}: {e}', flush=True)
        return None
    if slot is not None:
        name, offset, rec_shapes, idx = slot
        dst = _slot_arrays(name, offset, rec_shapes)
        for k in ('images', 'audio_clips'):
            if f[k].dtype != dst[k].dtype:                        # (cannot happen with the shapes make_dataset derives from the same config)
                return f
        for k in ('images', 'audio_clips'):
            dst[k][...] = f[k].reshape(dst[k].shape)
            f[k] = ('@slot', idx)
    return f


def _make_pool(workers, processes):
    if workers <= 0:
        return None
    if processes:
        import multiprocessing as mp
        return mp.get_context('spawn').Pool(workers)              # spawn, not fork: the parent usually holds an initialised GPU runtime
    from concurrent.futures import ThreadPoolExecutor
    return ThreadPoolExecutor(max_workers=workers)


class _RecordSlots:
    """Shared-memory slots for the float arrays of parsed records (parent side).  One segment of `count` slots; the parent names a free slot in every job
    it hands to the process pool and reads the record's images / audio_clips from its own mapping when the (small) result arrives.  A record keeps its
    slot until handle_batch has copied it into a batch.  None of this is required for correctness: with no segment (no /dev/shm room) or no free slot a
    job travels through the pool's pipe as before."""

    def __init__(self, config, count):
        self.shm, self.free, self.count = None, [], 0
        shp = _float_shapes(config, 1)
        self.rec_shapes = {k: (tuple(sh[1:]), dt) for k, (sh, dt) in shp.items()}
        self.rec_bytes = sum((int(np.prod(sh)) * np.dtype(dt).itemsize + 4095) // 4096 * 4096 for sh, dt in self.rec_shapes.values())
        try:
            st = os.statvfs('/dev/shm')
            room = st.f_bavail * st.f_frsize
        except OSError:
            room = 0
        count = int(min(count, (room // 2) // self.rec_bytes, (1 << 30) // self.rec_bytes))      # at most half of what is free, at most 1 GiB
        if count < 1:
            return
        try:
            from multiprocessing import shared_memory
            self.shm = shared_memory.SharedMemory(create=True, size=count * self.rec_bytes)
        except OSError:
            return
        self.count, self.free = count, list(range(count))

    def take(self):
        """(name, byte offset, per-record shapes, index) of a free slot for a job, or None."""
        if not self.free:
            return None
        i = self.free.pop()
        return (self.shm.name, i * self.rec_bytes, self.rec_shapes, i)

    def views(self, i):
        return _rec_views(self.shm.buf, i * self.rec_bytes, self.rec_shapes)

    def release(self, i):
        self.free.append(i)

    def close(self):
        if self.shm is not None:
            try:
                self.shm.close()
            except BufferError:                                   # a view of a slot is still alive somewhere: the mapping goes with the process
                pass
            try:
                self.shm.unlink()
            except FileNotFoundError:
                pass
            self.shm = None


def _close_pool(pool, processes):
    if pool is None:
        return
    if processes:
        pool.terminate()
        pool.join()
    else:
        pool.shutdown(wait=False)


def make_dataset(config, fns, batch_size, is_training=True, seed=None, token_is_valid=None, workers=0, device='cpu', processes=False, as_numpy=False,
                 slot_of=None, pool=None, rec_slots=None):
    """Generator of per-device batches from the shards `fns`: records of the shards interleaved round-robin (tf.data's parallel reads), passed through
    a shuffle buffer of config['device']['shuffle_buffer_size'] records when training, parsed, grouped into batches of batch_size with the remainder
    dropped.  `workers` > 0 parses in a pool: threads by default (PIL releases the GIL while it decodes and resamples; ~2x on 8 threads), or
    -- processes=True -- spawned worker PROCESSES that import neither torch nor the library and never touch the GPU; the pool parses chunk k + 1 while
    this process assembles the batches of chunk k and, since round 6, the workers hand their float arrays back through shared-memory record slots
    (_RecordSlots) instead of the pool's pipe.  Measured on the GPU box's host (EPYC 9575F, a job's share of ~16 cores; scripts/records_demo.py, frames
    stored at the corpus' 288 x 512, base grid; profiles/r06_reader_throughput.txt): 35.5 records / s on one core (8-bit image path; 29.3 with the float
    resampler), 256 with 8 processes, 354 with 16 (434 sustained) -- a base step consumes 4 records / 29 ms = 138 records / s per GPU.  (Round 5, 360 x
    640 frames, results through the pipe: 21.5 / 119 / 136.)
    The record -> random-stream assignment does not depend on the pool, so every mode yields the same batches for the same seed."""
    merged = merged_data_config(config)
    rng = np.random.default_rng(seed)
    token_is_valid = make_token_is_valid() if token_is_valid is None else token_is_valid
    shards = [iter_tfrecord(fn) for fn in ([fns] if isinstance(fns, (str, os.PathLike)) else list(fns))]      # streamed: one record per shard in memory
    order = (rec for group in zip_longest_skip(shards) for rec in group)
    buf_size = config.get('device', {}).get('shuffle_buffer_size', 256) if is_training else 1

    def shuffled():
        buf = []
        for rec in order:
            buf.append(rec)
            if len(buf) >= buf_size:
                yield buf.pop(int(rng.integers(0, len(buf))) if is_training else 0)
        while buf:
            yield buf.pop(int(rng.integers(0, len(buf))) if is_training else 0)

    use_bf16 = merged.get('use_bfloat16', True)
    own_pool = pool is None                                        # (input_fn_builder keeps ONE pool over its cycles: spawning 16 workers costs ~1-2 s)
    if own_pool:
        pool = _make_pool(workers, processes)
    chunk_size = max(batch_size, 2 * workers) if (pool is not None and processes) else batch_size      # keep every worker busy between two flushes
    # parsed records come back through shared memory when the pool is made of processes (see _parse_job): two chunks in flight / being batched + a
    # batch's worth of leftovers
    own_slots = rec_slots is None and pool is not None and processes
    if own_slots:
        rec_slots = _RecordSlots(config, 2 * chunk_size + batch_size)
    if not (pool is not None and processes):
        rec_slots = None
    try:
        done, chunk, pending = [], [], None

        def submit():
            """Hand the chunk to the pool and return a handle whose results are collected one chunk LATER: the workers parse chunk k + 1 while this
            process assembles the batches of chunk k (a blocking map left both sides idle half of the time: 16 workers gave what 8 did)."""
            slots = [None if rec_slots is None else rec_slots.take() for _ in chunk]
            jobs = [(rec, s, merged, token_is_valid, use_bf16) + (() if sl is None else (sl,)) for (rec, s), sl in zip(chunk, slots)]
            chunk.clear()
            if pool is None:
                return map(_parse_job, jobs), slots
            return (pool.map_async(_parse_job, jobs) if processes else pool.map(_parse_job, jobs)), slots

        def collect(handle):
            handle, slots = handle
            parsed = handle.get() if (pool is not None and processes) else handle
            for r, sl in zip(parsed, slots):
                if r is not None and sl is not None and isinstance(r['images'], tuple):
                    r.update(rec_slots.views(sl[3]))              # this process' views of the slot the worker filled
                    r['_rslot'] = sl[3]
                elif sl is not None:                              # a skipped record (or one that came through the pipe after all): the slot is free again
                    rec_slots.release(sl[3])
                if r is not None:
                    done.append(r)

        def ready():
            while len(done) >= batch_size:
                out = None if slot_of is None else slot_of()      # (ShardFeeder: a free shared-memory slot; blocks until the consumer released one)
                b = handle_batch(done[:batch_size], use_bfloat16=use_bf16, device=device, as_numpy=as_numpy, out=None if out is None else out[1])
                for r in done[:batch_size]:                       # handle_batch copied the float arrays: the records' slots are free
                    if '_rslot' in r:
                        r.pop('images'), r.pop('audio_clips')
                        rec_slots.release(r.pop('_rslot'))
                if out is not None:
                    b['_slot'] = out[0]
                yield b
                del done[:batch_size]
        for rec in shuffled():
            chunk.append((rec, int(rng.integers(0, 2 ** 63))))
            if len(chunk) == chunk_size:
                nxt = submit()
                if pending is not None:
                    collect(pending)
                    yield from ready()
                pending = nxt
        if chunk:
            nxt = submit()
            if pending is not None:
                collect(pending)
                yield from ready()
            pending = nxt
        if pending is not None:
            collect(pending)
        yield from ready()                                         # (drop_remainder=True)
    finally:
        if own_pool:
            _close_pool(pool, processes)
        if own_slots:
            done.clear()
            rec_slots.close()


def input_fn_builder(config, rank=0, world=1, seed=None, epochs=None, **kw):
    """dataloader.py:906-955 for one process per GPU: this rank's shards (file i belongs to rank i mod world, like the reference's hosts), shuffled
    every epoch, read n_fns_per_cycle at a time; batches of config['device']['batch_size'] // world records.  Endless unless `epochs` is given.
    Wrap it in loader.PrefetchLoader for the pinned-memory / copy-stream prefetch (the reference's prefetch_to_device)."""
    d, dev = config['data'], config.get('device', {})
    fns = [d['train_fns'].format(i) for i in range(d['num_train_files']) if i % world == rank]
    if not fns:
        raise ValueError(f'rank {rank} of {world} has no shard among {d["num_train_files"]} files')
    batch_size = max(dev.get('batch_size', world) // world, 1)
    per_cycle = min(dev.get('n_fns_per_cycle', 32), len(fns))
    while len(fns) % per_cycle != 0:
        per_cycle -= 1
    rng = np.random.default_rng(seed)
    epoch = 0
    workers, processes = kw.get('workers', 0), kw.get('processes', False)
    pool = _make_pool(workers, processes)                          # one pool for every cycle of every epoch
    try:
        while epochs is None or epoch < epochs:
            order = [fns[i] for i in rng.permutation(len(fns))]
            for s in range(0, len(order) - per_cycle + 1, per_cycle):
                try:
                    yield from make_dataset(config, order[s:s + per_cycle], batch_size, is_training=True, seed=int(rng.integers(0, 2 ** 63)), pool=pool, **kw)
                except Exception as e:                             # noqa: BLE001 -- an unreadable shard: reported, the cycle skipped (dataloader.py:948-951 catches Exception)
                    print(f'records: {type(e).__name__}: {e}', flush=True)
            epoch += 1
    finally:
        _close_pool(pool, processes)


def zip_longest_skip(lists):
    """Round-robin over lists of unequal length, exhausted lists dropping out."""
    its = [iter(l) for l in lists]
    while its:
        group, alive = [], []
        for it in its:
            try:
                group.append(next(it))
                alive.append(it)
            except StopIteration:
                pass
        its = alive
        if group:
            yield group


# ------------------------------------------------------------------------------------------------ the reader in a process of its own
def _float_shapes(config, batch_size):
    c = merged_data_config(config)
    h1, w1 = c['output_grid']
    images = (batch_size, c['num_segments'] * h1 * w1, c['vit_patch_size'] ** 2 * 3)
    audio = (batch_size, c['num_segments'] * c['num_audio_subsegments'] * c['audio_seq_length'], c['num_mels'] + 1)
    dt = np.uint16 if c.get('use_bfloat16', True) else np.float32
    return {'images': (images, dt), 'audio_clips': (audio, dt)}


def _slot_views(buf, shapes):
    out, off = {}, 0
    for k in ('images', 'audio_clips'):
        shape, dt = shapes[k]
        n = int(np.prod(shape)) * np.dtype(dt).itemsize
        out[k] = np.ndarray(shape, dtype=dt, buffer=buf, offset=off)
        off += (n + 4095) // 4096 * 4096
    return out, off


class _FeederStop(BaseException):
    """The consumer closed the feeder: raised inside make_dataset's slot request to unwind the generators.  Control flow, not an error -- a
    BaseException so that the per-cycle `except Exception` of input_fn_builder (the reference's log-and-continue) does not swallow it."""


def _feeder_main(config, rank, world, seed, epochs, kw, shm_names, shapes, full, free, stop):
    """Body of the feeder process: input_fn_builder with its parser pool, every batch's float arrays written into a shared-memory slot, the integer
    streams (a few hundred KB) sent through the queue.  No torch, no GPU."""
    shms = []
    try:
        from multiprocessing import shared_memory
        shms = [shared_memory.SharedMemory(name=n) for n in shm_names]
        views = [_slot_views(m.buf, shapes)[0] for m in shms]

        import queue

        def slot_of():
            while True:                                           # blocks while the consumer holds every slot; leaves when it closes the feeder
                if stop.is_set():
                    raise _FeederStop()
                try:
                    i = free.get(timeout=0.25)
                    return i, views[i]
                except queue.Empty:
                    pass
        gen = input_fn_builder(config, rank=rank, world=world, seed=seed, epochs=epochs, as_numpy=True, slot_of=slot_of, **kw)
        try:
            for b in gen:
                slot = b.pop('_slot')
                full.put((slot, {k: v for k, v in b.items() if k not in ('images', 'audio_clips')}))
                if stop.is_set():
                    break
            full.put(None)
        finally:
            gen.close()                                           # (unwinds make_dataset: its parser pool is terminated and joined)
    except _FeederStop:
        pass
    except BaseException as e:                                    # noqa: BLE001 -- reported to the consumer, which raises
        full.put(('error', f'{type(e).__name__}: {e}'))
    finally:
        for m in shms:
            m.close()


class ShardFeeder:
    """`input_fn_builder` in a PROCESS of its own, batches handed over through a ring of shared-memory slots.

    Why a process: the reader's Python (collecting and unpickling the parser pool's results, assembling batches: ~20 ms per base batch) would share the
    trainer's GIL with the step's host work (planner ~6 ms, graph launch ~4 ms) inside a 29 ms step.  Here the trainer's process only wraps a slot's float
    arrays as tensors (zero copy) and hands them to loader.PrefetchLoader, whose staging copy reads them once.  Measured end to end on the GPU box
    (scripts/records_feed_bench.py, profiles/r05_shard_fed_step.txt: shards on local disk -> 14 parser processes -> this feeder -> PrefetchLoader ->
    hipGraph replay, 150 steps over several shard cycles; round 6, profiles/r06_shard_fed_step.txt): 29.84-29.92 ms per step against 29.55 with resident batches
    (round 5: 31.4 against 29.1).

    A batch's float tensors are views of a slot that is reused `slots - 1` batches later: a consumer must have copied them by its next `next()` --
    PrefetchLoader does (it stages into pinned memory inside the call that fetched the batch).  Iterate it; `close()` (or leaving a `with`) stops the
    feeder and releases the shared memory."""

    def __init__(self, config, rank=0, world=1, seed=None, epochs=None, workers=16, slots=4, **kw):
        import multiprocessing as mp
        from multiprocessing import shared_memory
        self.config = config
        c = merged_data_config(config)
        bsz = max(config.get('device', {}).get('batch_size', world) // world, 1)
        self.shapes = _float_shapes(config, bsz)
        self.use_bf16 = bool(c.get('use_bfloat16', True))
        size = sum((int(np.prod(sh)) * np.dtype(dt).itemsize + 4095) // 4096 * 4096 for sh, dt in self.shapes.values())
        self.shms = [shared_memory.SharedMemory(create=True, size=size) for _ in range(slots)]
        self.views = [_slot_views(m.buf, self.shapes)[0] for m in self.shms]
        ctx = mp.get_context('spawn')
        self.full, self.free, self.stop = ctx.Queue(), ctx.Queue(), ctx.Event()
        for i in range(slots):
            self.free.put(i)
        kw = dict(kw, workers=workers, processes=workers > 0)
        self.proc = ctx.Process(target=_feeder_main, args=(config, rank, world, seed, epochs, kw, [m.name for m in self.shms], self.shapes, self.full,
                                                            self.free, self.stop), daemon=False)
        self.proc.start()
        self.held = None
        self.closed = self.exhausted = False

    def __iter__(self):
        return self

    def __next__(self):
        import torch
        if self.held is not None:                                 # the consumer is done with the previous batch (it asked for the next one)
            self.free.put(self.held)
            self.held = None
        import queue
        if self.exhausted:
            raise StopIteration
        while True:
            try:
                item = self.full.get(timeout=1.0)
                break
            except queue.Empty:
                if not self.proc.is_alive():
                    raise RuntimeError(f'the feeder process died (exit code {self.proc.exitcode})') from None
        if item is None:
            self.exhausted = True
            raise StopIteration
        if item[0] == 'error':
            self.exhausted = True
            raise RuntimeError(f'the feeder process failed: {item[1]}')
        slot, ints = item
        self.held = slot
        batch = dict(ints)
        for k, v in self.views[slot].items():
            batch[k] = torch.from_numpy(v.view(np.int16)).view(torch.bfloat16) if self.use_bf16 else torch.from_numpy(v)
        return batch

    def close(self):
        if self.closed:
            return
        self.closed = True
        self.stop.set()                                           # the feeder leaves its loop, closes its generator (parser pool included) and exits
        self.proc.join(timeout=15)
        if self.proc.is_alive():
            self.proc.terminate()
            self.proc.join(timeout=10)
        for q in (self.full, self.free):
            q.close()
            q.cancel_join_thread()
        self.views = None
        for m in self.shms:
            try:
                m.close()
                m.unlink()
            except (BufferError, FileNotFoundError):              # a batch view still alive in the caller: the segment goes with the process
                pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:                                         # noqa: BLE001 -- interpreter shutdown
            pass


# ------------------------------------------------------------------------------------------------ fixtures
def make_synthetic_record(config, rng, frame_hw=(90, 160), random_text_len=None, tokens_per_segment=(4, 14)):
    """A serialised record with the reference's feature layout and plausible content (smooth random frames and spectrograms as JPEGs, tokens
    with start / end times spread over each segment's span): what the tests and `scripts/records_demo.py` feed the reader, since no shard of the
    real corpus can be fetched here.  frame_hw: stored frame size (the reader rescales to the config's grid)."""
    from PIL import Image
    config = merged_data_config(config)
    nseg, nsub = config['num_segments'], config['num_audio_subsegments']
    seg_seconds = config['spec_size'] * config['fft_hop_length'] / config['sample_rate']
    feats = {}
    max_len = config.get('max_text_seq_len', config['seq_len'])
    random_text_len = random_text_len or int(max_len * 2.2)

    def jpeg(arr, mode):
        buf = io.BytesIO()
        Image.fromarray(arr, mode=mode).save(buf, format='JPEG', quality=90)
        return buf.getvalue()
    src = 0
    for i in range(nseg):
        p = f'c{i:02d}/'
        h, w = frame_hw
        low = rng.uniform(0, 255, size=(h // 8 + 1, w // 8 + 1, 3)).astype(np.float32)
        frame = np.asarray(Image.fromarray(low.astype(np.uint8)).resize((w, h), Image.Resampling.BICUBIC))
        feats[p + 'image/encoded'] = jpeg(frame, 'RGB')
        feats[p + 'image/format'] = b'jpeg'
        feats[p + 'image/height'], feats[p + 'image/width'] = h, w
        spec = np.clip(rng.normal(110, 40, size=(config['num_mels'], config['spec_size'])), 0, 255).astype(np.uint8)
        feats[p + 'spectrogram/encoded'] = jpeg(spec, 'L')
        feats[p + 'spectrogram/format'] = b'jpeg'
        feats[p + 'spectrogram/height'], feats[p + 'spectrogram/width'] = config['num_mels'], config['spec_size']
        feats[p + 'spectrogram/magic_number'] = np.float32(25.0)
        feats[p + 'youtube_id'] = b'synthetic00'
        if i > 0 and rng.random() < 0.05:
            src += 1
        feats[p + 'video_src_index'] = src
        feats[p + 'playback_speed'] = [1]
        feats[p + 'start_time'], feats[p + 'end_time'] = np.float32(i * seg_seconds), np.float32((i + 1) * seg_seconds)
        n = int(rng.integers(tokens_per_segment[0], tokens_per_segment[1] + 1))
        if rng.random() < 0.1:
            n = 0                                              # a silent segment: empty rows for reassign_empty_tokens to repair
        t0 = np.sort(rng.uniform(0.0, seg_seconds * 0.97, size=n)).astype(np.float32)
        feats[p + 'tok_ids'] = rng.integers(11, 32768, size=n)
        feats[p + 'tok_start_times'] = t0
        feats[p + 'tok_end_times'] = (t0 + rng.uniform(0.05, 0.4, size=n)).astype(np.float32)
        feats[p + 'random_text'] = rng.integers(11, 32768, size=random_text_len) if i < config['num_text_seqs_in_record'] else []
        if i == 0:
            feats[p + 'title'] = rng.integers(11, 32768, size=6)
            feats[p + 'description'] = rng.integers(11, 32768, size=20)
            feats[p + 'tags'] = rng.integers(11, 32768, size=5)
            feats[p + 'meta'] = b'{}'
    return make_example(feats)
