"""The real-data input path without TensorFlow (merlot_reserve_amd/records.py + csrc/hostio.cpp; pretrain/dataloader.py:449-789): the container
formats against known answers and round trips, the masking rules on hand-derived cases, the invariants a parsed record must satisfy, and
acceptance of a record-fed batch by the planner.  (The reference holds no recorded batch: structure is all there is to pin.)"""
import struct

import numpy as np
import pytest

from merlot_reserve_amd import records as R
from merlot_reserve_amd.config import Dims, load_config, tiny_config
from merlot_reserve_amd.synthetic import AUDIOSPAN, LTOVPOOL, MASK, MASKAUDIO, make_batch


def test_crc32c_known_answers():
    """RFC 3720 B.4 test vectors; the masked form of TFRecord files."""
    lib = R._lib()
    assert lib.mr_crc32c(b'123456789', 9, 0) == 0xE3069283
    assert lib.mr_crc32c(bytes(32), 32, 0) == 0x8A9136AA
    assert lib.mr_crc32c(bytes([0xff] * 32), 32, 0) == 0x62A8AB43
    assert lib.mr_crc32c(bytes(range(32)), 32, 0) == 0x46DD794E
    assert lib.mr_crc32c(bytes(range(31, -1, -1)), 32, 0) == 0x113FDB5C
    a = bytes(range(200)) * 7 + b'xyz'                                  # unaligned start, odd length, running value
    assert lib.mr_crc32c(a[5:], len(a) - 5, lib.mr_crc32c(a[:5], 5, 0)) == lib.mr_crc32c(a, len(a), 0)
    c = lib.mr_crc32c(b'123456789', 9, 0)
    assert lib.mr_crc32c_masked(b'123456789', 9) == ((((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff)


def test_tfrecord_framing_round_trip_and_corruption(tmp_path):
    recs = [b'', b'a', bytes(range(256)) * 33, b'tail']
    fn = tmp_path / 'x.tfrecord'
    R.write_tfrecord(fn, recs)
    assert R.read_tfrecord(fn) == recs and list(R.iter_tfrecord(fn)) == recs          # whole-file scanner (native) and the streaming reader
    raw = bytearray(fn.read_bytes())
    assert struct.unpack('<Q', raw[:8])[0] == 0 and len(raw) == sum(16 + len(r) for r in recs)
    bad = bytearray(raw)
    bad[16 + 16 + 1 + 12 + 100] ^= 1                                     # one bit inside the third record's data
    with pytest.raises(ValueError, match='record 2'):
        R.read_tfrecord(bytes(bad))
    assert R.read_tfrecord(bytes(bad), verify=False)[3] == b'tail'
    with pytest.raises(ValueError, match='past the end|truncated'):
        R.read_tfrecord(bytes(raw[:-3]))
    # a file cut 12..15 bytes into its last record (header whole, no room for the data checksum): the scanner's bound check must not wrap
    last = len(raw) - (16 + len(recs[-1]))
    for cut in (12, 13, 14, 15):
        for verify in (True, False):
            with pytest.raises(ValueError, match='past the end'):
                R.read_tfrecord(bytes(raw[:last + cut]), verify=verify)
    bad = bytearray(raw)
    bad[16] ^= 4                                                         # the second record's length field
    with pytest.raises(ValueError, match='record 1'):
        R.read_tfrecord(bytes(bad))
    fn2 = tmp_path / 'bad.tfrecord'
    fn2.write_bytes(bytes(bad))
    with pytest.raises(ValueError, match='record 1'):
        list(R.iter_tfrecord(fn2))
    fn2.write_bytes(bytes(raw[:-3]))
    with pytest.raises(ValueError, match='record 3'):
        list(R.iter_tfrecord(fn2))


def test_example_round_trip_and_unpacked_lists():
    feats = {'a/bytes': b'\x00\xffjpeg', 'b/ints': np.array([0, 1, -1, 2 ** 40, -2 ** 62]), 'c/floats': np.array([0.5, -3.25, 1e-8], np.float32),
             'd/empty': [], 'e/two': [b'x', b'yz'], 'f/scalar': 7}
    ex = R.parse_example(R.make_example(feats))
    assert ex['a/bytes'] == [b'\x00\xffjpeg'] and ex['e/two'] == [b'x', b'yz']
    assert ex['b/ints'].tolist() == [0, 1, -1, 2 ** 40, -2 ** 62] and ex['f/scalar'].tolist() == [7]
    assert np.array_equal(ex['c/floats'], feats['c/floats']) and len(ex['d/empty']) == 0
    # a writer that does not pack: one varint / one fixed32 per value
    ld = R._ld
    ints = ld(3, b''.join(b'\x08' + R._enc_varint(v) for v in (5, -2)))
    floats = ld(2, b''.join(b'\x0d' + struct.pack('<f', v) for v in (1.5, 2.5)))
    msg = ld(1, ld(1, ld(1, b'i') + ld(2, ints)) + ld(1, ld(1, b'f') + ld(2, floats)))
    ex = R.parse_example(msg)
    assert ex['i'].tolist() == [5, -2] and ex['f'].tolist() == [1.5, 2.5]


def test_ragged_shifts_by_hand():
    t = R.Ragged(np.arange(10, 20), [3, 0, 4, 3])                        # rows: [10 11 12] [] [13 14 15 16] [17 18 19]
    a = R.shift_ragged_tokens_at_positions(t, [2], right_to_left=True)   # row 2 hands its FIRST token to row 1
    assert [r.tolist() for r in a.rows()] == [[10, 11, 12], [13], [14, 15, 16], [17, 18, 19]]
    b = R.shift_ragged_tokens_at_positions(t, [0, 1], right_to_left=False)   # row 0 hands its LAST token to row 1; row 1 is empty: nothing to give
    assert [r.tolist() for r in b.rows()] == [[10, 11], [12], [13, 14, 15, 16], [17, 18, 19]]
    # an empty masked row takes from an unmasked neighbour with >= 2 tokens, never from a masked one or from a row of one
    rng = np.random.default_rng(0)
    c = R.reassign_empty_tokens(rng, t, mask_idx=[1])
    assert c.row_lengths[1] == 1 and c.row_lengths.sum() == 10 and (c.row_lengths[[0, 2]].tolist() in ([2, 4], [3, 3]))
    d = R.reassign_empty_tokens(rng, R.Ragged(np.arange(4), [1, 0, 3]), mask_idx=[1, 2])
    assert d.row_lengths.tolist() == [1, 0, 3]
    # increase_textmask: row 1 (masked, audio window 1.0 .. 2.0) takes the token at 2.05 s from its right neighbour but not the one at 0.2 s on the left
    t2 = R.Ragged(np.arange(6), [2, 1, 3])
    cent = np.array([0.2, 0.2, 1.5, 2.05, 2.6, 2.9], np.float32)
    se = np.array([[0, 1], [1, 2], [2, 3]], np.float32)
    e = R.increase_textmask(np.random.default_rng(1), t2, [1], cent, se, delta_thresh=0.125)
    assert e.row_lengths.tolist() == [2, 2, 2]


def test_mask_tokens_and_select_tokens():
    rng = np.random.default_rng(0)
    t = R.Ragged(np.arange(100, 112), [2, 3, 1, 2, 0, 4])
    spans, groups = R.mask_tokens(rng, t, [3, 1], do_audio_span=np.array([1, 1, 0, 1, 1, 1], bool), audio_token_length=2, text_span_start_counter=7,
                                  num_groups=2, padded_seq_len=8, do_audio_mask=False)
    assert [s.tolist() for s in spans] == [[102, 103, 104], [106, 107]]  # the ORIGINAL rows at the sorted indices 1, 3
    g0, g1 = groups
    assert g0.tolist() == [[AUDIOSPAN, 0, -1], [AUDIOSPAN, 0, -1], [MASK, 1, 7], [105, 2, -1]] + [[0, -1, -1]] * 4
    assert g1.tolist() == [[MASK, 3, 8], [AUDIOSPAN, 4, -1], [AUDIOSPAN, 4, -1], [AUDIOSPAN, 5, -1], [AUDIOSPAN, 5, -1]] + [[0, -1, -1]] * 3
    _, (g,) = R.mask_tokens(rng, t, [0], do_audio_mask=True)
    assert g[:2].tolist() == [[MASK, 0, 0], [MASKAUDIO, 0, 0]] and len(g) == 2 + 10
    # an over-long stream keeps every MASK and exactly the budget
    long = R.Ragged(rng.integers(11, 999, size=400), [10] * 40)
    _, (s,) = R.mask_tokens(rng, long, np.arange(0, 40, 4), num_groups=1, padded_seq_len=64)
    assert s.shape == (64, 3) and (s[:, 0] == MASK).sum() == 10 and (np.diff(s[:, 1]) >= 0).all()


def _check_record(f, cfg, d):
    ntrg, budget = d.ntrg, d.budget
    nseg, nsub = d.nseg, d.nas
    assert f['images'].shape == (nseg, d.hw, d.pp3) and f['images'].dtype == np.float32 and 0.0 <= f['images'].min() and f['images'].max() <= 1.0
    assert f['audio_clips'].shape == (nseg, nsub, d.a_raw, 65) and (f['audio_clips'][..., 64] == 1.0).all()
    t2a, a2t = f['text2audio'], f['audio2text']
    assert t2a.shape == a2t.shape == (d.ngroups, d.lang, 3)
    flat = t2a.reshape(-1, 3)
    m = flat[flat[:, 0] == MASK]
    assert len(m) == ntrg and sorted(m[:, 2].tolist()) == list(range(ntrg)) and (np.diff(m[:, 1]) > 0).all()
    ma = flat[flat[:, 0] == MASKAUDIO]
    assert set(map(tuple, ma[:, 1:].tolist())) <= set(map(tuple, m[:, 1:].tolist()))
    flat = a2t.reshape(-1, 3)
    m2 = flat[flat[:, 0] == MASK]
    assert len(m2) == ntrg and sorted(m2[:, 2].tolist()) == list(range(ntrg, 2 * ntrg))
    assert not set(m[:, 1].tolist()) & set(m2[:, 1].tolist()), 'the two directions mask different audio spans'
    for g in range(d.ngroups):
        for s in (t2a[g], a2t[g]):
            live = s[s[:, 1] >= 0]
            per = (nseg * nsub) // d.ngroups
            assert (np.diff(live[:, 1]) >= 0).all() and live[:, 1].min() >= g * per and live[:, 1].max() < (g + 1) * per
            assert (s[len(live):] == [0, -1, -1]).all()
    mt = f['audio_text_matching']
    assert mt.shape == (d.seq_len, 3) and (mt[:, 2] == -1).all()
    pools = mt[mt[:, 0] == LTOVPOOL]
    assert 0 < len(pools) <= nseg and pools[:, 1].tolist() == [i * nsub for i in range(len(pools))]
    rt = f['random_text']
    assert rt.shape == (1, d.seq_len, 3)
    mr = rt[0][rt[0][:, 0] == MASK]
    assert len(mr) == budget and mr[:, 2].tolist() == list(range(2 * ntrg, 2 * ntrg + budget))
    assert f['text_spans'].shape == (2 * ntrg + budget, d.span_len)
    assert f['video_src_index'].shape == (nseg,)


@pytest.mark.parametrize('name', ['tiny', 'base'])
def test_dataset_parser_invariants(name):
    cfg = tiny_config() if name == 'tiny' else load_config('base')
    d = Dims(cfg, 1)
    rng = np.random.default_rng(3)
    for k in range(2 if name == 'base' else 6):
        rec = R.make_synthetic_record(cfg, rng)
        f = R.dataset_parser(rec, cfg, rng=np.random.default_rng(100 + k))
        _check_record(f, cfg, d)
        g = R.dataset_parser(rec, cfg, rng=np.random.default_rng(100 + k))
        assert all(np.array_equal(f[k2], g[k2]) for k2 in f if isinstance(f[k2], np.ndarray)), 'same seed, same record'
    # token conservation in the text -> audio direction (when nothing had to be cut): unmasked tokens + masked spans = every token of the record
    ex = R.parse_example(rec)
    all_tok = np.concatenate([np.asarray(ex.get(f'c{i:02d}/tok_ids', []), dtype=np.int64) for i in range(d.nseg)])
    flat = f['text2audio'].reshape(-1, 3)
    kept = flat[(flat[:, 1] >= 0) & (flat[:, 0] != MASK) & (flat[:, 0] != MASKAUDIO), 0]
    spans = f['text_spans'][:d.ntrg]
    assert set(kept.tolist()) <= set(all_tok.tolist())
    if (f['text2audio'][:, -1, 1] == -1).all() and (spans[:, -1] == 0).all():     # no stream was cut by select_tokens, no span by the 15-token limit
        assert len(kept) + int((spans != 0).sum()) == len(all_tok)


def test_fast_image_resize_is_the_float_path_within_one_8bit_step():
    """data.fast_image_resize: frames resampled by PIL's 8-bit path (1.6x fewer host milliseconds per record) -- same draws, same geometry, values within
    a couple of 8-bit steps of the float path (a bf16 step between 0.5 and 1 is 1 / 256)."""
    cfg = tiny_config()
    rec = R.make_synthetic_record(cfg, np.random.default_rng(2), frame_hw=(120, 200))
    c = R.merged_data_config(cfg)
    a = R.dataset_parser(rec, dict(c, fast_image_resize=False), rng=np.random.default_rng(5))
    b = R.dataset_parser(rec, dict(c, fast_image_resize=True), rng=np.random.default_rng(5))
    assert a['images'].shape == b['images'].shape and b['images'].dtype == np.float32
    assert float(np.abs(a['images'] - b['images']).max()) < 0.02 and float(np.abs(a['images'] - b['images']).mean()) < 0.003
    assert all(np.array_equal(a[k], b[k]) for k in ('text2audio', 'audio2text', 'random_text', 'text_spans', 'audio_clips'))


def test_8bit_path_to_bf16_bits_without_float_frames():
    """Round 6: on the 8-bit path a worker maps the final uint8 patches to bf16 bit patterns through a 256-entry table (records.u8_to_bf16_bits) --
    exactly bf16(float32(v) / 255), i.e. what rounding the float form of the same 8-bit frames gives; audio likewise leaves as bf16 bits."""
    assert np.array_equal(R.u8_to_bf16_bits(np.arange(256, dtype=np.uint8)), R.bf16_bits(np.arange(256, dtype=np.float32) / np.float32(255.0)))
    cfg = tiny_config()
    rec = R.make_synthetic_record(cfg, np.random.default_rng(3), frame_hw=(120, 200))
    c = dict(R.merged_data_config(cfg), fast_image_resize=True)
    a = R.dataset_parser(rec, c, rng=np.random.default_rng(5))
    b = R.dataset_parser(rec, c, rng=np.random.default_rng(5), as_bf16_bits=True)
    assert a['images'].dtype == np.float32 and b['images'].dtype == np.uint16 and b['audio_clips'].dtype == np.uint16
    assert np.array_equal(R.bf16_bits(a['images']), b['images']) and np.array_equal(R.bf16_bits(a['audio_clips']), b['audio_clips'])
    assert np.array_equal(a['text2audio'], b['text2audio'])
    # a worker's job: a bf16 batch takes the 8-bit path unless the config says otherwise
    tiv = R.make_token_is_valid()
    merged = R.merged_data_config(cfg)
    j = R._parse_job((rec, 5, merged, tiv, True))
    assert np.array_equal(j['images'], b['images'])
    jf = R._parse_job((rec, 5, dict(merged, fast_image_resize=False), tiv, True))
    assert jf['images'].dtype == np.uint16 and not np.array_equal(jf['images'], b['images'])
    one_step = np.abs(R_bf16_to_f32(jf['images']) - R_bf16_to_f32(b['images'])).max()
    assert one_step <= 1.0 / 255 + 2.0 / 256, one_step


def R_bf16_to_f32(u16):
    return (u16.astype(np.uint32) << np.uint32(16)).view(np.float32)


def test_batch_matches_the_synthetic_layout_and_feeds_the_planner(tmp_path):
    from merlot_reserve_amd.planner import build_plan
    from merlot_reserve_amd.synthetic import make_draws
    cfg = tiny_config()
    cfg['device'] = dict(cfg.get('device', {}), shuffle_buffer_size=4, batch_size=2, n_fns_per_cycle=2)
    rng = np.random.default_rng(5)
    fns = []
    for s in range(2):
        fn = tmp_path / f'train{s:05d}of00002.tfrecord'
        R.write_tfrecord(fn, [R.make_synthetic_record(cfg, rng) for _ in range(3 + s)])
        fns.append(str(fn))
    B = 2
    batches = list(R.make_dataset(cfg, fns, B, is_training=True, seed=11))
    assert len(batches) == 3                                    # 7 records, remainder dropped
    ref = make_batch(cfg, B, seed=0)
    for b in batches:
        assert set(b) == set(ref)
        for k in ref:
            assert tuple(b[k].shape) == tuple(ref[k].shape) and b[k].dtype == ref[k].dtype, k
        d = Dims(cfg, B)
        splits, z = make_draws(cfg, B, seed=1)
        plan = build_plan(b, d, splits, z)           # the planner's own assertions on pointers and counts run here
        assert plan is not None
    again = list(R.make_dataset(cfg, fns, B, is_training=True, seed=11))
    assert all(np.array_equal(x['text2audio'], y['text2audio']) for x, y in zip(batches, again))
    threaded = list(R.make_dataset(cfg, fns, B, is_training=True, seed=11, workers=2))
    assert all(np.array_equal(x['audio2text'], y['audio2text']) and bool((x['images'] == y['images']).all()) for x, y in zip(batches, threaded))
    spawned = list(R.make_dataset(cfg, fns, B, is_training=True, seed=11, workers=2, processes=True))     # worker processes: no torch, no GPU, bf16 on the wire
    assert len(spawned) == 3 and all(np.array_equal(x['text_spans'], y['text_spans']) and bool((x['images'] == y['images']).all()) and
                                     bool((x['audio_clips'] == y['audio_clips']).all()) for x, y in zip(batches, spawned))
    # the driver-level iterator: this rank's shards, one epoch
    cfg['data'] = dict(cfg['data'], train_fns=str(tmp_path / 'train{:05d}of00002.tfrecord'), num_train_files=2)
    assert len(list(R.input_fn_builder(cfg, rank=0, world=1, seed=2, epochs=1))) == 3
    assert len(list(R.input_fn_builder(cfg, rank=1, world=2, seed=2, epochs=1))) == 4      # file 1 alone, batches of batch_size // world = 1


def test_more_than_one_sequence_per_kind():
    """num_text2audio_seqs / num_audio2text_seqs / num_text_seqs > 1 (dataloader.py:500-501, 547-597, 649): every sequence masks its own spans, the
    text pointers count through the sequences, and the planner accepts the batch."""
    from merlot_reserve_amd.planner import build_plan
    from merlot_reserve_amd.synthetic import make_draws
    cfg = tiny_config()
    cfg['data'].update(num_audio2text_seqs=2, num_text2audio_seqs=2, num_text_seqs=2, num_text_seqs_in_record=3)
    d = Dims(cfg, 2)
    rng = np.random.default_rng(8)
    recs = [R.dataset_parser(R.make_synthetic_record(cfg, rng), cfg, rng=np.random.default_rng(k)) for k in range(2)]
    f = recs[0]
    assert f['text2audio'].shape == (d.rows_t2a, d.lang, 3) and f['audio2text'].shape == (d.rows_a2t, d.lang, 3)
    assert f['random_text'].shape == (2, d.seq_len, 3) and f['text_spans'].shape == (d.ntext_spans, d.span_len)
    masks = [s[s[:, 0] == MASK] for s in (f['text2audio'].reshape(-1, 3), f['audio2text'].reshape(-1, 3), f['random_text'].reshape(-1, 3))]
    ptrs = np.concatenate([m[:, 2] for m in masks])
    assert sorted(ptrs.tolist()) == list(range(d.ntext_spans)), 'every text span is pointed at exactly once'
    spans_masked = np.concatenate([masks[0][:, 1], masks[1][:, 1]])
    assert len(set(spans_masked.tolist())) == 4 * d.ntrg1, 'the four sequences mask disjoint audio spans'
    batch = R.handle_batch(recs)
    splits, z = make_draws(cfg, 2, seed=1)
    assert build_plan(batch, d, splits, z)['n_pool'] > 0


@pytest.mark.parametrize('tokens_per_segment', [(0, 0), (40, 60)], ids=['silent_video', 'dense_speech'])
def test_parser_edge_cases(tokens_per_segment):
    """A video without a single token (every masked row stays empty: nothing to donate) and one whose streams overflow lang_seq_len (select_tokens cuts
    them: every MASK survives, rows keep their order, the stream is exactly full)."""
    cfg = tiny_config()
    d = Dims(cfg, 1)
    rng = np.random.default_rng(13)
    for k in range(3):
        rec = R.make_synthetic_record(cfg, rng, tokens_per_segment=tokens_per_segment)
        f = R.dataset_parser(rec, cfg, rng=np.random.default_rng(k))
        _check_record(f, cfg, d)
        t2a = f['text2audio'].reshape(-1, 3)
        if tokens_per_segment[1] == 0:
            assert not (f['text_spans'][:2 * d.ntrg1] != 0).any(), 'no token exists: the audio-stream spans are empty'
            assert set(t2a[t2a[:, 1] >= 0, 0].tolist()) <= {MASK, MASKAUDIO}
        else:
            assert (f['text2audio'][:, -1, 1] >= 0).all(), 'the dense streams fill lang_seq_len'


def _tf_example_classes():
    """tensorflow/core/example/{feature,example}.proto rebuilt with the protobuf runtime (field numbers and types as published): an independent
    encoder / decoder for the wire format records.parse_example / make_example implement by hand."""
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    F = descriptor_pb2.FieldDescriptorProto
    fd = descriptor_pb2.FileDescriptorProto(name='mr_test_example.proto', package='mrtest', syntax='proto3')

    def msg(name, fields):
        m = fd.message_type.add(name=name)
        for fname, num, ftype, label, tname, oneof in fields:
            f = m.field.add(name=fname, number=num, type=ftype, label=label)
            if tname:
                f.type_name = tname
            if oneof is not None:
                f.oneof_index = oneof
        return m
    REP, OPT = F.LABEL_REPEATED, F.LABEL_OPTIONAL
    msg('BytesList', [('value', 1, F.TYPE_BYTES, REP, '', None)])
    msg('FloatList', [('value', 1, F.TYPE_FLOAT, REP, '', None)])
    msg('Int64List', [('value', 1, F.TYPE_INT64, REP, '', None)])
    feat = msg('Feature', [('bytes_list', 1, F.TYPE_MESSAGE, OPT, '.mrtest.BytesList', 0), ('float_list', 2, F.TYPE_MESSAGE, OPT, '.mrtest.FloatList', 0),
                           ('int64_list', 3, F.TYPE_MESSAGE, OPT, '.mrtest.Int64List', 0)])
    feat.oneof_decl.add(name='kind')
    feats = msg('Features', [('feature', 1, F.TYPE_MESSAGE, REP, '.mrtest.Features.FeatureEntry', None)])
    entry = feats.nested_type.add(name='FeatureEntry')
    entry.field.add(name='key', number=1, type=F.TYPE_STRING, label=OPT)
    entry.field.add(name='value', number=2, type=F.TYPE_MESSAGE, label=OPT, type_name='.mrtest.Feature')
    entry.options.map_entry = True
    msg('Example', [('features', 1, F.TYPE_MESSAGE, OPT, '.mrtest.Features', None)])
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName('mrtest.Example'))


def test_example_codec_against_the_protobuf_runtime():
    Example = _tf_example_classes()
    rng = np.random.default_rng(4)
    ints = rng.integers(-2 ** 40, 2 ** 40, size=37)
    floats = rng.normal(size=21).astype(np.float32)
    blob = bytes(rng.integers(0, 256, size=5000, dtype=np.uint8))
    # protobuf encodes -> the hand-written parser reads
    ex = Example()
    ex.features.feature['c00/tok_ids'].int64_list.value.extend(int(v) for v in ints)
    ex.features.feature['c00/tok_start_times'].float_list.value.extend(float(v) for v in floats)
    ex.features.feature['c00/image/encoded'].bytes_list.value.append(blob)
    ex.features.feature['c00/title'].int64_list.value.extend([])
    got = R.parse_example(ex.SerializeToString())
    assert got['c00/tok_ids'].tolist() == ints.tolist() and np.array_equal(got['c00/tok_start_times'], floats)
    assert got['c00/image/encoded'] == [blob] and len(got['c00/title']) == 0
    # the hand-written encoder writes -> protobuf reads
    back = Example()
    back.ParseFromString(R.make_example({'c00/tok_ids': ints, 'c00/tok_start_times': floats, 'c00/image/encoded': blob, 'c00/two': [b'a', b'bc']}))
    f = back.features.feature
    assert list(f['c00/tok_ids'].int64_list.value) == ints.tolist() and list(f['c00/two'].bytes_list.value) == [b'a', b'bc']
    assert np.array_equal(np.array(f['c00/tok_start_times'].float_list.value, dtype=np.float32), floats) and f['c00/image/encoded'].bytes_list.value[0] == blob
    # and a whole synthetic record survives the round trip through the runtime
    cfg = tiny_config()
    rec = R.make_synthetic_record(cfg, rng)
    back = Example()
    back.ParseFromString(rec)
    mine = R.parse_example(rec)
    assert set(back.features.feature) == set(mine)
    assert R.parse_example(back.SerializeToString()).keys() == mine.keys()


def test_shard_feeder_process_delivers_the_same_batches(tmp_path):
    """records.ShardFeeder: the reader in a process of its own (shared-memory slots for the float arrays) yields what input_fn_builder yields in process,
    through loader.PrefetchLoader, and releases its shared memory."""
    import torch
    from merlot_reserve_amd.loader import PrefetchLoader
    cfg = tiny_config()
    rng = np.random.default_rng(21)
    for s in range(2):
        R.write_tfrecord(tmp_path / f'train{s:05d}of00002.tfrecord', [R.make_synthetic_record(cfg, rng) for _ in range(5)])
    cfg['data'] = dict(cfg['data'], train_fns=str(tmp_path / 'train{:05d}of00002.tfrecord'), num_train_files=2)
    cfg['device'] = dict(cfg.get('device', {}), batch_size=2, shuffle_buffer_size=4, n_fns_per_cycle=2)
    ref = list(R.input_fn_builder(cfg, rank=0, world=1, seed=5, epochs=1))
    assert len(ref) == 5
    with R.ShardFeeder(cfg, rank=0, world=1, seed=5, epochs=1, workers=2, slots=3) as feeder:
        names = [m.name for m in feeder.shms]
        got = []
        for b in PrefetchLoader(feeder, 'cpu', depth=2):          # the loader copies a slot's arrays before it asks for the next batch
            got.append({k: (v.clone() if torch.is_tensor(v) else v.copy()) for k, v in b.items()})
    assert len(got) == len(ref)
    for a, b in zip(got, ref):
        assert set(a) == set(b)
        for k in b:
            if torch.is_tensor(b[k]):
                assert a[k].dtype == b[k].dtype == torch.bfloat16 and torch.equal(a[k], b[k]), k
            else:
                assert np.array_equal(a[k], b[k]), k
    from multiprocessing import shared_memory
    for n in names:
        with pytest.raises(FileNotFoundError):
            shared_memory.SharedMemory(name=n)


def test_one_bad_record_costs_one_record(tmp_path, capsys):
    """The advisor's round-5 finding: a corrupt JPEG (PIL.UnidentifiedImageError / OSError) or a truncated protobuf (IndexError) used to end the
    cycle or the feeder; the reference's iterator catches Exception, logs and goes on (pretrain/dataloader.py:948-951).  Seven records, one with a
    garbled frame and one cut in the middle of the message: the five good ones still arrive."""
    cfg = tiny_config()
    cfg['device'] = dict(cfg.get('device', {}), shuffle_buffer_size=1, batch_size=1, n_fns_per_cycle=1)
    rng = np.random.default_rng(8)
    recs = [R.make_synthetic_record(cfg, rng) for _ in range(7)]
    ex = R.parse_example(recs[2])
    jpeg_key = next(k for k, v in ex.items() if isinstance(v, list) and v and isinstance(v[0], bytes) and v[0][:2] == b'\xff\xd8')
    ex[jpeg_key] = [b'\xff\xd8 not a jpeg at all' + bytes(50)]
    recs[2] = R.make_example(ex)
    recs[4] = recs[4][:len(recs[4]) // 2]                                 # the wire format ends inside a field
    fn = tmp_path / 'train00000of00001.tfrecord'
    R.write_tfrecord(fn, recs)
    for kw in ({}, {'workers': 2}):
        got = list(R.make_dataset(cfg, [str(fn)], 1, is_training=False, seed=3, **kw))
        assert len(got) == 5, len(got)
    out = capsys.readouterr().out
    assert out.count('skipping a record') == 4


def test_closing_an_endless_feeder_stops_it(tmp_path):
    """An endless stream (epochs=None) whose consumer stops early: close() must unwind the feeder process -- its stop signal travels as a BaseException,
    so that input_fn_builder's per-cycle `except Exception` (the reference's log-and-continue) cannot swallow it and spin (round 6: it did, for one commit)."""
    import time
    cfg = tiny_config()
    rng = np.random.default_rng(22)
    for s in range(2):
        R.write_tfrecord(tmp_path / f'train{s:05d}of00002.tfrecord', [R.make_synthetic_record(cfg, rng) for _ in range(4)])
    cfg['data'] = dict(cfg['data'], train_fns=str(tmp_path / 'train{:05d}of00002.tfrecord'), num_train_files=2)
    cfg['device'] = dict(cfg.get('device', {}), batch_size=2, shuffle_buffer_size=2, n_fns_per_cycle=2)
    feeder = R.ShardFeeder(cfg, rank=0, world=1, seed=5, epochs=None, workers=2, slots=2)
    names = [m.name for m in feeder.shms]
    it = iter(feeder)
    for _ in range(6):                      # more than one epoch of four batches
        next(it)
    t0 = time.time()
    feeder.close()
    assert time.time() - t0 < 14 and not feeder.proc.is_alive() and feeder.proc.exitcode == 0
    from multiprocessing import shared_memory
    for n in names:
        with pytest.raises(FileNotFoundError):
            shared_memory.SharedMemory(name=n)


def test_record_slots_fall_back_to_the_pipe_without_shared_memory_room(tmp_path, monkeypatch):
    """Parsed records travel through shared-memory record slots when the pool is made of processes (round 6); with no room under /dev/shm (or fewer slots
    than jobs) they travel through the pool's pipe as before -- same batches either way."""
    import os
    cfg = tiny_config()
    cfg['device'] = dict(cfg.get('device', {}), shuffle_buffer_size=2, batch_size=2, n_fns_per_cycle=1)
    rng = np.random.default_rng(31)
    fn = tmp_path / 'train00000of00001.tfrecord'
    R.write_tfrecord(fn, [R.make_synthetic_record(cfg, rng) for _ in range(6)])
    with_slots = list(R.make_dataset(cfg, [str(fn)], 2, is_training=True, seed=7, workers=2, processes=True))
    slots = R._RecordSlots(cfg, 5)
    assert slots.count == 5 and slots.take()[3] == 4 and len(slots.free) == 4
    slots.close()

    class NoRoom:
        f_bavail, f_frsize = 0, 4096
    monkeypatch.setattr(os, 'statvfs', lambda p: NoRoom())
    none = R._RecordSlots(cfg, 5)
    assert none.count == 0 and none.take() is None
    none.close()
    through_pipe = list(R.make_dataset(cfg, [str(fn)], 2, is_training=True, seed=7, workers=2, processes=True))
    assert len(with_slots) == len(through_pipe) == 3
    for a, b in zip(with_slots, through_pipe):
        assert np.array_equal(a['text2audio'], b['text2audio']) and bool((a['images'] == b['images']).all()) and bool((a['audio_clips'] == b['audio_clips']).all())
