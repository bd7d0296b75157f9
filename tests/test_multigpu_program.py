"""First-contact kit for the 8-GPU step (no hardware): Trainer._backward_reduce_update's collective PROGRAM at world = 8 for the base and
the large model, driven with a recording communicator, recording streams and a stub engine on a layout-only ('meta') parameter store --
bucket sizes, order (one program order on one stream, as RCCL requires), the fraction of gradient bytes reduced after the last backward
kernel, and the 240-CU tile plan set around the part of backward that overlaps the buckets and restored after it.
Collectives of the reference being matched: pretrain/pretrain_model.py:290 (all_gather), :329 (pmean of the gradients), :336 (pmean of loss_info)."""
import contextlib

import pytest
import torch

from merlot_reserve_amd import ops
from merlot_reserve_amd.config import Dims, load_config
from merlot_reserve_amd.params import ParamStore
from merlot_reserve_amd.trainer import Trainer


class _Stream:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def wait_stream(self, other):
        self.log.append(('wait', self.name, other.name))


class _Comm:
    capturable, backend = True, 'recording'

    def __init__(self, world, log, cur):
        self.world, self.rank, self.log, self.cur = world, 0, log, cur

    def allreduce_mean(self, t):
        self.log.append(('allreduce_mean', t.numel(), str(t.dtype).replace('torch.', ''), self.cur[0].name, ops.get_option('gemm_cus')))
        return t


class _Engine:
    """Calls layer_done the way TowerEngine.encoder_backward does: once per weight-gradient group, for every layer of the group."""

    def __init__(self, config, B, log, cur, streams):
        self.d, self.log, self.cur = Dims(config, B), log, cur
        self.side_stream = streams['side']

    def _on_side(self, fn):
        prev = self.cur[0]
        self.cur[0] = self.side_stream
        try:
            return fn()
        finally:
            self.cur[0] = prev

    def backward_stage_joint(self):
        self.log.append(('stage', 'joint', self.cur[0].name, ops.get_option('gemm_cus')))

    def backward_stage_audio(self):
        self.log.append(('stage', 'audio', self.cur[0].name, ops.get_option('gemm_cus')))

    def backward_stage_vision(self, layer_done=None):
        group = 2 if self.d.H <= 768 else 4
        self.log.append(('stage', 'vision', self.cur[0].name, ops.get_option('gemm_cus')))
        for hi in range(self.d.Lv, 0, -group):
            for l in range(hi - 1, max(hi - group, 0) - 1, -1):
                layer_done(l)


def _program(monkeypatch, name, world):
    config = load_config(name)
    log, cur = [], [None]
    streams = {k: _Stream(k, log) for k in ('main', 'side', 'comm')}
    cur[0] = streams['main']
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda *a, **k: cur[0])

    @contextlib.contextmanager
    def use(stream):
        prev, cur[0] = cur[0], stream
        try:
            yield
        finally:
            cur[0] = prev
    monkeypatch.setattr(torch.cuda, 'stream', use)
    monkeypatch.setattr(ops, 'nan_to_num_', lambda g: log.append(('nan_to_num', g.numel(), cur[0].name)))
    tr = Trainer.__new__(Trainer)
    tr.config, tr.B, tr.rank, tr.world = config, 4, 0, world
    tr.params = ParamStore(config, 'meta', init=False)             # layout only: offsets, tower ranges, no storage
    tr.engine = _Engine(config, 4, log, cur, streams)
    tr.comm, tr.use_comm, tr.comm_stream = _Comm(world, log, cur), True, streams['comm']

    class _State:
        def apply_range(self, lo, hi):
            log.append(('adam', lo, hi, cur[0].name))
    tr.state = _State()
    tr._make_buckets()
    before = ops.get_option('gemm_cus')
    tr._backward_reduce_update(update=True)
    assert ops.get_option('gemm_cus') == before, 'the 240-CU plan must be restored after backward'
    return tr, log


@pytest.mark.parametrize('name,nparams,last_frac', [('base', 316.4e6, (0.05, 0.12)), ('large', 853.1e6, (0.05, 0.14))])
def test_collective_program_at_world_8(monkeypatch, name, nparams, last_frac):
    tr, log = _program(monkeypatch, name, 8)
    keys = [str(b[0]) for b in tr.buckets]
    Lv = tr.engine.d.Lv
    assert keys == ['joint', 'audio', f"('vision', {Lv - Lv // 3})", f"('vision', {Lv - 2 * (Lv // 3)})", 'vision_end']
    # the buckets tile the flat gradient buffer exactly, each a whole number of Adam blocks
    assert sum(hi - lo for _, lo, hi in tr.buckets) == tr.params.total and abs(tr.params.total - nparams) / nparams < 0.01
    ar = [e for e in log if e[0] == 'allreduce_mean']
    # reduced in the order backward finishes them; audio rides behind the first vision bucket (it rarely ends before it)
    sizes = {str(k): hi - lo for k, lo, hi in tr.buckets}
    assert [e[1] for e in ar] == [sizes['joint'], sizes[keys[2]], sizes['audio'], sizes[keys[3]], sizes['vision_end']]
    assert tr.bucket_log == ['joint', tr.buckets[2][0], 'audio', tr.buckets[3][0], 'vision_end']
    assert all(e[2] == 'bfloat16' and e[3] == 'comm' for e in ar), 'every all-reduce on ONE stream in ONE program order, bf16 (pmean on bf16 grads, P:329)'
    # nan_to_num -> all-reduce -> Adam of the same range, per bucket, on the comm stream (P:328-329, O:180-190)
    seq = [e for e in log if e[0] in ('nan_to_num', 'allreduce_mean', 'adam')]
    for i in range(0, len(seq), 3):
        n2n, red, adam = seq[i:i + 3]
        assert (n2n[0], red[0], adam[0]) == ('nan_to_num', 'allreduce_mean', 'adam') and n2n[1] == red[1] == adam[2] - adam[1]
    # only the last bucket is reduced after the last backward kernel
    exposed = sizes['vision_end'] / tr.params.total
    assert last_frac[0] < exposed < last_frac[1], exposed
    # the 240-CU tile plan covers the stages that run beside a bucket in flight (audio + vision), not the joint stage before the first
    stages = {e[1]: e[3] for e in log if e[0] == 'stage'}
    assert stages == {'joint': 0, 'audio': 240, 'vision': 240}
    assert [e[4] for e in ar] == [0, 240, 240, 240, 240], 'the first bucket is handed over before the plan is switched'
    # the comm stream waits for the producer of each bucket; main joins side and comm at the end
    waits = [e for e in log if e[0] == 'wait']
    assert waits[-2:] == [('wait', 'main', 'side'), ('wait', 'main', 'comm')]
    assert ('wait', 'comm', 'side') in waits, "the audio bucket's producer is the side stream"


def test_single_rank_program_leaves_the_tile_plan_alone(monkeypatch):
    tr, log = _program(monkeypatch, 'base', 1)
    assert {e[1]: e[3] for e in log if e[0] == 'stage'} == {'joint': 0, 'audio': 0, 'vision': 0}
