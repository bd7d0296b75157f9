"""train_step (pretrain/pretrain_model.py:306-340) + construct_train_state (pretrain/optimization.py:158-195) on top
of the engine: one process per GPU; data parallelism = RCCL over xGMI (merlot_reserve_amd/dist.py).

Per step and rank:  plan (host) -> forward -> all-gather of the packed contrastive embeddings -> loss + dL/dE ->
reduce-scatter of dL/dE_all -> backward, during which every finished gradient BUCKET (five for the stock models) is
handed to a third stream: nan_to_num -> all-reduce(mean, bf16, like pmean at :329) -> fused bf16 Adam + decay + schedule +
apply on that range of the flat buffers, while the main / side streams continue with the rest of backward.  With the
library's RCCL communicator (dist.NativeComm) the collectives are ordinary stream-ordered launches, so the whole step,
collectives included, is ONE hipGraph.
"""
import os

import numpy as np
import torch

from . import ops
from .engine import PretrainEngine
from .params import ParamStore
from .planner import build_plan
from .synthetic import make_draws


def lr_scale_linearwarmup_cosinedecay(step, num_warmup_steps, num_train_steps, final_lr_scale=0.1):
    """pretrain/optimization.py:117-137, evaluated in float32 like jnp on an int32 step."""
    f = np.float32
    step = f(step)
    if step < num_warmup_steps:
        return float(step / f(num_warmup_steps))
    post = (step - f(num_warmup_steps)) / f(num_train_steps - num_warmup_steps + 1.0)
    post = min(post, f(1.0))
    post = f(1.0) - (f(1.0) - np.cos(f(np.pi) * post, dtype=np.float32)) / f(2.0)
    return float(f(final_lr_scale) + f(1.0 - final_lr_scale) * post)


class TrainState:
    """Counterpart of flax TrainState for this path: step counter, parameters (fp32 master + bf16 working copy),
    optimizer state (bf16 mu, cube-coded bf16 nu) -- all inside the ParamStore's flat buffers."""

    def __init__(self, params, opt_config):
        self.params = params
        self.opt_config = dict(opt_config)
        self.step = 0
        self.f32_grads = False          # use_bfloat16_grads = False: the chain reads params.grad32 (set by Trainer(bf16_grads=False))
        self.shards = None              # zero.MomentShards: mu / nu partitioned over the ranks (Trainer(shard_optimizer=True))

    def _scalars(self):
        """(sched, neg_lr, bias_corr1, bias_corr2) of the current step: scale_by_schedule uses its own count, evaluated
        BEFORE the increment (first update is zero); no bias correction unless the config asks (optimization.py:177)."""
        oc = self.opt_config
        assert oc.get('use_bfloat16_adam', True), 'only the bf16-state Adam of the reference configs is implemented'
        sched = lr_scale_linearwarmup_cosinedecay(self.step, oc['num_warmup_steps'], oc['num_train_steps'],
                                                  oc.get('final_lr_scale', 0.02))
        b1, b2 = oc.get('beta_1', 0.9), oc.get('beta_2', 0.98)
        bc1 = bc2 = 1.0
        if oc.get('do_bias_correction', False):
            bc1, bc2 = 1.0 - b1 ** (self.step + 1), 1.0 - b2 ** (self.step + 1)
        return sched, -oc['learning_rate'], bc1, bc2

    def apply_gradients(self):
        """optax chain of optimization.py:180-190 + apply_updates, one fused launch over the flat buffers."""
        oc, p = self.opt_config, self.params
        if self.shards is not None:      # partitioned moments: bucket by bucket, each rank its chunk (zero.py); a COLLECTIVE
            self.prepare_step()
            for key in self.shards.table:
                self.shards.update(key, self.apply_shard)
            self.finish_step()
            return
        if self.f32_grads:               # fp32 gradients: the device-scalar form of the same chain over the whole buffer
            self.prepare_step()
            self.apply_range(0, p.total)
            self.finish_step()
            return
        sched, neg_lr, bc1, bc2 = self._scalars()
        ops.adam_bf16_update(p.master, p.work, p.grad, p.mu, p.nu, p.decay_flags, oc.get('beta_1', 0.9), oc.get('beta_2', 0.98),
                             oc.get('eps', 1e-8), oc['weight_decay_rate'], sched, neg_lr, bc1, bc2)
        p.update_transposed()
        self.step += 1

    # ---- the same update in pieces: per-step scalars in a device vector (so the launches can sit inside a hipGraph) and
    # one launch per finished range of the flat gradient buffer, overlapped with the rest of backward (trainer.Trainer)
    def prepare_step(self):
        """Write this step's scalars into the device vector the captured Adam launches read: a ring of two pinned host
        vectors, each rewritten only after the copy that last read it has completed, so the host may run ahead of the stream."""
        dev = self.params.device
        if getattr(self, 'hyper', None) is None:
            self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
            self._hyper_host = [torch.zeros(4, dtype=torch.float32, pin_memory=dev.type == 'cuda') for _ in range(2)]
            self._hyper_ev, self._hyper_turn = [None, None], 0
        i = self._hyper_turn
        self._hyper_turn ^= 1
        if self._hyper_ev[i] is not None:
            self._hyper_ev[i].synchronize()
        sched, neg_lr, bc1, bc2 = self._scalars()
        self._hyper_host[i].copy_(torch.tensor([sched, neg_lr, 1.0 / bc1, 1.0 / bc2], dtype=torch.float32))
        self.hyper.copy_(self._hyper_host[i], non_blocking=True)
        if dev.type == 'cuda':
            self._hyper_ev[i] = torch.cuda.Event()
            self._hyper_ev[i].record()

    def apply_range(self, lo, hi, mu=None, nu=None, transposed=True):
        """mu / nu: the moments of [lo, hi) when they do not live at p.mu[lo:hi] (zero.MomentShards)."""
        oc, p = self.opt_config, self.params
        assert lo % 2048 == 0 and hi % 2048 == 0
        grad = p.grad32 if self.f32_grads else p.grad
        ops.adam_bf16_update_dev(p.master[lo:hi], p.work[lo:hi], grad[lo:hi], p.mu[lo:hi] if mu is None else mu,
                                 p.nu[lo:hi] if nu is None else nu, None,
                                 p.decay_flags[lo // 2048:hi // 2048], oc.get('beta_1', 0.9), oc.get('beta_2', 0.98),
                                 oc.get('eps', 1e-8), oc['weight_decay_rate'], self.hyper)
        if transposed:
            p.update_transposed(lo, hi)

    def apply_shard(self, lo, hi, mu, nu):
        self.apply_range(lo, hi, mu=mu, nu=nu, transposed=False)      # (MomentShards.update rewrites the bucket's transposed copies)

    def finish_step(self):
        self.step += 1

    # ---- checkpoint form: flax `to_state_dict(TrainState)` of the reference's optax chain (optimization.py:180-195):
    # opt_state = (ScaleByAdamState{count, mu, nu}, add_decayed_weights (empty), ScaleByScheduleState{count}, scale (empty))
    # serialised as dicts keyed '0'..'3' (merlot_reserve_amd/checkpoint.py writes / reads the msgpack file)
    def state_dict(self):
        p = self.params
        mu, nu = (p.mu, p.nu) if self.shards is None else self.shards.full_moments()      # (partitioned: a collective)
        return {'step': self.step, 'params': p.master_tree(),
                'opt_state': {'0': {'count': torch.tensor(self.step, dtype=torch.int32), 'mu': p._to_tree(mu), 'nu': p._to_tree(nu)},
                              '1': {}, '2': {'count': torch.tensor(self.step, dtype=torch.int32)}, '3': {}}}

    def load_state_dict(self, sd, reset_schedule=False):
        """reset_schedule = True is the resolution-adaptation restart (pretrain/train_fixres.py:94-113): parameters and the
        Adam moments are kept, the step and the schedule's count start again from 0."""
        p = self.params
        p.load_tree(sd['params'])
        self.step = int(sd.get('step', 0))
        opt = sd.get('opt_state')
        if opt:
            adam = opt['0']
            hosts = []
            for tree in (adam['mu'], adam['nu']):
                host = torch.zeros(p.total, dtype=torch.bfloat16)
                for name, fshape, *_ in p.specs:
                    o, n = p.offsets[name]
                    leaf = tree
                    for k in name.split('/'):
                        leaf = leaf[k]
                    host[o:o + n] = leaf.reshape(-1).to(torch.bfloat16)
                hosts.append(host)
            if self.shards is None:
                p.mu.copy_(hosts[0])
                p.nu.copy_(hosts[1])
            else:
                self.shards.store_moments(hosts[0], hosts[1])
            self.step = int(adam.get('count', self.step))
        if reset_schedule:
            self.step = 0


def construct_train_state(opt_config, params):
    return TrainState(params, opt_config)


class Trainer:
    f32 = False          # (instances built by __init__ set it from bf16_grads)
    shards = None        # (zero.MomentShards when built with shard_optimizer=True)

    def __init__(self, config, B, device, rank=0, world=1, seed=0, comm=None, bf16_grads=True, shard_optimizer=False):
        """bf16_grads = False is the reference's use_bfloat16_grads = False step (pretrain/pretrain_model.py:323-333; train.py:61-67): the
        fp32 master parameters are differentiated, the gradients stay fp32 through nan_to_num / pmean and enter the Adam chain as
        fp32 -- the fp32 program of the engine (mr_f32_* kernels, eager, several times slower than the bf16 step: the correctness
        path, not the benchmarked one).  Data parallel through either communicator (fp32 all-gather / reduce-scatter of the contrastive
        embeddings, fp32 all-reduce of the gradient buckets).
        shard_optimizer = True: the Adam moments are partitioned over the ranks of `comm` (pretrain/train_fixres.py:178-199; zero.py)."""
        self.config, self.B, self.rank, self.world = config, B, rank, world
        self.device = torch.device(device)
        self.f32 = not bf16_grads
        self.params = ParamStore(config, self.device, seed=seed, with_optimizer=not shard_optimizer)          # same seed on every rank: replicated init
        self.state = construct_train_state(config['optimizer'], self.params)
        self.state.f32_grads = self.f32
        self.engine = PretrainEngine(config, B, self.params, self.device, rank=rank, world=world,
                                     dtype=torch.float32 if self.f32 else torch.bfloat16, train=True)
        self.comm = comm
        # the collective path runs whenever a comm is given -- also with a single rank, which is how the RCCL calls
        # themselves are exercised on a 1-GPU box (tests/test_dist_gpu.py)
        assert world == 1 or comm is not None
        self.use_comm = comm is not None
        self.comm_stream = torch.cuda.Stream(device=self.device) if self.device.type == 'cuda' else None
        self.graph = None
        self._make_buckets()
        self.shards = None
        if shard_optimizer:
            assert self.use_comm, 'shard_optimizer needs a communicator (its world may be 1)'
            from .zero import MomentShards
            self.shards = self.state.shards = MomentShards(self.params, self.buckets, comm)
        if self.use_comm:
            assert comm.world == world and comm.rank == rank
            R, H = self.engine.R, self.engine.d.H
            z = lambda *s: torch.zeros(*s, dtype=self.engine.dtype, device=self.device)
            self.E_all, self.dE_all, self.dE_red = z(world, R, H), z(world, R, H), z(R, H)
            self.metrics = torch.zeros(8, dtype=torch.float32, device=self.device)

    def _make_buckets(self):
        """Gradient buckets = ranges of the flat buffers in the order backward makes them final (params.py lays the towers
        out in that order, and a tower's layers in ascending order, which backward walks downwards):
          'joint'   [scales, head, span, joint, token]        final after backward_stage_joint
          'audio'   the audio tower                            final when the side stream's audio backward ends
          ('vision', l)  vision layers >= l (+ final_ln, cls_proj, attention pool), down to the previous cut
          'vision_end'   the rest of the vision tower (first layers, pre_ln, cls, patch embedding)
        The vision tower (the last to finish) is cut at 2/3 and 1/3 of its depth, so only ~1/3 of its gradients -- < 10 % of
        all gradient bytes for base and large -- is reduced after the last backward kernel."""
        tr, offs, Lv = self.params.tower_ranges, self.params.offsets, self.engine.d.Lv
        a0, a1 = tr['audio_encoder']
        v0, v1 = tr['vision_encoder']
        assert a1 == v0 and v1 == self.params.total
        cuts = sorted({l for l in (Lv - Lv // 3, Lv - 2 * (Lv // 3)) if 0 < l < Lv}, reverse=True)
        lo_of = lambda l: offs[f'vision_encoder/transformer/layer_{l:02d}/pre_attn_ln/scale'][0]
        self.buckets = [('joint', 0, a0), ('audio', a0, a1)]
        hi = v1
        for l in cuts:
            self.buckets.append((('vision', l), lo_of(l), hi))
            hi = lo_of(l)
        self.buckets.append(('vision_end', v0, hi))
        self.vision_cuts = cuts
        assert all(lo % 2048 == 0 and hi_ % 2048 == 0 and lo < hi_ for _, lo, hi_ in self.buckets)
        assert sum(hi_ - lo for _, lo, hi_ in self.buckets) == self.params.total

    def plan(self, batch, draws=None):
        if draws is None:
            seed = int(batch['audio2text/text_ptr'].astype(np.uint32).sum() % (2 ** 31))   # pretrain_model.py:96
            draws = make_draws(self.config, self.B, seed=seed)
        return build_plan(batch, self.engine.d, draws[0], draws[1])

    # ---- the step as a launch sequence (eager, or captured once into a hipGraph and replayed) ----
    def _loss_and_exchange(self):
        eng = self.engine
        if self.use_comm:
            self.comm.gather_embeddings(eng.E, self.E_all)                  # pretrain_model.py:290
            eng.loss_and_grad_outputs(self.E_all, self.dE_all)
            self.comm.scatter_grad(self.dE_all, self.dE_red)                # transpose of the all-gather
            ops.add_(eng.dE.view(-1), self.dE_red.view(-1))
        else:
            eng.loss_and_grad_outputs()

    def _finish_bucket(self, key, producer, update):
        """The bucket's gradients are final on stream `producer`: on the comm stream, behind them, nan_to_num + all-reduce
        (pretrain_model.py:328-329) and the optimizer update of that range, while the producer goes on with backward.
        Buckets are enqueued in ONE program order on ONE stream, the same on every rank, as RCCL requires."""
        _, lo, hi = next(b for b in self.buckets if b[0] == key)
        self.bucket_log.append(key)
        # (MR_NO_COMM_STREAM=1, bench.py's exclusive timing pass: the bucket's kernels are issued in line -- beside a bucket's Adam kernel a
        # whole-CU GEMM of either tower waits for CUs, and its HIP-event duration would include that wait)
        cs = producer if os.environ.get('MR_NO_COMM_STREAM') == '1' else self.comm_stream
        cs.wait_stream(producer)
        tl = getattr(self, 'bucket_timeline', None)          # (eager steps only) HIP events around the bucket's collective: bench.py --gpus N

        def ev():
            if tl is None:
                return None
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        with torch.cuda.stream(cs):
            g = (self.params.grad32 if self.f32 else self.params.grad)[lo:hi]
            e_ready = ev()
            if self.use_comm:
                ops.nan_to_num_(g)
                self.comm.allreduce_mean(g)
            e_reduced = ev()
            if update and self.shards is not None:
                self.shards.update(key, self.state.apply_shard)
            elif update:
                self.state.apply_range(lo, hi)
            e_updated = ev()
        if tl is not None:
            tl.append((str(key), (hi - lo) * g.element_size() / 1e6, e_ready, e_reduced, e_updated))

    def _backward_reduce_update(self, update=True):
        eng = self.engine
        main = torch.cuda.current_stream()
        self.bucket_log = []
        eng.backward_stage_joint()
        self._finish_bucket('joint', main, update)
        # from here to the end of backward a bucket's all-reduce is in flight beside the GEMMs: leave its kernel two CUs per XCD
        with ops.gemm_cus(self.world if self.use_comm else 1):
            eng.side_stream.wait_stream(main)
            eng._on_side(eng.backward_stage_audio)
            audio_stream = main if os.environ.get('MR_NO_SIDE_STREAM') == '1' else eng.side_stream     # (A/B switch: issued in line)
            pending = ['audio']                     # enqueued behind the first vision bucket: audio rarely ends before it

            def layer_done(l):
                if l in self.vision_cuts:
                    self._finish_bucket(('vision', l), main, update)
                    if pending:
                        self._finish_bucket(pending.pop(), audio_stream, update)
            eng.backward_stage_vision(layer_done=layer_done)
            if pending:
                self._finish_bucket(pending.pop(), audio_stream, update)
            self._finish_bucket('vision_end', main, update)
        main.wait_stream(eng.side_stream)
        main.wait_stream(self.comm_stream)

    def _program(self, images, audio):
        self.engine.forward_device(images, audio)
        self._loss_and_exchange()
        self._backward_reduce_update(update=True)

    # ---- pieces, for tests and for the reference-API adapter (eager) ----
    def forward_and_loss(self, batch, plan=None, draws=None):
        eng = self.engine
        eng.forward(batch, plan=plan if plan is not None else self.plan(batch, draws))
        self._loss_and_exchange()
        return eng.loss_acc

    def backward_and_reduce(self, update=True):
        """Backward from the engine's current dE, bucket by bucket all-reduced; update=False leaves the averaged gradients in
        params.grad and the parameters untouched."""
        if update:
            self.state.prepare_step()
        self._backward_reduce_update(update=update)
        if update:
            self.state.finish_step()

    def train_step(self, batch, plan=None, draws=None):
        eng = self.engine
        if plan is None:
            plan = self.plan(batch, draws)
        self.state.prepare_step()
        eng.set_plan(plan)
        self._program(batch['images'], batch['audio_clips'])
        self.state.finish_step()
        return eng.loss_acc

    def train_step_timeline(self, batch, plan=None):
        """One EAGER step with HIP events around every gradient bucket's collective (events cannot be read back from a replayed graph):
        returns {'backward_ms': start of backward -> its last kernel on the main stream, 'buckets': [{bucket, mbytes, ready_ms (its
        gradients final = the all-reduce may start), reduced_ms, updated_ms}]}, times from the start of backward -- what the first run
        on more than one GPU needs to see whether the all-reduces hide under backward (pretrain/pretrain_model.py:329 is the pmean
        they implement).  A diagnostic: the step it times is a real optimizer step."""
        eng = self.engine
        if plan is None:
            plan = self.plan(batch)
        self.state.prepare_step()
        eng.set_plan(plan)
        eng.forward_device(batch['images'], batch['audio_clips'])
        self._loss_and_exchange()
        self.bucket_timeline = []
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        try:
            self._backward_reduce_update(update=True)
            t1.record()
            tl = self.bucket_timeline
        finally:
            self.bucket_timeline = None
        self.state.finish_step()
        torch.cuda.synchronize()
        return {'backward_ms': t0.elapsed_time(t1),
                'buckets': [{'bucket': k, 'mbytes': round(mb, 1), 'ready_ms': round(t0.elapsed_time(a), 3), 'reduced_ms': round(t0.elapsed_time(b), 3),
                             'updated_ms': round(t0.elapsed_time(c), 3)} for k, mb, a, b, c in tl]}

    # ---- hipGraph path: the step is a fixed launch sequence over fixed buffers; capture it once, replay per step ----
    def capture(self, batch):
        """Capture the WHOLE step -- forward, collectives, loss, backward, bucket reductions, optimizer -- into one hipGraph.
        Call after at least one eager train_step (buffers, plan capacities and the communicator's lazy state exist).
        A torch.distributed comm cannot be captured (its collectives are host-scheduled): such a trainer steps eagerly."""
        assert not self.use_comm or self.comm.capturable, 'only the library RCCL communicator (dist.NativeComm) can be captured'
        assert not self.f32, 'the fp32-gradient step runs eagerly (train_step)'
        eng = self.engine
        self.images_in = torch.zeros_like(batch['images'])
        self.audio_in = torch.zeros_like(batch['audio_clips'])
        eng.plan_frozen = True
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):      # other threads (RCCL proxies, torch's watchdog) may call HIP meanwhile
            self._program(self.images_in, self.audio_in)
        self.graph = g

    def train_step_graph(self, batch, plan):
        eng = self.engine
        self.state.prepare_step()
        self.images_in.copy_(batch['images'], non_blocking=True)
        self.audio_in.copy_(batch['audio_clips'], non_blocking=True)
        eng.set_plan(plan)
        self.graph.replay()
        self.state.finish_step()
        return eng.loss_acc

    def loss_info(self, reduce=True):
        """Host dict of the last step's losses.  With a comm and reduce=True it is the mean over ranks, like the reference's
        pmean of loss_info (pretrain_model.py:336) -- a COLLECTIVE: every rank must call it."""
        info = self.engine.loss_info()
        if self.use_comm and reduce and self.world > 1:
            keys = sorted(info)
            self.metrics[:len(keys)].copy_(torch.tensor([info[k] for k in keys], dtype=torch.float32))
            self.comm.allreduce_mean_f32(self.metrics)
            info = dict(zip(keys, self.metrics[:len(keys)].tolist()))
        return info
