"""train_step (pretrain/pretrain_model.py:306-340) + construct_train_state (pretrain/optimization.py:158-195) on top
of the engine: one process per GPU; data parallelism = RCCL over xGMI through torch.distributed ("nccl" backend).

Per step and rank:  plan (host) -> forward -> all-gather of the packed contrastive embeddings -> loss + dL/dE ->
reduce-scatter of dL/dE_all -> backward (tower gradient buckets are all-reduced (mean, bf16, like pmean at :329) on a side
stream as soon as the tower's backward finishes) -> fused nan_to_num + bf16 Adam + decay + schedule + apply.
"""
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
        sched, neg_lr, bc1, bc2 = self._scalars()
        ops.adam_bf16_update(p.master, p.work, p.grad, p.mu, p.nu, p.decay_flags, oc.get('beta_1', 0.9), oc.get('beta_2', 0.98),
                             oc.get('eps', 1e-8), oc['weight_decay_rate'], sched, neg_lr, bc1, bc2)
        self.step += 1

    # ---- the same update in pieces: per-step scalars in a device vector (so the launches can sit inside a hipGraph) and
    # one launch per finished range of the flat gradient buffer, overlapped with the rest of backward (trainer.Trainer)
    def prepare_step(self):
        if getattr(self, 'hyper', None) is None:
            self.hyper = torch.zeros(4, dtype=torch.float32, device=self.params.device)
        sched, neg_lr, bc1, bc2 = self._scalars()
        # a fresh pageable source every step (like the plan's index lists): the runtime stages it before returning, so the
        # host may run ahead of the stream without a later step's scalars overwriting an earlier step's pending copy
        self.hyper.copy_(torch.tensor([sched, neg_lr, 1.0 / bc1, 1.0 / bc2], dtype=torch.float32), non_blocking=True)

    def apply_range(self, lo, hi):
        oc, p = self.opt_config, self.params
        assert lo % 2048 == 0 and hi % 2048 == 0
        ops.adam_bf16_update_dev(p.master[lo:hi], p.work[lo:hi], p.grad[lo:hi], p.mu[lo:hi], p.nu[lo:hi], None,
                                 p.decay_flags[lo // 2048:hi // 2048], oc.get('beta_1', 0.9), oc.get('beta_2', 0.98),
                                 oc.get('eps', 1e-8), oc['weight_decay_rate'], self.hyper)

    def finish_step(self):
        self.step += 1

    # ---- checkpoint form: flax `to_state_dict(TrainState)` of the reference's optax chain (optimization.py:180-195):
    # opt_state = (ScaleByAdamState{count, mu, nu}, add_decayed_weights (empty), ScaleByScheduleState{count}, scale (empty))
    # serialised as dicts keyed '0'..'3' (merlot_reserve_amd/checkpoint.py writes / reads the msgpack file)
    def state_dict(self):
        p = self.params
        return {'step': self.step, 'params': p.master_tree(),
                'opt_state': {'0': {'count': torch.tensor(self.step, dtype=torch.int32), 'mu': p._to_tree(p.mu), 'nu': p._to_tree(p.nu)},
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
            for flat, tree in ((p.mu, adam['mu']), (p.nu, adam['nu'])):
                host = torch.zeros(p.total, dtype=torch.bfloat16)
                for name, fshape, *_ in p.specs:
                    o, n = p.offsets[name]
                    leaf = tree
                    for k in name.split('/'):
                        leaf = leaf[k]
                    host[o:o + n] = leaf.reshape(-1).to(torch.bfloat16)
                flat.copy_(host)
            self.step = int(adam.get('count', self.step))
        if reset_schedule:
            self.step = 0


def construct_train_state(opt_config, params):
    return TrainState(params, opt_config)


class Trainer:
    def __init__(self, config, B, device, rank=0, world=1, seed=0, comm=None):
        self.config, self.B, self.rank, self.world = config, B, rank, world
        self.device = torch.device(device)
        self.params = ParamStore(config, self.device, seed=seed)          # same seed on every rank: replicated init
        self.state = construct_train_state(config['optimizer'], self.params)
        self.engine = PretrainEngine(config, B, self.params, self.device, rank=rank, world=world)
        self.comm = comm
        tr_ = self.params.tower_ranges
        a0_, a1_ = tr_['audio_encoder']
        v0_, v1_ = tr_['vision_encoder']
        assert a1_ == v0_ and v1_ == self.params.total
        # the three ranges of the flat buffers whose gradients become final one backward stage after the other
        self.ranges = [(0, a0_), (a0_, a1_), (v0_, v1_)]
        # the collective path runs whenever a Comm is given -- also with a single rank, which is how the RCCL calls
        # themselves are exercised on a 1-GPU box (tests/test_dist_gpu.py)
        assert world == 1 or comm is not None
        self.use_comm = comm is not None
        if self.use_comm:
            assert comm.world == world and comm.rank == rank
            assert (B * self.engine.d.ntrg) % 8 == 0, 'world > 1 needs 8-aligned contrastive blocks (even B for the stock configs)'
            R, H = self.engine.R, self.engine.d.H
            z = lambda *s: torch.zeros(*s, dtype=torch.bfloat16, device=self.device)
            self.E_all, self.dE_all, self.dE_red = z(world, R, H), z(world, R, H), z(R, H)
            # gradient buckets = the three backward stages' ranges of the flat buffer (params.py lays towers out in
            # the order backward finishes them)
            tr = self.params.tower_ranges
            a0, a1 = tr['audio_encoder']
            v0, v1 = tr['vision_encoder']
            assert a1 == v0 and v1 == self.params.total
            # two gradient buckets: [scales, head, span, joint, token] -- reduced while the audio and vision towers' backward
            # runs (both towers concurrently, on the two streams, as in the single-rank step) -- and [audio, vision]
            self.buckets = [self.params.grad[:a0], self.params.grad[a0:v1]]
            self.bucket_ranges = [(0, a0), (a0, v1)]

    def plan(self, batch, draws=None):
        if draws is None:
            seed = int(batch['audio2text/text_ptr'].astype(np.uint32).sum() % (2 ** 31))   # pretrain_model.py:96
            draws = make_draws(self.config, self.B, seed=seed)
        return build_plan(batch, self.engine.d, draws[0], draws[1])

    # ---- hipGraph path: the step is a fixed launch sequence over fixed buffers; capture it once, replay per step ----
    def capture(self, batch):
        """Capture forward / loss / backward into hipGraphs (split at the collectives when world > 1, which stay eager).
        Call after at least one eager train_step (buffers and plan capacities exist)."""
        eng, d = self.engine, self.engine.d
        self.images_in = torch.zeros_like(batch['images'])
        self.audio_in = torch.zeros_like(batch['audio_clips'])
        eng.plan_frozen = True
        torch.cuda.synchronize()
        self.graphs = []
        segs = [lambda: eng.forward_device(self.images_in, self.audio_in)]
        if self.use_comm:
            b = self.buckets
            segs += [lambda: eng.loss_and_grad_outputs(self.E_all, self.dE_all),
                     lambda: (ops.add_(eng.dE.view(-1), self.dE_red.view(-1)), eng.backward_stage_joint(), ops.nan_to_num_(b[0])),
                     lambda: (self._backward_audio_vision(), ops.nan_to_num_(b[1]))]
        else:
            segs = [lambda: (eng.forward_device(self.images_in, self.audio_in), eng.loss_and_grad_outputs(), self._backward_and_update())]
        pool = None
        for fn in segs:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool, capture_error_mode='thread_local'):     # RCCL's watchdog thread polls events meanwhile
                fn()
            pool = g.pool()
            self.graphs.append(g)

    def _backward_and_update(self):
        """Single-rank backward with the optimizer folded in: each range of the flat buffers is updated as soon as its
        gradients are final -- [scales, head, span, joint, token] on the side stream ahead of the audio tower's backward,
        the audio range behind it (both beside the vision tower's backward on the main stream), the vision range last --
        so only the last launch (27 % of the parameters) is exposed.  A later stage never reads an earlier range's weights."""
        eng, st = self.engine, self.state
        eng.backward_stage_joint()
        main = torch.cuda.current_stream()
        eng.side_stream.wait_stream(main)

        def side_work():
            st.apply_range(*self.ranges[0])
            eng.backward_stage_audio()
            st.apply_range(*self.ranges[1])
        eng._on_side(side_work)
        eng.backward_stage_vision()
        st.apply_range(*self.ranges[2])
        main.wait_stream(eng.side_stream)

    def _backward_audio_vision(self):
        """The audio tower's backward on the side stream beside the vision tower's on the main stream (fork / join)."""
        eng = self.engine
        main = torch.cuda.current_stream()
        eng.side_stream.wait_stream(main)
        eng._on_side(eng.backward_stage_audio)
        eng.backward_stage_vision()
        main.wait_stream(eng.side_stream)

    def train_step_graph(self, batch, plan):
        eng = self.engine
        self.state.prepare_step()
        self.images_in.copy_(batch['images'], non_blocking=True)
        self.audio_in.copy_(batch['audio_clips'], non_blocking=True)
        eng.set_plan(plan)
        if self.use_comm:
            self.graphs[0].replay()
            self.comm.gather_embeddings(eng.E, self.E_all)
            self.graphs[1].replay()
            self.comm.scatter_grad(self.dE_all, self.dE_red)
            works = []
            for k in range(2):       # bucket 0 is all-reduced (pretrain_model.py:329) while the audio / vision towers' backward runs
                self.graphs[2 + k].replay()
                works.append(self.comm.allreduce_mean_async(self.buckets[k]))
            self._update_buckets(works)
        else:
            self.graphs[0].replay()
        self.state.finish_step()
        return eng.loss_acc

    def _update_buckets(self, works):
        """Every backward stage is already enqueued: wait for each bucket's all-reduce in turn and update its range while
        the later buckets are still being reduced."""
        for k, w in enumerate(works):
            if w is not None:
                w.wait()
            self.state.apply_range(*self.bucket_ranges[k])

    def train_step(self, batch, plan=None, draws=None):
        eng = self.engine
        if plan is None:
            plan = self.plan(batch, draws)
        self.state.prepare_step()
        eng.forward(batch, plan=plan)
        if self.use_comm:
            self.comm.gather_embeddings(eng.E, self.E_all)                  # pretrain_model.py:290
            eng.loss_and_grad_outputs(self.E_all, self.dE_all)
            self.comm.scatter_grad(self.dE_all, self.dE_red)                # transpose of the all-gather
            ops.add_(eng.dE.view(-1), self.dE_red.view(-1))
        else:
            eng.loss_and_grad_outputs()
        if self.use_comm:
            works = []
            for k, stage in enumerate((eng.backward_stage_joint, self._backward_audio_vision)):
                stage()
                ops.nan_to_num_(self.buckets[k])                            # pretrain_model.py:328, before the pmean
                works.append(self.comm.allreduce_mean_async(self.buckets[k]))   # :329 (bf16, like the reference)
            self._update_buckets(works)
        else:
            self._backward_and_update()
        self.state.finish_step()
        return eng.loss_acc

    def loss_info(self):
        return self.engine.loss_info()
