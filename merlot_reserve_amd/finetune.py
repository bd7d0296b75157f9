"""VCR finetuning step on the MI355X kernels (BASELINE config 5): the reference's
finetune/vcr/qa_qar_joint_finetune.py (F) + finetune/optimization.py (FO), name for name.

    reference                                                     here
    ------------------------------------------------------------  ----------------------------------------------------
    MerlotReserveVCR.from_config / __call__(batch)      F:144-170 MerlotReserveVCR.from_config(config, device=...) /
                                                                  .apply({'params': p}, batch) -> logits [B, 2, A] fp32
    train_loss_fn(state, params, batch)                 F:188-195 train_loss_fn(...) -> (loss, {'is_right', 'loss'})
    construct_finetuning_train_state(opt_config, model, FO:56-105 same name -> (state, tx_fns=None)
      params)
    finetune_train_step(state, batch, loss_fn, tx_fns)  FO:106-191 same name -> (state, loss_info)

One process per GPU; with a merlot_reserve_amd.dist.Comm the bf16 gradients are averaged over ranks (pmean, FO:143)
in two buckets, the first overlapped with the vision tower's backward.  The optimizer is replicated: the 8-way
sharding of the Adam state (FO:38-53, 148-171) is a 16 GB-TPU-core memory measure, not arithmetic.

batch (one device's slice): 'image' [B, h*w, 768] bf16 on the GPU, 'answers' [B, 2, A, T] int32 (numpy),
'labels' [B, 2] int32 (numpy).

Program: ViT (S = h*w + 1) + attention pool -> [B, V, H]; ONE index-driven gather builds the 2*A*B joint sequences
[T text tokens | V vision tokens] (the 8-fold `repeat` of the image rows, F:157, is just the gather list);
joint encoder with one validity code per position; the row at the first MASK of every sequence; Dense(1); softmax
cross-entropy over the A answers; hand-written backward through the same kernels as the pretraining step.
"""
import numpy as np
import torch

from . import ops
from .engine import BF16, F32, I32, TowerEngine, TowerState
from .params import VOCAB, ParamStore, decay_finetune, vcr_param_specs
from .planner import MASK, PADDING, csr_from_pairs, csr_gather, rot_scale_table, rotary_coords_2d


def lr_scale_linearwarmup_lineardecay(step, num_warmup_steps, num_train_steps):
    """pretrain/optimization.py:140-155 in float32"""
    f = np.float32
    step = f(step)
    if step < num_warmup_steps:
        return float(step / f(num_warmup_steps))
    post = (step - f(num_warmup_steps)) / f(num_train_steps - num_warmup_steps + 1.0)
    return float(f(1.0) - min(post, f(1.0)))


class VCRDims:
    def __init__(self, config, B):
        d, m = config['data'], config['model']
        self.B, self.H = B, m['hidden_size']
        self.nh = self.H // 64
        self.gh, self.gw = m['output_grid']
        self.hw = self.gh * self.gw
        self.pr = m['vit_pooling_ratio']
        self.V = self.hw // (self.pr ** 2)
        self.pp3 = m['vit_patch_size'] ** 2 * 3
        self.A = d.get('num_answers', 4)
        self.T = d['lang_seq_len']
        self.n = B * 2 * self.A
        self.Sv, self.Sj = self.hw + 1, self.T + self.V
        self.Lv, self.Lj = m['vit_num_layers'], m['joint_num_layers']


def build_vcr_plan(answers, d):
    """Every integer decision of MerlotReserveVCR.__call__ (F:150-168), from `answers` [B, 2, A, T] alone."""
    ans = np.asarray(answers).astype(np.int64).reshape(d.n, d.T)
    assert ans.min() >= 0 and ans.max() < VOCAB, 'token id out of range'
    b_of = np.arange(d.n) // (2 * d.A)
    codes = np.concatenate([ans, VOCAB + b_of[:, None] * d.V + np.arange(d.V)[None]], 1)            # [n, Sj]
    plan = {}
    plan['joint_gather_indptr'], plan['joint_gather_idx'] = csr_gather(codes)
    valid = np.concatenate([ans != PADDING, np.ones((d.n, d.V), bool)], 1)                          # M:726-728
    plan['joint_code'] = np.where(valid, 0, -1).astype(np.int32).reshape(-1)
    pool_idx = np.argmax((ans == MASK).astype(np.float32), 1)                                       # F:165: first MASK, else 0
    rows = np.arange(d.n) * d.Sj + pool_idx
    plan['pool_indptr'], plan['pool_idx'] = np.arange(d.n + 1, dtype=np.int32), rows.astype(np.int32)
    plan['poolT_indptr'], plan['poolT_idx'] = csr_from_pairs(rows, np.arange(d.n), d.n * d.Sj)
    # transpose of the embedding gather; PAD positions carry exactly zero gradient (masked keys, and never pooled)
    pos_rows = (np.arange(d.n)[:, None] * d.Sj + np.arange(d.T)[None])
    keep = ans != PADDING
    # ... unless the pooled position is itself PAD (a sequence with no MASK whose first token is PAD): that query row
    # attends uniformly to every position (all its keys carry the same -1e10, M:353-356), so the whole sequence gets gradient
    keep |= (ans[np.arange(d.n), pool_idx] == PADDING)[:, None]
    plan['embT_indptr'], plan['embT_idx'] = csr_from_pairs(ans[keep], pos_rows[keep], VOCAB)
    plan['pool_pos'] = pool_idx.astype(np.int32)
    return plan


class VCREngine(TowerEngine):
    def __init__(self, config, B, params, device):
        self.config, self.p, self.dev = config, params, torch.device(device)
        self.d = d = VCRDims(config, B)
        self.dtype, self.fwd_only, self.W, self.G = BF16, False, params.w, params.g
        dev, H = self.dev, d.H
        z = lambda *s: torch.zeros(*s, dtype=BF16, device=dev)
        f = lambda *s: torch.zeros(*s, dtype=F32, device=dev)
        self.tv = TowerState(B * d.Sv, H, d.Lv, B, d.Sv, dev)
        self.tj = TowerState(d.n * d.Sj, H, d.Lj, d.n, d.Sj, dev)
        # static tables: ViT rotary / pool rows; joint rotary (text: segment 0, token index 1..T; vision: pooled-grid (h, w),
        # segment 0 -- identical for every sequence: M:697-720 with token_segment_idx = 0, F:161)
        self.vit_rot = torch.from_numpy(rot_scale_table(np.concatenate([np.zeros((1, 2)), rotary_coords_2d(d.gh, d.gw)], 0))).to(dev)
        pr, h2, w2 = d.pr, d.gh // d.pr, d.gw // d.pr
        n_, i2, j2, di, dj = np.meshgrid(np.arange(B), np.arange(h2), np.arange(w2), np.arange(pr), np.arange(pr), indexing='ij')
        self.vit_pool_rows = torch.from_numpy((n_ * d.Sv + 1 + (i2 * pr + di) * d.gw + j2 * pr + dj).reshape(B * h2 * w2, pr * pr).astype(np.int32)).to(dev)
        coords = np.zeros((d.Sj, 4))
        coords[:d.T, 3] = (1.0 + np.arange(d.T)) / 1024.0
        coords[d.T:, :2] = rotary_coords_2d(h2, w2)
        self.joint_rot = torch.from_numpy(rot_scale_table(coords)).to(dev)
        # static transpose of the image-row tiling: d_imgs_seq[b*V + i] = sum_j Dj[(b*2A + j)*Sj + T + i]
        bb, ii, jj = np.meshgrid(np.arange(B), np.arange(d.V), np.arange(2 * d.A), indexing='ij')
        ip, ix = csr_from_pairs((bb * d.V + ii).reshape(-1), ((bb * 2 * d.A + jj) * d.Sj + d.T + ii).reshape(-1), B * d.V)
        self.visT = (torch.from_numpy(ip).to(dev), torch.from_numpy(ix).to(dev))
        self.unpad_v = self._unpad_csr(B, d.Sv)
        G = B * d.V
        self.v_qin, self.v_q, self.v_po, self.imgs_seq = z(G, H), z(G, H), z(G, H), z(G, H)
        self.v_k, self.v_v = z(self.tv.M, H), z(self.tv.M, H)
        self.v_probs = f(G, d.nh, pr * pr)
        self.v_cls = z(B, H)
        self.pooled_h, self.d_pooled = z(d.n, H), z(d.n, H)
        self.logits = f(d.n, 8)                       # column 0 = the Dense(1) output
        self.dlogits = z(d.n, 8)
        self.loss_acc = f(2)                          # [loss, is_right]
        self.labels_dev = torch.zeros(2 * B, dtype=I32, device=dev)
        self.Dv, self.Dj = z(self.tv.M, H), z(self.tj.M, H)
        self.d_imgs_seq, self.d_v_cls = z(G, H), z(B, H)
        Mmax = max(self.tv.M, self.tj.M)
        self.sc_main = self._make_scratch(Mmax, G, B * d.hw, H, d.nh)
        self.sc_side = self.sc_main                   # single-stream program: the two towers are strictly sequential
        self.cur = self.sc_main
        self.side_stream = None
        self.plan_dev, self._plan_caps, self._plan_views, self.plan_frozen = None, {}, {}, False

    def _idx_capacity(self):
        return self.tj.M + 64

    def forward(self, batch, plan=None):
        d, W = self.d, self.W
        if plan is None:
            plan = build_vcr_plan(batch['answers'], d)
        self.set_plan(plan)
        self.labels_dev.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(batch['labels']).astype(np.int32).reshape(-1))), non_blocking=True)
        return self.forward_device(batch['image'])

    def forward_device(self, image):
        d, W, tv, tj = self.d, self.W, self.tv, self.tj
        self._images2d = image.reshape(d.B * d.hw, d.pp3)
        self.fgemm(self._images2d, 'vision_encoder/embedding/kernel', tv.xin, bias=W['vision_encoder/embedding/bias'], row_map=(d.hw, d.Sv, 1))
        self._tower_with_pool_forward(tv, 'vision_encoder/transformer', 'vision_encoder/seq_attnpool', self.vit_rot, self.vit_pool_rows,
                                      self.v_qin, self.v_q, self.v_k, self.v_v, self.v_po, self.v_probs, self.imgs_seq, self.v_cls)
        ops.segment_sum([W['token_encoder/Embed_0/embedding'], self.imgs_seq], self._pl('joint_gather_indptr'), self._pl('joint_gather_idx'), tj.xin)
        self.encoder_forward(tj, 'joint_transformer', self.joint_rot, self._pl('joint_code'))
        ops.segment_sum([tj.xf], self._pl('pool_indptr'), self._pl('pool_idx'), self.pooled_h)
        self.gemm(self.pooled_h, W['proj/kernel'], self.logits, transB=True)          # Dense(1): [n, H] x [1, H]^T -> column 0
        return self.logits[:, 0].view(d.B, 2, d.A)

    def loss_and_grad_logits(self):
        """train_loss_fn (F:188-195) on the device + dL/dlogits."""
        d = self.d
        self.loss_acc.zero_()
        ops.softmax_xent(self.logits, d.A * 8, 8, self.labels_dev, 2 * d.B, d.A, 1.0 / (2 * d.B), self.loss_acc[0:1], self.loss_acc[1:2], self.dlogits)
        return self.loss_acc

    def backward(self):
        self.backward_stage_joint()
        self.backward_stage_vision()

    def gradient_buckets(self):
        """Ranges of the flat gradient buffer in the order backward makes them final (like trainer.Trainer._make_buckets):
        'joint' = [proj, joint tower, token table]; the vision tower cut at 2/3 and 1/3 of its depth; 'vision_end' = the rest."""
        p, Lv = self.p, self.d.Lv
        v0, v1 = p.tower_ranges['vision_encoder']
        assert v1 == p.total
        cuts = sorted({l for l in (Lv - Lv // 3, Lv - 2 * (Lv // 3)) if 0 < l < Lv}, reverse=True)
        lo_of = lambda l: p.offsets[f'vision_encoder/transformer/layer_{l:02d}/pre_attn_ln/scale'][0]
        buckets, hi = [('joint', 0, v0)], v1
        for l in cuts:
            buckets.append((('vision', l), lo_of(l), hi))
            hi = lo_of(l)
        buckets.append(('vision_end', v0, hi))
        assert sum(b[2] - b[1] for b in buckets) == p.total and all(b[1] % 2048 == 0 and b[2] % 2048 == 0 for b in buckets)
        return buckets, cuts

    def backward_stage_joint(self):
        d, W, G, tj = self.d, self.W, self.p.g, self.tj
        dl = self.dlogits[:, :1]
        self.gemm(dl, self.pooled_h, G['proj/kernel'], transA=True)                   # d proj [1, H] = dlogits^T . pooled_h
        self.gemm(dl, W['proj/kernel'], self.d_pooled)                                # d pooled_h = dlogits . proj^T
        ops.segment_sum([self.d_pooled], self._pl('poolT_indptr'), self._pl('poolT_idx'), self.Dj)   # zero except the pooled rows
        # the tower-level reductions (final / pre LayerNorm) are deferred into ONE mr_reduce_partials launch each side of the layers, as in
        # the pretraining engine (they were two-level immediate reductions here: 19 launches of ~68 us per VCR step, round-3 profile)
        tr = self._begin_tower_reductions()
        Dj = self.encoder_backward(tj, 'joint_transformer', self.joint_rot, self._pl('joint_code'), self.Dj, tr=tr)
        self._flush_tower_reductions(tr)
        ops.segment_sum([Dj], self._pl('embT_indptr'), self._pl('embT_idx'), G['token_encoder/Embed_0/embedding'])
        ops.segment_sum([Dj], self.visT[0], self.visT[1], self.d_imgs_seq)

    def backward_stage_vision(self, layer_done=None):
        d, G, tv = self.d, self.p.g, self.tv
        tr = self._begin_tower_reductions()
        Dv = self._tower_with_pool_backward(tv, 'vision_encoder/transformer', 'vision_encoder/seq_attnpool', self.vit_rot, self.vit_pool_rows,
                                            self.v_qin, self.v_q, self.v_k, self.v_v, self.v_po, self.v_probs, self.d_imgs_seq,
                                            self.d_v_cls, self.Dv, layer_done=layer_done, tr=tr)
        Dp = self.cur.Dpatch[:d.B * d.hw]
        ops.segment_sum([Dv], self.unpad_v[0], self.unpad_v[1], Dp)
        self._t_colsum(tr, Dp, G['vision_encoder/embedding/bias'])
        self._flush_tower_reductions(tr)
        self.gemm(self._images2d, Dp, G['vision_encoder/embedding/kernel'], transA=True)

    def loss_info(self):
        la = self.loss_acc.tolist()
        return {'loss': la[0], 'is_right': la[1]}


# ------------------------------------------------------------------------------------------------ reference-named API
class MerlotReserveVCR:
    """F:144-170"""

    def __init__(self, config, device='cuda:0', rank=0, world=1, comm=None, seed=0, shard_optimizer=False):
        """shard_optimizer = True: the Adam moments are partitioned over the ranks of `comm` (FO:148-171; zero.py)."""
        self.config, self.device, self.rank, self.world, self.comm, self.seed = config, torch.device(device), rank, world, comm, seed
        self.params_store, self.engine, self._last_tree = None, None, None
        self.shard_optimizer, self.shards = shard_optimizer, None
        assert not shard_optimizer or comm is not None, 'shard_optimizer needs a communicator (its world may be 1)'

    @classmethod
    def from_config(cls, config, **kwargs):
        if 'model' not in config or 'data' not in config:
            raise ValueError("config must have 'model' and 'data' sections (the reference's YAML schema)")
        return cls(config, **kwargs)

    def _ensure(self, batch):
        B = int(batch['image'].shape[0])
        if self.engine is None:
            self.params_store = ParamStore(self.config, self.device, seed=self.seed, specs=vcr_param_specs(self.config),
                                           decay_rule=decay_finetune, with_orig=True, with_optimizer=not self.shard_optimizer)
            self.engine = VCREngine(self.config, B, self.params_store, self.device)
            if self.shard_optimizer:
                from .zero import MomentShards
                self.shards = MomentShards(self.params_store, self.engine.gradient_buckets()[0], self.comm)
        elif self.engine.d.B != B:
            raise ValueError(f'this model was initialised for {self.engine.d.B} examples per device, got {B}')
        return self.engine

    def init_from_dummy_batch(self, dummy_batch):
        self._ensure(dummy_batch)
        self._last_tree = self.params_store.master_tree()
        return self._last_tree

    def _load(self, variables):
        if variables is None:
            return
        tree = variables['params'] if 'params' in variables else variables
        if tree is not self._last_tree:
            # a pretrained tree still holds audio_encoder / head / span_encoder (popped at F:181-183) -- they are not read
            self.params_store.load_tree(tree)
            self._last_tree = tree

    def apply(self, variables, batch):
        eng = self._ensure(batch)
        self._load(variables)
        return eng.forward(batch)

    def __call__(self, batch):
        return self.apply(None, batch)


class FinetuneTrainState:
    def __init__(self, model, opt_config):
        self._model, self.opt_config, self.step = model, dict(opt_config), 0
        self.apply_fn = model.apply

    @property
    def params(self):
        return self._model.params_store.master_tree()

    def _scalars(self):
        oc = self.opt_config
        assert oc.get('use_bfloat16_adam', True)
        sched = lr_scale_linearwarmup_lineardecay(self.step, oc['num_warmup_steps'], oc['num_train_steps'])
        b1, b2 = oc.get('beta_1', 0.9), oc.get('beta_2', 0.98)
        bc1 = bc2 = 1.0
        if oc.get('do_bias_correction', True):
            bc1, bc2 = 1.0 - b1 ** (self.step + 1), 1.0 - b2 ** (self.step + 1)
        return sched, -oc['learning_rate'], bc1, bc2

    def apply_gradients(self):
        """FO:77-90 + apply_updates, one fused launch over the flat buffers."""
        oc, p = self.opt_config, self._model.params_store
        if self._model.shards is not None:        # partitioned moments: bucket by bucket, each rank its chunk; a COLLECTIVE
            self.prepare_step()
            for key in self._model.shards.table:
                self._model.shards.update(key, self.apply_shard)
            self.step += 1
            return self
        sched, neg_lr, bc1, bc2 = self._scalars()
        ops.adam_bf16_update_finetune(p.master, p.work, p.grad, p.mu, p.nu, p.orig, p.decay_flags, oc.get('beta_1', 0.9), oc.get('beta_2', 0.98),
                                      oc.get('eps', 1e-6), oc['weight_decay_rate'], sched, neg_lr, bc1, bc2)
        p.update_transposed()
        self.step += 1
        return self

    # ---- the same update per gradient bucket with the step's scalars in a device vector, so that the launches can sit inside a
    # hipGraph and overlap the rest of backward (like trainer.TrainState.prepare_step / apply_range)
    def prepare_step(self):
        dev = self._model.params_store.device
        if getattr(self, 'hyper', None) is None:
            self.hyper = torch.zeros(4, dtype=torch.float32, device=dev)
            self._hyper_host = [torch.zeros(4, dtype=torch.float32, pin_memory=dev.type == 'cuda') for _ in range(2)]
            self._hyper_ev, self._hyper_turn = [None, None], 0
        i = self._hyper_turn
        self._hyper_turn ^= 1
        if self._hyper_ev[i] is not None:
            self._hyper_ev[i].synchronize()
        sched, neg_lr, bc1, bc2 = self._scalars()
        f32 = np.float32         # 1 / bias_correction in fp32, as mr_adam_bf16_update_finetune forms it from its float arguments
        self._hyper_host[i].copy_(torch.tensor([sched, neg_lr, float(f32(1.0) / f32(bc1)), float(f32(1.0) / f32(bc2))], dtype=torch.float32))
        self.hyper.copy_(self._hyper_host[i], non_blocking=True)
        if dev.type == 'cuda':
            self._hyper_ev[i] = torch.cuda.Event()
            self._hyper_ev[i].record()

    def apply_range(self, lo, hi, mu=None, nu=None, transposed=True):
        oc, p = self.opt_config, self._model.params_store
        ops.adam_bf16_update_dev(p.master[lo:hi], p.work[lo:hi], p.grad[lo:hi], p.mu[lo:hi] if mu is None else mu, p.nu[lo:hi] if nu is None else nu,
                                 p.orig[lo:hi], p.decay_flags[lo // 2048:hi // 2048], oc.get('beta_1', 0.9), oc.get('beta_2', 0.98), oc.get('eps', 1e-6),
                                 oc['weight_decay_rate'], self.hyper)
        if transposed:
            p.update_transposed(lo, hi)

    def apply_shard(self, lo, hi, mu, nu):
        self.apply_range(lo, hi, mu=mu, nu=nu, transposed=False)

    # ---- checkpoint form: flax `to_state_dict` of the finetuning chain (finetune/optimization.py:77-90): '0' = the bf16 Adam state
    # {count, mu, nu}, '1' = subtract_old_weights' orig_params (bf16), '2' = add_decayed_weights (empty), '3' = the schedule's count,
    # '4' = scale (empty).  (The reference notes that ITS finetuning state cannot be restarted; this one can.)  With partitioned
    # moments (shard_optimizer=True) state_dict() gathers them from every rank: a COLLECTIVE, see checkpoint.save_checkpoint(rank=).
    def state_dict(self):
        m, p = self._model, self._model.params_store
        mu, nu = (p.mu, p.nu) if m.shards is None else m.shards.full_moments()
        count = torch.tensor(self.step, dtype=torch.int32)
        return {'step': self.step, 'params': p.master_tree(),
                'opt_state': {'0': {'count': count, 'mu': p._to_tree(mu), 'nu': p._to_tree(nu)}, '1': {'orig_params': p._to_tree(p.orig)},
                              '2': {}, '3': {'count': count.clone()}, '4': {}}}

    def load_state_dict(self, sd):
        m, p = self._model, self._model.params_store
        p.load_tree(sd['params'])
        self.step = int(sd.get('step', 0))
        opt = sd.get('opt_state')
        if not opt:
            return self

        def flat(tree):
            host = torch.zeros(p.total, dtype=torch.bfloat16)
            for name, *_ in p.specs:
                o, n = p.offsets[name]
                leaf = tree
                for k in name.split('/'):
                    leaf = leaf[k]
                host[o:o + n] = leaf.reshape(-1).to(torch.bfloat16)
            return host
        mu, nu = flat(opt['0']['mu']), flat(opt['0']['nu'])
        if m.shards is None:
            p.mu.copy_(mu)
            p.nu.copy_(nu)
        else:
            m.shards.store_moments(mu.to(p.device), nu.to(p.device))
        if opt.get('1') and 'orig_params' in opt['1']:
            p.orig.copy_(flat(opt['1']['orig_params']))
        return self


def construct_finetuning_train_state(opt_config, model, params=None, only_state=False):
    """FO:56-105.  Returns (state, tx_fns) like the reference (tx_fns is None: the chain is one fused kernel)."""
    if model.engine is None:
        raise ValueError('call model.init_from_dummy_batch(batch) first')
    if params is not None:
        model._load({'params': params})
    state = FinetuneTrainState(model, opt_config)
    return state if only_state else (state, None)


def train_loss_fn(state, params, batch):
    """F:188-195: forward + softmax cross-entropy; returns (loss, {'is_right', 'loss'}) as host floats."""
    model = state._model
    model.apply(None if params is None else {'params': params}, batch)
    model.engine.loss_and_grad_logits()
    info = model.engine.loss_info()
    return info['loss'], info


def _backward_reduce_update(state):
    """Backward from the engine's dlogits with every finished gradient BUCKET handed to a second stream: nan_to_num ->
    all-reduce(mean) over ranks (FO:148-149) -> the optimizer chain on that range (FO:77-90), while the main stream goes on with
    the vision tower's backward.  Four buckets in ONE program order on ONE stream, the same on every rank; with the library's RCCL
    communicator the whole sequence is capturable (VCRGraphStep)."""
    model = state._model
    eng, comm, p = model.engine, model.comm, model.params_store
    main = torch.cuda.current_stream()
    if getattr(eng, 'comm_stream', None) is None:
        eng.comm_stream = torch.cuda.Stream(device=eng.dev)
    cs = eng.comm_stream
    buckets, cuts = eng.gradient_buckets()
    eng.bucket_log = []

    def finish(key):
        _, lo, hi = next(b for b in buckets if b[0] == key)
        eng.bucket_log.append(key)
        cs.wait_stream(main)
        with torch.cuda.stream(cs):
            if comm is not None:
                ops.nan_to_num_(p.grad[lo:hi])
                comm.allreduce_mean(p.grad[lo:hi])
            if model.shards is not None:
                model.shards.update(key, state.apply_shard)
            else:
                state.apply_range(lo, hi)
    eng.backward_stage_joint()
    finish('joint')
    with ops.gemm_cus(comm.world if comm is not None else 1):      # a bucket's all-reduce runs beside the vision tower's GEMMs
        eng.backward_stage_vision(layer_done=lambda l: finish(('vision', l)) if l in cuts else None)
        finish('vision_end')
    main.wait_stream(cs)


def finetune_train_step(state, batch, loss_fn=None, tx_fns=None, scan_minibatch=False):
    """FO:106-191: bf16 parameter copy -> forward -> loss -> backward -> nan_to_num -> mean over ranks -> optimizer.
    scan_minibatch (FO:125-146): gradients of the examples taken one at a time and SUMMED in bf16 (not averaged: as the reference
    notes, Adam rescales), the metrics averaged."""
    model = state._model
    eng = model._ensure(batch)
    if scan_minibatch:
        return _finetune_train_step_scanned(state, batch)
    eng.forward(batch)
    eng.loss_and_grad_logits()
    state.prepare_step()
    _backward_reduce_update(state)
    state.step += 1
    info = eng.loss_info()
    comm = model.comm
    if comm is not None and comm.world > 1:
        t = torch.tensor([info['loss'], info['is_right']], dtype=torch.float32, device=eng.dev)
        comm.allreduce_mean(t)
        info = {'loss': float(t[0]), 'is_right': float(t[1])}
    return state, info


def _finetune_train_step_scanned(state, batch):
    model = state._model
    eng, comm, p = model.engine, model.comm, model.params_store
    if getattr(model, '_engine1', None) is None:              # an engine for ONE example on the same parameter store
        model._engine1 = VCREngine(model.config, 1, p, model.device)
    e1 = model._engine1
    # Memory: the one-example engine keeps its own activations / scratch beside the B-example one (the price of scan_minibatch here:
    # ~1/B of the big engine's activations; free it with `model._engine1 = None` when scanning is switched off), and ONE persistent
    # parameter-sized bf16 accumulator, zeroed in place each step.
    if getattr(model, '_scan_acc', None) is None or model._scan_acc.shape != p.grad.shape:
        model._scan_acc = torch.zeros_like(p.grad)
    acc = model._scan_acc
    acc.zero_()
    losses = []
    for i in range(eng.d.B):
        micro = {'image': batch['image'][i:i + 1], 'answers': np.asarray(batch['answers'])[i:i + 1], 'labels': np.asarray(batch['labels'])[i:i + 1]}
        p.grad.zero_()                                        # a leaf this example's backward does not write must contribute 0, not the previous example's value
        e1.forward(micro)
        e1.loss_and_grad_logits()
        e1.backward()
        ops.add_(acc, p.grad)                                 # bf16 + bf16 -> bf16, like the scan's `a + b` on bf16 leaves
        losses.append(e1.loss_info())
    p.grad.copy_(acc)
    if comm is not None:
        ops.nan_to_num_(p.grad)
        comm.allreduce_mean(p.grad)
    state.prepare_step()
    if model.shards is not None:
        for key in model.shards.table:
            model.shards.update(key, state.apply_shard)
    else:
        state.apply_range(0, p.total)
    state.step += 1
    info = {k: float(np.mean([l[k] for l in losses])) for k in losses[0]}
    if comm is not None and comm.world > 1:
        t = torch.tensor([info['loss'], info['is_right']], dtype=torch.float32, device=eng.dev)
        comm.allreduce_mean(t)
        info = {'loss': float(t[0]), 'is_right': float(t[1])}
    return state, info


class VCRGraphStep:
    """hipGraph replay of finetune_train_step: the step is a fixed launch sequence over fixed buffers, so forward + loss + backward +
    the four bucket reductions + the optimizer are captured ONCE and replayed -- with the library's RCCL communicator
    (dist.NativeComm) the collectives are stream-ordered launches inside the same graph (a torch.distributed comm is host-scheduled
    and cannot be captured); per batch only the image, the labels, the plan's index lists and the step's four scalars are copied
    into their persistent buffers.  Same results as finetune_train_step (tests/test_vcr_gpu.py)."""

    def __init__(self, state, batch):
        self.state, self.model = state, state._model
        eng = self.eng = self.model._ensure(batch)
        comm = self.model.comm
        assert comm is None or getattr(comm, 'capturable', False), 'only the library RCCL communicator (dist.NativeComm) can be captured'
        finetune_train_step(state, batch)                  # eager step: every buffer and plan capacity exists afterwards
        self.image_in = torch.zeros_like(batch['image'])
        eng.plan_frozen = True
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
            eng.forward_device(self.image_in)
            eng.loss_and_grad_logits()
            _backward_reduce_update(state)

    def __call__(self, batch, plan=None):
        eng = self.eng
        self.state.prepare_step()
        self.image_in.copy_(batch['image'], non_blocking=True)
        eng.set_plan(plan if plan is not None else build_vcr_plan(batch['answers'], eng.d))
        eng.labels_dev.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(batch['labels']).astype(np.int32).reshape(-1))), non_blocking=True)
        self.graph.replay()
        self.state.step += 1
        return self.state


def make_vcr_batch(config, B, seed=0, device='cpu'):
    """Synthetic VCR batch with the structure of finetune/data: answers = [question tokens, MASK, answer tokens, PAD...]."""
    d = VCRDims(config, B)
    rng = np.random.default_rng(seed)
    ans = np.zeros((B, 2, d.A, d.T), dtype=np.int32)
    for idx in np.ndindex(B, 2, d.A):
        nq, na = int(rng.integers(4, d.T // 3)), int(rng.integers(2, d.T // 3))
        ans[idx][:nq] = rng.integers(10, VOCAB, size=nq)
        ans[idx][nq] = MASK
        ans[idx][nq + 1:nq + 1 + na] = rng.integers(10, VOCAB, size=na)
    g = torch.Generator().manual_seed(seed)
    return {'image': torch.rand(B, d.hw, d.pp3, generator=g).to(torch.bfloat16).to(device), 'answers': ans,
            'labels': rng.integers(0, d.A, size=(B, 2)).astype(np.int32)}
