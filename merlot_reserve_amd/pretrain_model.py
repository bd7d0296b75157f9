"""The reference's step-level interface, name for name, on top of the MI355X engine.

    reference (JAX/Flax)                                             here (same call shapes, torch tensors on the GPU)
    ---------------------------------------------------------------  --------------------------------------------------
    MerlotReservePretrainer.from_config(config)      P:38, M:584     MerlotReservePretrainer.from_config(config, ...)
    model.init_from_dummy_batch(dummy_batch)         M:636-649       -> nested dict of fp32 tensors, Flax names
    model.apply({'params': p}, batch) -> dict        P:38-259        -> {'imgs_to_audio': {'x','y'}, 'text_to_audio': {...}, ...}
    loss_fn_given_preds(preds) -> (loss, loss_info)  P:262-303       same keys in loss_info
    construct_train_state(opt_config, model, params) O:158-195       -> TrainState(step, params, apply_fn, apply_gradients)
    train_step(state, batch) -> (state, loss_info)   P:306-340       one full step (collectives included when world > 1)

(P = pretrain/pretrain_model.py, M = mreserve/modeling.py, O = pretrain/optimization.py of the reference.)

Differences that cannot be hidden: a batch is ONE device's slice ([B, ...], not [num_devices, B, ...]: one process per
GPU replaces pmap); arrays are torch tensors (bf16 / int32) resident on the GPU; the two JAX random draws of the
forward are explicit optional inputs (`split_from_here`, `gumbel_z`: see planner.py) with a seeded generator behind
them.  Everything numerical runs in the HIP library: there is no CPU path here, and a missing library raises.
"""
import torch

from . import ops, trainer as _trainer
from .trainer import Trainer


class _Preds(dict):
    """The outputs dict of the forward; remembers which engine produced it so that loss_fn_given_preds(preds) can run
    the loss kernels on the buffers the tensors are views of."""
    engine = None
    trainer = None


class MerlotReservePretrainer:
    def __init__(self, config, device='cuda:0', rank=0, world=1, comm=None, seed=0):
        self.config, self.device, self.rank, self.world, self.comm, self.seed = config, torch.device(device), rank, world, comm, seed
        # config['model']['use_bfloat16'] (M:594; train.py:61-67 clears it off-TPU): False = the whole step in fp32 -- the fp32 program of
        # the engine, fp32 gradients (train_step(use_bfloat16_grads=False), which train.py:106 ties to this flag)
        self.use_bfloat16 = bool(config['model'].get('use_bfloat16', True))
        self.trainer = None
        self._last_tree = None

    @classmethod
    def from_config(cls, config, **kwargs):
        """P:38 / M:584-598.  kwargs: device, rank, world, comm (merlot_reserve_amd.dist.Comm), seed."""
        if 'model' not in config or 'data' not in config:
            raise ValueError("config must have 'model' and 'data' sections (the reference's YAML schema)")
        from .config import Dims
        Dims(config, 1)             # refuses what this build does not implement (size_per_head != 64) before any buffer exists
        return cls(config, **kwargs)

    # -- parameters -------------------------------------------------------------------------------------------------
    def _ensure(self, batch):
        B = int(batch['images'].shape[0])
        if self.trainer is None:
            self.trainer = Trainer(self.config, B, self.device, rank=self.rank, world=self.world, seed=self.seed, comm=self.comm,
                                   bf16_grads=self.use_bfloat16)
        elif self.trainer.B != B:
            raise ValueError(f'this model was initialised for {self.trainer.B} records per device, got a batch of {B} '
                             '(buffers and hipGraphs are shape-specialised; build another model for another batch size)')
        return self.trainer

    def init_from_dummy_batch(self, dummy_batch):
        """M:636-649: builds the parameters (the reference's initialisers, fp32) for the batch's shapes and returns the
        Flax-named tree."""
        self._last_tree = self._ensure(dummy_batch).params.master_tree()
        return self._last_tree

    def _load(self, variables):
        if variables is None:
            return
        tree = variables['params'] if 'params' in variables else variables
        if tree is not self._last_tree:        # a tree this model has not seen (trees are host copies): upload it
            self.trainer.params.load_tree(tree)
            self._last_tree = tree

    # -- forward ----------------------------------------------------------------------------------------------------
    def apply(self, variables, batch, split_from_here=None, gumbel_z=None):
        """P:38-259 with variables = {'params': tree}: returns the dict of normalised, temperature-scaled embeddings
        per objective.  The tensors are views of the engine's packed buffer, valid until the next forward."""
        tr = self._ensure(batch)
        self._load(variables)
        draws = None
        if split_from_here is not None or gumbel_z is not None:
            if split_from_here is None or gumbel_z is None:
                raise ValueError('pass both split_from_here and gumbel_z, or neither')
            draws = (split_from_here, gumbel_z)
        plan = tr.plan(batch, draws)
        tr.engine.forward(batch, plan=plan)
        preds = _Preds(tr.engine.outputs())
        preds.engine, preds.trainer = tr.engine, tr
        return preds

    def __call__(self, batch):
        return self.apply(None, batch)


def loss_fn_given_preds(preds):
    """P:262-303.  Returns (loss, loss_info) with the reference's keys: the three objectives and the
    `_stuff_to_span_from_*` diagnostics; loss = sum of the keys that do not start with '_'."""
    if not isinstance(preds, _Preds) or preds.engine is None:
        raise TypeError('loss_fn_given_preds expects the dict returned by MerlotReservePretrainer.apply')
    tr = preds.trainer
    text_preds = preds.pop('text_preds', None)
    if tr.use_comm:
        tr.comm.gather_embeddings(preds.engine.E, tr.E_all)
        preds.engine.loss_and_grad_outputs(tr.E_all, tr.dE_all)
    else:
        preds.engine.loss_and_grad_outputs()
    info = preds.engine.loss_info()
    loss = info.pop('loss')
    if text_preds is not None:
        # P:265-274, the mask-LM special case: {'logits' [n, V], 'labels' [n]}, rows with label 0 masked out.  No forward of the reference (nor this
        # one) emits the entry; a caller that adds it gets the reference's value under the reference's key, and d loss / d logits in
        # text_preds['dlogits'] (fp32) -- the entry is outside the model's graph, so nothing flows back into the towers.
        logits = text_preds['logits'].to(torch.float32).contiguous()
        labels = torch.as_tensor(text_preds['labels']).to(device=logits.device, dtype=torch.int32).contiguous()
        out2 = torch.zeros(2, dtype=torch.float32, device=logits.device)
        text_preds['dlogits'] = torch.empty_like(logits)
        ops.masked_lm_xent(logits, labels, out2, dlogits=text_preds['dlogits'])
        info['audio2text'] = float(out2[0])
        loss = loss + info['audio2text']
    return loss, info


class TrainState:
    """flax.training.train_state.TrainState for this path: `.step`, `.params` (the Flax-named fp32 tree, views of the
    master buffer), `.apply_fn`, `.apply_gradients()`.  The optimizer state (bf16 mu, cube-coded bf16 nu) lives beside
    the parameters in the model's flat buffers (params.py), so a state is bound to its model."""

    def __init__(self, model, inner):
        self._model, self._inner = model, inner
        self.apply_fn = model.apply

    @property
    def step(self):
        return self._inner.step

    @property
    def params(self):
        return self._model.trainer.params.master_tree()

    @property
    def opt_config(self):
        return self._inner.opt_config

    def apply_gradients(self, grads=None):
        """O:180-195 on the gradients of the last backward (`grads` is accepted for signature parity and must be None or
        the model's own gradient tree: gradients never leave the flat bf16 buffer)."""
        self._inner.apply_gradients()
        return self


def construct_train_state(opt_config, model, params=None):
    """O:158-195: bf16-state Adam + weight decay (leaves with ndim > 1) + warmup/cosine schedule + learning rate."""
    if model.trainer is None:
        raise ValueError('call model.init_from_dummy_batch(batch) first (the reference does the same: train.py:99-100)')
    if params is not None:
        model._load({'params': params})
    model.trainer.state = _trainer.construct_train_state(opt_config, model.trainer.params)
    model.trainer.state.f32_grads = model.trainer.f32
    return TrainState(model, model.trainer.state)


def train_step(state, batch, use_bfloat16_grads=True, split_from_here=None, gumbel_z=None):
    """P:306-340: bf16 parameter copy -> forward -> loss -> backward -> nan_to_num -> mean over ranks -> optimizer.
    Returns (state, loss_info) with loss_info as host floats, averaged over ranks like the reference's pmean (:335).
    use_bfloat16_grads follows the model's precision, as in the reference's launcher (train.py:106 passes
    config['model']['use_bfloat16']): True with a bf16 model (the fast path: bf16 working copy, bf16 gradients), False with a model
    built from use_bfloat16 = False (fp32 program, fp32 gradients into the same Adam chain).  The two mixed combinations
    (fp32 accumulation of a bf16 model's gradients; an fp32 model on bf16-rounded parameters) are not built and raise."""
    model = state._model
    if bool(use_bfloat16_grads) != model.use_bfloat16:
        raise NotImplementedError(f"train_step(use_bfloat16_grads={bool(use_bfloat16_grads)}) on a model with config['model']['use_bfloat16'] = "
                                  f'{model.use_bfloat16}: only the combinations the reference launches (train.py:106: equal flags) are implemented')
    tr = model._ensure(batch)
    draws = None if split_from_here is None else (split_from_here, gumbel_z)
    tr.train_step(batch, draws=draws)
    info = tr.loss_info(reduce=True)           # mean over ranks (P:336)
    info.pop('loss', None)
    return state, info
