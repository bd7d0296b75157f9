"""Parameter store: the reference's Flax parameter tree (names and shapes of mreserve/modeling.py:600-634, see
SURVEY.md row P) laid out in FLAT device buffers sized for one fused optimizer launch and one bucketed all-reduce:

    master fp32 | working copy bf16 (what the forward reads: pretrain_model.py:323-324) | grads bf16 | mu bf16 | nu bf16

Every leaf starts at a multiple of 2048 elements (the Adam kernel's block = its weight-decay granularity) and leaves of
one tower are contiguous, in the order backward finishes them, so a tower's gradients form one all-reduce bucket.
Kernels see each leaf as a 2-D row-major view ([in, out] for every Dense kernel).
"""
import math
from collections import OrderedDict

import numpy as np
import torch

ALIGN = 2048
VOCAB = 32768


def _encoder_specs(prefix, H, L, has_cls):
    nh = H // 64
    s = []
    if has_cls:
        s.append((f'{prefix}/cls', (H,), (H,), 'normal02', None))
    s += [(f'{prefix}/pre_ln/scale', (H,), (H,), 'ones', None), (f'{prefix}/pre_ln/bias', (H,), (H,), 'zeros', None)]
    for i in range(L):
        p = f'{prefix}/layer_{i:02d}'
        s += [
            (f'{p}/pre_attn_ln/scale', (H,), (H,), 'ones', None), (f'{p}/pre_attn_ln/bias', (H,), (H,), 'zeros', None),
            (f'{p}/attention_layer/qkv/kernel', (H, 3 * nh, 64), (H, 3 * H), 'kernel', H),
            (f'{p}/attention_layer/qkv/bias', (3 * nh, 64), (3 * H,), 'zeros', None),
            (f'{p}/attention_layer/attn_proj/kernel', (nh, 64, H), (H, H), 'kernel', H),
            (f'{p}/pre_mlp_ln/scale', (H,), (H,), 'ones', None), (f'{p}/pre_mlp_ln/bias', (H,), (H,), 'zeros', None),
            (f'{p}/mlp_layer/intermediate/kernel', (H, 4 * H), (H, 4 * H), 'kernel', H),
            (f'{p}/mlp_layer/intermediate/bias', (4 * H,), (4 * H,), 'zeros', None),
            (f'{p}/mlp_layer/out/kernel', (4 * H, H), (4 * H, H), 'kernel', 4 * H),
        ]
    s += [(f'{prefix}/final_ln/scale', (H,), (H,), 'ones', None), (f'{prefix}/final_ln/bias', (H,), (H,), 'zeros', None)]
    if has_cls:
        s += [(f'{prefix}/cls_proj/kernel', (H, H), (H, H), 'kernel', H), (f'{prefix}/cls_proj/bias', (H,), (H,), 'zeros', None)]
    return s


def _attnpool_specs(prefix, H):
    nh = H // 64
    s = []
    for k in ('query', 'key', 'value'):
        s += [(f'{prefix}/{k}/kernel', (H, nh, 64), (H, H), 'lecun', H), (f'{prefix}/{k}/bias', (nh, 64), (H,), 'zeros', None)]
    s += [(f'{prefix}/out/kernel', (nh, 64, H), (H, H), 'lecun', H), (f'{prefix}/out/bias', (H,), (H,), 'zeros', None)]
    return s


def param_specs(config):
    """[(name, flax_shape, view_shape, init, fan_in)] in flat-buffer order.  Towers in the order their gradients
    complete in backward (loss side first): scales, head, span, joint, token, audio, vision."""
    m = config['model']
    H = m['hidden_size']
    specs = [('contrastive_scales', (3,), (3,), 'ones', None)]
    specs += [('head/kernel', (H, H), (H, H), 'kernel', H), ('head/bias', (H,), (H,), 'zeros', None)]
    specs += _encoder_specs('span_encoder/transformer', H, m['span_num_layers'], True)
    if not m.get('do_rotary', True):        # pretrain_model.py:146-148 drops the joint coordinates: the joint tower learns `pe` [seq_len, H] (modeling.py:335-341)
        specs += [('joint_transformer/pe', (config['data']['seq_len'], H), (config['data']['seq_len'], H), 'normal02', None)]
    specs += _encoder_specs('joint_transformer', H, m['joint_num_layers'], False)
    specs += [('token_encoder/Embed_0/embedding', (VOCAB, H), (VOCAB, H), 'embed', None)]
    specs += [('audio_encoder/embedding/kernel', (m['audio_patch_size'], 65, H), (m['audio_patch_size'] * 65, H), 'kernel', 130),
              ('audio_encoder/embedding/bias', (H,), (H,), 'zeros', None)]
    specs += _encoder_specs('audio_encoder/transformer', H, m['audio_num_layers'], True)
    specs += _attnpool_specs('audio_encoder/seq_attnpool', H)
    pp3 = m['vit_patch_size'] ** 2 * 3
    specs += [('vision_encoder/embedding/kernel', (pp3, H), (pp3, H), 'kernel', pp3),
              ('vision_encoder/embedding/bias', (H,), (H,), 'zeros', None)]
    specs += _encoder_specs('vision_encoder/transformer', H, m['vit_num_layers'], True)
    specs += _attnpool_specs('vision_encoder/seq_attnpool', H)
    return specs


def tower_of(name):
    return name.split('/')[0]


def _trunc_normal(shape, std, gen):
    t = torch.empty(shape, dtype=torch.float32)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=gen)
    return t * std


def init_leaf(kind, flax_shape, fan_in, H, gen):
    """Initialisers of mreserve/modeling.py:147-186 (kernel_init: truncated normal, std = min(18/in, 0.02)/sqrt(2)),
    :316 (cls), :526 (embedding), :634 (scales); flax defaults (lecun_normal, zero bias) for seq_attnpool."""
    if kind == 'zeros':
        return torch.zeros(flax_shape)
    if kind == 'ones':
        return torch.ones(flax_shape)
    if kind == 'normal02':
        return torch.randn(flax_shape, generator=gen) * 0.02
    if kind == 'normal001':                      # qa_qar_joint_finetune.py:184
        return torch.randn(flax_shape, generator=gen) * 0.01
    if kind == 'kernel':
        return _trunc_normal(flax_shape, min(18.0 / fan_in, 0.02) / math.sqrt(2.0), gen)
    if kind == 'lecun':
        return _trunc_normal(flax_shape, math.sqrt(1.0 / fan_in) / 0.87962566103423978, gen)
    if kind == 'embed':
        if H <= 768:
            return torch.randn(flax_shape, generator=gen) * 0.02
        a = math.sqrt(6.0 / (flax_shape[0] + flax_shape[1]))
        return (torch.rand(flax_shape, generator=gen) * 2 - 1) * a
    raise ValueError(kind)


def vcr_param_specs(config):
    """Parameter tree of MerlotReserveVCR (finetune/vcr/qa_qar_joint_finetune.py:144-185): the pretraining tree without
    audio_encoder / head / span_encoder (and contrastive_scales, which the finetuning graph never reads), plus
    proj/kernel [H, 1] (Dense(1), no bias; viewed [1, H] = its memory layout as an [N, K] operand).
    Order = the order backward finishes them."""
    m = config['model']
    H = m['hidden_size']
    pp3 = m['vit_patch_size'] ** 2 * 3
    specs = [('proj/kernel', (H, 1), (1, H), 'normal001', None)]
    specs += _encoder_specs('joint_transformer', H, m['joint_num_layers'], False)
    specs += [('token_encoder/Embed_0/embedding', (VOCAB, H), (VOCAB, H), 'embed', None)]
    specs += [('vision_encoder/embedding/kernel', (pp3, H), (pp3, H), 'kernel', pp3),
              ('vision_encoder/embedding/bias', (H,), (H,), 'zeros', None)]
    specs += _encoder_specs('vision_encoder/transformer', H, m['vit_num_layers'], True)
    specs += _attnpool_specs('vision_encoder/seq_attnpool', H)
    return specs


def decay_pretrain(flax_shape):
    """pretrain/optimization.py:182-184"""
    return len(flax_shape) > 1


def decay_finetune(flax_shape):
    """finetune/optimization.py:74-75"""
    return len(flax_shape) > 1 and int(np.prod(flax_shape)) > 4096


class ParamStore:
    def __init__(self, config, device, seed=0, with_optimizer=True, init=True, specs=None, decay_rule=decay_pretrain,
                 with_orig=False):
        self.config = config
        self.device = torch.device(device)
        self.H = config['model']['hidden_size']
        self.specs = param_specs(config) if specs is None else specs
        self.offsets = OrderedDict()
        off = 0
        flags = []
        self.tower_ranges = OrderedDict()
        for name, fshape, vshape, kind, fan in self.specs:
            n = int(np.prod(fshape))
            npad = (n + ALIGN - 1) // ALIGN * ALIGN
            self.offsets[name] = (off, n)
            flags += [1 if decay_rule(fshape) else 0] * (npad // ALIGN)     # the optax weight-decay mask, per leaf
            t = tower_of(name)
            lo, _ = self.tower_ranges.get(t, (off, off))
            self.tower_ranges[t] = (lo, off + npad)
            off += npad
        self.total = off
        self.master = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.work = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        self.grad = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        if with_optimizer:
            self.mu = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
            self.nu = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        # finetuning keeps a bf16 copy of the initial parameters (subtract_old_weights, finetune/optimization.py:20-34)
        self.orig = torch.zeros(off, dtype=torch.bfloat16, device=self.device) if with_orig else None
        self.decay_flags = torch.tensor(flags, dtype=torch.uint8, device=self.device)
        self.w, self.g, self.wm = {}, {}, {}       # bf16 working copy / bf16 grads / fp32 master, as 2-D kernel views
        for name, fshape, vshape, kind, fan in self.specs:
            o, n = self.offsets[name]
            self.w[name] = self.work[o:o + n].view(*vshape)
            self.wm[name] = self.master[o:o + n].view(*vshape)
            self.g[name] = self.grad[o:o + n].view(*vshape)
        # Transposed bf16 working copies of the Dense kernels ([in, out] -> [out, in]): what the FORWARD GEMMs read, so that
        # both of a GEMM's operands are contraction-contiguous (the dgrads get that from the flax layout as it is).  Same
        # offsets as `work`; rewritten after every change of `work` (update_transposed: one HBM-bound pass, 4 B / parameter).
        self.wT, tr = {}, []
        self.workT = None
        if self.device.type == 'cuda':
            ntile = 0
            for name, fshape, vshape, kind, fan in self.specs:
                if name.endswith('/kernel') and len(vshape) == 2 and vshape[0] % 64 == 0 and vshape[1] % 64 == 0:
                    o, n = self.offsets[name]
                    tr.append((o, vshape[0], vshape[1], ntile, name))
                    ntile += (vshape[0] // 64) * (vshape[1] // 64)
            if tr:
                self.workT = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
                self._tr = tr
                self._tr_ntile = ntile
                self._tr_dev = torch.tensor([t[:4] for t in tr], dtype=torch.int32, device=self.device)
                for o, rows, cols, _, name in tr:
                    self.wT[name] = self.workT[o:o + rows * cols].view(cols, rows)
        if init:
            self.load_tree(self.random_tree(seed))

    # ---- trees (nested dicts with the Flax names, flax shapes, fp32 CPU tensors) ----
    def random_tree(self, seed=0):
        gen = torch.Generator().manual_seed(seed)
        tree = {}
        for name, fshape, vshape, kind, fan in self.specs:
            _set(tree, name, init_leaf(kind, fshape, fan, self.H, gen))
        return tree

    def load_tree(self, tree):
        """master <- tree (fp32), working copy <- bf16(master)  (pretrain_model.py:323-324)."""
        host = torch.zeros(self.total, dtype=torch.float32)
        for name, fshape, *_ in self.specs:
            o, n = self.offsets[name]
            leaf = _get(tree, name)
            assert tuple(leaf.shape) == tuple(fshape), f'{name}: {tuple(leaf.shape)} != {fshape}'
            host[o:o + n] = leaf.reshape(-1).to(torch.float32)
        self.master.copy_(host)
        if self.orig is not None:
            self.orig.copy_(host.to(torch.bfloat16))
        self.refresh_work()

    def refresh_work(self):
        if self.master.is_cuda:
            from . import ops
            ops.cast_params(self.master, self.work)
            self.update_transposed()
        else:
            self.work.copy_(self.master.to(torch.bfloat16))

    def update_transposed(self, lo=0, hi=None):
        """Rewrite the transposed copies of the Dense kernels lying in [lo, hi) of the flat buffers (call after `work` changed)."""
        if self.workT is None:
            return
        from . import ops
        hi = self.total if hi is None else hi
        sel = [t for t in self._tr if lo <= t[0] < hi]
        if not sel:
            return
        last = sel[-1]
        ops.transpose_leaves(self.work, self.workT, self._tr_dev, len(self._tr), sel[0][3], last[3] + (last[1] // 64) * (last[2] // 64))

    def enable_f32_grads(self):
        """use_bfloat16_grads = False (pretrain/pretrain_model.py:323-333): the step differentiates the fp32 master parameters and the
        gradients stay fp32 -- a second flat gradient buffer (4 B / parameter) with the same offsets, allocated on first use."""
        if getattr(self, 'grad32', None) is None:
            self.grad32 = torch.zeros(self.total, dtype=torch.float32, device=self.device)
            self.g32 = {}
            for name, fshape, vshape, kind, fan in self.specs:
                o, n = self.offsets[name]
                self.g32[name] = self.grad32[o:o + n].view(*vshape)
        return self.grad32

    def _to_tree(self, flat):
        host = flat.detach().to('cpu')
        tree = {}
        for name, fshape, *_ in self.specs:
            o, n = self.offsets[name]
            _set(tree, name, host[o:o + n].reshape(fshape).clone())
        return tree

    def master_tree(self):
        return self._to_tree(self.master)

    def work_tree(self):
        return self._to_tree(self.work)

    def grad_tree(self):
        return self._to_tree(self.grad)

    def grad32_tree(self):
        return self._to_tree(self.grad32)


def _set(tree, path, val):
    ks = path.split('/')
    for k in ks[:-1]:
        tree = tree.setdefault(k, {})
    tree[ks[-1]] = val


def _get(tree, path):
    for k in path.split('/'):
        tree = tree[k]
    return tree
