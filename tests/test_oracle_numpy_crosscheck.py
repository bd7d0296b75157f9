"""Standing of the oracle (CPU): the reference ships no tests or golden vectors and JAX cannot be imported, so parity is
unpinned by the reference.  What can be done without it, and is done here:

* oracle/ref_numpy.py -- a second restatement of the same reference lines by a different route (float64 NumPy, explicit
  per-sequence / per-head / per-row loops, no torch) -- must agree with oracle/ref_torch.py (float64) to 1e-10 on every
  output tensor of the pretraining forward, on every integer decision, and on the loss for one and for two virtual devices;
* ref_torch's autograd gradients (what every GPU backward test is compared with) must agree with central finite differences
  of its own float64 loss along random directions of >= 20 parameter leaves.
"""
import numpy as np
import pytest
import torch

from merlot_reserve_amd.config import tiny_config
from merlot_reserve_amd.params import ParamStore
from merlot_reserve_amd.synthetic import make_batch, make_draws
from oracle import ref_numpy as N
from oracle import ref_torch as R
from tests.util import oracle_batch, oracle_draws, tree_to


# the config branches built in round 5: more than one sequence per kind (P:99-137) with the learned position table instead of rotary (P:146-148)
BRANCHES = {'data': dict(num_audio2text_seqs=2, num_text2audio_seqs=2, num_text_seqs=2), 'model': dict(do_rotary=False)}


def _setup(seed, B=1, flags=None):
    cfg = tiny_config(hidden_size=128, seq_len=80, lang_seq_len=40)
    for k, v in (flags or {}).items():
        cfg[k].update(v)
    store = ParamStore(cfg, 'cpu', seed=seed, with_optimizer=False)
    g = torch.Generator().manual_seed(seed + 100)
    tree = store.master_tree()

    def jitter(t):          # non-trivial biases / LayerNorm parameters / temperatures, larger embeddings
        if isinstance(t, dict):
            return {k: jitter(v) for k, v in t.items()}
        return t + 0.1 * torch.randn(t.shape, generator=g) if t.dim() == 1 else t * 3.0
    tree = tree_to(jitter(tree), torch.float64)
    batch = make_batch(cfg, B, seed=seed, device='cpu', float_dtype=torch.float32)
    splits, z = make_draws(cfg, B, seed=seed)
    splits[0][0] = 1                                        # a video-source split: the masking branch
    return cfg, tree, batch, splits, z


def _np_tree(t):
    return {k: _np_tree(v) for k, v in t.items()} if isinstance(t, dict) else t.numpy().astype(np.float64)


def _np_batch(batch):
    return {k: (v if isinstance(v, np.ndarray) else v.numpy().astype(np.float64)) for k, v in batch.items()}


@pytest.mark.parametrize('flags', [None, BRANCHES], ids=['stock', 'multi_seq_learned_pe'])
def test_numpy_and_torch_restatements_agree(flags):
    per_dev_t, per_dev_n = [], []
    for dev_i, seed in enumerate((5, 6)):
        cfg, tree, batch, splits, z = _setup(seed if dev_i == 0 else 5, B=1, flags=flags)
        if dev_i == 1:                                      # second virtual device: same parameters, another batch
            batch = make_batch(cfg, 1, seed=seed, device='cpu', float_dtype=torch.float32)
            splits, z = make_draws(cfg, 1, seed=seed)
        osp, oz = oracle_draws(splits, z)
        with torch.no_grad():
            pt = R.pretrain_forward(tree, cfg, oracle_batch(batch, torch.float64), osp, oz)
        pn = N.pretrain_forward(_np_tree(tree), cfg, _np_batch(batch), splits, z)
        for k in pt:
            for k2 in ('x', 'y', 'y_extra'):
                if k2 in pt[k]:
                    a, b = pt[k][k2].numpy(), pn[k][k2]
                    assert a.dtype == np.float64 and a.shape == b.shape
                    err = np.abs(a - b).max() / np.abs(a).max()
                    assert err < 1e-10, (k, k2, err)
        assert np.array_equal(pt['stuff_to_span']['_sources'].numpy(), pn['stuff_to_span']['_sources'])
        per_dev_t.append(pt)
        per_dev_n.append(pn)
    for devs_t, devs_n, rank in (([per_dev_t[0]], [per_dev_n[0]], 0), (per_dev_t, per_dev_n, 0), (per_dev_t, per_dev_n, 1)):
        lt, it = R.loss_fn_given_preds(devs_t, rank=rank)
        ln, inn = N.loss_fn_given_preds(devs_n, rank=rank)
        assert abs(float(lt) - ln) < 1e-10 * abs(ln), (float(lt), ln)
        assert set(it) == set(inn)
        for k in it:
            assert abs(float(it[k]) - inn[k]) < 1e-9 * max(abs(inn[k]), 1.0), (k, float(it[k]), inn[k])
    # the mask-LM special case of the loss (P:265-274)
    g = torch.Generator().manual_seed(1)
    logits, labels = torch.randn(9, 50, generator=g, dtype=torch.float64) * 3, torch.tensor([0, 3, 49, 0, 7, 7, 1, 0, 12])
    lt, it = R.loss_fn_given_preds([dict(per_dev_t[0], text_preds={'logits': logits, 'labels': labels})])
    ln, inn = N.loss_fn_given_preds([dict(per_dev_n[0], text_preds={'logits': logits.numpy(), 'labels': labels.numpy()})])
    assert abs(float(it['audio2text']) - inn['audio2text']) < 1e-12 * abs(inn['audio2text']) and abs(float(lt) - ln) < 1e-10 * abs(ln)


def test_known_answers_of_the_numpy_restatement():
    """The hand-derived vectors of SURVEY.md 8c, on the second restatement too."""
    assert N.rotary_coordinates(4).tolist() == [-2, -1, 1, 2] and N.rotary_coordinates(5).tolist() == [-2, -1, 1, 2, 3]
    assert N.rotary_coordinates(3, center_origin=False).tolist() == [1, 2, 3]
    c = N.rotary_coordinates_2d(12, 20)
    assert np.allclose(c[0], [-6 / 21, -10 / 21]) and np.allclose(c[1], [-6 / 21, -9 / 21]) and np.allclose(c[20], [-5 / 21, -10 / 21])
    x = np.arange(1.0, 65.0).reshape(1, 64)
    first, second = N.rotary_sinusoids(np.zeros((1, 1)))               # theta = 0 -> [-x0, x1, -x2, x3, ...]
    out = N.apply_rotary_head(x, first, second)
    exp = x.copy(); exp[:, 0:32:2] *= -1
    assert np.array_equal(out, exp)
    ln = N.layer_norm(np.array([[1.0, 2.0, 3.0, 4.0]]), {'scale': np.ones(4), 'bias': np.zeros(4)})
    assert np.allclose(ln, [[-1.341635, -0.447212, 0.447212, 1.341635]], atol=1e-5)
    assert np.allclose(N.unit_normalize(np.zeros((1, 8))), 0.0)
    px, cnt = N.one_hot_pool(np.array([True, False, True]), np.array([1, 1, -1]), np.ones((3, 2)), 2)
    assert px.tolist() == [[0, 0], [1, 1]] and cnt.tolist() == [0, 1]


def test_autograd_of_the_oracle_matches_finite_differences():
    cfg, tree, batch, splits, z = _setup(9, B=1)
    ob = oracle_batch(batch, torch.float64)
    osp, oz = oracle_draws(splits, z)
    params = R.tree_map(lambda t: t.clone().requires_grad_(True), tree)

    def loss_of(p):
        return R.loss_fn_given_preds([R.pretrain_forward(p, cfg, ob, osp, oz)])[0]
    loss_of(params).backward()
    leaves = list(R.tree_leaves(params))
    rng = np.random.default_rng(0)
    # every kind of leaf: embeddings, cls tokens, temperatures, LN, qkv / proj / MLP kernels and biases of all four towers, pools
    must = ['contrastive_scales', 'head/kernel', 'token_encoder/Embed_0/embedding', 'vision_encoder/embedding/kernel',
            'audio_encoder/embedding/kernel', 'vision_encoder/transformer/cls', 'span_encoder/transformer/cls_proj/kernel',
            'vision_encoder/seq_attnpool/query/kernel', 'audio_encoder/seq_attnpool/out/bias',
            'joint_transformer/layer_00/attention_layer/qkv/kernel', 'joint_transformer/layer_01/mlp_layer/out/kernel',
            'span_encoder/transformer/layer_00/pre_attn_ln/scale', 'audio_encoder/transformer/layer_01/attention_layer/qkv/bias',
            'vision_encoder/transformer/layer_00/attention_layer/attn_proj/kernel', 'joint_transformer/final_ln/bias']
    names = [n for n, _ in leaves]
    assert all(n in names for n in must), [n for n in must if n not in names]
    extra = [n for n in names if n not in must]
    chosen = must + [extra[i] for i in rng.choice(len(extra), size=10, replace=False)]
    assert len(chosen) >= 20
    by_name = dict(leaves)
    for name in chosen:
        t = by_name[name]
        assert t.grad is not None, name
        direction = torch.from_numpy(rng.standard_normal(tuple(t.shape)))
        direction = direction / direction.norm()
        eps = 1e-4 * max(float(t.detach().norm()), 1e-2)        # a unit-norm direction: rounding noise ~1e-15 / eps, truncation ~eps^2
        vals = []
        for sgn in (1.0, -1.0):
            with torch.no_grad():
                t.add_(sgn * eps * direction)
                vals.append(float(loss_of(params)))
                t.sub_(sgn * eps * direction)
        fd = (vals[0] - vals[1]) / (2 * eps)
        an = float((t.grad * direction).sum())
        assert abs(fd - an) <= 1e-4 * max(abs(an), abs(fd)) + 2e-8, (name, fd, an)


def test_bf16_storage_hook_is_off_by_default_and_rounds_at_bf16_scale():
    """oracle/ref_torch.py `bf16_storage()` (round 6: the oracle evaluated with the bf16 program's storage format, used by the tight forward bound of
    tests/test_trainer_fullsize_gpu.py): outside the context the restatement is bit-for-bit what it was (the float64 cross-check above runs on
    it); inside, every output is bf16-representable and within a few 2^-8 of the exact forward; the context restores the previous state."""
    cfg, tree, batch, splits, z = _setup(5, B=1)
    tree = tree_to(tree_to(tree, torch.bfloat16), torch.float32)          # bf16-representable weights, as the bf16 program reads them
    osp, oz = oracle_draws(splits, z)
    ob = oracle_batch(batch, torch.float32)
    with torch.no_grad():
        exact = R.pretrain_forward(tree, cfg, ob, osp, oz)
        with R.bf16_storage():
            stored = R.pretrain_forward(tree, cfg, ob, osp, oz)
        again = R.pretrain_forward(tree, cfg, ob, osp, oz)
    assert R._STORE is None
    n = 0
    for k in exact:
        for k2 in ('x', 'y', 'y_extra'):
            if k2 in exact[k]:
                a, s, b = exact[k][k2], stored[k][k2], again[k][k2]
                assert torch.equal(a, b), 'the hook must leave no trace outside its context'
                assert torch.equal(s, s.to(torch.bfloat16).to(torch.float32)), 'outputs are stored in bf16'
                e = float((a - s).norm() / a.norm())
                assert 1e-4 < e < 3e-2, (k, k2, e)
                n += 1
    assert n == 7
