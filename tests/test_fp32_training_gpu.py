"""The fp32 training step -- the reference's use_bfloat16 = False / train_step(use_bfloat16_grads=False) branch
(pretrain/pretrain_model.py:323-333, pretrain/train.py:61-67,106) -- against the fp32 oracle on identical inputs.

Both sides compute in fp32 on the same fp32 parameters, so the tolerances are those of fp32 sums taken in different orders
(stated per test), not the bf16 storage tolerances of tests/test_pretrain_gpu.py.
"""
import numpy as np
import pytest
import torch

from tests.util import oracle_batch, oracle_draws, relerr, tiny_setup

pytestmark = pytest.mark.gpu

SECTIONS = (('imgs_to_audio', 'x', 'i2a_x'), ('imgs_to_audio', 'y', 'i2a_y'), ('text_to_audio', 'x', 't2a_x'),
            ('text_to_audio', 'y', 't2a_y'), ('text_to_audio', 'y_extra', 't2a_ye'), ('stuff_to_span', 'x', 's2s_x'),
            ('stuff_to_span', 'y', 's2s_y'))


def _jitter(tree, gen):
    """Non-trivial biases / LayerNorm parameters / temperatures (the initialisers leave them at 0 / 1)."""
    if isinstance(tree, dict):
        return {k: _jitter(v, gen) for k, v in tree.items()}
    return tree + 0.05 * torch.randn(tree.shape, generator=gen) if tree.dim() == 1 else tree


def _leaf(tree, name):
    for part in name.split('/'):
        tree = tree[part]
    return tree


def _compare_grads(mine_tree, ref_grads, rel, floor, what):
    from oracle import ref_torch as R
    leaves = list(R.tree_leaves(ref_grads))
    gmax = max(float(g.norm()) for _, g in leaves)
    bad, worst = [], (0.0, '')
    for name, g in leaves:
        mine = _leaf(mine_tree, name)
        gn, err = float(g.norm()), float((mine.double() - g.double()).norm())
        if gn > 1e-2 * gmax:
            worst = max(worst, (err / gn, name))
        if err > rel * gn + floor * gmax:
            bad.append((name, err, gn))
    print(f'{what}: {len(leaves)} leaves, max|g| {gmax:.3e}, worst rel. error on a significant leaf {worst[0]:.3e} ({worst[1]})')
    assert not bad, f'{what}: ' + '; '.join(f'{n}: |d| {e:.3e} |g| {g:.3e}' for n, e, g in bad[:10])


@pytest.mark.parametrize('hidden_size,B,flags', [(128, 2, {}), (256, 1, {}), (128, 2, {'no_vision': True})], ids=['h128', 'h256', 'no_vision'])
def test_fp32_backward_parity(dev, hidden_size, B, flags):
    """Every parameter gradient of the fp32 program against autograd of the oracle,
    (a) for an injected upstream gradient dE, fp32 oracle: |d| <= 2e-5 |g| + 1e-6 max_leaf|g| per leaf (measured 7e-7);
    (b) of the contrastive loss itself, fp64 oracle: |d| <= 5e-3 |g| + 1e-4 max_leaf|g|.  At random init the towers emit nearly
        identical rows, dL/dE is a difference of near-equal vectors and the loss gradient is ill-conditioned in fp32 for ANY
        implementation: the fp32 oracle is itself 1.0e-3 (H = 128) away from its fp64 evaluation on the cls / cls_proj leaves, this
        program 6e-4 (measured) -- so (b) is checked against fp64, with a bound that only says "same function"."""
    from merlot_reserve_amd.config import Dims
    from merlot_reserve_amd.engine import PretrainEngine
    from merlot_reserve_amd.planner import build_plan
    from merlot_reserve_amd.synthetic import make_batch
    from oracle import ref_torch as R
    cfg, store, _b, splits, z = tiny_setup(B=B, seed=11, device=dev, hidden_size=hidden_size, model_flags=flags)
    batch = make_batch(cfg, B, seed=11, device=dev, float_dtype=torch.float32)
    store.load_tree(_jitter(store.master_tree(), torch.Generator().manual_seed(3)))
    eng = PretrainEngine(cfg, B, store, dev, dtype=torch.float32, train=True)
    eng.forward(batch, plan=build_plan(batch, Dims(cfg, B), splits, z))
    osp, oz = oracle_draws(splits, z)

    # (a) injected dE
    g = torch.Generator().manual_seed(1)
    dE = torch.randn(eng.R, eng.d.H, generator=g) * 1e-2
    eng.dE.copy_(dE.to(dev))
    eng.backward()
    torch.cuda.synchronize()
    params = R.tree_map(lambda t: t.clone().requires_grad_(True), store.master_tree())
    preds = R.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz)
    total = 0.0
    for k, k2, name in SECTIONS:
        o, n = eng.sec[name]
        total = total + (preds[k][k2] * dE[o:o + n]).sum()
    total.backward()
    grads = R.tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), params)
    assert store.grad32.dtype == torch.float32
    _compare_grads(store.grad32_tree(), grads, 2e-5, 1e-6, f'fp32 backward (injected dE, H={hidden_size})')

    # (b) the loss's own gradient
    eng.loss_and_grad_outputs()
    eng.backward()
    torch.cuda.synchronize()
    loss, info, _p, grads = R.loss_and_grads(R.tree_map(lambda t: t.double(), store.master_tree()), cfg, oracle_batch(batch, dtype=torch.float64), osp, oz)
    li = eng.loss_info()
    assert abs(li['loss'] - float(loss)) <= 1e-5 * abs(float(loss)), (li['loss'], float(loss))
    _compare_grads(store.grad32_tree(), grads, 5e-3, 1e-4, f'fp32 backward (loss vs fp64 oracle, H={hidden_size})')


def test_fp32_train_step_matches_the_oracle_chain(dev):
    """Three steps of train_step(state, batch, use_bfloat16_grads=False) through the reference-shaped API on an fp32 model.  Per step:
    the losses and every gradient leaf against value_and_grad of the oracle (fp64: see test_fp32_backward_parity (b)), then the
    oracle's restatement of the optax chain (bf16 mu, cube-coded bf16 nu, weight-decay mask, schedule) applied to the gradients the
    step produced, leaf by leaf: parameters to rtol 2e-6 (one fp32 rounding of the update), moments bit-exact up to one bf16 ulp
    (a product contracted into an fma on one side can land on the other side of a bf16 rounding)."""
    from merlot_reserve_amd import pretrain_model as P
    from merlot_reserve_amd.synthetic import make_batch, make_draws
    from merlot_reserve_amd.config import tiny_config
    from oracle import ref_torch as R
    B = 2
    cfg = tiny_config(hidden_size=128, seq_len=80, lang_seq_len=40)
    cfg['model']['use_bfloat16'] = False
    cfg['optimizer'].update(num_warmup_steps=2, num_train_steps=10, learning_rate=1e-3)
    model = P.MerlotReservePretrainer.from_config(cfg, device=dev, seed=5)
    batches = [make_batch(cfg, B, seed=20 + i, device=dev) for i in range(3)]
    draws = [make_draws(cfg, B, seed=30 + i) for i in range(3)]
    tree = model.init_from_dummy_batch(batches[0])
    tree = _jitter(tree, torch.Generator().manual_seed(4))
    state = P.construct_train_state(cfg['optimizer'], model, tree)
    with pytest.raises(NotImplementedError):
        P.train_step(state, batches[0], use_bfloat16_grads=True)        # an fp32 model with bf16 gradients is not a launched combination

    want = dict(R.tree_leaves(tree))
    mu = {n: torch.zeros_like(t, dtype=torch.bfloat16) for n, t in want.items()}
    nu = {n: torch.zeros_like(t, dtype=torch.bfloat16) for n, t in want.items()}

    def unflat(flat):
        out = {}
        for n, t in flat.items():
            d = out
            ks = n.split('/')
            for k in ks[:-1]:
                d = d.setdefault(k, {})
            d[ks[-1]] = t
        return out
    tr = model.trainer
    for count, (batch, (splits, z)) in enumerate(zip(batches, draws)):
        state, info = P.train_step(state, batch, use_bfloat16_grads=False, split_from_here=splits, gumbel_z=z)
        torch.cuda.synchronize()
        osp, oz = oracle_draws(splits, z)
        loss, oinfo, _p, grads = R.loss_and_grads(R.tree_map(lambda t: t.double(), unflat(want)), cfg, oracle_batch(batch, dtype=torch.float64), osp, oz)
        for k in ('imgs_to_audio', 'text_to_audio', 'stuff_to_span'):
            assert abs(info[k] - float(oinfo[k])) <= 1e-4 * abs(float(oinfo[k])), (count, k, info[k], float(oinfo[k]))
        _compare_grads(tr.params.grad32_tree(), grads, 5e-3, 1e-4, f'fp32 train step {count}: gradients vs fp64 oracle')
        gl = dict(R.tree_leaves(tr.params.grad32_tree()))             # the chain below runs on the gradients the step produced
        prev = dict(want)
        for n in want:
            want[n], mu[n], nu[n] = R.adam_bf16_apply(want[n], torch.nan_to_num(gl[n]), mu[n], nu[n], count, cfg['optimizer'])
        after = dict(R.tree_leaves(tr.params.master_tree()))
        got_mu, got_nu = dict(R.tree_leaves(tr.params._to_tree(tr.params.mu))), dict(R.tree_leaves(tr.params._to_tree(tr.params.nu)))
        step_max = max(float((want[n] - prev[n]).abs().max()) for n in want)
        for n, w in want.items():
            d = float((after[n] - w).abs().max())
            assert d <= 2e-6 * float(w.abs().max()) + 1e-12, (count, n, d, step_max)
            for name, got, ref in (('mu', got_mu[n], mu[n]), ('nu', got_nu[n], nu[n])):
                gf, rf = got.float().abs(), ref.float().abs()
                assert bool(((gf - rf).abs() <= 2.0 ** -7 * torch.maximum(gf, rf) + 1e-30).all()), (count, name, n)
            # continue the oracle chain from the device's state (so a one-ulp difference of a moment does not compound into the next step)
            want[n], mu[n], nu[n] = after[n].clone(), got_mu[n].clone(), got_nu[n].clone()
        if count == 0:
            assert step_max == 0.0, 'the schedule value of the first update is zero (optax.scale_by_schedule at count 0)'
        else:
            assert step_max > 0.0
    assert state.step == 3
    assert torch.equal(tr.params.work.float().cpu(), tr.params.master.cpu().to(torch.bfloat16).float()), 'working copy = bf16(master)'


def test_fp32_step_and_bf16_step_agree_to_bf16_accuracy(dev):
    """The same batch through the fp32 step and the bf16 step from the same (bf16-representable) parameters: the significant
    gradient leaves agree to the bf16 tolerances of tests/test_pretrain_gpu.py -- the two programs differentiate the same function."""
    from merlot_reserve_amd.config import tiny_config
    from merlot_reserve_amd.synthetic import make_batch, make_draws
    from merlot_reserve_amd.trainer import Trainer
    from oracle import ref_torch as R
    B = 2
    cfg = tiny_config(hidden_size=128, seq_len=80, lang_seq_len=40)
    batch = make_batch(cfg, B, seed=7, device=dev)
    draws = make_draws(cfg, B, seed=7)
    t16 = Trainer(cfg, B, dev, seed=2)
    t32 = Trainer(cfg, B, dev, seed=2, bf16_grads=False)
    t32.params.load_tree(R.tree_map(lambda t: t.float(), t16.params.work_tree()))
    g = torch.Generator().manual_seed(0)
    dE = (torch.randn(t16.engine.R, t16.engine.d.H, generator=g) * 1e-2).to(torch.bfloat16)
    for t in (t16, t32):
        t.forward_and_loss(batch, plan=t.plan(batch, draws))
        t.engine.dE.copy_(dE.to(dev))
        t.backward_and_reduce(update=False)
    torch.cuda.synchronize()
    g16, g32 = dict(R.tree_leaves(t16.params.grad_tree())), dict(R.tree_leaves(t32.params.grad32_tree()))
    gmax = max(float(v.norm()) for v in g32.values())
    for n, ref in g32.items():
        err, gn = float((g16[n].double() - ref.double()).norm()), float(ref.norm())
        assert err <= 8e-2 * gn + 1.5e-2 * gmax, (n, err, gn)
    assert abs(t16.engine.loss_info()['loss'] - t32.engine.loss_info()['loss']) <= 2e-2 * abs(t32.engine.loss_info()['loss'])
