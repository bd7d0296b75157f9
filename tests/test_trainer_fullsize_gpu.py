"""The Trainer at the sizes that are benchmarked (round 4, item 3c: the large model at B = 4 too).

* base and large, B = 4 records per GPU (BASELINE config 2: the workload of bench.py's headline; config 3's per-GPU workload): forward x / y tensors and loss, every
  parameter gradient of an injected dE through the bucketed backward, two optimizer steps leaf by leaf, and hipGraph replays equal
  to eager steps bit for bit -- against the oracle on the host cores.  At B = 4 (M = 15 424 rows) the GEMM dispatcher picks other
  kernels than at the B = 1 of tests/test_pretrain_gpu.py (ping-pong 256 x 256 / 256 x 192 tiles over three rounds of the CUs, two
  layers' weight gradients per launch).
* large and large-resadapt (B = 2): Trainer.capture + replays equal eager steps bit for bit.

Tolerances = 2-3 x the errors measured on MI355X (printed by the test): forward rel-L2 <= 2e-2 (measured 9.5e-3 with the jittered
LayerNorm parameters of this test), loss 2e-3 relative (measured 3e-6), gradients |d| <= 2.5e-2 |g| + 5e-3 max|g| and cosine >= 0.999
(measured: worst 9.5e-3 on a significant leaf, |d| <= 8.8e-3 max|g| on every leaf, lowest cosine 0.99996).
"""
import os

import numpy as np
import pytest
import torch

from tests.test_pretrain_gpu import SECTIONS
from tests.util import oracle_batch, oracle_draws, relerr, tree_to

pytestmark = pytest.mark.gpu

# Round 6 (review item 6).  Measured on MI355X, B = 4: program vs the oracle evaluated with bf16 storage 4.3e-3 .. 6.4e-3 (base), 6.5e-3 .. 8.2e-3 (large) -- about
# 0.65 of either one's distance to the exact forward (8.1e-3 .. 9.5e-3 base, 1.1e-2 .. 1.2e-2 large): two bf16 evaluations of a 24-layer-deep forward are
# two realisations of the same rounding noise (the softmax weights alone are rounded in another form by the kernels: unnormalised, per key tile), so their
# mutual distance cannot go far below sqrt(2) x 0.5 of that noise and the 3e-3 the review hoped for is not there to be had.  What the evaluation does give
# is a bound that SCALES with the noise: the program may not be further from the exact forward than the restatement's own bf16 evaluation is, times 1.25
# (measured ratios 1.01 .. 1.04) -- a systematic error of 0.75 x the noise (~0.7 % of a tensor's norm) fails it, where the fixed 2e-2 let 1.7 % through.
TOL_BF16_STORAGE = 1.5e-2      # rel-L2 of every x / y tensor against the oracle with bf16 storage
RATIO_BF16_STORAGE = 1.25      # (program - exact) / (bf16-storage oracle - exact), per tensor


def _snapshot(p):
    return {k: getattr(p, k).clone() for k in ('master', 'mu', 'nu')}


def _restore(tr, snap, step):
    p = tr.params
    for k, v in snap.items():
        getattr(p, k).copy_(v)
    p.refresh_work()
    tr.state.step = step


@pytest.mark.parametrize('model_name', ['base', 'large'])
def test_b4_trainer_step_against_oracle(dev, model_name):
    """base: BASELINE config 2, the headline workload; large: config 3's per-GPU workload (bench.py `secondary.large_b4`) -- both at the
    benchmarked B = 4 records per GPU."""
    from merlot_reserve_amd.config import load_config
    from merlot_reserve_amd.synthetic import make_batch, make_draws
    from merlot_reserve_amd.trainer import Trainer
    from oracle import ref_torch as R
    cfg = load_config(model_name)
    cfg['optimizer'].update(num_warmup_steps=2, learning_rate=1e-3)          # a visible second update (the first is zero: count starts at 0)
    B = 4
    tr = Trainer(cfg, B, dev, seed=0)
    p, eng = tr.params, tr.engine
    g = torch.Generator().manual_seed(7)

    def jitter(t):      # non-trivial LayerNorm / bias parameters; then every leaf bf16-representable: the oracle reads the same numbers
        return {k: jitter(v) for k, v in t.items()} if isinstance(t, dict) else (t + 0.05 * torch.randn(t.shape, generator=g) if t.dim() == 1 else t)
    p.load_tree(tree_to(tree_to(jitter(p.master_tree()), torch.bfloat16), torch.float32))
    batch = make_batch(cfg, B, seed=11, device=dev)
    batch2 = make_batch(cfg, B, seed=12, device=dev)
    draws = make_draws(cfg, B, seed=11)
    plan, plan2 = tr.plan(batch, draws), tr.plan(batch2)
    assert eng.tv.M == 15424 and eng.tj.M == 15360

    # ---- (1) forward + loss
    tr.forward_and_loss(batch, plan=plan)
    torch.cuda.synchronize()
    outs = {k: {k2: v.float().cpu().clone() for k2, v in d_.items() if torch.is_tensor(v)} for k, d_ in eng.outputs().items()}
    li = eng.loss_info()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    params = R.tree_map(lambda t: t.clone().requires_grad_(True), p.master_tree())
    osp, oz = oracle_draws(*draws)
    preds = R.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz)
    with torch.no_grad():
        loss, info = R.loss_fn_given_preds([preds])
    worst = 0.0
    for k, k2, _ in SECTIONS:
        e = relerr(outs[k][k2], preds[k][k2])
        worst = max(worst, e)
        assert e <= 2e-2, f'B = 4 forward {k}/{k2}: rel err {e:.3e}'
    assert abs(li['loss'] - float(loss)) <= 2e-3 * abs(float(loss)), (li['loss'], float(loss))
    print(f'{model_name} B=4 forward: worst rel-L2 {worst:.3e}; loss {li["loss"]:.5f} vs oracle {float(loss):.5f}')

    # ---- (1b) the same forward against the oracle evaluated WITH the program's storage format (oracle/ref_torch.py bf16_storage: bf16 where
    # engine.py stores bf16, fp32 accumulation in between).  Two bounds come out of it.  (i) program vs that evaluation: what is left is the
    # softmax weights' rounding (the kernels round the UNnormalised exp(s - running max) of a key tile, the restatement the normalised weights: same
    # magnitude, other bits), 24 layers deep -- half of the distance to the exact forward.  (ii) the program may not be further from the EXACT forward
    # than the restatement's own bf16 evaluation is (x RATIO_BF16_STORAGE): a mis-routed weight, epilogue or scale adds to the program's distance only.
    with torch.no_grad(), R.bf16_storage():
        preds_q = R.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz)
        loss_q, _ = R.loss_fn_given_preds([preds_q])
    worst_q, worst_ratio = 0.0, 0.0
    for k, k2, _ in SECTIONS:
        e = relerr(outs[k][k2], preds_q[k][k2])
        e_exact, q_exact = relerr(outs[k][k2], preds[k][k2]), relerr(preds_q[k][k2], preds[k][k2])
        worst_q, worst_ratio = max(worst_q, e), max(worst_ratio, e_exact / q_exact)
        print(f'  {k}/{k2}: program vs bf16-storage oracle {e:.3e}; vs exact: program {e_exact:.3e}, bf16-storage oracle {q_exact:.3e}')
        assert e <= TOL_BF16_STORAGE, f'B = 4 forward vs the bf16-storage oracle {k}/{k2}: rel err {e:.3e}'
        assert e_exact <= RATIO_BF16_STORAGE * q_exact, f'B = 4 forward {k}/{k2}: {e_exact:.3e} from the exact forward, the oracle in bf16 storage {q_exact:.3e}'
    assert abs(li['loss'] - float(loss_q)) <= 5e-4 * abs(float(loss_q)), (li['loss'], float(loss_q))
    print(f'{model_name} B=4 forward vs bf16-storage oracle: worst rel-L2 {worst_q:.3e}, worst distance ratio {worst_ratio:.2f}; loss {li["loss"]:.5f} vs {float(loss_q):.5f}')
    del preds_q

    # ---- (2) every gradient leaf of an injected dE through the Trainer's bucketed backward (no update)
    dE = (torch.randn(eng.R, eng.d.H, generator=g) * 1e-2).to(torch.bfloat16)
    eng.dE.copy_(dE.to(dev))
    tr.backward_and_reduce(update=False)
    torch.cuda.synchronize()
    total = 0.0
    for k, k2, name in SECTIONS:
        o, n = eng.sec[name]
        total = total + (preds[k][k2] * dE[o:o + n].float()).sum()
    total.backward()
    del preds, total
    gt = p.grad_tree()
    leaves = [(name, t.grad if t.grad is not None else torch.zeros_like(t)) for name, t in R.tree_leaves(params)]
    gmax = max(float(gr.norm()) for _, gr in leaves)
    bad, worst, wcos, wabs = [], (0.0, ''), 1.0, 0.0
    for name, gr in leaves:
        mine = gt
        for part in name.split('/'):
            mine = mine[part]
        gn, err = float(gr.norm()), float((mine.double() - gr.double()).norm())
        cos = float((mine.double().flatten() @ gr.double().flatten()) / (mine.double().norm() * gr.double().norm() + 1e-30))
        if gn > 5e-2 * gmax:
            worst = max(worst, (err / gn, name))
            wcos = min(wcos, cos)
        wabs = max(wabs, err / gmax)
        if err > 2.5e-2 * gn + 5e-3 * gmax or (gn > 5e-2 * gmax and cos < 0.999):
            bad.append((name, err, gn, cos))
    print(f'{model_name} B=4 backward: {len(leaves)} leaves, worst rel. error on a significant leaf {worst[0]:.3e} ({worst[1]}), lowest cosine {wcos:.6f}, '
          f'worst |d| / max|g| over all leaves {wabs:.3e}')
    assert not bad, bad[:10]
    del leaves, params

    if model_name != 'base':       # (3) and (4) do not depend on the model size beyond what test_large_trainer_capture_replay_equals_eager covers
        return
    # ---- (3) two optimizer steps on these gradients, leaf by leaf (pretrain/optimization.py:54-114, 180-195)
    before = p.master_tree()
    grads = tree_to(p.grad_tree(), torch.float32)
    mu = {n: torch.zeros_like(t, dtype=torch.bfloat16) for n, t in R.tree_leaves(before)}
    nu = {n: torch.zeros_like(t, dtype=torch.bfloat16) for n, t in R.tree_leaves(before)}
    want = dict(R.tree_leaves(before))
    gl = dict(R.tree_leaves(grads))
    for count in range(2):
        tr.state.apply_gradients()
        for n in want:
            want[n], mu[n], nu[n] = R.adam_bf16_apply(want[n], gl[n], mu[n], nu[n], count, cfg['optimizer'])
    torch.cuda.synchronize()
    after = dict(R.tree_leaves(p.master_tree()))
    moved = 0
    for n, w in want.items():
        assert torch.allclose(after[n], w, rtol=2e-6, atol=1e-9), (n, float((after[n] - w).abs().max()))
        moved += int((w != dict(R.tree_leaves(before))[n]).any())
    assert moved > 400, 'the second update must move (almost) every leaf'
    work = dict(R.tree_leaves(p.work_tree()))
    for n in ('vision_encoder/transformer/layer_03/mlp_layer/intermediate/kernel', 'joint_transformer/layer_11/attention_layer/qkv/kernel'):
        assert torch.equal(work[n], after[n].to(torch.bfloat16)), 'working copy = bf16(master)'
        o, cnt = p.offsets[n]
        assert torch.equal(p.wT[n].cpu(), work[n].reshape(p.w[n].shape).t().contiguous()), 'transposed working copy follows the update'

    # ---- (4) hipGraph replays == eager steps, bit for bit, at this size (two steps on two batches)
    snap, step0 = _snapshot(p), tr.state.step
    tr.train_step(batch, plan=plan)
    tr.train_step(batch2, plan=plan2)
    torch.cuda.synchronize()
    eager_master, eager_loss = p.master.clone(), eng.loss_acc.clone()
    _restore(tr, snap, step0)
    tr.capture(batch)
    tr.train_step_graph(batch, plan)
    tr.train_step_graph(batch2, plan2)
    torch.cuda.synchronize()
    assert torch.equal(eng.loss_acc, eager_loss)
    assert torch.equal(p.master, eager_master), 'graph replay and eager steps must agree bit for bit'


@pytest.mark.parametrize('case,B', [('large', 2), ('large_resadapt', 2)])
def test_large_trainer_capture_replay_equals_eager(dev, case, B):
    """Trainer.capture + replay == eager, bit for bit, on the large model (nh = 16, H = 1024: other tile widths and the one-layer
    weight-gradient launches) and on its resolution-adaptation variant (ViT S = 577, joint S = 1312)."""
    from merlot_reserve_amd.config import load_config, resadapt_config
    from merlot_reserve_amd.synthetic import make_batch
    from merlot_reserve_amd.trainer import Trainer
    cfg = resadapt_config('large', grid=(18, 32)) if case.endswith('resadapt') else load_config('large')
    cfg['optimizer'].update(num_warmup_steps=2)
    tr = Trainer(cfg, B, dev, seed=1)
    p, eng = tr.params, tr.engine
    batches = [make_batch(cfg, B, seed=40 + i, device=dev) for i in range(2)]
    plans = [tr.plan(b) for b in batches]
    tr.train_step(batches[0], plan=plans[0])                       # allocates everything; the first update is zero
    snap, step0 = _snapshot(p), tr.state.step
    for b, pl in zip(batches, plans):
        tr.train_step(b, plan=pl)
    torch.cuda.synchronize()
    eager_master, eager_loss = p.master.clone(), eng.loss_acc.clone()
    assert np.isfinite(tr.loss_info()['loss'])
    _restore(tr, snap, step0)
    tr.capture(batches[0])
    for b, pl in zip(batches, plans):
        tr.train_step_graph(b, pl)
    torch.cuda.synchronize()
    assert torch.equal(eng.loss_acc, eager_loss)
    assert torch.equal(p.master, eager_master)
    assert not torch.equal(p.master, snap['master']), 'the steps moved the parameters'
