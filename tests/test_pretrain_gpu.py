"""End-to-end parity of the HIP pretraining step against the oracle (fp32 restatement of the reference) on the tiny
configuration: forward outputs, loss, and every parameter gradient.

Tolerances (stated, bf16 compute vs fp32 oracle; the oracle reads the SAME bf16-rounded weights and inputs):
  forward x / y tensors: relative L2 <= 2e-2   (activations are stored in bf16: 2^-8 per rounding, a few dozen roundings)
  loss: |d| <= 2e-2 * |loss|
  gradients: relative L2 <= 8e-2 and cosine >= 0.995 per leaf (bf16 activations, bf16 gradient storage)
Integer decisions (selections, pooling targets) are bit-exact by construction: tests/test_planner_vs_oracle.py.
"""
import numpy as np
import pytest
import torch

from tests.util import oracle_batch, oracle_draws, relerr, tiny_setup, tree_to

pytestmark = pytest.mark.gpu


def _setup(dev, B=2, seed=3, hidden_size=128, model_flags=None, data_flags=None):
    from merlot_reserve_amd.config import Dims
    from merlot_reserve_amd.engine import PretrainEngine
    from merlot_reserve_amd.planner import build_plan
    cfg, store, batch, splits, z = tiny_setup(B=B, seed=seed, device=dev, hidden_size=hidden_size, model_flags=model_flags, data_flags=data_flags)
    eng = PretrainEngine(cfg, B, store, dev)
    plan = build_plan(batch, Dims(cfg, B), splits, z)
    eng.forward(batch, plan=plan)
    torch.cuda.synchronize()
    return cfg, store, eng, batch, splits, z


SECTIONS = (('imgs_to_audio', 'x', 'i2a_x'), ('imgs_to_audio', 'y', 'i2a_y'), ('text_to_audio', 'x', 't2a_x'),
            ('text_to_audio', 'y', 't2a_y'), ('text_to_audio', 'y_extra', 't2a_ye'), ('stuff_to_span', 'x', 's2s_x'),
            ('stuff_to_span', 'y', 's2s_y'))


# more than one sequence per kind and record (pretrain/pretrain_model.py:99-137: the vision input, the video-source table and the audio spans are tiled)
MULTI_SEQ = {'data': dict(num_audio2text_seqs=2, num_text2audio_seqs=3, num_text_seqs=2)}


# do_rotary = False: pretrain/pretrain_model.py:146-148 (no joint coordinates -> the learned `pe` of mreserve/modeling.py:335-341)
@pytest.mark.parametrize('flags', [{}, {'no_vision': True}, MULTI_SEQ, {'do_rotary': False}], ids=['stock', 'no_vision', 'multi_seq', 'learned_pe'])     # no_vision: pretrain/pretrain_model.py:61-63
def test_tiny_forward_and_loss_parity(dev, flags):
    from oracle import ref_torch as R
    cfg, store, eng, batch, splits, z = _setup(dev, model_flags={k: v for k, v in flags.items() if k != 'data'}, data_flags=flags.get('data'))
    eng.loss_and_grad_outputs()
    torch.cuda.synchronize()
    params = tree_to(store.work_tree(), torch.float32)
    osp, oz = oracle_draws(splits, z)
    with torch.no_grad():
        preds = R.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz)
        loss, info = R.loss_fn_given_preds([preds])
    outs = eng.outputs()
    for k, k2, _ in SECTIONS:
        e = relerr(outs[k][k2], preds[k][k2])
        print(f'forward {k}/{k2}: rel-L2 {e:.3e}')
        assert e <= 2e-2, f'{k}/{k2}: rel err {e:.3e}'
    li = eng.loss_info()
    print('loss', li['loss'], float(loss))
    for k in ('imgs_to_audio', 'text_to_audio', 'stuff_to_span'):
        assert abs(li[k] - float(info[k])) <= 2e-2 * abs(float(info[k])), (k, li[k], float(info[k]))
    assert abs(li['loss'] - float(loss)) <= 2e-2 * abs(float(loss))
    for t in ('text2audio', 'audio2text', 'random_text'):
        k = f'_stuff_to_span_from_{t}'
        assert abs(li[k] - float(info[k])) <= 3e-2 * abs(float(info[k])) + 1e-3, (k, li[k], float(info[k]))


def test_loss_gradient_wrt_outputs(dev):
    """loss_fn_given_preds and dL/d(x, y) on well-conditioned embeddings (a random-init tower emits near-identical
    rows, for which the contrastive gradient is a difference of near-equal vectors: ill-conditioned in bf16 for the
    reference too).  E holds bf16 values, the oracle reads the same values: tolerance 1e-2 (bf16 storage of dE)."""
    from oracle import ref_torch as R
    cfg, store, eng, batch, splits, z = _setup(dev)
    g = torch.Generator().manual_seed(0)
    E = torch.randn(eng.R, eng.d.H, generator=g)
    E = (E / E.norm(dim=-1, keepdim=True) * 1.6).to(torch.bfloat16)
    eng.E.copy_(E.to(dev))
    eng.loss_and_grad_outputs()
    torch.cuda.synchronize()
    Ef = E.float().requires_grad_(True)
    preds = {}
    for k, k2, name in SECTIONS:
        o, n = eng.sec[name]
        preds.setdefault(k, {})[k2] = Ef[o:o + n]
    preds['stuff_to_span']['_sources'] = torch.as_tensor(eng.plan['t2sp_src']).long()
    loss, info = R.loss_fn_given_preds([preds])
    loss.backward()
    li = eng.loss_info()
    assert abs(li['loss'] - float(loss)) <= 1e-3 * abs(float(loss)), (li['loss'], float(loss))
    for k, k2, name in SECTIONS:
        o, n = eng.sec[name]
        e = relerr(eng.dE[o:o + n], Ef.grad[o:o + n])
        print(f'dE {name}: {e:.3e}')
        assert e <= 1e-2, (name, e)


@pytest.mark.parametrize('flags', [{}, {'no_vision': True}, MULTI_SEQ, {'do_rotary': False}], ids=['stock', 'no_vision', 'multi_seq', 'learned_pe'])
def test_tiny_backward_parity(dev, flags):
    """Backward of the whole forward graph for a GIVEN upstream gradient dE (so the check is independent of the
    conditioning of the loss at random init): every parameter gradient against autograd of the oracle."""
    from oracle import ref_torch as R
    cfg, store, eng, batch, splits, z = _setup(dev, model_flags={k: v for k, v in flags.items() if k != 'data'}, data_flags=flags.get('data'))
    g = torch.Generator().manual_seed(1)
    dE = (torch.randn(eng.R, eng.d.H, generator=g) * 1e-2).to(torch.bfloat16)
    eng.dE.copy_(dE.to(dev))
    eng.backward()
    torch.cuda.synchronize()
    params = tree_to(store.work_tree(), torch.float32)
    params = R.tree_map(lambda t: t.clone().requires_grad_(True), params)
    osp, oz = oracle_draws(splits, z)
    preds = R.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz)
    total = 0.0
    for k, k2, name in SECTIONS:
        o, n = eng.sec[name]
        total = total + (preds[k][k2] * dE[o:o + n].float()).sum()
    total.backward()
    grads = R.tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), params)
    gt = store.grad_tree()
    gmax = max(float(g.norm()) for _, g in R.tree_leaves(grads))
    worst = []
    for name, g in R.tree_leaves(grads):
        mine = gt
        for part in name.split('/'):
            mine = mine[part]
        gn = float(g.norm())
        err = float((mine.double() - g.double()).norm())
        cos = float((mine.double().flatten() @ g.double().flatten()) / (mine.double().norm() * g.double().norm() + 1e-30))
        worst.append((err / (gn + 1e-30), cos, name, err, gn))
    worst.sort(key=lambda t: -t[3] / (8e-2 * t[4] + 1.5e-2 * gmax))
    msg = '\n'.join(f'rel={e:.3e} cos={c:.5f} abs={a:.3e} |g|={n2:.3e} {n}' for e, c, n, a, n2 in worst[:30])
    print('gmax', gmax)
    print('worst gradient leaves:\n' + msg)
    # |mine - ref| <= 8e-2 |ref| + 1.5e-2 max_leaf|ref|  (second term: bf16 noise floor of near-cancelling sums);
    # direction checked on every leaf that carries a non-negligible gradient
    bad = [(e, c, n) for e, c, n, a, n2 in worst if a > 8e-2 * n2 + 1.5e-2 * gmax or (n2 > 5e-2 * gmax and c < 0.995)]
    assert not bad, 'gradient parity failures:\n' + '\n'.join(f'{e:.3e} cos={c:.5f} {n}' for e, c, n in bad)


def test_two_rank_contrastive_path_on_one_gpu(dev):
    """The N > 1 loss path (rank-major all-gather of E, logits against every rank's keys, own-block offset, gradient wrt
    the gathered keys reduce-scattered back) driven for two ranks inside one process, collectives emulated with
    tensor copies, against the oracle's virtual-device loss_fn_given_preds."""
    from merlot_reserve_amd.config import Dims
    from merlot_reserve_amd.engine import PretrainEngine
    from merlot_reserve_amd.params import ParamStore
    from merlot_reserve_amd.planner import build_plan
    from oracle import ref_torch as R
    from tests.util import tiny_setup
    B, world = 8, 2
    engines, Es = [], []
    for r in range(world):
        cfg, store, batch, splits, z = tiny_setup(B=B, seed=3 + r, device=dev)
        eng = PretrainEngine(cfg, B, store, dev, rank=r, world=world)
        eng.set_plan(build_plan(batch, Dims(cfg, B), splits, z))
        g = torch.Generator().manual_seed(10 + r)
        E = torch.randn(eng.R, eng.d.H, generator=g)
        E = (E / E.norm(dim=-1, keepdim=True) * 1.6).to(torch.bfloat16)
        eng.E.copy_(E.to(dev))
        engines.append(eng)
        Es.append(E.float().requires_grad_(True))
    E_all = torch.stack([e.E for e in engines])
    dE_alls = [torch.zeros_like(E_all) for _ in range(world)]
    for r, eng in enumerate(engines):
        eng.loss_and_grad_outputs(E_all, dE_alls[r])
    torch.cuda.synchronize()
    red = sum(d.float() for d in dE_alls)                       # reduce-scatter(sum): rank r receives block r
    preds = []
    for r, eng in enumerate(engines):
        p = {}
        for k, k2, name in SECTIONS:
            o, n = eng.sec[name]
            p.setdefault(k, {})[k2] = Es[r][o:o + n]
        p['stuff_to_span']['_sources'] = torch.as_tensor(eng.plan['t2sp_src']).long()
        preds.append(p)
    losses = [R.loss_fn_given_preds(preds, rank=r) for r in range(world)]
    sum(l for l, _ in losses).backward()
    for r, eng in enumerate(engines):
        li = eng.loss_info()
        assert abs(li['loss'] - float(losses[r][0])) <= 1e-3 * abs(float(losses[r][0])), (r, li['loss'], float(losses[r][0]))
        for t in ('text2audio', 'audio2text', 'random_text'):
            k = f'_stuff_to_span_from_{t}'
            assert abs(li[k] - float(losses[r][1][k])) <= 2e-3 * abs(float(losses[r][1][k])) + 1e-4
        total = eng.dE.float() + red[r]
        for k, k2, name in SECTIONS:
            o, n = eng.sec[name]
            e = relerr(total[o:o + n], Es[r].grad[o:o + n])
            print(f'rank {r} dE {name}: {e:.3e}')
            assert e <= 1e-2, (r, name, e)


@pytest.mark.parametrize('hidden_size,B', [(128, 2), (256, 1)])
def test_fp32_forward_parity_1e3(dev, hidden_size, B):
    """The north-star forward bar: the fp32 program (mr_f32_* kernels on the fp32 master weights -- the reference's
    use_bfloat16 = False arithmetic) against the fp32 oracle on identical inputs: every x / y tensor of the three
    objectives and the loss within 1e-3 relative (measured ~1e-5); the integer decisions are shared (same plan)."""
    from merlot_reserve_amd.config import Dims
    from merlot_reserve_amd.engine import PretrainEngine
    from merlot_reserve_amd.planner import build_plan
    from merlot_reserve_amd.synthetic import make_batch
    from oracle import ref_torch as R
    cfg, store, _b, splits, z = tiny_setup(B=B, seed=11, device=dev, hidden_size=hidden_size)
    batch = make_batch(cfg, B, seed=11, device=dev, float_dtype=torch.float32)
    g = torch.Generator().manual_seed(3)                   # non-trivial biases / LN parameters / temperatures
    tree = store.master_tree()

    def jitter(t):
        return {k: jitter(v) for k, v in t.items()} if isinstance(t, dict) else (t + 0.05 * torch.randn(t.shape, generator=g) if t.dim() == 1 else t)
    store.load_tree(jitter(tree))
    eng = PretrainEngine(cfg, B, store, dev, dtype=torch.float32)
    eng.forward(batch, plan=build_plan(batch, Dims(cfg, B), splits, z))
    eng.loss_and_grad_outputs()
    torch.cuda.synchronize()
    osp, oz = oracle_draws(splits, z)
    with torch.no_grad():
        preds = R.pretrain_forward(store.master_tree(), cfg, oracle_batch(batch), osp, oz)
        loss, info = R.loss_fn_given_preds([preds])
    outs = eng.outputs()
    worst = 0.0
    for k, k2, _ in SECTIONS:
        assert outs[k][k2].dtype == torch.float32
        e = relerr(outs[k][k2], preds[k][k2])
        worst = max(worst, e)
        assert e <= 1e-3, f'{k}/{k2}: rel err {e:.3e}'
        emax = float(((outs[k][k2].cpu() - preds[k][k2]).abs().max() / preds[k][k2].abs().max()))
        assert emax <= 1e-3, f'{k}/{k2}: max rel err {emax:.3e}'
    li = eng.loss_info()
    print(f'fp32 forward parity H={hidden_size}: worst rel-L2 {worst:.2e}, loss {li["loss"]:.6f} vs {float(loss):.6f}')
    for k in ('imgs_to_audio', 'text_to_audio', 'stuff_to_span'):
        assert abs(li[k] - float(info[k])) <= 1e-3 * abs(float(info[k])), (k, li[k], float(info[k]))
    assert abs(li['loss'] - float(loss)) <= 1e-3 * abs(float(loss))
    with pytest.raises(AssertionError):
        eng.backward()                                      # built without train=True: forward + loss only


def test_training_reduces_the_contrastive_loss(dev):
    """1000 graph-replayed steps on two alternating tiny batches: the contrastive loss falls well below its chance level
    (7.34 = sum over objectives of ln(#candidates)): forward, backward and the folded optimizer work together over many
    steps.  (text_to_audio and stuff_to_span are memorised; imgs_to_audio stays at chance on iid-noise frames, whose ViT
    embeddings are nearly identical.)  The descent goes plateau by plateau (6.0 -> 5.6 -> 5.2 -> 4.6 -> 4.0 ...) and WHEN a
    plateau is left is sensitive to single-ulp differences of bf16 gradients (scripts/step_ab.py, scripts/conv_dbg.py: five
    reduction schedules whose gradients differ by one ulp in < 10 elements are anywhere between 5.66 and 4.2 after 300 steps, and
    between 4.15 and 3.78 after 1000), so the bounds are loose at 100 steps and on the 1000-step value, which every schedule seen
    so far passes with margin."""
    from merlot_reserve_amd.config import tiny_config
    from merlot_reserve_amd.synthetic import make_batch
    from merlot_reserve_amd.trainer import Trainer
    cfg = tiny_config(seq_len=80, lang_seq_len=40)
    cfg['optimizer'].update(num_warmup_steps=10, learning_rate=1e-3, num_train_steps=400)
    B = 2
    tr = Trainer(cfg, B, dev, seed=1)
    batches = [make_batch(cfg, B, seed=70 + i, device=dev) for i in range(2)]
    plans = [tr.plan(b) for b in batches]
    tr.train_step(batches[0], plan=plans[0])
    first = tr.loss_info()['loss']
    tr.capture(batches[0])
    hist = []
    for i in range(1000):
        tr.train_step_graph(batches[i % 2], plans[i % 2])
        if i % 100 == 99:
            hist.append(tr.loss_info()['loss'])
    assert all(np.isfinite(hist)), hist
    assert hist[0] < 0.9 * first and hist[-1] < 0.65 * first, (first, hist)
    assert tr.state.step == 1001


def _full_size_config(case):
    """BASELINE configs 2-4 at full width and sequence length.  `*_resadapt` = pretrain/train_fixres.py:78-90 (grid 18x32:
    ViT S = 577, joint S = 1312 -- at hidden 1024 the regime where the reference switches to jax.checkpoint attention,
    mreserve/modeling.py:202,231).  `depth` caps the layers per tower (only for the large resadapt BACKWARD check, whose
    full-depth autograd on the host would need ~60 GB); widths, heads, sequence lengths and batch structure stay."""
    from merlot_reserve_amd.config import load_config, resadapt_config
    name, _, variant = case.partition('_')
    grid = (24, 24) if variant.startswith('resadapt24') else (18, 32)      # the two grids train_fixres.py:78 alternates over processes
    cfg = resadapt_config(name, grid=grid) if variant.startswith('resadapt') else load_config(name)
    if variant.endswith('shallow'):
        cfg['model'].update(vit_num_layers=3, joint_num_layers=3, audio_num_layers=2, span_num_layers=1)
    return cfg


@pytest.mark.parametrize('case', ['base', 'large', 'base_resadapt', 'large_resadapt', 'base_resadapt24'])
def test_full_size_forward_parity(dev, case):
    """Full-size check (BASELINE config 2 / 3 / 4 models, one record): the fp32 program against the fp32 oracle on the
    host cores (1e-3), and the bf16 training program against the fp32 program on the same weights rounded to bf16 (2e-2)."""
    import os
    from merlot_reserve_amd.config import Dims
    from merlot_reserve_amd.engine import PretrainEngine
    from merlot_reserve_amd.params import ParamStore
    from merlot_reserve_amd.planner import build_plan
    from merlot_reserve_amd.synthetic import make_batch, make_draws
    from oracle import ref_torch as R
    cfg = _full_size_config(case)
    B = 1
    store = ParamStore(cfg, dev, seed=0, with_optimizer=False)
    store.load_tree(tree_to(store.work_tree(), torch.float32))          # master := bf16-representable values (both programs read the same numbers)
    batch16 = make_batch(cfg, B, seed=5, device=dev)
    batch32 = dict(batch16, images=batch16['images'].float(), audio_clips=batch16['audio_clips'].float())
    splits, z = make_draws(cfg, B, seed=5)
    d = Dims(cfg, B)
    if 'resadapt' in case:
        assert (d.Sv, d.Sj) == (577, 1312)         # 18 x 32 and 24 x 24 have the same patch count; the 2-D coordinates and the pooled grid differ
    plan = build_plan(batch16, d, splits, z)
    e32 = PretrainEngine(cfg, B, store, dev, dtype=torch.float32)
    e32.forward(batch32, plan=plan)
    e32.loss_and_grad_outputs()
    out32 = {k: {k2: v.clone() for k2, v in d_.items()} for k, d_ in e32.outputs().items()}
    loss32 = e32.loss_info()['loss']
    del e32
    torch.cuda.empty_cache()
    e16 = PretrainEngine(cfg, B, store, dev)
    e16.forward(batch16, plan=plan)
    e16.loss_and_grad_outputs()
    torch.cuda.synchronize()
    for k, k2, _ in SECTIONS:
        e = relerr(e16.outputs()[k][k2], out32[k][k2])
        assert e <= 2e-2, f'bf16 vs fp32 program, {k}/{k2}: {e:.3e}'
    assert abs(e16.loss_info()['loss'] - loss32) <= 2e-2 * abs(loss32)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    osp, oz = oracle_draws(splits, z)
    with torch.no_grad():
        preds = R.pretrain_forward(store.master_tree(), cfg, oracle_batch(batch32), osp, oz)
        loss, _ = R.loss_fn_given_preds([preds])
    for k, k2, _ in SECTIONS:
        e = relerr(out32[k][k2], preds[k][k2])
        assert e <= 1e-3, f'fp32 program vs oracle, {k}/{k2}: {e:.3e}'
    assert abs(loss32 - float(loss)) <= 1e-3 * abs(float(loss))
    print(f'{case}-size parity: loss fp32 {loss32:.6f} oracle {float(loss):.6f} bf16 {e16.loss_info()["loss"]:.6f}')


@pytest.mark.parametrize('case', ['base_resadapt', 'large_resadapt', 'base_resadapt24'])
def test_full_size_backward_parity(dev, case):
    """Every parameter gradient of the FULL-width model (one record) for an injected upstream gradient dE against autograd
    of the oracle on the host cores: |d| <= 3e-2 |g| + 6e-3 max|g| and cos >= 0.9995 on every leaf that carries gradient = 3 x
    what is measured (worst relative error on a significant leaf 0.6-1.2e-2, lowest cosine 0.99992; round 2 allowed 8e-2 / 0.995).  base / large: the 256-row GEMM tiles, grouped weight gradients (256 x 256
    tiles for large, nh = 16) and sequences of 241 / 640; *_resadapt: the flash backward at S = 577 / 1312.  large_resadapt (BASELINE config 4) runs at
    FULL depth since round 5 (24 / 24 / 12 / 4 layers: the host holds the autograd graph, ~60 GB); the stock large model's backward is checked at the
    benchmarked B = 4 in tests/test_trainer_fullsize_gpu.py."""
    import os
    from merlot_reserve_amd.config import Dims
    from merlot_reserve_amd.engine import PretrainEngine
    from merlot_reserve_amd.params import ParamStore
    from merlot_reserve_amd.planner import build_plan
    from merlot_reserve_amd.synthetic import make_batch, make_draws
    from oracle import ref_torch as R
    cfg = _full_size_config(case)
    B = 1
    store = ParamStore(cfg, dev, seed=0, with_optimizer=False)
    g = torch.Generator().manual_seed(2)
    tree = store.master_tree()

    def jitter(t):      # non-trivial LN / bias parameters
        return {k: jitter(v) for k, v in t.items()} if isinstance(t, dict) else (t + 0.05 * torch.randn(t.shape, generator=g) if t.dim() == 1 else t)
    store.load_tree(jitter(tree))
    batch = make_batch(cfg, B, seed=6, device=dev)
    splits, z = make_draws(cfg, B, seed=6)
    eng = PretrainEngine(cfg, B, store, dev)
    eng.forward(batch, plan=build_plan(batch, Dims(cfg, B), splits, z))
    dE = (torch.randn(eng.R, eng.d.H, generator=g) * 1e-2).to(torch.bfloat16)
    eng.dE.copy_(dE.to(dev))
    eng.backward()
    torch.cuda.synchronize()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    params = R.tree_map(lambda t: t.clone().requires_grad_(True), tree_to(store.work_tree(), torch.float32))
    osp, oz = oracle_draws(splits, z)
    preds = R.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz)
    total = 0.0
    for k, k2, name in SECTIONS:
        o, n = eng.sec[name]
        total = total + (preds[k][k2] * dE[o:o + n].float()).sum()
    total.backward()
    gt = store.grad_tree()
    leaves = [(name, t.grad if t.grad is not None else torch.zeros_like(t)) for name, t in R.tree_leaves(params)]
    del preds, total
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
        if err > 3e-2 * gn + 6e-3 * gmax or (gn > 5e-2 * gmax and cos < 0.9995):      # 3 x the errors measured on MI355X (printed below)
            bad.append((name, err, gn, cos))
    print(f'{case} backward parity: {len(leaves)} leaves, worst rel. error on a significant leaf {worst[0]:.3e} ({worst[1]}), lowest cosine {wcos:.6f}, '
          f'worst |d| / max|g| over all leaves {wabs:.3e}')
    assert not bad, bad[:10]


@pytest.mark.parametrize('B', [1, 3, 5])
def test_odd_batch_sizes_step(dev, B):
    """Record counts that are not multiples of the tile / vector sizes: one eager and two graph-replayed steps, finite
    loss equal to the oracle's on the same batch (forward tolerance)."""
    from merlot_reserve_amd.config import tiny_config
    from merlot_reserve_amd.synthetic import make_batch, make_draws
    from merlot_reserve_amd.trainer import Trainer
    from oracle import ref_torch as R
    cfg = tiny_config(seq_len=80, lang_seq_len=40)
    tr = Trainer(cfg, B, dev, seed=2)
    batch = make_batch(cfg, B, seed=30 + B, device=dev)
    draws = make_draws(cfg, B, seed=30 + B)
    params = tree_to(tr.params.work_tree(), torch.float32)
    plan = tr.plan(batch, draws)
    tr.train_step(batch, plan=plan)
    got = tr.loss_info()['loss']
    osp, oz = oracle_draws(*draws)
    with torch.no_grad():
        loss, _ = R.loss_fn_given_preds([R.pretrain_forward(params, cfg, oracle_batch(batch), osp, oz)])
    assert abs(got - float(loss)) <= 2e-2 * abs(float(loss)), (got, float(loss))
    tr.capture(batch)
    for _ in range(2):
        tr.train_step_graph(batch, plan)
    assert np.isfinite(tr.loss_info()['loss']) and tr.state.step == 3


@pytest.mark.parametrize('flags', [{'model': {'do_rotary': False}}, MULTI_SEQ], ids=['learned_pe', 'multi_seq'])
def test_config_branches_step_through_the_captured_trainer(dev, flags):
    """The two config branches built in round 5 through the whole trainer -- optimizer step, hipGraph capture, replay: bit for bit the eager steps."""
    from merlot_reserve_amd.config import tiny_config
    from merlot_reserve_amd.synthetic import make_batch
    from merlot_reserve_amd.trainer import Trainer
    cfg = tiny_config()
    for k, v in flags.items():
        cfg[k].update(v)
    cfg['optimizer'].update(num_warmup_steps=1, learning_rate=1e-3)
    B = 2
    bs = [make_batch(cfg, B, seed=10 + i, device=dev) for i in range(4)]
    tr, ref = Trainer(cfg, B, dev, seed=0), Trainer(cfg, B, dev, seed=0)
    for b in bs[:2]:
        tr.train_step(b, plan=tr.plan(b))
        ref.train_step(b, plan=ref.plan(b))
    tr.capture(bs[2])
    for b in bs[2:]:
        tr.train_step_graph(b, tr.plan(b))
        ref.train_step(b, plan=ref.plan(b))
    torch.cuda.synchronize()
    assert torch.equal(tr.params.master, ref.params.master) and tr.loss_info()['loss'] == ref.loss_info()['loss']
    assert np.isfinite(tr.loss_info()['loss']) and tr.state.step == 4


def test_tower_trains_under_an_arbitrary_attention_mask(dev):
    """Round 6 (the round-5 review's "missing" item 3): the reference differentiates TransformerEncoder under ANY boolean mask (mreserve/modeling.py:343-358).
    TowerEngine.encoder_forward / encoder_backward(dense_mask=...) on the span tower of a tiny pretrainer with a random mask that is not of the block form
    (an entirely masked row included): the tower's output and EVERY parameter gradient of the tower, and the gradient wrt its input, against autograd of
    the oracle's transformer_encoder(attention_mask=...) on the same input."""
    from merlot_reserve_amd.planner import rotary_coords_1d
    from oracle import ref_torch as R
    cfg, store, eng, batch, splits, z = _setup(dev)
    ts, prefix = eng.ts, 'span_encoder/transformer'
    g = torch.Generator().manual_seed(4)
    mask = torch.rand(ts.nseq, ts.S, ts.S, generator=g) < 0.6
    mask[:, torch.arange(ts.S), torch.arange(ts.S)] = True
    mask[1, 3, :] = False
    m8 = mask.to(torch.uint8).contiguous().to(dev)
    xin = ts.xin.clone()                                         # [nseq * S, H]: the CLS row already in place (engine.forward)
    eng.cur = eng.sc_main
    eng.encoder_forward(ts, prefix, eng.tables['span_rot'], None, dense_mask=m8)
    D = (torch.randn(ts.M, ts.H, generator=g) * 1e-2).to(torch.bfloat16).to(dev)
    store.grad.zero_()
    Din = eng.encoder_backward(ts, prefix, eng.tables['span_rot'], None, D.clone(), dense_mask=m8)
    torch.cuda.synchronize()
    # the oracle on the same rows: no CLS handling (the row is part of the input), the span tower's coordinates with the CLS position at 0
    p = R.tree_map(lambda t: t.clone().requires_grad_(True), tree_to(store.work_tree(), torch.float32)['span_encoder']['transformer'])
    x = xin.float().cpu().reshape(ts.nseq, ts.S, ts.H).clone().requires_grad_(True)
    coords = np.concatenate([np.zeros((1, 1)), rotary_coords_1d(eng.d.span_len, False)[:, None] / 16.0], 0)
    ref = R.transformer_encoder(p, x, cfg['model']['span_num_layers'], rotary_coords=coords, attention_mask=mask)['seq']
    assert relerr(ts.xf, ref.reshape(ts.M, ts.H)) < 2e-2
    (ref.reshape(ts.M, ts.H) * D.float().cpu()).sum().backward()
    assert relerr(Din, x.grad.reshape(ts.M, ts.H)) < 3e-2
    gt = store.grad_tree()['span_encoder']['transformer']
    leaves = [(n, t.grad) for n, t in R.tree_leaves(p) if t.grad is not None]
    assert len(leaves) >= 10 * cfg['model']['span_num_layers']
    gmax = max(float(gr.norm()) for _, gr in leaves)
    for name, gr in leaves:
        mine = gt
        for part in name.split('/'):
            mine = mine[part]
        err, gn = float((mine.double() - gr.double()).norm()), float(gr.norm())
        assert err <= 8e-2 * gn + 1.5e-2 * gmax, (name, err, gn)
