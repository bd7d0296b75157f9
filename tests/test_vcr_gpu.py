"""VCR finetuning step (BASELINE config 5; finetune/vcr/qa_qar_joint_finetune.py + finetune/optimization.py) on the MI355X
against the oracle: logits, loss / is_right, every parameter gradient (oracle autograd), the finetuning optimizer chain
leaf by leaf, and a short training run.  Tolerances as for the pretraining step (bf16 compute vs fp32 oracle reading the
same bf16-rounded weights and inputs): logits rel-L2 <= 2e-2, gradients |d| <= 8e-2 |g| + 1.5e-2 max|g| and cos >= 0.995."""
import numpy as np
import pytest
import torch

from merlot_reserve_amd.config import tiny_config
from oracle import ref_torch as R
from tests.util import relerr, tree_to

pytestmark = pytest.mark.gpu


def vcr_cfg(H=128):
    cfg = tiny_config(hidden_size=H, grid=(4, 6))
    cfg['data'].update(lang_seq_len=16, num_answers=4)
    cfg['optimizer'] = {'beta_2': 0.98, 'eps': 1e-6, 'learning_rate': 1e-3, 'num_train_steps': 100, 'num_warmup_steps': 2,
                        'use_bfloat16_adam': True, 'weight_decay_rate': 0.1, 'do_bias_correction': True}
    return cfg


def setup(dev, B=2, seed=0):
    from merlot_reserve_amd import finetune as F
    cfg = vcr_cfg()
    model = F.MerlotReserveVCR.from_config(cfg, device=dev, seed=seed)
    batch = F.make_vcr_batch(cfg, B, seed=seed, device=dev)
    batch['answers'][0, 1, 2, :] = 0                       # an empty sequence: no MASK, pooled at a PAD position
    batch['answers'][1, 0, 1, 5:] = 0
    batch['answers'][1, 0, 1, 2] = 3
    params = model.init_from_dummy_batch(batch)
    g = torch.Generator().manual_seed(1)                   # non-trivial biases / LN parameters; a larger Dense(1)
    def jit(t):
        return {k: jit(v) for k, v in t.items()} if isinstance(t, dict) else (t + 0.05 * torch.randn(t.shape, generator=g) if t.dim() == 1 else t)
    params = jit(params)
    params['proj']['kernel'] = torch.randn(128, 1, generator=g) * 0.3
    return F, cfg, model, batch, params


def oracle_batch(batch):
    return {'image': batch['image'].float().cpu(), 'answers': torch.from_numpy(batch['answers'].astype(np.int64)),
            'labels': torch.from_numpy(batch['labels'].astype(np.int64))}


def test_vcr_forward_loss_and_gradients(dev):
    F, cfg, model, batch, params = setup(dev)
    logits = model.apply({'params': params}, batch)
    eng, store = model.engine, model.params_store
    eng.loss_and_grad_logits()
    torch.cuda.synchronize()
    dl_loss = eng.dlogits[:, 0].float().cpu().view(2, 2, 4)
    # backward from an INJECTED dL/dlogits: at initialisation softmax is uniform, dL/dlogits = (-3/16, 1/16, 1/16, 1/16)
    # exactly, and every shared-parameter gradient is a sum of near-cancelling terms (ill-conditioned in bf16 for the
    # reference as well) -- same treatment as the contrastive loss in tests/test_pretrain_gpu.py
    g = torch.Generator().manual_seed(9)
    inj = (torch.randn(16, generator=g) * 0.2).to(torch.bfloat16)
    eng.dlogits[:, 0] = inj.to(dev)
    eng.backward()
    torch.cuda.synchronize()
    ob = oracle_batch(batch)
    wp = tree_to(store.work_tree(), torch.float32)
    wp = R.tree_map(lambda t: t.clone().requires_grad_(True), wp)
    ref = R.vcr_forward(wp, cfg, ob)
    assert logits.shape == ref.shape == (2, 2, 4)
    e = relerr(logits, ref)
    print('logits rel err', e, logits.flatten()[:4].tolist(), ref.flatten()[:4].tolist())
    assert e < 2e-2
    loss, info = R.vcr_train_loss(ref, ob['labels'])
    li = eng.loss_info()
    assert abs(li['loss'] - float(loss)) < 2e-2 * abs(float(loss)), (li, float(loss))
    assert abs(li['is_right'] - float(info['is_right'])) < 0.26          # argmax can flip on near-ties in bf16
    p_ref = torch.softmax(ref.detach(), -1)
    want_dl = (p_ref - torch.nn.functional.one_hot(ob['labels'], 4).float()) / 4.0
    assert relerr(dl_loss, want_dl) < 1e-2                                # the loss gradient (bf16 storage)
    (ref * inj.float().view(2, 2, 4)).sum().backward()
    grads = R.tree_map(lambda t: t.grad if t.grad is not None else torch.zeros_like(t), wp)
    gt = store.grad_tree()
    gmax = max(float(g.norm()) for _, g in R.tree_leaves(grads))
    bad = []
    for name, g in R.tree_leaves(grads):
        mine = gt
        for part in name.split('/'):
            mine = mine[part]
        gn, err = float(g.norm()), float((mine.double() - g.double()).norm())
        cos = float((mine.double().flatten() @ g.double().flatten()) / (mine.double().norm() * g.double().norm() + 1e-30))
        if err > 8e-2 * gn + 1.5e-2 * gmax or (gn > 5e-2 * gmax and cos < 0.995):
            bad.append((name, err, gn, cos))
    assert not bad, bad
    # the parameters the finetuning graph does not reach get exactly zero gradient (cls_proj of the ViT)
    assert float(gt['vision_encoder']['transformer']['cls_proj']['kernel'].abs().sum()) == 0.0


def test_finetune_optimizer_matches_oracle(dev):
    F, cfg, model, batch, params = setup(dev, seed=2)
    state, tx = F.construct_finetuning_train_state(cfg['optimizer'], model, params)
    store = model.params_store
    g = torch.Generator().manual_seed(5)
    for step in range(3):
        store.grad.copy_((torch.randn(store.total, generator=g) * 0.01).to(torch.bfloat16))
        before = {'p': store.master_tree(), 'mu': store._to_tree(store.mu), 'nu': store._to_tree(store.nu),
                  'g': store.grad_tree(), 'orig': store._to_tree(store.orig)}
        state.apply_gradients()
        torch.cuda.synchronize()
        after = {'p': store.master_tree(), 'mu': store._to_tree(store.mu), 'nu': store._to_tree(store.nu)}
        for name, p0 in R.tree_leaves(before['p']):
            get = lambda t: [t := t[k] for k in name.split('/')][-1]
            np_, nm, nv = R.finetune_adam_apply(p0, get(before['orig']), get(before['g']).float(), get(before['mu']), get(before['nu']), step, cfg['optimizer'])
            assert torch.allclose(get(after['p']), np_, rtol=2e-6, atol=1e-9), (name, step)
            # bf16 momentum: at most one ulp apart (the kernel contracts c1*g + b1*m into an fma)
            assert float((get(after['mu']).float() - nm.float()).abs().max()) <= 2 ** -7 * float(nm.float().abs().max()) + 1e-12, (name, step)
            # the cube-root codec may differ by one bf16 ulp where cbrt rounds differently on the two sides
            dv = (R.unsigned_bf16_decode(get(after['nu'])) - R.unsigned_bf16_decode(nv)).abs()
            assert float((dv / (R.unsigned_bf16_decode(nv).abs() + 1e-30)).max()) < 6e-3, (name, step)
    # decay mask: ndim > 1 and size > 4096 (FO:74-75): qkv bias [6, 64] and proj [128, 1] are NOT decayed, kernels are
    flags = {n: int(store.decay_flags[store.offsets[n][0] // 2048]) for n, *_ in store.specs}
    assert flags['joint_transformer/layer_00/attention_layer/qkv/bias'] == 0 and flags['proj/kernel'] == 0
    assert flags['joint_transformer/layer_00/attention_layer/qkv/kernel'] == 1 and flags['token_encoder/Embed_0/embedding'] == 1


def test_finetune_steps_reduce_loss(dev):
    """finetune_train_step end to end on one repeated batch: the loss falls and is_right reaches 1."""
    F, cfg, model, batch, params = setup(dev, seed=3)
    state, tx = F.construct_finetuning_train_state(cfg['optimizer'], model, params)
    hist = []
    for _ in range(30):
        state, info = F.finetune_train_step(state, batch, loss_fn=F.train_loss_fn, tx_fns=tx)
        hist.append(info['loss'])
    assert state.step == 30 and np.isfinite(hist).all()
    assert hist[-1] < 0.5 * hist[0], hist
    loss, info = F.train_loss_fn(state, None, batch)
    assert info['is_right'] == 1.0


def test_finetune_state_checkpoint_round_trip(dev, tmp_path):
    """FinetuneTrainState.state_dict / load_state_dict and checkpoint.save_checkpoint(rank=) (the advisor's round-5 finding: the finetuning state
    could not save its moments).  Three steps, the state dict taken, two more steps == a model with OTHER initial weights restored from that
    dict and stepped twice, bit for bit (parameters, both moments, the anchor the chain decays towards).  The file form: written by "rank 0"
    only, bf16 leaves (moments, anchor) come back exactly, fp32 parameters through the reference's fp16 file format."""
    import os
    from merlot_reserve_amd import checkpoint as C
    F, cfg, model, batch, params = setup(dev, seed=4)
    state, tx = F.construct_finetuning_train_state(cfg['optimizer'], model, params)
    for _ in range(3):
        state, _ = F.finetune_train_step(state, batch, loss_fn=F.train_loss_fn, tx_fns=tx)
    torch.cuda.synchronize()
    sd = state.state_dict()
    assert set(sd['opt_state']) == {'0', '1', '2', '3', '4'} and int(sd['opt_state']['3']['count']) == 3 and sd['step'] == 3
    skipped = C.save_checkpoint(state, str(tmp_path / 'r1'), rank=1)
    assert not os.path.exists(skipped), 'only rank 0 writes'
    fn = C.save_checkpoint(state, str(tmp_path / 'r0'), rank=0)
    assert os.path.exists(fn) and os.path.basename(fn) == 'ckpt_3'
    st = model.params_store
    at3 = {k: getattr(st, k).clone() for k in ('master', 'mu', 'nu', 'orig')}
    for _ in range(2):
        state, _ = F.finetune_train_step(state, batch, loss_fn=F.train_loss_fn, tx_fns=tx)
    torch.cuda.synchronize()
    want = {k: getattr(st, k).clone() for k in ('master', 'mu', 'nu', 'orig')}

    _, cfg2, model2, _, params2 = setup(dev, seed=9)                  # other initial weights: everything must come from the state dict
    state2, tx2 = F.construct_finetuning_train_state(cfg2['optimizer'], model2, params2)
    st2 = model2.params_store
    assert not torch.equal(st2.orig, at3['orig'])
    state2.load_state_dict(sd)
    assert state2.step == 3
    for k in at3:
        assert torch.equal(getattr(st2, k), at3[k]), k
    for _ in range(2):
        state2, _ = F.finetune_train_step(state2, batch, loss_fn=F.train_loss_fn, tx_fns=tx2)
    torch.cuda.synchronize()
    for k in want:
        assert torch.equal(getattr(st2, k), want[k]), f'{k}: the restored run must continue bit for bit'

    _, cfg3, model3, _, params3 = setup(dev, seed=10)
    state3, _ = F.construct_finetuning_train_state(cfg3['optimizer'], model3, params3)
    C.load_checkpoint(fn, state=state3)
    st3 = model3.params_store
    assert state3.step == 3
    for k in ('mu', 'nu', 'orig'):
        assert torch.equal(getattr(st3, k), at3[k]), f'{k}: bf16 leaves are stored as they are'
    assert float((st3.master - at3['master']).abs().max()) <= 2 ** -10 * float(at3['master'].abs().max()) + 1e-7      # fp16 file format


def test_graph_replay_matches_eager(dev):
    """VCRGraphStep (hipGraph replay) and finetune_train_step give the same parameters after the same batches."""
    F, cfg, model_a, batch, params = setup(dev, seed=4)
    _F, _c, model_b, _b, _p = setup(dev, seed=4)
    b2 = F.make_vcr_batch(cfg, 2, seed=11, device=dev)
    sa, _ = F.construct_finetuning_train_state(cfg['optimizer'], model_a, params)
    sb, _ = F.construct_finetuning_train_state(cfg['optimizer'], model_b, params)
    F.finetune_train_step(sa, batch)
    step = F.VCRGraphStep(sb, batch)                      # runs the first step eagerly, then captures
    for b in (b2, batch, b2):
        F.finetune_train_step(sa, b)
        step(b)
    torch.cuda.synchronize()
    assert sa.step == sb.step == 4
    pa, pb = model_a.params_store, model_b.params_store
    assert torch.equal(pa.master, pb.master) and torch.equal(pa.nu, pb.nu)
    assert model_a.engine.loss_info() == model_b.engine.loss_info()


@pytest.mark.parametrize('model_name,B', [('base', 1), ('large', 4)])
def test_vcr_full_size_forward_backward(dev, model_name, B):
    """The VCR step at the reference's shapes (base / large model -- BASELINE config 5 is the large one, checked at the benchmarked B = 4
    examples per GPU, where the dispatcher routes other GEMM kernels than at B = 1: M = 2308 -> the 128 x 128 geometry --, image grid 18x32
    -> ViT S = 577, answers [B, 2, 4, 144], joint [8B, 288]): logits and every gradient leaf (injected dL/dlogits)
    against the oracle on the host cores."""
    import os
    from merlot_reserve_amd import finetune as F
    from merlot_reserve_amd.config import load_config
    cfg = load_config(model_name)
    H = cfg['model']['hidden_size']
    cfg['model']['output_grid'] = [18, 32]
    cfg['data'].update(lang_seq_len=144, num_answers=4)
    model = F.MerlotReserveVCR.from_config(cfg, device=dev, seed=0)
    batch = F.make_vcr_batch(cfg, B, seed=0, device=dev)
    params = model.init_from_dummy_batch(batch)
    g = torch.Generator().manual_seed(1)
    params['proj']['kernel'] = torch.randn(H, 1, generator=g) * 0.3
    logits = model.apply({'params': params}, batch)
    eng, store = model.engine, model.params_store
    eng.loss_and_grad_logits()
    # Injected dL/dlogits with a non-zero mean.  The head is Dense(1), so the gradient entering the joint tower is rank one
    # (proj x inj) over 8 positions whose features are nearly identical at random initialisation: with zero-mean weights
    # every parameter gradient would be a difference of near-equal sums -- for the large model (scripts/debug_vcr_large.py)
    # already d proj = sum_i inj_i * pooled_h[i], a product of the FORWARD alone, then differs by 17 % between bf16 and fp32
    # storage of pooled_h (logits within 0.8 %), uniformly over the depth: conditioning of the test, not of the kernels.
    inj = (0.25 + 0.15 * torch.randn(8 * B, generator=g)).to(torch.bfloat16)
    eng.dlogits[:, 0] = inj.to(dev)
    eng.backward()
    torch.cuda.synchronize()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ob = oracle_batch(batch)
    wp = R.tree_map(lambda t: t.clone().requires_grad_(True), tree_to(store.work_tree(), torch.float32))
    ref = R.vcr_forward(wp, cfg, ob)
    e = relerr(logits, ref)
    assert logits.shape == (B, 2, 4) and e < 2e-2, e
    (ref * inj.float().view(B, 2, 4)).sum().backward()
    gt = store.grad_tree()
    leaves = [(n, t.grad if t.grad is not None else torch.zeros_like(t)) for n, t in R.tree_leaves(wp)]
    gmax = max(float(gr.norm()) for _, gr in leaves)
    bad = []
    for name, gr in leaves:
        mine = gt
        for part in name.split('/'):
            mine = mine[part]
        gn, err = float(gr.norm()), float((mine.double() - gr.double()).norm())
        cos = float((mine.double().flatten() @ gr.double().flatten()) / (mine.double().norm() * gr.double().norm() + 1e-30))
        if err > 8e-2 * gn + 1.5e-2 * gmax or (gn > 5e-2 * gmax and cos < 0.995):
            bad.append((name, err, gn, cos))
    print(f'VCR {model_name}-size B={B} parity: logits rel err {e:.2e}, {len(leaves)} gradient leaves checked')
    assert not bad, bad[:10]


def test_captured_step_with_the_rccl_communicator(dev):
    """The multi-GPU VCR step -- four gradient buckets (nan_to_num -> all-reduce(mean) -> optimizer on the range) overlapped with the
    vision tower's backward, finetune/optimization.py:106-191 -- captured with its collectives into ONE hipGraph through the library's
    RCCL communicator (one rank: RCCL refuses two ranks on one device).  Same parameters as the plain single-rank step, bit for bit
    (a one-rank mean is the identity), same bucket order eager and captured."""
    from merlot_reserve_amd.dist import NativeComm
    F, cfg, model_a, batch, params = setup(dev, seed=6)
    _F, _c, model_b, _b, _p = setup(dev, seed=6)
    comm = NativeComm()
    model_b.comm = comm
    b2 = F.make_vcr_batch(cfg, 2, seed=12, device=dev)
    sa, _ = F.construct_finetuning_train_state(cfg['optimizer'], model_a, params)
    sb, _ = F.construct_finetuning_train_state(cfg['optimizer'], model_b, params)
    F.finetune_train_step(sa, batch)
    step = F.VCRGraphStep(sb, batch)                      # eager step with the communicator, then the capture
    eager_order = list(model_b.engine.bucket_log)
    for b in (b2, batch, b2):
        F.finetune_train_step(sa, b)
        step(b)
    torch.cuda.synchronize()
    assert sa.step == sb.step == 4
    assert len(eager_order) == len(model_b.engine.gradient_buckets()[0]) >= 2 and eager_order[0] == 'joint' and eager_order[-1] == 'vision_end'
    from merlot_reserve_amd.config import load_config
    big = F.VCRDims(load_config('large'), 4)            # the stock models: 'joint', two cuts inside the vision tower, 'vision_end'
    assert sorted({l for l in (big.Lv - big.Lv // 3, big.Lv - 2 * (big.Lv // 3)) if 0 < l < big.Lv}) == [8, 16]
    assert model_a.engine.bucket_log == eager_order
    pa, pb = model_a.params_store, model_b.params_store
    assert torch.equal(pa.master, pb.master) and torch.equal(pa.nu, pb.nu) and torch.equal(pa.mu, pb.mu)
    assert model_a.engine.loss_info() == model_b.engine.loss_info()
    comm.close()


def test_scan_minibatch_sums_per_example_gradients(dev):
    """finetune_train_step(scan_minibatch=True), finetune/optimization.py:125-146: the examples one at a time, their bf16 gradients
    SUMMED (B x the full-batch mean gradient), the metrics averaged."""
    F, cfg, model, batch, params = setup(dev, seed=7)
    state, _ = F.construct_finetuning_train_state(cfg['optimizer'], model, params)
    eng, p = model._ensure(batch), model.params_store
    model._load({'params': params})
    eng.forward(batch)
    eng.loss_and_grad_logits()
    eng.backward()
    torch.cuda.synchronize()
    full, full_info = p.grad.float().clone(), eng.loss_info()
    state, info = F.finetune_train_step(state, batch, scan_minibatch=True)
    torch.cuda.synchronize()
    summed = p.grad.float()
    B = batch['image'].shape[0]
    assert state.step == 1 and abs(info['loss'] - full_info['loss']) < 2e-3 * abs(full_info['loss']) and info['is_right'] == full_info['is_right']
    assert relerr(summed, B * full) < 3e-2
