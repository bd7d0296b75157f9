"""The reference's own call sequence (pretrain/train.py:91-117) against the name-for-name API of
merlot_reserve_amd.pretrain_model, checked against the oracle: from_config -> init_from_dummy_batch ->
construct_train_state -> apply / loss_fn_given_preds -> train_step."""
import pytest
import torch

from tests.util import oracle_batch, oracle_draws, relerr, tree_to

pytestmark = pytest.mark.gpu


def test_train_py_call_sequence(dev):
    from merlot_reserve_amd.config import tiny_config
    from merlot_reserve_amd.pretrain_model import (MerlotReservePretrainer, construct_train_state, loss_fn_given_preds,
                                                   train_step)
    from merlot_reserve_amd.synthetic import make_batch, make_draws
    from oracle import ref_torch as R
    config = tiny_config()
    config['optimizer'].update(num_warmup_steps=1, learning_rate=1e-3)
    B = 2
    dummy_batch = make_batch(config, B, seed=5, device=dev)

    model = MerlotReservePretrainer.from_config(config, device=dev)                  # train.py:91
    params = model.init_from_dummy_batch(dummy_batch)                                # train.py:99
    state = construct_train_state(opt_config=config['optimizer'], model=model, params=params)    # train.py:100
    assert state.step == 0 and set(params) >= {'vision_encoder', 'audio_encoder', 'token_encoder', 'span_encoder',
                                               'joint_transformer', 'head', 'contrastive_scales'}

    # forward + loss through the reference's two functions, against the oracle on the same draws
    splits, z = make_draws(config, B, seed=11)
    preds = state.apply_fn({'params': params}, dummy_batch, split_from_here=splits, gumbel_z=z)
    loss, loss_info = loss_fn_given_preds(preds)
    bf16_params = tree_to(tree_to(params, torch.bfloat16), torch.float32)            # f32_to_bf16 at pretrain_model.py:323
    osp, oz = oracle_draws(splits, z)
    with torch.no_grad():
        opreds = R.pretrain_forward(bf16_params, config, oracle_batch(dummy_batch), osp, oz)
        oloss, oinfo = R.loss_fn_given_preds([opreds])
    assert set(preds) == set(opreds) == {'imgs_to_audio', 'text_to_audio', 'stuff_to_span'}
    for k in preds:
        for k2 in ('x', 'y'):
            assert relerr(preds[k][k2], opreds[k][k2]) <= 2e-2, (k, k2)
    assert set(loss_info) == set(oinfo)
    assert abs(loss - float(oloss)) <= 2e-2 * abs(float(oloss))

    # the mask-LM special case of the loss (pretrain_model.py:265-274): a caller-supplied 'text_preds' entry adds the masked token cross-entropy under the key
    # 'audio2text' -- against the oracle's restatement and against autograd for d loss / d logits
    gtp = torch.Generator().manual_seed(3)
    n, V = 37, 32768
    logits = (torch.randn(n, V, generator=gtp) * 3).to(dev)
    labels = torch.randint(1, V, (n,), generator=gtp)
    labels[::5] = 0                                                # masked rows
    preds2 = state.apply_fn({'params': params}, dummy_batch, split_from_here=splits, gumbel_z=z)
    tp = {'logits': logits, 'labels': labels}
    preds2['text_preds'] = tp                                      # (popped by the loss function, like the reference's preds.pop)
    loss2, info2 = loss_fn_given_preds(preds2)
    lg = logits.detach().cpu().double().requires_grad_(True)
    oloss2, oinfo2 = R.loss_fn_given_preds([dict(opreds, text_preds={'logits': lg, 'labels': labels})])
    assert set(info2) == set(oinfo2) == set(oinfo) | {'audio2text'}
    assert abs(info2['audio2text'] - float(oinfo2['audio2text'].detach())) <= 1e-5 * abs(float(oinfo2['audio2text'].detach()))
    assert abs(loss2 - (loss + info2['audio2text'])) <= 1e-6 * abs(loss2)
    oinfo2['audio2text'].backward()
    assert 'text_preds' not in preds2 and relerr(tp['dlogits'], lg.grad) <= 1e-5
    assert float(tp['dlogits'][::5].abs().max()) == 0.0

    # two optimizer steps through train_step; step 0 has schedule 0 (first update is zero, optimization.py:117-137)
    before = state.params
    state, info0 = train_step(state, dummy_batch)
    assert state.step == 1 and relerr(state.params['head']['kernel'], before['head']['kernel']) == 0.0
    state, info1 = train_step(state, make_batch(config, B, seed=6, device=dev))
    assert state.step == 2 and relerr(state.params['head']['kernel'], before['head']['kernel']) > 0.0
    assert set(info1) == set(oinfo) and all(v == v for v in info1.values())

    # the reference's error behaviour at this boundary
    with pytest.raises(ValueError):
        MerlotReservePretrainer.from_config({'model': {}})
    with pytest.raises(ValueError):
        model.apply({'params': params}, make_batch(config, B + 1, seed=7, device=dev))
    with pytest.raises(TypeError):
        loss_fn_given_preds({'imgs_to_audio': {}})
