"""Independent cross-checks of the oracle's restatements of the third-party layers the reference calls (flax 0.3.4 /
jax, absent here) against PyTorch's own implementations of the same published semantics.  The reference holds no
golden vectors for this path ("parity unpinned"); these checks at least tie the oracle's
LayerNorm / attention / multi-head attention pooling / strided Conv / Embed / log-softmax pieces to a second,
unrelated implementation."""
import math

import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_torch as R

torch.manual_seed(0)
D = torch.float64


def test_layer_norm_matches_torch():
    x = torch.randn(5, 7, 96, dtype=D) * 3 + 1.5
    p = {'scale': torch.randn(96, dtype=D), 'bias': torch.randn(96, dtype=D)}
    assert torch.allclose(R.layer_norm(x, p), F.layer_norm(x, (96,), p['scale'], p['bias'], 1e-5), atol=1e-10)


def test_attention_weights_match_sdpa():
    q, k, v = (torch.randn(3, 11, 4, 64, dtype=D) for _ in range(3))       # [B, L, heads, 64] (flax layout)
    mask = torch.rand(3, 1, 11, 11) > 0.3
    mask[..., 0] = True
    bias = torch.where(mask, 0.0, -1e10).to(D)
    w = R.dot_product_attention_weights(q, k, bias)
    got = torch.einsum('bhqk,bkhd->bqhd', w, v)
    ref = F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), attn_mask=mask).transpose(1, 2)
    assert torch.allclose(got, ref, atol=1e-9)


def test_multihead_attnpool_matches_torch_multiheadattention():
    """flax nn.MultiHeadDotProductAttention (q / k / v / out DenseGeneral with biases, M:424-426) vs torch.nn.MultiheadAttention."""
    H, nh, G, Rk = 128, 2, 6, 4
    p = {n: {'kernel': torch.randn(H, nh, 64, dtype=D) * 0.1, 'bias': torch.randn(nh, 64, dtype=D) * 0.1} for n in ('query', 'key', 'value')}
    p['out'] = {'kernel': torch.randn(nh, 64, H, dtype=D) * 0.1, 'bias': torch.randn(H, dtype=D) * 0.1}
    kv = torch.randn(G, Rk, H, dtype=D)
    q = kv.mean(-2, keepdim=True)
    got = R.multihead_attnpool(p, q, kv)
    mha = torch.nn.MultiheadAttention(H, nh, batch_first=True, dtype=D)
    with torch.no_grad():
        mha.in_proj_weight.copy_(torch.cat([p[n]['kernel'].reshape(H, H).T for n in ('query', 'key', 'value')], 0))
        mha.in_proj_bias.copy_(torch.cat([p[n]['bias'].reshape(H) for n in ('query', 'key', 'value')], 0))
        mha.out_proj.weight.copy_(p['out']['kernel'].reshape(H, H).T)
        mha.out_proj.bias.copy_(p['out']['bias'])
        ref, _ = mha(q, kv, kv, need_weights=False)
    assert torch.allclose(got, ref, atol=1e-9)


def test_audio_conv_is_strided_conv1d():
    """nn.Conv(features=H, kernel_size=[2], strides=[2], padding='SAME') on [N, 60, 65] (M:453-454) vs F.conv1d."""
    N, Hh = 3, 32
    x = torch.randn(N, 60, 65, dtype=D)
    k = torch.randn(2, 65, Hh, dtype=D)             # flax kernel [k, in, out]
    b = torch.randn(Hh, dtype=D)
    got = x.reshape(N, 30, 130) @ k.reshape(130, Hh) + b       # what audio_transformer does
    ref = F.conv1d(x.transpose(1, 2), k.permute(2, 1, 0), b, stride=2).transpose(1, 2)     # SAME pad = 0 for even length
    assert torch.allclose(got, ref, atol=1e-10)


def test_gelu_embed_unitnorm_onehot():
    x = torch.randn(50, dtype=D)
    assert torch.allclose(R.my_gelu(x), x * torch.sigmoid(1.702 * x))
    assert abs(float(R.my_gelu(torch.tensor(1.0, dtype=D))) - 0.84580) < 1e-5
    emb = torch.randn(100, 8, dtype=D)
    ids = torch.randint(0, 100, (4, 5))
    assert torch.equal(R.token_embedder({'Embed_0': {'embedding': emb}}, {'k': ids})['k'], F.embedding(ids, emb))
    v = torch.randn(6, 16, dtype=D)
    n = R.unit_normalize(v)
    assert torch.allclose(n, v / torch.sqrt((v * v).sum(-1, keepdim=True) + 1e-5))
    assert float(R.unit_normalize(torch.zeros(1, 4, dtype=D)).abs().sum()) == 0.0
    # one_hot_pool == index_add of the selected rows
    vals = torch.randn(2, 9, 5, dtype=D)
    idx = torch.randint(-1, 4, (2, 9))
    do = torch.rand(2, 9) > 0.4
    got = R.one_hot_pool(do, idx, vals, 4)['x']
    ref = torch.zeros(2, 4, 5, dtype=D)
    for b in range(2):
        for l in range(9):
            if do[b, l] and idx[b, l] >= 0:
                ref[b, idx[b, l]] += vals[b, l]
    assert torch.allclose(got, ref)


def test_contrastive_loss_is_symmetric_cross_entropy():
    """loss_fn_given_preds (P:262-303) on one device == mean over the two directions of cross-entropy with diagonal targets."""
    H = 16
    def unit(n):
        t = torch.randn(n, H, dtype=D)
        return t / t.norm(dim=-1, keepdim=True) * 3.0
    preds = {'imgs_to_audio': {'x': unit(6), 'y': unit(6)}, 'text_to_audio': {'x': unit(4), 'y': unit(4), 'y_extra': unit(5)},
             'stuff_to_span': {'x': unit(8), 'y': unit(8), '_sources': torch.tensor([0, 1, 2, 0, 1, 2, 0, 1])}}
    loss, info = R.loss_fn_given_preds([preds])
    want = 0.0
    for k, v in preds.items():
        x, y = v['x'], v['y']
        y_all = torch.cat([y, v['y_extra']], 0) if 'y_extra' in v else y
        tgt = torch.arange(x.shape[0])
        ce_xy = F.cross_entropy(x @ y_all.T, tgt)
        ce_yx = F.cross_entropy(y @ x.T, tgt)
        assert abs(float(info[k]) - float((ce_xy + ce_yx) / 2)) < 1e-9, k
        want = want + (ce_xy + ce_yx) / 2
    assert abs(float(loss) - float(want)) < 1e-9
