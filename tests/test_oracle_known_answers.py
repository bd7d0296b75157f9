"""Known-answer tests that pin the oracle (and the product's host formulas) to hand-derived values of the reference's
formulas.  The reference ships no tests or golden vectors: these are derived from the cited lines."""
import math

import numpy as np
import torch

from merlot_reserve_amd import planner
from merlot_reserve_amd.trainer import lr_scale_linearwarmup_cosinedecay as product_schedule
from oracle import ref_torch as R


def test_rotary_coordinates():
    # mreserve/modeling.py:21-35
    assert R.get_rotary_coordinates(4).tolist() == [-2, -1, 1, 2]
    assert R.get_rotary_coordinates(5).tolist() == [-2, -1, 1, 2, 3]
    assert R.get_rotary_coordinates(3, center_origin=False).tolist() == [1, 2, 3]
    c = R.get_rotary_coordinates_2d(12, 20)                       # :38-50: base_scale = 1/21, h first
    assert c.shape == (240, 2)
    assert np.allclose(c[0], [-6 / 21, -10 / 21]) and np.allclose(c[1], [-6 / 21, -9 / 21]) and np.allclose(c[20], [-5 / 21, -10 / 21])
    assert np.array_equal(planner.rotary_coords_2d(12, 20), c)
    assert np.array_equal(planner.rotary_coords_1d(7), R.get_rotary_coordinates(7))


def test_sinusoid_frequencies_and_rotary_identity():
    # :97: freqs = 2^linspace(0, log2(5), d): 1-D -> 5^(k/15), 2-D -> 5^(k/7), 4-D -> 5^(k/3)
    for nd, d in ((1, 16), (2, 8), (4, 4)):
        coords = np.zeros((1, nd)); coords[0, 0] = 1.0
        s = R.construct_rotary_sinusoids(coords)                  # [2, 1, 32]
        theta = np.arccos(np.clip(s[0, 0, 0:2 * d:2], -1, 1))     # cos(pi * f_k) for the first axis
        f = 5.0 ** (np.arange(d) / (d - 1))
        assert np.allclose(np.cos(np.pi * f), s[0, 0, 0:2 * d:2])
        assert np.allclose(s[0, 0, 0::2], s[0, 0, 1::2])          # pairs repeated (:112)
    # :133-142 quirk: theta = 0 -> [-x0, x1, -x2, x3, ...]; theta = pi/2 -> x
    x = torch.arange(1.0, 65.0, dtype=torch.float64).reshape(1, 1, 64)
    zero = torch.as_tensor(R.construct_rotary_sinusoids(np.zeros((1, 1))))
    out = R.apply_rotary(x, zero)
    exp = x.clone(); exp[..., 0:32:2] *= -1
    assert torch.equal(out, exp)
    half = torch.zeros(2, 1, 32, dtype=torch.float64); half[1] = 1.0   # cos = 0, sin = 1
    assert torch.equal(R.apply_rotary(x, half), x)
    # the product's diagonal table reproduces apply_rotary for arbitrary angles
    coords = np.random.default_rng(0).uniform(-1, 1, size=(5, 2))
    tab = planner.rot_scale_table(coords)
    q = torch.randn(5, 3, 64, dtype=torch.float64)
    ref = R.apply_rotary(q, torch.as_tensor(R.construct_rotary_sinusoids(coords)))
    mine = q.clone(); mine[..., :32] *= torch.as_tensor(tab, dtype=torch.float64)[:, None, :]
    assert torch.allclose(ref, mine, atol=1e-6)


def test_gelu_layernorm_unitnorm():
    assert abs(float(R.my_gelu(torch.tensor(1.0))) - 0.84580) < 1e-5               # :240-241
    p = {'scale': torch.ones(4), 'bias': torch.zeros(4)}
    y = R.layer_norm(torch.tensor([1.0, 2.0, 3.0, 4.0]), p)
    assert torch.allclose(y, torch.tensor([-1.341635, -0.447212, 0.447212, 1.341635]), atol=1e-5)
    assert torch.equal(R.unit_normalize(torch.zeros(2, 8)), torch.zeros(2, 8))      # :570-578
    v = torch.tensor([[3.0, 4.0]])
    assert torch.allclose(R.unit_normalize(v), v / math.sqrt(25 + 1e-5))


def test_attention_mask_semantics():
    # :353-356: -1e10 bias -> weight exactly 0 in fp32 on masked keys; fully masked row -> uniform
    q = torch.randn(1, 3, 1, 64); k = torch.randn(1, 3, 1, 64)
    bias = torch.tensor([[[[0.0, -1e10, 0.0], [0.0, 0.0, 0.0], [-1e10, -1e10, -1e10]]]])
    w = R.dot_product_attention_weights(q, k, bias)
    assert w[0, 0, 0, 1] == 0.0 and torch.allclose(w[0, 0, 2], torch.full((3,), 1 / 3))


def test_one_hot_pool():
    # :541-567: idx = -1 or do_pool False contributes nothing; real_bsize merges rows
    v = torch.arange(24.0).reshape(2, 3, 4)
    do_pool = torch.tensor([[True, False, True], [True, True, False]])
    idx = torch.tensor([[1, 0, -1], [0, 1, 2]])
    out = R.one_hot_pool(do_pool, idx, v, 3)
    assert torch.equal(out['x'][0], torch.stack([torch.zeros(4), v[0, 0], torch.zeros(4)]))
    assert torch.equal(out['x'][1], torch.stack([v[1, 0], v[1, 1], torch.zeros(4)]))
    merged = R.one_hot_pool(do_pool, idx, v, 3, real_bsize=1)
    assert torch.equal(merged['x'][0], torch.stack([v[1, 0], v[0, 0] + v[1, 1], torch.zeros(4)]))
    assert merged['idx_oh'].sum(1).tolist() == [[1.0, 2.0, 0.0]]


def test_schedule_known_values():
    # pretrain/optimization.py:117-137 with base.yaml's 3750 warm-up of 750000 and the 0.02 default of :187
    s = lambda t: float(R.lr_scale_linearwarmup_cosinedecay(t, 3750, 750000, 0.02))
    assert s(0) == 0.0                                       # count starts at 0: first update is zero
    assert abs(s(1875) - 0.5) < 1e-7
    assert abs(s(3750) - 1.0) < 1e-7
    assert abs(s(750000) - 0.02) < 1e-4
    mid = 3750 + (750000 - 3750 + 1) / 2
    assert abs(s(int(mid)) - 0.51) < 1e-3
    for t in (0, 1, 100, 3749, 3750, 3751, 100000, 750000, 800000):
        assert abs(product_schedule(t, 3750, 750000, 0.02) - s(t)) < 1e-7
    assert float(R.lr_scale_linearwarmup_lineardecay(10, 10, 110)) == 1.0


def test_cube_root_codec():
    # pretrain/optimization.py:36-51
    v = torch.logspace(-12, 2, 2000)
    dec = R.unsigned_bf16_decode(R.unsigned_bf16_encode(v))
    assert float(((dec - v).abs() / v).max()) < 2.0 ** -9 * 1.01 / 1.0          # 2^-9 on v^3 -> ~2^-9/3*... on v: bound
    assert float(((dec - v).abs() / v).max()) < 1.4e-3
    z = R.unsigned_bf16_encode(torch.zeros(1))
    assert z.dtype == torch.bfloat16 and torch.signbit(z).item()                 # enc(0) = -0.0 (strict '<' at :51)
    assert float(R.unsigned_bf16_decode(z)) == 0.0                               # (-0.0 >= 0) is True at :40
    # the sign bit carries the extra half-ulp: both signs occur
    enc = R.unsigned_bf16_encode(v)
    assert torch.signbit(enc).any() and (~torch.signbit(enc)).any()


def test_adam_first_step_is_zero_update():
    cfg = dict(learning_rate=4e-4, num_train_steps=750000, num_warmup_steps=3750, weight_decay_rate=0.1, beta_2=0.98, eps=1e-6)
    p = torch.randn(4, 4)
    g = torch.randn(4, 4)
    z = torch.zeros(4, 4, dtype=torch.bfloat16)
    newp, mu, nu = R.adam_bf16_apply(p, g, z, z, 0, cfg)
    assert torch.equal(newp, p)
    assert torch.allclose(mu.float(), (0.1 * g).to(torch.bfloat16).float())
    newp2, _, _ = R.adam_bf16_apply(p, g, mu, nu, 1, cfg)
    assert not torch.equal(newp2, p)
