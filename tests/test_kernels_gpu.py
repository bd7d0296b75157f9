"""Per-kernel numerics on the MI355X: each C-ABI entry point against a plain PyTorch fp32 reference of the same op
(inputs are bf16-representable, so the reference sees exactly the kernel's inputs).  Tolerances are stated per test:
outputs are bf16 (rel. precision 2^-8 = 3.9e-3), accumulation fp32."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def rnd(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF16).to(dev)


def relerr(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def assert_close(got, ref, rel, name=''):
    e = relerr(got, ref)
    assert math.isfinite(e) and e <= rel, f'{name}: rel-L2 err {e:.3e} > {rel:.1e}'
    # elementwise: bf16 output rounding (4e-3 relative) plus accumulated noise relative to the tensor's scale
    tol = 8e-3 * ref.float().abs() + 4 * rel * ref.float().abs().mean() + 1e-6
    bad = ((got.float() - ref.float()).abs() > tol).float().mean().item()
    assert bad < 1e-3, f'{name}: {bad * 100:.3f}% of elements outside tolerance'


GEMM_CASES = [
    # M, N, K, transA, transB
    (128, 128, 64, False, False),
    (300, 264, 200, False, False),   # ragged in every dim
    (241 * 4, 384, 128, False, False),
    (300, 264, 200, False, True),
    (304, 264, 200, True, False),
    (304, 264, 200, True, True),
    (128, 3072, 768, False, False),
    (1000, 768, 3072, False, True),
    (768, 3072, 1000, True, False),
    (64, 8, 136, False, False),      # tiny N, K = padded audio patch
]


@pytest.mark.parametrize('M,N,K,ta,tb', GEMM_CASES)
def test_gemm_plain(dev, M, N, K, ta, tb):
    from merlot_reserve_amd import ops
    a = rnd((K, M) if ta else (M, K), dev, seed=1)
    b = rnd((N, K) if tb else (K, N), dev, seed=2)
    out = torch.full((M, N), float('nan'), dtype=BF16, device=dev)
    ops.gemm(a, b, out, transA=ta, transB=tb)
    A = a.float().T if ta else a.float()
    B = b.float().T if tb else b.float()
    assert_close(out, A @ B, 3e-3, f'gemm {M}x{N}x{K} ta={ta} tb={tb}')


def test_gemm_asymmetric_identity(dev):
    """A = I with an asymmetric B catches a transposed C write."""
    from merlot_reserve_amd import ops
    n = 128
    a = torch.eye(n, dtype=BF16, device=dev)
    b = (torch.arange(n * n, device=dev).reshape(n, n) % 251).to(BF16)
    out = torch.zeros(n, n, dtype=BF16, device=dev)
    for ta in (False, True):
        for tb in (False, True):
            ops.gemm(a, b.T.contiguous() if tb else b, out, transA=ta, transB=tb)
            assert torch.equal(out, b), f'ta={ta} tb={tb}'


def test_gemm_f32_out_and_bias(dev):
    from merlot_reserve_amd import ops
    M, N, K = 100, 52, 768
    a, b = rnd((M, K), dev, seed=3), rnd((N, K), dev, seed=4)
    bias = rnd((N,), dev, seed=5)
    out = torch.zeros(M, N, dtype=F32, device=dev)
    ops.gemm(a, b, out, transB=True, bias=bias)
    ref = a.float() @ b.float().T + bias.float()
    assert relerr(out, ref) < 1e-5


def test_gemm_epilogues(dev):
    from merlot_reserve_amd import ops
    M, N, K = 482, 384, 128           # 2 sequences of 241, 2 heads x 3 x 64
    a, w = rnd((M, K), dev, seed=6), rnd((K, N), dev, scale=0.1, seed=7)
    bias = rnd((N,), dev, seed=8)
    tab = torch.rand(241, 32, device=dev) * 2 - 1
    # qkv-style: bias + rotary scale on the first 256 columns (q and k heads)
    out = torch.zeros(M, N, dtype=BF16, device=dev)
    ops.gemm(a, w, out, bias=bias, rot_tab=tab, rot_cols=256)
    ref = a.float() @ w.float() + bias.float()
    scale = torch.ones(M, N, device=dev)
    rows = torch.arange(M, device=dev) % 241
    for h in range(4):
        scale[:, h * 64:h * 64 + 32] = tab[rows]
    assert_close(out, ref * scale, 3e-3, 'rot epilogue')
    # mlp-in style: bias + gelu, pre-activation copy
    pre = torch.zeros(M, N, dtype=BF16, device=dev)
    ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU, c2=pre)
    sg = torch.sigmoid(1.702 * ref)
    assert_close(pre, sg + 1.702 * ref * sg * (1 - sg), 4e-3, 'c2 = gelu grad')
    assert_close(out, ref * sg, 4e-3, 'gelu')
    # residual
    res = rnd((M, N), dev, seed=9)
    ops.gemm(a, w, out, residual=res)
    assert_close(out, (a.float() @ w.float()).to(BF16).float() + res.float(), 3e-3, 'residual')
    # in-place residual (C == residual)
    buf = res.clone()
    ops.gemm(a, w, buf, residual=buf)
    assert torch.equal(buf, out)
    # gelu' multiply
    aux = rnd((M, N), dev, seed=10)
    ops.gemm(a, w, out, aux=aux)
    assert_close(out, (a.float() @ w.float()).to(BF16).float() * aux.float(), 4e-3, 'aux multiply')
    ops.gemm(a, w, pre, bias=bias, c2=out)
    assert_close(out, ref, 3e-3, 'c2 without activation')
    # row map: groups of 240 rows land after a CLS row
    M2 = 480
    a2 = rnd((M2, K), dev, seed=11)
    big = torch.zeros(2 * 241, N, dtype=BF16, device=dev)
    ops.gemm(a2, w, big, bias=bias, row_map=(240, 241, 1))
    ref2 = (a2.float() @ w.float() + bias.float())
    got = big.reshape(2, 241, N)
    assert torch.all(got[:, 0] == 0)
    assert_close(got[:, 1:].reshape(M2, N), ref2, 3e-3, 'row map')


@pytest.fixture
def ln_impl():
    from merlot_reserve_amd import ops

    def use(v):
        ops.set_option('ln_impl', v)
    yield use
    ops.set_option('ln_impl', 1)


@pytest.mark.parametrize('impl', [1, 0], ids=['round6', 'round5'])
@pytest.mark.parametrize('rows,H', [(5, 128), (1000, 768), (333, 1024), (64, 2048), (9000, 768)])
def test_layernorm(dev, rows, H, impl, ln_impl):
    """(impl: option ln_impl -- the round-6 backward kernel, whole grid resident, and round 5's, which H > 1024 still takes; 9000 rows: more rows than
    resident waves, so the kernels' row loops and prefetch run.)"""
    from merlot_reserve_amd import ops
    ln_impl(impl)
    x = rnd((rows, H), dev, scale=2.0, seed=1) + 0.5
    gamma, beta = rnd((H,), dev, seed=2) + 1, rnd((H,), dev, seed=3)
    y = torch.zeros_like(x)
    mean = torch.zeros(rows, device=dev)
    rstd = torch.zeros(rows, device=dev)
    ops.layernorm_fwd(x, gamma, beta, y, mean, rstd)
    xf = x.float()
    mu = xf.mean(-1, keepdim=True)
    var = (xf * xf).mean(-1, keepdim=True) - mu * mu
    ref = (xf - mu) * (torch.rsqrt(var + 1e-5) * gamma.float()) + beta.float()
    assert_close(y, ref, 3e-3, 'ln fwd')
    assert relerr(mean, mu[:, 0]) < 1e-5 and relerr(rstd, torch.rsqrt(var + 1e-5)[:, 0]) < 1e-5
    # backward vs autograd
    dy = rnd((rows, H), dev, seed=4)
    xr = xf.clone().requires_grad_(True)
    gr = gamma.float().clone().requires_grad_(True)
    br = beta.float().clone().requires_grad_(True)
    mu2 = xr.mean(-1, keepdim=True)
    var2 = (xr * xr).mean(-1, keepdim=True) - mu2 * mu2
    ((xr - mu2) * (torch.rsqrt(var2 + 1e-5) * gr) + br).backward(dy.float())
    dx = torch.zeros_like(x)
    dg, db = torch.zeros(H, dtype=BF16, device=dev), torch.zeros(H, dtype=BF16, device=dev)
    ws = ops.layernorm_bwd_workspace(H, dev)
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg, db, ws)
    assert_close(dx, xr.grad, 4e-3, 'ln dx')
    assert_close(dg, gr.grad, 4e-3, 'ln dgamma')
    assert_close(db, br.grad, 4e-3, 'ln dbeta')
    # accumulate form
    dx2 = dy.clone()
    ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx2, dg, db, ws, add_to_dx=True)
    assert_close(dx2, xr.grad + dy.float(), 4e-3, 'ln dx accumulate')


def test_colsum(dev):
    from merlot_reserve_amd import ops
    x = rnd((1234, 776), dev, seed=5)
    out = torch.zeros(776, dtype=BF16, device=dev)
    ops.colsum(x, out, ops.colsum_workspace(776, dev))
    assert_close(out, x.float().sum(0), 4e-3, 'colsum')


def test_deferred_reductions_in_one_launch(dev):
    """LayerNorm dgamma/dbeta and two column sums deferred into ONE mr_reduce_partials launch == the immediate forms."""
    from merlot_reserve_amd import ops
    rows, H = 5000, 256
    x = rnd((rows, H), dev, scale=2.0, seed=1) + 0.5
    gamma, dy = rnd((H,), dev, seed=2) + 1, rnd((rows, H), dev, seed=4)
    y, mean, rstd = torch.zeros_like(x), torch.zeros(rows, device=dev), torch.zeros(rows, device=dev)
    ops.layernorm_fwd(x, gamma, gamma, y, mean, rstd)
    a, b = rnd((777, 776), dev, seed=5), rnd((130, 3072), dev, seed=6)
    outs = []
    for deferred in (False, True):
        jobs = [] if deferred else None
        dx, dg, db = torch.zeros_like(x), torch.zeros(H, dtype=BF16, device=dev), torch.zeros(H, dtype=BF16, device=dev)
        ca, cb = torch.zeros(776, dtype=BF16, device=dev), torch.zeros(3072, dtype=BF16, device=dev)
        ws = [ops.layernorm_bwd_workspace(H, dev), ops.colsum_workspace(776, dev), ops.colsum_workspace(3072, dev)]   # kept alive
        ops.layernorm_bwd(dy, x, gamma, mean, rstd, dx, dg, db, ws[0], jobs=jobs)
        ops.colsum(a, ca, ws[1], jobs=jobs)
        ops.colsum(b, cb, ws[2], jobs=jobs)
        if deferred:
            assert len(jobs) == 3
            ops.reduce_partials(jobs)
            assert jobs == []
        outs.append((dg, db, ca, cb))
    assert_close(outs[1][2], a.float().sum(0), 4e-3, 'deferred colsum')
    for u, v in zip(*outs):
        assert relerr(u, v) < 4e-3      # same partial rows, different (both fixed) summation trees


def ref_attention(qkv, code, nseq, S, nh):
    H = nh * 64
    q, k, v = qkv.float().reshape(nseq, S, 3, nh, 64).unbind(2)
    s = torch.einsum('nqhd,nkhd->nhqk', q / 8.0, k)
    if code is not None:
        c = code.reshape(nseq, S)
        allowed = (c[:, :, None] == c[:, None, :]) & (c[:, :, None] >= 0)
        s = s + torch.where(allowed, 0.0, -1e10)[:, None]
    p = torch.softmax(s, -1)
    o = torch.einsum('nhqk,nkhd->nqhd', p, v).reshape(nseq * S, H)
    return o, torch.logsumexp(s, -1)


ATTN_CASES = [(3, 241, 2, False), (5, 31, 2, False), (2, 640, 3, True), (7, 16, 2, True), (2, 64, 1, False),
              (1, 130, 12, True),
              # resolution-adaptation shapes (pretrain/train_fixres.py:78-90) at the large model's 16 heads: joint S = 1312
              # (masked) and ViT S = 577 -- the regime where the reference recomputes attention (modeling.py:202,231)
              (2, 1312, 16, True), (2, 577, 16, False)]


# (case, attn_onepass): the library's own choice for every case (-1: the one-pass backward for 128 < S <= 256), plus both backward
# paths forced -- the dQ + dK / dV kernel pair (0) and the one-pass kernel (1) -- wherever one workgroup's 256 keys cover the sequence
ATTN_RUNS = [(c, -1) for c in ATTN_CASES] + [(c, v) for c in ATTN_CASES + [(2, 256, 2, False), (2, 200, 3, True), (3, 129, 2, False)]
                                             if c[1] <= 256 for v in (0, 1)]


@pytest.fixture
def attn_path():
    from merlot_reserve_amd import ops

    def use(v):
        ops.set_option('attn_onepass', v)
    yield use
    ops.set_option('attn_onepass', -1)


@pytest.mark.parametrize('case,onepass', ATTN_RUNS)
def test_attention_fwd_bwd(dev, case, onepass, attn_path):
    from merlot_reserve_amd import ops
    nseq, S, nh, masked = case
    attn_path(onepass)
    H = nh * 64
    qkv = rnd((nseq * S, 3 * H), dev, seed=1)
    code = None
    if masked:
        g = torch.Generator().manual_seed(7)
        c = torch.randint(0, 2, (nseq, S), generator=g)
        pad = torch.rand(nseq, S, generator=g) < 0.2
        c[pad] = -1
        c[:, 0] = 0
        code = c.to(torch.int32).reshape(-1).to(dev)
    out = torch.zeros(nseq * S, H, dtype=BF16, device=dev)
    lse = torch.zeros(nseq, nh, S, device=dev)
    ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    qr = qkv.float().clone().requires_grad_(True)
    ref_o, ref_lse = ref_attention(qr, code, nseq, S, nh)
    assert_close(out, ref_o, 4e-3, 'attn out')
    valid = torch.ones(nseq, S, dtype=torch.bool, device=dev) if code is None else (code.reshape(nseq, S) >= 0)
    vm = valid[:, None, :].expand(nseq, nh, S)
    assert relerr(lse[vm], ref_lse[vm]) < 1e-3
    # backward: upstream gradient zero on masked-out (padded) query rows, as on the real path
    dout = rnd((nseq * S, H), dev, seed=2) * valid.reshape(-1, 1).to(BF16)
    ref_o.backward(dout.float())
    dqkv = torch.full_like(qkv, float('nan'))
    delta = torch.zeros(nseq, nh, S, device=dev)
    ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, None, nseq, S, nh)
    g = qr.grad
    for name, sl in (('dq', slice(0, H)), ('dk', slice(H, 2 * H)), ('dv', slice(2 * H, 3 * H))):
        assert_close(dqkv[:, sl], g[:, sl], 1.5e-2, f'attn {name}')
    # rotary-scale on the way out, with the qkv bias gradient (column sums of the stored dqkv) folded into the kernels
    from merlot_reserve_amd import _lib
    tab = torch.rand(S, 32, device=dev) * 2 - 1
    dq2 = torch.zeros_like(qkv)
    rows = _lib.load().mr_attention_bwd_colsum_rows(nseq, S)
    ws = torch.full((rows * 3 * H,), float('nan'), device=dev)
    bias_grad = torch.zeros(3 * H, dtype=BF16, device=dev)
    jobs = []
    ops.attention_bwd(qkv, code, out, dout, lse, delta, dq2, tab, nseq, S, nh, colsum_ws=ws, bias_grad=bias_grad, jobs=jobs)
    ops.reduce_partials(jobs)
    assert torch.isfinite(ws).all()                                   # every partial written
    assert_close(bias_grad, dq2.float().sum(0), 6e-3, 'attn fused qkv bias gradient')
    scale = torch.ones(nseq * S, 3 * H, device=dev)
    rows = torch.arange(nseq * S, device=dev) % S
    for h in range(2 * nh):
        scale[:, h * 64:h * 64 + 32] = tab[rows]
    assert_close(dq2, dqkv.float() * scale, 5e-3, 'attn bwd rot')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16], ids=['fp32', 'bf16'])
def test_attention_bwd_dense_mask(dev, dtype):
    """mr_attention_bwd_dense_mask (round 6: the backward of TransformerEncoder under ANY mask, mreserve/modeling.py:343-358) against autograd of the
    reference's formula -- additive -1e10 where the mask is 0, so a query row with no allowed key is uniform over all keys and ITS gradient flows --
    on a causal mask and on a random mask with an entirely masked row; with and without the "rotary" scales on the way out; and equal to the
    code-based kernels on a mask of the block form."""
    from merlot_reserve_amd import ops
    nseq, S, nh = 3, 77, 2
    H = nh * 64
    g = torch.Generator().manual_seed(11)
    qkv = torch.randn(nseq * S, 3 * H, generator=g).to(dtype).to(dev)
    dout = torch.randn(nseq * S, H, generator=g).to(dtype).to(dev)
    causal = torch.tril(torch.ones(S, S, dtype=torch.bool))[None].repeat(nseq, 1, 1)
    rndm = torch.rand(nseq, S, S, generator=g) < 0.6
    rndm[:, torch.arange(S), torch.arange(S)] = True
    rndm[1, 5, :] = False                                   # a row with no allowed key
    tol_o, tol_g = (2e-5, 2e-4) if dtype == torch.float32 else (4e-3, 1.5e-2)
    for mask in (causal, rndm):
        m8 = mask.to(torch.uint8).contiguous().to(dev)
        out = torch.zeros(nseq * S, H, dtype=dtype, device=dev)
        ops.attention_fwd_dense_mask(qkv, m8, out, nseq, S, nh)
        qr = qkv.float().clone().requires_grad_(True)
        q, k, v = qr.reshape(nseq, S, 3, nh, 64).unbind(2)
        sc = torch.einsum('nqhd,nkhd->nhqk', q / 8.0, k) + torch.where(mask.to(dev), 0.0, -1e10)[:, None]
        o = torch.einsum('nhqk,nkhd->nqhd', torch.softmax(sc, -1), v).reshape(nseq * S, H)
        assert_close(out, o, tol_o, 'dense-mask attn out')
        o.backward(dout.float())
        dqkv = torch.full_like(qkv, float('nan'))
        ops.attention_bwd_dense_mask(qkv, m8, dout, dqkv, None, nseq, S, nh)
        assert torch.isfinite(dqkv.float()).all()
        for name, sl in (('dq', slice(0, H)), ('dk', slice(H, 2 * H)), ('dv', slice(2 * H, 3 * H))):
            assert_close(dqkv[:, sl], qr.grad[:, sl], tol_g, f'dense-mask attn {name}')
        assert float(qr.grad[1 * S + 5, :H].abs().max()) > 0, 'the empty row carries gradient in the reference'
        tab = torch.rand(S, 32, device=dev) * 2 - 1
        dq2 = torch.zeros_like(qkv)
        ops.attention_bwd_dense_mask(qkv, m8, dout, dq2, tab, nseq, S, nh)
        scale = torch.ones(nseq * S, 3 * H, device=dev)
        rows = torch.arange(nseq * S, device=dev) % S
        for h in range(2 * nh):
            scale[:, h * 64:h * 64 + 32] = tab[rows]
        assert_close(dq2, qr.grad * scale, tol_g, 'dense-mask attn bwd rot')
    if dtype != torch.bfloat16:
        return
    # a mask of the block form: the code-based kernels on the same problem
    c = torch.randint(0, 2, (nseq, S), generator=g)
    c[torch.rand(nseq, S, generator=g) < 0.2] = -1
    c[:, 0] = 0
    blockm = ((c[:, :, None] == c[:, None, :]) & (c[:, :, None] >= 0))
    valid = (c >= 0).reshape(-1, 1).to(dev)
    dout_v = dout * valid.to(dtype)
    code = c.to(torch.int32).reshape(-1).to(dev)
    out1, lse, delta = torch.zeros(nseq * S, H, dtype=dtype, device=dev), torch.zeros(nseq, nh, S, device=dev), torch.zeros(nseq, nh, S, device=dev)
    ops.attention_fwd(qkv, code, out1, lse, nseq, S, nh)
    d1, d2 = torch.zeros_like(qkv), torch.zeros_like(qkv)
    ops.attention_bwd(qkv, code, out1, dout_v, lse, delta, d1, None, nseq, S, nh)
    ops.attention_bwd_dense_mask(qkv, blockm.to(torch.uint8).contiguous().to(dev), dout_v, d2, None, nseq, S, nh)
    assert relerr(d2, d1) < 1e-2, relerr(d2, d1)


def _run_codes(S, runs):
    c = []
    for code, n in runs:
        c += [code] * n
    assert len(c) == S, (len(c), S)
    return c


# position codes in RUNS, as real sequences carry them (video sources one after the other, PAD gaps / tails; planner.joint_code of the
# bench batch: [(0,108),(-1,52),(0,480)] / [(0,527),(-1,113)]): boundaries off and on the 64-position tile grid, several sources, an all-valid
# and an entirely empty sequence -- the (wave, tile) classes of csrc/attention.hip (fast / skip / general) all occur
RUN_MASKS = {
    640: [[(0, 108), (-1, 52), (0, 480)], [(0, 527), (-1, 113)], [(0, 200), (1, 250), (-1, 62), (2, 128)], [(0, 640)], [(-1, 640)],
          [(0, 128), (-1, 64), (1, 256), (-1, 192)]],
    200: [[(0, 70), (-1, 30), (0, 100)], [(1, 64), (2, 64), (-1, 72)]],
    130: [[(0, 128), (-1, 2)], [(-1, 3), (5, 127)]],
    1312: [[(0, 1000), (-1, 312)], [(0, 400), (1, 400), (2, 400), (-1, 112)]],
}


@pytest.mark.parametrize('zero_pad_dout', [False, True])
@pytest.mark.parametrize('S,nh,onepass', [(640, 3, -1), (200, 2, 0), (200, 2, 1), (130, 2, -1), (130, 2, 0), (1312, 2, -1)])
def test_attention_run_structured_masks(dev, S, nh, onepass, zero_pad_dout, attn_path):
    """Masks made of runs: against the fp32 torch reference (PAD rows get a non-zero upstream gradient too), and BIT FOR BIT against the same
    kernels with the tile classification switched off (option attn_tile_modes = 0: the general path everywhere).  zero_pad_dout: the upstream
    gradient of the PAD rows is zero, which the dQ kernel detects per row (delta = -0.0) and both backward kernels use to skip dead waves / tiles."""
    from merlot_reserve_amd import ops
    attn_path(onepass)
    runs = RUN_MASKS[S]
    nseq, H = len(runs), nh * 64
    qkv = rnd((nseq * S, 3 * H), dev, seed=11)
    code = torch.tensor([_run_codes(S, r) for r in runs], dtype=torch.int32).reshape(-1).to(dev)
    dout = rnd((nseq * S, H), dev, seed=12)
    if zero_pad_dout:      # as in a training step: nothing reads the PAD rows' outputs (the backward kernels skip waves / tiles made of such rows)
        dout = dout * (code >= 0).reshape(-1, 1).to(BF16)
    res = {}
    try:
        for modes in (1, 0):
            ops.set_option('attn_tile_modes', modes)
            out = torch.zeros(nseq * S, H, dtype=BF16, device=dev)
            lse = torch.zeros(nseq, nh, S, device=dev)
            ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
            dqkv = torch.full_like(qkv, float('nan'))
            delta = torch.zeros(nseq, nh, S, device=dev)
            ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, None, nseq, S, nh)
            res[modes] = (out, lse, dqkv)
    finally:
        ops.set_option('attn_tile_modes', 1)
    for a, b, name in zip(res[1], res[0], ('out', 'lse', 'dqkv')):
        assert torch.equal(a, b), f'{name}: the classified tiles must reproduce the general path bit for bit'
    out, lse, dqkv = res[1]
    qr = qkv.float().clone().requires_grad_(True)
    ref_o, ref_lse = ref_attention(qr, code, nseq, S, nh)
    assert_close(out, ref_o, 4e-3, 'attn out (run masks)')
    valid = (code.reshape(nseq, S) >= 0)[:, None, :].expand(nseq, nh, S)
    assert relerr(lse[valid], ref_lse[valid]) < 1e-3
    ref_o.backward(dout.float())
    assert torch.isfinite(dqkv.float()).all()
    for name, sl in (('dq', slice(0, H)), ('dk', slice(H, 2 * H)), ('dv', slice(2 * H, 3 * H))):
        assert_close(dqkv[:, sl], qr.grad[:, sl], 1.5e-2, f'attn {name} (run masks)')


@pytest.mark.parametrize('onepass', [-1, 1])
@pytest.mark.parametrize('nseq,S,nh,masked', [(2, 241, 2, False), (1, 100, 2, True), (3, 31, 2, False)])
def test_attention_reads_nothing_past_the_sequence(dev, nseq, S, nh, masked, onepass, attn_path):
    """The K / V (Q / dO) tiles are fetched by LDS-DMA in whole 64-row tiles; rows past a sequence's end must come back as zeros (the
    buffer descriptor's extent), never as whatever lies behind: here the operands are leading views of NaN-filled allocations, so a
    read past the LAST sequence meets NaN; and the first sequence computed alone (followed by NaN instead of the second sequence's
    rows) must give bit-identical rows, so nothing of a neighbouring sequence leaks in either."""
    from merlot_reserve_amd import ops
    attn_path(onepass)
    H = nh * 64
    rows = nseq * S
    big = torch.full((rows + 256, 3 * H), float('nan'), dtype=BF16, device=dev)
    bigd = torch.full((rows + 256, H), float('nan'), dtype=BF16, device=dev)
    bigo = torch.full((rows + 256, H), float('nan'), dtype=BF16, device=dev)
    qkv, dout, out = big[:rows], bigd[:rows], bigo[:rows]
    qkv.copy_(rnd((rows, 3 * H), dev, seed=1))
    dout.copy_(rnd((rows, H), dev, seed=2))
    code = None
    if masked:
        c = torch.zeros(nseq, S, dtype=torch.int32)
        c[:, 5:9] = -1
        code = c.reshape(-1).to(dev)
    lse = torch.zeros(nseq, nh, S, device=dev)
    delta = torch.zeros(nseq, nh, S, device=dev)
    dqkv = torch.zeros(rows, 3 * H, dtype=BF16, device=dev)
    ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, None, nseq, S, nh)
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all() and torch.isfinite(lse).all()
    qr = qkv.float().clone().requires_grad_(True)
    ref_o, _ = ref_attention(qr, code, nseq, S, nh)
    assert_close(out, ref_o, 4e-3, 'attn out')
    if nseq > 1:       # the first sequence alone (what follows it is now NaN too) gives the same rows
        out1 = torch.full((S + 256, H), float('nan'), dtype=BF16, device=dev)
        big1 = torch.full((S + 256, 3 * H), float('nan'), dtype=BF16, device=dev)
        big1[:S].copy_(qkv[:S])
        lse1 = torch.zeros(1, nh, S, device=dev)
        ops.attention_fwd(big1[:S], None if code is None else code[:S].contiguous(), out1[:S], lse1, 1, S, nh)
        assert torch.equal(out1[:S], out[:S]) and torch.equal(lse1[0], lse[0])


@pytest.mark.parametrize('onepass', [-1, 0, 1])
@pytest.mark.parametrize('nseq,S,nh', [(2, 22, 2), (1, 130, 2), (3, 241, 2)])
def test_attention_bwd_through_pad_query_rows(dev, nseq, S, nh, onepass, attn_path):
    """A PAD query row has no allowed key: every score is exactly -1e10, the reference's softmax is uniform over all S
    keys and autodiff sends its upstream gradient to q, k and v (the VCR head can pool at such a row when a sequence has
    no MASK).  Its LSE (-1e10 + ln S) is not representable in fp32, so the backward must not recompute P from it."""
    from merlot_reserve_amd import ops
    attn_path(onepass)
    H = nh * 64
    qkv = rnd((nseq * S, 3 * H), dev, seed=4)
    c = torch.zeros(nseq, S, dtype=torch.int32)
    c[:, 3:9] = -1
    c[0, :] = -1                                   # an entirely empty sequence
    code = c.reshape(-1).to(dev)
    out = torch.zeros(nseq * S, H, dtype=BF16, device=dev)
    lse = torch.zeros(nseq, nh, S, device=dev)
    ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    qr = qkv.float().clone().requires_grad_(True)
    ref_o, _ = ref_attention(qr, code, nseq, S, nh)
    assert_close(out, ref_o, 4e-3, 'attn out (pad rows included)')
    dout = rnd((nseq * S, H), dev, seed=5)         # non-zero upstream gradient on the PAD rows too
    ref_o.backward(dout.float())
    dqkv = torch.full_like(qkv, float('nan'))
    delta = torch.zeros(nseq, nh, S, device=dev)
    ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, None, nseq, S, nh)
    assert torch.isfinite(dqkv.float()).all()
    for name, sl in (('dq', slice(0, H)), ('dk', slice(H, 2 * H)), ('dv', slice(2 * H, 3 * H))):
        assert_close(dqkv[:, sl], qr.grad[:, sl], 1.5e-2, f'attn {name} with pad query rows')


def test_poolattn(dev):
    from merlot_reserve_amd import ops
    nh, H, G, R, rows = 2, 128, 37, 5, 37 * 5 + 3
    q, k, v = rnd((G, H), dev, seed=1), rnd((rows, H), dev, seed=2), rnd((rows, H), dev, seed=3)
    perm = torch.randperm(rows, generator=torch.Generator().manual_seed(0))[:G * R].reshape(G, R).to(torch.int32).to(dev)
    out = torch.zeros(G, H, dtype=BF16, device=dev)
    probs = torch.zeros(G, nh, R, device=dev)
    ops.poolattn_fwd(q, k, v, perm, out, probs, nh)
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    kk = kr[perm.long()].reshape(G, R, nh, 64)
    vv = vr[perm.long()].reshape(G, R, nh, 64)
    s = torch.einsum('ghd,grhd->ghr', qr.reshape(G, nh, 64) / 8.0, kk)
    p = torch.softmax(s, -1)
    ref = torch.einsum('ghr,grhd->ghd', p, vv).reshape(G, H)
    assert_close(out, ref, 4e-3, 'poolattn out')
    assert relerr(probs, p) < 1e-4
    dout = rnd((G, H), dev, seed=4)
    ref.backward(dout.float())
    dq = torch.zeros_like(q)
    dk, dv = torch.zeros_like(k), torch.zeros_like(v)
    ops.poolattn_bwd(q, k, v, perm, probs, dout, dq, dk, dv, nh)
    assert_close(dq, qr.grad, 6e-3, 'poolattn dq')
    assert_close(dk, kr.grad, 6e-3, 'poolattn dk')
    assert_close(dv, vr.grad, 6e-3, 'poolattn dv')


def test_segment_sum_and_rows_mean(dev):
    from merlot_reserve_amd import ops
    H = 768
    t0, t1, t2 = rnd((50, H), dev, seed=1), rnd((20, H), dev, seed=2), rnd((30, H), dev, seed=3)
    g = torch.Generator().manual_seed(3)
    lens = torch.randint(0, 5, (40,), generator=g)
    indptr = torch.zeros(41, dtype=torch.int32)
    indptr[1:] = torch.cumsum(lens, 0)
    idx = torch.randint(0, 100, (int(indptr[-1]),), generator=g).to(torch.int32)
    dst = torch.full((40, H), float('nan'), dtype=BF16, device=dev)
    ops.segment_sum([t0, t1, t2], indptr.to(dev), idx.to(dev), dst)
    cat = torch.cat([t0, t1, t2]).float()
    ref = torch.stack([cat[idx[indptr[i]:indptr[i + 1]].long()].sum(0) if lens[i] > 0 else torch.zeros(H, device=dev)
                       for i in range(40)])
    assert_close(dst, ref, 4e-3, 'segment_sum')
    base = rnd((40, H), dev, seed=9)
    dst2 = base.clone()
    ops.segment_sum([t0, t1, t2], indptr.to(dev), idx.to(dev), dst2, scale=0.5, accumulate=True)
    assert_close(dst2, base.float() + 0.5 * ref, 4e-3, 'segment_sum accumulate')
    dst3 = torch.zeros(40, H, dtype=F32, device=dev)
    ops.segment_sum([t0, t1, t2], indptr.to(dev), idx.to(dev), dst3)
    assert relerr(dst3, ref) < 1e-6
    rows = torch.randperm(50, generator=g)[:48].reshape(12, 4).to(torch.int32).to(dev)
    m = torch.zeros(12, H, dtype=BF16, device=dev)
    ops.rows_mean_fwd(t0, rows, m)
    assert_close(m, t0.float()[rows.long()].mean(1), 4e-3, 'rows_mean')
    d = rnd((12, H), dev, seed=5)
    acc = t0.clone()
    ops.rows_mean_bwd(d, rows, acc)
    ref_acc = t0.float().clone()
    ref_acc[rows.long().reshape(-1)] += (d.float() / 4).repeat_interleave(4, 0)
    assert_close(acc, ref_acc, 4e-3, 'rows_mean bwd')


def test_small_helpers(dev):
    from merlot_reserve_amd import ops
    src = rnd((90, 130), dev, seed=1)
    dst = torch.full((90, 136), float('nan'), dtype=BF16, device=dev)
    ops.pad_cols(src, dst)
    assert torch.equal(dst[:, :130], src) and torch.all(dst[:, 130:] == 0)
    vec = rnd((128,), dev, seed=2)
    buf = torch.zeros(3 * 17, 128, dtype=BF16, device=dev)
    ops.fill_rows(vec, buf, 3, 17, 0)
    assert torch.equal(buf[0], vec) and torch.equal(buf[17], vec) and torch.equal(buf[34], vec) and torch.all(buf[1] == 0)
    x = rnd((3 * 17, 128), dev, seed=3)
    out = torch.zeros(128, dtype=BF16, device=dev)
    ops.sum_rows_strided(x, 3, 17, 0, out)
    assert_close(out, x.float()[[0, 17, 34]].sum(0), 4e-3, 'sum_rows_strided')
    x = rnd((70 * 5, 200), dev, seed=6)                     # more groups than the kernel's 16 row groups, ragged column count
    out = torch.zeros(200, dtype=BF16, device=dev)
    ops.sum_rows_strided(x, 70, 5, 2, out)
    assert_close(out, x.float()[2::5].sum(0), 4e-3, 'sum_rows_strided (70 groups)')
    a, b = rnd((64, 128), dev, seed=4), rnd((64, 128), dev, seed=5)
    y = torch.zeros_like(a)
    ops.add_(a, b, y)
    assert torch.equal(y, (a.float() + b.float()).to(BF16))
    f = torch.randn(1000, device=dev)
    o = torch.zeros(1000, dtype=BF16, device=dev)
    ops.cast_f32_to_bf16(f, o)
    assert torch.equal(o, f.to(BF16))


@pytest.mark.parametrize('ls', [1.0, 5.0])
def test_unit_norm_scale(dev, ls):
    from merlot_reserve_amd import ops
    rows, H = 77, 768
    x = rnd((rows, H), dev, scale=3.0, seed=1)
    lsb = torch.tensor([ls], dtype=BF16, device=dev)
    y = torch.zeros_like(x)
    inv = torch.zeros(rows, device=dev)
    ops.unit_norm_scale_fwd(x, lsb, y, inv)
    xr = x.float().clone().requires_grad_(True)
    lr = lsb.float().clone().requires_grad_(True)
    temp = torch.exp(torch.clamp(lr, max=math.log(100.0)) / 2)
    ref = xr / torch.sqrt((xr * xr).sum(-1, keepdim=True) + 1e-5) * temp
    assert_close(y, ref, 6e-3, 'unit_norm fwd')
    dy = rnd((rows, H), dev, seed=2)
    ref.backward(dy.float())
    dx = torch.zeros_like(x)
    dls = torch.full((1,), 123.0, device=dev)
    part = torch.zeros((rows + 3) // 4, device=dev)
    ops.unit_norm_scale_bwd(x, lsb, inv, dy, dx, dls, part)            # accumulate = False overwrites
    assert_close(dx, xr.grad, 6e-3, 'unit_norm dx')
    if ls < math.log(100.0):
        assert abs(dls.item() - lr.grad.item()) <= 2e-2 * abs(lr.grad.item()) + 1e-3
    else:
        assert dls.item() == 0.0
    first = dls.clone()
    ops.unit_norm_scale_bwd(x, lsb, inv, dy, dx, dls, part, accumulate=True)
    assert dls.item() == 2 * first.item()                               # fixed-order sums: bitwise reproducible


def test_contrastive_lse(dev):
    from merlot_reserve_amd import ops
    L, V, off = 48, 192, 96
    logits = torch.randn(L, V, device=dev) * 5
    src = torch.randint(-1, 3, (L,), generator=torch.Generator().manual_seed(1)).to(torch.int32).to(dev)
    lr = logits.clone().requires_grad_(True)
    lse = torch.logsumexp(lr, -1)
    numer = lr[torch.arange(L), off + torch.arange(L)]
    coef = 0.5 / L
    ref_loss = coef * (lse - numer).sum()
    ref_loss.backward()
    loss = torch.zeros(1, device=dev)
    diag = torch.zeros(6, device=dev)
    work = logits.clone()
    rows = torch.zeros(L, device=dev)
    ops.contrastive_lse(work, off, coef, src, loss, diag, rows)
    loss2, diag2 = torch.zeros(1, device=dev), torch.zeros(6, device=dev)
    ops.contrastive_lse(logits.clone(), off, coef, src, loss2, diag2, rows)
    assert torch.equal(loss, loss2) and torch.equal(diag, diag2)        # no float atomics: bitwise reproducible
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * abs(ref_loss.item()) + 1e-6
    assert relerr(work, lr.grad) < 1e-4
    for i in range(3):
        m = src == i
        assert abs(diag[i].item() - (lse - numer)[m].sum().item()) < 1e-3 and diag[3 + i].item() == m.sum().item()


def test_adam_bf16(dev):
    from merlot_reserve_amd import ops
    from oracle import ref_torch as R
    n = 2048 * 3
    g = torch.Generator().manual_seed(0)
    master = torch.randn(n, generator=g) * 0.05
    grad = (torch.randn(n, generator=g) * 1e-3).to(BF16)
    grad[5] = float('nan'); grad[6] = float('inf'); grad[7] = -float('inf')
    mu = (torch.randn(n, generator=g) * 1e-3).to(BF16)
    nu = R.unsigned_bf16_encode((torch.rand(n, generator=g) * 1e-3).float())
    nu[:10] = 0.0
    flags = torch.tensor([1, 0, 1], dtype=torch.uint8)
    cfg = dict(learning_rate=4e-4, num_train_steps=750000, num_warmup_steps=3750, weight_decay_rate=0.1, beta_2=0.98,
               eps=1e-6, use_bfloat16_adam=True)
    count = 100
    sched = float(R.lr_scale_linearwarmup_cosinedecay(count, 3750, 750000, 0.02))
    d_master, d_mu, d_nu = master.to(dev), mu.to(dev), nu.to(dev)
    work = torch.zeros(n, dtype=BF16, device=dev)
    ops.adam_bf16_update(d_master, work, grad.to(dev), d_mu, d_nu, flags.to(dev), 0.9, 0.98, 1e-6, 0.1, sched, -4e-4)
    gclean = torch.nan_to_num(grad.float(), nan=0.0, posinf=float(torch.finfo(BF16).max), neginf=-float(torch.finfo(BF16).max))
    for blk in range(3):
        sl = slice(blk * 2048, (blk + 1) * 2048)
        p = master[sl].reshape(2048, 1) if flags[blk] else master[sl]          # ndim > 1 <=> decay
        gg = gclean[sl].reshape(p.shape)
        newp, nm, nv = R.adam_bf16_apply(p, gg, mu[sl].reshape(p.shape), nu[sl].reshape(p.shape), count, cfg)
        finp = torch.isfinite(newp.reshape(-1))
        assert relerr(d_master[sl].cpu()[finp], newp.reshape(-1)[finp]) < 1e-6, blk
        finm = torch.isfinite(nm.reshape(-1).float())
        assert (d_mu[sl].cpu().float()[finm] - nm.reshape(-1).float()[finm]).abs().max() <= 1e-2 * nm.float()[finm.reshape(nm.shape)].abs().max()
        # the cube-root codec: decoded values must agree to codec precision
        dec_got = R.unsigned_bf16_decode(d_nu[sl].cpu())
        dec_ref = R.unsigned_bf16_decode(nv.reshape(-1))
        fin = torch.isfinite(dec_ref)                      # +-inf grads (clamped to bf16 max) overflow g^2 in both
        assert torch.equal(fin, torch.isfinite(dec_got))
        assert relerr(dec_got[fin], dec_ref[fin]) < 2e-3
        assert torch.equal(work[sl].cpu(), d_master[sl].cpu().to(BF16))


def test_adam_bf16_state_words_bit_exact(dev):
    """The optimizer state is byte data: mu = bf16(m'), nu = the sign-coded bf16 of v'^3 (pretrain/optimization.py:36-51 defines the
    code exactly).  The kernel's words must EQUAL the oracle's on >= 99.9 % of 65 536 elements; the rest -- an fp32 ulp from fma
    contraction / the device cbrt landing on the other side of a bf16 rounding or sign-code boundary -- must be neighbouring codes
    (one bf16 ulp in magnitude, or the same magnitude with the other sign code)."""
    from merlot_reserve_amd import ops
    from oracle import ref_torch as R
    n = 2048 * 32
    g = torch.Generator().manual_seed(1)
    master = torch.randn(n, generator=g) * 0.05
    grad = (torch.randn(n, generator=g) * 1e-3 * torch.exp(torch.randn(n, generator=g))).to(BF16)
    mu = (torch.randn(n, generator=g) * 1e-3).to(BF16)
    nu = R.unsigned_bf16_encode((torch.rand(n, generator=g) * 1e-3 * torch.exp(torch.randn(n, generator=g))).float())
    nu[:64] = 0.0
    flags = torch.ones(32, dtype=torch.uint8)
    cfg = dict(learning_rate=4e-4, num_train_steps=750000, num_warmup_steps=3750, weight_decay_rate=0.1, beta_2=0.98,
               eps=1e-6, use_bfloat16_adam=True)
    count = 4000
    sched = float(R.lr_scale_linearwarmup_cosinedecay(count, 3750, 750000, 0.02))
    d_master, d_mu, d_nu = master.to(dev), mu.to(dev), nu.to(dev)
    work = torch.zeros(n, dtype=BF16, device=dev)
    ops.adam_bf16_update(d_master, work, grad.to(dev), d_mu, d_nu, flags.to(dev), 0.9, 0.98, 1e-6, 0.1, sched, -4e-4)
    _, nm, nv = R.adam_bf16_apply(master.reshape(n, 1), grad.float().reshape(n, 1), mu.reshape(n, 1), nu.reshape(n, 1), count, cfg)
    bits = lambda t: t.reshape(-1).contiguous().view(torch.int16).to(torch.int32) & 0xFFFF
    for name, got, ref in (('mu', d_mu.cpu(), nm), ('nu', d_nu.cpu(), nv)):
        gb, rb = bits(got), bits(ref)
        neq = (gb != rb).nonzero().reshape(-1)
        frac = 1.0 - neq.numel() / n
        print(f'adam state {name}: {n - neq.numel()} of {n} bf16 words equal ({frac * 100:.4f} %); {neq.numel()} differ')
        assert frac >= 0.999, (name, frac)
        for i in neq.tolist():
            a, b = int(gb[i]), int(rb[i])
            mag = abs((a & 0x7FFF) - (b & 0x7FFF))
            assert mag <= 1, (name, i, hex(a), hex(b))          # neighbouring magnitude codes; (nu) possibly the other sign code


GEMM256_CASES = [
    # shapes that dispatch to the 256-row LDS-DMA kernel (M >= 512): ragged M, both tile widths, all layouts, split-K
    (15424 // 4, 768, 768, False, False),
    (1000, 2304, 768, False, False),
    (1000, 3072, 768, False, False),
    (777, 768, 3072, False, True),
    (520, 200, 128, False, True),       # N not a multiple of the tile
    (768, 3072, 1500, True, False),     # wgrad: split-K, ragged K
    (3072, 768, 1500, True, False),
    (768, 768, 5000, True, False),
    (640, 264, 704, True, True),
    (8000, 3072, 192, False, False),    # >= 512 tiles: XCD-blocked tile order
    (16000, 800, 128, False, True),     # same, ragged N and M
    (9000, 1544, 128, True, False),
]


@pytest.fixture(params=[0, 96, 128, 192, 256])
def tile_n(request):
    """Forces the output-tile width of the 256-row GEMM (0 = the library's own choice) for the duration of a test."""
    from merlot_reserve_amd import _lib
    _lib.check(_lib.load().mr_set_option(b'gemm_tile_n', request.param), 'mr_set_option')
    yield request.param
    _lib.load().mr_set_option(b'gemm_tile_n', 0)


@pytest.mark.parametrize('M,N,K,ta,tb', GEMM256_CASES)
def test_gemm256(dev, tile_n, M, N, K, ta, tb):
    from merlot_reserve_amd import ops
    ws = torch.zeros(32 * 1024 * 1024, device=dev)          # split-K partials (caller-owned, per call)
    a = rnd((K, M) if ta else (M, K), dev, seed=1)
    b = rnd((N, K) if tb else (K, N), dev, seed=2)
    out = torch.full((M, N), float('nan'), dtype=BF16, device=dev)
    ops.gemm(a, b, out, transA=ta, transB=tb, ws=ws)
    A = a.float().T if ta else a.float()
    B = b.float().T if tb else b.float()
    assert_close(out, A @ B, 3e-3, f'gemm256 {M}x{N}x{K} ta={ta} tb={tb}')


def test_gemm256_epilogues(dev, tile_n):
    from merlot_reserve_amd import ops
    M, N, K = 241 * 4, 384, 128
    a, w = rnd((M, K), dev, seed=6), rnd((K, N), dev, scale=0.1, seed=7)
    bias = rnd((N,), dev, seed=8)
    tab = torch.rand(241, 32, device=dev) * 2 - 1
    out = torch.zeros(M, N, dtype=BF16, device=dev)
    ops.gemm(a, w, out, bias=bias, rot_tab=tab, rot_cols=256)
    ref = a.float() @ w.float() + bias.float()
    scale = torch.ones(M, N, device=dev)
    rows = torch.arange(M, device=dev) % 241
    for h in range(4):
        scale[:, h * 64:h * 64 + 32] = tab[rows]
    assert_close(out, ref * scale, 3e-3, 'rot epilogue')
    pre = torch.zeros(M, N, dtype=BF16, device=dev)
    ops.gemm(a, w, out, bias=bias, act=ops.ACT_GELU, c2=pre)
    sg = torch.sigmoid(1.702 * ref)
    assert_close(pre, sg + 1.702 * ref * sg * (1 - sg), 4e-3, 'c2 = gelu grad')
    assert_close(out, ref * sg, 4e-3, 'gelu')
    res = rnd((M, N), dev, seed=9)
    buf = res.clone()
    ops.gemm(a, w, buf, residual=buf)
    assert_close(buf, (a.float() @ w.float()).to(BF16).float() + res.float(), 3e-3, 'residual in place')
    aux = rnd((M, N), dev, seed=10)
    ops.gemm(a, w, out, aux=aux)
    assert_close(out, (a.float() @ w.float()).to(BF16).float() * aux.float(), 4e-3, 'aux multiply')
    M2 = 240 * 4
    a2 = rnd((M2, K), dev, seed=11)
    big = torch.zeros(4 * 241, N, dtype=BF16, device=dev)
    ops.gemm(a2, w, big, bias=bias, row_map=(240, 241, 1))
    got = big.reshape(4, 241, N)
    assert torch.all(got[:, 0] == 0)
    assert_close(got[:, 1:].reshape(M2, N), a2.float() @ w.float() + bias.float(), 3e-3, 'row map')
    o32 = torch.zeros(M, N, dtype=F32, device=dev)
    ops.gemm(a, w, o32, bias=bias)
    assert relerr(o32, ref) < 1e-5


@pytest.mark.parametrize('M,N,K', [(15424 // 4, 3072, 768), (1500, 768, 512), (2304, 1024, 256)])
def test_gemm256_fused_colsum(dev, tile_n, M, N, K):
    """The aux-epilogue GEMM (d pre-activation = (dY . W2^T) * gelu') with the bias gradient folded in: per-(tile, wave) column
    sums of the STORED bf16 output, reduced by mr_reduce_partials -- equal to the separate column-sum pass over the output."""
    from merlot_reserve_amd import _lib, ops
    a, w = rnd((M, K), dev, seed=1), rnd((N, K), dev, scale=0.1, seed=2)
    aux = rnd((M, N), dev, seed=3)
    out = torch.full((M, N), float('nan'), dtype=BF16, device=dev)
    rows = _lib.load().mr_gemm_colsum_rows(M)
    ws = torch.full((rows * N,), float('nan'), device=dev)
    bias_grad = torch.zeros(N, dtype=BF16, device=dev)
    jobs = []
    fused = ops.gemm_colsum_job(a, w, out, ws, bias_grad, jobs, transB=True, aux=aux)
    assert fused and len(jobs) == 1
    ops.reduce_partials(jobs)
    ref = (a.float() @ w.float().T).to(BF16).float() * aux.float()
    assert_close(out, ref, 4e-3, 'aux GEMM')
    want = out.float().sum(0)                                   # column sums of what was stored
    assert_close(bias_grad, want, 6e-3, 'fused bias gradient')
    ref2 = torch.zeros(N, dtype=BF16, device=dev)
    ops.colsum(out, ref2, ops.colsum_workspace(N, dev))
    assert_close(bias_grad, ref2.float(), 6e-3, 'fused vs separate column sum')
    # problems the 256-row kernel does not take fall back to the separate pass
    small = rnd((300, K), dev, seed=4)
    o2 = torch.zeros(300, N, dtype=BF16, device=dev)
    assert not ops.gemm_colsum_job(small, w, o2, ws, bias_grad, [], transB=True, aux=aux[:300])


@pytest.mark.parametrize('group_tile', [0, 128, 256])
def test_gemm_grouped(dev, group_tile):
    """Four wgrad-shaped problems in one persistent launch == four separate GEMMs (both tile widths of the shared launch)."""
    from merlot_reserve_amd import _lib, ops
    _lib.check(_lib.load().mr_set_option(b'gemm_group_tile_n', group_tile), 'mr_set_option')
    Mtok, H = 1500, 512
    xs = [rnd((Mtok, 4 * H), dev, seed=1), rnd((Mtok, H), dev, seed=2), rnd((Mtok, H), dev, seed=3), rnd((Mtok, H), dev, seed=4)]
    ds = [rnd((Mtok, H), dev, seed=5), rnd((Mtok, 4 * H), dev, seed=6), rnd((Mtok, H), dev, seed=7), rnd((Mtok, 3 * H), dev, seed=8)]
    outs = [torch.full((x.shape[1], d.shape[1]), float('nan'), dtype=BF16, device=dev) for x, d in zip(xs, ds)]
    ops.gemm_grouped([ops.gemm_args(x, d, o, transA=True) for x, d, o in zip(xs, ds, outs)])
    for x, d, o in zip(xs, ds, outs):
        assert_close(o, x.float().T @ d.float(), 3e-3, f'grouped {tuple(o.shape)}')
    # a group that does not qualify (small M) falls back to separate launches with the same results
    small = [rnd((300, 128), dev, seed=9), rnd((300, 256), dev, seed=10)]
    so = [torch.zeros(128, 128, dtype=BF16, device=dev), torch.zeros(256, 128, dtype=BF16, device=dev)]
    dd = rnd((300, 128), dev, seed=11)
    ops.gemm_grouped([ops.gemm_args(x, dd, o, transA=True) for x, o in zip(small, so)])
    for x, o in zip(small, so):
        assert_close(o, x.float().T @ dd.float(), 3e-3, 'grouped fallback')
    _lib.load().mr_set_option(b'gemm_group_tile_n', 0)
