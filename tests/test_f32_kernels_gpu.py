"""fp32 forward kernels (mr_f32_*) on the MI355X against fp64 PyTorch references of the same op.
Tolerance: fp32 arithmetic with fp32 accumulation -> rel-L2 <= 2e-6 for GEMMs (K <= 3072), 1e-5 for softmax paths."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
F32, F64 = torch.float32, torch.float64


def rnd(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


def relerr(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-300)).item()


CASES = [(128, 128, 16, False, False), (300, 264, 200, False, False), (964, 384, 128, False, False),
         (300, 264, 200, False, True), (304, 264, 200, True, False), (304, 264, 200, True, True),
         (257, 3072, 768, False, False), (1000, 768, 3072, False, False), (60, 128, 130, False, False),
         (37, 12, 768, False, True)]


@pytest.mark.parametrize('M,N,K,ta,tb', CASES)
def test_f32_gemm(dev, M, N, K, ta, tb):
    from merlot_reserve_amd import ops
    a = rnd((K, M) if ta else (M, K), dev, seed=1)
    b = rnd((N, K) if tb else (K, N), dev, seed=2)
    out = torch.full((M, N), float('nan'), dtype=F32, device=dev)
    ops.gemm(a, b, out, transA=ta, transB=tb)
    A = a.double().T if ta else a.double()
    B = b.double().T if tb else b.double()
    e = relerr(out, A @ B)
    assert math.isfinite(e) and e < 2e-6, e


def test_f32_gemm_unaligned_views_and_identity(dev):
    from merlot_reserve_amd import ops
    n = 128
    a = torch.eye(n, dtype=F32, device=dev)
    b = (torch.arange(n * n, device=dev).reshape(n, n) % 251).to(F32)
    out = torch.zeros(n, n, dtype=F32, device=dev)
    for ta in (False, True):
        for tb in (False, True):
            ops.gemm(a, b.T.contiguous() if tb else b, out, transA=ta, transB=tb)
            assert torch.equal(out, b), (ta, tb)
    # operand with an odd leading dimension (audio conv input: 130 columns) -> scalar-load path
    x = rnd((90, 131), dev, seed=3)[:, :130]
    w = rnd((130, 64), dev, seed=4)
    o = torch.zeros(90, 64, dtype=F32, device=dev)
    ops.gemm(x, w, o)
    assert relerr(o, x.double() @ w.double()) < 2e-6


def test_f32_gemm_epilogue(dev):
    from merlot_reserve_amd import ops
    S, nseq, H = 13, 5, 128
    M, N, K = S * nseq, 3 * H, H
    a, w, bias = rnd((M, K), dev, seed=1), rnd((K, N), dev, 0.1, seed=2), rnd((N,), dev, seed=3)
    rot = rnd((S, 32), dev, seed=4)
    out = torch.zeros(M, N, dtype=F32, device=dev)
    ops.gemm(a, w, out, bias=bias, rot_tab=rot, rot_cols=2 * H)
    ref = a.double() @ w.double() + bias.double()
    sc = torch.ones(M, N, dtype=F64, device=dev)
    cols = torch.arange(N, device=dev)
    m = (cols < 2 * H) & ((cols & 63) < 32)
    sc[:, m] = rot.double().repeat(nseq, 1)[:, (cols[m] & 63)]
    assert relerr(out, ref * sc) < 2e-6
    # gelu + residual + CLS row map
    res = rnd((nseq * (S + 1), N), dev, seed=5)
    out2 = torch.zeros(nseq * (S + 1), N, dtype=F32, device=dev)
    ops.gemm(a, w, out2, bias=bias, act=ops.ACT_GELU, residual=res, row_map=(S, S + 1, 1))
    pre = a.double() @ w.double() + bias.double()
    g = pre * torch.sigmoid(1.702 * pre)
    ref2 = torch.zeros_like(out2, dtype=F64)
    ref2.view(nseq, S + 1, N)[:, 1:] = g.view(nseq, S, N) + res.double().view(nseq, S + 1, N)[:, 1:]
    assert relerr(out2, ref2) < 2e-6
    assert torch.equal(out2.view(nseq, S + 1, N)[:, 0], torch.zeros(nseq, N, device=dev))


def test_f32_layernorm(dev):
    from merlot_reserve_amd import ops
    x = rnd((77, 768), dev, 3.0, seed=1) + 0.7
    g, b = rnd((768,), dev, seed=2), rnd((768,), dev, seed=3)
    y = torch.zeros_like(x)
    ops.layernorm_fwd(x, g, b, y)
    ref = torch.nn.functional.layer_norm(x.double(), (768,), g.double(), b.double(), 1e-5)
    assert relerr(y, ref) < 2e-6
    # the reference's hand-derived known answer
    x4 = torch.tensor([[1., 2., 3., 4.]], device=dev)
    y4 = torch.zeros_like(x4)
    ops.layernorm_fwd(x4, torch.ones(4, device=dev), torch.zeros(4, device=dev), y4)
    assert torch.allclose(y4.cpu(), torch.tensor([[-1.341635, -0.447212, 0.447212, 1.341635]]), atol=2e-6)


def attn_ref(qkv, code, nseq, S, nh):
    H = nh * 64
    x = qkv.double().view(nseq, S, 3, nh, 64)
    q, k, v = x[:, :, 0], x[:, :, 1], x[:, :, 2]
    sc = torch.einsum('bqhd,bkhd->bhqk', q / 8.0, k)
    if code is not None:
        c = code.view(nseq, S)
        allowed = (c[:, :, None] == c[:, None, :]) & (c[:, :, None] >= 0)
        sc = (sc.float() + torch.where(allowed, 0.0, -1e10)[:, None].float()).double()   # the reference adds the bias in fp32
    p = torch.softmax(sc, -1)
    return torch.einsum('bhqk,bkhd->bqhd', p, v).reshape(nseq * S, H), torch.logsumexp(sc, -1)


@pytest.mark.parametrize('nseq,S,nh,masked', [(3, 16, 2, True), (2, 31, 2, False), (2, 241, 3, False), (2, 640, 2, True), (1, 100, 12, True)])
def test_f32_attention(dev, nseq, S, nh, masked):
    from merlot_reserve_amd import ops
    H = nh * 64
    qkv = rnd((nseq * S, 3 * H), dev, seed=S)
    code = None
    if masked:
        g = torch.Generator().manual_seed(S)
        code = torch.randint(0, 3, (nseq * S,), generator=g, dtype=torch.int32)
        code[torch.rand(nseq * S, generator=g) < 0.15] = -1
        code = code.to(dev)
    out = torch.full((nseq * S, H), float('nan'), dtype=F32, device=dev)
    lse = torch.zeros(nseq, nh, S, dtype=F32, device=dev)
    ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    ref, lse_ref = attn_ref(qkv, code, nseq, S, nh)
    if masked:
        ok = (code >= 0)
        assert relerr(out[ok], ref[ok]) < 1e-5
        # fully masked (pad) query rows: uniform softmax over all S keys, as in the reference
        pad = ~ok
        if pad.any():
            vmean = qkv.double().view(nseq, S, 3, H)[:, :, 2].mean(1)
            want = vmean[:, None].expand(nseq, S, H).reshape(nseq * S, H)
            assert relerr(out[pad], want[pad]) < 1e-5
    else:
        assert relerr(out, ref) < 1e-5
        assert relerr(lse, lse_ref) < 1e-6


def test_f32_rowops(dev):
    from merlot_reserve_amd import ops
    H, nh = 128, 2
    t0, t1 = rnd((50, H), dev, seed=1), rnd((20, H), dev, seed=2)
    indptr = torch.tensor([0, 1, 1, 3, 4], dtype=torch.int32, device=dev)
    idx = torch.tensor([49, 3, 50 + 19, 7], dtype=torch.int32, device=dev)
    dst = torch.full((4, H), float('nan'), device=dev)
    ops.segment_sum([t0, t1], indptr, idx, dst)
    want = torch.stack([t0[49], torch.zeros(H, device=dev), t0[3] + t1[19], t0[7]])
    assert torch.allclose(dst, want, atol=1e-6)
    rows = torch.tensor([[0, 1, 2, 3], [10, 11, 12, 13]], dtype=torch.int32, device=dev)
    m = torch.zeros(2, H, device=dev)
    ops.rows_mean_fwd(t0, rows, m)
    assert torch.allclose(m, torch.stack([t0[0:4].mean(0), t0[10:14].mean(0)]), atol=1e-6)
    # attention pooling core: one query, R keys
    q, k, v = rnd((2, H), dev, seed=3), rnd((50, H), dev, seed=4), rnd((50, H), dev, seed=5)
    o = torch.zeros(2, H, device=dev)
    ops.poolattn_fwd(q, k, v, rows, o, None, nh)
    for gi in range(2):
        kk, vv = k[rows[gi].long()].double().view(4, nh, 64), v[rows[gi].long()].double().view(4, nh, 64)
        s = torch.einsum('hd,rhd->hr', q[gi].double().view(nh, 64) / 8.0, kk)
        ref = torch.einsum('hr,rhd->hd', torch.softmax(s, -1), vv).reshape(H)
        assert relerr(o[gi], ref) < 1e-5
    # unit_normalize (+ temperature)
    y = torch.zeros_like(t0)
    ops.unit_norm_scale_fwd(t0, None, y)
    assert relerr(y, t0.double() / torch.sqrt((t0.double() ** 2).sum(-1, keepdim=True) + 1e-5)) < 1e-6
    ls = torch.tensor([5.0], device=dev)          # clipped at ln 100
    ops.unit_norm_scale_fwd(t0, ls, y)
    assert relerr(y, 10.0 * t0.double() / torch.sqrt((t0.double() ** 2).sum(-1, keepdim=True) + 1e-5)) < 1e-6
    z = torch.zeros(3 * 5, H, device=dev)
    ops.fill_rows(t0[7], z, 3, 5, 0)
    assert torch.equal(z.view(3, 5, H)[:, 0], t0[7].expand(3, H)) and float(z.view(3, 5, H)[:, 1:].abs().sum()) == 0.0


# ---------------------------------------------------------------------------------------------- fp32 backward kernels (csrc/f32bwd.hip)
def test_f32_layernorm_bwd(dev):
    """dx (+ the residual-path gradient), dgamma, dbeta against fp64 autograd; dx may alias dy (the engine's final / pre LayerNorm)."""
    from merlot_reserve_amd import ops
    rows, H = 77, 768
    x = rnd((rows, H), dev, 3.0, seed=1) + 0.7
    g = rnd((H,), dev, seed=2) + 1.0
    dy, add = rnd((rows, H), dev, seed=3), rnd((rows, H), dev, seed=4)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), torch.zeros(H, dtype=F64, device=dev, requires_grad=True)
    torch.nn.functional.layer_norm(xd, (H,), gd, bd, 1e-5).backward(dy.double())
    ws = torch.zeros(2 * rows, device=dev)
    dx, dgam, dbet = torch.zeros_like(x), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    mean = rstd = torch.zeros(rows, device=dev)                      # (the fp32 kernel recomputes the statistics)
    ops.layernorm_bwd(dy, x, g, mean, rstd, dx, dgam, dbet, ws, dx_add=add)
    assert relerr(dx, xd.grad + add.double()) < 2e-6
    assert relerr(dgam, gd.grad) < 2e-6 and relerr(dbet, bd.grad) < 2e-6
    buf = dy.clone()
    ops.layernorm_bwd(buf, x, g, mean, rstd, buf, dgam, dbet, ws)    # in place
    assert relerr(buf, xd.grad) < 2e-6


@pytest.mark.parametrize('nseq,S,nh,masked', [(3, 16, 2, True), (2, 31, 2, False), (2, 241, 3, False), (2, 200, 2, True)])
def test_f32_attention_bwd(dev, nseq, S, nh, masked):
    """dQ, dK, dV (with the "rotary" scaling of dq / dk undone) against fp64 autograd of the reference's attention, PAD query rows included:
    their scores are all -1e10, the softmax is uniform over the S keys, their dO flows into dV and -- autodiff does not know about the rounding --
    their dS into dq and dk (modeling.py:343-358)."""
    from merlot_reserve_amd import ops
    H = nh * 64
    qkv = rnd((nseq * S, 3 * H), dev, seed=S)
    dout = rnd((nseq * S, H), dev, seed=S + 1)
    code = None
    if masked:
        gen = torch.Generator().manual_seed(S)
        code = torch.randint(0, 3, (nseq * S,), generator=gen, dtype=torch.int32)
        code[torch.rand(nseq * S, generator=gen) < 0.15] = -1
        code = code.to(dev)
    out = torch.zeros(nseq * S, H, dtype=F32, device=dev)
    lse = torch.zeros(nseq, nh, S, dtype=F32, device=dev)
    ops.attention_fwd(qkv, code, out, lse, nseq, S, nh)
    delta = torch.zeros(nseq * nh * S, device=dev)
    dqkv = torch.full_like(qkv, float('nan'))
    ops.attention_bwd(qkv, code, out, dout, lse, delta, dqkv, None, nseq, S, nh)
    x = qkv.double().requires_grad_(True)
    xv = x.view(nseq, S, 3, nh, 64)
    sc = torch.einsum('bqhd,bkhd->bhqk', xv[:, :, 0] / 8.0, xv[:, :, 1])
    if code is not None:
        c = code.view(nseq, S)
        allowed = (c[:, :, None] == c[:, None, :]) & (c[:, :, None] >= 0)
        sc = sc + torch.where(allowed, 0.0, -1e10)[:, None].double()
        pad_q = (c < 0)[:, None, :, None]
        # a PAD query's scores all ROUND to -1e10 in fp32 (|q.k / 8| is far below the 1024 ulp): a uniform softmax in the forward -- but the
        # reference's autodiff still sees s = raw + bias, ds / draw = 1, so dS = P (dP - delta) of those rows flows into dq and dk: value 0, slope 1
        sc = torch.where(pad_q, sc - sc.detach(), sc)
    o = torch.einsum('bhqk,bkhd->bqhd', torch.softmax(sc, -1), xv[:, :, 2]).reshape(nseq * S, H)
    o.backward(dout.double())
    assert relerr(dqkv, x.grad) < 2e-5, relerr(dqkv, x.grad)


def test_f32_pool_rows_unitnorm_bwd(dev):
    from merlot_reserve_amd import ops
    H, nh = 128, 2
    q, k, v = rnd((2, H), dev, seed=3), rnd((50, H), dev, seed=4), rnd((50, H), dev, seed=5)
    rows = torch.tensor([[0, 1, 2, 3], [10, 11, 12, 13]], dtype=torch.int32, device=dev)
    do = rnd((2, H), dev, seed=6)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    outs = []
    for gi in range(2):
        kk, vv = kd[rows[gi].long()].view(4, nh, 64), vd[rows[gi].long()].view(4, nh, 64)
        s = torch.einsum('hd,rhd->hr', qd[gi].view(nh, 64) / 8.0, kk)
        outs.append(torch.einsum('hr,rhd->hd', torch.softmax(s, -1), vv).reshape(H))
    torch.stack(outs).backward(do.double())
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    ops.poolattn_bwd(q, k, v, rows, None, do, dq, dk, dv, nh)
    assert relerr(dq, qd.grad) < 1e-5 and relerr(dk, kd.grad) < 1e-5 and relerr(dv, vd.grad) < 1e-5
    # rows_mean backward: every row of a group gets ddst / R
    dsrc = torch.zeros(50, H, device=dev)
    ops.rows_mean_bwd(do, rows, dsrc)
    want = torch.zeros(50, H, dtype=F64, device=dev)
    for gi in range(2):
        want[rows[gi].long()] += do[gi].double() / 4
    assert relerr(dsrc, want) < 1e-6
    # unit_normalize * temperature backward, dx and the temperature's gradient (clipped and not)
    x, dy = rnd((37, H), dev, seed=7), rnd((37, H), dev, seed=8)
    for lsv in (0.3, 5.0):
        ls = torch.tensor([lsv], device=dev)
        xd, lsd = x.double().requires_grad_(True), ls.double().requires_grad_(True)
        y = xd / torch.sqrt((xd ** 2).sum(-1, keepdim=True) + 1e-5) * torch.exp(torch.clamp(lsd, max=math.log(100.0)) / 2)
        y.backward(dy.double())
        dx, dls, part = torch.zeros_like(x), torch.zeros(1, device=dev), torch.zeros(64, device=dev)
        ops.unit_norm_scale_bwd(x, ls, None, dy, dx, dls, part)
        assert relerr(dx, xd.grad) < 2e-6
        assert abs(float(dls) - float(lsd.grad)) <= 2e-6 * max(1.0, abs(float(lsd.grad))), (float(dls), float(lsd.grad))
    # strided row sums (the CLS parameter's gradient) and the column sums (bias gradients)
    z = rnd((3 * 5, H), dev, seed=9)
    o = torch.zeros(H, device=dev)
    ops.sum_rows_strided(z, 3, 5, 0, o)
    assert relerr(o, z.double().view(3, 5, H)[:, 0].sum(0)) < 1e-6
    cs = torch.zeros(H, device=dev)
    ops.colsum(z, cs, None)
    assert relerr(cs, z.double().sum(0)) < 1e-6
