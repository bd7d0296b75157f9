"""The ping-pong GEMM (csrc/gemm3.hip), the one-wave-per-SIMD GEMM (csrc/gemm4.hip) and the few-tile GEMM (csrc/gemm5.hip: 256 x 128 tiles with
two workgroups per CU, 128 x 128 tiles with the deep ring) forced onto every NT problem they can take, at
both tile widths (gemm3: both barrier schedules, one / two phases per k-tile), against a plain PyTorch fp32 reference of the same op: plain products (ragged M and N, one and two
k-tiles, several tiles per workgroup), every fused epilogue of the transformer layers (flax Dense / DenseGeneral, mreserve/modeling.py:
228-236, 252-255), and the grouped weight-gradient (TN) kernel.  By default these shapes would partly run on the one-barrier kernel
(few tiles), so the width is forced with mr_set_option('gemm3', width)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


def rnd(shape, dev, scale=1.0, seed=0):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(BF16).to(dev)


def assert_close(got, ref, rel, name=''):
    got, ref = got.float(), ref.float()
    e = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
    assert math.isfinite(e) and e <= rel, f'{name}: rel-L2 err {e:.3e} > {rel:.1e}'
    tol = 8e-3 * ref.abs() + 4 * rel * ref.abs().mean() + 1e-6        # bf16 output rounding + accumulated noise
    bad = ((got - ref).abs() > tol).float().mean().item()
    assert bad < 1e-3, f'{name}: {bad * 100:.3f}% of elements outside tolerance'


# (tile width, ping-pong phases per k-tile, one-wave-per-SIMD kernel for the epilogues it takes: bias / residual / plain, "gemm5" option:
#  1 = 256 x 128 tiles, two workgroups per CU; 3 = 128 x 128 tiles, deep ring -- every epilogue)
@pytest.fixture(params=[(256, 1, 0, 0), (192, 1, 0, 0), (256, 2, 0, 0), (192, 2, 0, 0), (256, 0, 1, 0), (192, 0, 1, 0), (256, 1, 0, 1), (256, 1, 0, 3)],
                ids=lambda p: f'g5-{"256x128" if p[3] == 1 else "128x128"}' if p[3] else f'w{p[0]}-' + ('g4' if p[2] else f'ph{p[1]}'))
def forced(request):
    from merlot_reserve_amd import _lib
    lib = _lib.load()
    _lib.check(lib.mr_set_option(b'gemm3', request.param[0]), 'mr_set_option')
    _lib.check(lib.mr_set_option(b'gemm3_phases', request.param[1]), 'mr_set_option')
    _lib.check(lib.mr_set_option(b'gemm4', request.param[2]), 'mr_set_option')
    _lib.check(lib.mr_set_option(b'gemm5', request.param[3]), 'mr_set_option')
    _lib.check(lib.mr_set_option(b'gemm_trace', 1), 'mr_set_option')
    yield request.param
    lib.mr_set_option(b'gemm3', 1)
    lib.mr_set_option(b'gemm3_phases', 0)
    lib.mr_set_option(b'gemm4', -1)
    lib.mr_set_option(b'gemm5', -1)
    lib.mr_set_option(b'gemm_trace', 0)


def routed_as_forced(forced, K):
    """The launch went to the kernel the fixture asked for (gemm5 takes K % 64 == 0, K >= 128; shorter K falls to the ping-pong kernel)."""
    from merlot_reserve_amd import _lib
    name = _lib.load().mr_last_gemm_kernel().decode()
    if forced[3] and K % 64 == 0 and K >= 128:
        assert name.startswith('g5::gemm5_kernel') and name.endswith(',128,8>') == (forced[3] == 3), name


@pytest.fixture
def dev():
    return torch.device('cuda:0')


NT_CASES = [
    # M, N, K
    (1024, 256, 128),        # one tile per workgroup at most
    (1312, 3072, 64),        # ONE k-tile per tile
    (2000, 1000, 192),       # ragged M and N, three k-tiles
    (40000, 192, 128),       # narrow: one tile column, 157 row tiles
    (15424 // 2, 2304, 768),
    (9000, 3072, 256),       # > 256 tiles: several tiles per workgroup, XCD-blocked order
    (4616, 768, 3072),       # long k-loop
]


@pytest.mark.parametrize('M,N,K', NT_CASES)
def test_nt_plain(dev, forced, M, N, K):
    from merlot_reserve_amd import ops
    a, b = rnd((M, K), dev, seed=1), rnd((N, K), dev, scale=0.05, seed=2)
    out = torch.full((M + 3, N), float('nan'), dtype=BF16, device=dev)       # three guard rows behind the output
    ops.gemm(a, b, out[:M], transB=True)
    routed_as_forced(forced, K)
    assert_close(out[:M], a.float() @ b.float().T, 3e-3, f'NT {M}x{N}x{K}')
    assert torch.isnan(out[M:].float()).all(), 'rows past M were written'


def test_nt_epilogues(dev, forced):
    from merlot_reserve_amd import _lib, ops
    S, nseq, N, K = 241, 6, 768, 256
    M = S * nseq
    a, w = rnd((M, K), dev, seed=6), rnd((N, K), dev, scale=0.1, seed=7)
    bias = rnd((N,), dev, seed=8)
    ref = a.float() @ w.float().T
    out = torch.zeros(M, N, dtype=BF16, device=dev)
    ops.gemm(a, w, out, transB=True, bias=bias)
    assert_close(out, ref + bias.float(), 3e-3, 'bias')
    tab = torch.rand(S, 32, device=dev) * 2 - 1                    # the "rotary" diagonal scaling of q and k (M:118-136)
    ops.gemm(a, w, out, transB=True, bias=bias, rot_tab=tab, rot_cols=512)
    scale = torch.ones(M, N, device=dev)
    rows = torch.arange(M, device=dev) % S
    for h in range(512 // 64):
        scale[:, h * 64:h * 64 + 32] = tab[rows]
    assert_close(out, (ref + bias.float()) * scale, 3e-3, 'bias + rot')
    pre = torch.zeros(M, N, dtype=BF16, device=dev)
    ops.gemm(a, w, out, transB=True, bias=bias, act=ops.ACT_GELU, c2=pre)
    routed_as_forced(forced, K)
    x = ref + bias.float()
    sg = torch.sigmoid(1.702 * x)
    assert_close(out, x * sg, 4e-3, 'gelu')
    assert_close(pre, sg + 1.702 * x * sg * (1 - sg), 4e-3, 'c2 = gelu grad')
    res = rnd((M, N), dev, seed=9)
    buf = res.clone()
    ops.gemm(a, w, buf, transB=True, residual=buf)
    assert_close(buf, ref.to(BF16).float() + res.float(), 3e-3, 'residual in place')
    aux = rnd((M, N), dev, seed=10)
    nrows = _lib.load().mr_gemm_colsum_rows(M)
    cs = torch.full((nrows, N), float('nan'), device=dev)
    ops.gemm(a, w, out, transB=True, aux=aux, colsum=cs)
    routed_as_forced(forced, K)
    assert_close(out, ref.to(BF16).float() * aux.float(), 4e-3, 'aux multiply')
    assert_close(cs.sum(0), out.float().sum(0), 2e-3, 'column sums of the stored output')
    ops.gemm(a, w, out, transB=True, aux=aux)
    assert_close(out, ref.to(BF16).float() * aux.float(), 4e-3, 'aux multiply, no column sums')
    M2 = 240 * nseq                                               # patch rows into [CLS | 240 patches] groups (M:396-404)
    a2 = rnd((M2, K), dev, seed=11)
    big = torch.zeros(nseq * 241, N, dtype=BF16, device=dev)
    ops.gemm(a2, w, big, transB=True, bias=bias, row_map=(240, 241, 1))
    got = big.reshape(nseq, 241, N)
    assert torch.all(got[:, 0] == 0)
    assert_close(got[:, 1:].reshape(M2, N), a2.float() @ w.float().T + bias.float(), 3e-3, 'row map')


@pytest.mark.parametrize('M,N,K', [(9000, 3072, 256), (15424, 3072, 768), (7000, 2048, 192)])
def test_aux_epilogue_over_several_tiles_per_workgroup(dev, M, N, K):
    """The aux + column-sum epilogue (fc2's dgrad) at shapes with several tiles per persistent workgroup (432 / 732 tiles), where a tile's stores are in
    flight under the next tile's first k-tiles and (round 6) leave as whole 128-byte lines: against the fp32 reference, and BIT FOR BIT between the 256-wide
    (full-line stores) and the 192-wide (half-line stores) instance of the same problem -- same k order per element."""
    from merlot_reserve_amd import _lib, ops
    lib = _lib.load()
    a, w = rnd((M, K), dev, seed=21), rnd((N, K), dev, scale=0.1, seed=22)
    aux = rnd((M, N), dev, seed=23)
    ref = (a.float() @ w.float().T).to(BF16).float() * aux.float()
    outs = []
    try:
        for width in (256, 192):
            lib.mr_set_option(b'gemm3', width)
            lib.mr_set_option(b'gemm_trace', 1)
            cs = torch.full((lib.mr_gemm_colsum_rows(M), N), float('nan'), device=dev)
            out = torch.full((M + 2, N), float('nan'), dtype=BF16, device=dev)
            ops.gemm(a, w, out[:M], transB=True, aux=aux, colsum=cs)
            assert f'gemm3_kernel<{width},4' in lib.mr_last_gemm_kernel().decode()
            assert_close(out[:M], ref, 4e-3, f'aux multiply, {width}-wide')
            assert torch.isnan(out[M:].float()).all(), 'rows past M were written'
            assert_close(cs.sum(0), out[:M].float().sum(0), 2e-3, 'column sums of the stored output')
            outs.append(out[:M].clone())
    finally:
        lib.mr_set_option(b'gemm3', 1)
        lib.mr_set_option(b'gemm_trace', 0)
    assert torch.equal(outs[0], outs[1])


def test_schedules_agree_bit_for_bit(dev):
    """The ping-pong schedules and the one-wave-per-SIMD kernel add a tile's k-blocks in the same order with the same MFMA: identical outputs."""
    from merlot_reserve_amd import _lib, ops
    lib = _lib.load()
    M, N, K = 5000, 1536, 768
    a, w = rnd((M, K), dev, seed=1), rnd((N, K), dev, scale=0.1, seed=2)
    outs = []
    try:
        for width in (256, 192):
            for ph, g4 in ((1, 0), (2, 0), (0, 1)):
                lib.mr_set_option(b'gemm3', width)
                lib.mr_set_option(b'gemm3_phases', ph)
                lib.mr_set_option(b'gemm4', g4)
                o = torch.zeros(M, N, dtype=BF16, device=dev)
                ops.gemm(a, w, o, transB=True)
                outs.append(o)
    finally:
        lib.mr_set_option(b'gemm3', 1)
        lib.mr_set_option(b'gemm3_phases', 0)
        lib.mr_set_option(b'gemm4', -1)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize('Mtok,H,layers', [(15424, 768, 2), (5000, 1024, 2), (4616, 1024, 4)])
def test_tn_grouped_weight_gradients(dev, Mtok, H, layers):
    """The weight gradients of two (base: 216 tiles) or four (large: 768 tiles = three rounds of the 256 CUs) transformer layers in
    ONE launch of the TN ping-pong kernel (dW = X^T . dY)."""
    from merlot_reserve_amd import _lib, ops
    lib = _lib.load()
    shapes = [(4 * H, H), (H, 4 * H), (H, H), (H, 3 * H)] * layers
    xs = [rnd((Mtok, m), dev, seed=20 + i) for i, (m, n) in enumerate(shapes)]
    dys = [rnd((Mtok, n), dev, scale=0.05, seed=40 + i) for i, (m, n) in enumerate(shapes)]
    outs = [torch.full((m, n), float('nan'), dtype=BF16, device=dev) for m, n in shapes]
    ws = torch.zeros(16 << 20, device=dev)
    lib.mr_set_option(b'gemm3', 256)          # (forced: the grouped launch must take all of them, whatever the tile count)
    try:
        ops.gemm_grouped([ops.gemm_args(x, dy, o, transA=True, ws=ws) for x, dy, o in zip(xs, dys, outs)])
    finally:
        lib.mr_set_option(b'gemm3', 1)
    for x, dy, o in zip(xs, dys, outs):
        assert_close(o, x.float().T @ dy.float(), 3e-3, f'wgrad {tuple(o.shape)}')


@pytest.mark.parametrize('cus', [240, 64])
def test_fewer_persistent_workgroups_same_result(dev, cus):
    """mr_set_option('gemm_cus', n): the persistent grids use n workgroups (n / 8 per XCD) -- what the data-parallel trainer asks for while a
    gradient bucket's RCCL kernel holds CUs.  Same tiles, same k order: bit-identical outputs, with and without the XCD partition."""
    from merlot_reserve_amd import _lib, ops
    lib = _lib.load()
    cases = [(15424, 3072, 768, True), (9000, 768, 1024, False), (3000, 1000, 256, False)]       # (M, N, K, bias): >= 2 rounds | 1-2 rounds | few tiles
    try:
        for M, N, K, with_bias in cases:
            a, w = rnd((M, K), dev, seed=1), rnd((N, K), dev, scale=0.1, seed=2)
            bias = rnd((N,), dev, seed=3) if with_bias else None
            outs = []
            for g4 in (0, 1):
                for n in (0, cus):
                    lib.mr_set_option(b'gemm3', 256 if M < 4000 else 1)          # the small case: forced onto the kernel
                    lib.mr_set_option(b'gemm4', g4)
                    lib.mr_set_option(b'gemm_cus', n)
                    o = torch.full((M, N), float('nan'), dtype=BF16, device=dev)
                    ops.gemm(a, w, o, transB=True, bias=bias)
                    outs.append(o)
            ref = a.float() @ w.float().T + (bias.float() if with_bias else 0)
            assert_close(outs[0], ref, 3e-3, f'{M}x{N}x{K}')
            for o in outs[1:]:
                assert torch.equal(o, outs[0]), f'{M}x{N}x{K}: outputs differ with gemm_cus = {cus}'
    finally:
        lib.mr_set_option(b'gemm3', 1)
        lib.mr_set_option(b'gemm4', -1)
        lib.mr_set_option(b'gemm_cus', 0)
