"""Kernels must give the same bits with and without another queue's MFMA waves on their SIMDs.

Round 3 found LayerNorm rows that changed from run to run while a second stream was active; round 4 reduced it to this (scripts/slp_repro.py): code
the SLP vectoriser packed into v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with op_sel / neg modifiers returns wrong values in lanes 48-63 of a wave WHILE
waves of an MFMA kernel (the 128 x 128 small-GEMM kernel: few, small workgroups) share its SIMD -- never alone, never beside whole-CU GEMMs, never in a
scalar (-fno-slp-vectorize) build.  Round 5: a second neighbour, the attention kernels' MFMA waves, alternates with it.  The HBM-bound files are built scalar (merlot_reserve_amd/build.py); the GEMM files keep packed epilogues (3.2 ms of
the step), so this test holds EVERY kernel family with vector epilogue or row code against its own lone launch, bit for bit, under that neighbour."""
import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


def _neighbour(dev, gen):
    from merlot_reserve_amd import ops
    a = torch.randn(5760, 136, generator=gen).to(BF16).to(dev)
    w = (torch.randn(136, 768, generator=gen) * 0.1).to(BF16).to(dev)
    o = torch.zeros(5760, 768, dtype=BF16, device=dev)
    a2 = torch.randn(192, 768, generator=gen).to(BF16).to(dev)
    w2 = (torch.randn(768, 768, generator=gen) * 0.05).to(BF16).to(dev)
    o2 = torch.zeros(192, 768, dtype=BF16, device=dev)

    def run():
        for _ in range(4):
            ops.gemm(a, w, o)                       # the audio tower's conv-as-GEMM shape: the small 128 x 128 MFMA kernel
            ops.gemm(a2, w2, o2, transB=True)
    return run


def _neighbour_attention(dev, gen):
    """Second neighbour (round 5): the attention kernels' MFMA waves -- 256-thread workgroups at two or three per CU that leave wave slots on every
    SIMD free, so the kernel under test really shares SIMDs with them (forward, and the short-sequence backward: one wave per (sequence, head))."""
    from merlot_reserve_amd import ops
    nseq, S, nh = 48, 31, 12
    H = nh * 64
    qkv = torch.randn(nseq * S, 3 * H, generator=gen).to(BF16).to(dev)
    dout = torch.randn(nseq * S, H, generator=gen).to(BF16).to(dev)
    out = torch.zeros(nseq * S, H, dtype=BF16, device=dev)
    lse, delta = torch.zeros(nseq, nh, S, device=dev), torch.zeros(nseq, nh, S, device=dev)
    dqkv = torch.zeros_like(qkv)
    qkv2 = torch.randn(8 * 241, 3 * H, generator=gen).to(BF16).to(dev)
    out2 = torch.zeros(8 * 241, H, dtype=BF16, device=dev)
    lse2 = torch.zeros(8, nh, 241, device=dev)

    def run():
        for _ in range(3):
            ops.attention_fwd(qkv, None, out, lse, nseq, S, nh)
            ops.attention_bwd(qkv, None, out, dout, lse, delta, dqkv, None, nseq, S, nh)
            ops.attention_fwd(qkv2, None, out2, lse2, 8, 241, nh)
    return run


def _hold(dev, launch, outputs, reps=12, per_rep=8):
    """launch(i) writes outputs[i]; the lone launches are the reference; then `reps` times: the neighbour loops on a second stream while the launches
    are repeated on the first; returns the number of launches whose output differed from the lone one."""
    gen = torch.Generator().manual_seed(5)
    nbs = [_neighbour(dev, gen), _neighbour_attention(dev, gen)]
    for i in range(per_rep):
        launch(i)
    torch.cuda.synchronize()
    refs = [[t.clone() for t in outputs[i]] for i in range(per_rep)]
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = 0
    for rep in range(reps):
        nb = nbs[rep % 2]                           # the two neighbours take turns
        for i in range(per_rep):
            for t in outputs[i]:
                t.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(sb):
            for _ in range(10):
                nb()
        with torch.cuda.stream(sa):
            for i in range(per_rep):
                launch(i)
        torch.cuda.synchronize()
        for i in range(per_rep):
            bad += int(any(not torch.equal(t, r) for t, r in zip(outputs[i], refs[i])))
    return bad


def test_layernorm_rows_beside_mfma_waves(dev):
    from merlot_reserve_amd import ops
    g = torch.Generator().manual_seed(0)
    rows, H = 15424, 768
    x = (torch.randn(rows, H, generator=g) * 2 + 0.3).to(BF16).to(dev)
    gam, bet = (torch.randn(H, generator=g) * 0.2 + 1).to(BF16).to(dev), (torch.randn(H, generator=g) * 0.1).to(BF16).to(dev)
    dy, add = torch.randn(rows, H, generator=g).to(BF16).to(dev), torch.randn(rows, H, generator=g).to(BF16).to(dev)
    n = 8
    outs = [(torch.zeros_like(x), torch.zeros_like(x), torch.zeros_like(x)) for _ in range(n)]
    mean, rstd = torch.zeros(rows, device=dev), torch.zeros(rows, device=dev)
    mean2, rstd2 = torch.zeros(rows, device=dev), torch.zeros(rows, device=dev)
    dg, db = torch.zeros(H, dtype=BF16, device=dev), torch.zeros(H, dtype=BF16, device=dev)
    ws = ops.layernorm_bwd_workspace(H, dev)

    def launch(i):
        y1, y2, dx = outs[i]
        ops.layernorm_fwd(x, gam, bet, y1, mean, rstd)
        ops.layernorm_fwd(y1, gam, bet, y2, mean2, rstd2)           # the step's opening: pre_ln, then layer 0's ln1 on its output
        ops.layernorm_bwd(dy, x, gam, mean, rstd, dx, dg, db, ws, dx_add=add)
    assert _hold(dev, launch, outs) == 0


@pytest.mark.parametrize('kernel', ['small', 'gemm5_256', 'gemm5_128', 'pingpong', 'onebarrier'])
def test_gemm_epilogues_beside_mfma_waves(dev, kernel):
    """bias + GELU + gelu' copy, x aux, bias + "rotary" scales: the vector-heavy epilogues, on every GEMM kernel family."""
    from merlot_reserve_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    M = {'small': 192, 'gemm5_256': 5952, 'gemm5_128': 3072, 'pingpong': 15424, 'onebarrier': 5952}[kernel]
    N, K = (768, 768) if kernel != 'pingpong' else (3072, 768)
    a = torch.randn(M, K, generator=g).to(BF16).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(BF16).to(dev)
    bias = torch.randn(N, generator=g).to(BF16).to(dev)
    aux = torch.randn(M, N, generator=g).to(BF16).to(dev)
    tab = (torch.rand(31, 32, generator=g) * 2 - 1).to(dev)
    n = 8
    outs = [tuple(torch.zeros(M, N, dtype=BF16, device=dev) for _ in range(4)) for _ in range(n)]
    opts = {'small': {}, 'gemm5_256': {'gemm5': 1}, 'gemm5_128': {'gemm5': 3}, 'pingpong': {'gemm5': 0}, 'onebarrier': {'gemm5': 0, 'gemm3': 0}}[kernel]
    prev = {k: ops.get_option(k) for k in opts}

    def launch(i):
        o, c2, o2, o3 = outs[i]
        ops.gemm(a, w, o, transB=True, bias=bias, act=ops.ACT_GELU, c2=c2)
        ops.gemm(a, w, o2, transB=True, aux=aux)
        ops.gemm(a, w, o3, transB=True, bias=bias, rot_tab=tab, rot_cols=512)
    try:
        for k, v in opts.items():
            ops.set_option(k, v)
        ops.set_option('gemm_trace', 1)
        launch(0)
        name = lib.mr_last_gemm_kernel().decode()
        want = {'small': 'gemm_bf16', 'gemm5_256': 'g5::gemm5_kernel<1,4>', 'gemm5_128': ',128,8>', 'pingpong': 'g3::gemm3_kernel', 'onebarrier': 'g256::gemm256'}[kernel]
        assert want in name, name
        assert _hold(dev, launch, outs) == 0
    finally:
        for k, v in prev.items():
            ops.set_option(k, v)
        ops.set_option('gemm_trace', 0)


def test_attention_and_row_kernels_beside_mfma_waves(dev):
    """The kernels whose small workgroups share CUs with other queues in the real step -- short-sequence attention forward / backward, attention pooling,
    gathers, unit-norm, the fused Adam -- each against its lone launch under the same neighbour."""
    from merlot_reserve_amd import ops
    g = torch.Generator().manual_seed(2)
    nseq, S, nh = 192, 31, 12
    H = nh * 64
    qkv = torch.randn(nseq * S, 3 * H, generator=g).to(BF16).to(dev)
    dout = torch.randn(nseq * S, H, generator=g).to(BF16).to(dev)
    tab = (torch.rand(S, 32, generator=g) * 2 - 1).to(dev)
    Gp = nseq * 6
    rows = (torch.arange(nseq)[:, None, None] * S + 1 + torch.arange(6)[None, :, None] * 5 + torch.arange(5)[None, None]).reshape(Gp, 5).to(torch.int32).to(dev)
    q = torch.randn(Gp, H, generator=g).to(BF16).to(dev)
    idxp = torch.arange(0, nseq * S + 1, dtype=torch.int32, device=dev)
    idx = torch.randperm(nseq * S, generator=g).to(torch.int32).to(dev)
    npar = 8 * 2048 * 64
    master0 = torch.randn(npar, generator=g).to(dev)
    grad = (torch.randn(npar, generator=g) * 1e-2).to(BF16).to(dev)
    flags = torch.ones(npar // 2048, dtype=torch.uint8, device=dev)
    hyper = torch.tensor([1.0, -1e-3, 1.0, 1.0], device=dev)
    n = 8

    def bufs():
        z = lambda *s: torch.zeros(*s, dtype=BF16, device=dev)
        f = lambda *s: torch.zeros(*s, dtype=F32, device=dev)
        return dict(out=z(nseq * S, H), lse=f(nseq * nh * S), delta=f(nseq * nh * S), dqkv=z(nseq * S, 3 * H), po=z(Gp, H), probs=f(Gp, nh, 5),
                    gath=z(nseq * S, H), un=z(nseq * S, H), inv=f(nseq * S), master=master0.clone(), work=z(npar), mu=z(npar), nu=z(npar))
    B = [bufs() for _ in range(n)]
    ls = torch.tensor([0.3], dtype=BF16, device=dev)

    def launch(i):
        b = B[i]
        ops.attention_fwd(qkv, None, b['out'], b['lse'], nseq, S, nh)
        ops.attention_bwd(qkv, None, b['out'], dout, b['lse'], b['delta'], b['dqkv'], tab, nseq, S, nh)
        ops.poolattn_fwd(q, qkv[:, H:2 * H], qkv[:, 2 * H:], rows, b['po'], b['probs'], nh)
        ops.segment_sum([dout], idxp, idx, b['gath'])
        ops.unit_norm_scale_fwd(dout, ls, b['un'], b['inv'])
        b['master'].copy_(master0); b['mu'].zero_(); b['nu'].zero_()
        ops.adam_bf16_update_dev(b['master'], b['work'], grad, b['mu'], b['nu'], None, flags, 0.9, 0.98, 1e-6, 0.1, hyper)
    outs = [tuple(B[i][k] for k in ('out', 'lse', 'dqkv', 'po', 'probs', 'gath', 'un', 'inv', 'master', 'work', 'mu', 'nu')) for i in range(n)]
    assert _hold(dev, launch, outs, reps=8) == 0
