"""Thin torch-tensor wrappers over the C-ABI (include/mreserve_hip.h).  torch is plumbing here: device memory and the
current HIP stream.  Every op launches asynchronously on torch's current stream; nothing allocates unless an output
tensor is not passed in.  All 2-D operands must have unit stride in the last dim; the row stride is the leading dim."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import GemmArgs, check

BF16, F32 = torch.bfloat16, torch.float32
MR_DT_BF16, MR_DT_F32 = 0, 1
ACT_NONE, ACT_GELU = 0, 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _ptr(t):
    return None if t is None else t.data_ptr()


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, f'need a row-major 2-D view, got {tuple(t.shape)} / {t.stride()}'
    return t.stride(0)


# Optional per-launch timing of the dominant kernel (bench.py): a list that receives (start_event, end_event, flops, shape tag,
# kernel name -- '' unless option 'gemm_trace' is on) for every mr_gemm launch, recorded with HIP events on the stream the kernel
# is launched on.
GEMM_PROFILE = None
# Optional per-launch timing of EVERY op by kernel family (bench.py's breakdown): a dict family -> list of (start, end)
# HIP events recorded on the stream the op is launched on.
FAMILY_PROFILE = None


def _timed(family):
    def deco(fn):
        def wrapper(*a, **k):
            if FAMILY_PROFILE is None or (family == 'gemm' and GEMM_PROFILE is not None):     # GEMMs: timed once, by GEMM_PROFILE
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = fn(*a, **k)
            e1.record()
            FAMILY_PROFILE.setdefault(family, []).append((e0, e1))
            return r
        wrapper.__name__, wrapper.__doc__ = fn.__name__, fn.__doc__
        return wrapper
    return deco


def set_option(name, value):
    """mr_set_option: an option of the calling thread's current handle, or of the process defaults when it has none
    (include/mreserve_hip.h lists the names)."""
    check(_lib.load().mr_set_option(name.encode(), int(value)), 'mr_set_option')


def get_option(name):
    v = C.c_int32(0)
    check(_lib.load().mr_get_option(name.encode(), C.byref(v)), 'mr_get_option')
    return v.value


class Handle:
    """mr_create / mr_destroy (include/mreserve_hip.h): an option set of its own (a copy of the process defaults at creation) and,
    with ws_bytes > 0, a split-K workspace on `device` that mr_gemm uses when the caller passes none.  `with handle:` makes it the
    calling THREAD's current handle for the block (launches inside run under its options) and restores the previous one after."""

    def __init__(self, device=0, ws_bytes=0):
        h = C.c_void_p()
        check(_lib.load().mr_create(int(device), int(ws_bytes), C.byref(h)), 'mr_create')
        self._h, self._prev = h, []

    def set_option(self, name, value):
        check(_lib.load().mr_handle_set_option(self._h, name.encode(), int(value)), 'mr_handle_set_option')

    def get_option(self, name):
        v = C.c_int32(0)
        check(_lib.load().mr_handle_get_option(self._h, name.encode(), C.byref(v)), 'mr_handle_get_option')
        return v.value

    def __enter__(self):
        lib = _lib.load()
        self._prev.append(lib.mr_get_current())
        check(lib.mr_make_current(self._h), 'mr_make_current')
        return self

    def __exit__(self, *exc):
        check(_lib.load().mr_make_current(self._prev.pop()), 'mr_make_current')
        return False

    def close(self):
        if self._h is not None:
            check(_lib.load().mr_destroy(self._h), 'mr_destroy')
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class gemm_cus:
    """with ops.gemm_cus(world): the persistent forward / dgrad GEMM grids launched inside use MR_COMM_GEMM_CUS workgroups (default 240 =
    30 per XCD) when world > 1, i.e. while a gradient bucket's RCCL kernel may hold CUs next to them: a 256-workgroup grid then runs its
    last workgroups as a second round (scripts/bench_contention.py: 16 CUs held -> 111 vs 79 us for the fc1 shape).  Launch parameters
    are fixed at capture time, so inside a captured step this shapes the graph's kernels, not the replay.  The option belongs to the
    calling thread's current handle (or the process defaults) and is put back to the value it HAD on exit, also on an exception."""

    def __init__(self, world):
        n = int(os.environ.get('MR_COMM_GEMM_CUS', '240'))
        self.n = n if world > 1 and 64 <= n < 256 else 0
        self.prev = None

    def __enter__(self):
        if self.n:
            self.prev = get_option('gemm_cus')
            set_option('gemm_cus', self.n)
        return self

    def __exit__(self, *exc):
        if self.n:
            set_option('gemm_cus', self.prev)
        return False


def gemm_args(a, b, out, transA=False, transB=False, bias=None, rot_tab=None, rot_cols=0, c2=None, act=ACT_NONE,
              residual=None, aux=None, row_map=None, ws=None, colsum=None):
    """Builds the mr_gemm_args of out[M,N] = op(a) @ op(b) (see gemm()).  ws: the CALLER's fp32 scratch for split-K
    partials (None disables split-K); it is per engine and per stream -- there is no module-level workspace, so two
    engines (or a captured graph and a later engine) can never share or outlive each other's scratch."""
    M, K = (a.shape[1], a.shape[0]) if transA else (a.shape[0], a.shape[1])
    Kb, N = (b.shape[1], b.shape[0]) if transB else (b.shape[0], b.shape[1])
    assert K == Kb, f'gemm: K mismatch {K} vs {Kb}'
    assert a.dtype == b.dtype and a.dtype in (BF16, F32)
    if a.dtype == F32:       # fp32 path (mr_f32_gemm): every operand fp32
        assert out.dtype == F32 and colsum is None
        assert all(t is None or t.dtype == F32 for t in (bias, residual, aux, c2))
    g = GemmArgs()
    g._f32_operands = a.dtype == F32
    g.M, g.N, g.K = M, N, K
    g.A, g.lda, g.transA = a.data_ptr(), _ld(a), int(transA)
    g.B, g.ldb, g.transB = b.data_ptr(), _ld(b), int(transB)
    g.C, g.ldc = out.data_ptr(), _ld(out)
    g.c_dtype = MR_DT_F32 if out.dtype == F32 else MR_DT_BF16
    g.bias = _ptr(bias)
    if rot_tab is not None:
        assert rot_tab.dtype == F32 and rot_tab.is_contiguous() and rot_tab.shape[-1] == 32
        g.rot_tab, g.rot_rows, g.rot_cols = rot_tab.data_ptr(), rot_tab.numel() // 32, rot_cols
    else:
        g.rot_tab, g.rot_rows, g.rot_cols = None, 0, 0
    g.c2 = _ptr(c2)
    if c2 is not None:
        assert _ld(c2) == g.ldc
    g.act = act
    g.residual, g.ldr = (_ptr(residual), _ld(residual)) if residual is not None else (None, 0)
    g.aux, g.ldaux = (_ptr(aux), _ld(aux)) if aux is not None else (None, 0)
    if row_map is not None:
        g.out_grp, g.out_grp_stride, g.out_grp_off = row_map
        assert out.shape[0] >= ((M - 1) // row_map[0]) * row_map[1] + row_map[2] + (M - 1) % row_map[0] + 1
    else:
        g.out_grp = g.out_grp_stride = g.out_grp_off = 0
        assert out.shape[0] >= M and out.shape[1] >= N
    if colsum is not None:       # fp32 [mr_gemm_colsum_rows(M), >= N]: per-(tile, wave) column sums of the stored output
        assert colsum.dtype == F32 and colsum.shape[0] >= _lib.load().mr_gemm_colsum_rows(M) and colsum.shape[1] >= N
        g.colsum, g.ldcs = colsum.data_ptr(), _ld(colsum)
    else:
        g.colsum, g.ldcs = None, 0
    if ws is not None:
        assert ws.dtype == F32 and ws.device == a.device
        g.workspace, g.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    else:
        g.workspace, g.workspace_bytes = None, 0
    return g


def gemm_bytes(g):
    """Algorithmic HBM bytes of one GEMM launch as issued: both operands once, the output once, every epilogue operand / second output once."""
    mn = g.M * g.N
    return (2 * (g.M * g.K + g.N * g.K) + (4 if g.c_dtype == MR_DT_F32 else 2) * mn + (2 * mn if g.c2 else 0) + (2 * mn if g.residual else 0) +
            (2 * mn if g.aux else 0) + (2 * g.N if g.bias else 0))


@_timed('gemm')
def gemm(a, b, out, **kw):
    """out[M,N] = op(a) @ op(b) with the fused epilogue of mr_gemm.  a: [M,K] (or [K,M] if transA); b: [K,N]
    (or [N,K] if transB).  row_map = (grp, grp_stride, grp_off) remaps output rows (out must be big enough)."""
    lib = _lib.load()
    g = gemm_args(a, b, out, **kw)
    if a.dtype == F32:
        g.workspace, g.workspace_bytes = None, 0
        check(lib.mr_f32_gemm(C.byref(g), _stream()), 'mr_f32_gemm')
        return out
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.mr_gemm(C.byref(g), _stream()), 'mr_gemm')
        e1.record()
        GEMM_PROFILE.append((e0, e1, 2.0 * g.M * g.N * g.K, (g.M, g.N, g.K, g.transA, g.transB, bool(g.bias), bool(g.rot_tab), bool(g.c2), g.act, bool(g.residual), bool(g.aux)),
                             lib.mr_last_gemm_kernel().decode(), gemm_bytes(g)))
        return out
    check(lib.mr_gemm(C.byref(g), _stream()), 'mr_gemm')
    return out


@_timed('gemm')
def gemm_grouped(arg_list):
    """Several independent GEMMs (built with gemm_args) in one persistent launch when they qualify."""
    lib = _lib.load()
    if arg_list and arg_list[0].c_dtype == MR_DT_F32 and getattr(arg_list[0], '_f32_operands', False):     # fp32 program: no grouped kernel
        for g in arg_list:
            g.workspace, g.workspace_bytes = None, 0
            check(lib.mr_f32_gemm(C.byref(g), _stream()), 'mr_f32_gemm')
        return
    arr = (GemmArgs * len(arg_list))(*arg_list)
    if GEMM_PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib.mr_gemm_grouped(arr, len(arg_list), _stream()), 'mr_gemm_grouped')
        e1.record()
        GEMM_PROFILE.append((e0, e1, sum(2.0 * g.M * g.N * g.K for g in arg_list), ('grouped', len(arg_list), arg_list[0].K), lib.mr_last_gemm_kernel().decode(),
                             sum(gemm_bytes(g) for g in arg_list)))
        return
    check(lib.mr_gemm_grouped(arr, len(arg_list), _stream()), 'mr_gemm_grouped')


def gemm_colsum_job(a, b, out, colsum_ws, bias_grad, jobs, **kw):
    """out = op(a) @ op(b) * aux with the bias gradient colsum(out) folded into the GEMM's epilogue when the library supports it
    for this problem (per-tile partial rows in colsum_ws, reduced later with the other deferred jobs); otherwise the GEMM
    and a separate column-sum pass over `out`.  Returns True when fused."""
    lib = _lib.load()
    if a.dtype == F32:
        gemm(a, b, out, **kw)
        return False
    M = a.shape[1] if kw.get('transA') else a.shape[0]
    N = out.shape[1]
    rows = lib.mr_gemm_colsum_rows(M)
    cs = colsum_ws[:rows * N].view(rows, N) if colsum_ws.numel() >= rows * N else None
    if cs is not None and jobs is not None:
        g = gemm_args(a, b, out, colsum=cs, **kw)
        if lib.mr_gemm_colsum_supported(C.byref(g)):
            gemm(a, b, out, colsum=cs, **kw)
            jobs.append(_lib.ReduceJob(cs.data_ptr(), rows, N, N, bias_grad.data_ptr(), bias_grad.data_ptr()))
            return True
    gemm(a, b, out, **kw)
    return False


@_timed('layernorm+reductions')
def layernorm_fwd(x, gamma, beta, y, mean=None, rstd=None, eps=1e-5):
    rows, H = x.shape
    if x.dtype == F32:
        check(_lib.load().mr_f32_layernorm_fwd(x.data_ptr(), _ld(x), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ld(y),
                                               rows, H, eps, _stream()), 'mr_f32_layernorm_fwd')
        return y
    check(_lib.load().mr_layernorm_fwd(x.data_ptr(), _ld(x), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _ld(y),
                                       _ptr(mean), _ptr(rstd), rows, H, eps, _stream()), 'mr_layernorm_fwd')
    return y


def layernorm_bwd_workspace(H, device):
    return torch.empty(_lib.load().mr_layernorm_bwd_workspace(H) // 4, dtype=F32, device=device)


@_timed('layernorm+reductions')
def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, partials, add_to_dx=False, dx_add=None, jobs=None):
    """dx = LN backward (+ dx_add; add_to_dx=True is shorthand for dx_add=dx).  With `jobs` (a list) the dgamma/dbeta
    reduction is deferred: a job is appended for reduce_partials() and `partials` must stay untouched until then."""
    rows, H = x.shape
    if add_to_dx:
        dx_add = dx
    lib = _lib.load()
    if x.dtype == F32:       # fp32 program: statistics recomputed, the parameter gradients reduced at once (`partials`: >= 2 * rows floats)
        assert partials.numel() >= 2 * rows
        check(lib.mr_f32_layernorm_bwd(dy.data_ptr(), _ld(dy), x.data_ptr(), _ld(x), gamma.data_ptr(), dx.data_ptr(), _ld(dx), _ptr(dx_add),
                                       0 if dx_add is None else _ld(dx_add), dgamma.data_ptr(), dbeta.data_ptr(), partials.data_ptr(), rows, H,
                                       1e-5, _stream()), 'mr_f32_layernorm_bwd')
        return dx
    defer = jobs is not None
    check(lib.mr_layernorm_bwd(dy.data_ptr(), _ld(dy), x.data_ptr(), _ld(x), gamma.data_ptr(), mean.data_ptr(),
                               rstd.data_ptr(), dx.data_ptr(), _ld(dx), _ptr(dx_add), 0 if dx_add is None else _ld(dx_add),
                               None if defer else dgamma.data_ptr(), None if defer else dbeta.data_ptr(), partials.data_ptr(),
                               rows, H, _stream()), 'mr_layernorm_bwd')
    if defer:
        jobs.append(_lib.ReduceJob(partials.data_ptr(), lib.mr_layernorm_bwd_nparts(rows), 2 * H, H, dgamma.data_ptr(), dbeta.data_ptr()))
    return dx


def colsum_workspace(N, device):
    return torch.empty(_lib.load().mr_colsum_workspace(N) // 4, dtype=F32, device=device)


@_timed('layernorm+reductions')
def colsum(x, out, partials, jobs=None):
    """out[n] = sum_m x[m, n]; with `jobs` the final reduction is deferred to reduce_partials() (see layernorm_bwd)."""
    rows, N = x.shape
    lib = _lib.load()
    if x.dtype == F32:
        check(lib.mr_f32_colsum(x.data_ptr(), _ld(x), rows, N, out.data_ptr(), _stream()), 'mr_f32_colsum')
        return out
    check(lib.mr_colsum(x.data_ptr(), _ld(x), rows, N, None if jobs is not None else out.data_ptr(), partials.data_ptr(), _stream()),
          'mr_colsum')
    if jobs is not None:
        jobs.append(_lib.ReduceJob(partials.data_ptr(), lib.mr_colsum_nparts(rows), N, N, out.data_ptr(), out.data_ptr()))
    return out


@_timed('layernorm+reductions')
def reduce_partials(jobs):
    """Runs the deferred column reductions (<= 16 per launch) and clears the list."""
    while jobs:
        chunk = jobs[:16]
        del jobs[:16]
        arr = (_lib.ReduceJob * len(chunk))(*chunk)
        check(_lib.load().mr_reduce_partials(arr, len(chunk), _stream()), 'mr_reduce_partials')


@_timed('attention')
def attention_fwd(qkv, code, out, lse, nseq, S, nh):
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.shape == (nseq * S, 3 * nh * 64)
    if qkv.dtype == F32:
        check(_lib.load().mr_f32_attention_fwd(qkv.data_ptr(), _ptr(code), out.data_ptr(), _ptr(lse), nseq, S, nh, _stream()),
              'mr_f32_attention_fwd')
        return out
    check(_lib.load().mr_attention_fwd(qkv.data_ptr(), _ptr(code), out.data_ptr(), lse.data_ptr(), nseq, S, nh, _stream()),
          'mr_attention_fwd')
    return out


@_timed('attention')
def attention_fwd_dense_mask(qkv, mask, out, nseq, S, nh):
    """Arbitrary boolean attention mask [nseq, S, S] (uint8, != 0 = allowed): forward only, bf16 or fp32."""
    assert qkv.is_contiguous() and out.is_contiguous() and qkv.shape == (nseq * S, 3 * nh * 64) and qkv.dtype in (BF16, F32)
    assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == nseq * S * S and mask.device == qkv.device
    check(_lib.load().mr_attention_fwd_dense_mask(qkv.data_ptr(), MR_DT_F32 if qkv.dtype == F32 else MR_DT_BF16, mask.data_ptr(), out.data_ptr(),
                                                  nseq, S, nh, _stream()), 'mr_attention_fwd_dense_mask')
    return out


@_timed('attention')
def attention_bwd_dense_mask(qkv, mask, dout, dqkv, rot_tab, nseq, S, nh, ws=None):
    """Backward of attention_fwd_dense_mask (any boolean mask [nseq, S, S]; mreserve/modeling.py:343-358): dqkv [nseq*S, 3H], every element written;
    the "rotary" scales applied to dq / dk when rot_tab is given.  ws: fp32 workspace of mr_attention_bwd_dense_mask_workspace bytes (allocated if None)."""
    assert qkv.is_contiguous() and dout.is_contiguous() and dqkv.is_contiguous() and qkv.shape == (nseq * S, 3 * nh * 64) and qkv.dtype in (BF16, F32)
    assert dout.shape == (nseq * S, nh * 64) and dqkv.shape == qkv.shape and dout.dtype == qkv.dtype == dqkv.dtype
    assert mask.dtype == torch.uint8 and mask.is_contiguous() and mask.numel() == nseq * S * S and mask.device == qkv.device
    lib = _lib.load()
    need = lib.mr_attention_bwd_dense_mask_workspace(nseq, S, nh) // 4
    if ws is None:
        ws = torch.empty(need, dtype=F32, device=qkv.device)
    assert ws.dtype == F32 and ws.numel() >= need
    if rot_tab is not None:
        assert rot_tab.dtype == F32 and rot_tab.is_contiguous() and rot_tab.shape[-1] == 32
    check(lib.mr_attention_bwd_dense_mask(qkv.data_ptr(), MR_DT_F32 if qkv.dtype == F32 else MR_DT_BF16, mask.data_ptr(), dout.data_ptr(), dqkv.data_ptr(),
                                          _ptr(rot_tab), 0 if rot_tab is None else rot_tab.numel() // 32, ws.data_ptr(), nseq, S, nh, _stream()),
          'mr_attention_bwd_dense_mask')
    return dqkv


@_timed('attention')
def attention_bwd(qkv, code, out, dout, lse, delta, dqkv, rot_tab, nseq, S, nh, colsum_ws=None, bias_grad=None, jobs=None):
    """With colsum_ws / bias_grad / jobs: the qkv bias gradient (column sums of dqkv) comes out of the backward kernels as
    per-block partial rows in colsum_ws, reduced later with the layer's other deferred jobs (no pass over dqkv)."""
    assert qkv.is_contiguous() and out.is_contiguous() and dout.is_contiguous() and dqkv.is_contiguous()
    rr = 0 if rot_tab is None else rot_tab.numel() // 32
    lib = _lib.load()
    if qkv.dtype == F32:
        check(lib.mr_f32_attention_bwd(qkv.data_ptr(), _ptr(code), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                                       dqkv.data_ptr(), _ptr(rot_tab), rr, nseq, S, nh, _stream()), 'mr_f32_attention_bwd')
        if bias_grad is not None and jobs is not None:
            check(lib.mr_f32_colsum(dqkv.data_ptr(), _ld(dqkv), dqkv.shape[0], dqkv.shape[1], bias_grad.data_ptr(), _stream()), 'mr_f32_colsum')
        return dqkv
    cs = None
    if jobs is not None and colsum_ws is not None:
        rows, n3 = lib.mr_attention_bwd_colsum_rows(nseq, S), dqkv.shape[1]
        assert colsum_ws.dtype == F32 and colsum_ws.numel() >= rows * n3
        cs = colsum_ws[:rows * n3].view(rows, n3)
    check(lib.mr_attention_bwd(qkv.data_ptr(), _ptr(code), out.data_ptr(), dout.data_ptr(), lse.data_ptr(),
                               delta.data_ptr(), dqkv.data_ptr(), _ptr(rot_tab), rr, _ptr(cs), nseq, S, nh, _stream()),
          'mr_attention_bwd')
    if cs is not None:
        jobs.append(_lib.ReduceJob(cs.data_ptr(), cs.shape[0], cs.shape[1], cs.shape[1], bias_grad.data_ptr(), bias_grad.data_ptr()))
    return dqkv


@_timed('rowops')
def poolattn_fwd(q, k, v, key_rows, out, probs, nh):
    G, R = key_rows.shape
    assert _ld(k) == _ld(v)
    if q.dtype == F32:
        check(_lib.load().mr_f32_poolattn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), _ld(k), key_rows.data_ptr(), out.data_ptr(),
                                              G, R, nh, _stream()), 'mr_f32_poolattn_fwd')
        return out
    check(_lib.load().mr_poolattn_fwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), _ld(k), key_rows.data_ptr(), out.data_ptr(),
                                      probs.data_ptr(), G, R, nh, _stream()), 'mr_poolattn_fwd')
    return out


@_timed('rowops')
def poolattn_bwd(q, k, v, key_rows, probs, dout, dq, dk, dv, nh):
    G, R = key_rows.shape
    assert _ld(k) == _ld(v) == _ld(dk) == _ld(dv)
    if q.dtype == F32:
        check(_lib.load().mr_f32_poolattn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), _ld(k), key_rows.data_ptr(), dout.data_ptr(),
                                              dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), G, R, nh, _stream()), 'mr_f32_poolattn_bwd')
        return
    check(_lib.load().mr_poolattn_bwd(q.data_ptr(), k.data_ptr(), v.data_ptr(), _ld(k), key_rows.data_ptr(), probs.data_ptr(),
                                      dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), G, R, nh, _stream()),
          'mr_poolattn_bwd')


@_timed('rowops')
def segment_sum(srcs, indptr, indices, dst, scale=1.0, accumulate=False):
    """srcs: list of 1..3 row tables; indices address their row-wise concatenation."""
    s = list(srcs) + [None] * (3 - len(srcs))
    n = [0 if t is None else t.shape[0] for t in s]
    ld = [0 if t is None else _ld(t) for t in s]
    n_dst, H = dst.shape
    assert indptr.numel() == n_dst + 1 and indptr.dtype == torch.int32 and indices.dtype == torch.int32
    if srcs[0].dtype == F32:
        assert dst.dtype == F32 and not accumulate
        check(_lib.load().mr_f32_segment_sum(_ptr(s[0]), ld[0], n[0], _ptr(s[1]), ld[1], n[1], _ptr(s[2]), ld[2], n[2],
                                             indptr.data_ptr(), indices.data_ptr(), dst.data_ptr(), _ld(dst), n_dst, H, scale,
                                             _stream()), 'mr_f32_segment_sum')
        return dst
    check(_lib.load().mr_segment_sum(_ptr(s[0]), ld[0], n[0], _ptr(s[1]), ld[1], n[1], _ptr(s[2]), ld[2], n[2],
                                     indptr.data_ptr(), indices.data_ptr(), dst.data_ptr(), _ld(dst),
                                     MR_DT_F32 if dst.dtype == F32 else MR_DT_BF16, n_dst, H, scale, int(accumulate),
                                     _stream()), 'mr_segment_sum')
    return dst


_PERIODIC = {}


def add_rows_periodic(x, table):
    """x[r] += table[r % S] for every row r of x [n * S, H] (S = table.shape[0]): the learned position table added to every sequence
    (mreserve/modeling.py:335-341) as ONE launch -- a segment sum with one static entry per row, accumulated onto x -- instead of one add per
    sequence.  fp32 tensors (the correctness program) keep the per-sequence adds."""
    S = table.shape[0]
    n = x.shape[0] // S
    assert x.shape[0] == n * S and x.shape[1] == table.shape[1]
    if x.dtype == F32:
        for i in range(n):
            add_(x[i * S:(i + 1) * S], table)
        return x
    key = (n, S, str(x.device))
    if key not in _PERIODIC:
        _PERIODIC[key] = (torch.arange(n * S + 1, dtype=torch.int32, device=x.device),
                          (torch.arange(n * S, dtype=torch.int32, device=x.device) % S).contiguous())
    indptr, idx = _PERIODIC[key]
    return segment_sum([table], indptr, idx, x, accumulate=True)


@_timed('rowops')
def rows_mean_fwd(src, rows, dst):
    G, R = rows.shape
    if src.dtype == F32:
        check(_lib.load().mr_f32_rows_mean_fwd(src.data_ptr(), _ld(src), rows.data_ptr(), dst.data_ptr(), G, R, dst.shape[1],
                                               _stream()), 'mr_f32_rows_mean_fwd')
        return dst
    check(_lib.load().mr_rows_mean_fwd(src.data_ptr(), _ld(src), rows.data_ptr(), dst.data_ptr(), G, R, dst.shape[1], _stream()),
          'mr_rows_mean_fwd')
    return dst


@_timed('rowops')
def rows_mean_bwd(ddst, rows, dsrc):
    G, R = rows.shape
    if ddst.dtype == F32:
        check(_lib.load().mr_f32_rows_mean_bwd(ddst.data_ptr(), rows.data_ptr(), dsrc.data_ptr(), _ld(dsrc), G, R, ddst.shape[1], _stream()),
              'mr_f32_rows_mean_bwd')
        return
    check(_lib.load().mr_rows_mean_bwd(ddst.data_ptr(), rows.data_ptr(), dsrc.data_ptr(), _ld(dsrc), G, R, ddst.shape[1],
                                       _stream()), 'mr_rows_mean_bwd')


@_timed('rowops')
def pad_cols(src, dst):
    assert src.is_contiguous() and dst.is_contiguous()
    check(_lib.load().mr_pad_cols(src.data_ptr(), src.shape[1], dst.data_ptr(), dst.shape[1], src.shape[0], _stream()),
          'mr_pad_cols')
    return dst


@_timed('rowops')
def fill_rows(vec, dst, ngroups, grp_stride, off):
    if dst.dtype == F32:
        check(_lib.load().mr_f32_fill_rows(vec.data_ptr(), dst.data_ptr(), _ld(dst), ngroups, grp_stride, off, dst.shape[1],
                                           _stream()), 'mr_f32_fill_rows')
        return
    check(_lib.load().mr_fill_rows(vec.data_ptr(), dst.data_ptr(), _ld(dst), ngroups, grp_stride, off, dst.shape[1], _stream()),
          'mr_fill_rows')


@_timed('rowops')
def sum_rows_strided(src, ngroups, grp_stride, off, out):
    if src.dtype == F32:
        check(_lib.load().mr_f32_sum_rows_strided(src.data_ptr(), _ld(src), ngroups, grp_stride, off, src.shape[1], out.data_ptr(), _stream()),
              'mr_f32_sum_rows_strided')
        return out
    check(_lib.load().mr_sum_rows_strided(src.data_ptr(), _ld(src), ngroups, grp_stride, off, src.shape[1], out.data_ptr(),
                                          _stream()), 'mr_sum_rows_strided')
    return out


@_timed('rowops')
def add_(a, b, y=None):
    y = a if y is None else y
    if a.dtype == F32:
        assert y is a
        check(_lib.load().mr_f32_axpby(a.data_ptr(), b.data_ptr(), 1.0, 1.0, a.numel(), _stream()), 'mr_f32_axpby')
        return y
    check(_lib.load().mr_add_bf16(a.data_ptr(), b.data_ptr(), y.data_ptr(), a.numel(), _stream()), 'mr_add_bf16')
    return y


@_timed('rowops')
def unit_norm_scale_fwd(x, log_scale, y, inv_norm=None):
    """y = unit_normalize(x) * exp(min(log_scale, ln 100) / 2); fp32 path: log_scale None = plain unit_normalize."""
    rows, H = x.shape
    if x.dtype == F32:
        check(_lib.load().mr_f32_unit_norm_scale_fwd(x.data_ptr(), _ld(x), _ptr(log_scale), y.data_ptr(), _ld(y), rows, H,
                                                     _stream()), 'mr_f32_unit_norm_scale_fwd')
        return y
    check(_lib.load().mr_unit_norm_scale_fwd(x.data_ptr(), _ld(x), log_scale.data_ptr(), y.data_ptr(), _ld(y),
                                             inv_norm.data_ptr(), rows, H, _stream()), 'mr_unit_norm_scale_fwd')
    return y


@_timed('rowops')
def unit_norm_scale_bwd(x, log_scale, inv_norm, dy, dx, dlog_scale, partials, accumulate=False):
    """dlog_scale[0] = (accumulate ? old : 0) + this call's temperature gradient; partials: >= (rows + 3) // 4 fp32 scratch."""
    rows, H = x.shape
    if x.dtype == F32:
        assert partials.dtype == F32 and partials.numel() >= rows
        if not accumulate:
            dlog_scale.zero_()
        check(_lib.load().mr_f32_unit_norm_scale_bwd(x.data_ptr(), _ld(x), log_scale.data_ptr(), dy.data_ptr(), _ld(dy), dx.data_ptr(), _ld(dx),
                                                     0, dlog_scale.data_ptr(), partials.data_ptr(), rows, H, _stream()),   # (dx overwritten; dls += )
              'mr_f32_unit_norm_scale_bwd')
        return dx
    assert partials.dtype == F32 and partials.numel() >= (rows + 3) // 4
    check(_lib.load().mr_unit_norm_scale_bwd(x.data_ptr(), _ld(x), log_scale.data_ptr(), inv_norm.data_ptr(), dy.data_ptr(),
                                             _ld(dy), dx.data_ptr(), _ld(dx), dlog_scale.data_ptr(), int(accumulate),
                                             partials.data_ptr(), rows, H, _stream()), 'mr_unit_norm_scale_bwd')
    return dx


@_timed('rowops')
def contrastive_lse(logits, own_off, coef, src, loss_out, diag, row_scratch):
    """loss_out[0] += coef * sum_l (lse_l - logits[l, own_off + l]); logits <- dL/dlogits; row_scratch: >= L fp32."""
    L, V = logits.shape
    assert row_scratch.dtype == F32 and row_scratch.numel() >= L
    check(_lib.load().mr_contrastive_lse(logits.data_ptr(), _ld(logits), L, V, own_off, coef, _ptr(src), loss_out.data_ptr(),
                                         _ptr(diag), row_scratch.data_ptr(), _stream()), 'mr_contrastive_lse')


@_timed('rowops')
def cast_f32_to_bf16(src, dst):
    check(_lib.load().mr_cast_f32_to_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _stream()), 'mr_cast_f32_to_bf16')
    return dst


@_timed('rowops')
def split_hilo(src, hi, lo):
    check(_lib.load().mr_split_f32_to_bf16_hilo(src.data_ptr(), hi.data_ptr(), lo.data_ptr(), src.numel(), _stream()),
          'mr_split_f32_to_bf16_hilo')


@_timed('rowops')
def split_hilo_rows(src, hi, lo):
    """src [rows, cols] fp32 -> hi, lo [rows, cols] bf16 views (any row strides): x = hi + lo to 16 mantissa bits."""
    rows, cols = src.shape
    assert hi.shape == lo.shape == src.shape and _ld(hi) == _ld(lo)
    check(_lib.load().mr_split_f32_to_bf16_hilo_rows(src.data_ptr(), _ld(src), hi.data_ptr(), lo.data_ptr(), _ld(hi), rows, cols,
                                                     _stream()), 'mr_split_f32_to_bf16_hilo_rows')


@_timed('optimizer')
def adam_bf16_update(master, work, grad, mu, nu, decay_flags, b1, b2, eps, weight_decay, sched, neg_lr, bc1=1.0, bc2=1.0):
    check(_lib.load().mr_adam_bf16_update(master.data_ptr(), work.data_ptr(), grad.data_ptr(), mu.data_ptr(), nu.data_ptr(),
                                          decay_flags.data_ptr(), master.numel(), b1, b2, eps, weight_decay, sched, neg_lr,
                                          bc1, bc2, _stream()), 'mr_adam_bf16_update')


@_timed('optimizer')
def nan_to_num_(g):
    if g.dtype == F32:
        check(_lib.load().mr_f32_nan_to_num(g.data_ptr(), g.numel(), _stream()), 'mr_f32_nan_to_num')
        return
    check(_lib.load().mr_nan_to_num_bf16(g.data_ptr(), g.numel(), _stream()), 'mr_nan_to_num_bf16')


@_timed('optimizer')
def transpose_leaves(work, workT, leaves_dev, nleaf, tile_lo, tile_hi):
    """workT <- per-leaf transposes of work for the 64 x 64 tiles [tile_lo, tile_hi) of the leaf table (params.ParamStore)."""
    check(_lib.load().mr_transpose_leaves(work.data_ptr(), workT.data_ptr(), leaves_dev.data_ptr(), nleaf, tile_lo, tile_hi, _stream()),
          'mr_transpose_leaves')


def masked_lm_xent(logits, labels, out2, dlogits=None, scratch=None):
    """P:265-274 (`text_preds`): logits [n, V] fp32, labels [n] int32; out2 = [masked mean NLL, number of unmasked rows]."""
    n, V = logits.shape
    assert logits.dtype == F32 and logits.stride(1) == 1 and labels.dtype == torch.int32 and labels.numel() == n
    assert dlogits is None or (dlogits.dtype == F32 and dlogits.stride() == logits.stride())
    scratch = torch.empty(3 * n, dtype=F32, device=logits.device) if scratch is None else scratch
    check(_lib.load().mr_masked_lm_xent(logits.data_ptr(), logits.stride(0), n, V, labels.data_ptr(), out2.data_ptr(), _ptr(dlogits),
                                        scratch.data_ptr(), _stream()), 'mr_masked_lm_xent')
    return out2


@_timed('optimizer')
def cast_params(master, work):
    check(_lib.load().mr_cast_f32_to_bf16_params(master.data_ptr(), work.data_ptr(), master.numel(), _stream()),
          'mr_cast_f32_to_bf16_params')


@_timed('optimizer')
def adam_bf16_update_finetune(master, work, grad, mu, nu, orig, decay_flags, b1, b2, eps, weight_decay, sched, neg_lr, bc1, bc2):
    check(_lib.load().mr_adam_bf16_update_finetune(master.data_ptr(), work.data_ptr(), grad.data_ptr(), mu.data_ptr(), nu.data_ptr(),
                                                   orig.data_ptr(), decay_flags.data_ptr(), master.numel(), b1, b2, eps,
                                                   weight_decay, sched, neg_lr, bc1, bc2, _stream()), 'mr_adam_bf16_update_finetune')


@_timed('rowops')
def softmax_xent(logits, row_stride, class_stride, labels, rows, C, coef, loss_out, correct_out=None, dlogits=None):
    check(_lib.load().mr_softmax_xent(logits.data_ptr(), row_stride, class_stride, labels.data_ptr(), rows, C, coef,
                                      loss_out.data_ptr(), _ptr(correct_out), _ptr(dlogits), _stream()), 'mr_softmax_xent')


@_timed('optimizer')
def adam_bf16_update_dev(master, work, grad, mu, nu, orig, decay_flags, b1, b2, eps, weight_decay, hyper):
    """Adam chain on (a sub-range of) the flat buffers with the per-step scalars in the device vector `hyper` [4]."""
    if grad.dtype == F32:       # use_bfloat16_grads = False: fp32 gradients straight into the chain
        assert orig is None
        check(_lib.load().mr_adam_f32grad_update_dev(master.data_ptr(), work.data_ptr(), grad.data_ptr(), mu.data_ptr(), nu.data_ptr(),
                                                     decay_flags.data_ptr(), master.numel(), b1, b2, eps, weight_decay, hyper.data_ptr(), _stream()),
              'mr_adam_f32grad_update_dev')
        return
    check(_lib.load().mr_adam_bf16_update_dev(master.data_ptr(), work.data_ptr(), grad.data_ptr(), mu.data_ptr(), nu.data_ptr(),
                                              _ptr(orig), decay_flags.data_ptr(), master.numel(), b1, b2, eps, weight_decay,
                                              hyper.data_ptr(), _stream()), 'mr_adam_bf16_update_dev')
