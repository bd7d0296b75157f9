"""Device program of one pretraining step: MerlotReservePretrainer.__call__ (pretrain/pretrain_model.py:38-259),
loss_fn_given_preds (:262-303) and their hand-written backward, as a fixed sequence of C-ABI kernel launches over
buffers allocated once (all shapes are static given config and B, so the whole step can be captured in a hipGraph).

Data layout in HBM (bf16 unless noted; M = sequences x positions of a tower):
  per tower and layer, kept for backward: X[l] layer input, ln1, qkv (post "rotary"), att, lse (fp32), xmid, ln2,
  hpre, hact  = 16 H bf16 / token / layer;  LN statistics fp32.
  joint input is ASSEMBLED by one index-driven gather from [token embedding | pooled audio | pooled vision] rows;
  the [S,S] attention mask never exists (one int32 code per position); the "rotary" is a [positions, 32] fp32 table.
"""
import math
import os

import numpy as np
import torch

from . import ops
from .config import Dims
from .planner import VOCAB, build_plan, static_tables

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


class TowerState:
    """Saved activations of one TransformerEncoder (modeling.py:283-376) over M = nseq*S rows."""

    def __init__(self, M, H, L, nseq, S, dev, dtype=BF16, keep_hpre=None):
        nh = H // 64
        z = lambda *s: torch.zeros(*s, dtype=dtype, device=dev)
        f = lambda *s: torch.zeros(*s, dtype=F32, device=dev)
        self.M, self.H, self.L, self.nseq, self.S = M, H, L, nseq, S
        self.xin, self.xf = z(M, H), z(M, H)
        self.X = z(L + 1, M, H)
        self.ln1, self.ln2, self.xmid, self.att = z(L, M, H), z(L, M, H), z(L, M, H), z(L, M, H)
        self.qkv = z(L, M, 3 * H)
        self.hact = z(L, M, 4 * H)
        keep_hpre = dtype == BF16 if keep_hpre is None else keep_hpre
        self.hpre = z(L, M, 4 * H) if keep_hpre else None           # gelu'(pre-activation), only kept for backward
        self.lse = f(L, nseq, nh, S)
        self.stats = f(2 * L + 2, 2, M)       # [ln index][mean|rstd][row]; index 0 = pre_ln, 1+2l / 2+2l = layer l, last = final


class TowerEngine:
    """What every program built from TransformerEncoder towers shares (pretraining step, VCR finetuning step): the
    encoder forward / hand-written backward over a TowerState, the CLS head + attention pool of the vision / audio
    towers, the two scratch sets (main / side stream) and the device copies of the per-batch plan.
    Subclasses set: self.p (ParamStore), self.W (weight views), self.dev, self.dtype, self.fwd_only, self.side_stream,
    self.sc_main / self.sc_side / self.cur (via _make_scratch), self.plan_dev / _plan_caps / _plan_views / plan_frozen."""

    def _make_scratch(self, Ms, Gs, Ps, H, nh):
        dev = self.dev
        z = lambda *s: torch.zeros(*s, dtype=self.dtype, device=dev)
        f = lambda *s: torch.zeros(*s, dtype=F32, device=dev)
        f32 = self.dtype == F32
        # fp32 program (use_bfloat16_grads = False): the reductions are immediate, a LayerNorm backward wants 2 floats per row
        ln_ws_shared = f(2 * Ms) if f32 else None
        ln_ws = (lambda: ln_ws_shared) if f32 else (lambda: ops.layernorm_bwd_workspace(H, dev))

        class Scratch:
            pass
        sc = Scratch()
        sc.T_a = z(Ms, H)
        # the weight gradients of up to WG layers share one launch (encoder_backward / _wgrad_group), so a layer's upstream gradients
        # (d qkv, d pre-activation, and the two [M, H] gradients of its residual stream) stay alive while the next layers' are produced:
        # WG sets of T_q / T_h, 2 WG [M, H] buffers (the caller's D joins their rotation)
        WG = self._wgrad_group(Ms, H)
        sc.T_qs, sc.T_hs = [z(Ms, 3 * H) for _ in range(WG)], [z(Ms, 4 * H) for _ in range(WG)]
        sc.T_ds = [z(Ms, H) for _ in range(2 * WG)]
        sc.WG = WG
        sc.delta = f(Ms * nh)
        sc.gemm_ws = f(32 * 1024 * 1024)                       # 128 MiB of fp32 split-K partials
        sc.ln_ws = ln_ws()
        sc.cs_ws = ops.colsum_workspace(4 * H, dev)
        # A layer's four deferred reductions (2 LayerNorm, 2 bias) keep their partial rows until ONE mr_reduce_partials launch per
        # weight-gradient GROUP of layers (engine._wgrad_group: 2 base, 4 large) -- the gradients are not needed before the group's
        # layer_done: one set of partial-row workspaces per position in the group
        sc.ln_wsA = [ln_ws() for _ in range(WG)]
        sc.ln_wsB = [ln_ws() for _ in range(WG)]
        sc.cs_wsA = [ops.colsum_workspace(4 * H, dev) for _ in range(WG)]
        sc.cs_wsB = [ops.colsum_workspace(4 * H, dev) for _ in range(WG)]
        # per-(tile, wave) column-sum partials written by the d(pre-activation) GEMM's epilogue (the fc1 bias gradient)
        sc.cs_fused = [f(4 * ((Ms + 255) // 256) * 4 * H) for _ in range(WG)]
        # per-(sequence, block) column-sum partials written by the attention backward kernels (the qkv bias gradient): at most
        # one row per 16 positions (short sequences) of [3H]
        sc.cs_attn = [f(((Ms + 15) // 16 + 64) * 3 * H) for _ in range(WG)]
        # tower-level reductions (pre / final LayerNorm, pooling and projection biases) keep their partial rows in these until
        # the tower's ONE mr_reduce_partials launch (they used to be two latency-bound launches each)
        sc.tower_ln_ws = [ln_ws() for _ in range(2)]
        sc.tower_cs_ws = [ops.colsum_workspace(H, dev) for _ in range(6)]
        sc.d_pool_q, sc.d_pool_po, sc.d_pool_qin = z(Gs, H), z(Gs, H), z(Gs, H)
        sc.d_k, sc.d_v = z(Ms, H), z(Ms, H)                    # CLS rows stay zero
        sc.Dpatch = z(Ps, H)
        return sc

    # GEMMs of this engine: split-K partials go to the scratch set being issued on (`self.cur`: main / side stream), which
    # belongs to THIS engine -- no module-level workspace (two engines, or a captured graph and a later engine, never share)
    def gemm(self, a, b, out, **kw):
        return ops.gemm(a, b, out, ws=None if self.cur is None else self.cur.gemm_ws, **kw)

    def fgemm(self, a, name, out, **kw):
        """Forward Dense: out = a @ W[name].  The bf16 program reads the TRANSPOSED working copy where the store keeps one
        (params.ParamStore.wT): both operands contraction-contiguous, the fast GEMM path; same values, same result."""
        wT = getattr(self.p, 'wT', None) if self.dtype == BF16 else None
        if wT and name in wT and os.environ.get('MR_NO_WT') != '1':
            return self.gemm(a, wT[name], out, transB=True, **kw)
        return self.gemm(a, self.W[name], out, **kw)

    def gemm_args(self, a, b, out, **kw):
        return ops.gemm_args(a, b, out, ws=None if self.cur is None else self.cur.gemm_ws, **kw)

    def fgemm_grouped(self, items):
        """Several independent forward Dense layers [(a, weight name, out, kwargs)] in one mr_gemm_grouped call (small problems of one
        operand layout share a launch; anything else falls back to one launch each inside the library)."""
        wT = getattr(self.p, 'wT', None) if self.dtype == BF16 else None
        if self.dtype != BF16 or os.environ.get('MR_NO_SMALL_GROUPING') == '1':
            for a, name, out, kw in items:
                self.fgemm(a, name, out, **kw)
            return
        args = []
        for a, name, out, kw in items:
            if wT and name in wT and os.environ.get('MR_NO_WT') != '1':
                args.append(self.gemm_args(a, wT[name], out, transB=True, **kw))
            else:
                args.append(self.gemm_args(a, self.W[name], out, **kw))
        ops.gemm_grouped(args)

    # tower-level deferred reductions: between _begin_tower_reductions() and _flush_tower_reductions() every LayerNorm /
    # colsum issued through _t_ln_bwd / _t_colsum leaves its partial rows in a workspace of its own
    def _begin_tower_reductions(self):
        off = os.environ.get('MR_NO_BATCH_REDUCE') == '1' or os.environ.get('MR_NO_TOWER_DEFER') == '1' or self.dtype == F32   # (A/B switches; fp32: immediate)
        return {'jobs': None if off else [], 'ln': 0, 'cs': 0, 'sc': self.cur}

    def _t_ln_bwd(self, tr, *a, **k):
        if tr is None or tr['jobs'] is None:
            return ops.layernorm_bwd(*a, self.cur.ln_ws, **k)
        ws = tr['sc'].tower_ln_ws[tr['ln']]
        tr['ln'] += 1
        return ops.layernorm_bwd(*a, ws, jobs=tr['jobs'], **k)

    def _t_colsum(self, tr, x, out):
        if tr is None or tr['jobs'] is None:
            return ops.colsum(x, out, self.cur.cs_ws)
        ws = tr['sc'].tower_cs_ws[tr['cs']]
        tr['cs'] += 1
        return ops.colsum(x, out, ws, jobs=tr['jobs'])

    def _flush_tower_reductions(self, tr):
        if tr is not None and tr['jobs']:
            ops.reduce_partials(tr['jobs'])          # (clears the list)

    def _on_side(self, fn):
        """Issue fn()'s kernels on the side stream (forked from / joined to the current stream by the caller)."""
        if self.fwd_only:
            if os.environ.get('MR_NO_SIDE_STREAM') == '1':
                return fn()
            with torch.cuda.stream(self.side_stream):
                return fn()
        self.cur = self.sc_side
        side_cus = int(os.environ.get('MR_SIDE_CUS', '0'))       # (experiment) persistent GEMM grids of the side stream's tower
        prev = ops.get_option('gemm_cus') if side_cus else 0
        try:
            if side_cus:
                ops.set_option('gemm_cus', side_cus)
            if os.environ.get('MR_NO_SIDE_STREAM') == '1':       # (A/B switch) same work, issued in line
                return fn()
            with torch.cuda.stream(self.side_stream):
                return fn()
        finally:
            if side_cus:
                ops.set_option('gemm_cus', prev)
            self.cur = self.sc_main

    def _unpad_csr(self, nseq, S):
        rows = (np.arange(nseq)[:, None] * S + 1 + np.arange(S - 1)[None]).reshape(-1).astype(np.int32)
        indptr = np.arange(len(rows) + 1, dtype=np.int32)
        return torch.as_tensor(indptr).to(self.dev), torch.as_tensor(rows).to(self.dev)

    # ------------------------------------------------------------------------------------------ plan upload
    def set_plan(self, plan):
        """Copy the host plan into ONE persistent device buffer (same addresses every step: graph-capturable) through a ring
        of two pinned host buffers and ONE asynchronous H2D copy per step.  A pinned buffer is rewritten only after the copy
        that last read it has completed (an event per buffer), so the host may plan step t+1 while the GPU runs step t and
        the copy engine never reads memory the host has since reused.  (Pageable sources made every copy a staged,
        stream-serialising one.)"""
        arrs = {k: np.ascontiguousarray(v) for k, v in plan.items() if isinstance(v, np.ndarray)}
        need_layout = any(k not in self._plan_caps or a.nbytes > self._plan_caps[k] for k, a in arrs.items())
        if need_layout:
            assert not self.plan_frozen, 'plan buffers would be reallocated after graph capture'
            # index lists are bounded by the number of joint + span positions; everything else has a fixed size
            off = 0
            self._plan_off, self._plan_caps = {}, {}
            for k, a in arrs.items():
                cap = max(self._idx_capacity() * a.itemsize if k.endswith('_idx') else a.nbytes, a.nbytes)
                cap = (cap + 255) // 256 * 256
                self._plan_off[k], self._plan_caps[k] = off, cap
                off += cap
            self.plan_dev = torch.zeros(off, dtype=torch.uint8, device=self.dev)
            pin = self.dev.type == 'cuda'
            self._plan_host = [torch.zeros(off, dtype=torch.uint8, pin_memory=pin) for _ in range(2)]
            self._plan_ev = [None, None]
            self._plan_turn = 0
        i = self._plan_turn
        self._plan_turn ^= 1
        if self._plan_ev[i] is not None:
            self._plan_ev[i].synchronize()
        host = self._plan_host[i].numpy()
        for k, a in arrs.items():
            o = self._plan_off[k]
            host[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
        self.plan_dev.copy_(self._plan_host[i], non_blocking=True)
        if self.dev.type == 'cuda':
            self._plan_ev[i] = torch.cuda.Event()
            self._plan_ev[i].record()
        for k, a in arrs.items():
            o = self._plan_off[k]
            self._plan_views[k] = self.plan_dev[o:o + a.nbytes].view(torch.from_numpy(a[:0].reshape(-1)).dtype).view(a.shape)
        self.plan = plan

    def _pl(self, k):
        return self._plan_views[k]

    def _joint_rot(self):
        """The joint tower's "rotary" table of the current plan, or None when the config drops the coordinates (do_rotary = False)."""
        return self._pl('joint_rot') if self.d.do_rotary else None

    def _idx_capacity(self):
        """Upper bound on the length of any index list of the plan (so the buffers never move after capture)."""
        return self.tj.M + self.ts.M + 64

    # ------------------------------------------------------------------------------------------ encoder
    def _names(self, prefix, l):
        p = f'{prefix}/layer_{l:02d}'
        return dict(g1=f'{p}/pre_attn_ln/scale', b1=f'{p}/pre_attn_ln/bias', wqkv=f'{p}/attention_layer/qkv/kernel',
                    bqkv=f'{p}/attention_layer/qkv/bias', wo=f'{p}/attention_layer/attn_proj/kernel',
                    g2=f'{p}/pre_mlp_ln/scale', b2=f'{p}/pre_mlp_ln/bias', w1=f'{p}/mlp_layer/intermediate/kernel',
                    bb1=f'{p}/mlp_layer/intermediate/bias', w2=f'{p}/mlp_layer/out/kernel')

    def encoder_forward(self, st, prefix, rot, code, dense_mask=None):
        """TransformerEncoder body (modeling.py:360-366) on st.xin (CLS row already in place) -> st.xf.
        dense_mask: uint8 [nseq, S, S] (!= 0 = allowed) -- ANY attention mask (modeling.py:303, 350-356) instead of the position codes; the
        attention then runs on the plain dense-mask kernels, forward here and backward in encoder_backward (round 6: a correctness path)."""
        W, H, nh = self.W, st.H, st.H // 64
        ops.layernorm_fwd(st.xin, W[f'{prefix}/pre_ln/scale'], W[f'{prefix}/pre_ln/bias'], st.X[0], st.stats[0, 0], st.stats[0, 1])
        for l in range(st.L):
            n = self._names(prefix, l)
            x = st.X[l]
            ops.layernorm_fwd(x, W[n['g1']], W[n['b1']], st.ln1[l], st.stats[1 + 2 * l, 0], st.stats[1 + 2 * l, 1])
            self.fgemm(st.ln1[l], n['wqkv'], st.qkv[l], bias=W[n['bqkv']], rot_tab=rot, rot_cols=2 * H)
            if dense_mask is not None:
                ops.attention_fwd_dense_mask(st.qkv[l], dense_mask, st.att[l], st.nseq, st.S, nh)
            else:
                ops.attention_fwd(st.qkv[l], code, st.att[l], st.lse[l], st.nseq, st.S, nh)
            self.fgemm(st.att[l], n['wo'], st.xmid[l], residual=x)
            ops.layernorm_fwd(st.xmid[l], W[n['g2']], W[n['b2']], st.ln2[l], st.stats[2 + 2 * l, 0], st.stats[2 + 2 * l, 1])
            self.fgemm(st.ln2[l], n['w1'], st.hact[l], bias=W[n['bb1']], act=ops.ACT_GELU, c2=None if st.hpre is None else st.hpre[l])
            self.fgemm(st.hact[l], n['w2'], st.X[l + 1], residual=st.xmid[l])
        k = 2 * st.L + 1
        ops.layernorm_fwd(st.X[st.L], W[f'{prefix}/final_ln/scale'], W[f'{prefix}/final_ln/bias'], st.xf, st.stats[k, 0], st.stats[k, 1])

    def _wgrad_group(self, M, H):
        """Layers whose weight gradients share one grouped launch of the one-tile-per-workgroup TN kernel (csrc/gemm3.hip): the count
        (<= 4) whose 256 x 256 output tiles fill the 256 CUs' rounds best, the smallest on a tie -- base model: 108 tiles per layer ->
        2 layers = 216 (one round at 84 %; 3 or 4 layers are no better); large: 192 per layer -> 4 layers = 768 = three full rounds
        (one layer per launch left a quarter of the chip idle: 24 rounds per tower instead of 18).  K = M tokens must be long
        enough to amortise a tile's prologue."""
        if os.environ.get('MR_WGRAD_PAIR') == '0' or M < 2048:
            return 1
        t = lambda m, n: ((m + 255) // 256) * ((n + 255) // 256)
        tiles = t(4 * H, H) + t(H, 4 * H) + t(H, H) + t(H, 3 * H)
        eff = lambda n: n / (((n + 255) // 256) * 256)
        best = 1
        for g in range(2, int(os.environ.get('MR_WGRAD_MAX', '4')) + 1):
            if eff(g * tiles) > eff(best * tiles) + 0.05:
                best = g
        return best

    def encoder_backward(self, st, prefix, rot, code, D, layer_done=None, tr=None, extra_wgrads=None, dense_mask=None):
        """D [M,H]: gradient wrt st.xf.  Returns the buffer holding the gradient wrt st.xin (D or one of the scratch
        buffers that rotate through the layers).  Weight gradients go to the flat grad buffer; the four weight
        gradients of a layer -- of TWO layers when that fills the chip better (_wgrad_group) -- are deferred to ONE grouped
        GEMM launch (no split-K).
        layer_done(l): called once every parameter gradient of layers >= l is final on the issuing stream (the data-parallel
        trainer reduces gradient buckets from there while backward continues).
        extra_wgrads: [(x, dy, out)] weight gradients of the tower's head over the SAME tokens (attention-pool key / value
        projections, the joint head): they join the FIRST group's launch when its tiles still fit the same number of CU rounds
        (base: 216 + 9 per extra of 256) -- alone each is a 9-tile split-K launch of ~65 us --, else they are launched here."""
        W, G, H, nh, M = self.W, self.G, st.H, st.H // 64, st.M
        T_a = self.cur.T_a[:M]
        # [M, H] gradient buffers: a layer reads Dcur, writes Dmid and Dnext; Dcur / Dmid (and the layer's T_q / T_h) stay untouched
        # until the group's weight gradients have been issued
        group = min(self._wgrad_group(M, H), len(self.cur.T_qs))
        T_qs, T_hs = [t[:M] for t in self.cur.T_qs], [t[:M] for t in self.cur.T_hs]
        free = [t[:M] for t in self.cur.T_ds[:2 * group]]
        Dcur = D
        pending, held, done_layers = [], [], []
        if extra_wgrads:
            t256 = lambda m, n: ((m + 255) // 256) * ((n + 255) // 256)
            tiles = group * (t256(4 * H, H) + t256(H, 4 * H) + t256(H, H) + t256(H, 3 * H))
            extra = sum(t256(x.shape[1], dy.shape[1]) for x, dy, _ in extra_wgrads)
            rounds = lambda n: (n + 255) // 256
            if M >= 2048 and 4 * group + len(extra_wgrads) <= 20 and rounds(tiles + extra) == rounds(tiles) and os.environ.get('MR_NO_WGRAD_EXTRA') != '1':
                pending += [self.gemm_args(x, dy, out, transA=True) for x, dy, out in extra_wgrads]
            else:
                for x, dy, out in extra_wgrads:
                    self.gemm(x, dy, out, transA=True)
        k = 2 * st.L + 1
        self._t_ln_bwd(tr, Dcur, st.X[st.L], W[f'{prefix}/final_ln/scale'], st.stats[k, 0], st.stats[k, 1], Dcur,
                       G[f'{prefix}/final_ln/scale'], G[f'{prefix}/final_ln/bias'])
        # everything above the layers (heads, pooling, final LayerNorm) is reduced here: a data-parallel trainer hands the
        # tower's LAST layers' gradient bucket -- which holds these leaves -- to the all-reduce at the first layer_done
        self._flush_tower_reductions(tr)
        jobs = None if os.environ.get('MR_NO_BATCH_REDUCE') == '1' or self.dtype == F32 else []      # (A/B switch) immediate reductions
        for l in reversed(range(st.L)):
            n = self._names(prefix, l)
            T_q, T_h = T_qs[len(done_layers)], T_hs[len(done_layers)]
            Dmid, Dnext = free.pop(), free.pop()
            gi = len(done_layers)               # position in the weight-gradient group: its own partial-row workspaces
            fused_bb1 = ops.gemm_colsum_job(Dcur, W[n['w2']], T_h, self.cur.cs_fused[gi], G[n['bb1']],
                                            jobs if os.environ.get('MR_NO_GEMM_COLSUM') != '1' else None, transB=True,
                                            aux=st.hpre[l], ws=self.cur.gemm_ws)            # d hpre (+ its column sums = d bias)
            self.gemm(T_h, W[n['w1']], T_a, transB=True)                                    # d ln2
            ops.layernorm_bwd(T_a, st.xmid[l], W[n['g2']], st.stats[2 + 2 * l, 0], st.stats[2 + 2 * l, 1], Dmid,
                              G[n['g2']], G[n['b2']], self.cur.ln_wsA[gi], dx_add=Dcur, jobs=jobs)    # Dmid = d xmid
            self.gemm(Dmid, W[n['wo']], T_a, transB=True)                                   # d att
            fuse_q = jobs is not None and os.environ.get('MR_NO_ATTN_COLSUM') != '1' and dense_mask is None          # (A/B switch)
            if dense_mask is not None:       # any mask (the forward ran mr_attention_fwd_dense_mask): the plain backward pair; d bias = a column-sum pass below
                ops.attention_bwd_dense_mask(st.qkv[l], dense_mask, T_a, T_q, rot, st.nseq, st.S, nh)
            else:
                ops.attention_bwd(st.qkv[l], code, st.att[l], T_a, st.lse[l], self.cur.delta, T_q, rot, st.nseq, st.S, nh,
                                  colsum_ws=self.cur.cs_attn[gi], bias_grad=G[n['bqkv']], jobs=jobs if fuse_q else None)   # (+ column sums of T_q = d bias)
            if not fused_bb1:
                ops.colsum(T_h, G[n['bb1']], self.cur.cs_wsA[gi], jobs=jobs)
            if not fuse_q:
                ops.colsum(T_q, G[n['bqkv']], self.cur.cs_wsB[gi], jobs=jobs)
            self.gemm(T_q, W[n['wqkv']], T_a, transB=True)                                  # d ln1
            pending += [self.gemm_args(st.hact[l], Dcur, G[n['w2']], transA=True),
                        self.gemm_args(st.ln2[l], T_h, G[n['w1']], transA=True),
                        self.gemm_args(st.att[l], Dmid, G[n['wo']], transA=True),
                        self.gemm_args(st.ln1[l], T_q, G[n['wqkv']], transA=True)]
            held += [Dcur, Dmid]          # (the caller's D joins the rotation once its layer's weight gradients are issued)
            done_layers.append(l)
            ops.layernorm_bwd(T_a, st.X[l], W[n['g1']], st.stats[1 + 2 * l, 0], st.stats[1 + 2 * l, 1], Dnext,
                              G[n['g1']], G[n['b1']], self.cur.ln_wsB[gi], dx_add=Dmid, jobs=jobs)   # Dnext = d X[l]
            if len(done_layers) == group or l == 0:
                if jobs is not None:
                    ops.reduce_partials(jobs)      # the group's 2 LayerNorm + 2 bias gradients per layer: one launch (<= 16 jobs)
                ops.gemm_grouped(pending)
                free += held
                if layer_done is not None:
                    for dl in done_layers:
                        layer_done(dl)
                pending, held, done_layers = [], [], []
            Dcur = Dnext
        self._t_ln_bwd(tr, Dcur, st.xin, W[f'{prefix}/pre_ln/scale'], st.stats[0, 0], st.stats[0, 1], Dcur,
                       G[f'{prefix}/pre_ln/scale'], G[f'{prefix}/pre_ln/bias'])
        return Dcur

    # ------------------------------------------------------------------------------------------ CLS tower head + pool
    def _cls_view(self, t, nseq, S):
        return t.view(nseq, S * t.shape[1])[:, :t.shape[1]]

    def _tower_with_pool_forward(self, st, prefix_t, prefix_pool, rot, pool_rows, qin, q, k, v, po, probs, out_seq, out_cls):
        W, nh = self.W, st.H // 64
        ops.fill_rows(W[f'{prefix_t}/cls'], st.xin, st.nseq, st.S, 0)
        self.encoder_forward(st, prefix_t, rot, None)
        ops.rows_mean_fwd(st.xf, pool_rows, qin)
        # the CLS head and the pool's query projection are independent small GEMMs (64 / 1152 rows): handed over as a group -- ONE launch when both fall to
        # the small-problem kernel (tiny configs); at base / large size the 1152-row query projection is claimed by the few-tile kernel (mr_gemm5_wanted)
        # and the library launches the two separately
        self.fgemm_grouped([(self._cls_view(st.xf, st.nseq, st.S), f'{prefix_t}/cls_proj/kernel', out_cls, dict(bias=W[f'{prefix_t}/cls_proj/bias'])),
                            (qin, f'{prefix_pool}/query/kernel', q, dict(bias=W[f'{prefix_pool}/query/bias']))])
        self.fgemm(st.xf, f'{prefix_pool}/key/kernel', k, bias=W[f'{prefix_pool}/key/bias'])
        self.fgemm(st.xf, f'{prefix_pool}/value/kernel', v, bias=W[f'{prefix_pool}/value/bias'])
        ops.poolattn_fwd(q, k, v, pool_rows, po, probs, nh)
        self.fgemm(po, f'{prefix_pool}/out/kernel', out_seq, bias=W[f'{prefix_pool}/out/bias'])

    def _tower_with_pool_backward(self, st, prefix_t, prefix_pool, rot, pool_rows, qin, q, k, v, po, probs, d_seq, d_cls, D,
                                  layer_done=None, tr=None):
        """d_seq: grad wrt the pooled sequence output; d_cls: grad wrt the cls output.  Returns D = grad wrt st.xin."""
        W, G, nh, M = self.W, self.G, st.H // 64, st.M
        Gn = qin.shape[0]
        d_po, d_q, d_qin = self.cur.d_pool_po[:Gn], self.cur.d_pool_q[:Gn], self.cur.d_pool_qin[:Gn]
        d_k, d_v = self.cur.d_k[:M], self.cur.d_v[:M]
        # the small weight gradients of the head (pool out / query projections, cls_proj: K = 1152 / 1152 / 64 rows) feed nothing in this
        # backward: they leave the chain of data gradients and run as ONE grouped launch at its end
        small_wgrads = []
        defer = self.dtype == BF16 and os.environ.get('MR_NO_SMALL_GROUPING') != '1'

        def wgrad(x, dy, out):
            if defer:
                small_wgrads.append(self.gemm_args(x, dy, out, transA=True))
            else:
                self.gemm(x, dy, out, transA=True)
        self._t_colsum(tr, d_seq, G[f'{prefix_pool}/out/bias'])
        wgrad(po, d_seq, G[f'{prefix_pool}/out/kernel'])
        self.gemm(d_seq, W[f'{prefix_pool}/out/kernel'], d_po, transB=True)
        ops.poolattn_bwd(q, k, v, pool_rows, probs, d_po, d_q, d_k, d_v, nh)
        self._t_colsum(tr, d_q, G[f'{prefix_pool}/query/bias'])
        wgrad(qin, d_q, G[f'{prefix_pool}/query/kernel'])
        self.gemm(d_q, W[f'{prefix_pool}/query/kernel'], d_qin, transB=True)
        self._t_colsum(tr, d_k, G[f'{prefix_pool}/key/bias'])
        self.gemm(d_k, W[f'{prefix_pool}/key/kernel'], D, transB=True)
        self._t_colsum(tr, d_v, G[f'{prefix_pool}/value/bias'])
        self.gemm(d_v, W[f'{prefix_pool}/value/kernel'], D, transB=True, residual=D)
        # the key / value weight gradients (K = all M tokens) ride in the encoder's first grouped weight-gradient launch
        extra = [(st.xf, d_k, G[f'{prefix_pool}/key/kernel']), (st.xf, d_v, G[f'{prefix_pool}/value/kernel'])]
        ops.rows_mean_bwd(d_qin, pool_rows, D)
        # cls head
        cls_in = self._cls_view(st.xf, st.nseq, st.S)
        self._t_colsum(tr, d_cls, G[f'{prefix_t}/cls_proj/bias'])
        wgrad(cls_in, d_cls, G[f'{prefix_t}/cls_proj/kernel'])
        Dc = self._cls_view(D, st.nseq, st.S)
        self.gemm(d_cls, W[f'{prefix_t}/cls_proj/kernel'], Dc, transB=True, residual=Dc)
        if small_wgrads:
            ops.gemm_grouped(small_wgrads)
        D = self.encoder_backward(st, prefix_t, rot, None, D, layer_done=layer_done, tr=tr, extra_wgrads=extra)
        ops.sum_rows_strided(D, st.nseq, st.S, 0, G[f'{prefix_t}/cls'])
        return D



class PretrainEngine(TowerEngine):
    def __init__(self, config, B, params, device, rank=0, world=1, dtype=BF16, train=None):
        """dtype = torch.float32 builds the fp32 program on the fp32 master weights through the mr_f32_* kernels (the reference's
        use_bfloat16 = False arithmetic): forward + loss only by default (the 1e-3 forward-parity check); with train=True also the
        backward, gradients in fp32 (params.grad32) -- the reference's use_bfloat16_grads = False step (pretrain_model.py:323-333), a
        plain correctness path (no fused reductions, no grouped launches).  The default bf16 program is the fast training path."""
        self.config, self.p, self.dev = config, params, torch.device(device)
        self.d = d = Dims(config, B)
        self.rank, self.world = rank, world
        self.dtype = dtype
        train = dtype == BF16 if train is None else train
        self.fwd_only = not train
        self.W = params.w if dtype == BF16 else params.wm
        if train and dtype == F32:
            params.enable_f32_grads()
        self.G = params.g if dtype == BF16 else getattr(params, 'g32', None)
        hp = train
        dev, H = self.dev, d.H
        self.tables = {k: torch.as_tensor(v).to(dev) for k, v in static_tables(d).items()}
        z = lambda *s: torch.zeros(*s, dtype=dtype, device=dev)
        f = lambda *s: torch.zeros(*s, dtype=F32, device=dev)

        self.tv = TowerState(d.Nv * d.Sv, H, d.Lv, d.Nv, d.Sv, dev, dtype, hp)
        self.ta = TowerState(d.Na * d.Sa, H, d.La, d.Na, d.Sa, dev, dtype, hp)
        self.tj = TowerState(d.Nj * d.Sj, H, d.Lj, d.Nj, d.Sj, dev, dtype, hp)
        if not d.do_rotary:                 # d pe: position p sums rows p, Sj + p, 2 Sj + p, ... of the tower-input gradient (a static segment list)
            idx = (np.arange(d.Nj)[None, :] * d.Sj + np.arange(d.Sj)[:, None]).reshape(-1).astype(np.int32)
            self.pe_lists = (torch.as_tensor(np.arange(d.Sj + 1, dtype=np.int32) * d.Nj).to(dev), torch.as_tensor(idx).to(dev))
        self.ts = TowerState(d.Ns * d.Ss, H, d.Ls, d.Ns, d.Ss, dev, dtype, hp)
        Mmax = max(t.M for t in (self.tv, self.ta, self.tj, self.ts))

        # attention pools
        self.Gv, self.Ga = d.Nv * d.hw4, d.Na * d.a_tok
        self.v_qin, self.v_q, self.v_po, self.imgs_seq = z(self.Gv, H), z(self.Gv, H), z(self.Gv, H), z(self.Gv, H)
        self.v_k, self.v_v = z(self.tv.M, H), z(self.tv.M, H)
        self.v_probs = f(self.Gv, d.nh, d.pr * d.pr)
        self.v_cls = z(d.Nv, H)
        self.a_pad_K = (d.a_patch * 65 + 7) // 8 * 8
        self.a_in = z(d.Na * d.a_len, self.a_pad_K)
        self.a_qin, self.a_q, self.a_po, self.audio_seq = z(self.Ga, H), z(self.Ga, H), z(self.Ga, H), z(self.Ga, H)
        self.a_k, self.a_v = z(self.ta.M, H), z(self.ta.M, H)
        self.a_probs = f(self.Ga, d.nh, d.a_pool)
        self.a_cls = z(d.Na, H)
        self.hj = z(self.tj.M, H)
        self.s_cls = z(d.Ns, H)

        # contrastive sections (rows per rank), order of the packed buffer E
        n_i2a, n_t2a, n_ext, n_s2s = B * d.nseg, B * d.ntrg, B * (d.nspans - d.ntrg), B * d.n_inc
        self.sec = {}
        off = 0
        for name, n in (('i2a_x', n_i2a), ('i2a_y', n_i2a), ('t2a_x', n_t2a), ('t2a_y', n_t2a), ('t2a_ye', n_ext),
                        ('s2s_x', n_s2s), ('s2s_y', n_s2s)):
            self.sec[name] = (off, n)
            off += n
        self.R = off
        self.n_pool = n_i2a + n_t2a + n_s2s
        self.Xpool = z(self.n_pool, H)
        self.acls_g = z(d.Na, H)
        self.E, self.dE = z(self.R, H), z(self.R, H)
        self.inv_norm = f(self.R)
        self.loss_acc = f(3)
        self.diag = f(2, 6)
        self.dls = f(3)
        self.dls_part = f(((self.R + 3) // 4 if dtype == BF16 else self.R) + 8)
        # objectives: (name, x section, y sections (gathered across ranks), scale index)
        self.objectives = [('imgs_to_audio', 'i2a_x', ('i2a_y',), 0), ('text_to_audio', 't2a_x', ('t2a_y', 't2a_ye'), 1),
                           ('stuff_to_span', 's2s_x', ('s2s_y',), 2)]
        # One contrastive problem per (objective, direction) (P:276-295): queries = a local section, keys = the matching
        # section(s) of EVERY rank, rank-major (all_gather(...).reshape(-1, H), P:290).  The gathered embeddings E_all
        # [world, R, H] are re-packed once per step into Kcat, where each problem's keys are contiguous rows (so logits,
        # d(query) and d(key) are ONE GEMM each for any world size) and appear twice, for the hi / lo split of dL/dlogits:
        #   rows [kb, kb + ldv): key c of the problem at row kb + c (zero rows for c >= V = world * nk), [kb + ldv, kb + 2 ldv): again.
        self.cprob = []
        kb = 0
        for oi, (name, xs, ys, _) in enumerate(self.objectives):
            xo, nx = self.sec[xs]
            yo, ny, ny0 = self.sec[ys[0]][0], sum(self.sec[y][1] for y in ys), self.sec[ys[0]][1]
            for di, (q_off, Lq, k_off, nk) in enumerate(((xo, nx, yo, ny), (yo, ny0, xo, nx))):
                V = world * nk
                ldv = (V + 7) // 8 * 8
                self.cprob.append(dict(oi=oi, di=di, name=name, q_off=q_off, Lq=Lq, k_off=k_off, nk=nk, V=V, ldv=ldv, kb=kb,
                                       logits=f(Lq, ldv), dl=None if self.fwd_only or dtype == F32 else z(Lq, 2 * ldv)))
                kb += 2 * ldv
        self.Kcat = z(kb, H)
        self.lse_rows = f(max(p['Lq'] for p in self.cprob))
        # static index lists of the re-pack (a gather from E_all's rows) and of its transpose (dE_all row <- DK hi + lo rows)
        ip, ix = [0], []
        src_of = {}
        for p in self.cprob:
            for half in range(2):
                for c in range(p['ldv']):
                    if c < p['V']:
                        r, j = divmod(c, p['nk'])
                        ix.append(r * self.R + p['k_off'] + j)
                        if half == 0:
                            src_of[r * self.R + p['k_off'] + j] = (p['kb'] + c, p['kb'] + p['ldv'] + c)
                    ip.append(len(ix))
        self.kpack = (torch.tensor(ip, dtype=I32, device=dev), torch.tensor(ix if ix else [0], dtype=I32, device=dev))
        assert sorted(src_of) == list(range(world * self.R)), 'every gathered row is a key of exactly one problem'
        if not self.fwd_only:
            self.DK = z(kb, H)
            self.dE_keys = z(self.R, H) if dtype == F32 and world == 1 else None
            self.kunpack = (torch.arange(0, 2 * world * self.R + 1, 2, dtype=I32, device=dev),
                            torch.tensor([v for e in range(world * self.R) for v in src_of[e]], dtype=I32, device=dev))

        self.side_stream = torch.cuda.Stream(device=dev) if dev.type == 'cuda' else None
        self.plan_dev, self._plan_caps, self._plan_views, self.plan_frozen = None, {}, {}, False
        self.cur = None
        if self.fwd_only:
            return
        # backward scratch
        self.Dv, self.Da, self.Dj, self.Ds = z(self.tv.M, H), z(self.ta.M, H), z(self.tj.M, H), z(self.ts.M, H)
        self.d_v_cls, self.d_s_cls = z(d.Nv, H), z(d.Ns, H)
        # Two scratch sets: the audio tower runs on a side stream concurrently with the vision tower (forward and
        # backward), so that one tower's kernels fill the CUs the other leaves idle (partial last rounds of the
        # persistent GEMMs, small kernels).  `self.cur` is the set the ops being ISSUED right now may use.
        self.sc_main = self._make_scratch(Mmax, max(self.Gv, self.Ga), max(d.Nv * d.hw, d.Na * d.a_len), H, d.nh)   # also serves audio when issued in line
        self.sc_side = self._make_scratch(self.ta.M, self.Ga, d.Na * d.a_len, H, d.nh)
        self.cur = self.sc_main
        self.dXpool = z(self.n_pool, H)
        self.d_hj = z(self.tj.M, H)
        self.d_acls_g, self.d_a_cls = z(d.Na, H), z(d.Na, H)
        self.d_audio_seq, self.d_imgs_seq = z(self.Ga, H), z(self.Gv, H)
        # static gather lists: patch rows of a [nseq, S] grid without the CLS rows
        self.unpad_v = self._unpad_csr(d.Nv, d.Sv)
        self.unpad_a = self._unpad_csr(d.Na, d.Sa)

        # (device copies of the per-batch plan -- plan_dev / _plan_caps / _plan_views, set above -- have fixed sizes where
        # possible; index lists are padded to capacity)

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, batch, plan=None, draws=None):
        """batch: dict of synthetic.make_batch layout (images / audio_clips on device, integer streams numpy)."""
        d, W, H = self.d, self.p.w, self.d.H
        if plan is None:
            from .synthetic import make_draws
            splits, z = draws if draws is not None else make_draws(self.config, d.B, seed=int(batch['audio2text/text_ptr'].astype(np.uint32).sum() % (2 ** 31)))
            plan = build_plan(batch, d, splits, z)
        self.set_plan(plan)
        return self.forward_device(batch['images'], batch['audio_clips'])

    def forward_device(self, images, audio_clips):
        """Device part of forward(): the plan is already in its device buffers."""
        d, W, H = self.d, self.W, self.d.H
        if images.dtype != self.dtype:          # the wire format is bf16 (pretrain/dataloader.py:786-788); the fp32 program casts up, like the
            images, audio_clips = images.to(self.dtype), audio_clips.to(self.dtype)     # reference model's first op (M: x.astype(self.dtype))
        batch = {'images': images, 'audio_clips': audio_clips}
        tv, ta, tj, ts = self.tv, self.ta, self.tj, self.ts

        # vision tower (main stream) and audio tower (side stream) are independent until the joint tower
        def audio_fwd():
            # audio tower (modeling.py:433-476): the stride-2 conv is a GEMM over 2 consecutive hops = 130 inputs
            audio = batch['audio_clips'].reshape(d.Na * d.a_len, d.a_patch * 65)
            if self.dtype == BF16:                       # 16-byte rows for the bf16 GEMM (the fp32 GEMM reads 130-wide rows as they are)
                ops.pad_cols(audio, self.a_in)
                a_view = self.a_in[:, :d.a_patch * 65]
            else:
                a_view = audio
            self._audio2d = a_view
            self.fgemm(a_view, 'audio_encoder/embedding/kernel', ta.xin, bias=W['audio_encoder/embedding/bias'], row_map=(d.a_len, d.Sa, 1))
            self._tower_with_pool_forward(ta, 'audio_encoder/transformer', 'audio_encoder/seq_attnpool', self.tables['audio_rot'],
                                          self.tables['audio_pool_rows'], self.a_qin, self.a_q, self.a_k, self.a_v, self.a_po,
                                          self.a_probs, self.audio_seq, self.a_cls)
        main = torch.cuda.current_stream()
        self.side_stream.wait_stream(main)
        self._on_side(audio_fwd)
        # vision tower (modeling.py:379-430)
        images = self._images2d = batch['images'].reshape(d.Nv * d.hw, d.pp3)
        self.fgemm(images, 'vision_encoder/embedding/kernel', tv.xin, bias=W['vision_encoder/embedding/bias'], row_map=(d.hw, d.Sv, 1))
        self._tower_with_pool_forward(tv, 'vision_encoder/transformer', 'vision_encoder/seq_attnpool', self.tables['vit_rot'],
                                      self.tables['vit_pool_rows'], self.v_qin, self.v_q, self.v_k, self.v_v, self.v_po,
                                      self.v_probs, self.imgs_seq, self.v_cls)
        if d.no_vision:                  # pretrain/pretrain_model.py:61-63: imgs_seq *= 0.0 (the cls output still feeds the contrastive loss)
            self.imgs_seq.zero_()
        main.wait_stream(self.side_stream)
        emb = W['token_encoder/Embed_0/embedding']

        def span_fwd():       # needs only the token table and the plan: runs beside the joint tower
            # span tower on the chosen spans (modeling.py:479-504)
            ops.segment_sum([emb], self._pl('span_gather_indptr'), self._pl('span_gather_idx'), ts.xin)
            ops.fill_rows(W['span_encoder/transformer/cls'], ts.xin, ts.nseq, ts.S, 0)
            self.encoder_forward(ts, 'span_encoder/transformer', self.tables['span_rot'], self._pl('span_code'))
            self.fgemm(self._cls_view(ts.xf, ts.nseq, ts.S), 'span_encoder/transformer/cls_proj/kernel', self.s_cls,
                     bias=W['span_encoder/transformer/cls_proj/bias'])
        self.side_stream.wait_stream(main)
        self._on_side(span_fwd)
        # joint tower: one gather assembles [token embeddings | audio spans | vision tokens | zero padding]
        ops.segment_sum([emb, self.audio_seq, self.imgs_seq], self._pl('joint_gather_indptr'), self._pl('joint_gather_idx'), tj.xin)
        if not d.do_rotary:                # pretrain_model.py:146-148 + modeling.py:335-341: no coordinates, x += pe (every position, padding included)
            ops.add_rows_periodic(tj.xin, W['joint_transformer/pe'])          # one launch for all d.Nj sequences
        self.encoder_forward(tj, 'joint_transformer', self._joint_rot(), self._pl('joint_code'))
        self.fgemm(tj.xf, 'head/kernel', self.hj, bias=W['head/bias'])
        ops.segment_sum([self.hj], self._pl('pool_indptr'), self._pl('pool_idx'), self.Xpool)
        ops.segment_sum([self.a_cls], self._pl('acls_indptr'), self._pl('acls_idx'), self.acls_g)
        main.wait_stream(self.side_stream)
        # unit-normalise * temperature into the packed buffer E (pretrain_model.py:239-257)
        for src, names, si in self._norm_sections():
            o, n = self.sec[names[0]][0], sum(self.sec[k][1] for k in names)
            ops.unit_norm_scale_fwd(src, W['contrastive_scales'][si:si + 1], self.E[o:o + n], self.inv_norm[o:o + n])
        return self.E

    def _norm_sections(self):
        n1, n2 = self.sec['i2a_x'][1], self.sec['t2a_x'][1]
        return [(self.Xpool[:n1], ('i2a_x',), 0), (self.v_cls, ('i2a_y',), 0),
                (self.Xpool[n1:n1 + n2], ('t2a_x',), 1), (self.acls_g, ('t2a_y', 't2a_ye'), 1),
                (self.Xpool[n1 + n2:], ('s2s_x',), 2), (self.s_cls, ('s2s_y',), 2)]

    def outputs(self):
        """The reference's output dict (pretrain_model.py:240-244 after :246-257), views of E."""
        s = lambda k: self.E[self.sec[k][0]:self.sec[k][0] + self.sec[k][1]]
        return {'imgs_to_audio': {'x': s('i2a_x'), 'y': s('i2a_y')},
                'text_to_audio': {'x': s('t2a_x'), 'y': s('t2a_y'), 'y_extra': s('t2a_ye')},
                'stuff_to_span': {'x': s('s2s_x'), 'y': s('s2s_y'), '_sources': torch.as_tensor(self.plan['t2sp_src'])}}

    # ------------------------------------------------------------------------------------------ loss (+ its backward)
    def loss_and_grad_outputs(self, E_all=None, dE_all=None):
        """loss_fn_given_preds (pretrain_model.py:262-303) for this rank and dL/dE.
        world > 1: E_all [world,R,H] is the rank-major all-gather of every rank's E (:290); the gradient wrt this rank's
        queries goes to self.dE, the gradient wrt EVERY rank's keys to dE_all [world,R,H] (every row written), which the
        caller reduce-scatters (sum) and adds to self.dE before backward().  world == 1: both land in self.dE.
        Per (objective, direction): one logits GEMM, the fp32 LSE, and -- dL/dlogits split into bf16 hi + lo side by side so
        that the products keep 16 mantissa bits (its rows sum to zero) with fp32 accumulation over BOTH halves -- one GEMM
        for d(queries) and one for d(keys); 30 launches + 2 re-packs per step whatever the world size.
        Returns the fp32 device vector loss_acc[3] (per-objective losses of this rank)."""
        world, rank, H = self.world, self.rank, self.d.H
        if world == 1:
            E_all = self.E[None]
        else:
            assert E_all is not None and dE_all is not None and E_all.shape[0] == world
        ops.segment_sum([E_all.view(world * self.R, H)], self.kpack[0], self.kpack[1], self.Kcat)
        self.dE.zero_()
        self.loss_acc.zero_()
        self.diag.zero_()
        # The six (objective, direction) problems are independent: each of the three products runs as ONE grouped launch over all of
        # them (bf16 program; alone, every one of these 18 GEMMs is a 17-23 us latency chain of k-tiles on one or two workgroups).
        grouped = not self.fwd_only and self.dtype == BF16 and os.environ.get('MR_NO_LOSS_GROUPING') != '1'
        qs = [self.E[p['q_off']:p['q_off'] + p['Lq']] for p in self.cprob]
        if grouped:
            ops.gemm_grouped([self.gemm_args(q, self.Kcat[p['kb']:p['kb'] + p['V']], p['logits'], transB=True) for p, q in zip(self.cprob, qs)])
        for p, q in zip(self.cprob, qs):
            kb, V, ldv = p['kb'], p['V'], p['ldv']
            logits = p['logits']
            if not grouped:
                self.gemm(q, self.Kcat[kb:kb + V], logits, transB=True)               # rank-major columns (:290)
            is_s2s = p['name'] == 'stuff_to_span'
            ops.contrastive_lse(logits[:, :V], rank * p['nk'], 0.5 / p['Lq'], self._pl('t2sp_src') if is_s2s else None,
                                self.loss_acc[p['oi']:p['oi'] + 1], self.diag[p['di']] if is_s2s else None, self.lse_rows)
            if self.fwd_only:
                continue
            if self.dtype == F32:        # fp32 program: dL/dlogits (left in `logits` by the LSE kernel) feeds both products as it is; the
                dlf = logits[:, :V]      # "lo" rows of DK stay zero
                self.gemm(dlf, self.Kcat[kb:kb + V], self.dE[p['q_off']:p['q_off'] + p['Lq']])
                self.gemm(dlf, q, self.DK[kb:kb + V], transA=True)
                continue
            dl = p['dl']                                                           # [Lq, hi(ldv) | lo(ldv)]; pad columns stay 0
            ops.split_hilo_rows(logits[:, :V], dl[:, :V], dl[:, ldv:ldv + V])
            if not grouped:
                self.gemm(dl, self.Kcat[kb:kb + 2 * ldv], self.dE[p['q_off']:p['q_off'] + p['Lq']])     # d(query side) = dlogits . keys
                self.gemm(dl, q, self.DK[kb:kb + 2 * ldv], transA=True)               # d(key side)   = dlogits^T . queries (hi rows, lo rows)
        if grouped:
            ops.gemm_grouped([self.gemm_args(p['dl'], self.Kcat[p['kb']:p['kb'] + 2 * p['ldv']], self.dE[p['q_off']:p['q_off'] + p['Lq']])
                              for p in self.cprob])                                # d(query side) = dlogits . keys
            ops.gemm_grouped([self.gemm_args(p['dl'], q, self.DK[p['kb']:p['kb'] + 2 * p['ldv']], transA=True)
                              for p, q in zip(self.cprob, qs)])                    # d(key side)   = dlogits^T . queries (hi rows, lo rows)
        if not self.fwd_only:
            if world == 1 and self.dtype == F32:
                ops.segment_sum([self.DK], self.kunpack[0], self.kunpack[1], self.dE_keys)
                ops.add_(self.dE, self.dE_keys)
            elif world == 1:
                ops.segment_sum([self.DK], self.kunpack[0], self.kunpack[1], self.dE, accumulate=True)
            else:
                ops.segment_sum([self.DK], self.kunpack[0], self.kunpack[1], dE_all.view(world * self.R, H))
        return self.loss_acc

    # ------------------------------------------------------------------------------------------ backward
    def backward(self):
        assert not self.fwd_only, 'this program was built forward-only (PretrainEngine(..., train=True) for the fp32 backward)'
        self._backward()

    def _backward(self):
        """Backward of forward() given self.dE; fills self.p.grad (bf16, per rank, un-reduced).
        Three stages, in the order the flat gradient buffer is laid out (params.py), so that a data-parallel caller can
        all-reduce each finished range while the next stage runs: [scales, head, span, joint, token] -> audio -> vision."""
        self.backward_stage_joint()
        main = torch.cuda.current_stream()
        self.side_stream.wait_stream(main)
        self._on_side(self.backward_stage_audio)
        self.backward_stage_vision()
        main.wait_stream(self.side_stream)

    def backward_stage_joint(self):
        d, W, G, H = self.d, self.W, self.G, self.d.H
        tv, ta, tj, ts = self.tv, self.ta, self.tj, self.ts
        self.dls.zero_()
        n1, n2 = self.sec['i2a_x'][1], self.sec['t2a_x'][1]
        dsts = [self.dXpool[:n1], self.d_v_cls, self.dXpool[n1:n1 + n2], self.d_acls_g, self.dXpool[n1 + n2:], self.d_s_cls]
        for (src, names, si), dst in zip(self._norm_sections(), dsts):
            o, n = self.sec[names[0]][0], sum(self.sec[k][1] for k in names)
            ops.unit_norm_scale_bwd(src, W['contrastive_scales'][si:si + 1], self.inv_norm[o:o + n], self.dE[o:o + n], dst,
                                    self.dls[si:si + 1], self.dls_part, accumulate=True)
        if self.dtype == F32:
            G['contrastive_scales'].copy_(self.dls)
        else:
            ops.cast_f32_to_bf16(self.dls, G['contrastive_scales'])

        def span_bwd():
            # span tower (only its cls output is used: gradient enters at the CLS rows)
            Ds = self.Ds
            Ds.zero_()
            tr = self._begin_tower_reductions()
            cls_in = self._cls_view(ts.xf, ts.nseq, ts.S)
            self._t_colsum(tr, self.d_s_cls, G['span_encoder/transformer/cls_proj/bias'])
            self.gemm(cls_in, self.d_s_cls, G['span_encoder/transformer/cls_proj/kernel'], transA=True)
            self.gemm(self.d_s_cls, W['span_encoder/transformer/cls_proj/kernel'], self._cls_view(Ds, ts.nseq, ts.S), transB=True)
            Ds = self.encoder_backward(ts, 'span_encoder/transformer', self.tables['span_rot'], self._pl('span_code'), Ds, tr=tr)
            self._flush_tower_reductions(tr)
            ops.sum_rows_strided(Ds, ts.nseq, ts.S, 0, G['span_encoder/transformer/cls'])
            if Ds.data_ptr() != self.Ds.data_ptr():            # the joint tower reuses the rotating scratch: keep a copy
                self.Ds.copy_(Ds)
                Ds = self.Ds

            return Ds
        main = torch.cuda.current_stream()
        self.side_stream.wait_stream(main)
        Ds = self._on_side(span_bwd)
        # joint tower
        ops.segment_sum([self.dXpool], self._pl('poolT_indptr'), self._pl('poolT_idx'), self.d_hj)
        trj = self._begin_tower_reductions()
        self._t_colsum(trj, self.d_hj, G['head/bias'])
        Dj = self.Dj
        self.gemm(self.d_hj, W['head/kernel'], Dj, transB=True)
        Dj = self.encoder_backward(tj, 'joint_transformer', self._joint_rot(), self._pl('joint_code'), Dj, tr=trj,
                                   extra_wgrads=[(tj.xf, self.d_hj, G['head/kernel'])])
        self._flush_tower_reductions(trj)
        if not d.do_rotary:                # d pe[p] = sum over the joint sequences of the tower-input gradient at position p
            ops.segment_sum([Dj], self.pe_lists[0], self.pe_lists[1], G['joint_transformer/pe'])
        main.wait_stream(self.side_stream)
        # scatter-adds of the joint / span inputs, as segment sums over the planner's inverted lists
        ops.segment_sum([Dj, Ds], self._pl('embT_indptr'), self._pl('embT_idx'), G['token_encoder/Embed_0/embedding'])
        ops.segment_sum([Dj], self._pl('audT_indptr'), self._pl('audT_idx'), self.d_audio_seq)
        ops.segment_sum([Dj], self._pl('visT_indptr'), self._pl('visT_idx'), self.d_imgs_seq)
        if d.no_vision:                  # d(imgs_seq * 0) = 0
            self.d_imgs_seq.zero_()
        ops.segment_sum([self.d_acls_g], self._pl('aclsT_indptr'), self._pl('aclsT_idx'), self.d_a_cls)

    def backward_stage_audio(self):
        d, W, G, H = self.d, self.W, self.G, self.d.H
        ta = self.ta
        tr = self._begin_tower_reductions()
        Da = self._tower_with_pool_backward(ta, 'audio_encoder/transformer', 'audio_encoder/seq_attnpool', self.tables['audio_rot'],
                                            self.tables['audio_pool_rows'], self.a_qin, self.a_q, self.a_k, self.a_v, self.a_po,
                                            self.a_probs, self.d_audio_seq, self.d_a_cls, self.Da, tr=tr)
        Dp = self.cur.Dpatch[:d.Na * d.a_len]
        ops.segment_sum([Da], self.unpad_a[0], self.unpad_a[1], Dp)
        self._t_colsum(tr, Dp, G['audio_encoder/embedding/bias'])
        self._flush_tower_reductions(tr)
        self.gemm(self._audio2d, Dp, G['audio_encoder/embedding/kernel'], transA=True)
    def backward_stage_vision(self, layer_done=None):
        d, W, G, H = self.d, self.W, self.G, self.d.H
        tv = self.tv
        tr = self._begin_tower_reductions()
        Dv = self._tower_with_pool_backward(tv, 'vision_encoder/transformer', 'vision_encoder/seq_attnpool', self.tables['vit_rot'],
                                            self.tables['vit_pool_rows'], self.v_qin, self.v_q, self.v_k, self.v_v, self.v_po,
                                            self.v_probs, self.d_imgs_seq, self.d_v_cls, self.Dv, layer_done=layer_done, tr=tr)
        Dp = self.cur.Dpatch[:d.Nv * d.hw]
        ops.segment_sum([Dv], self.unpad_v[0], self.unpad_v[1], Dp)
        self._t_colsum(tr, Dp, G['vision_encoder/embedding/bias'])
        self._flush_tower_reductions(tr)
        self.gemm(self._images2d, Dp, G['vision_encoder/embedding/kernel'], transA=True)

    def loss_info(self):
        """Host dict like the reference's loss_info (pretrain_model.py:263-303) from the device accumulators."""
        la = self.loss_acc.tolist()
        dg = self.diag.tolist()
        info = {k: la[i] for i, (k, *_r) in enumerate(self.objectives)}
        for i, t in enumerate(['text2audio', 'audio2text', 'random_text']):
            info[f'_stuff_to_span_from_{t}'] = sum(dg[di][i] / (dg[di][3 + i] + 1e-5) / 2.0 for di in range(2))
        info['loss'] = sum(la)
        return info
