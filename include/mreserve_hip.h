/* mreserve_hip.h -- C-ABI of libmreserve_hip.so: the MI355X (gfx950) kernels behind the
 * MERLOT Reserve pretraining step.
 *
 * The reference (rowanz/merlot_reserve) has no FFI layer: its boundary for this path is the
 * Python API (MerlotReservePretrainer.apply / loss_fn_given_preds / train_step), whose
 * arithmetic is delegated to XLA.  The entry points below are what a binding for that path
 * binds instead of XLA; each cites the reference lines whose arithmetic it replaces
 * (M = mreserve/modeling.py, P = pretrain/pretrain_model.py, O = pretrain/optimization.py).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / C++ types.
 *  - every function returns 0 on success, a negative MR_E* code on failure;
 *    mr_last_error() gives a thread-local message.
 *  - all pointers are DEVICE pointers unless the name ends in _host; memory is owned by the
 *    caller; the library never allocates, never synchronises, never throws.
 *  - `stream` is a hipStream_t passed as void*; kernels are enqueued on it asynchronously
 *    (capturable into a hipGraph).
 *  - bf16 = raw uint16 payload (IEEE bfloat16); matrices are row-major with explicit leading
 *    dimensions in ELEMENTS; leading dims and N of bf16 matrices must be multiples of 8
 *    (16-byte vectors) -- pad on the host otherwise.
 */
#ifndef MRESERVE_HIP_H
#define MRESERVE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MR_OK 0
#define MR_EINVAL (-1)   /* bad argument / unsupported shape */
#define MR_ELAUNCH (-2)  /* HIP launch error */

#define MR_DT_BF16 0
#define MR_DT_F32 1

/* activation applied in the GEMM epilogue */
#define MR_ACT_NONE 0
#define MR_ACT_GELU1702 1 /* x*sigmoid(1.702x): M:240-241 */

int mr_version(void);
const char* mr_last_error(void);
/* ---- Handles (mr_create / mr_destroy).  The reference has no FFI; what a handle replaces is the process-global state an
 *      XLA client keeps for a jitted step (pretrain/train.py:98-120: one jax device, one compiled step).  A handle owns what the library
 *      would otherwise keep per process: the tuning OPTION SET below and, optionally, a split-K workspace on its device.
 *        mr_create(device, ws_bytes, &h)   options = the process defaults at this moment; ws_bytes > 0: hipMalloc'ed here (the one
 *                                          allocation the library makes), used by mr_gemm when mr_gemm_args.workspace is NULL
 *        mr_make_current(h)                the CALLING THREAD launches under h's options from now on (NULL: the process defaults);
 *                                          per thread, so two trainers on two threads cannot disturb each other
 *        mr_handle_set_option / mr_handle_get_option(h, name, ...)    h = NULL addresses the process defaults
 *        mr_destroy(h)                     after the caller has synchronised the streams that used it
 *      mr_set_option / mr_get_option are the round-1..3 entry points, kept as shims: they address the calling thread's current handle,
 *      or the process defaults when it has none.
 * Option names:
 *   "gemm_tile_n"   0 = choose per problem (default) | 96 | 128 | 192 | 256 : forces the output-tile width of the 256-row GEMM
 *   "gemm_group_tile_n"  0 = choose (default) | 128 | 256 : tile width of mr_gemm_grouped's shared launch
 *   "gemm_v1_only"  1 = route every GEMM to the small-tile kernel
 *   "gemm3"         1 = default: NT problems with enough tiles run on the ping-pong kernel (gemm3.hip) | 0 = off | 256 / 192 = on
 *                   for EVERY NT problem it can take, with that tile width (tests, A/B)
 *   "gemm3_phases"  0 = choose per problem (default) | 1 | 2 : barrier phases per k-tile of the ping-pong kernel (tests, A/B)
 *   "gemm_cus"      0 / 256 = default: persistent GEMM grids fill all 256 CUs | 64 .. 248 (multiple of 8): the forward / dgrad kernels
 *                   (gemm3.hip, gemm4.hip, gemm5.hip) launch that many workgroups (x 2 for gemm5), e.g. 240 = two CUs per XCD left to a
 *                   resident RCCL kernel (the data-parallel trainer sets it while gradient buckets are in flight)
 *   "gemm4"         -1 = default (on) | 0 | 1 : the one-wave-per-SIMD kernel (gemm4.hip) for the bias / residual / plain problems
 *                   "gemm3" admits
 *   "gemm5"         -1 = default: the few-tile problems (gemm5.hip: at most one 128 x 128 tile per CU -> that geometry; at most one
 *                   256 x 128 tile per CU and K <= 1024 -> two workgroups per CU) | 0 = never | 1 = every NT problem it can take on 256 x 128
 *                   tiles | 3 = on 128 x 128 tiles (tests, A/B);  "gemm5_stagger"  -1 = default | 0..3 : the
 *                   start phase of two workgroups sharing a CU
 *   "attn_onepass"  -1 = default: mr_attention_bwd runs its one-pass kernel (dQ, dK, dV from one sweep, one workgroup per (sequence,
 *                   head)) for 128 < S <= 256 | 0 = never (the dQ + dK / dV kernel pair) | 1 = whenever S <= 256 (tests, A/B)
 *   "attn_tile_modes" 1 = default: the masked attention kernels classify every (wave, 64-position tile) pair by the position codes -- all pairs
 *                   allowed: the unmasked instruction sequence; none allowed and every query valid: the tile is skipped; otherwise the general
 *                   path -- bit-identical results | 0 = the general path everywhere (tests, A/B)
 *   "gemm_trace"    1 = every GEMM launch records the kernel it was routed to (mr_last_gemm_kernel; bench.py's per-kernel table)
 *   "ln_impl"       1 = default: mr_layernorm_bwd runs the round-6 kernel (<= 128 registers, 512-thread workgroups: the whole grid resident; at most
 *                   512 partial rows) | 0 = round 5's (A/B).  mr_layernorm_bwd_nparts follows the option.
 *   "gemm_xpx"      0 = default: the XCD partition px x (8 / px) of a persistent GEMM's tile grid comes from the cost model | 1 | 2 | 4 | 8 forces px;
 *   "gemm_xpanel"   0 = default (8): tile columns per panel of an XCD cell's walk (experiment knobs: scripts/xcd_walk_ab.sh)
 * Environment variables (MR_GEMM3, MR_G3_PH, ...: experiment scripts) are read only by a library built with -DMR_DEBUG_ENV
 * (MR_DEBUG_ENV=1 python -m merlot_reserve_amd.build); the product build ignores the environment.
 * Unknown names return MR_EINVAL. */
typedef struct mr_handle_s* mr_handle;
int mr_create(int32_t device, int64_t ws_bytes, mr_handle* out);
int mr_destroy(mr_handle h);
int mr_make_current(mr_handle h);
mr_handle mr_get_current(void);
int mr_handle_set_option(mr_handle h, const char* name, int32_t value);
int mr_handle_get_option(mr_handle h, const char* name, int32_t* value);
int mr_set_option(const char* name, int value);
int mr_get_option(const char* name, int32_t* value);
/* the kernel (template instance) the calling thread's last mr_gemm / mr_gemm_grouped launch went to; "" unless "gemm_trace" is on */
const char* mr_last_gemm_kernel(void);

/* ---- GEMM with fused epilogue (flax Dense / DenseGeneral: M:228-236, 252-255, 371, 402, 453, 631;
 *      and their dgrad / wgrad) --------------------------------------------------------------
 * C[M,N] = op(A) . op(B), bf16 inputs, fp32 accumulation on MFMA.
 *   transA = 0: A stored [M,K] (lda = row stride)      transA = 1: A stored [K,M]
 *   transB = 0: B stored [K,N] (flax kernel layout)    transB = 1: B stored [N,K]
 * Epilogue, in this order (each optional, NULL / 0 disables):
 *   v = acc + bias[n]                                  (bias bf16 [N])
 *   v *= rot_tab[(m % rot_rows)*32 + (n & 63)]          for n < rot_cols and (n & 63) < 32
 *        (the reference's "rotary": a per-position diagonal scaling, M:116-144; rot_tab fp32)
 *   if c2: c2[m,n] = bf16(act'(v))  (= bf16(v) when act is NONE)   (saved for backward, same ldc / row map)
 *   v = act(v)
 *   v = bf16(v) + residual[m,n]                        (residual bf16, ld = ldr)
 *   v = bf16(v) * aux[m,n]                             (aux bf16 = the act'(v) saved by the forward GEMM, ld = ldaux)
 * Output row map: row m is stored at row (m / out_grp) * out_grp_stride + out_grp_off + m % out_grp
 * when out_grp > 0 (used to leave room for the CLS row, M:311-320), else at row m.
 * c_dtype = MR_DT_F32 supports bias only (used for the contrastive logits, P:293).
 * With a workspace and no epilogue beyond bias, problems with few 128x128 output tiles and a long K (every weight
 * gradient: K = tokens) are split along K over gridDim.y and summed by a second kernel (fixed order: deterministic).
 */
typedef struct {
    int64_t M, N, K;
    const void* A; int64_t lda; int32_t transA;
    const void* B; int64_t ldb; int32_t transB;
    void* C; int64_t ldc; int32_t c_dtype;
    const void* bias;
    const float* rot_tab; int64_t rot_rows; int64_t rot_cols;
    void* c2;
    int32_t act;
    const void* residual; int64_t ldr;
    const void* aux; int64_t ldaux;
    int64_t out_grp, out_grp_stride, out_grp_off;
    void* workspace; int64_t workspace_bytes; /* optional fp32 scratch: enables split-K for few-tile / long-K problems */
    /* optional: column sums of the STORED output (the bias gradient of the layer that produced the upstream gradient, i.e.
     * colsum of d(pre-activation) = the aux-multiplied dgrad), as per-(256-row tile, 64-row wave) partial rows
     * colsum[(4 * (m / 256) + (m % 256) / 64), n] fp32 with row stride ldcs -- mr_gemm_colsum_rows(M) rows -- to be summed by
     * mr_reduce_partials in a fixed order.  Only for problems mr_gemm_colsum_supported() accepts (256-row kernel, aux
     * epilogue, bf16 output); mr_gemm returns MR_EINVAL otherwise. */
    void* colsum; int64_t ldcs;
} mr_gemm_args;
int mr_gemm(const mr_gemm_args* args, void* stream);
int64_t mr_gemm_colsum_rows(int64_t M);
int32_t mr_gemm_colsum_supported(const mr_gemm_args* args);
/* count independent GEMMs in one call.  Weight gradients (every problem transA = 1 / transB = 0, one K, bf16, no epilogue, <= 20
 * problems, enough 256 x 256 tiles to fill most of the chip) run as ONE launch of the TN ping-pong kernel -- the weight gradients of
 * two (base) or four (large) transformer layers; <= 4 problems with identical layouts, bias-only epilogues and N % 128 == 0 as one
 * launch of the one-barrier kernel; anything else as `count` calls of mr_gemm.  Every problem is validated like mr_gemm's (non-null,
 * 16-byte aligned operands, M / N / K > 0) BEFORE any kernel is launched.  Which kernel a problem lands on is a function of its shape
 * and of the options; kernels that take the bias as the accumulators' initial value (gemm3 / gemm4 / gemm5) and those that add it in
 * the epilogue (gemm256, gemm.hip) round in a different order, so the same problem may differ by one bf16 ulp between routes. */
int mr_gemm_grouped(const mr_gemm_args* list, int32_t count, void* stream);

/* ---- LayerNorm (flax nn.LayerNorm eps=1e-5, fp32 stats, var = E[x^2]-E[x]^2: M:272,277,360,366) ---- */
int mr_layernorm_fwd(const void* x, int64_t ldx, const void* gamma, const void* beta, void* y, int64_t ldy,
                     float* mean, float* rstd, int64_t rows, int64_t H, float eps, void* stream);
/* dx = LN backward (+ dx_add when not NULL: the other gradient path of the residual stream; dx_add may alias dx);
 * dgamma/dbeta are reduced over rows into bf16 [H] each.  partials: fp32 workspace of mr_layernorm_bwd_workspace(H) bytes.
 * dgamma = dbeta = NULL defers that reduction: `partials` then holds mr_layernorm_bwd_nparts(rows) fp32 rows of
 * [dgamma(H) | dbeta(H)] for a later mr_reduce_partials job {partials, nparts, 2H, split = H, dgamma, dbeta}. */
int64_t mr_layernorm_bwd_workspace(int64_t H);
int64_t mr_layernorm_bwd_nparts(int64_t rows);
int mr_layernorm_bwd(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* gamma,
                     const float* mean, const float* rstd, void* dx, int64_t lddx, const void* dx_add, int64_t ldadd,
                     void* dgamma, void* dbeta, void* partials, int64_t rows, int64_t H, void* stream);

/* ---- column sum: out[n] = bf16(sum_m x[m,n])  (bias gradients) ----
 * out = NULL defers the final reduction: `partials` holds mr_colsum_nparts(rows) fp32 rows of [N]. */
int64_t mr_colsum_workspace(int64_t N);
int64_t mr_colsum_nparts(int64_t rows);
int mr_colsum(const void* x, int64_t ldx, int64_t rows, int64_t N, void* out, void* partials, void* stream);

/* ---- several deferred column reductions in ONE launch (the 2 LayerNorm + 2 bias gradients of a transformer layer):
 * out0[c] = bf16(sum_p partials[p, c]) for c < split, out1[c - split] for split <= c < ncols; fixed order. count <= 16. */
typedef struct {
    const float* partials;
    int32_t nparts, ncols, split;
    void* out0;
    void* out1;
} mr_reduce_job;
int mr_reduce_partials(const mr_reduce_job* jobs, int32_t count, void* stream);

/* ---- fused multi-head attention (M:188-200 + flax dot_product_attention_weights) ----
 * qkv: [nseq*S, 3*H] bf16 rows = [Q(H) | K(H) | V(H)], head h at columns h*64..h*64+63, q/k already
 * "rotary"-scaled by the QKV GEMM epilogue.  scores = (q/8).k + bias, bias = 0 where allowed else -1e10
 * (M:353-356), allowed(i,j) = code[i] == code[j] && code[i] >= 0; code == NULL means no mask.
 * out: [nseq*S, H] bf16; lse: [nseq, nh, S] fp32 (natural-log LSE of the biased scores).
 */
int mr_attention_fwd(const void* qkv, const int32_t* code, void* out, float* lse,
                     int64_t nseq, int64_t S, int64_t nh, void* stream);
/* dqkv [nseq*S, 3H] bf16 = gradient wrt the PRE-"rotary" qkv when rot_tab != NULL (the diagonal
 * scaling is applied to dq, dk on the way out), else wrt qkv as given.  delta: fp32 [nseq, nh, S] workspace (rowsum(dO * O),
 * computed by the dQ kernel for the dK / dV kernel; a row whose dO is entirely zero is marked -0.0 there -- an internal hand-off, not an output).  colsum (optional): fp32 [mr_attention_bwd_colsum_rows(nseq, S), 3H] -- per (sequence, 64- or
 * 128-position block) column sums of the stored dqkv, i.e. partial rows of the qkv bias gradient (M:228) to be summed in a
 * fixed order by mr_reduce_partials; every element is written. */
int mr_attention_bwd(const void* qkv, const int32_t* code, const void* out, const void* dout, const float* lse,
                     float* delta, void* dqkv, const float* rot_tab, int64_t rot_rows, float* colsum,
                     int64_t nseq, int64_t S, int64_t nh, void* stream);
int64_t mr_attention_bwd_colsum_rows(int64_t nseq, int64_t S);

/* ---- attention pooling core (flax MultiHeadDotProductAttention with 1 query and R keys: M:419-427, 467-472)
 * q [G, H]; k, v rows gathered by key_rows [G, R] (row indices into k/v, ld = ldkv); out [G, H].
 * probs [G, nh, R] fp32 saved for backward.
 */
int mr_poolattn_fwd(const void* q, const void* k, const void* v, int64_t ldkv, const int32_t* key_rows,
                    void* out, float* probs, int64_t G, int64_t R, int64_t nh, void* stream);
/* dq [G,H]; dk, dv [rows, ldkv]: every k/v row belongs to at most one group, rows not referenced must be
 * pre-zeroed by the caller. */
int mr_poolattn_bwd(const void* q, const void* k, const void* v, int64_t ldkv, const int32_t* key_rows,
                    const float* probs, const void* dout, void* dq, void* dk, void* dv,
                    int64_t G, int64_t R, int64_t nh, void* stream);

/* ---- row gather / segment sum (nn.Embed M:527, audio-span substitution M:685-695, vision tiling P:104,
 *      one_hot_pool M:541-567, gathers P:183-190, 233-236, and all of their transposes) ----
 * dst[i,:] = scale * sum_{j in [indptr[i], indptr[i+1])} src(indices[j]) for i < n_dst, H columns, bf16.
 * src(code): code < n0 -> src0 row code; code < n0+n1 -> src1 row code-n0; else src2 row code-n0-n1.
 * (src1/src2 may be NULL when n1 / n2 are 0).  Empty list -> zero row.  Accumulation in fp32, fixed order.
 * If accumulate != 0, dst += result (dst read as bf16).  dst_dtype MR_DT_BF16 or MR_DT_F32.
 */
int mr_segment_sum(const void* src0, int64_t ld0, int64_t n0, const void* src1, int64_t ld1, int64_t n1,
                   const void* src2, int64_t ld2, int64_t n2,
                   const int32_t* indptr, const int32_t* indices, void* dst, int64_t ldd, int32_t dst_dtype,
                   int64_t n_dst, int64_t H, float scale, int32_t accumulate, void* stream);

/* mean over R rows: dst[g,:] = (1/R) sum_r src[rows[g,r],:]  (M:423, 471); bwd scatters dst-grad/R back
 * (adds into dsrc, which holds the other gradient path already). */
int mr_rows_mean_fwd(const void* src, int64_t lds, const int32_t* rows, void* dst, int64_t G, int64_t R, int64_t H,
                     void* stream);
int mr_rows_mean_bwd(const void* ddst, const int32_t* rows, void* dsrc, int64_t lds, int64_t G, int64_t R, int64_t H,
                     void* stream);

/* ---- elementwise helpers ---- */
/* dst[r, 0:cols_out] = src[r, 0:cols_in] zero-padded (audio conv input 130 -> 136 columns, M:453) */
int mr_pad_cols(const void* src, int64_t cols_in, void* dst, int64_t cols_out, int64_t rows, void* stream);
/* dst[g*grp_stride + off, :] = vec  for g < ngroups  (CLS rows, M:316-320) */
int mr_fill_rows(const void* vec, void* dst, int64_t ldd, int64_t ngroups, int64_t grp_stride, int64_t off, int64_t H,
                 void* stream);
/* out[n] = bf16(sum_g src[g*grp_stride + off, n])   (gradient of the CLS parameter) */
int mr_sum_rows_strided(const void* src, int64_t lds, int64_t ngroups, int64_t grp_stride, int64_t off, int64_t H,
                        void* out, void* stream);
/* y = a + b (bf16, fp32 add) over n elements (n % 8 == 0) */
int mr_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream);

/* ---- unit_normalize * exp(min(log_scale, ln 100)/2)  (M:570-578, P:239-257) ----
 * x [rows, H] bf16 -> y bf16; log_scale: pointer to ONE bf16 (the working copy of contrastive_scales[i]).
 * inv_norm [rows] fp32 saved.  bwd returns dx and the temperature gradient: dlog_scale[0] (fp32) = (accumulate ? old : 0)
 * + this call's sum, formed from one partial per block of 4 rows (`partials`: (rows + 3) / 4 floats of caller scratch)
 * in a fixed order -- bitwise reproducible, no float atomics. */
int mr_unit_norm_scale_fwd(const void* x, int64_t ldx, const void* log_scale, void* y, int64_t ldy, float* inv_norm,
                           int64_t rows, int64_t H, void* stream);
int mr_unit_norm_scale_bwd(const void* x, int64_t ldx, const void* log_scale, const float* inv_norm, const void* dy,
                           int64_t lddy, void* dx, int64_t lddx, float* dlog_scale, int32_t accumulate, float* partials,
                           int64_t rows, int64_t H, void* stream);

/* ---- contrastive loss pieces (P:276-295) ----
 * logits [L, V] fp32 (from mr_gemm); numer[l] = logits[l, own_off + l]; lse over the V columns (fp32).
 * loss_out[0] += coef * sum_l (lse[l] - numer[l]);  when src != NULL, per-source sums / counts are added to
 * diag[0..5] are SET to the per-source sums / counts (three sums then three counts, P:296-300).  dlogits (in place) =
 * coef * (softmax - onehot(own)).  row_scratch: L floats of caller scratch (per-row lse - numer; the sums over rows are
 * taken from it in a fixed order: bitwise reproducible, no float atomics).
 */
int mr_contrastive_lse(float* logits, int64_t ldl, int64_t L, int64_t V, int64_t own_off, float coef,
                       const int32_t* src, float* loss_out, float* diag, float* row_scratch, void* stream);
/* ---- mask-LM branch of the loss (pretrain/pretrain_model.py:265-274: `text_preds`; no forward of the reference emits it) ----
 * logits [n, V] fp32 (leading dimension ldl), labels [n] int32 in [0, V); rows with label 0 are masked out.
 * out2[0] = -sum_r mask_r log_softmax(logits[r])[labels[r]] / sum_r mask_r;  out2[1] = sum_r mask_r.
 * dlogits (nullable, [n, V] fp32, ldl) = d out2[0] / d logits.  row_scratch: 3 n floats.  Sums in a fixed order (no atomics). */
int mr_masked_lm_xent(const float* logits, int64_t ldl, int64_t n, int64_t V, const int32_t* labels, float* out2,
                      float* dlogits, float* row_scratch, void* stream);
/* bf16 copy of an fp32 array */
int mr_cast_f32_to_bf16(const float* src, void* dst, int64_t n, void* stream);
/* hi = bf16(x), lo = bf16(x - hi): 16-bit-mantissa split of dL/dlogits, so that the two bf16 MFMA GEMMs on hi and lo
 * reproduce an fp32-operand product (the rows of dL/dlogits sum to zero; 8-bit rounding would break that cancellation) */
int mr_split_f32_to_bf16_hilo(const float* src, void* hi, void* lo, int64_t n, void* stream);
/* the same for a [rows, cols] matrix with row stride lds (elements) into two bf16 matrices with row stride ldo */
int mr_split_f32_to_bf16_hilo_rows(const float* src, int64_t lds, void* hi, void* lo, int64_t ldo, int64_t rows, int64_t cols,
                                   void* stream);

/* ---- optimizer: nan_to_num + bf16-state Adam + weight decay + schedule + apply (P:328, O:54-114, 180-195) ----
 * Flat buffers of n elements (n % 2048 == 0): master fp32 params, bf16 grads, bf16 mu, bf16 cube-coded nu.
 * decay_flag_per_block: one uint8 per 2048-element block (1 = weight decay applies: leaf ndim > 1, O:182-184).
 * sched = schedule(count) (O:117-137) and neg_lr = -learning_rate are computed on the host and applied in the
 * reference's order, u = (u * sched) * neg_lr.  bias_corr1/2 = 1 - beta^count_inc, or 1.0 when bias correction is off.
 * Writes master, the bf16 working copy (what the next forward reads, P:323-324), mu, nu.
 */
int mr_adam_bf16_update(float* master, void* work_bf16, const void* grad_bf16, void* mu_bf16, void* nu_bf16,
                        const uint8_t* decay_flag_per_block, int64_t n, double b1, double b2, float eps,
                        float weight_decay, float sched, float neg_lr, float bias_corr1, float bias_corr2, void* stream);
/* Finetuning chain (finetune/optimization.py:77-90: Adam with bias correction, subtract_old_weights, add_decayed_weights,
 * linear schedule, -lr): as above plus `orig_bf16`, the bf16 copy of the initial parameters; under the decay mask
 * (leaf ndim > 1 and size > 4096: finetune/optimization.py:74-75) u = u - wd * orig + wd * param. */
int mr_adam_bf16_update_finetune(float* master, void* work_bf16, const void* grad_bf16, void* mu_bf16, void* nu_bf16,
                                 const void* orig_bf16, const uint8_t* decay_flag_per_block, int64_t n, double b1, double b2,
                                 float eps, float weight_decay, float sched, float neg_lr, float bias_corr1,
                                 float bias_corr2, void* stream);
/* Either chain with the four per-step scalars read from DEVICE memory: hyper_dev = {sched, neg_lr, 1/bias_corr1,
 * 1/bias_corr2} (fp32).  The launch can then be part of a captured hipGraph, and a caller can run the optimizer on a
 * sub-range of the flat buffers (pointers offset by a multiple of 2048 elements, decay flags by the same number of blocks)
 * as soon as that range's gradients are final -- overlapped with the rest of backward.  orig_bf16 = NULL selects the
 * pretraining chain, non-NULL the finetuning chain. */
int mr_adam_bf16_update_dev(float* master, void* work_bf16, const void* grad_bf16, void* mu_bf16, void* nu_bf16,
                            const void* orig_bf16, const uint8_t* decay_flag_per_block, int64_t n, double b1, double b2,
                            float eps, float weight_decay, const float* hyper_dev, void* stream);
/* ---- softmax cross-entropy over C <= 64 classes (finetune/vcr/qa_qar_joint_finetune.py:188-195) ----
 * logits[r * row_stride + c * class_stride] fp32; loss_out += coef * sum_r -log_softmax(logits[r])[labels[r]];
 * correct_out (nullable) += coef * #(argmax == label); dlogits_bf16 (nullable, same strides) = coef * (softmax - onehot). */
int mr_softmax_xent(const float* logits, int64_t row_stride, int64_t class_stride, const int32_t* labels, int64_t rows,
                    int64_t C, float coef, float* loss_out, float* correct_out, void* dlogits_bf16, void* stream);
/* grads = nan_to_num(grads) in place (P:328), bf16, n % 8 == 0 */
int mr_nan_to_num_bf16(void* g, int64_t n, void* stream);
/* work_bf16 = bf16(master) (P:323-324) */
int mr_cast_f32_to_bf16_params(const float* master, void* work_bf16, int64_t n, void* stream);
/* Transposed working copies of Dense kernels: for every leaf of the table -- int32 quadruples {element offset in the flat buffers,
 * rows (= in features K), cols (= out features N), index of the leaf's first 64 x 64 tile}, rows and cols multiples of 64, leaves
 * in ascending tile order, in DEVICE memory -- workT[off + n * rows + k] = work[off + k * cols + n] for the tiles
 * [tile_lo, tile_hi) (a range of leaves: one gradient bucket).  Forward GEMMs (P:323-324 casts the params once per step; the
 * reference's XLA picks its own operand layouts) then read B = W^T [N, K] with transB = 1, like the dgrads read W [K, N]. */
int mr_transpose_leaves(const void* work_bf16, void* workT_bf16, const int32_t* leaves_dev, int32_t nleaf, int32_t tile_lo,
                        int32_t tile_hi, void* stream);

/* ---- fp32 forward path (use_bfloat16 = false: M:594; every zero-shot / feature caller runs fp32, M:999-1000) ----
 * Same operations as above with fp32 storage and fp32 arithmetic (v_mfma_f32_16x16x4_f32 / fp32 VALU): the forward; the backward entry points
 * (mr_f32_*_bwd, round 4) follow below.
 * Leading dims and H must be multiples of 4 (16-byte vectors) except mr_f32_gemm's lda / ldb (unaligned operands take a
 * scalar-load path). */
/* mr_gemm_args with A, B, C, bias, residual, c2, aux = fp32; c_dtype must be MR_DT_F32; workspace unused (NULL).
 * Epilogue order: + bias, * rot_tab, gelu1702 (with the gelu' copy to c2 when given: the fp32 training step), + residual (residual addressed with
 * the mapped output row), * aux. */
int mr_f32_gemm(const mr_gemm_args* args, void* stream);
int mr_f32_layernorm_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y, int64_t ldy,
                         int64_t rows, int64_t H, float eps, void* stream);
/* Attention forward under an ARBITRARY boolean mask (mreserve/modeling.py:303, 350-356: TransformerEncoder's attention_mask [*, L, L]; bias 0 where
 * mask != 0, -1e10 elsewhere; a row without any allowed key is uniform over the L keys, as in the reference).  qkv / out as mr_attention_fwd, in bf16
 * (MR_DT_BF16) or fp32 (MR_DT_F32); mask: uint8 [nseq, S, S], shared by the heads.  Forward only (the zero-shot API surface): every mask the model builds
 * itself has the block form the code-based kernels take. */
int mr_attention_fwd_dense_mask(const void* qkv, int32_t dtype, const uint8_t* mask, void* out, int64_t nseq, int64_t S, int64_t nh, void* stream);
/* Backward of mr_attention_fwd_dense_mask (round 6; the reference differentiates TransformerEncoder under any mask: mreserve/modeling.py:343-358).
 * dout [nseq*S, H], dqkv [nseq*S, 3H] in `dtype`; every element of dqkv is written.  rot_tab (optional, fp32 [rot_rows, 32]): dq / dk are multiplied by
 * the "rotary" scales on the way out, like mr_attention_bwd.  workspace: mr_attention_bwd_dense_mask_workspace(nseq, S, nh) bytes (the probabilities
 * and score gradients, fp32 [nseq, nh, S, S] each).  The bias gradient is the column sum of dqkv (mr_colsum).  S <= 1920. */
int64_t mr_attention_bwd_dense_mask_workspace(int64_t nseq, int64_t S, int64_t nh);
int mr_attention_bwd_dense_mask(const void* qkv, int32_t dtype, const uint8_t* mask, const void* dout, void* dqkv, const float* rot_tab,
                                int64_t rot_rows, void* workspace, int64_t nseq, int64_t S, int64_t nh, void* stream);
/* qkv [nseq*S, 3H] fp32 (layout of mr_attention_fwd); lse may be NULL */
int mr_f32_attention_fwd(const float* qkv, const int32_t* code, float* out, float* lse, int64_t nseq, int64_t S,
                         int64_t nh, void* stream);
int mr_f32_poolattn_fwd(const float* q, const float* k, const float* v, int64_t ldkv, const int32_t* key_rows, float* out,
                        int64_t G, int64_t R, int64_t nh, void* stream);
int mr_f32_segment_sum(const float* src0, int64_t ld0, int64_t n0, const float* src1, int64_t ld1, int64_t n1,
                       const float* src2, int64_t ld2, int64_t n2, const int32_t* indptr, const int32_t* indices,
                       float* dst, int64_t ldd, int64_t n_dst, int64_t H, float scale, void* stream);
int mr_f32_rows_mean_fwd(const float* src, int64_t lds, const int32_t* rows, float* dst, int64_t G, int64_t R, int64_t H,
                         void* stream);
/* y = x / sqrt(sum x^2 + 1e-5) * exp(min(*log_scale, ln 100) / 2); log_scale NULL -> plain unit_normalize (M:570-578) */
int mr_f32_unit_norm_scale_fwd(const float* x, int64_t ldx, const float* log_scale, float* y, int64_t ldy, int64_t rows,
                               int64_t H, void* stream);
int mr_f32_fill_rows(const float* vec, float* dst, int64_t ldd, int64_t ngroups, int64_t grp_stride, int64_t off,
                     int64_t H, void* stream);

/* ---- fp32 BACKWARD (round 4): the reference's `use_bfloat16 = False` training arithmetic -- its GPU-debug mode, pretrain/train.py:61-67;
 *      the `use_bfloat16_grads = False` branch of train_step, pretrain/pretrain_model.py:323-333.  Every operand fp32; fixed-order
 *      reductions; written for exactness, not speed.  mr_f32_gemm (above) takes the training epilogues too: c2 = gelu'(v) beside
 *      act = GELU (v itself without an activation), aux = elementwise multiplier of the stored output. */
/* LayerNorm backward (M:272,277,360,366): statistics recomputed from x; dx may alias dy; dx_add (optional) is added to dx;
 * dgamma / dbeta (both or neither) are full column reductions in row order; stat_ws: 2 * rows floats of scratch. */
int mr_f32_layernorm_bwd(const float* dy, int64_t lddy, const float* x, int64_t ldx, const float* gamma, float* dx, int64_t lddx,
                         const float* dx_add, int64_t ldadd, float* dgamma, float* dbeta, float* stat_ws, int64_t rows, int64_t H,
                         float eps, void* stream);
int mr_f32_colsum(const float* x, int64_t ldx, int64_t rows, int64_t N, float* out, void* stream);
/* backward of mr_f32_attention_fwd (M:188-200 with the -1e10 bias of M:353-356): dqkv [nseq*S, 3H] (q / k thirds scaled by rot_tab's
 * first 32 dims per head when given, like mr_attention_bwd); delta [nseq, nh, S] scratch. */
int mr_f32_attention_bwd(const float* qkv, const int32_t* code, const float* out, const float* dout, const float* lse, float* delta,
                         float* dqkv, const float* rot_tab, int64_t rot_rows, int64_t nseq, int64_t S, int64_t nh, void* stream);
/* backward of mr_f32_poolattn_fwd (probabilities recomputed); dk / dv are written at the groups' key rows only */
int mr_f32_poolattn_bwd(const float* q, const float* k, const float* v, int64_t ldkv, const int32_t* key_rows, const float* dout,
                        float* dq, float* dk, float* dv, int64_t G, int64_t R, int64_t nh, void* stream);
/* dst[rows[g, r]] += dsrc[g] / R */
int mr_f32_rows_mean_bwd(const float* dsrc, const int32_t* rows, float* dst, int64_t ldd, int64_t G, int64_t R, int64_t H, void* stream);
/* backward of mr_f32_unit_norm_scale_fwd: dx (+= when accumulate), *dls += d(log_scale); dls_rows: `rows` floats of scratch */
int mr_f32_unit_norm_scale_bwd(const float* x, int64_t ldx, const float* log_scale, const float* dy, int64_t lddy, float* dx, int64_t lddx,
                               int32_t accumulate, float* dls, float* dls_rows, int64_t rows, int64_t H, void* stream);
int mr_f32_sum_rows_strided(const float* x, int64_t ldx, int64_t ngroups, int64_t grp_stride, int64_t off, int64_t H, float* out, void* stream);
/* y = a * x + b * y;  jnp.nan_to_num in place (P:328) */
int mr_f32_axpby(float* y, const float* x, float a, float b, int64_t n, void* stream);
int mr_f32_nan_to_num(float* y, int64_t n, void* stream);
/* mr_adam_bf16_update_dev with fp32 gradients (no bf16 round trip, nan_to_num at fp32 range): the pretraining chain of O:180-190 */
int mr_adam_f32grad_update_dev(float* master, void* work_bf16, const float* grad_f32, void* mu_bf16, void* nu_bf16,
                               const uint8_t* decay_flag_per_block, int64_t n, double b1, double b2, float eps, float weight_decay,
                               const float* hyper_dev, void* stream);

/* ---- data-parallel collectives over RCCL / xGMI (one process per GPU) ------------------------------------------------
 * Replace the XLA collectives of the pmap'ed step: jax.lax.all_gather of the packed contrastive embeddings (P:290) and its
 * transpose in backward (a reduce-scatter), jax.lax.pmean of the bf16 gradient pytree (P:329) and of the fp32 metrics
 * (P:336).  A communicator belongs to the process's CURRENT device at mr_comm_init time; every call is asynchronous on the
 * given stream, never synchronises, and may be captured into a hipGraph together with the kernels around it.
 * RCCL (librccl.so.1) is resolved at run time from the copy already in the process; calls fail with MR_ELAUNCH and a
 * message if it is absent.  Counts are in elements.
 */
#define MR_COMM_UNIQUE_ID_BYTES 128
typedef struct mr_comm mr_comm;
/* rank 0 creates the id (ncclGetUniqueId) and hands its 128 bytes to every rank out of band */
int mr_comm_unique_id(void* id_out);
/* collective over all ranks: each calls it with the same id, its own rank, on its own device */
int mr_comm_init(int32_t rank, int32_t world, const void* unique_id, mr_comm** out);
int mr_comm_destroy(mr_comm* comm);
int32_t mr_comm_rank(const mr_comm* comm);
int32_t mr_comm_world(const mr_comm* comm);
/* buf[n] bf16 <- mean over ranks, in place (bf16 like pmean on the bf16 gradients, P:323-329) */
int mr_allreduce_mean_bf16(mr_comm* comm, void* buf, int64_t n, void* stream);
/* buf[n] fp32 <- mean over ranks (loss_info, P:336) */
int mr_allreduce_mean_f32(mr_comm* comm, float* buf, int64_t n, void* stream);
/* recv[world * n_per_rank] bf16: block r = rank r's send[n_per_rank]  (rank-major, as all_gather(...).reshape(-1, H)) */
int mr_allgather(mr_comm* comm, const void* send, void* recv, int64_t n_per_rank, void* stream);
/* recv[n_per_rank] bf16 = sum over ranks of their send[rank * n_per_rank ...]  (transpose of mr_allgather) */
int mr_reducescatter_sum(mr_comm* comm, const void* send, void* recv, int64_t n_per_rank, void* stream);
/* the same on fp32 elements (the fp32 training step, P:323-333 with use_bfloat16_grads = False; its all-gather moves bytes: mr_allgather on
 * 2 * n_per_rank) */
int mr_reducescatter_sum_f32(mr_comm* comm, const float* send, float* recv, int64_t n_per_rank, void* stream);

/* ---- host-side record I/O of the real-data input path (pretrain/dataloader.py:884 `tf.data.TFRecordDataset`; no GPU involved) ----
 * CRC-32C (Castagnoli), crc = running value (0 to start); the masked form TFRecord files store is rotr(crc, 15) + 0xa282ead8. */
uint32_t mr_crc32c(const void* data, int64_t n, uint32_t crc);
uint32_t mr_crc32c_masked(const void* data, int64_t n);
/* Walks a TFRecord byte image (uint64 length | masked crc of the length | data | masked crc of the data, little endian): returns the number of
 * records and writes the first `cap` (offset of the data, its length) pairs; verify != 0 checks both checksums of every record.
 * < 0 (MR_EINVAL, text in mr_last_error) for a truncated or corrupted image. */
int64_t mr_tfrecord_scan(const void* buf, int64_t n, int64_t* offsets, int64_t* lengths, int64_t cap, int32_t verify);

#ifdef __cplusplus
}
#endif
#endif
