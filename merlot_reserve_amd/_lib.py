"""ctypes binding of libmreserve_hip.so (include/mreserve_hip.h).  There is NO fallback: if the library is missing
or a call fails, this raises -- the product path never routes through a CPU implementation."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MR_LIB', os.path.join(HERE, 'libmreserve_hip.so'))    # MR_LIB: A/B a kernel variant in one run

i64, i32, f32, f64, vp = C.c_int64, C.c_int32, C.c_float, C.c_double, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [('M', i64), ('N', i64), ('K', i64),
                ('A', vp), ('lda', i64), ('transA', i32),
                ('B', vp), ('ldb', i64), ('transB', i32),
                ('C', vp), ('ldc', i64), ('c_dtype', i32),
                ('bias', vp),
                ('rot_tab', vp), ('rot_rows', i64), ('rot_cols', i64),
                ('c2', vp),
                ('act', i32),
                ('residual', vp), ('ldr', i64),
                ('aux', vp), ('ldaux', i64),
                ('out_grp', i64), ('out_grp_stride', i64), ('out_grp_off', i64),
                ('workspace', vp), ('workspace_bytes', i64),
                ('colsum', vp), ('ldcs', i64)]


class ReduceJob(C.Structure):
    """mirrors mr_reduce_job"""
    _fields_ = [('partials', vp), ('nparts', i32), ('ncols', i32), ('split', i32), ('out0', vp), ('out1', vp)]


# name -> (restype, argtypes); every symbol declared in include/mreserve_hip.h
PROTOTYPES = {
    'mr_version': (i32, []),
    'mr_set_option': (i32, [C.c_char_p, i32]),
    'mr_get_option': (i32, [C.c_char_p, C.POINTER(i32)]),
    'mr_create': (i32, [i32, i64, C.POINTER(vp)]),
    'mr_destroy': (i32, [vp]),
    'mr_make_current': (i32, [vp]),
    'mr_get_current': (vp, []),
    'mr_handle_set_option': (i32, [vp, C.c_char_p, i32]),
    'mr_handle_get_option': (i32, [vp, C.c_char_p, C.POINTER(i32)]),
    'mr_last_gemm_kernel': (C.c_char_p, []),
    'mr_last_error': (C.c_char_p, []),
    'mr_gemm': (i32, [C.POINTER(GemmArgs), vp]),
    'mr_gemm_grouped': (i32, [C.POINTER(GemmArgs), i32, vp]),
    'mr_gemm_colsum_rows': (i64, [i64]),
    'mr_gemm_colsum_supported': (i32, [C.POINTER(GemmArgs)]),
    'mr_layernorm_fwd': (i32, [vp, i64, vp, vp, vp, i64, vp, vp, i64, i64, f32, vp]),
    'mr_layernorm_bwd_workspace': (i64, [i64]),
    'mr_layernorm_bwd': (i32, [vp, i64, vp, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, i64, i64, vp]),
    'mr_colsum_workspace': (i64, [i64]),
    'mr_colsum_nparts': (i64, [i64]),
    'mr_layernorm_bwd_nparts': (i64, [i64]),
    'mr_reduce_partials': (i32, [C.POINTER(ReduceJob), i32, vp]),
    'mr_colsum': (i32, [vp, i64, i64, i64, vp, vp, vp]),
    'mr_attention_fwd': (i32, [vp, vp, vp, vp, i64, i64, i64, vp]),
    'mr_attention_bwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, i64, i64, vp]),
    'mr_attention_bwd_colsum_rows': (i64, [i64, i64]),
    'mr_poolattn_fwd': (i32, [vp, vp, vp, i64, vp, vp, vp, i64, i64, i64, vp]),
    'mr_poolattn_bwd': (i32, [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    'mr_segment_sum': (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, i64, i32, i64, i64, f32, i32, vp]),
    'mr_rows_mean_fwd': (i32, [vp, i64, vp, vp, i64, i64, i64, vp]),
    'mr_rows_mean_bwd': (i32, [vp, vp, vp, i64, i64, i64, i64, vp]),
    'mr_pad_cols': (i32, [vp, i64, vp, i64, i64, vp]),
    'mr_fill_rows': (i32, [vp, vp, i64, i64, i64, i64, i64, vp]),
    'mr_sum_rows_strided': (i32, [vp, i64, i64, i64, i64, i64, vp, vp]),
    'mr_add_bf16': (i32, [vp, vp, vp, i64, vp]),
    'mr_unit_norm_scale_fwd': (i32, [vp, i64, vp, vp, i64, vp, i64, i64, vp]),
    'mr_unit_norm_scale_bwd': (i32, [vp, i64, vp, vp, vp, i64, vp, i64, vp, i32, vp, i64, i64, vp]),
    'mr_contrastive_lse': (i32, [vp, i64, i64, i64, i64, f32, vp, vp, vp, vp, vp]),
    'mr_masked_lm_xent': (i32, [vp, i64, i64, i64, vp, vp, vp, vp, vp]),
    'mr_split_f32_to_bf16_hilo_rows': (i32, [vp, i64, vp, vp, i64, i64, i64, vp]),
    'mr_cast_f32_to_bf16': (i32, [vp, vp, i64, vp]),
    'mr_split_f32_to_bf16_hilo': (i32, [vp, vp, vp, i64, vp]),
    'mr_adam_bf16_update': (i32, [vp, vp, vp, vp, vp, vp, i64, f64, f64, f32, f32, f32, f32, f32, f32, vp]),
    'mr_adam_bf16_update_finetune': (i32, [vp, vp, vp, vp, vp, vp, vp, i64, f64, f64, f32, f32, f32, f32, f32, f32, vp]),
    'mr_adam_bf16_update_dev': (i32, [vp, vp, vp, vp, vp, vp, vp, i64, f64, f64, f32, f32, vp, vp]),
    'mr_softmax_xent': (i32, [vp, i64, i64, vp, i64, i64, f32, vp, vp, vp, vp]),
    'mr_nan_to_num_bf16': (i32, [vp, i64, vp]),
    'mr_cast_f32_to_bf16_params': (i32, [vp, vp, i64, vp]),
    'mr_transpose_leaves': (i32, [vp, vp, vp, i32, i32, i32, vp]),
    'mr_f32_gemm': (i32, [C.POINTER(GemmArgs), vp]),
    'mr_f32_layernorm_fwd': (i32, [vp, i64, vp, vp, vp, i64, i64, i64, f32, vp]),
    'mr_f32_attention_fwd': (i32, [vp, vp, vp, vp, i64, i64, i64, vp]),
    'mr_attention_fwd_dense_mask': (i32, [vp, i32, vp, vp, i64, i64, i64, vp]),
    'mr_attention_bwd_dense_mask_workspace': (i64, [i64, i64, i64]),
    'mr_attention_bwd_dense_mask': (i32, [vp, i32, vp, vp, vp, vp, i64, vp, i64, i64, i64, vp]),
    'mr_f32_poolattn_fwd': (i32, [vp, vp, vp, i64, vp, vp, i64, i64, i64, vp]),
    'mr_f32_segment_sum': (i32, [vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, i64, i64, i64, f32, vp]),
    'mr_f32_rows_mean_fwd': (i32, [vp, i64, vp, vp, i64, i64, i64, vp]),
    'mr_f32_unit_norm_scale_fwd': (i32, [vp, i64, vp, vp, i64, i64, i64, vp]),
    'mr_f32_fill_rows': (i32, [vp, vp, i64, i64, i64, i64, i64, vp]),
    'mr_f32_layernorm_bwd': (i32, [vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, vp, vp, i64, i64, f32, vp]),
    'mr_f32_colsum': (i32, [vp, i64, i64, i64, vp, vp]),
    'mr_f32_attention_bwd': (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i64, i64, i64, i64, vp]),
    'mr_f32_poolattn_bwd': (i32, [vp, vp, vp, i64, vp, vp, vp, vp, vp, i64, i64, i64, vp]),
    'mr_f32_rows_mean_bwd': (i32, [vp, vp, vp, i64, i64, i64, i64, vp]),
    'mr_f32_unit_norm_scale_bwd': (i32, [vp, i64, vp, vp, i64, vp, i64, i32, vp, vp, i64, i64, vp]),
    'mr_f32_sum_rows_strided': (i32, [vp, i64, i64, i64, i64, i64, vp, vp]),
    'mr_f32_axpby': (i32, [vp, vp, f32, f32, i64, vp]),
    'mr_f32_nan_to_num': (i32, [vp, i64, vp]),
    'mr_adam_f32grad_update_dev': (i32, [vp, vp, vp, vp, vp, vp, i64, f64, f64, f32, f32, vp, vp]),
    'mr_comm_unique_id': (i32, [vp]),
    'mr_comm_init': (i32, [i32, i32, vp, C.POINTER(vp)]),
    'mr_comm_destroy': (i32, [vp]),
    'mr_comm_rank': (i32, [vp]),
    'mr_comm_world': (i32, [vp]),
    'mr_allreduce_mean_bf16': (i32, [vp, vp, i64, vp]),
    'mr_allreduce_mean_f32': (i32, [vp, vp, i64, vp]),
    'mr_allgather': (i32, [vp, vp, vp, i64, vp]),
    'mr_reducescatter_sum': (i32, [vp, vp, vp, i64, vp]),
    'mr_reducescatter_sum_f32': (i32, [vp, vp, vp, i64, vp]),
    'mr_crc32c': (C.c_uint32, [vp, i64, C.c_uint32]),
    'mr_crc32c_masked': (C.c_uint32, [vp, i64]),
    'mr_tfrecord_scan': (i64, [vp, i64, vp, vp, i64, i32]),
}

_lib = None


class MreserveHipError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises if it has not been built: run `python -m merlot_reserve_amd.build`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MreserveHipError(f'{LIB_PATH} not found: the HIP extension is not built '
                               f'(python -m merlot_reserve_amd.build). There is no CPU fallback.')
    # torch ships its own HIP runtime: it must be in the process BEFORE this library resolves libamdhip64, otherwise two
    # runtimes coexist and launches from here see no device
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)       # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mr_last_error()
        raise MreserveHipError(f'{what} failed ({rc}): {msg.decode() if msg else ""}')
