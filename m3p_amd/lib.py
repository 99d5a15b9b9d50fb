"""ctypes binding of libm3p_hip.so (C ABI declared in include/m3p_hip.h).

The HIP library is the product: there is NO fallback.  If the shared object is missing
or a kernel reports an error, the call raises — a silent eager/PyTorch path would void
every parity and performance claim made for this package.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# M3P_HIP_LIB: developer override used for A/B runs of two builds inside one GPU session
LIB_PATH = os.environ.get('M3P_HIP_LIB') or os.path.join(_HERE, 'libm3p_hip.so')

EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_DROP_RES, EPI_RES, EPI_DGELU, EPI_MUL, EPI_MULQ, EPI_BIAS_GELUQ, EPI_BIAS_LSE = range(10)
# kernel ids of m3p_gemm_nt_plan / m3p_gemm_wgrad_plan (include/m3p_hip.h)
KERN_NT_SKINNY, KERN_NT_W8, KERN_NT_W8_QUEUE, KERN_NT_W4, KERN_NT_RING, KERN_NT_128 = range(1, 7)
KERN_WGRAD_W4_CHUNKS, KERN_WGRAD_W4_TILES, KERN_WGRAD_RING, KERN_WGRAD_128 = range(10, 14)


class M3PError(RuntimeError):
    pass


class Epilogue(C.Structure):
    """struct M3PEpilogue (include/m3p_hip.h)."""
    _fields_ = [
        ('bias', C.c_void_p), ('aux', C.c_void_p), ('out2', C.c_void_p), ('colsum', C.c_void_p),
        ('ld_aux', C.c_int32), ('ld_out2', C.c_int32), ('scale_cols', C.c_int32), ('scale', C.c_float),
        ('alpha', C.c_float), ('seed', C.c_uint32), ('thresh24', C.c_uint32), ('inv_keep', C.c_float),
        ('descale_a', C.c_void_p), ('descale_b', C.c_void_p),
        ('out8', C.c_void_p), ('scale8', C.c_void_p), ('amax8', C.c_void_p), ('ld_out8', C.c_int32), ('out8_bf8', C.c_int32),
    ]


_p, _i, _f, _u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint32

# name -> (restype, argtypes); mirrors include/m3p_hip.h one to one
SIGNATURES = {
    'm3p_version': (C.c_char_p, []),
    'm3p_gemm_nt_bf16': (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, C.POINTER(Epilogue), _p]),
    'm3p_gemm_nt_fp8': (_i, [_p, _i, _i, _p, _i, _p, _i, _i, _i, _i, _i, C.POINTER(Epilogue), _p]),
    'm3p_quant_fp8': (_i, [_p, _i, _p, _i, _i, _i, _p, _p, _i, _p]),
    'm3p_quant_fp8_batch': (_i, [_p, _i, _i, _p]),
    'm3p_gemm_nt_streamk_f32': (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _f, _p]),
    'm3p_gemm_nn_streamk_f32': (_i, [_p, _i, _p, _i, _i, _p, _i, _i, _i, _i, _f, _p]),
    'm3p_gemm_nn_w4_f32': (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _f, _p, C.c_size_t, _p]),
    'm3p_gemm_wgrad_workspace_bytes': (C.c_size_t, []),
    'm3p_gemm_wgrad_bf16': (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _f, _p, C.c_size_t, _p]),
    'm3p_gemm_wgrad_store_bf16': (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _i, _f, _p, C.c_size_t, _p]),
    'm3p_gemm_wgrad_pair_bf16': (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _i, _i, _f, _p, C.c_size_t, _p]),
    'm3p_layernorm_fwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _p]),
    'm3p_layernorm_bwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _u32, _u32, _f, _p]),
    'm3p_attn_fwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _u32, _u32, _f, _p]),
    'm3p_attn_bwd': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _u32, _u32, _f, _p]),
    'm3p_attn_query_fwd': (_i, [_p, _i, _p, C.c_longlong, _i, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    'm3p_attn_rows_fwd': (_i, [_p, _i, _p, C.c_longlong, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _u32, _u32, _f, _p]),
    'm3p_attn_rows_bwd': (_i, [_p, _i, _p, C.c_longlong, _i, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _i, _i, _f, _u32, _u32, _f, _p]),
    'm3p_cast_f32_bf16': (_i, [_p, _p, C.c_longlong, _p]),
    'm3p_embed_assemble_fwd': (_i, [_p] * 19 + [_i, _i, _i, _i, _u32, _u32, _u32, _f, _p, _p]),
    'm3p_embed_image_rows_fwd': (_i, [_p] * 10 + [_i, _i, _i, _u32, _u32, _f, _p]),
    'm3p_embed_assemble_bwd': (_i, [_p] * 24 + [_i, _i, _i, _i, _i, _u32, _u32, _u32, _f, _i, _p]),
    'm3p_dropout_rows': (_i, [_p, _i, _p, _i, _p, _i, _i, _i, _u32, _u32, _u32, _u32, _f, _p]),
    'm3p_glu_fwd': (_i, [_p, _i, _p, _i, _i, _p]),
    'm3p_glu_bwd': (_i, [_p, _i, _p, _p, _i, _i, _p]),
    'm3p_gather_rows': (_i, [_p, _p, _p, _i, _i, _p]),
    'm3p_scatter_add_rows': (_i, [_p, _p, _p, _i, _i, _p]),
    'm3p_scatter_add_token_rows': (_i, [_p, _i, _p, _p, _i, _i, _i, _p]),
    'm3p_ce_fwd_bwd': (_i, [_p, _i, _i, _i, _p, _p, _p, _f, _f, _p]),
    'm3p_ce_colsum_workspace_bytes': (C.c_size_t, [_i, _i]),
    'm3p_ce_fwd_bwd_colsum': (_i, [_p, _i, _i, _i, _p, _p, _p, _f, _p, _p, C.c_size_t, _p]),
    'm3p_ce_lse_from_blocks': (_i, [_p, _i, _i, _p, _i, _p, _p, _p, _p, _p]),
    'm3p_ce_bwd_colsum': (_i, [_p, _i, _i, _i, _p, _p, _f, _p, _p, C.c_size_t, _p]),
    'm3p_colsum_bf16': (_i, [_p, _i, _i, _i, _p, _p, _p]),
    'm3p_sumsq_f32': (_i, [_p, C.c_longlong, _p, _p]),
    'm3p_sumsq_ranges_f32': (_i, [_p, _p, _p, _i, _p, _p]),
    'm3p_adam_step': (_i, [_p, _p, _p, _p, _p, C.c_longlong, _f, _f, _f, _f, _f, _f, _p, _f, _f, _i, _p]),
    'm3p_adam_step_ranges': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _f, _f, _f, _f, _p, _f, _f, _p]),
    'm3p_itm_score_fwd': (_i, [_p, _p, _p, _p, _p, _i, _i, _p]),
    'm3p_itm_score_bwd': (_i, [_p, _p, _p, _p, _p, _i, _p, _p, _p, _i, _i, _p]),
    'm3p_gelu_bwd': (_i, [_p, _p, _p, C.c_longlong, _p]),
    'm3p_mse_fwd_bwd': (_i, [_p, _i, _p, _i, _p, _p, _i, _i, _f, _p]),
    'm3p_gelu_fwd': (_i, [_p, _p, _p, C.c_longlong, _p]),
    'm3p_gelu_fwd_gq': (_i, [_p, _p, _p, _i, _i, _p]),
    'm3p_gelu_fwd_q8': (_i, [_p, _p, _p, C.c_longlong, _p, _p, _p]),
    'm3p_transpose_batch_bf16': (_i, [_p, _i, _i, _p]),
    'm3p_transpose_bf16': (_i, [_p, _p, _i, _i, _i, _i, _p]),
    'm3p_set_persistent_grid': (_i, [_i]),
    'm3p_set_tile_queue': (_i, [_p, _i]),
    'm3p_gemm_nt_plan': (_i, [_i, _i, _i, _i]),
    'm3p_gemm_wgrad_plan': (_i, [_i, _i, _i]),
    'm3p_debug_set_variant': (None, [_i]),
    'm3p_debug_attn_variant': (None, [_i]),
    'm3p_seq_masks': (_i, [_p, _p, _i, _i, _p, _p, _p]),
    'm3p_mask_to_rows': (_i, [_p, _i, _i, C.c_longlong, C.c_longlong, C.c_longlong, _i, _p, _i, _p]),
    'm3p_cast_rows_f32_bf16': (_i, [_p, C.c_longlong, C.c_longlong, _i, _i, _i, _p, _p]),
    'm3p_scale_bf16_dev': (_i, [_p, _i, _p, _p, C.c_longlong, _p]),
    'm3p_axpy_dev_f32': (_i, [_p, _p, _p, C.c_longlong, _p]),
    'm3p_itm_loss_fwd_bwd': (_i, [_p, _p, _i, _i, _f, _f, _p, _p, _p]),
    'm3p_probe_mfma_16x16x32': (_i, [_p, _p, _p, _p, _p]),
    'm3p_probe_mfma_fp8_16x16x128': (_i, [_p, _p, _p, _i, _p]),
    'm3p_probe_tr16': (_i, [_p, _p, _p]),
    'm3p_probe_permlane16_swap': (_i, [_p, _p]),
}

_lib = None


def load():
    """Load the library once; raises M3PError if it has not been built
    (``python -c 'import __graft_entry__ as g; g.build()'`` or ``make -C m3p_amd/csrc``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise M3PError('libm3p_hip.so not found at %s — build it first (make -C m3p_amd/csrc); '
                       'there is no fallback path' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so is stale
        fn.restype = res
        fn.argtypes = args
    _lib_tq = os.environ.get('M3P_TILE_QUEUE', '0') != '0'     # developer switch: dynamic tile queues without data parallelism
    if os.environ.get('M3P_VARIANT'):      # developer switch between GEMM kernel generations (A/B runs)
        lib.m3p_debug_set_variant(int(os.environ['M3P_VARIANT']))
    if os.environ.get('M3P_ATTN_VARIANT'):
        lib.m3p_debug_attn_variant(int(os.environ['M3P_ATTN_VARIANT']))
    _lib = lib
    if _lib_tq and torch.cuda.is_available():
        from . import ops
        ops.set_tile_queue(True)
    return lib


def check(code, what):
    if code != 0:
        kind = {-1: 'M3P_EINVAL (bad shape/alignment)', -2: 'M3P_ENOTIMPL', -3: 'M3P_ENOMEM'}.get(code, 'hipError_t %d' % code)
        raise M3PError('%s failed: %s' % (what, kind))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def num_cus():
    """Compute units of the current device, rounded down to whole XCD octets like the kernels' own count."""
    n = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    return max(n - n % 8, 8)


def thresh24(p):
    """Dropout probability -> 24-bit drop threshold used by every kernel (csrc/common.hpp)."""
    return int(round(float(p) * (1 << 24)))
