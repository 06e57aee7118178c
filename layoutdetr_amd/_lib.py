"""ctypes binding of the C-ABI kernel library (include/ldetr_hip.h).

The product path has no CPU fallback: if `libldetr_hip.so` is missing or a symbol is absent the
import of any op raises.  (The reference silently falls back to its `_ref` path only for non-CUDA
tensors, torch_utils/ops/bias_act.py:85-87; here a non-GPU tensor is an error.)
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_ubyte, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDETR_LIB: development aid (tools/build_variant.sh) -- an alternative build of the same library, e.g. to A/B a kernel change on one box
ABI_VERSION = 24   # include/ldetr_hip.h; csrc/ldetr_core.cpp
LIB_PATH = os.environ.get('LDETR_LIB') or os.path.join(_HERE, 'lib', 'libldetr_hip.so')


class Tensor4(Structure):
    _fields_ = [('N', c_int), ('C', c_int), ('H', c_int), ('W', c_int),
                ('sn', c_int64), ('sc', c_int64), ('sh', c_int64), ('sw', c_int64)]


class GemmDesc(Structure):
    _fields_ = [('A', c_void_p), ('lda', c_int64), ('ta', c_int), ('B', c_void_p), ('ldb', c_int64), ('tb', c_int),
                ('C', c_void_p), ('ldc', c_int64), ('M', c_int), ('N', c_int), ('K', c_int), ('splitk', c_int),
                ('ep', c_void_p), ('pix_per_sample', c_int)]


class P3Epilogue(Structure):
    _fields_ = [('alpha', c_float), ('col_scale', c_void_p), ('col_bias', c_void_p), ('residual_p3', c_void_p),
                ('residual_f32', c_void_p), ('relu_mask_p3', c_void_p), ('relu', c_int)]


class Epilogue(Structure):
    _fields_ = [('alpha', c_float), ('col_scale', c_void_p), ('col_bias', c_void_p), ('samp_scale', c_void_p),
                ('samp_ld', c_int64), ('residual', c_void_p), ('ldr', c_int64), ('act', c_int),
                ('act_alpha', c_float), ('act_gain', c_float), ('mask_src', c_void_p), ('ldm', c_int64),
                ('mask_mode', c_int), ('out_scale', c_float), ('p_drop', c_float), ('seed', c_uint64),
                ('seed_ptr', c_void_p), ('accumulate', c_int), ('a_rowsum', c_void_p)]


class LnArgs(Structure):
    """ldetr_ln_args (include/ldetr_hip.h): one LayerNorm problem of a group launch."""
    _fields_ = [('x', c_void_p), ('r', c_void_p), ('gamma', c_void_p), ('beta', c_void_p),
                ('y', c_void_p), ('z', c_void_p), ('mean', c_void_p), ('rstd', c_void_p),
                ('dy', c_void_p), ('dx', c_void_p), ('dr', c_void_p), ('dgamma', c_void_p), ('dbeta', c_void_p),
                ('rows', c_int64), ('D', c_int), ('eps', c_float), ('p_drop', c_float), ('seed', c_uint64), ('seed_ptr', c_void_p),
                ('pos', c_void_p), ('pos_rows', c_int64), ('ypos', c_void_p), ('dy2', c_void_p),
                ('r_parts', c_int), ('r_part_stride', c_int64), ('r_bias', c_void_p),
                ('dy_parts', c_void_p), ('dy_nparts', c_int), ('dy_part_stride', c_int64)]


class FfnArgs(Structure):
    """ldetr_ffn_args."""
    _fields_ = [('x', c_void_p), ('ldx', c_int64), ('w1', c_void_p), ('b1', c_void_p), ('w2', c_void_p), ('h', c_void_p), ('ypart', c_void_p),
                ('M', c_int), ('F', c_int), ('p_drop', c_float), ('seed', c_uint64), ('seed_ptr', c_void_p),
                ('dy', c_void_p), ('dxpart', c_void_p), ('dh', c_void_p)]


class MhaSmallArgs(Structure):
    """ldetr_mha_small_args."""
    _fields_ = [('x', c_void_p), ('ldx', c_int64), ('w_in', c_void_p), ('b_in', c_void_p), ('w_out', c_void_p), ('kpm', c_void_p),
                ('qkv', c_void_p), ('o', c_void_p), ('lse', c_void_p), ('ypart', c_void_p), ('B', c_int), ('L', c_int),
                ('scale', c_float), ('p_drop', c_float), ('seed', c_uint64), ('seed_ptr', c_void_p),
                ('dr', c_void_p), ('dqkv', c_void_p), ('dxpart', c_void_p)]


class MhaCrossArgs(Structure):
    """ldetr_mha_cross_args."""
    _fields_ = [('x', c_void_p), ('ldx', c_int64), ('w_q', c_void_p), ('b_q', c_void_p), ('k', c_void_p), ('ldk', c_int64), ('v', c_void_p), ('ldv', c_int64),
                ('w_out', c_void_p), ('kpm', c_void_p), ('q', c_void_p), ('o', c_void_p), ('lse', c_void_p), ('ypart', c_void_p),
                ('B', c_int), ('Lq', c_int), ('Lk', c_int), ('scale', c_float), ('p_drop', c_float), ('seed', c_uint64), ('seed_ptr', c_void_p),
                ('dr', c_void_p), ('dq', c_void_p), ('dk', c_void_p), ('lddk', c_int64), ('dv', c_void_p), ('lddv', c_int64), ('dxpart', c_void_p)]


class WgradDesc(Structure):
    """ldetr_wgrad_desc."""
    _fields_ = [('A', c_void_p), ('lda', c_int64), ('B', c_void_p), ('ldb', c_int64), ('dW', c_void_p), ('ldw', c_int64), ('db', c_void_p),
                ('M', c_int), ('rows', c_int), ('cols', c_int)]


_P = c_void_p
_I = c_int
_L = c_int64
_F = c_float
_T4 = POINTER(Tensor4)
_EP = POINTER(Epilogue)

# name -> argtypes; the list doubles as the "every declared symbol is exported" check in tests.
SIGNATURES = {
    'ldetr_set_workspace': [_P, _L],
    'ldetr_bias_act_f32': [_P, _P, _P, _P, _P, _P, _L, _I, _L, _I, _I, _F, _F, _F, _P],
    'ldetr_upfirdn2d_f32': [_P, _P, _P, _I, _I, _I, _I, POINTER(c_int64), _I, _I, _L, _L, _I, _I, _I, _I, _I, _I, _I, _I,
                            _I, _F, _I, _I, POINTER(c_int64), _P, _I, _F, _F, _P],
    'ldetr_gemm_f32': [_P, _L, _I, _P, _L, _I, _P, _L, _I, _I, _I, _I, _EP, _I, _P],
    'ldetr_conv2d_fwd_f32': [_P, _T4, _P, _I, _I, _I, _I, _I, _P, _L, _I, _I, _P, _L, _EP, _P],
    'ldetr_conv2d_bwd_data_f32': [_P, _T4, _P, _I, _I, _I, _I, _I, _P, _L, _I, _I, _P, _L, _EP, _P],
    'ldetr_conv2d_bwd_weight_f32': [_P, _T4, _P, _T4, _P, _I, _I, _I, _I, _I, _P, _L, _P, _L, _I, _P],
    'ldetr_conv_transpose2d_fwd_f32': [_P, _T4, _P, _I, _I, _I, _I, _I, _P, _L, _I, _I, _P, _L, _EP, _P],
    'ldetr_conv_transpose2d_bwd_data_f32': [_P, _T4, _P, _I, _I, _I, _I, _I, _P, _L, _I, _I, _P, _L, _EP, _P],
    'ldetr_conv_transpose2d_bwd_weight_f32': [_P, _T4, _P, _T4, _P, _I, _I, _I, _I, _I, _P, _L, _P, _L, _I, _P],
    'ldetr_attention_fwd_f32': [_P, _L, _P, _L, _P, _L, _P, _P, _L, _P, _I, _I, _I, _I, _I, _F, _F, c_uint64, _P, _I, _P],
    'ldetr_attention_bwd_f32': [_P, _L, _P, _L, _P, _L, _P, _P, _L, _P, _P, _L, _P, _L, _P, _L, _P, _L,
                                _I, _I, _I, _I, _I, _F, _F, c_uint64, _P, _I, _P],
    'ldetr_layernorm_fwd_f32': [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, c_uint64, _P, _P],
    'ldetr_layernorm_bwd_f32': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, c_uint64, _P, _P],
    'ldetr_layernorm_fwd_pos_f32': [_P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, c_uint64, _P, _P, _L, _P, _P],
    'ldetr_layernorm_fwd_parts_f32': [_P, _P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _F, c_uint64, _P, _P, _L, _P, _P],
    'ldetr_mha_small_fwd_f32': [_P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _F, c_uint64, _P, _P],
    'ldetr_mha_cross_fwd_f32': [_P, _L, _P, _P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, c_uint64, _P, _P],
    'ldetr_ffn_fwd_f32': [_P, _L, _P, _P, _P, _P, _P, _L, _I, _I, _F, c_uint64, _P, _P],
    'ldetr_ffn_bwd_f32': [_P, _P, _L, _P, _P, _P, _P, _P, _L, _I, _I, _F, _P],
    'ldetr_layernorm_bwd_parts_f32': [_P, _P, _P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, c_uint64, _P, _P],
    'ldetr_layernorm_bwd2_f32': [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, c_uint64, _P, _P],
    'ldetr_colsum_f32': [_P, _P, _I, _L, _I, _P],
    'ldetr_act_bwd_reduce_f32': [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _F, _F, _P],
    'ldetr_mul_reduce_f32': [_P, _P, _P, _P, _P, _I, _L, _I, _P],
    'ldetr_torgb_fwd_f32': [_P, _P, _P, _P, _P, _I, _L, _I, _P],
    'ldetr_torgb_bwd_f32': [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _P],
    'ldetr_maxpool3x3s2_fwd_f32': [_P, _P, _P, _I, _I, _I, _I, _P],
    'ldetr_maxpool3x3s2_bwd_f32': [_P, _P, _P, _I, _I, _I, _I, _P],
    'ldetr_grad_sanitize_f32': [_P, _L, _F, _F, _F, _F, _P],
    'ldetr_adam_step_f32': [_P, _P, _P, _P, _L, _L, _F, _F, _F, _F, _I, _F, _F, _F, _F, _P],
    'ldetr_adam_ema_step_f32': [_P, _P, _P, _P, _L, _L, _F, _F, _F, _F, _I, _F, _F, _F, _F, _P, _F, _P],
    'ldetr_ema_lerp_f32': [_P, _P, _L, _F, _P],
    'ldetr_lsap_f64': [_P, _I, _I, _I, _P, _P, _P],
    'ldetr_box_giou_pairwise_f32': [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, c_double, _P],
    'ldetr_bmm_strided_f32': [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    'ldetr_softmax_xent_fwd_f32': [_P, _L, _P, _P, _P, _P, _L, _I, _L, _F, _P],
    'ldetr_softmax_xent_bwd_f32': [_P, _L, _P, _P, _P, _P, _P, _L, _L, _I, _L, _F, _P],
    'ldetr_embedding_fwd_f32': [_P, _P, _P, _P, _L, _I, _I, _I, _P],
    'ldetr_embedding_bwd_f32': [_P, _P, _P, _L, _I, _I, _L, _P],
    'ldetr_debug_trace_tiles': [_P],
    'ldetr_set_split_bf16': [c_int],
    'ldetr_engine_launch_counts': [POINTER(c_int64), POINTER(c_int64)],
    'ldetr_layernorm_fwd_group_f32': [_P, _I, _P],
    'ldetr_layernorm_bwd_group_f32': [_P, _I, _P],
    'ldetr_sum_parts_f32': [_P, _P, _I, _L, _P, _L, _P],
    'ldetr_ffn_fwd_group_f32': [_P, _I, _P],
    'ldetr_ffn_bwd_group_f32': [_P, _I, _P],
    'ldetr_mha_small_fwd_group_f32': [_P, _I, _P],
    'ldetr_mha_small_bwd_group_f32': [_P, _I, _P],
    'ldetr_mha_cross_bwd_f32': [_P, _P],
    'ldetr_wgrad_multi_f32': [_P, _I, _P],
    'ldetr_loss_combine_fwd_f32': [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    'ldetr_loss_combine_bwd_f32': [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P],
    'ldetr_masked_mse_fwd_f32': [_P, _P, _P, _L, _I, _I, _P, _P],
    'ldetr_masked_mse_bwd_f32': [_P, _P, _P, _L, _I, _I, _P, _P, _P, _P],
    'ldetr_torgb_bwd_finish_f32': [_P, _P, _P, _I, _I, _P, _P, _P],
    'ldetr_gemm_pair_f32': [_P, _P, _P],
    'ldetr_gemm_pair_is_single_launch': [_P, _P],
    'ldetr_demod_fwd_f32': [_P, _L, _L, _L, _L, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P],
    'ldetr_demod_bwd_f32': [_P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P],
    'ldetr_layout_losses_f32': [_P, _P, _P, _I, _I, _P, _P, _P],
    'ldetr_layout_losses_bwd_f32': [_P, _P, _I, _I, _P, _P],
    'ldetr_resample_coeffs': [_I, _I, _P, _P, _L, _P],
    'ldetr_p3_split_f32': [_P, _L, _P, _L, _I, _P],
    'ldetr_p3_merge_f32': [_P, _P, _L, _L, _I, _P],
    'ldetr_p3_weight_bwd': [_P, _P, _P, _I, _I, _I, _I, _P],
    'ldetr_p3_conv2d_bwd_data': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    'ldetr_p3_conv2d_fwd': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    'ldetr_p3_conv2d_bwd_weight': [_P, _I, _I, _I, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    'ldetr_p3_weight_prep': [_P, _I, _I, _P],
    'ldetr_p3_conv2d_fwd_dual': [_P, _P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    'ldetr_p3_conv2d_bwd_pair': [_P, _I, _I, _I, _I, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P],
    'ldetr_p3_last_launch': [_P],
    'ldetr_engine_last_launch': [_P],
    'ldetr_struct_sizes': [_P],
    'ldetr_resize_normalize_u8': [_P, _L, _I, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _P, _F, _F, _F, _F, _F, _F, _P],
}

_lib = None


def load():
    """Load the library (once).  Raises if it is missing: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: build the gfx950 kernels first (python -m layoutdetr_amd.build or '
            f'__graft_entry__.build()).  layoutdetr_amd has no CPU/PyTorch fallback path.')
    # torch first: its wheel carries its own HIP runtime; loading this library before it pulls in /opt/rocm's copy as a SECOND runtime
    # in the process (seen as hipGetDevice failing in ldetr_set_workspace when __graft_entry__.build() ran before smoke())
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    lib.ldetr_last_error.restype = c_char_p
    lib.ldetr_last_error.argtypes = []
    lib.ldetr_abi_version.restype = c_int
    lib.ldetr_abi_version.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = c_int
        fn.argtypes = argtypes
    if lib.ldetr_abi_version() != ABI_VERSION:
        raise RuntimeError('libldetr_hip.so ABI version mismatch; rebuild it')
    sizes = (ctypes.c_int32 * 6)()
    lib.ldetr_struct_sizes(sizes)
    mirror = [ctypes.sizeof(t) for t in (LnArgs, FfnArgs, MhaSmallArgs, MhaCrossArgs, WgradDesc, P3Epilogue)]
    if list(sizes) != mirror:
        raise RuntimeError(f'libldetr_hip.so argument blocks {list(sizes)} do not match the ctypes mirror {mirror}; rebuild')
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().ldetr_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'{what}: {msg}' if what else msg)
