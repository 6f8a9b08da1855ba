"""ctypes binding of libdsrl_hip.so (include/dsrl_hip.h).

The library is the product's only arithmetic path: if it cannot be loaded, or a call returns an error code,
a RuntimeError is raised - there is no CPU or stock-PyTorch fallback behind these wrappers.
"""
import ctypes as C
import os

import torch  # noqa: F401  (imported first so the HIP runtime the allocator uses is the one the library binds to)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libdsrl_hip.so')

fp = C.c_void_p          # device pointers travel as integers
i32 = C.c_int
i64 = C.c_int64
u64 = C.c_uint64
u32 = C.c_uint32
f32 = C.c_float
sz = C.c_size_t
stream_t = C.c_void_p

_conv_shape = [i32] * 10                      # N,H,W,C,K,R,S,stride,pad,dil


class WgradProblem(C.Structure):              # include/dsrl_hip.h: dsrl_wgrad_problem
    _fields_ = [('x', C.c_void_p), ('dy', C.c_void_p), ('dw', C.c_void_p)] + [(n, C.c_int32) for n in
                ('ldx', 'lddy', 'N', 'H', 'W', 'C', 'K', 'R', 'S', 'stride', 'pad', 'dil')] + [('x_amax', C.c_void_p), ('dy_amax', C.c_void_p)]


PROTOTYPES = {
    'dsrl_version': (i32, []),
    'dsrl_last_error': (C.c_char_p, []),
    'dsrl_device_check': (i32, [C.POINTER(i32)]),
    'dsrl_conv2d_fwd_workspace_bytes': (sz, _conv_shape),
    'dsrl_conv2d_fwd': (i32, [fp, i32, fp, fp, fp, i32] + _conv_shape + [fp, sz, stream_t]),
    'dsrl_conv2d_fwd_stats_parts': (i32, _conv_shape),
    'dsrl_conv2d_fwd_stats': (i32, [fp, i32, fp, fp, fp, i32] + _conv_shape + [fp, sz, fp, i32, stream_t]),
    'dsrl_conv2d_dgrad_workspace_bytes': (sz, _conv_shape),
    'dsrl_conv2d_transposed_filter_floats': (sz, [i32] * 4),
    'dsrl_conv2d_transpose_filter': (i32, [fp, fp, i32, i32, i32, i32, stream_t]),
    'dsrl_conv2d_transpose_filters_batched': (i32, [fp, i32, i64, stream_t]),
    'dsrl_conv2d_filters_amax_segment_floats': (i32, []),
    'dsrl_conv2d_filters_amax_batched': (i32, [fp, i64, stream_t]),
    'dsrl_conv2d_dgrad': (i32, [fp, i32, fp, fp, fp, i32] + _conv_shape + [fp, sz, stream_t]),
    'dsrl_conv2d_dgrad_stats_parts': (i32, _conv_shape),
    'dsrl_conv2d_dgrad_bnstats': (i32, [fp, i32, fp, fp, fp, i32] + _conv_shape + [fp, sz, fp, i32, fp, i32, fp, fp, i32, fp, i32, i32, stream_t]),
    'dsrl_conv2d_dgrad_accumulate': (i32, [fp, i32, fp, fp, fp, i32] + _conv_shape + [fp, sz, stream_t]),
    'dsrl_conv2d_wgrad_workspace_bytes': (sz, _conv_shape),
    'dsrl_conv2d_wgrad': (i32, [fp, i32, fp, i32, fp] + _conv_shape + [fp, sz, stream_t]),
    'dsrl_conv2d_wgrad_group_table_bytes': (sz, [i32]),
    'dsrl_conv2d_wgrad_group_workspace_bytes': (sz, [fp, i32]),
    'dsrl_conv2d_wgrad_group_plan': (i32, [fp, i32, fp, sz, fp, fp, sz]),
    'dsrl_conv2d_wgrad_group_launch': (i32, [fp, fp, stream_t]),
    'dsrl_amax': (i32, [fp, i32, i64, i32, fp, stream_t]),
    'dsrl_conv2d_fwd_amax': (i32, [fp, i32, fp, fp, fp, fp, fp, fp, i32] + _conv_shape + [fp, sz, fp, i32, stream_t]),
    'dsrl_conv2d_dgrad_amax': (i32, [fp, i32, fp, fp, fp, fp, fp, fp, i32] + _conv_shape + [fp, sz, fp, i32, fp, i32, fp, fp, i32, fp, i32, i32, stream_t]),
    'dsrl_conv2d_split_filters_batched': (i32, [fp, i32, i64, stream_t]),
    'dsrl_planes_lo_offset': (sz, [i64]),
    'dsrl_planes_bytes': (sz, [i64, i32]),
    'dsrl_split_planes': (i32, [fp, i32, i64, i32, fp, fp, i32, stream_t]),
    'dsrl_conv2d_filter_planes_batched': (i32, [fp, i32, i64, stream_t]),
    'dsrl_conv2d_fwd_planes': (i32, [fp, i32, fp, fp, fp, fp, fp, fp, fp, fp, i32] + _conv_shape + [fp, sz, fp, i32, stream_t]),
    'dsrl_conv2d_dgrad_planes': (i32, [fp, i32, fp, fp, fp, fp, fp, fp, fp, fp, i32] + _conv_shape + [fp, sz, fp, i32, fp, i32, fp, fp, i32, fp, i32, i32, stream_t]),
    'dsrl_conv2d_dgrad_planes_drop': (i32, [fp, i32, fp, fp, fp, fp, fp, fp, fp, fp, i32] + _conv_shape + [fp, sz, fp, i32, fp, i32, fp, fp, i32, f32, fp, i32, i32, stream_t]),
    'dsrl_conv2d_wgrad_amax': (i32, [fp, i32, fp, fp, i32, fp, fp] + _conv_shape + [fp, sz, stream_t]),
    'dsrl_conv_precision': (i32, [i32]),
    'dsrl_conv2d_inbounds_macs': (i64, _conv_shape),
    'dsrl_conv2d_rowfold_fwd_workspace_bytes': (sz, [i32] * 9),
    'dsrl_conv2d_rowfold_fwd': (i32, [fp, i32, fp, fp, fp, i32] + [i32] * 9 + [i64, fp, sz, stream_t]),
    'dsrl_conv2d_rowfold_wgrad_workspace_bytes': (sz, [i32] * 9),
    'dsrl_conv2d_rowfold_wgrad': (i32, [fp, i32, fp, i32, fp] + [i32] * 9 + [i64, fp, sz, stream_t]),
    'dsrl_pad_image_nhwc': (i32, [fp, i64, i64, i64, i64, fp] + [i32] * 9 + [stream_t]),
    'dsrl_colsum_workspace_bytes': (sz, [i64, i32]),
    'dsrl_colsum': (i32, [fp, i32, i64, i32, fp, fp, sz, stream_t]),
    'dsrl_bn_workspace_bytes': (sz, [i64, i32]),
    'dsrl_bn_stats_floats': (sz, [i32, i32, i32]),
    'dsrl_bn_stats': (i32, [fp, i32, i64, i32, f32, f32, fp, fp, fp, fp, fp, sz, stream_t]),
    'dsrl_bn_invstd_from_var': (i32, [fp, i32, f32, fp, stream_t]),
    'dsrl_bn_apply': (i32, [fp, i32, fp, i32, i64, i32, fp, fp, fp, fp, fp, i32, i32, f32, u64, u32, fp, stream_t]),
    'dsrl_bn_fused_max_blocks': (i32, [i32]),
    'dsrl_bn_fused_barrier_timeouts': (i32, [C.POINTER(i64)]),
    'dsrl_bn_train_fwd_from_stats': (i32, [fp, i32, fp, i32, i64, i32, f32, f32, fp, fp, fp, fp, fp, fp, fp, i32, i32, f32, u64, u32, fp, i32, fp, stream_t]),
    'dsrl_bn_train_fwd': (i32, [fp, i32, fp, i32, i64, i32, f32, f32, fp, fp, fp, fp, fp, fp, fp, i32, i32, f32, u64, u32, fp, sz, fp, stream_t]),
    'dsrl_bn_bwd': (i32, [fp, i32, fp, i32, fp, i32, fp, i32, fp, i32, i64, i32, fp, fp, fp, fp, fp, i32, f32, i32, fp, sz, fp, stream_t]),
    'dsrl_bn_bwd_from_stats': (i32, [fp, i32, fp, i32, fp, i32, fp, i32, fp, i32, i64, i32, fp, fp, fp, fp, fp, i32, i32, fp, i32, fp, stream_t]),
    'dsrl_bn_bwd_from_stats_drop': (i32, [fp, i32, fp, i32, fp, i32, fp, i32, fp, i32, i64, i32, fp, fp, fp, fp, fp, i32, f32, i32, fp, i32, fp, stream_t]),
    'dsrl_bn_bwd_from_stats_res_parts': (i32, [i64, i32, i32]),
    'dsrl_bn_bwd_from_stats_res': (i32, [fp, i32, fp, i32, fp, i32, fp, i32, fp, i32, i64, i32, fp, fp, fp, fp, fp, i32, f32, i32, fp, i32, fp, fp, i32, fp, fp, fp, i32, stream_t]),
    'dsrl_dropout_fwd': (i32, [fp, i32, fp, i32, i64, i32, f32, u64, u32, stream_t]),
    'dsrl_dropout_bwd': (i32, [fp, i32, fp, i32, i64, i32, f32, u64, u32, stream_t]),
    'dsrl_bilinear_ac_fwd': (i32, [fp, i32, fp, i32, i32, i32, i32, i32, i32, i32, stream_t]),
    'dsrl_bilinear_ac_bwd_workspace_bytes': (sz, [i32] * 6),
    'dsrl_bilinear_ac_bwd': (i32, [fp, i32, fp, i32, i32, i32, i32, i32, i32, i32, fp, sz, stream_t]),
    'dsrl_global_avgpool_fwd': (i32, [fp, i32, fp, i32, i32, i32, stream_t]),
    'dsrl_global_avgpool_bwd': (i32, [fp, fp, i32, i32, i32, i32, stream_t]),
    'dsrl_maxpool3x3s2_fwd': (i32, [fp, fp, fp, i32, i32, i32, i32, stream_t]),
    'dsrl_maxpool3x3s2_bwd': (i32, [fp, fp, fp, i32, i32, i32, i32, stream_t]),
    'dsrl_convt2x2_fwd': (i32, [fp, fp, fp, fp, i32, i32, i32, i32, i32, stream_t]),
    'dsrl_convt2x2_bwd_workspace_bytes': (sz, [i32] * 5),
    'dsrl_convt2x2_bwd': (i32, [fp, fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, fp, sz, stream_t]),
    'dsrl_convt2x2_fwd_ce_supported': (i32, [fp, fp, i32, i32, i32, i32, i32]),
    'dsrl_convt2x2_fwd_ce_workspace_bytes': (sz, [i32, i32, i32]),
    'dsrl_convt2x2_fwd_ce': (i32, [fp, fp, fp, fp, i32, i32, i32, i32, i32, fp, i32, fp, fp, fp, sz, stream_t]),
    'dsrl_convt2x2_bwd_ce_supported': (i32, [fp, fp, fp, i32, i32, i32, i32, i32]),
    'dsrl_convt2x2_bwd_ce': (i32, [fp, fp, fp, fp, i32, fp, fp, fp, i32, fp, fp, fp, i32, i32, i32, i32, i32, fp, sz, stream_t]),
    'dsrl_pixel_shuffle_fwd': (i32, [fp, fp, i32, i32, i32, i32, i32, stream_t]),
    'dsrl_pixel_shuffle_bwd': (i32, [fp, fp, i32, i32, i32, i32, i32, stream_t]),
    'dsrl_pointwise_strided_fwd': (i32, [fp, fp, fp, i32, i32, i32, i32, i32, stream_t]),
    'dsrl_pointwise_strided_bwd_workspace_bytes': (sz, [i32] * 5),
    'dsrl_pointwise_strided_bwd': (i32, [fp, fp, fp, fp, fp, i32, i32, i32, i32, i32, i32, fp, sz, stream_t]),
    'dsrl_copy2d': (i32, [fp, i32, fp, i32, i64, i32, stream_t]),
    'dsrl_cat_channels_supported': (i32, [fp, fp, fp, i32, fp, i32, i64]),          # srcs / lds / cs: host arrays (ctypes)
    'dsrl_cat_channels': (i32, [fp, fp, fp, i32, fp, i32, i64, fp, stream_t]),
    'dsrl_nchw_to_nhwc': (i32, [fp, fp, i32, i32, i32, i32, i32, stream_t]),
    'dsrl_ce_workspace_bytes': (sz, [i64]),
    'dsrl_ce_fwd': (i32, [fp, i32, fp, i64, i32, i32, fp, fp, sz, stream_t]),
    'dsrl_ce_bwd': (i32, [fp, i32, fp, i64, i32, i32, fp, fp, fp, i32, stream_t]),
    'dsrl_mse_workspace_bytes': (sz, [i64]),
    'dsrl_mse_fwd': (i32, [fp, fp, i64, fp, fp, sz, stream_t]),
    'dsrl_mse_bwd': (i32, [fp, fp, i64, fp, fp, stream_t]),
    'dsrl_ce_fused_workspace_bytes': (sz, [i64]),
    'dsrl_ce_fused': (i32, [fp, i32, fp, i64, i32, i32, fp, i32, fp, fp, fp, sz, stream_t]),
    'dsrl_mse_fused': (i32, [fp, fp, i64, f32, fp, fp, fp, fp, sz, stream_t]),
    'dsrl_loss_mix': (i32, [fp, fp, fp, f32, f32, fp, fp, stream_t]),
    'dsrl_fa_saved_floats': (sz, [i32] * 5),
    'dsrl_fa_workspace_bytes': (sz, [i32] * 5),
    'dsrl_fa_fwd': (i32, [fp, fp, i32, i32, i32, i32, i64, i64, i64, i64, i32, i32, fp, fp, fp, sz, stream_t]),
    'dsrl_fa_bwd': (i32, [fp, fp, i32, i32, i32, i32, i64, i64, i64, i64, i32, i32, fp, fp, fp, fp, fp, sz, stream_t]),
    'dsrl_seg_metrics': (i32, [fp, i32, fp, i64, i32, i32, fp, stream_t]),
    'dsrl_prepare_batch': (i32, [fp, fp, fp, C.POINTER(f32), C.POINTER(f32), fp, fp, fp, i32, i32, i32, i32, i32, stream_t]),
    'dsrl_sgd_step': (i32, [fp, fp, fp, i64, f32, f32, f32, f32, stream_t]),
    'dsrl_sgd_step_dev': (i32, [fp, fp, fp, i64, fp, stream_t]),
    'dsrl_sgd_step_dev_segments': (i32, [fp, fp, fp, fp, i64, fp, stream_t]),
    'dsrl_nan_check': (i32, [fp, i64, fp, stream_t]),
    'dsrl_rng_bind_device_key': (i32, [fp]),
    'dsrl_rng_advance_key': (i32, [fp, stream_t]),
    'dsrl_prof_enable': (i32, [i32]),
    'dsrl_prof_read': (i32, [i32, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'dsrl_prof_read_bytes': (i32, [i32, C.POINTER(C.c_double)]),
    'dsrl_prof_kernel_name': (C.c_char_p, [i32]),
}

_lib = None


class DsrlHipError(RuntimeError):
    pass


def load():
    """Loads libdsrl_hip.so (once). Raises DsrlHipError with the build hint when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise DsrlHipError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                           f'or `make -C {os.path.join(_HERE, "csrc")}` - there is no fallback path')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here means header and library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.dsrl_version() != 1:
        raise DsrlHipError(f'libdsrl_hip.so ABI version {lib.dsrl_version()} != 1')
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = _lib.dsrl_last_error().decode('utf-8', 'replace') if _lib is not None else ''
        raise DsrlHipError(f'{what} failed with code {code}: {msg}')


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)


def query(name, *args):
    return getattr(load(), name)(*args)
