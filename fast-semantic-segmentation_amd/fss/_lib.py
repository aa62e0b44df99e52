"""ctypes binding of libfss_gfx950.so (declared in include/fss.h).

The library is mandatory: importing this module raises if the shared object is
missing or does not export a declared symbol -- there is no CPU fallback.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# FSS_LIB: another build of the same library (kernel experiments)
LIB_PATH = os.environ.get('FSS_LIB') or os.path.join(os.path.dirname(_HERE), 'libfss_gfx950.so')

F32, BF16 = 0, 1
F32_BF16X3 = 2      # split-bfloat16 tensors (two bf16 numbers per value), three MFMAs per product
F16 = 3             # IEEE half precision tensors (convolution entry points only)
F16_W2 = 4          # fss_conv_desc.in_dtype only: F16 activations, weights as two f16 numbers each
PYRAMID_STATIC_MAX_BYTES = 32 << 20     # FSS_PYRAMID_STATIC_MAX_BYTES
ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
ACT_CODES = {None: ACT_NONE, 'relu': ACT_RELU, 'relu6': ACT_RELU6}

c_int, c_size_t, c_void_p = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p
c_float_p = ctypes.POINTER(ctypes.c_float)
c_int_p = ctypes.POINTER(ctypes.c_int)


class ConvDesc(ctypes.Structure):
    """Mirror of `fss_conv_desc`."""
    _fields_ = [(n, ctypes.c_int32) for n in (
        'n', 'h', 'w', 'cin', 'cout', 'kh', 'kw', 'stride', 'dilation',
        'pad_top', 'pad_left', 'ho', 'wo', 'act', 'in_dtype', 'out_dtype',
        'y_ld', 'res_ld', 'k_pad', 'cout_pad', 'config')]


# name -> (restype, argtypes); must list every symbol of include/fss.h
PROTOTYPES = {
    'fss_last_error': (ctypes.c_char_p, []),
    'fss_version': (c_int, []),
    'fss_device_count': (c_int, [c_int_p]),
    'fss_set_device': (c_int, [c_int]),
    'fss_device_name': (c_int, [ctypes.c_char_p, c_int]),
    'fss_malloc': (c_int, [ctypes.POINTER(c_void_p), c_size_t]),
    'fss_free': (c_int, [c_void_p]),
    'fss_memcpy_h2d': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'fss_memcpy_d2h': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    'fss_memset': (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    'fss_stream_create': (c_int, [ctypes.POINTER(c_void_p)]),
    'fss_stream_destroy': (c_int, [c_void_p]),
    'fss_stream_sync': (c_int, [c_void_p]),
    'fss_device_sync': (c_int, []),
    'fss_event_create': (c_int, [ctypes.POINTER(c_void_p)]),
    'fss_event_destroy': (c_int, [c_void_p]),
    'fss_event_record': (c_int, [c_void_p, c_void_p]),
    'fss_stream_wait_event': (c_int, [c_void_p, c_void_p]),
    'fss_event_sync': (c_int, [c_void_p]),
    'fss_event_elapsed_ms': (c_int, [c_void_p, c_void_p, c_float_p]),
    'fss_capture_begin': (c_int, [c_void_p]),
    'fss_capture_end': (c_int, [c_void_p, ctypes.POINTER(c_void_p)]),
    'fss_graph_launch': (c_int, [c_void_p, c_void_p]),
    'fss_graph_destroy': (c_int, [c_void_p]),
    'fss_conv_pack_dims': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int_p, c_int_p]),
    'fss_conv_config_count': (c_int, []),
    'fss_conv_config_info': (c_int, [c_int, c_int_p]),
    'fss_conv2d_nhwc': (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p]),
    'fss_conv2d_splitk_nhwc': (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    'fss_conv2d_dual_nhwc': (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    'fss_conv2d_dual_head_nhwc': (c_int, [ctypes.POINTER(ConvDesc), c_void_p, c_void_p, c_int, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                          c_int, c_void_p, c_int, c_void_p]),
    'fss_conv1x1_chain_nhwc': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                       c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                       c_int, c_void_p, c_int, c_int, ctypes.c_longlong, c_int, c_void_p]),
    'fss_conv3x3_c64_nhwc': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 13 + [c_void_p]),
    'fss_conv3x3_c128_pack_index': (c_int, [c_int_p]),
    'fss_conv3x3_c128_nhwc': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    'fss_zoom2_border_gather': (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    'fss_zoom2_conv_commute': (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    'fss_stem_pack_dims': (c_int, [c_int, c_int, c_int, c_int, c_int_p, c_int_p]),
    'fss_stem_conv_u8': (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_float_p, c_float_p] + [c_int] * 12 + [c_void_p]),
    'fss_stem2_conv_u8': (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_float_p, c_float_p] + [c_int] * 14 + [c_void_p]),
    'fss_dwconv2d_nhwc': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 14 + [c_void_p]),
    'fss_maxpool_nhwc': (c_int, [c_void_p, c_void_p] + [c_int] * 11 + [c_void_p]),
    'fss_subsample_nhwc': (c_int, [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p]),
    'fss_avgpool_bins_nhwc': (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    'fss_resize_bilinear_nhwc': (c_int, [c_void_p, c_void_p] + [c_int] * 9 + [c_void_p]),
    'fss_avgpool_pyramid_nhwc': (c_int, [c_void_p, ctypes.POINTER(c_void_p), c_int_p, c_int_p, c_int_p, c_int_p,
                                         c_int, c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'fss_avgpool_pyramid_nhwc_static': (c_int, [c_void_p, ctypes.POINTER(c_void_p), c_int_p, c_int_p, c_int_p,
                                                c_int_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'fss_pyramid_add_nhwc': (c_int, [c_void_p, ctypes.POINTER(c_void_p), c_int_p, c_int_p,
                                     c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'fss_add_act': (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    'fss_to_half': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    'fss_copy_channels': (c_int, [c_void_p, c_int, c_void_p, c_int, c_size_t, c_int, c_int, c_void_p]),
    'fss_preprocess_u8': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_float_p, c_float_p, c_int, c_void_p]),
    'fss_upsample_argmax': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'fss_label_lut': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'fss_confusion_accumulate': (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p, c_void_p]),
}


class FssError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libfss_gfx950.so not found at %s; build it with '
            '`make -C fast-semantic-segmentation_amd/csrc` (hipcc, gfx950). '
            'There is no fallback path.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


lib = _load()


def check(status, what=''):
    if status != 0:
        msg = lib.fss_last_error()
        raise FssError('%s failed (%d): %s' % (what or 'libfss call', status,
                                              msg.decode() if msg else ''))


def call(name, *args):
    check(getattr(lib, name)(*args), name)


CONV3X3_C128_IMAGE_ELEMS = 294912
_c128_index = None


def conv3x3_c128_index():
    """[294912, 2] int32: (row, k) of fss_conv2d_nhwc's packed [256][1152] matrix behind every element
    of the filter image fss_conv3x3_c128_nhwc reads."""
    global _c128_index
    if _c128_index is None:
        import numpy as np
        idx = np.empty((CONV3X3_C128_IMAGE_ELEMS, 2), np.int32)
        call('fss_conv3x3_c128_pack_index', idx.ctypes.data_as(c_int_p))
        _c128_index = idx
    return _c128_index


def conv_pack_dims(cin, kh, kw, cout, dtype):
    k_pad, cout_pad = c_int(0), c_int(0)
    call('fss_conv_pack_dims', cin, kh, kw, cout, dtype,
         ctypes.byref(k_pad), ctypes.byref(cout_pad))
    return k_pad.value, cout_pad.value
