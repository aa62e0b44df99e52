"""numpy restatement of the TensorFlow 1.8 ops the reference graph instantiates.

Test infrastructure only (see oracle/__init__.py).  All tensors are NHWC
float32 unless `dtype=np.float64` is requested for a tighter reference.
Each function names the reference call site it stands in for; paths are
relative to the reference checkout.
"""
import numpy as np


# --------------------------------------------------------------------------
# padding helpers
# --------------------------------------------------------------------------
def same_padding(in_size, k, stride, rate=1):
    """TF 'SAME' rule: out = ceil(in/stride); the odd pad goes to the end."""
    k_eff = k + (k - 1) * (rate - 1)
    out = -(-in_size // stride)
    pad_total = max((out - 1) * stride + k_eff - in_size, 0)
    pad_beg = pad_total // 2
    return out, pad_beg, pad_total - pad_beg


def fixed_padding(k, rate=1):
    """Explicit symmetric pad of `conv2d_same` (third_party/resnet_utils.py:115-120)."""
    k_eff = k + (k - 1) * (rate - 1)
    pad_total = k_eff - 1
    pad_beg = pad_total // 2
    return pad_beg, pad_total - pad_beg


def _pad_hw(x, pt, pb, pl, pr, value=0.0):
    if pt == pb == pl == pr == 0:
        return x
    return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)),
                  mode='constant', constant_values=value)


# --------------------------------------------------------------------------
# convolutions  (slim.conv2d: third_party/dilated_resnet_v1.py:90-102,
#                libs/compressible_ops.py:20-24)
# --------------------------------------------------------------------------
def conv2d(x, w, stride=1, rate=1, padding='SAME', dtype=np.float32):
    """Cross-correlation, NHWC x HWIO.  padding: 'SAME' | 'VALID' |
    (pad_top, pad_bottom, pad_left, pad_right) applied before a VALID conv."""
    x = np.asarray(x, dtype=dtype)
    w = np.asarray(w, dtype=dtype)
    kh, kw, cin, cout = w.shape
    n, h, wd, c = x.shape
    assert c == cin, (x.shape, w.shape)
    if padding == 'SAME':
        ho, pt, pb = same_padding(h, kh, stride, rate)
        wo, pl, pr = same_padding(wd, kw, stride, rate)
    elif padding == 'VALID':
        pt = pb = pl = pr = 0
    else:
        pt, pb, pl, pr = padding
    xp = _pad_hw(x, pt, pb, pl, pr)
    hp, wp = xp.shape[1], xp.shape[2]
    keh = kh + (kh - 1) * (rate - 1)
    kew = kw + (kw - 1) * (rate - 1)
    ho = (hp - keh) // stride + 1
    wo = (wp - kew) // stride + 1
    out = np.zeros((n * ho * wo, cout), dtype=dtype)
    for r in range(kh):
        for s in range(kw):
            xs = xp[:, r * rate: r * rate + (ho - 1) * stride + 1: stride,
                    s * rate: s * rate + (wo - 1) * stride + 1: stride, :]
            out += np.ascontiguousarray(xs).reshape(-1, cin) @ w[r, s]
    return out.reshape(n, ho, wo, cout)


def conv2d_same(x, w, stride, rate=1, dtype=np.float32):
    """third_party/resnet_utils.py:77-122."""
    k = w.shape[0]
    if stride == 1:
        return conv2d(x, w, 1, rate, 'SAME', dtype)
    pb, pe = fixed_padding(k, rate)
    return conv2d(x, w, stride, rate, (pb, pe, pb, pe), dtype)


def depthwise_conv2d(x, w, stride=1, rate=1, dtype=np.float32):
    """slim.separable_conv2d(num_outputs=None, depth_multiplier=1), SAME
    (third_party/conv_blocks.py:240-249).  w: [kh, kw, C, 1]."""
    x = np.asarray(x, dtype=dtype)
    w = np.asarray(w, dtype=dtype)
    kh, kw, c, mult = w.shape
    assert mult == 1 and x.shape[3] == c
    n, h, wd, _ = x.shape
    ho, pt, pb = same_padding(h, kh, stride, rate)
    wo, pl, pr = same_padding(wd, kw, stride, rate)
    xp = _pad_hw(x, pt, pb, pl, pr)
    out = np.zeros((n, ho, wo, c), dtype=dtype)
    for r in range(kh):
        for s in range(kw):
            xs = xp[:, r * rate: r * rate + (ho - 1) * stride + 1: stride,
                    s * rate: s * rate + (wo - 1) * stride + 1: stride, :]
            out += xs * w[r, s, :, 0]
    return out


def batch_norm(x, gamma, beta, mean, var, eps=1e-5):
    """slim.batch_norm, is_training=False: (x-mean)*gamma/sqrt(var+eps)+beta."""
    dt = x.dtype
    inv = (1.0 / np.sqrt(np.asarray(var, dt) + dt.type(eps)))
    if gamma is not None:
        inv = inv * np.asarray(gamma, dt)
    out = (x - np.asarray(mean, dt)) * inv
    if beta is not None:
        out = out + np.asarray(beta, dt)
    return out


def relu(x):
    return np.maximum(x, 0)


def relu6(x):
    return np.minimum(np.maximum(x, 0), 6)


# --------------------------------------------------------------------------
# pooling
# --------------------------------------------------------------------------
def max_pool_same(x, k=3, stride=2):
    """slim.max_pool2d(net, [3, 3], stride=2) under padding='SAME'
    (third_party/dilated_resnet_v1.py:175, third_party/resnet_utils.py:274)."""
    n, h, wd, c = x.shape
    ho, pt, pb = same_padding(h, k, stride)
    wo, pl, pr = same_padding(wd, k, stride)
    xp = _pad_hw(x, pt, pb, pl, pr, value=-np.inf)
    out = np.full((n, ho, wo, c), -np.inf, dtype=x.dtype)
    for r in range(k):
        for s in range(k):
            xs = xp[:, r: r + (ho - 1) * stride + 1: stride,
                    s: s + (wo - 1) * stride + 1: stride, :]
            out = np.maximum(out, xs)
    return out


def subsample(x, factor):
    """max_pool2d [1,1] stride=factor (third_party/resnet_utils.py:59-74)."""
    if factor == 1:
        return x
    return x[:, ::factor, ::factor, :]


def avg_pool_valid(x, kh, kw):
    """slim.avg_pool2d(x, (kh, kw), stride=(kh, kw)), VALID: trailing rows and
    columns that do not fill a window are dropped
    (architectures/icnet_architecture.py:195-200)."""
    n, h, wd, c = x.shape
    ho, wo = (h - kh) // kh + 1, (wd - kw) // kw + 1
    xs = x[:, :ho * kh, :wo * kw, :].reshape(n, ho, kh, wo, kw, c)
    return xs.sum(axis=(2, 4), dtype=x.dtype) / x.dtype.type(kh * kw)


# --------------------------------------------------------------------------
# tf.image.resize_bilinear (TF 1.x legacy kernel, no half-pixel centres)
# --------------------------------------------------------------------------
def _interp_weights(out_size, in_size, align_corners):
    if align_corners and out_size > 1:
        scale = np.float32(in_size - 1) / np.float32(out_size - 1)
    else:
        scale = np.float32(in_size) / np.float32(out_size)
    src = np.arange(out_size, dtype=np.float32) * scale      # float32 product
    lo = np.floor(src).astype(np.int64)
    hi = np.minimum(lo + 1, in_size - 1)
    lerp = src - lo.astype(np.float32)
    return lo, hi, lerp.astype(np.float32)


def resize_bilinear(x, out_h, out_w, align_corners=True):
    """architectures/icnet_architecture.py:201-204,264-266;
    third_party/dilated_resnet_v1.py:45-47; libs/evaluator.py:68-71."""
    n, h, wd, c = x.shape
    if (out_h, out_w) == (h, wd):
        return x.copy()
    dt = x.dtype
    ylo, yhi, yl = _interp_weights(out_h, h, align_corners)
    xlo, xhi, xl = _interp_weights(out_w, wd, align_corners)
    xl = xl.astype(dt)[None, None, :, None]
    yl = yl.astype(dt)[None, :, None, None]
    top_rows = x[:, ylo]
    bot_rows = x[:, yhi]
    tl, tr = top_rows[:, :, xlo], top_rows[:, :, xhi]
    bl, br = bot_rows[:, :, xlo], bot_rows[:, :, xhi]
    top = tl + (tr - tl) * xl
    bot = bl + (br - bl) * xl
    return top + (bot - top) * yl


def interp_size(in_size, s_factor=1.0, z_factor=1.0):
    """`_dynamic_interpolation` size rule (architectures/icnet_architecture.py:255-266)."""
    shrink = (in_size - 1) * s_factor + 1
    zoom = shrink + (shrink - 1) * (z_factor - 1)
    return int(zoom)


# --------------------------------------------------------------------------
# glue ops  (libs/exporter.py:14-80)
# --------------------------------------------------------------------------
def pad_to_bounding_box(x, target_h, target_w):
    """tf.image.pad_to_bounding_box(x, 0, 0, H, W): zeros bottom/right."""
    n, h, wd, c = x.shape
    out = np.zeros((n, target_h, target_w, c), dtype=x.dtype)
    out[:, :h, :wd] = x
    return out


def argmax_channels(x):
    """tf.argmax(x, 3): int64, lowest index on ties."""
    return np.argmax(x, axis=3).astype(np.int64)


def map_to_colored_labels(labels, color_map):
    """one_hot(label) @ color_table (libs/exporter.py:14-29): float32 [N,H,W,C]."""
    table = np.asarray([list(c) for c in color_map], dtype=np.float32)
    return table[labels]
