"""Bit-level forms of weights and activations the kernels read: bfloat16 / IEEE-half bits,
split-bfloat16 tensors and rows, two-f16 weight rows, the raw-byte weights of the uint8 stem."""
import numpy as np


def f32_to_bf16_bits(a):
    """float32 ndarray -> uint16 ndarray of bfloat16 bits (round to nearest even)."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return u.astype(np.uint16)


def f32_to_f16_bits(a):
    """float32 ndarray -> uint16 ndarray of IEEE half bits (round to nearest even, saturating)."""
    a = np.clip(np.ascontiguousarray(a, dtype=np.float32), -65504.0, 65504.0)
    return a.astype(np.float16).view(np.uint16)


def f32_split_f16_rows(Wp):
    """Packed float32 weights [cout_pad][k_pad] -> uint16 [cout_pad][2 * k_pad]: every row's
    IEEE-half high parts followed by its remainders (the FSS_F16_W2 convolutions)."""
    hi = f32_to_f16_bits(Wp)
    lo = f32_to_f16_bits(np.asarray(Wp, np.float32) - hi.view(np.float16).astype(np.float32))
    return np.ascontiguousarray(np.concatenate([hi, lo], axis=1))


def stem_pack_f16x2(W, mean, scale, cout_pad, k_pad):
    """Weights of fss_stem_conv_u8 in its FSS_F32_BF16X3 mode from the filter W [kh,kw,3,cout]
    (batch norm folded in): the kernel multiplies the raw frame bytes and a 1 per pixel inside the
    padded image, so scale goes into the three channel weights and -sum_c W_c mean_c scale_c into
    the fourth; returned as uint16 [2 * cout_pad][k_pad]: IEEE-half high parts, then remainders."""
    kh, kw, _, cout = W.shape
    W = np.asarray(W, np.float64)
    ms = np.asarray(mean, np.float64) * np.asarray(scale, np.float64)
    W4 = np.zeros((kh, kw, 4, cout))
    W4[:, :, :3] = W * np.asarray(scale, np.float64)[None, None, :, None]
    W4[:, :, 3] = -(W * ms[None, None, :, None]).sum(axis=2)
    Wp = np.zeros((cout_pad, k_pad))
    Wp[:cout, :kh * kw * 4] = W4.transpose(3, 0, 1, 2).reshape(cout, -1)
    hi = f32_to_f16_bits(Wp.astype(np.float32))
    lo = f32_to_f16_bits((Wp - hi.view(np.float16).astype(np.float64)).astype(np.float32))
    return np.concatenate([hi, lo])


def split_bf16_decode(raw):
    """float32 ndarray holding split-bfloat16 groups along its last axis (a multiple of 8 slots:
    8 bf16 high parts, then 8 bf16 remainders, per 8 elements) -> the float32 values."""
    raw = np.ascontiguousarray(raw, dtype=np.float32)
    u = raw.view(np.uint16).reshape(raw.shape[:-1] + (raw.shape[-1] // 8, 2, 8)).astype(np.uint32) << 16
    f = u.view(np.float32)
    return (f[..., 0, :] + f[..., 1, :]).reshape(raw.shape)


def split_bf16_encode(a):
    """float32 ndarray (last axis a multiple of 8) -> float32-typed array holding its
    split-bfloat16 form (same shape and bytes per element)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    return f32_split_bf16_rows(a.reshape(-1, a.shape[-1])).view(np.float32).reshape(a.shape)


def f32_split_bf16_rows(wp):
    """Packed float32 weights [cout_pad, k_pad] -> the FSS_F32_BF16X3 form: per 8 consecutive k
    values of a row their 8 bfloat16 high parts, then their 8 bfloat16 remainders (uint16
    array of shape [cout_pad, 2 * k_pad], the same bytes as the float32 matrix)."""
    wp = np.ascontiguousarray(wp, dtype=np.float32)
    hi = f32_to_bf16_bits(wp)
    hi_f = (hi.astype(np.uint32) << 16).view(np.float32)
    lo = f32_to_bf16_bits(wp - hi_f)
    r, k = wp.shape
    out = np.empty((r, k // 8, 2, 8), np.uint16)
    out[:, :, 0, :] = hi.reshape(r, k // 8, 8)
    out[:, :, 1, :] = lo.reshape(r, k // 8, 8)
    return out.reshape(r, 2 * k)
