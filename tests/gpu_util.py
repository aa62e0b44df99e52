"""Helpers for the GPU parity tests: thin numpy -> C-ABI -> numpy wrappers.
Every call goes through libfss_gfx950.so via ctypes (fss._lib)."""
import ctypes

import numpy as np
import torch

from fss import _lib
from fss.plan import (f32_to_bf16_bits, f32_to_f16_bits, f32_split_bf16_rows, f32_split_f16_rows,
                      split_bf16_decode, split_bf16_encode)

# 'f32x3': split-bfloat16 tensors (FSS_F32_BF16X3: 4-byte slots, per 8 channels 8 bf16 high parts
# then 8 bf16 remainders); convolutions form their products from them as three bf16 MFMAs
DT = {'f32': (_lib.F32, torch.float32, 4), 'bf16': (_lib.BF16, torch.bfloat16, 8),
      'f32x3': (_lib.F32_BF16X3, torch.float32, 8), 'f16': (_lib.F16, torch.float16, 8),
      # convolution input only: f16 activations against weights kept as two f16 numbers each
      'f16w': (_lib.F16_W2, torch.float16, 8)}


def x3_round(a):
    """float32 -> the nearest value a split-bfloat16 tensor holds (about 16 significant bits)."""
    a = np.asarray(a, np.float32)
    return split_bf16_decode(split_bf16_encode(a.reshape(-1, 8))).reshape(a.shape)


def half_bits(a, dt):
    """float32 array -> the uint16 bit patterns of its bfloat16 / half-precision rounding."""
    return f32_to_f16_bits(a) if dt == 'f16' else f32_to_bf16_bits(a)


def f16_round(a):
    """float32 -> nearest half-precision value, as float32."""
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def conv_in_code(dt):
    return DT[dt][0]


def bf16_round(a):
    """float32 -> nearest bfloat16 value, as float32."""
    bits = f32_to_bf16_bits(np.asarray(a, np.float32)).astype(np.uint32) << 16
    return bits.view(np.float32).reshape(np.shape(a))


def to_dev(a, dt):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if dt == 'f32x3':
        a = split_bf16_encode(a)
    t = torch.from_numpy(a).cuda()
    return t.to(DT[dt][1]).contiguous()


def to_host(t, dt=None, c=None, fill=-77.0):
    """Device tensor -> float32 host array.  dt 'f32x3': decoded from its split-bfloat16 form.
    c: logical channel count of a tensor with a wider pixel stride; the channels past it must
    still hold `fill` (checked on the raw bytes), and only the first c are returned."""
    torch.cuda.synchronize()
    raw = t.float().cpu().numpy()
    if c is not None and c != raw.shape[-1]:
        assert np.all(raw[..., c:] == fill), 'kernel wrote outside its channels'
        raw = np.ascontiguousarray(raw[..., :c])
    return split_bf16_decode(raw) if dt == 'f32x3' else raw


def pack_conv_weights(w, dt, cin_phys=None):
    kh, kw, cin, cout = w.shape
    pc = cin_phys or cin
    if pc != cin:
        w = np.pad(w, ((0, 0), (0, 0), (0, pc - cin), (0, 0)))
    k_pad, cout_pad = _lib.conv_pack_dims(pc, kh, kw, cout, DT[dt][0])
    wp = np.zeros((cout_pad, k_pad), np.float32)
    wp[:cout, :kh * kw * pc] = w.transpose(3, 0, 1, 2).reshape(cout, -1)
    if dt in ('bf16', 'f16'):
        dev = torch.from_numpy(half_bits(wp, dt).view(np.int16)).cuda()
    elif dt == 'f32x3':
        dev = torch.from_numpy(f32_split_bf16_rows(wp).view(np.int16)).cuda()
    elif dt == 'f16w':
        dev = torch.from_numpy(f32_split_f16_rows(wp).view(np.int16)).cuda()
    else:
        dev = torch.from_numpy(wp).cuda()
    return dev, k_pad, cout_pad


def conv2d(x, w, bias=None, residual=None, stride=1, rate=1, pads=(0, 0), out_hw=None,
           act=None, dt='f32', out_dt=None, y_ld=None, config=0, splits=1):
    """x: [n,h,w,cin] float32 host array; returns float32 host array [n,ho,wo,cout].
    splits > 1: through fss_conv2d_splitk_nhwc."""
    out_dt = out_dt or dt
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    ho, wo = out_hw
    xd = to_dev(x, dt)
    wdev, k_pad, cout_pad = pack_conv_weights(w, dt)
    bd = torch.from_numpy(np.asarray(bias, np.float32)).cuda() if bias is not None else None
    # the residual has the input's type; for half-precision input, the output's
    rd = to_dev(residual, out_dt if dt in ('f16', 'f16w') else dt) if residual is not None else None
    ld = y_ld or cout
    y = torch.full((n, ho, wo, ld), -77.0, dtype=DT[out_dt][1], device='cuda')
    d = _lib.ConvDesc(n=n, h=h, w=wd, cin=cin, cout=cout, kh=kh, kw=kw, stride=stride,
                      dilation=rate, pad_top=pads[0], pad_left=pads[1], ho=ho, wo=wo,
                      act=_lib.ACT_CODES[act], in_dtype=conv_in_code(dt), out_dtype=DT[out_dt][0],
                      y_ld=ld, res_ld=cout, k_pad=k_pad, cout_pad=cout_pad, config=config)
    torch.cuda.synchronize()
    if splits > 1:
        ws = torch.full((splits * n * ho * wo * ((cout + 3) // 4 * 4) + 64,), float('nan'), device='cuda')
        _lib.call('fss_conv2d_splitk_nhwc', ctypes.byref(d), xd.data_ptr(), wdev.data_ptr(),
                  bd.data_ptr() if bd is not None else None,
                  rd.data_ptr() if rd is not None else None, y.data_ptr(), splits,
                  ws.data_ptr(), (ws.numel() - 64) * 4, None)
        _lib.call('fss_device_sync')
        assert bool(torch.isnan(ws[-64:]).all()), 'split-K launch wrote past its slabs'
    else:
        _lib.call('fss_conv2d_nhwc', ctypes.byref(d), xd.data_ptr(), wdev.data_ptr(),
                  bd.data_ptr() if bd is not None else None,
                  rd.data_ptr() if rd is not None else None, y.data_ptr(), None)
    _lib.call('fss_device_sync')
    return to_host(y, out_dt, cout)


def conv2d_dual(x, w, x2, w2, bias, stride=1, rate=1, pads=(0, 0), out_hw=None, act=None, dt='bf16',
                config=0, out_dt=None):
    """act(conv(x, w) + conv1x1(x2, w2) + bias) through fss_conv2d_dual_nhwc; host float32
    arrays in, float32 [n,ho,wo,cout] out."""
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    cin2 = x2.shape[3]
    ho, wo = out_hw
    k = kh * kw * cin + cin2
    k_pad, cout_pad = (k + 63) // 64 * 64, (cout + 255) // 256 * 256
    wp = np.zeros((cout_pad, k_pad), np.float32)
    wp[:cout, :kh * kw * cin] = w.transpose(3, 0, 1, 2).reshape(cout, -1)
    wp[:cout, kh * kw * cin:k] = w2.reshape(cin2, cout).T
    if dt in ('bf16', 'f16'):
        wdev = torch.from_numpy(half_bits(wp, dt).view(np.int16)).cuda()
    elif dt == 'f32x3':
        wdev = torch.from_numpy(f32_split_bf16_rows(wp).view(np.int16)).cuda()
    elif dt == 'f16w':
        wdev = torch.from_numpy(f32_split_f16_rows(wp).view(np.int16)).cuda()
    else:
        wdev = torch.from_numpy(wp).cuda()
    xd, x2d = to_dev(x, dt), to_dev(x2, dt)
    bd = torch.from_numpy(np.asarray(bias, np.float32)).cuda()
    out_dt = out_dt or dt
    y = torch.full((n, ho, wo, cout), -77.0, dtype=DT[out_dt][1], device='cuda')
    d = _lib.ConvDesc(n=n, h=h, w=wd, cin=cin, cout=cout, kh=kh, kw=kw, stride=stride,
                      dilation=rate, pad_top=pads[0], pad_left=pads[1], ho=ho, wo=wo,
                      act=_lib.ACT_CODES[act], in_dtype=conv_in_code(dt), out_dtype=DT[out_dt][0],
                      y_ld=cout, res_ld=0, k_pad=k_pad, cout_pad=cout_pad, config=config)
    torch.cuda.synchronize()
    _lib.call('fss_conv2d_dual_nhwc', ctypes.byref(d), xd.data_ptr(), x2d.data_ptr(), cin2,
              wdev.data_ptr(), bd.data_ptr(), None, y.data_ptr(), None)
    _lib.call('fss_device_sync')
    return to_host(y, out_dt)


def conv2d_dual_head(x, w, x2, w2, bias, w_head, b_head, rate=1, pads=(0, 0), out_hw=None, act='relu',
                     config=0, y_ld=None, residual=None, zoom_residual=None, dt='bf16'):
    """bf16(act(conv(x, w) + conv1x1(x2, w2) + bias)) times the [cout, classes] matrix w_head
    + b_head through fss_conv2d_dual_head_nhwc (dt 'bf16': bfloat16 activations; 'f32x3': float
    tensors, split-bf16 products, nothing rounded to bfloat16); x2 / w2 may be None.
    Returns float32 [n,ho,wo,classes]."""
    assert dt in ('bf16', 'f32x3', 'f16w')
    rdt = 'f32x3' if dt == 'f16w' else dt          # type of the residual
    n, h, wd, cin = x.shape
    kh, kw, _, cout = w.shape
    cin2 = x2.shape[3] if x2 is not None else 0
    ncls = w_head.shape[1]
    ho, wo = out_hw
    k = kh * kw * cin + cin2
    k_pad, cout_pad = (k + 63) // 64 * 64, (cout + 255) // 256 * 256
    wp = np.zeros((cout_pad, k_pad), np.float32)
    wp[:cout, :kh * kw * cin] = w.transpose(3, 0, 1, 2).reshape(cout, -1)
    if cin2:
        wp[:cout, kh * kw * cin:k] = w2.reshape(cin2, cout).T
    wh = np.zeros((32, cout), np.float32)
    wh[:ncls] = w_head.T
    if dt == 'bf16':
        wdev = torch.from_numpy(f32_to_bf16_bits(wp).view(np.int16)).cuda()
        wh_bits = f32_to_bf16_bits(wh)
    else:
        rows = f32_split_f16_rows if dt == 'f16w' else f32_split_bf16_rows
        wdev = torch.from_numpy(rows(wp).view(np.int16)).cuda()
        hi = f32_to_bf16_bits(wh)
        wh_bits = np.concatenate([hi, f32_to_bf16_bits(wh - (hi.astype(np.uint32) << 16).view(np.float32))])
    whdev = torch.from_numpy(np.ascontiguousarray(wh_bits).view(np.int16)).cuda()
    xd = to_dev(x, dt)
    x2d = to_dev(x2, dt) if cin2 else None
    bd = torch.from_numpy(np.asarray(bias, np.float32)).cuda()
    bh = torch.from_numpy(np.asarray(b_head, np.float32)).cuda()
    rd = to_dev(residual, rdt) if residual is not None else None
    # the interpolated sources of the split mode are plain float tensors
    zd = [to_dev(t, 'f32' if dt != 'bf16' else dt) for t in zoom_residual] if zoom_residual is not None else None
    zp = (ctypes.c_void_p * 4)(*[t.data_ptr() for t in zd]) if zd else None
    zh, zw = (zd[0].shape[1], zd[0].shape[2]) if zd else (0, 0)
    ld = y_ld or ncls
    y = torch.full((n, ho, wo, ld), -77.0, dtype=torch.float32, device='cuda')
    d = _lib.ConvDesc(n=n, h=h, w=wd, cin=cin, cout=cout, kh=kh, kw=kw, stride=1,
                      dilation=rate, pad_top=pads[0], pad_left=pads[1], ho=ho, wo=wo,
                      act=_lib.ACT_CODES[act], in_dtype=conv_in_code(dt), out_dtype=DT[rdt][0],
                      y_ld=cout, res_ld=cout, k_pad=k_pad, cout_pad=cout_pad, config=config)
    torch.cuda.synchronize()
    _lib.call('fss_conv2d_dual_head_nhwc', ctypes.byref(d), xd.data_ptr(),
              x2d.data_ptr() if cin2 else None, cin2, wdev.data_ptr(), bd.data_ptr(),
              rd.data_ptr() if rd is not None else None, zp, zh, zw, whdev.data_ptr(), bh.data_ptr(),
              ncls, y.data_ptr(), ld, None)
    _lib.call('fss_device_sync')
    out = to_host(y)
    if ld != ncls:
        assert np.all(out[..., ncls:] == -77.0), 'kernel wrote outside its channel slice'
    return out[..., :ncls]


def conv1x1_chain(x, x2, w, bias, residual, act1, w2, bias2, act2, y_ld=None, y2_ld=None):
    """act1(conv1x1([x | x2], w) + bias + residual) and act2(conv1x1(of that, as stored, w2) + bias2)
    through fss_conv1x1_chain_nhwc (bfloat16).  x, x2, residual: [n,h,w,c]; w: [cin (+cin2), cout1];
    w2: [cout1, cout2].  Returns the two outputs as float32."""
    n, h, wd, cin = x.shape
    cin2 = x2.shape[3] if x2 is not None else 0
    cout1, cout2 = w.shape[1], w2.shape[1]
    wdev = torch.from_numpy(f32_to_bf16_bits(np.ascontiguousarray(w.T)).view(np.int16)).cuda()
    w2dev = torch.from_numpy(f32_to_bf16_bits(np.ascontiguousarray(w2.T)).view(np.int16)).cuda()
    xd = to_dev(x, 'bf16')
    x2d = to_dev(x2, 'bf16') if cin2 else None
    rd = to_dev(residual, 'bf16') if residual is not None else None
    bd = torch.from_numpy(np.asarray(bias, np.float32)).cuda() if bias is not None else None
    b2d = torch.from_numpy(np.asarray(bias2, np.float32)).cuda() if bias2 is not None else None
    ld1, ld2 = y_ld or cout1, y2_ld or cout2
    y = torch.full((n, h, wd, ld1), -77.0, dtype=torch.bfloat16, device='cuda')
    y2 = torch.full((n, h, wd, ld2), -77.0, dtype=torch.bfloat16, device='cuda')
    torch.cuda.synchronize()
    _lib.call('fss_conv1x1_chain_nhwc', xd.data_ptr(), cin, cin, x2d.data_ptr() if cin2 else None, cin2, cin2,
              wdev.data_ptr(), bd.data_ptr() if bd is not None else None,
              rd.data_ptr() if rd is not None else None, cout1 if rd is not None else 0,
              _lib.ACT_CODES[act1], y.data_ptr(), ld1, cout1, w2dev.data_ptr(),
              b2d.data_ptr() if b2d is not None else None, _lib.ACT_CODES[act2], y2.data_ptr(), ld2, cout2,
              n * h * wd, _lib.BF16, None)
    _lib.call('fss_device_sync')
    o1, o2 = to_host(y), to_host(y2)
    assert np.all(o1[..., cout1:] == -77.0) and np.all(o2[..., cout2:] == -77.0), 'wrote outside its channels'
    return o1[..., :cout1], o2[..., :cout2]


def conv3x3_c64(x, w, bias, stride, rate, pads, out_hw, act, y_ld=None, dt='bf16'):
    """act(conv3x3(x, w) + bias) through fss_conv3x3_c64_nhwc (bfloat16, 64 input channels, filter
    resident in LDS).  x: [n,h,w,64]; w: [3,3,64,cout].  Returns float32 [n,ho,wo,cout]."""
    n, h, wd, cin = x.shape
    cout = w.shape[3]
    ho, wo = out_hw
    wdev = torch.from_numpy(half_bits(np.ascontiguousarray(w.transpose(3, 0, 1, 2).reshape(cout, -1)), dt)
                            .view(np.int16)).cuda()
    xd = to_dev(x, dt)
    bd = torch.from_numpy(np.asarray(bias, np.float32)).cuda() if bias is not None else None
    ld = y_ld or cout
    y = torch.full((n, ho, wo, ld), -77.0, dtype=DT[dt][1], device='cuda')
    torch.cuda.synchronize()
    _lib.call('fss_conv3x3_c64_nhwc', xd.data_ptr(), cin, wdev.data_ptr(), bd.data_ptr() if bd is not None else None,
              y.data_ptr(), ld, n, h, wd, ho, wo, cout, stride, rate, pads[0], pads[1], _lib.ACT_CODES[act],
              DT[dt][0], None)
    _lib.call('fss_device_sync')
    out = to_host(y)
    assert np.all(out[..., cout:] == -77.0), 'wrote outside its channels'
    return out[..., :cout]


def conv3x3_c128_image(w, dt):
    """[3,3,128,256] filter -> device tensor holding the LDS image fss_conv3x3_c128_nhwc streams."""
    bits = half_bits(np.ascontiguousarray(w.transpose(3, 0, 1, 2).reshape(256, -1)), dt)
    idx = _lib.conv3x3_c128_index()
    return torch.from_numpy(np.ascontiguousarray(bits[idx[:, 0], idx[:, 1]]).view(np.int16)).cuda()


def conv3x3_c128(x, w, bias, act, y_ld=None, x_ld=None, dt='f16'):
    """act(conv3x3(x, w) + bias), stride 1, SAME, through fss_conv3x3_c128_nhwc (input tile + halo in
    LDS).  x: [n,h,w,128]; w: [3,3,128,256].  Returns float32 [n,h,w,256]."""
    n, h, wd, cin = x.shape
    wdev = conv3x3_c128_image(w, dt)
    xd = to_dev(x, dt)
    if x_ld and x_ld != cin:
        wide = torch.full((n, h, wd, x_ld), 3.0, dtype=DT[dt][1], device='cuda')
        wide[..., :cin] = xd
        xd = wide
    bd = torch.from_numpy(np.asarray(bias, np.float32)).cuda() if bias is not None else None
    ld = y_ld or 256
    y = torch.full((n, h, wd, ld), -77.0, dtype=DT[dt][1], device='cuda')
    torch.cuda.synchronize()
    _lib.call('fss_conv3x3_c128_nhwc', xd.data_ptr(), xd.shape[3], wdev.data_ptr(),
              bd.data_ptr() if bd is not None else None, y.data_ptr(), ld, n, h, wd, _lib.ACT_CODES[act],
              DT[dt][0], None)
    _lib.call('fss_device_sync')
    out = to_host(y)
    assert np.all(out[..., 256:] == -77.0), 'wrote outside its channels'
    return out[..., :256]


def zoomed_dilated_conv(x, w, dt='f32'):
    """conv3x3 with dilation 2 (SAME) of the align_corners x2 zoom of x, computed at the source
    resolution: fss_zoom2_border_gather, four fss_conv2d_nhwc launches (x and the three crop
    tensors, same filter, dilation 1, the crops unpadded across their 3 rows / columns),
    fss_zoom2_conv_commute.  Returns float32
    [n,2h-1,2w-1,cout]."""
    n, h, wd, cin = x.shape
    cout = w.shape[3]
    code, tdt, _ = DT[dt]
    xd = to_dev(x, dt)
    wdev, k_pad, cout_pad = pack_conv_weights(w, dt)
    mk = lambda *shape: torch.full(shape, -77.0, dtype=tdt, device='cuda')
    tb, lr, cn = mk(2 * n, 3, wd, cin), mk(2 * n, h, 3, cin), mk(4 * n, 3, 3, cin)
    _lib.call('fss_zoom2_border_gather', xd.data_ptr(), tb.data_ptr(), lr.data_ptr(), cn.data_ptr(),
              n, h, wd, cin, code, None)
    outs = []
    for t, pt, pl in ((xd, 1, 1), (tb, 0, 1), (lr, 1, 0), (cn, 0, 0)):
        nn, hh, ww, _ = t.shape
        oh, ow = (hh if pt else 1), (ww if pl else 1)
        y = mk(nn, oh, ow, cout)
        d = _lib.ConvDesc(n=nn, h=hh, w=ww, cin=cin, cout=cout, kh=3, kw=3, stride=1, dilation=1,
                          pad_top=pt, pad_left=pl, ho=oh, wo=ow, act=_lib.ACT_CODES[None], in_dtype=code,
                          out_dtype=code, y_ld=cout, res_ld=0, k_pad=k_pad, cout_pad=cout_pad, config=0)
        _lib.call('fss_conv2d_nhwc', ctypes.byref(d), t.data_ptr(), wdev.data_ptr(), None, None,
                  y.data_ptr(), None)
        outs.append(y)
    r = mk(n, 2 * h - 1, 2 * wd - 1, cout)
    _lib.call('fss_zoom2_conv_commute', outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
              outs[3].data_ptr(), r.data_ptr(), n, h, wd, cout, code, None)
    _lib.call('fss_device_sync')
    return to_host(r, dt)


def pack_stem_weights(w, dt, mean=None, scale=None):
    """[kh,kw,3,cout] -> device [cout_pad][k_pad], k = tap*4 + channel ('f32x3': the two f16
    matrices of the raw-byte form, with `mean` and `scale` folded in)."""
    kh, kw, cin, cout = w.shape
    assert cin == 3
    k_pad, cout_pad = ctypes.c_int(0), ctypes.c_int(0)
    _lib.call('fss_stem_pack_dims', kh, kw, cout, conv_in_code(dt), ctypes.byref(k_pad), ctypes.byref(cout_pad))
    if dt in ('f32x3', 'f16w'):
        from fss import plan
        return torch.from_numpy(plan.stem_pack_f16x2(w, mean, scale, cout_pad.value, k_pad.value).view(np.int16)).cuda()
    wp = np.zeros((cout_pad.value, k_pad.value), np.float32)
    w4 = np.pad(w, ((0, 0), (0, 0), (0, 1), (0, 0)))
    wp[:cout, :kh * kw * 4] = w4.transpose(3, 0, 1, 2).reshape(cout, -1)
    if dt in ('bf16', 'f16'):
        return torch.from_numpy(half_bits(wp, dt).view(np.int16)).cuda()
    return torch.from_numpy(wp).cuda()


def stem_conv(frames, w, bias, mean, scale, sub, logical_hw, stride, rate, pads, out_hw, act, dt,
              y_ld=None):
    """frames: uint8 [n,hr,wr,3]; returns float32 host array [n,ho,wo,cout]."""
    n, hr, wr, _ = frames.shape
    kh, kw, _, cout = w.shape
    ho, wo = out_hw
    xd = torch.from_numpy(frames).cuda()
    wdev = pack_stem_weights(w, dt, mean, scale)
    bd = torch.from_numpy(np.asarray(bias, np.float32)).cuda()
    ld = y_ld or cout
    y = torch.full((n, ho, wo, ld), -77.0, dtype=DT[dt][1], device='cuda')
    m3 = (ctypes.c_float * 3)(*mean)
    s3 = (ctypes.c_float * 3)(*scale)
    torch.cuda.synchronize()
    _lib.call('fss_stem_conv_u8', xd.data_ptr(), wdev.data_ptr(), bd.data_ptr(), y.data_ptr(),
              n, hr, wr, sub, logical_hw[0], logical_hw[1], m3, s3, cout, kh, kw, stride, rate,
              pads[0], pads[1], ho, wo, _lib.ACT_CODES[act], conv_in_code(dt), ld, None)
    _lib.call('fss_device_sync')
    return to_host(y, dt, cout)


def stem2_conv(frames, w1, b1, w2, b2, mean, scale, sub, logical_hw, pads1, hw1, act1, pads2, hw2, act2,
               y_ld=None, dt='bf16'):
    """Two fused 3x3 stride-2 convolutions from uint8 frames [n,hr,wr,3] (bfloat16);
    returns the float32 host array [n,h2,w2,c2]."""
    n, hr, wr, _ = frames.shape
    c1, c2 = w1.shape[3], w2.shape[3]
    xd = torch.from_numpy(frames).cuda()
    w1d = pack_stem_weights(w1, dt)
    w2d, _, _ = pack_conv_weights(w2, dt)
    b1d = torch.from_numpy(np.asarray(b1, np.float32)).cuda()
    b2d = torch.from_numpy(np.asarray(b2, np.float32)).cuda()
    ld = y_ld or c2
    y = torch.full((n, hw2[0], hw2[1], ld), -77.0, dtype=DT[dt][1], device='cuda')
    m3 = (ctypes.c_float * 3)(*mean)
    s3 = (ctypes.c_float * 3)(*scale)
    torch.cuda.synchronize()
    _lib.call('fss_stem2_conv_u8', xd.data_ptr(), w1d.data_ptr(), b1d.data_ptr(), w2d.data_ptr(),
              b2d.data_ptr(), y.data_ptr(), n, hr, wr, sub, logical_hw[0], logical_hw[1], m3, s3,
              c1, pads1[0], pads1[1], hw1[0], hw1[1], _lib.ACT_CODES[act1],
              c2, pads2[0], pads2[1], hw2[0], hw2[1], _lib.ACT_CODES[act2], DT[dt][0], ld, None)
    _lib.call('fss_device_sync')
    out = to_host(y)
    if ld != c2:
        assert np.all(out[..., c2:] == -77.0), 'kernel wrote outside its channel slice'
    return out[..., :c2]


def avgpool_pyramid(x, levels, dt):
    """x: [n,h,w,c] host array; levels: list of (kh, kw); returns the list of bin maps."""
    n, h, w, c = x.shape
    xd = to_dev(x, dt)
    L = len(levels)
    outs = [torch.full((n, h // kh, w // kw, c), -77.0, dtype=DT[dt][1], device='cuda') for kh, kw in levels]
    ptrs = (ctypes.c_void_p * L)(*[o.data_ptr() for o in outs])
    kh = (ctypes.c_int * L)(*[l[0] for l in levels])
    kw = (ctypes.c_int * L)(*[l[1] for l in levels])
    bh = (ctypes.c_int * L)(*[h // l[0] for l in levels])
    bw = (ctypes.c_int * L)(*[w // l[1] for l in levels])
    work = torch.empty((n * h * sum(bw) * c,), dtype=torch.float32, device='cuda')
    torch.cuda.synchronize()
    _lib.call('fss_avgpool_pyramid_nhwc', xd.data_ptr(), ptrs, kh, kw, bh, bw, L, work.data_ptr(),
              work.numel() * 4, n, h, w, c, DT[dt][0], None)
    _lib.call('fss_device_sync')
    return [to_host(o, dt) for o in outs]
