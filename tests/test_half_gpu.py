"""Half-precision (FSS_F16) forms of the convolution kernels against the numpy oracle.

Inputs are rounded to half precision first, so with float output a kernel must match the
float64 oracle as tightly as the float kernels do (2e-4 of the output scale); with half
output the limit is one rounding of the result (2^-11 of its magnitude <= 1/1024 of the
scale).  The two forms that join a half-precision stretch of a graph to the float one are
covered as well: split-bf16 arithmetic on float input with half output, half arithmetic with
float output."""
import ctypes
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ops as O           # noqa: E402
from test_kernels_gpu import CONV_CASES, STEM2_CASES, _close, _conv_pads, _conv_ref, _gu, _lib   # noqa: E402

HALF_TOL = {'f32': 2e-4, 'f16': 1.0 / 1024}
COMBOS = [('f16', 'f16'), ('f16', 'f32'), ('f32x3', 'f16')]


def _rnd(gu, v, dt):
    return gu.f16_round(v) if dt == 'f16' else v


@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[3] % 8 == 0])
@pytest.mark.parametrize('dt,out_dt', COMBOS)
def test_conv2d_half(case, dt, out_dt):
    gu = _gu()
    n, h, wd, cin, cout, k, stride, rate, mode = case
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    x = _rnd(gu, rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32), dt)
    w = _rnd(gu, rng.normal(0, 1 / np.sqrt(k * k * cin), (k, k, cin, cout)).astype(np.float32), dt)
    bias = rng.normal(0, 0.5, cout).astype(np.float32)
    pads, out_hw = _conv_pads(h, wd, k, stride, rate, mode)
    ref = _conv_ref(x, w, stride, rate, mode) + bias
    got = gu.conv2d(x, w, bias, None, stride, rate, pads, out_hw, None, dt, out_dt=out_dt)
    _close(got, ref, HALF_TOL[out_dt], 'conv %s -> %s %s' % (dt, out_dt, case,))


@pytest.mark.parametrize('dt,out_dt', COMBOS)
@pytest.mark.parametrize('act', [None, 'relu', 'relu6'])
def test_conv2d_half_epilogue(dt, out_dt, act):
    """bias + residual (of the input's type) + activation, output in a wider pixel stride; every
    16x16 configuration gives the same bits, the 32x32 ones are refused."""
    gu, L = _gu(), _lib()
    rng = np.random.default_rng(7)
    n, h, wd, cin, cout = 2, 21, 19, 64, 136
    x = _rnd(gu, rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32), dt)
    w = _rnd(gu, rng.normal(0, 0.1, (3, 3, cin, cout)).astype(np.float32), dt)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    res = _rnd(gu, rng.normal(0, 2, (n, h, wd, cout)).astype(np.float32), dt)
    ref = O.conv2d(x, w, 1, 1, 'SAME', np.float64) + bias + res
    ref = {'relu': O.relu, 'relu6': O.relu6, None: lambda v: v}[act](ref)
    first = None
    for cfg in range(0, L.lib.fss_conv_config_count() + 1):
        info = (ctypes.c_int * 7)()
        if cfg:
            L.call('fss_conv_config_info', cfg, info)
            if info[6] != 16:
                with pytest.raises(L.FssError, match='16x16'):
                    gu.conv2d(x, w, bias, res, 1, 1, (1, 1), (h, wd), act, dt, out_dt=out_dt, config=cfg)
                continue
            if dt == 'f32x3' and (info[4] // 64) % 2:
                continue
        got = gu.conv2d(x, w, bias, res, 1, 1, (1, 1), (h, wd), act, dt, out_dt=out_dt, y_ld=cout + 16, config=cfg)
        _close(got, ref, HALF_TOL[out_dt], 'conv epilogue %s -> %s %s config %d' % (dt, out_dt, act, cfg))
        if cfg:
            first = got if first is None else first
            assert np.array_equal(got, first), 'config %d differs' % cfg


def test_conv2d_half_saturates():
    """Half-precision outputs clamp at +-65504 instead of becoming infinite."""
    gu = _gu()
    x = np.full((1, 8, 8, 64), 16.0, np.float32)
    w = np.full((1, 1, 64, 64), 128.0, np.float32)
    w[..., 1::2] = -128.0
    got = gu.conv2d(x, w, None, None, 1, 1, (0, 0), (8, 8), None, 'f16', out_dt='f16')
    assert np.all(got[..., 0::2] == 65504.0) and np.all(got[..., 1::2] == -65504.0)


@pytest.mark.parametrize('out_dt', ['f16', 'f32'])
def test_conv2d_half_splitk(out_dt):
    gu = _gu()
    rng = np.random.default_rng(9)
    n, h, wd, cin, cout, rate = 1, 33, 65, 256, 256, 4
    x = gu.f16_round(rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32))
    w = gu.f16_round(rng.normal(0, 0.02, (3, 3, cin, cout)).astype(np.float32))
    bias = rng.normal(0, 0.5, cout).astype(np.float32)
    res = gu.f16_round(rng.normal(0, 1, (n, h, wd, cout)).astype(np.float32))
    ref = O.relu(O.conv2d(x, w, 1, rate, 'SAME', np.float64) + bias + res)
    got = gu.conv2d(x, w, bias, res, 1, rate, (rate, rate), (h, wd), 'relu', 'f16', out_dt=out_dt, y_ld=cout + 8,
                    config=22, splits=4)
    _close(got, ref, HALF_TOL[out_dt], 'split-K half conv')


@pytest.mark.parametrize('out_dt', ['f16', 'f32'])
def test_conv2d_dual_half(out_dt):
    """A bottleneck's projection shortcut + last 1x1 convolution as one launch on half tensors."""
    gu = _gu()
    rng = np.random.default_rng(21)
    n, h, wd, cin, cin2, cout = 2, 23, 37, 128, 64, 264
    r = lambda *s: gu.f16_round(rng.normal(0, 1, s).astype(np.float32))
    x, x2 = r(n, h, wd, cin), r(n, h, wd, cin2)
    w = gu.f16_round(rng.normal(0, 0.05, (1, 1, cin, cout)).astype(np.float32))
    w2 = gu.f16_round(rng.normal(0, 0.1, (1, 1, cin2, cout)).astype(np.float32))
    bias = rng.normal(0, 1, cout).astype(np.float32)
    ref = O.relu(O.conv2d(x, w, 1, 1, 'SAME', np.float64) + O.conv2d(x2, w2, 1, 1, 'SAME', np.float64) + bias)
    for cfg in (0, 16, 28):
        out = gu.conv2d_dual(x, w, x2, w2, bias, 1, 1, (0, 0), (h, wd), 'relu', 'f16', config=cfg, out_dt=out_dt)
        _close(out, ref, HALF_TOL[out_dt], 'dual half conv config %d' % cfg)


@pytest.mark.parametrize('n,h,wd,cout,stride,rate,mode', [
    (2, 19, 31, 64, 1, 1, 'SAME'),
    (1, 33, 17, 128, 2, 1, 'SAME'),
    (2, 9, 13, 64, 2, 1, 'FIXED'),
    (1, 21, 20, 128, 1, 2, 'SAME'),
    (1, 40, 67, 128, 2, 1, 'SAME'),       # wide rows at stride 2: taps shared between lanes
    (2, 19, 40, 128, 1, 1, 'SAME'),       # 128 filters at stride 1
])
def test_conv3x3_c64_half(n, h, wd, cout, stride, rate, mode):
    gu = _gu()
    rng = np.random.default_rng(61)
    x = gu.f16_round(rng.normal(0, 1, (n, h, wd, 64)).astype(np.float32))
    w = gu.f16_round(rng.normal(0, 0.05, (3, 3, 64, cout)).astype(np.float32))
    bias = rng.normal(0, 1, cout).astype(np.float32)
    ref = O.relu(_conv_ref(x, w, stride, rate, mode) + bias)
    pads, (ho, wo) = _conv_pads(h, wd, 3, stride, rate, mode)
    out = gu.conv3x3_c64(x, w, bias, stride, rate, pads, (ho, wo), 'relu', y_ld=cout + 8, dt='f16')
    _close(out, ref, 1.0 / 1024, 'resident 3x3, half precision')
    plain = gu.conv2d(x, w, bias=bias, stride=stride, rate=rate, pads=pads, out_hw=(ho, wo), act='relu', dt='f16')
    assert np.mean(plain != out) < 0.01
    _close(out, plain, 1.0 / 1024, 'resident 3x3 vs implicit GEMM, half precision')


@pytest.mark.parametrize('case', STEM2_CASES)
def test_stem2_conv_u8_half(case):
    """fss_stem2_conv_u8 in half precision against the oracle ops chained with the image, the
    weights and the intermediate rounded to half precision."""
    gu = _gu()
    (hr, wr), (hp, wp), sub, c1, c2, act1, act2, y_ld = case
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    frames = rng.integers(0, 256, (2, hr, wr, 3)).astype(np.uint8)
    mean, scale = (123.68, 116.779, 103.939), (1.0, 1.0, 1.0) if sub == 1 else (1.0, 0.5, 2.0)
    w1 = rng.normal(0, 0.02, (3, 3, 3, c1)).astype(np.float32)
    b1 = rng.normal(0, 1, c1).astype(np.float32)
    w2 = rng.normal(0, 1 / np.sqrt(9 * c1), (3, 3, c1, c2)).astype(np.float32)
    b2 = rng.normal(0, 0.5, c2).astype(np.float32)
    img = O.pad_to_bounding_box(frames.astype(np.float32), hp, wp)
    img = ((img - np.asarray(mean, np.float32)) * np.asarray(scale, np.float32)).astype(np.float32)
    if sub == 2:
        img = img[:, ::2, ::2, :]
    hl, wl = img.shape[1:3]
    fn = {'relu': O.relu, 'relu6': O.relu6, None: lambda v: v}
    h1, pt1, pb1 = O.same_padding(hl, 3, 2, 1)
    w1o, pl1, pr1 = O.same_padding(wl, 3, 2, 1)
    h2, pt2, pb2 = O.same_padding(h1, 3, 2, 1)
    w2o, pl2, pr2 = O.same_padding(w1o, 3, 2, 1)
    y1 = fn[act1](O.conv2d(gu.f16_round(img), gu.f16_round(w1), 2, 1, (pt1, pb1, pl1, pr1), np.float64) + b1)
    y1 = gu.f16_round(y1.astype(np.float32))
    ref = fn[act2](O.conv2d(y1, gu.f16_round(w2), 2, 1, (pt2, pb2, pl2, pr2), np.float64) + b2)
    got = gu.stem2_conv(frames, w1, b1, w2, b2, mean, scale, sub, (hl, wl), (pt1, pl1), (h1, w1o), act1,
                        (pt2, pl2), (h2, w2o), act2, y_ld=y_ld, dt='f16')
    _close(got, ref, 1.0 / 1024, 'stem2 half %s' % (case,))


def test_half_outputs_round_to_nearest_even():
    """The half-precision store of the convolution epilogue is numpy's float16 rounding, bit for
    bit (values with 16 significant bits pass through a split-bf16 identity convolution exactly)."""
    gu = _gu()
    rng = np.random.default_rng(3)
    x = (rng.integers(0, 65536, (1, 16, 16, 64)) / 4096.0).astype(np.float32) * rng.choice([-1.0, 1.0], (1, 16, 16, 64))
    x = x.astype(np.float32)
    w = np.eye(64, dtype=np.float32).reshape(1, 1, 64, 64)
    got = gu.conv2d(x, w, None, None, 1, 1, (0, 0), (16, 16), None, 'f32x3', out_dt='f16')
    assert np.array_equal(got, x.astype(np.float16).astype(np.float32))


WIDE_TOL = {'f32x3': 2e-5, 'f16': 1.0 / 1024}


@pytest.mark.parametrize('case', [c for c in CONV_CASES if c[3] % 8 == 0])
@pytest.mark.parametrize('out_dt', ['f32x3', 'f16'])
def test_conv2d_half_activations_wide_weights(case, out_dt):
    """FSS_F16_W2: half-precision activations against float weights kept as two f16 numbers.
    The weights are NOT rounded here, so with split-bfloat16 output the kernel must match the
    float64 oracle to the 16 bits that output holds - an order of magnitude tighter than the float
    kernels' 2e-4, and out of reach with weights rounded to half precision."""
    gu = _gu()
    n, h, wd, cin, cout, k, stride, rate, mode = case
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    x = gu.f16_round(rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32))
    w = rng.normal(0, 1 / np.sqrt(k * k * cin), (k, k, cin, cout)).astype(np.float32)
    bias = rng.normal(0, 0.5, cout).astype(np.float32)
    pads, out_hw = _conv_pads(h, wd, k, stride, rate, mode)
    ref = _conv_ref(x, w, stride, rate, mode) + bias
    if out_dt == 'f32x3' and cout % 8:
        pytest.skip('split-bfloat16 outputs come in groups of 8 channels')
    got = gu.conv2d(x, w, bias, None, stride, rate, pads, out_hw, None, 'f16w', out_dt=out_dt)
    _close(got, ref, WIDE_TOL[out_dt], 'conv f16 x two-f16 weights -> %s %s' % (out_dt, case,))


def test_conv2d_wide_weights_epilogue_and_configurations():
    """Residual (split-bfloat16, the output's type) + relu, output in a wider pixel stride; every
    configuration whose ring fits the LDS with two weight tiles per stage gives the same bits,
    the others are refused; and the two-input form (projection shortcut + last 1x1)."""
    gu, L = _gu(), _lib()
    rng = np.random.default_rng(9)
    n, h, wd, cin, cout = 2, 21, 19, 128, 264
    x = gu.f16_round(rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32))
    w = rng.normal(0, 0.1, (1, 1, cin, cout)).astype(np.float32)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    res = gu.x3_round(rng.normal(0, 2, (n, h, wd, cout)).astype(np.float32))
    ref = O.relu(O.conv2d(x, w, 1, 1, 'SAME', np.float64) + bias + res)
    first, ran, refused = None, 0, 0
    info = (ctypes.c_int * 7)()
    for cfg in range(1, L.lib.fss_conv_config_count() + 1):
        L.call('fss_conv_config_info', cfg, info)
        fits = info[6] == 16 and info[5] * (info[0] + 2 * info[1]) * info[4] < 160 * 1024 and cfg != 35
        if not fits:
            with pytest.raises(L.FssError, match='16x16'):
                gu.conv2d(x, w, bias, res, 1, 1, (0, 0), (h, wd), 'relu', 'f16w', out_dt='f32x3', config=cfg)
            refused += 1
            continue
        got = gu.conv2d(x, w, bias, res, 1, 1, (0, 0), (h, wd), 'relu', 'f16w', out_dt='f32x3', y_ld=cout + 16,
                        config=cfg)
        _close(got, ref, 2e-5, 'two-f16-weight conv, config %d' % cfg)
        first = got if first is None else first
        assert np.array_equal(got, first), 'config %d differs' % cfg
        ran += 1
    assert ran >= 10 and refused >= 10
    x2 = gu.f16_round(rng.normal(0, 1, (n, h, wd, 64)).astype(np.float32))
    w2 = rng.normal(0, 0.1, (1, 1, 64, cout)).astype(np.float32)
    ref2 = O.relu(O.conv2d(x, w, 1, 1, 'SAME', np.float64) + O.conv2d(x2, w2, 1, 1, 'SAME', np.float64) + bias)
    for cfg in (0, 14, 27):
        out = gu.conv2d_dual(x, w, x2, w2, bias, 1, 1, (0, 0), (h, wd), 'relu', 'f16w', config=cfg, out_dt='f32x3')
        _close(out, ref2, 2e-5, 'two-input two-f16-weight conv, config %d' % cfg)


def test_maxpool_and_depthwise_half():
    """fss_maxpool_nhwc (exact: selection only) and fss_dwconv2d_nhwc (float arithmetic, one
    rounding of the result) on half-precision tensors, both depthwise kernels."""
    import torch
    gu, L = _gu(), _lib()
    rng = np.random.default_rng(14)
    x = gu.f16_round(rng.normal(0, 1, (2, 37, 41, 64)).astype(np.float32))
    ho, pt, _ = O.same_padding(37, 3, 2)
    wo, pl, _ = O.same_padding(41, 3, 2)
    y = torch.empty((2, ho, wo, 64), dtype=torch.float16, device='cuda')
    L.call('fss_maxpool_nhwc', gu.to_dev(x, 'f16').data_ptr(), y.data_ptr(), 2, 37, 41, 64, 3, 2, pt, pl, ho, wo,
           L.F16, None)
    assert np.array_equal(gu.to_host(y), O.max_pool_same(x, 3, 2))
    for (n, h, wd, c, stride, rate) in [(8, 129, 257, 96, 1, 4), (12, 200, 258, 64, 2, 1), (2, 20, 26, 24, 2, 1),
                                        (1, 15, 22, 16, 1, 3)]:
        x = gu.f16_round(rng.normal(0, 1, (n, h, wd, c)).astype(np.float32))
        w = rng.normal(0, 0.4, (3, 3, c, 1)).astype(np.float32)
        bias = rng.normal(0, 0.5, c).astype(np.float32)
        ref = O.relu6(O.depthwise_conv2d(x, w, stride, rate, np.float64) + bias)
        ho, pt, _ = O.same_padding(h, 3, stride, rate)
        wo, pl, _ = O.same_padding(wd, 3, stride, rate)
        wdv = torch.from_numpy(w[:, :, :, 0].reshape(9, c).copy()).cuda()
        bd = torch.from_numpy(bias).cuda()
        y = torch.empty((n, ho, wo, c), dtype=torch.float16, device='cuda')
        L.call('fss_dwconv2d_nhwc', gu.to_dev(x, 'f16').data_ptr(), wdv.data_ptr(), bd.data_ptr(), y.data_ptr(),
               n, h, wd, c, 3, 3, stride, rate, pt, pl, ho, wo, L.ACT_RELU6, L.F16, None)
        _close(gu.to_host(y), ref, 1.0 / 1024, 'half-precision depthwise %s' % ((n, h, wd, c, stride, rate),))


@pytest.mark.parametrize('k,stride,cout,sub', [(7, 2, 64, 2), (3, 2, 32, 1), (3, 1, 24, 1)])
def test_stem_conv_u8_raw_bytes_half_output(k, stride, cout, sub):
    """fss_stem_conv_u8 with dtype FSS_F16_W2: the raw frame bytes against two-f16 weights (mean
    and scale folded into them) with half-precision output - one rounding of the result away
    from the float64 oracle, borders of the padded image included."""
    gu = _gu()
    rng = np.random.default_rng(15)
    frames = rng.integers(0, 256, (2, 45, 77, 3)).astype(np.uint8)
    hp, wp = 47, 80
    mean, scale = (123.68, 116.779, 103.939), (1.0, 0.5, 2.0 / 255)
    w = rng.normal(0, 0.05, (k, k, 3, cout)).astype(np.float32)
    bias = rng.normal(0, 0.5, cout).astype(np.float32)
    img = (O.pad_to_bounding_box(frames.astype(np.float64), hp, wp) - np.asarray(mean)) * np.asarray(scale)
    img = img[:, ::sub, ::sub]
    hl, wl = img.shape[1:3]
    ho, pt, pb = O.same_padding(hl, k, stride, 1)
    wo, pl, pr = O.same_padding(wl, k, stride, 1)
    ref = O.relu6(O.conv2d(img, w, stride, 1, (pt, pb, pl, pr), np.float64) + bias)
    got = gu.stem_conv(frames, w, bias, mean, scale, sub, (hl, wl), stride, 1, (pt, pl), (ho, wo), 'relu6', 'f16w')
    _close(got, ref, 1.0 / 1024, 'raw-byte stem, half output')


@pytest.mark.parametrize('n,h,wd,dt,act', [
    (2, 9, 70, 'f16', 'relu6'),       # one full band, a ragged band of 6 columns, a 1-row last tile
    (1, 13, 129, 'f16', 'relu'),      # two bands and a single-column band (one tall tile)
    (3, 6, 37, 'bf16', 'relu6'),      # ragged band only
    (2, 8, 64, 'f16', None),          # no ragged band
    (1, 129, 65, 'f16', 'relu6'),     # the real height: 33 row tiles and the 129-row column tile
    (12, 21, 257, 'bf16', 'relu'),    # the real width; 300 tiles: some blocks take a second one
])
def test_conv3x3_c128_matches_implicit_gemm(n, h, wd, dt, act):
    """fss_conv3x3_c128_nhwc (persistent blocks, input tile + halo in LDS) gives the bits of fss_conv2d_nhwc
    on the same operands: regular tiles, ragged right-hand bands, partial last tiles, wide pixel strides."""
    gu = _gu()
    rng = np.random.RandomState(n * 1000 + h * 10 + wd)
    x = rng.randn(n, h, wd, 128).astype(np.float32)
    w = (rng.randn(3, 3, 128, 256) * 0.03).astype(np.float32)
    bias = rng.randn(256).astype(np.float32)
    ref = gu.conv2d(x, w, bias, None, 1, 1, (1, 1), (h, wd), act, dt=dt)
    out = gu.conv3x3_c128(x, w, bias, act, y_ld=256 + 8, x_ld=128 + 16 if wd == 70 else None, dt=dt)
    assert out.shape == ref.shape
    assert np.array_equal(out, ref), np.abs(out - ref).max()
