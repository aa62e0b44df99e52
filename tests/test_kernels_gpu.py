"""Parity of every C-ABI kernel against the numpy oracle (seeded inputs).

Float tolerances (stated per test): float32 kernels must match the float64
oracle to 2e-4 of the output scale (the reference's float32 runtime itself is
only that exact); bfloat16 kernels are fed bf16-representable inputs, so with
float32 output they must match as tightly, and with bf16 output to 1/128
(one bf16 rounding).  Integer outputs (labels, confusion matrix) are bit-exact.
"""
import ctypes
import zlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import ops as O           # noqa: E402
from oracle import metrics as OM      # noqa: E402


def _lib():
    from fss import _lib as L
    return L


def _gu():
    import gpu_util
    return gpu_util


def _close(got, ref, rtol, what=''):
    scale = max(float(np.abs(ref).max()), 1e-6)
    err = float(np.abs(got - ref).max()) / scale
    assert err <= rtol, '%s: max err %.3g of scale (limit %.3g)' % (what, err, rtol)


CONV_CASES = [
    # n, h, w, cin, cout, k, stride, rate, mode
    (1, 17, 23, 8, 19, 1, 1, 1, 'SAME'),       # classifier-like, N tile 32
    (2, 13, 17, 16, 64, 3, 1, 1, 'SAME'),      # N tile 64
    (1, 33, 65, 32, 96, 3, 1, 4, 'SAME'),      # dilation 4, 64x64 tiles
    (1, 33, 65, 64, 128, 3, 1, 16, 'SAME'),    # dilation 16 on a 33-row map
    (1, 40, 52, 8, 40, 3, 2, 1, 'SAME'),       # strided SAME on even sizes (asymmetric pad)
    (1, 41, 53, 8, 24, 3, 2, 1, 'FIXED'),      # conv2d_same explicit padding, stride 2
    (1, 37, 45, 8, 32, 7, 2, 1, 'FIXED'),      # 7x7/2 root conv shape
    (3, 96, 100, 64, 256, 1, 1, 1, 'SAME'),    # M = 28800: 128x128 tiles
    (1, 65, 129, 40, 72, 3, 1, 2, 'SAME'),     # cin, cout not multiples of 32
    (1, 9, 9, 24, 144, 1, 1, 1, 'SAME'),       # K smaller than one K tile
]


def _conv_ref(x, w, stride, rate, mode, dtype=np.float64):
    if mode == 'SAME':
        return O.conv2d(x, w, stride, rate, 'SAME', dtype)
    return O.conv2d_same(x, w, stride, rate, dtype)


def _conv_pads(h, wd, k, stride, rate, mode):
    if mode == 'SAME':
        ho, pt, _ = O.same_padding(h, k, stride, rate)
        wo, pl, _ = O.same_padding(wd, k, stride, rate)
        return (pt, pl), (ho, wo)
    pb, pe = O.fixed_padding(k, rate)
    keff = k + (k - 1) * (rate - 1)
    return (pb, pb), ((h + pb + pe - keff) // stride + 1, (wd + pb + pe - keff) // stride + 1)


@pytest.mark.parametrize('case', CONV_CASES)
@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_conv2d(case, dt):
    gu = _gu()
    n, h, wd, cin, cout, k, stride, rate, mode = case
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    x = rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32)
    w = rng.normal(0, 1 / np.sqrt(k * k * cin), (k, k, cin, cout)).astype(np.float32)
    bias = rng.normal(0, 0.5, cout).astype(np.float32)
    if dt == 'bf16':
        x, w = gu.bf16_round(x), gu.bf16_round(w)
    pads, out_hw = _conv_pads(h, wd, k, stride, rate, mode)
    ref = _conv_ref(x, w, stride, rate, mode) + bias
    assert ref.shape[1:3] == out_hw
    got = gu.conv2d(x, w, bias, None, stride, rate, pads, out_hw, None, dt,
                    out_dt='f32')
    _close(got, ref, 2e-4, 'conv %s %s' % (dt, case,))


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
@pytest.mark.parametrize('act', [None, 'relu', 'relu6'])
def test_conv2d_epilogue(dt, act):
    """bias + residual + activation, output written into a wider pixel stride."""
    gu = _gu()
    rng = np.random.default_rng(7)
    n, h, wd, cin, cout = 2, 21, 19, 32, 48
    x = rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32)
    w = rng.normal(0, 0.2, (3, 3, cin, cout)).astype(np.float32)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    res = rng.normal(0, 2, (n, h, wd, cout)).astype(np.float32)
    if dt == 'bf16':
        x, w, res = gu.bf16_round(x), gu.bf16_round(w), gu.bf16_round(res)
    ref = O.conv2d(x, w, 1, 1, 'SAME', np.float64) + bias + res
    ref = {'relu': O.relu, 'relu6': O.relu6, None: lambda v: v}[act](ref)
    got = gu.conv2d(x, w, bias, res, 1, 1, (1, 1), (h, wd), act, dt, y_ld=cout + 16)
    _close(got, ref, 2e-4 if dt != 'bf16' else 1.0 / 128, 'conv epilogue %s %s' % (dt, act))


SPLITK_CASES = [
    # n, h, w, cin, cout, k, rate, splits, config
    (1, 33, 65, 256, 256, 3, 4, 4, 22),     # backbone block at one frame; taps skipped at the border
    (1, 33, 65, 512, 96, 1, 1, 2, 21),      # 1x1
    (2, 9, 11, 64, 40, 3, 16, 8, 22),       # most taps read only padding: fewer K tiles than splits
    (1, 17, 23, 24, 72, 3, 2, 3, 0),        # cin below the K tile (per-lane tap walk), automatic config
]


@pytest.mark.parametrize('case', SPLITK_CASES)
@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_conv2d_splitk_matches_the_plain_launch(case, dt):
    """fss_conv2d_splitk_nhwc = fss_conv2d_nhwc up to the order of the float additions, with bias,
    residual, activation and a wider output pixel stride; also against the oracle."""
    gu = _gu()
    n, h, wd, cin, cout, k, rate, splits, config = case
    if dt == 'f32x3' and config == 21:
        config = 22                          # the split-bf16 mode needs an even number of k steps
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    x = rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32)
    w = rng.normal(0, 1 / np.sqrt(k * k * cin), (k, k, cin, cout)).astype(np.float32)
    bias = rng.normal(0, 0.5, cout).astype(np.float32)
    res = rng.normal(0, 1, (n, h, wd, cout)).astype(np.float32)
    if dt == 'bf16':
        x, w, res = gu.bf16_round(x), gu.bf16_round(w), gu.bf16_round(res)
    pads, out_hw = _conv_pads(h, wd, k, 1, rate, 'SAME')
    ref = O.relu(_conv_ref(x, w, 1, rate, 'SAME') + bias + res)
    plain = gu.conv2d(x, w, bias, res, 1, rate, pads, out_hw, 'relu', dt, y_ld=cout + 8, config=config)
    split = gu.conv2d(x, w, bias, res, 1, rate, pads, out_hw, 'relu', dt, y_ld=cout + 8, config=config,
                      splits=splits)
    tol = 1.0 / 128 if dt == 'bf16' else 2e-4
    _close(split, ref, tol, 'split-K conv %s %s' % (dt, case,))
    # same products, another summation order: one output rounding step apart at most
    _close(split, plain, 1.0 / 200 if dt == 'bf16' else 1e-5, 'split-K vs plain %s %s' % (dt, case,))
    # float output of bf16 inputs (the classifier form of the plain entry point)
    if dt == 'bf16':
        s32 = gu.conv2d(x, w, bias, None, 1, rate, pads, out_hw, None, dt, out_dt='f32', config=config,
                        splits=splits)
        _close(s32, _conv_ref(x, w, 1, rate, 'SAME') + bias, 2e-4, 'split-K float output %s' % (case,))


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_conv2d_configs_agree_bitwise(dt):
    """Tile configurations with the same MFMA shape accumulate each output over k
    in the same order: results must be bit-identical within a shape class, and
    every configuration (and the automatic choice) must match the oracle."""
    gu, L = _gu(), _lib()
    rng = np.random.default_rng(11)
    n, h, wd, cin, cout = 1, 37, 41, 64, 136
    x = rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32)
    w = rng.normal(0, 0.1, (3, 3, cin, cout)).astype(np.float32)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    ref = O.relu(O.conv2d(gu.bf16_round(x) if dt == 'bf16' else x,
                          gu.bf16_round(w) if dt == 'bf16' else w, 1, 2, 'SAME', np.float64) + bias)
    by_shape = {}
    for cfg in range(0, L.lib.fss_conv_config_count() + 1):
        info = (ctypes.c_int * 7)()
        if cfg:
            L.call('fss_conv_config_info', cfg, info)
            if dt == 'f32x3' and (info[4] // (16 * (64 // info[6]))) % 2:
                # the split mode pairs k steps: an odd number per tile is rejected
                with pytest.raises(L.FssError, match='even number of k steps'):
                    gu.conv2d(x, w, bias, None, 1, 2, (2, 2), (h, wd), 'relu', dt, config=cfg)
                continue
        out = gu.conv2d(x, w, bias, None, 1, 2, (2, 2), (h, wd), 'relu', dt, config=cfg)
        _close(out, ref, 2e-4 if dt != 'bf16' else 1.0 / 128, 'conv config %d' % cfg)
        if cfg:
            assert info[0] % 32 == 0 and info[1] % 32 == 0 and info[6] in (16, 32)
            first = by_shape.setdefault(info[6], out)
            assert np.array_equal(out, first), 'config %d differs inside its shape class' % cfg
    assert sorted(by_shape) == [16, 32]


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_conv2d_dual_is_the_sum_of_two_convolutions(dt):
    """fss_conv2d_dual_nhwc = relu(conv3x3_dilated(x) + conv1x1(x2) + bias) (the cascade
    feature fusion) against the oracle's two convolutions added in float64; every tile
    configuration of a MFMA shape class must give the same bits."""
    gu, L = _gu(), _lib()
    rng = np.random.default_rng(21)
    n, h, wd, cin, cin2, cout = 2, 23, 37, 128, 64, 136
    x = rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32)
    x2 = rng.normal(0, 1, (n, h, wd, cin2)).astype(np.float32)
    w = rng.normal(0, 0.05, (3, 3, cin, cout)).astype(np.float32)
    w2 = rng.normal(0, 0.1, (1, 1, cin2, cout)).astype(np.float32)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    if dt == 'bf16':
        x, x2, w, w2 = (gu.bf16_round(v) for v in (x, x2, w, w2))
    ref = O.relu(O.conv2d(x, w, 1, 2, 'SAME', np.float64) + O.conv2d(x2, w2, 1, 1, 'SAME', np.float64) + bias)
    by_shape = {}
    for cfg in range(0, L.lib.fss_conv_config_count() + 1):
        info = (ctypes.c_int * 7)()
        if cfg:
            L.call('fss_conv_config_info', cfg, info)
            if dt == 'f32x3' and (info[4] // (16 * (64 // info[6]))) % 2:
                continue        # odd number of k steps per tile: not offered in the split mode
        out = gu.conv2d_dual(x, w, x2, w2, bias, 1, 2, (2, 2), (h, wd), 'relu', dt, config=cfg)
        _close(out, ref, 2e-4 if dt != 'bf16' else 1.0 / 128, 'dual conv config %d' % cfg)
        if cfg:
            first = by_shape.setdefault(info[6], out)
            assert np.array_equal(out, first), 'config %d differs inside its shape class' % cfg


@pytest.mark.parametrize('dt', ['bf16', 'f32x3'])
@pytest.mark.parametrize('cout,cin2,with_res', [(256, 64, False), (128, 0, True), (96, 128, False)])
def test_conv2d_head_is_conv_then_classifier(cout, cin2, with_res, dt):
    """fss_conv2d_dual_head_nhwc = the two-input convolution (bf16: rounded to bfloat16 as the
    unfused plan stores it; float32x3: not rounded), followed by the 1x1 classifier, against
    the oracle in float64; every configuration that can hold the layer must agree to float
    summation order, the others must refuse."""
    gu, L = _gu(), _lib()
    rng = np.random.default_rng(22)
    n, h, wd, cin, ncls = 2, 19, 31, 64, 19
    rnd = gu.bf16_round if dt == 'bf16' else (lambda v: v)
    x = rnd(rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32))
    w = rnd(rng.normal(0, 0.05, (3, 3, cin, cout)).astype(np.float32))
    x2 = w2 = None
    conv = O.conv2d(x, w, 1, 2, 'SAME', np.float64)
    if cin2:
        x2 = rnd(rng.normal(0, 1, (n, h, wd, cin2)).astype(np.float32))
        w2 = rnd(rng.normal(0, 0.1, (1, 1, cin2, cout)).astype(np.float32))
        conv = conv + O.conv2d(x2, w2, 1, 1, 'SAME', np.float64)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    w_head = rnd(rng.normal(0, 0.1, (cout, ncls)).astype(np.float32))
    b_head = rng.normal(0, 1, ncls).astype(np.float32)
    res = None
    if with_res:
        res = rnd(rng.normal(0, 1, (n, h, wd, cout)).astype(np.float32))
        conv = conv + res
    act = O.relu(conv + bias)
    if dt == 'bf16':
        act = gu.bf16_round(act.astype(np.float32)).astype(np.float64)
    ref = act.reshape(-1, cout) @ w_head.astype(np.float64) + b_head
    ref = ref.reshape(n, h, wd, ncls)
    ran = 0
    for cfg in range(0, L.lib.fss_conv_config_count() + 1):
        info = (ctypes.c_int * 7)()
        if cfg:
            L.call('fss_conv_config_info', cfg, info)
        # float32x3: only tiles with an even number of k steps (128 bytes of K per row)
        can = info[6] == 16 and info[1] >= max(cout, 128) and (dt == 'bf16' or info[4] == 128)
        if cfg and not can:
            with pytest.raises(RuntimeError, match='16x16x32'):
                gu.conv2d_dual_head(x, w, x2, w2, bias, w_head, b_head, 2, (2, 2), (h, wd), config=cfg, dt=dt)
            continue
        out = gu.conv2d_dual_head(x, w, x2, w2, bias, w_head, b_head, 2, (2, 2), (h, wd), config=cfg,
                                  y_ld=20 if cfg % 2 else None, residual=res, dt=dt)
        if dt == 'bf16':
            # a bfloat16 rounding of an activation that sits on a tie may go the other way
            # (the conv sum is formed in another order than the oracle's): 2^-8 * |w_head| * act
            _close(out, ref, 2e-2, 'head conv config %d' % cfg)
            assert np.mean(np.abs(out - ref) > 2e-3) < 0.01, 'config %d' % cfg
        else:
            _close(out, ref, 2e-4, 'float32x3 head conv config %d' % cfg)
        ran += 1
    assert ran >= (3 if dt == 'bf16' else 1)


@pytest.mark.parametrize('n,h,wd,cout,stride,rate,mode', [
    (2, 19, 31, 64, 1, 1, 'SAME'),        # block1 conv2; pixels do not fill the last pass
    (1, 33, 17, 128, 2, 1, 'SAME'),       # `Conv_2`: stride 2 on odd sizes
    (1, 16, 24, 64, 2, 1, 'SAME'),        # stride 2 on even sizes (asymmetric padding)
    (2, 9, 13, 64, 2, 1, 'FIXED'),        # conv2d_same: explicit symmetric padding, then VALID
    (1, 21, 20, 128, 1, 2, 'SAME'),       # dilation 2
    (3, 40, 67, 64, 1, 1, 'SAME'),        # several passes per block, rows that wrap inside a pixel group
    (1, 3, 3, 64, 1, 1, 'SAME'),          # every tap of most pixels on the padding
    (1, 40, 67, 128, 2, 1, 'SAME'),       # rows of 34 outputs at stride 2: groups that do not wrap,
    (2, 21, 70, 64, 2, 1, 'SAME'),        # i.e. tap 2 taken from the next lane (DPP), 64 and 128 filters
    (2, 19, 40, 128, 1, 1, 'SAME'),       # 128 filters at stride 1, dilation 1 (no weight pipelining)
])
def test_conv3x3_c64_matches_oracle(n, h, wd, cout, stride, rate, mode):
    """fss_conv3x3_c64_nhwc (filter resident in LDS) against the oracle's convolution in float64 and
    against fss_conv2d_nhwc on the same inputs (same products, another summation order)."""
    gu = _gu()
    rng = np.random.default_rng(61)
    x = gu.bf16_round(rng.normal(0, 1, (n, h, wd, 64)).astype(np.float32))
    w = gu.bf16_round(rng.normal(0, 0.05, (3, 3, 64, cout)).astype(np.float32))
    bias = rng.normal(0, 1, cout).astype(np.float32)
    ref = O.relu(_conv_ref(x, w, stride, rate, mode) + bias)
    pads, (ho, wo) = _conv_pads(h, wd, 3, stride, rate, mode)
    assert ref.shape[1:3] == (ho, wo)
    out = gu.conv3x3_c64(x, w, bias, stride, rate, pads, (ho, wo), 'relu', y_ld=cout + 8)
    _close(out, ref, 1.0 / 128, 'resident 3x3')
    plain = gu.conv2d(x, w, bias=bias, stride=stride, rate=rate, pads=pads, out_hw=(ho, wo), act='relu', dt='bf16')
    assert np.mean(plain != out) < 0.01
    _close(out, plain, 1.0 / 128, 'resident 3x3 vs implicit GEMM')


@pytest.mark.parametrize('n,h,wd,dt', [(2, 9, 70, 'bf16'), (1, 13, 129, 'f16'), (1, 3, 3, 'f16')])
def test_conv3x3_c128_matches_oracle(n, h, wd, dt):
    """fss_conv3x3_c128_nhwc (persistent blocks, input tile and halo in LDS) against the oracle's
    convolution in float64: a full band with a ragged one, the single-column band, a frame smaller
    than the filter's reach."""
    gu = _gu()
    rng = np.random.default_rng(67)
    rnd = gu.bf16_round if dt == 'bf16' else gu.f16_round
    x = rnd(rng.normal(0, 1, (n, h, wd, 128)).astype(np.float32))
    w = rnd(rng.normal(0, 0.03, (3, 3, 128, 256)).astype(np.float32))
    bias = rng.normal(0, 1, 256).astype(np.float32)
    ref = np.minimum(O.relu(_conv_ref(x, w, 1, 1, 'SAME') + bias), 6.0)
    out = gu.conv3x3_c128(x, w, bias, 'relu6', y_ld=256 + 8, dt=dt)
    _close(out, ref, 1.0 / 128 if dt == 'bf16' else 1.0 / 1024, 'halo 3x3')


@pytest.mark.parametrize('cin,cin2,cout1,cout2,with_res,pixels', [
    (64, 0, 256, 64, True, (2, 19, 31)),       # ICNet-1.0 block1: conv3 + identity shortcut -> conv1
    (64, 64, 256, 64, False, (1, 9, 17)),      # ... conv3 + projection shortcut -> conv1
    (32, 0, 128, 32, True, (3, 7, 5)),         # the half-width model; 105 pixels: a partial wave
    (32, 32, 128, 32, False, (1, 1, 1)),       # one pixel
    (128, 0, 256, 32, False, (2, 40, 67)),     # several passes per block
])
def test_conv1x1_chain_is_two_convolutions(cin, cin2, cout1, cout2, with_res, pixels):
    """fss_conv1x1_chain_nhwc against the oracle: y = relu(conv1x1([x | x2]) + bias + residual)
    in float64, rounded to bfloat16 as it is stored, then y2 = relu6(conv1x1(y) + bias2); the
    first output must also equal the plain two-input convolution bit for bit up to summation
    order, and the padding channels of wider output rows stay untouched."""
    gu = _gu()
    rng = np.random.default_rng(51)
    n, h, wd = pixels
    r16 = lambda *shape: gu.bf16_round(rng.normal(0, 1, shape).astype(np.float32))
    x = r16(n, h, wd, cin)
    x2 = r16(n, h, wd, cin2) if cin2 else None
    w = gu.bf16_round(rng.normal(0, 0.1, (cin + cin2, cout1)).astype(np.float32))
    w2 = gu.bf16_round(rng.normal(0, 0.1, (cout1, cout2)).astype(np.float32))
    bias, bias2 = rng.normal(0, 1, cout1).astype(np.float32), rng.normal(0, 1, cout2).astype(np.float32)
    res = r16(n, h, wd, cout1) if with_res else None
    xin = x if x2 is None else np.concatenate([x, x2], axis=3)
    pre = xin.reshape(-1, cin + cin2).astype(np.float64) @ w.astype(np.float64) + bias
    if with_res:
        pre = pre + res.reshape(-1, cout1)
    ref1 = O.relu(pre)
    y, y2 = gu.conv1x1_chain(x, x2, w, bias, res, 'relu', w2, bias2, 'relu6', y_ld=cout1 + 8, y2_ld=cout2 + 16)
    _close(y.reshape(-1, cout1), ref1, 1.0 / 128, 'chain: first output')
    # the second layer reads the first output as stored
    ref2 = np.clip(y.reshape(-1, cout1).astype(np.float64) @ w2.astype(np.float64) + bias2, 0, 6)
    _close(y2.reshape(-1, cout2), ref2, 1.0 / 128, 'chain: second output')
    # and equals the unfused kernels on the same inputs
    if cin % 64 == 0 and cin2 % 64 == 0:
        w4 = w[:cin].reshape(1, 1, cin, cout1)
        if cin2:
            plain = gu.conv2d_dual(x, w4, x2, w[cin:].reshape(1, 1, cin2, cout1), bias, 1, 1, (0, 0), (h, wd),
                                   'relu', 'bf16')
        else:
            plain = gu.conv2d(x, w4, bias=bias, residual=res, out_hw=(h, wd), act='relu', dt='bf16')
        assert np.mean(plain != y) < 0.01, 'first output differs from the plain convolution on too many values'
        _close(y, plain, 1.0 / 128, 'chain vs plain convolution')


@pytest.mark.parametrize('dt', ['f32', 'bf16'])
@pytest.mark.parametrize('shape', [(2, 4, 4, 8, 24), (1, 5, 9, 16, 40), (2, 33, 17, 64, 64)])
def test_dilated_conv_of_a_zoom_at_the_source_resolution(shape, dt):
    """conv3x3_dilation2(zoom_x2(x)) from fss_zoom2_border_gather + the plain conv3x3 of x and
    of its border crops + fss_zoom2_conv_commute, against the oracle's zoom followed by the
    dilated convolution (float64), including the smallest legal map (4x4) where every output
    pixel is near a border."""
    gu = _gu()
    n, h, wd, cin, cout = shape
    rng = np.random.default_rng(31)
    x = rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32)
    w = rng.normal(0, 0.1, (3, 3, cin, cout)).astype(np.float32)
    if dt == 'bf16':
        x, w = gu.bf16_round(x), gu.bf16_round(w)
    up = O.resize_bilinear(x.astype(np.float64), 2 * h - 1, 2 * wd - 1, True)
    ref = O.conv2d(up, w, 1, 2, 'SAME', np.float64)
    out = gu.zoomed_dilated_conv(x, w, dt)
    assert out.shape == ref.shape
    # bf16: the source-resolution results are rounded to bfloat16 before they are averaged
    _close(out, ref, 2e-5 if dt == 'f32' else 1.0 / 128, 'zoomed dilated conv')
    if dt == 'f32':     # the border rows / columns are as exact as the interior
        for sl in (np.s_[:, 1], np.s_[:, -2], np.s_[:, :, 1], np.s_[:, :, -2]):
            _close(out[sl], ref[sl], 2e-5, 'border')


@pytest.mark.parametrize('dt', ['bf16', 'f32x3', 'f16w'])
@pytest.mark.parametrize('n,zh,zw,cout', [(2, 6, 9, 128), (3, 4, 4, 128), (3, 9, 40, 256), (2, 5, 70, 128)])
def test_head_conv_interpolates_its_residual_from_the_source_resolution(n, zh, zw, cout, dt):
    """fss_conv2d_dual_head_nhwc with zoom_residual = {y, ytb, ylr, ycn}: the epilogue adds the
    corrected x2 zoom of y (border rows / columns from the three substitute tensors) exactly
    as fss_zoom2_conv_commute would have stored it; checked against float64.  The shapes cover
    the smallest legal map (7 pixels wide: several row wraps inside 16 consecutive pixels),
    frames that end inside a wave's pixel group, and rows wide enough that whole groups lie on
    one even output row (two sources instead of four)."""
    gu = _gu()
    rng = np.random.default_rng(41)
    cin, ncls = 64, 19
    h, wd = 2 * zh - 1, 2 * zw - 1
    rnd = gu.bf16_round if dt == 'bf16' else (lambda v: v)
    r16 = lambda *shape: rnd(rng.normal(0, 1, shape).astype(np.float32))
    x, w = r16(n, h, wd, cin), rnd(rng.normal(0, 0.1, (1, 1, cin, cout)).astype(np.float32))
    if dt == 'f16w':        # half-precision activations, weights of float quality
        x = gu.f16_round(x)
    y, ytb, ylr, ycn = r16(n, zh, zw, cout), r16(2 * n, 1, zw, cout), r16(2 * n, zh, 1, cout), r16(4 * n, 1, 1, cout)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    w_head = rnd(rng.normal(0, 0.1, (cout, ncls)).astype(np.float32))
    b_head = rng.normal(0, 1, ncls).astype(np.float32)
    # the corrected zoom in float64
    z = np.zeros((n, h, wd, cout))
    for p in range(h):
        for q in range(wd):
            srcs = []
            for i in range(p // 2, p // 2 + 1 + p % 2):
                for j in range(q // 2, q // 2 + 1 + q % 2):
                    rf = 1 if (p == 1 and i == 1) else 2 if (p == h - 2 and i == zh - 2) else 0
                    cf = 1 if (q == 1 and j == 1) else 2 if (q == wd - 2 and j == zw - 2) else 0
                    if not rf and not cf:
                        srcs.append(y[:, i, j])
                    elif not cf:
                        srcs.append(ytb[(rf - 1) * n:rf * n, 0, j])
                    elif not rf:
                        srcs.append(ylr[(cf - 1) * n:cf * n, i, 0])
                    else:
                        k = (rf - 1) * 2 + (cf - 1)
                        srcs.append(ycn[k * n:(k + 1) * n, 0, 0])
            z[:, p, q] = np.mean(np.asarray(srcs, np.float64), axis=0)
    conv = O.conv2d(x, w, 1, 1, 'SAME', np.float64) + bias + z
    act = O.relu(conv)
    if dt == 'bf16':
        act = gu.bf16_round(act.astype(np.float32)).astype(np.float64)
    ref = (act.reshape(-1, cout) @ w_head.astype(np.float64) + b_head).reshape(n, h, wd, ncls)
    out = gu.conv2d_dual_head(x, w, None, None, bias, w_head, b_head, 1, (0, 0), (h, wd),
                              zoom_residual=(y, ytb, ylr, ycn), dt=dt)
    if dt == 'bf16':
        _close(out, ref, 2e-2, 'head conv with zoomed residual')
        assert np.mean(np.abs(out - ref) > 5e-3) < 0.01
    else:
        _close(out, ref, 2e-4, '%s head conv with zoomed residual' % dt)


def test_entry_points_reject_bad_arguments():
    """Every entry point validates its arguments on the host and returns a non-zero
    status with a message (no launch, no fault): null pointers, packed-weight dimensions
    that do not match the descriptor, unsupported channel counts, unaligned pointers."""
    gu, L = _gu(), _lib()
    x = torch.zeros((1, 8, 8, 64), dtype=torch.bfloat16, device='cuda')
    w = torch.zeros((256, 576 + 64), dtype=torch.bfloat16, device='cuda')
    y = torch.zeros((1, 8, 8, 64), dtype=torch.bfloat16, device='cuda')
    b = torch.zeros(64, dtype=torch.float32, device='cuda')

    def desc(**kw):
        d = dict(n=1, h=8, w=8, cin=64, cout=64, kh=3, kw=3, stride=1, dilation=1, pad_top=1, pad_left=1,
                 ho=8, wo=8, act=0, in_dtype=L.BF16, out_dtype=L.BF16, y_ld=64, res_ld=0, k_pad=576,
                 cout_pad=256, config=0)
        d.update(kw)
        return L.ConvDesc(**d)

    def rejected(name, *args):
        rc = getattr(L.lib, name)(*args)
        msg = L.lib.fss_last_error().decode()
        assert rc != 0 and msg, (name, rc, msg)
        return msg

    good = desc()
    assert L.lib.fss_conv2d_nhwc(ctypes.byref(good), x.data_ptr(), w.data_ptr(), b.data_ptr(), None,
                                 y.data_ptr(), None) == 0
    assert 'null' in rejected('fss_conv2d_nhwc', ctypes.byref(good), None, w.data_ptr(), b.data_ptr(), None,
                              y.data_ptr(), None)
    assert 'mismatch' in rejected('fss_conv2d_nhwc', ctypes.byref(desc(k_pad=512)), x.data_ptr(), w.data_ptr(),
                                  b.data_ptr(), None, y.data_ptr(), None)
    assert 'config' in rejected('fss_conv2d_nhwc', ctypes.byref(desc(config=999)), x.data_ptr(), w.data_ptr(),
                                b.data_ptr(), None, y.data_ptr(), None)
    assert 'unaligned' in rejected('fss_conv2d_nhwc', ctypes.byref(good), x.data_ptr() + 2, w.data_ptr(),
                                   b.data_ptr(), None, y.data_ptr(), None)
    assert 'y_ld' in rejected('fss_conv2d_nhwc', ctypes.byref(desc(y_ld=32)), x.data_ptr(), w.data_ptr(),
                              b.data_ptr(), None, y.data_ptr(), None)
    # two-input form: x2 and cin2 go together, channel counts in multiples of 64
    assert 'together' in rejected('fss_conv2d_dual_nhwc', ctypes.byref(desc(k_pad=640)), x.data_ptr(), None, 64,
                                  w.data_ptr(), b.data_ptr(), None, y.data_ptr(), None)
    assert 'multiples of 64' in rejected('fss_conv2d_dual_nhwc', ctypes.byref(desc(k_pad=640)), x.data_ptr(),
                                         x.data_ptr(), 32, w.data_ptr(), b.data_ptr(), None, y.data_ptr(), None)
    # classifier-head form: bfloat16, at most 20 classes, a 16x16-MFMA configuration
    yh = torch.zeros((1, 8, 8, 20), dtype=torch.float32, device='cuda')
    head = [x.data_ptr(), None, 0, w.data_ptr(), b.data_ptr(), None, None, 0, 0, w.data_ptr(), b.data_ptr(),
            19, yh.data_ptr(), 20, None]
    assert L.lib.fss_conv2d_dual_head_nhwc(ctypes.byref(good), *head) == 0
    bad = list(head); bad[11] = 21
    assert 'cout_head' in rejected('fss_conv2d_dual_head_nhwc', ctypes.byref(good), *bad)
    bad = list(head); bad[13] = 16
    assert 'y_head_ld' in rejected('fss_conv2d_dual_head_nhwc', ctypes.byref(good), *bad)
    zp = (ctypes.c_void_p * 4)(*[x.data_ptr()] * 4)
    bad = list(head); bad[6] = zp; bad[7] = 4; bad[8] = 4          # 8x8 output is not 7x7
    assert 'zoom_residual' in rejected('fss_conv2d_dual_head_nhwc', ctypes.byref(good), *bad)
    bad = list(head); bad[9] = None
    assert 'null' in rejected('fss_conv2d_dual_head_nhwc', ctypes.byref(good), *bad)
    assert 'FSS_BF16' in rejected('fss_conv2d_dual_head_nhwc', ctypes.byref(desc(in_dtype=L.F32, out_dtype=L.F32)),
                              *head)
    assert '16x16x32' in rejected('fss_conv2d_dual_head_nhwc', ctypes.byref(desc(config=1)), *head)
    # fused stem pair: only bfloat16, 32 or 64 channels
    m3 = (ctypes.c_float * 3)(0, 0, 0)
    s3 = (ctypes.c_float * 3)(1, 1, 1)
    frame = torch.zeros((1, 16, 16, 3), dtype=torch.uint8, device='cuda')
    args = [frame.data_ptr(), w.data_ptr(), b.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), 1, 16, 16, 1,
            16, 16, m3, s3, 64, 1, 1, 8, 8, 0, 64, 1, 1, 4, 4, 0, L.BF16, 64, None]
    bad = list(args); bad[14] = 48
    assert 'c1' in rejected('fss_stem2_conv_u8', *bad)
    bad = list(args); bad[26] = L.F32
    assert 'bfloat16' in rejected('fss_stem2_conv_u8', *bad)
    # depthwise / argmax / confusion
    assert 'null' in rejected('fss_upsample_argmax', None, y.data_ptr(), None, 1, 8, 8, 19, 20, 29, 29, None)
    assert 'class count' in rejected('fss_upsample_argmax', x.data_ptr(), y.data_ptr(), None, 1, 8, 8, 300, 300,
                                     29, 29, None)
    assert 'multiple of 16 bytes' in rejected('fss_dwconv2d_nhwc', x.data_ptr(), b.data_ptr(), b.data_ptr(),
                                              y.data_ptr(), 1, 8, 8, 12, 3, 3, 1, 1, 1, 1, 8, 8, 0, L.BF16, None)
    # the two-layer 1x1 launch: x [64 px, 64], w [256][64], y [64 px, 256], w2 [64][256], y2 [64 px, 64]
    y1 = torch.zeros((64, 256), dtype=torch.bfloat16, device='cuda')

    def chain(**kw):
        a = dict(x=x.data_ptr(), cin=64, x_ld=64, x2=None, cin2=0, x2_ld=0, w=w.data_ptr(), bias=None, res=None,
                 res_ld=0, act1=1, y=y1.data_ptr(), y_ld=256, cout1=256, w2=w.data_ptr(), bias2=None, act2=1,
                 y2=y.data_ptr(), y2_ld=64, cout2=64, pixels=64, dtype=L.BF16)
        a.update(kw)
        return rejected('fss_conv1x1_chain_nhwc', *a.values(), None)
    assert 'null' in chain(w2=None)
    assert 'bfloat16' in chain(dtype=L.F32)
    assert 'cin' in chain(cin=48, x_ld=48)
    assert 'cin2 = 0 without' in chain(cin2=64)
    assert '32, 64 or 128' in chain(cin=64, x2=x.data_ptr(), cin2=32, x2_ld=32)
    assert 'cout1' in chain(cout1=192)
    assert 'cout2' in chain(cout2=128, y2_ld=128)
    assert 'pixel strides' in chain(y_ld=250)
    assert 'pixel strides' in chain(res=y1.data_ptr(), res_ld=128)
    assert 'unaligned' in chain(y=y1.data_ptr() + 2)
    assert 'bad act' in chain(act2=7)
    assert 'bad shape' in chain(pixels=0)
    # the LDS-resident 3x3 convolution of 64 channels

    def c3(**kw):
        a = dict(x=x.data_ptr(), x_ld=64, w=w.data_ptr(), bias=None, y=y1.data_ptr(), y_ld=64, n=1, h=8, wd=8, ho=8,
                 wo=8, cout=64, stride=1, dilation=1, pad_top=1, pad_left=1, act=1, dtype=L.BF16)
        a.update(kw)
        return rejected('fss_conv3x3_c64_nhwc', *a.values(), None)
    assert 'null' in c3(w=None)
    assert 'bfloat16' in c3(dtype=L.F32)
    assert 'cout' in c3(cout=96, y_ld=96)
    assert 'stride' in c3(stride=5)
    assert 'output size' in c3(ho=12)
    assert 'pixel strides' in c3(x_ld=60)
    assert 'unaligned' in c3(x=x.data_ptr() + 2)
    assert 'bad act' in c3(act=9)
    # the 128 -> 256 channel 3x3 convolution with its input tile in LDS

    def c128(**kw):
        a = dict(x=x.data_ptr(), x_ld=128, w=w.data_ptr(), bias=None, y=y1.data_ptr(), y_ld=256, n=1, h=4, wd=8,
                 act=1, dtype=L.F16)
        a.update(kw)
        return rejected('fss_conv3x3_c128_nhwc', *a.values(), None)
    assert 'null' in c128(w=None)
    assert 'bfloat16 and half' in c128(dtype=L.F32)
    assert 'pixel strides' in c128(x_ld=64)
    assert 'pixel strides' in c128(y_ld=250)
    assert 'unaligned' in c128(y=y1.data_ptr() + 2)
    assert 'bad act' in c128(act=9)
    assert 'bad shape' in c128(h=0)
    L.call('fss_device_sync')


STEM_CASES = [
    # frame hw, padded hw, sub, k, stride, rate, cout, act, y_ld
    ((37, 70), (41, 75), 1, 3, 2, 1, 32, 'relu', None),      # ICNet 'Conv' (full-resolution branch)
    ((37, 70), (41, 75), 2, 7, 2, 1, 64, 'relu', None),      # resnet conv1 behind the 0.5 resize
    ((40, 64), (41, 65), 1, 7, 2, 1, 64, 'relu', None),      # PSPNet: conv1 on the padded frame
    ((23, 45), (23, 45), 1, 3, 1, 2, 24, 'relu6', 32),       # ragged cout, dilation, channel slice
    ((9, 11), (9, 11), 1, 1, 1, 1, 5, None, None),           # 1x1, tiny
]


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
@pytest.mark.parametrize('case', STEM_CASES)
def test_stem_conv_u8_matches_preprocess_resize_conv(case, dt):
    """fss_stem_conv_u8 == pad_to_bounding_box -> (x - mean) * scale -> [align_corners
    half-size resize] -> conv (explicit symmetric padding as conv2d_same / SAME) ->
    bias -> activation of the oracle, built from the separate oracle ops."""
    gu = _gu()
    (hr, wr), (hp, wp), sub, k, stride, rate, cout, act, y_ld = case
    if dt == 'f32x3' and cout % 8:
        pytest.skip('split-bf16 tensors hold channels in groups of 8')
    rng = np.random.default_rng(zlib.crc32(repr(case).encode()))
    frames = rng.integers(0, 256, (2, hr, wr, 3)).astype(np.uint8)
    mean, scale = (123.68, 116.779, 103.939), (1.0, 0.5, 2.0)
    w = rng.normal(0, 0.02, (k, k, 3, cout)).astype(np.float32)
    bias = rng.normal(0, 1, cout).astype(np.float32)
    img = O.pad_to_bounding_box(frames.astype(np.float32), hp, wp)
    img = ((img - np.asarray(mean, np.float32)) * np.asarray(scale, np.float32)).astype(np.float32)
    if sub == 2:
        hl, wl = (hp - 1) // 2 + 1, (wp - 1) // 2 + 1
        half = O.resize_bilinear(img, hl, wl, align_corners=True)
        assert np.array_equal(half, img[:, ::2, ::2, :])     # the resize is a pure subsampling
        img = half
    hl, wl = img.shape[1:3]
    keff = k + (k - 1) * (rate - 1)
    if stride > 1 and k > 1:     # conv2d_same: explicit symmetric-ish padding, then VALID
        pt = (keff - 1) // 2
        pb = keff - 1 - pt
        pads = (pt, pb, pt, pb)
        ho, wo = (hl + keff - 1 - keff) // stride + 1, (wl + keff - 1 - keff) // stride + 1
    else:
        ho, pt, pb = O.same_padding(hl, k, stride, rate)
        wo, pl, pr = O.same_padding(wl, k, stride, rate)
        pads = (pt, pb, pl, pr)
    xin, win = (gu.bf16_round(img), gu.bf16_round(w)) if dt == 'bf16' else (img, w)
    ref = O.conv2d(xin, win, stride, rate, pads, np.float64) + bias
    ref = {'relu': O.relu, 'relu6': O.relu6, None: lambda v: v}[act](ref)
    assert ref.shape[1:3] == (ho, wo)
    got = gu.stem_conv(frames, w, bias, mean, scale, sub, (hl, wl), stride, rate, (pads[0], pads[2]),
                       (ho, wo), act, dt, y_ld=y_ld)
    # f32x3: raw bytes against two-f16 weights, nothing rounded but the split-bfloat16 store
    _close(got, ref, {'bf16': 1.0 / 128, 'f32x3': 2e-5}.get(dt, 2e-4), 'stem conv %s' % dt)


STEM2_CASES = [
    # frame hw, padded hw, sub, c1, c2, act1, act2, y_ld
    ((37, 70), (41, 75), 1, 64, 64, 'relu6', 'relu6', None),    # ICNet-1.0 'Conv' + 'Conv_1'
    ((150, 300), (161, 321), 1, 64, 64, 'relu6', 'relu6', None),  # interior tiles (no masks) too
    ((64, 128), (65, 129), 1, 32, 32, 'relu6', 'relu6', None),  # ICNet-0.5 widths, odd sizes
    ((40, 66), (40, 66), 1, 32, 64, 'relu', None, 72),          # even sizes (asymmetric SAME pad), slice
    ((45, 77), (47, 80), 2, 64, 32, None, 'relu', None),        # behind a half-size resize
    ((5, 6), (5, 6), 1, 64, 64, 'relu', 'relu', None),          # smaller than one tile
]


@pytest.mark.parametrize('case', STEM2_CASES)
def test_stem2_conv_u8_matches_two_layers(case):
    """fss_stem2_conv_u8 (two 3x3 stride-2 SAME convolutions fused, the first output kept
    in LDS) against the oracle ops chained with the intermediate rounded to bfloat16
    (tolerance 1/128 of the output scale: one bf16 rounding), and against the unfused
    kernels fss_stem_conv_u8 -> fss_conv2d_nhwc: the fused kernel adds the products of the
    first layer in another order and its bias inside the MFMA, so a few intermediate values
    round to the neighbouring bfloat16; at most 1 % of the outputs may differ, each by no
    more than one bf16 step (1/128 of the scale)."""
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
    y1 = fn[act1](O.conv2d(gu.bf16_round(img), gu.bf16_round(w1), 2, 1, (pt1, pb1, pl1, pr1), np.float64) + b1)
    y1 = gu.bf16_round(y1.astype(np.float32))
    ref = fn[act2](O.conv2d(y1, gu.bf16_round(w2), 2, 1, (pt2, pb2, pl2, pr2), np.float64) + b2)
    assert ref.shape[1:3] == (h2, w2o)
    got = gu.stem2_conv(frames, w1, b1, w2, b2, mean, scale, sub, (hl, wl), (pt1, pl1), (h1, w1o), act1,
                        (pt2, pl2), (h2, w2o), act2, y_ld=y_ld)
    _close(got, ref, 1.0 / 128, 'stem2 %s' % (case,))
    # the two stand-alone kernels (16x16 MFMA shape, as the plan runs them in bfloat16)
    mid = gu.stem_conv(frames, w1, b1, mean, scale, sub, (hl, wl), 2, 1, (pt1, pl1), (h1, w1o), act1, 'bf16')
    two = gu.conv2d(mid, w2, b2, None, 2, 1, (pt2, pl2), (h2, w2o), act2, 'bf16', config=20)
    _close(got, two, 1.0 / 128, 'stem2 vs the two stand-alone kernels')
    assert np.mean(got != two) <= 0.01, 'too many outputs differ from the two stand-alone kernels'


AGREE_SHAPES = [
    # n, h, w, cin, cout, k, stride, rate, residual
    (2, 33, 65, 128, 256, 3, 1, 4, False),     # dilated, partial M tiles, taps skipped per block
    (2, 33, 65, 64, 512, 3, 1, 16, False),     # d=16: most blocks skip taps, different per tile size
    (1, 65, 129, 64, 256, 1, 1, 1, True),      # pointwise + residual
    (3, 40, 52, 32, 72, 3, 2, 1, False),       # strided, cout not a multiple of the tiles
    (1, 29, 31, 256, 264, 1, 1, 1, True),      # cout a multiple of 8 only: widest vector path edge
    (1, 65, 129, 192, 320, 3, 1, 2, True),     # 33 x 2 tiles of 256x256, an odd number of K tiles (27)
    (1, 33, 65, 2048, 512, 1, 1, 1, False),    # long pointwise K (32 K tiles), two N tiles
]


@pytest.mark.parametrize('shape', AGREE_SHAPES)
def test_conv2d_bf16_configs_agree_bitwise_on_odd_shapes(shape):
    """The plan's tuner may pick any configuration of one MFMA shape class for a layer:
    they must all give the same bits, also when tiles are partial, when blocks of
    different sizes skip different padding taps, and with a residual."""
    gu, L = _gu(), _lib()
    n, h, wd, cin, cout, k, stride, rate, has_res = shape
    rng = np.random.default_rng(zlib.crc32(repr(shape).encode()))
    x = gu.bf16_round(rng.normal(0, 1, (n, h, wd, cin)).astype(np.float32))
    w = gu.bf16_round(rng.normal(0, 0.05, (k, k, cin, cout)).astype(np.float32))
    bias = rng.normal(0, 1, cout).astype(np.float32)
    ho, pt, _ = O.same_padding(h, k, stride, rate)
    wo, pl, _ = O.same_padding(wd, k, stride, rate)
    res = gu.bf16_round(rng.normal(0, 1, (n, ho, wo, cout)).astype(np.float32)) if has_res else None
    ref = O.conv2d(x, w, stride, rate, 'SAME', np.float64) + bias + (res if has_res else 0.0)
    ref = O.relu(ref)
    by_shape = {}
    for cfg in range(1, L.lib.fss_conv_config_count() + 1):
        out = gu.conv2d(x, w, bias, res, stride, rate, (pt, pl), (ho, wo), 'relu', 'bf16', config=cfg)
        _close(out, ref, 1.0 / 128, 'conv config %d' % cfg)
        info = (ctypes.c_int * 7)()
        L.call('fss_conv_config_info', cfg, info)
        first = by_shape.setdefault(info[6], out)
        assert np.array_equal(out, first), 'config %d differs inside its shape class' % cfg


def test_conv2d_rejects_bad_arguments():
    L = _lib()
    d = L.ConvDesc(n=1, h=8, w=8, cin=3, cout=8, kh=3, kw=3, stride=1, dilation=1, pad_top=1,
                   pad_left=1, ho=8, wo=8, act=0, in_dtype=L.F32, out_dtype=L.F32, y_ld=8,
                   res_ld=0, k_pad=64, cout_pad=128)
    buf = torch.zeros(4096, device='cuda')
    rc = L.lib.fss_conv2d_nhwc(ctypes.byref(d), buf.data_ptr(), buf.data_ptr(), None, None,
                               buf.data_ptr(), None)
    assert rc != 0 and b'cin' in L.lib.fss_last_error()


def _call(name, *args):
    _lib().call(name, *args)
    _lib().call('fss_device_sync')


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_depthwise(dt):
    gu, L = _gu(), _lib()
    rng = np.random.default_rng(3)
    # (the large cases run the register-blocked 3x3 kernel, the small ones the generic kernel)
    for (n, h, wd, c, stride, rate) in [(8, 129, 257, 96, 1, 4), (12, 200, 258, 64, 2, 1),
                                        (8, 130, 259, 144, 1, 2), (8, 120, 163, 80, 1, 1),
                                        (1, 33, 65, 96, 1, 4), (2, 20, 26, 24, 2, 1),
                                        (1, 17, 33, 144, 1, 2), (1, 9, 11, 32, 2, 1),
                                        (2, 12, 19, 40, 1, 1),       # plain 3x3, ragged run at the row end
                                        (1, 15, 22, 16, 1, 3)]:      # a dilation only the generic kernel has
        x = rng.normal(0, 1, (n, h, wd, c)).astype(np.float32)
        w = rng.normal(0, 0.4, (3, 3, c, 1)).astype(np.float32)
        bias = rng.normal(0, 0.5, c).astype(np.float32)
        if dt != 'f32':
            x = gu.bf16_round(x) if dt == 'bf16' else gu.x3_round(x)
        ref = O.relu6(O.depthwise_conv2d(x, w, stride, rate, np.float64) + bias)
        ho, pt, _ = O.same_padding(h, 3, stride, rate)
        wo, pl, _ = O.same_padding(wd, 3, stride, rate)
        xd = gu.to_dev(x, dt)
        wdv = torch.from_numpy(w[:, :, :, 0].reshape(9, c).copy()).cuda()
        bd = torch.from_numpy(bias).cuda()
        y = torch.empty((n, ho, wo, c), dtype=gu.DT[dt][1], device='cuda')
        _call('fss_dwconv2d_nhwc', xd.data_ptr(), wdv.data_ptr(), bd.data_ptr(), y.data_ptr(),
              n, h, wd, c, 3, 3, stride, rate, pt, pl, ho, wo, L.ACT_RELU6, gu.DT[dt][0], None)
        _close(gu.to_host(y, dt), ref, {'f32': 2e-4, 'f32x3': 2e-5}.get(dt, 1.0 / 128), 'depthwise')


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_pools_and_subsample(dt):
    gu = _gu()
    rng = np.random.default_rng(4)
    code, tdt, epc = gu.DT[dt]
    for (n, h, wd, c) in [(1, 37, 41, 64), (2, 40, 52, 16)]:
        x = rng.normal(0, 1, (n, h, wd, c)).astype(np.float32)
        if dt != 'f32':
            x = gu.bf16_round(x) if dt == 'bf16' else gu.x3_round(x)
        xd = gu.to_dev(x, dt)
        # max pool 3x3/2 SAME: exact (selection only)
        ref = O.max_pool_same(x, 3, 2)
        ho, pt, _ = O.same_padding(h, 3, 2)
        wo, pl, _ = O.same_padding(wd, 3, 2)
        y = torch.empty((n, ho, wo, c), dtype=tdt, device='cuda')
        _call('fss_maxpool_nhwc', xd.data_ptr(), y.data_ptr(), n, h, wd, c, 3, 2, pt, pl, ho, wo,
              code, None)
        assert np.array_equal(gu.to_host(y, dt), ref)
        # subsample: exact
        ref = O.subsample(x, 2)
        y = torch.empty(ref.shape, dtype=tdt, device='cuda')
        _call('fss_subsample_nhwc', xd.data_ptr(), y.data_ptr(), n, h, wd, c, 2, ref.shape[1],
              ref.shape[2], code, None)
        assert np.array_equal(gu.to_host(y, dt), ref)
        # avg pool bins, including dropped remainder rows / columns
        for f in (1, 2, 3, 4, 6):
            kh, kw = int(h / f), int(wd / f)
            ref = O.avg_pool_valid(x.astype(np.float64), kh, kw)
            y = torch.empty(ref.shape, dtype=tdt, device='cuda')
            _call('fss_avgpool_bins_nhwc', xd.data_ptr(), y.data_ptr(), n, h, wd, c, kh, kw,
                  ref.shape[1], ref.shape[2], code, None)
            _close(gu.to_host(y, dt), ref, {'f32': 1e-5, 'f32x3': 2e-5}.get(dt, 1.0 / 128), 'avgpool f=%d' % f)
        # all levels of a pyramid pooling module in one pass (dropped remainders included)
        for factors in ((1, 2, 3, 4), (1, 2, 3, 6), (2,)):
            levels = [(int(h / f), int(wd / f)) for f in factors]
            got = gu.avgpool_pyramid(x, levels, dt)
            for (kh, kw), g in zip(levels, got):
                ref = O.avg_pool_valid(x.astype(np.float64), kh, kw)
                assert g.shape == ref.shape
                _close(g, ref, {'f32': 1e-5, 'f32x3': 2e-5}.get(dt, 1.0 / 128), 'pyramid avgpool %dx%d' % (kh, kw))


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_avgpool_pyramid_large_batch_path(dt):
    """fss_avgpool_pyramid_nhwc switches its first stage with the problem size (column pieces
    for small batches, whole rows for large ones).  The small form is checked against the
    oracle above; here a batch large enough for the row form must agree with the per-level
    kernel fss_avgpool_bins_nhwc (itself checked against the oracle)."""
    gu, L = _gu(), _lib()
    code, tdt, epc = gu.DT[dt]
    n, h, wd, c = 40, 33, 65, 2048
    assert n * h * (c // epc) >= 262144
    g = torch.Generator(device='cuda').manual_seed(1)
    xd = torch.randn((n, h, wd, c), device='cuda', generator=g).to(tdt)
    if dt == 'f32x3':
        # a valid split-bfloat16 tensor made on the device: bf16 high parts, zero remainders
        hi = xd.to(torch.bfloat16).view(n, h, wd, c // 8, 1, 8)
        xd = torch.cat([hi, torch.zeros_like(hi)], dim=4).contiguous().view(torch.float32).view(n, h, wd, c)
    levels = [(int(h / f), int(wd / f)) for f in (1, 2, 3, 4)]
    Lv = len(levels)
    outs = [torch.full((n, h // kh, wd // kw, c), -77.0, dtype=tdt, device='cuda') for kh, kw in levels]
    ptrs = (ctypes.c_void_p * Lv)(*[o.data_ptr() for o in outs])
    khs = (ctypes.c_int * Lv)(*[l[0] for l in levels])
    kws = (ctypes.c_int * Lv)(*[l[1] for l in levels])
    bhs = (ctypes.c_int * Lv)(*[h // l[0] for l in levels])
    bws = (ctypes.c_int * Lv)(*[wd // l[1] for l in levels])
    work = torch.empty((n * h * sum(bws) * c,), dtype=torch.float32, device='cuda')
    _call('fss_avgpool_pyramid_nhwc', xd.data_ptr(), ptrs, khs, kws, bhs, bws, Lv, work.data_ptr(),
          work.numel() * 4, n, h, wd, c, code, None)
    for (kh, kw), o in zip(levels, outs):
        ref = torch.empty_like(o)
        _call('fss_avgpool_bins_nhwc', xd.data_ptr(), ref.data_ptr(), n, h, wd, c, kh, kw, h // kh, wd // kw,
              code, None)
        _close(gu.to_host(o, dt), gu.to_host(ref, dt), {'f32': 1e-5, 'f32x3': 2e-5}.get(dt, 1.0 / 128), 'pyramid rows %dx%d' % (kh, kw))


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_resize_bilinear(dt):
    gu = _gu()
    rng = np.random.default_rng(5)
    code, tdt, epc = gu.DT[dt]
    cases = [((1, 33, 65, 16), (65, 129), True), ((2, 17, 21, 8), (9, 11), True),
             ((1, 3, 3, 32), (33, 65), True), ((1, 1, 1, 8), (9, 17), True),
             ((1, 6, 6, 8), (90, 90), True), ((1, 10, 12, 8), (23, 31), False),
             ((1, 1025 // 8, 2049 // 8, 8), (512 // 8 + 1, 129), True)]
    for shape, (ho, wo), align in cases:
        x = rng.normal(0, 1, shape).astype(np.float32)
        if dt != 'f32':
            x = gu.bf16_round(x) if dt == 'bf16' else gu.x3_round(x)
        ref = O.resize_bilinear(x, ho, wo, align)
        xd = gu.to_dev(x, dt)
        n, h, wd, c = shape
        y = torch.empty((n, ho, wo, c), dtype=tdt, device='cuda')
        _call('fss_resize_bilinear_nhwc', xd.data_ptr(), y.data_ptr(), n, h, wd, c, ho, wo, c,
              1 if align else 0, code, None)
        # same float32 formula as the oracle: exact for f32 up to fma contraction
        _close(gu.to_host(y, dt), ref, {'f32': 2e-6, 'f32x3': 2e-5}.get(dt, 1.0 / 128), 'resize %s' % (shape,))
    # float32 logits with 19 channels (scalar path) into a strided destination
    x = rng.normal(0, 1, (1, 9, 11, 19)).astype(np.float32)
    ref = O.resize_bilinear(x, 17, 21, True)
    xd = gu.to_dev(x, 'f32')
    y = torch.zeros((1, 17, 21, 24), dtype=torch.float32, device='cuda')
    _call('fss_resize_bilinear_nhwc', xd.data_ptr(), y.data_ptr(), 1, 9, 11, 19, 17, 21, 24, 1,
          gu.DT['f32'][0], None)
    got = gu.to_host(y)
    _close(got[..., :19], ref, 2e-6, 'resize c=19')
    assert np.all(got[..., 19:] == 0)


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_pyramid_add(dt):
    gu = _gu()
    rng = np.random.default_rng(6)
    code, tdt, epc = gu.DT[dt]
    n, h, wd, c = 2, 33, 65, 64
    x = rng.normal(0, 1, (n, h, wd, c)).astype(np.float32)
    if dt != 'f32':
        x = gu.bf16_round(x) if dt == 'bf16' else gu.x3_round(x)
    pooled = []
    ref = x.astype(np.float32)
    for f in (1, 2, 3, 4):
        p = O.avg_pool_valid(x, int(h / f), int(wd / f)).astype(np.float32)
        if dt != 'f32':
            p = gu.bf16_round(p) if dt == 'bf16' else gu.x3_round(p)
        pooled.append(p)
        ref = ref + O.resize_bilinear(p, h, wd, True)
    xd = gu.to_dev(x, dt)
    pd = [gu.to_dev(p, dt) for p in pooled]
    y = torch.empty((n, h, wd, c), dtype=tdt, device='cuda')
    ptrs = (ctypes.c_void_p * 4)(*[p.data_ptr() for p in pd])
    bh = (ctypes.c_int * 4)(*[p.shape[1] for p in pooled])
    bw = (ctypes.c_int * 4)(*[p.shape[2] for p in pooled])
    _call('fss_pyramid_add_nhwc', xd.data_ptr(), ptrs, bh, bw, 4, y.data_ptr(), n, h, wd, c, code,
          None)
    _close(gu.to_host(y, dt), ref, {'f32': 2e-6, 'f32x3': 2e-5}.get(dt, 1.0 / 128), 'pyramid')
    # without a base tensor: the sum of the resized maps alone
    y.fill_(-5.0)
    _call('fss_pyramid_add_nhwc', None, ptrs, bh, bw, 4, y.data_ptr(), n, h, wd, c, code, None)
    _close(gu.to_host(y, dt), ref - x, {'f32': 2e-6, 'f32x3': 2e-5}.get(dt, 1.0 / 128), 'pyramid sum without a base')


@pytest.mark.parametrize('dt', ['f32', 'bf16', 'f32x3'])
def test_add_copy_preprocess(dt):
    gu, L = _gu(), _lib()
    rng = np.random.default_rng(8)
    code, tdt, epc = gu.DT[dt]
    a = rng.normal(0, 3, (2, 5, 7, 16)).astype(np.float32)
    b = rng.normal(0, 3, (2, 5, 7, 16)).astype(np.float32)
    if dt != 'f32':
        a, b = (gu.bf16_round(a), gu.bf16_round(b)) if dt == 'bf16' else (gu.x3_round(a), gu.x3_round(b))
    ad, bd = gu.to_dev(a, dt), gu.to_dev(b, dt)
    y = torch.empty_like(ad)
    _call('fss_add_act', ad.data_ptr(), bd.data_ptr(), y.data_ptr(), a.size, L.ACT_RELU, code, None)
    _close(gu.to_host(y, dt), O.relu(a + b), {'f32': 1e-6, 'f32x3': 2e-5}.get(dt, 1.0 / 128), 'add')
    # channel copy into a concat slice
    big = torch.zeros((2, 5, 7, 48), dtype=tdt, device='cuda')
    esz = big.element_size()
    _call('fss_copy_channels', ad.data_ptr(), 16, big.data_ptr() + 16 * esz, 48, 2 * 5 * 7, 16, code,
          None)
    got = gu.to_host(big, dt)
    assert np.array_equal(got[..., 16:32], a) and np.all(got[..., :16] == 0) and np.all(got[..., 32:] == 0)
    # preprocess: zero pad BEFORE mean subtraction
    img = rng.integers(0, 256, (2, 10, 12, 3)).astype(np.uint8)
    mean = np.array([123.68, 116.779, 103.939], np.float32)
    scale = np.array([1.0, 1.0, 1.0], np.float32)
    ref = O.pad_to_bounding_box(img.astype(np.float32), 11, 13) - mean
    xd = torch.from_numpy(img).cuda()
    y = torch.empty((2, 11, 13, epc), dtype=tdt, device='cuda')
    _call('fss_preprocess_u8', xd.data_ptr(), y.data_ptr(), 2, 10, 12, 11, 13, epc,
          mean.ctypes.data_as(L.c_float_p), scale.ctypes.data_as(L.c_float_p), code, None)
    got = gu.to_host(y, dt)
    _close(got[..., :3], ref, {'f32': 1e-6, 'f32x3': 2e-5}.get(dt, 1.0 / 128), 'preprocess')
    assert np.all(got[..., 3:] == 0)


def test_upsample_argmax_lut_confusion():
    gu, L = _gu(), _lib()
    rng = np.random.default_rng(9)
    n, h, wd, c = 2, 17, 33, 19
    x = rng.normal(0, 1, (n, h, wd, c)).astype(np.float32)
    x[0, 3, 4, :] = 1.0             # exact ties: lowest index must win
    x[1, :, :, 5] = x[1, :, :, 2]   # tie between classes 2 and 5 everywhere
    ho, wo = 65, 129
    up = O.resize_bilinear(x, ho, wo, True)
    ref = O.argmax_channels(up)
    xd = torch.from_numpy(x).cuda()
    lab = torch.empty((n, ho, wo), dtype=torch.uint8, device='cuda')
    full = torch.empty((n, ho, wo, c), dtype=torch.float32, device='cuda')
    _call('fss_upsample_argmax', xd.data_ptr(), lab.data_ptr(), full.data_ptr(), n, h, wd, c, c, ho,
          wo, None)
    got_full = full.cpu().numpy()
    _close(got_full, up, 2e-6, 'upsampled logits')
    got = lab.cpu().numpy().astype(np.int64)
    # labels must be the exact argmax of the logits the kernel itself produced ...
    assert np.array_equal(got, np.argmax(got_full, axis=3))
    # ... and agree with the oracle wherever the oracle's top-2 margin exceeds rounding noise
    s = np.sort(up, axis=3)
    clear = (s[..., -1] - s[..., -2]) > 1e-5
    assert np.array_equal(got[clear], ref[clear])
    assert got[0, 0, 0] == ref[0, 0, 0]
    # same-size call = plain argmax (tie rows included): bit-exact
    lab2 = torch.empty((n, h, wd), dtype=torch.uint8, device='cuda')
    _call('fss_upsample_argmax', xd.data_ptr(), lab2.data_ptr(), None, n, h, wd, c, c, h, wd, None)
    assert np.array_equal(lab2.cpu().numpy().astype(np.int64), O.argmax_channels(x))
    # power-of-two zoom fast path (pixel stride 20 floats): source coordinates are exact,
    # so labels are bit-exact against the oracle's argmax wherever its float32 logits
    # have no exact tie broken differently by fma contraction
    xp = np.zeros((n, h, wd, 20), np.float32)
    xp[..., :c] = x
    xp[..., c] = 1e9               # the pad channel must never win
    xpd = torch.from_numpy(xp).cuda()
    for z in (2, 4, 8):
        hz, wz = (h - 1) * z + 1, (wd - 1) * z + 1
        upz = O.resize_bilinear(x, hz, wz, True)
        labz = torch.full((n, hz, wz), 255, dtype=torch.uint8, device='cuda')
        _call('fss_upsample_argmax', xpd.data_ptr(), labz.data_ptr(), None, n, h, wd, c, 20, hz, wz,
              None)
        gz = labz.cpu().numpy().astype(np.int64)
        sz = np.sort(upz, axis=3)
        clearz = (sz[..., -1] - sz[..., -2]) > 1e-5
        assert gz.max() < c
        assert np.array_equal(gz[clearz], O.argmax_channels(upz)[clearz])
        assert clearz.mean() > 0.4
    # smooth logits (large regions with one winning class): most cells take the kernel's
    # shortcut "four corners agree => the whole cell has that label"; the result must still
    # be the oracle's argmax of the interpolated logits
    hs, ws = 19, 45
    coarse = rng.normal(0, 1, (n, 3, 4, c)).astype(np.float32)
    xs = O.resize_bilinear(coarse, hs, ws, True) * 4 + rng.normal(0, 0.01, (n, hs, ws, c)).astype(np.float32)
    xs = xs.astype(np.float32)
    xsp = np.zeros((n, hs, ws, 20), np.float32)
    xsp[..., :c] = xs
    xsp[..., c] = 1e9
    xsd = torch.from_numpy(xsp).cuda()
    for z in (2, 4, 8):
        hz, wz = (hs - 1) * z + 1, (ws - 1) * z + 1
        upz = O.resize_bilinear(xs, hz, wz, True)
        labz = torch.full((n, hz, wz), 255, dtype=torch.uint8, device='cuda')
        _call('fss_upsample_argmax', xsd.data_ptr(), labz.data_ptr(), None, n, hs, ws, c, 20, hz, wz, None)
        gz = labz.cpu().numpy().astype(np.int64)
        sz = np.sort(upz, axis=3)
        clearz = (sz[..., -1] - sz[..., -2]) > 1e-5
        assert np.array_equal(gz[clearz], O.argmax_channels(upz)[clearz])
        assert clearz.mean() > 0.95
        am = O.argmax_channels(xs)
        uniform = (am[:, :-1, :-1] == am[:, 1:, :-1]) & (am[:, :-1, :-1] == am[:, :-1, 1:]) & \
                  (am[:, :-1, :-1] == am[:, 1:, 1:])
        assert 0.5 < uniform.mean() < 1.0        # both kinds of cells are present
    # colour LUT with crop
    from libs.constants import CITYSCAPES_LABEL_COLORS, CITYSCAPES_LABEL_IDS
    for table in (CITYSCAPES_LABEL_COLORS, CITYSCAPES_LABEL_IDS):
        t = np.asarray(table, np.float32)
        td = torch.from_numpy(t).cuda()
        out = torch.empty((n, ho - 1, wo - 1, t.shape[1]), dtype=torch.float32, device='cuda')
        _call('fss_label_lut', lab.data_ptr(), out.data_ptr(), td.data_ptr(), n, ho, wo, ho - 1,
              wo - 1, t.shape[1], t.shape[0], None)
        refc = O.map_to_colored_labels(got, table)[:, :ho - 1, :wo - 1]
        assert np.array_equal(out.cpu().numpy(), refc)
    # confusion matrix: bit-exact, accumulates across calls, ignore label dropped
    labels = rng.integers(0, 19, (n, ho, wo)).astype(np.uint8)
    labels[rng.random(labels.shape) < 0.05] = 255
    labels[0, :8] = 3               # long uniform runs
    ld = torch.from_numpy(labels).cuda()
    cm = torch.zeros((19, 19), dtype=torch.int64, device='cuda')
    for _ in range(2):
        _call('fss_confusion_accumulate', lab.data_ptr(), ld.data_ptr(), labels.size, 19, 255,
              cm.data_ptr(), None)
    refcm = OM.confusion_matrix(got, labels, 19, 255)
    assert np.array_equal(cm.cpu().numpy(), 2 * refcm)
    # empty input is a no-op; all-ignore adds nothing
    _call('fss_confusion_accumulate', lab.data_ptr(), ld.data_ptr(), 0, 19, 255, cm.data_ptr(), None)
    allign = torch.full_like(ld, 255)
    _call('fss_confusion_accumulate', lab.data_ptr(), allign.data_ptr(), labels.size, 19, 255,
          cm.data_ptr(), None)
    assert np.array_equal(cm.cpu().numpy(), 2 * refcm)
