"""The oracle against an independent second implementation (torch CPU ops) and
against the known answers the reference publishes or implies.

The reference ships no tests or golden vectors for this path, so these are the
pins the oracle has: published parameter counts (docs/model_zoo.md:26-28),
endpoint shapes, and op-level known answers for the TF semantics that decide
parity (SAME-padding asymmetry, conv2d_same, legacy bilinear sampling,
VALID avg-pool remainder, argmax ties, mIoU corner cases).
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import metrics as OM
from oracle import nets
from oracle import ops as O


def _t(x):      # NHWC numpy -> NCHW torch
    return torch.from_numpy(np.ascontiguousarray(x.transpose(0, 3, 1, 2)))


def _n(t):      # NCHW torch -> NHWC numpy
    return t.numpy().transpose(0, 2, 3, 1)


@pytest.mark.parametrize('h,w,k,stride,rate', [(13, 17, 3, 1, 1), (12, 16, 3, 2, 1), (13, 16, 3, 2, 1),
                                               (33, 65, 3, 1, 16), (9, 11, 3, 1, 4), (14, 14, 7, 2, 1),
                                               (10, 10, 1, 1, 1)])
def test_conv2d_same_padding_vs_torch(h, w, k, stride, rate):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(2, h, w, 5)).astype(np.float32)
    wt = rng.normal(size=(k, k, 5, 7)).astype(np.float32)
    got = O.conv2d(x, wt, stride, rate, 'SAME')
    # TF SAME: the extra pad element goes to the bottom / right
    ho, pt, pb = O.same_padding(h, k, stride, rate)
    wo, pl, pr = O.same_padding(w, k, stride, rate)
    xp = F.pad(_t(x), (pl, pr, pt, pb))
    ref = F.conv2d(xp, torch.from_numpy(wt.transpose(3, 2, 0, 1).copy()), stride=stride, dilation=rate)
    assert got.shape == (2, ho, wo, 7)
    np.testing.assert_allclose(got, _n(ref), rtol=1e-4, atol=1e-4)


def test_same_padding_asymmetry_known_answer():
    # even input, stride 2, k 3: TF pads 0 before and 1 after; conv2d_same pads 1 and 1
    assert O.same_padding(12, 3, 2) == (6, 0, 1)
    assert O.same_padding(13, 3, 2) == (7, 1, 1)
    assert O.fixed_padding(3) == (1, 1) and O.fixed_padding(7) == (3, 3)
    assert O.fixed_padding(3, rate=2) == (2, 2)
    x = np.arange(16, dtype=np.float32).reshape(1, 4, 4, 1)
    w = np.ones((3, 3, 1, 1), np.float32)
    same = O.conv2d(x, w, 2, 1, 'SAME')[0, :, :, 0]
    fixed = O.conv2d_same(x, w, 2)[0, :, :, 0]
    assert same.shape == (2, 2) and fixed.shape == (2, 2)
    assert same[0, 0] == x[0, 0:3, 0:3, 0].sum()          # window starts at pixel 0
    assert fixed[0, 0] == x[0, 0:2, 0:2, 0].sum()         # window starts at pixel -1


def test_conv2d_same_equals_stride1_then_subsample():
    """The identity the reference's conv2d_same docstring states
    (third_party/resnet_utils.py:83-90)."""
    rng = np.random.default_rng(1)
    for h, w in [(12, 16), (13, 17)]:
        x = rng.normal(size=(1, h, w, 4)).astype(np.float32)
        wt = rng.normal(size=(3, 3, 4, 6)).astype(np.float32)
        a = O.conv2d_same(x, wt, 2)
        b = O.subsample(O.conv2d(x, wt, 1, 1, 'SAME'), 2)
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-5)


def test_depthwise_vs_torch():
    rng = np.random.default_rng(2)
    for h, w, stride, rate in [(11, 13, 1, 1), (12, 14, 2, 1), (17, 9, 1, 4), (10, 10, 1, 2)]:
        x = rng.normal(size=(2, h, w, 6)).astype(np.float32)
        wt = rng.normal(size=(3, 3, 6, 1)).astype(np.float32)
        got = O.depthwise_conv2d(x, wt, stride, rate)
        ho, pt, pb = O.same_padding(h, 3, stride, rate)
        wo, pl, pr = O.same_padding(w, 3, stride, rate)
        ref = F.conv2d(F.pad(_t(x), (pl, pr, pt, pb)),
                       torch.from_numpy(wt.transpose(2, 3, 0, 1).copy()), stride=stride,
                       dilation=rate, groups=6)
        np.testing.assert_allclose(got, _n(ref), rtol=1e-4, atol=1e-5)


def test_pools_vs_torch():
    rng = np.random.default_rng(3)
    x = rng.normal(size=(2, 13, 16, 4)).astype(np.float32)
    got = O.max_pool_same(x, 3, 2)
    ho, pt, pb = O.same_padding(13, 3, 2)
    wo, pl, pr = O.same_padding(16, 3, 2)
    ref = F.max_pool2d(F.pad(_t(x), (pl, pr, pt, pb), value=float('-inf')), 3, 2)
    np.testing.assert_array_equal(got, _n(ref))
    # VALID avg pool drops the remainder: 13 rows / k=4 -> 3 bins, row 12 unused
    got = O.avg_pool_valid(x, 4, 5)
    ref = F.avg_pool2d(_t(x), (4, 5), (4, 5))
    assert got.shape == (2, 3, 3, 4)
    np.testing.assert_allclose(got, _n(ref), rtol=1e-5, atol=1e-6)
    x2 = x.copy()
    x2[:, 12:, :, :] = 1e6
    x2[:, :, 15:, :] = 1e6
    np.testing.assert_array_equal(O.avg_pool_valid(x2, 4, 5), got)
    np.testing.assert_array_equal(O.subsample(x, 2), x[:, ::2, ::2])


def test_resize_bilinear_align_corners_vs_torch():
    rng = np.random.default_rng(4)
    for (h, w), (ho, wo) in [((5, 7), (9, 13)), ((33, 65), (65, 129)), ((3, 3), (33, 65)),
                             ((6, 6), (90, 90)), ((17, 33), (9, 17))]:
        x = rng.normal(size=(2, h, w, 3)).astype(np.float32)
        got = O.resize_bilinear(x, ho, wo, True)
        ref = F.interpolate(_t(x), size=(ho, wo), mode='bilinear', align_corners=True)
        np.testing.assert_allclose(got, _n(ref), rtol=1e-4, atol=1e-5)


def test_resize_bilinear_known_answers():
    x = np.arange(2 * 3 * 5 * 1, dtype=np.float32).reshape(2, 3, 5, 1)
    # x2 zoom with align_corners: even outputs are input pixels, odd ones midpoints
    up = O.resize_bilinear(x, 5, 9, True)
    np.testing.assert_array_equal(up[:, ::2, ::2], x)
    np.testing.assert_array_equal(up[:, 1, 0, 0], (x[:, 0, 0, 0] + x[:, 1, 0, 0]) / 2)
    # x0.5 with (in-1) even is a pure stride-2 pick
    big = np.random.default_rng(5).normal(size=(1, 9, 13, 2)).astype(np.float32)
    np.testing.assert_array_equal(O.resize_bilinear(big, 5, 7, True), big[:, ::2, ::2])
    # 1x1 input broadcasts; same size is the identity
    one = np.full((1, 1, 1, 2), 3.5, np.float32)
    np.testing.assert_array_equal(O.resize_bilinear(one, 4, 6, True), np.full((1, 4, 6, 2), 3.5))
    np.testing.assert_array_equal(O.resize_bilinear(big, 9, 13, False), big)
    # legacy (not half-pixel) sampling when align_corners is False: src = dst * in/out
    y = O.resize_bilinear(x[:1, :, :1], 6, 1, False)[0, :, 0, 0]
    np.testing.assert_allclose(y, [0, 2.5, 5, 7.5, 10, 10])
    assert O.interp_size(1025, s_factor=0.5) == 513 and O.interp_size(129, z_factor=2.0) == 257
    assert O.interp_size(257, z_factor=4) == 1025 and O.interp_size(90, z_factor=8.0) == 713


def test_argmax_ties_and_glue():
    x = np.zeros((1, 1, 3, 4), np.float32)
    x[0, 0, 1] = [1, 5, 5, 2]
    x[0, 0, 2] = [7, 7, 7, 7]
    assert O.argmax_channels(x).tolist() == [[[0, 1, 0]]]
    assert O.argmax_channels(x).dtype == np.int64
    img = np.full((1, 2, 3, 3), 9, np.uint8)
    p = O.pad_to_bounding_box(img, 3, 4)
    assert p.shape == (1, 3, 4, 3) and p[0, 2].sum() == 0 and p[0, :, 3].sum() == 0
    # zero padding happens before mean subtraction: the padded border is -mean
    pre = nets.preprocess(p.astype(np.float32), 'dilated_resnet50')
    np.testing.assert_allclose(pre[0, 2, 3], -np.asarray(nets.RESNET_MEANS, np.float32))
    np.testing.assert_allclose(nets.preprocess(p.astype(np.float32), 'dilated_mobilenet')[0, 2, 3], -1.0)
    cm = O.map_to_colored_labels(np.array([[[0, 18]]]), [(i, i + 1, i + 2) for i in range(19)])
    assert cm.dtype == np.float32 and cm[0, 0].tolist() == [[0, 1, 2], [18, 19, 20]]


def test_mean_iou_known_answers():
    # all pixels ignored -> 0
    cm = OM.confusion_matrix(np.zeros(10), np.full(10, 255), 19)
    assert cm.sum() == 0 and OM.mean_iou(cm) == 0
    # perfect prediction on two present classes: absent classes are left out of the mean
    lab = np.array([0, 0, 1, 1, 255])
    cm = OM.confusion_matrix(lab.clip(0, 18), lab, 19)
    assert cm.sum() == 4 and OM.mean_iou(cm) == 1.0
    # one wrong pixel: IoU(0) = 2/3, IoU(1) = 1/2
    cm = OM.confusion_matrix(np.array([0, 0, 0, 1]), np.array([0, 0, 1, 1]), 19)
    np.testing.assert_allclose(OM.mean_iou(cm), np.float32((2 / 3 + 1 / 2) / 2), rtol=1e-6)
    assert cm[1, 0] == 1 and cm[0, 0] == 2        # rows = label, columns = prediction
    # a class that is only ever (wrongly) predicted still counts with IoU 0
    cm = OM.confusion_matrix(np.array([2, 0]), np.array([0, 0]), 19)
    np.testing.assert_allclose(OM.mean_iou(cm), np.float32(0.25))
    # zero-padded label pixels are class 0 with weight 1 (they are not ignore_label)
    assert OM.confusion_matrix(np.array([3]), np.array([0]), 19)[0, 3] == 1


@pytest.mark.parametrize('name,fn,fs,published', [
    ('icnet-1.0', nets.icnet_predict, 1.0, 26.82), ('icnet-0.5', nets.icnet_predict, 2.0, 6.71),
    ('pspnet50', nets.pspnet_predict, 1.0, 46.53)])
def test_published_parameter_counts(name, fn, fs, published):
    """docs/model_zoo.md:26-28 of the reference."""
    P = nets.Params(None)
    out = fn(P, np.zeros((1, 257, 513, 3), np.float32), filter_scale=fs)
    assert out.shape == (1, 257, 513, 19)
    assert round(P.conv_param_count() / 1e6, 2) == published


def test_endpoint_shapes_and_dilations():
    P = nets.Params(None)
    x = np.zeros((1, 257, 513, 3), np.float32)
    _, taps = nets.icnet_predict(P, x, return_all=True)
    assert taps['half_res'].shape == (1, 17, 33, 256)
    assert taps['quarter_res'].shape == (1, 9, 17, 2048)
    assert taps['psp_aux'].shape == (1, 9, 17, 512)
    assert taps['full_res'].shape == (1, 33, 65, 256)
    assert taps['logits_lowres'].shape == (1, 65, 129, 19)
    _, taps = nets.icnet_predict(nets.Params(None), x, extractor='dilated_mobilenet', return_all=True)
    assert taps['half_res'].shape == (1, 17, 33, 32)
    assert taps['quarter_res'].shape == (1, 9, 17, 160)
    _, taps = nets.pspnet_predict(nets.Params(None), np.zeros((1, 193, 193, 3), np.float32),
                                  return_all=True)
    assert taps['backbone'].shape == (1, 25, 25, 2048)
