"""Writes the golden fixtures tests/golden/*.npz.

A SECOND implementation of the reference's inference graphs, independent of oracle/ and of the
product: torch CPU ops in float64 (F.conv2d, F.max_pool2d, F.avg_pool2d, F.interpolate), written
from the reference's graph-building code, which each function cites.  Run in the development
container only (it is not imported by any test); the tests compare both the numpy oracle and the
HIP path against the arrays it writes.  Nothing in the reference pins numbers (it has no tests
and no fixtures), so these fixtures remove the single-author risk of the oracle but not the
"parity unpinned" status.

Weights come from fss.synth.make_weights(variables, seed=1234), a deterministic function of the
variable names; only inputs and expected outputs are stored (each file < 1 MB).

    python tests/golden/make_golden.py
"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd')]

RESNET_MEANS = (123.68, 116.779, 103.939)      # extractors/pspnet_icnet_resnet_v1.py:20
EPS = 1e-5                                     # hyperparams / resnet_arg_scope batch norm epsilon


class Net(object):
    """Holds the weights (name -> array) and records which variables were used."""

    def __init__(self, weights):
        self.w = weights
        self.used = set()

    def get(self, name):
        self.used.add(name)
        return torch.from_numpy(np.asarray(self.w[name], np.float64))

    # ---- TF / slim op semantics on NCHW float64 tensors
    @staticmethod
    def _same_pads(size, k, stride, rate):
        out = -(-size // stride)
        keff = k + (k - 1) * (rate - 1)
        total = max((out - 1) * stride + keff - size, 0)
        return total // 2, total - total // 2            # the extra row / column goes last

    def conv(self, x, scope, stride=1, rate=1, explicit=False, bn=True, act='relu'):
        """slim.conv2d [+ batch_norm] [+ activation].  explicit: resnet_utils.conv2d_same
        (third_party/resnet_utils.py:77-122): SAME at stride 1, else pad k_eff - 1 and VALID."""
        w = self.get(scope + '/weights')                  # [kh, kw, cin, cout]
        k = w.shape[0]
        if explicit and stride > 1:
            total = k + (k - 1) * (rate - 1) - 1
            pt = pl = total // 2
            pb = pr = total - total // 2
        else:
            pt, pb = self._same_pads(x.shape[2], k, stride, rate)
            pl, pr = self._same_pads(x.shape[3], k, stride, rate)
        x = F.pad(x, (pl, pr, pt, pb))
        y = F.conv2d(x, w.permute(3, 2, 0, 1).contiguous(), stride=stride, dilation=rate)
        if bn:
            y = self.bn(y, scope)
        else:
            y = y + self.get(scope + '/biases').view(1, -1, 1, 1)
        return self.act(y, act)

    def depthwise(self, x, scope, stride, rate):
        """slim.separable_conv2d(num_outputs=None) + batch norm + relu6
        (third_party/conv_blocks.py:240-249)."""
        w = self.get(scope + '/depthwise_weights')        # [3, 3, c, 1]
        c = w.shape[2]
        pt, pb = self._same_pads(x.shape[2], 3, stride, rate)
        pl, pr = self._same_pads(x.shape[3], 3, stride, rate)
        x = F.pad(x, (pl, pr, pt, pb))
        y = F.conv2d(x, w.permute(2, 3, 0, 1).contiguous(), stride=stride, dilation=rate, groups=c)
        return self.act(self.bn(y, scope), 'relu6')

    def bn(self, y, scope):
        g, b = self.get(scope + '/BatchNorm/gamma'), self.get(scope + '/BatchNorm/beta')
        m, v = self.get(scope + '/BatchNorm/moving_mean'), self.get(scope + '/BatchNorm/moving_variance')
        s = g / torch.sqrt(v + EPS)
        return y * s.view(1, -1, 1, 1) + (b - m * s).view(1, -1, 1, 1)

    @staticmethod
    def act(y, kind):
        if kind == 'relu':
            return torch.clamp(y, min=0)
        if kind == 'relu6':
            return torch.clamp(y, 0, 6)
        return y

    @staticmethod
    def resize(x, h, w):
        """tf.image.resize_bilinear(align_corners=True)."""
        if (h, w) == tuple(x.shape[2:]):
            return x
        return F.interpolate(x, size=(h, w), mode='bilinear', align_corners=True)

    @staticmethod
    def interp(x, s=1.0, z=1.0):
        """_dynamic_interpolation (architectures/icnet_architecture.py:255-266)."""
        h, w = x.shape[2], x.shape[3]
        sh, sw = (h - 1) * s + 1, (w - 1) * s + 1
        return Net.resize(x, int(sh + (sh - 1) * (z - 1)), int(sw + (sw - 1) * (z - 1)))

    @staticmethod
    def max_pool_3x3_s2_same(x):
        pt, pb = Net._same_pads(x.shape[2], 3, 2, 1)
        pl, pr = Net._same_pads(x.shape[3], 3, 2, 1)
        return F.max_pool2d(F.pad(x, (pl, pr, pt, pb), value=-math.inf), 3, 2)


# ---------------------------------------------------------------- dilated ResNet-50 v1
def bottleneck(net, x, scope, depth, depth_bottleneck, stride, rate):
    """third_party/dilated_resnet_v1.py:50-113; rate = rate * overide_rate."""
    if depth == x.shape[1]:
        shortcut = x[:, :, ::stride, ::stride]            # resnet_utils.subsample = 1x1 max pool
    else:
        shortcut = net.conv(x, scope + '/shortcut', stride, act=None)
    r = net.conv(x, scope + '/conv1')
    r = net.conv(r, scope + '/conv2', stride, rate, explicit=True)
    r = net.conv(r, scope + '/conv3', act=None)
    return torch.clamp(shortcut + r, min=0)


def dilated_resnet50(net, x, fs, mid_downsample, taps):
    """third_party/dilated_resnet_v1.py:151-231 with output_stride 8, through
    resnet_utils.stack_blocks_dense (third_party/resnet_utils.py:126-219)."""
    root = 'SharedFeatureExtractor/resnet_v1_50'
    x = net.conv(x, root + '/conv1', 2, explicit=True)
    x = Net.max_pool_3x3_s2_same(x)
    blocks = [('block1', 64 // fs, 3, 2, 1)]
    if mid_downsample:
        blocks.append(('downsample_block', None, 0, 1, 1))
    blocks += [('block2', 128 // fs, 4, 2, 1), ('block3', 256 // fs, 6, 2, 2), ('block4', 512 // fs, 3, 1, 4)]
    current_stride, rate, target = 1, 1, 8 // 4
    ends = {}
    for name, base, units, block_stride, overide in blocks:
        if name == 'downsample_block':
            x = Net.interp(x, s=0.5)
            continue
        for u in range(units):
            stride = block_stride if u == units - 1 else 1
            scope = '%s/%s/unit_%d/bottleneck_v1' % (root, name, u + 1)
            if current_stride == target:
                x = bottleneck(net, x, scope, base * 4, base, 1, rate * overide)
                rate *= stride
            else:
                x = bottleneck(net, x, scope, base * 4, base, stride, 1 * overide)
                current_stride *= stride
        ends[name] = x
    return ends['block1'], ends['block4'], ends['block2']


# ---------------------------------------------------------------------- MobileNetV2
def make_divisible(v, divisor, min_value=None):
    """third_party/mobilenet.py:62-69."""
    min_value = divisor if min_value is None else min_value
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def mobilenet_v2(net, x, fs, mid_downsample, taps):
    """third_party/mobilenet_v2.py:55-98 run by mobilenet_base (third_party/mobilenet.py:
    148-294) with depth multiplier 1, min_depth = divisible_by = 8, output_stride 8, up to
    layer_18; expanded_conv = third_party/conv_blocks.py:162-314."""
    root = 'SharedFeatureExtractor/MobilenetV2'
    spec = [('conv', 2, 32), ('expanded', 1, 16)]
    spec += [('expanded', s, d / fs) for s, d in
             [(2, 24), (1, 24), (2, 32), (1, 32), (1, 32), (2, 64), (1, 64), (1, 64), (1, 64),
              (1, 96), (1, 96), (1, 96), (2, 160), (1, 160), (1, 160), (1, 320)]]
    if mid_downsample:
        spec.insert(5, ('downsample', 1, None))
    current_stride, rate, n_exp, ends = 1, 1, 0, {}
    for i, (kind, stride, depth) in enumerate(spec):
        if current_stride == 8:
            layer_stride, layer_rate = 1, rate
            rate *= stride
        else:
            layer_stride, layer_rate = stride, 1
            current_stride *= stride
        if kind == 'conv':
            x = net.conv(x, root + '/Conv', layer_stride, layer_rate, act='relu6')
        elif kind == 'downsample':
            x = Net.interp(x, s=0.5)
        else:
            scope = root + ('/expanded_conv' if n_exp == 0 else '/expanded_conv_%d' % n_exp)
            cin = x.shape[1]
            inner = make_divisible(cin * 1, 1) if n_exp == 0 else make_divisible(cin * 6, 8)
            n_exp += 1
            y = x
            if inner > cin:
                y = net.conv(y, scope + '/expand', act='relu6')
            y = net.depthwise(y, scope + '/depthwise', layer_stride, layer_rate)
            y = net.conv(y, scope + '/project', act=None)
            cout = make_divisible(depth * 1.0, 8, 8)
            assert y.shape[1] == cout, (scope, y.shape, cout)
            x = y + x if (layer_stride == 1 and cout == cin) else y
        ends['layer_%d' % (i + 1)] = x
        if i + 1 == 18:
            break
    return ends['layer_5'], ends['layer_18'], ends['layer_8']


# ----------------------------------------------------------------------------- heads
def preprocess(img_u8, pad_hw, extractor):
    """libs/exporter.py:52-53,37 (zero padding of the uint8 frame, then to float) and the
    extractor's preprocess (mean subtraction / scaling to [-1, 1])."""
    h, w = pad_hw
    x = np.zeros((1, h, w, 3), np.float64)
    x[0, :img_u8.shape[0], :img_u8.shape[1]] = img_u8
    if extractor == 'dilated_resnet50':
        x = x - np.asarray(RESNET_MEANS)
    else:
        x = (x - 127.5) / 127.5
    return torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()


def icnet(net, x, extractor, fs, pooling_factors, taps):
    """architectures/icnet_architecture.py:110-174 (is_training=False)."""
    backbone = dilated_resnet50 if extractor == 'dilated_resnet50' else mobilenet_v2
    half, quarter, _ = backbone(net, Net.interp(x, s=0.5), fs, True, taps)
    taps['half_res'], taps['quarter_res'] = half, quarter
    h, w = quarter.shape[2], quarter.shape[3]
    added = quarter                                       # _icnet_pspmodule :176-211
    for f in pooling_factors:
        k = (int(h / f), int(w / f))
        added = added + Net.resize(F.avg_pool2d(quarter, k, k), h, w)
    pooled = net.conv(added, 'FastPSPModule/Conv', act='relu6')
    taps['psp_module'] = pooled
    b = net.conv(x, 'Conv', 2, act='relu6')               # _third_feature_branch :213-226
    b = net.conv(b, 'Conv_1', 2, act='relu6')
    b = net.conv(b, 'Conv_2', 2, act='relu6')
    full = net.conv(b, 'Conv_3', 1, act='relu6')

    def cff(first, second, scope):                        # _cascade_feature_fusion :228-253
        d = net.conv(Net.interp(first, z=2.0), scope + '/DilatedConv', 1, 2, act=None)
        c = net.conv(second, scope + '/Conv', act=None)
        return torch.clamp(c + Net.resize(d, c.shape[2], c.shape[3]), min=0)

    f1 = cff(pooled, half, 'CascadeFeatureFusion')
    f2 = cff(f1, full, 'CascadeFeatureFusion_1')
    taps['first_fusion'], taps['second_fusion'] = f1, f2
    low = net.conv(Net.interp(f2, z=2.0), 'Predictions/Conv', bn=False, act=None)
    taps['logits_lowres'] = low
    return Net.interp(low, z=4.0)


def pspnet(net, x, extractor, fs, pooling_factors, taps):
    """architectures/pspnet_architecture.py:88-147 (is_training=False)."""
    backbone = dilated_resnet50 if extractor == 'dilated_resnet50' else mobilenet_v2
    _, feats, _ = backbone(net, x, fs, False, taps)
    taps['backbone'] = feats
    h, w = feats.shape[2], feats.shape[3]
    branches = [feats]
    for i, f in enumerate(pooling_factors):
        k = (int(h / f), int(w / f))
        p = net.conv(F.avg_pool2d(feats, k, k), 'PSPModule/Conv' + ('_%d' % i if i else ''), act='relu6')
        branches.append(Net.resize(p, h, w))
    merged = net.conv(torch.cat(branches, 1), 'PSPModule/Conv_%d' % len(pooling_factors), act='relu6')
    taps['psp_module'] = merged
    low = net.conv(merged, 'Predictions/Conv', bn=False, act=None)
    taps['logits_lowres'] = low
    return Net.interp(low, z=8.0)


def variables_of(arch, ext, fs, pf, in_shape, pad):
    """Variable names / shapes of the model (from the product's graph builder: shapes only)."""
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import graph as G
    cfg = pipeline_pb2.PipelineConfig()
    text_format.Merge("""model { %s { num_classes: 19 filter_scale: %s pooling_factors: %s
      feature_extractor { type: '%s' }
      hyperparams { batch_norm { train: true scale: true center: true decay: 0.9997 epsilon: 0.00001 }
                    regularizer { l2_regularizer { weight: 0.0001 } }
                    initializer { variance_scaling_initializer {} } }
      loss { classification_loss { softmax {} } use_auxiliary_loss: true } } }""" % (arch, fs, pf, ext), cfg)
    g = G.reset_default_graph()
    _, model = model_builder.build(cfg.model, is_training=False)
    model.predict(model.preprocess(G.to_float(G.placeholder((1,) + pad + (3,)))))
    return dict(g.variables)


CASES = [
    # file stem, architecture, extractor, filter_scale (config value), pooling factors, frame, padded size
    ('icnet_resnet50_0.5_129x257', 'icnet', 'dilated_resnet50', 0.5, [1.0, 2.0, 3.0, 4.0], (128, 256), (129, 257)),
    ('pspnet_resnet50_1.0_97x97', 'pspnet', 'dilated_resnet50', 1.0, [1.0, 2.0, 3.0, 6.0], (97, 97), (97, 97)),
    ('icnet_mobilenetv2_1.0_129x257', 'icnet', 'dilated_mobilenet', 1.0, [1.0, 2.0, 3.0, 4.0], (128, 256), (129, 257)),
]


def main():
    from fss import synth
    torch.set_num_threads(8)
    for stem, arch, ext, fs_cfg, pf, frame, pad in CASES:
        variables = variables_of(arch, ext, fs_cfg, pf, frame, pad)
        istd = 70.0 if ext == 'dilated_resnet50' else 0.55
        weights = synth.make_weights(variables, seed=1234, input_std=istd)
        img = np.random.default_rng(2024).integers(0, 256, frame + (3,)).astype(np.uint8)
        net = Net(weights)
        taps = {}
        fs = int(round(1.0 / fs_cfg))
        with torch.no_grad():
            x = preprocess(img, pad, ext)
            logits = (icnet if arch == 'icnet' else pspnet)(net, x, ext, fs, pf, taps)
        unused = sorted(set(weights) - net.used)
        assert not unused, 'variables the second implementation never read: %s' % unused[:5]
        out = {'image': img, 'pad': np.asarray(pad), 'filter_scale': np.asarray(fs_cfg),
               'pooling_factors': np.asarray(pf), 'seed': np.asarray(1234), 'input_std': np.asarray(istd),
               'labels': logits.argmax(1)[0].numpy().astype(np.uint8)}
        keep = ['half_res', 'quarter_res', 'psp_module', 'first_fusion', 'second_fusion', 'logits_lowres',
                'backbone']
        for k in keep:
            if k in taps:
                t = taps[k].permute(0, 2, 3, 1).numpy().astype(np.float32)
                if t.size * 4 > 200000:       # big taps: a strided sample of pixels keeps files small
                    step = int(math.ceil(math.sqrt(t.size * 4 / 200000.0)))
                    out['tap_stride/' + k] = np.asarray(step)
                    t = t[:, ::step, ::step]
                out['tap/' + k] = np.ascontiguousarray(t)
        path = os.path.join(HERE, stem + '.npz')
        np.savez_compressed(path, **out)
        print('%s: %d variables, logits scale %.3g, %.0f KB' % (
            stem, len(weights), float(logits.abs().max()), os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
    main()
