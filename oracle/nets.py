"""numpy restatement of the reference's inference graphs (ICNet, PSPNet).

Test infrastructure only (see oracle/__init__.py).  The functions below follow
the reference's graph-building code function by function and address weights
by the TensorFlow variable names that code would create, so a weight
dictionary keyed by checkpoint names drives them directly.

`Params` has two modes: with a weight dict it looks tensors up (and checks
their shapes); with `weights=None` it only records the shapes requested,
which is how the published parameter counts are reproduced.
"""
import numpy as np

from . import ops


class Params(object):
    def __init__(self, weights=None, dtype=np.float32):
        self.weights = weights
        self.dtype = dtype
        self.shapes = {}          # name -> shape, in creation order

    def get(self, name, shape):
        shape = tuple(int(s) for s in shape)
        if name in self.shapes:
            assert self.shapes[name] == shape, (name, self.shapes[name], shape)
        self.shapes[name] = shape
        if self.weights is None:
            return np.zeros(shape, dtype=self.dtype)
        arr = np.asarray(self.weights[name])
        assert tuple(arr.shape) == shape, (name, arr.shape, shape)
        return arr.astype(self.dtype)

    def conv_param_count(self):
        """Parameters in conv kernels only ('weights' / 'depthwise_weights')."""
        return sum(int(np.prod(s)) for n, s in self.shapes.items()
                   if n.endswith('weights'))


class Hyper(object):
    """What builders/hyperparams_builder.py:45-68 puts in the arg scope."""

    def __init__(self, bn_scale=True, bn_center=True, bn_epsilon=1e-5):
        self.bn_scale = bn_scale
        self.bn_center = bn_center
        self.bn_epsilon = bn_epsilon


_ACT = {None: lambda v: v, 'relu': ops.relu, 'relu6': ops.relu6}


def _bn(P, x, scope, hyper, scale=None):
    c = x.shape[3]
    use_scale = hyper.bn_scale if scale is None else scale
    gamma = P.get(scope + '/BatchNorm/gamma', (c,)) if use_scale else None
    beta = P.get(scope + '/BatchNorm/beta', (c,)) if hyper.bn_center else None
    mean = P.get(scope + '/BatchNorm/moving_mean', (c,))
    var = P.get(scope + '/BatchNorm/moving_variance', (c,))
    return ops.batch_norm(x, gamma, beta, mean, var, hyper.bn_epsilon)


def conv_bn_act(P, x, scope, cout, k, stride=1, rate=1, act='relu',
                hyper=None, padding='SAME', same=False, bn=True):
    """slim.conv2d with normalizer_fn=batch_norm (no bias) or, with bn=False,
    a plain biased conv (libs/compressible_ops.py:16-24)."""
    cin = x.shape[3]
    w = P.get(scope + '/weights', (k, k, cin, int(cout)))
    if same:
        y = ops.conv2d_same(x, w, stride, rate, dtype=P.dtype)
    else:
        y = ops.conv2d(x, w, stride, rate, padding, dtype=P.dtype)
    if bn:
        y = _bn(P, y, scope, hyper)
    else:
        y = y + P.get(scope + '/biases', (int(cout),))
    return _ACT[act](y)


# --------------------------------------------------------------------------
# dilated ResNet-50 v1  (third_party/dilated_resnet_v1.py, resnet_utils.py)
# --------------------------------------------------------------------------
def _bottleneck(P, x, scope, depth, depth_bottleneck, stride, overide_rate,
                rate, hyper):
    """third_party/dilated_resnet_v1.py:50-113."""
    s = scope + '/bottleneck_v1'
    if depth == x.shape[3]:
        shortcut = ops.subsample(x, stride)
    else:
        shortcut = conv_bn_act(P, x, s + '/shortcut', depth, 1, stride,
                               act=None, hyper=hyper)
    r = conv_bn_act(P, x, s + '/conv1', depth_bottleneck, 1, 1, hyper=hyper)
    r = conv_bn_act(P, r, s + '/conv2', depth_bottleneck, 3, stride,
                    rate=rate * overide_rate, hyper=hyper, same=True)
    r = conv_bn_act(P, r, s + '/conv3', depth, 1, 1, act=None, hyper=hyper)
    return ops.relu(shortcut + r)


def _resnet_block(scope, base_depth, num_units, stride, rate=1):
    """third_party/dilated_resnet_v1.py:116-141."""
    unit = dict(depth=int(base_depth * 4), depth_bottleneck=int(base_depth),
                overide_rate=rate)
    return (scope, 'bottleneck',
            [dict(unit, stride=1)] * (num_units - 1) + [dict(unit, stride=stride)])


def dilated_resnet_v1_50(P, x, filter_scale, mid_downsample, output_stride,
                         hyper, root='SharedFeatureExtractor/resnet_v1_50'):
    """third_party/dilated_resnet_v1.py:151-231 with global_pool=False,
    num_classes=None; returns the end-point dict keyed by block scope."""
    blocks = [_resnet_block('block1', 64 / filter_scale, 3, 2)]
    if mid_downsample:
        blocks.append(('downsample_block', 'downsample', [dict(s_factor=0.5)]))
    blocks += [_resnet_block('block2', 128 / filter_scale, 4, 2),
               _resnet_block('block3', 256 / filter_scale, 6, 2, rate=2),
               _resnet_block('block4', 512 / filter_scale, 3, 1, rate=4)]
    output_stride = output_stride / 4          # root block: dilated_resnet_v1.py:173
    net = conv_bn_act(P, x, root + '/conv1', 64 // filter_scale, 7, 2,
                      hyper=hyper, same=True)
    net = ops.max_pool_same(net, 3, 2)
    # stack_blocks_dense: third_party/resnet_utils.py:176-219
    end_points = {}
    current_stride, rate = 1, 1
    for scope, kind, units in blocks:
        for i, unit in enumerate(units):
            uscope = '%s/%s/unit_%d' % (root, scope, i + 1)
            ustride = unit.get('stride', 1)
            if current_stride == output_stride:
                args, urate = dict(unit, stride=1), rate
                rate *= ustride
            else:
                args, urate = unit, 1
                current_stride *= ustride
                assert current_stride <= output_stride
            if kind == 'downsample':     # dilated_resnet_v1.py:38-47
                h, w = net.shape[1], net.shape[2]
                net = ops.resize_bilinear(
                    net, int((h - 1) * args['s_factor'] + 1),
                    int((w - 1) * args['s_factor'] + 1), True)
            else:
                net = _bottleneck(P, net, uscope, rate=urate, hyper=hyper, **args)
        end_points[root + '/' + scope] = net
    assert current_stride == output_stride
    return net, end_points


def resnet_extractor(P, x, filter_scale, mid_downsample, hyper):
    """extractors/pspnet_icnet_resnet_v1.py:47-68 (resnet_arg_scope forces
    BN scale=True, epsilon=1e-5 and relu)."""
    h = Hyper(True, hyper.bn_center, 1e-5)
    _, ep = dilated_resnet_v1_50(P, x, filter_scale, mid_downsample, 8, h)
    root = 'SharedFeatureExtractor/resnet_v1_50'
    return ep[root + '/block1'], ep[root + '/block4'], ep[root + '/block2']


RESNET_MEANS = (123.68, 116.779, 103.939)


# --------------------------------------------------------------------------
# MobileNetV2 (third_party/mobilenet_v2.py, mobilenet.py, conv_blocks.py)
# --------------------------------------------------------------------------
def _make_divisible(v, divisor, min_value=None):
    """third_party/mobilenet.py:62-69."""
    if min_value is None:
        min_value = divisor
    new_v = max(min_value, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def _expanded_conv(P, x, scope, num_outputs, stride, rate, expansion, hyper):
    """third_party/conv_blocks.py:162-314, depthwise_location='expansion'."""
    cin = x.shape[3]
    inner = expansion(cin)
    net = x
    if inner > cin:
        net = conv_bn_act(P, net, scope + '/expand', inner, 1, act='relu6',
                          hyper=hyper)
    c = net.shape[3]
    w = P.get(scope + '/depthwise/depthwise_weights', (3, 3, c, 1))
    net = ops.depthwise_conv2d(net, w, stride, rate, dtype=P.dtype)
    net = ops.relu6(_bn(P, net, scope + '/depthwise', hyper))
    net = conv_bn_act(P, net, scope + '/project', num_outputs, 1, act=None,
                      hyper=hyper)
    if stride == 1 and net.shape[3] == cin:
        net = net + x
    return net


def mobilenet_v2_base(P, x, filter_scale, mid_downsample, output_stride,
                      final_endpoint, hyper,
                      root='SharedFeatureExtractor/MobilenetV2'):
    """third_party/mobilenet_v2.py:55-98 + third_party/mobilenet.py:238-285
    with multiplier 1.0, min_depth = divisible_by = 8."""
    fs = filter_scale
    spec = [('conv', dict(stride=2, num_outputs=32)),
            ('expanded', dict(num_outputs=16, expand1=True))]
    for stride, depth in [(2, 24), (1, 24), (2, 32), (1, 32), (1, 32), (2, 64),
                          (1, 64), (1, 64), (1, 64), (1, 96), (1, 96), (1, 96),
                          (2, 160), (1, 160), (1, 160), (1, 320)]:
        spec.append(('expanded', dict(stride=stride, num_outputs=depth / fs)))
    spec.append(('conv1x1', dict(num_outputs=1280)))
    if mid_downsample:
        spec.insert(5, ('downsample', dict(s_factor=0.5)))
    # normalizer_params of the hyperparams arg scope reach every conv here
    # (builders/hyperparams_builder.py:59-66), so BN scale/center/epsilon
    # are the config's.
    h = hyper
    current_stride, rate = 1, 1
    net, end_points, n_expanded = x, {}, 0
    for i, (kind, params) in enumerate(spec):
        params = dict(params)
        if 'num_outputs' in params:
            params['num_outputs'] = _make_divisible(params['num_outputs'] * 1.0, 8, 8)
        stride = params.get('stride', 1)
        if current_stride == output_stride:
            layer_stride, layer_rate = 1, rate
            rate *= stride
        else:
            layer_stride, layer_rate = stride, 1
            current_stride *= stride
        if kind == 'conv':
            net = conv_bn_act(P, net, root + '/Conv', params['num_outputs'], 3,
                              layer_stride, layer_rate, act='relu6', hyper=h)
        elif kind == 'conv1x1':
            net = conv_bn_act(P, net, root + '/Conv_1', params['num_outputs'], 1,
                              act='relu6', hyper=h)
        elif kind == 'downsample':
            hh, ww = net.shape[1], net.shape[2]
            net = ops.resize_bilinear(net, int((hh - 1) * 0.5 + 1),
                                      int((ww - 1) * 0.5 + 1), True)
        else:
            scope = root + ('/expanded_conv' if n_expanded == 0
                            else '/expanded_conv_%d' % n_expanded)
            n_expanded += 1
            if params.get('expand1'):
                expansion = lambda c: _make_divisible(c * 1, 1)
            else:
                expansion = lambda c: _make_divisible(c * 6, 8)
            net = _expanded_conv(P, net, scope, params['num_outputs'],
                                 layer_stride, layer_rate, expansion, h)
        end_points['layer_%d' % (i + 1)] = net
        if final_endpoint == 'layer_%d' % (i + 1):
            break
    return net, end_points


def mobilenet_extractor(P, x, filter_scale, mid_downsample, hyper):
    """extractors/pspnet_icnet_mobilenet_v2.py:48-72."""
    _, ep = mobilenet_v2_base(P, x, filter_scale, mid_downsample, 8,
                              'layer_18', hyper)
    return ep['layer_5'], ep['layer_18'], ep['layer_8']


MOBILENET_MEANS = (127.5, 127.5, 127.5)


def preprocess(x, extractor):
    """extractors/pspnet_icnet_resnet_v1.py:43-45 /
    extractors/pspnet_icnet_mobilenet_v2.py:44-46."""
    if extractor == 'dilated_resnet50':
        return x - np.asarray(RESNET_MEANS, x.dtype)
    m = np.asarray(MOBILENET_MEANS, x.dtype)
    return (x - m) / m


_EXTRACTORS = {'dilated_resnet50': resnet_extractor,
               'dilated_mobilenet': mobilenet_extractor}


# --------------------------------------------------------------------------
# ICNet  (architectures/icnet_architecture.py)
# --------------------------------------------------------------------------
def _interp(x, s_factor=1.0, z_factor=1.0):
    return ops.resize_bilinear(x, ops.interp_size(x.shape[1], s_factor, z_factor),
                               ops.interp_size(x.shape[2], s_factor, z_factor), True)


def icnet_predict(P, preprocessed, num_classes=19, filter_scale=1.0,
                  pooling_factors=(1.0, 2.0, 3.0, 4.0),
                  extractor='dilated_resnet50', hyper=None,
                  pretrain_single_branch_mode=False, return_all=False):
    """architectures/icnet_architecture.py:110-174, is_training=False.
    `filter_scale` is the inverted value the builder passes (1 or 2)."""
    hyper = hyper or Hyper()
    fs = filter_scale
    x = preprocessed
    taps = {}
    ext_in = x if pretrain_single_branch_mode else _interp(x, s_factor=0.5)
    half, quarter, aux = _EXTRACTORS[extractor](
        P, ext_in, fs, not pretrain_single_branch_mode, hyper)
    taps.update(half_res=half, quarter_res=quarter, psp_aux=aux)
    # _icnet_pspmodule (:176-211)
    h, w = quarter.shape[1], quarter.shape[2]
    added = quarter
    for f in pooling_factors:
        pooled = ops.avg_pool_valid(quarter, int(h / f), int(w / f))
        pooled = ops.resize_bilinear(pooled, h, w, True)
        added = added + pooled
    pooled_quarter = conv_bn_act(P, added, 'FastPSPModule/Conv', 512 // fs, 1,
                                 act='relu6', hyper=hyper)
    taps['psp_module'] = pooled_quarter
    if not pretrain_single_branch_mode:
        # _third_feature_branch (:213-226)
        b = conv_bn_act(P, x, 'Conv', 64 // fs, 3, 2, act='relu6', hyper=hyper)
        b = conv_bn_act(P, b, 'Conv_1', 64 // fs, 3, 2, act='relu6', hyper=hyper)
        b = conv_bn_act(P, b, 'Conv_2', 128 // fs, 3, 2, act='relu6', hyper=hyper)
        full = conv_bn_act(P, b, 'Conv_3', 256 // fs, 3, 1, act='relu6', hyper=hyper)
        taps['full_res'] = full

        def cff(first, second, scope):       # :228-253
            up = _interp(first, z_factor=2.0)
            d = conv_bn_act(P, up, scope + '/DilatedConv', 256 // fs, 3, 1,
                            rate=2, act=None, hyper=hyper)
            c = conv_bn_act(P, second, scope + '/Conv', 256 // fs, 1, act=None,
                            hyper=hyper)
            # resize_bilinear(d, shape(c)) at :249 is a same-size resize
            assert d.shape == c.shape
            return ops.relu(c + d)

        f1 = cff(pooled_quarter, half, 'CascadeFeatureFusion')
        f2 = cff(f1, full, 'CascadeFeatureFusion_1')
        taps.update(first_fusion=f1, second_fusion=f2)
        final = _interp(f2, z_factor=2.0)
        zoom = 4
    else:
        final, zoom = pooled_quarter, 8
    logits = conv_bn_act(P, final, 'Predictions/Conv', num_classes, 1, act=None,
                         bn=False)
    taps['logits_lowres'] = logits
    logits = _interp(logits, z_factor=zoom)
    return (logits, taps) if return_all else logits


# --------------------------------------------------------------------------
# PSPNet  (architectures/pspnet_architecture.py)
# --------------------------------------------------------------------------
def pspnet_predict(P, preprocessed, num_classes=19, filter_scale=1.0,
                   pooling_factors=(1.0, 2.0, 3.0, 6.0),
                   extractor='dilated_resnet50', hyper=None, return_all=False):
    """architectures/pspnet_architecture.py:88-147, is_training=False."""
    hyper = hyper or Hyper()
    fs = filter_scale
    _, feats, aux = _EXTRACTORS[extractor](P, preprocessed, fs, False, hyper)
    taps = dict(backbone=feats, psp_aux=aux)
    h, w = feats.shape[1], feats.shape[2]
    branches = [feats]
    for i, f in enumerate(pooling_factors):
        pooled = ops.avg_pool_valid(feats, int(h / f), int(w / f))
        scope = 'PSPModule/Conv' if i == 0 else 'PSPModule/Conv_%d' % i
        pooled = conv_bn_act(P, pooled, scope, 512, 1, act='relu6', hyper=hyper)
        branches.append(ops.resize_bilinear(pooled, h, w, True))
    merged = np.concatenate(branches, axis=-1)
    n = len(pooling_factors)
    final = conv_bn_act(P, merged, 'PSPModule/Conv_%d' % n, 512 // fs, 3, 1,
                        act='relu6', hyper=hyper)
    taps['psp_module'] = final
    logits = conv_bn_act(P, final, 'Predictions/Conv', num_classes, 1, act=None,
                         bn=False)
    taps['logits_lowres'] = logits
    logits = _interp(logits, z_factor=8.0)
    return (logits, taps) if return_all else logits


# --------------------------------------------------------------------------
# inference / evaluation glue
# --------------------------------------------------------------------------
def inference_outputs(predict_fn, image_u8, input_shape, pad_to_shape,
                      label_color_map, extractor='dilated_resnet50'):
    """libs/exporter.py:32-80: uint8 HWC image -> float32 [1,h,w,C] colour map
    (or int64 [1,h,w,1] labels when label_color_map is None)."""
    x = np.asarray(image_u8)
    if x.ndim == 3:
        x = x[None]
    if pad_to_shape is not None:
        x = ops.pad_to_bounding_box(x, pad_to_shape[0], pad_to_shape[1])
    x = preprocess(x.astype(np.float32), extractor)
    labels = ops.argmax_channels(predict_fn(x))[..., None]
    out = labels
    if label_color_map is not None:
        out = ops.map_to_colored_labels(labels[..., 0], label_color_map)
    if pad_to_shape is not None:
        out = out[:, :input_shape[0], :input_shape[1]]
    return out
