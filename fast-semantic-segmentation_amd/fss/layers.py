"""slim-style layer constructors on top of `fss.graph`.

Mirrors the parts of `tf.contrib.slim` the reference's model code calls
(`arg_scope`, `conv2d`, `separable_conv2d`, `batch_norm` parameters,
`max_pool2d`, `avg_pool2d`), with slim's defaults, variable names and
scope-uniquification rules, so the architecture files read like the
reference's and produce the same checkpoint variable names.
"""
import contextlib

from fss import graph as G

relu = 'relu'
relu6 = 'relu6'
batch_norm = 'batch_norm'

# slim.batch_norm defaults (TF 1.8 contrib/layers/python/layers/layers.py)
_BN_DEFAULTS = {'center': True, 'scale': False, 'epsilon': 0.001}
_CONV_DEFAULTS = {'stride': 1, 'padding': 'SAME', 'rate': 1, 'activation_fn': relu,
                  'normalizer_fn': None, 'normalizer_params': None}

_ARG_SCOPE = [{}]   # stack of {op_name: {kwarg: value}}

_OPS = ('conv2d', 'separable_conv2d', 'conv2d_transpose', 'batch_norm',
        'max_pool2d', 'avg_pool2d')


def _op_name(op):
    name = op if isinstance(op, str) else getattr(op, '__name__', None)
    if name not in _OPS:
        raise ValueError('arg_scope: unsupported op %r' % (op,))
    return name


@contextlib.contextmanager
def arg_scope(list_ops_or_scope, **kwargs):
    """`slim.arg_scope`: either re-enters a captured scope (a dict) or sets
    default kwargs for the listed ops; inner scopes override outer ones."""
    current = {k: dict(v) for k, v in _ARG_SCOPE[-1].items()}
    if isinstance(list_ops_or_scope, dict):
        if kwargs:
            raise ValueError('kwargs are not allowed when re-entering a scope')
        for op, vals in list_ops_or_scope.items():
            current.setdefault(op, {}).update(vals)
    else:
        for op in list_ops_or_scope:
            current.setdefault(_op_name(op), {}).update(kwargs)
    _ARG_SCOPE.append(current)
    try:
        yield {k: dict(v) for k, v in current.items()}
    finally:
        _ARG_SCOPE.pop()


def _resolve(op, explicit, defaults):
    out = dict(defaults)
    out.update(_ARG_SCOPE[-1].get(op, {}))
    out.update({k: v for k, v in explicit.items() if v is not _UNSET})
    return out


_UNSET = object()


def _same_out(in_size, stride):
    return -(-in_size // stride)


def _same_pads(in_size, k, stride, rate):
    k_eff = k + (k - 1) * (rate - 1)
    out = _same_out(in_size, stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    return total // 2, total - total // 2


def _pair(v):
    if isinstance(v, (list, tuple)):
        return int(v[0]), int(v[1])
    return int(v), int(v)


def _bn_variables(g, channels, params):
    """Creates the BatchNorm variables slim would and returns their names."""
    p = dict(_BN_DEFAULTS)
    p.update(_ARG_SCOPE[-1].get('batch_norm', {}))
    p.update(params or {})
    if p.get('is_training'):
        raise NotImplementedError('batch_norm(is_training=True) is not supported')
    with g.variable_scope('BatchNorm'):
        names = {
            'gamma': g.variable('gamma', (channels,)) if p['scale'] else None,
            'beta': g.variable('beta', (channels,)) if p['center'] else None,
            'mean': g.variable('moving_mean', (channels,)),
            'var': g.variable('moving_variance', (channels,)),
            'epsilon': float(p['epsilon'])}
    return names


def conv2d(inputs, num_outputs, kernel_size, stride=_UNSET, padding=_UNSET, rate=_UNSET,
           activation_fn=_UNSET, normalizer_fn=_UNSET, normalizer_params=_UNSET,
           explicit_padding=None, scope=None, **unused):
    """slim.conv2d.  `explicit_padding=(top, bottom, left, right)` zero-pads
    before a VALID convolution (the tf.pad + VALID pair of conv2d_same)."""
    a = _resolve('conv2d', dict(stride=stride, padding=padding, rate=rate,
                                activation_fn=activation_fn, normalizer_fn=normalizer_fn,
                                normalizer_params=normalizer_params), _CONV_DEFAULTS)
    g = G.get_default_graph()
    kh, kw = _pair(kernel_size)
    s, r = int(a['stride']), int(a['rate'])
    if r > 1 and s > 1:
        raise ValueError('rate > 1 requires stride 1')
    n, h, w, cin = inputs.shape
    cout = int(num_outputs)
    if explicit_padding is not None:
        pt, pb, pl, pr = [int(v) for v in explicit_padding]
        keh, kew = kh + (kh - 1) * (r - 1), kw + (kw - 1) * (r - 1)
        ho, wo = (h + pt + pb - keh) // s + 1, (w + pl + pr - kew) // s + 1
    elif a['padding'] == 'SAME':
        (pt, pb), (pl, pr) = _same_pads(h, kh, s, r), _same_pads(w, kw, s, r)
        ho, wo = _same_out(h, s), _same_out(w, s)
    elif a['padding'] == 'VALID':
        pt = pb = pl = pr = 0
        keh, kew = kh + (kh - 1) * (r - 1), kw + (kw - 1) * (r - 1)
        ho, wo = (h - keh) // s + 1, (w - kew) // s + 1
    else:
        raise ValueError('unknown padding %r' % (a['padding'],))
    with g.variable_scope(scope, default_name='Conv'):
        weights = g.variable('weights', (kh, kw, cin, cout))
        bn, biases = None, None
        if a['normalizer_fn'] is not None:
            if a['normalizer_fn'] != batch_norm:
                raise ValueError('only batch_norm is supported as normalizer_fn')
            bn = _bn_variables(g, cout, a['normalizer_params'])
        else:
            biases = g.variable('biases', (cout,))
    return g.add('conv2d', [inputs], (n, ho, wo, cout), weights=weights, bn=bn, biases=biases,
                 kh=kh, kw=kw, stride=s, rate=r, pad_top=pt, pad_left=pl,
                 activation=a['activation_fn'])


def separable_conv2d(inputs, num_outputs, kernel_size, depth_multiplier=1, stride=_UNSET,
                     padding=_UNSET, rate=_UNSET, activation_fn=_UNSET, normalizer_fn=_UNSET,
                     normalizer_params=_UNSET, scope=None, **unused):
    """slim.separable_conv2d with num_outputs=None (depthwise only)."""
    if num_outputs is not None or depth_multiplier != 1:
        raise NotImplementedError('only depthwise (num_outputs=None, multiplier 1) is supported')
    a = _resolve('separable_conv2d', dict(stride=stride, padding=padding, rate=rate,
                                          activation_fn=activation_fn,
                                          normalizer_fn=normalizer_fn,
                                          normalizer_params=normalizer_params), _CONV_DEFAULTS)
    if a['padding'] != 'SAME':
        raise NotImplementedError('depthwise convolution supports SAME padding only')
    g = G.get_default_graph()
    kh, kw = _pair(kernel_size)
    s, r = int(a['stride']), int(a['rate'])
    n, h, w, c = inputs.shape
    (pt, _), (pl, _) = _same_pads(h, kh, s, r), _same_pads(w, kw, s, r)
    with g.variable_scope(scope, default_name='SeparableConv2d'):
        weights = g.variable('depthwise_weights', (kh, kw, c, 1))
        bn, biases = None, None
        if a['normalizer_fn'] is not None:
            bn = _bn_variables(g, c, a['normalizer_params'])
        else:
            biases = g.variable('biases', (c,))
    return g.add('depthwise', [inputs], (n, _same_out(h, s), _same_out(w, s), c),
                 weights=weights, bn=bn, biases=biases, kh=kh, kw=kw, stride=s, rate=r,
                 pad_top=pt, pad_left=pl, activation=a['activation_fn'])


def max_pool2d(inputs, kernel_size, stride=2, padding=_UNSET, scope=None):
    a = _resolve('max_pool2d', dict(padding=padding), {'padding': 'VALID'})
    kh, kw = _pair(kernel_size)
    if kh != kw:
        raise NotImplementedError('square max-pool windows only')
    n, h, w, c = inputs.shape
    s = int(stride)
    if a['padding'] == 'SAME':
        (pt, _), (pl, _) = _same_pads(h, kh, s, 1), _same_pads(w, kw, s, 1)
        ho, wo = _same_out(h, s), _same_out(w, s)
    else:
        pt = pl = 0
        ho, wo = (h - kh) // s + 1, (w - kw) // s + 1
    return G.get_default_graph().add('max_pool', [inputs], (n, ho, wo, c), k=kh, stride=s,
                                     pad_top=pt, pad_left=pl)


def avg_pool2d(inputs, kernel_size, stride, padding='VALID', scope=None):
    kh, kw = _pair(kernel_size)
    if _pair(stride) != (kh, kw) or padding != 'VALID':
        raise NotImplementedError('avg_pool2d supports VALID pooling with stride == kernel')
    n, h, w, c = inputs.shape
    if kh < 1 or kw < 1 or kh > h or kw > w:
        raise ValueError('avg_pool2d window %s does not fit input %s' % ((kh, kw), (h, w)))
    return G.get_default_graph().add('avg_pool', [inputs],
                                     (n, (h - kh) // kh + 1, (w - kw) // kw + 1, c),
                                     kh=kh, kw=kw)
