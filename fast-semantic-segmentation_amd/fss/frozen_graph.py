"""Weights from a TensorFlow frozen inference graph (`frozen_inference_graph.pb`).

The reference's export.py:82-137 freezes the inference graph with
freeze_graph_with_def_protos: every variable becomes a `Const` node that keeps the variable's
name (`.../weights`, `.../BatchNorm/gamma`, ...), and the Model Zoo archives ship that file
(docs/model_zoo.md:5-12, "with folded batch norm parameters").  Two forms are read:

  * plain: the constants carry the checkpoint's variable names - the result is the same
    name -> array dict a checkpoint gives;
  * batch norm folded: a convolution's `BatchNorm/*` constants are gone and a constant vector is
    added to the convolution's output (BiasAdd / Add, possibly behind the BatchToSpaceND of a
    dilated convolution).  That vector is returned as `<conv scope>/folded_bias`, and the plan
    compiler then uses the `weights` constant as it is (scale 1) with that bias.

A GraphDef is parsed with the minimal protobuf decoder of fss.checkpoint
(tensorflow/core/framework/{graph,node_def,attr_value,tensor}.proto field numbers).  No frozen
graph written by TensorFlow is available offline: the reader is pinned by the published message
layout and by this module's own writer only (format unpinned).
"""
import struct

import numpy as np

from fss.checkpoint import _DTYPES, _DTYPE_CODES, _parse_fields, _varint


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _parse_tensor(buf):
    dtype, dims, content, floats, ints = 1, [], None, [], []
    for f, wt, v in _parse_fields(buf):
        if f == 1:
            dtype = v
        elif f == 2:
            for f2, _, v2 in _parse_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _parse_fields(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >= (1 << 63) else v3
                    dims.append(size)
        elif f == 4:
            content = bytes(v)
        elif f == 5:                      # float_val: packed, or one fixed32 per entry
            if wt == 2:
                floats += list(np.frombuffer(bytes(v), '<f4'))
            else:
                floats.append(struct.unpack('<f', bytes(v))[0])
        elif f in (7, 10):                # int_val / int64_val
            if wt == 2:
                pos, b = 0, bytes(v)
                while pos < len(b):
                    x, shift = 0, 0
                    while True:
                        c = b[pos]
                        pos += 1
                        x |= (c & 0x7F) << shift
                        shift += 7
                        if not c & 0x80:
                            break
                    ints.append(x - (1 << 64) if x >= (1 << 63) else x)
            else:
                ints.append(v - (1 << 64) if v >= (1 << 63) else v)
    if dtype not in _DTYPES:
        return None
    np_dt = np.dtype(_DTYPES[dtype])
    count = int(np.prod(dims)) if dims else 1
    if content is not None:
        arr = np.frombuffer(content, dtype=np_dt.newbyteorder('<'))
    else:
        vals = floats if np_dt.kind == 'f' else ints
        if not vals:
            vals = [0]
        arr = np.asarray(vals, dtype=np_dt)
        if arr.size == 1 and count > 1:   # a single value stands for a constant-filled tensor
            arr = np.full(count, arr[0], dtype=np_dt)
    return arr.reshape(dims).astype(np_dt)


def parse_graph_def(data):
    """bytes of a GraphDef -> list of nodes: dict(name, op, inputs, tensor) with `tensor` the
    ndarray of a Const node's `value` attribute (None for other ops / unsupported dtypes)."""
    nodes = []
    for f, _, v in _parse_fields(memoryview(data)):
        if f != 1:
            continue
        node = dict(name='', op='', inputs=[], tensor=None)
        for f2, _, v2 in _parse_fields(v):
            if f2 == 1:
                node['name'] = bytes(v2).decode('utf-8')
            elif f2 == 2:
                node['op'] = bytes(v2).decode('utf-8')
            elif f2 == 3:
                node['inputs'].append(bytes(v2).decode('utf-8'))
            elif f2 == 5:
                key, val = None, None
                for f3, _, v3 in _parse_fields(v2):
                    if f3 == 1:
                        key = bytes(v3)
                    elif f3 == 2:
                        val = v3
                if key == b'value' and val is not None:
                    for f4, _, v4 in _parse_fields(val):
                        if f4 == 8:
                            node['tensor'] = _parse_tensor(v4)
        nodes.append(node)
    return nodes


def _base(name):
    name = name[1:] if name.startswith('^') else name
    return name.split(':')[0]


def load_variables(path, wanted=None):
    """{variable name: ndarray} from a frozen graph.  wanted: names or {name: shape} that must
    be resolvable - directly, or for `<scope>/BatchNorm/*` through a `<scope>/folded_bias`
    (returned in their place).  None = every float / integer constant named like a variable."""
    with open(path, 'rb') as f:
        nodes = parse_graph_def(f.read())
    by_name = {n['name']: n for n in nodes}
    consts = {n['name']: n['tensor'] for n in nodes if n['op'] == 'Const' and n['tensor'] is not None}

    def source_const(name):
        """The Const a tensor name resolves to through Identity nodes (the `/read` ops)."""
        seen = 0
        n = by_name.get(_base(name))
        while n is not None and n['op'] == 'Identity' and n['inputs'] and seen < 8:
            n = by_name.get(_base(n['inputs'][0]))
            seen += 1
        return n if n is not None and n['op'] == 'Const' else None

    consumers = {}
    for n in nodes:
        for i in n['inputs']:
            consumers.setdefault(_base(i), []).append(n)

    folded = {}
    for n in nodes:
        if n['op'] not in ('Conv2D', 'DepthwiseConv2dNative') or len(n['inputs']) < 2:
            continue
        w = source_const(n['inputs'][1])
        if w is None or not (w['name'].endswith('/weights') or w['name'].endswith('/depthwise_weights')):
            continue
        scope = w['name'].rsplit('/', 1)[0]
        if scope + '/BatchNorm/moving_variance' in consts or scope + '/biases' in consts:
            continue
        cur, hops = n, 0
        while hops < 4:                                   # BatchToSpaceND of a dilated convolution
            nxt = consumers.get(cur['name'], [])
            if len(nxt) != 1:
                break
            cur, hops = nxt[0], hops + 1
            if cur['op'] in ('BiasAdd', 'Add', 'AddV2'):
                others = [c for c in (source_const(i) for i in cur['inputs'])
                          if c is not None and c['tensor'] is not None]
                if others:
                    folded[scope + '/folded_bias'] = np.asarray(others[0]['tensor']).reshape(-1)
                break
            if cur['op'] != 'BatchToSpaceND':
                break
    out = {}
    if wanted is None:
        leaves = ('weights', 'depthwise_weights', 'biases', 'gamma', 'beta', 'moving_mean', 'moving_variance')
        out = {k: v for k, v in consts.items() if k.rsplit('/', 1)[-1] in leaves}
        out.update(folded)
        return out
    shapes = wanted if isinstance(wanted, dict) else {}
    missing = []
    for name in wanted:
        if name in consts:
            arr = consts[name]
            if name in shapes and tuple(shapes[name]) != tuple(arr.shape):
                raise ValueError('constant %s has shape %s in the graph, expected %s'
                                 % (name, arr.shape, tuple(shapes[name])))
            out[name] = arr
        elif '/BatchNorm/' in name and name.split('/BatchNorm/')[0] + '/folded_bias' in folded:
            key = name.split('/BatchNorm/')[0] + '/folded_bias'
            out[key] = folded[key]
        else:
            missing.append(name)
    if missing:
        raise KeyError('frozen graph %s is missing %d variables, e.g. %s' % (path, len(missing), missing[:3]))
    return out


def load_weights(path, wanted=None):
    """Weights from either source the Model Zoo archives hold: a frozen graph (`*.pb`) or a
    checkpoint prefix (`model.ckpt`)."""
    if str(path).endswith('.pb'):
        return load_variables(path, wanted)
    from fss import checkpoint
    return checkpoint.load_variables(path, wanted)


# -------------------------------------------------------------------- writer
def _tensor_proto(arr):
    arr = np.ascontiguousarray(arr)
    shape = b''.join(_ld(2, _varint(1 << 3) + _varint(int(d))) for d in arr.shape)
    return (_varint(1 << 3) + _varint(_DTYPE_CODES[arr.dtype]) + _ld(2, shape)
            + _ld(4, arr.astype(arr.dtype.newbyteorder('<')).tobytes()))


def _node(name, op, inputs=(), tensor=None):
    msg = _ld(1, name.encode('utf-8')) + _ld(2, op.encode('utf-8'))
    for i in inputs:
        msg += _ld(3, i.encode('utf-8'))
    if tensor is not None:
        attr_value = _ld(8, _tensor_proto(tensor))
        msg += _ld(5, _ld(1, b'value') + _ld(2, attr_value))
    return _ld(1, msg)


def save_frozen_graph(path, variables, convs=(), fold_batch_norm=False, epsilon=1e-5):
    """Writes a GraphDef shaped like a frozen inference graph: one Const (+ Identity `read`) per
    variable and, for every entry of `convs`, a convolution node reading the scope's kernel.
    An entry is (scope, dilated[, epsilon[, depthwise]]): `epsilon` is that layer's batch norm
    epsilon (default: the `epsilon` argument), `depthwise` makes the node a
    DepthwiseConv2dNative reading `<scope>/depthwise_weights`.
    fold_batch_norm: the BatchNorm constants of those convolutions are folded into the kernels
    and a bias that a BiasAdd applies (behind a BatchToSpaceND when `dilated`).  A scope whose
    batch norm is folded already (`<scope>/folded_bias` among the variables, as load_variables
    returns it) is written in that form whatever `fold_batch_norm` says: kernel untouched,
    the vector behind a BiasAdd."""
    variables = dict(variables)
    out = b''
    extra = []
    for entry in convs:
        scope, dilated = entry[0], entry[1]
        eps = entry[2] if len(entry) > 2 and entry[2] is not None else epsilon
        depthwise = len(entry) > 3 and entry[3]
        wname = scope + ('/depthwise_weights' if depthwise else '/weights')
        op = 'DepthwiseConv2dNative' if depthwise else 'Conv2D'
        conv_in = 'input_of/' + scope
        extra.append(_node(conv_in, 'Placeholder'))
        conv_node = scope + ('/depthwise' if depthwise else '/Conv2D')
        bias = None
        if scope + '/folded_bias' in variables:
            bias = np.asarray(variables.pop(scope + '/folded_bias'), np.float32)
        elif fold_batch_norm and scope + '/BatchNorm/moving_variance' in variables:
            g = variables.pop(scope + '/BatchNorm/gamma', None)
            b = variables.pop(scope + '/BatchNorm/beta', None)
            m = variables.pop(scope + '/BatchNorm/moving_mean')
            v = variables.pop(scope + '/BatchNorm/moving_variance')
            sc = 1.0 / np.sqrt(v.astype(np.float64) + eps)
            if g is not None:
                sc = sc * g
            bias = (-m * sc + (b if b is not None else 0.0)).astype(np.float32)
            w = variables[wname].astype(np.float64)
            # depthwise kernels are [kh, kw, c, 1]: the scale runs along axis 2
            variables[wname] = (w * (sc.reshape(-1, 1) if depthwise else sc)).astype(np.float32)
        if bias is not None:
            tail = conv_node
            variables[scope + '/folded/bias_const'] = bias
            if dilated:
                extra.append(_node(scope + '/BatchToSpaceND', 'BatchToSpaceND', [conv_node]))
                tail = scope + '/BatchToSpaceND'
            extra.append(_node(scope + '/BiasAdd', 'BiasAdd', [tail, scope + '/folded/bias_const']))
        extra.append(_node(conv_node, op, [conv_in, wname + '/read']))
    for name in sorted(variables):
        out += _node(name, 'Const', tensor=np.asarray(variables[name]))
        out += _node(name + '/read', 'Identity', [name])
    out += b''.join(extra)
    with open(path, 'wb') as f:
        f.write(out)


def unfold_batch_norm(weights, bn_layers):
    """Variables of the unfolded graph for weights whose batch norm is folded already: for every
    (scope, bn) of `bn_layers` (bn = the layer's batch norm variable names and epsilon) that has
    a `<scope>/folded_bias` in `weights`, BatchNorm statistics under which the layer computes
    exactly conv + that vector - mean = -bias, variance = 1 - epsilon, gamma = 1, beta = 0 - take
    the vector's place.  A checkpoint written from the result loads like any trained one."""
    out = dict(weights)
    for scope, bn in bn_layers:
        key = scope + '/folded_bias'
        if key not in out:
            continue
        bias = np.asarray(out.pop(key), np.float32)
        # float32(1 - eps) + eps is 1 to a few 1e-8: the folded scale is 1 to float precision
        out[bn['var']] = np.full(bias.shape, 1.0 - bn['epsilon'], np.float32)
        out[bn['mean']] = -bias
        if bn['gamma'] is not None:
            out[bn['gamma']] = np.ones(bias.shape, np.float32)
        if bn['beta'] is not None:
            out[bn['beta']] = np.zeros(bias.shape, np.float32)
    return out
