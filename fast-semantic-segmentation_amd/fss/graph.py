"""Define-then-run graph: symbolic NHWC tensors, ops and TF-style variable scopes.

The model code (architectures/, extractors/, third_party/) is written against
this module the way the reference is written against TensorFlow: calls build
nodes, nothing is computed until `fss.plan.compile` lowers the graph to HIP
kernel launches.  Variable names follow TensorFlow's variable-scope rules
(explicit scopes are taken verbatim, default-named scopes are made unique with
`_1`, `_2`, ... per parent scope) so weights are addressed by the names found
in checkpoints written by the reference.
"""
import contextlib


class Tensor(object):
    """Symbolic tensor. shape is (n, h, w, c); dtype in {'uint8','float32','int64'}."""

    def __init__(self, node, shape, dtype='float32'):
        self.node = node
        self.shape = tuple(int(s) for s in shape)
        self.dtype = dtype

    def get_shape(self):
        return self

    def as_list(self):
        return list(self.shape)

    def __repr__(self):
        return 'Tensor(%s#%d, %s, %s)' % (self.node.kind, self.node.index, self.shape, self.dtype)


class Node(object):
    def __init__(self, index, kind, inputs, attrs):
        self.index = index
        self.kind = kind
        self.inputs = list(inputs)
        self.attrs = attrs
        self.output = None


class Graph(object):
    def __init__(self):
        self.nodes = []
        self.variables = {}        # name -> shape (creation order)
        self._scope = []           # current variable-scope path
        self._scope_counts = {}    # full scope name -> times opened

    # ---- nodes
    def add(self, kind, inputs, shape, dtype='float32', **attrs):
        node = Node(len(self.nodes), kind, inputs, attrs)
        node.scope = self.scope_name
        node.output = Tensor(node, shape, dtype)
        self.nodes.append(node)
        return node.output

    # ---- variable scopes
    @property
    def scope_name(self):
        return '/'.join(self._scope)

    def _full(self, name):
        return '/'.join(self._scope + [name])

    @contextlib.contextmanager
    def variable_scope(self, name=None, default_name=None):
        if name is None:
            if default_name is None:
                raise ValueError('variable_scope needs a name or a default_name')
            name, i = default_name, 0
            while self._scope_counts.get(self._full(name), 0) > 0:
                i += 1
                name = '%s_%d' % (default_name, i)
        parts = name.split('/')
        saved = list(self._scope)
        for p in parts:
            self._scope.append(p)
            full = self.scope_name
            self._scope_counts[full] = self._scope_counts.get(full, 0) + 1
        try:
            yield self.scope_name
        finally:
            self._scope = saved

    def variable(self, name, shape):
        full = self._full(name)
        shape = tuple(int(s) for s in shape)
        if full in self.variables:
            if self.variables[full] != shape:
                raise ValueError('variable %s re-declared with shape %s (was %s)'
                                 % (full, shape, self.variables[full]))
        else:
            self.variables[full] = shape
        return full

    def consumers(self):
        """tensor-producing node index -> list of consumer nodes."""
        out = {n.index: [] for n in self.nodes}
        for n in self.nodes:
            for t in n.inputs:
                out[t.node.index].append(n)
        return out


_GRAPH_STACK = [Graph()]


def get_default_graph():
    return _GRAPH_STACK[-1]


def reset_default_graph():
    _GRAPH_STACK[-1] = Graph()
    return _GRAPH_STACK[-1]


@contextlib.contextmanager
def as_default(graph):
    _GRAPH_STACK.append(graph)
    try:
        yield graph
    finally:
        _GRAPH_STACK.pop()


def variable_scope(name=None, default_name=None):
    return get_default_graph().variable_scope(name, default_name)


# --------------------------------------------------------------------------
# ops (the subset of tf.* the inference / evaluation path uses)
# --------------------------------------------------------------------------
def placeholder(shape, dtype='uint8', name='inputs'):
    if len(shape) == 3:
        shape = (1,) + tuple(shape)
        squeeze = True
    else:
        squeeze = False
    if any(s is None for s in shape):
        raise ValueError('placeholder shape must be fully defined, got %s' % (shape,))
    return get_default_graph().add('placeholder', [], shape, dtype, name=name,
                                   rank3=squeeze)


def pad_to_bounding_box(x, offset_height, offset_width, target_height, target_width):
    if offset_height != 0 or offset_width != 0:
        raise ValueError('only zero offsets are supported')
    n, h, w, c = x.shape
    if target_height < h or target_width < w:
        raise ValueError('target size must be >= the input size')
    return get_default_graph().add('pad_to_bbox', [x], (n, target_height, target_width, c), x.dtype)


def crop_to_bounding_box(x, offset_height, offset_width, target_height, target_width):
    if offset_height != 0 or offset_width != 0:
        raise ValueError('only zero offsets are supported')
    n, h, w, c = x.shape
    if target_height > h or target_width > w:
        raise ValueError('crop size must be <= the input size')
    return get_default_graph().add('crop', [x], (n, target_height, target_width, c), x.dtype)


def expand_dims_batch(x):
    """Placeholders of rank 3 already carry a leading batch of 1."""
    return x


def to_float(x):
    if x.dtype == 'float32':
        return x
    return get_default_graph().add('to_float', [x], x.shape, 'float32')


def channel_affine(x, mean, scale):
    """(x - mean[c]) * scale[c], per channel."""
    if x.dtype != 'float32':
        raise ValueError('channel_affine expects a float32 tensor')
    if len(mean) != x.shape[3] or len(scale) != x.shape[3]:
        raise ValueError('mean/scale length must equal the channel count')
    return get_default_graph().add('affine', [x], x.shape, 'float32',
                                   mean=tuple(float(m) for m in mean),
                                   scale=tuple(float(s) for s in scale))


def resize_bilinear(x, size, align_corners=False):
    n, h, w, c = x.shape
    return get_default_graph().add('resize', [x], (n, int(size[0]), int(size[1]), c), x.dtype,
                                   align_corners=bool(align_corners))


def add(a, b):
    if a.shape != b.shape:
        raise ValueError('add: shape mismatch %s vs %s' % (a.shape, b.shape))
    return get_default_graph().add('add', [a, b], a.shape)


def relu(x):
    return get_default_graph().add('relu', [x], x.shape)


def concat(tensors, axis=-1):
    if axis not in (-1, 3):
        raise ValueError('only channel concat is supported')
    n, h, w, _ = tensors[0].shape
    for t in tensors:
        if t.shape[:3] != (n, h, w):
            raise ValueError('concat: spatial shape mismatch')
    return get_default_graph().add('concat', list(tensors),
                                   (n, h, w, sum(t.shape[3] for t in tensors)))


def argmax(x, axis=3):
    if axis != 3:
        raise ValueError('only channel argmax is supported')
    n, h, w, c = x.shape
    return get_default_graph().add('argmax', [x], (n, h, w, 1), 'int64')


def map_to_colored_labels(labels, color_map):
    """one_hot(labels) @ color_table: float32 [n,h,w,len(color_map[0])]."""
    table = tuple(tuple(float(v) for v in color) for color in color_map)
    n, h, w, _ = labels.shape
    return get_default_graph().add('lut', [labels], (n, h, w, len(table[0])), 'float32',
                                   table=table)
