"""L1-norm filter pruning of a model's weights ("Pruning Filters for Efficient ConvNets").

Counterpart of the reference's libs/filter_pruner.py:54-416: the same traversal, pruning specs
and index rules, run on this package's graph (fss.graph) and a name -> ndarray weights dict
instead of a TensorFlow GraphDef and session.  Host-side numpy only; nothing here runs on the
GPU.  The graph is viewed through the TensorFlow node names the reference's prune configs use:

    conv2d node with scope S  ->  S/Conv2D [-> S/BatchToSpaceND if dilated]
                                  -> S/BatchNorm/FusedBatchNorm | S/BiasAdd  [-> S/Relu | S/Relu6]
    every other node          ->  one node named <kind>_<index> with a TensorFlow-like op name

A spec (target conv, optional source conv, following nodes) keeps the int(n * factor) filters of
the target with the largest L1 norm - or the source's kept set - and removes the same channels
from the following batch norms (gamma, beta, moving statistics) and from the input channels of
the following convolutions (filter_pruner.py:163-297).  Convolutions without a hand-written
spec get theirs from the traversal: their own batch norm and the next convolutions downstream
(:300-374).
"""
import collections
import os

import numpy as np

FilterPrunerNodeSpec = collections.namedtuple('FilterPrunerNodeSpec', ['source', 'target', 'following'])

_OPS = {'add': 'Add', 'relu': 'Relu', 'resize': 'ResizeBilinear', 'max_pool': 'MaxPool',
        'avg_pool': 'AvgPool', 'concat': 'ConcatV2', 'affine': 'Sub', 'to_float': 'Cast',
        'pad_to_bbox': 'Pad', 'placeholder': 'Placeholder', 'argmax': 'ArgMax', 'depthwise':
        'DepthwiseConv2dNative', 'lut': 'MatMul', 'crop': 'Slice'}


class _TFNode(object):
    def __init__(self, name, op, scope=None):
        self.name, self.op, self.scope = name, op, scope
        self.next = []


def tf_view(graph, input_name='Preprocessor/sub'):
    """{TensorFlow-style node name: _TFNode} of an fss graph; the channel-affine preprocessing
    node carries `input_name` (the reference's `Preprocessor/sub`)."""
    nodes, first, last = {}, {}, {}

    def new(name, op, scope=None):
        n = _TFNode(name, op, scope)
        nodes[name] = n
        return n

    for nd in graph.nodes:
        if nd.kind == 'conv2d':
            scope = nd.attrs['weights'].rsplit('/', 1)[0]
            chain = [new(scope + '/Conv2D', 'Conv2D', scope)]
            if nd.attrs['rate'] > 1:
                chain.append(new(scope + '/BatchToSpaceND', 'BatchToSpaceND', scope))
            if nd.attrs['bn'] is not None:
                chain.append(new(scope + '/BatchNorm/FusedBatchNorm', 'FusedBatchNorm', scope))
            else:
                chain.append(new(scope + '/BiasAdd', 'BiasAdd', scope))
            if nd.attrs['activation'] is not None:
                op = 'Relu6' if nd.attrs['activation'] == 'relu6' else 'Relu'
                chain.append(new(scope + '/' + op, op, scope))
            for a, b in zip(chain, chain[1:]):
                a.next.append(b.name)
            first[nd.index], last[nd.index] = chain[0], chain[-1]
        else:
            name = input_name if nd.kind == 'affine' else '%s_%d' % (nd.kind, nd.index)
            first[nd.index] = last[nd.index] = new(name, _OPS.get(nd.kind, nd.kind))
    for nd in graph.nodes:
        for t in nd.inputs:
            last[t.node.index].next.append(first[nd.index].name)
    return nodes


class FilterPruner(object):
    """Same constructor arguments as the reference's class (filter_pruner.py:56-87); `compress`
    takes the model's graph and weights instead of a checkpoint path with a `.meta` graph."""

    def __init__(self, input_node, output_node, compression_factor, init_pruner_specs=None,
                 skippable_nodes=(), interactive_mode=False, pruner_mode='ICNetPruner',
                 soft_apply=False, **unused):
        if pruner_mode != 'ICNetPruner':
            raise ValueError('Currently only the pruner mode `ICNetPruner` is implemented for '
                             'pruning ICNet filters.')
        if interactive_mode:
            raise NotImplementedError('interactive (plotting) mode is not supported')
        self.input_node, self.output_node = input_node, output_node
        self.compression_factor = compression_factor
        self.init_pruner_specs = dict(init_pruner_specs or {})
        self.skippable_nodes = [s for s in skippable_nodes if s]
        self.soft_apply = soft_apply
        self.pruner_specs = []
        self.nodes_map, self.values_map, self.output_values_map = {}, {}, {}

    # ---- graph walking (filter_pruner.py:112-160, 300-374)
    def _reachable_neighbours(self):
        """consumer lists restricted to nodes from which the output node can be reached (the
        reference builds its adjacency backwards from the output)."""
        preds = collections.defaultdict(list)
        for n in self.nodes_map.values():
            for m in n.next:
                preds[m].append(n.name)
        live, stack = set(), [self.output_node]
        while stack:
            name = stack.pop()
            if name not in live:
                live.add(name)
                stack.extend(preds[name])
        return {name: [m for m in n.next if m in live] for name, n in self.nodes_map.items() if name in live}

    def _next_op_instances(self, name, op):
        if self.nodes_map[name].op == op:
            return [name]
        out = []
        for nxt in self.neighbors.get(name, []):
            for r in self._next_op_instances(nxt, op):
                if r not in out:
                    out.append(r)
        return out

    def _following_bn_and_convs(self, name):
        convs, bns = [], []
        for nxt in self.neighbors.get(name, []):
            node = self.nodes_map[nxt]
            if node.op == 'BatchToSpaceND':
                found = self._next_op_instances(nxt, 'FusedBatchNorm')
                if len(found) > 1:
                    raise ValueError('Something went wrong with BatchNorms...')
                nxt = found[0]
                node = self.nodes_map[nxt]
            if node.op == 'FusedBatchNorm':
                bns = [nxt]
                convs = self._next_op_instances(nxt, 'Conv2D')
            elif node.op in ('Relu', 'Relu6'):
                convs = self._next_op_instances(nxt, 'Conv2D')
            elif node.op == 'Conv2D':
                convs = [nxt]
            elif nxt == self.output_node:
                return None
            else:
                raise ValueError('Incompatable model file.')
        return bns + convs

    def _create_specs(self, start):
        visited, stack = set(), [start]
        # depth-first in consumer order, as the reference's recursion
        def visit(name):
            if name in visited:
                return
            visited.add(name)
            node = self.nodes_map[name]
            if name not in self.skippable_nodes:
                if name in self.init_pruner_specs:
                    self.pruner_specs.append(self.init_pruner_specs[name])
                elif node.op == 'Conv2D':
                    dependants = self._following_bn_and_convs(name)
                    if dependants:
                        self.pruner_specs.append(FilterPrunerNodeSpec(source='', target=name, following=dependants))
            for nxt in self.neighbors.get(name, []):
                visit(nxt)
        import sys
        limit = sys.getrecursionlimit()
        sys.setrecursionlimit(max(limit, 10000))
        try:
            visit(start)
        finally:
            sys.setrecursionlimit(limit)

    # ---- applying the specs (filter_pruner.py:163-297)
    def _weights_name(self, conv_name):
        return self.nodes_map[conv_name].scope + '/weights'

    def _prune_idxs(self, weights_name):
        w = self.values_map[weights_name]
        n = w.shape[-1]
        keep = int(n * self.compression_factor)
        magnitude = np.abs(w).sum((0, 1, 2))
        top = np.sort(np.argsort(-magnitude)[:keep])
        idxs = np.zeros(n, dtype=bool)
        idxs[top] = True
        return idxs

    def _current(self, name):
        return self.output_values_map[name] if name in self.output_values_map else self.values_map[name]

    def _prune_conv(self, conv_name, idxs=None):
        wn = self._weights_name(conv_name)
        w = self._current(wn)
        keep = self._prune_idxs(wn) if idxs is None else idxs
        out = np.copy(w)
        if self.soft_apply:
            out[:, :, :, ~keep] = 0
        else:
            out = out[:, :, :, keep]
        self.output_values_map[wn] = out
        return keep

    def _prune_bn(self, bn_name, keep):
        base = self.nodes_map[bn_name].scope + '/BatchNorm/'
        for leaf in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            if base + leaf not in self.values_map:
                continue                               # batch norm without scale / centre
            v = np.copy(self.values_map[base + leaf])
            if self.soft_apply:
                v[~keep] = 0
            else:
                v = v[keep]
            self.output_values_map[base + leaf] = v

    def _prune_conv_inputs(self, conv_name, src_keep):
        wn = self._weights_name(conv_name)
        w = self._current(wn)
        cin = w.shape[2]
        keep = np.copy(src_keep)
        if cin != len(src_keep):
            # behind a channel concat (the last convolution of the PSP module): the other
            # branches' channels follow the pruned ones and are all kept
            keep = np.resize(keep, cin)
            keep[len(src_keep):] = True
        out = np.copy(w)
        if self.soft_apply:
            out[:, :, ~keep, :] = 0
        else:
            out = out[:, :, keep, :]
        self.output_values_map[wn] = out

    def _apply(self):
        self.output_values_map = {}
        kept = {}
        for spec in self.pruner_specs:
            node = self.nodes_map[spec.target]
            if node.op != 'Conv2D':
                raise ValueError('Only Conv nodes can be prunned with the FilterPruner compressor.')
            src = None
            if spec.source:
                src = kept[spec.source] if spec.source in kept else self._prune_idxs(self._weights_name(spec.source))
            keep = self._prune_conv(spec.target, src)
            kept[spec.target] = keep
            for name in spec.following:
                f = self.nodes_map[name]
                if f.op == 'FusedBatchNorm':
                    self._prune_bn(name, keep)
                elif f.op == 'Conv2D':
                    self._prune_conv_inputs(name, keep)
                else:
                    raise ValueError('Following nodes should only be BatchNorms or Convs...')

    def compress(self, graph, weights, skip_apply=False):
        """graph: the model's fss graph (inference form); weights: {variable name: ndarray}."""
        self.nodes_map = tf_view(graph, self.input_node)
        for name in [self.input_node, self.output_node] + list(self.init_pruner_specs):
            if name not in self.nodes_map:
                raise KeyError('the graph has no node named %s' % name)
        self.values_map = {k: np.asarray(v) for k, v in weights.items()}
        self.neighbors = self._reachable_neighbours()
        self.pruner_specs = []
        self._create_specs(self.input_node)
        if not skip_apply:
            self._apply()
        return self.pruned_variables()

    def pruned_variables(self):
        """Every variable of the input, pruned where a spec touched it (filter_pruner.py:388-401)."""
        return {k: (self.output_values_map[k] if k in self.output_values_map else v)
                for k, v in self.values_map.items()}

    def save(self, output_checkpoint_dir, output_checkpoint_name):
        from fss import checkpoint
        prefix = os.path.join(output_checkpoint_dir, output_checkpoint_name)
        checkpoint.save_variables(prefix, self.pruned_variables())
        checkpoint.write_checkpoint_state(output_checkpoint_dir, [prefix])
        print('Saving pruned model checkpoint to {}'.format(prefix))
        return prefix
