"""Builds the filter pruner from a CompressionStrategy config.

Counterpart of the reference's builders/compressor_builder.py:13-77: node names of a spec are
completed with the config's `node_scope` (or the spec's own), a leading "..." marks an absolute
name."""
import functools
import os

from libs.filter_pruner import FilterPruner, FilterPrunerNodeSpec
from protos import compressor_pb2


def _complete_node_scope(name, parent_scope, overide_scope=None):
    if not name:
        return ''
    if name[:3] == '...':
        return name[3:]
    main_scope = overide_scope if overide_scope is not None else parent_scope
    return os.path.join(main_scope, name)


def _build_filter_pruning_compressor(config, skippable_nodes, compression_factor, interactive_mode,
                                     soft_apply):
    skip = [n.name for n in config.skip_node] + list(skippable_nodes)
    specs = {}
    for node in config.node:
        overide = node.node_scope if node.node_scope != 'null' else None
        complete = functools.partial(_complete_node_scope, parent_scope=config.node_scope,
                                     overide_scope=overide)
        spec = FilterPrunerNodeSpec(source=complete(node.source.name), target=complete(node.target.name),
                                    following=[complete(f.name) for f in node.following])
        specs[spec.target] = spec
    return FilterPruner(input_node=config.input.name, output_node=config.output.name,
                        compression_factor=compression_factor, init_pruner_specs=specs,
                        skippable_nodes=skip, interactive_mode=interactive_mode, soft_apply=soft_apply)


def build(compression_config, skippable_nodes, compression_factor, interactive_mode=False,
          soft_apply=False):
    if not isinstance(compression_config, compressor_pb2.CompressionStrategy):
        raise ValueError('pruner_config not of type compressor_pb2.CompressionStrategy.')
    strategy = compression_config.WhichOneof('compression_strategy')
    if strategy == 'filter_pruner':
        return _build_filter_pruning_compressor(compression_config.filter_pruner, skippable_nodes,
                                                compression_factor, interactive_mode, soft_apply)
    raise ValueError('Unknown compression strategy: {}'.format(strategy))
