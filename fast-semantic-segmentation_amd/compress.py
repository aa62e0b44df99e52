r"""Filter pruning of a trained checkpoint (host-side).

Same flags as the reference's compress.py:27-60 (--input_checkpoint, --prune_config,
--compression_factor, --output_dir, --skippable_nodes, --soft_apply) plus --config_path: the
reference reads the model graph from the checkpoint's `.meta` file, this package rebuilds it
from the pipeline config.  Writes <output_dir>/pruned_model.ckpt (+ the `checkpoint` state file);
the result loads into the model whose filter_scale equals the compression factor.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--input_checkpoint', required=True)
    ap.add_argument('--prune_config', required=True)
    ap.add_argument('--config_path', required=True, help='pipeline config of the model in the checkpoint')
    ap.add_argument('--compression_factor', type=float, default=0.5)
    ap.add_argument('--output_dir', default='')
    ap.add_argument('--skippable_nodes', default='')
    ap.add_argument('--soft_apply', action='store_true')
    flags = ap.parse_args(argv)

    from google.protobuf import text_format
    from builders import compressor_builder, model_builder
    from fss import frozen_graph, graph as G
    from protos import compressor_pb2, pipeline_pb2

    if not os.path.exists(flags.input_checkpoint + '.index') and not flags.input_checkpoint.endswith('.pb'):
        print('The input checkpoint prefix specified from `FLAGS.input_checkpoint` must point to a '
              'location with valid checkpoint files.')
        return -1
    compression_config = compressor_pb2.CommpressionConfig()
    with open(flags.prune_config) as f:
        text_format.Merge(f.read(), compression_config)
    pipeline_config = pipeline_pb2.PipelineConfig()
    with open(flags.config_path) as f:
        text_format.Merge(f.read(), pipeline_config)
    graph = G.reset_default_graph()
    _, model = model_builder.build(pipeline_config.model, is_training=False)
    model.predict(model.preprocess(G.to_float(G.placeholder((1, 129, 257, 3)))))
    weights = frozen_graph.load_weights(flags.input_checkpoint, graph.variables)
    compressor = compressor_builder.build(compression_config.compression_strategy,
                                          flags.skippable_nodes.replace(' ', '').split(','),
                                          flags.compression_factor, soft_apply=flags.soft_apply)
    os.makedirs(flags.output_dir or '.', exist_ok=True)
    compressor.compress(graph, weights)
    compressor.save(flags.output_dir or '.', 'pruned_model.ckpt')
    return 0


if __name__ == '__main__':
    sys.exit(main())
