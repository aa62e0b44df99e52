r"""Export a segmentation model: inference checkpoint + frozen inference graph.

Same flags as the reference's export.py:25-45 (--input_shape --pad_to_shape --config_path
--trained_checkpoint --output_dir --output_colours) plus --fold_batch_norm.  Like the reference
(export.py:82-137) it builds the inference graph, prints the parameter / FLOP analysis of the
model (trainable parameters without batch norm; float operations of the convolutions as tfprof
counts them, 2 per multiply-add), and writes into --output_dir

  model.ckpt.{index,data-00000-of-00001}, checkpoint
        the variables of the inference graph only (no optimizer slots, no aux heads),
        TF checkpoint-V2 format;
  frozen_inference_graph.pb
        the variables as Const nodes under their variable names (+ the Conv2D node that reads each
        `weights`), with --fold_batch_norm in the form the Model Zoo archives describe: batch norm
        folded into the kernels and a BiasAdd constant (docs/model_zoo.md:8-12);
  export_graph.pbtxt
        a text listing of the graph's nodes (name, op, inputs, shape).

The frozen graph holds what this package's readers (fss.frozen_graph, inference.py, eval.py)
take from such a file, i.e. the weights; the arithmetic nodes of a TensorFlow GraphDef are not
written, so a TensorFlow runtime cannot execute it.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def profile_inference_graph(graph):
    """(trainable parameters without batch norm, float operations of the convolutions), printed
    the way export.py:60-79 asks tfprof for them; returns both totals."""
    params, flops = 0, 0
    rows = []
    for nd in graph.nodes:
        if nd.kind not in ('conv2d', 'depthwise'):
            continue
        a = nd.attrs
        shape = graph.variables[a['weights']]
        n = int(np.prod(shape))
        if a.get('biases') is not None:
            n += int(np.prod(graph.variables[a['biases']]))
        _, ho, wo, cout = nd.output.shape
        if nd.kind == 'conv2d':
            kh, kw, cin, _ = shape
            f = 2 * ho * wo * cout * kh * kw * cin
        else:
            kh, kw, c, _ = shape
            f = 2 * ho * wo * c * kh * kw
        f *= nd.output.shape[0]
        params += n
        flops += f
        rows.append((a['weights'].rsplit('/', 1)[0], n, f))
    for name, n, f in rows:
        print('  %-90s %10d params %14d flops' % (name, n, f))
    print('total trainable parameters (without batch norm): %d (%.2f M)' % (params, params / 1e6))
    print('total float operations of the convolutions: %d (%.2f GFLOPs)' % (flops, flops / 1e9))
    return params, flops


def export_inference_graph(pipeline_config, trained_checkpoint_prefix, output_directory,
                           input_shape=None, pad_to_shape=None, output_colours=False,
                           output_collection_name='predictions', fold_batch_norm=False):
    from builders import model_builder
    from fss import checkpoint, frozen_graph, graph as G
    from libs.constants import CITYSCAPES_LABEL_COLORS
    from libs.exporter import deploy_segmentation_inference_graph

    graph = G.reset_default_graph()
    _, segmentation_model = model_builder.build(pipeline_config.model, is_training=False)
    os.makedirs(output_directory, exist_ok=True)
    frozen_graph_path = os.path.join(output_directory, 'frozen_inference_graph.pb')
    eval_graphdef_path = os.path.join(output_directory, 'export_graph.pbtxt')
    model_path = os.path.join(output_directory, 'model.ckpt')

    outputs, placeholder_tensor = deploy_segmentation_inference_graph(
        model=segmentation_model, input_shape=input_shape, pad_to_shape=pad_to_shape,
        label_color_map=(CITYSCAPES_LABEL_COLORS if output_colours else None),
        output_collection_name=output_collection_name)
    totals = profile_inference_graph(graph)

    def name_of(nd):
        scope = nd.attrs['weights'].rsplit('/', 1)[0] + '/' if 'weights' in nd.attrs else ''
        return '%s%s_%d' % (scope, nd.kind, nd.index)

    with open(eval_graphdef_path, 'w') as f:
        for nd in graph.nodes:
            f.write('node {\n  name: "%s"\n  op: "%s"\n' % (name_of(nd), nd.kind))
            for t in nd.inputs:
                f.write('  input: "%s"\n' % name_of(t.node))
            f.write('  shape: "%s"\n}\n' % (tuple(nd.output.shape),))

    # the inference variables, from a checkpoint prefix or another frozen graph
    weights = frozen_graph.load_weights(trained_checkpoint_prefix, graph.variables)
    layers = [nd for nd in graph.nodes if nd.kind in ('conv2d', 'depthwise')]
    # a source with folded batch norm: the checkpoint (and the unfolded frozen graph) get
    # statistics that reproduce the folded bias, so both load like a trained model's
    plain = frozen_graph.unfold_batch_norm(
        weights, [(nd.attrs['weights'].rsplit('/', 1)[0], nd.attrs['bn']) for nd in layers if nd.attrs['bn']])
    checkpoint.save_variables(model_path, plain)
    checkpoint.write_checkpoint_state(output_directory, [model_path])
    # every layer folds with its own batch norm epsilon, as the plan compiler does
    convs = [(nd.attrs['weights'].rsplit('/', 1)[0], nd.attrs['rate'] > 1,
              nd.attrs['bn']['epsilon'] if nd.attrs['bn'] else None, nd.kind == 'depthwise') for nd in layers]
    frozen_graph.save_frozen_graph(frozen_graph_path, weights if fold_batch_norm else plain, convs,
                                   fold_batch_norm=fold_batch_norm)
    print('Done!')
    return totals


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--input_shape', default=None,
                    help='[batch,] height, width, channels of the placeholder')
    ap.add_argument('--pad_to_shape', default=None, help='height, width the input is padded to')
    ap.add_argument('--config_path', required=True)
    ap.add_argument('--trained_checkpoint', required=True,
                    help='checkpoint prefix (path/to/model.ckpt) or a frozen_inference_graph.pb')
    ap.add_argument('--output_dir', required=True)
    ap.add_argument('--output_colours', action='store_true')
    ap.add_argument('--fold_batch_norm', action='store_true',
                    help='fold batch norm into the kernels of the frozen graph (the Model Zoo form)')
    flags = ap.parse_args(argv)

    from google.protobuf import text_format
    from protos import pipeline_pb2
    pipeline_config = pipeline_pb2.PipelineConfig()
    with open(flags.config_path) as f:
        text_format.Merge(f.read(), pipeline_config)
    if not flags.input_shape:
        raise ValueError('Must supply `input_shape`: placeholder shapes are static here')
    input_shape = [int(d) for d in flags.input_shape.split(',')]
    pad_to_shape = [int(d) for d in flags.pad_to_shape.split(',')] if flags.pad_to_shape else None
    return export_inference_graph(pipeline_config, flags.trained_checkpoint, flags.output_dir,
                                  input_shape, pad_to_shape, flags.output_colours,
                                  fold_batch_norm=flags.fold_batch_norm)


if __name__ == '__main__':
    main()
