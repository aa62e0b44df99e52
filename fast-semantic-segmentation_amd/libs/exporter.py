"""Inference graph deployment.

Same surface as the reference's libs/exporter.py:32-80: a uint8 image
placeholder is zero-padded to `pad_to_shape`, converted to float, preprocessed,
run through the model, arg-maxed, optionally mapped through a colour / id table
and cropped back to `input_shape`.  The returned `outputs` / `placeholder`
pair is symbolic; `Session` (below) plays the role of `tf.Session` +
`Saver.restore` + `sess.run` of the reference's inference.py:88-98.
"""
import numpy as np

from fss import graph as G


def _map_to_colored_labels(segmentation_map, shape_list, color_map):
    return G.map_to_colored_labels(segmentation_map, color_map)


def _get_outputs_from_inputs(model, input_tensors):
    inputs = G.to_float(G.expand_dims_batch(input_tensors))
    preprocessed_inputs = model.preprocess(inputs)
    outputs_dict = model.predict(preprocessed_inputs)
    output_tensors = outputs_dict[model.main_class_predictions_key]
    return G.argmax(output_tensors, 3)          # [n, h, w, 1] int64


def _image_tensor_input_placeholder(input_shape=None, pad_to_shape=None):
    if input_shape is None or any(d is None for d in input_shape):
        raise ValueError('`input_shape` must be fully defined (got %r): kernels '
                         'and buffers are planned for one image size' % (input_shape,))
    placeholder_tensor = G.placeholder(tuple(input_shape), dtype='uint8', name='inputs')
    if pad_to_shape is not None:
        input_tensor = G.pad_to_bounding_box(placeholder_tensor, 0, 0,
                                             pad_to_shape[0], pad_to_shape[1])
    else:
        input_tensor = placeholder_tensor
    return placeholder_tensor, input_tensor


def deploy_segmentation_inference_graph(model, input_shape, pad_to_shape=None,
                                        label_color_map=None,
                                        output_collection_name='predictions'):
    placeholder_tensor, input_tensor = _image_tensor_input_placeholder(input_shape, pad_to_shape)
    outputs = _get_outputs_from_inputs(model, input_tensor)
    if label_color_map is not None:
        outputs = _map_to_colored_labels(outputs, outputs.shape, label_color_map)
    if pad_to_shape is not None:
        h, w = (input_shape[0], input_shape[1]) if len(input_shape) == 3 else \
            (input_shape[1], input_shape[2])
        outputs = G.crop_to_bounding_box(outputs, 0, 0, h, w)
    return outputs, placeholder_tensor


class Session(object):
    """Compiles `outputs` for the GPU with the given weights; `run` is the
    counterpart of `sess.run(outputs, feed_dict={placeholder: image})`."""

    def __init__(self, outputs, placeholder, weights, dtype='mixed', graph=None, **plan_kwargs):
        from fss import plan      # needs the GPU library; imported on first use
        self._outputs = outputs if isinstance(outputs, (list, tuple)) else [outputs]
        self._single = not isinstance(outputs, (list, tuple))
        self.placeholder = placeholder
        # batch norm statistics may be absent when the weights come from a frozen graph that
        # has them folded into the kernels (then `<scope>/folded_bias` stands in for them)
        missing = [n for n in (graph or G.get_default_graph()).variables if n not in weights
                   and not ('/BatchNorm/' in n and n.split('/BatchNorm/')[0] + '/folded_bias' in weights)]
        if missing:
            raise KeyError('checkpoint is missing %d variables, e.g. %s'
                           % (len(missing), missing[:3]))
        self.plan = plan.CompiledGraph(graph or G.get_default_graph(), self._outputs, weights,
                                       dtype=dtype, **plan_kwargs)

    def run(self, image):
        res = self.plan.run(image)
        res = [r.astype(np.int64) if t.dtype == 'int64' else r
               for r, t in zip(res, self._outputs)]
        return res[0] if self._single else res
