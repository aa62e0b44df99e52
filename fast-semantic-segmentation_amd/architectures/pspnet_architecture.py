"""PSPNet architecture and the feature-extractor plug-in base class.

Same public surface as the reference's architectures/pspnet_architecture.py
(`PSPNetArchitecture` :29-211, `PSPNetFeatureExtractor` :214-251); the graph
is built with `fss.graph` / `fss.layers` and executed by HIP kernels.
Inference only: `loss` and training-mode heads raise NotImplementedError.
"""
from fss import graph as G
from fss import layers
from libs import base_model as model
from libs import compressible_ops as ops


def dynamic_interpolation(features, s_factor=1.0, z_factor=1.0):
    """Bilinear (align_corners) resize to ((in-1)*s+1) zoomed by z."""
    _, h, w, _ = features.shape
    shrink_h, shrink_w = (h - 1) * s_factor + 1, (w - 1) * s_factor + 1
    zoom_h = shrink_h + (shrink_h - 1) * (z_factor - 1)
    zoom_w = shrink_w + (shrink_w - 1) * (z_factor - 1)
    return G.resize_bilinear(features, [int(zoom_h), int(zoom_w)], align_corners=True)


class PSPNetArchitecture(model.FastSegmentationModel):
    def __init__(self, is_training, model_arg_scope, num_classes, feature_extractor,
                 classification_loss, filter_scale, pooling_factors, use_aux_loss=True,
                 main_loss_weight=1, aux_loss_weight=0, upsample_train_logits=False,
                 add_summaries=True, scope=None):
        super(PSPNetArchitecture, self).__init__(num_classes=num_classes)
        if is_training:
            raise NotImplementedError('training graphs are not part of this build')
        self._is_training = is_training
        self._model_arg_scope = model_arg_scope
        self._feature_extractor = feature_extractor
        self._classification_loss = classification_loss
        self._filter_scale = filter_scale
        self._pooling_factors = list(pooling_factors)
        self._output_zoom_factor = 8.0
        self._scope = scope

    @property
    def main_class_predictions_key(self):
        return 'class_predictions'

    @property
    def aux_predictions_key(self):
        return 'aux_predictions'

    @property
    def main_loss_key(self):
        return 'loss'

    @property
    def aux_loss_key(self):
        return 'aux_loss'

    def preprocess(self, inputs):
        if inputs.dtype != 'float32':
            raise ValueError('`preprocess` expects a tf.float32 tensor')
        return self._feature_extractor.preprocess(inputs)

    def predict(self, preprocessed_inputs):
        with layers.arg_scope(self._model_arg_scope):
            _, backbone, _ = self._feature_extractor.extract_features(
                preprocessed_inputs, scope=self.shared_feature_extractor_scope)
            final = self._pspnet_pspmodule(backbone)
            with G.variable_scope('Predictions'):
                predictions = ops.conv2d(final, self._num_classes, 1, 1, prediction_output=True)
                predictions = dynamic_interpolation(predictions,
                                                    z_factor=self._output_zoom_factor)
            return {self.main_class_predictions_key: predictions}

    def _pspnet_pspmodule(self, input_features, scope=None):
        """Pyramid pooling: per factor avg-pool -> 1x1 conv -> resize back;
        concat with the input; 3x3 conv."""
        _, h, w, _ = input_features.shape
        branches = [input_features]
        with G.variable_scope(scope, default_name='PSPModule'):
            for factor in self._pooling_factors:
                window = (int(h / factor), int(w / factor))
                pooled = layers.avg_pool2d(input_features, window, stride=window)
                pooled = ops.conv2d(pooled, 512, 1, stride=1)
                branches.append(G.resize_bilinear(pooled, (h, w), align_corners=True))
            merged = G.concat(branches, axis=-1)
            return ops.conv2d(merged, 512, 3, stride=1, compression_ratio=self._filter_scale)

    def loss(self, prediction_dict):
        raise NotImplementedError('training losses are not part of this build')


class PSPNetFeatureExtractor(object):
    """Backbone plug-in: subclasses implement `preprocess` and
    `_extract_features` -> (half_res, quarter_res, psp_aux)."""

    def __init__(self, is_training, features_stride, batch_norm_trainable=False,
                 reuse_weights=None, weight_decay=0.0):
        self._is_training = is_training
        self._features_stride = features_stride
        self._train_batch_norm = batch_norm_trainable and is_training
        self._reuse_weights = reuse_weights
        self._weight_decay = weight_decay

    def preprocess(self, raw_inputs):
        raise NotImplementedError

    def extract_features(self, preprocessed_inputs, scope=None):
        with G.variable_scope(scope):
            return self._extract_features(preprocessed_inputs, scope)

    def _extract_features(self, preprocessed_inputs, scope):
        raise NotImplementedError
