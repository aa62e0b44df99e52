"""ICNet architecture (http://arxiv.org/abs/1704.08545), inference graph.

Same public surface as the reference's architectures/icnet_architecture.py
(`ICNetArchitecture` :21-346): half-resolution backbone, fast pyramid pooling
module (:176-211), full-resolution branch (:213-226), two cascade feature
fusions (:228-253), 1x1 classifier and the final x4 zoom (:131-144).
Inference only: `loss`, the aux heads and `restore_map` for fine-tuning are
training features and raise NotImplementedError.
"""
from fss import graph as G
from fss import layers
from architectures.pspnet_architecture import dynamic_interpolation
from libs import base_model as model
from libs import compressible_ops as ops


class ICNetArchitecture(model.FastSegmentationModel):
    def __init__(self, is_training, model_arg_scope, num_classes, feature_extractor,
                 classification_loss, filter_scale, pooling_factors,
                 pretrain_single_branch_mode=False, use_aux_loss=True, main_loss_weight=1.0,
                 first_branch_loss_weight=0.0, second_branch_loss_weight=0.0,
                 upsample_train_logits=False, add_summaries=True, no_add_n_op=False,
                 scope=None):
        super(ICNetArchitecture, self).__init__(num_classes=num_classes)
        if is_training:
            raise NotImplementedError('training graphs are not part of this build')
        self._is_training = is_training
        self._model_arg_scope = model_arg_scope
        self._feature_extractor = feature_extractor
        self._classification_loss = classification_loss
        self._filter_scale = filter_scale
        self._pooling_factors = list(pooling_factors)
        self._pretrain_single_branch_mode = pretrain_single_branch_mode
        self._output_zoom_factor = 8 if pretrain_single_branch_mode else 4
        self._scope = scope

    @property
    def main_class_predictions_key(self):
        return 'class_predictions'

    @property
    def first_aux_predictions_key(self):
        return 'first_aux_predictions'

    @property
    def second_aux_predictions_key(self):
        return 'second_aux_predictions'

    @property
    def single_branch_mode_predictions_key(self):
        return 'single_branch_mode_predictions'

    @property
    def main_loss_key(self):
        return 'loss'

    def preprocess(self, inputs):
        if inputs.dtype != 'float32':
            raise ValueError('`preprocess` expects a tf.float32 tensor')
        return self._feature_extractor.preprocess(inputs)

    def _extract_shared_features(self, preprocessed_inputs, scope):
        extractor_inputs = preprocessed_inputs
        if not self._pretrain_single_branch_mode:
            extractor_inputs = dynamic_interpolation(preprocessed_inputs, s_factor=0.5)
        return self._feature_extractor.extract_features(extractor_inputs, scope=scope)

    def predict(self, preprocessed_inputs):
        with layers.arg_scope(self._model_arg_scope):
            half_res, quarter_res, _ = self._extract_shared_features(
                preprocessed_inputs, scope=self.shared_feature_extractor_scope)
            pooled_quarter_res = self._icnet_pspmodule(quarter_res)
            if not self._pretrain_single_branch_mode:
                full_res = self._third_feature_branch(preprocessed_inputs)
                first_fusion, _ = self._cascade_feature_fusion(pooled_quarter_res, half_res)
                second_fusion, _ = self._cascade_feature_fusion(first_fusion, full_res)
                final_logits = dynamic_interpolation(second_fusion, z_factor=2.0)
            else:
                final_logits = pooled_quarter_res
            with G.variable_scope('Predictions'):
                predictions = ops.conv2d(final_logits, self._num_classes, 1, 1,
                                         prediction_output=True)
                predictions = dynamic_interpolation(predictions,
                                                    z_factor=self._output_zoom_factor)
            return {self.main_class_predictions_key: predictions}

    def _icnet_pspmodule(self, input_features, scope=None):
        """x + sum over pooling factors of resize(avg_pool(x)), then 1x1 conv."""
        _, h, w, _ = input_features.shape
        added = input_features
        with G.variable_scope(scope, default_name='FastPSPModule'):
            for factor in self._pooling_factors:
                window = (int(h / factor), int(w / factor))
                pooled = layers.avg_pool2d(input_features, window, stride=window)
                pooled = G.resize_bilinear(pooled, (h, w), align_corners=True)
                added = G.add(added, pooled)
            return ops.conv2d(added, 512, 1, stride=1, compression_ratio=self._filter_scale)

    def _third_feature_branch(self, preprocessed_inputs):
        fs = self._filter_scale
        net = ops.conv2d(preprocessed_inputs, 64, (3, 3), stride=2, compression_ratio=fs)
        net = ops.conv2d(net, 64, (3, 3), stride=2, compression_ratio=fs)
        net = ops.conv2d(net, 128, (3, 3), stride=2, compression_ratio=fs)
        return ops.conv2d(net, 256, (3, 3), stride=1, compression_ratio=fs)

    def _cascade_feature_fusion(self, first_feature_map, second_feature_map, scope=None):
        """relu(conv1x1(second) + dilated_conv3x3(upsample x2(first))); both convs
        are linear (batch norm, no activation)."""
        with G.variable_scope(scope, default_name='CascadeFeatureFusion'):
            upsampled = dynamic_interpolation(first_feature_map, z_factor=2.0)
            dilated_conv = ops.conv2d(upsampled, 256, (3, 3), stride=1,
                                      compression_ratio=self._filter_scale, rate=2,
                                      activation_fn=None, scope='DilatedConv')
            conv = ops.conv2d(second_feature_map, 256, (1, 1),
                              compression_ratio=self._filter_scale, activation_fn=None,
                              scope='Conv')
            dilated_conv = G.resize_bilinear(dilated_conv, conv.shape[1:3])
            return G.relu(G.add(conv, dilated_conv)), upsampled

    def loss(self, prediction_dict):
        raise NotImplementedError('training losses are not part of this build')

    def restore_map(self, fine_tune_checkpoint_type='segmentation'):
        raise NotImplementedError('fine-tune restore maps are a training feature')
