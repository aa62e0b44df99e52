"""Dilated ResNet-50 v1 feature extractor for ICNet / PSPNet.

Same surface as the reference's extractors/pspnet_icnet_resnet_v1.py:16-91:
mean subtraction (:43-45) and the block1 / block4 / block2 end points (:47-68).
"""
from fss import graph as G
from fss import layers
from architectures import pspnet_architecture
from third_party import dilated_resnet_v1
from third_party import resnet_utils


class PSPNetICNetResnetV1FeatureExtractor(pspnet_architecture.PSPNetFeatureExtractor):
    _channel_means = [123.68, 116.779, 103.939]     # RGB ImageNet means

    def __init__(self, architecture, resnet_model, is_training, filter_scale, features_stride,
                 mid_downsample=False, batch_norm_trainable=False, reuse_weights=None,
                 weight_decay=0.0):
        if features_stride != 8:
            raise ValueError('`features_stride` must be 8 for ICNet and PSPNet.')
        self._filter_scale = filter_scale
        self._architecture = architecture
        self._resnet_model = resnet_model
        self._mid_downsample = mid_downsample
        super(PSPNetICNetResnetV1FeatureExtractor, self).__init__(
            is_training, features_stride, batch_norm_trainable, reuse_weights, weight_decay)

    def preprocess(self, raw_inputs):
        return G.channel_affine(raw_inputs, self._channel_means, [1.0, 1.0, 1.0])

    def _extract_features(self, preprocessed_inputs, scope):
        prefix = scope + '/' + self._architecture
        with layers.arg_scope(resnet_utils.resnet_arg_scope(
                batch_norm_epsilon=1e-5, batch_norm_scale=True,
                weight_decay=self._weight_decay)):
            _, activations = self._resnet_model(
                preprocessed_inputs, filter_scale=self._filter_scale,
                mid_downsample=self._mid_downsample, num_classes=None, is_training=False,
                global_pool=False, output_stride=self._features_stride)
        return (activations[prefix + '/block1'], activations[prefix + '/block4'],
                activations[prefix + '/block2'])


class PSPNetICNetDilatedResnet50FeatureExtractor(PSPNetICNetResnetV1FeatureExtractor):
    def __init__(self, is_training, filter_scale=1.0, mid_downsample=False, features_stride=8,
                 batch_norm_trainable=False, reuse_weights=None, weight_decay=0.0):
        super(PSPNetICNetDilatedResnet50FeatureExtractor, self).__init__(
            'resnet_v1_50', dilated_resnet_v1.dilated_resnet_v1_50, is_training, filter_scale,
            features_stride, mid_downsample, batch_norm_trainable, reuse_weights, weight_decay)
