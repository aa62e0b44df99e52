"""MobileNetV2 feature extractor for ICNet / PSPNet.

Same surface as the reference's extractors/pspnet_icnet_mobilenet_v2.py:15-95:
[-1, 1] normalisation (:44-46) and the layer_5 / layer_18 / layer_8 end points
(:48-72).
"""
from fss import graph as G
from architectures import pspnet_architecture
from third_party import mobilenet_v2


class PSPNetICNetMobilenetV2FeatureExtractor(pspnet_architecture.PSPNetFeatureExtractor):
    _channel_means = [127.5, 127.5, 127.5]

    def __init__(self, architecture, mobilenet_model, is_training, filter_scale, features_stride,
                 mid_downsample=False, batch_norm_trainable=False, reuse_weights=None,
                 weight_decay=0.0):
        if features_stride != 8:
            raise ValueError('`features_stride` must be 8 for ICNet and PSPNet.')
        self._filter_scale = filter_scale
        self._architecture = architecture
        self._mobilenet_model = mobilenet_model
        self._mid_downsample = mid_downsample
        self._depth_multiplier = 1.0
        super(PSPNetICNetMobilenetV2FeatureExtractor, self).__init__(
            is_training, features_stride, batch_norm_trainable, reuse_weights, weight_decay)

    def preprocess(self, raw_inputs):
        m = self._channel_means
        return G.channel_affine(raw_inputs, m, [1.0 / v for v in m])

    def _extract_features(self, preprocessed_inputs, scope):
        conv_defs = mobilenet_v2.make_conv_defs(filter_scale=self._filter_scale,
                                                mid_downsample=self._mid_downsample)
        exact = self._depth_multiplier == 1.0
        _, activations = self._mobilenet_model(
            preprocessed_inputs, conv_defs=conv_defs, depth_multiplier=self._depth_multiplier,
            min_depth=8 if exact else 1, divisible_by=8 if exact else 1,
            output_stride=self._features_stride, final_endpoint='layer_18')
        return activations['layer_5'], activations['layer_18'], activations['layer_8']


class PSPNetICNetMobilenetFeatureExtractor(PSPNetICNetMobilenetV2FeatureExtractor):
    def __init__(self, is_training, filter_scale=1.0, mid_downsample=False, features_stride=8,
                 batch_norm_trainable=False, reuse_weights=None, weight_decay=0.0):
        super(PSPNetICNetMobilenetFeatureExtractor, self).__init__(
            'MobilenetV2', mobilenet_v2.mobilenet_base, is_training, filter_scale,
            features_stride, mid_downsample, batch_norm_trainable, reuse_weights, weight_decay)
