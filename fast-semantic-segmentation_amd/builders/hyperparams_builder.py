"""Builds the layer defaults ("arg scope") from a Hyperparams config.

Same surface as the reference's builders/hyperparams_builder.py:45-68: every
conv / separable conv gets relu6 and batch norm with the configured
center / scale / epsilon.  Regularizer and initializer settings only matter for
training; they are validated and otherwise ignored.
"""
from fss import layers


def _check_regularizer(regularizer):
    kind = regularizer.WhichOneof('regularizer_oneof')
    if kind not in ('l1_regularizer', 'l2_regularizer'):
        raise ValueError('Unknown regularizer function: {}'.format(kind))


def _check_initializer(initializer):
    kind = initializer.WhichOneof('initializer_oneof')
    if kind not in ('truncated_normal_initializer', 'variance_scaling_initializer'):
        raise ValueError('Unknown initializer function: {}'.format(kind))


def build(hyperparams_config, is_training):
    if not hyperparams_config.HasField('batch_norm'):
        raise ValueError('hyperparams.batch_norm must be set.')
    bn = hyperparams_config.batch_norm
    batch_norm_params = {'decay': bn.decay, 'center': bn.center, 'scale': bn.scale,
                         'epsilon': bn.epsilon, 'is_training': is_training and bn.train}
    _check_regularizer(hyperparams_config.regularizer)
    _check_initializer(hyperparams_config.initializer)
    affected_ops = [layers.conv2d, layers.separable_conv2d, 'conv2d_transpose']
    with layers.arg_scope(affected_ops, activation_fn=layers.relu6,
                          normalizer_fn=layers.batch_norm,
                          normalizer_params=batch_norm_params):
        with layers.arg_scope([layers.batch_norm], **batch_norm_params) as sc:
            return sc
