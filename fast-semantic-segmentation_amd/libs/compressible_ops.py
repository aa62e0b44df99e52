"""Convolution whose width can be divided by a pruning compression ratio.

Same call surface as the reference's libs/compressible_ops.py:9-30.
"""
from fss import layers


def convolution2d_compressible(inputs, num_outputs, kernel_size, stride=1,
                               compression_ratio=1.0, prediction_output=False, **kwargs):
    if prediction_output:       # plain biased conv: no activation, no batch norm
        kwargs['activation_fn'] = None
        kwargs['normalizer_fn'] = None
    return layers.conv2d(inputs, int(num_outputs // compression_ratio), kernel_size, stride,
                         **kwargs)


compressible_conv2d = convolution2d_compressible
conv2d_compressible = convolution2d_compressible
conv2d = convolution2d_compressible
