"""Message classes of the reference's protos/hyperparams.proto, built at import time
(see protos/_schema.py); stands in for the protoc-generated module."""
from protos import _schema

_schema.export(globals(), ['Hyperparams', 'Regularizer', 'L1Regularizer', 'L2Regularizer', 'Initializer', 'TruncatedNormalInitializer', 'VarianceScalingInitializer', 'BatchNorm'])
