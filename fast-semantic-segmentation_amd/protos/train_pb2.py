"""Message classes of the reference's protos/train.proto, built at import time
(see protos/_schema.py); stands in for the protoc-generated module."""
from protos import _schema

_schema.export(globals(), ['TrainConfig'])
