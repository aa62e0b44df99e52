"""Builds the evaluation input from an InputReader config.

Counterpart of the reference's builders/dataset_builder.py:65-106.  The
reference returns a dict of dataset tensors; here `build` returns a generator of
(image uint8 [h,w,3], label uint8 [h,w]) pairs decoded from the TFRecord files
(`image/encoded` + `image/segmentation/class/encoded`, PNG or JPEG), in file
order.  Shuffling is a training feature and is ignored with a note.
"""
import glob
import io
import logging

import numpy as np

from fss import tfrecord
from libs import standard_fields as fields
from protos import input_reader_pb2


def decode_example(serialized):
    from PIL import Image
    ex = tfrecord.parse_example(serialized)
    enc = ex.get(fields.TFRecordFields.image_encoded)
    lab = ex.get(fields.TFRecordFields.segmentation_class_encoded)
    if not enc or not lab:
        raise ValueError('"tensor_dict" must have both imageand label fields')
    image = np.asarray(Image.open(io.BytesIO(enc[0])).convert('RGB'), dtype=np.uint8)
    label = np.asarray(Image.open(io.BytesIO(lab[0])))
    if label.ndim == 3:
        label = label[..., 0]
    return image, label.astype(np.uint8)


def build(input_reader_config, shard=None, limit=None):
    """shard = (rank, world): only records whose index (in file order, over all files) is
    congruent to rank modulo world are decoded and yielded, the others are skipped before
    any PNG / JPEG decoding; limit = number of records (over all ranks) to consider."""
    if not isinstance(input_reader_config, input_reader_pb2.InputReader):
        raise ValueError('input_reader_config not of type input_reader_pb2.InputReader.')
    reader_config = input_reader_config.tf_record_input_reader
    paths = []
    for pattern in reader_config.input_path:
        paths += sorted(glob.glob(pattern))
    if not paths:
        raise ValueError('At least one input path must be specified in `input_reader_config`.')
    if input_reader_config.shuffle:
        logging.info('input_reader.shuffle is ignored: evaluation reads records in file order')

    rank, world = shard if shard is not None else (0, 1)

    def generator():
        index = 0
        for path in paths:
            for record in tfrecord.read_records(path):
                if limit is not None and index >= limit:
                    return
                if index % world == rank:
                    yield decode_example(record)
                index += 1
    return generator()
