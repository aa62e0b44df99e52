r"""Build a TFRecord for the Cityscapes semantic segmentation dataset, without TensorFlow.

Same flags, file patterns, output name and feature keys as the reference's
dataset_tools/create_cityscapes_tfrecord.py:16-25,33-41,63-92,106-137: the record written for
`--split_type val` is `<output_dir>/cityscapes_val.record`, one tf.Example per image with the
encoded PNG bytes of image and label map (the files are not re-encoded), in sorted file order.
The framing (length, masked CRC-32C, payload, masked CRC-32C) and the Example encoding come from
fss.tfrecord; builders/dataset_builder.py reads the result for eval.py.
"""
import argparse
import glob
import io
import logging
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

_DEFAULT_PATTEN = {
    'input': '*_leftImg8bit.png',
    'annot': '*_gtFine_labelTrainIds.png',
}

_DEFAULT_DIR = {
    'image': 'leftImg8bit',
    'label': 'gtFine',
}


def _open_file(full_path):
    import PIL.Image
    with open(full_path, 'rb') as fid:
        encoded_file = fid.read()
    image = PIL.Image.open(io.BytesIO(encoded_file))
    return image, encoded_file


def create_tf_example(image_path, label_path, image_dir='', is_jpeg=False):
    """Serialized tf.Example (bytes) of one image / label pair."""
    from fss import tfrecord
    from libs.standard_fields import TFRecordFields as F
    file_format = 'jpeg' if is_jpeg else 'png'
    full_image_path = os.path.join(image_dir, image_path)
    full_label_path = os.path.join(image_dir, label_path)
    image, encoded_image = _open_file(full_image_path)
    label, encoded_label = _open_file(full_label_path)
    height, width = image.height, image.width
    if height != label.height or width != label.width:
        raise ValueError('Input and annotated images must have same dims.'
                         'verify the matching pair for {}'.format(full_image_path))
    return tfrecord.make_example({
        F.image_encoded: encoded_image,
        F.image_filename: full_image_path.encode('utf8'),
        F.image_format: file_format.encode('utf8'),
        F.image_height: height,
        F.image_width: width,
        F.image_channels: 3,
        F.segmentation_class_encoded: encoded_label,
        F.segmentation_class_format: 'png'.encode('utf8'),
    })


def _create_tf_record(images, labels, output_path):
    from fss import tfrecord

    def examples():
        for idx, image in enumerate(images):
            if idx % 100 == 0:
                logging.info('On image %d of %d', idx, len(images))
            yield create_tf_example(image, labels[idx], is_jpeg=False)
    tfrecord.write_records(output_path, examples())
    logging.info('Finished writing!')


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--input_pattern', default='', help='Pattern matching input images for Cityscapes.')
    ap.add_argument('--annot_pattern', default='', help='Pattern matching ground truth images for Cityscapes.')
    ap.add_argument('--cityscapes_dir', default='', help='Cityscapes dataset root folder.')
    ap.add_argument('--split_type', default='', help='Type of split: `train`, `test` or `val`.')
    ap.add_argument('--output_dir', default='', help='Output data directory.')
    flags = ap.parse_args(argv)
    logging.getLogger().setLevel(logging.INFO)
    assert flags.output_dir, '`output_dir` missing.'
    assert flags.split_type, '`split_type` missing.'
    assert flags.cityscapes_dir or (flags.input_pattern and flags.annot_pattern), \
        'Must specify either `cityscapes_dir` or `input_pattern` and `annot_pattern`.'
    os.makedirs(flags.output_dir, exist_ok=True)
    output_path = os.path.join(flags.output_dir, 'cityscapes_{}.record'.format(flags.split_type))
    if flags.cityscapes_dir:
        image_filenames = glob.glob(os.path.join(
            flags.cityscapes_dir, _DEFAULT_DIR['image'], flags.split_type, '*', _DEFAULT_PATTEN['input']))
        annot_filenames = glob.glob(os.path.join(
            flags.cityscapes_dir, _DEFAULT_DIR['label'], flags.split_type, '*', _DEFAULT_PATTEN['annot']))
    else:
        image_filenames = glob.glob(flags.input_pattern)
        annot_filenames = glob.glob(flags.annot_pattern)
    if len(image_filenames) != len(annot_filenames):
        raise ValueError('Supplied patterns do not have image counts.')
    _create_tf_record(sorted(image_filenames), sorted(annot_filenames), output_path=output_path)
    return output_path


if __name__ == '__main__':
    main()
