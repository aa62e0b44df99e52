"""Keys of the TFRecord features and of the decoded example dict
(same strings as the reference's libs/standard_fields.py:7-25)."""


class TFRecordFields(object):
    image_encoded = 'image/encoded'
    image_filename = 'image/filename'
    image_format = 'image/format'
    image_height = 'image/height'
    image_width = 'image/width'
    image_channels = 'image/channels'
    segmentation_class_encoded = 'image/segmentation/class/encoded'
    segmentation_class_format = 'image/segmentation/class/format'


class GroundtruthFields(object):
    input_image = 'input_image'
    input_image_path = 'input_image_path'
    input_image_height = 'input_image_height'
    input_image_width = 'input_image_width'
    output_mask = 'label_mask'
