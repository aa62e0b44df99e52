"""Abstract segmentation model: the interface the exporter and evaluator call.

Same surface as the reference's libs/base_model.py:26-102.
"""
import abc


class FastSegmentationModel(abc.ABC):
    def __init__(self, num_classes):
        self._num_classes = num_classes
        self._groundtruth_labels = {}

    @property
    def shared_feature_extractor_scope(self):
        return 'SharedFeatureExtractor'

    @property
    def num_classes(self):
        return self._num_classes

    @abc.abstractmethod
    def preprocess(self, inputs):
        """float32 [batch, h, w, 3] image tensor in 0..255 -> network input."""

    @abc.abstractmethod
    def predict(self, preprocessed_inputs):
        """network input -> dict of prediction tensors."""

    @abc.abstractmethod
    def loss(self, prediction_dict):
        """training loss (not available in this inference-only build)."""

    def provide_groundtruth(self, groundtruth_labels):
        self._groundtruth_labels = groundtruth_labels
