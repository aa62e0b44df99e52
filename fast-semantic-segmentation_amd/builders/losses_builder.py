"""Loss config validation (reference: builders/losses_builder.py).  Losses are
only evaluated in training, which this build does not include; the returned
callable documents that."""


def build(loss_config):
    kind = loss_config.classification_loss.WhichOneof('loss_type')
    if kind != 'softmax':
        raise ValueError('Empty loss config.')

    def classification_loss(predictions, labels):
        raise NotImplementedError('training losses are not part of this build')
    return classification_loss
