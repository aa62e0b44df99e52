"""Builds a segmentation model from a FastSegmentationModel config.

Same surface and error behaviour as the reference's builders/model_builder.py
(`build` :109-123, `_build_pspnet_icnet_model` :41-106, extractor registry
:15-20).  One deliberate difference: the extractor key 'dialted_mobilenet',
under which the reference's own MobileNet ICNet config is shipped
(configs/single_stage_icnet_1.0_1025_mobilenet_v2.config:15), is accepted as an
alias of 'dilated_mobilenet' so that config builds.
"""
from architectures import icnet_architecture
from architectures import pspnet_architecture
from builders import hyperparams_builder
from builders import losses_builder
from extractors import pspnet_icnet_mobilenet_v2
from extractors import pspnet_icnet_resnet_v1
from protos import model_pb2

PSPNET_ICNET_FEATURE_EXTRACTER = {
    'dilated_resnet50': pspnet_icnet_resnet_v1.PSPNetICNetDilatedResnet50FeatureExtractor,
    'dilated_mobilenet': pspnet_icnet_mobilenet_v2.PSPNetICNetMobilenetFeatureExtractor,
}
_EXTRACTOR_ALIASES = {'dialted_mobilenet': 'dilated_mobilenet'}


def _build_pspnet_icnet_extractor(feature_extractor_config, filter_scale, is_training,
                                  mid_downsample=False, reuse_weights=None):
    feature_type = feature_extractor_config.type
    feature_type = _EXTRACTOR_ALIASES.get(feature_type, feature_type)
    if feature_type not in PSPNET_ICNET_FEATURE_EXTRACTER:
        raise ValueError('Unknown ICNet feature_extractor: {}'.format(feature_type))
    return PSPNET_ICNET_FEATURE_EXTRACTER[feature_type](
        is_training, batch_norm_trainable=is_training, filter_scale=filter_scale,
        mid_downsample=mid_downsample, reuse_weights=reuse_weights)


def _build_pspnet_icnet_model(model_config, is_training, add_summaries,
                              build_baseline_psp=False):
    num_classes = model_config.num_classes
    if not num_classes:
        raise ValueError('"num_classes" must be greater than 0.')
    in_filter_scale = model_config.filter_scale
    if in_filter_scale > 1 or in_filter_scale < 0:
        raise ValueError('"filter_scale" must be in the range (0,1].')
    if in_filter_scale == 0:
        raise ValueError('"filter_scale" must be in the range (0,1].')
    filter_scale = 1.0 / in_filter_scale

    pretrain_single_branch_mode = False
    should_downsample_extractor = False
    if not build_baseline_psp:
        pretrain_single_branch_mode = model_config.pretrain_single_branch_mode
        should_downsample_extractor = not pretrain_single_branch_mode

    feature_extractor = _build_pspnet_icnet_extractor(
        model_config.feature_extractor, filter_scale, is_training,
        mid_downsample=should_downsample_extractor)
    model_arg_scope = hyperparams_builder.build(model_config.hyperparams, is_training)
    loss_config = model_config.loss
    common_kwargs = {
        'is_training': is_training,
        'num_classes': num_classes,
        'model_arg_scope': model_arg_scope,
        'feature_extractor': feature_extractor,
        'classification_loss': losses_builder.build(loss_config),
        'use_aux_loss': loss_config.use_auxiliary_loss,
        'upsample_train_logits': loss_config.upsample_logits,
        'add_summaries': add_summaries,
        'filter_scale': filter_scale,
        'pooling_factors': list(model_config.pooling_factors),
    }
    if build_baseline_psp:
        if loss_config.use_auxiliary_loss:
            common_kwargs.update(main_loss_weight=1.0, aux_loss_weight=0.4)
        return num_classes, pspnet_architecture.PSPNetArchitecture(**common_kwargs)
    if loss_config.use_auxiliary_loss:
        common_kwargs.update(
            main_loss_weight=model_config.main_branch_loss_weight,
            second_branch_loss_weight=model_config.second_branch_loss_weight,
            first_branch_loss_weight=model_config.first_branch_loss_weight)
    return num_classes, icnet_architecture.ICNetArchitecture(
        pretrain_single_branch_mode=pretrain_single_branch_mode,
        no_add_n_op=model_config.mobile_ops_only, **common_kwargs)


def build(model_config, is_training, add_summaries=True):
    if not isinstance(model_config, model_pb2.FastSegmentationModel):
        raise ValueError('model_config not of type model_pb2.FastSegmentationModel.')
    if is_training:
        raise NotImplementedError('only inference / evaluation models are built '
                                  '(is_training must be False)')
    model = model_config.WhichOneof('model')
    if model == 'pspnet':
        return _build_pspnet_icnet_model(model_config.pspnet, is_training, add_summaries,
                                         build_baseline_psp=True)
    elif model == 'icnet':
        return _build_pspnet_icnet_model(model_config.icnet, is_training, add_summaries)
    raise ValueError('Unknown model: {}'.format(model))
