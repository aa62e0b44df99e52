"""Runtime-built protobuf message classes for the pipeline config schema.

The reference compiles protos/*.proto with protoc (Makefile:26-27); protoc is
not required here: the same proto2 schema (package
fast_semantic_segmentation.protos; field names, numbers, types and defaults as
in the reference's protos/{pipeline,model,icnet,pspnet,hyperparams,losses,eval,
input_reader,train,optimizer,preprocessor,compressor}.proto) is declared below
as data and turned into message classes with the protobuf runtime, so
`text_format.Merge` parses the reference's .config files unchanged.
"""
from google.protobuf import descriptor_pb2
from google.protobuf import descriptor_pool
from google.protobuf import message_factory

_PKG = 'fast_semantic_segmentation.protos'
_F = descriptor_pb2.FieldDescriptorProto
_TYPES = {'int32': _F.TYPE_INT32, 'uint32': _F.TYPE_UINT32, 'float': _F.TYPE_FLOAT,
          'bool': _F.TYPE_BOOL, 'string': _F.TYPE_STRING}

# message -> list of (name, number, type, label, default, oneof)
#   type: scalar name, or a message / enum name of this package
_O, _R = 'optional', 'repeated'
_SCHEMA = {
    'PipelineConfig': [
        ('model', 1, 'FastSegmentationModel', _O, None, None),
        ('train_config', 2, 'TrainConfig', _O, None, None),
        ('train_input_reader', 3, 'InputReader', _O, None, None),
        ('eval_config', 4, 'EvalConfig', _O, None, None),
        ('eval_input_reader', 5, 'InputReader', _O, None, None)],
    'FastSegmentationModel': [
        ('pspnet', 1, 'PSPNet', _O, None, 'model'),
        ('icnet', 2, 'ICNet', _O, None, 'model')],
    'ICNet': [
        ('num_classes', 1, 'int32', _O, None, None),
        ('feature_extractor', 2, 'PSPNetICNetFeatureExtractor', _O, None, None),
        ('hyperparams', 3, 'Hyperparams', _O, None, None),
        ('loss', 4, 'Loss', _O, None, None),
        ('main_branch_loss_weight', 5, 'float', _O, '1.0', None),
        ('second_branch_loss_weight', 6, 'float', _O, '0.4', None),
        ('first_branch_loss_weight', 7, 'float', _O, '0.4', None),
        ('filter_scale', 8, 'float', _O, None, None),
        ('pooling_factors', 9, 'float', _R, None, None),
        ('pretrain_single_branch_mode', 10, 'bool', _O, 'false', None),
        ('mobile_ops_only', 11, 'bool', _O, 'false', None)],
    'PSPNet': [
        ('num_classes', 1, 'int32', _O, None, None),
        ('feature_extractor', 2, 'PSPNetICNetFeatureExtractor', _O, None, None),
        ('hyperparams', 3, 'Hyperparams', _O, None, None),
        ('loss', 4, 'Loss', _O, None, None),
        ('filter_scale', 5, 'float', _O, None, None),
        ('pooling_factors', 6, 'float', _R, None, None)],
    'PSPNetICNetFeatureExtractor': [('type', 1, 'string', _O, None, None)],
    'Hyperparams': [
        ('regularizer', 1, 'Regularizer', _O, None, None),
        ('initializer', 2, 'Initializer', _O, None, None),
        ('batch_norm', 3, 'BatchNorm', _O, None, None)],
    'Regularizer': [
        ('l1_regularizer', 1, 'L1Regularizer', _O, None, 'regularizer_oneof'),
        ('l2_regularizer', 2, 'L2Regularizer', _O, None, 'regularizer_oneof')],
    'L1Regularizer': [('weight', 1, 'float', _O, '1.0', None)],
    'L2Regularizer': [('weight', 1, 'float', _O, '1.0', None)],
    'Initializer': [
        ('truncated_normal_initializer', 1, 'TruncatedNormalInitializer', _O, None,
         'initializer_oneof'),
        ('variance_scaling_initializer', 2, 'VarianceScalingInitializer', _O, None,
         'initializer_oneof')],
    'TruncatedNormalInitializer': [
        ('mean', 1, 'float', _O, '0.0', None), ('stddev', 2, 'float', _O, '1.0', None)],
    'VarianceScalingInitializer': [
        ('factor', 1, 'float', _O, '2.0', None), ('uniform', 2, 'bool', _O, 'false', None),
        ('mode', 3, 'VarianceScalingInitializer.Mode', _O, 'FAN_IN', None)],
    'BatchNorm': [
        ('decay', 1, 'float', _O, '0.9997', None), ('center', 2, 'bool', _O, 'true', None),
        ('scale', 3, 'bool', _O, 'true', None), ('epsilon', 4, 'float', _O, '0.00001', None),
        ('train', 5, 'bool', _O, 'true', None)],
    'Loss': [
        ('classification_loss', 1, 'ClassificationLoss', _O, None, None),
        ('use_auxiliary_loss', 2, 'bool', _O, 'false', None),
        ('ignore_label', 3, 'uint32', _O, '255', None),
        ('upsample_logits', 4, 'bool', _O, 'false', None)],
    'ClassificationLoss': [('softmax', 1, 'SoftmaxClassificationLoss', _O, None, 'loss_type')],
    'SoftmaxClassificationLoss': [],
    'EvalConfig': [
        ('num_examples', 1, 'uint32', _O, '5000', None),
        ('ignore_label', 2, 'uint32', _O, '255', None),
        ('eval_input_type', 3, 'EvalInputType', _O, None, None)],
    'EvalInputType': [
        ('cropped_eval_input', 1, 'CroppedEvalInput', _O, None, 'eval_input_type_oneof'),
        ('padded_eval_input', 2, 'PaddedEvalInput', _O, None, 'eval_input_type_oneof')],
    'CroppedEvalInput': [('height', 1, 'int32', _O, None, None), ('width', 2, 'int32', _O, None, None)],
    'PaddedEvalInput': [('height', 1, 'int32', _O, None, None), ('width', 2, 'int32', _O, None, None)],
    'InputReader': [
        ('shuffle', 1, 'bool', _O, 'true', None),
        ('queue_capacity', 2, 'uint32', _O, '2000', None),
        ('min_after_dequeue', 3, 'uint32', _O, '1000', None),
        ('num_epochs', 4, 'uint32', _O, '0', None),
        ('num_readers', 5, 'uint32', _O, '1', None),
        ('num_parallel_calls', 6, 'uint32', _O, '1', None),
        ('shuffle_buffer', 7, 'uint32', _O, '2048', None),
        ('tf_record_input_reader', 8, 'TFRecordInputReader', _O, None, None),
        ('num_examples', 9, 'uint32', _O, None, None)],
    'TFRecordInputReader': [('input_path', 1, 'string', _R, None, None)],
    'TrainConfig': [
        ('batch_size', 1, 'uint32', _O, '32', None),
        ('preprocessor_step', 2, 'PreprocessingStep', _R, None, None),
        ('keep_checkpoint_every_n_hours', 3, 'uint32', _O, '1000', None),
        ('optimizer', 4, 'Optimizer', _O, None, None),
        ('fine_tune_checkpoint', 5, 'string', _O, '', None),
        ('fine_tune_checkpoint_type', 6, 'string', _O, '', None),
        ('freeze_fine_tune_backbone', 7, 'bool', _O, 'false', None),
        ('num_steps', 8, 'uint32', _O, '0', None),
        ('batch_queue_capacity', 9, 'int32', _O, '150', None),
        ('num_batch_queue_threads', 10, 'int32', _O, '8', None),
        ('prefetch_queue_capacity', 11, 'int32', _O, '5', None),
        ('add_regularization_loss', 12, 'bool', _O, 'true', None),
        ('quantize_with_delay', 13, 'int32', _O, None, None),
        ('num_queue_threads', 14, 'int32', _O, '8', None)],
    'Optimizer': [
        ('momentum_optimizer', 1, 'MomentumOptimizer', _O, None, 'optimizer'),
        ('adam_optimizer', 2, 'AdamOptimizer', _O, None, 'optimizer')],
    'MomentumOptimizer': [
        ('learning_rate', 1, 'LearningRate', _O, None, None),
        ('momentum_optimizer_value', 2, 'float', _O, '0.9', None)],
    'AdamOptimizer': [('learning_rate', 1, 'LearningRate', _O, None, None)],
    'LearningRate': [
        ('constant_learning_rate', 1, 'ConstantLearningRate', _O, None, 'learning_rate'),
        ('exponential_decay_learning_rate', 2, 'ExponentialDecayLearningRate', _O, None,
         'learning_rate'),
        ('polynomial_decay_learning_rate', 3, 'PolynomialDecayLearningRate', _O, None,
         'learning_rate')],
    'ConstantLearningRate': [('learning_rate', 1, 'float', _O, '0.002', None)],
    'ExponentialDecayLearningRate': [
        ('initial_learning_rate', 1, 'float', _O, '0.002', None),
        ('decay_steps', 2, 'uint32', _O, '4000000', None),
        ('decay_factor', 3, 'float', _O, '0.95', None),
        ('staircase', 4, 'bool', _O, 'true', None)],
    'PolynomialDecayLearningRate': [
        ('initial_learning_rate', 1, 'float', _O, '0.0001', None),
        ('decay_steps', 2, 'uint32', _O, '4000000', None),
        ('power', 3, 'float', _O, '0.9', None)],
    'PreprocessingStep': [
        ('set_fixed_image_size', 1, 'SetFixedImageSize', _O, None, 'preprocessing_step'),
        ('aspect_ratio_image_resize', 2, 'AspectRatioImageResize', _O, None, 'preprocessing_step'),
        ('random_image_scale', 3, 'RandomImageScale', _O, None, 'preprocessing_step'),
        ('random_image_crop', 4, 'RandomCropImage', _O, None, 'preprocessing_step'),
        ('random_horizontal_flip', 5, 'RandomHorizontalFlip', _O, None, 'preprocessing_step')],
    'SetFixedImageSize': [
        ('fixed_height', 1, 'int32', _O, None, None), ('fixed_width', 2, 'int32', _O, None, None),
        ('images_channel_dim', 3, 'int32', _O, '3', None),
        ('labels_channel_dim', 4, 'int32', _O, '1', None)],
    'AspectRatioImageResize': [
        ('min_dimension', 1, 'int32', _O, '600', None),
        ('max_dimension', 2, 'int32', _O, '1024', None),
        ('resize_method', 3, 'ResizeType', _O, 'BILINEAR', None),
        ('pad_to_max_dimension', 4, 'bool', _O, 'false', None)],
    'RandomImageScale': [
        ('min_scale_ratio', 1, 'float', _O, '0.5', None),
        ('max_scale_ratio', 2, 'float', _O, '2.0', None)],
    'RandomCropImage': [
        ('crop_height', 1, 'int32', _O, None, None), ('crop_width', 2, 'int32', _O, None, None),
        ('images_channel_dim', 3, 'int32', _O, '3', None),
        ('labels_channel_dim', 4, 'int32', _O, '1', None),
        ('int_seed', 5, 'int32', _O, None, None)],
    'RandomHorizontalFlip': [],
    'CommpressionConfig': [('compression_strategy', 1, 'CompressionStrategy', _O, None, None)],
    'CompressionStrategy': [('filter_pruner', 1, 'FilterPruner', _O, None, 'compression_strategy')],
    'FilterPruner': [
        ('input', 1, 'NodeName', _O, None, None), ('output', 2, 'NodeName', _O, None, None),
        ('node_scope', 3, 'string', _O, None, None),
        ('compression_factor', 4, 'float', _O, '0.5', None),
        ('skip_node', 5, 'NodeName', _R, None, None),
        ('node', 6, 'SingleNodePruningSpec', _R, None, None)],
    'SingleNodePruningSpec': [
        ('target', 1, 'NodeName', _O, None, None), ('source', 2, 'NodeName', _O, None, None),
        ('following', 3, 'NodeName', _R, None, None),
        ('node_scope', 4, 'string', _O, 'null', None)],
    'NodeName': [('name', 1, 'string', _O, None, None)],
}
_NESTED_ENUMS = {'VarianceScalingInitializer': {'Mode': [('FAN_IN', 0), ('FAN_OUT', 1),
                                                         ('FAN_AVG', 2)]}}
_TOP_ENUMS = {'ResizeType': [('BILINEAR', 0), ('NEAREST_NEIGHBOR', 1), ('BICUBIC', 2)]}
_ENUM_NAMES = {'ResizeType', 'VarianceScalingInitializer.Mode'}


def _build():
    fdp = descriptor_pb2.FileDescriptorProto()
    fdp.name = 'fss/protos/schema.proto'
    fdp.package = _PKG
    fdp.syntax = 'proto2'
    for ename, values in _TOP_ENUMS.items():
        e = fdp.enum_type.add()
        e.name = ename
        for vname, num in values:
            v = e.value.add()
            v.name, v.number = vname, num
    for mname, fields in _SCHEMA.items():
        m = fdp.message_type.add()
        m.name = mname
        for ename, values in _NESTED_ENUMS.get(mname, {}).items():
            e = m.enum_type.add()
            e.name = ename
            for vname, num in values:
                v = e.value.add()
                v.name, v.number = vname, num
        oneofs = []
        for fname, number, ftype, label, default, oneof in fields:
            f = m.field.add()
            f.name, f.number = fname, number
            f.label = _F.LABEL_REPEATED if label == _R else _F.LABEL_OPTIONAL
            if ftype in _TYPES:
                f.type = _TYPES[ftype]
            elif ftype in _ENUM_NAMES:
                f.type = _F.TYPE_ENUM
                f.type_name = '.%s.%s' % (_PKG, ftype)
            else:
                f.type = _F.TYPE_MESSAGE
                f.type_name = '.%s.%s' % (_PKG, ftype)
            if default is not None:
                f.default_value = default
            if oneof is not None:
                if oneof not in oneofs:
                    oneofs.append(oneof)
                    m.oneof_decl.add().name = oneof
                f.oneof_index = oneofs.index(oneof)
    pool = descriptor_pool.DescriptorPool()
    fd = pool.Add(fdp) if hasattr(pool, 'Add') and not hasattr(pool, 'AddSerializedFile') \
        else pool.AddSerializedFile(fdp.SerializeToString())
    classes = {}
    for mname in _SCHEMA:
        desc = pool.FindMessageTypeByName('%s.%s' % (_PKG, mname))
        classes[mname] = message_factory.GetMessageClass(desc)
    return classes


MESSAGES = _build()


def export(namespace, names):
    for n in names:
        namespace[n] = MESSAGES[n]
