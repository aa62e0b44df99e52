"""mIoU evaluation of a segmentation model on the GPU.

Same surface as the reference's libs/evaluator.py (`create_evaluation_input`
:21-46, `create_predictions_and_labels` :49-88, `eval_segmentation_model(_once)`
:91-216).  Differences that follow from not running inside TensorFlow:

  * `create_input_fn()` returns an iterable of (image uint8 [h,w,3],
    label uint8 [h,w] or [h,w,1]) pairs instead of a dict of dataset tensors;
  * weights come from `weights=` (a name -> ndarray dict) or from the checkpoint
    named by `evaluate_single_checkpoint` / found in `train_dir`;
  * examples are sharded over ranks when a torch.distributed process group is
    initialised (example i goes to rank i % world_size) and the int64 confusion
    matrix is all-reduced once at the end;
  * summaries / TensorBoard events are not written.
Image and label are zero-padded to `input_dimensions` exactly as
pad_to_specific_size does: padded label pixels are class 0 and do count.
"""
import functools
import inspect
import itertools
import logging

import numpy as np

from fss import graph as G


def _accepts(fn, names):
    try:
        params = inspect.signature(fn).parameters
    except (TypeError, ValueError):
        return False
    return all(n in params for n in names)


def create_evaluation_input(create_input_dict_fn, input_height, input_width,
                            cropped_eval=False):
    """Returns a generator of (image uint8 [h,w,3], label uint8 [H,W]) with the
    label already zero-padded to (input_height, input_width)."""
    if cropped_eval:
        raise NotImplementedError('cropped (random-crop) evaluation is not supported')

    def generator():
        for image, label in create_input_dict_fn():
            image = np.asarray(image)
            label = np.asarray(label)
            if label.ndim == 3:
                label = label[..., 0]
            if image.ndim != 3 or label.ndim != 2:
                raise ValueError('Images or Labels in tensor_dict should be rank 4')
            if image.shape[0] > input_height or image.shape[1] > input_width:
                raise ValueError('example larger than the padded evaluation size')
            padded = np.zeros((input_height, input_width), np.uint8)
            padded[:label.shape[0], :label.shape[1]] = label
            yield image.astype(np.uint8), padded
    return generator()


def create_predictions_and_labels(model, image_shape, input_height, input_width,
                                  cropped_eval=False, batch_size=1):
    """Builds the evaluation graph: placeholder -> pad -> float -> preprocess ->
    predict -> (same-size resize) -> argmax.  Returns (predictions, placeholder)."""
    if cropped_eval:
        raise NotImplementedError('cropped (random-crop) evaluation is not supported')
    placeholder = G.placeholder((batch_size,) + tuple(image_shape), dtype='uint8', name='inputs')
    eval_images = G.to_float(G.pad_to_bounding_box(placeholder, 0, 0, input_height, input_width))
    mean_subtracted_inputs = model.preprocess(eval_images)
    output_dict = model.predict(mean_subtracted_inputs)
    padded_predictions = output_dict[model.main_class_predictions_key]
    padded_predictions = G.resize_bilinear(padded_predictions, (input_height, input_width),
                                           align_corners=True)
    eval_predictions = G.argmax(padded_predictions, 3)
    return eval_predictions, placeholder


def eval_segmentation_model_once(checkpoint_path, create_model_fn, create_input_fn,
                                 input_dimensions, eval_config, eval_dir,
                                 cropped_evaluation=False, image_summaries=False, verbose=False,
                                 sess_config=None, **kwargs):
    return eval_segmentation_model(
        create_model_fn, create_input_fn, input_dimensions, eval_config, train_dir=None,
        eval_dir=eval_dir, cropped_evaluation=cropped_evaluation,
        evaluate_single_checkpoint=checkpoint_path, image_summaries=image_summaries,
        verbose=verbose, sess_config=sess_config, **kwargs)


def eval_segmentation_model(create_model_fn, create_input_fn, input_dimensions, eval_config,
                            train_dir, eval_dir, cropped_evaluation=False,
                            evaluate_single_checkpoint=None, image_summaries=False,
                            verbose=False, sess_config=None, weights=None, dtype='mixed',
                            batch_size=1, return_confusion=False):
    import torch
    import torch.distributed as dist
    from fss import plan, metrics

    ignore_label = eval_config.ignore_label
    total_eval_examples = eval_config.num_examples
    input_height, input_width = input_dimensions
    rank, world = 0, 1
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()

    # example i (of the first num_examples) belongs to rank i % world.  An input function that
    # takes `shard` / `limit` (dataset_builder.build) skips the other ranks' records before
    # decoding them; any other iterable is decoded in full and sliced here.
    expected = len(range(rank, total_eval_examples, world))
    if _accepts(create_input_fn, ('shard', 'limit')):
        mine = create_evaluation_input(
            functools.partial(create_input_fn, shard=(rank, world), limit=total_eval_examples),
            input_height, input_width, cropped_evaluation)
    else:
        examples = create_evaluation_input(create_input_fn, input_height, input_width,
                                           cropped_evaluation)
        examples = itertools.islice(examples, total_eval_examples)
        mine = itertools.islice(examples, rank, None, world)
    first = next(mine, None)

    graph = G.reset_default_graph()
    num_classes, segmentation_model = create_model_fn()
    miou = metrics.StreamingMeanIoU(num_classes, ignore_label)
    count = 0
    if first is not None:
        predictions, placeholder = create_predictions_and_labels(
            segmentation_model, first[0].shape, input_height, input_width,
            cropped_evaluation, batch_size)
        if weights is None:
            from fss import checkpoint, frozen_graph
            path = evaluate_single_checkpoint or checkpoint.latest_checkpoint(train_dir)
            if path is None:
                raise ValueError('No checkpoints found in `train_dir`.')
            weights = frozen_graph.load_weights(path, graph.variables)
        cg = plan.CompiledGraph(graph, [predictions], weights, dtype=dtype)
        pred_val = cg.value(predictions)
        labels_dev = torch.zeros((batch_size, input_height, input_width), dtype=torch.uint8,
                                 device=cg.device)
        logging.info('Evaluating over %d samples...', total_eval_examples)
        batch = []
        for example in itertools.chain([first], mine):
            batch.append(example)
            if len(batch) < batch_size:
                continue
            count += _eval_batch(cg, miou, pred_val, labels_dev, batch, ignore_label)
            batch = []
            if verbose:
                print('EvalMetrics/mIoU - [%s]' % miou.result())
        if batch:       # ragged tail: pad the batch with all-ignore labels
            count += _eval_batch(cg, miou, pred_val, labels_dev, batch, ignore_label)
        cg.sync()
    if count < expected:
        # the reference's reader repeats the dataset; this one stops at its end
        logging.warning('rank %d evaluated %d examples, eval_config.num_examples asked for %d of them: '
                        'the input holds fewer records', rank, count, expected)
    miou.all_reduce()
    metric_results = miou.result()
    logging.info('Evaluation over. Eval values: %s', metric_results)
    if return_confusion:
        return metric_results, miou.confusion_matrix()
    return metric_results


def _eval_batch(cg, miou, pred_val, labels_dev, batch, ignore_label):
    import torch
    n = labels_dev.shape[0]
    images = np.zeros(cg.input_val.shape, np.uint8)
    labels = np.full(tuple(labels_dev.shape), ignore_label, np.uint8)
    for i, (image, label) in enumerate(batch):
        images[i] = image
        labels[i] = label
    cg.set_input(images)
    with torch.cuda.stream(cg.stream):
        labels_dev.copy_(torch.from_numpy(labels))
    cg.launch()
    miou.update(pred_val.buf.view(-1), labels_dev.view(-1), cg.sh)
    cg.sync()
    return len(batch)
