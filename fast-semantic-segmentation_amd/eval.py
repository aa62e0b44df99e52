r"""mIoU evaluation of checkpoints (MI355X path).

Same flags as the reference's eval.py:25-53 (--evaluate_all_from_checkpoint,
--train_dir, --eval_dir, --config_path, --image_summaries, --verbose,
--limit_gpu_mem) plus --dtype / --batch_size.  Launch one process per GPU with
torch.distributed.run to shard the examples; the confusion matrix is all-reduced
over RCCL.  Rank 0 writes <eval_dir>/eval_results.json: per checkpoint the mIoU and
the int64 confusion matrix.  Rehearsal knobs for several ranks on one GPU:
FSS_DIST_BACKEND=gloo FSS_DEVICE=0.
"""
import argparse
import functools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def get_checkpoints_from_path(initial_checkpoint_path, checkpoint_dir):
    from fss import checkpoint
    state = checkpoint.get_checkpoint_state(checkpoint_dir)
    if state is None:
        raise ValueError('No checkpoints found in `train_dir`.')
    all_checkpoints = state[1]
    checkpoints_to_evaluate = None
    for idx, ckpt in enumerate(all_checkpoints):
        print(idx, ' ', str(ckpt))
        full_init_ckpt_path = os.path.join(os.path.dirname(ckpt), initial_checkpoint_path)
        if str(ckpt) == full_init_ckpt_path:
            checkpoints_to_evaluate = all_checkpoints[idx:]
            break
    if checkpoints_to_evaluate is None:
        raise ValueError('Checkpoint not found. Exiting.')
    return checkpoints_to_evaluate


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--evaluate_all_from_checkpoint', default=None)
    ap.add_argument('--train_dir', required=True)
    ap.add_argument('--eval_dir', required=True)
    ap.add_argument('--config_path', required=True)
    ap.add_argument('--image_summaries', action='store_true')
    ap.add_argument('--verbose', action='store_true')
    ap.add_argument('--limit_gpu_mem', action='store_true')
    ap.add_argument('--dtype', default='mixed', choices=['mixed', 'float32x3', 'bfloat16', 'float32'],
                    help='mixed (default): float32 tensors with products as three bf16 MFMAs, the 3x3 convolutions of the '
                         'bottleneck units and the full-resolution branch in half precision - meets the float32 tolerances '
                         '(logits 1e-3, labels 99.9 %%); float32x3: the same without half precision, the results of a float32 '
                         'framework at 2/3 of the speed; bfloat16: twice the throughput, labels differ on 0.1 - 2 %% of the pixels')
    ap.add_argument('--batch_size', type=int, default=1)
    flags = ap.parse_args(argv)

    import torch
    import torch.distributed as dist
    from google.protobuf import text_format
    from builders import dataset_builder, model_builder
    from protos import pipeline_pb2
    from libs.evaluator import eval_segmentation_model, eval_segmentation_model_once

    device = int(os.environ.get('FSS_DEVICE', os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(device)
    if 'WORLD_SIZE' in os.environ and int(os.environ['WORLD_SIZE']) > 1:
        backend = os.environ.get('FSS_DIST_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', device))
        else:
            dist.init_process_group(backend)
    rank0 = not dist.is_initialized() or dist.get_rank() == 0
    results = []
    os.makedirs(flags.eval_dir, exist_ok=True)
    if not os.path.isdir(flags.train_dir):
        raise ValueError('`train_dir` must be a valid directory containing model checkpoints '
                         'from training.')
    pipeline_config = pipeline_pb2.PipelineConfig()
    with open(flags.config_path) as f:
        text_format.Merge(f.read(), pipeline_config)
    eval_config = pipeline_config.eval_config
    create_input_fn = functools.partial(dataset_builder.build,
                                        input_reader_config=pipeline_config.eval_input_reader)
    create_model_fn = functools.partial(model_builder.build, model_config=pipeline_config.model,
                                        is_training=False)
    input_type = eval_config.eval_input_type.WhichOneof('eval_input_type_oneof')
    if input_type == 'cropped_eval_input':
        c = eval_config.eval_input_type.cropped_eval_input
        input_dims, cropped_evaluation = (c.height, c.width), True
    elif input_type == 'padded_eval_input':
        p = eval_config.eval_input_type.padded_eval_input
        input_dims, cropped_evaluation = (p.height, p.width), False
    else:
        raise ValueError('Must specify an `eval_input_type` for evaluation.')
    extra = dict(dtype=flags.dtype, batch_size=flags.batch_size, return_confusion=True)
    if flags.evaluate_all_from_checkpoint is not None:
        for curr_checkpoint in get_checkpoints_from_path(flags.evaluate_all_from_checkpoint,
                                                         flags.train_dir):
            result, cm = eval_segmentation_model_once(
                curr_checkpoint, create_model_fn, create_input_fn, input_dims, eval_config,
                eval_dir=flags.eval_dir, cropped_evaluation=cropped_evaluation,
                image_summaries=flags.image_summaries, verbose=flags.verbose, **extra)
            results.append(dict(checkpoint=str(curr_checkpoint), mean_iou=float(result),
                                confusion_matrix=cm.tolist()))
            if rank0:
                print('Evaluation of `{}` over. Eval values: {}'.format(curr_checkpoint, result))
    else:
        result, cm = eval_segmentation_model(
            create_model_fn, create_input_fn, input_dims, eval_config, train_dir=flags.train_dir,
            eval_dir=flags.eval_dir, cropped_evaluation=cropped_evaluation,
            evaluate_single_checkpoint=flags.evaluate_all_from_checkpoint,
            image_summaries=flags.image_summaries, verbose=flags.verbose, **extra)
        from fss import checkpoint
        results.append(dict(checkpoint=str(checkpoint.latest_checkpoint(flags.train_dir)),
                            mean_iou=float(result), confusion_matrix=cm.tolist()))
        if rank0:
            print('Evaluation over. Eval values: {}'.format(result))
    if rank0:
        import json
        with open(os.path.join(flags.eval_dir, 'eval_results.json'), 'w') as f:
            json.dump(dict(dtype=flags.dtype, world_size=dist.get_world_size() if dist.is_initialized() else 1,
                           results=results), f)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
