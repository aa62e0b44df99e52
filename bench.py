"""Headline benchmark: ICNet-1.0 (dilated ResNet-50) forward pass + mIoU
accumulation on synthetic 1025x2049 Cityscapes-shaped frames, frames/s.

    python bench.py --gpus N --steps K --warmup W          (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the hot path over one batch of `--batch` frames that
already sit in HBM: uint8 frames -> preprocess -> ICNet forward -> x4 zoom +
argmax -> confusion-matrix accumulation against synthetic labels.  Every rank
runs its own shard of frames (weak scaling, no data-path collective); the int64
confusion matrix is all-reduced once, inside the timed region, after the last
step.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, 'fast-semantic-segmentation_amd')
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

CONFIG = 'single_stage_icnet_1.0_1025_resnet_v1.config'
INPUT_SHAPE = (1024, 2048, 3)
PAD_SHAPE = (1025, 2049)
# dense peaks (MI355X): bf16 MFMA ~2.5 PFLOP/s, f32 MFMA 157.3 TFLOP/s
PEAK_TFLOPS = {'bfloat16': 2500.0, 'float32': 157.3}


def measured_conv_traffic(batch, n_conv, dtype):
    """HBM bytes per conv launch from the committed PMC passes (TCC FETCH_SIZE /
    WRITE_SIZE collected by rocprofv3 in separate runs of this script, with the
    gfx950 fetch correction already applied); None when no measurement of this
    dtype is on disk.  Counters cannot be read from inside the process."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', '*conv_traffic*.json'))):
        with open(path) as f:
            d = json.load(f)
        if d.get('dtype', 'bfloat16') == dtype:
            best = d
    if best is None:
        return None
    per_frame = best['conv_hbm_bytes_per_step'] / best['frames_per_step']
    return per_frame * batch / n_conv


def build_plan(batch, dtype, with_labels=True):
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import graph as G, synth, plan
    from libs import evaluator
    cfg = pipeline_pb2.PipelineConfig()
    with open(os.path.join(PKG, 'configs', CONFIG)) as f:
        text_format.Merge(f.read(), cfg)
    g = G.reset_default_graph()
    num_classes, model = model_builder.build(cfg.model, is_training=False)
    predictions, placeholder = evaluator.create_predictions_and_labels(
        model, INPUT_SHAPE, PAD_SHAPE[0], PAD_SHAPE[1], False, batch)
    weights = synth.make_weights(g.variables, seed=1234)
    cg = plan.CompiledGraph(g, [predictions], weights, dtype=dtype)
    return cg, predictions, num_classes, weights


def cpu_baseline(weights):
    """The numpy oracle (a CPU port of the reference graph) on one full-size
    frame, timed on this machine's host cores."""
    import numpy as np
    from oracle import nets, ops as O
    threads = int(os.environ.get('OMP_NUM_THREADS', os.cpu_count() or 1))
    threads = max(1, min(threads, os.cpu_count() or 1))
    img = np.random.default_rng(100).integers(0, 256, (1,) + INPUT_SHAPE).astype(np.uint8)
    t0 = time.perf_counter()
    x = nets.preprocess(O.pad_to_bounding_box(img.astype(np.float32), *PAD_SHAPE),
                        'dilated_resnet50')
    logits = nets.icnet_predict(nets.Params(weights), x)
    O.argmax_channels(logits)
    dt = time.perf_counter() - t0
    return {'value': 1.0 / dt, 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': '1 frame 1025x2049 ICNet-1.0 float32, numpy/OpenBLAS oracle, %.1f s' % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='frames per step per GPU')
    ap.add_argument('--dtype', default='bfloat16', choices=['bfloat16', 'float32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-latency', action='store_true')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus:
        raise SystemExit('--gpus %d does not match WORLD_SIZE %d' % (args.gpus, world))
    # rehearsal knobs (several ranks on one GPU box): FSS_BENCH_BACKEND=gloo FSS_BENCH_DEVICE=0
    backend = os.environ.get('FSS_BENCH_BACKEND', 'nccl')
    if 'FSS_BENCH_DEVICE' in os.environ:
        local_rank = int(os.environ['FSS_BENCH_DEVICE'])
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    from fss import metrics
    B = args.batch
    cg, predictions, num_classes, weights = build_plan(B, args.dtype)
    pred_val = cg.value(predictions)
    miou = metrics.StreamingMeanIoU(num_classes, 255)

    # synthetic frames + labels of this rank's shard, resident in HBM before timing
    rng = np.random.default_rng(100 + rank)
    cg.set_input(rng.integers(0, 256, (B,) + INPUT_SHAPE).astype(np.uint8))
    labels = rng.integers(0, num_classes, (B,) + PAD_SHAPE).astype(np.uint8)
    labels[rng.random(labels.shape) < 0.05] = 255
    labels[:, PAD_SHAPE[0] - 1, :] = 0        # zero-padded label row / column = class 0
    labels[:, :, PAD_SHAPE[1] - 1] = 0
    with torch.cuda.stream(cg.stream):
        labels_dev = torch.from_numpy(labels).to(cg.device)
    cg.sync()

    def step():
        cg.launch()
        miou.update(pred_val.buf.view(-1), labels_dev.view(-1), cg.sh)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    cg.sync()
    with torch.cuda.stream(cg.stream):
        miou.all_reduce()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = world * B * args.steps

    out = None
    if rank == 0:
        # live per-kernel timing (HIP events on the plan's stream) for the roofline
        prof = cg.profile(repeats=3)
        conv_ms = sum(p['ms'] for p in prof if p['kind'] == 'conv')
        conv_flops = sum(p['flops'] for p in prof if p['kind'] == 'conv')
        graph_flops = sum(p['graph_flops'] for p in prof if p['kind'] == 'conv')
        n_conv = sum(1 for p in prof if p['kind'] == 'conv')
        total_ms = sum(p['ms'] for p in prof)
        other_bytes = sum(p['bytes'] for p in prof if p['kind'] != 'conv')
        other_ms = total_ms - conv_ms
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.dtype]
        out = {
            'metric': 'ICNet 1025x2049 frames/sec',
            'value': frames / elapsed,
            'unit': 'frames/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'bf16' if args.dtype == 'bfloat16' else 'f32',
            'data': 'synthetic',
            'config': {'workload': 'ICNet-1.0 dilated-ResNet50 1025x2049 forward + argmax + '
                                   'confusion matrix (single_stage_icnet_1.0_1025_resnet_v1)',
                       'frames_per_step_per_gpu': B, 'weights': 'synthetic seed 1234',
                       'parallelism': 'frames sharded over %d GPU(s)' % world},
            'roofline': {'bound': 'mfma', 'kernel': 'conv_igemm_kernel',
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak,
                         'traffic': measured_conv_traffic(B, n_conv, args.dtype),
                         'algorithmic_bytes': sum(p['bytes'] for p in prof if p['kind'] == 'conv') / n_conv,
                         'launches_per_step': n_conv,
                         'avg_launch_ms': conv_ms / n_conv,
                         'algorithmic_gflop_per_frame': conv_flops / B / 1e9,
                         # the same launches priced with the FLOP of the graph as written (the
                         # fusion convolutions run on a quarter of the pixels after a rewrite)
                         'graph_gflop_per_frame': graph_flops / B / 1e9,
                         'graph_tflops': graph_flops / (conv_ms * 1e-3) / 1e12},
            'hbm_kernels': {'ms_per_step': other_ms,
                            'achieved_gbps': other_bytes / (other_ms * 1e-3) / 1e9 if other_ms > 0 else None,
                            'peak_gbps': 8000.0},
            'kernel_ms_per_step': total_ms,
        }
        kinds = {}
        for p in prof:
            kinds[p['kind']] = kinds.get(p['kind'], 0.0) + p['ms']
        out['ms_by_kind'] = {k: round(v, 4) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])}
        if os.environ.get('FSS_BENCH_LAYERS'):
            with open(os.environ['FSS_BENCH_LAYERS'], 'w') as f:
                for p in prof:
                    tf = p['flops'] / (p['ms'] * 1e-3) / 1e12 if p['ms'] > 0 else 0
                    gb = p['bytes'] / (p['ms'] * 1e-3) / 1e9 if p['ms'] > 0 else 0
                    f.write('%-10s %-70s %8.4f ms %8.1f TFLOP/s %8.1f GB/s cfg=%s\n'
                            % (p['kind'], p['name'], p['ms'], tf, gb, p.get('config')))
    del cg
    if rank == 0 and world == 1 and not args.no_latency:
        # bs=1 latency of the same model (median wall time of a launch + sync)
        cg1, pred1, _, _ = build_plan(1, args.dtype)
        cg1.set_input(np.random.default_rng(0).integers(0, 256, (1,) + INPUT_SHAPE).astype(np.uint8))
        for _ in range(10):
            cg1.launch()
        cg1.sync()
        ts = []
        for _ in range(50):
            t1 = time.perf_counter()
            cg1.launch()
            cg1.sync()
            ts.append(time.perf_counter() - t1)
        out['bs1_ms'] = float(np.median(ts) * 1e3)
        del cg1
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(weights)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
