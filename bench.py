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
# dense peaks (MI355X): bf16 MFMA ~2.5 PFLOP/s, f32 MFMA 157.3 TFLOP/s.  float32x3 issues three
# bf16 MFMAs per product: its convolution FLOP are priced x3 against the bf16 peak.
PEAK_TFLOPS = {'bfloat16': 2500.0, 'float32': 157.3, 'float32x3': 2500.0, 'mixed': 2500.0}
MFMA_PER_PRODUCT = {'bfloat16': 1, 'float32': 1, 'float32x3': 3, 'mixed': 1}
DTYPE_TAG = {'bfloat16': 'bf16', 'float32': 'f32', 'float32x3': 'f32 (bf16x3 products)', 'mixed': 'f32 (bf16x3 products) + f16'}


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


def build_plan(batch, dtype, with_labels=True, config=CONFIG, streams=2, ksplit=False):
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import graph as G, synth, plan
    from libs import evaluator
    cfg = pipeline_pb2.PipelineConfig()
    with open(os.path.join(PKG, 'configs', config)) as f:
        text_format.Merge(f.read(), cfg)
    g = G.reset_default_graph()
    num_classes, model = model_builder.build(cfg.model, is_training=False)
    predictions, placeholder = evaluator.create_predictions_and_labels(
        model, INPUT_SHAPE, PAD_SHAPE[0], PAD_SHAPE[1], False, batch)
    weights = synth.make_weights(g.variables, seed=1234)
    cg = plan.CompiledGraph(g, [predictions], weights, dtype=dtype, streams=streams, ksplit=ksplit)
    return cg, predictions, num_classes, weights


def _blas_threads():
    """Threads the BLAS behind numpy will actually use (not the machine's core count)."""
    try:
        from threadpoolctl import threadpool_info
        n = [d.get('num_threads') for d in threadpool_info() if d.get('user_api') == 'blas']
        if n:
            return int(max(n))
    except Exception:
        pass
    return int(os.environ.get('OMP_NUM_THREADS', 1))


def cpu_baseline(weights, runs=3):
    """The numpy oracle (a CPU port of the reference graph) on one full-size frame, timed on
    this machine's host cores: median of `runs`.  Also returns the oracle's label map of that
    frame, which the parity figures of the bench line are measured against."""
    import numpy as np
    from oracle import nets, ops as O
    img = np.random.default_rng(100).integers(0, 256, (1,) + INPUT_SHAPE).astype(np.uint8)
    ts, labels = [], None
    for _ in range(runs):
        t0 = time.perf_counter()
        x = nets.preprocess(O.pad_to_bounding_box(img.astype(np.float32), *PAD_SHAPE),
                            'dilated_resnet50')
        logits = nets.icnet_predict(nets.Params(weights), x)
        labels = O.argmax_channels(logits)
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    return ({'value': 1.0 / dt, 'unit': 'frames/s', 'cores': _blas_threads(), 'kind': 'port',
             'sample': '1 frame 1025x2049 ICNet-1.0 float32, numpy/OpenBLAS oracle, median of %d runs, %.1f s'
                       % (runs, dt)}, img, labels)


def timed_steps(step, sync, barrier, warmup, steps):
    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    return t0


def host_fed(cg, pred_val, steps):
    """Frames/s when every batch comes from (pinned) host memory and every label map goes back
    to it, as the reference's timer around sess.run sees it (inference.py:96-101), minus image
    decoding: H2D of batch k+1 and D2H of the label maps of batch k-1 run on two copy streams
    beside the forward pass of batch k.  Output = uint8 label maps, not the reference's float
    colour map (12x the bytes)."""
    import numpy as np
    import torch
    B = cg.input_val.shape[0]
    rng = np.random.default_rng(7)
    pin_in = [torch.from_numpy(rng.integers(0, 256, tuple(cg.input_val.buf.shape)).astype(np.uint8)).pin_memory()
              for _ in range(2)]
    pin_out = [torch.empty(tuple(pred_val.buf.shape), dtype=torch.uint8).pin_memory() for _ in range(2)]
    stage_in = torch.empty_like(cg.input_val.buf)
    stage_out = torch.empty_like(pred_val.buf)
    s_in, s_out = torch.cuda.Stream(device=cg.device), torch.cuda.Stream(device=cg.device)
    ev_h2d, ev_in_free = torch.cuda.Event(), torch.cuda.Event()
    ev_out_ready, ev_out_free = torch.cuda.Event(), torch.cuda.Event()

    def run(n):
        with torch.cuda.stream(s_in):
            stage_in.copy_(pin_in[0], non_blocking=True)
            ev_h2d.record(s_in)
        for k in range(n):
            with torch.cuda.stream(cg.stream):
                cg.stream.wait_event(ev_h2d)
                cg.input_val.buf.copy_(stage_in)
                ev_in_free.record(cg.stream)
            if k + 1 < n:
                with torch.cuda.stream(s_in):
                    s_in.wait_event(ev_in_free)
                    stage_in.copy_(pin_in[(k + 1) % 2], non_blocking=True)
                    ev_h2d.record(s_in)
            cg.launch()
            with torch.cuda.stream(cg.stream):
                if k > 0:
                    cg.stream.wait_event(ev_out_free)
                stage_out.copy_(pred_val.buf)
                ev_out_ready.record(cg.stream)
            with torch.cuda.stream(s_out):
                s_out.wait_event(ev_out_ready)
                pin_out[k % 2].copy_(stage_out, non_blocking=True)
                ev_out_free.record(s_out)
        torch.cuda.synchronize()

    run(3)
    t0 = time.perf_counter()
    run(steps)
    dt = time.perf_counter() - t0
    return {'value': B * steps / dt, 'unit': 'frames/s', 'ms_per_step': dt / steps * 1e3,
            'in': 'pinned host uint8 frames', 'out': 'uint8 label maps to pinned host memory',
            'overlap': 'H2D / D2H on two copy streams beside the forward pass'}


def optional_leg(fn):
    """A side figure of the bench line: its value, or the error that kept it from being measured
    (the headline figure is printed either way)."""
    try:
        return fn()
    except Exception as e:
        return {'error': '%s: %s' % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=128, help='frames per step per GPU')
    ap.add_argument('--dtype', default='bfloat16', choices=['bfloat16', 'float32', 'float32x3', 'mixed'])
    ap.add_argument('--no-conformant', action='store_true',
                    help='skip the float32x3 (all tolerances met) figure and the parity figures')
    ap.add_argument('--no-host-fed', action='store_true')
    ap.add_argument('--streams', type=int, default=2, help='1: single-stream graph (profiling)')
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
    cg, predictions, num_classes, weights = build_plan(B, args.dtype, streams=args.streams)
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
        # live per-kernel timing (HIP events on the plan's stream, mean of 5 passes) for the roofline
        prof = cg.profile(repeats=5)
        mpp = MFMA_PER_PRODUCT[args.dtype]
        conv_ms = sum(p['mean_ms'] for p in prof if p['kind'] == 'conv')
        conv_flops = sum(p['flops'] for p in prof if p['kind'] == 'conv')
        graph_flops = sum(p['graph_flops'] for p in prof if p['kind'] == 'conv')
        n_conv = sum(1 for p in prof if p['kind'] == 'conv')
        total_ms = sum(p['mean_ms'] for p in prof)
        other_bytes = sum(p['bytes'] for p in prof if p['kind'] != 'conv')
        other_ms = total_ms - conv_ms
        achieved = mpp * conv_flops / (conv_ms * 1e-3) / 1e12
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
            'dtype': DTYPE_TAG[args.dtype],
            'data': 'synthetic',
            'config': {'workload': 'ICNet-1.0 dilated-ResNet50 1025x2049 forward + argmax + '
                                   'confusion matrix (single_stage_icnet_1.0_1025_resnet_v1)',
                       'frames_per_step_per_gpu': B, 'weights': 'synthetic seed 1234',
                       'parallelism': 'frames sharded over %d GPU(s)' % world,
                       'streams': cg.stream_choice},
            'roofline': {'bound': 'mfma', 'kernel': 'conv_igemm_kernel (+ conv8_kernel, its 8-phase form)',
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s',
                         'frac': achieved / peak,
                         'mfma_per_product': mpp,
                         'traffic': measured_conv_traffic(B, n_conv, args.dtype),
                         'algorithmic_bytes': sum(p['bytes'] for p in prof if p['kind'] == 'conv') / n_conv,
                         'launches_per_step': n_conv,
                         'avg_launch_ms': conv_ms / n_conv,
                         'avg_launch_ms_best_of_5': sum(p['ms'] for p in prof if p['kind'] == 'conv') / n_conv,
                         'algorithmic_gflop_per_frame': conv_flops / B / 1e9,
                         # the same launches priced with the FLOP of the graph as written (the
                         # fusion convolutions run on a quarter of the pixels after a rewrite)
                         'graph_gflop_per_frame': graph_flops / B / 1e9,
                         'graph_tflops': mpp * graph_flops / (conv_ms * 1e-3) / 1e12},
            'hbm_kernels': {'ms_per_step': other_ms,
                            'achieved_gbps': other_bytes / (other_ms * 1e-3) / 1e9 if other_ms > 0 else None,
                            'peak_gbps': 8000.0},
            'kernel_ms_per_step': total_ms,
        }
        kinds = {}
        for p in prof:
            kinds[p['kind']] = kinds.get(p['kind'], 0.0) + p['mean_ms']
        out['ms_by_kind'] = {k: round(v, 4) for k, v in sorted(kinds.items(), key=lambda kv: -kv[1])}
        if os.environ.get('FSS_BENCH_LAYERS'):
            with open(os.environ['FSS_BENCH_LAYERS'], 'w') as f:
                for p in prof:
                    tf = p['flops'] / (p['mean_ms'] * 1e-3) / 1e12 if p['mean_ms'] > 0 else 0
                    gb = p['bytes'] / (p['mean_ms'] * 1e-3) / 1e9 if p['mean_ms'] > 0 else 0
                    f.write('%-10s %-70s %8.4f ms %8.1f TFLOP/s %8.1f GB/s cfg=%s\n'
                            % (p['kind'], p['name'], p['mean_ms'], tf, gb, p.get('config')))
        if world == 1 and not args.no_host_fed:
            out['host_fed'] = optional_leg(lambda: host_fed(cg, pred_val, args.steps))
    del cg, pred_val, labels_dev
    torch.cuda.empty_cache()
    solo = rank == 0 and world == 1
    oracle_img = oracle_labels = None
    if solo and not args.no_cpu_baseline:
        try:
            out['cpu_baseline'], oracle_img, oracle_labels = cpu_baseline(weights)
        except Exception as e:      # the headline line is printed whatever happens to a side figure
            out['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, e)}

    def bs1(dtype):
        """bs=1 latency (median wall time of launch + sync) and, when the oracle's label map of
        the CPU baseline frame is at hand, the share of pixels on which the plan agrees with it."""
        cg1, pred1, _, _ = build_plan(1, dtype, ksplit=True)     # the low-latency plan
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
        agree = None
        if oracle_labels is not None:
            lab = cg1.run(oracle_img, [pred1])[0]
            agree = float((lab[..., 0] == oracle_labels).mean())
        del cg1
        return float(np.median(ts) * 1e3), agree

    if solo and not args.no_latency:
        try:
            out['bs1_ms'], agree = bs1(args.dtype)
            # argmax agreement with the float32 CPU oracle on one full-size frame (synthetic weights
            # whose classifier is zero-mean: near-ties are far denser than in a trained network)
            out['parity'] = {'argmax_agreement_vs_cpu_oracle': agree, 'required': 0.999,
                             'frame': '1 x 1025x2049, seed 100'}
        except Exception as e:
            out['bs1_ms'] = None
            out['parity'] = {'error': '%s: %s' % (type(e).__name__, e)}

    def conformant_leg():
            # the compute mode that meets every stated tolerance (logits 1e-3, argmax 99.9 %), timed
            # the same way as the headline: same frames per step, same step, same number of steps
            cgc, predc, _, _ = build_plan(B, 'float32x3')
            pv = cgc.value(predc)
            miou_c = metrics.StreamingMeanIoU(num_classes, 255)
            rng = np.random.default_rng(100)
            cgc.set_input(rng.integers(0, 256, (B,) + INPUT_SHAPE).astype(np.uint8))
            with torch.cuda.stream(cgc.stream):
                lab_c = torch.from_numpy(labels).to(cgc.device)
            cgc.sync()

            def step_c():
                cgc.launch()
                miou_c.update(pv.buf.view(-1), lab_c.view(-1), cgc.sh)

            t0 = timed_steps(step_c, cgc.sync, torch.cuda.synchronize, args.warmup, args.steps)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            profc = cgc.profile(repeats=3)
            cms = sum(p['mean_ms'] for p in profc if p['kind'] == 'conv')
            cfl = sum(p['flops'] for p in profc if p['kind'] == 'conv')
            del cgc, pv, lab_c
            torch.cuda.empty_cache()
            lat, agree = bs1('float32x3') if not args.no_latency else (None, None)
            return {
                'dtype': DTYPE_TAG['float32x3'], 'value': B * args.steps / el, 'unit': 'frames/s',
                'ms_per_step': el / args.steps * 1e3, 'steps': args.steps, 'frames_per_step': B,
                'conv_mfma_tflops': 3 * cfl / (cms * 1e-3) / 1e12, 'conv_frac_of_bf16_peak': 3 * cfl / (cms * 1e-3) / 1e12 / 2500.0,
                'bs1_ms': lat, 'argmax_agreement_vs_cpu_oracle': agree,
                'note': 'float32 tensors, convolution products as three bf16 MFMAs of hi/lo splits'}
    if solo and not args.no_conformant and args.dtype == 'bfloat16':
        out['conformant'] = optional_leg(conformant_leg)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
