"""Throughput / latency of every shipped model configuration (synthetic weights).

usage: python tools/bench_configs.py [--dtype bfloat16|float32] [names...]
Prints one line per (config, batch): frames/s, ms/step, conv TFLOP/s and the
per-kind kernel time; optionally writes the per-layer table to --layers DIR.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'fast-semantic-segmentation_amd')
sys.path[:0] = [ROOT, PKG]

import numpy as np

CASES = {
    # name: (config file, input shape, pad shape, batches)
    'icnet10': ('single_stage_icnet_1.0_1025_resnet_v1.config', (1024, 2048, 3), (1025, 2049), (1, 16, 32, 64, 128, 192)),
    'pspnet713': ('pspnet_1.0_713_resnet_v1.config', (713, 713, 3), (713, 713), (1, 8, 16)),
    'icnet05': ('two_stage_icnet_0.5_1025_resnet_v1_stage_2.config', (1024, 2048, 3), (1025, 2049), (1, 32, 64)),
    'icnet_mnv2': ('single_stage_icnet_1.0_1025_mobilenet_v2.config', (1024, 2048, 3), (1025, 2049), (1, 16, 64)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('names', nargs='*', default=list(CASES))
    ap.add_argument('--dtype', default='bfloat16')
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--layers', default=None)
    ap.add_argument('--batches', default='', help='comma-separated batch sizes (default: the case\'s list)')
    ap.add_argument('--ksplit', action='store_true', help='split-K launches where they are faster (small batches)')
    args = ap.parse_args()
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import graph as G, synth, plan
    from libs import evaluator
    for name in args.names:
        cfg_file, in_shape, pad_shape, batches = CASES[name]
        cfg = pipeline_pb2.PipelineConfig()
        with open(os.path.join(PKG, 'configs', cfg_file)) as f:
            text_format.Merge(f.read(), cfg)
        if args.batches:
            batches = [int(b) for b in args.batches.split(',')]
        for b in batches:
            g = G.reset_default_graph()
            _, model = model_builder.build(cfg.model, is_training=False)
            preds, _ = evaluator.create_predictions_and_labels(model, in_shape, pad_shape[0],
                                                               pad_shape[1], False, b)
            istd = 0.55 if 'mobilenet' in cfg_file else 70.0
            weights = synth.make_weights(g.variables, input_std=istd)
            cg = plan.CompiledGraph(g, [preds], weights, dtype=args.dtype, ksplit=args.ksplit)
            cg.set_input(np.random.default_rng(0).integers(0, 256, (b,) + in_shape).astype(np.uint8))
            for _ in range(3):
                cg.launch()
            cg.sync()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                cg.launch()
            cg.sync()
            dt = (time.perf_counter() - t0) / args.steps
            lat = []
            for _ in range(20):
                t1 = time.perf_counter()
                cg.launch()
                cg.sync()
                lat.append(time.perf_counter() - t1)
            prof = cg.profile()
            kinds = {}
            for p in prof:
                kinds[p['kind']] = kinds.get(p['kind'], 0.0) + p['ms']
            conv_ms = kinds.get('conv', 0.0)
            conv_fl = sum(p['flops'] for p in prof if p['kind'] == 'conv')
            print('%-11s %-8s batch %2d: %8.1f frames/s  %7.3f ms/step  median launch+sync %7.3f ms  '
                  'conv %6.1f TFLOP/s (%.1f GFLOP/frame)  kernels: %s' %
                  (name, args.dtype, b, b / dt, dt * 1e3, float(np.median(lat)) * 1e3,
                   conv_fl / conv_ms / 1e9 if conv_ms else 0, conv_fl / b / 1e9,
                   ' '.join('%s=%.3f' % kv for kv in sorted(kinds.items(), key=lambda kv: -kv[1]))),
                  flush=True)
            if args.layers:
                os.makedirs(args.layers, exist_ok=True)
                with open(os.path.join(args.layers, '%s_%s_b%d.txt' % (name, args.dtype, b)), 'w') as f:
                    for p in prof:
                        tf = p['flops'] / (p['ms'] * 1e-3) / 1e12 if p['ms'] > 0 else 0
                        gb = p['bytes'] / (p['ms'] * 1e-3) / 1e9 if p['ms'] > 0 else 0
                        f.write('%-10s %-70s %8.4f ms %8.1f TFLOP/s %8.1f GB/s cfg=%s\n'
                                % (p['kind'], p['name'], p['ms'], tf, gb, p.get('config')))
            del cg


if __name__ == '__main__':
    main()
