"""Times the conv tile configurations for every layer shape of the shipped model
configurations and writes the winners to a table the plan compiler loads at import
(fast-semantic-segmentation_amd/fss/conv_tuned.json).

usage (on a GPU box): python tools/tune_convs.py [--out gpurun_out/conv_tuned.json]
"""
import argparse
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'fast-semantic-segmentation_amd')
sys.path[:0] = [ROOT, PKG, os.path.join(ROOT, 'tools')]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'conv_tuned.json'))
    ap.add_argument('--fresh', action='store_true', help='ignore the shipped table')
    ap.add_argument('--drop', default='', help='time again the shapes whose table key contains this text')
    ap.add_argument('--dtypes', default='bfloat16,float32,float32x3,mixed',
                    help='compute modes whose layer shapes are tuned')
    args = ap.parse_args()
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import graph as G, synth, plan
    from libs import evaluator
    from bench_configs import CASES
    import torch
    if args.fresh:
        plan.CompiledGraph._TUNE_CACHE = {}
    if args.drop:
        plan.CompiledGraph._tune_load()
        for key in [k for k in plan.CompiledGraph._TUNE_CACHE if args.drop in k]:
            del plan.CompiledGraph._TUNE_CACHE[key]
    jobs = []
    for name, (cfg_file, in_shape, pad_shape, batches) in CASES.items():
        for dtype, edges in (('bfloat16', (16, 32)), ('float32', (32, 16)), ('float32x3', (16, 32)),
                             ('mixed', (16,))):
            if dtype not in args.dtypes.split(','):
                continue
            for b in batches:
                if dtype == 'float32' and name not in ('icnet10',):      # float32x3: every model
                    continue
                for e in edges:
                    jobs.append((name, cfg_file, in_shape, pad_shape, b, dtype, e))
    # the smoke-test / small shapes
    for name, cfg_file, in_shape, pad_shape, b, dtype, e in jobs:
        cfg = pipeline_pb2.PipelineConfig()
        with open(os.path.join(PKG, 'configs', cfg_file)) as f:
            text_format.Merge(f.read(), cfg)
        g = G.reset_default_graph()
        _, model = model_builder.build(cfg.model, is_training=False)
        preds, _ = evaluator.create_predictions_and_labels(model, in_shape, pad_shape[0], pad_shape[1],
                                                           False, b)
        weights = synth.make_weights(g.variables, input_std=0.55 if 'mobilenet' in cfg_file else 70.0)
        for ksplit in ((False, True) if b == 1 else (False,)):       # one frame: the low-latency plan too
            cg = plan.CompiledGraph(g, [preds], weights, dtype=dtype, mfma_edge=e, use_hip_graph=False,
                                    ksplit=ksplit)
            print('%s %s b%d edge %d%s: %d shapes timed' % (name, dtype, b, e, ' split-K' if ksplit else '',
                                                            cg.tuned_now), flush=True)
            del cg
            gc.collect()
            torch.cuda.empty_cache()
        plan.CompiledGraph.save_tuning(args.out)
    print('wrote', args.out, len(plan.CompiledGraph._TUNE_CACHE), 'entries')


if __name__ == '__main__':
    main()
