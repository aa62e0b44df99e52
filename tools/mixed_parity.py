"""Label agreement / logit error of the compute modes on the small model cases of
tests/test_models_gpu.py and (optionally) one full-size ICNet frame.

    python tools/mixed_parity.py [--full] [--dtypes mixed,float32x3] [--half-scopes re1,re2]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd'), os.path.join(ROOT, 'tests')]
import numpy as np

from test_models_gpu import build, oracle_logits, CASES
from fss import plan
from oracle import ops as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--full', action='store_true')
    ap.add_argument('--dtypes', default='mixed')
    ap.add_argument('--half-scopes', default=None)
    ap.add_argument('--cases', default='0,1,2,3,4')
    args = ap.parse_args()
    scopes = args.half_scopes.split(',') if args.half_scopes else None
    cases = [CASES[int(i)] for i in args.cases.split(',') if i != '']
    if args.full:
        cases += [('icnet', 'dilated_resnet50', 1.0, [1.0, 2.0, 3.0, 4.0], (1, 1024, 2048, 3), (1025, 2049)),
                  ('icnet', 'dilated_resnet50', 0.5, [1.0, 2.0, 3.0, 4.0], (1, 1024, 2048, 3), (1025, 2049))]
    for arch, ext, fs, pf, in_shape, pad in cases:
        g, cfg, outs, weights = build(arch, ext, fs, pf, in_shape, pad)
        img = np.random.default_rng(0).integers(0, 256, in_shape).astype(np.uint8)
        ref = oracle_logits(arch, ext, fs, pf, img, pad, weights)
        scale = float(np.abs(ref).max())
        lab = O.argmax_channels(ref)
        for dtype in args.dtypes.split(','):
            kw = dict(half_scopes=scopes) if dtype == 'mixed' and scopes else {}
            cg = plan.CompiledGraph(g, outs, weights, dtype=dtype, commute_zoom=True, commute_pyramid=True, **kw)
            logits, labels = cg.run(img)
            n16 = sum(1 for v in cg.vals.values() if v.dt == 'f16')
            print('%-7s %-18s %.1f %-12s %-10s logit err %.3g of scale, labels %.5f (%d half tensors)' % (
                arch, ext, fs, in_shape[1:3], dtype, np.abs(logits - ref).max() / scale,
                (labels[..., 0] == lab).mean(), n16), flush=True)
            del cg


if __name__ == '__main__':
    main()
