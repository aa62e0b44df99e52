"""For the heaviest convolution launches of the benchmarked plan: every tile configuration of the
plan's MFMA shape timed IN the step (all launches in order, HIP events between them), against the
configuration the shipped table names (which was timed as repeated launches of the one kernel).
    python tools/tune_insitu.py [batch] [dtype] [min_ms]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd')]
import bench                                   # noqa: E402
from fss import _lib                           # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    dtype = sys.argv[2] if len(sys.argv) > 2 else 'mixed'
    min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.8
    cg, _, _, _ = bench.build_plan(batch, dtype)
    info = (ctypes.c_int * 7)()
    cands = []
    for cfg in range(1, _lib.lib.fss_conv_config_count() + 1):
        _lib.call('fss_conv_config_info', cfg, info)
        if info[6] == cg.mfma_edge:
            cands.append(cfg)
    base = cg.profile(repeats=5)
    print('step %.3f ms' % sum(p['mean_ms'] for p in base), flush=True)
    gain = 0.0
    for k, s in enumerate(cg.steps):
        if s.kind != 'conv' or base[k]['ms'] < min_ms or cg.step_fname[k] == 'fss_conv2d_splitk_nhwc':
            continue
        d = s.keep[0] if isinstance(s.keep, tuple) else s.keep
        shipped = d.config
        res = {}
        for cfg in cands:
            d.config = cfg
            if s.fn(*s.args) != 0:
                continue
            _lib.call('fss_device_sync')
            res[cfg] = cg.profile(repeats=3)[k]['ms']
        d.config = shipped
        best = min(res, key=res.get)
        top = sorted(res.items(), key=lambda kv: kv[1])[:4]
        print('%-78s shipped %2d %.3f  best %2d %.3f  %s' % (
            s.name[-78:], shipped, res.get(shipped, float('nan')), best, res[best],
            ' '.join('%d:%.3f' % kv for kv in top)), flush=True)
        gain += res.get(shipped, res[best]) - res[best]
    print('sum of (shipped - best): %.3f ms' % gain)


if __name__ == '__main__':
    main()
