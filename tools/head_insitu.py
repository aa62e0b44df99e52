"""The classifier launch of the benchmarked plan timed IN the step (every launch of the step in order,
HIP events between them) for each tile configuration that takes it, next to the time of the whole
step.    python tools/head_insitu.py [batch] [dtype]"""
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
    cg, _, _, _ = bench.build_plan(batch, dtype)
    k = [i for i, f in enumerate(cg.step_fname) if f == 'fss_conv2d_dual_head_nhwc'][0]
    s = cg.steps[k]
    d = s.keep[0] if isinstance(s.keep, tuple) else s.keep
    shipped = d.config
    for cfg in range(1, _lib.lib.fss_conv_config_count() + 1):
        d.config = cfg
        if s.fn(*s.args) != 0:
            continue
        _lib.call('fss_device_sync')
        prof = cg.profile(repeats=5)
        print('cfg %2d%s  head %.3f ms (mean %.3f)   step %.3f ms' % (
            cfg, ' (shipped)' if cfg == shipped else '', prof[k]['ms'], prof[k]['mean_ms'],
            sum(p['mean_ms'] for p in prof)), flush=True)
    d.config = shipped


if __name__ == '__main__':
    main()
