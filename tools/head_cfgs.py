"""Times the classifier launch of the benchmarked plan (last fusion + `Predictions`, interpolated
residual) with every tile configuration that takes it.    python tools/head_cfgs.py [batch] [dtype]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd')]
import bench                                   # noqa: E402
from fss import _lib                           # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    dtype = sys.argv[2] if len(sys.argv) > 2 else 'mixed'
    cg, _, _, _ = bench.build_plan(batch, dtype)
    for s, fname in zip(cg.steps, cg.step_fname):
        if fname != 'fss_conv2d_dual_head_nhwc':
            continue
        d = s.keep[0] if isinstance(s.keep, tuple) else s.keep
        shipped = d.config
        for cfg in range(1, _lib.lib.fss_conv_config_count() + 1):
            d.config = cfg
            if s.fn(*s.args) != 0:
                continue
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(10):
                    s.fn(*s.args)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) * 100.0)
            print('%s cfg %2d%s  %.3f ms' % (s.name, cfg, ' (shipped)' if cfg == shipped else '', best), flush=True)
        d.config = shipped


if __name__ == '__main__':
    main()
