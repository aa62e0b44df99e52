"""Times the last cascade feature fusion + classifier of the benchmarked plan in its forms:
classifier in the epilogue with the interpolated residual (default), with a stored zoomed residual,
and as separate launches.    python tools/head_micro.py [batch] [dtype]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd')]
import bench
from fss import plan


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    dtype = sys.argv[2] if len(sys.argv) > 2 else 'mixed'
    init = plan.CompiledGraph.__init__
    for tag, kw in (('fused head, interpolated residual', {}),
                    ('fused head, stored zoomed residual', dict(fuse_zoom_residual=False)),
                    ('separate classifier', dict(fuse_head=False))):
        def patched(self, *a, **k):
            k.update(kw)
            init(self, *a, **k)
        plan.CompiledGraph.__init__ = patched
        cg, _, _, _ = bench.build_plan(batch, dtype)
        plan.CompiledGraph.__init__ = init
        prof = cg.profile(repeats=3)
        print(tag)
        tot = 0.0
        for p in prof:
            if 'CascadeFeatureFusion_1' in p['name'] or 'Predictions' in p['name']:
                print('   %-60s %.4f ms cfg=%s' % (p['name'], p['ms'], p.get('config')))
                tot += p['ms']
        print('   total %.4f ms; step %.3f ms' % (tot, sum(p['ms'] for p in prof)))
        del cg


if __name__ == '__main__':
    main()
