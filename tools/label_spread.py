"""Label agreement of the mixed mode with the CPU oracle on several different full-size frames
of one of bench.py's other configurations (how far the figure moves from frame to frame).

    python tools/label_spread.py [pspnet713_bs8 | icnet05_bs32 | icnet_mnv2_bs64]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd')]
import numpy as np

import bench


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'icnet_mnv2_bs64'
    cfg_file, in_shape, pad, _, arch, ext, fs, istd = bench.OTHER_CONFIGS[name]
    b = 8
    cg, pred, _, w = bench.build_plan(b, 'mixed', config=cfg_file, input_shape=in_shape, pad_shape=pad,
                                      input_std=istd)
    frames = np.random.default_rng(77).integers(0, 256, (b,) + in_shape, dtype=np.uint8)
    lab = cg.run(frames, [pred])[0]
    agree = []
    for i in range(0, b, 2):
        ref = bench.oracle_labels(w, frames[i:i + 1], arch, ext, fs, pad)
        agree.append(float((lab[i, ..., 0] == ref[0]).mean()))
    print(name, os.environ.get('FSS_HALF_ACT_SCOPES', 'default scopes'), ['%.5f' % a for a in agree])


if __name__ == '__main__':
    main()
