"""Which storage / product precision keeps the argmax labels of the synthetic-weight models?

CPU experiment on the numpy oracle: every convolution's input tensor (= what the producing kernel
stored) and its weights are rounded to a chosen number of mantissa bits before the product, which
is then exact float32.  Compared with the unrounded oracle on the small model cases of
tests/test_models_gpu.py.  bf16 = 8 bits, fp16 = 11 bits (range not modelled), float32 = 24.

    python tools/precision_emulation.py [case ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd'), os.path.join(ROOT, 'tests')]
import numpy as np

from test_models_gpu import build, CASES
from oracle import nets, ops as O


def round_bits(a, bits):
    """Round float32 to `bits` significant bits (round to nearest even), keeping the exponent range."""
    if bits >= 24:
        return a
    a = np.ascontiguousarray(a, np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    drop = 24 - bits
    u = (u + ((1 << (drop - 1)) - 1) + ((u >> drop) & 1)) >> drop << drop
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def fp16_round(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


MODE = dict(act=24, w=24)
_conv2d, _dw = O.conv2d, O.depthwise_conv2d


def conv2d(x, w, *a, **k):
    return _conv2d(round_bits(x, MODE['act']), round_bits(w, MODE['w']), *a, **k)


def dwconv(x, w, *a, **k):
    return _dw(round_bits(x, MODE['act']), round_bits(w, MODE['w']), *a, **k)


def main():
    which = [int(v) for v in sys.argv[1:]] or [0]
    O.conv2d, O.depthwise_conv2d = conv2d, dwconv
    for ci in which:
        arch, ext, fs, pf, in_shape, pad = CASES[ci]
        g, cfg, outs, weights = build(arch, ext, fs, pf, in_shape, pad)
        img = np.random.default_rng(0).integers(0, 256, in_shape).astype(np.uint8)
        x = nets.preprocess(O.pad_to_bounding_box(img.astype(np.float32), pad[0], pad[1]), ext)
        fn = nets.icnet_predict if arch == 'icnet' else nets.pspnet_predict
        run = lambda: fn(nets.Params(weights), x, filter_scale=1.0 / fs, pooling_factors=pf, extractor=ext)
        MODE.update(act=24, w=24)
        ref = run()
        scale = float(np.abs(ref).max())
        lab = O.argmax_channels(ref)
        print('case %d %s %s %s' % (ci, arch, ext, fs))
        for act, w in ((8, 8), (11, 11), (11, 24), (12, 24), (13, 24), (14, 24), (16, 24), (24, 11), (16, 11),
                       (16, 16)):
            MODE.update(act=act, w=w)
            got = run()
            print('  act %2d bits, weights %2d bits: logit err %.3g of scale, labels %.5f' % (
                act, w, np.abs(got - ref).max() / scale, (O.argmax_channels(got) == lab).mean()), flush=True)


if __name__ == '__main__':
    main()
