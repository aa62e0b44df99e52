"""Times fss_conv3x3_c128_nhwc against fss_conv2d_nhwc (every tile configuration that takes the
layer) at `Conv_3`'s shape: frames x 129 x 257 x 128 -> 256, half precision."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'fast-semantic-segmentation_amd'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from fss import _lib                      # noqa: E402
import gpu_util as gu                     # noqa: E402


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    h, wd = 129, 257
    rng = np.random.RandomState(0)
    w = (rng.randn(3, 3, 128, 256) * 0.03).astype(np.float32)
    x = torch.randn(frames, h, wd, 128, device='cuda').half()
    y = torch.empty(frames, h, wd, 256, device='cuda', dtype=torch.float16)
    y2 = torch.empty_like(y)
    bias = torch.randn(256, device='cuda')
    wimg = gu.conv3x3_c128_image(w, 'f16')
    wdev, k_pad, cout_pad = gu.pack_conv_weights(w, 'f16')
    flop = 2.0 * frames * h * wd * 256 * 1152
    s = torch.cuda.current_stream().cuda_stream

    def halo():
        _lib.call('fss_conv3x3_c128_nhwc', x.data_ptr(), 128, wimg.data_ptr(), bias.data_ptr(), y.data_ptr(), 256,
                  frames, h, wd, _lib.ACT_CODES['relu6'], _lib.F16, s)
    ms = timed(halo)
    print('halo            %.3f ms  %.0f TFLOP/s' % (ms, flop / ms * 1e-9), flush=True)
    for cfg in (() if len(sys.argv) > 2 else (0, 26, 27, 28, 33, 34, 35)):
        d = _lib.ConvDesc(n=frames, h=h, w=wd, cin=128, cout=256, kh=3, kw=3, stride=1, dilation=1, pad_top=1,
                          pad_left=1, ho=h, wo=wd, act=_lib.ACT_CODES['relu6'], in_dtype=_lib.F16,
                          out_dtype=_lib.F16, y_ld=256, res_ld=0, k_pad=k_pad, cout_pad=cout_pad, config=cfg)

        def igemm():
            _lib.call('fss_conv2d_nhwc', ctypes.byref(d), x.data_ptr(), wdev.data_ptr(), bias.data_ptr(), None,
                      y2.data_ptr(), s)
        try:
            ms = timed(igemm)
        except _lib.FssError as e:
            print('igemm cfg %2d   rejected: %s' % (cfg, e))
            continue
        print('igemm cfg %2d    %.3f ms  %.0f TFLOP/s' % (cfg, ms, flop / ms * 1e-9), flush=True)
    print('equal:', bool(torch.equal(y, y2)))


if __name__ == '__main__':
    main()
