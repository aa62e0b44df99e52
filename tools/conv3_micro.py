"""fss_conv3x3_c64_nhwc beside fss_conv2d_nhwc at the 64-input-channel 3x3 layers of ICNet-1.0
(bf16, random data): largest difference between the two results and ms per launch.

    python tools/conv3_micro.py [--frames 128] [--iters 20]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'fast-semantic-segmentation_amd'))
from fss import _lib  # noqa: E402

SHAPES = {
    # name: (h, w, cout, stride, pad_top, pad_left, ho, wo, generic configs to try)
    'block1_conv2': (129, 257, 64, 1, 1, 1, 129, 257, (21, 20, 19, 32)),
    'block1_u3_conv2': (129, 257, 64, 2, 1, 1, 65, 129, (32, 21, 20)),
    'Conv_2': (257, 513, 128, 2, 1, 1, 129, 257, (26, 30, 28)),
}


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('names', nargs='*', default=list(SHAPES))
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    s = torch.cuda.current_stream().cuda_stream
    n = args.frames
    for name in args.names:
        h, w, cout, stride, pt, pl, ho, wo, cfgs = SHAPES[name]
        x = torch.randn((n, h, w, 64), device='cuda').to(torch.bfloat16)
        wp = torch.zeros((256, 576), dtype=torch.bfloat16, device='cuda')
        wp[:cout] = (torch.randn((cout, 576), device='cuda') * 0.05).to(torch.bfloat16)
        b = torch.randn(cout, device='cuda')
        y0 = torch.empty((n, ho, wo, cout), dtype=torch.bfloat16, device='cuda')
        y1 = torch.empty_like(y0)
        best = None
        for cfg in cfgs:
            d = _lib.ConvDesc(n=n, h=h, w=w, cin=64, cout=cout, kh=3, kw=3, stride=stride, dilation=1, pad_top=pt,
                              pad_left=pl, ho=ho, wo=wo, act=1, in_dtype=_lib.BF16, out_dtype=_lib.BF16, y_ld=cout,
                              res_ld=0, k_pad=576, cout_pad=256, config=cfg)

            def generic():
                _lib.call('fss_conv2d_nhwc', ctypes.byref(d), x.data_ptr(), wp.data_ptr(), b.data_ptr(), None,
                          y0.data_ptr(), s)
            ms = timed(generic, args.iters)
            if best is None or ms < best[0]:
                best = (ms, cfg)

        def resident():
            _lib.call('fss_conv3x3_c64_nhwc', x.data_ptr(), 64, wp.data_ptr(), b.data_ptr(), y1.data_ptr(), cout,
                      n, h, w, ho, wo, cout, stride, 1, pt, pl, 1, _lib.BF16, s)
        ms1 = timed(resident, args.iters)
        diff = (y0.float() - y1.float()).abs().max().item()
        scale = y0.float().abs().max().item()
        flops = 2.0 * n * ho * wo * cout * 576
        print('%-16s generic %.4f ms (cfg %d, %.0f TFLOP/s)  resident %.4f ms (%.0f TFLOP/s, %.2fx)  max diff %.3g of %.3g'
              % (name, best[0], best[1], flops / best[0] / 1e9, ms1, flops / ms1 / 1e9, best[0] / ms1, diff, scale))


if __name__ == '__main__':
    main()
