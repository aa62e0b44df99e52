"""Times fss_conv1x1_chain_nhwc at the shapes of ICNet-1.0's first block (bf16) beside the two
launches it replaces (fss_conv2d_dual_nhwc / fss_conv2d_nhwc with a residual, then the next
1x1 convolution), on random data, and prints ms and GB/s of algorithmic bytes.

    python tools/chain_micro.py [--frames 128] [--iters 20]
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'fast-semantic-segmentation_amd'))
from fss import _lib  # noqa: E402


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
    ap.add_argument('--frames', type=int, default=128)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    m = args.frames * 129 * 257
    s = torch.cuda.current_stream().cuda_stream
    bf = lambda *shape: torch.randn(shape, device='cuda').to(torch.bfloat16)
    for cin, cin2, with_res in ((64, 64, False), (64, 0, True)):
        cout1, cout2 = 256, 64
        k = cin + cin2
        x, x2 = bf(m, cin), (bf(m, cin2) if cin2 else None)
        res = bf(m, cout1) if with_res else None
        w, w2 = bf(cout1, k) * 0.1, bf(cout2, cout1) * 0.1
        b, b2 = torch.zeros(cout1, device='cuda'), torch.zeros(cout2, device='cuda')
        y, y2 = torch.empty((m, cout1), dtype=torch.bfloat16, device='cuda'), torch.empty((m, cout2), dtype=torch.bfloat16, device='cuda')

        def chain():
            _lib.call('fss_conv1x1_chain_nhwc', x.data_ptr(), cin, cin, x2.data_ptr() if cin2 else None, cin2, cin2,
                      w.data_ptr(), b.data_ptr(), res.data_ptr() if with_res else None, cout1 if with_res else 0,
                      1, y.data_ptr(), cout1, cout1, w2.data_ptr(), b2.data_ptr(), 1, y2.data_ptr(), cout2, cout2,
                      m, _lib.BF16, s)
        ms = timed(chain, args.iters)
        nbytes = m * (k + cout1 * (2 if with_res else 1) + cout2) * 2
        print('chain  K=%3d res=%d: %.4f ms  %.0f GB/s' % (k, with_res, ms, nbytes / ms / 1e6))
        # the unfused pair: packed weights of the generic kernel
        wp = torch.zeros((256, (k + 63) // 64 * 64), dtype=torch.bfloat16, device='cuda')
        wp[:cout1, :k] = w
        wp2 = torch.zeros((256, 256), dtype=torch.bfloat16, device='cuda')
        wp2[:cout2, :cout1] = w2
        d1 = _lib.ConvDesc(n=args.frames, h=129, w=257, cin=cin, cout=cout1, kh=1, kw=1, stride=1, dilation=1, pad_top=0, pad_left=0,
                           ho=129, wo=257, act=1, in_dtype=_lib.BF16, out_dtype=_lib.BF16, y_ld=cout1,
                           res_ld=cout1 if with_res else 0, k_pad=wp.shape[1], cout_pad=256, config=26)
        d2 = _lib.ConvDesc(n=args.frames, h=129, w=257, cin=cout1, cout=cout2, kh=1, kw=1, stride=1, dilation=1, pad_top=0, pad_left=0,
                           ho=129, wo=257, act=1, in_dtype=_lib.BF16, out_dtype=_lib.BF16, y_ld=cout2, res_ld=0,
                           k_pad=256, cout_pad=256, config=20)

        def pair():
            if cin2:
                _lib.call('fss_conv2d_dual_nhwc', ctypes.byref(d1), x.data_ptr(), x2.data_ptr(), cin2, wp.data_ptr(),
                          b.data_ptr(), None, y.data_ptr(), s)
            else:
                _lib.call('fss_conv2d_nhwc', ctypes.byref(d1), x.data_ptr(), wp.data_ptr(), b.data_ptr(),
                          res.data_ptr(), y.data_ptr(), s)
            _lib.call('fss_conv2d_nhwc', ctypes.byref(d2), y.data_ptr(), wp2.data_ptr(), b2.data_ptr(), None,
                      y2.data_ptr(), s)
        ms2 = timed(pair, args.iters)
        print('pair   K=%3d res=%d: %.4f ms  (chain is %.2fx)' % (k, with_res, ms2, ms2 / ms))


if __name__ == '__main__':
    main()
