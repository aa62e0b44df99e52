"""Micro-benchmark of fss_conv2d_nhwc on one layer shape (random data).

usage: python tools/conv_micro.py [name] [--dtype bf16|f16|f32|f32x3] [--iters N] [--batch B]
Shapes are ICNet-1.0 layers at 1025x2049.  Prints ms and TFLOP/s (HIP events).
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd'), os.path.join(ROOT, 'tests')]

import numpy as np
import torch

SHAPES = {  # name: (h, w, cin, cout, k, stride, rate, residual)
    'cff1_dil': (129, 257, 256, 256, 3, 1, 2, False),
    'cff0_dil': (65, 129, 512, 256, 3, 1, 2, False),
    'conv_3': (129, 257, 128, 256, 3, 1, 1, False),
    'b4_conv2': (33, 65, 512, 512, 3, 1, 4, False),
    'b4_conv3': (33, 65, 512, 2048, 1, 1, 1, True),
    'b4_conv1': (33, 65, 2048, 512, 1, 1, 1, False),
    'b3_conv2': (33, 65, 256, 256, 3, 1, 4, False),
    'b1_conv3': (129, 257, 64, 256, 1, 1, 1, True),
    'b3_conv3': (65, 129, 256, 1024, 1, 1, 1, True),
    'b3_conv1': (65, 129, 1024, 256, 1, 1, 1, False),
    'cff1_conv': (129, 257, 256, 256, 1, 1, 1, True),
    'branch0': (1025, 2049, 8, 64, 3, 2, 1, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('names', nargs='*', default=list(SHAPES))
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--cfg', type=int, default=0)
    ap.add_argument('--cfgs', default='', help='comma-separated configurations to sweep (overrides --cfg)')
    ap.add_argument('--nores', action='store_true', help='drop the residual input')
    ap.add_argument('--zeros', action='store_true', help='zero-filled operands (no data toggling: the clock-limited ceiling)')
    args = ap.parse_args()
    from fss import _lib
    import gpu_util as gu
    code, tdt, _ = gu.DT[args.dtype]
    for name in args.names:
        h, w, cin, cout, k, stride, rate, has_res = SHAPES[name]
        has_res = has_res and not args.nores
        n = args.batch
        keff = k + (k - 1) * (rate - 1)
        ho, wo = -(-h // stride), -(-w // stride)
        pt = max((ho - 1) * stride + keff - h, 0) // 2
        pl = max((wo - 1) * stride + keff - w, 0) // 2
        g = torch.Generator(device='cuda').manual_seed(0)
        x = torch.randn((n, h, w, cin), device='cuda', generator=g).to(tdt)
        wt = np.random.default_rng(0).normal(0, 1 / np.sqrt(k * k * cin), (k, k, cin, cout)).astype(np.float32)
        wdev, k_pad, cout_pad = gu.pack_conv_weights(wt, args.dtype)
        bias = torch.randn(cout, device='cuda')
        if args.zeros:
            x.zero_()
            wdev.zero_()
        res = torch.randn((n, ho, wo, cout), device='cuda', generator=g).to(tdt) if has_res else None
        y = torch.empty((n, ho, wo, cout), device='cuda', dtype=tdt)
        d = _lib.ConvDesc(n=n, h=h, w=w, cin=cin, cout=cout, kh=k, kw=k, stride=stride, dilation=rate,
                          pad_top=pt, pad_left=pl, ho=ho, wo=wo, act=1, in_dtype=gu.conv_in_code(args.dtype), out_dtype=code,
                          y_ld=cout, res_ld=cout, k_pad=k_pad, cout_pad=cout_pad, config=args.cfg)
        torch.cuda.synchronize()

        def run():
            _lib.call('fss_conv2d_nhwc', ctypes.byref(d), x.data_ptr(), wdev.data_ptr(), bias.data_ptr(),
                      res.data_ptr() if res is not None else None, y.data_ptr(), None)
        for cfg in ([int(c) for c in args.cfgs.split(',')] if args.cfgs else [args.cfg]):
            d.config = cfg
            for _ in range(3):
                run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                run()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.iters
            flops = 2.0 * n * ho * wo * cout * k * k * cin
            nbytes = (x.numel() + y.numel() * (2 if has_res else 1)) * x.element_size()
            print('%-10s cfg %2d M=%7d N=%4d K=%5d  %8.4f ms  %7.1f TFLOP/s  %7.1f GB/s' %
                  (name, cfg, n * ho * wo, cout, k * k * cin, ms, flops / ms / 1e9, nbytes / ms / 1e6), flush=True)


if __name__ == '__main__':
    main()
