"""Micro-benchmark of fss_stem_conv_u8 on the two ICNet stem shapes (random frames).

usage: python tools/stem_micro.py [--dtype bf16|f32] [--batch B] [--iters N]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd'), os.path.join(ROOT, 'tests')]

import numpy as np
import torch

SHAPES = {  # name: (sub, logical hw, k, stride, pad, cout, out hw)
    'branch_conv': (1, (1025, 2049), 3, 2, 1, 32, (513, 1025)),
    'resnet_conv1': (2, (513, 1025), 7, 2, 3, 64, (257, 513)),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('names', nargs='*', default=list(SHAPES) + ['branch_conv2'])
    ap.add_argument('--dtype', default='bf16')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--batch', type=int, default=16)
    args = ap.parse_args()
    from fss import _lib
    import gpu_util as gu
    code, tdt, _ = gu.DT[args.dtype]
    n = args.batch
    frames = torch.randint(0, 256, (n, 1024, 2048, 3), dtype=torch.uint8, device='cuda')
    for name in args.names:
        if name == 'branch_conv2':      # the fused pair 'Conv' + 'Conv_1' (bf16 only)
            c1 = c2 = 64
            rng = np.random.default_rng(0)
            w1 = gu.pack_stem_weights(rng.normal(0, 0.05, (3, 3, 3, c1)).astype(np.float32), 'bf16')
            w2, _, _ = gu.pack_conv_weights(rng.normal(0, 0.05, (3, 3, c1, c2)).astype(np.float32), 'bf16')
            b1, b2 = torch.randn(c1, device='cuda'), torch.randn(c2, device='cuda')
            y = torch.empty((n, 257, 513, c2), device='cuda', dtype=torch.bfloat16)
            m3 = (ctypes.c_float * 3)(123.68, 116.779, 103.939)
            s3 = (ctypes.c_float * 3)(1.0, 1.0, 1.0)

            def run():
                _lib.call('fss_stem2_conv_u8', frames.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                          b2.data_ptr(), y.data_ptr(), n, 1024, 2048, 1, 1025, 2049, m3, s3, c1, 1, 1, 513, 1025,
                          2, c2, 1, 1, 257, 513, 2, _lib.BF16, c2, None)
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
            flops = 2.0 * n * (513 * 1025 * c1 * 27 + 257 * 513 * c2 * 9 * c1)
            print('%-13s %8.4f ms  %7.1f TFLOP/s' % (name, ms, flops / ms / 1e9), flush=True)
            continue
        sub, (h, w), k, stride, pad, cout, (ho, wo) = SHAPES[name]
        wt = np.random.default_rng(0).normal(0, 0.05, (k, k, 3, cout)).astype(np.float32)
        wdev = gu.pack_stem_weights(wt, args.dtype, (123.68, 116.779, 103.939), (1.0, 1.0, 1.0))
        bias = torch.randn(cout, device='cuda')
        y = torch.empty((n, ho, wo, cout), device='cuda', dtype=tdt)
        m3 = (ctypes.c_float * 3)(123.68, 116.779, 103.939)
        s3 = (ctypes.c_float * 3)(1.0, 1.0, 1.0)

        def run():
            _lib.call('fss_stem_conv_u8', frames.data_ptr(), wdev.data_ptr(), bias.data_ptr(), y.data_ptr(),
                      n, 1024, 2048, sub, h, w, m3, s3, cout, k, k, stride, 1, pad, pad, ho, wo, 1, code,
                      cout, None)
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
        nbytes = n * (1024 // sub) * (2048 // sub) * 3 + y.numel() * y.element_size()
        print('%-13s %8.4f ms  %7.1f GB/s' % (name, ms, nbytes / ms / 1e6), flush=True)


if __name__ == '__main__':
    main()
