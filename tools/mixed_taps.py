"""Where the error of the 'mixed' mode is born: relative rms error of intermediate tensors of ICNet
(architectures/icnet_architecture.py:110-174) against the float32 oracle, for the GPU plan (needs a
GPU) or for the CPU emulation that rounds the operands of the chosen convolutions to 11 bits.

    python tools/mixed_taps.py gpu|cpu "<half scope regex>[,<regex>...]"
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd'), os.path.join(ROOT, 'tests'),
                os.path.join(ROOT, 'tools')]
import numpy as np

from test_models_gpu import build, CASES
from oracle import nets, ops as O

NAMES = ['quarter_res', 'psp_module', 'first_fusion', 'second_fusion']


def main():
    where, scopes = sys.argv[1], sys.argv[2].split(',')
    arch, ext, fs, pf, in_shape, pad = CASES[0]
    g, cfg, outs, weights = build(arch, ext, fs, pf, in_shape, pad)
    img = np.random.default_rng(0).integers(0, 256, in_shape).astype(np.uint8)
    x = nets.preprocess(O.pad_to_bounding_box(img.astype(np.float32), pad[0], pad[1]), ext)
    ref, rt = nets.icnet_predict(nets.Params(weights), x, return_all=True)
    lab = O.argmax_channels(ref)
    if where == 'gpu':
        from bf16_error_taps import find_taps
        from fss import plan
        taps = find_taps(g)
        cg = plan.CompiledGraph(g, outs + [taps[n] for n in NAMES], weights, dtype='mixed', half_scopes=scopes,
                                commute_zoom=True, commute_pyramid=True)
        got = cg.run(img)
        vals, logits = dict(zip(NAMES, got[2:])), got[0]
    else:
        from precision_emulation import round_bits
        pats = [re.compile(p) for p in scopes]
        hit = lambda s: any(p.search(s) for p in pats)
        cur = [24]
        conv2d, cba, bott = O.conv2d, nets.conv_bn_act, nets._bottleneck
        O.conv2d = lambda xx, w, *a, **k: conv2d(round_bits(xx, cur[0]), round_bits(w, cur[0]), *a, **k)

        def cba2(P, xx, scope, *a, **k):
            cur[0] = 11 if hit(scope) else 24
            r = cba(P, xx, scope, *a, **k)
            cur[0] = 24
            return r
        nets.conv_bn_act = cba2
        nets._bottleneck = lambda P, xx, scope, *a, **k: round_bits(bott(P, xx, scope, *a, **k),
                                                                   11 if hit(scope + '/') else 24)
        logits, vals = nets.icnet_predict(nets.Params(weights), x, return_all=True)
    for n in NAMES:
        v, r = vals[n], rt[n]
        print('  %-14s rms err / rms %.3g   max err / max %.3g' % (
            n, np.sqrt(((v - r) ** 2).mean() / (r ** 2).mean()), np.abs(v - r).max() / np.abs(r).max()))
    print('  logits: rms err / rms %.3g, max err / scale %.3g, flips %.4f%%' % (
        np.sqrt(((logits - ref) ** 2).mean() / (ref ** 2).mean()), np.abs(logits - ref).max() / np.abs(ref).max(),
        100 * (O.argmax_channels(logits) != lab).mean()))


if __name__ == '__main__':
    main()
