"""Where is the bfloat16 logit error of ICNet born?  Runs the bf16 (and float32) plan with the
intermediate tensors of architectures/icnet_architecture.py:110-174 fetched as extra outputs and
compares each against the numpy oracle's taps (oracle.nets.icnet_predict(return_all=True)):
max error / scale and rms error / rms per stage, argmax agreement of the final labels and the
share of pixels whose top-2 margin is below a given fraction of the logit scale.

    python tools/bf16_error_taps.py [synth|survey] [H W]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd'), os.path.join(ROOT, 'tests')]
import numpy as np

from test_models_gpu import build
from fss import plan, synth
from oracle import nets, ops as O


def find_taps(g):
    by_w = {}
    for nd in g.nodes:
        if nd.kind == 'conv2d':
            by_w[nd.attrs['weights']] = nd
    t = {}
    t['psp_module'] = by_w['FastPSPModule/Conv/weights'].output
    t['full_res'] = by_w['Conv_3/weights'].output
    t['half_res'] = by_w['CascadeFeatureFusion/Conv/weights'].inputs[0]
    first_avg = [nd for nd in g.nodes if nd.kind == 'avg_pool'][0]
    t['quarter_res'] = first_avg.inputs[0]
    t['first_fusion'] = by_w['CascadeFeatureFusion_1/DilatedConv/weights'].inputs[0].node.inputs[0]
    t['second_fusion'] = by_w['Predictions/Conv/weights'].inputs[0].node.inputs[0]
    return t


def main():
    recipe = sys.argv[1] if len(sys.argv) > 1 else 'synth'
    h, w = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 512)
    in_shape, pad = (1, h, w, 3), (h + 1, w + 1)
    g, cfg, outs, weights = build('icnet', 'dilated_resnet50', 1.0, [1.0, 2.0, 3.0, 4.0], in_shape, pad)
    if recipe == 'survey':
        weights = synth.make_weights_survey(g.variables, seed=1234)
    img = np.random.default_rng(0).integers(0, 256, in_shape).astype(np.uint8)
    x = nets.preprocess(O.pad_to_bounding_box(img.astype(np.float32), pad[0], pad[1]), 'dilated_resnet50')
    ref, rt = nets.icnet_predict(nets.Params(weights), x, return_all=True)
    taps = find_taps(g)
    names = ['half_res', 'quarter_res', 'psp_module', 'full_res', 'first_fusion', 'second_fusion']
    scale = float(np.abs(ref).max())
    s = np.sort(ref, axis=3)
    margin = (s[..., -1] - s[..., -2]) / scale
    print('recipe %s, %dx%d: logit scale %.3g; pixels with top-2 margin below 1e-3 / 3e-3 / 1e-2 / 3e-2 of scale: '
          '%.4f %.4f %.4f %.4f' % (recipe, h, w, scale, (margin < 1e-3).mean(), (margin < 3e-3).mean(),
                                   (margin < 1e-2).mean(), (margin < 3e-2).mean()))
    for dtype in ('float32', 'bfloat16'):
        cg = plan.CompiledGraph(g, outs + [taps[n] for n in names], weights, dtype=dtype)
        got = cg.run(img)
        print(dtype)
        for n, v in zip(names, got[2:]):
            r = rt[n]
            print('  %-14s max err / scale %.3g   rms err / rms %.3g' % (
                n, np.abs(v - r).max() / np.abs(r).max(), np.sqrt(((v - r) ** 2).mean() / (r ** 2).mean())))
        lg, lab = got[0], got[1]
        print('  %-14s max err / scale %.3g   rms err / rms %.3g   argmax agreement %.5f' % (
            'logits', np.abs(lg - ref).max() / scale, np.sqrt(((lg - ref) ** 2).mean() / (ref ** 2).mean()),
            (lab[..., 0] == O.argmax_channels(ref)).mean()))
        # the default plan (taps not fetched, all fusions on)
        cg2 = plan.CompiledGraph(g, outs, weights, dtype=dtype)
        lg, lab = cg2.run(img)
        print('  %-14s max err / scale %.3g   rms err / rms %.3g   argmax agreement %.5f' % (
            'fused plan', np.abs(lg - ref).max() / scale, np.sqrt(((lg - ref) ** 2).mean() / (ref ** 2).mean()),
            (lab[..., 0] == O.argmax_channels(ref)).mean()))


if __name__ == '__main__':
    main()
