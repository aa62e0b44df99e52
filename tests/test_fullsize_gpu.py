"""The headline workload at its full size (ICNet-1.0, 1024x2048 frames padded to
1025x2049): one float32 frame against the numpy oracle (a few seconds of host
time), and size-independent properties of the bf16 path the bench measures:
replay determinism, batch invariance, label / colour / confusion-matrix
consistency."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import ops as O, metrics as OM        # noqa: E402
from test_models_gpu import build, oracle_logits   # noqa: E402

CASE = ('icnet', 'dilated_resnet50', 1.0, [1.0, 2.0, 3.0, 4.0])
PAD = (1025, 2049)


FULL_SIZE = [
    # model, backbone, filter scale, pooling factors, frame shape, padded size
    (CASE, (1, 1024, 2048, 3), PAD),                                               # the headline workload
    (('pspnet', 'dilated_resnet50', 1.0, [1.0, 2.0, 3.0, 6.0]), (1, 713, 713, 3), (713, 713)),
    (('icnet', 'dilated_mobilenet', 1.0, [1.0, 2.0, 3.0, 4.0]), (1, 1024, 2048, 3), PAD),
]


@pytest.mark.parametrize('case,in_shape,pad', FULL_SIZE, ids=['icnet_resnet50', 'pspnet50_713', 'icnet_mobilenetv2'])
def test_full_size_float32_frame_matches_oracle(case, in_shape, pad):
    """One full-size float32 frame of each model family the path serves against the numpy
    oracle: the stated float32 tolerances (logits 1e-3 of scale, labels >= 99.9 %)."""
    from fss import plan
    g, cfg, outs, weights = build(*case, in_shape, pad)
    img = np.random.default_rng(11).integers(0, 256, in_shape).astype(np.uint8)
    ref = oracle_logits(*case, img, pad, weights)
    cg = plan.CompiledGraph(g, outs, weights, dtype='float32')
    logits, labels = cg.run(img)
    scale = float(np.abs(ref).max())
    err = float(np.abs(logits - ref).max()) / scale
    agree = float((labels[..., 0] == O.argmax_channels(ref)).mean())
    print('full size float32 %s: logit err %.3g of scale, argmax agreement %.5f' % (case[:2], err, agree))
    assert logits.shape == (1, pad[0], pad[1], 19)
    assert err <= 1e-3            # the stated float32 tolerance
    assert agree >= 0.999


def test_full_size_bf16_properties():
    from fss import plan, metrics
    from libs.constants import CITYSCAPES_LABEL_COLORS
    import torch
    rng = np.random.default_rng(12)
    frames = rng.integers(0, 256, (2, 1024, 2048, 3)).astype(np.uint8)
    g, cfg, outs, weights = build(*CASE, (2, 1024, 2048, 3), PAD, CITYSCAPES_LABEL_COLORS)
    cg2 = plan.CompiledGraph(g, outs, weights, dtype='bfloat16')
    logits2, labels2, colors2 = cg2.run(frames)
    # 1. a replay of the captured graph is bit-identical
    cg2.launch()
    cg2.sync()
    assert np.array_equal(cg2.fetch(outs[0]), logits2)
    # 2. labels are the argmax (first index on ties) of the logits, colours are the table
    assert labels2.max() < 19
    assert np.array_equal(labels2[..., 0], np.argmax(logits2, axis=3))
    assert np.array_equal(colors2, O.map_to_colored_labels(labels2[..., 0], CITYSCAPES_LABEL_COLORS)
                          [:, :1024, :2048])
    # 3. confusion matrix on the device == the oracle's on the same label maps; every
    #    non-ignored pixel is counted once
    gt = rng.integers(0, 19, (2,) + PAD).astype(np.uint8)
    gt[rng.random(gt.shape) < 0.05] = 255
    m = metrics.StreamingMeanIoU(19, 255)
    pred_val = cg2.value(outs[1])
    with torch.cuda.stream(cg2.stream):
        gt_dev = torch.from_numpy(gt).to(cg2.device)
    m.update(pred_val.buf.view(-1), gt_dev.view(-1), cg2.sh)
    cg2.sync()
    miou, cm = m.result(), m.confusion_matrix()
    ref_cm = OM.confusion_matrix(labels2[..., 0], gt, 19, 255)
    assert np.array_equal(np.asarray(cm), ref_cm)
    assert int(np.asarray(cm).sum()) == int((gt != 255).sum())
    assert miou == OM.mean_iou(ref_cm)
    del cg2
    # 4. batch invariance: a frame alone gives the bits it gives inside a batch (tile
    #    configurations differ with the batch, results inside one MFMA shape class do not)
    g1, _, outs1, _ = build(*CASE, (1, 1024, 2048, 3), PAD)
    cg1 = plan.CompiledGraph(g1, outs1, weights, dtype='bfloat16')
    logits1, labels1 = cg1.run(frames[1:2])
    del cg1
    assert np.array_equal(logits1[0], logits2[1])
    assert np.array_equal(labels1[0], labels2[1])
    # 5. the low-latency plan (split-K launches in the late backbone blocks: another order of the
    #    float additions) agrees with it within bf16 rounding steps
    cgk = plan.CompiledGraph(g1, outs1, weights, dtype='bfloat16', ksplit=True)
    logitsk, labelsk = cgk.run(frames[1:2])
    nsplit = sum(f == 'fss_conv2d_splitk_nhwc' for f in cgk.step_fname)
    scale = float(np.abs(logits1).max())
    err = float(np.abs(logitsk - logits1).max()) / scale
    agree = float((labelsk == labels1).mean())
    print('split-K plan vs plain plan, one bf16 frame: %d split-K launches, logit diff %.3g of scale, '
          'labels equal %.5f' % (nsplit, err, agree))
    assert nsplit > 0 and err <= 2e-2 and agree >= 0.99


def _scores_of(cg):
    """The classifier scores the fused zoom + argmax launch reads, as a host array
    [n, h, w, classes] (the tensor the plan computes in place of the reference's zoomed logits)."""
    import torch
    i = [k for k, f in enumerate(cg.step_fname) if f == 'fss_upsample_argmax'][0]
    a = cg.steps[i].args
    ptr, n, h, w, c, ld = a[0], a[3], a[4], a[5], a[6], a[7]
    for v in cg.vals.values():
        if v.buf.data_ptr() == ptr and v.buf.numel() == n * h * w * ld:     # (dead tensors may share its memory)
            cg.sync()
            torch.cuda.synchronize()
            return v.buf.reshape(n, h, w, ld)[..., :c].float().cpu().numpy()
    raise AssertionError('scores buffer not found')


@pytest.mark.parametrize('config,fs', [('single_stage_icnet_1.0_1025_resnet_v1.config', 1.0),
                                       ('two_stage_icnet_0.5_1025_resnet_v1_stage_2.config', 0.5)],
                         ids=['icnet_1.0', 'icnet_0.5'])
def test_benchmarked_plan_at_full_size_matches_oracle(config, fs):
    """The plan bench.py times - evaluator graph, 1025x2049, 8 frames per launch, so that the
    fusion convolutions run in front of their zoom, the classifier sits in the last fusion's
    epilogue with the interpolated residual, the two zooms are one and two streams are on -
    against the numpy oracle on its first and last frame: classifier scores (zoomed x2 on the
    host, which is where the reference applies the classifier) within the stated tolerance of
    the dtype, labels as stated for float32x3 and the mixed mode and as measured for bfloat16; and against the
    small-batch plan (fusion convolutions on the zoomed grid, stored residual)."""
    import bench
    from fss import plan
    from oracle import nets
    B = 8
    rng = np.random.default_rng(21)
    frames = rng.integers(0, 256, (B, 1024, 2048, 3)).astype(np.uint8)
    refs = {}
    for dtype, tol, floor in (('bfloat16', 2e-2, 0.97), ('float32x3', 1e-3, 0.999), ('mixed', 1e-3, 0.999)):
        cg, pred, _, weights = bench.build_plan(B, dtype, config=config)
        names = [s.name for s in cg.steps]
        assert any(n.endswith('@source') for n in names), 'fusion convs must run at the source resolution'
        assert cg.two_streams or cg.stream_choice.startswith('single: two streams'), cg.stream_choice
        if dtype == 'bfloat16':
            head = [k for k, f in enumerate(cg.step_fname) if f == 'fss_conv2d_dual_head_nhwc']
            assert head and cg.steps[head[0]].args[7] is not None, 'classifier epilogue with interpolated residual'
        labels = cg.run(frames)[0]
        scores = _scores_of(cg)
        del cg
        for f in (0, B - 1):
            if f not in refs:
                x = nets.preprocess(O.pad_to_bounding_box(frames[f:f + 1].astype(np.float32), *PAD), 'dilated_resnet50')
                refs[f] = nets.icnet_predict(nets.Params(weights), x, filter_scale=int(round(1.0 / fs)),
                                             return_all=True)
            ref, taps = refs[f]
            low = taps['logits_lowres']
            scale = float(np.abs(low).max())
            zoomed = O.resize_bilinear(scores[f:f + 1], low.shape[1], low.shape[2], True)
            err = float(np.abs(zoomed - low).max()) / scale
            ref_lab = O.argmax_channels(ref)
            agree = float((labels[f, ..., 0] == ref_lab[0]).mean())
            print('%s %s frame %d of %d: score err %.3g of scale, argmax agreement %.5f' % (config[:22], dtype, f, B, err, agree))
            assert err <= tol
            assert agree >= floor
            if dtype == 'bfloat16':
                s_ = np.sort(ref, axis=3)
                clear = (s_[..., -1] - s_[..., -2]) > 2 * 2e-2 * float(np.abs(ref).max())
                assert np.array_equal(labels[f, ..., 0][clear[0]], ref_lab[0][clear[0]])
        if dtype == 'bfloat16':
            # the small-batch plan of the same model: same scores to a few bf16 roundings
            cg1, _, _, _ = bench.build_plan(1, dtype, config=config)
            assert not any(s.name.endswith('@source') for s in cg1.steps)
            labels1 = cg1.run(frames[:1])[0]
            scores1 = _scores_of(cg1)
            del cg1
            sc = float(np.abs(scores1).max())
            d = float(np.abs(scores1[0] - scores[0]).max()) / sc
            same = float((labels1[0] == labels[0]).mean())
            print('  batch-8 plan vs batch-1 plan: scores differ by %.3g of scale, labels agree %.5f' % (d, same))
            assert d <= 2e-2 and same >= 0.97


@pytest.mark.parametrize('name', ['pspnet713_bs8', 'icnet05_bs32'])
def test_other_configurations_at_their_stated_batch(name):
    """PSPNet-50 713x713 at 8 frames per launch and the half-width ICNet at 32 (the batch sizes
    their published figures are quoted on), in the default compute mode: the first and the last
    frame of the batch - two different frames - against the numpy oracle at the stated label
    floor, and a copy of the first frame in the middle of the batch gives the first frame's
    labels bit for bit (no frame depends on its position or its neighbours)."""
    import bench
    cfg_file, in_shape, pad, b, arch, ext, fs, istd = bench.OTHER_CONFIGS[name]
    cg, pred, _, weights = bench.build_plan(b, 'mixed', config=cfg_file, input_shape=in_shape, pad_shape=pad,
                                            input_std=istd)
    frames = np.random.default_rng(31).integers(0, 256, (b,) + in_shape, dtype=np.uint8)
    frames[b // 2] = frames[0]
    labels = cg.run(frames, [pred])[0]
    del cg
    assert labels.shape[0] == b
    assert np.array_equal(labels[b // 2], labels[0])
    for f in (0, b - 1):
        ref = bench.oracle_labels(weights, frames[f:f + 1], arch, ext, fs, pad)
        agree = float((labels[f, ..., 0] == ref[0]).mean())
        print('%s frame %d of %d: argmax agreement %.5f' % (name, f, b, agree))
        assert agree >= 0.999


def test_headline_plan_takes_the_intended_routes():
    """The plan bench.py times by default (mixed mode, its default frames per launch): every
    layer shape comes from the shipped tuning table (a stale table would be ignored silently and
    every plan build would time its candidates), and the precision routes of the mode are the
    ones taken - two-f16-weight launches for conv3 of the bottleneck units, the first unit and the
    classifier launch, the raw-byte stem with half-precision output, a half-precision max-pool -
    on a fraction of the memory a buffer per tensor would take."""
    import re
    import bench
    from fss import _lib
    # bench.py's own default
    batch = int(re.search(r"'--batch', type=int, default=(\d+)", open(bench.__file__).read()).group(1))
    cg, pred, _, _ = bench.build_plan(batch, 'mixed')
    assert cg.tuned_now == 0, '%d layer shapes were not in fss/conv_tuned.json' % cg.tuned_now
    descs = [(s.keep[0] if isinstance(s.keep, tuple) else s.keep) for s in cg.steps]
    wide = [d for d in descs if isinstance(d, _lib.ConvDesc) and d.in_dtype == _lib.F16_W2]
    assert len(wide) >= 14
    fn = dict(zip(cg.step_fname, cg.steps))
    assert fn['fss_conv2d_dual_head_nhwc'].keep[0].in_dtype == _lib.F16_W2
    assert fn['fss_stem_conv_u8'].args[22] == _lib.F16_W2
    assert fn['fss_maxpool_nhwc'].args[12] == _lib.F16
    assert 0 < cg.activation_bytes < 0.6 * cg.activation_bytes_unshared
    del cg
    for dtype in ('bfloat16', 'float32x3'):
        cg, _, _, _ = bench.build_plan(batch, dtype)
        assert cg.tuned_now == 0, dtype
        del cg
