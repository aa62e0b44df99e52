"""End-to-end parity: the HIP forward pass vs the numpy oracle on the same
synthetic weights and seeded inputs, at sizes the oracle finishes in seconds.

Tolerances are the ones the task states: logits within 1e-3 (float32) / 2e-2
(bfloat16) of the logit scale, argmax agreement >= 99.9 % in float32; label
maps, colour maps, confusion matrix and mIoU bit-exact given identical labels.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import nets, ops as O, metrics as OM     # noqa: E402

CFG = """
model { %(arch)s { num_classes: 19 filter_scale: %(fs)s pooling_factors: %(pf)s
  feature_extractor { type: '%(ext)s' }
  hyperparams { batch_norm { train: true scale: true center: true decay: 0.9997 epsilon: 0.00001 }
                regularizer { l2_regularizer { weight: 0.0001 } }
                initializer { variance_scaling_initializer {} } }
  loss { classification_loss { softmax {} } use_auxiliary_loss: true } } }
eval_config: { num_examples: 4 eval_input_type { padded_eval_input { height: %(h)d width: %(w)d } } }
"""


def build(arch, ext, fs, pf, in_shape, pad_shape, color_map=None):
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import graph as G, synth
    from libs import exporter
    cfg = pipeline_pb2.PipelineConfig()
    text_format.Merge(CFG % dict(arch=arch, ext=ext, fs=fs, pf=pf, h=pad_shape[0], w=pad_shape[1]), cfg)
    g = G.reset_default_graph()
    _, model = model_builder.build(cfg.model, is_training=False)
    placeholder, inp = exporter._image_tensor_input_placeholder(in_shape, pad_shape)
    pre = model.preprocess(G.to_float(inp))
    logits = model.predict(pre)[model.main_class_predictions_key]
    labels = G.argmax(logits, 3)
    outs = [logits, labels]
    if color_map is not None:
        n_in = in_shape if len(in_shape) == 3 else in_shape[1:]
        outs.append(G.crop_to_bounding_box(G.map_to_colored_labels(labels, color_map), 0, 0,
                                           n_in[0], n_in[1]))
    istd = 70.0 if ext == 'dilated_resnet50' else 0.55
    weights = synth.make_weights(g.variables, seed=1234, input_std=istd)
    return g, cfg, outs, weights


def oracle_logits(arch, ext, fs, pf, img, pad_shape, weights):
    x = O.pad_to_bounding_box(img.astype(np.float32), pad_shape[0], pad_shape[1])
    x = nets.preprocess(x, ext)
    P = nets.Params(weights)
    fn = nets.icnet_predict if arch == 'icnet' else nets.pspnet_predict
    return fn(P, x, filter_scale=1.0 / fs, pooling_factors=pf, extractor=ext)


CASES = [
    ('icnet', 'dilated_resnet50', 1.0, [1.0, 2.0, 3.0, 4.0], (2, 256, 512, 3), (257, 513)),
    ('icnet', 'dilated_resnet50', 0.5, [1.0, 2.0, 3.0, 4.0], (1, 320, 448, 3), (321, 449)),
    ('pspnet', 'dilated_resnet50', 1.0, [1.0, 2.0, 3.0, 6.0], (2, 193, 193, 3), (193, 193)),
    ('icnet', 'dilated_mobilenet', 1.0, [1.0, 2.0, 3.0, 4.0], (1, 256, 512, 3), (257, 513)),
    ('pspnet', 'dilated_mobilenet', 1.0, [1.0, 2.0, 3.0, 6.0], (1, 193, 193, 3), (193, 193)),
]


@pytest.mark.parametrize('case', CASES, ids=lambda c: '%s-%s-%s' % (c[0], c[1], c[2]))
@pytest.mark.parametrize('dtype', ['float32', 'float32x3', 'mixed', 'bfloat16'])
def test_forward_parity(case, dtype):
    from fss import plan
    from libs.constants import CITYSCAPES_LABEL_COLORS
    arch, ext, fs, pf, in_shape, pad_shape = case
    g, cfg, outs, weights = build(arch, ext, fs, pf, in_shape, pad_shape, CITYSCAPES_LABEL_COLORS)
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, in_shape).astype(np.uint8)
    ref = oracle_logits(arch, ext, fs, pf, img, pad_shape, weights)
    # commute_zoom=True: the plan large batches get (fusion convs in front of their zoom); the
    # small-batch plan is compared with it in test_layer_fusions_agree_with_the_unfused_plan
    cg = plan.CompiledGraph(g, outs, weights, dtype=dtype, commute_zoom=True, commute_pyramid=True)
    logits, labels, colors = cg.run(img)
    assert logits.shape == ref.shape
    scale = float(np.abs(ref).max())
    err = float(np.abs(logits - ref).max()) / scale
    ref_lab = O.argmax_channels(ref)
    agree = float((labels[..., 0] == ref_lab).mean())
    print('%s %s: logit err %.3g of scale, argmax agreement %.5f' % (case[:3], dtype, err, agree))
    if dtype != 'bfloat16':
        # float32 MFMAs, float32 tensors with split-bf16 products, and the mixed mode (the same with
        # the 3x3 convolutions of the bottleneck units and ICNet's full-resolution branch in half
        # precision): every stated tolerance, the float32 one for the logits included
        assert err <= 1e-3
        assert agree >= 0.999
    else:
        # bfloat16 meets the logit tolerance but NOT the 99.9 % label gate on these weights: the
        # classifier of synth.make_weights is zero-mean, so ~1 % of the pixels have a top-2
        # margin below 1e-3 of the logit scale and ~9 % below 1e-2, and bf16 rounding noise
        # (0.5-0.8 % rms of every activation from the first residual block on) flips 1-2 % of
        # them.  The floor below is what is measured, not what is required; 'float32x3' is the
        # mode that meets the gate (asserted above).
        assert err <= 2e-2
        assert agree >= 0.98
        s = np.sort(ref, axis=3)
        clear = (s[..., -1] - s[..., -2]) > 2 * 2e-2 * scale
        assert np.array_equal(labels[..., 0][clear], ref_lab[clear])
    # labels are exactly the argmax of the logits the GPU produced
    assert np.array_equal(labels[..., 0], np.argmax(logits, axis=3))
    # colour map: bit-exact given the label map
    h, w = in_shape[1], in_shape[2]
    assert np.array_equal(colors, O.map_to_colored_labels(labels[..., 0], CITYSCAPES_LABEL_COLORS)[:, :h, :w])
    # the hipGraph replay gives the same result as the eager launch sequence
    cg.launch()
    cg.sync()
    assert np.array_equal(cg.fetch(outs[1]), labels)


@pytest.mark.parametrize('case', CASES[:2], ids=lambda c: '%s-%s-%s' % (c[0], c[1], c[2]))
def test_bf16_labels_on_plain_he_normal_weights(case):
    """The same comparison on weights drawn by the plain recipe (synth.make_weights_survey: He-normal
    kernels, batch norm gamma ~ U(0.5, 1.5), nothing zero-mean): label regions are larger, margins
    wider; bfloat16 still misses the 99.9 % label gate (0.996-0.999 measured), so it is held to the logit
    tolerance and a floor of 99 %, float32x3 to every stated tolerance.  (ICNet only: with this recipe
    PSPNet's pre-activations reach 1e4 and its relu6 features are almost all 0 or 6, so one flipped
    feature moves a logit by a sizeable share of the logit scale in any 8-bit-mantissa arithmetic.)"""
    from fss import plan, synth
    arch, ext, fs, pf, in_shape, pad_shape = case
    g, cfg, outs, _ = build(arch, ext, fs, pf, in_shape, pad_shape)
    weights = synth.make_weights_survey(g.variables, seed=1234)
    img = np.random.default_rng(0).integers(0, 256, in_shape).astype(np.uint8)
    ref = oracle_logits(arch, ext, fs, pf, img, pad_shape, weights)
    ref_lab = O.argmax_channels(ref)
    scale = float(np.abs(ref).max())
    for dtype, tol, floor in (('bfloat16', 2e-2, 0.99), ('float32x3', 1e-3, 0.999)):
        logits, labels = plan.CompiledGraph(g, outs, weights, dtype=dtype).run(img)
        err = float(np.abs(logits - ref).max()) / scale
        agree = float((labels[..., 0] == ref_lab).mean())
        print('%s %s plain weights: logit err %.3g of scale, argmax agreement %.5f' % (case[:3], dtype, err, agree))
        assert err <= tol
        assert agree >= floor


def test_classifier_commute_is_exact_enough():
    """Moving the 1x1 classifier in front of the x2 zoom changes logits by
    rounding only."""
    from fss import plan
    case = CASES[0]
    g, cfg, outs, weights = build(*case)
    img = np.random.default_rng(1).integers(0, 256, case[4]).astype(np.uint8)
    a = plan.CompiledGraph(g, outs, weights, dtype='float32', commute_classifier=True).run(img)[0]
    b = plan.CompiledGraph(g, outs, weights, dtype='float32', commute_classifier=False).run(img)[0]
    assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max()


def test_two_zooms_before_the_argmax_run_as_one():
    """The evaluator's label map is argmax(zoom x4(zoom x2(classifier scores))), both zooms
    align_corners: the plan interpolates once, x8, from the scores (inside one source cell the
    first zoom samples a bilinear function, which the second reproduces).  Labels must agree
    with the two-step plan and with the oracle's two-step float32 computation to the stated
    argmax tolerance; only near-ties may differ."""
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import graph as G, synth, plan
    from libs import evaluator
    arch, ext, fs, pf, in_shape, pad_shape = CASES[0]
    cfg = pipeline_pb2.PipelineConfig()
    text_format.Merge(CFG % dict(arch=arch, ext=ext, fs=fs, pf=pf, h=pad_shape[0], w=pad_shape[1]), cfg)
    g = G.reset_default_graph()
    _, model = model_builder.build(cfg.model, is_training=False)
    preds, ph = evaluator.create_predictions_and_labels(model, in_shape[1:], pad_shape[0], pad_shape[1],
                                                        False, in_shape[0])
    weights = synth.make_weights(g.variables, seed=1234, input_std=70.0)
    img = np.random.default_rng(8).integers(0, 256, in_shape).astype(np.uint8)
    one = plan.CompiledGraph(g, [preds], weights, dtype='float32')
    two = plan.CompiledGraph(g, [preds], weights, dtype='float32', collapse_zooms=False)
    n_resize = lambda cg: sum(1 for s in cg.steps if s.kind == 'resize')
    assert n_resize(one) == n_resize(two) - 1
    la, lb = one.run(img)[0][..., 0], two.run(img)[0][..., 0]
    assert la.shape == (in_shape[0],) + tuple(pad_shape)
    assert float((la == lb).mean()) >= 0.9999
    ref = oracle_logits(arch, ext, fs, pf, img, pad_shape, weights)      # after the x2 zoom
    ref = O.resize_bilinear(ref, pad_shape[0], pad_shape[1], True)
    assert float((la == O.argmax_channels(ref)).mean()) >= 0.999


def test_inference_session_matches_oracle():
    """libs.exporter surface: uint8 HWC image in, float32 [1,h,w,3] colours out."""
    from libs import exporter
    from libs.constants import CITYSCAPES_LABEL_COLORS, CITYSCAPES_LABEL_IDS
    from fss import graph as G, synth
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    cfg = pipeline_pb2.PipelineConfig()
    text_format.Merge(CFG % dict(arch='icnet', ext='dilated_resnet50', fs=1.0,
                                 pf=[1.0, 2.0, 3.0, 4.0], h=257, w=513), cfg)
    img = np.random.default_rng(2).integers(0, 256, (256, 512, 3)).astype(np.uint8)
    for table in (CITYSCAPES_LABEL_COLORS, CITYSCAPES_LABEL_IDS):
        g = G.reset_default_graph()
        _, model = model_builder.build(cfg.model, is_training=False)
        outputs, placeholder = exporter.deploy_segmentation_inference_graph(
            model=model, input_shape=[256, 512, 3], pad_to_shape=[257, 513], label_color_map=table)
        weights = synth.make_weights(g.variables)
        sess = exporter.Session(outputs, placeholder, weights, dtype='float32')
        got = sess.run(img)
        P = nets.Params(weights)
        ref = nets.inference_outputs(lambda x: nets.icnet_predict(P, x), img, (256, 512, 3),
                                     (257, 513), table)
        assert got.shape == ref.shape == (1, 256, 512, len(table[0])) and got.dtype == np.float32
        assert (np.all(got == ref, axis=-1)).mean() >= 0.999


def _eval_examples(n, h, w, seed=5):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        img = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
        lab = rng.integers(0, 19, (h, w)).astype(np.uint8)
        lab[rng.random((h, w)) < 0.05] = 255
        out.append((img, lab))
    return out


@pytest.mark.parametrize('batch_size', [1, 2])
def test_evaluator_confusion_matrix_and_miou_bit_exact(batch_size):
    """libs.evaluator surface: mIoU over padded examples.  Given the GPU's own
    label maps, the confusion matrix and mIoU must equal the oracle's exactly;
    zero-padded label pixels count as class 0; a ragged last batch is handled."""
    import functools
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import graph as G, synth, plan
    from libs import evaluator
    cfg = pipeline_pb2.PipelineConfig()
    text_format.Merge(CFG % dict(arch='icnet', ext='dilated_resnet50', fs=0.5,
                                 pf=[1.0, 2.0, 3.0, 4.0], h=257, w=513), cfg)
    cfg.eval_config.num_examples = 5
    examples = _eval_examples(7, 256, 512)          # only the first 5 may be used
    create_model_fn = functools.partial(model_builder.build, model_config=cfg.model, is_training=False)
    g = G.reset_default_graph()
    model_builder.build(cfg.model, is_training=False)[1].predict(
        G.channel_affine(G.to_float(G.placeholder((1, 257, 513, 3))), [0, 0, 0], [1, 1, 1]))
    weights = synth.make_weights(g.variables)
    miou, cm = evaluator.eval_segmentation_model(
        create_model_fn, lambda: iter(examples), (257, 513), cfg.eval_config, train_dir=None,
        eval_dir=None, weights=weights, dtype='float32', batch_size=batch_size,
        return_confusion=True)
    # independent pass: label maps from a plain batch-1 plan, metric from the oracle
    g = G.reset_default_graph()
    _, model = model_builder.build(cfg.model, is_training=False)
    preds, ph = evaluator.create_predictions_and_labels(model, (256, 512, 3), 257, 513, False, 1)
    cg = plan.CompiledGraph(g, [preds], weights, dtype='float32')
    ref_cm = np.zeros((19, 19), np.int64)
    for img, lab in examples[:5]:
        labels = cg.run(img)[0][0, :, :, 0]
        padded = np.zeros((257, 513), np.uint8)
        padded[:256, :512] = lab
        ref_cm += OM.confusion_matrix(labels, padded, 19, 255)
    assert np.array_equal(cm, ref_cm)
    assert miou == OM.mean_iou(ref_cm)
    assert cm[0].sum() > 257 + 513 - 1 - 1        # the padded row / column landed in class 0


def test_cli_inference_from_checkpoint(tmp_path):
    """inference.py surface: config file + checkpoint prefix + PNG in, PNG out."""
    from PIL import Image
    from fss import checkpoint, graph as G, synth
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from libs.constants import CITYSCAPES_LABEL_IDS
    import inference
    cfg_text = CFG % dict(arch='icnet', ext='dilated_resnet50', fs=0.5, pf=[1.0, 2.0, 3.0, 4.0],
                          h=257, w=513)
    cfg_path = tmp_path / 'model.config'
    cfg_path.write_text(cfg_text)
    cfg = pipeline_pb2.PipelineConfig()
    text_format.Merge(cfg_text, cfg)
    g = G.reset_default_graph()
    _, model = model_builder.build(cfg.model, is_training=False)
    model.predict(model.preprocess(G.to_float(G.placeholder((1, 257, 513, 3)))))
    weights = synth.make_weights(g.variables)
    extra = dict(weights)
    extra['global_step'] = np.asarray(7, np.int64)                  # ignored by the loader
    extra['AuxOutput/weights'] = np.zeros((1, 1, 128, 19), np.float32)
    prefix = str(tmp_path / 'train' / 'model.ckpt-7')
    checkpoint.save_variables(prefix, extra)
    img = np.random.default_rng(3).integers(0, 256, (256, 512, 3)).astype(np.uint8)
    (tmp_path / 'in').mkdir()
    Image.fromarray(img).save(str(tmp_path / 'in' / 'frame.png'))
    G.reset_default_graph()
    inference.main(['--input_path', str(tmp_path / 'in'), '--input_shape', '256,512,3',
                    '--pad_to_shape', '257,513', '--config_path', str(cfg_path),
                    '--trained_checkpoint', prefix, '--output_dir', str(tmp_path / 'out'),
                    '--label_ids', '--dtype', 'float32'])
    got = np.asarray(Image.open(str(tmp_path / 'out' / 'frame.png')))
    P = nets.Params(weights)
    ref = nets.inference_outputs(lambda x: nets.icnet_predict(P, x, filter_scale=2.0), img,
                                 (256, 512, 3), (257, 513), CITYSCAPES_LABEL_IDS)
    assert got.shape == (256, 512) and got.dtype == np.uint8
    assert (got == ref[0, :, :, 0].astype(np.uint8)).mean() >= 0.999
    # the archive's other weight source: frozen_inference_graph.pb, with the variables as
    # constants (same label map, bit for bit) and with batch norm folded into the kernels
    from fss import frozen_graph
    convs = [(nd.attrs['weights'].rsplit('/', 1)[0], nd.attrs['rate'] > 1) for nd in g.nodes if nd.kind == 'conv2d']
    for fold in (False, True):
        pb = str(tmp_path / ('frozen_%d.pb' % fold))
        frozen_graph.save_frozen_graph(pb, extra, convs, fold_batch_norm=fold)
        G.reset_default_graph()
        inference.main(['--input_path', str(tmp_path / 'in'), '--input_shape', '256,512,3',
                        '--pad_to_shape', '257,513', '--config_path', str(cfg_path),
                        '--trained_checkpoint', pb, '--output_dir', str(tmp_path / ('out_pb%d' % fold)),
                        '--label_ids', '--dtype', 'float32'])
        got_pb = np.asarray(Image.open(str(tmp_path / ('out_pb%d' % fold) / 'frame.png')))
        if fold:
            assert (got_pb == got).mean() >= 0.999
        else:
            assert np.array_equal(got_pb, got)


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_side_stream_and_stem_fusion_do_not_change_results(dtype):
    """The two-stream schedule only reorders launches along true data dependencies, and
    the plan picks tile configurations inside one MFMA shape class, so logits must be
    bit-identical to the single-stream plan; the fused uint8 stem convolution must agree
    with the unfused preprocess -> resize -> conv path to rounding."""
    from fss import plan
    arch, ext, fs, pf, in_shape, pad_shape = CASES[0]
    g, cfg, outs, weights = build(arch, ext, fs, pf, in_shape, pad_shape)
    img = np.random.default_rng(3).integers(0, 256, in_shape).astype(np.uint8)
    got = {}
    for key, kw in (('two', dict(streams=2)), ('one', dict(streams=1)),
                    ('eager', dict(streams=2, use_hip_graph=False)),
                    ('unfused', dict(streams=1, fuse_stem=False))):
        cg = plan.CompiledGraph(g, outs, weights, dtype=dtype, **kw)
        if key == 'two':
            assert cg.two_streams or cg.stream_choice.startswith('single: two streams'), cg.stream_choice
            assert any(s.stream == 1 for s in cg.steps), 'ICNet must have a side chain'
            assert any(s.kind == 'stem' for s in cg.steps)
        if key == 'unfused':
            assert not any(s.kind == 'stem' for s in cg.steps)
        got[key] = cg.run(img)
        if key == 'two':        # replays of the two-stream graph are deterministic
            for _ in range(5):
                cg.launch()
                cg.sync()
                assert np.array_equal(cg.fetch(outs[0]), got[key][0])
        del cg
    for key in ('one', 'eager'):
        assert np.array_equal(got['two'][0], got[key][0]), key
        assert np.array_equal(got['two'][1], got[key][1]), key
    scale = float(np.abs(got['unfused'][0]).max())
    err = float(np.abs(got['two'][0] - got['unfused'][0]).max()) / scale
    assert err <= (1e-5 if dtype == 'float32' else 2e-2), err


@pytest.mark.parametrize('dtype', ['float32', 'bfloat16'])
def test_layer_fusions_agree_with_the_unfused_plan(dtype):
    """Each cross-layer fusion of the plan compiler (the two stride-2 convolutions of the
    full-resolution branch in one kernel; the sum of two convolutions as one GEMM; the
    classifier in the epilogue of the last convolution; the dilated convolution of a cascade
    feature fusion computed in front of the x2 zoom) is
    switched off in turn; logits must agree with the fully fused plan to rounding (the
    fused forms keep float32 where the unfused ones round an intermediate to the storage
    type, so bfloat16 differs by at most a few bf16 steps of the logit scale)."""
    from fss import plan
    arch, ext, fs, pf, in_shape, pad_shape = CASES[0]
    g, cfg, outs, weights = build(arch, ext, fs, pf, in_shape, pad_shape)
    img = np.random.default_rng(5).integers(0, 256, in_shape).astype(np.uint8)
    ref_plan = plan.CompiledGraph(g, outs, weights, dtype=dtype, streams=1, commute_zoom=True, commute_pyramid=True)
    kinds = [s.kind for s in ref_plan.steps]
    fnames = list(ref_plan.step_fname)
    assert 'fss_conv2d_dual_nhwc' in fnames, 'ICNet must use the two-input convolution'
    if dtype == 'bfloat16':
        assert 'stem2' in kinds, 'ICNet in bfloat16 must fuse the first two branch convolutions'
        assert 'fss_conv2d_dual_head_nhwc' in fnames, 'ICNet in bfloat16 must fuse the classifier'
    ref = ref_plan.run(img)[0]
    scale = float(np.abs(ref).max())
    assert 'fss_zoom2_conv_commute' in fnames, 'ICNet must run the fusion\'s dilated conv before the zoom'
    assert any(s.name.endswith('/pyramid_sum') for s in ref_plan.steps), \
        'ICNet must apply the pyramid module\'s 1x1 convolution to the pooled maps'
    if dtype == 'bfloat16':
        assert 'fss_conv1x1_chain_nhwc' in fnames, 'ResNet block1 in bfloat16 must chain conv3 and the next conv1'
    for kw in (dict(fuse_stem2=False), dict(fuse_dual=False), dict(fuse_head=False), dict(commute_zoom=False),
               dict(commute_pyramid=False), dict(fuse_chain=False),
               dict(fuse_stem2=False, fuse_dual=False, fuse_head=False, commute_zoom=False,
                    commute_pyramid=False, fuse_chain=False)):
        kw.setdefault('commute_zoom', True)
        kw.setdefault('commute_pyramid', True)
        cg = plan.CompiledGraph(g, outs, weights, dtype=dtype, streams=1, **kw)
        if not kw.get('commute_pyramid', True):
            assert not any(s.name.endswith('/pyramid_sum') for s in cg.steps)
        if not kw.get('fuse_chain', True):
            assert 'fss_conv1x1_chain_nhwc' not in cg.step_fname
        if not kw.get('fuse_head', True):
            assert 'fss_conv2d_dual_head_nhwc' not in cg.step_fname
        if not kw.get('commute_zoom', True):
            assert 'fss_zoom2_conv_commute' not in cg.step_fname
        if not kw.get('fuse_dual', True):
            assert 'fss_conv2d_dual_nhwc' not in cg.step_fname
        if not kw.get('fuse_stem2', True):
            assert not any(s.kind == 'stem2' for s in cg.steps)
        got = cg.run(img)[0]
        err = float(np.abs(got - ref).max()) / scale
        assert err <= (1e-5 if dtype == 'float32' else 2e-2), (kw, err)
        del cg


def test_every_launch_writes_only_its_own_outputs():
    """Runs the launches of an ICNet plan one at a time and checks that no activation
    buffer other than the launch's declared outputs changes (the two-stream schedule
    derives its dependencies from exactly these read / write sets)."""
    import ctypes
    import torch
    from fss import plan
    from fss._lib import check
    arch, ext, fs, pf, in_shape, pad_shape = CASES[0]
    g, cfg, outs, weights = build(arch, ext, fs, pf, in_shape, pad_shape)
    cg = plan.CompiledGraph(g, outs, weights, dtype='bfloat16', streams=1, use_hip_graph=False)
    cg.run(np.random.default_rng(4).integers(0, 256, in_shape).astype(np.uint8))
    bufs = {}
    for v in list(cg.vals.values()) + cg.temps + [cg.input_val]:
        bufs[v.buf.data_ptr()] = v.buf
    for t in cg.workspaces:
        bufs[t.data_ptr()] = t

    def buf_of(ptr):
        for bp, b in bufs.items():
            if bp <= ptr < bp + b.numel() * b.element_size():
                return bp
        return None

    for i, s in enumerate(cg.steps):
        mine = set()
        for ai in plan._OUT_ARGS[cg.step_fname[i]]:
            a = s.args[ai]
            for q in (list(a) if isinstance(a, ctypes.Array) else [a]):
                if q:
                    mine.add(buf_of(q))
        assert None not in mine, 'output of %s is not a plan buffer' % s.name
        before = {bp: b.clone() for bp, b in bufs.items() if bp not in mine}
        check(s.fn(*s.args), s.name)
        torch.cuda.synchronize()
        for bp, b in before.items():
            assert torch.equal(b.view(torch.uint8), bufs[bp].view(torch.uint8)), \
                '%s (%s) wrote outside its outputs' % (s.name, s.kind)


def test_side_chain_started_in_the_middle_of_the_pass():
    """The side chain (here headed by the uint8 stem convolution of the full-resolution branch)
    is released right before the pyramid pooling kernels, so that it shares the compute units
    with them - the situation in which a build WITH packed-f32 VALU math (hipcc's SLP
    vectorizer) lost blend terms in lanes 48-63 of pyramid_add_kernel's waves.  The shipped
    build must reproduce the single-stream outputs of every launch bit for bit, launched
    eagerly on two streams and replayed as a two-stream graph, and re-capturing (which tears
    the previous executable graph down) must keep working."""
    import torch
    from fss import plan
    arch, ext, fs, pf = 'icnet', 'dilated_resnet50', 1.0, [1.0, 2.0, 3.0, 4.0]
    in_shape, pad_shape = (2, 1024, 2048, 3), (1025, 2049)
    g, cfg, outs, weights = build(arch, ext, fs, pf, in_shape, pad_shape)
    img = np.random.default_rng(12).integers(0, 256, in_shape).astype(np.uint8)
    cg = plan.CompiledGraph(g, outs, weights, dtype='bfloat16', streams=1, commute_zoom=False,
                            fuse_stem2=False)
    cg.run(img)
    bufs = {}
    for v in list(cg.vals.values()) + cg.temps:
        bufs[v.buf.data_ptr()] = v.buf
    torch.cuda.synchronize()
    ref = {p: b.clone() for p, b in bufs.items()}
    fork = [i for i, s in enumerate(cg.steps) if s.kind == 'avgpool'][0] - 1
    assert cg.steps[fork].kind == 'conv'
    cg._setup_streams(tune=False)
    assert cg._assign_streams(fork)
    side = [s for s in cg.steps if s.stream == 1]
    assert side and side[0].kind == 'stem'
    cg.two_streams = True

    def check(tag):
        for it in range(4):
            cg.launch()
            cg.sync()
            torch.cuda.synchronize()
            for p, b in bufs.items():
                assert torch.equal(b.view(torch.uint8), ref[p].view(torch.uint8)), \
                    '%s, pass %d: a buffer of shape %s differs from the single-stream pass' % (tag, it, tuple(b.shape))

    saved, cg.graph_exec = cg.graph_exec, None
    check('eager two streams')
    cg.graph_exec = saved
    for round_ in range(3):                 # each capture destroys the previous executable
        cg._capture()
        check('two-stream graph %d' % round_)


def test_pruned_weights_run_in_the_half_width_model():
    """libs/filter_pruner on the ICNet-1.0 weights (factor 0.5, shipped prune config) gives weights
    for the filter_scale 0.5 model - the path from a 1.0 checkpoint to the compressed model of
    BASELINE's configuration 3: the HIP forward pass on them matches the oracle's."""
    import os
    from google.protobuf import text_format
    from protos import compressor_pb2
    from builders import compressor_builder
    from conftest import PKG
    from fss import plan
    arch, ext, pf, in_shape, pad = 'icnet', 'dilated_resnet50', [1.0, 2.0, 3.0, 4.0], (1, 128, 256, 3), (129, 257)
    g1, _, _, w1 = build(arch, ext, 1.0, pf, in_shape, pad)
    c = compressor_pb2.CommpressionConfig()
    with open(os.path.join(PKG, 'configs', 'compression', 'icnet_resnet_v1_pruner_v2.prune_config')) as f:
        text_format.Merge(f.read(), c)
    pruned = compressor_builder.build(c.compression_strategy, [], 0.5).compress(g1, w1)
    g, cfg, outs, _ = build(arch, ext, 0.5, pf, in_shape, pad)
    assert all(tuple(pruned[k].shape) == tuple(v) for k, v in g.variables.items())
    img = np.random.default_rng(5).integers(0, 256, in_shape).astype(np.uint8)
    ref = oracle_logits(arch, ext, 0.5, pf, img, pad, pruned)
    logits, labels = plan.CompiledGraph(g, outs, pruned, dtype='float32').run(img)
    assert float(np.abs(logits - ref).max()) <= 1e-3 * float(np.abs(ref).max())
    assert (labels[..., 0] == O.argmax_channels(ref)).mean() >= 0.999


@pytest.mark.parametrize('dtype,tol', [('float32', 1e-3), ('mixed', 1e-3), ('bfloat16', 2e-2)])
def test_icnet_pretrain_single_branch_mode(dtype, tol):
    """ICNet with pretrain_single_branch_mode (architectures/icnet_architecture.py:64-79: the
    full-resolution backbone with its pyramid module alone, no second or third branch, logits
    zoomed x8) through the HIP path against the oracle."""
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import graph as G, synth, plan
    from libs import exporter
    in_shape, pad_shape = (1, 192, 256, 3), (193, 257)
    cfg = pipeline_pb2.PipelineConfig()
    text_format.Merge(CFG % dict(arch='icnet', ext='dilated_resnet50', fs=1.0, pf=[1.0, 2.0, 3.0, 4.0],
                                 h=pad_shape[0], w=pad_shape[1]), cfg)
    cfg.model.icnet.pretrain_single_branch_mode = True
    g = G.reset_default_graph()
    _, model = model_builder.build(cfg.model, is_training=False)
    _, inp = exporter._image_tensor_input_placeholder(in_shape, pad_shape)
    logits = model.predict(model.preprocess(G.to_float(inp)))[model.main_class_predictions_key]
    outs = [logits, G.argmax(logits, 3)]
    weights = synth.make_weights(g.variables, seed=1234, input_std=70.0)
    assert 'Conv/weights' not in weights and 'CascadeFeatureFusion/Conv/weights' not in weights
    img = np.random.default_rng(5).integers(0, 256, in_shape).astype(np.uint8)
    x = nets.preprocess(O.pad_to_bounding_box(img.astype(np.float32), *pad_shape), 'dilated_resnet50')
    ref = nets.icnet_predict(nets.Params(weights), x, pretrain_single_branch_mode=True)
    got, labels = plan.CompiledGraph(g, outs, weights, dtype=dtype).run(img)
    assert got.shape == ref.shape == (1,) + pad_shape + (19,)
    err = float(np.abs(got - ref).max()) / float(np.abs(ref).max())
    agree = float((labels[..., 0] == O.argmax_channels(ref)).mean())
    print('single-branch ICNet %s: logit err %.3g of scale, argmax agreement %.5f' % (dtype, err, agree))
    assert err <= tol
    if dtype != 'bfloat16':
        assert agree >= 0.999


@pytest.mark.parametrize('dtype', ['mixed', 'bfloat16'])
def test_shared_activation_memory_changes_nothing(dtype):
    """Tensors with disjoint lifetimes share device memory (the lifetimes are read off the
    launches of a first lowering pass): same bits as the plan with a buffer per tensor, for the
    logits and the labels, on the two-stream graph and on eager replays, in less memory."""
    from fss import plan
    case = CASES[0]
    arch, ext, fs, pf, in_shape, pad_shape = case
    g, cfg, outs, weights = build(arch, ext, fs, pf, (8,) + tuple(in_shape[1:]), pad_shape)
    img = np.random.default_rng(3).integers(0, 256, (8,) + tuple(in_shape[1:])).astype(np.uint8)
    ref = plan.CompiledGraph(g, outs, weights, dtype=dtype, reuse_buffers=False)
    want = ref.run(img)
    assert ref.activation_bytes == 0
    del ref
    cg = plan.CompiledGraph(g, outs, weights, dtype=dtype, reuse_buffers=True)
    got = cg.run(img)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    assert 0 < cg.activation_bytes < 0.7 * cg.activation_bytes_unshared
    for _ in range(2):          # replays, and the launch-by-launch path
        cg.launch()
    cg.sync()
    assert np.array_equal(cg.fetch(outs[0]), want[0])
    cg.run_eager()
    cg.sync()
    assert np.array_equal(cg.fetch(outs[1]), want[1])
    # a side chain survives the sharing: its buffers are shared among its own launches only
    assert any(s.stream == 1 for s in cg.steps) or not cg.stream_choice.startswith('two')
