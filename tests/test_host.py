"""Host-side logic that needs no GPU: config parsing, model building, variable
naming, C-ABI exports, the mIoU formula and the multi-rank reduction."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import PKG, ROOT


def _cfg(name):
    from google.protobuf import text_format
    from protos import pipeline_pb2
    cfg = pipeline_pb2.PipelineConfig()
    with open(os.path.join(PKG, 'configs', name)) as f:
        text_format.Merge(f.read(), cfg)
    return cfg


def _build(name, hw):
    from builders import model_builder
    from fss import graph as G
    cfg = _cfg(name)
    g = G.reset_default_graph()
    nc, model = model_builder.build(cfg.model, is_training=False)
    x = G.placeholder((1,) + hw + (3,), 'uint8')
    out = model.predict(model.preprocess(G.to_float(x)))
    return g, model, out[model.main_class_predictions_key]


@pytest.mark.parametrize('name,hw,mparams', [
    ('single_stage_icnet_1.0_1025_resnet_v1.config', (1025, 2049), 26.82),
    ('two_stage_icnet_0.5_1025_resnet_v1_stage_2.config', (1025, 2049), 6.71),
    ('pspnet_1.0_713_resnet_v1.config', (713, 713), 46.53),
    ('single_stage_icnet_1.0_1025_mobilenet_v2.config', (1025, 2049), 3.65)])
def test_graph_matches_published_counts(name, hw, mparams):
    g, model, logits = _build(name, hw)
    assert logits.shape == (1,) + hw + (19,)
    n = sum(int(np.prod(s)) for v, s in g.variables.items() if v.endswith('weights'))
    assert round(n / 1e6, 2) == mparams


@pytest.mark.parametrize('name,ext,arch', [
    ('single_stage_icnet_1.0_1025_resnet_v1.config', 'dilated_resnet50', 'icnet'),
    ('single_stage_icnet_1.0_1025_mobilenet_v2.config', 'dilated_mobilenet', 'icnet'),
    ('pspnet_1.0_713_resnet_v1.config', 'dilated_resnet50', 'pspnet'),
    ('pspnet_1.0_713_mobilenet_v2.config', 'dilated_mobilenet', 'pspnet')])
def test_variable_names_match_the_oracle(name, ext, arch):
    """Two independent derivations of the TF checkpoint variable names / shapes."""
    from oracle import nets
    g, _, _ = _build(name, (257, 513))
    P = nets.Params(None)
    fn = nets.icnet_predict if arch == 'icnet' else nets.pspnet_predict
    fn(P, np.zeros((1, 257, 513, 3), np.float32), extractor=ext)
    assert P.shapes == dict(g.variables)
    assert 'Predictions/Conv/biases' in g.variables
    if arch == 'icnet':
        for v in ('FastPSPModule/Conv/weights', 'Conv_3/BatchNorm/moving_variance',
                  'CascadeFeatureFusion_1/DilatedConv/weights'):
            assert v in g.variables


def test_resnet_dilations_compound():
    """block3 runs with dilation 2*2 and block4 with 4*4 (stack_blocks_dense rate x
    the block's own overide_rate)."""
    g, _, _ = _build('single_stage_icnet_1.0_1025_resnet_v1.config', (257, 513))
    rates = {}
    for nd in g.nodes:
        if nd.kind == 'conv2d' and nd.attrs['weights'].endswith('conv2/weights'):
            block = re.search(r'(block\d)', nd.attrs['weights']).group(1)
            rates.setdefault(block, set()).add((nd.attrs['rate'], nd.attrs['stride']))
    assert rates['block1'] == {(1, 1), (1, 2)}
    assert rates['block2'] == {(1, 1)} and rates['block3'] == {(4, 1)} and rates['block4'] == {(16, 1)}


def test_reference_configs_parse_with_runtime_schema():
    """text_format syntax used by the reference's configs: list literals,
    `name: {`, comments, training blocks."""
    from google.protobuf import text_format
    from protos import pipeline_pb2
    text = '''
    # comment
    model { icnet { num_classes: 19 filter_scale: 0.5 pooling_factors: [1.0, 2.0]
            feature_extractor { type: 'dilated_resnet50' } pretrain_single_branch_mode: true } }
    train_config: { batch_size: 16 num_steps: 90000 fine_tune_checkpoint_type: "classification"
        optimizer { momentum_optimizer: { learning_rate: { polynomial_decay_learning_rate {
            initial_learning_rate: 0.001 decay_steps: 90000 power: 0.9 } } momentum_optimizer_value: 0.9 } }
        preprocessor_step { random_image_scale { min_scale_ratio: 0.5 max_scale_ratio: 2.0 } }
        preprocessor_step { random_image_crop { crop_height: 1025 crop_width: 1025 } }
        preprocessor_step { random_horizontal_flip {} } }
    train_input_reader: { num_examples: 2975 shuffle: true tf_record_input_reader { input_path: "/tmp/a.record" } }
    eval_config: { num_examples: 500 eval_input_type { padded_eval_input { height: 1025 width: 2049 } } }
    '''
    cfg = pipeline_pb2.PipelineConfig()
    text_format.Merge(text, cfg)
    assert cfg.model.WhichOneof('model') == 'icnet'
    assert list(cfg.model.icnet.pooling_factors) == [1.0, 2.0]
    assert cfg.eval_config.ignore_label == 255 and cfg.eval_config.num_examples == 500
    assert cfg.model.icnet.hyperparams.batch_norm.epsilon == pytest.approx(1e-5)
    assert cfg.train_config.optimizer.momentum_optimizer.momentum_optimizer_value == pytest.approx(0.9)
    assert cfg.model.icnet.main_branch_loss_weight == 1.0


def test_model_builder_errors():
    from builders import model_builder
    from protos import model_pb2
    cfg = _cfg('single_stage_icnet_1.0_1025_resnet_v1.config')
    with pytest.raises(ValueError, match='not of type'):
        model_builder.build(cfg, is_training=False)
    with pytest.raises(NotImplementedError):
        model_builder.build(cfg.model, is_training=True)
    bad = model_pb2.FastSegmentationModel()
    bad.CopyFrom(cfg.model)
    bad.icnet.filter_scale = 1.5
    with pytest.raises(ValueError, match='filter_scale'):
        model_builder.build(bad, is_training=False)
    bad.icnet.filter_scale = 1.0
    bad.icnet.num_classes = 0
    with pytest.raises(ValueError, match='num_classes'):
        model_builder.build(bad, is_training=False)
    bad.icnet.num_classes = 19
    bad.icnet.feature_extractor.type = 'vgg'
    with pytest.raises(ValueError, match='Unknown ICNet feature_extractor'):
        model_builder.build(bad, is_training=False)
    with pytest.raises(ValueError, match='Unknown model'):
        model_builder.build(model_pb2.FastSegmentationModel(), is_training=False)
    # the reference's own (mistyped) extractor key builds
    bad.icnet.feature_extractor.type = 'dialted_mobilenet'
    model_builder.build(bad, is_training=False)


def test_preprocess_requires_float():
    from fss import graph as G
    g, model, _ = _build('single_stage_icnet_1.0_1025_resnet_v1.config', (257, 513))
    with pytest.raises(ValueError, match='float32'):
        model.preprocess(G.placeholder((1, 257, 513, 3), 'uint8'))


def test_pretrain_single_branch_mode_graph():
    from builders import model_builder
    from fss import graph as G
    cfg = _cfg('single_stage_icnet_1.0_1025_resnet_v1.config')
    cfg.model.icnet.pretrain_single_branch_mode = True
    g = G.reset_default_graph()
    _, model = model_builder.build(cfg.model, is_training=False)
    x = G.placeholder((1, 257, 513, 3), 'uint8')
    out = model.predict(model.preprocess(G.to_float(x)))['class_predictions']
    assert out.shape == (1, 257, 513, 19)
    assert 'Conv/weights' not in g.variables and 'FastPSPModule/Conv/weights' in g.variables


def test_cabi_exports_every_declared_symbol():
    """The shared library loads and exports exactly what include/fss.h declares
    (no compute calls: there is no GPU here)."""
    header = open(os.path.join(ROOT, 'include', 'fss.h')).read()
    header = re.sub(r'/\*.*?\*/', '', header, flags=re.S)
    declared = set(re.findall(r'\b(fss_[a-z0-9_]+)\s*\(', header))
    so = os.path.join(PKG, 'libfss_gfx950.so')
    assert os.path.exists(so), 'build the library first (python -c "import __graft_entry__ as g; g.build()")'
    lib = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(lib, name), name
    from fss import _lib
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert _lib.lib.fss_version() >= 100
    assert _lib.conv_pack_dims(8, 3, 3, 19, _lib.BF16) == (128, 256)
    assert _lib.lib.fss_conv_pack_dims(0, 3, 3, 19, 0, None, None) != 0
    assert b'null' in _lib.lib.fss_last_error()


def test_library_has_no_packed_f32_valu():
    """No v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 in the shipped code objects (the build
    switches hipcc's SLP vectorizer off): kernels that used them lost results in the last
    quarter of a wave when they shared compute units with another stream's kernel."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import check_isa
    assert check_isa.packed_f32_kernels(os.path.join(PKG, 'libfss_gfx950.so')) == {}


def test_mean_iou_formula_matches_oracle():
    from fss import metrics
    from oracle import metrics as OM
    rng = np.random.default_rng(0)
    for _ in range(20):
        cm = rng.integers(0, 10 ** 9, (19, 19)) * (rng.random((19, 19)) < 0.3)
        assert metrics.mean_iou_from_confusion(cm) == OM.mean_iou(cm)
    assert metrics.mean_iou_from_confusion(np.zeros((19, 19), np.int64)) == 0


def test_bf16_rounding_helper():
    import torch
    from fss.plan import f32_to_bf16_bits
    x = np.random.default_rng(1).normal(size=4096).astype(np.float32)
    x[:4] = [1.0, 1.00390625, 1.01171875, -0.0]      # exact, tie -> even (down), tie -> even (up)
    ref = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    np.testing.assert_array_equal(f32_to_bf16_bits(x), ref)


def test_confusion_allreduce_world_size_2_gloo():
    """N>1 path: each rank owns a shard of frames, matrices are summed once."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dist_worker.py')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29611')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                          '--nproc-per-node=2', '--master-addr', '127.0.0.1', '--master-port',
                          '29611', script], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert 'DIST_OK' in out.stdout


def test_filter_pruner_gives_the_half_width_model(tmp_path):
    """compress.py on ICNet-1.0 / PSPNet-1.0 weights with the shipped prune configs and factor 0.5:
    every pruned variable has the shape the filter_scale 0.5 model declares, the pruned ICNet has
    the published 6.71 M convolution parameters (docs/model_zoo.md:28), and the kept filters of
    a convolution are those with the largest L1 norms, in their original order
    (libs/filter_pruner.py:163-177)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import checkpoint, graph as G, synth
    import compress

    def model(arch, fs, pf):
        cfg = ("model { %s { num_classes: 19 filter_scale: %s pooling_factors: %s "
               "feature_extractor { type: 'dilated_resnet50' } "
               "hyperparams { batch_norm { train: true scale: true center: true decay: 0.9997 epsilon: 0.00001 } "
               "regularizer { l2_regularizer { weight: 0.0001 } } initializer { variance_scaling_initializer {} } } "
               "loss { classification_loss { softmax {} } use_auxiliary_loss: true } } }" % (arch, fs, pf))
        pc = pipeline_pb2.PipelineConfig()
        text_format.Merge(cfg, pc)
        g = G.reset_default_graph()
        _, m = model_builder.build(pc.model, is_training=False)
        m.predict(m.preprocess(G.to_float(G.placeholder((1, 129, 257, 3)))))
        return cfg, dict(g.variables)

    for arch, pf, prune, params in (('icnet', [1.0, 2.0, 3.0, 4.0], 'icnet_resnet_v1_pruner_v2', 6708032),
                                    ('pspnet', [1.0, 2.0, 3.0, 6.0], 'pspnet50_resnet_v1_pruner', None)):
        cfg_text, full_vars = model(arch, 1.0, pf)
        _, half_vars = model(arch, 0.5, pf)
        weights = synth.make_weights(full_vars)
        d = tmp_path / arch
        d.mkdir()
        (d / 'pipeline.config').write_text(cfg_text)
        checkpoint.save_variables(str(d / 'model.ckpt'), weights)
        rc = compress.main(['--input_checkpoint', str(d / 'model.ckpt'), '--config_path', str(d / 'pipeline.config'),
                            '--prune_config', os.path.join(PKG, 'configs', 'compression', prune + '.prune_config'),
                            '--compression_factor', '0.5', '--output_dir', str(d / 'out')])
        assert rc == 0
        pruned = checkpoint.load_variables(str(d / 'out' / 'pruned_model.ckpt'), half_vars)
        assert set(pruned) == set(half_vars)
        if params:
            assert sum(v.size for k, v in pruned.items() if k.endswith('/weights')) == params
        # a convolution pruned by its own norms: the root convolution
        name = 'SharedFeatureExtractor/resnet_v1_50/conv1/weights'
        w = weights[name]
        keep = np.sort(np.argsort(-np.abs(w).sum((0, 1, 2)))[:w.shape[3] // 2])
        np.testing.assert_array_equal(pruned[name], w[..., keep])
        np.testing.assert_array_equal(pruned[name.replace('weights', 'BatchNorm/gamma')],
                                      weights[name.replace('weights', 'BatchNorm/gamma')][keep])
        # ... and the next convolutions lost the same input channels
        nxt = 'SharedFeatureExtractor/resnet_v1_50/block1/unit_1/bottleneck_v1/conv1/weights'
        assert pruned[nxt].shape[2] == len(keep)
    assert compress.main(['--input_checkpoint', str(tmp_path / 'nothing'), '--config_path', 'x',
                          '--prune_config', 'y']) == -1


def test_shipped_conv_table_was_measured_with_this_library():
    """fss/conv_tuned.json carries a stamp of the library version and the tile-configuration
    table it was measured with; the plan compiler ignores a table whose stamp differs, and
    would then time every candidate at every plan build."""
    import json
    from fss import plan
    with open(os.path.join(PKG, 'fss', 'conv_tuned.json')) as f:
        table = json.load(f)
    assert table['stamp'] == plan.CompiledGraph._tune_stamp(), 'retune: python tools/tune_convs.py --fresh'
    assert len(table['entries']) > 1000
