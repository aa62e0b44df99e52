"""Checkpoint-V2 and TFRecord readers: round trips against the writers in this
repo and known answers of the primitives (crc32c, varint, block layout).

Format parity is UNPINNED against TensorFlow-written files: none ship with the
reference (the Model Zoo is a set of remote links, docs/model_zoo.md:5-12) and
TensorFlow is not installed, so only self-consistency and the published
primitives' test vectors are checked here.
"""
import io
import os
import struct

import numpy as np
import pytest

from fss import checkpoint, tfrecord


def test_crc32c_known_answers():
    # RFC 3720 appendix B.4 test vectors
    assert checkpoint.crc32c(b'\x00' * 32) == 0x8A9136AA
    assert checkpoint.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert checkpoint.crc32c(bytes(range(32))) == 0x46DD794E
    assert checkpoint.crc32c(b'123456789') == 0xE3069283
    # LevelDB mask: rotate right 15, add constant
    c = checkpoint.crc32c(b'foo')
    assert checkpoint.masked_crc32c(b'foo') == ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def test_checkpoint_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    variables = {}
    for i in range(300):            # enough entries for several index blocks + prefix sharing
        name = 'SharedFeatureExtractor/resnet_v1_50/block%d/unit_%d/bottleneck_v1/conv%d/weights' % (
            i % 4 + 1, i // 12 + 1, i % 3 + 1)
        variables[name + '_%d' % i] = rng.normal(size=(1, 1, i % 7 + 1, 3)).astype(np.float32)
    variables['global_step'] = np.asarray(1234, np.int64)
    variables['Predictions/Conv/biases'] = rng.normal(size=(19,)).astype(np.float32)
    variables['big'] = rng.normal(size=(300, 300)).astype(np.float32)      # > 64 KiB
    prefix = str(tmp_path / 'model.ckpt-7')
    checkpoint.save_variables(prefix, variables)
    assert os.path.exists(prefix + '.index') and os.path.exists(prefix + '.data-00000-of-00001')
    with open(prefix + '.index', 'rb') as f:
        assert struct.unpack('<Q', f.read()[-8:])[0] == 0xdb4775248b80fb57
    got = checkpoint.load_variables(prefix)
    assert set(got) == set(variables)
    for k, v in variables.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape
        np.testing.assert_array_equal(got[k], v)
    # subset load tolerates extra variables, checks shapes, reports missing ones
    sub = checkpoint.load_variables(prefix, {'Predictions/Conv/biases': (19,)})
    assert list(sub) == ['Predictions/Conv/biases']
    with pytest.raises(KeyError, match='missing 1 variables'):
        checkpoint.load_variables(prefix, ['nope/weights'])
    with pytest.raises(ValueError, match='expected'):
        checkpoint.load_variables(prefix, {'Predictions/Conv/biases': (20,)})
    # corruption is detected
    raw = bytearray(open(prefix + '.index', 'rb').read())
    raw[10] ^= 0xFF
    open(prefix + '.index', 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum'):
        checkpoint.load_variables(prefix)


def test_checkpoint_state_file(tmp_path):
    d = str(tmp_path)
    assert checkpoint.latest_checkpoint(d) is None
    checkpoint.write_checkpoint_state(d, [os.path.join(d, 'model.ckpt-1'), os.path.join(d, 'model.ckpt-2')])
    latest, allp = checkpoint.get_checkpoint_state(d)
    assert latest == os.path.join(d, 'model.ckpt-2')
    assert allp == [os.path.join(d, 'model.ckpt-1'), os.path.join(d, 'model.ckpt-2')]
    import eval as eval_cli
    assert eval_cli.get_checkpoints_from_path('model.ckpt-2', d) == [os.path.join(d, 'model.ckpt-2')]
    with pytest.raises(ValueError, match='Checkpoint not found'):
        eval_cli.get_checkpoints_from_path('model.ckpt-9', d)


def test_tfrecord_example_round_trip(tmp_path):
    from PIL import Image
    from builders import dataset_builder
    from protos import input_reader_pb2
    rng = np.random.default_rng(1)
    records, expected = [], []
    for i in range(3):
        img = rng.integers(0, 256, (12, 16, 3)).astype(np.uint8)
        lab = rng.integers(0, 19, (12, 16)).astype(np.uint8)
        lab[0, 0] = 255
        bi, bl = io.BytesIO(), io.BytesIO()
        Image.fromarray(img).save(bi, 'PNG')
        Image.fromarray(lab).save(bl, 'PNG')
        records.append(tfrecord.make_example({
            'image/encoded': bi.getvalue(), 'image/filename': b'x.png', 'image/format': b'png',
            'image/height': 12, 'image/width': 16, 'image/channels': 3,
            'image/segmentation/class/encoded': bl.getvalue(),
            'image/segmentation/class/format': b'png'}))
        expected.append((img, lab))
    path = str(tmp_path / 'val.record')
    tfrecord.write_records(path, records)
    ex = tfrecord.parse_example(next(tfrecord.read_records(path, verify=True)))
    assert ex['image/height'] == [12] and ex['image/format'] == [b'png']
    cfg = input_reader_pb2.InputReader()
    cfg.shuffle = False
    cfg.tf_record_input_reader.input_path.append(path)
    got = list(dataset_builder.build(cfg))
    assert len(got) == 3
    for (gi, gl), (ei, el) in zip(got, expected):
        np.testing.assert_array_equal(gi, ei)
        np.testing.assert_array_equal(gl, el)
    # record-level sharding: record i belongs to rank i % world; records of other ranks and
    # records past `limit` are skipped before they are decoded
    calls = []
    real = dataset_builder.decode_example
    dataset_builder.decode_example = lambda r: (calls.append(1), real(r))[1]
    try:
        r0 = list(dataset_builder.build(cfg, shard=(0, 2)))
        r1 = list(dataset_builder.build(cfg, shard=(1, 2)))
        assert len(r0) == 2 and len(r1) == 1 and len(calls) == 3
        np.testing.assert_array_equal(r0[1][0], expected[2][0])
        np.testing.assert_array_equal(r1[0][1], expected[1][1])
        assert len(list(dataset_builder.build(cfg, shard=(0, 2), limit=2))) == 1
        assert len(list(dataset_builder.build(cfg, limit=2))) == 2
    finally:
        dataset_builder.decode_example = real
    with pytest.raises(ValueError, match='At least one input path'):
        cfg2 = input_reader_pb2.InputReader()
        dataset_builder.build(cfg2)
    with pytest.raises(ValueError, match='not of type'):
        dataset_builder.build(object())


def test_crc32c_known_answers_and_large_tensor_corruption(tmp_path):
    """CRC-32C check values (RFC 3720 B.4), the blocked numpy form against the byte loop, and a
    flipped byte in a LARGE tensor of a checkpoint (every tensor is checksummed)."""
    assert checkpoint.crc32c(b'123456789') == 0xE3069283
    assert checkpoint.crc32c(bytes(32)) == 0x8A9136AA
    assert checkpoint.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    rng = np.random.default_rng(0)
    for n in (0, 1, 4095, 16384, 16385, 4096 * 37 + 11):
        d = rng.integers(0, 256, n).astype(np.uint8).tobytes()
        assert checkpoint.crc32c(d) == checkpoint._crc_bytes(0xFFFFFFFF, d) ^ 0xFFFFFFFF
    big = rng.normal(0, 1, (3, 3, 64, 128)).astype(np.float32)          # 295 KB
    prefix = str(tmp_path / 'model.ckpt')
    checkpoint.save_variables(prefix, {'a/weights': big, 'b/biases': np.zeros(4, np.float32)})
    np.testing.assert_array_equal(checkpoint.load_variables(prefix)['a/weights'], big)
    path = prefix + '.data-00000-of-00001'
    raw = bytearray(open(path, 'rb').read())
    raw[200000] ^= 0x10
    open(path, 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum mismatch'):
        checkpoint.load_variables(prefix)


def test_frozen_graph_reader(tmp_path):
    """GraphDef of constants (plain and batch-norm-folded form) -> the weights dict; the plan
    compiler's folding of (weights, folded_bias) equals its folding of the batch norm."""
    from fss import frozen_graph
    rng = np.random.default_rng(3)
    v = {
        'A/Conv/weights': rng.normal(0, 1, (3, 3, 8, 16)).astype(np.float32),
        'A/Conv/BatchNorm/gamma': rng.uniform(0.5, 1.5, 16).astype(np.float32),
        'A/Conv/BatchNorm/beta': rng.normal(0, 1, 16).astype(np.float32),
        'A/Conv/BatchNorm/moving_mean': rng.normal(0, 1, 16).astype(np.float32),
        'A/Conv/BatchNorm/moving_variance': rng.uniform(0.5, 1.5, 16).astype(np.float32),
        'A/Dil/weights': rng.normal(0, 1, (3, 3, 16, 8)).astype(np.float32),
        'A/Dil/BatchNorm/gamma': rng.uniform(0.5, 1.5, 8).astype(np.float32),
        'A/Dil/BatchNorm/beta': rng.normal(0, 1, 8).astype(np.float32),
        'A/Dil/BatchNorm/moving_mean': rng.normal(0, 1, 8).astype(np.float32),
        'A/Dil/BatchNorm/moving_variance': rng.uniform(0.5, 1.5, 8).astype(np.float32),
        'Predictions/Conv/weights': rng.normal(0, 1, (1, 1, 8, 19)).astype(np.float32),
        'Predictions/Conv/biases': rng.normal(0, 1, 19).astype(np.float32),
        'global_step': np.asarray(7, np.int64),
    }
    convs = [('A/Conv', False), ('A/Dil', True), ('Predictions/Conv', False)]
    plain = str(tmp_path / 'plain.pb')
    frozen_graph.save_frozen_graph(plain, v, convs)
    wanted = {k: a.shape for k, a in v.items() if k != 'global_step'}
    got = frozen_graph.load_weights(plain, wanted)
    assert set(got) == set(wanted)
    for k in wanted:
        np.testing.assert_array_equal(got[k], v[k])
    assert 'global_step' not in frozen_graph.load_variables(plain)
    with pytest.raises(KeyError, match='missing 1 variables'):
        frozen_graph.load_variables(plain, ['A/Conv/weights', 'B/weights'])
    with pytest.raises(ValueError, match='has shape'):
        frozen_graph.load_variables(plain, {'A/Conv/weights': (1, 1, 8, 16)})
    # node structure of the file: constants, their read ops and the convolutions
    nodes = frozen_graph.parse_graph_def(open(plain, 'rb').read())
    ops = {n['name']: n['op'] for n in nodes}
    assert ops['A/Conv/weights'] == 'Const' and ops['A/Conv/weights/read'] == 'Identity'
    assert ops['A/Dil/Conv2D'] == 'Conv2D'
    # folded form: BatchNorm constants are gone, a bias is added behind the convolution
    folded = str(tmp_path / 'folded.pb')
    frozen_graph.save_frozen_graph(folded, v, convs, fold_batch_norm=True)
    got = frozen_graph.load_weights(folded, wanted)
    assert not any('BatchNorm' in k for k in got)
    for scope in ('A/Conv', 'A/Dil'):
        s = v[scope + '/BatchNorm/gamma'] / np.sqrt(v[scope + '/BatchNorm/moving_variance'].astype(np.float64) + 1e-5)
        np.testing.assert_allclose(got[scope + '/weights'], v[scope + '/weights'] * s, rtol=1e-6)
        np.testing.assert_allclose(got[scope + '/folded_bias'],
                                   v[scope + '/BatchNorm/beta'] - v[scope + '/BatchNorm/moving_mean'] * s, rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(got['Predictions/Conv/biases'], v['Predictions/Conv/biases'])
    # every layer folds with its own epsilon; depthwise kernels fold along their channel axis
    v2 = dict(v)
    v2['D/depthwise/depthwise_weights'] = rng.normal(0, 1, (3, 3, 8, 1)).astype(np.float32)
    for leaf, lo, hi in (('beta', -1, 1), ('moving_mean', -1, 1), ('moving_variance', 0.5, 1.5)):
        v2['D/depthwise/BatchNorm/' + leaf] = rng.uniform(lo, hi, 8).astype(np.float32)
    convs2 = [('A/Conv', False, 1e-3), ('A/Dil', True, None), ('D/depthwise', False, 0.01, True)]
    frozen_graph.save_frozen_graph(folded, v2, convs2, fold_batch_norm=True)
    got = frozen_graph.load_variables(folded)
    s = v['A/Conv/BatchNorm/gamma'] / np.sqrt(v['A/Conv/BatchNorm/moving_variance'].astype(np.float64) + 1e-3)
    np.testing.assert_allclose(got['A/Conv/weights'], v['A/Conv/weights'] * s, rtol=1e-6)
    s = 1.0 / np.sqrt(v2['D/depthwise/BatchNorm/moving_variance'].astype(np.float64) + 0.01)    # no gamma
    np.testing.assert_allclose(got['D/depthwise/depthwise_weights'][..., 0],
                               v2['D/depthwise/depthwise_weights'][..., 0] * s, rtol=1e-6)
    np.testing.assert_allclose(got['D/depthwise/folded_bias'],
                               v2['D/depthwise/BatchNorm/beta'] - v2['D/depthwise/BatchNorm/moving_mean'] * s,
                               rtol=1e-5, atol=1e-6)
    # a folded graph written again stays folded, bit for bit, whatever the flag says
    again = str(tmp_path / 'again.pb')
    frozen_graph.save_frozen_graph(again, got, convs2, fold_batch_norm=False)
    got2 = frozen_graph.load_variables(again)
    assert set(got2) == set(got)
    for k in got:
        np.testing.assert_array_equal(got2[k], got[k])
    ops = {n['name']: n['op'] for n in frozen_graph.parse_graph_def(open(again, 'rb').read())}
    assert ops['D/depthwise/depthwise'] == 'DepthwiseConv2dNative' and ops['D/depthwise/BiasAdd'] == 'BiasAdd'


def test_export_cli_writes_inference_checkpoint_and_frozen_graph(tmp_path):
    """export.py surface (reference export.py:82-137): parameter / FLOP analysis equal to the
    published figures, an inference-only checkpoint, a frozen graph the readers take the same
    weights from (plain and batch-norm-folded), a graph listing."""
    import sys
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'fast-semantic-segmentation_amd')
    sys.path.insert(0, pkg)
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import frozen_graph, graph as G, synth
    from libs.exporter import deploy_segmentation_inference_graph
    import export
    cfg_path = os.path.join(pkg, 'configs', 'two_stage_icnet_0.5_1025_resnet_v1_stage_2.config')
    cfg = pipeline_pb2.PipelineConfig()
    with open(cfg_path) as f:
        text_format.Merge(f.read(), cfg)
    g = G.reset_default_graph()
    _, model = model_builder.build(cfg.model, is_training=False)
    deploy_segmentation_inference_graph(model, [1024, 2048, 3], [1025, 2049])
    weights = synth.make_weights(g.variables, seed=5)
    extra = dict(weights)
    extra['global_step'] = np.asarray(11, np.int64)
    extra['Conv/weights/Momentum'] = np.zeros_like(weights['Conv/weights'])
    src = str(tmp_path / 'train' / 'model.ckpt-11')
    os.makedirs(os.path.dirname(src))
    checkpoint.save_variables(src, extra)
    out = str(tmp_path / 'export')
    params, flops = export.main(['--config_path', cfg_path, '--trained_checkpoint', src, '--output_dir', out,
                                 '--input_shape', '1024,2048,3', '--pad_to_shape', '1025,2049'])
    assert params == 6708032 + 19                  # docs/model_zoo.md: 6.71 M (kernels + the classifier's biases)
    assert abs(flops / 1e9 - 56.33) < 0.01         # 2 x the multiply-adds of the half-width model's convolutions
    assert sorted(os.listdir(out)) == ['checkpoint', 'export_graph.pbtxt', 'frozen_inference_graph.pb',
                                       'model.ckpt.data-00000-of-00001', 'model.ckpt.index']
    assert checkpoint.latest_checkpoint(out) == os.path.join(out, 'model.ckpt')
    saved = checkpoint.load_variables(os.path.join(out, 'model.ckpt'))
    assert set(saved) == set(weights)              # inference variables only
    frozen = frozen_graph.load_weights(os.path.join(out, 'frozen_inference_graph.pb'), g.variables)
    assert set(frozen) == set(weights)
    for k in weights:
        np.testing.assert_array_equal(frozen[k], weights[k])
        np.testing.assert_array_equal(saved[k], weights[k])
    listing = open(os.path.join(out, 'export_graph.pbtxt')).read()
    assert listing.count('op: "conv2d"') == sum(1 for nd in g.nodes if nd.kind == 'conv2d')
    # the Model Zoo form, exported from the frozen graph just written
    out2 = str(tmp_path / 'export_folded')
    export.main(['--config_path', cfg_path, '--trained_checkpoint', os.path.join(out, 'frozen_inference_graph.pb'),
                 '--output_dir', out2, '--input_shape', '1024,2048,3', '--pad_to_shape', '1025,2049',
                 '--fold_batch_norm'])
    folded = frozen_graph.load_weights(os.path.join(out2, 'frozen_inference_graph.pb'), g.variables)
    assert not any('/BatchNorm/' in k for k in folded) and any(k.endswith('/folded_bias') for k in folded)
    scope = 'CascadeFeatureFusion/DilatedConv'
    s = weights[scope + '/BatchNorm/gamma'] / np.sqrt(weights[scope + '/BatchNorm/moving_variance'].astype(np.float64) + 1e-5)
    np.testing.assert_allclose(folded[scope + '/weights'], weights[scope + '/weights'] * s, rtol=1e-6)
    # round trip from the folded graph: its export is a checkpoint (and a plain frozen graph) that
    # load like a trained model's, with batch norm statistics that give the folded bias back ...
    out3 = str(tmp_path / 'export_from_folded')
    export.main(['--config_path', cfg_path, '--trained_checkpoint', os.path.join(out2, 'frozen_inference_graph.pb'),
                 '--output_dir', out3, '--input_shape', '1024,2048,3', '--pad_to_shape', '1025,2049'])
    for src3 in (os.path.join(out3, 'model.ckpt'), os.path.join(out3, 'frozen_inference_graph.pb')):
        back = frozen_graph.load_weights(src3, g.variables)
        assert set(back) == set(weights)
        np.testing.assert_array_equal(back[scope + '/weights'], folded[scope + '/weights'])
        sc = back[scope + '/BatchNorm/gamma'] / np.sqrt(back[scope + '/BatchNorm/moving_variance'].astype(np.float64) + 1e-5)
        np.testing.assert_allclose(sc, 1.0, rtol=0, atol=1e-7)
        np.testing.assert_allclose(back[scope + '/BatchNorm/beta'] - back[scope + '/BatchNorm/moving_mean'] * sc,
                                   folded[scope + '/folded_bias'], rtol=1e-6, atol=1e-7)
    # ... and folding a folded graph again changes nothing
    out4 = str(tmp_path / 'export_folded_twice')
    export.main(['--config_path', cfg_path, '--trained_checkpoint', os.path.join(out2, 'frozen_inference_graph.pb'),
                 '--output_dir', out4, '--input_shape', '1024,2048,3', '--pad_to_shape', '1025,2049',
                 '--fold_batch_norm'])
    twice = frozen_graph.load_weights(os.path.join(out4, 'frozen_inference_graph.pb'), g.variables)
    assert set(twice) == set(folded)
    for k in folded:
        np.testing.assert_array_equal(twice[k], folded[k])


def test_create_cityscapes_tfrecord_tool(tmp_path):
    """dataset_tools/create_cityscapes_tfrecord.py surface (the reference's flags, directory
    layout, file patterns, output name and feature keys): the record it writes is what
    builders.dataset_builder reads back, in sorted file order, with the PNG bytes untouched."""
    import sys
    from PIL import Image
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'fast-semantic-segmentation_amd')
    sys.path.insert(0, pkg)
    from builders import dataset_builder
    from dataset_tools import create_cityscapes_tfrecord as tool
    from protos import input_reader_pb2
    rng = np.random.default_rng(9)
    root = tmp_path / 'cityscapes'
    frames = {}
    for city, stem in (('ulm', 'ulm_000001_000019'), ('aachen', 'aachen_000003_000019'), ('ulm', 'ulm_000000_000019')):
        img = rng.integers(0, 256, (12, 20, 3)).astype(np.uint8)
        lab = rng.integers(0, 19, (12, 20)).astype(np.uint8)
        lab[rng.random(lab.shape) < 0.1] = 255
        for kind, arr, suffix in (('leftImg8bit', img, '_leftImg8bit.png'), ('gtFine', lab, '_gtFine_labelTrainIds.png')):
            d = root / kind / 'val' / city
            d.mkdir(parents=True, exist_ok=True)
            Image.fromarray(arr).save(str(d / (stem + suffix)))
        # a file of the same frame the label pattern must not pick up
        Image.fromarray(lab).save(str(root / 'gtFine' / 'val' / city / (stem + '_gtFine_labelIds.png')))
        frames[stem] = (img, lab)
    out = tool.main(['--cityscapes_dir', str(root), '--split_type', 'val', '--output_dir', str(tmp_path / 'records')])
    assert out == str(tmp_path / 'records' / 'cityscapes_val.record')
    records = list(tfrecord.read_records(out, verify=True))
    assert len(records) == 3
    order = sorted(frames, key=lambda s: str(root / 'leftImg8bit' / 'val' / s.split('_')[0] / s))
    for rec, stem in zip(records, order):
        ex = tfrecord.parse_example(rec)
        path = str(root / 'leftImg8bit' / 'val' / stem.split('_')[0] / (stem + '_leftImg8bit.png'))
        assert ex['image/filename'] == [path.encode()]
        assert ex['image/format'] == [b'png'] and ex['image/segmentation/class/format'] == [b'png']
        assert (ex['image/height'], ex['image/width'], ex['image/channels']) == ([12], [20], [3])
        assert ex['image/encoded'] == [open(path, 'rb').read()]
    cfg = input_reader_pb2.InputReader()
    cfg.tf_record_input_reader.input_path.append(out)
    for (img, lab), stem in zip(dataset_builder.build(cfg), order):
        np.testing.assert_array_equal(img, frames[stem][0])
        np.testing.assert_array_equal(lab, frames[stem][1])
    # explicit patterns; mismatching counts are refused
    out2 = tool.main(['--input_pattern', str(root / 'leftImg8bit' / 'val' / 'ulm' / '*_leftImg8bit.png'),
                      '--annot_pattern', str(root / 'gtFine' / 'val' / 'ulm' / '*_labelTrainIds.png'),
                      '--split_type', 'train', '--output_dir', str(tmp_path / 'records')])
    assert os.path.basename(out2) == 'cityscapes_train.record' and len(list(tfrecord.read_records(out2))) == 2
    with pytest.raises(ValueError, match='do not have image counts'):
        tool.main(['--input_pattern', str(root / 'leftImg8bit' / 'val' / '*' / '*.png'),
                   '--annot_pattern', str(root / 'gtFine' / 'val' / 'ulm' / '*_labelTrainIds.png'),
                   '--split_type', 'x', '--output_dir', str(tmp_path / 'records')])
