"""eval.py end to end on the GPU: TFRecord of PNG examples + a `checkpoint` state file naming
two checkpoints -> mIoU per checkpoint, single process and two ranks (one process each, as
torch.distributed.run starts them) whose int64 confusion matrices are all-reduced; plus the
RCCL all-reduce itself on a one-rank process group.

Reference: eval.py:56-147 (checkpoint iteration, flags), builders/dataset_builder.py:18-62
(TFRecord fields), libs/evaluator.py:126-199 (metric over num_examples)."""
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from conftest import PKG, ROOT                       # noqa: E402
from oracle import nets, ops as O, metrics as OM     # noqa: E402
from test_models_gpu import CFG                      # noqa: E402

H, W = 128, 256            # frames; evaluated padded to (H + 1, W + 1)
N_RECORDS, N_EVAL = 9, 7   # the record file holds more examples than eval_config.num_examples


def _png(a):
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(a).save(b, format='PNG')
    return b.getvalue()


def _examples():
    out = []
    for i in range(N_RECORDS):
        rng = np.random.default_rng(100 + i)
        img = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        lab = rng.integers(0, 19, (H, W)).astype(np.uint8)
        lab[rng.random(lab.shape) < 0.05] = 255
        out.append((img, lab))
    return out


def _setup(tmp_path):
    """Writes config, TFRecord and two checkpoints (+ state file); returns (config path,
    train dir, examples, [weights of ckpt-1, weights of ckpt-2])."""
    from google.protobuf import text_format
    from protos import pipeline_pb2
    from builders import model_builder
    from fss import checkpoint, graph as G, synth, tfrecord
    from libs import standard_fields as fields
    examples = _examples()
    rec = str(tmp_path / 'val.record')
    F = fields.TFRecordFields
    tfrecord.write_records(rec, [tfrecord.make_example({
        F.image_encoded: [_png(img)], F.image_format: [b'png'], F.image_filename: [b'f%d' % i],
        F.image_height: [H], F.image_width: [W], F.image_channels: [3],
        F.segmentation_class_encoded: [_png(lab)], F.segmentation_class_format: [b'png']})
        for i, (img, lab) in enumerate(examples)])
    cfg_text = CFG % dict(arch='icnet', ext='dilated_resnet50', fs=0.5, pf=[1.0, 2.0, 3.0, 4.0],
                          h=H + 1, w=W + 1)
    cfg_text = cfg_text.replace('num_examples: 4', 'num_examples: %d' % N_EVAL)
    cfg_text += "\neval_input_reader: { shuffle: false tf_record_input_reader { input_path: '%s' } }\n" % rec
    cfg_path = tmp_path / 'pipeline.config'
    cfg_path.write_text(cfg_text)
    cfg = pipeline_pb2.PipelineConfig()
    text_format.Merge(cfg_text, cfg)
    g = G.reset_default_graph()
    _, model = model_builder.build(cfg.model, is_training=False)
    model.predict(model.preprocess(G.to_float(G.placeholder((1, H + 1, W + 1, 3)))))
    train = tmp_path / 'train'
    train.mkdir()
    all_w, prefixes = [], []
    for step, seed in ((1, 1234), (2, 99)):
        w = synth.make_weights(g.variables, seed=seed)
        prefix = str(train / ('model.ckpt-%d' % step))
        checkpoint.save_variables(prefix, dict(w, global_step=np.asarray(step, np.int64)))
        all_w.append(w)
        prefixes.append(prefix)
    checkpoint.write_checkpoint_state(str(train), prefixes)
    G.reset_default_graph()
    return str(cfg_path), str(train), examples, all_w


def _oracle_labels(weights, img):
    x = nets.preprocess(O.pad_to_bounding_box(img[None].astype(np.float32), H + 1, W + 1), 'dilated_resnet50')
    return O.argmax_channels(nets.icnet_predict(nets.Params(weights), x, filter_scale=2.0))[0]


def _padded(lab):
    p = np.zeros((H + 1, W + 1), np.uint8)
    p[:H, :W] = lab
    return p


def test_eval_cli_tfrecord_checkpoints_single_and_two_ranks(tmp_path):
    cfg_path, train, examples, all_w = _setup(tmp_path)
    import eval as eval_cli
    args = ['--train_dir', train, '--config_path', cfg_path, '--evaluate_all_from_checkpoint', 'model.ckpt-1',
            '--dtype', 'float32', '--batch_size', '2']
    eval_cli.main(args + ['--eval_dir', str(tmp_path / 'eval1')])
    one = json.load(open(str(tmp_path / 'eval1' / 'eval_results.json')))
    assert [os.path.basename(r['checkpoint']) for r in one['results']] == ['model.ckpt-1', 'model.ckpt-2']
    for r, w in zip(one['results'], all_w):
        cm = np.asarray(r['confusion_matrix'], np.int64)
        # every non-ignored pixel of the first N_EVAL examples (padding included) is counted once
        assert cm.sum() == sum(int((_padded(lab) != 255).sum()) for _, lab in examples[:N_EVAL])
        # mIoU is the oracle's formula on that matrix, bit for bit
        assert np.float32(r['mean_iou']) == OM.mean_iou(cm)
        # the matrix against the oracle's forward pass: the float32 label gate (>= 99.9 % of pixels)
        ref = np.zeros((19, 19), np.int64)
        for img, lab in examples[:N_EVAL]:
            ref += OM.confusion_matrix(_oracle_labels(w, img), _padded(lab), 19, 255)
        assert np.abs(cm - ref).sum() <= 2 * 1e-3 * ref.sum()
    assert one['results'][0]['confusion_matrix'] != one['results'][1]['confusion_matrix']

    # two ranks (fresh processes started before any GPU call), both on this box's one GPU,
    # reduction over gloo: the reduced matrix must equal the single-process one bit for bit
    env = dict(os.environ, FSS_DIST_BACKEND='gloo', FSS_DEVICE='0', MASTER_ADDR='127.0.0.1',
               PYTHONPATH=os.pathsep.join([ROOT, PKG, os.environ.get('PYTHONPATH', '')]))
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                          '--master-addr', '127.0.0.1', '--master-port', '29631',
                          os.path.join(PKG, 'eval.py')] + args + ['--eval_dir', str(tmp_path / 'eval2')],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    two = json.load(open(str(tmp_path / 'eval2' / 'eval_results.json')))
    assert two['world_size'] == 2
    for a, b in zip(one['results'], two['results']):
        assert a['confusion_matrix'] == b['confusion_matrix']
        assert a['mean_iou'] == b['mean_iou']


def test_rccl_all_reduce_of_the_confusion_matrix_one_rank():
    """The nccl (= RCCL) branch of StreamingMeanIoU.all_reduce on the GPU, in a process group of
    one rank (this box has one GPU): the collective must run and leave the matrix unchanged."""
    code = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
from fss import metrics
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
assert dist.get_backend() == 'nccl'
m = metrics.StreamingMeanIoU(19, 255)
rng = np.random.default_rng(5)
pred = torch.from_numpy(rng.integers(0, 19, 100000).astype(np.uint8)).cuda()
lab = torch.from_numpy(rng.integers(0, 19, 100000).astype(np.uint8)).cuda()
m.update(pred, lab)
torch.cuda.synchronize()
before = m.confusion_matrix().copy()
m.all_reduce()
torch.cuda.synchronize()
assert before.sum() == 100000 and np.array_equal(m.confusion_matrix(), before)
dist.destroy_process_group()
print('RCCL_OK')
'''
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29633',
               PYTHONPATH=os.pathsep.join([ROOT, PKG, os.environ.get('PYTHONPATH', '')]))
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'RCCL_OK' in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_bench_starts_its_own_ranks():
    """Plain `python bench.py --gpus 2` (no launcher around it): the script starts two fresh ranks
    before any GPU call and relays rank 0's line.  Both ranks compute on this box's one GPU and
    exchange the confusion matrix over gloo; an 8-GPU node runs the same code over RCCL."""
    env = dict(os.environ, FSS_BENCH_BACKEND='gloo', FSS_BENCH_DEVICE='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--batch', '2',
                          '--steps', '2', '--warmup', '1', '--quick'],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['scaling'] == 'weak'
    assert d['value'] > 0 and d['config']['frames_per_step_per_gpu'] == 2
    assert d['roofline']['frac'] > 0 and d['roofline']['launches_per_step'] > 0
