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
    assert np.array_equal(logits1[0], logits2[1])
    assert np.array_equal(labels1[0], labels2[1])
