"""world_size-2 gloo rehearsal of the sharded evaluation: frame i -> rank i % W,
per-rank int64 confusion matrices, one all-reduce, mIoU on every rank.
The per-rank matrices come from the oracle here (no GPU); what is under test is
the sharding rule and the reduction path of fss.metrics."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'fast-semantic-segmentation_amd')]

from oracle import metrics as OM            # noqa: E402


def main():
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    frames = 7                              # ragged: ranks get 4 and 3 frames
    full = np.zeros((19, 19), np.int64)
    mine = np.zeros((19, 19), np.int64)
    for i in range(frames):
        rng = np.random.default_rng(100 + i)
        pred = rng.integers(0, 19, 5000)
        lab = rng.integers(0, 19, 5000)
        lab[rng.random(5000) < 0.05] = 255
        cm = OM.confusion_matrix(pred, lab, 19)
        full += cm
        if i % world == rank:
            mine += cm
    from fss import metrics

    class HostMeanIoU(metrics.StreamingMeanIoU):       # same reduction code, CPU storage
        def __init__(self):
            self.num_classes, self.ignore_label = 19, 255
            self.cm = torch.from_numpy(mine.copy())

    m = HostMeanIoU()
    m.all_reduce()
    assert np.array_equal(m.confusion_matrix(), full), 'sharded sum != single-pass matrix'
    assert m.result() == OM.mean_iou(full)
    dist.barrier()
    if rank == 0:
        print('DIST_OK')
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
