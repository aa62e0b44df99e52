"""Cityscapes train-id -> colour / label-id tables (values as in the
reference's libs/constants.py:6-47, which inference.py:133-134 selects from)."""

CITYSCAPES_LABEL_COLORS = [
    (128, 64, 128), (244, 35, 231), (69, 69, 69), (102, 102, 156), (190, 153, 153),
    (153, 153, 153), (250, 170, 29), (219, 219, 0), (106, 142, 3), (152, 250, 152),
    (69, 129, 180), (219, 19, 60), (255, 0, 0), (0, 0, 142), (0, 0, 69), (0, 60, 100),
    (0, 79, 100), (0, 0, 230), (119, 10, 3)]

CITYSCAPES_LABEL_IDS = [(i,) for i in (7, 8, 11, 12, 13, 17, 19, 20, 21, 22, 23, 24, 25, 26, 27,
                                       28, 31, 32, 33)]
