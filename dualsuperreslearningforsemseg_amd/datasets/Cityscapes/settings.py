"""Cityscapes constants the hot path consumes (reference datasets/Cityscapes/settings.py:3-30, restated as data)."""
NUM_CLASSES = 19
MEAN = (0.28690, 0.32513, 0.28389)
STD = (0.17614, 0.18099, 0.17772)
IGNORE_CLASS_LABEL = 255
# raw Cityscapes label id -> train id (datasets/Cityscapes/settings.py:9-18)
LABEL_MAPPING_DICT = {
    **{k: IGNORE_CLASS_LABEL for k in (0, 1, 2, 3, 4, 5, 6, 9, 10, 14, 15, 16, 18, 29, 30, -1)},
    7: 0, 8: 1, 11: 2, 12: 3, 13: 4, 17: 5, 19: 6, 20: 7, 21: 8, 22: 9, 23: 10, 24: 11, 25: 12, 26: 13, 27: 14, 28: 15,
    31: 16, 32: 17, 33: 18,
}
