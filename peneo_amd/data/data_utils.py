"""Box / text helpers of the data path (reference: data/data_utils.py:7-195).

Host-side integer and string work that sits in front of the device path: everything here runs per document in the DataLoader
workers, the results (token boxes in 0..1000, reading order) are what the embedding and bias kernels index with, so each
function keeps the reference's arithmetic to the last integer (``int()`` truncation, numpy's default argsort for ties, the
jitter's draw order)."""
from __future__ import annotations

import random
from typing import Dict, List, Sequence, Tuple, Union

import numpy as np



def box_two_point_convert(box: Union[List[float], Dict[str, float]]) -> List[float]:
    """8-value polygon (list x0,y0,x1,y1,... or dict with 'x'/'y' in the key names) -> [left, top, right, bottom];
    4-value lists pass through (data/data_utils.py:7-28)."""
    if isinstance(box, list) and len(box) == 4:
        return box
    assert len(box) == 8, "Box should be List or Dict that contains 4 or 8 values."
    if isinstance(box, list):
        xs, ys = box[0::2], box[1::2]
    else:
        xs = [v for k, v in box.items() if "x" in k]
        ys = [v for k, v in box.items() if "x" not in k]
    return [min(xs), min(ys), max(xs), max(ys)]


def normalize_bbox(box: Sequence[float], size: Tuple[float, float]) -> List[int]:
    """Pixel box -> the 0..1000 grid the 2-D position tables are indexed with (data/data_utils.py:31-58):
    ``int(1000 * v / extent)`` truncated toward zero, then clipped."""
    width, height = size
    out = []
    for v, extent in zip(box, (width, height, width, height)):
        out.append(min(max(int((v / extent) * 1000), 0), 1000))
    assert out[2] >= out[0]
    assert out[3] >= out[1]
    return out


def merge_bbox(bbox_list: Sequence[Sequence[float]]) -> List[float]:
    """Smallest box around all of ``bbox_list`` (data/data_utils.py:61-75)."""
    cols = list(zip(*bbox_list))
    return [min(cols[0]), min(cols[1]), max(cols[2]), max(cols[3])]


def sort_boxes(sample: Sequence[Sequence[float]]) -> List[int]:
    """Reading order of line boxes (data/data_utils.py:78-117): order by centre y, cut into rows wherever two consecutive
    centres are at least half the mean box height apart, order each row by centre x.  Returns indices into ``sample``.
    Ties resolve the way numpy's default (unstable) argsort resolves them in the reference, hence the same calls."""
    if len(sample) == 0:
        return []
    boxes = np.array(sample)
    cx = (boxes[:, 0] + boxes[:, 2]) / 2.0
    cy = (boxes[:, 1] + boxes[:, 3]) / 2.0
    half_mean_h = np.sum(boxes[:, 3] - boxes[:, 1]) / (2.0 * float(len(boxes)))
    order = np.argsort(cy)
    gaps = np.diff(cy[order])
    row_starts = [0] + [k + 1 for k in range(len(gaps)) if not (gaps[k] < half_mean_h)] + [len(order)]
    for lo, hi in zip(row_starts[:-1], row_starts[1:]):
        row = order[lo:hi]
        order[lo:hi] = row[np.argsort(cx[row])]
    return order.tolist()


def box_augmentation(bbox: Sequence[float], image_w: int, image_h: int) -> Tuple[int, int, int, int]:
    """Training-time box jitter (data/data_utils.py:120-169): shift by up to 10 % of the width and 30 % of the height,
    clip to the page, round.  Draw order (x direction, y direction, x ratio, y ratio) and the reference's behaviour of
    shifting y DOWN for both y directions are kept, so the same ``random.seed`` gives the same boxes."""
    left, top, right, bot = bbox
    x_sign = 1 if random.randint(0, 1) else -1
    random.randint(0, 1)  # the y direction is drawn but both of its branches add (data/data_utils.py:153-158)
    dx = (right - left) * (random.randint(0, 10) / 100)
    dy = (bot - top) * (random.randint(0, 30) / 100)
    xs = np.clip([left + x_sign * dx, right + x_sign * dx], 0, image_w)
    ys = np.clip([top + dy, bot + dy], 0, image_h)
    return int(round(xs[0])), int(round(ys[0])), int(round(xs[1])), int(round(ys[1]))


_FULLWIDTH = {0x3000: " "}
_FULLWIDTH.update({code: chr(code - 0xFEE0) for code in range(0xFF01, 0xFF5F)})


def string_f2h(text: str) -> str:
    """Full-width forms -> ASCII (U+3000 -> space, U+FF01..U+FF5E -> minus 0xFEE0; data/data_utils.py:172-195)."""
    return text.translate(_FULLWIDTH)
