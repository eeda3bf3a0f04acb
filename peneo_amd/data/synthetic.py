"""Synthetic RFUND-shaped batches.

The reference's ``RFUNDDataset`` (data/datasets/rfund.py:244-419) turns an annotated
document into ``seq_len`` tokens, a per-token bbox that is the *line* box replicated
over the line's tokens (:257-258) and five lists of ``(i, j, tag)`` spots; its collator
(data/collator.py:156-230) scatters the spots into five dense ``[B, P]`` int64 label
maps over the packed upper triangle, ``P = N (N + 1) / 2`` with ``N = seq_len - 1``.
This module produces batches with exactly those keys / dtypes / value ranges from a
seed, with no tokenizer, images or dataset on disk.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

TAG_KEYS = (
    "line_extraction_shaking_tag",
    "ent_linking_head_rel_shaking_tag",
    "ent_linking_tail_rel_shaking_tag",
    "line_grouping_head_rel_shaking_tag",
    "line_grouping_tail_rel_shaking_tag",
)


def spots_to_shaking_tag(spots: Sequence[Tuple[int, int, int]], n: int) -> torch.Tensor:
    """(i, j, tag) spots -> packed label map, p(i, j) = i*n - i(i-1)/2 + (j - i)
    (reference: HandshakingTaggingScheme.spots2shaking_tag4batch, model/peneo_decoder.py:35-73)."""
    tag = torch.zeros(n * (n + 1) // 2, dtype=torch.int64)
    for i, j, t in spots:
        assert 0 <= i <= j < n
        tag[i * n - i * (i - 1) // 2 + (j - i)] = t
    return tag


def _one_doc(rng: np.random.Generator, seq_len: int, n_lines: int, vocab: int, ragged: bool,
             add_sep: bool, pad_id: int):
    max_tok = seq_len - (2 if add_sep else 1)
    ntok = int(rng.integers(max(n_lines, int(0.6 * max_tok)), max_tok + 1)) if ragged else max_tok
    n_lines = min(n_lines, ntok)
    ids = np.full(seq_len, pad_id, dtype=np.int64)
    ids[0] = 0
    ids[1:1 + ntok] = rng.integers(3, vocab, size=ntok)
    used = 1 + ntok
    if add_sep:
        ids[used] = 2
        used += 1
    mask = np.zeros(seq_len, dtype=np.int64)
    mask[:used] = 1
    # lines: n_lines - 1 cut points in the ntok tokens
    cuts = np.sort(rng.choice(np.arange(1, ntok), size=n_lines - 1, replace=False)) if n_lines > 1 else np.array([], dtype=np.int64)
    starts = np.concatenate([[0], cuts]).astype(np.int64)
    ends = np.concatenate([cuts, [ntok]]).astype(np.int64) - 1  # inclusive
    bbox = np.zeros((seq_len, 4), dtype=np.int64)
    step = max(1, 990 // max(n_lines, 1))
    for k in range(n_lines):
        x0 = int(rng.integers(0, 801))
        box = [x0, step * k, x0 + 150, min(1000, step * k + 8)]
        bbox[1 + starts[k]:1 + ends[k] + 1] = box
    # label spots (token indices are CLS-dropped: token t of the doc is index t)
    le = [(int(starts[k]), int(ends[k]), 1) for k in range(0, n_lines, 2)]
    n_links = max(1, n_lines // 4)

    def links():
        h2h, t2t = [], []
        for _ in range(n_links):
            a, b = rng.choice(n_lines, size=2, replace=False)
            tag = 1
            if a > b:
                a, b, tag = b, a, 2
            h2h.append((int(starts[a]), int(starts[b]), tag))
            t2t.append((int(ends[a]), int(ends[b]), tag))
        return h2h, t2t

    el_h, el_t = links()
    lg_h, lg_t = links()
    return ids, mask, bbox, (le, el_h, el_t, lg_h, lg_t)


def synthetic_rfund_batch(batch_size: int, seq_len: int = 512, n_lines: int = 128, vocab_size: int = 50265,
                          seed: int = 0, ragged: bool = False, with_image: bool = True, add_sep: bool = True,
                          pad_id: int = 1, image_size: int = 224, pad_to_longest: bool = False) -> Dict[str, torch.Tensor]:
    """Batch with the keys ``DataCollatorForPEneo`` emits (data/collator.py:205-230).  ``pad_to_longest`` (with ``ragged``):
    the batch is cut to its longest document rounded up to a multiple of 8, as the reference's collator pads
    (``padding="longest"``, ``pad_to_multiple_of=8``, data/collator.py:110-116); the label maps follow that length."""
    rng = np.random.default_rng(seed)
    ids, masks, boxes, all_spots = [], [], [], []
    for _ in range(batch_size):
        i, m, b, spots = _one_doc(rng, seq_len, n_lines, vocab_size, ragged, add_sep, pad_id)
        ids.append(i)
        masks.append(m)
        boxes.append(b)
        all_spots.append(spots)
    L = seq_len
    if pad_to_longest:
        L = min(seq_len, -(-max(int(m.sum()) for m in masks) // 8) * 8)
        ids, masks, boxes = [i[:L] for i in ids], [m[:L] for m in masks], [b[:L] for b in boxes]
    n = L - 1
    tags = [[spots_to_shaking_tag(sp[h], n) for sp in all_spots] for h in range(5)]
    batch = {
        "input_ids": torch.from_numpy(np.stack(ids)),
        "attention_mask": torch.from_numpy(np.stack(masks)),
        "bbox": torch.from_numpy(np.stack(boxes)),
    }
    batch["orig_bbox"] = batch["bbox"].clone()
    for k, t in zip(TAG_KEYS, tags):
        batch[k] = torch.stack(t)
    if with_image:
        g = torch.Generator().manual_seed(seed)
        batch["image"] = torch.randn(batch_size, 3, image_size, image_size, generator=g)
    return batch
