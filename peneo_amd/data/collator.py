"""Items -> one batch (reference: data/collator.py:10-230, image half transformers' LayoutLMv3ImageProcessor).

Same constructor, same batch: ``input_ids`` / ``attention_mask`` / ``bbox`` / ``orig_bbox`` int64 tensors padded to the
longest item rounded up to ``pad_to_multiple_of`` (8) or to ``max_length``; ``fname`` / ``image_path`` / ``text`` /
``relations`` as Python lists; the five ``*_shaking_tag`` label maps ``[B, P]`` int64 with P = N(N+1)/2, N = padded length
minus the CLS slot (:156-204); ``image`` ``[B, 3, 224, 224]`` fp32 when the backbone has visual embeddings.

Differences in HOW, not in what:
  * the reference builds a Python list of all P pairs plus an N x N list-of-lists per BATCH (O(N^2) host objects, 130 816
    tuples at N = 511) just to look up p(i, j); here p(i, j) = i N - i (i - 1) / 2 + (j - i) is evaluated on the handful of
    spots (``HandshakingTaggingScheme.spots2shaking_tag4batch(seq_len=...)``);
  * ``sparse_tags=True`` (not in the reference) skips the dense maps altogether and ships ``*_matrix_spots`` as
    ``[n, 4]`` int32 rows (b, i, j, tag) + ``shaking_seq_len``: the model scatters them on the device
    (``peneo_spots_to_tags``), so the 5 x 8 B x P label bytes per document never cross PCIe;
  * padding is done here (``pad_token_id`` / ``padding_side`` of the tokenizer) instead of through ``tokenizer.pad``.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ..model.peneo_decoder import HandshakingTaggingScheme

SPOT_KEYS = ("line_extraction", "ent_linking_head_rel", "ent_linking_tail_rel", "line_grouping_head_rel",
             "line_grouping_tail_rel")


class PEneoImageProcessor:
    """Page image -> ``pixel_values``: RGB, bilinear resize to ``size`` x ``size`` (PIL), x / 255, (x - mean) / std with
    mean = std = 0.5, channels first — what ``LayoutLMv3ImageProcessor(apply_ocr=False)`` computes for the reference
    (call site data/collator.py:225-228).  Call signature of an HF image processor: ``proc(images, return_tensors="pt")``."""

    def __init__(self, size: int = 224, image_mean: Sequence[float] = (0.5, 0.5, 0.5),
                 image_std: Sequence[float] = (0.5, 0.5, 0.5), rescale_factor: float = 1 / 255) -> None:
        self.size = size
        self.mean = np.asarray(image_mean, dtype=np.float64).reshape(1, 1, 3)
        self.std = np.asarray(image_std, dtype=np.float64).reshape(1, 1, 3)
        self.rescale_factor = rescale_factor

    def __call__(self, images, return_tensors: Optional[str] = "pt") -> Dict[str, object]:
        from PIL import Image
        if not isinstance(images, (list, tuple)):
            images = [images]
        out = []
        for im in images:
            im = im.convert("RGB").resize((self.size, self.size), resample=Image.BILINEAR)
            x = np.asarray(im).astype(np.float64) * self.rescale_factor
            x = ((x - self.mean) / self.std).astype(np.float32)
            out.append(np.ascontiguousarray(x.transpose(2, 0, 1)))
        arr = np.stack(out)
        return {"pixel_values": torch.from_numpy(arr) if return_tensors == "pt" else arr}


class DataCollatorForPEneo:
    PADDING_TYPE = ["longest", "max_length"]
    NO_BATCH_KEYS: List[str] = []
    NO_TENSOR_KEYS = ["text", "relations"] + [f"{k}_shaking_tag" for k in SPOT_KEYS]

    def __init__(self, tokenizer, image_processor=None, padding: str = "longest", max_length: int = 512,
                 pad_to_multiple_of: int = 8, label_pad_token_id: int = -100, require_image: bool = True,
                 add_cls_token: bool = True, add_sep_token: bool = True, sparse_tags: bool = False) -> None:
        if require_image:
            assert image_processor is not None, "image_processor must be provided if require_image is True"
        assert padding in self.PADDING_TYPE, f"invalid padding type {padding}, must be in {self.PADDING_TYPE}"
        if padding == "max_length":
            assert max_length > 0, f"invalid max_length {max_length}, must be positive"
        self.tokenizer, self.image_processor, self.require_image = tokenizer, image_processor, require_image
        self.padding = padding
        self.max_length = max_length if padding == "max_length" else None
        self.pad_to_multiple_of = pad_to_multiple_of
        self.label_pad_token_id = label_pad_token_id
        self.add_cls_token, self.add_sep_token = add_cls_token, add_sep_token
        self.sparse_tags = sparse_tags

    def _padded_length(self, lengths: Sequence[int]) -> int:
        n = max(lengths) if self.max_length is None else self.max_length
        m = self.pad_to_multiple_of
        if m is not None and n % m != 0:
            n = (n // m + 1) * m
        return n

    def __call__(self, features: List[dict]) -> Dict[str, object]:
        left = getattr(self.tokenizer, "padding_side", "right") != "right"
        pad_id = self.tokenizer.pad_token_id
        lengths = [len(f["input_ids"]) for f in features]
        S = self._padded_length(lengths)

        def pad(seq, filler):
            fill = [filler] * (S - len(seq))
            return fill + list(seq) if left else list(seq) + fill

        batch: Dict[str, object] = {}
        for key in features[0]:  # key order of the dataset item, like BatchEncoding
            if key.endswith("_matrix_spots"):
                continue
            col = [f[key] for f in features]
            if key == "input_ids":
                batch[key] = torch.tensor([pad(v, pad_id) for v in col], dtype=torch.int64)
            elif key in ("bbox", "orig_bbox"):
                batch[key] = torch.tensor([pad(v, [0, 0, 0, 0]) for v in col], dtype=torch.int64)
            elif key == "labels":
                batch[key] = torch.tensor([pad(v, self.label_pad_token_id) for v in col], dtype=torch.int64)
            else:
                batch[key] = col
        batch["attention_mask"] = torch.tensor([pad([1] * n, 0) for n in lengths], dtype=torch.int64)

        N = S - 1 if self.add_cls_token else S
        for key in SPOT_KEYS:
            spots = [f[f"{key}_matrix_spots"] for f in features]
            if self.sparse_tags:
                rows = [(b, i, j, t) for b, doc in enumerate(spots) for (i, j, t) in doc]
                batch[f"{key}_matrix_spots"] = torch.tensor(rows, dtype=torch.int32).reshape(-1, 4)
            else:
                batch[f"{key}_shaking_tag"] = HandshakingTaggingScheme.spots2shaking_tag4batch(spots, seq_len=N)
        if self.sparse_tags:
            batch["shaking_seq_len"] = N

        if self.require_image:
            from PIL import Image
            images = [Image.open(f["image_path"]).convert("RGB") for f in features]
            batch["image"] = self.image_processor(images, return_tensors="pt")["pixel_values"]
        return batch
