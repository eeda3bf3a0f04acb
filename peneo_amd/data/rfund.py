"""RFUND documents -> model-ready items (reference: data/datasets/rfund.py:12-471, schema docs/documentation.md:196-240).

One item is one page: the token ids of its text lines in reading order, one box per token (the line's box, in pixels and
on the 0..1000 grid), the source substring of every token, the key/value strings of the page, and five lists of sparse
``(i, j, tag)`` spots — the only form of the labels this build needs (``DataCollatorForPEneo`` turns them into the dense
``[B, P]`` maps of the reference contract, or hands them to the device scatter kernel ``peneo_spots_to_tags``).

What an item holds, and why it is this and nothing else, is the parity contract of BASELINE config 1; the behaviours below
are the reference's and decide which spots exist, so they are kept as executed:

  * a line whose fetched token list is empty is dropped; an entity is "empty" only if NO line of the page has been kept
    by the time it ends (:195-197 tests the page-wide list), and its "last line" is the last line kept SO FAR (:199-202);
  * lines are laid out in ``sort_boxes`` order until the next one would reach ``max_token_len`` (``>=``, :236), the rest
    of the page is cut — entities and links that lost their first/last line are skipped, an entity that lost a middle
    line still counts;
  * only question / answer lines get a line-extraction spot (start, end inclusive, 1);
  * links live in the upper triangle: (min, max, 1) when the source precedes the target, else (min, max, 2); equal
    positions give tag 2;
  * head maps join START tokens of first lines (entity links) or of the two lines (line grouping); tail maps join the
    LAST tokens of last lines / of the two lines.
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

from torch.utils.data import Dataset

from .data_utils import box_augmentation, normalize_bbox, sort_boxes, string_f2h

Spot = Tuple[int, int, int]

# characters the tokenizers' vocabularies do not cover in RFUND (check boxes, private-use glyphs, a Greek omicron inside
# "Topic", accented vowels, the en dash): removed or folded before tokenising (:93-107)
_TEXT_REPAIRS = (("\u2610", ""), ("\u2611", ""), ("\uf702", ""), ("\uf703", ""), ("T\u03bfpic", "Topic"), ("\u00e1", "a"),
                 ("\u00e9", "e"), ("\u00ed", "i"), ("\u00f3", "o"), ("\u00fa", "u"), ("\u00fc", "u"), ("\u2013", "-"))


@dataclass
class LineInfo:
    coords: List[float]
    tokens: List[str]
    sos_processed_tokens: List[str]
    category: str
    orig_entity_id: object
    orig_line_id: object
    orig_next_line: Optional[int] = None
    sorted_start_token: Optional[int] = None
    sorted_end_token: Optional[int] = None  # exclusive


@dataclass
class _Page:
    """What the first pass over ``entities`` leaves behind."""
    lines: List[LineInfo] = field(default_factory=list)
    empty_lines: set = field(default_factory=set)
    empty_entities: set = field(default_factory=set)
    entity_text: Dict[object, str] = field(default_factory=dict)
    first_line_of: Dict[object, object] = field(default_factory=dict)
    last_line_of: Dict[object, object] = field(default_factory=dict)
    entity_of_line: Dict[object, object] = field(default_factory=dict)


def _upper(src: int, dst: int) -> Spot:
    """A directed link as an upper-triangle spot."""
    return (src, dst, 1) if src < dst else (dst, src, 2)


class RFUNDDataset(Dataset):
    """``RFUNDDataset(data_root, split, language, tokenizer, tokenizer_fetcher, max_token_len, add_cls_token,
    add_sep_token, apply_box_aug)`` over ``<data_root>/<language>.<train|val>.json`` and ``<data_root>/images/<language>/``."""

    LANG_LIST = ["en", "zh", "ja", "es", "fr", "de", "it", "pt"]
    SPLIT_LIST = ["train", "dev", "test"]
    ENTITY_LABEL_LIST = ["other", "header", "question", "answer"]
    LABEL_LIST = ["O"] + [f"{p}-{name}" for name in ENTITY_LABEL_LIST[1:] for p in ("B", "I")]
    LABEL_NAME2ID = {name: k for k, name in enumerate(LABEL_LIST)}
    LABEL_ID2NAME = {k: name for k, name in enumerate(LABEL_LIST)}

    def __init__(self, data_root: str, split: str, language: str, tokenizer, tokenizer_fetcher: Optional[Callable] = None,
                 max_token_len: int = 511, add_cls_token: bool = False, add_sep_token: bool = False,
                 apply_box_aug: bool = False, **kwargs) -> None:
        super().__init__()
        assert language in self.LANG_LIST, f"Language {language} not supported, should be one of {self.LANG_LIST}"
        assert split in self.SPLIT_LIST, f"Split {split} not supported, should be one of {self.SPLIT_LIST}"
        self.language, self.split = language, split
        if isinstance(tokenizer, str):
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tokenizer)
        elif hasattr(tokenizer, "tokenizer") and not hasattr(tokenizer, "convert_tokens_to_ids"):
            tokenizer = tokenizer.tokenizer  # a processor wrapping (tokenizer, image_processor)
        self.tokenizer = tokenizer
        self.tokenizer_fetcher = tokenizer_fetcher
        self.image_root = os.path.join(data_root, "images", language)
        self.annotation_dir = os.path.join(data_root, f"{language}.{'train' if split == 'train' else 'val'}.json")
        with open(self.annotation_dir, "r", encoding="utf-8") as f:
            self.annotation = json.load(f)["documents"]
        self.max_token_len = max_token_len
        self.add_cls_token, self.add_sep_token = add_cls_token, add_sep_token
        self.apply_box_aug = apply_box_aug

    def __len__(self) -> int:
        return len(self.annotation)

    def _special_text_replace(self, line_text: str) -> str:
        for bad, good in _TEXT_REPAIRS:
            line_text = line_text.replace(bad, good)
        return string_f2h(line_text)

    # ---- what differs between the data sets: where documents come from, how a line's text and box are read ---------------
    def _document(self, index: int) -> dict:
        return self.annotation[index]

    def _image_path(self, fname: str) -> str:
        return os.path.join(self.image_root, fname)

    def _line_text(self, text: str, first: bool) -> str:
        """Later lines of an entity get a leading blank (not in zh / ja), then the character repairs (:130-134)."""
        glue = "" if first or self.language in ("zh", "ja") else " "
        return self._special_text_replace(glue + text)

    def _line_box(self, bbox) -> list:
        return list(bbox)

    # ---- pass 1: tokenise the lines entity by entity (:123-203) -----------------------------------------------------------
    def _collect(self, doc: dict) -> _Page:
        page = _Page()
        width, height = doc["img"]["width"], doc["img"]["height"]
        for entity in doc["entities"]:
            kept_text: List[str] = []
            for line in entity["lines"]:
                text = self._line_text(line["text"], first=not kept_text)
                tokens = self.tokenizer.tokenize(text)
                pieces = self.tokenizer_fetcher(text, tokens) if self.tokenizer_fetcher is not None else tokens
                if len(pieces) == 0:
                    page.empty_lines.add(line["id"])
                    continue
                box = self._line_box(line["bbox"])
                if self.apply_box_aug:
                    box = list(box_augmentation(tuple(box), width, height))
                    for lo, hi in ((0, 2), (1, 3)):  # keep the jittered box at least one pixel wide and high
                        if box[lo] >= box[hi]:
                            if box[hi] == 0:
                                box[lo], box[hi] = 0, 1
                            else:
                                box[lo] = box[hi] - 1
                if not kept_text:
                    page.first_line_of[entity["id"]] = line["id"]
                kept_text.append(text)
                page.lines.append(LineInfo(coords=box, tokens=tokens, sos_processed_tokens=pieces, category=entity["label"],
                                           orig_entity_id=entity["id"], orig_line_id=line["id"]))
                page.entity_of_line[line["id"]] = entity["id"]
            if not page.lines:  # nothing kept on the page yet, not "nothing kept in this entity"
                page.empty_entities.add(entity["id"])
                continue
            page.last_line_of[entity["id"]] = page.lines[-1].orig_line_id
            page.entity_text[entity["id"]] = "".join(kept_text)
        return page

    def __getitem__(self, index):
        doc = self._document(index)
        fname = doc["img"]["fname"]
        size = (doc["img"]["width"], doc["img"]["height"])
        page = self._collect(doc)
        ordered = [page.lines[k] for k in sort_boxes([ln.coords for ln in page.lines])]

        # ---- pass 2: lay the lines out in reading order up to the token budget (:205-275) -------------------------------
        input_ids: List[int] = []
        bbox: List[List[int]] = []
        orig_bbox: List[List[float]] = []
        texts: List[str] = []
        line_spots: List[Spot] = []
        placed: Dict[object, LineInfo] = {}     # line id -> its record, for the lines that fit
        seen_entities = set()
        for ln in ordered:
            ids = self.tokenizer.convert_tokens_to_ids(ln.tokens)
            if len(input_ids) + len(ids) >= self.max_token_len:
                break
            ln.sorted_start_token = len(input_ids)
            ln.sorted_end_token = len(input_ids) + len(ids)
            placed[ln.orig_line_id] = ln
            seen_entities.add(ln.orig_entity_id)
            norm = normalize_bbox(ln.coords, size)
            input_ids.extend(ids)
            bbox.extend([norm] * len(ids))
            orig_bbox.extend([ln.coords] * len(ids))
            texts.extend(ln.sos_processed_tokens)
            if ln.category in ("question", "answer"):
                line_spots.append((ln.sorted_start_token, ln.sorted_end_token - 1, 1))

        # ---- pass 3: links between what was placed (:277-419) -------------------------------------------------------------
        ent_head: List[Spot] = []
        ent_tail: List[Spot] = []
        relations: List[Dict[str, str]] = []
        kv = doc["relations"]["kv_entity"]
        for link in kv:
            q, a = link["from_id"], link["to_id"]
            if q in page.empty_entities or a in page.empty_entities or q not in seen_entities or a not in seen_entities:
                continue
            ends = [page.first_line_of[q], page.first_line_of[a], page.last_line_of[q], page.last_line_of[a]]
            if any(line_id not in placed for line_id in ends):
                continue
            q_first, a_first, q_last, a_last = (placed[line_id] for line_id in ends)
            ent_head.append(_upper(q_first.sorted_start_token, a_first.sorted_start_token))
            ent_tail.append(_upper(q_last.sorted_end_token - 1, a_last.sorted_end_token - 1))
        for link in kv:  # the strings of the metric: every pair whose two entities are (partly) on the page (:421-436)
            q, a = link["from_id"], link["to_id"]
            if q in page.entity_text and a in page.entity_text and q not in page.empty_entities \
                    and a not in page.empty_entities and q in seen_entities and a in seen_entities:
                relations.append({"key": page.entity_text[q], "value": page.entity_text[a]})

        grp_head: List[Spot] = []
        grp_tail: List[Spot] = []
        for link in doc["relations"]["line_grouping"]:
            src, dst = link["from_id"], link["to_id"]
            if src in page.empty_lines or dst in page.empty_lines:
                continue
            if page.entity_of_line.get(src, -1) not in seen_entities or page.entity_of_line.get(dst, -1) not in seen_entities:
                continue
            if src not in placed or dst not in placed:
                continue
            s, d = placed[src], placed[dst]
            grp_head.append(_upper(s.sorted_start_token, d.sorted_start_token))
            grp_tail.append(_upper(s.sorted_end_token - 1, d.sorted_end_token - 1))

        if self.add_cls_token:
            input_ids = [self.tokenizer.cls_token_id] + input_ids
            bbox = [[0, 0, 0, 0]] + bbox
            orig_bbox = [[0, 0, 0, 0]] + orig_bbox
        if self.add_sep_token:
            input_ids = input_ids + [self.tokenizer.sep_token_id]
            bbox = bbox + [[0, 0, 0, 0]]
            orig_bbox = orig_bbox + [[0, 0, 0, 0]]
        assert len(input_ids) == len(bbox), f"bbox_length mismatch {fname}"
        assert len(input_ids) == len(orig_bbox), f"orig_bbox length mismatch {fname}"
        assert len(ent_head) == len(ent_tail), f"entity relation length mismatch {fname}"
        assert len(grp_head) == len(grp_tail), f"line relation length mismatch {fname}"
        return {
            "fname": fname,
            "image_path": self._image_path(fname),
            "input_ids": input_ids,
            "bbox": bbox,
            "orig_bbox": orig_bbox,
            "text": texts,
            "relations": relations,
            "line_extraction_matrix_spots": line_spots,
            "ent_linking_head_rel_matrix_spots": ent_head,
            "ent_linking_tail_rel_matrix_spots": ent_tail,
            "line_grouping_head_rel_matrix_spots": grp_head,
            "line_grouping_tail_rel_matrix_spots": grp_tail,
        }


class SIBRDataset(RFUNDDataset):
    """SIBR (reference: data/datasets/sibr.py:25-460): one json per page under ``<data_root>/converted_label/``, listed by
    ``<data_root>/<split>.txt``, images under ``<data_root>/images/``.  The item is built exactly like an RFUND item (the two
    reference classes share everything from the reading-order pass on); what differs: no language, line texts are tokenised
    as they are (no leading blank for later lines, no character repairs: :118-121) and box coordinates are truncated to
    integers (:137-143)."""

    SPLIT_LIST = ["train", "test"]

    def __init__(self, data_root: str, split: str, tokenizer, tokenizer_fetcher: Optional[Callable] = None,
                 max_token_len: int = 511, add_cls_token: bool = False, add_sep_token: bool = False,
                 apply_box_aug: bool = False, **kwargs) -> None:
        Dataset.__init__(self)
        assert split in self.SPLIT_LIST, f"Split {split} not supported, should be one of {self.SPLIT_LIST}"
        self.split, self.language = split, None
        if isinstance(tokenizer, str):
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(tokenizer)
        elif hasattr(tokenizer, "tokenizer") and not hasattr(tokenizer, "convert_tokens_to_ids"):
            tokenizer = tokenizer.tokenizer
        self.tokenizer, self.tokenizer_fetcher = tokenizer, tokenizer_fetcher
        self.image_root = os.path.join(data_root, "images")
        self.annotation_root = os.path.join(data_root, "converted_label")
        with open(os.path.join(data_root, f"{split}.txt"), "r") as f:
            self.annotation_fname_list = [os.path.basename(x.strip()) for x in f.readlines()]
        self.max_token_len = max_token_len
        self.add_cls_token, self.add_sep_token = add_cls_token, add_sep_token
        self.apply_box_aug = apply_box_aug

    def __len__(self) -> int:
        return len(self.annotation_fname_list)

    def _document(self, index: int) -> dict:
        with open(os.path.join(self.annotation_root, self.annotation_fname_list[index]), "r", encoding="utf-8") as f:
            return json.load(f)

    def _line_text(self, text: str, first: bool) -> str:
        return text

    def _line_box(self, bbox) -> list:
        return [int(v) for v in bbox]
