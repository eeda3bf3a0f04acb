"""Spots -> lines -> key/value pairs: the host side of PEneo's decoding (second half).

Counterpart of the reference's ``pipeline/decode.py`` (``parse_matrix_spots`` :9-69, ``sample_decode_peneo`` :72-378,
``decode_peneo`` :381-511) with the same call signatures and return values, written against the COMPACT spot lists of
``HandshakingTaggingScheme.get_spots_from_shaking_tag``: on device tensors that call is one fused
softmax / argmax / compaction launch per score map (``peneo_spots_compact``) plus one small copy, instead of the
reference's Python loop over ``nonzero`` with three ``.item()`` syncs per spot; everything after it walks a few hundred
spots on the host and is not worth a kernel.

Semantics kept from the reference (they decide which pairs come out, so they are part of the parity contract):
  * predictions keep ONE successor per head and one head per successor (highest score wins, first wins ties);
    ground truth (``decode_gt``) keeps the first listed successor;
  * tag 2 in a link map means the link runs j -> i (the maps only store the upper triangle);
  * an entity is a chain of lines: follow line_grouping head->head while the tail->tail map agrees with the line
    extraction's tail of the next line; chains stop at self-links and after 1000 hops;
  * a (key, value) pair is emitted only when the LAST tails of both chains are linked in the entity tail->tail map.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from ..model.peneo_decoder import HandshakingTaggingScheme

Spot = Tuple[int, int, int, float]
_MAX_HOPS = 1000


def merge_bbox(boxes: Sequence[Sequence[int]]) -> List[int]:
    """Smallest box around `boxes` (reference: data/data_utils.py:62-76)."""
    xs0, ys0, xs1, ys1 = zip(*boxes)
    return [min(xs0), min(ys0), max(xs1), max(ys1)]


def parse_matrix_spots(matrix_spots: Sequence[Spot], top_score_only: bool = False, triu_mode: bool = False,
                       score_thresh: float = 0) -> Dict[int, object]:
    """[(i, j, tag, score)] -> {head: tail} (``top_score_only``) or {head: [tails]} (reference :9-69)."""
    links: Dict[int, object] = {}
    for i, j, tag, score in matrix_spots:
        if tag == 0 or score < score_thresh:
            continue
        head, tail = (j, i) if (triu_mode and tag == 2) else (i, j)
        if not top_score_only:
            links.setdefault(head, []).append(tail)
        elif head not in links or score > links[head][1]:
            links[head] = (tail, score)
    if not top_score_only:
        return links
    # one outgoing link per head (above), now one incoming link per tail: the best-scored head keeps it
    best_head: Dict[int, Tuple[int, float]] = {}
    for head, (tail, score) in links.items():
        if tail not in best_head or score > best_head[tail][1]:
            best_head[tail] = (head, score)
    return {head: tail for tail, (head, _) in best_head.items()}


def _entity_chain(first_head: int, first_tail: int, line_tail_of: Dict[int, int], next_head_of: Dict[int, int],
                  next_tail_of: Dict[int, int]) -> List[Tuple[int, int]]:
    """Lines [(head, tail)] of the entity that starts with line (first_head, first_tail) (reference :239-288, :300-343)."""
    lines = [(first_head, first_tail)]
    head, tail = first_head, first_tail
    nxt = next_head_of.get(head)
    hops = 0
    while nxt is not None:
        hops += 1
        if hops > _MAX_HOPS or nxt == head:
            break
        nxt_tail = line_tail_of.get(nxt)                 # tail of the next line according to the line extraction
        if nxt_tail is None or next_tail_of.get(tail) != nxt_tail:   # ... must agree with the tail->tail grouping link
            break
        lines.append((nxt, nxt_tail))
        head, tail = nxt, nxt_tail
        nxt = next_head_of.get(head)
    return lines


def sample_decode_peneo(handshaking_tagger: HandshakingTaggingScheme, text: List[str], line_extraction_shaking: torch.Tensor,
                        ent_linking_h2h_shaking: torch.Tensor, ent_linking_t2t_shaking: torch.Tensor,
                        line_grouping_h2h_shaking: torch.Tensor, line_grouping_t2t_shaking: torch.Tensor,
                        bbox: Optional[torch.Tensor] = None, seq_len: Optional[int] = None,
                        shaking_ind2matrix_ind: Optional[List[Tuple[int, int]]] = None, decode_gt: bool = False,
                        score_thresh: float = 0) -> Tuple:
    """One document: five score maps ([P, C] logits) or label maps ([P] tags) -> (kv pairs, lines, and the five link
    dictionaries), exactly the 7-tuple of the reference (:72-378).  ``shaking_ind2matrix_ind`` is accepted for signature
    compatibility; only its length is used (the pair index is closed-form)."""
    if seq_len is None:
        assert shaking_ind2matrix_ind is not None, "seq_len or shaking_ind2matrix_ind must be given"
        P = len(shaking_ind2matrix_ind)
        seq_len = int(((8 * P + 1) ** 0.5 - 1) // 2)
    spots = [handshaking_tagger.get_spots_from_shaking_tag(m, seq_len=seq_len)
             for m in (line_extraction_shaking, ent_linking_h2h_shaking, ent_linking_t2t_shaking,
                       line_grouping_h2h_shaking, line_grouping_t2t_shaking)]
    le_spots, el_h2h_spots, el_t2t_spots, lg_h2h_spots, lg_t2t_spots = spots
    single = not decode_gt
    line_tail_of = parse_matrix_spots(le_spots, top_score_only=single, triu_mode=False, score_thresh=score_thresh)
    lg_tail = parse_matrix_spots(lg_t2t_spots, top_score_only=single, triu_mode=True, score_thresh=score_thresh)
    lg_head = parse_matrix_spots(lg_h2h_spots, top_score_only=single, triu_mode=True, score_thresh=score_thresh)
    if decode_gt:                                         # ground truth: the first listed successor
        line_tail_of = {k: v[0] for k, v in line_tail_of.items()}
        lg_tail = {k: v[0] for k, v in lg_tail.items()}
        lg_head = {k: v[0] for k, v in lg_head.items()}
    boxes = bbox.tolist() if bbox is not None else None

    def span_text(h: int, t: int) -> str:
        return "".join(text[h:t + 1])

    parsed_lines = []
    for h, t in line_tail_of.items():
        parsed_lines.append((span_text(h, t), merge_bbox(boxes[h:t + 1])) if boxes is not None else span_text(h, t))

    el_tail = parse_matrix_spots(el_t2t_spots, top_score_only=False, triu_mode=True, score_thresh=score_thresh)
    el_head: Dict[int, List[int]] = {}
    kv_pairs = []
    for i, j, tag, score in el_h2h_spots:
        if tag == 0 or score < score_thresh:
            continue
        key_head, value_head = (j, i) if tag == 2 else (i, j)
        el_head.setdefault(key_head, []).append(value_head)
        key_tail, value_tail = line_tail_of.get(key_head), line_tail_of.get(value_head)
        if key_tail is None or value_tail is None:
            continue
        key_lines = _entity_chain(key_head, key_tail, line_tail_of, lg_head, lg_tail)
        value_lines = _entity_chain(value_head, value_tail, line_tail_of, lg_head, lg_tail)
        linked_tails = el_tail.get(key_lines[-1][1])
        if linked_tails is None or value_lines[-1][1] not in linked_tails:
            continue
        key_text = "".join(span_text(h, t) for h, t in key_lines).strip()
        value_text = "".join(span_text(h, t) for h, t in value_lines).strip()
        if boxes is not None:
            key_box = merge_bbox([merge_bbox(boxes[h:t + 1]) for h, t in key_lines])
            value_box = merge_bbox([merge_bbox(boxes[h:t + 1]) for h, t in value_lines])
            kv_pairs.append((key_text, value_text, key_box, value_box))
        else:
            kv_pairs.append((key_text, value_text))
    return kv_pairs, parsed_lines, line_tail_of, el_head, el_tail, lg_head, lg_tail


def decode_peneo(handshaking_tagger: HandshakingTaggingScheme, texts, line_extraction_shaking_outputs,
                 ent_linking_h2h_shaking_outputs, ent_linking_t2t_shaking_outputs, line_grouping_h2h_shaking_outputs,
                 line_grouping_t2t_shaking_outputs, line_extraction_shaking_tags, ent_linking_h2h_shaking_tags,
                 ent_linking_t2t_shaking_tags, line_grouping_h2h_shaking_tags, line_grouping_t2t_shaking_tags, orig_bboxes,
                 file_ids):
    """Batch form (reference :381-511): predictions and ground truth of every document -> (preds, gts, file ids)."""
    preds, gts, ids = [], [], []
    outs = (line_extraction_shaking_outputs, ent_linking_h2h_shaking_outputs, ent_linking_t2t_shaking_outputs,
            line_grouping_h2h_shaking_outputs, line_grouping_t2t_shaking_outputs)
    tags = (line_extraction_shaking_tags, ent_linking_h2h_shaking_tags, ent_linking_t2t_shaking_tags,
            line_grouping_h2h_shaking_tags, line_grouping_t2t_shaking_tags)
    for d, (text, orig_bbox, file_id) in enumerate(zip(texts, orig_bboxes, file_ids)):
        if len(texts) == 0:
            continue
        n = len(orig_bbox)
        preds.append(sample_decode_peneo(handshaking_tagger, text, *[o[d] for o in outs], seq_len=n, decode_gt=False))
        gts.append(sample_decode_peneo(handshaking_tagger, text, *[t[d] for t in tags], seq_len=n, decode_gt=True))
        ids.append(file_id)
    return preds, gts, ids
