"""Key/value-pair metrics over decoded documents (reference: pipeline/evaluation.py:6-665).

Input: what ``decode_peneo`` returns — per document a 7-tuple (kv pairs, lines, line texts, entity-link heads, entity-link
tails, line-grouping heads, line-grouping tails) for the prediction and for the ground truth.  Output: micro-averaged
precision / recall / F1 over the data set (counts summed over documents, one count per file name: a distributed sampler
pads the last batch by repeating documents) and a ``detail`` dictionary with per-document counts and TP / FP / FN lists.

Under ``torch.distributed`` every rank scores its own documents and the per-file COUNTS (a few integers per document, via
``all_gather_object``) are the only thing exchanged — the one collective of the eval path.

The six tasks of the detailed metric are one table here instead of six copies of the same arithmetic."""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple, Union

import torch.distributed as dist


def _prf(num_correct: float, num_pred: float, num_gt: float) -> Tuple[float, float, float]:
    precision = num_correct / num_pred if num_pred > 0 else 0.0
    recall = num_correct / num_gt if num_gt > 0 else 0.0
    f1 = (2 * precision * recall) / (precision + recall) if precision + recall > 0 else 0.0
    return precision, recall, f1


def _calculate_linking_metric_core(pred: Union[Dict, List], gt: Union[Dict, List]):
    """(precision, recall, f1, #pred, #gt, #correct) of (head, tail) links; dicts are read as {head: tail} (:6-42)."""
    pred = list(pred.items()) if isinstance(pred, dict) else pred
    gt = list(gt.items()) if isinstance(gt, dict) else gt
    num_correct = float(sum(1 for item in pred if item in gt))
    return (*_prf(num_correct, float(len(pred)), float(len(gt))), float(len(pred)), float(len(gt)), num_correct)


def _calculate_KV_metric_core(pred: List, gt: List, return_detail: bool = False):
    """Same six numbers for lists of predictions; with ``return_detail`` a seventh: one {"status": TP|FP, "pred"} per
    prediction in order, then one {"status": FN, "gt"} per ground truth that no prediction matched (:45-95)."""
    hits = [p in gt for p in pred]
    num_correct = float(sum(hits))
    out = (*_prf(num_correct, float(len(pred)), float(len(gt))), float(len(pred)), float(len(gt)), num_correct)
    if not return_detail:
        return out
    detail = [{"status": "TP" if hit else "FP", "pred": p} for p, hit in zip(pred, hits)]
    matched = [p for p, hit in zip(pred, hits) if hit]
    detail += [{"status": "FN", "gt": g} for g in gt if g not in matched]
    return (*out, detail)


def _gather_rows(rows: List[list]) -> List[list]:
    """Per-file count rows of all ranks, first occurrence of a file name wins (:150-178)."""
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, rows)
    else:
        parts = [rows]
    seen, out = set(), []
    for part in parts:
        for row in part:
            if row[0] not in seen:
                seen.add(row[0])
                out.append(row)
    return out


def _pairs_of_lists(links: Dict[int, Sequence[int]]) -> List[Tuple[int, int]]:
    return [(head, tail) for head, tails in links.items() for tail in tails]


def _pairs(links: Dict[int, int]) -> List[Tuple[int, int]]:
    return list(links.items())


# (name in `detail`, index in the decoded 7-tuple, how to list its items, scoring core)
_TASKS = (
    ("kv_pair", 0, list, _calculate_KV_metric_core),
    ("line_extraction", 1, list, _calculate_KV_metric_core),
    ("ent_linking_head", 3, _pairs_of_lists, _calculate_linking_metric_core),
    ("ent_linking_tail", 4, _pairs_of_lists, _calculate_linking_metric_core),
    ("line_grouping_head", 5, _pairs, _calculate_linking_metric_core),
    ("line_grouping_tail", 6, _pairs, _calculate_linking_metric_core),
)


def _stats(p: float, r: float, f1: float, n_pred: float, n_gt: float, n_correct: float, counts_first: bool) -> dict:
    counts = {"num_pred": n_pred, "num_gt": n_gt, "num_correct": n_correct}
    scores = {"precision": p, "recall": r, "f1": f1}
    return {**counts, **scores} if counts_first else {**scores, **counts}


def calculate_KVPE_metric(all_pred: List[Tuple], all_gt: List[Tuple], all_fname: List[str]):
    """({"precision", "recall", "f1"} of the key/value pairs, detail) (:98-207)."""
    samples, rows = [], []
    for fname, pred, gt in zip(all_fname, all_pred, all_gt):
        p, r, f1, n_pred, n_gt, n_correct, info = _calculate_KV_metric_core(pred[0], gt[0], return_detail=True)
        samples.append({"fname": fname, **_stats(p, r, f1, n_pred, n_gt, n_correct, True), "detail": info})
        rows.append([fname, n_pred, n_gt, n_correct])
    rows = _gather_rows(rows)
    n_pred = sum((row[1] for row in rows), 0.0)
    n_gt = sum((row[2] for row in rows), 0.0)
    n_correct = sum((row[3] for row in rows), 0.0)
    p, r, f1 = _prf(n_correct, n_pred, n_gt)
    detail = {**_stats(p, r, f1, n_pred, n_gt, n_correct, False), "num_sample_processed": len(rows), "detail": samples}
    return {"precision": p, "recall": r, "f1": f1}, detail


def calculate_detail_KVPE_metric(all_pred: List[Tuple], all_gt: List[Tuple], all_fname: List[str]):
    """As above plus the five intermediate tasks; the metric dict has ``precision`` / ``recall`` / ``f1`` for the pairs and
    ``<task>_precision`` ... for the others (:210-665)."""
    samples, rows = [], []
    for fname, pred, gt in zip(all_fname, all_pred, all_gt):
        sample = {"fname": fname}
        row = [fname]
        kv_info = None
        for name, slot, items, core in _TASKS:
            if name == "kv_pair":
                *six, kv_info = core(items(pred[slot]), items(gt[slot]), return_detail=True)
            else:
                six = core(items(pred[slot]), items(gt[slot]))
            sample[name] = _stats(*six, True)
            row += [six[3], six[4], six[5]]
        sample["detail"] = kv_info
        samples.append(sample)
        rows.append(row)
    rows = _gather_rows(rows)
    metric, detail = {}, {}
    for k, (name, _, _, _) in enumerate(_TASKS):
        n_pred = sum((row[1 + 3 * k] for row in rows), 0.0)
        n_gt = sum((row[2 + 3 * k] for row in rows), 0.0)
        n_correct = sum((row[3 + 3 * k] for row in rows), 0.0)
        p, r, f1 = _prf(n_correct, n_pred, n_gt)
        detail[name] = _stats(p, r, f1, n_pred, n_gt, n_correct, False)
        prefix = "" if name == "kv_pair" else name + "_"
        metric.update({prefix + "precision": p, prefix + "recall": r, prefix + "f1": f1})
    detail["detail"] = samples
    return metric, detail
