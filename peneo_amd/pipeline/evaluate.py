"""The evaluation pass: batches -> model -> decode -> metrics (reference: ``PEneoTrainer.prediction_loop``
pipeline/trainer.py:57-211 and the ``compute_metrics`` closure of start/run_rfund.py:242-301).

The reference runs this inside a HuggingFace ``Trainer`` subclass; the model-facing part of it is small and is what
BASELINE config 1 exercises end to end (RFUND json -> ``RFUNDDataset`` -> ``DataCollatorForPEneo`` -> ``PEneoModel`` ->
``decode_peneo`` -> ``calculate_KVPE_metric``), so it is restated here without the Trainer: the same per-batch reads of the
model output (``orig_bbox.tolist()``, the five score maps split along the batch dimension, ``<x>_loss.mean().item()`` of
the LAST batch), the same arguments to ``decode_peneo`` and the same metric keys.  The score maps stay on the device: the
decode front end (``peneo_spots_compact``) reduces each ``[P, C]`` map to its few spots there."""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional

import torch

from ..model.peneo_decoder import HandshakingTaggingScheme
from .decode import decode_peneo
from .evaluation import calculate_detail_KVPE_metric, calculate_KVPE_metric

_OUTPUTS = ("line_extraction", "ent_linking_h2h", "ent_linking_t2t", "line_grouping_h2h", "line_grouping_t2t")
_TAGS = ("line_extraction", "ent_linking_head_rel", "ent_linking_tail_rel", "line_grouping_head_rel", "line_grouping_tail_rel")


def _to_device(batch: Dict[str, object], device) -> Dict[str, object]:
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


def make_compute_metrics(detail_eval: bool = False, start_eval_epoch: int = 0,
                         on_detail: Optional[Callable[[dict], None]] = None) -> Callable:
    """The ``compute_metrics(p, epoch)`` of start/run_rfund.py:242-301: decode predictions and ground truth, score the
    key/value pairs (``detail_eval``: all six tasks); before ``start_eval_epoch`` the metric is all zeros.  ``on_detail``
    receives the detail dictionary (the reference dumps it to ``detail.json`` on the main process)."""
    tagger = HandshakingTaggingScheme()

    def compute_metrics(p, epoch: int = 0) -> Dict[str, float]:
        predictions, label_ids, file_ids = p
        if epoch < start_eval_epoch:
            return {"precision": 0.0, "recall": 0.0, "f1": 0.0}
        tags, (gt_relations, orig_bboxes, texts) = label_ids[:5], label_ids[5:]
        all_pred, all_gt, all_fname = decode_peneo(
            handshaking_tagger=tagger, texts=texts,
            line_extraction_shaking_outputs=predictions[0], ent_linking_h2h_shaking_outputs=predictions[1],
            ent_linking_t2t_shaking_outputs=predictions[2], line_grouping_h2h_shaking_outputs=predictions[3],
            line_grouping_t2t_shaking_outputs=predictions[4],
            line_extraction_shaking_tags=tags[0], ent_linking_h2h_shaking_tags=tags[1], ent_linking_t2t_shaking_tags=tags[2],
            line_grouping_h2h_shaking_tags=tags[3], line_grouping_t2t_shaking_tags=tags[4],
            orig_bboxes=orig_bboxes, file_ids=file_ids)
        score = calculate_detail_KVPE_metric if detail_eval else calculate_KVPE_metric
        metric, detail = score(all_pred=all_pred, all_gt=all_gt, all_fname=all_fname)
        if on_detail is not None:
            on_detail(detail)
        return metric

    return compute_metrics


@torch.no_grad()
def prediction_loop(model, dataloader: Iterable[Dict[str, object]], compute_metrics: Optional[Callable] = None,
                    epoch: int = 0, metric_key_prefix: str = "eval", device=None) -> Dict[str, float]:
    """One pass over ``dataloader`` (batches of ``DataCollatorForPEneo``) in eval mode; returns the metric dict with every
    key prefixed ``<prefix>_`` (pipeline/trainer.py:57-211).  As in the reference the reported losses are those of the last
    batch, and ``<prefix>_line_grouping_h2h_loss`` ends up holding the tail->tail loss (:196-201 assigns that key twice);
    callers that want the separate values read them from the model output."""
    if device is None:
        device = next(model.parameters()).device
    compute_metrics = compute_metrics or make_compute_metrics()
    was_training = model.training
    model.eval()
    file_ids: List[str] = []
    orig_bboxes: List[list] = []
    text: List[list] = []
    relations: List[list] = []
    maps: List[List[torch.Tensor]] = [[] for _ in _OUTPUTS]
    tags: List[List[torch.Tensor]] = [[] for _ in _TAGS]
    outputs = None
    for batch in dataloader:
        if "text" not in batch:
            raise ValueError("No text given in evaluation")
        inputs = _to_device(batch, device)
        outputs = model(**inputs)
        orig_bboxes += outputs.orig_bbox.tolist()
        text += batch["text"]
        file_ids += batch.get("fname", [])
        relations += batch["relations"]
        for k, name in enumerate(_OUTPUTS):
            maps[k] += list(getattr(outputs, name + "_shaking_outputs"))
        for k, name in enumerate(_TAGS):
            key = name + "_shaking_tag"
            if key in batch:
                tags[k] += list(batch[key])
            else:  # sparse labels: rebuild the per-document maps on the device for the ground-truth side of the decode
                from .. import ops
                B, N = inputs["input_ids"].shape[0], outputs.orig_bbox.shape[1]
                tags[k] += list(ops.spots_to_tags(inputs[name + "_matrix_spots"], N, device, B=B))
    model.train(was_training)
    if outputs is None:
        return {}
    metrics = dict(compute_metrics((tuple(maps), (*tags, relations, orig_bboxes, text), file_ids), epoch))
    pre = metric_key_prefix + "_"
    metrics[pre + "loss"] = outputs.loss.mean().item()
    metrics[pre + "line_extraction_loss"] = outputs.line_extraction_loss.mean().item()
    metrics[pre + "ent_linking_h2h_loss"] = outputs.ent_linking_h2h_loss.mean().item()
    metrics[pre + "ent_linking_t2t_loss"] = outputs.ent_linking_t2t_loss.mean().item()
    metrics[pre + "line_grouping_h2h_loss"] = outputs.line_grouping_t2t_loss.mean().item()
    return {(k if k.startswith(pre) else pre + k): v for k, v in metrics.items()}
