"""CPU oracle for the PEneo forward/backward hot path.

TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this package; the shipped
``peneo_amd`` package never does (its ops raise if the HIP library is missing).

This is a from-scratch, *functional* restatement in plain fp32 PyTorch of what the
reference computes on its CPU path.  It deliberately shares no code with
``peneo_amd``: it consumes a reference-layout ``state_dict`` (keys ``backbone.*`` /
``peneo_decoder.*``) plus a plain config dict, and returns plain tensors.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference
(``/root/reference``, transformers-5 compat shims in ``tests/golden/_ref_import.py``)
in the build container, runs it on seeded inputs and stores inputs/outputs as fixtures
under ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this oracle against
those fixtures (the reference ships no tests / golden vectors of its own).

Reference citations (relative to the upstream repo root):

* LayoutLMv3 embeddings ............ model/backbone/layoutlmv3/modeling_layoutlmv3.py:131-227
* patch embedding / forward_image .. :51-84, :910-931
* visual bbox ...................... :879-908
* relative position buckets / bias . :586-676
* self-attention (cogview softmax) . :308-410
* Roberta SelfOutput/Intermediate/Output (transformers 4.40.1, un-vendored)
* LayoutLMv3Model.forward .......... :934-1164
* LiLT ............................. model/backbone/lilt/modeling_lilt.py
* PEneoModel.forward (crop) ........ model/modeling_peneo.py:108-175
* HandshakingKernel ................ model/peneo_decoder.py:118-177
* PEneoDecoder ..................... model/peneo_decoder.py:201-443
* CrossEntropyLossOHEM fast path ... model/custom_loss.py:189-202
* CrossEntropyLossOHEM OHEM branch . model/custom_loss.py:204-288 (``ohem_ce``)
* spots decoding ................... model/peneo_decoder.py:76-115
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

HEAD_NAMES = (
    "line_extraction",
    "ent_linking_h2h",
    "ent_linking_t2t",
    "line_grouping_h2h",
    "line_grouping_t2t",
)
# kwarg names under which the five label maps reach PEneoDecoder.forward
# (model/peneo_decoder.py:338-348)
TAG_KEYS = (
    "line_extraction_shaking_tag",
    "ent_linking_head_rel_shaking_tag",
    "ent_linking_tail_rel_shaking_tag",
    "line_grouping_head_rel_shaking_tag",
    "line_grouping_tail_rel_shaking_tag",
)


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _linear(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    b = sd.get(prefix + ".bias")
    return F.linear(x, sd[prefix + ".weight"], b)


def _layer_norm(x: Tensor, sd: Dict[str, Tensor], prefix: str, eps: float) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def _dropout(x: Tensor, p: float, training: bool) -> Tensor:
    return F.dropout(x, p, training) if (training and p > 0) else x


# --------------------------------------------------------------------------------------
# relative-position buckets (modeling_layoutlmv3.py:586-613)
# --------------------------------------------------------------------------------------
def relative_position_bucket(relative_position: Tensor, num_buckets: int, max_distance: int) -> Tensor:
    """Bidirectional T5-style bucket of an int64 tensor of signed distances."""
    num_buckets //= 2
    ret = (relative_position > 0).long() * num_buckets
    n = torch.abs(relative_position)
    max_exact = num_buckets // 2
    is_small = n < max_exact
    val_if_large = max_exact + (
        torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (num_buckets - max_exact)
    ).to(torch.long)
    val_if_large = torch.min(val_if_large, torch.full_like(val_if_large, num_buckets - 1))
    return ret + torch.where(is_small, n, val_if_large)


def visual_bbox(grid: int = 14, max_len: int = 1000) -> Tensor:
    """[1 + grid*grid, 4] int64: cls box then the patch grid (:879-901)."""
    xs = torch.div(torch.arange(0, max_len * (grid + 1), max_len), grid, rounding_mode="trunc")
    ys = xs.clone()
    box = torch.stack(
        [
            xs[:-1].repeat(grid, 1),
            ys[:-1].repeat(grid, 1).transpose(0, 1),
            xs[1:].repeat(grid, 1),
            ys[1:].repeat(grid, 1).transpose(0, 1),
        ],
        dim=-1,
    ).view(-1, 4)
    cls_box = torch.tensor([[1, 1, max_len - 1, max_len - 1]])
    return torch.cat([cls_box, box], dim=0)


# --------------------------------------------------------------------------------------
# LayoutLMv3 backbone
# --------------------------------------------------------------------------------------
def layoutlmv3_text_embeddings(sd, cfg, input_ids: Tensor, bbox: Tensor, training=False, p="backbone.") -> Tensor:
    """K1 — modeling_layoutlmv3.py:131-227."""
    pad = cfg["pad_token_id"]
    mask = input_ids.ne(pad).int()
    position_ids = (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad
    e = p + "embeddings."
    emb = F.embedding(input_ids, sd[e + "word_embeddings.weight"], padding_idx=pad)
    emb = emb + F.embedding(torch.zeros_like(input_ids), sd[e + "token_type_embeddings.weight"])
    emb = emb + F.embedding(position_ids, sd[e + "position_embeddings.weight"], padding_idx=pad)
    if not (torch.all(0 <= bbox) and torch.all(bbox <= 1023)):
        raise IndexError("The :obj:`bbox` coordinate values should be within 0-1000 range.")
    x_w, y_w = sd[e + "x_position_embeddings.weight"], sd[e + "y_position_embeddings.weight"]
    h_w, w_w = sd[e + "h_position_embeddings.weight"], sd[e + "w_position_embeddings.weight"]
    spatial = torch.cat(
        [
            F.embedding(bbox[:, :, 0], x_w),
            F.embedding(bbox[:, :, 1], y_w),
            F.embedding(bbox[:, :, 2], x_w),
            F.embedding(bbox[:, :, 3], y_w),
            F.embedding(torch.clip(bbox[:, :, 3] - bbox[:, :, 1], 0, 1023), h_w),
            F.embedding(torch.clip(bbox[:, :, 2] - bbox[:, :, 0], 0, 1023), w_w),
        ],
        dim=-1,
    )
    emb = emb + spatial
    emb = _layer_norm(emb, sd, e + "LayerNorm", cfg["layer_norm_eps"])
    return _dropout(emb, cfg["hidden_dropout_prob"], training)


def layoutlmv3_image_embeddings(sd, cfg, image: Tensor, p="backbone.") -> Tensor:
    """K2 — PatchEmbed (:69-84) + forward_image (:910-931)."""
    x = F.conv2d(image, sd[p + "patch_embed.proj.weight"], sd[p + "patch_embed.proj.bias"], stride=16)
    x = x.flatten(2).transpose(1, 2)
    cls = sd[p + "cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat((cls, x), dim=1) + sd[p + "pos_embed"]
    return F.layer_norm(x, (x.shape[-1],), sd[p + "norm.weight"], sd[p + "norm.bias"], 1e-6)


def layoutlmv3_rel_bias(sd, cfg, position_ids: Tensor, bbox: Tensor, as_executed=False, p="backbone.") -> Optional[Tensor]:
    """K4 — summed 1-D + 2-D bias [B, heads, T, T] (before the /sqrt(d)) (:615-676)."""
    out = None
    enc = p + "encoder."
    if cfg.get("has_relative_attention_bias", False):
        rel = position_ids.unsqueeze(-2) - position_ids.unsqueeze(-1)
        bk = relative_position_bucket(rel, cfg["rel_pos_bins"], cfg["max_rel_pos"])
        w = sd[enc + "rel_pos_bias.weight"]  # [heads, bins]
        if as_executed:
            out = F.linear(F.one_hot(bk, cfg["rel_pos_bins"]).to(w.dtype), w).permute(0, 3, 1, 2).contiguous()
        else:
            out = w.t()[bk].permute(0, 3, 1, 2)
    if cfg.get("has_spatial_attention_bias", False):
        x = bbox[:, :, 0]
        y = bbox[:, :, 3]
        bx = relative_position_bucket(x.unsqueeze(-2) - x.unsqueeze(-1), cfg["rel_2d_pos_bins"], cfg["max_rel_2d_pos"])
        by = relative_position_bucket(y.unsqueeze(-2) - y.unsqueeze(-1), cfg["rel_2d_pos_bins"], cfg["max_rel_2d_pos"])
        wx, wy = sd[enc + "rel_pos_x_bias.weight"], sd[enc + "rel_pos_y_bias.weight"]
        if as_executed:
            rx = F.linear(F.one_hot(bx, cfg["rel_2d_pos_bins"]).to(wx.dtype), wx).permute(0, 3, 1, 2).contiguous()
            ry = F.linear(F.one_hot(by, cfg["rel_2d_pos_bins"]).to(wy.dtype), wy).permute(0, 3, 1, 2).contiguous()
        else:
            rx = wx.t()[bx].permute(0, 3, 1, 2)
            ry = wy.t()[by].permute(0, 3, 1, 2)
        r2 = rx + ry
        out = r2 if out is None else out + r2
    return out


def _cogview_softmax(scores: Tensor, alpha: float = 32.0) -> Tensor:
    """modeling_layoutlmv3.py:308-321 (numerically == softmax)."""
    scaled = scores / alpha
    mx = scaled.amax(dim=-1).unsqueeze(-1)
    return torch.softmax((scaled - mx) * alpha, dim=-1)


def layoutlmv3_layer(sd, cfg, x: Tensor, ext_mask: Tensor, rel_bias: Optional[Tensor], prefix: str,
                     training=False, capture: Optional[dict] = None) -> Tensor:
    """K5–K8 — one encoder layer (:323-410, :443-529 + Roberta blocks)."""
    B, T, H = x.shape
    nh = cfg["num_attention_heads"]
    d = H // nh
    a = prefix + "attention."
    q = _linear(x, sd, a + "self.query").view(B, T, nh, d).permute(0, 2, 1, 3)
    k = _linear(x, sd, a + "self.key").view(B, T, nh, d).permute(0, 2, 1, 3)
    v = _linear(x, sd, a + "self.value").view(B, T, nh, d).permute(0, 2, 1, 3)
    scores = torch.matmul(q / math.sqrt(d), k.transpose(-1, -2))
    if rel_bias is not None:
        scores = scores + rel_bias / math.sqrt(d)
    scores = scores + ext_mask
    probs = _cogview_softmax(scores)
    probs = _dropout(probs, cfg["attention_probs_dropout_prob"], training)
    ctx = torch.matmul(probs, v).permute(0, 2, 1, 3).contiguous().view(B, T, H)
    if capture is not None:
        capture["ctx"] = ctx
    h = _dropout(_linear(ctx, sd, a + "output.dense"), cfg["hidden_dropout_prob"], training)
    attn_out = _layer_norm(h + x, sd, a + "output.LayerNorm", cfg["layer_norm_eps"])
    inter = F.gelu(_linear(attn_out, sd, prefix + "intermediate.dense"))
    h = _dropout(_linear(inter, sd, prefix + "output.dense"), cfg["hidden_dropout_prob"], training)
    return _layer_norm(h + attn_out, sd, prefix + "output.LayerNorm", cfg["layer_norm_eps"])


def layoutlmv3_forward(sd, cfg, input_ids: Tensor, bbox: Tensor, attention_mask: Tensor,
                       image: Optional[Tensor] = None, training=False, as_executed=False,
                       capture: Optional[dict] = None, p="backbone.") -> Tensor:
    """LayoutLMv3Model.forward for the PEneo call pattern (:934-1164) -> [B, T, H]."""
    B, S = input_ids.shape
    dev = input_ids.device
    emb = layoutlmv3_text_embeddings(sd, cfg, input_ids, bbox, training, p)
    if capture is not None:
        capture["text_emb"] = emb
    has_bias = cfg.get("has_relative_attention_bias", False) or cfg.get("has_spatial_attention_bias", False)
    final_bbox = final_pos = None
    if image is not None:
        vis = layoutlmv3_image_embeddings(sd, cfg, image, p)
        nv = vis.shape[1]
        attention_mask = torch.cat([attention_mask, torch.ones((B, nv), dtype=attention_mask.dtype, device=dev)], dim=1)
        if has_bias:
            grid = int(cfg.get("input_size", 224) / 16)
            vb = visual_bbox(grid).to(dev).unsqueeze(0).repeat(B, 1, 1)
            final_bbox = torch.cat([bbox, vb], dim=1)
            final_pos = torch.cat(
                [torch.arange(S, device=dev).unsqueeze(0).expand(B, S),
                 torch.arange(nv, device=dev).unsqueeze(0).repeat(B, 1)], dim=1)
        emb = torch.cat([emb, vis], dim=1)
        emb = _layer_norm(emb, sd, p + "LayerNorm", cfg["layer_norm_eps"])
        emb = _dropout(emb, cfg["hidden_dropout_prob"], training)
    elif has_bias:
        final_bbox = bbox
        final_pos = torch.arange(S, device=dev).unsqueeze(0).expand(B, S)
    if capture is not None:
        capture["emb"] = emb
    ext_mask = (1.0 - attention_mask[:, None, None, :].to(emb.dtype)) * torch.finfo(emb.dtype).min
    rel_bias = layoutlmv3_rel_bias(sd, cfg, final_pos, final_bbox, as_executed, p) if has_bias else None
    if capture is not None and rel_bias is not None:
        capture["rel_bias"] = rel_bias
    x = emb
    for i in range(cfg["num_hidden_layers"]):
        cap = {} if (capture is not None and i == 0) else None
        x = layoutlmv3_layer(sd, cfg, x, ext_mask, rel_bias, f"{p}encoder.layer.{i}.", training, cap)
        if cap is not None:
            capture["layer0_ctx"] = cap["ctx"]
            capture["layer0_out"] = x
    return x


# --------------------------------------------------------------------------------------
# LiLT backbone (model/backbone/lilt/modeling_lilt.py)
# --------------------------------------------------------------------------------------
def lilt_forward(sd, cfg, input_ids: Tensor, bbox: Tensor, attention_mask: Tensor, training=False,
                 p="backbone.") -> Tensor:
    """LiltModel.forward (:855-997) -> cat(text, layout) [B, S, H + H/r].

    Text stream: RoBERTa embeddings (:39-130).  Layout stream: six 2-D position
    embeddings -> Linear -> + box position embeddings -> LN (:133-210).  Each layer
    (:269-429, :540-660): both streams build Q/K/V; the score matrices are summed
    (BiACM) and each stream soft-maxes the *same* summed scores.
    """
    B, S = input_ids.shape
    H = cfg["hidden_size"]
    r = cfg["channel_shrink_ratio"]
    Hl = H // r
    nh = cfg["num_attention_heads"]
    d, dl = H // nh, Hl // nh
    eps = cfg["layer_norm_eps"]
    pd = cfg["hidden_dropout_prob"]
    pad = cfg["pad_token_id"]

    # text embeddings (:75-110)
    mask = input_ids.ne(pad).int()
    position_ids = (torch.cumsum(mask, dim=1).type_as(mask) * mask).long() + pad
    e = p + "embeddings."
    x = F.embedding(input_ids, sd[e + "word_embeddings.weight"], padding_idx=pad)
    x = x + F.embedding(torch.zeros_like(input_ids), sd[e + "token_type_embeddings.weight"])
    x = x + F.embedding(position_ids, sd[e + "position_embeddings.weight"], padding_idx=pad)
    x = _dropout(_layer_norm(x, sd, e + "LayerNorm", eps), pd, training)

    # layout embeddings (:160-210)
    le = p + "layout_embeddings."
    xw, yw = sd[le + "x_position_embeddings.weight"], sd[le + "y_position_embeddings.weight"]
    hw, ww = sd[le + "h_position_embeddings.weight"], sd[le + "w_position_embeddings.weight"]
    spatial = torch.cat(
        [
            F.embedding(bbox[:, :, 0], xw),
            F.embedding(bbox[:, :, 1], yw),
            F.embedding(bbox[:, :, 2], xw),
            F.embedding(bbox[:, :, 3], yw),
            F.embedding(bbox[:, :, 3] - bbox[:, :, 1], hw),
            F.embedding(bbox[:, :, 2] - bbox[:, :, 0], ww),
        ],
        dim=-1,
    )
    l = _linear(spatial, sd, le + "box_linear_embeddings")
    l = l + F.embedding(position_ids, sd[le + "box_position_embeddings.weight"], padding_idx=pad)
    l = _dropout(_layer_norm(l, sd, le + "LayerNorm", eps), pd, training)

    ext_mask = (1.0 - attention_mask[:, None, None, :].to(x.dtype)) * torch.finfo(x.dtype).min

    def heads(t, n, dd):
        return t.view(B, S, n, dd).permute(0, 2, 1, 3)

    pa = cfg["attention_probs_dropout_prob"]
    for i in range(cfg["num_hidden_layers"]):
        lp = f"{p}encoder.layer.{i}."
        a = lp + "attention.self."
        q, k, v = (heads(_linear(x, sd, a + n), nh, d) for n in ("query", "key", "value"))
        ql, kl, vl = (heads(_linear(l, sd, a + n), nh, dl) for n in ("layout_query", "layout_key", "layout_value"))
        s_t = torch.matmul(q, k.transpose(-1, -2))
        s_l = torch.matmul(ql, kl.transpose(-1, -2))
        tmp_t = s_t / math.sqrt(d)
        tmp_l = s_l / math.sqrt(dl)
        s_t = tmp_t + tmp_l
        s_l = tmp_l + tmp_t
        p_l = _dropout(torch.softmax(s_l + ext_mask, dim=-1), pa, training)
        ctx_l = torch.matmul(p_l, vl).permute(0, 2, 1, 3).contiguous().view(B, S, Hl)
        p_t = _dropout(torch.softmax(s_t + ext_mask, dim=-1), pa, training)
        ctx_t = torch.matmul(p_t, v).permute(0, 2, 1, 3).contiguous().view(B, S, H)
        # self-output, both streams (:432-470)
        o = lp + "attention."
        at = _layer_norm(_dropout(_linear(ctx_t, sd, o + "output.dense"), pd, training) + x, sd, o + "output.LayerNorm", eps)
        al = _layer_norm(_dropout(_linear(ctx_l, sd, o + "layout_output.dense"), pd, training) + l, sd, o + "layout_output.LayerNorm", eps)
        # feed-forward, both streams (:620-660)
        it = F.gelu(_linear(at, sd, lp + "intermediate.dense"))
        x = _layer_norm(_dropout(_linear(it, sd, lp + "output.dense"), pd, training) + at, sd, lp + "output.LayerNorm", eps)
        il = F.gelu(_linear(al, sd, lp + "layout_intermediate.dense"))
        l = _layer_norm(_dropout(_linear(il, sd, lp + "layout_output.dense"), pd, training) + al, sd, lp + "layout_output.LayerNorm", eps)
    return torch.cat([x, l], dim=-1)


# --------------------------------------------------------------------------------------
# PEneo decoder
# --------------------------------------------------------------------------------------
def pair_index(n: int) -> Tuple[Tensor, Tensor]:
    """Row-major upper-triangular (i <= j) enumeration used by the reference
    (peneo_decoder.py:129-147): p(i, j) = i*n - i*(i-1)/2 + (j - i)."""
    idx = torch.triu_indices(n, n)
    return idx[0], idx[1]


def handshaking(sd, seq: Tensor, as_executed=False, p="peneo_decoder.") -> Tensor:
    """K11 — HandshakingKernel.forward (peneo_decoder.py:149-177) -> [B, P, D]."""
    B, N, D = seq.shape
    w = sd[p + "handshaking_kernel.combine_fc.weight"]
    b = sd[p + "handshaking_kernel.combine_fc.bias"]
    ii, jj = pair_index(N)
    ii, jj = ii.to(seq.device), jj.to(seq.device)
    if as_executed:
        m = torch.cat([seq.unsqueeze(2).repeat(1, 1, N, 1), seq.unsqueeze(1).repeat(1, N, 1, 1)], dim=-1)
        m = m.permute(0, 3, 1, 2).flatten(-2)[..., N * ii + jj].permute(0, 2, 1)
        return F.silu(F.linear(m, w, b))
    a = F.linear(seq, w[:, :D])
    c = F.linear(seq, w[:, D:], b)
    return F.silu(a[:, ii] + c[:, jj])


def _classifier(sd, x: Tensor, prefix: str, num_layers: int, pdrop: float, training: bool) -> Tensor:
    """build_classifier (peneo_decoder.py:231-271)."""
    if num_layers == 1:
        return _linear(x, sd, prefix)
    idx = 0
    for _ in range(num_layers - 1):
        x = _dropout(F.silu(_linear(x, sd, f"{prefix}.{idx}")), pdrop, training)
        idx += 3
    return _linear(x, sd, f"{prefix}.{idx}")


def weighted_ce(logits: Tensor, target: Tensor, weight: Optional[Tensor]) -> Tensor:
    """CrossEntropyLossOHEM fast path (custom_loss.py:189-202): class-weighted mean."""
    return F.cross_entropy(logits.float().view(-1, logits.shape[-1]), target.view(-1), weight=weight)


def ohem_ce(logits: Tensor, target: Tensor, weight: Optional[Tensor], num_hard_positive: int, num_hard_negative: int) -> Tensor:
    """CrossEntropyLossOHEM.forward with OHEM active, reduction "mean", random=False (custom_loss.py:204-288), stated with
    explicit ranks instead of the reference's tensor indexing:

    * per-element weighted CE; positives = target != 0, negatives = target == 0, each kept in flattened order (:236-238);
    * each class is sorted by descending loss; k = min(count, num_hard) (:259-262, :269-272);
    * k <= 0 -> the class keeps all its elements; k < count -> the reference evaluates ``sorted[idx[:k]]`` (:265-267,
      :275-277), i.e. for each of the k hardest elements, with position j inside its class list, the j-th largest loss of
      the class is what enters the sum (the sorted array is indexed with positions of the unsorted one);
    * loss = (kept positives + kept negatives) / (k_pos + k_neg) with the k's as computed, even when <= 0 (:279-283).
    """
    ce = F.cross_entropy(logits.float().view(-1, logits.shape[-1]), target.view(-1), weight=weight, reduction="none")
    tgt = target.view(-1)
    total, ks = 0.0, []
    for cls_mask, num_hard in ((tgt != 0, num_hard_positive), (tgt == 0, num_hard_negative)):
        vals = ce[cls_mask]                                   # flattened order
        n = vals.shape[0]
        k = min(n, num_hard)
        ks.append(k)
        if 0 < k < n:
            order = torch.argsort(vals, descending=True, stable=True)     # order[r] = list position of the r-th largest
            by_rank = vals[order]                                          # by_rank[r] = r-th largest loss
            vals = by_rank[order[:k]]                                      # ranks taken from list positions: as executed
        total = total + vals.sum()
    return total / (ks[0] + ks[1])


def decoder_forward(sd, pcfg, seq: Tensor, tags: Optional[Sequence[Tensor]] = None, training=False,
                    as_executed=False, capture: Optional[dict] = None, p="peneo_decoder.") -> Dict[str, Tensor]:
    """PEneoDecoder.forward (peneo_decoder.py:338-443)."""
    bcfg = pcfg["backbone_config"]
    pdrop = bcfg["hidden_dropout_prob"]
    if pcfg.get("peneo_decoder_shrink", True):
        seq = _dropout(F.silu(_linear(seq, sd, p + "shrink_projection.0")), pdrop, training)
        seq = _dropout(F.silu(_linear(seq, sd, p + "shrink_projection.3")), pdrop, training)
    if capture is not None:
        capture["shrunk"] = seq
    shaking = handshaking(sd, seq, as_executed, p)
    if capture is not None:
        capture["shaking"] = shaking
    nl = pcfg.get("peneo_classifier_num_layers", 2)
    out: Dict[str, Tensor] = {}
    for name in HEAD_NAMES:
        out[name + "_shaking_outputs"] = _classifier(sd, shaking, f"{p}{name}_fc", nl, pdrop, training)
    if tags is None:
        return out
    cw = pcfg.get("peneo_category_weights", [1.0, 1.0, 1.0])
    link_w = torch.tensor(cw, dtype=torch.float32, device=seq.device)
    le_w = torch.tensor(cw[:-1], dtype=torch.float32, device=seq.device)
    ratios = pcfg.get("peneo_loss_ratio") or [1.0] * 5
    total = 0.0
    for name, tag, ratio in zip(HEAD_NAMES, tags, ratios):
        w = le_w if name == "line_extraction" else link_w
        hp, hn = pcfg.get("peneo_ohem_num_positive", -1), pcfg.get("peneo_ohem_num_negative", -1)
        if hp == -1 and hn == -1:
            l = weighted_ce(out[name + "_shaking_outputs"], tag, w)
        else:
            l = ohem_ce(out[name + "_shaking_outputs"], tag, w, hp, hn)
        out[name + "_loss"] = l
        total = total + ratio * l
    out["loss"] = total
    return out


# --------------------------------------------------------------------------------------
# PEneoModel
# --------------------------------------------------------------------------------------
def peneo_forward(sd: Dict[str, Tensor], pcfg: dict, batch: Dict[str, Tensor], training=False,
                  as_executed=False, capture: Optional[dict] = None) -> Dict[str, Tensor]:
    """PEneoModel.forward (model/modeling_peneo.py:108-175).

    ``pcfg`` is the PEneoConfig as a dict (``backbone_name``, ``backbone_config`` dict, peneo_*).
    ``batch`` holds input_ids, bbox, orig_bbox, attention_mask, optionally image and the five tags.
    """
    name = pcfg["backbone_name"].lower()
    bcfg = pcfg["backbone_config"]
    ids, bbox, mask = batch["input_ids"], batch["bbox"], batch["attention_mask"]
    S = ids.shape[1]
    if "layoutlmv3" in name:
        hidden = layoutlmv3_forward(sd, bcfg, ids, bbox, mask, batch.get("image"), training, as_executed, capture)
        hidden = hidden[:, 1:S]                     # drop CLS and the visual tokens (:138-147)
    elif "lilt" in name:
        hidden = lilt_forward(sd, bcfg, ids, bbox, mask, training)
        hidden = hidden[:, 1:]                      # drop CLS (:156-163)
    else:
        raise ValueError(f"oracle has no backbone {name}")
    if capture is not None:
        capture["sequence_output"] = hidden
    hidden = _dropout(hidden, bcfg["hidden_dropout_prob"], training)
    tags = None
    if all(k in batch and batch[k] is not None for k in TAG_KEYS):
        tags = [batch[k] for k in TAG_KEYS]
    out = decoder_forward(sd, pcfg, hidden, tags, training, as_executed, capture)
    ob = batch.get("orig_bbox")
    out["orig_bbox"] = ob[:, 1:S] if ob is not None else None
    return out


# --------------------------------------------------------------------------------------
# K14 — spots from a score map (peneo_decoder.py:76-115)
# --------------------------------------------------------------------------------------
def spots_from_logits(logits: Tensor) -> List[Tuple[int, int, int, float]]:
    """[P, C] logits -> [(i, j, tag, score)] for argmax != 0, in increasing p order."""
    P = logits.shape[0]
    n = int((math.isqrt(8 * P + 1) - 1) // 2)
    ii, jj = pair_index(n)
    prob = logits.softmax(-1)
    pred = prob.argmax(-1)
    score = prob.max(-1)[0]
    out = []
    for pidx in torch.nonzero(pred)[:, 0].tolist():
        out.append((int(ii[pidx]), int(jj[pidx]), int(pred[pidx]), float(score[pidx])))
    return out


def spots_to_tag(spots, n: int) -> Tensor:
    """HandshakingTaggingScheme.spots2shaking_tag4batch for one sample (peneo_decoder.py:35-73)."""
    tag = torch.zeros(n * (n + 1) // 2, dtype=torch.long)
    for (i, j, t) in spots:
        tag[i * n - i * (i - 1) // 2 + (j - i)] = t
    return tag
