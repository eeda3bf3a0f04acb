"""LiLT backbone (BASELINE config 5) on libpeneo_hip kernels.

Reference: model/backbone/lilt/modeling_lilt.py — text stream (hidden H) and layout stream (H / r) with
shared attention scores (BiACM, :398-404).  Parameter names reproduce the reference
(``embeddings.*``, ``layout_embeddings.*``, ``encoder.layer.{i}.attention.self.{query,key,value,layout_query,...}``,
``attention.output`` / ``attention.layout_output``, ``intermediate`` / ``layout_intermediate``,
``output`` / ``layout_output``).

Device mapping: per layer two fused QKV GEMMs (text, layout); q/k/v of both streams are concatenated per head
(``peneo_head_concat``, q pre-scaled by 1/sqrt(d) resp. 1/sqrt(d_l)) so ONE flash-attention call over head dim
d + d_l produces the summed scores, one softmax and both context streams; the rest is the same
GEMM(+bias/GELU/residual/dropout) + LayerNorm chain as LayoutLMv3, once per stream.  Deviation in *train* mode
only: the reference draws two independent attention-dropout masks for the two (identical) probability
matrices; here both streams share one mask.
"""
from __future__ import annotations

import math
import os
from typing import List

import torch
import torch.nn as nn

from .. import ops
from ..hip import ACT_GELU, PeneoHipError
from .configuration_peneo import LiltConfig
from .engine import DropoutSeeds, WeightCache, zeros_like_param, zeros_like_params
from .engine import can_defer, defer_join, join_pending
from .engine import side_stream as engine_side_stream


class _SelfParams(nn.Module):
    def __init__(self, H: int, Hl: int):
        super().__init__()
        self.query, self.key, self.value = nn.Linear(H, H), nn.Linear(H, H), nn.Linear(H, H)
        self.layout_query, self.layout_key, self.layout_value = nn.Linear(Hl, Hl), nn.Linear(Hl, Hl), nn.Linear(Hl, Hl)


class _DenseLN(nn.Module):
    def __init__(self, fan_in: int, fan_out: int, eps: float):
        super().__init__()
        self.dense = nn.Linear(fan_in, fan_out)
        self.LayerNorm = nn.LayerNorm(fan_out, eps=eps)


class _Dense(nn.Module):
    def __init__(self, fan_in: int, fan_out: int):
        super().__init__()
        self.dense = nn.Linear(fan_in, fan_out)


class _AttentionParams(nn.Module):
    def __init__(self, cfg, H, Hl):
        super().__init__()
        self.self = _SelfParams(H, Hl)
        self.output = _DenseLN(H, H, cfg.layer_norm_eps)
        self.layout_output = _DenseLN(Hl, Hl, cfg.layer_norm_eps)


class LiltLayer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H, r = cfg.hidden_size, cfg.channel_shrink_ratio
        Hl, I, Il = H // r, cfg.intermediate_size, cfg.intermediate_size // r
        self.attention = _AttentionParams(cfg, H, Hl)
        self.intermediate = _Dense(H, I)
        self.output = _DenseLN(I, H, cfg.layer_norm_eps)
        self.layout_intermediate = _Dense(Hl, Il)
        self.layout_output = _DenseLN(Il, Hl, cfg.layer_norm_eps)


class LiltEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([LiltLayer(cfg) for _ in range(cfg.num_hidden_layers)])


class LiltTextEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H = cfg.hidden_size
        self.word_embeddings = nn.Embedding(cfg.vocab_size, H, padding_idx=cfg.pad_token_id)
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, H, padding_idx=cfg.pad_token_id)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, H)
        self.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(cfg.max_position_embeddings).expand((1, -1)))


class LiltLayoutEmbeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H, Hl = cfg.hidden_size, cfg.hidden_size // cfg.channel_shrink_ratio
        for n in ("x", "y", "h", "w"):
            setattr(self, f"{n}_position_embeddings", nn.Embedding(cfg.max_2d_position_embeddings, H // 6))
        self.box_position_embeddings = nn.Embedding(cfg.max_position_embeddings, Hl, padding_idx=cfg.pad_token_id)
        self.box_linear_embeddings = nn.Linear(H, Hl)
        self.LayerNorm = nn.LayerNorm(Hl, eps=cfg.layer_norm_eps)


# ------------------------------------------------------------------------------------------------
# shared per-stream pieces: everything of a transformer layer after the attention context
# ------------------------------------------------------------------------------------------------
def _post_attn_fwd(wc: WeightCache, key: str, dt, eps, seeds: DropoutSeeds, site: int, x, att, params):
    wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2 = params
    Wo, Wi, Wo2 = wc.cast(key + ".o", wo, dt), wc.cast(key + ".i", wi, dt), wc.cast(key + ".o2", wo2, dt)
    h1 = ops.gemm(att, Wo, bias=bo, residual=x, drop_p=seeds.p_hidden, drop_seed=seeds.seed(site))
    a, m1, r1 = ops.layernorm_fwd(h1, g1, b1, eps)
    zi = torch.empty((x.shape[0], wi.shape[0]), dtype=dt, device=x.device)
    inter = ops.gemm(a, Wi, bias=bi, act=ACT_GELU, preact=zi)
    h2 = ops.gemm(inter, Wo2, bias=bo2, residual=a, drop_p=seeds.p_hidden, drop_seed=seeds.seed(site + 1))
    out, m2, r2 = ops.layernorm_fwd(h2, g2, b2, eps)
    return out, (att, h1, m1, r1, a, zi, inter, h2, m2, r2)


def _post_attn_bwd(wc: WeightCache, key: str, dt, seeds: DropoutSeeds, site: int, d_out, saved, params, on_side):
    """-> (d_att, d_x_residual, grads in parameter order).  ``on_side(fn, operands)`` runs the parameter-gradient work
    (wgrad GEMMs, bias column sums) on the stage's second stream, off the activation-gradient critical path; ``operands``
    are the tensors that work READS (kept alive until the join).  Its outputs are never kept: they are returned as
    gradients, and an extra reference would make AccumulateGrad clone them on the main stream instead of storing them."""
    wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2 = params
    att, h1, m1, r1, a, zi, inter, h2, m2, r2 = saved
    Wo, Wi, Wo2 = wc.cast(key + ".o", wo, dt), wc.cast(key + ".i", wi, dt), wc.cast(key + ".o2", wo2, dt)
    dev = d_out.device
    Hs, Is = wo.shape[0], wi.shape[0]
    wgrad = lambda dy, xin: ops.gemm(dy, xin, a_kmajor=False, b_kmajor=False, out_dtype=torch.float32)
    pool = torch.zeros(4 * Hs + (Hs + Is + Hs), dtype=torch.float32, device=dev)   # one fill for all small accumulators
    dg2, db2, dg1, db1 = pool[:Hs], pool[Hs:2 * Hs], pool[2 * Hs:3 * Hs], pool[3 * Hs:4 * Hs]
    dbo2, dbi, dbo = pool[4 * Hs:5 * Hs], pool[5 * Hs:5 * Hs + Is], pool[5 * Hs + Is:]
    d_dense2 = torch.empty_like(h2) if seeds.p_hidden > 0 else None
    d_h2 = ops.layernorm_bwd(d_out.contiguous(), h2, g2, m2, r2, dg2, db2, dx_dropped=d_dense2, drop2_p=seeds.p_hidden,
                             drop2_seed=seeds.seed(site + 1))
    if d_dense2 is None:
        d_dense2 = d_h2
    _, dwo2 = on_side(lambda: (ops.colsum(d_dense2, out=dbo2, accumulate=True), wgrad(d_dense2, inter)),
                      (d_dense2, inter))
    d_zi = ops.gemm(d_dense2, Wo2, b_kmajor=False, grad_src=zi, grad_act=ACT_GELU)
    _, dwi = on_side(lambda: (ops.colsum(d_zi, out=dbi, accumulate=True), wgrad(d_zi, a)), (d_zi, a))
    d_a = ops.gemm(d_zi, Wi, b_kmajor=False, residual=d_h2)
    d_dense1 = torch.empty_like(h1) if seeds.p_hidden > 0 else None
    d_h1 = ops.layernorm_bwd(d_a, h1, g1, m1, r1, dg1, db1, dx_dropped=d_dense1, drop2_p=seeds.p_hidden,
                             drop2_seed=seeds.seed(site))
    if d_dense1 is None:
        d_dense1 = d_h1
    _, dwo = on_side(lambda: (ops.colsum(d_dense1, out=dbo, accumulate=True), wgrad(d_dense1, att)),
                     (d_dense1, att))
    d_att = ops.gemm(d_dense1, Wo, b_kmajor=False)
    return d_att, d_h1, (dwo, dbo, dg1, db1, dwi, dbi, dwo2, dbo2, dg2, db2)


class _State:
    def __init__(self):
        self.key_bias = None
        self.seeds = None
        self.dtype = torch.float32
        self.dims = None


class _LiltEmbedStage(torch.autograd.Function):
    """text: word + type + pos -> LN (+dropout); layout: cat(6 tables) -> Linear -> + box_pos[pid] -> LN (+dropout)."""

    @staticmethod
    def forward(ctx, model, st, input_ids, bbox, *params):
        (word, type_w, pos_w, ln_g, ln_b, xw, yw, hw, ww, boxpos, lin_w, lin_b, lln_g, lln_b) = params
        cfg, wc, dt = model.config, model.weight_cache, st.dtype
        B, S = input_ids.shape
        H, Hl = cfg.hidden_size, cfg.hidden_size // cfg.channel_shrink_ratio
        dev = input_ids.device
        seeds = st.seeds
        # the bbox range check of the reference is the layout embedding kernel's sticky device flag (see LayoutLMv3's _EmbedStage):
        # check_inputs = True reads it at once (one host sync), "deferred" leaves it to raise_on_bad_inputs(), False ignores it
        check = getattr(model, "check_inputs", True)
        seeds.prepare_attn_words(cfg.num_hidden_layers, B, cfg.num_attention_heads, S, dev)
        pid = ops.position_ids(input_ids, cfg.pad_token_id)
        x0 = torch.empty((B * S, H), dtype=dt, device=dev)
        ops.embed_fwd(dt, x0, B, S, H, input_ids=input_ids, pos_ids=pid, word=word, type0=type_w[0], pos=pos_w)
        x, m1, r1 = ops.layernorm_fwd(x0, ln_g, ln_b, cfg.layer_norm_eps, drop_p=seeds.p_hidden, drop_seed=seeds.seed(1))
        # layout stream
        status = model.input_status(dev) if check else None
        sp = torch.empty((B * S, H), dtype=dt, device=dev)
        ops.embed_fwd(dt, sp, B, S, H, bbox=bbox, x=xw, y=yw, h=hw, w=ww, clip_hw=False, status=status)
        bp = torch.empty((B * S, Hl), dtype=dt, device=dev)
        zero_type = model.zeros("type", Hl, dev)
        zero_pos = model.zeros("pos", (1, Hl), dev)
        zpid = model.zeros_i32("zpid", (B, S), dev)
        ops.embed_fwd(dt, bp, B, S, Hl, input_ids=pid.to(torch.int64), pos_ids=zpid, word=boxpos, type0=zero_type, pos=zero_pos)
        Wl = wc.cast("boxlin", lin_w, dt)
        l0 = ops.gemm(sp, Wl, bias=lin_b, residual=bp)
        l, m2, r2 = ops.layernorm_fwd(l0, lln_g, lln_b, cfg.layer_norm_eps, drop_p=seeds.p_hidden, drop_seed=seeds.seed(2))
        if check is True:
            model.raise_on_bad_inputs()
        ctx.model, ctx.st = model, st
        ctx.saved = (pid, x0, m1, r1, sp, l0, m2, r2)
        ctx.inputs = (input_ids, bbox)
        ctx.params = params
        return x, l

    @staticmethod
    def backward(ctx, d_x, d_l):
        model, st = ctx.model, ctx.st
        (word, type_w, pos_w, ln_g, ln_b, xw, yw, hw, ww, boxpos, lin_w, lin_b, lln_g, lln_b) = ctx.params
        pid, x0, m1, r1, sp, l0, m2, r2 = ctx.saved
        input_ids, bbox = ctx.inputs
        cfg, wc, dt = model.config, model.weight_cache, st.dtype
        B, S = input_ids.shape
        H, Hl = cfg.hidden_size, cfg.hidden_size // cfg.channel_shrink_ratio
        seeds = st.seeds
        g = zeros_like_params(ctx.params)
        d_x0 = ops.layernorm_bwd(d_x.contiguous(), x0, ln_g, m1, r1, g[id(ln_g)], g[id(ln_b)], drop_p=seeds.p_hidden,
                                 drop_seed=seeds.seed(1))
        ops.embed_bwd(d_x0, B, S, H, input_ids=input_ids, pos_ids=pid, g_word=g[id(word)], g_pos=g[id(pos_w)],
                      pad_id=cfg.pad_token_id)
        ops.colsum(d_x0, out=g[id(type_w)][0], accumulate=True)
        d_l0 = ops.layernorm_bwd(d_l.contiguous(), l0, lln_g, m2, r2, g[id(lln_g)], g[id(lln_b)], drop_p=seeds.p_hidden,
                                 drop_seed=seeds.seed(2))
        # box position table: rows = position ids (pad rows skipped, like padding_idx)
        scratch_pos = torch.zeros((1, Hl), dtype=torch.float32, device=d_l0.device)
        ops.embed_bwd(d_l0, B, S, Hl, input_ids=pid.to(torch.int64), pos_ids=model.zeros_i32("zpid", (B, S), d_l0.device),
                      g_word=g[id(boxpos)], g_pos=scratch_pos, pad_id=cfg.pad_token_id)
        ops.colsum(d_l0, out=g[id(lin_b)], accumulate=True)
        ops.gemm(d_l0, sp, a_kmajor=False, b_kmajor=False, out=g[id(lin_w)], accumulate=True)
        d_sp = ops.gemm(d_l0, wc.cast("boxlin", lin_w, dt), b_kmajor=False)
        ops.embed_bwd(d_sp, B, S, H, bbox=bbox, g_x=g[id(xw)], g_y=g[id(yw)], g_h=g[id(hw)], g_w=g[id(ww)], clip_hw=False,
                      pad_id=cfg.pad_token_id)
        join_pending()     # weight-gradient work the layer stages left on the side stream
        grads = tuple(g[id(p)] if p.requires_grad else None for p in ctx.params)
        return (None, None, None, None) + grads


# ------------------------------------------------------------------------------------------------
# bf16: the two streams in lock step, every pair of same-role Linear layers as ONE grouped launch (ops.gemm_group with
# per-problem epilogues).  The layout stream's GEMMs ([4096, 192] x [192, 576] ...) are all fixed cost on their own (11-18 us
# for 1-2 us of MFMA work), and a LiLT step issued launch by launch is host-bound (18.5 ms of host against 20 ms of device time).
# ------------------------------------------------------------------------------------------------
def _use_groups(dt, *dims) -> bool:
    return dt == torch.bfloat16 and all(d % 8 == 0 for d in dims) and os.environ.get("PENEO_LILT_GROUPS", "1") != "0"


def _post_attn_fwd2(wc, idx, dt, eps, seeds, site_t, site_l, x, att, tp, l, latt, lp):
    (wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2), (lwo, lbo, lg1, lb1, lwi, lbi, lwo2, lbo2, lg2, lb2) = tp, lp
    kt, kl = f"L{idx}.t", f"L{idx}.l"
    Wo, Wi, Wo2 = wc.cast(kt + ".o", wo, dt), wc.cast(kt + ".i", wi, dt), wc.cast(kt + ".o2", wo2, dt)
    lWo, lWi, lWo2 = wc.cast(kl + ".o", lwo, dt), wc.cast(kl + ".i", lwi, dt), wc.cast(kl + ".o2", lwo2, dt)
    h1, lh1 = ops.gemm_group([(att, Wo, None, dict(bias=bo, residual=x, drop_p=seeds.p_hidden, drop_seed=seeds.seed(site_t))),
                              (latt, lWo, None, dict(bias=lbo, residual=l, drop_p=seeds.p_hidden, drop_seed=seeds.seed(site_l)))])
    a, m1, r1 = ops.layernorm_fwd(h1, g1, b1, eps)
    la, lm1, lr1 = ops.layernorm_fwd(lh1, lg1, lb1, eps)
    zi = torch.empty((x.shape[0], wi.shape[0]), dtype=dt, device=x.device)
    lzi = torch.empty((l.shape[0], lwi.shape[0]), dtype=dt, device=x.device)
    inter, linter = ops.gemm_group([(a, Wi, None, dict(bias=bi, act=ACT_GELU, preact=zi)),
                                    (la, lWi, None, dict(bias=lbi, act=ACT_GELU, preact=lzi))])
    h2, lh2 = ops.gemm_group([(inter, Wo2, None, dict(bias=bo2, residual=a, drop_p=seeds.p_hidden, drop_seed=seeds.seed(site_t + 1))),
                              (linter, lWo2, None, dict(bias=lbo2, residual=la, drop_p=seeds.p_hidden, drop_seed=seeds.seed(site_l + 1)))])
    out, m2, r2 = ops.layernorm_fwd(h2, g2, b2, eps)
    lout, lm2, lr2 = ops.layernorm_fwd(lh2, lg2, lb2, eps)
    return out, lout, (att, h1, m1, r1, a, zi, inter, h2, m2, r2), (latt, lh1, lm1, lr1, la, lzi, linter, lh2, lm2, lr2)


def _post_attn_bwd2(wc, idx, dt, seeds, site_t, site_l, d_out, d_lout, sv_t, sv_l, tp, lp, on_side):
    """Both streams' post-attention backward in lock step: the three dgrad GEMM pairs as grouped launches on the main stream, the
    text weight gradients one by one (split-k) and the FOUR layout weight gradients of this half as one grouped launch (each tile
    with its full K = tokens: 38 long workgroups instead of four split-k launches + four reductions) on the side stream."""
    (wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2), (lwo, lbo, lg1, lb1, lwi, lbi, lwo2, lbo2, lg2, lb2) = tp, lp
    att, h1, m1, r1, a, zi, inter, h2, m2, r2 = sv_t
    latt, lh1, lm1, lr1, la, lzi, linter, lh2, lm2, lr2 = sv_l
    kt, kl = f"L{idx}.t", f"L{idx}.l"
    Wo, Wi, Wo2 = wc.cast(kt + ".o", wo, dt), wc.cast(kt + ".i", wi, dt), wc.cast(kt + ".o2", wo2, dt)
    lWo, lWi, lWo2 = wc.cast(kl + ".o", lwo, dt), wc.cast(kl + ".i", lwi, dt), wc.cast(kl + ".o2", lwo2, dt)
    dev = d_out.device
    Hs, Is, Hl, Il = wo.shape[0], wi.shape[0], lwo.shape[0], lwi.shape[0]
    wgrad = lambda dy, xin: ops.gemm(dy, xin, a_kmajor=False, b_kmajor=False, out_dtype=torch.float32)
    pool = torch.zeros(4 * Hs + (Hs + Is + Hs) + 4 * Hl + (Hl + Il + Hl), dtype=torch.float32, device=dev)   # one fill for all small accumulators
    dg2, db2, dg1, db1 = pool[:Hs], pool[Hs:2 * Hs], pool[2 * Hs:3 * Hs], pool[3 * Hs:4 * Hs]
    o = 4 * Hs
    dbo2, dbi, dbo = pool[o:o + Hs], pool[o + Hs:o + Hs + Is], pool[o + Hs + Is:o + 2 * Hs + Is]
    o += 2 * Hs + Is
    ldg2, ldb2, ldg1, ldb1 = pool[o:o + Hl], pool[o + Hl:o + 2 * Hl], pool[o + 2 * Hl:o + 3 * Hl], pool[o + 3 * Hl:o + 4 * Hl]
    o += 4 * Hl
    ldbo2, ldbi, ldbo = pool[o:o + Hl], pool[o + Hl:o + Hl + Il], pool[o + Hl + Il:]
    drop = seeds.p_hidden > 0
    d_dense2 = torch.empty_like(h2) if drop else None
    d_ldense2 = torch.empty_like(lh2) if drop else None
    d_h2 = ops.layernorm_bwd(d_out.contiguous(), h2, g2, m2, r2, dg2, db2, dx_dropped=d_dense2, drop2_p=seeds.p_hidden,
                             drop2_seed=seeds.seed(site_t + 1))
    d_lh2 = ops.layernorm_bwd(d_lout.contiguous(), lh2, lg2, lm2, lr2, ldg2, ldb2, dx_dropped=d_ldense2, drop2_p=seeds.p_hidden,
                              drop2_seed=seeds.seed(site_l + 1))
    if not drop:
        d_dense2, d_ldense2 = d_h2, d_lh2
    d_zi, d_lzi = ops.gemm_group([(d_dense2, Wo2, None, dict(grad_src=zi, grad_act=ACT_GELU)),
                                  (d_ldense2, lWo2, None, dict(grad_src=lzi, grad_act=ACT_GELU))], b_kmajor=False)
    d_a, d_la = ops.gemm_group([(d_zi, Wi, None, dict(residual=d_h2)), (d_lzi, lWi, None, dict(residual=d_lh2))], b_kmajor=False)
    d_dense1 = torch.empty_like(h1) if drop else None
    d_ldense1 = torch.empty_like(lh1) if drop else None
    d_h1 = ops.layernorm_bwd(d_a, h1, g1, m1, r1, dg1, db1, dx_dropped=d_dense1, drop2_p=seeds.p_hidden, drop2_seed=seeds.seed(site_t))
    d_lh1 = ops.layernorm_bwd(d_la, lh1, lg1, lm1, lr1, ldg1, ldb1, dx_dropped=d_ldense1, drop2_p=seeds.p_hidden,
                              drop2_seed=seeds.seed(site_l))
    if not drop:
        d_dense1, d_ldense1 = d_h1, d_lh1
    d_att, d_latt = ops.gemm_group([(d_dense1, Wo, None), (d_ldense1, lWo, None)], b_kmajor=False)

    # Round 5: only the bias column sums start here.  The six weight gradients of this half leave as operand pairs: the caller
    # runs them behind the attention backward, together with the two QKV weight gradients, as TWO grouped launches (four text,
    # four layout; each tile with its full K) instead of four split-k launches + reductions and two groups - the schedule that
    # measured best for the LayoutLMv3 layer (csrc/stages.hip), and 7 launches per layer less for a step that is close to host-bound
    def side_work():
        ops.colsum(d_dense2, out=dbo2, accumulate=True)
        ops.colsum(d_zi, out=dbi, accumulate=True)
        ops.colsum(d_dense1, out=dbo, accumulate=True)
        ops.colsum(d_ldense2, out=ldbo2, accumulate=True)
        ops.colsum(d_lzi, out=ldbi, accumulate=True)
        ops.colsum(d_ldense1, out=ldbo, accumulate=True)
    on_side(side_work, (d_dense2, d_zi, d_dense1, d_ldense2, d_lzi, d_ldense1))
    wg_t = [(d_zi, a), (d_dense2, inter), (d_dense1, att)]                 # -> dwi, dwo2, dwo
    wg_l = [(d_lzi, la), (d_ldense2, linter), (d_ldense1, latt)]          # -> ldwi, ldwo2, ldwo
    gt = (None, dbo, dg1, db1, None, dbi, None, dbo2, dg2, db2)
    gl = (None, ldbo, ldg1, ldb1, None, ldbi, None, ldbo2, ldg2, ldb2)
    return d_att, d_h1, gt, d_latt, d_lh1, gl, wg_t, wg_l


class _LiltLayerStage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, st, idx, x, l, *params):
        (wq, bq, wk, bk, wv, bv, lwq, lbq, lwk, lbk, lwv, lbv, *rest) = params
        tp, lp = rest[:10], rest[10:]
        cfg, wc, dt = model.config, model.weight_cache, st.dtype
        B, S = st.dims
        H, nh = cfg.hidden_size, cfg.num_attention_heads
        Hl = H // cfg.channel_shrink_ratio
        d, dl = H // nh, Hl // nh
        dc = d + dl
        seeds = st.seeds
        site = 32 * (idx + 1)
        dev = x.device
        Wqkv = wc.cat_rows(f"L{idx}.qkv", [wq, wk, wv], dt)
        bqkv = wc.get((f"L{idx}.bqkv",), [bq, bk, bv], lambda: torch.cat([bq.detach(), bk.detach(), bv.detach()]))
        Wlqkv = wc.cat_rows(f"L{idx}.lqkv", [lwq, lwk, lwv], dt)
        blqkv = wc.get((f"L{idx}.blqkv",), [lbq, lbk, lbv], lambda: torch.cat([lbq.detach(), lbk.detach(), lbv.detach()]))
        grouped = _use_groups(dt, H, Hl, tp[4].shape[0], lp[4].shape[0])
        if grouped:
            qkv, lqkv = ops.gemm_group([(x, Wqkv, None, dict(bias=bqkv)), (l, Wlqkv, None, dict(bias=blqkv))])
        else:
            qkv = ops.gemm(x, Wqkv, bias=bqkv)          # [R, 3H]
            lqkv = ops.gemm(l, Wlqkv, bias=blqkv)       # [R, 3Hl]
        R = x.shape[0]
        cat = torch.empty((R, 3 * nh * dc), dtype=dt, device=dev)
        ops.head_concat(qkv[:, :H], lqkv[:, :Hl], nh, cat[:, :nh * dc], 1.0 / math.sqrt(d), 1.0 / math.sqrt(dl))
        ops.head_concat(qkv[:, H:], lqkv[:, Hl:], 2 * nh, cat[:, nh * dc:])      # k and v in one launch: 2 nh "heads"
        qc, kc, vc = cat[:, :nh * dc], cat[:, nh * dc:2 * nh * dc], cat[:, 2 * nh * dc:]
        attc, lse = ops.attn_fwd(qc, kc, vc, B, nh, S, dc, 1.0, None, st.key_bias, drop_p=seeds.p_attn,
                                 drop_words=seeds.attn_words(idx, cfg.num_hidden_layers, B, nh, S, dev))
        att = torch.empty((R, H), dtype=dt, device=dev)
        latt = torch.empty((R, Hl), dtype=dt, device=dev)
        ops.head_split(attc, nh, att, latt)
        if grouped:
            xo, lo, sv_t, sv_l = _post_attn_fwd2(wc, idx, dt, cfg.layer_norm_eps, seeds, site + 2, site + 6, x, att, tp, l, latt, lp)
        else:
            xo, sv_t = _post_attn_fwd(wc, f"L{idx}.t", dt, cfg.layer_norm_eps, seeds, site + 2, x, att, tp)
            lo, sv_l = _post_attn_fwd(wc, f"L{idx}.l", dt, cfg.layer_norm_eps, seeds, site + 6, l, latt, lp)
        ctx.grouped = grouped
        ctx.model, ctx.st, ctx.idx = model, st, idx
        ctx.saved = (x, l, cat, attc, lse, sv_t, sv_l)
        ctx.params = params
        return xo, lo

    @staticmethod
    def backward(ctx, d_xo, d_lo):
        model, st, idx = ctx.model, ctx.st, ctx.idx
        (wq, bq, wk, bk, wv, bv, lwq, lbq, lwk, lbk, lwv, lbv, *rest) = ctx.params
        tp, lp = rest[:10], rest[10:]
        x, l, cat, attc, lse, sv_t, sv_l = ctx.saved
        cfg, wc, dt = model.config, model.weight_cache, st.dtype
        B, S = st.dims
        H, nh = cfg.hidden_size, cfg.num_attention_heads
        Hl = H // cfg.channel_shrink_ratio
        d, dl = H // nh, Hl // nh
        dc = d + dl
        seeds = st.seeds
        site = 32 * (idx + 1)
        dev = x.device
        R = x.shape[0]
        main = torch.cuda.current_stream()
        side = model.side_stream(dev)

        # operand tensors of the side-stream work, referenced until the (deferred) join.  Only what the side stream READS:
        # a reference to one of its OUTPUTS (the bias-gradient slices are returned as gradients) would push that tensor's
        # use count to 2, AccumulateGrad would clone it on the main stream instead of storing it, and the clone would race
        # the side stream's column sums
        kept = []

        def on_side(fn, operands=()):
            kept.extend(operands)
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                return fn()

        wg_t = wg_l = None
        if ctx.grouped:
            d_att, d_x_res, gt, d_latt, d_l_res, gl, wg_t, wg_l = _post_attn_bwd2(wc, idx, dt, seeds, site + 2, site + 6, d_xo, d_lo, sv_t, sv_l, tp, lp, on_side)
        else:
            d_att, d_x_res, gt = _post_attn_bwd(wc, f"L{idx}.t", dt, seeds, site + 2, d_xo, sv_t, tp, on_side)
            d_latt, d_l_res, gl = _post_attn_bwd(wc, f"L{idx}.l", dt, seeds, site + 6, d_lo, sv_l, lp, on_side)
        d_attc = torch.empty((R, nh * dc), dtype=dt, device=dev)
        ops.head_concat(d_att, d_latt, nh, d_attc)
        qc, kc, vc = cat[:, :nh * dc], cat[:, nh * dc:2 * nh * dc], cat[:, 2 * nh * dc:]
        dcat = torch.empty_like(cat)
        ops.attn_bwd(qc, kc, vc, attc, d_attc, lse, B, nh, S, dc, 1.0, None, st.key_bias, dcat, None, drop_p=seeds.p_attn,
                     drop_words=seeds.attn_words(idx, cfg.num_hidden_layers, B, nh, S, dcat.device))
        dqkv = torch.empty((R, 3 * H), dtype=dt, device=dev)
        dlqkv = torch.empty((R, 3 * Hl), dtype=dt, device=dev)
        ops.head_split(dcat[:, :nh * dc], nh, dqkv[:, :H], dlqkv[:, :Hl], 1.0 / math.sqrt(d), 1.0 / math.sqrt(dl))
        ops.head_split(dcat[:, nh * dc:], 2 * nh, dqkv[:, H:], dlqkv[:, Hl:])
        Wqkv = wc.cat_rows(f"L{idx}.qkv", [wq, wk, wv], dt)
        Wlqkv = wc.cat_rows(f"L{idx}.lqkv", [lwq, lwk, lwv], dt)
        wg = lambda dy, xin: ops.gemm(dy, xin, a_kmajor=False, b_kmajor=False, out_dtype=torch.float32)
        if ctx.grouped:
            def all_wgrads():
                gt_ = ops.gemm_group([(dqkv, x, None)] + [(dy, xin, None) for dy, xin in wg_t], a_kmajor=False, b_kmajor=False,
                                     out_dtype=torch.float32)
                gl_ = ops.gemm_group([(dlqkv, l, None)] + [(dy, xin, None) for dy, xin in wg_l], a_kmajor=False, b_kmajor=False,
                                     out_dtype=torch.float32)
                return ops.colsum(dqkv), ops.colsum(dlqkv), gt_, gl_
            dbqkv, dblqkv, gt_, gl_ = on_side(all_wgrads, (dqkv, dlqkv, x, l) + tuple(t for pr in wg_t + wg_l for t in pr))
            dwqkv, dwi_, dwo2_, dwo_ = gt_
            dwlqkv, ldwi_, ldwo2_, ldwo_ = gl_
            gt = (dwo_,) + gt[1:4] + (dwi_,) + gt[5:6] + (dwo2_,) + gt[7:]
            gl = (ldwo_,) + gl[1:4] + (ldwi_,) + gl[5:6] + (ldwo2_,) + gl[7:]
        else:
            dbqkv, dblqkv, dwqkv, dwlqkv = on_side(lambda: (ops.colsum(dqkv), ops.colsum(dlqkv), wg(dqkv, x), wg(dlqkv, l)),
                                                   (dqkv, dlqkv, x, l))
        if ctx.grouped:
            d_x, d_l = ops.gemm_group([(dqkv, Wqkv, None, dict(residual=d_x_res)), (dlqkv, Wlqkv, None, dict(residual=d_l_res))], b_kmajor=False)
        else:
            d_x = ops.gemm(dqkv, Wqkv, b_kmajor=False, residual=d_x_res)
            d_l = ops.gemm(dlqkv, Wlqkv, b_kmajor=False, residual=d_l_res)
        if os.environ.get("PENEO_DEFER_JOIN", "1") != "0" and can_defer(ctx.params):
            defer_join(side, keep=kept)   # joined one stage later: the critical path does not wait for the QKV wgrads (engine.py)
        else:
            main.wait_stream(side)
        grads = (dwqkv[:H], dbqkv[:H], dwqkv[H:2 * H], dbqkv[H:2 * H], dwqkv[2 * H:], dbqkv[2 * H:],
                 dwlqkv[:Hl], dblqkv[:Hl], dwlqkv[Hl:2 * Hl], dblqkv[Hl:2 * Hl], dwlqkv[2 * Hl:], dblqkv[2 * Hl:]) + gt + gl
        grads = tuple(gr if p.requires_grad else None for gr, p in zip(grads, ctx.params))
        return (None, None, None, d_x, d_l) + grads


class _CatStage(torch.autograd.Function):
    """last_hidden_state = cat(text, layout) along the feature dim (:987)."""

    @staticmethod
    def forward(ctx, x, l):
        out = torch.empty((x.shape[0], x.shape[1] + l.shape[1]), dtype=x.dtype, device=x.device)
        ops.copy2d(x, out[:, :x.shape[1]])
        ops.copy2d(l, out[:, x.shape[1]:])
        ctx.hx = x.shape[1]
        return out

    @staticmethod
    def backward(ctx, d):
        return ops.copy2d(d[:, :ctx.hx]), ops.copy2d(d[:, ctx.hx:])


def _post_params(dln: _DenseLN, inter: _Dense, out: _DenseLN) -> List[torch.Tensor]:
    return [dln.dense.weight, dln.dense.bias, dln.LayerNorm.weight, dln.LayerNorm.bias, inter.dense.weight, inter.dense.bias,
            out.dense.weight, out.dense.bias, out.LayerNorm.weight, out.LayerNorm.bias]


class LiltModel(nn.Module):
    """Counterpart of the reference ``LiltModel`` for the PEneo call pattern:
    ``forward(input_ids, bbox, attention_mask)`` -> ``(cat(text, layout) [B, S, H + H/r],)``."""

    config_class = LiltConfig

    def __init__(self, config: LiltConfig):
        super().__init__()
        self.config = config
        H, nh, r = config.hidden_size, config.num_attention_heads, config.channel_shrink_ratio
        if H % 6 or H % nh or (H // r) % nh:
            raise ValueError("LiLT needs hidden_size divisible by 6, by num_attention_heads and (H / r) by the heads")
        self.embeddings = LiltTextEmbeddings(config)
        self.layout_embeddings = LiltLayoutEmbeddings(config)
        self.encoder = LiltEncoder(config)
        self.weight_cache = WeightCache()
        self.compute_dtype = torch.float32
        self._consts = {}

    def side_stream(self, device) -> "torch.cuda.Stream":
        return engine_side_stream(device, "wgrad")


    def input_status(self, dev) -> torch.Tensor:
        """int32 [1] on `dev`, sticky: set to 1 by the layout embedding kernel when a box coordinate is out of range."""
        return self.zeros_i32("input_status", 1, dev)

    def raise_on_bad_inputs(self) -> None:
        """Read (one host sync per flag) and clear the input flags; raises the reference's IndexError if any forward since the last
        call saw a bbox coordinate outside its table (reference modeling_lilt.py)."""
        for key, st in self._consts.items():
            if key[0] == "input_status" and int(st) != 0:
                st.zero_()
                raise IndexError("The :obj:`bbox`coordinate values should be within 0-1000 range.")

    def zeros(self, key, shape, dev):
        k = (key, str(shape), str(dev))
        if k not in self._consts:
            self._consts[k] = torch.zeros(shape, dtype=torch.float32, device=dev)
        return self._consts[k]

    def zeros_i32(self, key, shape, dev):
        k = (key, str(shape), str(dev))
        if k not in self._consts:
            self._consts[k] = torch.zeros(shape, dtype=torch.int32, device=dev)
        return self._consts[k]

    def _init_weights(self, module) -> None:
        std = self.config.initializer_range
        if isinstance(module, nn.Linear):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.bias is not None:
                module.bias.data.zero_()
        elif isinstance(module, nn.Embedding):
            module.weight.data.normal_(mean=0.0, std=std)
            if module.padding_idx is not None:
                module.weight.data[module.padding_idx].zero_()
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)

    def forward(self, input_ids=None, bbox=None, attention_mask=None, **unused):
        if input_ids is None:
            raise ValueError("You have to specify either input_ids or inputs_embeds")
        if not input_ids.is_cuda:
            raise PeneoHipError("peneo_amd runs on the GPU only: move the model and the batch to 'cuda' "
                                "(there is no CPU fallback; the CPU oracle lives in oracle/ for tests)")
        cfg = self.config
        B, S = input_ids.shape
        dev = input_ids.device
        if bbox is None:
            bbox = torch.zeros((B, S, 4), dtype=torch.long, device=dev)
        if attention_mask is None:
            attention_mask = torch.ones((B, S), dtype=torch.long, device=dev)
        st = _State()
        st.dtype = self.compute_dtype
        st.dims = (B, S)
        st.seeds = DropoutSeeds(self.training, cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob)
        kb = torch.zeros((B, ops.attn_padded_len(S)), dtype=torch.float32, device=dev)
        kb[:, :S].masked_fill_(attention_mask == 0, -1.0e30)
        st.key_bias = kb
        e, le = self.embeddings, self.layout_embeddings
        eparams = [e.word_embeddings.weight, e.token_type_embeddings.weight, e.position_embeddings.weight,
                   e.LayerNorm.weight, e.LayerNorm.bias, le.x_position_embeddings.weight, le.y_position_embeddings.weight,
                   le.h_position_embeddings.weight, le.w_position_embeddings.weight, le.box_position_embeddings.weight,
                   le.box_linear_embeddings.weight, le.box_linear_embeddings.bias, le.LayerNorm.weight, le.LayerNorm.bias]
        x, l = _LiltEmbedStage.apply(self, st, input_ids.contiguous(), bbox.contiguous(), *eparams)
        for i, layer in enumerate(self.encoder.layer):
            s = layer.attention.self
            params = [s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias,
                      s.layout_query.weight, s.layout_query.bias, s.layout_key.weight, s.layout_key.bias,
                      s.layout_value.weight, s.layout_value.bias]
            params += _post_params(layer.attention.output, layer.intermediate, layer.output)
            params += _post_params(layer.attention.layout_output, layer.layout_intermediate, layer.layout_output)
            x, l = _LiltLayerStage.apply(self, st, i, x, l, *params)
        out = _CatStage.apply(x, l)
        return (out.view(B, S, out.shape[1]),)
