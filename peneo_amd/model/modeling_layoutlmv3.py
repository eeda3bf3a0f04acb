"""LayoutLMv3 backbone on libpeneo_hip kernels.

Module / parameter names reproduce the reference (model/backbone/layoutlmv3/modeling_layoutlmv3.py)
so state dicts interchange (``backbone.embeddings.word_embeddings.weight``,
``backbone.encoder.layer.{i}.attention.self.query.weight`` ...).  The ``nn.Module``s below only
*hold* fp32 master parameters; the math runs in three kinds of autograd stages whose forward and
backward are sequences of HIP kernel launches on the current stream:

  _EmbedStage   K1 text gather+LN, K2 patch embed, K3 concat LN, K4 rel-pos bias (once per forward)
  _LayerStage   K5 fused QKV GEMM, K6 flash attention with bias, K7/K8 GEMMs with fused
                bias/GELU/residual(/dropout) epilogues + LayerNorm
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
import torch.nn as nn

import ctypes as C

from .. import ops
from ..hip import ACT_GELU, ACT_NONE, EncoderLayer, EncoderLayerGrads, PeneoHipError, check, ptr
from ..hip import lib as hip_lib
from ..hip import stream as hip_stream
from .configuration_peneo import LayoutLMv3Config
from .engine import DropoutSeeds, can_defer, WeightCache, defer_join, join_pending, zeros_like_param, zeros_like_params
from .engine import side_stream as engine_side_stream
from .relpos import bucket_lut, visual_xy


# ------------------------------------------------------------------------------------------------
# parameter containers (names == reference)
# ------------------------------------------------------------------------------------------------
class _SelfAttentionParams(nn.Module):
    def __init__(self, hidden: int):
        super().__init__()
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(hidden, hidden)
        self.value = nn.Linear(hidden, hidden)


class _DenseLN(nn.Module):
    """RobertaSelfOutput / RobertaOutput: dense + LayerNorm (dropout has no parameters)."""

    def __init__(self, fan_in: int, fan_out: int, eps: float):
        super().__init__()
        self.dense = nn.Linear(fan_in, fan_out)
        self.LayerNorm = nn.LayerNorm(fan_out, eps=eps)


class _Dense(nn.Module):
    def __init__(self, fan_in: int, fan_out: int):
        super().__init__()
        self.dense = nn.Linear(fan_in, fan_out)


class _AttentionParams(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.self = _SelfAttentionParams(cfg.hidden_size)
        self.output = _DenseLN(cfg.hidden_size, cfg.hidden_size, cfg.layer_norm_eps)


class LayoutLMv3Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.attention = _AttentionParams(cfg)
        self.intermediate = _Dense(cfg.hidden_size, cfg.intermediate_size)
        self.output = _DenseLN(cfg.intermediate_size, cfg.hidden_size, cfg.layer_norm_eps)


class LayoutLMv3Encoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([LayoutLMv3Layer(cfg) for _ in range(cfg.num_hidden_layers)])
        if cfg.has_relative_attention_bias:
            self.rel_pos_bias = nn.Linear(cfg.rel_pos_bins, cfg.num_attention_heads, bias=False)
        if cfg.has_spatial_attention_bias:
            self.rel_pos_x_bias = nn.Linear(cfg.rel_2d_pos_bins, cfg.num_attention_heads, bias=False)
            self.rel_pos_y_bias = nn.Linear(cfg.rel_2d_pos_bins, cfg.num_attention_heads, bias=False)


class LayoutLMv3Embeddings(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        H = cfg.hidden_size
        self.word_embeddings = nn.Embedding(cfg.vocab_size, H, padding_idx=cfg.pad_token_id)
        self.token_type_embeddings = nn.Embedding(cfg.type_vocab_size, H)
        self.LayerNorm = nn.LayerNorm(H, eps=cfg.layer_norm_eps)
        self.register_buffer("position_ids", torch.arange(cfg.max_position_embeddings).expand((1, -1)))
        self.position_embeddings = nn.Embedding(cfg.max_position_embeddings, H, padding_idx=cfg.pad_token_id)
        self.x_position_embeddings = nn.Embedding(cfg.max_2d_position_embeddings, cfg.coordinate_size)
        self.y_position_embeddings = nn.Embedding(cfg.max_2d_position_embeddings, cfg.coordinate_size)
        self.h_position_embeddings = nn.Embedding(cfg.max_2d_position_embeddings, cfg.shape_size)
        self.w_position_embeddings = nn.Embedding(cfg.max_2d_position_embeddings, cfg.shape_size)


class PatchEmbed(nn.Module):
    def __init__(self, embed_dim: int):
        super().__init__()
        self.proj = nn.Conv2d(3, embed_dim, kernel_size=16, stride=16)


# Bias gradients are column sums on the weight-gradient stream.  Two fusions were built, measured slower and removed from the
# model: out of the LayerNorm backward (peneo_layernorm_bwd's dx_colsum; round 3: 24 launches fewer, step +0.1 ms: the extra
# reduction sits on the main stream) and out of the weight-gradient GEMM (peneo_gemm's a_colsum; round 4: 53 launches fewer,
# step +0.1 .. +0.3 ms: HBM-bound side work became MFMA work beside the main stream -- profiles/r04_bias_gradient_fusion_ab.txt).


class _FwdState:
    """Per-forward scratch shared by the stages (rel-pos bias, its gradient accumulator, seeds)."""

    def __init__(self) -> None:
        self.bias = None       # [B, nh, T, T] working dtype, already divided by sqrt(d)
        self.g_bias = None     # fp32 accumulator of dS over the layers (training only)
        self.buckets = (None, None, None)
        self.bucket_inputs = None   # (pos_t, xs, ys, lut1, lut2): what the transposed maps are rebuilt from in backward
        self.ds_layers = None       # bf16 [L, B, nh, T, Tp]: per-layer dS^T of the single-pass attention backward
        self.rel_grads = None       # fp32 (dw1, dwx, dwy) filled group by group from ds_layers on the side stream
        self.bucket_t = None        # transposed bucket maps (built when the first group is reduced)
        self.key_mask = None   # int32 [B, T]
        self.key_bias = None   # fp32 [B, Tp] additive mask, only when there is no bias tensor
        self.seeds: Optional[DropoutSeeds] = None
        self.dtype = torch.float32
        self.dims = None       # (B, S, T)
        self.grad_pools = None  # [layers, pool] fp32 accumulators of the layer stages' backward (one zero fill per step)
        self.grad_mode = False  # torch.is_grad_enabled() of the model call (inside an autograd.Function it is always off)
        self.fill_event = None  # the side-stream zero fill of the embedding stage's gradient buffer (main-pool memory)


# ------------------------------------------------------------------------------------------------
# stage 1: embeddings + relative-position bias
# ------------------------------------------------------------------------------------------------
class _EmbedStage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, st, input_ids, bbox, attention_mask, image, *params):
        cfg, wc, dt = model.config, model.weight_cache, st.dtype
        (word, type_w, pos_w, xw, yw, hw, ww, ln_g, ln_b, proj_w, proj_b, cls, pos_embed, norm_g, norm_b, LN_g, LN_b,
         *rel) = params
        B, S = input_ids.shape
        H = cfg.hidden_size
        dev = input_ids.device
        seeds = st.seeds
        has_img = image is not None
        nv = 0
        if has_img:
            grid = image.shape[2] // 16
            nv = grid * (image.shape[3] // 16) + 1
            if nv != pos_embed.shape[1]:
                raise ValueError(f"image gives {nv} visual tokens but pos_embed has {pos_embed.shape[1]}")
        T = S + nv
        st.dims = (B, S, T)
        seeds.prepare_attn_words(cfg.num_hidden_layers, B, cfg.num_attention_heads, T, dev)
        # The bbox range check of the reference (modeling_layoutlmv3.py:133) is made by the embedding kernel itself (a token whose
        # id / position / box coordinate falls outside its table sets a sticky device flag and gets a zero row, never an
        # out-of-bounds read).  check_inputs = True (default) reads the flag at once and raises like the reference (one host sync);
        # "deferred" leaves it on the device for raise_on_bad_inputs() -- a training loop calls that where it synchronises anyway
        # (bench.py: after the timed steps); False ignores it.
        check = getattr(model, "check_inputs", True)
        status = model.input_status(dev) if check else None
        pid = ops.position_ids(input_ids, cfg.pad_token_id)
        x0 = torch.empty((B, S, H), dtype=dt, device=dev)
        ops.embed_fwd(dt, x0, B, S, H, input_ids=input_ids, pos_ids=pid, bbox=bbox, word=word, type0=type_w[0],
                      pos=pos_w, x=xw, y=yw, h=hw, w=ww, clip_hw=True, status=status)
        if check is True:
            model.raise_on_bad_inputs()
        cat = torch.empty((B, T, H), dtype=dt, device=dev)
        # text rows: LayerNorm (+dropout) of the summed embeddings, written in place of the concat
        _, m1, r1 = ops.layernorm_fwd(x0, ln_g, ln_b, cfg.layer_norm_eps, out=cat[:, :S],
                                      drop_p=seeds.p_hidden, drop_seed=seeds.seed(1))
        saved = dict(pid=pid, x0=x0, m1=m1, r1=r1, has_img=has_img)
        if has_img:
            patches = ops.im2col_patch16(image.contiguous(), dt)
            wproj = wc.cast("patch_proj", proj_w.view(H, -1), dt)
            proj = ops.gemm(patches, wproj, bias=proj_b)
            vis0 = ops.visual_assemble_fwd(proj, cls.view(-1), pos_embed.view(-1, H), B)
            _, mv, rv = ops.layernorm_fwd(vis0, norm_g, norm_b, 1e-6, out=cat[:, S:])
            emb, m2, r2 = ops.layernorm_fwd(cat, LN_g, LN_b, cfg.layer_norm_eps,
                                            drop_p=seeds.p_hidden, drop_seed=seeds.seed(2))
            saved.update(patches=patches, vis0=vis0, mv=mv, rv=rv, cat=cat, m2=m2, r2=r2)
        else:
            emb = cat

        # key mask over text + visual tokens (visual tokens are always attended: :1075-1080) and the per-token inputs of the
        # bucket maps, one launch
        nh = cfg.num_attention_heads
        d = H // nh
        use1, use2 = cfg.has_relative_attention_bias, cfg.has_spatial_attention_bias
        vx, vy = model.visual_xy(dev, nv) if (use2 and nv > 0) else (None, None)
        km, pos_t, xs, ys = ops.relpos_inputs(attention_mask, bbox, vx, vy, B, S, nv, use1, use2)
        st.key_mask = km

        # K4: bucket maps + summed bias, shared by every layer
        if use1 or use2:
            lut1 = model.lut("1d", cfg.rel_pos_bins, cfg.max_rel_pos, dev) if use1 else None
            lut2 = model.lut("2d", cfg.rel_2d_pos_bins, cfg.max_rel_2d_pos, dev) if use2 else None
            st.buckets = ops.relpos_buckets(pos_t, xs, ys, B, T, lut1, cfg.rel_pos_bins // 2, lut2, cfg.rel_2d_pos_bins // 2)
            st.bucket_inputs = (pos_t, xs, ys, lut1, lut2)
            ri = iter(rel)
            w1 = next(ri) if use1 else None
            wx = next(ri) if use2 else None
            wy = next(ri) if use2 else None
            st.bias = ops.relpos_bias_fwd(dt, st.buckets[0], st.buckets[1], st.buckets[2], w1, wx, wy,
                                          1.0 / math.sqrt(d), B, nh, T, key_mask=km)   # padding mask folded in
        else:
            # no bias tensor: the padding mask travels as an additive per-key row
            kb = torch.zeros((B, ops.attn_padded_len(T)), dtype=torch.float32, device=dev)
            kb[:, :T].masked_fill_(km == 0, -1.0e30)
            st.key_bias = kb
        ctx.model, ctx.st, ctx.saved = model, st, saved
        ctx.inputs = (input_ids, bbox)
        ctx.params = params
        # the gradient buffers of this stage (the 154 MB word table among them) are zero-filled NOW, on a side stream beside the
        # forward: filled in the backward they sat on the critical path at the very end of the step (128 us)
        ctx.g_pre = None
        if st.grad_mode and any(ctx.needs_input_grad[6:]):   # (needs_input_grad is set under no_grad too)
            ctx.g_pre = zeros_like_params(params, fill_stream=model.side_stream(dev, "rel"))
            st.fill_event = ctx.g_pre[1]
        return emb.view(B * T, H)

    @staticmethod
    def backward(ctx, d_emb):
        model, st, sv = ctx.model, ctx.st, ctx.saved
        cfg, dt = model.config, st.dtype
        (word, type_w, pos_w, xw, yw, hw, ww, ln_g, ln_b, proj_w, proj_b, cls, pos_embed, norm_g, norm_b, LN_g, LN_b,
         *rel) = ctx.params
        input_ids, bbox = ctx.inputs
        B, S, T = st.dims
        H = cfg.hidden_size
        seeds = st.seeds
        d_emb = d_emb.contiguous().view(B, T, H)
        if ctx.g_pre is not None:
            g, ready = ctx.g_pre
            ctx.g_pre = None
            torch.cuda.current_stream(d_emb.device).wait_event(ready)
        else:
            g = zeros_like_params(ctx.params)
        if sv["has_img"]:
            d_cat = ops.layernorm_bwd(d_emb, sv["cat"], LN_g, sv["m2"], sv["r2"], g[id(LN_g)], g[id(LN_b)],
                                      drop_p=seeds.p_hidden, drop_seed=seeds.seed(2))
            d_vis0 = torch.empty_like(sv["vis0"])
            ops.layernorm_bwd(d_cat[:, S:], sv["vis0"], norm_g, sv["mv"], sv["rv"], g[id(norm_g)], g[id(norm_b)], dx=d_vis0)
            d_proj = ops.visual_assemble_bwd(d_vis0, g[id(cls)].view(-1), g[id(pos_embed)].view(-1, H))
            ops.colsum(d_proj, out=g[id(proj_b)], accumulate=True)
            ops.gemm(d_proj, sv["patches"], a_kmajor=False, b_kmajor=False, out=g[id(proj_w)].view(H, -1), accumulate=True)
        else:
            d_cat = d_emb
        d_x0 = torch.empty_like(sv["x0"])
        ops.layernorm_bwd(d_cat[:, :S], sv["x0"], ln_g, sv["m1"], sv["r1"], g[id(ln_g)], g[id(ln_b)], dx=d_x0,
                          drop_p=seeds.p_hidden, drop_seed=seeds.seed(1))
        ops.embed_bwd(d_x0, B, S, H, input_ids=input_ids, pos_ids=sv["pid"], bbox=bbox, g_word=g[id(word)],
                      g_pos=g[id(pos_w)], g_x=g[id(xw)], g_y=g[id(yw)], g_h=g[id(hw)], g_w=g[id(ww)], clip_hw=True,
                      pad_id=cfg.pad_token_id)
        ops.colsum(d_x0.view(B * S, H), out=g[id(type_w)][0], accumulate=True)
        # rel-pos tables: every layer has accumulated its dS into st.g_bias by now
        if (st.g_bias is not None or st.ds_layers is not None) and rel:
            use1, use2 = cfg.has_relative_attention_bias, cfg.has_spatial_attention_bias
            ri = iter(rel)
            w1 = next(ri) if use1 else None
            wx = next(ri) if use2 else None
            wy = next(ri) if use2 else None
            d = H // cfg.num_attention_heads
            gw = (g[id(w1)] if w1 is not None else None, g[id(wx)] if wx is not None else None,
                  g[id(wy)] if wy is not None else None)
            if st.ds_layers is not None:
                # bf16 path: the layer stages already reduced their dS^T slabs group by group (side stream, joined by every
                # layer stage before it returned) into st.rel_grads
                torch.cuda.current_stream().wait_stream(model.side_stream(d_emb.device, "rel"))
                for dst, src in zip(gw, st.rel_grads):
                    if dst is not None and src is not None:
                        dst.add_(src)
                st.ds_layers, st.rel_grads, st.bucket_t = None, None, None
            else:
                ops.relpos_bias_bwd(st.g_bias, st.buckets[0], st.buckets[1], st.buckets[2], gw[0], gw[1], gw[2],
                                    1.0 / math.sqrt(d))
                st.g_bias = None
        join_pending()     # weight-gradient work the layer stages left on the side stream
        grads = tuple(g[id(p)] if p.requires_grad else None for p in ctx.params)
        return (None, None, None, None, None, None) + grads



# ------------------------------------------------------------------------------------------------
# composite layer calls (csrc/stages.hip): ONE C call per layer forward / backward instead of 7 / ~25 (bf16, default
# schedule).  PENEO_STAGE_CALLS=0: the per-kernel sequence below (also what fp32 parity mode and the optional schedules run).
# ------------------------------------------------------------------------------------------------
STAGE_CALLS = os.environ.get("PENEO_STAGE_CALLS", "1") != "0"
_WS_BYTES: dict = {}


def _layer_ws_bytes(rows: int, H: int, I: int, which: int) -> int:
    key = (rows, H, I, which)
    if key not in _WS_BYTES:
        _WS_BYTES[key] = int(hip_lib().peneo_encoder_layer_workspace_bytes(rows, H, I, which))
    return _WS_BYTES[key]


def _use_stage_calls(model, st, H: int, I: int) -> bool:
    return STAGE_CALLS and st.dtype == torch.bfloat16 and H % 8 == 0 and I % 8 == 0


class _LayerBuffers:
    """Activations of one layer forward carved from ONE bf16 and ONE fp32 allocation (the C call takes raw pointers)."""
    __slots__ = ("flat", "stats", "R", "H", "I", "has_zi", "o")

    def __init__(self, R: int, H: int, I: int, nlse: int, has_zi: bool, dev) -> None:
        self.R, self.H, self.I, self.has_zi = R, H, I, has_zi
        n = R * (3 * H + 4 * H + I + (I if has_zi else 0))     # qkv | att | h1 | a | h2 | inter | zi
        self.flat = torch.empty(n, dtype=torch.bfloat16, device=dev)
        self.stats = torch.empty(nlse + 4 * R, dtype=torch.float32, device=dev)   # lse | m1 | r1 | m2 | r2
        self.o = nlse

    def fill(self, L) -> None:
        R, H, I = self.R, self.H, self.I
        b, e = self.flat.data_ptr(), 2
        L.qkv = b
        L.att = b + e * R * 3 * H
        L.h1 = L.att + e * R * H
        L.a = L.h1 + e * R * H
        L.h2 = L.a + e * R * H
        L.inter = L.h2 + e * R * H
        L.zi = (L.inter + e * R * I) if self.has_zi else None
        s = self.stats.data_ptr()
        L.lse = s
        L.m1 = s + 4 * self.o
        L.r1 = L.m1 + 4 * R
        L.m2 = L.r1 + 4 * R
        L.r2 = L.m2 + 4 * R


def _describe_layer(model, st, idx, params, x, bufs):
    (wq, bq, wk, bk, wv, bv, wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2) = params
    cfg, wc, dt = model.config, model.weight_cache, st.dtype
    B, S, T = st.dims
    H, nh = cfg.hidden_size, cfg.num_attention_heads
    seeds = st.seeds
    site = 16 * (idx + 1)
    Wqkv = wc.cat_rows(f"L{idx}.qkv", [wq, wk, wv], dt)
    bqkv = wc.get((f"L{idx}.bqkv",), [bq, bk, bv], lambda: torch.cat([bq.detach(), bk.detach(), bv.detach()]))
    Wo, Wi, Wo2 = wc.cast(f"L{idx}.o", wo, dt), wc.cast(f"L{idx}.i", wi, dt), wc.cast(f"L{idx}.o2", wo2, dt)
    L = EncoderLayer()
    L.Wqkv, L.bqkv, L.Wo, L.bo, L.g1, L.b1 = Wqkv.data_ptr(), bqkv.data_ptr(), Wo.data_ptr(), bo.data_ptr(), g1.data_ptr(), b1.data_ptr()
    L.Wi, L.bi, L.Wo2, L.bo2, L.g2, L.b2 = Wi.data_ptr(), bi.data_ptr(), Wo2.data_ptr(), bo2.data_ptr(), g2.data_ptr(), b2.data_ptr()
    if st.bias is not None:
        L.bias, L.bias_ld = st.bias.data_ptr(), st.bias.shape[-1]
    if st.key_bias is not None:
        L.key_bias = st.key_bias.data_ptr()
    words = seeds.attn_words(idx, cfg.num_hidden_layers, B, nh, T, x.device)
    if words is not None:
        L.drop_words = words.data_ptr()
    L.x = x.data_ptr()
    bufs.fill(L)
    L.B, L.T, L.H, L.nh, L.I = B, T, H, nh, cfg.intermediate_size
    L.eps, L.attn_scale, L.p_hidden, L.p_attn = cfg.layer_norm_eps, 1.0 / math.sqrt(H // nh), seeds.p_hidden, seeds.p_attn
    L.seed_o, L.seed_o2 = seeds.seed(site + 2), seeds.seed(site + 3)
    return L, (Wqkv, bqkv, Wo, Wi, Wo2, words)      # (the working copies stay referenced while the call is in flight)


def _stage_forward(ctx, model, st, idx, x, params):
    cfg = model.config
    B, S, T = st.dims
    H, nh, I = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
    R, dev = B * T, x.device
    if not x.is_contiguous():
        x = x.contiguous()
    # (needs_input_grad reflects requires_grad under torch.no_grad() too: an eval forward must not allocate / store zi)
    bufs = _LayerBuffers(R, H, I, B * nh * T, st.grad_mode and any(ctx.needs_input_grad), dev)
    L, keep = _describe_layer(model, st, idx, params, x, bufs)
    out = torch.empty((R, H), dtype=st.dtype, device=dev)
    wsb = _layer_ws_bytes(R, H, I, 0)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
    check(hip_lib().peneo_encoder_layer_fwd(C.byref(L), out.data_ptr(), ptr(ws), wsb, hip_stream()), "peneo_encoder_layer_fwd")
    ctx.model, ctx.st, ctx.idx = model, st, idx
    ctx.saved = (x, bufs)
    ctx.params = params
    ctx.stage_call = True
    return out


def _stage_backward(ctx, d_out):
    model, st, idx = ctx.model, ctx.st, ctx.idx
    params = ctx.params
    x, bufs = ctx.saved
    cfg, dt = model.config, st.dtype
    B, S, T = st.dims
    H, nh, I = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size
    R, dev = B * T, x.device
    seeds = st.seeds
    d_out = d_out.contiguous()
    L, keep = _describe_layer(model, st, idx, params, x, bufs)
    main = torch.cuda.current_stream()
    side = model.side_stream(dev) if model.wgrad_on_side_stream else None
    # small fp32 accumulators of all layers: one zero fill per step (the last layer runs backward first)
    psz = 4 * H + (H + I + H + 3 * H)
    nl = len(model.encoder.layer)
    if getattr(st, "grad_pools", None) is None or idx == nl - 1:
        st.grad_pools = torch.zeros((nl, psz), dtype=torch.float32, device=dev)
    pool = st.grad_pools[idx]
    dg2, db2, dg1, db1 = pool[:H], pool[H:2 * H], pool[2 * H:3 * H], pool[3 * H:4 * H]
    o_ = 4 * H
    dbo2, dbi, dbo, dbqkv = pool[o_:o_ + H], pool[o_ + H:o_ + H + I], pool[o_ + H + I:o_ + 2 * H + I], pool[o_ + 2 * H + I:]
    if st.bias is not None and model.rel_tables_need_grad():
        # this layer's dS^T goes to its own bf16 slab; the three bias tables are reduced from all slabs once (layer 0, below)
        if st.ds_layers is None:
            st.ds_layers = torch.empty((cfg.num_hidden_layers, B, nh, T, st.bias.shape[-1]), dtype=dt, device=dev)
        ds = st.ds_layers[idx]
    else:   # nobody wants the slab but the dQ kernel of this call
        ds = torch.empty((B, nh, T, ops.attn_padded_len(T)), dtype=dt, device=dev)
    drop = seeds.p_hidden > 0
    # activation-gradient scratch: d_h2 | d_a | d_h1 | d_att | (d_dense2 | d_dense1) | d_zi | dqkv   (one allocation)
    nsc = R * (4 * H + (2 * H if drop else 0) + I + 3 * H)
    scratch = torch.empty(nsc, dtype=dt, device=dev)
    delta = torch.empty(B * nh * T, dtype=torch.float32, device=dev)
    d_x = torch.empty((R, H), dtype=dt, device=dev)
    dwqkv = torch.empty((3 * H, H), dtype=torch.float32, device=dev)
    dwo = torch.empty((H, H), dtype=torch.float32, device=dev)
    dwi = torch.empty((I, H), dtype=torch.float32, device=dev)
    dwo2 = torch.empty((H, I), dtype=torch.float32, device=dev)
    G = EncoderLayerGrads()
    G.d_out, G.d_x = d_out.data_ptr(), d_x.data_ptr()
    b, e = scratch.data_ptr(), 2
    G.d_h2 = b
    G.d_a = b + e * R * H
    G.d_h1 = G.d_a + e * R * H
    G.d_att = G.d_h1 + e * R * H
    nxt = G.d_att + e * R * H
    if drop:
        G.d_dense2, G.d_dense1 = nxt, nxt + e * R * H
        nxt += 2 * e * R * H
    G.d_zi = nxt
    G.dqkv = nxt + e * R * I
    G.delta = delta.data_ptr()
    G.ds_out = ds.data_ptr()
    G.dwqkv, G.dwo, G.dwi, G.dwo2 = dwqkv.data_ptr(), dwo.data_ptr(), dwi.data_ptr(), dwo2.data_ptr()
    p0 = pool.data_ptr()
    G.dg2, G.db2, G.dg1, G.db1 = p0, p0 + 4 * H, p0 + 8 * H, p0 + 12 * H
    G.dbo2 = p0 + 4 * o_
    G.dbi = G.dbo2 + 4 * H
    G.dbo = G.dbi + 4 * I
    G.dbqkv = G.dbo + 4 * H
    wmb, wsb = _layer_ws_bytes(R, H, I, 1), _layer_ws_bytes(R, H, I, 2)
    wm = torch.empty(wmb, dtype=torch.uint8, device=dev) if wmb else None
    wsd = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
    check(hip_lib().peneo_encoder_layer_bwd(C.byref(L), C.byref(G), ptr(wm), wmb, ptr(wsd), wsb, main.cuda_stream,
                                            side.cuda_stream if side is not None else None), "peneo_encoder_layer_bwd")
    if idx == 0 and st.ds_layers is not None:
        # the bias-table gradient of all layers from their dS^T slabs, on its own stream (joined by _EmbedStage.backward)
        rel_stream = model.side_stream(dev, "rel")
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(rel_stream):
            rel_stream.wait_event(ev)
            model.reduce_rel_group(st, 0, cfg.num_hidden_layers, B, T)
    if side is not None:
        if model.defer_wgrad_join and can_defer(params):
            defer_join(side, keep=(scratch, bufs.flat, x, wsd, *keep))     # what the side stream still reads
        else:
            main.wait_stream(side)
    grads = (dwqkv[:H], dbqkv[:H], dwqkv[H:2 * H], dbqkv[H:2 * H], dwqkv[2 * H:], dbqkv[2 * H:], dwo, dbo, dg1, db1,
             dwi, dbi, dwo2, dbo2, dg2, db2)
    grads = tuple(gr if p.requires_grad else None for gr, p in zip(grads, params))
    return (None, None, None, d_x) + grads

# ------------------------------------------------------------------------------------------------
# stage 2: one encoder layer
# ------------------------------------------------------------------------------------------------
class _LayerStage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, st, idx, x, *params):
        ctx.stage_call = False
        if _use_stage_calls(model, st, model.config.hidden_size, model.config.intermediate_size):
            return _stage_forward(ctx, model, st, idx, x, params)
        (wq, bq, wk, bk, wv, bv, wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2) = params
        cfg, wc, dt = model.config, model.weight_cache, st.dtype
        B, S, T = st.dims
        H, nh = cfg.hidden_size, cfg.num_attention_heads
        d = H // nh
        seeds = st.seeds
        site = 16 * (idx + 1)
        Wqkv = wc.cat_rows(f"L{idx}.qkv", [wq, wk, wv], dt)
        bqkv = wc.get((f"L{idx}.bqkv",), [bq, bk, bv], lambda: torch.cat([bq.detach(), bk.detach(), bv.detach()]))
        Wo, Wi, Wo2 = wc.cast(f"L{idx}.o", wo, dt), wc.cast(f"L{idx}.i", wi, dt), wc.cast(f"L{idx}.o2", wo2, dt)

        qkv = ops.gemm(x, Wqkv, bias=bqkv)
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        att, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 1.0 / math.sqrt(d), st.bias, st.key_bias, drop_p=seeds.p_attn,
                                drop_words=seeds.attn_words(idx, cfg.num_hidden_layers, B, nh, T, x.device))
        h1 = ops.gemm(att, Wo, bias=bo, residual=x, drop_p=seeds.p_hidden, drop_seed=seeds.seed(site + 2))
        a, m1, r1 = ops.layernorm_fwd(h1, g1, b1, cfg.layer_norm_eps)
        # the GELU pre-activation is only needed by the backward: inference (no input needs a gradient) skips its 35 MB store
        zi = torch.empty((B * T, cfg.intermediate_size), dtype=dt, device=x.device) if (st.grad_mode and any(ctx.needs_input_grad)) else None
        inter = ops.gemm(a, Wi, bias=bi, act=ACT_GELU, preact=zi)
        h2 = ops.gemm(inter, Wo2, bias=bo2, residual=a, drop_p=seeds.p_hidden, drop_seed=seeds.seed(site + 3))
        out, m2, r2 = ops.layernorm_fwd(h2, g2, b2, cfg.layer_norm_eps)
        ctx.model, ctx.st, ctx.idx = model, st, idx
        ctx.saved = (x, qkv, att, lse, h1, m1, r1, a, zi, inter, h2, m2, r2)
        ctx.params = params
        return out

    @staticmethod
    def backward(ctx, d_out):
        if ctx.stage_call:
            return _stage_backward(ctx, d_out)
        model, st, idx = ctx.model, ctx.st, ctx.idx
        (wq, bq, wk, bk, wv, bv, wo, bo, g1, b1, wi, bi, wo2, bo2, g2, b2) = ctx.params
        x, qkv, att, lse, h1, m1, r1, a, zi, inter, h2, m2, r2 = ctx.saved
        cfg, wc, dt = model.config, model.weight_cache, st.dtype
        B, S, T = st.dims
        H, nh = cfg.hidden_size, cfg.num_attention_heads
        d = H // nh
        seeds = st.seeds
        site = 16 * (idx + 1)
        dev = x.device
        Wqkv = wc.cat_rows(f"L{idx}.qkv", [wq, wk, wv], dt)
        Wo, Wi, Wo2 = wc.cast(f"L{idx}.o", wo, dt), wc.cast(f"L{idx}.i", wi, dt), wc.cast(f"L{idx}.o2", wo2, dt)
        f32 = lambda p: torch.zeros(p.shape, dtype=torch.float32, device=dev)
        d_out = d_out.contiguous()

        # The parameter-gradient work (wgrad GEMMs, bias column sums) is off the activation-gradient critical path: it
        # goes to a second HIP stream and fills the CUs the small / tail-heavy critical-path kernels leave idle.  The
        # main stream joins the side stream before the stage returns (autograd consumes the gradients on main).
        main = torch.cuda.current_stream()
        side = model.side_stream(dev) if model.wgrad_on_side_stream else None

        def on_side(fn):
            if side is None:
                return fn()
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                return fn()

        def wgrad(dy, xin):
            return ops.gemm(dy, xin, a_kmajor=False, b_kmajor=False, out_dtype=torch.float32)
        # all small fp32 accumulators of the stage (LayerNorm and bias gradients) carved from ONE zero-filled buffer:
        # one fill launch instead of eight fills / memsets per layer
        I = cfg.intermediate_size
        psz = 4 * H + (H + I + H + 3 * H)
        nl = len(model.encoder.layer)
        if getattr(st, "grad_pools", None) is None or idx == nl - 1:   # the last layer runs backward first: it zero-fills all pools
            st.grad_pools = torch.zeros((nl, psz), dtype=torch.float32, device=dev)
        pool = st.grad_pools[idx]
        dg2, db2, dg1, db1 = pool[:H], pool[H:2 * H], pool[2 * H:3 * H], pool[3 * H:4 * H]
        o_ = 4 * H
        dbo2, dbi, dbo, dbqkv = pool[o_:o_ + H], pool[o_ + H:o_ + H + I], pool[o_ + H + I:o_ + 2 * H + I], pool[o_ + 2 * H + I:]

        # LayerNorm backward writes d_h (for the residual branch) and, in the same pass, d_h through the dropout mask of
        # the dense layer that fed the LayerNorm (for that layer's dgrad / wgrad)
        def ln_bwd(dy, hx, g, m, r, dg, db, dxd, seed_):
            return ops.layernorm_bwd(dy, hx, g, m, r, dg, db, dx_dropped=dxd, drop2_p=seeds.p_hidden, drop2_seed=seed_)

        d_dense2 = torch.empty_like(h2) if seeds.p_hidden > 0 else None
        d_h2 = ln_bwd(d_out, h2, g2, m2, r2, dg2, db2, d_dense2, seeds.seed(site + 3))
        if d_dense2 is None:
            d_dense2 = d_h2
        d_zi = ops.gemm(d_dense2, Wo2, b_kmajor=False, grad_src=zi, grad_act=ACT_GELU)
        d_a = ops.gemm(d_zi, Wi, b_kmajor=False, residual=d_h2)
        d_dense1 = torch.empty_like(h1) if seeds.p_hidden > 0 else None
        d_h1 = ln_bwd(d_a, h1, g1, m1, r1, dg1, db1, d_dense1, seeds.seed(site + 2))
        if d_dense1 is None:
            d_dense1 = d_h1
        d_att = ops.gemm(d_dense1, Wo, b_kmajor=False)

        dqkv = torch.empty_like(qkv)
        q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
        ds_out = None
        if st.bias is not None and model.rel_tables_need_grad():
            if dt == torch.bfloat16 and H % 8 == 0:
                # single-pass kernel: this layer's dS^T goes to its own bf16 slab (12 x 105 MB at B = 8: nothing against
                # 288 GB); the three tables are reduced from all slabs once, in _EmbedStage.backward
                if st.ds_layers is None:
                    L = cfg.num_hidden_layers
                    st.ds_layers = torch.empty((L, B, nh, T, st.bias.shape[-1]), dtype=dt, device=dev)   # every element is written
                ds_out = st.ds_layers[idx]
            elif st.g_bias is None:
                st.g_bias = torch.zeros(st.bias.shape, dtype=torch.float32, device=dev)
        # The FFN / output-projection weight gradients and bias column sums do not start beside the dgrad GEMMs that produce
        # their operands (two MFMA-bound kernels then share the CUs and both take longer) but when the attention backward
        # starts: its second, thin round of workgroups leaves most CUs idle (+0.7 %; the event sits behind the d_att GEMM)
        _, dwo2, _, dwi, _, dwo = on_side(lambda: (ops.colsum(d_dense2, out=dbo2, accumulate=True), wgrad(d_dense2, inter),
                                                   ops.colsum(d_zi, out=dbi, accumulate=True), wgrad(d_zi, a),
                                                   ops.colsum(d_dense1, out=dbo, accumulate=True), wgrad(d_dense1, att)))
        ops.attn_bwd(q, k, v, att, d_att, lse, B, nh, T, d, 1.0 / math.sqrt(d), st.bias, st.key_bias, dqkv,
                     st.g_bias if ds_out is None else None, drop_p=seeds.p_attn,
                     drop_words=seeds.attn_words(idx, cfg.num_hidden_layers, B, nh, T, q.device), ds_out=ds_out)
        _, dwqkv = on_side(lambda: (ops.colsum(dqkv, out=dbqkv, accumulate=True), wgrad(dqkv, x)))
        d_x = ops.gemm(dqkv, Wqkv, b_kmajor=False, residual=d_h1)
        if ds_out is not None and idx == 0:
            # the bias-table gradient of all layers (their dS^T slabs are complete), on its own stream (joined only by
            # _EmbedStage.backward): LDS-atomic and HBM bound.  It goes out BEHIND this stage's last dgrad GEMM: started in front
            # of it, the 277 us reduction stretched that GEMM from 54 to 247 us at the tail of the step
            rel_stream = model.side_stream(dev, "rel")
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(rel_stream):
                rel_stream.wait_event(ev)
                model.reduce_rel_group(st, idx, cfg.num_hidden_layers, B, T)
        if side is not None:
            if model.defer_wgrad_join and can_defer(ctx.params):
                # joined one stage later (engine.py): the critical path does not wait for dW_qkv
                defer_join(side, keep=(d_dense2, inter, d_zi, a, d_dense1, att, dqkv, x))
            else:
                main.wait_stream(side)    # gradients are accumulated into existing .grad tensors on main right after this
        grads = (dwqkv[:H], dbqkv[:H], dwqkv[H:2 * H], dbqkv[H:2 * H], dwqkv[2 * H:], dbqkv[2 * H:], dwo, dbo, dg1, db1,
                 dwi, dbi, dwo2, dbo2, dg2, db2)
        grads = tuple(gr if p.requires_grad else None for gr, p in zip(grads, ctx.params))
        return (None, None, None, d_x) + grads


def layer_params(layer: LayoutLMv3Layer) -> List[torch.Tensor]:
    s, o = layer.attention.self, layer.attention.output
    return [s.query.weight, s.query.bias, s.key.weight, s.key.bias, s.value.weight, s.value.bias, o.dense.weight,
            o.dense.bias, o.LayerNorm.weight, o.LayerNorm.bias, layer.intermediate.dense.weight,
            layer.intermediate.dense.bias, layer.output.dense.weight, layer.output.dense.bias,
            layer.output.LayerNorm.weight, layer.output.LayerNorm.bias]


# ------------------------------------------------------------------------------------------------
# the backbone module
# ------------------------------------------------------------------------------------------------
class LayoutLMv3Model(nn.Module):
    """Counterpart of the reference ``LayoutLMv3Model`` for the PEneo call pattern
    (``forward(input_ids, bbox, attention_mask, image)`` -> ``(last_hidden_state [B, T, H],)``)."""

    config_class = LayoutLMv3Config

    def __init__(self, config: LayoutLMv3Config):
        super().__init__()
        self.config = config
        H = config.hidden_size
        if 4 * config.coordinate_size + 2 * config.shape_size != H:
            raise ValueError("4 * coordinate_size + 2 * shape_size must equal hidden_size")
        self.embeddings = LayoutLMv3Embeddings(config)
        self.encoder = LayoutLMv3Encoder(config)
        if config.visual_embed:
            self.patch_embed = PatchEmbed(H)
            size = int(config.input_size / 16)
            self.cls_token = nn.Parameter(torch.zeros(1, 1, H))
            self.pos_embed = nn.Parameter(torch.zeros(1, size * size + 1, H))
            self.LayerNorm = nn.LayerNorm(H, eps=config.layer_norm_eps)
            self.norm = nn.LayerNorm(H, eps=1e-6)
        self.weight_cache = WeightCache()
        self.compute_dtype = torch.float32
        self.wgrad_on_side_stream = os.environ.get("PENEO_WGRAD_STREAM", "1") != "0"
        self.defer_wgrad_join = os.environ.get("PENEO_DEFER_JOIN", "1") != "0"
        self._luts = {}

    # ---- small host-side constants ---------------------------------------------------------
    def lut(self, kind: str, bins: int, max_dist: int, dev) -> torch.Tensor:
        key = (kind, bins, max_dist, str(dev))
        if key not in self._luts:
            self._luts[key] = bucket_lut(bins, max_dist, 1024).to(dev)
        return self._luts[key]

    # ---- the input range check as a device flag (see _EmbedStage) ----------------------------
    def input_status(self, dev) -> torch.Tensor:
        """int32 [1] on `dev`, sticky: set to 1 by the embedding kernel when an id / position / box coordinate is out of range."""
        st = self._luts.get(("input_status", str(dev)))
        if st is None:
            st = self._luts[("input_status", str(dev))] = torch.zeros(1, dtype=torch.int32, device=dev)
        return st

    def raise_on_bad_inputs(self) -> None:
        """Read (one host sync per flag) and clear the input flags; raises the reference's IndexError if any forward since the last
        call saw a bbox coordinate outside its table (reference modeling_layoutlmv3.py:133 / modeling_lilt.py)."""
        for key, st in self._luts.items():
            if key[0] == "input_status" and int(st) != 0:
                st.zero_()
                raise IndexError("The :obj:`bbox` coordinate values should be within 0-1000 range.")

    def side_stream(self, device, which: str = "wgrad") -> "torch.cuda.Stream":
        return engine_side_stream(device, which)

    def reduce_rel_group(self, st, lo: int, hi: int, B: int, T: int) -> None:
        """Bias-table gradients of layers lo..hi-1 from their bf16 dS^T slabs (runs on the caller's current stream)."""
        cfg = self.config
        d = cfg.hidden_size // cfg.num_attention_heads
        nh = cfg.num_attention_heads
        dev = st.ds_layers.device
        if st.bucket_t is None:   # transposed bucket maps = the bucket kernel on negated positions
            pos_t, xs, ys, lut1, lut2 = st.bucket_inputs
            neg = lambda t: (-t).contiguous() if t is not None else None
            st.bucket_t = ops.relpos_buckets(neg(pos_t), neg(xs), neg(ys), B, T, lut1, cfg.rel_pos_bins // 2, lut2,
                                             cfg.rel_2d_pos_bins // 2)
            z = lambda on, bins: torch.zeros((nh, bins), dtype=torch.float32, device=dev) if on else None
            st.rel_grads = (z(cfg.has_relative_attention_bias, cfg.rel_pos_bins),
                            z(cfg.has_spatial_attention_bias, cfg.rel_2d_pos_bins),
                            z(cfg.has_spatial_attention_bias, cfg.rel_2d_pos_bins))
        bt, rg = st.bucket_t, st.rel_grads
        ops.relpos_bias_bwd_layers(st.ds_layers[lo:hi], bt[0], bt[1], bt[2], rg[0], rg[1], rg[2], 1.0 / math.sqrt(d))

    def rel_tables_need_grad(self) -> bool:
        enc = self.encoder
        return any(p.requires_grad for p in (getattr(enc, "rel_pos_bias", None), getattr(enc, "rel_pos_x_bias", None),
                                             getattr(enc, "rel_pos_y_bias", None)) if p is not None for p in p.parameters())

    def visual_xy(self, dev, nv: int):
        key = ("vxy", nv, str(dev))
        if key not in self._luts:
            grid = int(round(math.sqrt(nv - 1)))
            vx, vy = visual_xy(grid)
            self._luts[key] = (vx.to(dev), vy.to(dev))
        return self._luts[key]

    def _init_weights(self, module) -> None:
        """Reference rule (modeling_layoutlmv3.py:260-274)."""
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

    def embed_params(self) -> List[torch.Tensor]:
        e, cfg = self.embeddings, self.config
        ps = [e.word_embeddings.weight, e.token_type_embeddings.weight, e.position_embeddings.weight,
              e.x_position_embeddings.weight, e.y_position_embeddings.weight, e.h_position_embeddings.weight,
              e.w_position_embeddings.weight, e.LayerNorm.weight, e.LayerNorm.bias]
        if cfg.visual_embed:
            ps += [self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.cls_token, self.pos_embed,
                   self.norm.weight, self.norm.bias, self.LayerNorm.weight, self.LayerNorm.bias]
        else:
            raise PeneoHipError("LayoutLMv3 without visual_embed is not wired up (PEneo always has it)")
        if cfg.has_relative_attention_bias:
            ps.append(self.encoder.rel_pos_bias.weight)
        if cfg.has_spatial_attention_bias:
            ps += [self.encoder.rel_pos_x_bias.weight, self.encoder.rel_pos_y_bias.weight]
        return ps

    def refresh_working_weights(self, dt: torch.dtype) -> None:
        """bf16 working copies of all encoder-layer matrices (fused QKV [3H, H], output, intermediate, output2) in ONE launch when
        any parameter has changed since the last forward (85 M of the 127 M parameters; the per-key lazy casts of WeightCache cost
        72 launches = 0.6 ms per step behind an optimizer update).  The entries land in the weight cache under the keys and stamps
        the layer stages look up."""
        if dt != torch.bfloat16:
            return
        wc = self.weight_cache
        groups = []                                            # (cache key, [parameters], rows)
        for i, layer in enumerate(self.encoder.layer):
            s_, o_ = layer.attention.self, layer.attention.output
            groups.append(((f"L{i}.qkv", dt), [s_.query.weight, s_.key.weight, s_.value.weight]))
            groups.append(((f"L{i}.o", dt), [o_.dense.weight]))
            groups.append(((f"L{i}.i", dt), [layer.intermediate.dense.weight]))
            groups.append(((f"L{i}.o2", dt), [layer.output.dense.weight]))
        params = [p for _, ps in groups for p in ps]
        stamp = WeightCache._stamp(params)
        if getattr(self, "_wstamp", None) == stamp:
            return
        plan = getattr(self, "_wplan", None)
        ptrs = tuple(p.data_ptr() for p in params)
        if plan is None or plan[0] != ptrs:
            bufs, pairs = [], []
            for _, ps in groups:
                buf = torch.empty((sum(p.shape[0] for p in ps), ps[0].shape[1]), dtype=dt, device=ps[0].device)
                r = 0
                for p in ps:
                    pairs.append((p.detach(), buf[r:r + p.shape[0]]))
                    r += p.shape[0]
                bufs.append(buf)
            plan = (ptrs, ops.CastPlan(pairs), bufs)
            self._wplan = plan
        plan[1].run()
        for (key, ps), buf in zip(groups, plan[2]):
            wc._store[key] = (WeightCache._stamp(ps), buf)
        self._wstamp = stamp

    def forward(self, input_ids=None, bbox=None, attention_mask=None, image=None, **unused):
        if input_ids is None:
            raise ValueError("You have to specify input_ids")
        if not input_ids.is_cuda:
            raise PeneoHipError("peneo_amd runs on the GPU only: move the model and the batch to 'cuda' "
                                "(there is no CPU fallback; the CPU oracle lives in oracle/ for tests)")
        cfg = self.config
        B, S = input_ids.shape
        if bbox is None:
            bbox = torch.zeros((B, S, 4), dtype=torch.long, device=input_ids.device)
        if attention_mask is not None and attention_mask.dtype != torch.int64:
            attention_mask = attention_mask.long()       # (None = every text token attended: the mask kernel takes NULL)
        if bbox.dtype != torch.int64:
            bbox = bbox.long()
        self.refresh_working_weights(self.compute_dtype)
        st = _FwdState()
        st.dtype = self.compute_dtype
        st.grad_mode = torch.is_grad_enabled()
        st.seeds = DropoutSeeds(self.training, cfg.hidden_dropout_prob, cfg.attention_probs_dropout_prob)
        x = _EmbedStage.apply(self, st, input_ids.contiguous(), bbox.contiguous(),
                              attention_mask.contiguous() if attention_mask is not None else None, image,
                              *self.embed_params())
        _, _, T = st.dims
        for i, layer in enumerate(self.encoder.layer):
            x = _LayerStage.apply(self, st, i, x, *layer_params(layer))
        if st.fill_event is not None:
            # the gradient buffer belongs to the MAIN stream's pool: if the graph is dropped without a backward (a validation
            # pass run in grad mode) its block may be handed out again, so the main stream must have seen the fill complete.
            # The fill ended long ago (it ran beside the embedding stage): the wait is free
            torch.cuda.current_stream(input_ids.device).wait_event(st.fill_event)
        return (x.view(B, T, cfg.hidden_size),)
