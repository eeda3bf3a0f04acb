"""PEneo decoder on libpeneo_hip kernels (reference: model/peneo_decoder.py).

``HandshakingTaggingScheme`` (label map <-> spots, :12-115), ``HandshakingKernel`` (:118-177),
``PEneoOutput`` (:180-198) and ``PEneoDecoder`` (:201-443) keep the reference's names, parameter
layout (``shrink_projection.{0,3}``, ``handshaking_kernel.combine_fc``, ``<head>_fc.{0,3}``,
``link_loss.weight`` / ``le_loss.weight`` buffers) and outputs.  The forward never materialises the
[B, P, D] pair tensor: K10 runs as two GEMMs with fused bias+SiLU(+dropout), K11+K12+K13 as the fused
pair-heads kernel; the backward re-creates the pair activations chunk by chunk (rows of the
triangle) so that its three GEMMs per chunk stay Infinity-Cache resident.
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn
from transformers.modeling_outputs import ModelOutput

from .. import ops
from ..hip import ACT_NONE, ACT_SILU, PeneoHipError
from .engine import DropoutSeeds, WeightCache, big_acquire, big_clear, big_release, can_defer, defer_join, join_pending, mark_late

HEAD_NAMES = ("line_extraction", "ent_linking_h2h", "ent_linking_t2t", "line_grouping_h2h", "line_grouping_t2t")
TAG_KWARGS = ("line_extraction_shaking_tag", "ent_linking_head_rel_shaking_tag", "ent_linking_tail_rel_shaking_tag",
              "line_grouping_head_rel_shaking_tag", "line_grouping_tail_rel_shaking_tag")
HEAD_CLASSES = (2, 3, 3, 3, 3)
_HEAD_OF_CLASS_SLOT: dict = {}   # device -> int64 [sum(HEAD_CLASSES)]: head index of every class slot


class HandshakingTaggingScheme:
    """Label-map <-> spot conversion (reference :12-115).  The packed index of pair (i, j), i <= j, of a
    length-n sequence is p = i*n - i*(i-1)/2 + (j - i)."""

    @staticmethod
    def spots2shaking_tag(spots: List[Tuple], seq_len: int) -> torch.Tensor:
        """NOTE: the reference builds an all-zero index map here, so every spot lands on p = 0
        (:26-31); only ``spots2shaking_tag4batch`` is used by its collator.  Reproduced as is."""
        tag = torch.zeros(seq_len * (seq_len + 1) // 2).long()
        for sp in spots:
            tag[0] = sp[2]
        return tag

    @staticmethod
    def spots2shaking_tag4batch(batch_spots, shaking_ind2matrix_ind=None, matrix_ind2shaking_ind=None,
                                seq_len: int = None) -> torch.Tensor:
        if shaking_ind2matrix_ind is not None and matrix_ind2shaking_ind is not None:
            seq_len = len(matrix_ind2shaking_ind)
        elif seq_len is None:
            raise ValueError("If shaking_ind2matrix_ind and matrix_ind2shaking_ind are not provided,seq_len must be given")
        n = seq_len
        out = torch.zeros(len(batch_spots), n * (n + 1) // 2).long()
        for b, spots in enumerate(batch_spots):
            for sp in spots:
                i, j = sp[0], sp[1]
                if matrix_ind2shaking_ind is not None:
                    p = matrix_ind2shaking_ind[i][j]
                else:
                    p = i * n - i * (i - 1) // 2 + (j - i) if i <= j else 0  # reference's table is 0 below the diagonal
                out[b][p] = sp[2]
        return out

    @staticmethod
    def spots2shaking_tag4batch_device(batch_spots, seq_len: int, device) -> torch.Tensor:
        """Same result as ``spots2shaking_tag4batch(batch_spots, seq_len=seq_len)`` but built by one device kernel from the
        sparse spots: no O(N^2) host table, no 5.2 MB/document label copy."""
        return ops.spots_to_tags(batch_spots, seq_len, device)

    @staticmethod
    def get_spots_from_shaking_tag(shaking_tag: torch.Tensor, shaking_ind2matrix_ind=None, seq_len: int = None):
        """[P, C] logits (or [P] tags) -> [(i, j, tag, score)] in increasing p order (reference :76-115).
        Device tensors go through the fused softmax/argmax/compaction kernel (K14) — one launch and one
        copy instead of a Python loop with three ``.item()`` syncs per spot."""
        if shaking_ind2matrix_ind is None and seq_len is None:
            raise ValueError("If shaking_ind2matrix_ind and matrix_ind2shaking_ind are not provided,seq_len must be given")
        P = shaking_tag.shape[0]
        if seq_len is None:
            seq_len = int(((8 * P + 1) ** 0.5 - 1) // 2)
        n = seq_len
        if shaking_tag.dim() > 1 and shaking_tag.shape[-1] > 1 and shaking_tag.is_cuda:
            spots, scores = ops.spots_compact(shaking_tag.float().contiguous(), n)
            sp, sc = spots.cpu().tolist(), scores.cpu().tolist()
            return [(a, b, t, s) for (a, b, t), s in zip(sp, sc)]
        if shaking_tag.dim() > 1 and shaking_tag.shape[-1] > 1:
            prob = shaking_tag.softmax(-1)
            pred, score = prob.argmax(-1), prob.max(-1)[0]
        else:
            pred, score = shaking_tag.view(-1), torch.ones_like(shaking_tag.view(-1), dtype=torch.float32)
        out = []
        for p in torch.nonzero(pred)[:, 0].tolist():
            if shaking_ind2matrix_ind is not None:
                i, j = shaking_ind2matrix_ind[p]
            else:
                i = int((2 * n + 1 - ((2 * n + 1) ** 2 - 8 * p) ** 0.5) // 2)
                while i > 0 and i * n - i * (i - 1) // 2 > p:
                    i -= 1
                while i < n - 1 and (i + 1) * n - (i + 1) * i // 2 <= p:
                    i += 1
                j = i + p - (i * n - i * (i - 1) // 2)
            out.append((i, j, int(pred[p]), float(score[p])))
        return out


@dataclass
class PEneoOutput(ModelOutput):
    loss: Optional[torch.Tensor] = None

    line_extraction_loss: Optional[torch.Tensor] = None
    ent_linking_h2h_loss: Optional[torch.Tensor] = None
    ent_linking_t2t_loss: Optional[torch.Tensor] = None
    line_grouping_h2h_loss: Optional[torch.Tensor] = None
    line_grouping_t2t_loss: Optional[torch.Tensor] = None

    line_extraction_shaking_outputs: Optional[torch.Tensor] = None
    ent_linking_h2h_shaking_outputs: Optional[torch.Tensor] = None
    ent_linking_t2t_shaking_outputs: Optional[torch.Tensor] = None
    line_grouping_h2h_shaking_outputs: Optional[torch.Tensor] = None
    line_grouping_t2t_shaking_outputs: Optional[torch.Tensor] = None

    attentions: Optional[Tuple[torch.FloatTensor]] = None
    hidden_states: Optional[Tuple[torch.FloatTensor]] = None
    orig_bbox: Optional[torch.Tensor] = None


class HandshakingKernel(nn.Module):
    """Parameter holder for ``combine_fc`` (Linear(2D -> D)); the pair expansion itself is fused into the
    pair-heads kernel via combine_fc(cat(h_i, h_j)) = W[:, :D] h_i + (W[:, D:] h_j + b)."""

    def __init__(self, hidden_size: int) -> None:
        super().__init__()
        self.combine_fc = nn.Linear(hidden_size * 2, hidden_size)
        self.activation = nn.SiLU()


class _ClassWeightedCE(nn.Module):
    """Holds the ``weight`` buffer of the reference's CrossEntropyLossOHEM (model/custom_loss.py:104-202)."""

    def __init__(self, weight: torch.Tensor, num_hard_positive: int, num_hard_negative: int):
        super().__init__()
        self.register_buffer("weight", weight)
        # -1 / -1: plain class-weighted mean (custom_loss.py:189-202, fused into the pair-heads kernel); anything else runs
        # the OHEM branch (:204-288) through peneo_ohem_ce on the logit maps
        self.num_hard_positive, self.num_hard_negative = int(num_hard_positive), int(num_hard_negative)

    @property
    def ohem(self) -> bool:
        return self.num_hard_positive != -1 or self.num_hard_negative != -1


def _row_chunks(n: int, max_pairs: int) -> List[Tuple[int, int]]:
    """Split rows 0..n of the pair triangle into [i0, i1) ranges of at most ~max_pairs pairs."""
    out, i0, acc = [], 0, 0
    for i in range(n):
        row = n - i
        if acc > 0 and acc + row > max_pairs:
            out.append((i0, i))
            i0, acc = i, 0
        acc += row
    out.append((i0, n))
    return out


class _DecoderStage(torch.autograd.Function):
    """shrink MLP -> [a | b] projection -> fused pair heads (+ CE).  Inputs: cropped sequence output
    [B*N, Hin]; outputs: total loss, 5 per-head losses, 5 logit maps."""

    @staticmethod
    def forward(ctx, dec, seq, B, N, tags, want_logits, need_grad, *params):
        wc, dt = dec.weight_cache, seq.dtype
        it = iter(params)
        if dec.decoder_shrink:
            w0, b0, w3, b3 = next(it), next(it), next(it), next(it)
        wc_w, wc_b = next(it), next(it)
        kcls = dec.num_cls_layers
        heads = [tuple(next(it) for _ in range(2 * kcls)) for _ in HEAD_NAMES]
        D = wc_w.shape[0]
        seeds = DropoutSeeds(dec.training, dec.dropout_p, 0.0)
        dev = seq.device
        saved = dict(seeds=seeds)
        h = seq
        if dec.decoder_shrink:
            W0, W3 = wc.cast("dec.s0", w0, dt), wc.cast("dec.s3", w3, dt)
            z1 = torch.empty((h.shape[0], w0.shape[0]), dtype=dt, device=dev)
            s1 = ops.gemm(h, W0, bias=b0, act=ACT_SILU, preact=z1, drop_p=seeds.p_hidden, drop_seed=seeds.seed(901))
            z2 = torch.empty((h.shape[0], w3.shape[0]), dtype=dt, device=dev)
            s2 = ops.gemm(s1, W3, bias=b3, act=ACT_SILU, preact=z2, drop_p=seeds.p_hidden, drop_seed=seeds.seed(902))
            saved.update(z1=z1, s1=s1, z2=z2)
            h = s2
        saved["s2"] = h
        # [a | b] = h [Wc[:, :D]; Wc[:, D:]]^T + [0 | bc]
        Wab = dec.stacked_combine_weight(wc_w, dt)
        bab = wc.get(("dec.bab",), [wc_b], lambda: torch.cat([torch.zeros_like(wc_b.detach()), wc_b.detach()]))
        ab = ops.gemm(h, Wab, bias=bab).view(B, N, 2 * D)
        cws = [dec.le_loss.weight] + [dec.link_loss.weight] * 4 if tags is not None else None
        if kcls != 2:
            # classifier depths other than the shipped 2 (reference :253-271): materialising per-document path
            logits, outs, extra = _generic_heads_forward(dec, ab, heads, tags, cws, seeds, need_grad, want_logits)
            saved.update(extra)
            saved.update(ab=ab, B=B, N=N, D=D, seq=seq)
            ctx.dec, ctx.saved, ctx.params = dec, saved, params
            ctx.has_loss = tags is not None
            ctx.mark_non_differentiable(*[lg for lg in logits if lg is not None])
            ctx.set_materialize_grads(False)
            return tuple(outs) + tuple(logits)
        w1s, b1s = [hd[0] for hd in heads], [hd[1] for hd in heads]
        w2s, b2s = [hd[2] for hd in heads], [hd[3] for hd in heads]
        wp = wc.get(("dec.pack", dt), w1s + w2s, lambda: ops.pair_heads_pack(dt, [w.detach() for w in w1s],
                                                                             [w.detach() for w in w2s]))
        b1cat = wc.get(("dec.b1",), b1s, lambda: torch.cat([b.detach() for b in b1s]))
        b2cat = wc.get(("dec.b2",), b2s, lambda: torch.cat([b.detach() for b in b2s]))
        # the Dropout between the two layers of every classifier (reference :261) acts on the [B, P, 5D] hidden inside the
        # kernel; the backward regenerates the mask from (p, seed)
        saved["k12_drop"] = (seeds.p_hidden, seeds.seed(903))
        # a step that will run the fused backward lets the forward leave the classifiers' pre-activations (f16, dropout applied) and
        # x in the backward's block order: what autograd under autocast would have saved of reference :253-271, 2 bytes per pair and
        # hidden unit (4.2 GB at 8 x 511 tokens) - the backward then neither repeats the first-layer product nor the dropout chain
        save = (need_grad and tags is not None and dec.fused_bwd and dec.save_pair_act and ops.pair_bwd_supported(dt, D, len(HEAD_NAMES))
                and ops.pair_save_supported(dt, D, len(HEAD_NAMES)))
        bufs = None
        if save:
            # (step-sized buffers are kept between steps: engine.big_acquire.)  The saved form is an optimisation that costs memory:
            # when its buffers do not fit, the step continues on the recomputing backward, which needs none of them
            try:
                act, act_h = big_acquire("pair_act", (ops.pair_save_bytes(B, N, len(HEAD_NAMES), D),), torch.uint8, dev)
                try:
                    xr, xr_h = big_acquire("pair_x", (B * ops.pair_bwd_rows(N), D), dt, dev)
                except torch.cuda.OutOfMemoryError:
                    big_release("pair_act", act_h)
                    raise
                bufs = (act, xr)
                saved["pair_act"], saved["pair_x"] = (act, act_h), (xr, xr_h)
            except torch.cuda.OutOfMemoryError:
                big_clear(dev)
                save = False
        res = ops.pair_heads_fwd(ab, wp, b1cat, b2cat, HEAD_CLASSES, want_logits=want_logits,
                                 tags=tags, class_weights=cws, want_dlogits=need_grad and tags is not None,
                                 drop_p=seeds.p_hidden, drop_seed=seeds.seed(903), save=save, save_buffers=bufs)
        logits, partials, dlog = res[:3]
        outs = []
        if tags is not None and dec.le_loss.ohem:
            # OHEM: the kept pairs of each head are chosen from its finished logit map; the un-normalised dlogits of the
            # dropped pairs are zeroed in place, so the backward below runs unchanged with scale_h = ratio_h / (k_pos + k_neg)
            ratio = dec.loss_ratio_tensor(dev)
            out8 = torch.empty((len(HEAD_NAMES), 8), dtype=torch.float32, device=dev)
            dls = torch.zeros(sum(HEAD_CLASSES), dtype=torch.float32, device=dev)
            ws, off = None, 0
            for h, c in enumerate(HEAD_CLASSES):
                lossmod = dec.le_loss if h == 0 else dec.link_loss
                _, ws = ops.ohem_ce(logits[h], tags[h], cws[h], lossmod.num_hard_positive, lossmod.num_hard_negative,
                                    dlogits=dlog[h] if dlog is not None else None, out8=out8[h], dl_sum=dls[off:off + c],
                                    workspace=ws)
                off += c
            losses, scale = ops.ohem_finish(out8, ratio)
            outs = [losses[5]] + [losses[i] for i in range(5)]
            saved.update(scale=scale[0], inv_den=scale[1], dlog=dlog, dls=dls)
        elif tags is not None:
            ratio = dec.loss_ratio_tensor(dev)
            losses, scale, dls = ops.loss_finish(partials, ratio, sum(HEAD_CLASSES))
            outs = [losses[5]] + [losses[i] for i in range(5)]
            saved.update(scale=scale[0], inv_den=scale[1], dlog=dlog, dls=dls)
        else:
            outs = [None] * 6
        saved.update(ab=ab, B=B, N=N, D=D, seq=seq)
        ctx.dec, ctx.saved, ctx.params = dec, saved, params
        ctx.has_loss = tags is not None
        if logits is None:
            logits = [None] * 5
        # ONE call: each call replaces ctx.non_differentiable.  The logits carry no gradient (the CE is fused into the
        # kernel that produces them), so a loss built on them outside the model raises instead of training on zeros.
        ctx.mark_non_differentiable(*[lg for lg in logits if lg is not None])
        ctx.set_materialize_grads(False)             # unused loss outputs arrive as None in backward, not as zero tensors
        return tuple(outs) + tuple(logits)

    @staticmethod
    def backward(ctx, d_loss, d_le=None, d_elh=None, d_elt=None, d_lgh=None, d_lgt=None, *unused):
        dec, sv, params = ctx.dec, ctx.saved, ctx.params
        if not ctx.has_loss:
            raise PeneoHipError("backward through the PEneo decoder needs the five *_shaking_tag label maps")
        wc = dec.weight_cache
        it = iter(params)
        if dec.decoder_shrink:
            w0, b0, w3, b3 = next(it), next(it), next(it), next(it)
        wc_w, wc_b = next(it), next(it)
        kcls = dec.num_cls_layers
        heads = [tuple(next(it) for _ in range(2 * kcls)) for _ in HEAD_NAMES]
        B, N, D = sv["B"], sv["N"], sv["D"]
        ab, seeds = sv["ab"], sv["seeds"]
        dt, dev = ab.dtype, ab.device
        nh = len(HEAD_NAMES)
        # dlogits are un-normalised: scale_h = ratio_h / den_h, times the incoming d(loss); the five per-head losses are
        # ordinary differentiable outputs as in the reference (peneo_decoder.py:375-428): their own incoming gradients
        # enter head by head as d(head loss) / den_h = d_head * scale_h / ratio_h
        scale = sv["scale"] * (d_loss.to(torch.float32) if d_loss is not None else 0.0)
        d_heads = [d_le, d_elh, d_elt, d_lgh, d_lgt]
        if any(d is not None for d in d_heads):
            extra = torch.stack([d.to(torch.float32).reshape(()) if d is not None else torch.zeros((), device=dev)
                                 for d in d_heads])
            scale = scale + extra * sv["inv_den"]
        if kcls != 2:
            d_ab, head_grads = _generic_heads_backward(dec, sv, heads, scale)
            return _DecoderStage._front_backward(ctx, dec, sv, params, d_ab, head_grads)
        w1s, b1s = [hd[0] for hd in heads], [hd[1] for hd in heads]
        w2s, b2s = [hd[2] for hd in heads], [hd[3] for hd in heads]
        W1cat = wc.cat_rows("dec.w1cat", w1s, dt)                      # [nh*D, D]
        b1cat = wc.get(("dec.b1",), b1s, lambda: torch.cat([b.detach() for b in b1s]))
        dW1cat = torch.zeros((nh * D, D), dtype=torch.float32, device=dev)
        P = N * (N + 1) // 2
        w2d = [w.detach().contiguous() for w in w2s]
        drop_p, drop_seed = sv["k12_drop"]
        use_fused_bwd = dec.fused_bwd and ops.pair_bwd_supported(dt, D, nh)   # (LDS: the head count enters, D = 512 holds 5 heads)
        # rows the dz producers spread their partial sums over: 256 for the fused kernels / the GEMM epilogue, the full
        # 1024 for the stand-alone peneo_pair_dz (the chunked path with the classifier dropout active)
        dz_ws = ops.pair_dz_workspace(nh, D, dev, slots=256 if (use_fused_bwd or drop_p == 0.0) else None)
        # the fused kernel overwrites d_ab; the chunked path accumulates into it
        d_ab = (torch.empty if use_fused_bwd else torch.zeros)((B, N, 2 * D), dtype=torch.float32, device=dev)
        if use_fused_bwd:
            # ONE kernel for the whole batch: x, z, dz, du = dz W1 and the sums into d_a / d_b never leave the chip except
            # dz and x themselves (block order), which the one remaining GEMM dW1 = dz^T x reads back once
            wp2 = wc.get(("dec.pack2", dt), w1s, lambda: ops.pair_bwd_pack([w.detach() for w in w1s]))
            rows = ops.pair_bwd_rows(N)
            # (a kept buffer may still be read by an earlier decoder backward of THIS autograd run - two forwards summed into one loss -
            # whose side-stream join is deferred to the end of the backward: join first; nothing is pending in the usual step)
            join_pending()
            dzbuf, dz_h = big_acquire("pair_dz", (B * rows, nh * D), dt, dev)
            dza = ops.pair_dz_args(D, HEAD_CLASSES, sv["dlog"], w2d, scale, drop_p=drop_p, drop_seed=drop_seed)
            if "pair_act" in sv:
                xbuf1, x_h = sv.pop("pair_x")
                act, act_h = sv.pop("pair_act")
                ops.pair_bwd_saved(ab, wp2, dza, act, dzbuf, d_ab, dz_ws)
                big_release("pair_act", act_h)           # (read by that launch only: the next forward's stores are ordered behind it)
            else:
                xbuf1, x_h = big_acquire("pair_x", (B * rows, D), dt, dev)
                ops.pair_bwd_fused(ab, wp2, b1cat, dza, dzbuf, xbuf1, d_ab, dz_ws)
            # dz and x are read by the dW1 GEMM below, on the side stream or here: whoever takes them next (the next step's forward /
            # backward, on the main stream) is issued after this backward has ended, i.e. behind the join of that side stream
            big_release("pair_dz", dz_h)
            big_release("pair_x", x_h)
            # dW1 = dz^T x (2 ms of pure MFMA work, needed by nobody until the optimizer) runs on the side stream beside the
            # shrink-MLP backward and the first encoder layers, whose short kernels leave CUs idle; joined one stage later
            if dec.dw1_on_side:
                main = torch.cuda.current_stream()
                side = dec.side_stream(dev)
                ev = torch.cuda.Event()
                ev.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ev)
                    # few, long workgroups: the split a GEMM gets when it runs alone fills all 512 resident slots, and the
                    # main stream's kernels then find no CU to be dispatched to until it ends (measured: the stage after the
                    # decoder stood still for 1.8 ms).  About one workgroup on every second CU runs ~7 ms beside the whole
                    # encoder backward instead and is joined when the backward ends (split 11 / 7 / 5 / 4 / 3: 18.56 /
                    # 18.79 / 18.34 / 18.16 / 18.10 ms per step on one box).
                    # Only when the join can wait for the end of the backward (fresh .grad tensors, no DDP hooks reading them).
                    can_hold = dec.dw1_hold and can_defer(params)
                    # (D = 512: 80 tiles, i.e. ONE workgroup per tile from that target, would run 23 ms - longer than the 24-layer
                    # encoder backward it hides behind; no workgroup gets more than dw1_side_rows rows of K: split 2 / 3 / 4 / 6 at
                    # config 4: 30.5 / 31.1 / 32.8 / 32.7 ms per step)
                    split = None if not can_hold else \
                        dec.dw1_side_split or max(1, dec.dw1_side_wgs // (-(-nh * D // 128) * -(-D // 128)),
                                                  -(-B * rows // dec.dw1_side_rows))
                    with ops.kernel_timer("dw1_gemm"):
                        ops.gemm(dzbuf, xbuf1, a_kmajor=False, b_kmajor=False, out=dW1cat, split_k=split)
                ctx.side_work = (side, (dzbuf, xbuf1, dW1cat, ab), can_hold)
            else:
                with ops.kernel_timer("dw1_gemm"):
                    ops.gemm(dzbuf, xbuf1, a_kmajor=False, b_kmajor=False, out=dW1cat)
            chunks = []
        else:
            chunks = _row_chunks(N, dec.bwd_chunk_pairs)
        if not chunks:
            return _DecoderStage._finish_backward(ctx, dec, sv, params, scale, dW1cat, dz_ws, d_ab, heads, w1s, b1s, w2s, b2s)
        maxp = max((i1 * N - i1 * (i1 - 1) // 2) - (i0 * N - i0 * (i0 - 1) // 2) for i0, i1 in chunks)
        # Two-stage pipeline over the pair chunks on two HIP streams: stage 1 (x, z GEMM whose epilogue turns z into dz:
        # VALU-bound) runs one chunk ahead of stage 2 (dW1 and dx GEMMs + the scatter into d_ab: MFMA-bound), so the two
        # kinds of work share the CUs instead of alternating.  Buffers are double-buffered; events order the hand-offs.
        xbuf = [torch.empty((maxp, D), dtype=dt, device=dev) for _ in range(2)]
        prebuf = [torch.empty((maxp, D), dtype=dt, device=dev) for _ in range(2)]   # a_i + b_j: SiLU' source of the dx GEMM
        zbuf = [torch.empty((maxp, nh * D), dtype=dt, device=dev) for _ in range(2)]
        dxbuf = torch.empty((maxp, D), dtype=dt, device=dev)
        # (with the classifier dropout active the un-fused form runs z = x W1^T + b1 as a plain GEMM and the stand-alone
        # peneo_pair_dz kernel: the GEMM's pair-dz epilogue does not regenerate the mask)
        fused_dz = (dec.fused_dz and dt == torch.bfloat16 and D % 32 == 0
                    and D // 16 in (2, 4, 6, 8, 12, 16, 24, 32))
        if fused_dz:
            wp = wc.get(("dec.pack", dt), w1s + w2s, lambda: ops.pair_heads_pack(dt, [w.detach() for w in w1s],
                                                                                 [w.detach() for w in w2s]))
        main = torch.cuda.current_stream()
        side = main if os.environ.get("PENEO_DEC_STREAMS", "2") == "1" else dec.side_stream(dev)   # "1": strictly serial
        ready = [torch.cuda.Event() for _ in range(2)]
        done = [torch.cuda.Event() for _ in range(2)]
        side.wait_stream(main)                       # d_ab / dW1cat zero fills, ab, weights
        idx = 0
        for b in range(B):
            for (i0, i1) in chunks:
                k = idx & 1
                p0 = i0 * N - i0 * (i0 - 1) // 2
                p1 = i1 * N - i1 * (i1 - 1) // 2
                npairs = p1 - p0
                x, z, dx, pre = xbuf[k][:npairs], zbuf[k][:npairs], dxbuf[:npairs], prebuf[k][:npairs]
                if idx >= 2:
                    main.wait_event(done[k])         # stage 2 of chunk idx-2 has released x[k] / z[k]
                dza = ops.pair_dz_args(D, HEAD_CLASSES, [sv["dlog"][h][b, p0:p1] for h in range(nh)], w2d, scale,
                                       drop_p=drop_p, drop_seed=drop_seed, drop_doc=b, drop_pair0=p0)
                if fused_dz:
                    # dz straight from ab: x lives in registers, z in the MFMA accumulators; x / pre are only needed by
                    # the dW1 / dx GEMMs (measured best split of the two queues: x and dW1 on the main stream)
                    ops.pair_x_fwd(ab[b], i0, i1, x, pre)
                    ops.pair_dz_fused(ab[b], i0, i1, wp, b1cat, dza, z, dz_ws)
                    ops.gemm(z, x, a_kmajor=False, b_kmajor=False, out=dW1cat, accumulate=True)
                else:
                    ops.pair_x_fwd(ab[b], i0, i1, x, pre)
                    if drop_p > 0.0:
                        ops.gemm(x, W1cat, bias=b1cat, out=z)
                        ops.pair_dz(z, npairs, D, HEAD_CLASSES, None, None, dz_ws, None, args=dza)
                    else:
                        # z = x W1^T + b1 and, in the same kernel's epilogue, z -> dz plus the dW2 / db1 partial sums
                        ops.gemm(x, W1cat, bias=b1cat, out=z, pair_dz=dza, pair_dz_ws=dz_ws)
                ready[k].record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ready[k])
                    if idx > 0:
                        side.wait_event(done_x)      # dxbuf is single-buffered: the previous chunk's scatter has read it
                    if not fused_dz:
                        ops.gemm(z, x, a_kmajor=False, b_kmajor=False, out=dW1cat, accumulate=True)
                    # du = (dz W1) * SiLU'(a_i + b_j) in the GEMM epilogue, then plain segmented sums into d_a / d_b
                    ops.gemm(z, W1cat, b_kmajor=False, out=dx, grad_src=pre, grad_act=ACT_SILU)
                    ops.pair_x_bwd(ab[b], i0, i1, dx, d_ab[b], premultiplied=True)
                    done[k].record(side)
                    done_x = done[k]
                idx += 1
        main.wait_stream(side)
        return _DecoderStage._finish_backward(ctx, dec, sv, params, scale, dW1cat, dz_ws, d_ab, heads, w1s, b1s, w2s, b2s)

    @staticmethod
    def _finish_backward(ctx, dec, sv, params, scale, dW1cat, dz_ws, d_ab, heads, w1s, b1s, w2s, b2s):
        """Second half of the two-layer backward: parameter gradients of the heads from the accumulated sums."""
        D = sv["D"]
        nh = len(HEAD_NAMES)
        dw2, db1cat = ops.pair_dz_finish(dz_ws, nh, D, HEAD_CLASSES)
        # db2 of head h = its dlogit sums x scale_h: one multiply for all heads (was one tiny kernel per head on the main stream)
        hidx = _HEAD_OF_CLASS_SLOT.get(scale.device)
        if hidx is None:
            hidx = _HEAD_OF_CLASS_SLOT[scale.device] = torch.tensor([h for h, c in enumerate(HEAD_CLASSES) for _ in range(c)],
                                                                    device=scale.device)
        db2cat = sv["dls"] * scale[hidx]
        head_grads = []
        off = 0
        for h in range(nh):
            c = HEAD_CLASSES[h]
            head_grads += [dW1cat[h * D:(h + 1) * D], db1cat[h * D:(h + 1) * D], dw2[h], db2cat[off:off + c]]
            off += c
        return _DecoderStage._front_backward(ctx, dec, sv, params, d_ab, head_grads, w1s)

    @staticmethod
    def _front_backward(ctx, dec, sv, params, d_ab, head_grads, w1s=()):
        """Back through the [a | b] projection and the shrink MLP; `head_grads`: the heads' parameter gradients in
        stage_params() order."""
        wc = dec.weight_cache
        B, N, D = sv["B"], sv["N"], sv["D"]
        ab, seeds = sv["ab"], sv["seeds"]
        dt, dev = ab.dtype, ab.device
        it = iter(params)
        if dec.decoder_shrink:
            w0, b0, w3, b3 = next(it), next(it), next(it), next(it)
        wc_w, wc_b = next(it), next(it)
        d_ab2 = ops.cast(d_ab.view(B * N, 2 * D), dt) if dt != torch.float32 else d_ab.view(B * N, 2 * D)
        s2 = sv["s2"]
        Wab = dec.stacked_combine_weight(wc_w, dt)
        dWab = ops.gemm(d_ab2, s2, a_kmajor=False, b_kmajor=False, out_dtype=torch.float32)     # [2D, D]
        d_wc = torch.cat([dWab[:D], dWab[D:]], dim=1)
        d_bc = ops.colsum(d_ab2[:, D:])
        grads = []
        if dec.decoder_shrink:
            W0, W3 = wc.cast("dec.s0", w0, dt), wc.cast("dec.s3", w3, dt)
            d_z2 = ops.gemm(d_ab2, Wab, b_kmajor=False, grad_src=sv["z2"], grad_act=ACT_SILU,
                            drop_p=seeds.p_hidden, drop_seed=seeds.seed(902))
            dw3 = ops.gemm(d_z2, sv["s1"], a_kmajor=False, b_kmajor=False, out_dtype=torch.float32)
            db3 = ops.colsum(d_z2)
            d_z1 = ops.gemm(d_z2, W3, b_kmajor=False, grad_src=sv["z1"], grad_act=ACT_SILU,
                            drop_p=seeds.p_hidden, drop_seed=seeds.seed(901))
            dw0 = ops.gemm(d_z1, sv["seq"], a_kmajor=False, b_kmajor=False, out_dtype=torch.float32)
            db0 = ops.colsum(d_z1)
            d_seq = ops.gemm(d_z1, W0, b_kmajor=False)
            grads += [dw0, db0, dw3, db3]
        else:
            d_seq = ops.gemm(d_ab2, Wab, b_kmajor=False)
        grads += [d_wc, d_bc]
        grads += list(head_grads)
        grads = tuple(g if p.requires_grad else None for g, p in zip(grads, params))
        side_work = getattr(ctx, "side_work", None)
        if side_work is not None:
            side, keep, can_hold = side_work
            ctx.side_work = None
            if can_defer(params):
                if can_hold:   # the data-parallel wrapper lays these out last: their data is complete only at the end of the backward
                    mark_late(w1s)
                defer_join(side, keep=keep, hold=can_hold)
            else:
                torch.cuda.current_stream().wait_stream(side)
        return (None, d_seq, None, None, None, None, None) + grads


def _generic_heads_forward(dec, ab, heads, tags, cws, seeds, need_grad, want_logits):
    """Classifier heads of any depth over the materialised pair activations, one document at a time: x = SiLU(a_i + b_j)
    [P, D] (peneo_pair_x_fwd), per head (Linear + SiLU + Dropout) x (k - 1) as GEMMs with fused epilogues, the D -> C layer as
    a GEMM into fp32 logits, class-weighted CE by peneo_weighted_ce (reference model/peneo_decoder.py:253-271,315-336).
    Nothing but the un-normalised dlogits is kept for the backward (it recomputes the activations document by document)."""
    B, N, D2 = ab.shape
    D = D2 // 2
    dt, dev = ab.dtype, ab.device
    wc = dec.weight_cache
    nh, k = len(HEAD_NAMES), dec.num_cls_layers
    P = N * (N + 1) // 2
    logits = [torch.empty((B, P, c), dtype=torch.float32, device=dev) for c in HEAD_CLASSES]
    num = torch.zeros(nh, dtype=torch.float32, device=dev)
    den = torch.zeros(nh, dtype=torch.float32, device=dev)
    dlog = [torch.empty((B, P, c), dtype=torch.float32, device=dev) for c in HEAD_CLASSES] if (need_grad and tags is not None) else None
    x = torch.empty((P, D), dtype=dt, device=dev)
    for b in range(B):
        ops.pair_x_fwd(ab[b], 0, N, x)
        for h in range(nh):
            cur = x
            for l in range(k - 1):
                W = wc.cast(f"dec.h{h}.{l}", heads[h][2 * l], dt)
                cur = ops.gemm(cur, W, bias=heads[h][2 * l + 1], act=ACT_SILU, drop_p=seeds.p_hidden,
                               drop_seed=seeds.seed(1000 + ((b * nh + h) * 8 + l)))
            Wl = wc.cast(f"dec.h{h}.{k - 1}", heads[h][2 * (k - 1)], dt)
            ops.gemm(cur, Wl, bias=heads[h][2 * (k - 1) + 1], out=logits[h][b])
            if tags is not None:
                n_, d_, dl_ = ops.weighted_ce(logits[h][b], tags[h][b], cws[h], want_dlogits=dlog is not None)
                num[h] += n_[0]
                den[h] += d_[0]
                if dlog is not None:
                    dlog[h][b].copy_(dl_)
    outs = [None] * 6
    extra = {}
    if tags is not None:
        ratio = dec.loss_ratio_tensor(dev)
        losses = num / den
        outs = [(losses * ratio).sum()] + [losses[i] for i in range(nh)]
        extra = dict(scale=ratio / den, inv_den=1.0 / den, dlog=dlog)
    return logits, outs, extra


def _generic_heads_backward(dec, sv, heads, scale):
    """Backward of _generic_heads_forward: per document and head the hidden activations are recomputed, then
    dlogits -> (dW, db) of every layer and dx, summed over the heads; d_ab = pair_x_bwd(dx).  -> (d_ab, head gradients in
    stage_params() order)."""
    ab, seeds = sv["ab"], sv["seeds"]
    B, N, D2 = ab.shape
    D = D2 // 2
    dt, dev = ab.dtype, ab.device
    wc = dec.weight_cache
    nh, k = len(HEAD_NAMES), dec.num_cls_layers
    P = N * (N + 1) // 2
    dlog = sv["dlog"]
    gW = [[torch.zeros(heads[h][2 * l].shape, dtype=torch.float32, device=dev) for l in range(k)] for h in range(nh)]
    gb = [[torch.zeros(heads[h][2 * l + 1].shape, dtype=torch.float32, device=dev) for l in range(k)] for h in range(nh)]
    d_ab = torch.zeros((B, N, D2), dtype=torch.float32, device=dev)
    x = torch.empty((P, D), dtype=dt, device=dev)
    dx32 = torch.empty((P, D), dtype=torch.float32, device=dev)
    for b in range(B):
        ops.pair_x_fwd(ab[b], 0, N, x)
        dx32.zero_()
        for h in range(nh):
            Ws = [wc.cast(f"dec.h{h}.{l}", heads[h][2 * l], dt) for l in range(k)]
            acts, pres = [x], []
            for l in range(k - 1):
                z = torch.empty((P, D), dtype=dt, device=dev)
                acts.append(ops.gemm(acts[-1], Ws[l], bias=heads[h][2 * l + 1], act=ACT_SILU, preact=z, drop_p=seeds.p_hidden,
                                     drop_seed=seeds.seed(1000 + ((b * nh + h) * 8 + l))))
                pres.append(z)
            g = (dlog[h][b] * scale[h]).to(dt)                                   # [P, C]: d loss / d logits of this document
            ops.gemm(g, acts[k - 1], a_kmajor=False, b_kmajor=False, out=gW[h][k - 1], accumulate=True)
            ops.colsum(g, out=gb[h][k - 1], accumulate=True)
            for l in range(k - 2, -1, -1):
                up = g if l == k - 2 else dz
                dz = ops.gemm(up, Ws[l + 1], b_kmajor=False, grad_src=pres[l], grad_act=ACT_SILU, drop_p=seeds.p_hidden,
                              drop_seed=seeds.seed(1000 + ((b * nh + h) * 8 + l)))
                ops.gemm(dz, acts[l], a_kmajor=False, b_kmajor=False, out=gW[h][l], accumulate=True)
                ops.colsum(dz, out=gb[h][l], accumulate=True)
            ops.gemm(g if k == 1 else dz, Ws[0], b_kmajor=False, out=dx32, accumulate=True)
        ops.pair_x_bwd(ab[b], 0, N, ops.cast(dx32, dt) if dt != torch.float32 else dx32, d_ab[b])
    head_grads = []
    for h in range(nh):
        for l in range(k):
            head_grads += [gW[h][l], gb[h][l]]
    return d_ab, head_grads


class PEneoDecoder(nn.Module):
    """PEneo pair extraction downstream head (reference :201-443)."""

    def train(self, mode: bool = True):
        """Leaving training mode hands the step-sized buffers the saved-activation backward keeps between steps (engine.big_acquire:
        9.2 GB at 8 x 511 tokens) back to the allocator: inference needs none of them, the next training step takes them again."""
        if not mode:
            big_clear()
        return super().train(mode)

    def side_stream(self, device, which: int = 0) -> "torch.cuda.Stream":
        """Extra HIP streams of the chunked backward (created once per device)."""
        from .engine import side_stream
        return side_stream(device, "dec" if which == 0 else f"dec{which}")

    def __init__(self, config, input_size: int) -> None:
        super().__init__()
        self.decoder_shrink = config.peneo_decoder_shrink
        hidden = config.backbone_config["hidden_size"]
        self.dropout_p = config.backbone_config["hidden_dropout_prob"]
        self.num_cls_layers = int(config.peneo_classifier_num_layers)
        if self.num_cls_layers < 1:
            raise ValueError("peneo_classifier_num_layers must be >= 1")
        if self.decoder_shrink:
            D = hidden // 2
            self.shrink_projection = nn.Sequential(
                nn.Linear(input_size, hidden), nn.SiLU(), nn.Dropout(self.dropout_p),
                nn.Linear(hidden, D), nn.SiLU(), nn.Dropout(self.dropout_p))
        else:
            D = input_size
        if D % 32 != 0:
            raise ValueError(f"decoder hidden size {D} must be a multiple of 32 for the MFMA pair-heads kernel")
        self.decoder_hidden_size = D
        self.handshaking_kernel = HandshakingKernel(D)
        self.inference_mode = config.inference_mode

        def build_classifier(out_size: int) -> nn.Module:
            """Reference :231-271: one Linear for a single layer, else (Linear, SiLU, Dropout) x (k - 1) + Linear; the shipped
            k = 2 runs in the fused pair kernels, every other depth in the materialising per-document path below."""
            if self.num_cls_layers == 1:
                return nn.Linear(D, out_size)
            mods = []
            for _ in range(self.num_cls_layers - 1):
                mods += [nn.Linear(D, D), nn.SiLU(), nn.Dropout(self.dropout_p)]
            mods.append(nn.Linear(D, out_size))
            return nn.Sequential(*mods)

        self.line_extraction_fc = build_classifier(2)
        self.ent_linking_h2h_fc = build_classifier(3)
        self.ent_linking_t2t_fc = build_classifier(3)
        self.line_grouping_h2h_fc = build_classifier(3)
        self.line_grouping_t2t_fc = build_classifier(3)

        self.loss_ratio = config.peneo_loss_ratio
        if self.loss_ratio is not None:
            assert len(self.loss_ratio) == 5, "loss_ratio must be a list of 5 elements"
        cw = config.peneo_category_weights
        assert cw is not None and len(cw) == 3, "category_weights must be a list of 3 elements"
        self.link_loss = _ClassWeightedCE(torch.tensor(cw).float(), config.peneo_ohem_num_positive,
                                          config.peneo_ohem_num_negative)
        self.le_loss = _ClassWeightedCE(torch.tensor(cw[:-1]).float(), config.peneo_ohem_num_positive,
                                        config.peneo_ohem_num_negative)
        self.weight_cache = WeightCache()
        self.train_logits = False    # True: PEneoOutput.*_shaking_outputs are also written in training steps
        # pairs per backward chunk: the z / dz buffer is chunk x 5D (a whole base document is 0.5 GB in bf16, small against
        # 288 GB of HBM), and long launches amortise tile tails and the split-k reduction of the weight-gradient GEMM
        self.bwd_chunk_pairs = int(os.environ.get("PENEO_BWD_CHUNK_PAIRS", 1 << 18))
        self.fused_dz = os.environ.get("PENEO_DZ_FUSED", "1") != "0"            # bf16: dz without x / z in memory
        self.fused_bwd = os.environ.get("PENEO_BWD_FUSED", "1") != "0"          # bf16: the whole pair-space backward in one kernel
        self.save_pair_act = os.environ.get("PENEO_PAIR_SAVE", "1") != "0"      # D = 384: the forward saves the classifiers' pre-activations
        self.dw1_on_side = os.environ.get("PENEO_DW1_SIDE", "1") != "0"          # its dW1 GEMM beside the following stages
        self.dw1_hold = os.environ.get("PENEO_DW1_HOLD", "1") != "0"             # ... joined at the end of the backward only
        self.dw1_side_split = int(os.environ.get("PENEO_DW1_SPLIT", "0"))   # fixed split-k of that GEMM; 0: from the targets below
        self.dw1_side_rows = 600_000     # ... none of them with more rows of K than this
        self.dw1_side_wgs = 144          # ... of about this many long workgroups (measured 18.10 ms per step against 18.56 at 495)
        self._ratio = {}

    def stacked_combine_weight(self, wc_w: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
        """[Wc[:, :D]; Wc[:, D:]] as a [2D, D] working-precision matrix, so one GEMM yields [a | b]."""
        D = wc_w.shape[0]
        return self.weight_cache.get(("dec.ab", dt), [wc_w], lambda: ops.cast(
            torch.cat([wc_w.detach()[:, :D], wc_w.detach()[:, D:]], dim=0).contiguous(), dt))

    def loss_ratio_tensor(self, dev) -> torch.Tensor:
        key = str(dev)
        if key not in self._ratio:
            r = self.loss_ratio if self.loss_ratio is not None else [1.0] * 5
            self._ratio[key] = torch.tensor(r, dtype=torch.float32, device=dev)
        return self._ratio[key]

    def stage_params(self) -> List[torch.Tensor]:
        ps: List[torch.Tensor] = []
        if self.decoder_shrink:
            ps += [self.shrink_projection[0].weight, self.shrink_projection[0].bias,
                   self.shrink_projection[3].weight, self.shrink_projection[3].bias]
        ps += [self.handshaking_kernel.combine_fc.weight, self.handshaking_kernel.combine_fc.bias]
        for name in HEAD_NAMES:
            for lin in self.head_linears(name):
                ps += [lin.weight, lin.bias]
        return ps

    def head_linears(self, name: str) -> List[nn.Linear]:
        fc = getattr(self, name + "_fc")
        return [fc] if isinstance(fc, nn.Linear) else [m for m in fc if isinstance(m, nn.Linear)]

    def forward(self, sequence_output: torch.Tensor, orig_bbox: torch.Tensor = None, line_extraction_shaking_tag=None,
                ent_linking_head_rel_shaking_tag=None, ent_linking_tail_rel_shaking_tag=None,
                line_grouping_head_rel_shaking_tag=None, line_grouping_tail_rel_shaking_tag=None, **kwargs):
        """``sequence_output``: [B, N, Hin] in the working dtype, already cropped (CLS / visual tokens removed)."""
        B, N, Hin = sequence_output.shape
        tags = [line_extraction_shaking_tag, ent_linking_head_rel_shaking_tag, ent_linking_tail_rel_shaking_tag,
                line_grouping_head_rel_shaking_tag, line_grouping_tail_rel_shaking_tag]
        P = N * (N + 1) // 2
        if all(t is None for t in tags) and not self.inference_mode and "line_extraction_matrix_spots" in kwargs:
            # sparse labels of DataCollatorForPEneo(sparse_tags=True): (b, i, j, tag) rows, scattered on the device
            assert kwargs.get("shaking_seq_len", N) == N, "spots were collated for another padded length"
            tags = [ops.spots_to_tags(kwargs[f"{k}_matrix_spots"], N, sequence_output.device, B=B)
                    for k in ("line_extraction", "ent_linking_head_rel", "ent_linking_tail_rel", "line_grouping_head_rel",
                              "line_grouping_tail_rel")]
        if all(t is not None for t in tags):
            for t in tags:
                assert t.shape == (B, P), f"label map shape {tuple(t.shape)} != {(B, P)}"
            tags = [t.contiguous() for t in tags]
        elif self.inference_mode:
            tags = None
        else:
            # the reference asserts on `pred.shape[:-1] == target.shape` with target None -> AttributeError
            raise AssertionError("the five *_shaking_tag label maps are required unless config.inference_mode is set")
        if self.inference_mode:
            tags = None
        params = self.stage_params()
        # autograd is off inside Function.forward, so decide here whether the backward will need dlogits
        need_grad = torch.is_grad_enabled() and (sequence_output.requires_grad or any(p.requires_grad for p in params))
        # The logit maps of a TRAINING step (7.3 MB per document in fp32) are read by nobody: the reference's trainer takes
        # outputs["loss"] and decodes predictions under model.eval() only (pipeline/trainer.py:116-156).  They stay in the
        # kernel's registers unless `train_logits` asks for them; eval / inference always returns them.
        want_logits = bool(self.train_logits) or not (self.training and need_grad and tags is not None) \
            or (tags is not None and self.le_loss.ohem)
        res = _DecoderStage.apply(self, sequence_output.reshape(B * N, Hin), B, N, tags, want_logits, need_grad, *params)
        loss, l_le, l_elh, l_elt, l_lgh, l_lgt, o_le, o_elh, o_elt, o_lgh, o_lgt = res
        if self.inference_mode:
            # NB the reference's tuple order differs from the dataclass field order (:365-373)
            return (o_le, o_elh, o_elt, o_lgh, o_lgt, orig_bbox)
        return PEneoOutput(
            loss=loss, line_extraction_loss=l_le, ent_linking_h2h_loss=l_elh, ent_linking_t2t_loss=l_elt,
            line_grouping_h2h_loss=l_lgh, line_grouping_t2t_loss=l_lgt, line_extraction_shaking_outputs=o_le,
            ent_linking_h2h_shaking_outputs=o_elh, ent_linking_t2t_shaking_outputs=o_elt,
            line_grouping_h2h_shaking_outputs=o_lgh, line_grouping_t2t_shaking_outputs=o_lgt, orig_bbox=orig_bbox)
