"""Functional wrappers: torch tensors in, libpeneo_hip.so kernels on the current stream, tensors out.

These are 1:1 with the C entry points of include/peneo_hip.h (no autograd here; the
autograd.Functions in peneo_amd/model compose them).  Nothing falls back to PyTorch math.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import hip
from .hip import ACT_GELU, ACT_NONE, ACT_SILU, BF16, F32, check, dtype_code, lib, ptr, stream

_i64p = C.POINTER(C.c_int64)


class KernelTimer:
    """Optional HIP-event timing of individual launches on the current stream (bench.py's roofline leg).
    ``with ops.kernel_timer("name"):`` brackets a launch with two events; ``durations_ms()`` after a
    device synchronise gives the per-launch times."""

    def __init__(self) -> None:
        self.enabled = False
        self.events = {}

    def reset(self, enabled: bool) -> None:
        self.enabled = enabled
        self.events = {}

    def durations_ms(self, name: str):
        return [a.elapsed_time(b) for a, b in self.events.get(name, [])]


TIMER = KernelTimer()


class kernel_timer:
    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if TIMER.enabled:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if TIMER.enabled:
            self.b.record()
            TIMER.events.setdefault(self.name, []).append((self.a, self.b))
        return False


def _c(t: torch.Tensor) -> torch.Tensor:
    if not t.is_contiguous():
        raise hip.PeneoHipError("tensor must be contiguous")
    return t


# ----------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------
def _fill_epilogue(ep, out: torch.Tensor, *, bias=None, act=ACT_NONE, residual=None, preact=None, grad_src=None, grad_act=ACT_NONE,
                   alpha: float = 1.0, accumulate: bool = False, drop_p: float = 0.0, drop_seed: int = 0) -> None:
    ep.bias = ptr(bias)
    ep.act = act
    if preact is not None:
        assert preact.dtype == out.dtype and preact.shape == out.shape
        ep.preact, ep.ld_preact = ptr(preact), preact.stride(0)
    if grad_src is not None:
        assert grad_src.dtype == out.dtype and grad_src.shape == out.shape
        ep.grad_src, ep.ld_grad, ep.grad_act = ptr(grad_src), grad_src.stride(0), grad_act
    if residual is not None:
        assert residual.dtype == out.dtype and residual.shape == out.shape
        ep.residual, ep.ld_res = ptr(residual), residual.stride(0)
    ep.alpha = alpha
    ep.accumulate = 1 if accumulate else 0
    ep.drop_p, ep.drop_seed = drop_p, drop_seed & 0xFFFFFFFF


def gemm_group(problems: Sequence, *, a_kmajor: bool = True, b_kmajor: bool = True, accumulate: bool = False,
               out_dtype: Optional[torch.dtype] = None) -> List[torch.Tensor]:
    """Up to 4 independent bf16 GEMMs out_i = epilogue_i(op(A_i) op(B_i)) (one operand layout) in ONE launch, each tile with its
    full K.  A problem is (A, B, out) or (A, B, out, epilogue-dict) with out = None to allocate [M, N] of `out_dtype` (default
    bf16; all outputs share the dtype) and the epilogue keys of `gemm` (bias, act, residual, preact, grad_src, grad_act, drop_p,
    drop_seed, alpha).  Returns the outputs."""
    n = len(problems)
    arr = (hip.GemmProblem * n)()
    eps, outs = [], []
    for q, prob in zip(arr, problems):
        A, B, out = prob[0], prob[1], prob[2]
        kw = prob[3] if len(prob) > 3 else None
        M = A.shape[0] if a_kmajor else A.shape[1]
        K = A.shape[1] if a_kmajor else A.shape[0]
        N = B.shape[0] if b_kmajor else B.shape[1]
        if out is None:
            out = torch.empty((M, N), dtype=out_dtype or torch.bfloat16, device=A.device)
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and out.shape == (M, N)
        q.M, q.N, q.K = M, N, K
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc = ptr(A), A.stride(0), ptr(B), B.stride(0), ptr(out), out.stride(0)
        q.accumulate = int(accumulate)
        if kw:
            ep = hip.GemmEpilogue()
            _fill_epilogue(ep, out, **kw)
            eps.append(ep)                            # (alive until the call has returned)
            q.ep = C.cast(C.pointer(ep), C.c_void_p)
        outs.append(out)
    with kernel_timer("gemm_group"):
        check(lib().peneo_gemm_group(dtype_code(torch.bfloat16), int(a_kmajor), int(b_kmajor), dtype_code(outs[0].dtype),
                                     arr, n, stream()), "peneo_gemm_group")
    return outs


def choose_split_k(M: int, N: int, K: int, dtype: torch.dtype) -> int:
    """Split the reduction only when the output grid cannot fill the 256 CUs."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    kt = (K + (63 if dtype == torch.bfloat16 else 31)) // (64 if dtype == torch.bfloat16 else 32)
    if tiles >= 192 or kt < 8:
        return 1
    # 512 = resident workgroups (2 per CU): never spill a few workgroups into a second, almost empty round
    return max(1, min(kt // 4, 512 // tiles, 16))


def gemm(a: torch.Tensor, b: torch.Tensor, *, a_kmajor: bool = True, b_kmajor: bool = True,
         bias: Optional[torch.Tensor] = None, act: int = ACT_NONE, residual: Optional[torch.Tensor] = None,
         preact: Optional[torch.Tensor] = None, grad_src: Optional[torch.Tensor] = None, grad_act: int = ACT_NONE,
         out: Optional[torch.Tensor] = None, out_dtype: Optional[torch.dtype] = None, accumulate: bool = False,
         alpha: float = 1.0, drop_p: float = 0.0, drop_seed: int = 0, split_k: Optional[int] = None,
         pair_dz=None, pair_dz_ws: Optional[torch.Tensor] = None, a_colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = epilogue(alpha * A.B^T); a, b are 2-D (row stride may exceed the row length).
    a_colsum (a_kmajor=False only): fp32 [M] that receives += the column sums of a = [K, M] -- the bias gradient beside a
    weight gradient, from the tiles the product holds in LDS anyway."""
    assert a.dim() == 2 and b.dim() == 2 and a.dtype == b.dtype
    assert a.stride(1) == 1 and b.stride(1) == 1
    if a_kmajor:
        M, K = a.shape
    else:
        K, M = a.shape
    if b_kmajor:
        N, Kb = b.shape
    else:
        Kb, N = b.shape
    assert K == Kb, f"inner dims differ: {K} vs {Kb}"
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or a.dtype, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    ep = hip.GemmEpilogue()
    ep.bias = ptr(bias)
    ep.act = act
    if preact is not None:
        assert preact.dtype == out.dtype and preact.shape == out.shape
        ep.preact, ep.ld_preact = ptr(preact), preact.stride(0)
    if grad_src is not None:
        assert grad_src.dtype == out.dtype and grad_src.shape == out.shape
        ep.grad_src, ep.ld_grad, ep.grad_act = ptr(grad_src), grad_src.stride(0), grad_act
    if residual is not None:
        assert residual.dtype == out.dtype and residual.shape == out.shape
        ep.residual, ep.ld_res = ptr(residual), residual.stride(0)
    ep.alpha = alpha
    ep.accumulate = 1 if accumulate else 0
    ep.drop_p, ep.drop_seed = drop_p, drop_seed & 0xFFFFFFFF
    if a_colsum is not None:
        assert not a_kmajor and a_colsum.dtype == torch.float32 and a_colsum.numel() == M and a_colsum.is_contiguous()
        ep.a_colsum = ptr(a_colsum)
    if pair_dz is not None:      # hip.PairDzArgs: the tile is z of the pair heads; store dz, accumulate dW2 / db1 partials
        ep.pair_dz, ep.pair_dz_ws = C.cast(C.pointer(pair_dz), C.c_void_p), ptr(pair_dz_ws)
        split_k = 1
    if split_k is None:
        split_k = choose_split_k(M, N, K, a.dtype)
    ws, ws_bytes = None, 0
    if split_k > 1:
        ws_bytes = lib().peneo_gemm_workspace_bytes(M, N, K, split_k)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=a.device)
    check(lib().peneo_gemm(dtype_code(a.dtype), int(a_kmajor), int(b_kmajor), M, N, K, ptr(a), a.stride(0), ptr(b),
                           b.stride(0), ptr(out), out.stride(0), dtype_code(out.dtype), C.byref(ep), split_k, ptr(ws),
                           ws_bytes, stream()), "peneo_gemm")
    return out


# ----------------------------------------------------------------------------------------------
# element-wise plumbing
# ----------------------------------------------------------------------------------------------
def cast(src: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _c(src)
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    check(lib().peneo_cast(ptr(src), dtype_code(src.dtype), ptr(out), dtype_code(out.dtype), src.numel(), stream()),
          "peneo_cast")
    return out


class CastPlan:
    """Device tables of a fixed set of (fp32 source, bf16 destination) pairs for ``peneo_cast_multi``: built once, launched
    whenever the sources have changed."""

    def __init__(self, pairs: Sequence[Tuple[torch.Tensor, torch.Tensor]]):
        import numpy as np
        chunk = int(lib().peneo_cast_multi_chunk_elems())
        items = (hip.CastItem * len(pairs))()
        ci, cx = [], []
        for k, (src, dst) in enumerate(pairs):
            assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.is_contiguous() and dst.is_contiguous()
            assert src.numel() == dst.numel()
            items[k].src, items[k].dst, items[k].numel = src.data_ptr(), dst.data_ptr(), src.numel()
            n = (src.numel() + chunk - 1) // chunk
            ci += [k] * n
            cx += list(range(n))
        dev = pairs[0][0].device
        raw = np.frombuffer(bytes(items), dtype=np.uint8).copy()
        self.table = torch.from_numpy(raw).to(dev)
        self.chunk_item = torch.tensor(ci, dtype=torch.int32, device=dev)
        self.chunk_index = torch.tensor(cx, dtype=torch.int32, device=dev)
        self.n_chunks = len(ci)
        self.keep = list(pairs)                     # the tensors whose addresses the table holds

    def run(self) -> None:
        check(lib().peneo_cast_multi(ptr(self.table), ptr(self.chunk_item), ptr(self.chunk_index), self.n_chunks, stream()),
              "peneo_cast_multi")


def copy2d(src: torch.Tensor, out: Optional[torch.Tensor] = None, drop_p: float = 0.0, drop_seed: int = 0) -> torch.Tensor:
    assert src.dim() == 2 and src.stride(1) == 1
    if out is None:
        out = torch.empty(src.shape, dtype=src.dtype, device=src.device)
    assert out.shape == src.shape and out.stride(1) == 1
    check(lib().peneo_copy2d(dtype_code(src.dtype), ptr(src), src.stride(0), ptr(out), out.stride(0), src.shape[0],
                             src.shape[1], drop_p, drop_seed & 0xFFFFFFFF, stream()), "peneo_copy2d")
    return out


def copy_rows(src: torch.Tensor, dst: torch.Tensor, drop_p: float = 0.0, drop_seed: int = 0) -> torch.Tensor:
    """Copy between [B, R, C] views whose last dim is contiguous (either side may be a slice of a larger buffer)."""
    assert src.dim() == 3 and dst.dim() == 3 and src.shape == dst.shape and src.dtype == dst.dtype
    assert src.stride(2) == 1 and dst.stride(2) == 1
    Bn, R, Cn = src.shape
    check(lib().peneo_copy_rows(dtype_code(src.dtype), ptr(src), R, src.stride(0), src.stride(1), ptr(dst), R, dst.stride(0),
                                dst.stride(1), Bn * R, Cn, drop_p, drop_seed & 0xFFFFFFFF, stream()), "peneo_copy_rows")
    return dst


def head_concat(a: torch.Tensor, b: torch.Tensor, nh: int, out: torch.Tensor, scale_a: float = 1.0, scale_b: float = 1.0):
    """out[r, h*(da+db) + c] = [scale_a * a_h | scale_b * b_h]; a, b, out are 2-D views with unit column stride."""
    da, db = a.shape[1] // nh, b.shape[1] // nh
    assert out.shape[1] == nh * (da + db) and a.shape[0] == b.shape[0] == out.shape[0]
    check(lib().peneo_head_concat(dtype_code(a.dtype), ptr(a), a.stride(0), da, scale_a, ptr(b), b.stride(0), db, scale_b,
                                  ptr(out), out.stride(0), a.shape[0], nh, stream()), "peneo_head_concat")
    return out


def head_split(x: torch.Tensor, nh: int, a: torch.Tensor, b: torch.Tensor, scale_a: float = 1.0, scale_b: float = 1.0):
    da, db = a.shape[1] // nh, b.shape[1] // nh
    assert x.shape[1] == nh * (da + db) and a.shape[0] == b.shape[0] == x.shape[0]
    check(lib().peneo_head_split(dtype_code(x.dtype), ptr(x), x.stride(0), ptr(a), a.stride(0), da, scale_a, ptr(b),
                                 b.stride(0), db, scale_b, x.shape[0], nh, stream()), "peneo_head_split")
    return a, b


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    assert x.dim() == 2 and x.stride(1) == 1
    if out is None:
        out = torch.empty(x.shape[1], dtype=torch.float32, device=x.device)
        accumulate = False
    check(lib().peneo_colsum(dtype_code(x.dtype), ptr(x), x.stride(0), x.shape[0], x.shape[1], ptr(out), int(accumulate),
                             stream()), "peneo_colsum")
    return out


# ----------------------------------------------------------------------------------------------
# LayerNorm
# ----------------------------------------------------------------------------------------------
def _rowmap(t: torch.Tensor, H: int) -> Tuple[int, int, int]:
    """(rows, rows_per_batch, batch_stride) for a [rows, H] or a sliced [B, R, H] tensor."""
    assert t.stride(-1) == 1 and t.shape[-1] == H
    if t.dim() == 2:
        assert t.stride(0) == H
        return t.shape[0], 0, 0
    assert t.dim() == 3 and t.stride(1) == H
    return t.shape[0] * t.shape[1], t.shape[1], t.stride(0)


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float, out: Optional[torch.Tensor] = None,
                  drop_p: float = 0.0, drop_seed: int = 0):
    H = x.shape[-1]
    rows, xr, xb = _rowmap(x, H)
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    orows, yr, yb = _rowmap(out, H)
    assert orows == rows and out.dtype == x.dtype
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().peneo_layernorm_fwd(dtype_code(x.dtype), ptr(x), xr, xb, ptr(out), yr, yb, ptr(gamma), ptr(beta), eps,
                                    ptr(mean), ptr(rstd), rows, H, drop_p, drop_seed & 0xFFFFFFFF, stream()),
          "peneo_layernorm_fwd")
    return out, mean, rstd


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                  dgamma: torch.Tensor, dbeta: torch.Tensor, dx: Optional[torch.Tensor] = None, drop_p: float = 0.0,
                  drop_seed: int = 0, dx_dropped: Optional[torch.Tensor] = None, drop2_p: float = 0.0,
                  drop2_seed: int = 0, dx_colsum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dx_colsum (fp32 [H], accumulated into): column sums of dx_dropped (of dx without it) = the bias gradient of the Linear in
    front of this LayerNorm."""
    H = x.shape[-1]
    assert dx_colsum is None or (dx_colsum.dtype == torch.float32 and dx_colsum.numel() == H and dx_colsum.is_contiguous())
    rows, xr, xb = _rowmap(x, H)
    drows, dr, db = _rowmap(dy, H)
    assert drows == rows and dy.dtype == x.dtype
    if dx is None:
        dx = torch.empty(dy.shape, dtype=x.dtype, device=x.device)
    _, gr, gb = _rowmap(dx, H)
    check(lib().peneo_layernorm_bwd(dtype_code(x.dtype), ptr(dy), dr, db, ptr(x), xr, xb, ptr(dx), gr, gb, ptr(gamma),
                                    ptr(mean), ptr(rstd), ptr(dgamma), ptr(dbeta), rows, H, drop_p,
                                    drop_seed & 0xFFFFFFFF, ptr(dx_dropped), drop2_p, drop2_seed & 0xFFFFFFFF, ptr(dx_colsum),
                                    stream()),
          "peneo_layernorm_bwd")
    return dx


def layernorm_bwd_partial(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor, rstd: torch.Tensor,
                          dx: Optional[torch.Tensor] = None, drop_p: float = 0.0, drop_seed: int = 0,
                          dx_dropped: Optional[torch.Tensor] = None, drop2_p: float = 0.0, drop2_seed: int = 0):
    """LayerNorm backward with the parameter gradients left as per-workgroup partial sums: -> (dx, partials [P, 2H] fp32 with
    [dgamma | dbeta] per row) or (None, None) when this dtype / row length has no such form.  ``colsum(partials)`` gives
    [dgamma | dbeta]; the model runs that reduction on its weight-gradient stream."""
    H = x.shape[-1]
    rows, xr, xb = _rowmap(x, H)
    drows, dr, db = _rowmap(dy, H)
    assert drows == rows and dy.dtype == x.dtype
    P = int(lib().peneo_layernorm_bwd_partial_rows(dtype_code(x.dtype), rows, H))
    if P <= 0:
        return None, None
    if dx is None:
        dx = torch.empty(dy.shape, dtype=x.dtype, device=x.device)
    _, gr, gb = _rowmap(dx, H)
    partials = torch.empty((P, 2 * H), dtype=torch.float32, device=x.device)
    check(lib().peneo_layernorm_bwd_partial(dtype_code(x.dtype), ptr(dy), dr, db, ptr(x), xr, xb, ptr(dx), gr, gb, ptr(gamma),
                                            ptr(mean), ptr(rstd), ptr(partials), P, rows, H, drop_p,
                                            drop_seed & 0xFFFFFFFF, ptr(dx_dropped), drop2_p, drop2_seed & 0xFFFFFFFF, stream()),
          "peneo_layernorm_bwd_partial")
    return dx, partials


# ----------------------------------------------------------------------------------------------
# embeddings
# ----------------------------------------------------------------------------------------------
def position_ids(input_ids: torch.Tensor, pad_id: int) -> torch.Tensor:
    _c(input_ids)
    B, S = input_ids.shape
    out = torch.empty((B, S), dtype=torch.int32, device=input_ids.device)
    check(lib().peneo_position_ids(ptr(input_ids), B, S, pad_id, ptr(out), stream()), "peneo_position_ids")
    return out


def embed_fwd(dtype: torch.dtype, out: torch.Tensor, B: int, S: int, H: int, *, input_ids=None, pos_ids=None, bbox=None,
              word=None, type0=None, pos=None, x=None, y=None, h=None, w=None, clip_hw: bool = True,
              status: Optional[torch.Tensor] = None) -> torch.Tensor:
    tab = hip.EmbedTables()
    tab.word, tab.type0, tab.pos = ptr(word), ptr(type0), ptr(pos)
    tab.x, tab.y, tab.h, tab.w = ptr(x), ptr(y), ptr(h), ptr(w)
    if x is not None:
        tab.coord_size, tab.shape_size, tab.max_2d = x.shape[1], h.shape[1], x.shape[0]
    if word is not None:
        tab.vocab, tab.max_pos = word.shape[0], pos.shape[0]
    rows, rpb, bs = _rowmap(out, H)
    assert rows == B * S and out.dtype == dtype
    check(lib().peneo_embed_text_fwd(dtype_code(dtype), ptr(input_ids), ptr(pos_ids), ptr(bbox), C.byref(tab), B, S, H,
                                     int(clip_hw), ptr(out), rpb, bs, ptr(status), stream()), "peneo_embed_text_fwd")
    return out


def embed_bwd(d_out: torch.Tensor, B: int, S: int, H: int, *, input_ids=None, pos_ids=None, bbox=None, g_word=None,
              g_pos=None, g_x=None, g_y=None, g_h=None, g_w=None, clip_hw: bool = True, pad_id: int = 1) -> None:
    g = hip.EmbedGrads()
    g.word, g.pos = ptr(g_word), ptr(g_pos)
    g.x, g.y, g.h, g.w = ptr(g_x), ptr(g_y), ptr(g_h), ptr(g_w)
    cs = g_x.shape[1] if g_x is not None else 0
    ss = g_h.shape[1] if g_h is not None else 0
    m2 = g_x.shape[0] if g_x is not None else 0
    rows, rpb, bs = _rowmap(d_out, H)
    assert rows == B * S
    with kernel_timer("embed_bwd"):
        check(lib().peneo_embed_text_bwd(dtype_code(d_out.dtype), ptr(d_out), rpb, bs, ptr(input_ids), ptr(pos_ids), ptr(bbox),
                                         C.byref(g), cs, ss, m2, B, S, H, int(clip_hw), pad_id, stream()),
              "peneo_embed_text_bwd")


def im2col_patch16(image: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    _c(image)
    assert image.dtype == torch.float32
    B, Cc, Hi, Wi = image.shape
    out = torch.empty((B * (Hi // 16) * (Wi // 16), Cc * 256), dtype=dtype, device=image.device)
    check(lib().peneo_im2col_patch16(dtype_code(dtype), ptr(image), B, Cc, Hi, Wi, ptr(out), stream()),
          "peneo_im2col_patch16")
    return out


def visual_assemble_fwd(proj: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor, B: int) -> torch.Tensor:
    npch, H = proj.shape[0] // B, proj.shape[1]
    vis = torch.empty((B, npch + 1, H), dtype=proj.dtype, device=proj.device)
    check(lib().peneo_visual_assemble_fwd(dtype_code(proj.dtype), ptr(_c(proj)), ptr(cls), ptr(pos), B, npch, H, ptr(vis),
                                          stream()), "peneo_visual_assemble_fwd")
    return vis


def visual_assemble_bwd(d_vis: torch.Tensor, d_cls: Optional[torch.Tensor], d_pos: Optional[torch.Tensor]) -> torch.Tensor:
    B, t, H = d_vis.shape
    d_proj = torch.empty((B * (t - 1), H), dtype=d_vis.dtype, device=d_vis.device)
    check(lib().peneo_visual_assemble_bwd(dtype_code(d_vis.dtype), ptr(_c(d_vis)), B, t - 1, H, ptr(d_proj), ptr(d_cls),
                                          ptr(d_pos), stream()), "peneo_visual_assemble_bwd")
    return d_proj


# ----------------------------------------------------------------------------------------------
# relative-position bias
# ----------------------------------------------------------------------------------------------
def relpos_inputs(attention_mask: Optional[torch.Tensor], bbox: Optional[torch.Tensor], vx: Optional[torch.Tensor],
                  vy: Optional[torch.Tensor], B: int, S: int, nv: int, want_pos: bool, want_xy: bool):
    """-> (key_mask, pos, xs, ys): int32 [B, S + nv] each (pos / xs, ys None when not wanted), one launch."""
    dev = (attention_mask if attention_mask is not None else bbox).device
    mk = lambda: torch.empty((B, S + nv), dtype=torch.int32, device=dev)
    km, pos = mk(), (mk() if want_pos else None)
    xs, ys = (mk(), mk()) if want_xy else (None, None)
    if attention_mask is not None:
        assert attention_mask.dtype == torch.int64 and attention_mask.is_contiguous()
    if want_xy:
        assert bbox.dtype == torch.int64 and bbox.is_contiguous()
    check(lib().peneo_relpos_inputs(ptr(attention_mask), ptr(bbox), ptr(vx), ptr(vy), B, S, nv, ptr(km), ptr(pos), ptr(xs), ptr(ys),
                                    stream()), "peneo_relpos_inputs")
    return km, pos, xs, ys


def relpos_buckets(pos: Optional[torch.Tensor], xs: Optional[torch.Tensor], ys: Optional[torch.Tensor], B: int, T: int,
                   lut1: Optional[torch.Tensor], half1: int, lut2: Optional[torch.Tensor], half2: int):
    dev = (pos if pos is not None else xs).device
    mk = lambda: torch.empty((B, T, T), dtype=torch.uint8, device=dev)
    bk1 = mk() if pos is not None else None
    bkx = mk() if xs is not None else None
    bky = mk() if ys is not None else None
    check(lib().peneo_relpos_buckets(ptr(pos), ptr(xs), ptr(ys), B, T, ptr(lut1), lut1.numel() if lut1 is not None else 0,
                                     half1, ptr(lut2), lut2.numel() if lut2 is not None else 0, half2, ptr(bk1), ptr(bkx),
                                     ptr(bky), stream()), "peneo_relpos_buckets")
    return bk1, bkx, bky


def attn_padded_len(T: int) -> int:
    return (T + 63) // 64 * 64


def attn_padded_dim(d: int) -> int:
    return (d + 31) // 32 * 32


def relpos_bias_fwd(dtype: torch.dtype, bk1, bkx, bky, w1, wx, wy, scale: float, B: int, nh: int, T: int,
                    key_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> bias [B, nh, T, Tp] (Tp = T rounded up to 64); padding columns and masked keys hold -1e30."""
    dev = (bk1 if bk1 is not None else bkx).device
    Tp = attn_padded_len(T)
    bias = torch.empty((B, nh, T, Tp), dtype=dtype, device=dev)
    check(lib().peneo_relpos_bias_fwd(dtype_code(dtype), ptr(bk1), ptr(bkx), ptr(bky), ptr(w1),
                                      w1.shape[1] if w1 is not None else 0, ptr(wx), ptr(wy),
                                      wx.shape[1] if wx is not None else 0, scale, B, nh, T, Tp, ptr(key_mask), ptr(bias),
                                      stream()), "peneo_relpos_bias_fwd")
    return bias


def relpos_bias_bwd_layers(ds: torch.Tensor, bk1_t, bkx_t, bky_t, dw1, dwx, dwy, scale: float) -> None:
    """ds: bf16 [L, B, nh, T, Tp] per-layer dS^T (attn_bwd's ds_out); bk*_t: transposed bucket maps, [B, T, T] (padded
    here to the kernel's row stride Tp) or already [B, T, Tp]."""
    L, B, nh, T, Tp = ds.shape

    def padded(t):
        if t is None or t.shape[-1] == Tp:
            return t
        out = torch.zeros((B, T, Tp), dtype=torch.uint8, device=t.device)
        out[:, :, :T] = t
        return out
    bk1_t, bkx_t, bky_t = padded(bk1_t), padded(bkx_t), padded(bky_t)
    check(lib().peneo_relpos_bias_bwd_layers(ptr(_c(ds)), L, ds.stride(0), ptr(bk1_t), ptr(bkx_t), ptr(bky_t), ptr(dw1),
                                             dw1.shape[1] if dw1 is not None else 0, ptr(dwx), ptr(dwy),
                                             dwx.shape[1] if dwx is not None else 0, scale, B, nh, T, Tp, stream()),
          "peneo_relpos_bias_bwd_layers")


def relpos_bias_bwd(g: torch.Tensor, bk1, bkx, bky, dw1, dwx, dwy, scale: float) -> None:
    B, nh, T, ldg = g.shape
    check(lib().peneo_relpos_bias_bwd(ptr(_c(g)), ldg, ptr(bk1), ptr(bkx), ptr(bky), ptr(dw1),
                                      dw1.shape[1] if dw1 is not None else 0, ptr(dwx), ptr(dwy),
                                      dwx.shape[1] if dwx is not None else 0, scale, B, nh, T, stream()),
          "peneo_relpos_bias_bwd")


# ----------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------
def head_transpose(x: torch.Tensor, B: int, nh: int, T: int, d: int) -> torch.Tensor:
    """x: 2-D view [B*T, nh*d] (row stride may be larger) -> [B, nh, DP, Tp] per-head transposed, zero padded."""
    assert x.dim() == 2 and x.stride(1) == 1
    out = torch.empty((B, nh, attn_padded_dim(d), attn_padded_len(T)), dtype=x.dtype, device=x.device)
    check(lib().peneo_head_transpose(dtype_code(x.dtype), ptr(x), x.stride(0), B, nh, T, d, ptr(out), stream()),
          "peneo_head_transpose")
    return out


def attn_drop_words_shape(B: int, nh: int, T: int, sets: int = 1):
    nqb, tk = C.c_int(0), C.c_int(0)
    lib().peneo_attn_drop_words_dims(T, C.byref(nqb), C.byref(tk))
    return (sets, B * nh, nqb.value, tk.value)


def attn_drop_words(B: int, nh: int, T: int, drop_p: float, drop_seed: int, device=None, sets: int = 1,
                    out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Keep bits of the attention dropout (peneo_attn_drop_words): int32 [sets, B * nh, query blocks, key slots], one launch for
    `sets` independent calls of the same shape (e.g. all layers of a step); hand set i to attn_fwd / attn_bwd of call i.
    `out`: a buffer of attn_drop_words_shape(...) allocated by the caller (e.g. on another stream than the one that fills it)."""
    shape = attn_drop_words_shape(B, nh, T, sets)
    words = out if out is not None else torch.empty(shape, dtype=torch.int32, device=device or torch.device("cuda"))
    assert tuple(words.shape) == shape and words.dtype == torch.int32 and words.is_contiguous()
    check(lib().peneo_attn_drop_words(ptr(words), sets * B, nh, T, drop_p, drop_seed & 0xFFFFFFFF, stream()), "peneo_attn_drop_words")
    return words


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, B: int, nh: int, T: int, d: int, scale: float,
             bias: Optional[torch.Tensor], key_bias: Optional[torch.Tensor] = None, drop_p: float = 0.0, drop_seed: int = 0,
             vt: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, lse: Optional[torch.Tensor] = None,
             drop_words: Optional[torch.Tensor] = None):
    """q/k/v: 2-D views [B*T, nh*d] with a common row stride (e.g. slices of a fused QKV buffer).
    bias: [B, nh, T, Tp] from relpos_bias_fwd (masking folded in); key_bias: fp32 [B, Tp] additive (0 / -1e30).
    out [B*T, nh*d] / lse [B, nh, T] may be given (e.g. row slices of larger buffers).
    Dropout: `drop_words` = one set of attn_drop_words (the backward of this call takes the same set); without it the words
    are made here from (drop_p, drop_seed) - attn_bwd with the same pair regenerates the same bits."""
    if drop_p > 0 and drop_words is None:
        drop_words = attn_drop_words(B, nh, T, drop_p, drop_seed, q.device)
    assert q.stride(0) == k.stride(0) == v.stride(0) and q.stride(1) == 1
    if vt is None and q.dtype != torch.bfloat16:   # bf16 reads V in place (transpose reads); fp32 needs the transposed copy
        vt = head_transpose(v, B, nh, T, d)
    if out is None:
        out = torch.empty((B * T, nh * d), dtype=q.dtype, device=q.device)
    if lse is None:
        lse = torch.empty((B, nh, T), dtype=torch.float32, device=q.device)
    assert out.shape == (B * T, nh * d) and out.stride(1) == 1 and lse.is_contiguous() and lse.shape == (B, nh, T)
    check(lib().peneo_attn_fwd(dtype_code(q.dtype), ptr(q), ptr(k), ptr(v), q.stride(0), ptr(vt), B, nh, T, d, scale, ptr(bias),
                               bias.shape[-1] if bias is not None else 0, ptr(key_bias), ptr(out), out.stride(0), ptr(lse),
                               drop_p, ptr(drop_words) if drop_p > 0 else None, stream()), "peneo_attn_fwd")
    return out, lse


def attn_bwd(q, k, v, out, d_out, lse, B: int, nh: int, T: int, d: int, scale: float, bias, key_bias,
             dqkv: torch.Tensor, g_bias: Optional[torch.Tensor], drop_p: float = 0.0, drop_seed: int = 0,
             single_pass: Optional[bool] = None, ds_out: Optional[torch.Tensor] = None,
             dq_atomic: bool = False, drop_words: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dqkv: [B*T, 3*nh*d] buffer receiving dq | dk | dv (same layout as the fused QKV activations).
    bf16 runs the single-pass kernel (fp32 dQ accumulator, no transposed copies); fp32 the dQ + dK/dV pair."""
    H = nh * d
    dq, dk, dv = dqkv[:, :H], dqkv[:, H:2 * H], dqkv[:, 2 * H:]
    delta = torch.empty((B, nh, T), dtype=torch.float32, device=q.device)
    if drop_p > 0 and drop_words is None:
        drop_words = attn_drop_words(B, nh, T, drop_p, drop_seed, q.device)
    assert out.stride(0) == d_out.stride(0)
    if single_pass is None:
        single_pass = (q.dtype == torch.bfloat16 and H % 8 == 0 and dqkv.stride(0) % 8 == 0
                       and dqkv.data_ptr() % 16 == 0)
    kt = qt = dot = dq_acc = None
    if single_pass:
        assert q.dtype == torch.bfloat16, "the single-pass attention backward is bf16 only"
        if dq_atomic:      # dQ through fp32 atomics into an accumulator (kept for comparison: a quarter of the kernel's time)
            dq_acc = torch.empty((B * T, H), dtype=torch.float32, device=q.device)
        elif ds_out is None:   # dQ by a second kernel from the layer's dS^T: needs the slab even when nobody else wants it
            ds_out = torch.empty((B, nh, T, attn_padded_len(T)), dtype=q.dtype, device=q.device)
    else:
        kt = head_transpose(k, B, nh, T, d)
        qt = head_transpose(q, B, nh, T, d)
        dot = head_transpose(d_out, B, nh, T, d)
    check(lib().peneo_attn_bwd(dtype_code(q.dtype), ptr(q), ptr(k), ptr(v), q.stride(0), ptr(kt), ptr(qt), ptr(dot),
                               ptr(out), ptr(d_out), out.stride(0), ptr(lse), B, nh, T, d, scale, ptr(bias),
                               bias.shape[-1] if bias is not None else 0, ptr(key_bias), ptr(dq), ptr(dk), ptr(dv),
                               dqkv.stride(0), ptr(g_bias), ptr(delta), ptr(dq_acc), ptr(ds_out), drop_p,
                               ptr(drop_words) if drop_p > 0 else None, stream()), "peneo_attn_bwd")
    return dqkv


# ----------------------------------------------------------------------------------------------
# pair heads
# ----------------------------------------------------------------------------------------------
def _ptr_list(ts: Sequence[Optional[torch.Tensor]]):
    arr = (C.c_void_p * len(ts))()
    for i, t in enumerate(ts):
        arr[i] = ptr(t)
    return arr


def pair_heads_pack(dtype: torch.dtype, w1: Sequence[torch.Tensor], w2: Sequence[torch.Tensor]) -> torch.Tensor:
    """w1[h]: [D, D] fp32, w2[h]: [C_h, D] fp32 -> one packed byte buffer (both layers, MFMA fragment order)."""
    nh, D = len(w1), w1[0].shape[1]
    dc = dtype_code(dtype)
    dev = w1[0].device
    packed = torch.empty(lib().peneo_pair_heads_packed_bytes(dc, nh, D), dtype=torch.uint8, device=dev)
    classes = (C.c_int * nh)(*[w.shape[0] for w in w2])
    check(lib().peneo_pair_heads_pack(dc, _ptr_list([_c(w) for w in w1]), _ptr_list([_c(w) for w in w2]), classes, nh, D,
                                      ptr(packed), stream()), "peneo_pair_heads_pack")
    return packed


def pair_heads_fwd(ab: torch.Tensor, wp: torch.Tensor, b1: torch.Tensor, b2: torch.Tensor,
                   classes: Sequence[int], *, want_logits: bool = True, tags: Optional[Sequence[torch.Tensor]] = None,
                   class_weights: Optional[Sequence[Optional[torch.Tensor]]] = None, want_dlogits: bool = False,
                   drop_p: float = 0.0, drop_seed: int = 0, save: bool = False, save_buffers=None):
    """ab: [B, N, 2D].  Returns (logits list | None, loss partials [n, 32] | None, dlogits list | None).
    drop_p / drop_seed: the Dropout between the two classifier layers (train mode, model/peneo_decoder.py:261).
    save (pair_save_supported): the launch also leaves the hidden activations the backward needs; returns a fourth value
    (act [bytes] uint8, x_rows [B * pair_bwd_rows(N), D]) for pair_bwd_saved; save_buffers = (act, x_rows) supplies them."""
    _c(ab)
    B, N, D2 = ab.shape
    D = D2 // 2
    P = N * (N + 1) // 2
    nh = len(classes)
    desc = hip.PairHeadsDesc()
    desc.num_heads, desc.D = nh, D
    for h, c in enumerate(classes):
        desc.classes[h] = c
    desc.w_packed, desc.b1, desc.b2 = ptr(wp), ptr(b1), ptr(b2)
    desc.drop_p, desc.drop_seed = float(drop_p), int(drop_seed) & 0xFFFFFFFF
    logits = [torch.empty((B, P, c), dtype=torch.float32, device=ab.device) for c in classes] if want_logits else None
    lp = _ptr_list(logits) if logits is not None else None
    loss = None
    partials = None
    dlog = None
    if tags is not None:
        loss = hip.PairLoss()
        nrows = lib().peneo_pair_loss_partials_save(B, N) if save else lib().peneo_pair_loss_partials(B, N)
        partials = torch.empty((nrows, 32), dtype=torch.float32, device=ab.device)
        if want_dlogits:
            dlog = [torch.empty((B, P, c), dtype=torch.float32, device=ab.device) for c in classes]
        for h in range(nh):
            assert tags[h].dtype == torch.int64 and tags[h].shape == (B, P)
            loss.tags[h] = ptr(_c(tags[h]))
            loss.class_weight[h] = ptr(class_weights[h]) if class_weights is not None else None
            if dlog is not None:
                loss.dlogits[h] = ptr(dlog[h])
        loss.partials = ptr(partials)
    if save:
        if save_buffers is not None:
            act, x_rows = save_buffers
            assert act.dtype == torch.uint8 and act.numel() >= lib().peneo_pair_save_bytes(B, N, nh, D) and act.is_contiguous()
            assert x_rows.shape == (B * pair_bwd_rows(N), D) and x_rows.dtype == ab.dtype and x_rows.is_contiguous()
        else:
            act = torch.empty(lib().peneo_pair_save_bytes(B, N, nh, D), dtype=torch.uint8, device=ab.device)
            x_rows = torch.empty((B * pair_bwd_rows(N), D), dtype=ab.dtype, device=ab.device)
        with kernel_timer("pair_heads_fwd"):
            check(lib().peneo_pair_heads_fwd_save(dtype_code(ab.dtype), ptr(ab), B, N, C.byref(desc), lp,
                                                  C.byref(loss) if loss is not None else None, ptr(act), ptr(x_rows), stream()),
                  "peneo_pair_heads_fwd_save")
        return logits, partials, dlog, (act, x_rows)
    with kernel_timer("pair_heads_fwd"):
        check(lib().peneo_pair_heads_fwd(dtype_code(ab.dtype), ptr(ab), B, N, C.byref(desc), lp,
                                         C.byref(loss) if loss is not None else None, stream()), "peneo_pair_heads_fwd")
    return logits, partials, dlog


def pair_save_bytes(B: int, N: int, num_heads: int, D: int) -> int:
    return int(lib().peneo_pair_save_bytes(int(B), int(N), int(num_heads), int(D)))


def pair_save_supported(dtype, D: int, num_heads: int) -> bool:
    """True when peneo_pair_heads_fwd_save / peneo_pair_bwd_saved exist for this shape (bf16, D = 384)."""
    return bool(lib().peneo_pair_save_supported(dtype_code(dtype), int(D), int(num_heads)))


def pair_x_fwd(ab_doc: torch.Tensor, i0: int, i1: int, out: torch.Tensor, pre: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x = SiLU(a_i + b_j) for the pairs of rows i0..i1; `pre` (same shape) optionally receives a_i + b_j."""
    N, D2 = ab_doc.shape
    check(lib().peneo_pair_x_fwd(dtype_code(ab_doc.dtype), ptr(_c(ab_doc)), N, D2 // 2, i0, i1, ptr(out), ptr(pre), stream()),
          "peneo_pair_x_fwd")
    return out


def pair_x_bwd(ab_doc: torch.Tensor, i0: int, i1: int, dx: torch.Tensor, d_ab_doc: torch.Tensor,
               premultiplied: bool = False) -> None:
    N, D2 = ab_doc.shape
    assert d_ab_doc.dtype == torch.float32 and d_ab_doc.shape == ab_doc.shape
    check(lib().peneo_pair_x_bwd(dtype_code(ab_doc.dtype), ptr(_c(ab_doc)), N, D2 // 2, i0, i1, ptr(dx), ptr(_c(d_ab_doc)),
                                 int(premultiplied), stream()), "peneo_pair_x_bwd")


def pair_dz_workspace(nh: int, D: int, device, slots: Optional[int] = None) -> torch.Tensor:
    """Zeroed [slots, 4 * nh * D] fp32 accumulator for pair_dz (sum its rows with colsum at the end).  The stand-alone
    peneo_pair_dz needs the full size (default); the GEMM epilogue and peneo_pair_dz_fused only touch the first 256 rows."""
    full = lib().peneo_pair_dz_workspace_bytes(nh, D) // (16 * nh * D)
    return torch.zeros((full if slots is None else min(slots, full), 4 * nh * D), dtype=torch.float32, device=device)


def pair_dz_args(D: int, classes: Sequence[int], dlogits: Sequence[torch.Tensor], w2: Sequence[torch.Tensor],
                 scale: torch.Tensor, drop_p: float = 0.0, drop_seed: int = 0, drop_doc: int = 0,
                 drop_pair0: int = 0) -> "hip.PairDzArgs":
    """drop_*: the forward's classifier dropout; drop_doc / drop_pair0 locate a chunk (document, packed index of its first pair)."""
    a = hip.PairDzArgs()
    a.drop_p, a.drop_seed, a.drop_doc, a.drop_pair0 = float(drop_p), int(drop_seed) & 0xFFFFFFFF, int(drop_doc), int(drop_pair0)
    a.num_heads, a.D = len(classes), D
    for h, c in enumerate(classes):
        a.classes[h] = c
        a.dlogits[h], a.w2[h] = ptr(dlogits[h]), ptr(w2[h])
    a.scale = ptr(scale)
    return a


def pair_dz(z: torch.Tensor, npairs: int, D: int, classes: Sequence[int], dlogits: Sequence[torch.Tensor],
            w2: Sequence[torch.Tensor], workspace: torch.Tensor, scale: torch.Tensor,
            args: Optional["hip.PairDzArgs"] = None) -> None:
    a = args if args is not None else pair_dz_args(D, classes, dlogits, w2, scale)
    check(lib().peneo_pair_dz(dtype_code(z.dtype), ptr(z), npairs, C.byref(a), ptr(workspace), stream()), "peneo_pair_dz")


def pair_dz_fused(ab_doc: torch.Tensor, i0: int, i1: int, wp: torch.Tensor, b1: torch.Tensor, args: "hip.PairDzArgs",
                  dz: torch.Tensor, workspace: torch.Tensor, x: Optional[torch.Tensor] = None,
                  pre: Optional[torch.Tensor] = None) -> None:
    """dz [npairs, nh*D] of rows i0..i1 of one document straight from ab (bf16): no x / z round trip through memory.
    `x` / `pre` ([npairs, D], together) optionally receive what pair_x_fwd would write."""
    N, D2 = ab_doc.shape
    with kernel_timer("pair_dz_fused"):
        check(lib().peneo_pair_dz_fused(dtype_code(ab_doc.dtype), ptr(_c(ab_doc)), N, D2 // 2, i0, i1, ptr(wp), ptr(b1),
                                        C.byref(args), ptr(dz), ptr(workspace), ptr(x), ptr(pre), stream()),
              "peneo_pair_dz_fused")


def pair_bwd_supported(dtype: torch.dtype, D: int, num_heads: int = 0) -> bool:
    """The fused pair-space backward exists for (dtype, D) and its LDS image fits `num_heads` heads (0: do not ask)."""
    return dtype == torch.bfloat16 and bool(lib().peneo_pair_bwd_supported(BF16, D, num_heads))


def pair_bwd_rows(N: int) -> int:
    """Rows per document of the dz / x buffers of ``pair_bwd_fused`` (pairs in 8 x 16 blocks of the triangle)."""
    return int(lib().peneo_pair_bwd_rows(N))


def pair_bwd_pack(w1: Sequence[torch.Tensor]) -> torch.Tensor:
    """First-layer weights of all heads ([D, D] fp32 each) -> bf16 fragment stream of ``pair_bwd_fused``."""
    nh, D = len(w1), w1[0].shape[0]
    out = torch.empty(lib().peneo_pair_bwd_packed_bytes(nh, D), dtype=torch.uint8, device=w1[0].device)
    check(lib().peneo_pair_bwd_pack(_ptr_list([_c(w) for w in w1]), nh, D, ptr(out), stream()), "peneo_pair_bwd_pack")
    return out


def pair_bwd_fused(ab: torch.Tensor, wp: torch.Tensor, b1: torch.Tensor, args: "hip.PairDzArgs", dz: torch.Tensor,
                   x: torch.Tensor, d_ab: torch.Tensor, workspace: torch.Tensor) -> None:
    """Whole-batch decoder backward through the pair space (see include/peneo_hip.h): ab [B, N, 2D] bf16 ->
    dz [B * rows, nh*D], x [B * rows, D] (bf16, block order), d_ab [B, N, 2D] fp32 (overwritten), dW2 / db1 sums in `workspace`."""
    B, N, D2 = ab.shape
    assert d_ab.dtype == torch.float32 and d_ab.shape == ab.shape and dz.dtype == x.dtype == torch.bfloat16
    rows = pair_bwd_rows(N)
    assert dz.shape[0] == B * rows and x.shape == (B * rows, D2 // 2)
    partials = torch.empty(lib().peneo_pair_bwd_partial_bytes(B, N, D2 // 2) // 4, dtype=torch.float32, device=ab.device)
    with kernel_timer("pair_bwd_fused"):
        check(lib().peneo_pair_bwd_fused(dtype_code(ab.dtype), ptr(_c(ab)), B, N, D2 // 2, ptr(wp), ptr(b1), C.byref(args),
                                         ptr(_c(dz)), ptr(_c(x)), ptr(_c(d_ab)), ptr(workspace), ptr(partials), stream()),
              "peneo_pair_bwd_fused")


def pair_bwd_saved(ab: torch.Tensor, wp: torch.Tensor, args: "hip.PairDzArgs", act: torch.Tensor, dz: torch.Tensor,
                   d_ab: torch.Tensor, workspace: torch.Tensor) -> None:
    """pair_bwd_fused for a forward that saved its hidden activations (pair_heads_fwd(save=True)): same outputs except x, which
    the forward already wrote."""
    B, N, D2 = ab.shape
    assert d_ab.dtype == torch.float32 and d_ab.shape == ab.shape and dz.dtype == torch.bfloat16
    assert dz.shape[0] == B * pair_bwd_rows(N)
    partials = torch.empty(lib().peneo_pair_bwd_partial_bytes(B, N, D2 // 2) // 4, dtype=torch.float32, device=ab.device)
    with kernel_timer("pair_bwd_saved"):
        check(lib().peneo_pair_bwd_saved(dtype_code(ab.dtype), ptr(_c(ab)), B, N, D2 // 2, ptr(wp), C.byref(args), ptr(act),
                                         ptr(_c(dz)), ptr(_c(d_ab)), ptr(workspace), ptr(partials), stream()),
              "peneo_pair_bwd_saved")


def pair_dz_finish(workspace: torch.Tensor, nh: int, D: int, classes: Sequence[int]):
    """-> (dw2 list of [C_h, D] fp32, db1 [nh * D] fp32) from the accumulated workspace."""
    vec = colsum(workspace)                     # [4 * nh * D]
    ncol = nh * D
    blk = vec.view(4, ncol)
    dw2 = [blk[:c, h * D:(h + 1) * D] for h, c in enumerate(classes)]
    return dw2, blk[3]


def loss_finish(partials: torch.Tensor, ratio: torch.Tensor, total_classes: int):
    """-> (out [nh + 1] per-head losses then total, scale [2, nh] = (ratio_h / den_h, 1 / den_h), dl_sum [total_classes])."""
    nh = ratio.numel()
    out = torch.empty(nh + 1, dtype=torch.float32, device=partials.device)
    scale = torch.empty((2, nh), dtype=torch.float32, device=partials.device)
    dls = torch.empty(total_classes, dtype=torch.float32, device=partials.device)
    check(lib().peneo_loss_finish(ptr(partials), partials.shape[0], ptr(ratio), nh, total_classes, ptr(out), ptr(scale[0]),
                                  ptr(dls), ptr(scale[1]), stream()), "peneo_loss_finish")
    return out, scale, dls


def ohem_ce(logits: torch.Tensor, tags: torch.Tensor, cw: Optional[torch.Tensor], num_hard_positive: int, num_hard_negative: int,
            dlogits: Optional[torch.Tensor] = None, out8: Optional[torch.Tensor] = None, dl_sum: Optional[torch.Tensor] = None,
            workspace: Optional[torch.Tensor] = None):
    """OHEM cross entropy of one head over its flattened pairs (reference custom_loss.py:204-288, as executed).
    -> out8 = [loss, kept sum, k_pos + k_neg, n_pos, n_neg, k_pos, k_neg, 0] (device); `dlogits` is masked in place."""
    _c(logits)
    C_ = logits.shape[-1]
    n = logits.numel() // C_
    assert logits.dtype == torch.float32 and tags.dtype == torch.int64 and tags.numel() == n
    need = lib().peneo_ohem_workspace_bytes(n)
    if workspace is None or workspace.numel() < need:
        workspace = torch.empty(need, dtype=torch.uint8, device=logits.device)
    if out8 is None:
        out8 = torch.empty(8, dtype=torch.float32, device=logits.device)
    if dlogits is not None:
        assert dlogits.dtype == torch.float32 and dlogits.numel() == logits.numel()
        _c(dlogits)
    check(lib().peneo_ohem_ce(ptr(logits), ptr(_c(tags)), ptr(cw), n, C_, int(num_hard_positive), int(num_hard_negative),
                              ptr(dlogits), ptr(out8), ptr(dl_sum), ptr(workspace), workspace.numel(), stream()), "peneo_ohem_ce")
    return out8, workspace


def ohem_finish(out8: torch.Tensor, ratio: torch.Tensor):
    """[nh, 8] per-head OHEM results -> (out [nh + 1], scale [2, nh]) like ``loss_finish``."""
    nh = ratio.numel()
    out = torch.empty(nh + 1, dtype=torch.float32, device=out8.device)
    scale = torch.empty((2, nh), dtype=torch.float32, device=out8.device)
    check(lib().peneo_ohem_finish(ptr(_c(out8)), ptr(ratio), nh, ptr(out), ptr(scale[0]), ptr(scale[1]), stream()),
          "peneo_ohem_finish")
    return out, scale


def weighted_ce(logits: torch.Tensor, tags: torch.Tensor, cw: Optional[torch.Tensor], want_dlogits: bool = False):
    _c(logits)
    C_ = logits.shape[-1]
    rows = logits.numel() // C_
    num = torch.zeros(1, dtype=torch.float32, device=logits.device)
    den = torch.zeros(1, dtype=torch.float32, device=logits.device)
    dl = torch.empty_like(logits) if want_dlogits else None
    check(lib().peneo_weighted_ce(ptr(logits), ptr(_c(tags)), ptr(cw), rows, C_, ptr(num), ptr(den), ptr(dl), stream()),
          "peneo_weighted_ce")
    return num, den, dl


def spots_compact(logits: torch.Tensor, N: int, max_spots: int = 4096):
    """[P, C] fp32 logits -> (spots int32 [n, 3] (i, j, tag), scores fp32 [n]) in increasing p order."""
    _c(logits)
    P, C_ = logits.shape
    spots = torch.empty((max_spots, 3), dtype=torch.int32, device=logits.device)
    scores = torch.empty(max_spots, dtype=torch.float32, device=logits.device)
    count = torch.zeros(1, dtype=torch.int32, device=logits.device)
    check(lib().peneo_spots_compact(ptr(logits), P, C_, N, ptr(spots), ptr(scores), ptr(count), max_spots, stream()),
          "peneo_spots_compact")
    n = int(count.item())
    if n > max_spots:
        return spots_compact(logits, N, max_spots=n)
    return spots[:n], scores[:n]


def spots_to_tags(batch_spots, N: int, device, B: int = None) -> torch.Tensor:
    """Sparse spots -> dense label maps [B, P] int64 built on the device (K13 input side).  ``batch_spots``: a list with
    one [(i, j, tag), ...] list per document, or an int32 tensor [n, 4] of (b, i, j, tag) rows (then ``B`` is required;
    what ``DataCollatorForPEneo(sparse_tags=True)`` ships)."""
    if torch.is_tensor(batch_spots):
        assert B is not None and batch_spots.dim() == 2 and batch_spots.shape[1] == 4
        n = batch_spots.shape[0]
        sp = batch_spots.to(device=device, dtype=torch.int32).contiguous() if n else None
    else:
        B = len(batch_spots)
        flat = [(b, int(sp[0]), int(sp[1]), int(sp[2])) for b, spots in enumerate(batch_spots) for sp in spots]
        n = len(flat)
        sp = torch.tensor(flat, dtype=torch.int32).view(-1, 4).to(device) if flat else None
    tags = torch.empty((B, N * (N + 1) // 2), dtype=torch.int64, device=device)
    status = torch.zeros(1, dtype=torch.int32, device=device)
    check(lib().peneo_spots_to_tags(ptr(sp), n, B, N, ptr(tags), ptr(status), stream()), "peneo_spots_to_tags")
    if n and int(status) != 0:
        raise IndexError("spot outside the [0, N) x [0, N) pair matrix")
    return tags
