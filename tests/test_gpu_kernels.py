"""Per-kernel parity: every libpeneo_hip.so entry point against a plain fp32 PyTorch statement of the same op.

fp32 mode (exact fp32 MFMA) is held to ~1e-5 relative; bf16 mode is compared with the fp32
result computed from the *same bf16-rounded inputs* (so only accumulation order and the output
rounding differ) at ~1e-2 relative.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


@pytest.fixture(scope="module")
def ops():
    from peneo_amd import ops as o
    from peneo_amd import hip
    hip.load_library()
    return o


def rel_err(a, b):
    a, b = a.float(), b.float()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-6))


def tol(dtype):
    return 2e-5 if dtype == torch.float32 else 2e-2


DTYPES = [torch.float32, torch.bfloat16]


# ---------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K", [(709, 200, 136), (128, 128, 64), (300, 768, 768), (33, 24, 8)])
def test_gemm_layouts(ops, dtype, ak, bk, M, N, K):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K)
    a = torch.randn(M, K, generator=g).to(DEV).to(dtype)
    b = torch.randn(N, K, generator=g).to(DEV).to(dtype)
    ref = a.float() @ b.float().t()
    A = a if ak else a.t().contiguous()
    Bm = b if bk else b.t().contiguous()
    out = ops.gemm(A, Bm, a_kmajor=ak, b_kmajor=bk, out_dtype=torch.float32)
    assert rel_err(out, ref) < tol(dtype), (rel_err(out, ref))


@pytest.mark.parametrize("ak,bk", [(True, True), (True, False), (False, True), (False, False)])
@pytest.mark.parametrize("M,N,K,split", [(712, 200, 136, 1), (128, 128, 64, 1), (1000, 264, 200, 1), (256, 384, 1064, 3),
                                         (8, 8, 8, 1), (136, 520, 72, 1), (768, 768, 5672, 8)])
def test_gemm_bf16_lds_dma_path(ops, ak, bk, M, N, K, split):
    """Shapes that qualify for the LDS-DMA kernel (16-byte aligned chunks) with ragged M/N edges, a k tail and
    split-k; operands are strided views of wider buffers so that leading dims differ from the extents."""
    g = torch.Generator(device="cpu").manual_seed(M + 3 * N + 5 * K)
    a = torch.randn(M, K, generator=g).to(DEV).to(torch.bfloat16)
    b = torch.randn(N, K, generator=g).to(DEV).to(torch.bfloat16)
    ref = a.float() @ b.float().t()

    def view(t):   # embed in a wider buffer: ld = cols + 16
        buf = torch.full((t.shape[0], t.shape[1] + 16), 7.0, device=DEV, dtype=t.dtype)
        buf[:, :t.shape[1]] = t
        return buf[:, :t.shape[1]]
    A = view(a if ak else a.t().contiguous())
    Bm = view(b if bk else b.t().contiguous())
    out = ops.gemm(A, Bm, a_kmajor=ak, b_kmajor=bk, out_dtype=torch.float32, split_k=split)
    # bf16 inputs are exact in fp32 and the output stays fp32: only the accumulation order differs from torch
    assert rel_err(out, ref) < 1e-5, rel_err(out, ref)


def test_gemm_bf16_lds_dma_kernel_all_layouts(ops):
    """The LDS-DMA kernel on every operand layout, ragged shapes with a k tail and split-k (fp32 output: only the accumulation
    order differs from torch)."""
    torch.manual_seed(0)
    for (M, N, K, split) in [(712, 200, 136, 1), (256, 384, 1064, 3), (136, 520, 72, 1), (768, 768, 5672, 8), (8, 8, 8, 1)]:
        a = torch.randn(M, K).to(DEV).to(torch.bfloat16)
        b = torch.randn(N, K).to(DEV).to(torch.bfloat16)
        ref = a.float() @ b.float().t()
        for ak in (True, False):
            for bk in (True, False):
                A = a if ak else a.t().contiguous()
                B = b if bk else b.t().contiguous()
                out = ops.gemm(A, B, a_kmajor=ak, b_kmajor=bk, out_dtype=torch.float32, split_k=split)
                assert rel_err(out, ref) < 1e-5, (M, N, K, ak, bk)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_is_transpose_detecting(ops, dtype):
    # A = identity against an asymmetric B catches swapped C rows/cols
    n = 128
    a = torch.eye(n, device=DEV, dtype=dtype)
    b = (torch.arange(n * n, device=DEV, dtype=torch.float32).view(n, n) % 251).to(dtype)
    out = ops.gemm(a, b, out_dtype=torch.float32)
    assert torch.equal(out, b.float().t())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogues(ops, dtype):
    from peneo_amd.hip import ACT_GELU, ACT_SILU
    M, N, K = 260, 192, 96
    g = torch.Generator().manual_seed(5)
    a = torch.randn(M, K, generator=g).to(DEV).to(dtype)
    b = (torch.randn(N, K, generator=g) * 0.2).to(DEV).to(dtype)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV).to(dtype)
    z_ref = a.float() @ b.float().t() + bias
    pre = torch.empty(M, N, device=DEV, dtype=dtype)
    out = ops.gemm(a, b, bias=bias, act=ACT_GELU, residual=res, preact=pre)
    assert rel_err(pre, z_ref) < tol(dtype)
    assert rel_err(out, F.gelu(z_ref) + res.float()) < tol(dtype)
    # dgrad-style: multiply by SiLU'(src)
    src = torch.randn(M, N, generator=g).to(DEV).to(dtype)
    out2 = ops.gemm(a, b, grad_src=src, grad_act=ACT_SILU)
    s = torch.sigmoid(src.float())
    assert rel_err(out2, (a.float() @ b.float().t()) * (s * (1 + src.float() * (1 - s)))) < tol(dtype)
    # accumulate into fp32 + split-k
    acc = torch.ones(M, N, device=DEV)
    ops.gemm(a, b, out=acc, accumulate=True, split_k=3)
    assert rel_err(acc, 1 + a.float() @ b.float().t()) < tol(dtype)
    # alpha
    out3 = ops.gemm(a, b, alpha=0.25, out_dtype=torch.float32)
    assert rel_err(out3, 0.25 * (a.float() @ b.float().t())) < tol(dtype)


def test_gemm_dropout_mask_is_reproducible(ops):
    M, N, K = 256, 128, 64
    a = torch.randn(M, K, device=DEV)
    b = torch.randn(N, K, device=DEV)
    base = ops.gemm(a, b)
    d1 = ops.gemm(a, b, drop_p=0.25, drop_seed=77)
    d2 = ops.gemm(a, b, drop_p=0.25, drop_seed=77)
    d3 = ops.gemm(a, b, drop_p=0.25, drop_seed=78)
    assert torch.equal(d1, d2) and not torch.equal(d1, d3)
    kept = d1 != 0
    frac = float(kept.float().mean())
    assert 0.72 < frac < 0.78
    assert rel_err(d1[kept], base[kept] / 0.75) < 1e-5


def test_gemm_rejects_bad_arguments(ops):
    from peneo_amd.hip import PeneoHipError
    a = torch.randn(8, 8, device=DEV)
    with pytest.raises(PeneoHipError):
        ops.gemm(a, a, out=torch.empty(8, 8, device=DEV, dtype=torch.bfloat16), accumulate=True)


# ---------------------------------------------------------------------------------------------- element-wise
def test_cast_copy_colsum(ops):
    x = torch.randn(1000, 77, device=DEV)
    xb = ops.cast(x, torch.bfloat16)
    assert torch.equal(xb, x.to(torch.bfloat16))
    assert torch.equal(ops.cast(xb, torch.float32), xb.float())
    big = torch.randn(50, 200, device=DEV)
    sub = big[:, 10:90]
    assert torch.equal(ops.copy2d(sub), sub)
    for dt in DTYPES:
        y = torch.randn(1300, 200, device=DEV).to(dt)
        assert rel_err(ops.colsum(y), y.float().sum(0)) < 1e-4                      # 16-byte vector kernel
        assert rel_err(ops.colsum(y[:, 8:80]), y[:, 8:80].float().sum(0)) < 1e-4    # ... on a strided slice
        assert rel_err(ops.colsum(y[:, 3:80]), y[:, 3:80].float().sum(0)) < 1e-4    # unaligned -> scalar kernel
        acc = torch.ones(200, device=DEV)
        ops.colsum(y, out=acc, accumulate=True)
        assert rel_err(acc, 1 + y.float().sum(0)) < 1e-4


# ---------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H", [768, 64, 1024, 24, 256, 192, 384, 96])
def test_layernorm_fwd_bwd(ops, dtype, H):
    rows = 517
    g = torch.Generator().manual_seed(H)
    x = (torch.randn(rows, H, generator=g) * 2 + 0.3).to(DEV).to(dtype)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV)
    beta = (0.1 * torch.randn(H, generator=g)).to(DEV)
    dy = torch.randn(rows, H, generator=g).to(DEV).to(dtype)
    xr = x.float().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    yr = F.layer_norm(xr, (H,), gr, br, 1e-5)
    yr.backward(dy.float())
    y, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    assert rel_err(y, yr) < tol(dtype)
    dg = torch.zeros(H, device=DEV)
    db = torch.zeros(H, device=DEV)
    dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db)
    assert rel_err(dx, xr.grad) < tol(dtype)
    assert rel_err(dg, gr.grad) < 5e-4 and rel_err(db, br.grad) < 5e-4


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H", [768, 192, 96])
def test_layernorm_bwd_column_sums_of_the_dropped_output(ops, dtype, H):
    """dx_colsum of peneo_layernorm_bwd = column sums of its second (dropped) output, or of dx without one: the bias gradient of
    the Linear in front of the LayerNorm, which used to be a separate column-sum launch (fast half-wave kernels and the generic one)."""
    R = 777
    g = torch.Generator().manual_seed(H)
    x = torch.randn(R, H, generator=g).to(DEV).to(dtype)
    dy = torch.randn(R, H, generator=g).to(DEV).to(dtype)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV)
    beta = torch.zeros(H, device=DEV)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    for with_drop in (False, True):
        dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
        col = torch.full((H,), 0.5, device=DEV)        # accumulated into
        dxd = torch.empty_like(x) if with_drop else None
        dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dx_dropped=dxd, drop2_p=0.2 if with_drop else 0.0, drop2_seed=9,
                               dx_colsum=col)
        src = dxd if with_drop else dx
        ref = src.float().sum(0) + 0.5
        scale = float(src.float().abs().sum(0).max())
        assert float((col - ref).abs().max()) < (2e-5 if dtype == torch.float32 else 6e-3) * scale, (dtype, H, with_drop)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("H", [768, 48, 192])
def test_layernorm_dropout_fwd_bwd_share_the_mask(ops, dtype, H):
    rows = 300
    g = torch.Generator().manual_seed(H + 1)
    x = torch.randn(rows, H, generator=g).to(DEV).to(dtype)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV)
    beta = (0.5 + 0.1 * torch.randn(H, generator=g)).to(DEV)
    y0, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    y1, _, _ = ops.layernorm_fwd(x, gamma, beta, 1e-5, drop_p=0.3, drop_seed=11)
    y2, _, _ = ops.layernorm_fwd(x, gamma, beta, 1e-5, drop_p=0.3, drop_seed=11)
    assert torch.equal(y1, y2)
    keep = y1 != 0
    assert 0.65 < float(keep.float().mean()) < 0.75
    assert rel_err(y1[keep], y0[keep].float() / 0.7) < tol(dtype)
    dy = torch.randn(rows, H, generator=g).to(DEV).to(dtype)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dx = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, drop_p=0.3, drop_seed=11)
    dy_masked = (dy.float() * keep / 0.7).to(dtype)
    dg2, db2 = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dx2 = ops.layernorm_bwd(dy_masked, x, gamma, mean, rstd, dg2, db2)
    assert rel_err(dx, dx2) < tol(dtype) and rel_err(dg, dg2) < tol(dtype) and rel_err(db, db2) < tol(dtype)
    # second output: dx through another dropout mask (the producer GEMM's) == a separate masking pass over dx
    dxd = torch.empty_like(x)
    dg3, db3 = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dx3 = ops.layernorm_bwd(dy_masked, x, gamma, mean, rstd, dg3, db3, dx_dropped=dxd, drop2_p=0.25, drop2_seed=77)
    assert torch.equal(dx3, dx2)
    ref_d = ops.copy2d(dx2, drop_p=0.25, drop_seed=77)
    assert torch.equal(dxd == 0, ref_d == 0)                       # the same mask ...
    assert rel_err(dxd, ref_d) < (1e-6 if dtype == torch.float32 else 1e-2)   # ... bf16: one rounding instead of two


def test_layernorm_strided_rows_fast_path(ops):
    B, T, S, H = 3, 50, 20, 256
    for dt in DTYPES:
        buf = torch.randn(B, T, H, device=DEV).to(dt)
        gamma, beta = torch.rand(H, device=DEV) + 0.5, torch.randn(H, device=DEV)
        ref = buf.clone()
        ref[:, :S] = F.layer_norm(buf[:, :S].float(), (H,), gamma, beta, 1e-5).to(dt)
        ops.layernorm_fwd(buf[:, :S], gamma, beta, 1e-5, out=buf[:, :S])
        assert rel_err(buf, ref) < tol(dt)


def test_layernorm_strided_rows(ops):
    B, T, S, H = 3, 50, 20, 64
    buf = torch.randn(B, T, H, device=DEV)
    gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    ref = buf.clone()
    ref[:, :S] = F.layer_norm(buf[:, :S], (H,), gamma, beta, 1e-5)
    ops.layernorm_fwd(buf[:, :S], gamma, beta, 1e-5, out=buf[:, :S])
    assert rel_err(buf, ref) < 2e-5


# ---------------------------------------------------------------------------------------------- embeddings
def test_position_ids(ops):
    ids = torch.randint(3, 100, (4, 57), device=DEV)
    ids[0, 40:] = 1
    ids[2, 10:] = 1
    ids[3, :] = 1
    mask = ids.ne(1).int()
    ref = (torch.cumsum(mask, 1).type_as(mask) * mask).long() + 1
    assert torch.equal(ops.position_ids(ids, 1).long(), ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_fwd_bwd(ops, dtype):
    B, S, H, cs, ss, V, MP = 2, 37, 64, 11, 10, 300, 66
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(3, V, (B, S), generator=g)
    ids[1, 30:] = 1
    bbox = torch.randint(0, 500, (B, S, 4), generator=g)
    bbox[..., 2] += bbox[..., 0]
    bbox[..., 3] += bbox[..., 1]
    tabs = {n: torch.randn(s, generator=g).to(DEV) for n, s in
            dict(word=(V, H), type=(1, H), pos=(MP, H), x=(1024, cs), y=(1024, cs), h=(1024, ss), w=(1024, ss)).items()}
    ids, bbox = ids.to(DEV), bbox.to(DEV)
    pid = ops.position_ids(ids, 1)
    leaf = {n: t.clone().requires_grad_(True) for n, t in tabs.items()}
    ref = (F.embedding(ids, leaf["word"], padding_idx=1) + leaf["type"][0] + F.embedding(pid.long(), leaf["pos"], padding_idx=1)
           + torch.cat([F.embedding(bbox[..., 0], leaf["x"]), F.embedding(bbox[..., 1], leaf["y"]),
                        F.embedding(bbox[..., 2], leaf["x"]), F.embedding(bbox[..., 3], leaf["y"]),
                        F.embedding((bbox[..., 3] - bbox[..., 1]).clip(0, 1023), leaf["h"]),
                        F.embedding((bbox[..., 2] - bbox[..., 0]).clip(0, 1023), leaf["w"])], -1))
    out = torch.empty(B, S, H, device=DEV, dtype=dtype)
    status = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.embed_fwd(dtype, out, B, S, H, input_ids=ids, pos_ids=pid, bbox=bbox, word=tabs["word"], type0=tabs["type"][0],
                  pos=tabs["pos"], x=tabs["x"], y=tabs["y"], h=tabs["h"], w=tabs["w"], status=status)
    assert int(status) == 0
    assert rel_err(out, ref) < (1e-6 if dtype == torch.float32 else 1e-2)
    d_out = torch.randn(B, S, H, generator=g).to(DEV).to(dtype)
    ref.backward(d_out.float())
    grads = {n: torch.zeros_like(t) for n, t in tabs.items()}
    ops.embed_bwd(d_out, B, S, H, input_ids=ids, pos_ids=pid, bbox=bbox, g_word=grads["word"], g_pos=grads["pos"],
                  g_x=grads["x"], g_y=grads["y"], g_h=grads["h"], g_w=grads["w"], pad_id=1)
    for n in ("word", "pos", "x", "y", "h", "w"):
        assert rel_err(grads[n], leaf[n].grad) < 1e-4, n
    # out-of-range coordinate -> status flag (the reference raises IndexError)
    bad = bbox.clone()
    bad[0, 3, 2] = 1500
    ops.embed_fwd(dtype, out, B, S, H, input_ids=ids, pos_ids=pid, bbox=bad, word=tabs["word"], type0=tabs["type"][0],
                  pos=tabs["pos"], x=tabs["x"], y=tabs["y"], h=tabs["h"], w=tabs["w"], status=status)
    assert int(status) == 1


@pytest.mark.parametrize("dtype", DTYPES)
def test_embed_bwd_with_the_collisions_of_real_batches(ops, dtype):
    """Embedding-table gradients under the collision pattern of real batches - line boxes replicated over a line's tokens, two
    heights / widths in the whole batch, position rows shared by all documents - against torch's index_add.  (An LDS-image form of
    the small tables was tried against the token-per-wave atomics: 0.28-0.33 ms against 0.185 ms, not kept.)"""
    B, S, H, cs, ss, V = 3, 200, 192, 32, 32, 500
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(3, V, (B, S), generator=g)
    ids[2, 150:] = 1
    lines = torch.randint(0, 20, (B, S), generator=g)                     # 20 line boxes per document
    x0, y0 = (lines * 37) % 800, lines * 40
    bbox = torch.stack([x0, y0, x0 + 150 - 20 * (lines % 2), y0 + 8 + 4 * (lines % 2)], -1)
    tabs = {n: torch.zeros(sh, device=DEV) for n, sh in
            dict(word=(V, H), pos=(S + 2, H), x=(1024, cs), y=(1024, cs), h=(1024, ss), w=(1024, ss)).items()}
    ids, bbox = ids.to(DEV), bbox.to(DEV)
    pid = ops.position_ids(ids, 1)
    d_out = torch.randn(B, S, H, generator=g).to(DEV).to(dtype)
    grads = {n: torch.full_like(t, 0.25) for n, t in tabs.items()}          # accumulated into
    ops.embed_bwd(d_out, B, S, H, input_ids=ids, pos_ids=pid, bbox=bbox, g_word=grads["word"], g_pos=grads["pos"],
                  g_x=grads["x"], g_y=grads["y"], g_h=grads["h"], g_w=grads["w"], pad_id=1)
    d = d_out.float().view(B * S, H)
    ref = {n: torch.full_like(t, 0.25) for n, t in tabs.items()}
    keep = (ids.view(-1) != 1)
    ref["word"].index_add_(0, ids.view(-1)[keep], d[keep])
    keep_p = (pid.view(-1) != 1)
    ref["pos"].index_add_(0, pid.view(-1).long()[keep_p], d[keep_p])
    bb = bbox.view(-1, 4)
    ref["x"].index_add_(0, bb[:, 0], d[:, 0:cs]); ref["x"].index_add_(0, bb[:, 2], d[:, 2 * cs:3 * cs])
    ref["y"].index_add_(0, bb[:, 1], d[:, cs:2 * cs]); ref["y"].index_add_(0, bb[:, 3], d[:, 3 * cs:4 * cs])
    ref["h"].index_add_(0, (bb[:, 3] - bb[:, 1]).clip(0, 1023), d[:, 4 * cs:4 * cs + ss])
    ref["w"].index_add_(0, (bb[:, 2] - bb[:, 0]).clip(0, 1023), d[:, 4 * cs + ss:4 * cs + 2 * ss])
    for n in tabs:
        assert rel_err(grads[n], ref[n]) < 2e-5, n


@pytest.mark.parametrize("dtype", DTYPES)
def test_patch_embed_pieces(ops, dtype):
    B, H = 2, 64
    img = torch.randn(B, 3, 224, 224, device=DEV)
    w = torch.randn(H, 3, 16, 16, device=DEV) * 0.05
    bias = torch.randn(H, device=DEV)
    cls, pos = torch.randn(H, device=DEV), torch.randn(197, H, device=DEV)
    patches = ops.im2col_patch16(img, dtype)
    proj = ops.gemm(patches, w.view(H, -1).to(dtype).contiguous(), bias=bias)
    ref = F.conv2d(img.to(dtype).float(), w.to(dtype).float(), bias, stride=16).flatten(2).transpose(1, 2)
    assert rel_err(proj.view(B, 196, H), ref) < tol(dtype)
    vis = ops.visual_assemble_fwd(proj, cls, pos, B)
    refv = torch.cat([cls.expand(B, 1, H), proj.float().view(B, 196, H)], 1) + pos
    assert rel_err(vis, refv) < tol(dtype)
    dv = torch.randn(B, 197, H, device=DEV).to(dtype)
    dc, dp = torch.zeros(H, device=DEV), torch.zeros(197, H, device=DEV)
    dproj = ops.visual_assemble_bwd(dv, dc, dp)
    assert torch.equal(dproj.view(B, 196, H), dv[:, 1:])
    assert rel_err(dp, dv.float().sum(0)) < 1e-5 and rel_err(dc, dv.float()[:, 0].sum(0)) < 1e-5


def test_relpos_inputs_one_launch(ops):
    """peneo_relpos_inputs: key mask over text + visual tokens and the per-token inputs of the bucket maps (the reference builds
    them with arange / cat / slicing: modeling_layoutlmv3.py:1052-1080, 586-676)."""
    B, S, nv = 3, 40, 17
    g = torch.Generator().manual_seed(2)
    am = (torch.rand(B, S, generator=g) > 0.2).long().to(DEV)
    bbox = torch.randint(0, 1000, (B, S, 4), generator=g).to(DEV)
    vx = torch.randint(0, 1000, (nv,), generator=g).int().to(DEV)
    vy = torch.randint(0, 1000, (nv,), generator=g).int().to(DEV)
    km, pos, xs, ys = ops.relpos_inputs(am, bbox, vx, vy, B, S, nv, True, True)
    assert torch.equal(km, torch.cat([am.int(), torch.ones(B, nv, dtype=torch.int32, device=DEV)], 1))
    assert torch.equal(pos, torch.cat([torch.arange(S), torch.arange(nv)]).int().to(DEV).expand(B, -1))
    assert torch.equal(xs, torch.cat([bbox[..., 0].int(), vx.expand(B, -1)], 1))
    assert torch.equal(ys, torch.cat([bbox[..., 3].int(), vy.expand(B, -1)], 1))
    km2, pos2, xs2, ys2 = ops.relpos_inputs(None, bbox, None, None, B, S, 0, False, True)     # no mask given, no visual tokens, no 1-D part
    assert pos2 is None and bool((km2 == 1).all()) and torch.equal(xs2, bbox[..., 0].int()) and torch.equal(ys2, bbox[..., 3].int())


def test_attention_drop_words_sets_are_independent_and_reproducible(ops):
    """One launch makes the keep bits of several calls (layers): same seed -> same bits, another seed or another set -> other bits,
    every set at the requested keep rate."""
    B, nh, T, p_drop = 2, 3, 300, 0.1
    w1 = ops.attn_drop_words(B, nh, T, p_drop, 123, sets=4)
    w2 = ops.attn_drop_words(B, nh, T, p_drop, 123, sets=4)
    w3 = ops.attn_drop_words(B, nh, T, p_drop, 124, sets=4)
    assert torch.equal(w1, w2) and not torch.equal(w1, w3)
    assert w1.shape[:2] == (4, B * nh) and w1.shape[2] * 32 >= T and w1.shape[3] >= T
    bits = ((w1.long().unsqueeze(-1) >> torch.arange(32, device=DEV)) & 1).float()
    for s_ in range(4):
        rate = float(bits[s_].mean())
        assert abs(rate - (1 - p_drop)) < 2e-3, (s_, rate)
        if s_:
            assert not torch.equal(w1[s_], w1[0])



# ---------------------------------------------------------------------------------------------- rel-pos bias
def test_relpos_buckets_and_bias(ops):
    from oracle import peneo_oracle as O
    from peneo_amd.model.relpos import bucket_lut
    B, T, nh = 2, 237, 4
    g = torch.Generator().manual_seed(11)
    pos = torch.cat([torch.arange(40), torch.arange(197)]).repeat(B, 1)
    xs = torch.randint(0, 1001, (B, T), generator=g)
    ys = torch.randint(0, 1001, (B, T), generator=g)
    lut1, lut2 = bucket_lut(32, 128, 1024).to(DEV), bucket_lut(64, 256, 1024).to(DEV)
    bk1, bkx, bky = ops.relpos_buckets(pos.int().to(DEV), xs.int().to(DEV), ys.int().to(DEV), B, T, lut1, 16, lut2, 32)
    r1 = O.relative_position_bucket(pos.unsqueeze(-2) - pos.unsqueeze(-1), 32, 128)
    rx = O.relative_position_bucket(xs.unsqueeze(-2) - xs.unsqueeze(-1), 64, 256)
    ry = O.relative_position_bucket(ys.unsqueeze(-2) - ys.unsqueeze(-1), 64, 256)
    assert torch.equal(bk1.cpu().long(), r1) and torch.equal(bkx.cpu().long(), rx) and torch.equal(bky.cpu().long(), ry)
    w1 = torch.randn(nh, 32, generator=g).to(DEV)
    wx, wy = torch.randn(nh, 64, generator=g).to(DEV), torch.randn(nh, 64, generator=g).to(DEV)
    scale = 0.25
    km = torch.ones(B, T, dtype=torch.int32)
    km[1, 30:37] = 0
    bias = ops.relpos_bias_fwd(torch.float32, bk1, bkx, bky, w1, wx, wy, scale, B, nh, T, key_mask=km.to(DEV))
    Tp = bias.shape[-1]
    assert Tp % 64 == 0 and Tp >= T
    ref = scale * (w1.t()[r1.to(DEV)] + wx.t()[rx.to(DEV)] + wy.t()[ry.to(DEV)]).permute(0, 3, 1, 2)
    valid = km.bool().to(DEV)[:, None, None, :].expand(B, nh, T, T)
    assert rel_err(bias[..., :T][valid], ref[valid]) < 1e-6
    assert bool((bias[..., T:] < -1e29).all()) and bool((bias[..., :T][~valid] < -1e29).all())
    gg = torch.zeros(B, nh, T, Tp, device=DEV)
    gg[..., :T] = torch.randn(B, nh, T, T, generator=g).to(DEV)
    d1, dx, dy = torch.zeros_like(w1), torch.zeros_like(wx), torch.zeros_like(wy)
    ops.relpos_bias_bwd(gg, bk1, bkx, bky, d1, dx, dy, scale)
    gg = gg[..., :T]
    w1r, wxr, wyr = (t.clone().requires_grad_(True) for t in (w1, wx, wy))
    (scale * (w1r.t()[r1.to(DEV)] + wxr.t()[rx.to(DEV)] + wyr.t()[ry.to(DEV)]).permute(0, 3, 1, 2) * gg).sum().backward()
    assert rel_err(d1, w1r.grad) < 1e-4 and rel_err(dx, wxr.grad) < 1e-4 and rel_err(dy, wyr.grad) < 1e-4


# ---------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, bias, mask, scale):
    s = torch.einsum("bhqd,bhkd->bhqk", q, k) * scale
    if bias is not None:
        s = s + bias
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
    p = torch.softmax(s, -1)
    return torch.einsum("bhqk,bhkd->bhqd", p, v)


def _padded_bias(bias, mask):
    """[B, nh, T, T] natural bias + [B, T] int mask -> the kernel's [B, nh, T, Tp] layout with -1e30 masking."""
    B, nh, T, _ = bias.shape
    Tp = (T + 63) // 64 * 64
    out = torch.full((B, nh, T, Tp), -1.0e30, dtype=bias.dtype, device=bias.device)
    out[..., :T] = bias
    if mask is not None:
        out[..., :T].masked_fill_((mask == 0)[:, None, None, :], -1.0e30)
    return out


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,nh,T,d", [(2, 3, 237, 16), (1, 2, 709, 64), (2, 2, 64, 30), (1, 2, 130, 80)])
def test_attention_fwd_bwd(ops, dtype, B, nh, T, d):
    g = torch.Generator().manual_seed(T + d)
    H = nh * d
    qkv = (torch.randn(B * T, 3 * H, generator=g)).to(DEV).to(dtype)
    bias = (0.5 * torch.randn(B, nh, T, T, generator=g)).to(DEV).to(dtype)
    mask = torch.ones(B, T, dtype=torch.int32)
    mask[0, T // 3: T // 2] = 0
    mask = mask.to(DEV)
    scale = 1.0 / math.sqrt(d)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    leaf = qkv.float().clone().requires_grad_(True)
    br = bias.float().clone().requires_grad_(True)
    hd = lambda t: t.view(B, T, nh, d).permute(0, 2, 1, 3)
    ref = _attn_ref(hd(leaf[:, :H]), hd(leaf[:, H:2 * H]), hd(leaf[:, 2 * H:]), br, mask, scale)
    ref2d = ref.permute(0, 2, 1, 3).reshape(B * T, H)
    pb = _padded_bias(bias, mask)
    out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, scale, pb)
    assert rel_err(out, ref2d) < tol(dtype), rel_err(out, ref2d)
    d_out = torch.randn(B * T, H, generator=g).to(DEV).to(dtype)
    ref2d.backward(d_out.float())
    dqkv = torch.empty_like(qkv)
    gbias = torch.zeros(pb.shape, dtype=torch.float32, device=DEV)
    ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, scale, pb, None, dqkv, gbias)
    if dtype == torch.bfloat16:   # the two-kernel path must agree with the single-pass default
        dqkv2 = torch.empty_like(qkv)
        gbias2 = torch.zeros_like(gbias)
        ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, scale, pb, None, dqkv2, gbias2, single_pass=False)
        assert rel_err(dqkv, dqkv2) < 2e-2 and rel_err(gbias, gbias2) < 2e-2
    t = 3e-5 if dtype == torch.float32 else 4e-2
    assert rel_err(dqkv[:, 2 * H:], leaf.grad[:, 2 * H:]) < t, "dv"
    assert rel_err(dqkv[:, H:2 * H], leaf.grad[:, H:2 * H]) < t, "dk"
    assert rel_err(dqkv[:, :H], leaf.grad[:, :H]) < t, "dq"
    assert rel_err(gbias[..., :T], br.grad) < t, "dbias"
    assert float(gbias[..., T:].abs().max()) == 0.0 if gbias.shape[-1] > T else True
    # accumulation semantics of the bias gradient
    ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, scale, pb, None, dqkv, gbias)
    assert rel_err(gbias[..., :T], 2 * br.grad) < t


@pytest.mark.parametrize("d", [64, 80])
def test_attention_bf16_single_pass_matches_two_kernel_path_with_dropout(ops, d):
    """Both backward implementations regenerate the forward's dropout mask from (seed, element index)."""
    B, nh, T = 2, 2, 200
    H = nh * d
    g = torch.Generator().manual_seed(d)
    qkv = torch.randn(B * T, 3 * H, generator=g).to(DEV).to(torch.bfloat16)
    bias = _padded_bias((0.5 * torch.randn(B, nh, T, T, generator=g)).to(DEV).to(torch.bfloat16), None)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 0.11, bias, None, drop_p=0.2, drop_seed=9)
    d_out = torch.randn(B * T, H, generator=g).to(DEV).to(torch.bfloat16)
    res = []
    for single, atomic in ((True, False), (False, False), (True, True)):
        dqkv = torch.zeros_like(qkv)
        gb = torch.zeros(bias.shape, dtype=torch.float32, device=DEV)
        ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.11, bias, None, dqkv, gb, drop_p=0.2, drop_seed=9, single_pass=single,
                     dq_atomic=atomic)
        res.append((dqkv.float(), gb))
    assert rel_err(res[0][0], res[1][0]) < 2e-2
    assert rel_err(res[0][1], res[1][1]) < 2e-2
    assert rel_err(res[2][0], res[1][0]) < 2e-2 and rel_err(res[2][1], res[1][1]) < 2e-2   # dQ through fp32 atomics
    # per-layer dS^T copy (key-major bf16) instead of the fp32 accumulator; padding columns stay zero
    Tp = bias.shape[-1]
    ds = torch.full((B, nh, T, Tp), 7.0, device=DEV, dtype=torch.bfloat16)
    dq2 = torch.zeros_like(qkv)
    ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.11, bias, None, dq2, None, drop_p=0.2, drop_seed=9, ds_out=ds)
    assert rel_err(dq2.float(), res[0][0]) < 1e-6   # dQ from the stored dS^T either way
    assert rel_err(ds[..., :T].float().transpose(2, 3), res[0][1][..., :T]) < 1e-2
    assert float(ds[..., T:].abs().max()) == 0.0
    # ... and the table gradients reduced from two "layers" of it equal those from the fp32 accumulator
    w = [torch.zeros(nh, 32, device=DEV), torch.zeros(nh, 64, device=DEV), torch.zeros(nh, 64, device=DEV)]
    w2 = [torch.zeros_like(t) for t in w]
    gen = torch.Generator().manual_seed(3)
    bk = [torch.randint(0, n, (B, T, T), generator=gen, dtype=torch.uint8).to(DEV) for n in (32, 64, 64)]
    ops.relpos_bias_bwd(2 * res[0][1], bk[0], bk[1], bk[2], w[0], w[1], w[2], 0.3)
    ds2 = torch.stack([ds, ds]).contiguous()
    bkt = [t.transpose(1, 2).contiguous() for t in bk]
    ops.relpos_bias_bwd_layers(ds2, bkt[0], bkt[1], bkt[2], w2[0], w2[1], w2[2], 0.3)
    for a_, b_ in zip(w, w2):
        assert rel_err(b_, a_) < 1e-2


@pytest.mark.parametrize("B,nh,T,d", [(2, 3, 200, 64), (1, 2, 129, 80), (2, 2, 70, 32)])
def test_attention_dropout_mask_of_the_forward_is_the_mask_of_the_backward(ops, B, nh, T, d):
    """The forward's keep mask is read off with one-hot V blocks (out = dropped probabilities), its rate is checked, and an
    autograd statement of softmax -> that mask -> PV must give the gradients of the fused backward (same seed)."""
    p_drop, seed = 0.2, 1234
    H = nh * d
    g = torch.Generator().manual_seed(5)
    qkv = (0.7 * torch.randn(B * T, 3 * H, generator=g)).to(DEV).to(torch.bfloat16)
    bias = _padded_bias((0.5 * torch.randn(B, nh, T, T, generator=g)).to(DEV).to(torch.bfloat16), None)
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    scale = 0.11
    dropped = torch.zeros(B, nh, T, T, device=DEV)
    for j in range(0, T, d):
        vj = torch.zeros(B, T, nh, d, device=DEV)
        n = min(d, T - j)
        vj[:, j:j + n] = torch.eye(d, device=DEV)[:n].view(1, n, 1, d)
        qkvj = qkv.clone()
        qkvj[:, 2 * H:] = vj.view(B * T, H).to(torch.bfloat16)
        oj, _ = ops.attn_fwd(qkvj[:, :H], qkvj[:, H:2 * H], qkvj[:, 2 * H:], B, nh, T, d, scale, bias, None, drop_p=p_drop,
                             drop_seed=seed)
        dropped[..., j:j + n] = oj.float().view(B, T, nh, d).permute(0, 2, 1, 3)[..., :n]
    keep = dropped != 0          # probabilities are > 0 everywhere (no masked keys here)
    rate = float(keep.float().mean())
    assert abs(rate - (1 - p_drop)) < 4 * math.sqrt(p_drop * (1 - p_drop) / keep.numel()) + 1e-4, rate
    # neighbouring keys decide independently
    n2 = T // 2 * 2
    assert 0.05 < float((keep[..., 0:n2:2] != keep[..., 1:n2:2]).float().mean()) < 2 * p_drop
    # a different seed gives a different mask, the same seed the same one
    o1, lse = ops.attn_fwd(q, k, v, B, nh, T, d, scale, bias, None, drop_p=p_drop, drop_seed=seed)
    o2, _ = ops.attn_fwd(q, k, v, B, nh, T, d, scale, bias, None, drop_p=p_drop, drop_seed=seed)
    o3, _ = ops.attn_fwd(q, k, v, B, nh, T, d, scale, bias, None, drop_p=p_drop, drop_seed=seed + 1)
    assert torch.equal(o1, o2) and not torch.equal(o1, o3)
    # autograd with the extracted mask
    leaf = qkv.float().clone().requires_grad_(True)
    hd = lambda t: t.view(B, T, nh, d).permute(0, 2, 1, 3)
    sc = hd(leaf[:, :H]) @ hd(leaf[:, H:2 * H]).transpose(-1, -2) * scale + bias[..., :T].float()
    pr = torch.softmax(sc, -1) * keep.float() / (1 - p_drop)
    ref = (pr @ hd(leaf[:, 2 * H:])).permute(0, 2, 1, 3).reshape(B * T, H)
    assert rel_err(o1, ref) < 2e-2
    d_out = torch.randn(B * T, H, generator=g).to(DEV).to(torch.bfloat16)
    ref.backward(d_out.float())
    for single in (True, False):
        dqkv = torch.zeros_like(qkv)
        ops.attn_bwd(q, k, v, o1, d_out, lse, B, nh, T, d, scale, bias, None, dqkv, None, drop_p=p_drop, drop_seed=seed,
                     single_pass=single)
        assert rel_err(dqkv, leaf.grad) < 4e-2, (single, rel_err(dqkv, leaf.grad))


@pytest.mark.parametrize("B,nh,T,d", [(2, 2, 70, 16), (2, 2, 200, 64), (4, 12, 1389, 64), (1, 3, 17, 32), (2, 2, 129, 80), (1, 2, 257, 128)])
def test_attention_keep_words_are_the_forward_mask_in_both_precisions(ops, B, nh, T, d):
    """peneo_attn_drop_words defines the mask: bit (q & 31) of words[b * nh + h][q >> 5][kslot(key)].  The forward's mask is read
    off with one-hot V blocks for the fp32 and the bf16 kernel and must be exactly that bit map (the two kernels schedule the
    SGPR mask moves differently: a missing hazard pad between v_readlane and the select once gave the fp32 kernel wrong bits).
    The last case has 528 workgroups: the bf16 forward then runs its 32-key tiles (three workgroups per CU)."""
    p_drop = 0.2
    H = nh * d
    g = torch.Generator().manual_seed(5)
    qkv = (0.7 * torch.randn(B * T, 3 * H, generator=g)).to(DEV)
    Tp = ops.attn_padded_len(T)
    w = ops.attn_drop_words(B, nh, T, p_drop, 77)[0]
    kb = torch.zeros(B, Tp, device=DEV)
    kb[:, T:] = -1e30
    Tk = w.shape[-1]
    kslot = lambda k: (k & ~31) | (((k >> 3) & 3) << 3) | ((k & 3) << 1) | ((k >> 2) & 1)
    ks = torch.tensor([kslot(k) for k in range(T)], device=DEV)
    q = torch.arange(T, device=DEV)
    ww = w.view(B, nh, -1, Tk).long() & 0xFFFFFFFF
    ref = ((ww[:, :, (q >> 5)][:, :, :, ks] >> (q & 31).view(1, 1, T, 1)) & 1).bool()
    assert abs(float(ref.float().mean()) - (1 - p_drop)) < 4 * math.sqrt(p_drop * (1 - p_drop) / ref.numel()) + 2e-3
    for dt in (torch.float32, torch.bfloat16):
        dropped = torch.zeros(B, nh, T, T, device=DEV)
        for j in range(0, T, d):
            vj = torch.zeros(B, T, nh, d, device=DEV)
            n = min(d, T - j)
            vj[:, j:j + n] = torch.eye(d, device=DEV)[:n].view(1, n, 1, d)
            x = qkv.clone()
            x[:, 2 * H:] = vj.view(B * T, H)
            x = x.to(dt)
            o, _ = ops.attn_fwd(x[:, :H], x[:, H:2 * H], x[:, 2 * H:], B, nh, T, d, 0.2, None, kb, drop_p=p_drop, drop_words=w)
            dropped[..., j:j + n] = o.float().view(B, T, nh, d).permute(0, 2, 1, 3)[..., :n]
        assert torch.equal(dropped != 0, ref), dt


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_key_bias_only(ops, dtype):
    """LiLT-style: no bias tensor, padding mask as an additive per-key row."""
    B, nh, T, d = 2, 2, 100, 32
    g = torch.Generator().manual_seed(4)
    H = nh * d
    qkv = torch.randn(B * T, 3 * H, generator=g).to(DEV).to(dtype)
    mask = torch.ones(B, T, dtype=torch.int32)
    mask[1, 60:] = 0
    mask = mask.to(DEV)
    kb = torch.zeros(B, 128, device=DEV)
    kb[:, :T].masked_fill_(mask == 0, -1.0e30)
    leaf = qkv.float().clone().requires_grad_(True)
    hd = lambda t: t.view(B, T, nh, d).permute(0, 2, 1, 3)
    ref = _attn_ref(hd(leaf[:, :H]), hd(leaf[:, H:2 * H]), hd(leaf[:, 2 * H:]), None, mask, 0.2)
    ref2d = ref.permute(0, 2, 1, 3).reshape(B * T, H)
    out, lse = ops.attn_fwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, T, d, 0.2, None, kb)
    assert rel_err(out, ref2d) < tol(dtype)
    d_out = torch.randn(B * T, H, generator=g).to(DEV).to(dtype)
    ref2d.backward(d_out.float())
    dqkv = torch.empty_like(qkv)
    ops.attn_bwd(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], out, d_out, lse, B, nh, T, d, 0.2, None, kb, dqkv, None)
    assert rel_err(dqkv, leaf.grad) < (3e-5 if dtype == torch.float32 else 4e-2)


@pytest.mark.parametrize("B,nh,T,drop", [(1, 2, 709, 0.0), (2, 3, 709, 0.1), (2, 2, 200, 0.2), (1, 1, 64, 0.1), (2, 2, 33, 0.0),
                                         (1, 16, 1221, 0.1), (1, 2, 129, 0.1), (33, 16, 64, 0.1), (8, 12, 709, 0.1)])
def test_attention_fwd_pipelined_kernel_is_the_staged_kernel_bit_for_bit(ops, B, nh, T, drop):
    """attn_fwd_pipe.hip (LDS-DMA ring, one barrier per 32-key tile; what a bf16 call with a bias tensor at head dim 64 runs) against
    attn_fwd_kernel, which the same call still runs when it hands over a transposed copy of V: the same online softmax over 32-key
    blocks in the same order, so the output and lse must be IDENTICAL (attn_fwd_kernel itself is held to fp32 autograd by
    test_attention_fwd_bwd, and so is the new kernel through that test's (1, 2, 709, 64) case).  Ragged tails, masked keys,
    dropout words, grids below and above the resident slots."""
    d, H = 64, nh * 64
    g = torch.Generator().manual_seed(T + int(drop * 100) + 1)
    qkv = torch.randn(B * T, 3 * H, generator=g).to(DEV).to(torch.bfloat16)
    Tp = ops.attn_padded_len(T)
    bias = torch.full((B, nh, T, Tp), -1.0e30, dtype=torch.bfloat16, device=DEV)
    bias[..., :T] = (0.5 * torch.randn(B, nh, T, T, generator=g)).to(DEV).to(torch.bfloat16)
    bias[0, :, :, T // 3: T // 2] = -1.0e30
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    w = ops.attn_drop_words(B, nh, T, drop, 5)[0] if drop > 0 else None
    out_new, lse_new = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, None, drop_p=drop, drop_words=w)
    out_old, lse_old = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, None, drop_p=drop, drop_words=w,
                                    vt=ops.head_transpose(v, B, nh, T, d))
    assert bool(torch.isfinite(out_new.float()).all())
    assert torch.equal(out_new, out_old), "attention output"
    assert torch.equal(lse_new, lse_old), "lse"
    if T % 32:
        # peneo_attn_fwd's contract says nothing about the padding columns of the bias: keys >= T are masked by the kernel itself
        # (round 6), so zeros or NaNs there change nothing
        for pad in (0.0, float("nan")):
            b2 = bias.clone()
            b2[..., T:] = pad
            out_pad, lse_pad = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, b2, None, drop_p=drop, drop_words=w)
            assert torch.equal(out_pad, out_old) and torch.equal(lse_pad, lse_old), f"padding columns = {pad}"


@pytest.mark.parametrize("B,nh,T,drop", [(1, 2, 709, 0.0), (2, 3, 709, 0.1), (2, 2, 200, 0.2), (1, 1, 64, 0.1), (2, 2, 33, 0.0),
                                         (1, 16, 1221, 0.1), (1, 2, 129, 0.1), (33, 16, 64, 0.1), (20, 16, 140, 0.1),
                                         (20, 10, 300, 0.0), (8, 12, 709, 0.1)])
def test_attention_bwd_pipelined_kernel_is_the_fused_kernel_bit_for_bit(ops, B, nh, T, drop):
    """attn_bwd_pipe.hip (LDS-DMA ring, one barrier per 32-query tile; what a bias + dS^T-slab call at head dim 64 runs) against
    attn_bwd_fused_kernel, which a call that also asks for the fp32 bias gradient still runs: same arithmetic in the same order, so
    dq | dk | dv and the dS^T slab must be IDENTICAL (the fused kernel itself is held to fp32 autograd by test_attention_fwd_bwd).
    Ragged tails in both directions (T = 709 / 1221 / 129 / 33 / 140 / 300), masked keys, key blocks whose last waves lie wholly past
    T (idle waves that only serve the DMA stream), the padding columns of the slab, grids below and above the 512 resident slots
    (33 x 16 heads; the benchmark's 8 x 12 at T = 709: 576 workgroups)."""
    d, H = 64, nh * 64
    g = torch.Generator().manual_seed(T + int(drop * 100))
    qkv = torch.randn(B * T, 3 * H, generator=g).to(DEV).to(torch.bfloat16)
    Tp = ops.attn_padded_len(T)
    bias = torch.full((B, nh, T, Tp), -1.0e30, dtype=torch.bfloat16, device=DEV)
    bias[..., :T] = (0.5 * torch.randn(B, nh, T, T, generator=g)).to(DEV).to(torch.bfloat16)
    bias[0, :, :, T // 3: T // 2] = -1.0e30
    q, k, v = qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:]
    out, lse = ops.attn_fwd(q, k, v, B, nh, T, d, 0.125, bias, None, drop_p=drop, drop_seed=5)
    d_out = torch.randn(B * T, H, generator=g).to(DEV).to(torch.bfloat16)
    res = []
    for fused in (True, False):
        dqkv = torch.full_like(qkv, 3.0)
        ds = torch.full((B, nh, T, Tp), 7.0, device=DEV, dtype=torch.bfloat16)
        gb = torch.zeros(bias.shape, dtype=torch.float32, device=DEV) if fused else None
        ops.attn_bwd(q, k, v, out, d_out, lse, B, nh, T, d, 0.125, bias, None, dqkv, gb, drop_p=drop, drop_seed=5, ds_out=ds)
        res.append((dqkv, ds))
    assert bool(torch.isfinite(res[1][0].float()).all())
    assert torch.equal(res[0][0], res[1][0]), "dq | dk | dv"
    assert torch.equal(res[0][1], res[1][1]), "dS^T slab"
    if Tp > T:
        assert float(res[1][1][..., T:].float().abs().max()) == 0.0


def test_head_transpose(ops):
    B, nh, T, d = 2, 3, 70, 30
    x = torch.randn(B * T, nh * d + 6, device=DEV)
    xt = ops.head_transpose(x[:, :nh * d], B, nh, T, d)
    assert xt.shape == (B, nh, 32, 128)
    ref = x[:, :nh * d].view(B, T, nh, d).permute(0, 2, 3, 1)
    assert torch.equal(xt[:, :, :d, :T], ref)
    assert float(xt[:, :, d:].abs().max()) == 0 and float(xt[..., T:].abs().max()) == 0


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nh,da,db", [(4, 48, 12), (12, 64, 16), (24, 64, 16)])   # scalar form; 16-byte form (LiLT-base q, k+v)
def test_head_concat_split(ops, dtype, nh, da, db):
    """LiLT's [text | layout] per-head packing and its inverse, on strided slices of fused buffers."""
    R = 77
    g = torch.Generator().manual_seed(9)
    a = torch.randn(R, 3 * nh * da, generator=g).to(DEV).to(dtype)
    b = torch.randn(R, 3 * nh * db, generator=g).to(DEV).to(dtype)
    out = torch.zeros(R, 2 * nh * (da + db), device=DEV, dtype=dtype)
    av, bv = a[:, nh * da:2 * nh * da], b[:, nh * db:2 * nh * db]
    ops.head_concat(av, bv, nh, out[:, nh * (da + db):], 0.5, 2.0)
    ref = torch.cat([0.5 * av.float().view(R, nh, da), 2.0 * bv.float().view(R, nh, db)], dim=2).reshape(R, -1)
    assert torch.equal(out[:, nh * (da + db):].float(), ref.to(dtype).float())
    assert float(out[:, :nh * (da + db)].abs().max()) == 0
    a2 = torch.zeros(R, nh * da + 8, device=DEV, dtype=dtype)
    b2 = torch.zeros(R, nh * db, device=DEV, dtype=dtype)
    ops.head_split(out[:, nh * (da + db):], nh, a2[:, :nh * da], b2, 2.0, 0.5)
    assert torch.equal(a2[:, :nh * da], av) and torch.equal(b2, bv)
    assert float(a2[:, nh * da:].abs().max()) == 0


@pytest.mark.parametrize("mode", [256, 384, 128])
@pytest.mark.parametrize("bk", [True, False])
def test_gemm_big_tiles_match_fp32_matmul(ops, mode, bk):
    """gemm_big.hip (one 8-wave workgroup per CU; 256 x 256 / 384 x 192 / 256 x 128 tiles) forced for a ragged problem
    (M, N not multiples of any tile; 5 and 13 k-tiles: ring wrap-around and the short-loop prologue), both B layouts, with
    every fused epilogue option, against an fp32 matmul of the same bf16 operands and against the 128 x 128 kernel."""
    import ctypes
    from peneo_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    g = torch.Generator().manual_seed(mode + bk)
    dt = torch.bfloat16
    try:
        for (M, N, K) in [(2101, 1032, 320), (2800, 776, 832), (3000, 2304, 128)]:
            a = torch.randn(M, K, generator=g).to(DEV).to(dt)
            b = (torch.randn(N, K, generator=g) if bk else torch.randn(K, N, generator=g)).to(DEV).to(dt)
            bias = torch.randn(N, generator=g).to(DEV)
            res = torch.randn(M, N, generator=g).to(DEV).to(dt)
            src = torch.randn(M, N, generator=g).to(DEV).to(dt)
            z = a.float() @ (b.float().t() if bk else b.float()) / math.sqrt(K) * math.sqrt(K)
            outs = {}
            for m_ in (mode, 0):
                lib.peneo_gemm_set_big_mode(m_)
                pre = torch.empty(M, N, device=DEV, dtype=dt)
                o1 = ops.gemm(a, b, b_kmajor=bk, bias=bias, act=1, preact=pre)                       # bias + GELU + pre-activation
                o2 = ops.gemm(a, b, b_kmajor=bk, bias=bias, residual=res, drop_p=0.1, drop_seed=77)   # bias + dropout + residual
                o3 = ops.gemm(a, b, b_kmajor=bk, grad_src=src, grad_act=2)                            # x SiLU'(src)
                o4 = ops.gemm(a, b, b_kmajor=bk, out_dtype=torch.float32)
                outs[m_] = (pre, o1, o2, o3, o4)
            pre, o1, o2, o3, o4 = outs[mode]
            assert rel_err(pre, z + bias) < 2e-2 and rel_err(o1, F.gelu(z + bias)) < 2e-2
            sg = torch.sigmoid(src.float())
            assert rel_err(o3, z * (sg * (1 + src.float() * (1 - sg)))) < 2e-2
            assert rel_err(o4, z) < 2e-3
            kept = (o2.float() - res.float()).abs() > 0
            assert 0.85 < float(kept.float().mean()) < 0.95
            # same dropout mask function and same arithmetic as the 128 x 128 kernel: only the summation order differs
            for x, y in zip(outs[mode], outs[0]):
                assert rel_err(x, y) < 1e-2
            assert torch.equal((outs[0][2].float() - res.float()) != 0, kept)
    finally:
        lib.peneo_gemm_set_big_mode(1)


@pytest.mark.parametrize("mode", [5128, 4256, 6256, 8128, 7128, 105128, 105256, 104128, 108128])
@pytest.mark.parametrize("bk", [True])
def test_gemm_stream_k_matches_fp32_matmul(ops, mode, bk):
    """gemm_sk.hip (one persistent launch, every workgroup a contiguous range of (tile, k-stage) units; mode = F * 1000 + N extent of
    the tile, + 100000 for ranges that cut tiles = stream-k) forced for ragged problems: tiles cut once, twice and three times by
    range boundaries (few tiles x many k-stages), ranges of one unit, M / N that are no multiple of the tile, every fused epilogue
    option -- against an fp32 matmul of the same bf16 operands and
    against the 128 x 128 kernel (same dropout mask function, same epilogue arithmetic: only the summation order differs).
    Every problem is launched three times with fresh operands: the slab flags are reset by their consumers, so a stale flag or a
    stale slab from the launch before shows as a wrong tile."""
    import ctypes
    from peneo_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    g = torch.Generator().manual_seed(mode + bk)
    dt = torch.bfloat16
    try:
        for (M, N, K) in [(2101, 1032, 320), (2800, 776, 832), (300, 136, 128), (1500, 768, 3072), (5672, 768, 768), (777, 264, 1920)]:
            for rep in range(3):
                a = torch.randn(M, K, generator=g).to(DEV).to(dt)
                b = (torch.randn(N, K, generator=g) if bk else torch.randn(K, N, generator=g)).to(DEV).to(dt)
                bias = torch.randn(N, generator=g).to(DEV)
                res = torch.randn(M, N, generator=g).to(DEV).to(dt)
                src = torch.randn(M, N, generator=g).to(DEV).to(dt)
                z = a.float() @ (b.float().t() if bk else b.float())
                outs = {}
                for m_ in (mode, 0):
                    lib.peneo_gemm_set_sk_mode(m_)
                    lib.peneo_gemm_set_big_mode(1 if m_ else 0)
                    pre = torch.empty(M, N, device=DEV, dtype=dt)
                    o1 = ops.gemm(a, b, b_kmajor=bk, bias=bias, act=1, preact=pre, split_k=1)                       # bias + GELU + pre-activation
                    o2 = ops.gemm(a, b, b_kmajor=bk, bias=bias, residual=res, drop_p=0.1, drop_seed=77, split_k=1)   # bias + dropout + residual
                    o3 = ops.gemm(a, b, b_kmajor=bk, grad_src=src, grad_act=2, split_k=1)                            # x SiLU'(src)
                    o4 = ops.gemm(a, b, b_kmajor=bk, out_dtype=torch.float32, split_k=1)
                    o5 = ops.gemm(a, b, b_kmajor=bk, bias=bias, split_k=1)                                            # bias only (compact epilogue)
                    o6 = ops.gemm(a, b, b_kmajor=bk, bias=bias, residual=res, split_k=1)                              # bias + residual, no dropout
                    outs[m_] = (pre, o1, o2, o3, o4, o5, o6)
                pre, o1, o2, o3, o4, o5, o6 = outs[mode]
                assert rel_err(o5, z + bias) < 2e-2 and rel_err(o6, z + bias + res.float()) < 2e-2, (M, N, K, rep)
                assert rel_err(pre, z + bias) < 2e-2 and rel_err(o1, F.gelu(z + bias)) < 2e-2, (M, N, K, rep)
                sg = torch.sigmoid(src.float())
                assert rel_err(o3, z * (sg * (1 + src.float() * (1 - sg)))) < 2e-2, (M, N, K, rep)
                assert rel_err(o4, z) < 2e-3, (M, N, K, rep, rel_err(o4, z))
                kept = (o2.float() - res.float()).abs() > 0
                assert 0.85 < float(kept.float().mean()) < 0.95
                for x, y in zip(outs[mode], outs[0]):
                    assert rel_err(x, y) < 1e-2, (M, N, K, rep)
                # (a kept value that rounds into its residual reads as dropped: a handful of elements differ with the summation order)
                assert float((((outs[0][2].float() - res.float()) != 0) != kept).float().mean()) < 1e-4
    finally:
        lib.peneo_gemm_set_sk_mode(1)
        lib.peneo_gemm_set_big_mode(1)


def test_gemm_stream_k_hand_off_under_uneven_load(ops):
    """The stream-k slab hand-off (write-through slab stores, vm drain, flag; relaxed poll + one agent-scope acquire; flags reset by
    their consumer) checked the way the CDNA guide asks for: every word of every result, many launches in a row with fresh operands
    (a stale flag or a stale slab of the launch before shows), and beside a second stream that keeps part of the CUs busy with long
    workgroups, so that the ranges of a launch start at uneven times and a finisher meets slabs that are not there yet."""
    import ctypes
    from peneo_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    g = torch.Generator().manual_seed(99)
    dt = torch.bfloat16
    shapes = [(1500, 768, 3072), (2442, 1024, 4096), (777, 264, 1920), (5672, 768, 768)]
    ops_ab = {}
    for (M, N, K) in shapes:
        ops_ab[(M, N, K)] = [((torch.randn(M, K, generator=g)).to(DEV).to(dt), (torch.randn(N, K, generator=g) * 0.05).to(DEV).to(dt),
                              torch.randn(N, generator=g).to(DEV)) for _ in range(3)]
    side = torch.cuda.Stream()
    big_a = torch.randn(8192, 4096, generator=g).to(DEV).to(dt)
    big_b = torch.randn(4096, 4096, generator=g).to(DEV).to(dt)
    try:
        refs = {}
        lib.peneo_gemm_set_sk_mode(0)
        for key, sets in ops_ab.items():
            refs[key] = [ops.gemm(a, b, bias=bias, split_k=1).float() for a, b, bias in sets]
        torch.cuda.synchronize()
        bad = 0
        for it in range(60):
            if it % 4 == 0:
                with torch.cuda.stream(side):               # long tiled workgroups on another queue: uneven CU availability
                    lib.peneo_gemm_set_sk_mode(0)
                    ops.gemm(big_a, big_b, split_k=1)
            key = shapes[it % len(shapes)]
            a, b, bias = ops_ab[key][it % 3]
            lib.peneo_gemm_set_sk_mode(105128 if it % 2 == 0 else 104256)
            out = ops.gemm(a, b, bias=bias, split_k=1)
            err = rel_err(out, refs[key][it % 3])
            bad += int(not (err < 1e-2))
        torch.cuda.synchronize()
        assert bad == 0, f"{bad} of 60 stream-k launches differ from the tiled kernel"
    finally:
        lib.peneo_gemm_set_sk_mode(1)


def test_cast_multi_equals_single_casts(ops):
    """peneo_cast_multi (all weight copies of a step in one launch) == peneo_cast tensor by tensor, ragged sizes included."""
    g = torch.Generator().manual_seed(5)
    srcs = [torch.randn(n, generator=g).to(DEV) for n in (768 * 768, 16384, 16385, 7, 3072 * 768 + 24, 1)]
    whole = torch.zeros(sum(t.numel() for t in srcs) + 64, dtype=torch.bfloat16, device=DEV)
    dsts, off = [], 0
    for t in srcs:
        dsts.append(whole[off:off + t.numel()])
        off += (t.numel() + 7) // 8 * 8
    plan = ops.CastPlan(list(zip(srcs, dsts)))
    plan.run()
    for t, d in zip(srcs, dsts):
        assert torch.equal(d, t.to(torch.bfloat16))
    assert float(whole[off:].abs().max()) == 0
    srcs[2].mul_(3.0)
    plan.run()                                               # same tables, new values
    assert torch.equal(dsts[2], srcs[2].to(torch.bfloat16))


# ---------------------------------------------------------------------------------------------- pair heads
def _pair_ref(ab, w1, b1, w2, b2):
    B, N, D2 = ab.shape
    D = D2 // 2
    ii, jj = torch.triu_indices(N, N, device=ab.device)
    x = F.silu(ab[:, ii, :D] + ab[:, jj, D:])
    return [F.linear(F.silu(F.linear(x, a, b)), c, d) for a, b, c, d in zip(w1, b1, w2, b2)], x


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,D", [(39, 32), (71, 96), (130, 384), (70, 512), (33, 384)])
def test_pair_heads_fwd_and_loss(ops, dtype, N, D):
    B, classes = 2, [2, 3, 3, 3, 3]
    g = torch.Generator().manual_seed(N)
    ab = torch.randn(B, N, 2 * D, generator=g).to(DEV).to(dtype)
    w1 = [(torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV) for _ in classes]
    b1 = [0.1 * torch.randn(D, generator=g).to(DEV) for _ in classes]
    w2 = [(torch.randn(c, D, generator=g) / math.sqrt(D)).to(DEV) for c in classes]
    b2 = [0.1 * torch.randn(c, generator=g).to(DEV) for c in classes]
    rd = lambda t: t.to(dtype).float()
    ref, _ = _pair_ref(ab.float(), [rd(w) for w in w1], b1, [rd(w) for w in w2], b2)
    wp = ops.pair_heads_pack(dtype, w1, w2)
    P = N * (N + 1) // 2
    tags = [torch.randint(0, c, (B, P), generator=g).to(DEV) for c in classes]
    cw = [torch.tensor([1.0, 10.0, 10.0][:c], device=DEV) for c in classes]
    logits, partials, dlog = ops.pair_heads_fwd(ab, wp, torch.cat(b1), torch.cat(b2), classes, tags=tags,
                                                class_weights=cw, want_dlogits=True)
    tot = partials.sum(0)
    num, den = tot[:5], tot[8:13]
    out, scale, dls = ops.loss_finish(partials, torch.ones(5, device=DEV), 14)
    for h in range(5):
        assert rel_err(logits[h], ref[h]) < tol(dtype), (h, rel_err(logits[h], ref[h]))
        lr = logits[h].clone().requires_grad_(True)
        loss_sum = F.cross_entropy(lr.view(-1, classes[h]), tags[h].view(-1), weight=cw[h], reduction="sum")
        loss_sum.backward()
        wsum = cw[h][tags[h].view(-1)].sum()
        assert abs(float(num[h]) - float(loss_sum)) / float(loss_sum) < 1e-4
        assert abs(float(den[h]) - float(wsum)) / float(wsum) < 1e-5
        assert rel_err(dlog[h], lr.grad) < 1e-4
    assert rel_err(dls, torch.cat([d.sum((0, 1)) for d in dlog])) < 1e-3
    assert abs(float(out[5]) - float((num / den).sum())) < 1e-4 and rel_err(scale[0], 1.0 / den) < 1e-5 and rel_err(scale[1], 1.0 / den) < 1e-5
    # loss-only call (no logits written) agrees
    _, part2, _ = ops.pair_heads_fwd(ab, wp, torch.cat(b1), torch.cat(b2), classes, want_logits=False,
                                     tags=tags, class_weights=cw)
    assert rel_err(part2.sum(0)[:5], num) < 1e-5


@pytest.mark.parametrize("N,D", [(511, 384), (301, 512)])
def test_pair_heads_fwd_hand_kernel_is_repeatable_at_size(ops, N, D):
    """The hand-interleaved forward kernel (bf16, D = 384 / 512) issues its MFMAs as inline-asm statements: the compiler knows neither
    their latency nor that a partner wave shares the matrix pipe.  A hazard there shows up as launch-to-launch differences
    (as the packed-fp32 one did): three launches of the train-mode call on the same inputs must agree bit for bit in the dlogits and
    the loss partials, and the eval logits must agree with the fp32 torch computation."""
    B, classes, dtype = 2, [2, 3, 3, 3, 3], torch.bfloat16
    g = torch.Generator().manual_seed(N + D)
    ab = torch.randn(B, N, 2 * D, generator=g).to(DEV).to(dtype)
    P = N * (N + 1) // 2
    w1 = [(torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV) for _ in classes]
    w2 = [(torch.randn(c, D, generator=g) / math.sqrt(D)).to(DEV) for c in classes]
    b1, b2 = (0.1 * torch.randn(len(classes) * D, generator=g)).to(DEV), torch.randn(14, generator=g).to(DEV)
    wp = ops.pair_heads_pack(dtype, w1, w2)
    tags = [torch.randint(0, c, (B, P), generator=g).to(DEV) for c in classes]
    cw = [torch.tensor([1.0, 10.0, 10.0][:c], device=DEV) for c in classes]
    ref = None
    for _ in range(3):
        _, partials, dlog = ops.pair_heads_fwd(ab, wp, b1, b2, classes, want_logits=False, tags=tags, class_weights=cw, want_dlogits=True,
                                               drop_p=0.1, drop_seed=77)
        torch.cuda.synchronize()
        cur = [partials.clone()] + [d.clone() for d in dlog]
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(cur, ref))
    logits, _, _ = ops.pair_heads_fwd(ab, wp, b1, b2, classes)
    ii, jj = torch.triu_indices(N, N, device=DEV)
    xq = F.silu(ab.float()[:, ii, :D] + ab.float()[:, jj, D:]).to(dtype).float()
    off = 0
    for h, c in enumerate(classes):
        y = F.silu(F.linear(xq, w1[h].to(dtype).float(), b1[h * D:(h + 1) * D])).to(dtype).float()
        want = F.linear(y, w2[h].to(dtype).float(), b2[off:off + c])
        assert (logits[h] - want).abs().max() < 3e-2, h
        off += c


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,D", [(39, 32), (71, 96), (130, 384), (70, 512)])
def test_pair_heads_fwd_classifier_dropout(ops, dtype, N, D):
    """Train mode: Dropout(p) between the two classifier layers (model/peneo_decoder.py:261) inside the fused kernel.  The mask
    is a pure function of (seed, document, pair, hidden column), restated on the host (tests/dropout_ref.py): logits must
    equal the explicit computation with that mask, and the realised keep rate is 1 - round(p 2^16) / 2^16."""
    from dropout_ref import k12_keep, k12_scale
    B, classes, p_drop, seed = 2, [2, 3, 3, 3, 3], 0.1, 0xC0FFEE + N
    g = torch.Generator().manual_seed(N + 7)
    ab = torch.randn(B, N, 2 * D, generator=g).to(DEV).to(dtype)
    w1 = [(torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV) for _ in classes]
    b1 = [0.1 * torch.randn(D, generator=g).to(DEV) for _ in classes]
    w2 = [(torch.randn(c, D, generator=g) / math.sqrt(D)).to(DEV) for c in classes]
    b2 = [0.1 * torch.randn(c, generator=g).to(DEV) for c in classes]
    rd = lambda t: t.to(dtype).float()
    P = N * (N + 1) // 2
    keep = torch.stack([k12_keep(seed, b, 0, P, len(classes) * D, p_drop) for b in range(B)]).to(DEV)   # [B, P, nh*D]
    assert abs(float(keep.float().mean()) - (1 - 6554 / 65536)) < 4 / math.sqrt(keep.numel())
    ii, jj = torch.triu_indices(N, N, device=DEV)
    x = F.silu(ab.float()[:, ii, :D] + ab.float()[:, jj, D:])
    if dtype == torch.bfloat16:
        x = x.to(dtype).float()
    ref = []
    for h, (a, b_, c, d) in enumerate(zip(w1, b1, w2, b2)):
        y = F.silu(F.linear(x, rd(a), b_)) * keep[:, :, h * D:(h + 1) * D] * k12_scale(p_drop)
        ref.append(F.linear(y, rd(c), d))
    wp = ops.pair_heads_pack(dtype, w1, w2)
    logits, _, _ = ops.pair_heads_fwd(ab, wp, torch.cat(b1), torch.cat(b2), classes, drop_p=p_drop, drop_seed=seed)
    plain, _, _ = ops.pair_heads_fwd(ab, wp, torch.cat(b1), torch.cat(b2), classes)
    for h in range(5):
        assert rel_err(logits[h], ref[h]) < tol(dtype), (h, rel_err(logits[h], ref[h]))
        assert rel_err(plain[h], ref[h]) > 0.1              # the mask really changes the result
    again, _, _ = ops.pair_heads_fwd(ab, wp, torch.cat(b1), torch.cat(b2), classes, drop_p=p_drop, drop_seed=seed)
    other, _, _ = ops.pair_heads_fwd(ab, wp, torch.cat(b1), torch.cat(b2), classes, drop_p=p_drop, drop_seed=seed + 1)
    assert all(torch.equal(a_, b_) for a_, b_ in zip(logits, again)) and not torch.equal(logits[1], other[1])


@pytest.mark.parametrize("dtype", DTYPES)
def test_pair_backward_blocks(ops, dtype):
    N, D, classes = 45, 128, [2, 3, 3, 3, 3]
    g = torch.Generator().manual_seed(9)
    ab = torch.randn(N, 2 * D, generator=g).to(DEV).to(dtype)
    ii, jj = torch.triu_indices(N, N, device=DEV)
    i0, i1 = 7, 30
    p0, p1 = i0 * N - i0 * (i0 - 1) // 2, i1 * N - i1 * (i1 - 1) // 2
    npairs = p1 - p0
    abr = ab.float().clone().requires_grad_(True)
    xr = F.silu(abr[ii, :D] + abr[jj, D:])[p0:p1]
    x = torch.empty(npairs, D, device=DEV, dtype=dtype)
    ops.pair_x_fwd(ab, i0, i1, x)
    assert rel_err(x, xr) < tol(dtype)
    dx = torch.randn(npairs, D, generator=g).to(DEV).to(dtype)
    xr.backward(dx.float())
    dab = torch.zeros(N, 2 * D, device=DEV)
    ops.pair_x_bwd(ab, i0, i1, dx, dab)
    assert rel_err(dab, abr.grad) < (1e-4 if dtype == torch.float32 else 2e-2)
    # split form used by the model: pre = a_i + b_j stored, SiLU' applied elsewhere (GEMM epilogue), plain segmented sums here
    pre = torch.empty_like(x)
    x2 = torch.empty_like(x)
    ops.pair_x_fwd(ab, i0, i1, x2, pre)
    assert torch.equal(x2, x)
    prr = (ab.float()[ii, :D] + ab.float()[jj, D:])[p0:p1]
    assert rel_err(pre, prr) < tol(dtype)
    sg = torch.sigmoid(pre.float())
    du = (dx.float() * (sg * (1 + pre.float() * (1 - sg)))).to(dtype)
    dab2 = torch.zeros(N, 2 * D, device=DEV)
    ops.pair_x_bwd(ab, i0, i1, du, dab2, premultiplied=True)
    assert rel_err(dab2, abr.grad) < (1e-4 if dtype == torch.float32 else 3e-2)
    # dz block
    nh = len(classes)
    z = torch.randn(npairs, nh * D, generator=g).to(DEV).to(dtype)
    w2 = [torch.randn(c, D, generator=g).to(DEV) for c in classes]
    dl = [torch.randn(npairs, c, generator=g).to(DEV) for c in classes]
    scale = torch.rand(nh, generator=g).to(DEV) + 0.5
    zr = z.float().clone().requires_grad_(True)
    w2r = [w.clone().requires_grad_(True) for w in w2]
    tot = 0
    for h in range(nh):
        y = F.silu(zr[:, h * D:(h + 1) * D])
        tot = tot + ((y @ w2r[h].t()) * dl[h] * scale[h]).sum()
    tot.backward()
    zz = z.clone()
    ws = ops.pair_dz_workspace(nh, D, DEV)
    ops.pair_dz(zz, npairs, D, classes, dl, w2, ws, scale)
    dw2, db1 = ops.pair_dz_finish(ws, nh, D, classes)
    t = 1e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(zz, zr.grad) < t
    assert rel_err(db1, zr.grad.sum(0)) < t
    for h in range(nh):
        assert rel_err(dw2[h], w2r[h].grad) < t
    # ... with the forward's classifier dropout (document 3, chunk starting at pair p0): y and dz masked and scaled
    from dropout_ref import k12_keep, k12_scale
    keep = (k12_keep(77, 3, p0, p1, nh * D, 0.1).to(DEV) * k12_scale(0.1)).float()
    zr2 = z.float().clone().requires_grad_(True)
    w2r2 = [w.clone().requires_grad_(True) for w in w2]
    tot = 0
    for h in range(nh):
        y = F.silu(zr2[:, h * D:(h + 1) * D]) * keep[:, h * D:(h + 1) * D]
        tot = tot + ((y @ w2r2[h].t()) * dl[h] * scale[h]).sum()
    tot.backward()
    zd = z.clone()
    ws_d = ops.pair_dz_workspace(nh, D, DEV)
    ops.pair_dz(zd, npairs, D, classes, None, None, ws_d, None,
                args=ops.pair_dz_args(D, classes, dl, w2, scale, drop_p=0.1, drop_seed=77, drop_doc=3, drop_pair0=p0))
    dw2d, db1d = ops.pair_dz_finish(ws_d, nh, D, classes)
    assert rel_err(zd, zr2.grad) < t and rel_err(db1d, zr2.grad.sum(0)) < t
    for h in range(nh):
        assert rel_err(dw2d[h], w2r2[h].grad) < t
    # the same block fused into the epilogue of the z GEMM: z = x W1^T + b1 -> dz, with the dW2 / db1 partial sums
    w1cat = (torch.randn(nh * D, D, generator=g) / math.sqrt(D)).to(DEV).to(dtype)
    b1cat = (0.1 * torch.randn(nh * D, generator=g)).to(DEV)
    zfull = ops.gemm(x, w1cat, bias=b1cat)
    ws_a = ops.pair_dz_workspace(nh, D, DEV)
    za = zfull.clone()
    ops.pair_dz(za, npairs, D, classes, dl, w2, ws_a, scale)
    ws_b = ops.pair_dz_workspace(nh, D, DEV)
    zb = torch.empty_like(zfull)
    ops.gemm(x, w1cat, bias=b1cat, out=zb, pair_dz=ops.pair_dz_args(D, classes, dl, w2, scale), pair_dz_ws=ws_b)
    # unfused path rounds z to the storage dtype before the activation derivative; fused keeps fp32 -> bf16-level agreement
    assert rel_err(zb, za) < (1e-5 if dtype == torch.float32 else 3e-2)
    dw2a, db1a = ops.pair_dz_finish(ws_a, nh, D, classes)
    dw2b, db1b = ops.pair_dz_finish(ws_b, nh, D, classes)
    assert rel_err(db1b, db1a) < (1e-4 if dtype == torch.float32 else 2e-2)
    for h in range(nh):
        assert rel_err(dw2b[h], dw2a[h]) < (1e-4 if dtype == torch.float32 else 2e-2)


@pytest.mark.parametrize("N,D,rows", [(45, 128, (7, 30)), (45, 128, (0, 45)), (70, 384, (0, 70)), (33, 64, (30, 33)),
                                      (40, 96, (5, 6)), (37, 512, (0, 37)), (20, 32, (0, 20))])
def test_pair_dz_fused_matches_the_gemm_epilogue(ops, N, D, rows):
    """peneo_pair_dz_fused (x and z never in memory) against pair_x_fwd + the z GEMM with the pair-dz epilogue."""
    dtype, classes = torch.bfloat16, [2, 3, 3, 3, 3]
    nh = len(classes)
    g = torch.Generator().manual_seed(N * 1000 + D)
    ab = torch.randn(N, 2 * D, generator=g).to(DEV).to(dtype)
    i0, i1 = rows
    p0, p1 = i0 * N - i0 * (i0 - 1) // 2, i1 * N - i1 * (i1 - 1) // 2
    npairs = p1 - p0
    w1 = [(torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV) for _ in classes]
    w2 = [torch.randn(c, D, generator=g).to(DEV) for c in classes]
    b1cat = (0.1 * torch.randn(nh * D, generator=g)).to(DEV)
    dl = [torch.randn(npairs, c, generator=g).to(DEV) for c in classes]
    scale = torch.rand(nh, generator=g).to(DEV) + 0.5
    w1cat = torch.cat(w1).to(dtype)
    wp = ops.pair_heads_pack(dtype, w1, w2)
    args = ops.pair_dz_args(D, classes, dl, w2, scale)
    x = torch.empty(npairs, D, device=DEV, dtype=dtype)
    ops.pair_x_fwd(ab, i0, i1, x)
    ws_a = ops.pair_dz_workspace(nh, D, DEV)
    za = torch.empty(npairs, nh * D, device=DEV, dtype=dtype)
    ops.gemm(x, w1cat, bias=b1cat, out=za, pair_dz=args, pair_dz_ws=ws_a)
    ws_b = ops.pair_dz_workspace(nh, D, DEV)
    zb = torch.full((npairs + 3, nh * D), 7.0, device=DEV, dtype=dtype)     # guard rows behind the chunk
    ops.pair_dz_fused(ab, i0, i1, wp, b1cat, args, zb, ws_b)
    torch.cuda.synchronize()
    assert bool((zb[npairs:] == 7.0).all())
    # the same call can leave x and a_i + b_j (what pair_x_fwd writes) for the dW1 / dx GEMMs
    pre = torch.empty_like(x)
    ops.pair_x_fwd(ab, i0, i1, x, pre)
    zc = torch.empty_like(za)
    xc = torch.full((npairs + 2, D), 3.0, device=DEV, dtype=dtype)
    pc = torch.full((npairs + 2, D), 3.0, device=DEV, dtype=dtype)
    ops.pair_dz_fused(ab, i0, i1, wp, b1cat, args, zc, ops.pair_dz_workspace(nh, D, DEV), xc, pc)
    assert torch.equal(zc, zb[:npairs]) and torch.equal(xc[:npairs], x) and torch.equal(pc[:npairs], pre)
    assert bool((xc[npairs:] == 3.0).all()) and bool((pc[npairs:] == 3.0).all())
    assert rel_err(zb[:npairs], za) < 2e-2
    dw2a, db1a = ops.pair_dz_finish(ws_a, nh, D, classes)
    dw2b, db1b = ops.pair_dz_finish(ws_b, nh, D, classes)
    assert rel_err(db1b, db1a) < 2e-2
    for h in range(nh):
        assert rel_err(dw2b[h], dw2a[h]) < 2e-2
    # and against fp32 autograd of the same block
    abf = ab.float()
    ii, jj = torch.triu_indices(N, N, device=DEV)
    xr = F.silu(abf[ii, :D] + abf[jj, D:])[p0:p1].to(dtype).float()
    zr = (xr @ w1cat.float().t() + b1cat).requires_grad_(True)
    w2r = [w.clone().requires_grad_(True) for w in w2]
    tot = 0
    for h in range(nh):
        tot = tot + ((F.silu(zr[:, h * D:(h + 1) * D]) @ w2r[h].t()) * dl[h] * scale[h]).sum()
    tot.backward()
    assert rel_err(zb[:npairs], zr.grad) < 2e-2
    assert rel_err(db1b, zr.grad.sum(0)) < 2e-2
    for h in range(nh):
        assert rel_err(dw2b[h], w2r[h].grad) < 2e-2


@pytest.mark.parametrize("layout", ["nn_t", "nt", "tn"])
def test_gemm_group_matches_single_launches(ops, layout):
    """peneo_gemm_group: several problems in one launch == the same problems one by one (and fp32 matmul)."""
    g = torch.Generator().manual_seed(11)
    dt = torch.bfloat16
    T = 1000                                               # ragged against the 128 tile and the 64 k-tile
    shapes = [(384, 256), (128, 128), (520, 136), (72, 264)]   # (rows of out, cols of out)
    probs, refs = [], []
    for (m, n) in shapes:
        if layout == "nn_t":      # wgrad: out[m, n] = dy^T x, both operands stored [tokens, *]
            A = torch.randn(T, m, generator=g).to(DEV).to(dt); B = torch.randn(T, n, generator=g).to(DEV).to(dt)
            ref = A.float().t() @ B.float(); kw = dict(a_kmajor=False, b_kmajor=False)
        elif layout == "nt":      # forward: out = x W^T
            A = torch.randn(m, T, generator=g).to(DEV).to(dt); B = torch.randn(n, T, generator=g).to(DEV).to(dt)
            ref = A.float() @ B.float().t(); kw = dict(a_kmajor=True, b_kmajor=True)
        else:                     # dgrad: out = dy W
            A = torch.randn(m, T, generator=g).to(DEV).to(dt); B = torch.randn(T, n, generator=g).to(DEV).to(dt)
            ref = A.float() @ B.float(); kw = dict(a_kmajor=True, b_kmajor=False)
        probs.append((A, B, torch.full((m, n), 5.0, device=DEV, dtype=torch.float32)))
        refs.append(ref)
    for k in (1, 3, 4):
        for _, _, o in probs: o.fill_(5.0)
        ops.gemm_group(probs[:k], **kw)
        for i, ((A, B, o), ref) in enumerate(zip(probs, refs)):
            if i < k:
                single = ops.gemm(A, B, out_dtype=torch.float32, **kw)
                assert rel_err(o, ref) < 2e-3 and rel_err(o, single) < 1e-5
            else:
                assert bool((o == 5.0).all())
    ops.gemm_group(probs, accumulate=True, **kw)            # out += A B
    for (A, B, o), ref in zip(probs, refs):
        assert rel_err(o, 2 * ref) < 2e-3
    with pytest.raises(Exception):
        ops.gemm_group(probs + probs[:1], **kw)             # more than 4 problems


def test_gemm_group_with_per_problem_epilogues(ops):
    """Two Linear layers of different width in ONE launch, each with its own fused epilogue (LiLT's text + layout streams,
    modeling_lilt.py:269-429): bias + GELU + pre-activation store on one, bias + dropout + residual on the other; then the
    dgrad pair with x GELU'(src) / + residual.  Each problem must equal its single peneo_gemm launch (to one bf16 step)."""
    from peneo_amd.hip import ACT_GELU
    g = torch.Generator().manual_seed(12)
    dt = torch.bfloat16
    R = 1000
    x = torch.randn(R, 768, generator=g).to(DEV).to(dt); l = torch.randn(R, 192, generator=g).to(DEV).to(dt)
    wi = (torch.randn(3072, 768, generator=g) * 0.05).to(DEV).to(dt); lwi = (torch.randn(768, 192, generator=g) * 0.05).to(DEV).to(dt)
    bi = torch.randn(3072, generator=g).to(DEV); lbi = torch.randn(768, generator=g).to(DEV)
    res = torch.randn(R, 768, generator=g).to(DEV).to(dt)
    zi, lzi = torch.empty(R, 3072, device=DEV, dtype=dt), torch.empty(R, 768, device=DEV, dtype=dt)
    o1, o2 = ops.gemm_group([(x, wi, None, dict(bias=bi, act=ACT_GELU, preact=zi)),
                             (l, lwi, None, dict(bias=lbi, residual=res, drop_p=0.1, drop_seed=77))])
    zi1 = torch.empty_like(zi)
    s1 = ops.gemm(x, wi, bias=bi, act=ACT_GELU, preact=zi1)
    s2 = ops.gemm(l, lwi, bias=lbi, residual=res, drop_p=0.1, drop_seed=77)
    assert rel_err(o1, s1) < 4e-3 and rel_err(zi, zi1) < 4e-3 and rel_err(o2, s2) < 4e-3     # (one bf16 step: the single launch may pick another tile shape)
    assert rel_err(o1, F.gelu(x.float() @ wi.float().t() + bi)) < 2e-2
    # dgrad pair (B stored [K, N]): d_zi = (dy Wo2) * GELU'(zi), d_l = dly lWo2 + residual
    dy = torch.randn(R, 768, generator=g).to(DEV).to(dt); dly = torch.randn(R, 192, generator=g).to(DEV).to(dt)
    wo2 = (torch.randn(768, 3072, generator=g) * 0.05).to(DEV).to(dt); lwo2 = (torch.randn(192, 768, generator=g) * 0.05).to(DEV).to(dt)
    d1, d2 = ops.gemm_group([(dy, wo2, None, dict(grad_src=zi, grad_act=ACT_GELU)), (dly, lwo2, None, dict(residual=res))], b_kmajor=False)
    assert rel_err(d1, ops.gemm(dy, wo2, b_kmajor=False, grad_src=zi, grad_act=ACT_GELU)) < 4e-3
    assert rel_err(d2, ops.gemm(dly, lwo2, b_kmajor=False, residual=res)) < 4e-3


def test_weighted_ce_and_spots(ops):
    from oracle import peneo_oracle as O
    N = 40
    P = N * (N + 1) // 2
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(P, 3, generator=g)
    logits[:, 0] += 2.5
    tags = torch.randint(0, 3, (P,), generator=g)
    cw = torch.tensor([1.0, 10.0, 10.0])
    num, den, dl = ops.weighted_ce(logits.to(DEV), tags.to(DEV), cw.to(DEV), want_dlogits=True)
    lr = logits.clone().requires_grad_(True)
    ls = F.cross_entropy(lr, tags, weight=cw, reduction="sum")
    ls.backward()
    assert abs(float(num) - float(ls)) / float(ls) < 1e-5 and rel_err(dl.cpu(), lr.grad) < 1e-5
    # ignore_index (-100) rows carry no loss, no weight and no gradient, like F.cross_entropy(ignore_index=-100)
    tags_i = tags.clone(); tags_i[::7] = -100
    num_i, den_i, dl_i = ops.weighted_ce(logits.to(DEV), tags_i.to(DEV), cw.to(DEV), want_dlogits=True)
    lr2 = logits.clone().requires_grad_(True)
    ls2 = F.cross_entropy(lr2, tags_i, weight=cw, reduction="sum", ignore_index=-100)
    ls2.backward()
    assert abs(float(num_i) - float(ls2)) / float(ls2) < 1e-5 and rel_err(dl_i.cpu(), lr2.grad) < 1e-5
    assert abs(float(den_i) - float(cw[tags_i[tags_i >= 0]].sum())) < 1e-3
    spots, scores = ops.spots_compact(logits.to(DEV), N, max_spots=16)  # forces the regrow path
    ref = O.spots_from_logits(logits)
    assert [tuple(r) for r in spots.cpu().tolist()] == [(i, j, t) for i, j, t, _ in ref]
    assert rel_err(scores.cpu(), torch.tensor([s for *_, s in ref])) < 1e-5


def test_spots_to_tags_matches_the_host_loop(ops):
    from peneo_amd.model import HandshakingTaggingScheme as S
    N = 37
    g = torch.Generator().manual_seed(2)
    batch = []
    for b in range(3):
        spots = []
        for _ in range(40):
            i = int(torch.randint(0, N, (1,), generator=g)); j = int(torch.randint(i, N, (1,), generator=g))
            spots.append((i, j, int(torch.randint(1, 3, (1,), generator=g))))
        spots.append(spots[3][:2] + (2,)); spots.append(spots[3][:2] + (1,))      # duplicates: the last one wins
        batch.append(spots)
    batch.append([])                                                             # a document without spots
    want = S.spots2shaking_tag4batch(batch, seq_len=N)
    got = S.spots2shaking_tag4batch_device(batch, N, DEV)
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    with pytest.raises(IndexError):
        S.spots2shaking_tag4batch_device([[(0, N, 1)]], N, DEV)


def test_fused_adamw_matches_torch_adamw_over_the_reference_groups(ops):
    """Four groups of pipeline/trainer.py:286-322 (decoder lr x ratio, no decay on biases / LayerNorm), 5 steps."""
    from peneo_amd.optim import FusedAdamW, peneo_param_groups

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Sequential(torch.nn.Linear(33, 70), torch.nn.LayerNorm(70))
            self.peneo_decoder = torch.nn.Sequential(torch.nn.Linear(70, 5001), torch.nn.LayerNorm(5001))
    torch.manual_seed(0)
    a, b = Toy().to(DEV), Toy().to(DEV)
    b.load_state_dict(a.state_dict())
    ga = peneo_param_groups(a, 1e-3, 0.05, 30.0)
    gb = peneo_param_groups(b, 1e-3, 0.05, 30.0)
    assert [len(g["params"]) for g in ga] == [1, 3, 1, 3] and ga[0]["lr"] == pytest.approx(0.03) and ga[1]["weight_decay"] == 0.0
    oa = FusedAdamW(ga, betas=(0.9, 0.98), eps=1e-6)
    ob = torch.optim.AdamW(gb, betas=(0.9, 0.98), eps=1e-6)
    g = torch.Generator().manual_seed(1)
    for step in range(5):
        for (_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            gr = torch.randn(pa.shape, generator=g).to(DEV)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        if step == 3:                       # scheduler-style lr change between steps
            for grp in oa.param_groups + ob.param_groups:
                grp["lr"] *= 0.5
        oa.step(); ob.step()
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert rel_err(pa.detach(), pb.detach()) < 2e-6, n


def test_fused_adamw_clips_the_global_gradient_norm_like_clip_grad_norm(ops):
    """max_grad_norm (HF Trainer's default 1.0, the reference's training loop start/run_rfund.py:307-321): the fused step
    with clipping against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW over the reference's four groups; steps whose
    norm is above AND below the threshold; the gradients themselves stay un-clipped; the reported norm is torch's."""
    from peneo_amd.optim import FusedAdamW, peneo_param_groups

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = torch.nn.Sequential(torch.nn.Linear(33, 70), torch.nn.LayerNorm(70))
            self.peneo_decoder = torch.nn.Sequential(torch.nn.Linear(70, 5001), torch.nn.LayerNorm(5001))
    torch.manual_seed(0)
    a, b = Toy().to(DEV), Toy().to(DEV)
    b.load_state_dict(a.state_dict())
    oa = FusedAdamW(peneo_param_groups(a, 1e-3, 0.05, 30.0), betas=(0.9, 0.98), eps=1e-6, max_grad_norm=1.0)
    ob = torch.optim.AdamW(peneo_param_groups(b, 1e-3, 0.05, 30.0), betas=(0.9, 0.98), eps=1e-6)
    g = torch.Generator().manual_seed(1)
    for step, scale in enumerate([1.0, 1e-3, 0.3, 1e-4, 5.0]):          # norms ~ 600, 0.6, 180, 0.06, 3000
        for (_, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
            gr = (torch.randn(pa.shape, generator=g) * scale).to(DEV)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        keep = [p.grad.clone() for p in a.parameters()]
        want_norm = torch.nn.utils.clip_grad_norm_(b.parameters(), 1.0)
        oa.step(); ob.step()
        assert rel_err(oa.last_grad_norm(), want_norm.reshape(1)) < 1e-5, (step, float(oa.last_grad_norm()), float(want_norm))
        assert all(torch.equal(p.grad, k) for p, k in zip(a.parameters(), keep))
        assert (float(want_norm) > 1.0) == (scale >= 0.3)
    for (n, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert rel_err(pa.detach(), pb.detach()) < 3e-6, n
    with pytest.raises(ValueError):
        FusedAdamW(peneo_param_groups(a, 1e-3, 0.05, 30.0), max_grad_norm=0.0)


def test_ohem_ce_matches_reference_cases_and_oracle(ops):
    """peneo_ohem_ce (OHEM branch, custom_loss.py:204-288) against the fixture produced by the real reference, then against
    the CPU oracle at the size of a full config-2 head (B * P = 8 * 130 816 pairs)."""
    from conftest import load_golden
    from oracle import peneo_oracle as O
    fx = load_golden("ohem")
    for c in fx["cases"]:
        lg, tg, w = c["logits"].to(DEV), c["target"].to(DEV), c["weight"].to(DEV)
        # un-normalised dlogits as peneo_pair_heads_fwd writes them: w_y (softmax - onehot)
        dl = (lg.softmax(-1) - F.one_hot(tg, lg.shape[-1])) * w[tg][:, None]
        dls = torch.zeros(lg.shape[-1], device=DEV)
        out8, _ = ops.ohem_ce(lg, tg, w, c["num_hard_positive"], c["num_hard_negative"], dlogits=dl, dl_sum=dls)
        o = out8.cpu()
        key = (c["num_hard_positive"], c["num_hard_negative"], tuple(lg.shape))
        n_pos = int((c["target"] != 0).sum())
        assert int(o[3]) == n_pos and int(o[4]) == lg.shape[0] - n_pos, key
        assert int(o[5]) == min(n_pos, c["num_hard_positive"]) and int(o[6]) == min(lg.shape[0] - n_pos, c["num_hard_negative"])
        assert abs(float(o[0]) - float(c["loss"])) <= 2e-5 * abs(float(c["loss"])) + 1e-6, (key, float(o[0]), float(c["loss"]))
        if c["grad"] is not None:
            got = (dl / o[2].item()).cpu()
            assert (got - c["grad"]).abs().max() <= 2e-5 * c["grad"].abs().max() + 1e-8, key
            assert rel_err(dls, dl.sum(0)) < 1e-4
    # full-size head
    g = torch.Generator().manual_seed(21)
    n = 8 * 130816
    lg = torch.randn(n, 3, generator=g) * 1.5
    tg = (torch.rand(n, generator=g) < 2e-4).long() * torch.randint(1, 3, (n,), generator=g)
    w = torch.tensor([1.0, 10.0, 10.0])
    lr = lg.clone().requires_grad_(True)
    ref = O.ohem_ce(lr, tg, w, 64, 20000)
    ref.backward()
    dl = ((lg.softmax(-1) - F.one_hot(tg, 3)) * w[tg][:, None]).to(DEV)
    out8, _ = ops.ohem_ce(lg.to(DEV), tg.to(DEV), w.to(DEV), 64, 20000, dlogits=dl)
    o = out8.cpu()
    assert abs(float(o[0]) - float(ref)) <= 2e-5 * abs(float(ref))
    got = (dl / o[2].item()).cpu()
    kept_g, kept_r = got.abs().sum(-1) > 0, lr.grad.abs().sum(-1) > 0
    assert int(kept_g.sum()) == int(kept_r.sum()) == 64 + 20000
    # the kept set is a function of the ORDER of 1 M losses (mean gap between neighbours: ~70 ulp, so ~1 % of neighbours lie
    # within one ulp): where this kernel's and torch's log-softmax differ in the last bit two neighbours swap ranks and a
    # kept row is replaced by its neighbour of (nearly) equal loss — measured 0.5 % of the kept rows; the rest is exact
    assert int((kept_g != kept_r).sum()) <= 0.02 * (64 + 20000), int((kept_g != kept_r).sum())
    both = kept_g & kept_r
    assert (got[both] - lr.grad[both]).abs().max() <= 2e-5 * lr.grad.abs().max()


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,drop", [(2, 150, 0.1), (1, 40, 0.0), (3, 97, 0.2), (1, 511, 0.1)])
def test_pair_saved_activations_path_is_the_recomputing_path(ops, B, N, drop):
    """peneo_pair_heads_fwd_save + peneo_pair_bwd_saved (the forward leaves the classifiers' pre-activations as f16, the backward reads
    them) against peneo_pair_heads_fwd + peneo_pair_bwd_fused (the backward rebuilds them): logits, dlogits and the x rows are the same
    numbers bit for bit (same arithmetic per pair, another walk of the triangle); dz, d_ab and the dW2 / db1 sums differ by the f16
    rounding of the saved z only.  Reference: model/peneo_decoder.py:231-292 forward and its autograd graph."""
    D, classes, nh = 384, [2, 3, 3, 3, 3], 5
    dt, dev = torch.bfloat16, "cuda"
    assert ops.pair_save_supported(dt, D, nh) and not ops.pair_save_supported(torch.float32, D, nh) and not ops.pair_save_supported(dt, 512, nh)
    g = torch.Generator().manual_seed(N + int(100 * drop))
    P = N * (N + 1) // 2
    ab = torch.randn(B, N, 2 * D, generator=g).to(dev).to(dt)
    w1 = [(torch.randn(D, D, generator=g) / math.sqrt(D)).to(dev) for _ in classes]
    w2 = [(torch.randn(c, D, generator=g) / math.sqrt(D)).to(dev) for c in classes]
    b1, b2 = (0.1 * torch.randn(nh * D, generator=g)).to(dev), (0.1 * torch.randn(14, generator=g)).to(dev)
    wp = ops.pair_heads_pack(dt, w1, w2)
    tags = [torch.randint(0, c, (B, P), generator=g).to(dev) for c in classes]
    cw = [(torch.rand(c, generator=g) + 0.5).to(dev) for c in classes]
    kw = dict(tags=tags, class_weights=cw, want_dlogits=True, want_logits=True, drop_p=drop, drop_seed=77)
    lg0, pt0, dl0 = ops.pair_heads_fwd(ab, wp, b1, b2, classes, **kw)
    lg1, pt1, dl1, (act, xr) = ops.pair_heads_fwd(ab, wp, b1, b2, classes, save=True, **kw)
    for h in range(nh):
        assert torch.equal(lg0[h], lg1[h]) and torch.equal(dl0[h], dl1[h])
    s0, s1 = pt0.sum(0), pt1.sum(0)
    assert float((s0 - s1).abs().max() / s0.abs().max()) < 1e-5          # the same terms in other partial rows
    wp2 = ops.pair_bwd_pack(w1)
    rows = ops.pair_bwd_rows(N)
    scale = (torch.rand(nh, generator=g) + 0.5).to(dev)
    outs = []
    for saved in (False, True):
        dz = torch.full((B * rows, nh * D), 3.0, device=dev, dtype=dt)
        d_ab = torch.zeros(B, N, 2 * D, device=dev)
        ws = ops.pair_dz_workspace(nh, D, dev, slots=256)
        args = ops.pair_dz_args(D, classes, dl0, w2, scale, drop_p=drop, drop_seed=77)
        if saved:
            ops.pair_bwd_saved(ab, wp2, args, act, dz, d_ab, ws)
            x = xr
        else:
            x = torch.empty(B * rows, D, device=dev, dtype=dt)
            ops.pair_bwd_fused(ab, wp2, b1, args, dz, x, d_ab, ws)
        torch.cuda.synchronize()
        outs.append((dz.float(), x, d_ab, ws.sum(0)))
    assert torch.equal(outs[0][1], outs[1][1])                            # x rows: bit for bit
    for i, tol in ((0, 2e-3), (2, 1e-3), (3, 1e-3)):                       # dz (bf16 values from an f16-rounded z), d_ab, column sums
        a, b_ = outs[0][i], outs[1][i]
        assert torch.isfinite(b_).all()
        assert float((a - b_).norm() / (a.norm() + 1e-30)) < tol, (i, float((a - b_).norm() / a.norm()))
    # a dropped unit is dropped in both: the zero pattern of dz is the same
    if drop > 0:
        assert torch.equal(outs[0][0] == 0, outs[1][0] == 0)



@pytest.mark.parametrize("B,N,D,p_drop", [(2, 45, 128, 0.0), (1, 70, 384, 0.0), (3, 23, 32, 0.0), (1, 130, 64, 0.0),
                                          (2, 16, 384, 0.0), (1, 9, 128, 0.0), (2, 45, 128, 0.1), (1, 70, 384, 0.1),
                                          (3, 23, 32, 0.25), (2, 511, 384, 0.0), (2, 511, 384, 0.1),
                                          (2, 37, 512, 0.0), (1, 70, 512, 0.1), (1, 1023, 512, 0.1)])
def test_pair_bwd_fused_matches_autograd(ops, B, N, D, p_drop):
    """peneo_pair_bwd_fused: dz / x in block order, d_ab (= d_a | d_b), dW2 / db1 sums and (through the one GEMM it leaves)
    dW1, against fp32 autograd through x = SiLU(a_i + b_j) -> z = x W1^T + b1 -> SiLU -> W2 with given dlogits."""
    dtype, classes = torch.bfloat16, [2, 3, 3, 3, 3]
    nh = len(classes)
    g = torch.Generator().manual_seed(B * 100000 + N * 1000 + D)
    ab = torch.randn(B, N, 2 * D, generator=g).to(DEV).to(dtype)
    P = N * (N + 1) // 2
    w1 = [(torch.randn(D, D, generator=g) / math.sqrt(D)).to(DEV) for _ in classes]
    w2 = [torch.randn(c, D, generator=g).to(DEV) for c in classes]
    b1cat = (0.1 * torch.randn(nh * D, generator=g)).to(DEV)
    dl = [torch.randn(B, P, c, generator=g).to(DEV) for c in classes]
    scale = torch.rand(nh, generator=g).to(DEV) + 0.5
    assert ops.pair_bwd_supported(dtype, D)
    rows = ops.pair_bwd_rows(N)
    assert rows % 128 == 0 and rows >= P
    wp2 = ops.pair_bwd_pack(w1)
    from dropout_ref import k12_keep, k12_scale
    seed = 4242 + N
    args = ops.pair_dz_args(D, classes, dl, w2, scale, drop_p=p_drop, drop_seed=seed)
    dz = torch.full((B * rows + 2, nh * D), 7.0, device=DEV, dtype=dtype)
    x = torch.full((B * rows + 2, D), 7.0, device=DEV, dtype=dtype)
    d_ab = torch.full((B, N, 2 * D), 5.0, device=DEV)            # overwritten, not accumulated
    ws = ops.pair_dz_workspace(nh, D, DEV, slots=256)
    ops.pair_bwd_fused(ab, wp2, b1cat, args, dz[:B * rows], x[:B * rows], d_ab, ws)
    torch.cuda.synchronize()
    assert bool((dz[B * rows:] == 7.0).all()) and bool((x[B * rows:] == 7.0).all())
    dW1 = ops.gemm(dz[:B * rows], x[:B * rows], a_kmajor=False, b_kmajor=False, out_dtype=torch.float32)
    dw2, db1 = ops.pair_dz_finish(ws, nh, D, classes)
    # fp32 autograd of the same block (bf16-rounded operands where the kernel rounds: ab, x, W1)
    abr = ab.float().clone().requires_grad_(True)
    ii, jj = torch.triu_indices(N, N, device=DEV)
    xr = F.silu(abr[:, ii, :D] + abr[:, jj, D:])                               # [B, P, D]
    xq = xr + (xr.detach().to(dtype).float() - xr.detach())                     # straight-through bf16 rounding of x
    w1r = [w.to(dtype).float().clone().requires_grad_(True) for w in w1]
    w2r = [w.clone().requires_grad_(True) for w in w2]
    b1r = b1cat.clone().requires_grad_(True)
    keep_all = torch.stack([k12_keep(seed, b, 0, P, nh * D, p_drop) for b in range(B)]).to(DEV) if p_drop > 0 else None
    for h in range(nh):                                                          # head by head: [B, P, D] temporaries only
        z = xq @ w1r[h].t() + b1r[h * D:(h + 1) * D]
        y = F.silu(z)
        if p_drop > 0:
            y = y * keep_all[:, :, h * D:(h + 1) * D] * k12_scale(p_drop)
        (((y @ w2r[h].t()) * dl[h] * scale[h]).sum()).backward(retain_graph=h + 1 < nh)
        del z, y
    t = 3e-2
    assert rel_err(d_ab, abr.grad) < t, rel_err(d_ab, abr.grad)
    assert rel_err(db1, b1r.grad) < t
    for h in range(nh):
        assert rel_err(dw2[h], w2r[h].grad) < t, h
        assert rel_err(dW1[h * D:(h + 1) * D], w1r[h].grad) < t, h
    # rows outside the triangle are exactly zero in dz and finite in x; the valid ones carry every pair exactly once
    nz = (dz[:B * rows].float().abs().sum(-1) > 0).view(B, rows).sum(-1)
    assert int(nz.max()) <= P and bool(torch.isfinite(x[:B * rows].float()).all())
    if N >= 500:
        # the size the benchmark runs at (1056 blocks per document, the 256-slot workspace wrapped 4x per document): the
        # kernel must be bit-reproducible launch to launch in dz / x / d_ab (packed fp32 VALU beside a partner
        # wave's MFMAs returned sporadic wrong 16-byte pieces of dz; only the dW2 / db1 atomics may differ in the last bits)
        for _ in range(3):
            dz2, x2, d2 = torch.empty_like(dz[:B * rows]), torch.empty_like(x[:B * rows]), torch.empty_like(d_ab)
            ops.pair_bwd_fused(ab, wp2, b1cat, args, dz2, x2, d2, ops.pair_dz_workspace(nh, D, DEV, slots=256))
            assert torch.equal(dz2, dz[:B * rows]) and torch.equal(x2, x[:B * rows]) and torch.equal(d2, d_ab)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_bwd_partial_sums_equal_the_atomic_form(ops, dtype):
    """peneo_layernorm_bwd_partial: same dx (and dropped second output) as peneo_layernorm_bwd, and its per-workgroup partial
    rows column-sum to dgamma | dbeta."""
    R, H = 5672 if dtype == torch.bfloat16 else 1000, 768
    g = torch.Generator().manual_seed(3)
    x = torch.randn(R, H, generator=g).to(DEV).to(dtype)
    dy = torch.randn(R, H, generator=g).to(DEV).to(dtype)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(DEV)
    beta = torch.zeros(H, device=DEV)
    _, mean, rstd = ops.layernorm_fwd(x, gamma, beta, 1e-5)
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    dxd1 = torch.empty_like(x)
    dx1 = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dx_dropped=dxd1, drop2_p=0.1, drop2_seed=5)
    dxd2 = torch.empty_like(x)
    dx2, part = ops.layernorm_bwd_partial(dy, x, gamma, mean, rstd, dx_dropped=dxd2, drop2_p=0.1, drop2_seed=5)
    assert part is not None and part.shape[1] == 2 * H and part.shape[0] == min((R + 7) // 8, 1024)
    assert torch.equal(dx1, dx2) and torch.equal(dxd1, dxd2)
    sums = ops.colsum(part)
    assert rel_err(sums[:H], dg) < 1e-5 and rel_err(sums[H:], db) < 1e-5
    # no partial form for a row length outside the fast instantiations
    assert ops.layernorm_bwd_partial(dy[:, :40].contiguous(), x[:, :40].contiguous(), gamma[:40], mean, rstd)[0] is None


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("shape", [(5672, 2304, 768), (1000, 192, 520), (709, 768, 3072), (64, 40, 24)])
def test_gemm_wgrad_with_fused_bias_gradient(ops, dtype, shape):
    """dW = dy^T x with db = dy^T 1 out of the same launch (peneo_gemm's a_colsum; autograd of every nn.Linear of the encoder,
    modeling_layoutlmv3.py:292-294,335-360): the sums are ADDED to the destination, every split-k slice contributes, shapes
    that do not run the LDS-DMA kernel (fp32, ragged) fall back to a column-sum pass inside the same C call."""
    K, M, N = shape                      # K = tokens, M = output features (rows of dW), N = input features
    g = torch.Generator().manual_seed(7)
    dy = torch.randn(K, M, generator=g).to(DEV).to(dtype)
    x = torch.randn(K, N, generator=g).to(DEV).to(dtype)
    base = torch.randn(M, generator=g).to(DEV)
    for split in (None, 1, 3):
        db = base.clone()
        dw = ops.gemm(dy, x, a_kmajor=False, b_kmajor=False, out_dtype=torch.float32, a_colsum=db, split_k=split)
        want_w = dy.float().t() @ x.float()
        want_b = base + dy.float().sum(0)
        tol = 2e-2 if dtype == torch.bfloat16 else 2e-5
        assert rel_err(dw, want_w) < tol, (shape, split)
        assert rel_err(db, want_b) < 1e-4, (shape, split, rel_err(db, want_b))
    with pytest.raises(Exception):
        ops.gemm(dy.t().contiguous(), x.t().contiguous(), a_colsum=base.clone())     # a_colsum needs A as [K, M]
