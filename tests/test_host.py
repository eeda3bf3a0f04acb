"""CPU-side checks: the C ABI library loads and exports every symbol of include/peneo_hip.h, and the host-side
mirror of the reference API (configs, registry, state-dict layout, label-map helpers) behaves like the reference."""
import ctypes
import os
import re

import pytest
import torch

from conftest import ROOT, load_golden


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "peneo_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(peneo_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from peneo_amd import hip
    assert os.path.exists(hip.LIB_PATH), "build libpeneo_hip.so first (python -c 'import __graft_entry__ as g; g.build()')"
    lib = ctypes.CDLL(hip.LIB_PATH)
    declared = _declared_functions()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/peneo_hip.h but not exported"
    # and the ctypes binding table covers exactly the header
    assert sorted(hip.SIGNATURES) == declared
    assert lib.peneo_version() >= 100


def test_struct_layouts_match_the_header_sizes():
    from peneo_amd import hip
    # spot checks of the by-value / by-pointer structs shared with C (natural alignment on x86-64)
    assert ctypes.sizeof(hip.GemmEpilogue) == 112
    lib = ctypes.CDLL(hip.LIB_PATH)
    lib.peneo_struct_bytes.restype = ctypes.c_size_t
    # the C side's own sizeof of the structs a binding fills field by field
    assert [lib.peneo_struct_bytes(i) for i in range(3)] == [ctypes.sizeof(hip.GemmEpilogue), ctypes.sizeof(hip.EncoderLayer),
                                                             ctypes.sizeof(hip.EncoderLayerGrads)]
    assert ctypes.sizeof(hip.PairHeadsDesc) == 8 + 4 * 8 + 3 * 8 + 8
    assert ctypes.sizeof(hip.PairLoss) == 8 * 8 * 2 + 8 + 8 * 8
    assert ctypes.sizeof(hip.PairDzArgs) == 8 + 4 * 8 + 8 * 8 * 2 + 8 + 16 + 8


def test_fused_pair_backward_query_knows_the_lds_limit():
    """peneo_pair_bwd_supported(dtype, D, num_heads) is a host-side question (no GPU call): the fused launch's LDS image grows with
    the head count, so D = 512 holds the reference's 5 heads but not 6 (the caller then runs the per-document chain instead of
    getting PENEO_ERR_INVALID from peneo_pair_bwd_fused), D = 384 holds 5 of the 8 heads PENEO_MAX_HEADS allows."""
    from peneo_amd import hip
    lib = ctypes.CDLL(hip.LIB_PATH)
    f = lib.peneo_pair_bwd_supported
    f.restype, f.argtypes = ctypes.c_int, [ctypes.c_int] * 3
    BF16, F32 = hip.BF16, hip.F32
    assert f(BF16, 512, 0) == 1 and f(BF16, 512, 5) == 1 and f(BF16, 512, 6) == 0
    assert f(BF16, 384, 5) == 1 and f(BF16, 384, 8) == 0 and f(BF16, 384, 9) == 0
    assert f(BF16, 128, 8) == 1 and f(BF16, 64, 8) == 1
    assert f(F32, 384, 5) == 0 and f(BF16, 400, 5) == 0


def test_no_cpu_fallback():
    from peneo_amd import ops
    from peneo_amd.hip import PeneoHipError
    a = torch.randn(4, 8)
    with pytest.raises(PeneoHipError):
        ops.gemm(a, a)


def test_product_never_imports_the_oracle():
    for base, _, files in os.walk(os.path.join(ROOT, "peneo_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(base, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(base, f)


def test_state_dict_layout_matches_reference():
    from peneo_amd.model import PEneoConfig, PEneoModel
    fx = load_golden("lmv3_tiny")
    m = PEneoModel(PEneoConfig(**{k: v for k, v in fx["config"].items() if k != "model_type"}))
    sd = m.state_dict()
    assert set(sd) == set(fx["state_dict"])
    for k, v in fx["state_dict"].items():
        assert sd[k].shape == v.shape and sd[k].dtype == v.dtype, k
    m.load_state_dict(fx["state_dict"], strict=True)
    # optimizer hook of pipeline/trainer.py:280-322: decoder parameters are found by name
    assert sum("peneo_decoder" in n for n, _ in m.named_parameters()) == 26


def test_model_api_errors():
    from peneo_amd.model import BACKBONE_MAPPING, PEneoConfig, PEneoModel
    with pytest.raises(ValueError):
        PEneoModel(PEneoConfig(backbone_name="layoutlmv3-base", backbone_config=None))
    assert list(BACKBONE_MAPPING) == ["lilt-infoxlm-base", "lilt-roberta-en-base", "layoutxlm-base",
                                      "layoutlmv2-base-uncased", "layoutlmv3-base-chinese", "layoutlmv3-base"]
    info = BACKBONE_MAPPING["layoutlmv3-base"]
    assert (info.max_token_len, info.add_cls_token, info.add_sep_token, info.has_visual_embeds) == (510, True, True, True)
    info = BACKBONE_MAPPING["lilt-roberta-en-base"]
    assert (info.max_token_len, info.add_cls_token, info.add_sep_token, info.has_visual_embeds) == (511, True, False, False)
    with pytest.raises(NotImplementedError):
        BACKBONE_MAPPING["layoutxlm-base"].model(None)


def test_config_roundtrip(tmp_path):
    from peneo_amd.model import PEneoConfig
    fx = load_golden("lmv3_tiny")
    cfg = PEneoConfig(**{k: v for k, v in fx["config"].items() if k != "model_type"})
    cfg.save_pretrained(tmp_path)
    back = PEneoConfig.from_pretrained(tmp_path)
    assert back.backbone_config["hidden_size"] == 64 and back.peneo_category_weights == [1.0, 10.0, 10.0]
    assert back.model_type == "peneo"


def test_model_save_and_from_pretrained_roundtrip(tmp_path):
    """``PEneoModel.save_pretrained`` / ``from_pretrained(dir, config=config)`` (start/run_rfund.py:199-202, Trainer.save_model):
    the checkpoint keeps the reference's 241-tensor layout and comes back bit for bit (no GPU needed: nothing is launched)."""
    from peneo_amd.model import PEneoConfig, PEneoModel
    fx = load_golden("lmv3_tiny")
    cfg = PEneoConfig(**{k: v for k, v in fx["config"].items() if k != "model_type"})
    model = PEneoModel(cfg)
    model.load_state_dict(fx["state_dict"], strict=True)
    model.save_pretrained(tmp_path)
    back = PEneoModel.from_pretrained(tmp_path, config=PEneoConfig.from_pretrained(tmp_path))
    a, b = model.state_dict(), back.state_dict()
    assert list(a) == list(b) == list(fx["state_dict"])
    assert all(torch.equal(a[k], b[k]) for k in a)
    # decoder parameters keep the prefix the reference's optimizer groups key on (pipeline/trainer.py:280-284)
    assert sum(1 for n, _ in back.named_parameters() if "peneo_decoder" in n) == 26


def test_backbone_from_checkpoint_path(tmp_path, caplog):
    """``PEneoModel(config, backbone_name_or_path=dir)`` (reference model/modeling_peneo.py:58-79): backbone weights come
    from the checkpoint directory (HF layout, keys prefixed ``layoutlmv3.``), ``config.backbone_config`` is filled from
    its config.json when absent, the decoder is freshly initialised, and a path that cannot be read raises OSError (the
    reference's hub fallback needs the network) instead of training from random weights."""
    import json
    from safetensors.torch import save_file
    from peneo_amd.model import PEneoConfig, PEneoModel
    fx = load_golden("lmv3_tiny")
    full = {k: v for k, v in fx["config"].items() if k != "model_type"}
    bb = {k[len("backbone."):]: v for k, v in fx["state_dict"].items() if k.startswith("backbone.")}
    ck = tmp_path / "ckpt"
    ck.mkdir()
    save_file({"layoutlmv3." + k: v.contiguous() for k, v in bb.items() if v.dtype.is_floating_point}, str(ck / "model.safetensors"))
    json.dump(dict(full["backbone_config"], model_type="layoutlmv3", architectures=["LayoutLMv3Model"]), open(ck / "config.json", "w"))
    for given in (None, full["backbone_config"]):
        cfg = PEneoConfig(**dict(full, backbone_config=given))
        m = PEneoModel(cfg, backbone_name_or_path=str(ck))
        sd = m.backbone.state_dict()
        assert all(torch.equal(sd[k], v) for k, v in bb.items() if v.dtype.is_floating_point)
        assert cfg.backbone_config["hidden_size"] == full["backbone_config"]["hidden_size"]
        w = m.peneo_decoder.line_extraction_fc[0].weight
        assert 0.01 < float(w.std()) < 0.03 and float(m.peneo_decoder.line_extraction_fc[0].bias.abs().max()) == 0.0
    for bad in (str(tmp_path / "nowhere"), "auto"):
        with pytest.raises(OSError):
            PEneoModel(PEneoConfig(**full), backbone_name_or_path=bad)


def test_init_weights_rule():
    """Reference rule (modeling_layoutlmv3.py:260-274 via model/modeling_peneo.py:105-106): Linear / Embedding N(0, 0.02),
    biases and the padding row zero, LayerNorm (1, 0)."""
    from peneo_amd.model import PEneoConfig, PEneoModel
    fx = load_golden("lmv3_tiny")
    torch.manual_seed(1)
    m = PEneoModel(PEneoConfig(**{k: v for k, v in fx["config"].items() if k != "model_type"}))
    e = m.backbone.embeddings
    assert float(e.word_embeddings.weight[e.word_embeddings.padding_idx].abs().max()) == 0.0
    assert abs(float(e.word_embeddings.weight.std()) - 0.02) < 0.003
    lyr = m.backbone.encoder.layer[0]
    assert abs(float(lyr.intermediate.dense.weight.std()) - 0.02) < 0.003 and float(lyr.intermediate.dense.bias.abs().max()) == 0.0
    ln = lyr.output.LayerNorm
    assert torch.equal(ln.weight, torch.ones_like(ln.weight)) and torch.equal(ln.bias, torch.zeros_like(ln.bias))
    for name in ("line_extraction_fc", "ent_linking_h2h_fc"):
        fc = getattr(m.peneo_decoder, name)
        assert abs(float(fc[0].weight.std()) - 0.02) < 0.004 and float(fc[3].bias.abs().max()) == 0.0


def test_tagging_scheme_matches_oracle():
    from oracle import peneo_oracle as O
    from peneo_amd.data import spots_to_shaking_tag
    from peneo_amd.model import HandshakingTaggingScheme as H
    n = 11
    spots = [(0, 0, 1), (2, 7, 2), (10, 10, 1), (3, 4, 1), (0, 10, 2)]
    tag = H.spots2shaking_tag4batch([spots, []], seq_len=n)
    assert torch.equal(tag[0], O.spots_to_tag(spots, n)) and torch.equal(tag[0], spots_to_shaking_tag(spots, n))
    assert int(tag[1].sum()) == 0
    logits = torch.full((tag.shape[1], 3), -4.0)
    logits[torch.arange(tag.shape[1]), tag[0]] = 4.0
    got = H.get_spots_from_shaking_tag(logits, seq_len=n)          # CPU tensor: plain loop
    want = O.spots_from_logits(logits)
    assert [(i, j, t) for i, j, t, _ in got] == [(i, j, t) for i, j, t, _ in want]
    assert max(abs(a[3] - b[3]) for a, b in zip(got, want)) < 1e-6
    # integer tag input (no class dim)
    got2 = H.get_spots_from_shaking_tag(tag[0], seq_len=n)
    assert sorted((i, j, t) for i, j, t, _ in got2) == sorted(spots)


def test_row_chunks_cover_triangle():
    from peneo_amd.model.peneo_decoder import _row_chunks
    for n, cap in [(511, 32768), (39, 100), (5, 1), (1023, 32768)]:
        ch = _row_chunks(n, cap)
        assert ch[0][0] == 0 and ch[-1][1] == n and all(a[1] == b[0] for a, b in zip(ch, ch[1:]))
        for i0, i1 in ch:
            pairs = sum(n - i for i in range(i0, i1))
            assert pairs <= cap or i1 - i0 == 1


def test_synthetic_batch_contract():
    from peneo_amd.data import synthetic_rfund_batch
    b = synthetic_rfund_batch(3, 64, 9, 500, seed=3, ragged=True)
    n = 63
    assert b["input_ids"].shape == (3, 64) and b["bbox"].shape == (3, 64, 4) and b["image"].shape == (3, 3, 224, 224)
    for k in ("line_extraction_shaking_tag", "ent_linking_head_rel_shaking_tag", "line_grouping_tail_rel_shaking_tag"):
        assert b[k].shape == (3, n * (n + 1) // 2) and b[k].dtype == torch.int64
    assert int(b["bbox"].max()) <= 1000 and int(b["input_ids"][:, 0].sum()) == 0
    assert torch.equal(b["attention_mask"], (b["input_ids"] != 1).long())
    b2 = synthetic_rfund_batch(3, 64, 9, 500, seed=3, ragged=True)
    assert all(torch.equal(b[k], b2[k]) for k in b)


def _same_decode(got, want):
    """7-tuple of sample_decode_peneo: kv pairs, lines, five link dictionaries."""
    assert len(got) == len(want) == 7
    assert [tuple(map(lambda v: tuple(v) if isinstance(v, list) else v, kv)) for kv in got[0]] == \
           [tuple(map(lambda v: tuple(v) if isinstance(v, list) else v, kv)) for kv in want[0]]
    norm = lambda ln: (ln[0], tuple(ln[1])) if isinstance(ln, (tuple, list)) else ln
    assert [norm(l) for l in got[1]] == [norm(l) for l in want[1]]
    for a, b in zip(got[2:], want[2:]):
        assert dict(a) == dict(b)


def test_decode_graph_walk_matches_reference():
    """peneo_amd.pipeline.decode (spots -> lines -> kv pairs) on CPU tensors against the fixture produced by the reference's
    pipeline/decode.py (tests/golden/decode.pt): predictions (one successor per node, best score wins), a score threshold,
    ground-truth decoding, and the batch form."""
    from peneo_amd.model import HandshakingTaggingScheme
    from peneo_amd.pipeline import decode_peneo, parse_matrix_spots, sample_decode_peneo
    fx = load_golden("decode")
    T = HandshakingTaggingScheme()
    for d in fx["docs"]:
        _same_decode(sample_decode_peneo(T, d["text"], *d["logits"], bbox=d["bbox"], seq_len=d["n"]), d["pred"])
        _same_decode(sample_decode_peneo(T, d["text"], *d["logits"], seq_len=d["n"], score_thresh=0.6), d["pred_thr"])
        _same_decode(sample_decode_peneo(T, d["text"], *d["tags"], bbox=d["bbox"], seq_len=d["n"], decode_gt=True), d["gt"])
        assert len(d["gt"][0]) >= 2                                # the hand-built documents do contain pairs
    b = fx["docs"][:2]
    preds, gts, ids = decode_peneo(T, [d["text"] for d in b], *[[d["logits"][h] for d in b] for h in range(5)],
                                   *[[d["tags"][h] for d in b] for h in range(5)], [d["bbox"].tolist() for d in b],
                                   ["a.json", "b.json"])
    assert ids == fx["batch"][2]
    for got, want in zip(preds, fx["batch"][0]):
        _same_decode(got, want)
    for got, want in zip(gts, fx["batch"][1]):
        _same_decode(got, want)
    # tie / direction rules of parse_matrix_spots
    assert parse_matrix_spots([(1, 4, 1, 0.9), (1, 5, 1, 0.95), (2, 5, 1, 0.5)], top_score_only=True) == {1: 5}
    assert parse_matrix_spots([(1, 4, 2, 0.9), (0, 4, 1, 0.3)], triu_mode=True) == {4: [1], 0: [4]}


def test_bench_launch_plan_never_relabels_a_smaller_job():
    """`bench.py --gpus N`: no launcher environment -> spawn N fresh ranks; under torch.distributed.run with WORLD_SIZE == N ->
    run as a rank; any other WORLD_SIZE -> refuse (reference launch: torchrun --nproc_per_node N, README.md:206-218)."""
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    assert bench.plan_launch(1, {}) == ("run",)
    assert bench.plan_launch(8, {}) == ("spawn", 8)
    assert bench.plan_launch(8, {"WORLD_SIZE": "8", "RANK": "3"}) == ("run",)
    assert bench.plan_launch(8, {"WORLD_SIZE": "1"})[0] == "fail"
    assert bench.plan_launch(1, {"WORLD_SIZE": "2", "RANK": "0"})[0] == "fail"
    # the refusal is an exit code, taken before anything touches a GPU (so it can be checked on a CPU box)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=dict(os.environ, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr, (r.returncode, r.stderr[-400:])


def test_step_sized_buffers_are_kept_and_handed_out_once():
    """engine.big_acquire / big_release (the decoder's multi-gigabyte buffers of a train step): a released buffer comes back for the
    same or a smaller request, never to two holders at once, and a larger request drops the smaller kept ones."""
    import torch
    from peneo_amd.model import engine
    engine._BIG_FREE.clear()
    a, ha = engine.big_acquire("t", (4, 8), torch.float32, "cpu")
    b, hb = engine.big_acquire("t", (4, 8), torch.float32, "cpu")
    assert a.shape == (4, 8) and a.dtype == torch.float32 and ha is not hb and a.data_ptr() != b.data_ptr()
    engine.big_release("t", ha)
    engine.big_release("t", ha)                       # (released twice: kept once)
    c, hc = engine.big_acquire("t", (2, 8), torch.bfloat16, "cpu")
    assert hc is ha and c.shape == (2, 8) and c.dtype == torch.bfloat16 and c.data_ptr() == a.data_ptr()
    d, hd = engine.big_acquire("t", (4, 8), torch.float32, "cpu")
    assert hd is not ha and hd is not hb              # nothing free: a new one
    engine.big_release("t", hb); engine.big_release("t", hc); engine.big_release("t", hd)
    assert len(engine._BIG_FREE[("t", "cpu")]) == 2   # at most two kept per tag
    e, he = engine.big_acquire("t", (1 << 22,), torch.uint8, "cpu")       # larger than anything kept: the kept ones go
    assert he.numel() >= 1 << 22 and engine._BIG_FREE[("t", "cpu")] == []
    # big_clear: the pool of one device, or all of it, is dropped and reported; a buffer still held is untouched
    engine.big_release("t", he)
    engine.big_release("u", hb)
    held = engine.big_acquire("v", (16,), torch.uint8, "cpu")[1]
    assert engine.big_clear("cuda:7") == 0 and len(engine._BIG_FREE[("t", "cpu")]) == 1
    assert engine.big_clear("cpu") == he.numel() + hb.numel() and not any(k[1] == "cpu" for k in engine._BIG_FREE)
    assert held.numel() > 0
    engine.big_release("t", he)
    assert engine.big_clear() == he.numel() and engine._BIG_FREE == {}


def test_saved_activation_sizes_follow_the_block_walk():
    """Host-side size queries of the saving forward (no GPU): one 2 KiB record per document, block of 8 x 16 pairs, 32-unit slab and
    group of 32 pairs; one loss-partial row per workgroup of two blocks; bf16 at D = 384 only (include/peneo_hip.h)."""
    from peneo_amd import hip
    lib = hip.lib()
    rows = lib.peneo_pair_bwd_rows(511)
    assert rows % 128 == 0 and rows >= 511 * 512 // 2
    assert lib.peneo_pair_save_bytes(8, 511, 5, 384) == 8 * (rows // 128) * (5 * 384 // 32) * 4 * 2048 == 4152360960
    assert lib.peneo_pair_loss_partials_save(8, 511) == 8 * ((rows // 128 + 1) // 2)
    assert lib.peneo_pair_save_bytes(0, 511, 5, 384) == 0
    assert lib.peneo_pair_save_supported(hip.BF16, 384, 5) == 1
    assert lib.peneo_pair_save_supported(hip.F32, 384, 5) == 0 and lib.peneo_pair_save_supported(hip.BF16, 512, 5) == 0
    assert lib.peneo_pair_save_supported(hip.BF16, 384, 0) == 0
