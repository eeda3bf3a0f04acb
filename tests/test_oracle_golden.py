"""Pin the CPU oracle against fixtures produced by the real reference (tests/golden/make_golden.py)."""
import pytest
import torch

from conftest import load_golden
from oracle import peneo_oracle as O

HEADS = O.HEAD_NAMES
TINY = ["lmv3_tiny", "lmv3_tiny_s24", "lilt_tiny", "lmv3_tiny_cls1", "lmv3_tiny_cls3"]


def _req_grad(sd):
    out = {}
    for k, v in sd.items():
        out[k] = v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("_loss.weight") else v
    return out


@pytest.mark.parametrize("name", TINY)
@pytest.mark.parametrize("as_executed", [False, True])
def test_forward_matches_reference(name, as_executed):
    fx = load_golden(name)
    cap = {}
    with torch.no_grad():
        out = O.peneo_forward(fx["state_dict"], fx["config"], fx["batch"], as_executed=as_executed, capture=cap)
    ref = fx["outputs"]
    for h in HEADS:
        k = h + "_shaking_outputs"
        assert out[k].shape == ref[k].shape
        assert (out[k] - ref[k]).abs().max() < 2e-5, k
        assert torch.equal(out[k].argmax(-1), ref[k].argmax(-1)), k
        assert abs(float(out[h + "_loss"]) - float(ref[h + "_loss"])) < 1e-5
    assert abs(float(out["loss"]) - float(ref["loss"])) < 1e-5
    assert torch.equal(out["orig_bbox"], ref["orig_bbox"])
    for k, v in fx["captures"].items():
        if k in cap:
            assert (cap[k] - v).abs().max() < 2e-5, k


@pytest.mark.parametrize("name", TINY)
def test_gradients_match_reference(name):
    fx = load_golden(name)
    sd = _req_grad(fx["state_dict"])
    out = O.peneo_forward(sd, fx["config"], fx["batch"])
    out["loss"].backward()
    checked = 0
    for n, g in fx["grads"].items():
        mine = sd[n].grad
        assert mine is not None, n
        # key.bias gradients are analytically zero (softmax shift invariance): allow an absolute floor
        assert (mine - g).abs().max() <= 2e-4 * g.abs().max() + 1e-7, n
        checked += 1
    assert checked > 50


def test_pair_index_is_row_major_triu():
    n = 7
    ii, jj = O.pair_index(n)
    p = 0
    for i in range(n):
        for j in range(i, n):
            assert (int(ii[p]), int(jj[p])) == (i, j)
            assert p == i * n - i * (i - 1) // 2 + (j - i)
            p += 1


def test_spots_roundtrip():
    n = 9
    spots = [(0, 0, 1), (2, 5, 2), (8, 8, 1), (3, 4, 1)]
    tag = O.spots_to_tag(spots, n)
    logits = torch.full((tag.numel(), 3), -5.0)
    logits[torch.arange(tag.numel()), tag] = 5.0
    got = O.spots_from_logits(logits)
    assert sorted((i, j, t) for i, j, t, _ in got) == sorted(spots)
    assert all(s > 0.99 for *_, s in got)


def test_bucket_function_small_cases():
    rp = torch.tensor([-300, -128, -17, -8, -7, -1, 0, 1, 7, 8, 9, 17, 127, 128, 300])
    b = O.relative_position_bucket(rp, 32, 128)
    assert b.tolist()[6] == 0 and b.tolist()[5] == 1 and b.tolist()[7] == 17
    assert b.max() == 31 and b[0] == 15 and b[-1] == 31


def test_ohem_cases_match_reference():
    """CrossEntropyLossOHEM with OHEM active (custom_loss.py:204-288), run by the real reference: loss and d loss / d logits."""
    fx = load_golden("ohem")
    assert len(fx["cases"]) >= 10
    for c in fx["cases"]:
        lg = c["logits"].clone().requires_grad_(True)
        loss = O.ohem_ce(lg, c["target"], c["weight"], c["num_hard_positive"], c["num_hard_negative"])
        key = (c["num_hard_positive"], c["num_hard_negative"], c["logits"].shape)
        assert abs(float(loss) - float(c["loss"])) <= 1e-5 * abs(float(c["loss"])) + 1e-6, key
        if c["grad"] is not None:
            loss.backward()
            assert (lg.grad - c["grad"]).abs().max() <= 1e-5 * c["grad"].abs().max() + 1e-8, key


def test_ohem_model_matches_reference():
    fx = load_golden("ohem")["model"]
    base = load_golden(fx["base_fixture"])
    sd = _req_grad(base["state_dict"])
    out = O.peneo_forward(sd, fx["config"], base["batch"])
    for k, v in fx["losses"].items():
        assert abs(float(out[k]) - float(v)) < 1e-5 * max(1.0, abs(float(v))), k
    assert abs(float(out["loss"]) - float(base["outputs"]["loss"])) > 1e-3      # OHEM really changed the loss
    out["loss"].backward()
    for n, g in fx["grads"].items():
        assert (sd[n].grad - g).abs().max() <= 2e-4 * g.abs().max() + 1e-7, n
