"""world_size-2 checks of the data-parallel plumbing on the gloo backend (no GPU needed)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _LateFirst(torch.nn.Module):
    """Parameter order != gradient order, like the real model: `embed` is registered LAST (it lands in the upper half of the
    flat buffer) but used FIRST, so its gradient is the last one the backward produces."""

    def __init__(self):
        super().__init__()
        self.body = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.SiLU(), torch.nn.Linear(16, 3))
        self.embed = torch.nn.Linear(8, 16)

    def forward(self, x):
        return self.body(torch.tanh(self.embed(x)))


def _make_net(kind):
    if kind == "late_first":
        return _LateFirst()
    return torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.SiLU(), torch.nn.Linear(16, 3))


def _worker(rank, world, port, q, impl, kind="seq"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from peneo_amd.parallel import (all_gather_counts, init_distributed, max_over_ranks, shard_documents,
                                    wrap_data_parallel)
    r, lr, w = init_distributed("gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    net = _make_net(kind)
    chunked = impl == "flat2"                        # chunks of a few hundred bytes: several collectives during the backward
    if chunked:
        os.environ["PENEO_DP_CHUNK_MB"] = str(200 / (1 << 20))
        impl = "flat"
    elif impl == "flat":
        os.environ["PENEO_DP_CHUNKS"] = "1"          # one all-reduce after the backward
    ddp = wrap_data_parallel(net, device_ids=None, bucket_cap_mb=1, impl=impl)
    docs = torch.arange(10 * 8, dtype=torch.float32).view(10, 8) / 50.0
    mine = shard_documents(10, rank, world)
    x = docs[list(mine)]
    for _ in range(3):                               # step 1 learns the arrival order, steps 2 and 3 run chunked
        for p in net.parameters():
            p.grad = None
        loss = ddp(x).pow(2).mean()
        loss.backward()
    if chunked:
        assert ddp.sync_calls == 3
        assert len(ddp.chunks) >= 2, ddp.chunks
        assert ddp.early_calls == 2 * (len(ddp.chunks) - 1)      # every chunk but the last, in steps 2 and 3
        names = [n for n, _ in net.named_parameters()]
        laid_out = [names[i] for i in ddp.order]
        if kind == "late_first":                     # registered last, produced last: the layout follows the backward
            assert laid_out[-1].startswith("embed") and laid_out[0].startswith("body.2"), laid_out
    for n, p in net.named_parameters():
        assert p.grad is not None and float(p.grad.abs().max()) > 0, n
    g = torch.cat([p.grad.flatten() for p in net.parameters()])
    # reference: average of the two ranks' local gradients computed without DDP
    ref_net = _make_net(kind)
    ref_net.load_state_dict(net.state_dict())
    acc = None
    for rr in range(world):
        ref_net.zero_grad()
        ref_net(docs[list(shard_documents(10, rr, world))]).pow(2).mean().backward()
        gg = torch.cat([p.grad.flatten() for p in ref_net.parameters()])
        acc = gg if acc is None else acc + gg
    ok = torch.allclose(g, acc / world, atol=1e-6)
    counts = all_gather_counts([rank, 10 + rank])
    mx = max_over_ranks(1.0 + rank)
    q.put((rank, ok, counts, mx, list(mine)))
    dist.destroy_process_group()


@pytest.mark.parametrize("impl,kind", [("flat", "seq"), ("flat2", "seq"), ("flat2", "late_first"), ("flat", "late_first"),
                                       ("ddp", "seq")])
def test_ddp_gradient_average_and_gathers(impl, kind):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, impl, kind)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    assert all(r[1] for r in res), "DDP-averaged gradients differ from the mean of the per-rank gradients"
    assert res[0][2] == [[0, 10], [1, 11]] and res[1][2] == [[0, 10], [1, 11]]
    assert res[0][3] == 2.0 and res[1][3] == 2.0
    assert res[0][4] == [0, 1, 2, 3, 4] and res[1][4] == [5, 6, 7, 8, 9]


def test_shard_documents_is_a_partition():
    from peneo_amd.parallel import shard_documents
    for n, w in [(64, 8), (10, 3), (5, 8), (1, 1)]:
        seen = []
        for r in range(w):
            seen += list(shard_documents(n, r, w))
        assert seen == list(range(n))


def _wire_worker(rank, world, port, q, wire):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), PENEO_DP_CHUNKS="1")
    from peneo_amd.parallel import FlatGradDataParallel, init_distributed
    init_distributed("gloo")
    net = torch.nn.Linear(64, 64, bias=False)
    ddp = FlatGradDataParallel(net, wire_dtype=wire)
    # every rank contributes its own gradient of realistic spread (a few orders of magnitude, both signs)
    g = torch.Generator().manual_seed(100 + rank)
    mine = torch.randn(64, 64, generator=g) * torch.logspace(-4, 0, 64).unsqueeze(1)
    x = torch.eye(64)
    (ddp(x) * mine.t()).sum().backward()             # d/dW = mine
    allg = [torch.randn(64, 64, generator=torch.Generator().manual_seed(100 + r)) * torch.logspace(-4, 0, 64).unsqueeze(1)
            for r in range(world)]
    exact = torch.stack(allg).double().mean(0)
    scale = torch.stack(allg).abs().double().mean(0)           # the magnitude the rounding errors scale with
    err = float(((net.weight.grad.double() - exact).abs() / scale).max())
    q.put((rank, err))
    dist.destroy_process_group()


@pytest.mark.parametrize("wire,bound", [(torch.bfloat16, 8 * 2.0 ** -8), (torch.float32, 1e-6)])
def test_wire_precision_of_an_eight_rank_sum(wire, bound):
    """The reference's DDP sums fp32 gradients; the bf16 wire format rounds every rank's contribution to 8 bits and the
    collective accumulates in bf16: each of the 7 additions and the 8 input roundings adds at most 2^-9 of the running
    magnitude, so |error| <= 8 * 2^-8 of the mean |gradient| (measured ~1e-2); PENEO_DP_WIRE=fp32 is exact to fp32 rounding."""
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_wire_worker, args=(r, world, port, q, wire)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    errs = [e for _, e in res]
    assert max(errs) <= bound, errs
    if wire == torch.bfloat16:
        assert max(errs) > 1e-4                      # the bf16 wire really was in use
