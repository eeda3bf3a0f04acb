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
    if impl == "flat2":                              # the upper half of the gradients goes on the wire during the backward
        os.environ["PENEO_DP_CHUNKS"] = "2"
        impl = "flat"
    ddp = wrap_data_parallel(net, device_ids=None, bucket_cap_mb=1, impl=impl)
    if os.environ.get("PENEO_DP_CHUNKS") == "2":
        assert ddp._split is not None
    docs = torch.arange(10 * 8, dtype=torch.float32).view(10, 8) / 50.0
    mine = shard_documents(10, rank, world)
    x = docs[list(mine)]
    for _ in range(2):                               # the second step checks that the wrapper re-arms itself
        for p in net.parameters():
            p.grad = None
        loss = ddp(x).pow(2).mean()
        loss.backward()
    if os.environ.get("PENEO_DP_CHUNKS") == "2":
        assert ddp.sync_calls == 2
        if kind == "seq":                            # every upper-half gradient is there before the backward ends
            assert ddp.early_calls == 2
        names = [n for n, _ in net.named_parameters()]
        if kind == "late_first":                     # the late gradient sits in the upper half: this is the case under test
            assert any(n.startswith("embed") for n in names[ddp._split:])
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
