#!/usr/bin/env python3
"""docs/sec, forward + backward, LayoutLMv3-base seq 512 / 128 lines, on N MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = forward + loss + backward of `--docs-per-gpu` synthetic RFUND-shaped documents per GPU in
train mode (every dropout site on, incl. the one inside the pair classifiers), bf16 MFMA inputs / fp32 accumulate, and for
N > 1 the RCCL all-reduce of all 127 M gradients (one flat bf16 buffer in stage order, ~64 MB chunks).  No optimizer step (the
metric is fwd+bwd): the bf16 working copies of the fp32 master weights are therefore built once and reused; what
refreshing them after an optimizer step costs is reported beside the metric (`with_weight_recast`).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# The step uses 4 HIP streams per process (main, weight gradients, decoder stage 2, bias-table reduction) plus RCCL's.
# ROCm maps streams onto GPU_MAX_HW_QUEUES (default 4) hardware queues; one process per GPU runs the same with 2 or 8,
# but two processes sharing ONE GPU (the gloo smoke test of the multi-process path) collapse 30x at the default.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402  (importing torch does not touch the GPU; nothing below does before main() has decided how to launch)

# algorithmic GFLOP per document, SURVEY §8(d) / BASELINE.md §3: (forward total, what one pair_heads_fwd launch computes
# = heads L1 + L2) for config 2 (LayoutLMv3-base S512), config 4 (large S1024), config 5 (LiLT-base S512)
ALGO_GFLOP = {("layoutlmv3", "base", 512): (334.7, 192.90 + 1.41),
              ("layoutlmv3", "large", 1024): (2269.2, 1373.05 + 7.51),
              ("lilt", "base", 512): (300.1, 192.90 + 1.41)}
PEAK_BF16_TFLOPS = 2500.0           # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic_bytes(kernel_key: str, docs_per_gpu: int):
    """HBM bytes per launch of the roofline kernel from the committed PMC collection (profiles/pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes of tools/run_pair.py, the same launch this benchmark
    times, with the gfx950 corrections of MI355X_MICROARCH.md).  bench.py cannot run rocprofv3 on itself, so the number
    is the recorded one and only reported when it was taken at this workload; otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get(kernel_key)
        if rec and rec.get("docs_per_launch") == docs_per_gpu:
            return int(rec["traffic_bytes"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def build_model(size: str, dtype, backbone: str = "layoutlmv3", vocab: int = 0):
    """vocab = 250002: the XLM-R-vocabulary registry entries (layoutlmv3-base-chinese / lilt-infoxlm-base,
    model/backbone_mapping.py:277-288,325-336): same encoder, a 768 MB fp32 word table."""
    from seeded import layoutlmv3_config, lilt_config, peneo_config
    from peneo_amd.model import PEneoConfig, PEneoModel
    xlmr = vocab == 250002
    if backbone == "lilt":
        pcfg = peneo_config("lilt-infoxlm-base" if xlmr else "lilt-roberta-en-base", lilt_config(size))
    else:
        pcfg = peneo_config("layoutlmv3-base-chinese" if xlmr else "layoutlmv3-base", layoutlmv3_config(size))
    if vocab:
        pcfg["backbone_config"]["vocab_size"] = vocab
    model = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"}))
    return model, pcfg


def cpu_baseline(pcfg, seq_len, n_lines, seed):
    """The oracle (CPU restatement of the reference, pinned to reference-generated goldens) timed on the host cores for ONE
    document, forward + backward, fp32 — in the reference's AS-EXECUTED form (materialised [N, N, 2D] handshaking input and
    one-hot bias GEMMs, peneo_decoder.py:164-175) as `value`, and in the algebraically reduced form (a_i + b_j) the GPU path
    is built on as `decomposed_value`.  The thread count is capped: oversubscribing the 256-thread host made the
    as-executed form take 181 s instead of 3.5 s."""
    from oracle import peneo_oracle as O
    from peneo_amd.data import synthetic_rfund_batch
    from peneo_amd.model import PEneoConfig, PEneoModel
    threads = max(1, min(os.cpu_count() or 1, 32))
    torch.set_num_threads(threads)
    m = PEneoModel(PEneoConfig(**{k: v for k, v in pcfg.items() if k != "model_type"}))
    sd = {k: (v.detach().clone().requires_grad_(True) if v.is_floating_point() and not k.endswith("_loss.weight") else v)
          for k, v in m.state_dict().items()}
    del m
    batch = synthetic_rfund_batch(1, seq_len, n_lines, pcfg["backbone_config"]["vocab_size"], seed=seed)

    def timed(as_executed, n):
        times = []
        for _ in range(n):
            for v in sd.values():
                if v.is_floating_point() and v.requires_grad:
                    v.grad = None
            t0 = time.perf_counter()
            out = O.peneo_forward(sd, pcfg, batch, training=False, as_executed=as_executed)
            out["loss"].backward()
            times.append(time.perf_counter() - t0)
        return times

    asx = timed(True, 4)                              # one warm-up + three timed passes, ~3.5 s each
    red = timed(False, 3)                             # one warm-up + two timed passes, ~2.6 s each
    dt, dr = sorted(asx[1:])[1], min(red[1:])
    return {"value": round(1.0 / dt, 4), "unit": "docs/s", "cores": threads, "kind": "port",
            "decomposed_value": round(1.0 / dr, 4),
            "sample": f"1 document seq{seq_len}/{n_lines} lines, fwd+bwd fp32 on {threads} of {os.cpu_count()} host threads: the "
                      f"reference's as-executed form (materialised [N,N,2D] handshaking, one-hot bias GEMMs), median of 3 after 1 "
                      f"warm-up ({dt:.2f}s; cold {asx[0]:.1f}s); decomposed_value = the algebraically reduced form (a_i + b_j), "
                      f"best of 2 after 1 warm-up ({dr:.2f}s)"}


def plan_launch(gpus: int, environ) -> tuple:
    """How `bench.py --gpus N` becomes N ranks (reference: `torchrun --nproc_per_node N start/run_rfund.py`, README.md:206-218).
    ("run",)          this process IS a rank (N = 1, or torch.distributed.run / torchrun started it: WORLD_SIZE == N);
    ("spawn", N)      no launcher in the environment and N > 1: the caller starts N fresh rank processes itself;
    ("fail", reason)  a launcher environment whose WORLD_SIZE is not N: never run a 1-GPU job under an N-GPU label."""
    if "WORLD_SIZE" in environ or "RANK" in environ:
        world = int(environ.get("WORLD_SIZE", "1"))
        if world != gpus:
            return ("fail", f"--gpus {gpus} but the launcher set WORLD_SIZE={world}")
        return ("run",)
    if gpus > 1:
        return ("spawn", gpus)
    return ("run",)


def spawn_ranks(n: int, argv) -> int:
    """Start `n` fresh rank processes of this script under torch.distributed.run and relay their output and exit code.  The
    parent never touches the GPU (no HIP call, no torch.cuda.is_available()): the ranks are children, not re-execs."""
    import subprocess
    import uuid
    # the launcher picks AND HOLDS the rendezvous port itself (c10d store on 127.0.0.1:0): a port found by bind / close here could be
    # taken by another process before the ranks connect (concurrent bench runs or tests on one box)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--rdzv-backend=c10d",
           "--rdzv-endpoint=127.0.0.1:0", f"--rdzv-id=peneo-{uuid.uuid4().hex[:12]}", "--local-addr", "127.0.0.1",
           os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--docs-per-gpu", type=int, default=8)
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--lines", type=int, default=128)
    ap.add_argument("--size", default="base", choices=["tiny", "base", "large"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--backbone", default="layoutlmv3", choices=["layoutlmv3", "lilt"],
                    help="lilt = BASELINE config 5 (side measurement; the headline metric is layoutlmv3 base)")
    ap.add_argument("--vocab", type=int, default=0, help="250002 = the XLM-R-vocabulary backbones (side measurement: word-table "
                    "gather / scatter kernels against the HBM roofline)")
    ap.add_argument("--no-ragged", action="store_true", help="skip the ragged side line")
    ap.add_argument("--trained-agree-lr", type=float, default=5e-5, help="backbone learning rate of that training (decoder: x 30)")
    ap.add_argument("--trained-agree-steps", type=int, default=-1,
                    help="optimizer steps on one batch before the indices_agree_trained side metric (0: skip; default: 2000 = ~40 s, "
                         "or 0 with --no-cpu-baseline, i.e. in the quick runs of tools/)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eval-forward", action="store_true", help="time eval forward only (reported as a side metric)")
    args = ap.parse_args()
    if args.trained_agree_steps < 0:
        args.trained_agree_steps = 0 if args.no_cpu_baseline else 2000

    plan = plan_launch(args.gpus, os.environ)
    if plan[0] == "fail":
        print(f"bench.py: {plan[1]}", file=sys.stderr)
        sys.exit(2)
    if plan[0] == "spawn":
        sys.exit(spawn_ranks(plan[1], sys.argv[1:]))

    from peneo_amd import ops
    from peneo_amd.data import synthetic_rfund_batch
    from peneo_amd.parallel import init_distributed, max_over_ranks, wrap_data_parallel
    import torch.distributed as dist

    FWD_GFLOP_PER_DOC, PAIR_HEADS_GFLOP_PER_DOC = ALGO_GFLOP.get((args.backbone, args.size, args.seq_len), (0.0, 0.0))
    rank, local_rank, world = init_distributed()
    ranks = dist.get_world_size() if dist.is_initialized() else 1
    if world != args.gpus or ranks != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}, process group of {ranks} rank(s)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    # The job runs on a HIGH-priority stream (the training loop's choice, like any stream): the step's side streams (weight
    # gradients, dW1, table reductions) stay at normal priority, so the dispatcher hands CU slots to the main chain first and the
    # side work fills what is left instead of sharing round-robin: 17.57 against 17.72 ms per step (tools/run_prio.py).
    if os.environ.get("PENEO_BENCH_PRIORITY", "1") != "0":
        torch.cuda.set_stream(torch.cuda.Stream(device=dev, priority=-1))

    torch.manual_seed(1234)
    model, pcfg = build_model(args.size, dtype, args.backbone, args.vocab)
    model = model.to(dev).set_compute_dtype(dtype).train()
    # the bbox range check of the reference stays ON: as the embedding kernel's sticky device flag, read once after the timed steps
    # (and after every side measurement below) instead of two host syncs per forward
    model.backbone.check_inputs = "deferred"
    net = wrap_data_parallel(model, device_ids=[local_rank]) if world > 1 else model
    B = args.docs_per_gpu
    vocab = pcfg["backbone_config"]["vocab_size"]

    def make_batch(step):
        b = synthetic_rfund_batch(B, args.seq_len, args.lines, vocab, seed=1000 * rank + step)
        if args.backbone == "lilt":
            b.pop("image", None)
        return {k: v.to(dev, non_blocking=True) for k, v in b.items()}

    batches = [make_batch(s) for s in range(4)]     # inputs resident in HBM before the timed region

    def step(i):
        for p in model.parameters():
            p.grad = None
        out = net(**batches[i % len(batches)])
        out["loss"].backward()
        return out["loss"].detach()       # (what a training loop logs; holding the loss itself would keep the step's graph alive through the next step)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ops.TIMER.reset(True)
    torch.cuda.synchronize()
    dbg_alloc = os.environ.get("PENEO_BENCH_ALLOC_STATS") == "1"       # tools/: device allocations and host time per step inside the timed region
    if dbg_alloc:
        ms0 = torch.cuda.memory_stats(dev); host_t = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
        if dbg_alloc:
            host_t.append(time.perf_counter())
    torch.cuda.synchronize()
    if dbg_alloc:
        print("dW1 GEMM ms per launch (side stream: beside the encoder backward):", [round(v, 2) for v in ops.TIMER.durations_ms("dw1_gemm")], file=sys.stderr)
        ms1 = torch.cuda.memory_stats(dev)
        print("alloc stats over the timed region:", {k: ms1[k] - ms0[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams")},
              "reserved GB", round(ms1["reserved_bytes.all.current"] / 1e9, 2), "peak allocated GB", round(ms1["allocated_bytes.all.peak"] / 1e9, 2),
              "host ms per step", [round((b - a) * 1e3, 2) for a, b in zip([t0] + host_t[:-1], host_t)], file=sys.stderr)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dev)
    model.backbone.raise_on_bad_inputs()        # the deferred input check of the warm-up and timed steps (outside the timed region: one sync)
    ph = ops.TIMER.durations_ms("pair_heads_fwd")
    pb = ops.TIMER.durations_ms("pair_bwd_fused")     # fused pair-space backward + its partial-row reduction (one C call)
    pbs = ops.TIMER.durations_ms("pair_bwd_saved")    # ... the form that reads the pre-activations the forward saved (D = 384)
    eb = ops.TIMER.durations_ms("embed_bwd")          # word / position / box table scatter (fp32 atomics)
    ops.TIMER.reset(False)
    loss_val = float(loss.detach())

    # side line: ragged documents (300..510 tokens), batch cut to its longest document rounded up to a multiple
    # of 8 like the reference's collator (data/collator.py:110-116): N varies from batch to batch (masked tails everywhere)
    ragged = None
    if not args.no_ragged and world == 1:
        rb = []
        for s_ in range(4):
            b_ = synthetic_rfund_batch(B, args.seq_len, args.lines, vocab, seed=7000 + s_, ragged=True, pad_to_longest=True)
            if args.backbone == "lilt":
                b_.pop("image", None)
            rb.append({k: v.to(dev) for k, v in b_.items()})

        def rstep(i):
            for p in model.parameters():
                p.grad = None
            out = net(**rb[i % 4])
            out["loss"].backward()
        for i in range(len(rb)):                  # every padded length once: the caching allocator has seen all buffer sizes
            rstep(i)
        torch.cuda.synchronize()
        tr = time.perf_counter()
        nr = max(4, args.steps // 2)
        for i in range(nr):
            rstep(i)
        torch.cuda.synchronize()
        r_ms = (time.perf_counter() - tr) * 1e3 / nr
        ragged = {"docs_per_s": round(B * 1e3 / r_ms, 1), "ms_per_step": round(r_ms, 3),
                  "padded_lengths": [int(b_["input_ids"].shape[1]) for b_ in rb],
                  "mean_tokens": round(float(sum(float(b_["attention_mask"].sum()) for b_ in rb) / (4 * B)), 1)}

    # side metric: the same step when the parameters have changed since the last forward (i.e. behind an optimizer step): every
    # working-precision weight copy is re-cast / re-packed first (engine.WeightCache); not part of `value` (no optimizer step)
    from peneo_amd.model.engine import bump_param_epoch
    nrc = max(4, args.steps // 2)
    for i in range(2):
        bump_param_epoch(); step(i)
    torch.cuda.synchronize()
    trc = time.perf_counter()
    for i in range(nrc):
        bump_param_epoch(); step(i)
    torch.cuda.synchronize()
    recast_ms = (time.perf_counter() - trc) * 1e3 / nrc

    # side metric: eval forward only (the "encoder + pair-head forward" roofline target of BASELINE.md §5)
    model.eval()
    with torch.no_grad():
        for i in range(2):
            net_out = model(**batches[i])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        nf = max(3, args.steps // 2)
        for i in range(nf):
            model(**batches[i % len(batches)])
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t1) * 1e3 / nf

    # side metric (north_star: "BIO/link indices bit-exact"): how many of the 5 x B x 130 816 pair-tag indices (argmax over the
    # classes, model/peneo_decoder.py:98-114) the bf16 path BENCHMARKED here decides differently from the fp32 parity path of
    # the same weights on the same batch; `clear` counts only pairs whose fp32 margin (top1 - top2 logit) exceeds 0.05
    indices_agree = None
    if world == 1:
        with torch.no_grad():
            o16 = model(**batches[0])
            model.set_compute_dtype(torch.float32)
            o32 = model(**batches[0])
            model.set_compute_dtype(dtype)
            maps, flips, clear, total = {}, 0, 0, 0
            for k in o32:
                if not k.endswith("_shaking_outputs"):
                    continue
                a16, a32 = o16[k].argmax(-1), o32[k].argmax(-1)
                diff = a16 != a32
                top2 = o32[k].float().topk(2, dim=-1).values
                dc = diff & ((top2[..., 0] - top2[..., 1]) > 0.05)
                maps[k[:-len("_shaking_outputs")]] = {"flips": int(diff.sum()), "clear_flips": int(dc.sum()), "of": diff.numel()}
                flips += int(diff.sum()); clear += int(dc.sum()); total += diff.numel()
            indices_agree = {"bf16_vs_fp32_flips": flips, "clear_flips": clear, "of": total,
                             "flip_rate": round(flips / max(1, total), 9), "maps": maps}
            del o16, o32

    # side metric: the optimizer step that completes a training iteration (fused multi-tensor AdamW, reference groups);
    # not part of `value` (the metric is fwd + bwd)
    from peneo_amd.optim import FusedAdamW, peneo_param_groups
    model.train()
    step(0)
    opt = FusedAdamW(peneo_param_groups(model, 5e-5, 0.01, 30.0), max_grad_norm=1.0)   # HF Trainer's default clipping, fused
    opt.step()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(3):
        opt.step()
    torch.cuda.synchronize()
    opt_ms = (time.perf_counter() - t2) * 1e3 / 3

    # side metric: `indices_agree` above is taken on random-init weights, where every logit sits in a near-tie
    # band.  What the reference decodes in practice (model/peneo_decoder.py:98-114) are separated logits: train this model on ONE
    # synthetic batch with the fused AdamW (reference groups: decoder lr x 30) until the positive spots separate, then count the
    # pair-tag indices the bf16 path decides differently from the fp32 path of the same trained weights, and compare the decoded
    # spot lists (get_spots_from_shaking_tag) of the two precisions.  Outside the timed region; not part of `value`.
    indices_agree_trained = None
    if world == 1 and args.trained_agree_steps > 0 and args.backbone == "layoutlmv3":
        from peneo_amd.model.peneo_decoder import HandshakingTaggingScheme
        # (eval mode: no dropout - the run is the same every time up to the order of fp32 atomics, and the positives separate in
        # fewer steps; the gradient path is the one the parity tests use)
        model.eval()
        weights_before = {k_: v_.detach().clone() for k_, v_ in model.state_dict().items()}    # restored below: later readers see the benchmarked weights
        tb = batches[0]
        base_lrs = [g_["lr"] * (args.trained_agree_lr / 5e-5) for g_ in opt.param_groups]   # (built at 5e-5 / 1.5e-3 for the timing above)
        nst = args.trained_agree_steps
        for it in range(nst):
            # the reference recipe's schedule shape (README.md:218-241: warm-up, then linear decay): without the decay the last
            # steps still move the decision boundary of a few positives and the count below depends on where the run happens to stop
            f_ = min(1.0, (it + 1) / max(1, nst // 20)) * max(0.0, 1.0 - it / nst)
            for g_, lr0 in zip(opt.param_groups, base_lrs):
                g_["lr"] = lr0 * f_
            for p_ in model.parameters():
                p_.grad = None
            out_t = net(**tb)
            out_t["loss"].backward()
            opt.step()
        loss_tr = float(out_t["loss"].detach())
        model.eval()
        Nn = tb["input_ids"].shape[1] - 1
        with torch.no_grad():
            o16 = model(**tb)
            model.set_compute_dtype(torch.float32)
            o32 = model(**tb)
            model.set_compute_dtype(dtype)
            flips = total = pos32 = pos_flips = 0
            edges = (2e-3, 1e-2, 5e-2, 2e-1)
            above = [0] * len(edges)
            n_spots, spots_differ, max_flip_margin, max_dlogit = 0, 0, 0.0, 0.0
            for k in o32:
                if not k.endswith("_shaking_outputs"):
                    continue
                a16, a32 = o16[k].argmax(-1), o32[k].argmax(-1)
                diff = a16 != a32
                top2 = o32[k].float().topk(2, dim=-1).values
                margin = top2[..., 0] - top2[..., 1]
                flips += int(diff.sum()); total += diff.numel()
                for i_, e_ in enumerate(edges):
                    above[i_] += int((diff & (margin > e_)).sum())
                if int(diff.sum()):
                    max_flip_margin = max(max_flip_margin, float(margin[diff].max()))
                max_dlogit = max(max_dlogit, float((o16[k].float() - o32[k].float()).abs().max()))
                pos = a32 != 0
                pos32 += int(pos.sum()); pos_flips += int((diff & (pos | (a16 != 0))).sum())
                for b_ in range(B):       # decoded (i, j, tag) lists, the reference's definition of "indices"
                    s16 = HandshakingTaggingScheme.get_spots_from_shaking_tag(o16[k][b_], seq_len=Nn)
                    s32 = HandshakingTaggingScheme.get_spots_from_shaking_tag(o32[k][b_], seq_len=Nn)
                    n_spots += len(s32)
                    spots_differ += len(set(tuple(x[:3]) for x in s16) ^ set(tuple(x[:3]) for x in s32))
            indices_agree_trained = {"train_steps": args.trained_agree_steps, "lr": args.trained_agree_lr, "loss_after": round(loss_tr, 6),
                                     "bf16_vs_fp32_flips": flips, "of": total, "flip_rate": round(flips / max(1, total), 9),
                                     "flips_with_fp32_margin_above": {str(e_): n_ for e_, n_ in zip(edges, above)},
                                     "largest_fp32_margin_of_a_flip": round(max_flip_margin, 5),
                                     "max_abs_logit_difference_bf16_vs_fp32": round(max_dlogit, 5),
                                     "fp32_positive_tags": pos32, "flips_touching_a_positive": pos_flips,
                                     "decoded_spots_fp32": n_spots, "decoded_spots_differing": spots_differ}
            del o16, o32
        model.load_state_dict(weights_before)
        del weights_before
        model.backbone.raise_on_bad_inputs()
        model.train()

    if rank == 0:
        docs = world * B * args.steps
        ms_per_step = elapsed * 1e3 / args.steps
        ph_ms = sum(ph) / max(1, len(ph))
        achieved = PAIR_HEADS_GFLOP_PER_DOC * B / ph_ms if ph_ms > 0 else 0.0      # GFLOP / ms = TFLOP/s
        # the dominant kernel of the step since round 2: the fused pair-space backward (z recomputed + du = dz W1: two
        # first-layer contractions of all pairs, 2 x 5 * 2 * P * D^2 FLOP per document; dW1 = dz^T x stays a GEMM)
        pb_ms = sum(pb) / max(1, len(pb))
        l1 = PAIR_HEADS_GFLOP_PER_DOC - 2.0 * (args.seq_len - 1) * args.seq_len / 2 * (pcfg["backbone_config"]["hidden_size"] // 2) * 14 / 1e9
        pb_gflop = 2.0 * l1
        pb_achieved = pb_gflop * B / pb_ms if pb_ms > 0 else 0.0
        ks_f = pcfg['backbone_config']['hidden_size'] // 32      # D / 16; bf16 at 24 / 32: the hand-interleaved kernel (unless switched off)
        hand = args.dtype == "bf16" and ks_f in (24, 32) and os.environ.get("PENEO_PAIR_FWD_HAND", "1") != "0"
        fwd_roof = {"bound": "mfma", "kernel": (f"pair_heads_fwd_hand_kernel<{ks_f}>" + (" (saving form: + 2 B per pair and hidden unit written)" if pbs else "")) if hand else f"pair_heads_fwd_kernel<{'bf16' if args.dtype == 'bf16' else 'f32'},{ks_f}>",
                    "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4),
                    "traffic": pmc_traffic_bytes(("pair_heads_fwd_save_" if pbs else "pair_heads_fwd_train_") + args.dtype, B) if args.size == "base" and args.seq_len == 512 else None,
                    "traffic_source": "profiles/pmc_traffic.json (recorded rocprofv3 --pmc passes, not measured by this run)",
                    "avg_launch_ms": round(ph_ms, 4), "launches": len(ph)}
        if pbs:
            # The saved-activation form of the pair-space backward.  SURVEY 8(d) bounds the pair kernels by the MFMA roof: the launch
            # contracts du = dz W1 for every pair (the first-layer FLOPs of 8(d)), so `roofline` is those FLOPs over the launch time
            # against the dense bf16 peak.  What the DESIGN pays for having one contraction instead of two -- the forward leaves z
            # (f16) per pair and hidden unit, this kernel reads it back and writes dz -- is reported beside it as `design_traffic`,
            # with the bytes the operation would move if nothing were saved (ab, dlogits, weights in; d_ab out).
            pbs_ms = sum(pbs) / len(pbs)
            Pn = (args.seq_len - 1) * args.seq_len // 2
            Nn = args.seq_len - 1
            Dd = pcfg["backbone_config"]["hidden_size"] // 2
            design_bytes = B * (2 * Pn * 5 * Dd * 2 + Pn * 14 * 4)
            io_bytes = B * (Pn * 14 * 4 + Nn * 2 * Dd * 2 + Nn * 2 * Dd * 4) + 5 * Dd * Dd * 2 + 14 * Dd * 4
            mfma_tf = l1 * B / pbs_ms
            dom_roof = {"bound": "mfma", "kernel": f"pair_bwd_sv_kernel<{Dd // 16}> (+ pair_bwd_reduce_kernel)",
                        "achieved": round(mfma_tf, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(mfma_tf / PEAK_BF16_TFLOPS, 4),
                        "traffic": pmc_traffic_bytes("pair_bwd_saved_" + args.dtype, B) if args.size == "base" and args.seq_len == 512 else None,
                        "traffic_source": "profiles/pmc_traffic.json (recorded rocprofv3 --pmc passes, not measured by this run)",
                        "design_traffic": {"bytes": design_bytes, "GBps": round(design_bytes / pbs_ms / 1e6, 1),
                                           "frac_of_hbm": round(design_bytes / pbs_ms / 1e6 / 8000.0, 4),
                                           "algorithmic_io_bytes": io_bytes, "ratio": round(design_bytes / io_bytes, 1)},
                        "avg_launch_ms": round(pbs_ms, 4), "launches": len(pbs)}
        elif pb_ms > 0:
            ks = pcfg['backbone_config']['hidden_size'] // 32       # D / 16: 24 -> the wave-specialised kernel, 32 -> the one-wave kernel
            dom_roof = {"bound": "mfma", "kernel": f"pair_bwd_{'one' if ks == 32 else 'ws'}_kernel<{ks}> (+ pair_bwd_reduce_kernel)",
                        "achieved": round(pb_achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(pb_achieved / PEAK_BF16_TFLOPS, 4),
                        "traffic": pmc_traffic_bytes("pair_bwd_fused_" + args.dtype, B) if args.size == "base" and args.seq_len == 512 else None,
                        "traffic_source": "profiles/pmc_traffic.json (recorded rocprofv3 --pmc passes, not measured by this run)",
                        "avg_launch_ms": round(pb_ms, 4), "launches": len(pb)}
        else:
            dom_roof = fwd_roof
        res = {
            "metric": "docs/sec fwd+bwd, LayoutLMv3-base seq512 L128" if (args.backbone, args.size, args.seq_len, args.lines) ==
                      ("layoutlmv3", "base", 512, 128) else f"docs/sec fwd+bwd, {args.backbone}-{args.size} seq{args.seq_len} L{args.lines}",
            "value": round(docs / elapsed, 2),
            "unit": "docs/s",
            "n_gpus": world,
            "rccl_ranks": ranks,                     # dist.get_world_size() of the process group the gradients were reduced over
            "dist_backend": dist.get_backend() if dist.is_initialized() else None,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{'LayoutLMv3' if args.backbone == 'layoutlmv3' else 'LiLT'}-{args.size} PEneo, synthetic RFUND-shaped batch seq{args.seq_len}/"
                                   f"{args.lines} lines, {B} docs/GPU, train mode (dropout 0.1), fwd+loss+bwd"
                                   f"{' + RCCL grad all-reduce (one flat bf16 buffer per step)' if world > 1 else ''}",
                       "docs_per_gpu": B, "seq_len": args.seq_len, "lines": args.lines,
                       "parallelism": f"dp{world}", "final_loss": round(loss_val, 5)},
            "input_checks": True,               # bbox range check on in the timed region (device flag, read after it)
            "roofline": dom_roof,
            "roofline_pair_heads_fwd": fwd_roof,
            "forward_only": {"ms_per_batch": round(fwd_ms, 3), "docs_per_s": round(B * 1e3 / fwd_ms, 1),
                             "tflops_algorithmic": round(FWD_GFLOP_PER_DOC * B / fwd_ms, 1),
                             "frac_of_mfma_peak": round(FWD_GFLOP_PER_DOC * B / fwd_ms / PEAK_BF16_TFLOPS, 4)},
            "indices_agree": indices_agree,
            "indices_agree_trained": indices_agree_trained,
            "ragged": ragged,
            "embed_bwd": ({"avg_launch_ms": round(sum(eb) / len(eb), 4), "launches": len(eb), "vocab": vocab,
                           # algorithmic bytes: every token reads its d_x row (H * 2 B) and adds H fp32 values into 6 tables
                           "GBps_algorithmic": round(B * args.seq_len * pcfg["backbone_config"]["hidden_size"] * (2 + 2 * 4 * 2)
                                                     / (sum(eb) / len(eb) * 1e-3) / 1e9, 1), "peak_GBps": 8000.0} if eb else None),
            "with_weight_recast": {"ms_per_step": round(recast_ms, 3), "docs_per_s": round(world * B * 1e3 / recast_ms, 1)},
            "optimizer_step_ms": round(opt_ms, 3),
            "train_tflops_algorithmic": round(3 * FWD_GFLOP_PER_DOC * docs / elapsed / 1e3, 1),
        }
        if not args.no_cpu_baseline and world == 1 and args.size == "base" and args.backbone == "layoutlmv3":
            res["cpu_baseline"] = cpu_baseline(pcfg, args.seq_len, args.lines, seed=7)
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
