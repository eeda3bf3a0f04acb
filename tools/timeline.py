#!/usr/bin/env python3
"""Timeline view of one training step from a rocprofv3 kernel trace (csv).

    python tools/timeline.py <..._kernel_trace.csv> [step_index_from_end=2 | +K = K-th train step from the start] [--list] [--gaps]

Splits the trace into steps at every `pair_heads_fwd_kernel` / `pair_heads_fwd_hand_kernel` launch of a train step (one per step), takes one steady-state
step and prints: wall span, union busy time, idle time, per-queue busy time, and per kernel name the total duration, the
time during which it was the ONLY kernel running (exposed time) and the launch count.  `--list` dumps the step's launches
in start order (offset, duration, queue, name) to follow the critical path by eye."""
import csv
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("peneo::", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:70]


def main():
    path = sys.argv[1]
    arg = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "2"
    back = int(arg) if not arg.startswith("+") else None      # "+K": the K-th train step from the START of the trace (0-based)
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "pair_heads_fwd_kernel" in r[3] or "pair_heads_fwd_hand_kernel" in r[3]]
    # train steps are the pair_heads launches followed by a pair_dz kernel before the next mark
    train = []
    for a, b in zip(marks, marks[1:] + [len(rows)]):
        if any(("pair_dz" in rows[k][3] or "pair_bwd" in rows[k][3]) for k in range(a, b)):
            train.append(a)
    if back is None:
        k = int(arg[1:])
        if len(train) < k + 2:
            print("not enough train steps in the trace", len(train))
            return
        a, b = train[k], train[k + 1]
    else:
        if len(train) < back + 1:
            print("not enough train steps in the trace", len(train))
            return
        a, b = train[-back - 1], train[-back]
    # a step starts at the first kernel after the previous step's last backward kernel: walk back from the pair_heads launch
    # to the embedding kernel of the same forward
    def step_start(i):
        j = i
        while j > 0 and "embed_text_fwd" not in rows[j][3]:
            j -= 1
        while j > 0 and rows[j][0] - rows[j - 1][1] < 200_000 and not any(s in rows[j - 1][3] for s in ("embed_text_bwd", "relpos_bias_bwd", "adamw")):
            j -= 1
        return j
    s0, s1 = step_start(a), step_start(b)
    step = rows[s0:s1]
    t0, t1 = step[0][0], max(r[1] for r in step)
    span = (t1 - t0) / 1e3
    ev = []
    for s, e, q, n in step:
        ev.append((s, 1, n)); ev.append((e, -1, n))
    ev.sort()
    busy = 0.0
    active = defaultdict(int)
    exposed = defaultdict(float)
    last = t0
    nact = 0
    gaps = []
    prev_name = ""
    for t, d, n in ev:
        if nact == 0 and t > last:
            gaps.append((t - last, last - t0, prev_name, n))
        prev_name = n
        if nact > 0:
            busy += t - last
            if nact == 1:
                only = next(k for k, v in active.items() if v > 0)
                exposed[only] += t - last
        active[n] += d
        nact += d
        last = t
    tot = defaultdict(float); cnt = defaultdict(int); qbusy = defaultdict(float)
    for s, e, q, n in step:
        tot[short(n)] += e - s; cnt[short(n)] += 1; qbusy[q] += e - s
    exp2 = defaultdict(float)
    for n, v in exposed.items():
        exp2[short(n)] += v
    print(f"step: {len(step)} launches, span {span:.1f} us, busy (union) {busy / 1e3:.1f} us, idle {span - busy / 1e3:.1f} us")
    print("queue busy (us):", {q: round(v / 1e3, 1) for q, v in sorted(qbusy.items())})
    print(f"{'total_us':>10} {'alone_us':>10} {'calls':>6}  name")
    for n in sorted(tot, key=lambda k: -tot[k])[:45]:
        print(f"{tot[n] / 1e3:10.1f} {exp2[n] / 1e3:10.1f} {cnt[n]:6d}  {n}")
    if "--gaps" in sys.argv:
        print("largest idle gaps (us, at offset us, after kernel -> before kernel):")
        for g, off, p, n in sorted(gaps, reverse=True)[:25]:
            print(f"{g / 1e3:8.1f} @{off / 1e3:9.1f}  {short(p)[:45]} -> {short(n)[:45]}")
        print(f"gaps: {len(gaps)}, total {sum(g for g, *_ in gaps) / 1e3:.1f} us")
    if "--list" in sys.argv:
        for s, e, q, n in step:
            print(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} q{q} {short(n)}")


if __name__ == "__main__":
    main()
