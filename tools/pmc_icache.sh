#!/bin/bash
# Instruction-cache behaviour per kernel over a short train run (rocprofv3 --pmc, one pass per counter group; kernel-trace only).
# usage (through gpurun): bash tools/pmc_icache.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/icache; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_INSTS_ALL\|SQ_WAIT_INST_ANY" | sort -u > $OUT/avail.txt
cat $OUT/avail.txt | tr '\n' ' '; echo
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  rm -rf $OUT/tmp
  rocprofv3 --pmc $C -d $OUT/tmp -o p -f csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-ragged --trained-agree-steps 0 "$@" > /dev/null 2>&1
  python3 - <<'PY'
import csv, glob, collections
fs = glob.glob("gpurun_out/icache/tmp/**/*counter_collection.csv", recursive=True)
if not fs:
    print("no counter file"); raise SystemExit
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    n[(k, r["Counter_Name"])] += 1
names = sorted({c for v in acc.values() for c in v})
print("per launch:", " ".join(f"{c:>18s}" for c in names), " launches  kernel")
rows = []
for k, v in acc.items():
    cnt = max(n[(k, c)] for c in names if (k, c) in n)
    rows.append((sum(v.values()), k, [v.get(c, 0.0) / max(1, n[(k, c)]) for c in names], cnt))
for _, k, vals, cnt in sorted(rows, reverse=True)[:28]:
    print("           ", " ".join(f"{x:18.0f}" for x in vals), f"{cnt:8d}  {k}")
PY
done
