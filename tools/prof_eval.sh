#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/evalprof; rm -rf $OUT; mkdir -p $OUT
python tools/run_eval_fwd.py
rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o run -- python3 tools/run_eval_fwd.py > $OUT/line.txt 2>&1
T=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/prof_summary_csv.py $T 30 > $OUT/summary.txt 2>&1
python - $T <<'PY' > $OUT/gaps.txt
import csv, sys
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1])))
# last forward: from the last embed_text_fwd to the end
idx = [i for i, r in enumerate(rows) if "embed_text_fwd" in r[2]]
a = idx[-1]
seg = rows[a - 3 if a >= 3 else 0:]
span = (seg[-1][1] - seg[0][0]) / 1e3
busy = sum(e - s for s, e, _ in seg) / 1e3
gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3, seg[i][2][:50], seg[i + 1][2][:50]) for i in range(len(seg) - 1))
print(f"last forward: {len(seg)} launches, span {span:.1f} us, kernel time {busy:.1f} us, gaps {span - busy:.1f} us")
print("largest gaps:")
for g in gaps[-12:]: print(f"  {g[0]:7.1f} us  after {g[1]}  before {g[2]}")
PY
rm -rf $OUT/prof; cat $OUT/summary.txt | head -34; cat $OUT/gaps.txt
